// Micro-probe: does independent VALU work co-issue with v_mfma_f32_32x32x2_f32 on a SIMD?  Each wave loops over
// 16 MFMAs (4 accumulator chains, constant operands) followed by NV independent v_fma_f32 (8 chains).  If the MFMA rate
// holds until NV * 4 cycles approaches 16 * 64 cycles, the two pipes overlap; if it falls from NV = 0 on, they share issue.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_mix_probe tools/probes/valu_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float *out, const float *in, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = in[threadIdx.x + 32 * j];
    const float c = in[threadIdx.x + 7], d = in[threadIdx.x + 9];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NV; n++) v[n & 7] = fmaf(v[n & 7], c, d);
    }
    float s = 0;
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) s += acc[j][r];
    for (int j = 0; j < 8; j++) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// interleaved variant: KV independent VALU ops pinned right behind EACH MFMA (same wave, MFMA shadow)
template <int KV, int WPS>
__global__ __launch_bounds__(256, WPS) void ki(float *out, const float *in, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = in[threadIdx.x + 32 * j];
    const float c = in[threadIdx.x + 7], d = in[threadIdx.x + 9];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < KV; n++) v[n & 7] = fmaf(v[n & 7], c, d);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) s += acc[j][r];
    for (int j = 0; j < 8; j++) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KV, int WPS>
void runi(float *out, float *in) {
    const int iters = 4000, blocks = 256 * WPS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((ki<KV, WPS>), dim3(blocks), dim3(256), 0, 0, out, in, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((ki<KV, WPS>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 16;
    const double tf = mf * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("interleaved: waves/SIMD %d, %2d VALU behind each MFMA: %.3f ms  %.1f TFLOP/s\n", WPS, KV, ms, tf);
}

// OP: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_add_f32, 3 v_add_f32, 4 v_add_u32, 5 v_mov_b32, 6 v_pk_mul_f32
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP, int NV>
__global__ __launch_bounds__(256, 3) void kop(float *out, const float *in, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    v2f v[8];
    for (int j = 0; j < 8; j++) v[j] = v2f{in[threadIdx.x + 32 * j], in[threadIdx.x + 32 * j + 1]};
    v2f c = {in[threadIdx.x + 7], in[threadIdx.x + 8]}, d = {in[threadIdx.x + 9], in[threadIdx.x + 10]};
    __shared__ float lds[4096];
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned ldsaddr = (unsigned)(threadIdx.x & 63) * 16;
    unsigned sreg = 0;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f wide[2] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NV; n++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7].x) : "v"(c.x), "v"(d.x));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(c), "v"(d));
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[n & 7]) : "v"(c));
            if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[n & 7].x) : "v"(c.x));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[n & 7].x) : "v"(c.x));
            if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(v[n & 7].x) : "v"(c.x));
            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[n & 7]) : "v"(c));
            if (OP == 7) asm volatile("ds_read_b32 %0, %1" : "=v"(v[n & 7].y) : "v"(ldsaddr) : "memory");
            if (OP == 8) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
            if (OP == 9) asm volatile("s_nop 0");
            if (OP == 10) asm volatile("ds_read_b128 %0, %1" : "=v"(wide[n & 1]) : "v"(ldsaddr) : "memory");
        }
    }
    float s = 0;
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 16; r++) s += acc[j][r];
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int j = 0; j < 8; j++) s += v[j].x + v[j].y;
    s += wide[0].x + wide[1].y + (float)sreg;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int NV>
void runop(const char *name, float *out, float *in, float base_ms) {
    const int iters = 4000, blocks = 768;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kop<OP, NV>), dim3(blocks), dim3(256), 0, 0, out, in, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kop<OP, NV>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles stolen per instruction, in units of MFMA-pipe cycles (16 MFMAs = 1024 cycles = base_ms per iteration set)
    printf("%-14s %3d per 16 MFMA: %.3f ms  -> %.2f MFMA-cycles per instruction\n", name, NV, ms,
           (ms / base_ms - 1.0) * 1024.0 / NV);
}

template <int NV, int WPS>
void run(float *out, float *in) {
    const int iters = 4000, blocks = 256 * WPS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, WPS>), dim3(blocks), dim3(256), 0, 0, out, in, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, WPS>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 16;
    const double tf = mf * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d, %3d VALU per 16 MFMA: %.3f ms  %.1f TFLOP/s\n", WPS, NV, ms, tf);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22);
    hipMemset(in, 0, 1 << 20);
    run<0, 1>(out, in); run<32, 1>(out, in); run<64, 1>(out, in); run<128, 1>(out, in); run<256, 1>(out, in);
    run<0, 3>(out, in); run<32, 3>(out, in); run<64, 3>(out, in); run<128, 3>(out, in); run<192, 3>(out, in); run<256, 3>(out, in);
    {
        const float base = 5.16f;  // 3 waves/SIMD, MFMA only (measured above)
        runop<0, 64>("v_fma_f32", out, in, base); runop<1, 64>("v_pk_fma_f32", out, in, base); runop<2, 64>("v_pk_add_f32", out, in, base);
        runop<3, 64>("v_add_f32", out, in, base); runop<4, 64>("v_add_u32", out, in, base); runop<5, 64>("v_mov_b32", out, in, base);
        runop<6, 64>("v_pk_mul_f32", out, in, base);
        runop<7, 64>("ds_read_b32", out, in, base); runop<10, 64>("ds_read_b128", out, in, base);
        runop<9, 64>("s_nop 0", out, in, base);
    }
    runi<0, 1>(out, in); runi<2, 1>(out, in); runi<4, 1>(out, in); runi<8, 1>(out, in); runi<12, 1>(out, in); runi<16, 1>(out, in);
    runi<0, 3>(out, in); runi<2, 3>(out, in); runi<4, 3>(out, in); runi<8, 3>(out, in); runi<12, 3>(out, in); runi<16, 3>(out, in);
    return 0;
}
