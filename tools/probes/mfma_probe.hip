// Micro-probe: issue rate of v_mfma_f32_32x32x2_f32 from ONE wave per SIMD under different operand sources.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_probe tools/probes/mfma_probe.hip ; run on the GPU box
// (results of round 1: const operands 150-155 TFLOP/s; operand rewritten right behind its MFMA 111; double-buffered
// operand registers 137; LDS value consumed by an FMA right behind its ds_read 77)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void k(float *out, const float *in, int iters) {
    __shared__ float lds[8192];
    for (int j = threadIdx.x; j < 8192; j += 256) lds[j] = in[j & 1023];
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    float a[NACC], b = in[threadIdx.x], c = in[threadIdx.x + 256];
    for (int j = 0; j < NACC; j++) a[j] = in[threadIdx.x + 32 * j];
    const int base = (threadIdx.x & 31);
    if (MODE == 4 || MODE == 5) {
        // double-buffered operands: the A value of iteration it+1 is produced (VALU, optionally from LDS) right AFTER
        // the MFMA j of iteration it was issued, into a register no in-flight MFMA reads
        float cur[NACC], nxt[NACC];
        for (int j = 0; j < NACC; j++) cur[j] = a[j];
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[j], b, acc[j], 0, 0, 0);
                const float x = MODE == 5 ? lds[base + ((it * NACC + j) & 127) * 32] : a[j];
                nxt[j] = fmaf(c, x, b);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(nxt[j], b, acc[j], 0, 0, 0);
                const float x = MODE == 5 ? lds[base + ((it * NACC + j + 64) & 127) * 32] : a[j];
                cur[j] = fmaf(c, x, b);
                __builtin_amdgcn_sched_barrier(0);
            }
            c += 1e-9f;
        }
    } else
    for (int it = 0; it < iters; it++) {
        float x[NACC];
        if (MODE >= 2) {
#pragma unroll
            for (int j = 0; j < NACC; j++) x[j] = lds[base + ((it * NACC + j) & 127) * 32];
        }
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            float av = a[j];
            if (MODE == 1) av = fmaf(c, a[j], b);            // VALU-produced operand, register inputs
            if (MODE >= 2) av = fmaf(c, x[j], a[j]);          // LDS-fed + VALU
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[j], 0, 0, 0);
            if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) { c += 1e-9f; }
    }
    float s = 0;
    for (int j = 0; j < NACC; j++)
        for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char *name, float *out, float *in, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NACC;  // MFMAs per wave
    const double tf = mf * 4096.0 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("%-34s blocks %d: %.3f ms  %.1f ns/MFMA  %.1f TFLOP/s\n", name, blocks, ms, ms * 1e6 / mf, tf);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22);
    float *h = (float *)malloc(1 << 20);
    for (int i = 0; i < (1 << 18); i++) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
    hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<0, 9>("const operands, 9 acc", out, in, blocks);
        run<1, 9>("VALU-produced A, 9 acc", out, in, blocks);
        run<2, 9>("LDS+VALU A, 9 acc", out, in, blocks);
        run<3, 9>("LDS+VALU A, 9 acc, pinned order", out, in, blocks);
        run<4, 9>("VALU A, double-buffered regs", out, in, blocks);
        run<5, 9>("LDS+VALU A, double-buffered regs", out, in, blocks);
        run<0, 4>("const operands, 4 acc", out, in, blocks);
        run<0, 1>("const operands, 1 acc", out, in, blocks);
    }
    return 0;
}
