// Probe: cycles per v_mfma_f32_32x32x16_bf16 for the four placements of (accumulator, weight operand) in the unified
// register file, one wave per SIMD, 6 independent accumulators, 18 MFMAs per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_operand_probe tools/probes/mfma_operand_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define MF_VAV(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(W), "v"(X))
#define MF_AAV(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "a"(W), "v"(X))
#define MF_AVV(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X))
#define MF_VVV(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(W), "v"(X))
#define MF_VVA(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(X), "a"(W))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const int *in, float *out, long long *cyc, int iters) {
    i32x4 w[18], x[3];
    f32x16 acc[6];
    for (int i = 0; i < 18; i++) w[i] = *reinterpret_cast<const i32x4 *>(in + ((threadIdx.x + i * 64) & 1023) * 4);
    for (int i = 0; i < 3; i++) x[i] = *reinterpret_cast<const i32x4 *>(in + ((threadIdx.x * 3 + i) & 1023) * 4);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    if (MODE == 0 || MODE == 1 || MODE == 4) {
        for (int i = 0; i < 18; i++) asm volatile("" : "+a"(w[i]));
    } else {
        for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w[i]));
    }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 18; i++) {
            if (MODE == 0) MF_VAV(acc[i % 6], w[i], x[i % 3]);
            if (MODE == 1) MF_AAV(acc[i % 6], w[i], x[i % 3]);
            if (MODE == 2) MF_AVV(acc[i % 6], w[i], x[i % 3]);
            if (MODE == 3) MF_VVV(acc[i % 6], w[i], x[i % 3]);
            if (MODE == 4) MF_VVA(acc[i % 6], w[i], x[i % 3]);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (MODE == 1 || MODE == 2) {
        for (int i = 0; i < 6; i++) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[i]));
    } else {
        for (int i = 0; i < 6; i++) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[i]));
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int *in; float *out; long long *cyc;
    const int nb = 256, iters = 2000;
    hipMalloc(&in, 1024 * 16); hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    std::vector<int> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = 0x3f803f80 + (i * 2654435761u >> 12 & 0x007f007f);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const char *names[5] = {"acc VGPR, W AGPR (srcA), X VGPR", "acc AGPR, W AGPR (srcA), X VGPR", "acc AGPR, W VGPR, X VGPR",
                            "acc VGPR, W VGPR, X VGPR", "acc VGPR, X VGPR (srcA), W AGPR (srcB)"};
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 5; m++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, in, out, cyc, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, in, out, cyc, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, in, out, cyc, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, in, out, cyc, iters);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, in, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(nb);
            hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : c) avg += v; avg /= nb;
            // s_memtime ticks at a fixed 100 MHz; report wall-derived ns per MFMA and the tick count
            printf("%-44s %.3f ms  %.2f ns/MFMA  (memtime ticks/MFMA %.3f)\n", names[m], ms, ms * 1e6 / (iters * 18.0),
                   avg / (iters * 18.0));
        }
    return 0;
}
