// Fused global-norm clip + SGD(momentum, nesterov, weight decay) on flat fp32 buffers.
// HBM-bound: sumsq reads g once; step reads p,g,buf and writes p,buf (20 B / parameter).
#include "common.h"

namespace mvd {

__global__ void k_sumsq(const float *__restrict__ g, double *__restrict__ partial, long n) {
    __shared__ double red[16];
    double acc[1] = {0.0};
    const long n4 = n / 4;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 q = g4[i];
        acc[0] += (double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        float v = g[n4 * 4 + threadIdx.x];
        acc[0] += (double)v * v;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}

__device__ inline float sgd_one(float &p, float g, float &b, float gscale, float clip, float lr, float mom, float wd,
                             int first) {
    g = g * gscale;  // data-parallel mean (1/world; exactly the separate `grad *= 1/world` pass it replaces)
    g = g * clip;
    g = g + wd * p;
    b = first ? g : (mom * b + g);
    g = g + mom * b;
    p = p - lr * g;
    return p;
}

// HYPER: the step's scalars come from a 6-float device array {lr, momentum, weight decay, max_norm, grad_scale, first step
// (0/1)} instead of kernel arguments, so that a hipGraph-captured step follows the learning-rate schedule without
// re-capture (nnUNetTrainerMI355's graphed train_step; PolyLR changes lr once per epoch, nnUNetTrainer.py:880)
template <bool HYPER>
__global__ void k_sgd(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf,
                      const float *__restrict__ sumsq, long n, float lr, float mom, float wd, float max_norm,
                      float gscale, int first, const float *__restrict__ hyper) {
    if (HYPER) {
        lr = hyper[0]; mom = hyper[1]; wd = hyper[2]; max_norm = hyper[3]; gscale = hyper[4];
        first = hyper[5] != 0.f;
    }
    float clip = 1.0f;
    if (max_norm > 0.f) {
        float total = sqrtf(sumsq[0]) * gscale;  // sumsq is over the un-scaled buffer
        clip = max_norm / (total + 1e-6f);  // torch.nn.utils.clip_grad_norm_
        if (clip > 1.0f) clip = 1.0f;
    }
    const long n4 = n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(p);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float4 *b4 = reinterpret_cast<float4 *>(buf);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pp = p4[i], gg = g4[i], bb = first ? make_float4(0, 0, 0, 0) : b4[i];
        sgd_one(pp.x, gg.x, bb.x, gscale, clip, lr, mom, wd, first);
        sgd_one(pp.y, gg.y, bb.y, gscale, clip, lr, mom, wd, first);
        sgd_one(pp.z, gg.z, bb.z, gscale, clip, lr, mom, wd, first);
        sgd_one(pp.w, gg.w, bb.w, gscale, clip, lr, mom, wd, first);
        p4[i] = pp;
        b4[i] = bb;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        long i = n4 * 4 + threadIdx.x;
        float pp = p[i], bb = first ? 0.f : buf[i];
        sgd_one(pp, g[i], bb, gscale, clip, lr, mom, wd, first);
        p[i] = pp;
        buf[i] = bb;
    }
}

}  // namespace mvd

using namespace mvd;

static inline long sumsq_blocks(long n) {
    long b = cdiv(n / 4 + 1, 256);
    return b > 1024 ? 1024 : (b < 1 ? 1 : b);
}

extern "C" {

size_t mvd_sumsq_workspace_bytes(long n) { return (size_t)sumsq_blocks(n) * sizeof(double) + 256; }

int mvd_grad_sumsq(const float *g, float *out, long n, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(g && out && ws && n > 0, "grad_sumsq: bad arguments");
    MVD_REQUIRE(((uintptr_t)g & 15) == 0, "grad_sumsq: g must be 16-byte aligned");
    MVD_REQUIRE(ws_bytes >= mvd_sumsq_workspace_bytes(n), "grad_sumsq: workspace too small");
    long bx = sumsq_blocks(n);
    double *partial = reinterpret_cast<double *>(ws);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_sumsq, dim3(bx), dim3(256), 0, s, g, partial, n);
    if (check_launch("grad_sumsq")) return 1;
    return reduce_partials(partial, out, (int)bx, 1, s);
}

int mvd_sgd_nesterov_step(float *p, const float *g, float *buf, const float *sumsq, long n, float lr, float momentum,
                          float weight_decay, float max_norm, float grad_scale, int first_step, void *stream) {
    MVD_REQUIRE(p && g && buf && n > 0 && (max_norm <= 0.f || sumsq), "sgd_step: bad arguments");
    MVD_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0, "sgd_step: buffers must be 16-byte aligned");
    long bx = cdiv(n / 4 + 1, 256);
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(k_sgd<false>, dim3(bx), dim3(256), 0, as_stream(stream), p, g, buf, sumsq, n, lr, momentum,
                       weight_decay, max_norm, grad_scale, first_step, (const float *)nullptr);
    return check_launch("sgd_step");
}

int mvd_sgd_nesterov_step_dev(float *p, const float *g, float *buf, const float *sumsq, long n, const float *hyper,
                              void *stream) {
    MVD_REQUIRE(p && g && buf && sumsq && hyper && n > 0, "sgd_step_dev: bad arguments");
    MVD_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0, "sgd_step_dev: buffers must be 16-byte aligned");
    long bx = cdiv(n / 4 + 1, 256);
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(k_sgd<true>, dim3(bx), dim3(256), 0, as_stream(stream), p, g, buf, sumsq, n, 0.f, 0.f, 0.f, 0.f, 0.f, 0,
                       hyper);
    return check_launch("sgd_step_dev");
}
}
