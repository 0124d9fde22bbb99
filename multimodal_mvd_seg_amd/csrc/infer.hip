// Sliding-window inference helpers (predict_from_raw_data.py:562-588, :643-714): mirror test-time
// augmentation and Gaussian-weighted accumulation of tile logits.  All planar fp32 [C][D][H][W]; HBM-bound streaming
// kernels, one pass each, no reductions (deterministic by construction: every output element has one writer).
#include "common.h"

namespace mvd {

// dst[c][z][y][x] (+)= src[c][fz][fy][fx], f = flipped index on the axes of `mask` (bit 0: D, 1: H, 2: W).
// One thread per 4 consecutive x (W % 4 == 0 fast path is not assumed: scalar tail-free formulation).
__global__ void k_flip_add(const float *__restrict__ src, float *__restrict__ dst, int C, int D, int H, int W, int mask,
                           int accumulate) {
    const long total = (long)C * D * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int z = (int)(r % D);
        const int c = (int)(r / D);
        const int fz = (mask & 1) ? D - 1 - z : z, fy = (mask & 2) ? H - 1 - y : y, fx = (mask & 4) ? W - 1 - x : x;
        const float v = src[(((size_t)c * D + fz) * H + fy) * W + fx];
        dst[idx] = accumulate ? dst[idx] + v : v;
    }
}

// logits[k][oz+z][oy+y][ox+x] += tile[k][z][y][x] * scale * g[z][y][x];  npred[...] += g   (g == nullptr: weight 1)
__global__ void k_sw_accumulate(const float *__restrict__ tile, const float *__restrict__ g, float scale,
                                float *__restrict__ logits, float *__restrict__ npred, int K, int pd, int ph, int pw,
                                int D, int H, int W, int oz, int oy, int ox) {
    const long pv = (long)pd * ph * pw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < pv; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int x = (int)(r % pw); r /= pw;
        const int y = (int)(r % ph);
        const int z = (int)(r / ph);
        const float wgt = g ? g[idx] : 1.f;
        const size_t o = (((size_t)(oz + z)) * H + (oy + y)) * W + (ox + x);
        npred[o] += wgt;
        const size_t V = (size_t)D * H * W;
        for (int k = 0; k < K; k++) logits[(size_t)k * V + o] += tile[(size_t)k * pv + idx] * scale * wgt;
    }
}

__global__ void k_sw_normalize(float *__restrict__ logits, const float *__restrict__ npred, int K, long V) {
    const long total = (long)K * V;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
        logits[idx] = logits[idx] / npred[idx % V];
}

}  // namespace mvd

using namespace mvd;

static inline unsigned sw_grid(long n) {
    long b = cdiv(n, 256);
    if (b > 16384) b = 16384;
    return (unsigned)(b < 1 ? 1 : b);
}

extern "C" {

int mvd_flip_add(const float *src, float *dst, int C, int D, int H, int W, int mask, int accumulate, void *stream) {
    MVD_REQUIRE(src && dst && src != dst && C > 0 && D > 0 && H > 0 && W > 0 && mask >= 0 && mask < 8, "flip_add: bad arguments");
    hipLaunchKernelGGL(k_flip_add, dim3(sw_grid((long)C * D * H * W)), dim3(256), 0, as_stream(stream), src, dst, C, D, H, W,
                       mask, accumulate);
    return check_launch("flip_add");
}

int mvd_sw_accumulate(const float *tile, const float *gauss, float scale, float *logits, float *npred, int K, int pd, int ph,
                      int pw, int D, int H, int W, int oz, int oy, int ox, void *stream) {
    MVD_REQUIRE(tile && logits && npred && K > 0 && pd > 0 && ph > 0 && pw > 0, "sw_accumulate: bad arguments");
    MVD_REQUIRE(oz >= 0 && oy >= 0 && ox >= 0 && oz + pd <= D && oy + ph <= H && ox + pw <= W,
                "sw_accumulate: tile (%d,%d,%d)+(%d,%d,%d) outside the volume (%d,%d,%d)", oz, oy, ox, pd, ph, pw, D, H, W);
    hipLaunchKernelGGL(k_sw_accumulate, dim3(sw_grid((long)pd * ph * pw)), dim3(256), 0, as_stream(stream), tile, gauss, scale,
                       logits, npred, K, pd, ph, pw, D, H, W, oz, oy, ox);
    return check_launch("sw_accumulate");
}

int mvd_sw_normalize(float *logits, const float *npred, int K, long V, void *stream) {
    MVD_REQUIRE(logits && npred && K > 0 && V > 0, "sw_normalize: bad arguments");
    hipLaunchKernelGGL(k_sw_normalize, dim3(sw_grid((long)K * V)), dim3(256), 0, as_stream(stream), logits, npred, K, V);
    return check_launch("sw_normalize");
}
}
