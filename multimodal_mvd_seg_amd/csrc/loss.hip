// Segmentation head (1x1x1 conv NDHWC -> planar logits), fused softmax+CE+soft-Dice, argmax counts, KL distillation.
// All HBM-bound single-pass kernels; reductions are fp64 per thread -> fixed-order block tree -> partials -> finalize.
#include "common.h"

namespace mvd {

constexpr int KMAX = 8;    // classes (K = 5 on the reference's data, MVDTrainer.py:98)
constexpr int KLCMAX = 32; // channels of the KL softmax (logits <= 8, features 32)

// =============================================================================================== seg head fwd
// block: 256 threads = 64 voxels x 4 class slots; channels staged through LDS in chunks of 32 (coalesced 128-B rows)
template <bool XB>
__global__ void k_seghead_fwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                              float *__restrict__ logits, long V, int C, int K) {
    __shared__ float tile[64][33];
    const int n = blockIdx.y;
    const long v0 = (long)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int vox = t & 63, kq = t >> 6;
    float acc[2] = {0.f, 0.f};
    const size_t xn = (size_t)n * V * C;
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
        for (int j = t; j < 64 * 32; j += 256) {
            int vv = j >> 5, ch = j & 31;
            long v = v0 + vv;
            tile[vv][ch] = (v < V && c0 + ch < C) ? ld1<XB>(x, xn + (size_t)v * C + c0 + ch) : 0.f;
        }
        __syncthreads();
        const int cn = (C - c0 < 32) ? (C - c0) : 32;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int k = kq + 4 * i;
            if (k < K) {
                const float *wk = w + (size_t)k * C + c0;
                float a = acc[i];
                for (int ch = 0; ch < cn; ch++) a += tile[vox][ch] * wk[ch];
                acc[i] = a;
            }
        }
    }
    long v = v0 + vox;
    if (v < V) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int k = kq + 4 * i;
            if (k < K) logits[((size_t)n * K + k) * V + v] = acc[i] + bias[k];
        }
    }
}

// dx[n][v][c] (+)= sum_k dl[n][k][v] * w[k][c]
template <bool XB>
__global__ void k_seghead_dx(const float *__restrict__ dl, const float *__restrict__ w, float *__restrict__ dx, long V,
                             int C, int K, int accumulate) {
    extern __shared__ float sm[];  // w_s[K][C], dl_s[K][64]
    float *w_s = sm;
    float *dl_s = sm + (size_t)K * C;
    const int n = blockIdx.y;
    const long v0 = (long)blockIdx.x * 64;
    const int t = threadIdx.x;
    for (int j = t; j < K * C; j += 256) w_s[j] = w[j];
    for (int j = t; j < K * 64; j += 256) {
        int k = j >> 6, vv = j & 63;
        long v = v0 + vv;
        dl_s[j] = (v < V) ? dl[((size_t)n * K + k) * V + v] : 0.f;
    }
    __syncthreads();
    const size_t dxn = ((size_t)n * V + v0) * C;
    const long nv = (V - v0 < 64) ? (V - v0) : 64;
    for (long j = t; j < nv * C; j += 256) {
        int vv = (int)(j / C), c = (int)(j % C);
        float a = 0.f;
        for (int k = 0; k < K; k++) a += dl_s[k * 64 + vv] * w_s[k * C + c];
        st1<XB>(dx, dxn + j, accumulate ? ld1<XB>(dx, dxn + j) + a : a);
    }
}

// C % 4 == 0 fast paths (HBM-bound: the tiled kernels above ran at 2-3 TB/s on the 128^3 head).
// fwd: one thread per voxel, the C channels as C/4 16-byte loads all in flight, weights as scalar operands; the
// accumulation order per class is the channel order of k_seghead_fwd (same result).  Planar logits: coalesced stores.
// PRO (round 3, ops.NormActSegHeadFn): x is the RAW bf16 output of the last decoder conv; the head reads
// a = bf16(lrelu(fma(x, scale[n][c], shift[n][c]))) -- the arithmetic of k_in_apply_ss16, rounded to bf16 like the tensor that
// pass would have written -- so the activated tensor of the last block is never materialised.
__device__ inline float4 seg_pro4(float4 q, const float *__restrict__ sc, const float *__restrict__ sh, float slope) {
    float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float z = fmaf(f[i], sc[i], sh[i]);
        const float a = fmaxf(z, z * slope);
        f[i] = __uint_as_float((unsigned)f2bf(a) << 16);
    }
    return make_float4(f[0], f[1], f[2], f[3]);
}
// fp32 form (PRO = 2): the expression of k_in_apply_rows<false, false> -- ((x - mean) * rstd) * gamma + beta, z > 0 ? z : z * slope
__device__ inline float4 seg_pro4f(float4 q, const float *__restrict__ mu, const float *__restrict__ rs,
                                   const float *__restrict__ ga, const float *__restrict__ be, float slope) {
    float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float xh = (f[i] - mu[i]) * rs[i];
        const float z = fmaf(xh, ga[i], be[i]);  // (explicit: the same rounding in every kernel that re-computes z)
        f[i] = z > 0.f ? z : z * slope;
    }
    return make_float4(f[0], f[1], f[2], f[3]);
}
template <bool XB, int PRO = 0>
__global__ __launch_bounds__(256) void k_seghead_fwd_vox(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ logits, long V,
                                                         int C, int K, const float *__restrict__ scale = nullptr,
                                                         const float *__restrict__ shift = nullptr, float slope = 0.f,
                                                         const float *__restrict__ gamma = nullptr,
                                                         const float *__restrict__ beta = nullptr) {
    const int n = blockIdx.y;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) acc[k] = 0.f;
    const size_t xo = ((size_t)n * V + v) * C;
    const bool full = (C & 31) == 0;  // every 32-channel chunk whole: the loads need no predicate (a predicate around a load
                                      // makes the compiler wait for the loads before it: eight round trips per chunk)
    for (int c0 = 0; c0 < C; c0 += 32) {
        float4 q[8];
        if (full) {
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = ld4<XB>(x, xo + c0 + 4 * j);
            if (PRO == 1) {   // (host: C % 32 == 0) scale / shift of sample n: wave-uniform, scalar loads
#pragma unroll
                for (int j = 0; j < 8; j++)
                    q[j] = seg_pro4(q[j], scale + (size_t)n * C + c0 + 4 * j, shift + (size_t)n * C + c0 + 4 * j, slope);
            }
            if (PRO == 2) {   // fp32: scale = mean, shift = rstd of sample n
#pragma unroll
                for (int j = 0; j < 8; j++)
                    q[j] = seg_pro4f(q[j], scale + (size_t)n * C + c0 + 4 * j, shift + (size_t)n * C + c0 + 4 * j, gamma + c0 + 4 * j,
                                     beta + c0 + 4 * j, slope);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = (c0 + 4 * j < C) ? ld4<XB>(x, xo + c0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                const float *wk = w + (size_t)k * C + c0;  // wave-uniform: scalar loads
                float a = acc[k];
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (c0 + 4 * j < C) {
                        a = fmaf(q[j].x, wk[4 * j], a);
                        a = fmaf(q[j].y, wk[4 * j + 1], a);
                        a = fmaf(q[j].z, wk[4 * j + 2], a);
                        a = fmaf(q[j].w, wk[4 * j + 3], a);
                    }
                acc[k] = a;
            }
    }
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        if (k < K) logits[((size_t)n * K + k) * V + v] = acc[k] + bias[k];
}

// Deep levels (many channels, few voxels: 320 channels at 8^3 is 1024 threads looping over 320 channels each): P adjacent
// lanes share a voxel, lane p takes the 32-channel chunks p, p + P, ...; the K partial sums meet by xor-shuffles (fixed
// tree: deterministic).  A lane reads 128 (64) contiguous bytes per chunk, the P lanes of a voxel consecutive chunks.
template <bool XB, int P>
__global__ __launch_bounds__(256) void k_seghead_fwd_voxp(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ logits, long V,
                                                          int C, int K) {
    const int n = blockIdx.y;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const long v = gt / P;
    const int p = (int)(gt - v * P);
    const bool live = v < V;
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) acc[k] = 0.f;
    const size_t xo = ((size_t)n * V + (live ? v : 0)) * C;
    // C % 32 == 0 (host): every load below is unconditional -- a dead lane reads voxel 0 and drops its result.  (With a
    // predicate on each load the compiler waited for load j before it issued load j + 1: eight round trips per chunk.)
    for (int c0 = p * 32; c0 < C; c0 += 32 * P) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = ld4<XB>(x, xo + c0 + 4 * j);
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {  // uniform
                const float *wk = w + (size_t)k * C + c0;
                float a = acc[k];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wk + 4 * j);
                    a += q[j].x * w4.x;
                    a += q[j].y * w4.y;
                    a += q[j].z * w4.z;
                    a += q[j].w * w4.w;
                }
                acc[k] = a;
            }
    }
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        if (k < K) {
#pragma unroll
            for (int o = 1; o < P; o <<= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        }
    if (live && p == 0) {
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) logits[((size_t)n * K + k) * V + v] = acc[k] + bias[k];
    }
}

// dx: one thread per (quad of 4 consecutive voxels, 4-channel group): the K class gradients of the quad come in as one
// float4 per class (planar dl), the thread's K x 4 weights stay in registers over its grid-stride loop, and the C/4 lanes
// of a voxel write one full 128-byte line.  k order as in k_seghead_dx.  V % 4 == 0.
template <bool XB>
__global__ __launch_bounds__(256) void k_seghead_dx4(const float *__restrict__ dl, const float *__restrict__ w,
                                                     float *__restrict__ dx, long V, int C, int K, int accumulate) {
    const int n = blockIdx.y;
    const int CG = C >> 2;
    const int g = threadIdx.x % CG, r = threadIdx.x / CG, R = 256 / CG;
    if (r >= R) return;
    float4 wk[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        wk[k] = k < K ? *reinterpret_cast<const float4 *>(w + (size_t)k * C + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long nq = V >> 2;
    for (long q = (long)blockIdx.x * R + r; q < nq; q += (long)gridDim.x * R) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        // all class gradients of the quad in flight at once: classes >= K read class K - 1 again (their weights are zero and
        // the arithmetic below skips them); behind `k < K` each load waited for the one before it
        float4 dq[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            dq[k] = *reinterpret_cast<const float4 *>(dl + ((size_t)n * K + (k < K ? k : K - 1)) * V + 4 * q);
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                const float4 d = dq[k];
                const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a[u].x += dv[u] * wk[k].x;
                    a[u].y += dv[u] * wk[k].y;
                    a[u].z += dv[u] * wk[k].z;
                    a[u].w += dv[u] * wk[k].w;
                }
            }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t o = ((size_t)n * V + 4 * q + u) * C + (size_t)g * 4;
            if (accumulate) {
                const float4 old = ld4<XB>(dx, o);
                a[u].x += old.x; a[u].y += old.y; a[u].z += old.z; a[u].w += old.w;
            }
            st4<XB>(dx, o, a[u].x, a[u].y, a[u].z, a[u].w);
        }
    }
}

// partial[b][k*C + c] = sum over the block's voxels of dl[k][v]*x[v][c]; partial[b][K*C + k] = sum dl[k][v]
template <bool XB>
__global__ void k_seghead_dw(const float *__restrict__ x, const float *__restrict__ dl, double *__restrict__ partial,
                             int N, long V, int C, int K, long chunk) {
    __shared__ float xs[64][33];
    __shared__ float ds[KMAX][64];
    const int t = threadIdx.x;
    const int c_l = t & 31, kq = t >> 5;  // 32 channels x 8 class slots
    const long total = (long)N * V;
    const long g0 = (long)blockIdx.x * chunk;
    long g1 = g0 + chunk;
    if (g1 > total) g1 = total;
    const int nv_out = K * C + K;
    double *po = partial + (size_t)blockIdx.x * nv_out;
    for (int c0 = 0; c0 < C; c0 += 32) {
        double acc = 0.0, accb = 0.0;
        for (long base = g0; base < g1; base += 64) {
            __syncthreads();
            for (int j = t; j < 64 * 32; j += 256) {
                int vv = j >> 5, ch = j & 31;
                long g = base + vv;
                xs[vv][ch] = (g < g1 && c0 + ch < C) ? ld1<XB>(x, (size_t)g * C + c0 + ch) : 0.f;
            }
            for (int j = t; j < K * 64; j += 256) {
                int k = j >> 6, vv = j & 63;
                long g = base + vv;
                float val = 0.f;
                if (g < g1) {
                    long n = g / V, v = g % V;
                    val = dl[((size_t)n * K + k) * V + v];
                }
                ds[k][vv] = val;
            }
            __syncthreads();
            if (kq < K) {
                float a = 0.f, b = 0.f;
#pragma unroll 8
                for (int vv = 0; vv < 64; vv++) {
                    a += ds[kq][vv] * xs[vv][c_l];
                    b += ds[kq][vv];
                }
                acc += (double)a;
                accb += (double)b;
            }
        }
        if (kq < K && c0 + c_l < C) po[(size_t)kq * C + c0 + c_l] = acc;
        if (c0 == 0 && kq < K && c_l == 0) po[(size_t)K * C + kq] = accb;
    }
}

// streaming variant for C % 4 == 0: thread = (4-channel group, row); each thread keeps K x 4 accumulators in
// registers while walking its rows with float4 loads; rows are combined through LDS in a fixed order.
template <bool XB>
__global__ void k_seghead_dw4(const float *__restrict__ x, const float *__restrict__ dl, double *__restrict__ partial,
                              int N, long V, int C, int K, long chunk) {
    extern __shared__ float smf[];  // [R][K*C + K]
    const int t = threadIdx.x;
    const int CG = C / 4, R = blockDim.x / CG;
    const int g = t % CG, r = t / CG;
    const long total = (long)N * V;
    const long g0 = (long)blockIdx.x * chunk;
    long g1 = g0 + chunk;
    if (g1 > total) g1 = total;
    const int nv_out = K * C + K;
    float acc[KMAX][4], accb[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        accb[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) acc[k][j] = 0.f;
    }
    if (r < R) {
        for (long i = g0 + r; i < g1; i += R) {
            const long n = i / V, v = i - n * V;
            const float4 q = ld4<XB>(x, (size_t)i * C + g * 4);
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if (k < K) {
                    const float d = dl[((size_t)n * K + k) * V + v];
                    acc[k][0] += d * q.x; acc[k][1] += d * q.y; acc[k][2] += d * q.z; acc[k][3] += d * q.w;
                    accb[k] += d;
                }
        }
        float *o = smf + (size_t)r * nv_out;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
#pragma unroll
                for (int j = 0; j < 4; j++) o[k * C + g * 4 + j] = acc[k][j];
                if (g == 0) o[K * C + k] = accb[k];
            }
    }
    __syncthreads();
    double *po = partial + (size_t)blockIdx.x * nv_out;
    for (int j = t; j < nv_out; j += blockDim.x) {
        double s = 0;
        for (int rr = 0; rr < R; rr++) s += (double)smf[(size_t)rr * nv_out + j];
        po[j] = s;
    }
}

// V % 4 == 0 variant: a thread walks QUADS of consecutive voxels, so the planar dlogits come in as one float4 per
// class (instead of 4 scalar loads) next to the 4 channel rows; no per-voxel division (n, v advance incrementally)
template <bool XB, int PRO = 0>
__global__ void k_seghead_dw4v(const float *__restrict__ x, const float *__restrict__ dl, double *__restrict__ partial,
                               int N, long V, int C, int K, long chunk, const float *__restrict__ scale = nullptr,
                               const float *__restrict__ shift = nullptr, float slope = 0.f,
                               const float *__restrict__ gamma = nullptr, const float *__restrict__ beta = nullptr) {
    extern __shared__ float smf[];  // [R][K*C + K]
    const int t = threadIdx.x;
    const int CG = C / 4, R = blockDim.x / CG;
    const int g = t % CG, r = t / CG;
    const long total = (long)N * V;
    const long g0 = (long)blockIdx.x * chunk;
    long g1 = g0 + chunk;
    if (g1 > total) g1 = total;
    const int nv_out = K * C + K;
    float acc[KMAX][4], accb[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        accb[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) acc[k][j] = 0.f;
    }
    if (r < R) {
        long i = g0 + 4L * r;
        long n = i / V, v = i - n * V;
        const long step = 4L * R;
        for (; i < g1; i += step) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = ld4<XB>(x, (size_t)(i + u) * C + g * 4);
            if (PRO) {   // the four voxels i .. i + 3 belong to one sample (V % 4 == 0): its scale / shift of this lane's channels
                const float *sc = scale + (size_t)n * C + g * 4, *sh = shift + (size_t)n * C + g * 4;
                const float s4[4] = {sc[0], sc[1], sc[2], sc[3]}, h4[4] = {sh[0], sh[1], sh[2], sh[3]};
                if (PRO == 1) {
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = seg_pro4(q[u], s4, h4, slope);
                } else {   // fp32: (mean, rstd) of the sample, gamma / beta of the channels
                    const float g4[4] = {gamma[g * 4], gamma[g * 4 + 1], gamma[g * 4 + 2], gamma[g * 4 + 3]};
                    const float b4[4] = {beta[g * 4], beta[g * 4 + 1], beta[g * 4 + 2], beta[g * 4 + 3]};
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = seg_pro4f(q[u], s4, h4, g4, b4, slope);
                }
            }
            float4 dq[KMAX];  // (all classes in flight at once, see k_seghead_dx4)
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                dq[k] = *reinterpret_cast<const float4 *>(dl + ((size_t)n * K + (k < K ? k : K - 1)) * V + v);
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if (k < K) {
                    const float4 d = dq[k];
                    // (explicit fma chains: the contraction the compiler picks otherwise differs between the instantiations of
                    // this kernel -- with and without the loader prologue -- and with it the last bit of dW)
                    acc[k][0] = fmaf(d.w, q[3].x, fmaf(d.z, q[2].x, fmaf(d.y, q[1].x, fmaf(d.x, q[0].x, acc[k][0]))));
                    acc[k][1] = fmaf(d.w, q[3].y, fmaf(d.z, q[2].y, fmaf(d.y, q[1].y, fmaf(d.x, q[0].y, acc[k][1]))));
                    acc[k][2] = fmaf(d.w, q[3].z, fmaf(d.z, q[2].z, fmaf(d.y, q[1].z, fmaf(d.x, q[0].z, acc[k][2]))));
                    acc[k][3] = fmaf(d.w, q[3].w, fmaf(d.z, q[2].w, fmaf(d.y, q[1].w, fmaf(d.x, q[0].w, acc[k][3]))));
                    accb[k] += (d.x + d.y) + (d.z + d.w);
                }
            v += step;
            while (v >= V) {
                v -= V;
                n++;
            }
        }
        float *o = smf + (size_t)r * nv_out;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
#pragma unroll
                for (int j = 0; j < 4; j++) o[k * C + g * 4 + j] = acc[k][j];
                if (g == 0) o[K * C + k] = accb[k];
            }
    }
    __syncthreads();
    double *po = partial + (size_t)blockIdx.x * nv_out;
    for (int j = t; j < nv_out; j += blockDim.x) {
        double s = 0;
        for (int rr = 0; rr < R; rr++) s += (double)smf[(size_t)rr * nv_out + j];
        po[j] = s;
    }
}

// =============================================================================================== DC + CE
__device__ inline int label_of(float t, int K) {
    int y = (int)t;  // .long() truncation (robust_ce_loss.py:16)
    return y < 0 ? 0 : (y >= K ? K - 1 : y);
}

__global__ void k_dcce_fwd(const float *__restrict__ logits, const float *__restrict__ target,
                           double *__restrict__ partial, int N, long V, int K) {
    __shared__ double red[(3 * KMAX + 1) * 16];
    const int n = blockIdx.y;
    const float *ln = logits + (size_t)n * K * V;
    const float *tn = target + (size_t)n * V;
    double acc[3 * KMAX + 1];
#pragma unroll
    for (int i = 0; i < 3 * KMAX + 1; i++) acc[i] = 0.0;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float z[KMAX];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; k++) z[k] = ln[(size_t)(k < K ? k : K - 1) * V + v];  // all classes in flight (no predicate)
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) m = fmaxf(m, z[k]);
        const int y = label_of(tn[v], K);
        float s = 0.f, zy = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                float d = z[k] - m;
                if (k == y) zy = d;
                z[k] = expf(d);
                s += z[k];
            }
        const float inv = 1.0f / s;
        acc[3 * KMAX] += (double)(logf(s) - zy);  // -log_softmax(z)[y]
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                float p = z[k] * inv;
                if (k == y) {
                    acc[3 * k + 0] += (double)p;
                    acc[3 * k + 2] += 1.0;
                }
                acc[3 * k + 1] += (double)p;
            }
    }
    block_sum<3 * KMAX + 1>(acc, red);
    if (threadIdx.x == 0) {
        // partial[b][n][3K+1]
        double *po = partial + ((size_t)blockIdx.x * N + n) * (3 * K + 1);
#pragma unroll
        for (int k = 0; k < KMAX; k++)  // static indices: a run-time-indexed acc[] lives in scratch memory (208 B per lane,
            if (k < K) {                // every accumulation a scratch load + store: 162 us for the 128^3 level)
                po[3 * k + 0] = acc[3 * k + 0];
                po[3 * k + 1] = acc[3 * k + 1];
                po[3 * k + 2] = acc[3 * k + 2];
            }
        po[3 * K] = acc[3 * KMAX];
    }
}

__global__ void k_dcce_finalize(const float *__restrict__ stats, int N, const float *__restrict__ dstats, int Nd,
                                float *__restrict__ loss, float *__restrict__ coef, long V, int K, int batch_dice,
                                int do_bg, float smooth, float w_ce, float w_dice) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int S = 3 * K + 1;
    double ce = 0;
    for (int n = 0; n < N; n++) ce += (double)stats[(size_t)n * S + 3 * K];
    ce /= ((double)N * (double)V);
    const int k0 = do_bg ? 0 : 1;
    double dcsum = 0;
    for (size_t i = 0; i < (size_t)Nd * K * 2; i++) coef[i] = 0.f;
    if (batch_dice) {
        const double cnt = (double)(K - k0);
        for (int k = k0; k < K; k++) {
            float I = 0, P = 0, G = 0;  // torch sums the per-sample fp32 values in fp32
            for (int n = 0; n < Nd; n++) {
                I += dstats[(size_t)n * S + 3 * k + 0];
                P += dstats[(size_t)n * S + 3 * k + 1];
                G += dstats[(size_t)n * S + 3 * k + 2];
            }
            float den = G + P + smooth;
            bool clipped = den < 1e-8f;
            if (clipped) den = 1e-8f;
            float num = 2.f * I + smooth;
            dcsum += (double)(num / den);
            float cI = (float)(-(1.0 / cnt) * 2.0 / (double)den);
            float cP = clipped ? 0.f : (float)((1.0 / cnt) * (double)num / ((double)den * (double)den));
            for (int n = 0; n < Nd; n++) {
                coef[((size_t)n * K + k) * 2 + 0] = w_dice * cI;
                coef[((size_t)n * K + k) * 2 + 1] = w_dice * cP;
            }
        }
        dcsum /= cnt;
    } else {
        const double cnt = (double)Nd * (double)(K - k0);
        for (int n = 0; n < Nd; n++)
            for (int k = k0; k < K; k++) {
                float I = dstats[(size_t)n * S + 3 * k + 0], P = dstats[(size_t)n * S + 3 * k + 1],
                      G = dstats[(size_t)n * S + 3 * k + 2];
                float den = G + P + smooth;
                bool clipped = den < 1e-8f;
                if (clipped) den = 1e-8f;
                float num = 2.f * I + smooth;
                dcsum += (double)(num / den);
                coef[((size_t)n * K + k) * 2 + 0] = w_dice * (float)(-(1.0 / cnt) * 2.0 / (double)den);
                coef[((size_t)n * K + k) * 2 + 1] =
                    clipped ? 0.f : w_dice * (float)((1.0 / cnt) * (double)num / ((double)den * (double)den));
            }
        dcsum /= cnt;
    }
    loss[1] = (float)ce;
    loss[2] = (float)(-dcsum);
    loss[0] = w_ce * (float)ce + w_dice * (float)(-dcsum);
}

__global__ void k_dcce_bwd(const float *__restrict__ logits, const float *__restrict__ target,
                           const float *__restrict__ coef, const float *__restrict__ gscale_dev, float gscale_host,
                           float *__restrict__ dlogits, int N, long V, int K, float w_ce) {
    const int n = blockIdx.y;
    const float *ln = logits + (size_t)n * K * V;
    const float *tn = target + (size_t)n * V;
    float *dn = dlogits + (size_t)n * K * V;
    const float g = gscale_host * (gscale_dev ? gscale_dev[0] : 1.0f);
    const float cew = w_ce / ((float)N * (float)V);
    float cI[KMAX], cP[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        if (k < K) {
            cI[k] = coef[((size_t)n * K + k) * 2 + 0];
            cP[k] = coef[((size_t)n * K + k) * 2 + 1];
        }
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float z[KMAX];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; k++) z[k] = ln[(size_t)(k < K ? k : K - 1) * V + v];  // all classes in flight (no predicate)
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) m = fmaxf(m, z[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                z[k] = expf(z[k] - m);
                s += z[k];
            }
        const float inv = 1.0f / s;
        const int y = label_of(tn[v], K);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                z[k] *= inv;  // p_k
                float q = cP[k] + (k == y ? cI[k] : 0.f);
                dot += z[k] * q;
            }
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                float q = cP[k] + (k == y ? cI[k] : 0.f);
                float d = cew * (z[k] - (k == y ? 1.f : 0.f)) + z[k] * (q - dot);
                dn[(size_t)k * V + v] = g * d;
            }
    }
}

__global__ void k_argmax_counts(const float *__restrict__ logits, const float *__restrict__ target,
                                unsigned long long *__restrict__ counts, long V, int K) {
    __shared__ unsigned int c_s[KMAX * 3];
    const int n = blockIdx.y;
    if (threadIdx.x < KMAX * 3) c_s[threadIdx.x] = 0;
    __syncthreads();
    const float *ln = logits + (size_t)n * K * V;
    const float *tn = target + (size_t)n * V;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        int best = 0;
        float bv = ln[v];
        for (int k = 1; k < K; k++) {
            float z = ln[(size_t)k * V + v];
            if (z > bv) {  // first maximum wins (torch.argmax)
                bv = z;
                best = k;
            }
        }
        int y = label_of(tn[v], K);
        if (best == y)
            atomicAdd(&c_s[3 * y + 0], 1u);
        else {
            atomicAdd(&c_s[3 * best + 1], 1u);
            atomicAdd(&c_s[3 * y + 2], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < K * 3 && c_s[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)c_s[threadIdx.x]);
}

// =============================================================================================== softmax channel select
template <bool BWD>
__global__ void k_softmax_select(const float *__restrict__ logits, const float *__restrict__ g, float *__restrict__ out,
                                 long V, int K, int sel) {
    const int n = blockIdx.y;
    const float *ln = logits + (size_t)n * K * V;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
        float z[KMAX];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; k++) z[k] = ln[(size_t)(k < K ? k : K - 1) * V + v];  // all classes in flight (no predicate)
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) m = fmaxf(m, z[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k < K) {
                z[k] = expf(z[k] - m);
                s += z[k];
            }
        const float inv = 1.0f / s;
        float ps = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
            if (k == sel) ps = z[k] * inv;
        if (!BWD) {
            out[(size_t)n * V + v] = ps;
        } else {
            const float gg = g[(size_t)n * V + v] * ps;
#pragma unroll
            for (int k = 0; k < KMAX; k++)
                if (k < K) out[((size_t)n * K + k) * V + v] = gg * ((k == sel ? 1.f : 0.f) - z[k] * inv);
        }
    }
}

__global__ void k_label_mask(const float *__restrict__ labels, float *__restrict__ mask, long n, float value) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        mask[i] = (labels[i] == value) ? 1.f : 0.f;
}

// =============================================================================================== KL distillation
struct KlIdx {
    long sn, sc, sv;
};

template <bool BWD>
__global__ void k_kl(const float *__restrict__ ys, const float *__restrict__ yt, double *__restrict__ partial,
                     float *__restrict__ gs, float *__restrict__ gt, const float *__restrict__ gscale_dev,
                     float gscale_host, int N, int C, long V, KlIdx ix, float T, float eps_s, int pad) {
    __shared__ double red[16];
    const int Ce = pad ? C + 1 : C;
    const float invT = 1.0f / T;
    const double coef = (double)T * (double)T / ((double)N * (double)Ce * (double)V);
    const float g = BWD ? gscale_host * (gscale_dev ? gscale_dev[0] : 1.0f) * (float)coef * invT : 0.f;
    double acc[1] = {0.0};
    const long total = (long)N * V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / V, v = i % V;
        const size_t base = (size_t)n * ix.sn + (size_t)v * ix.sv;
        float a[KLCMAX + 1], b[KLCMAX + 1];
#pragma unroll
        for (int c = 0; c < KLCMAX + 1; c++) a[c] = b[c] = 0.f;  // (defined everywhere: the arrays stay in registers)
        float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
        for (int c = 0; c < KLCMAX + 1; c++)
            if (c < Ce) {
                float s_ = (c < C) ? ys[base + (size_t)c * ix.sc] : 0.f;
                float t_ = (c < C) ? yt[base + (size_t)c * ix.sc] : 0.f;
                a[c] = s_ * invT + eps_s;
                b[c] = t_ * invT;
                ma = fmaxf(ma, a[c]);
                mb = fmaxf(mb, b[c]);
            }
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int c = 0; c < KLCMAX + 1; c++)
            if (c < Ce) {
                sa += expf(a[c] - ma);
                sb += expf(b[c] - mb);
            }
        const float lsa = logf(sa), lsb = logf(sb);
        float kl = 0.f;
#pragma unroll
        for (int c = 0; c < KLCMAX + 1; c++)
            if (c < Ce) {
                float lq = a[c] - ma - lsa;  // log_softmax(student)
                float lp = b[c] - mb - lsb;  // log softmax(teacher)
                float p = expf(lp);
                float d = lp - lq;
                kl += (p > 0.f) ? p * d : 0.f;
                if (BWD) {
                    a[c] = expf(lq) - p;  // dL/du_s (times 1/T folded in g)
                    b[c] = d;             // c_j
                }
            }
        if (BWD) {
            if (gs) {
#pragma unroll
                for (int c = 0; c < KLCMAX; c++)
                    if (c < C) gs[base + (size_t)c * ix.sc] = g * a[c];
            }
            if (gt) {
                // p_k (c_k - sum_j p_j c_j); p_j = q_j - a_j ... recompute p from lp: p = exp(lp)
                float dot = kl;  // sum_j p_j c_j
#pragma unroll
                for (int c = 0; c < KLCMAX; c++)
                    if (c < C) {
                        float lp = (yt[base + (size_t)c * ix.sc] * invT) - mb - lsb;
                        float p = expf(lp);
                        gt[base + (size_t)c * ix.sc] = g * p * (b[c] - dot);
                    }
            }
        } else {
            acc[0] += (double)kl;
        }
    }
    if (!BWD) {
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = acc[0] * coef;
    }
}

// Dense channel-last rows (the NDHWC feature maps of the feature distillation: [N*V rows][C], C in {4,8,16,32}):
// C/4 adjacent lanes share a voxel, each holds 4 channels from one 16-byte (fp32) / 8-byte (bf16) load -- fully
// coalesced -- and the softmax max / sums go across those lanes with xor-shuffles (fixed tree: deterministic).
// XB: inputs (and the gradients written back) are bf16; arithmetic is fp32 either way.
template <bool BWD, bool XB>
__global__ void k_kl_rows(const void *__restrict__ ys, const void *__restrict__ yt, double *__restrict__ partial,
                          void *__restrict__ gs, void *__restrict__ gt, const float *__restrict__ gscale_dev,
                          float gscale_host, int N, int C, long V, float T, float eps_s) {
    __shared__ double red[16];
    const int parts = C >> 2;  // 1, 2, 4 or 8 lanes per voxel
    const int part = threadIdx.x & (parts - 1);
    const long rows = (long)N * V;
    const int rpb = blockDim.x / parts;
    const float invT = 1.0f / T;
    const double coef = (double)T * (double)T / ((double)N * (double)C * (double)V);
    const float g = BWD ? gscale_host * (gscale_dev ? gscale_dev[0] : 1.0f) * (float)coef * invT : 0.f;
    double acc[1] = {0.0};
    // every lane runs the same number of iterations (the shuffles need the whole voxel group alive)
    for (long r0 = (long)blockIdx.x * rpb; r0 < rows; r0 += (long)gridDim.x * rpb) {
        const long row = r0 + threadIdx.x / parts;
        const bool ok = row < rows;
        const size_t e = (size_t)(ok ? row : 0) * C + part * 4;
        const float4 s4 = ld4<XB>(ys, e), t4 = ld4<XB>(yt, e);
        float a[4] = {s4.x * invT + eps_s, s4.y * invT + eps_s, s4.z * invT + eps_s, s4.w * invT + eps_s};
        float b[4] = {t4.x * invT, t4.y * invT, t4.z * invT, t4.w * invT};
        float ma = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), mb = fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]));
        for (int o = 1; o < parts; o <<= 1) {
            ma = fmaxf(ma, __shfl_xor(ma, o, 64));
            mb = fmaxf(mb, __shfl_xor(mb, o, 64));
        }
        float sa = (expf(a[0] - ma) + expf(a[1] - ma)) + (expf(a[2] - ma) + expf(a[3] - ma));
        float sb = (expf(b[0] - mb) + expf(b[1] - mb)) + (expf(b[2] - mb) + expf(b[3] - mb));
        for (int o = 1; o < parts; o <<= 1) {
            sa += __shfl_xor(sa, o, 64);
            sb += __shfl_xor(sb, o, 64);
        }
        const float lsa = logf(sa), lsb = logf(sb);
        float p[4], d[4], kl = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float lq = a[c] - ma - lsa, lp = b[c] - mb - lsb;
            p[c] = expf(lp);
            d[c] = lp - lq;
            kl += (p[c] > 0.f) ? p[c] * d[c] : 0.f;
            if (BWD) a[c] = expf(lq) - p[c];
        }
        for (int o = 1; o < parts; o <<= 1) kl += __shfl_xor(kl, o, 64);
        if (BWD) {
            if (ok && gs) st4<XB>(gs, e, g * a[0], g * a[1], g * a[2], g * a[3]);
            if (ok && gt)
                st4<XB>(gt, e, g * p[0] * (d[0] - kl), g * p[1] * (d[1] - kl), g * p[2] * (d[2] - kl), g * p[3] * (d[3] - kl));
        } else if (ok && part == 0) {
            acc[0] += (double)kl;
        }
    }
    if (!BWD) {
        block_sum<1>(acc, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = acc[0] * coef;
    }
}

static inline bool kl_rows_ok(int C, long V, long sn, long sc, long sv, int pad) {
    return !pad && (C == 4 || C == 8 || C == 16 || C == 32) && sc == 1 && sv == C && sn == V * (long)C;
}

// ---- plain MSE: l2_loss(channel_wise=False) = mean(|a-b|^2) (other_loss.py:77-78); elementwise, layout-agnostic
__global__ void k_mse_fwd(const float *__restrict__ a, const float *__restrict__ b, double *__restrict__ partial, long n) {
    __shared__ double red[16];
    double acc[1] = {0.0};
    const long n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 x = a4[i], y = b4[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc[0] += (double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2 + (double)d3 * d3;
    }
    if (blockIdx.x == 0 && (long)threadIdx.x < (n & 3)) {
        const float d = a[n4 * 4 + threadIdx.x] - b[n4 * 4 + threadIdx.x];
        acc[0] += (double)d * d;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
// out[0] = (sum_b partial[b]) / n, summed in a fixed order by one wave
__global__ void k_mse_finish(const double *__restrict__ partial, float *__restrict__ out, int nblk, long n) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = (float)(s / (double)n);
}
__global__ void k_mse_bwd(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ g,
                          float *__restrict__ ga, float *__restrict__ gb, long n) {
    const float c = g[0] * (float)(2.0 / (double)n);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = c * (a[i] - b[i]);
        if (ga) ga[i] = v;
        if (gb) gb[i] = -v;
    }
}

}  // namespace mvd

using namespace mvd;

static inline long grid_for(long n, long cap) {
    long b = cdiv(n, 256);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return b;
}
static inline long cap_per_n(long total_blocks, int N) {
    long c = total_blocks / (N > 0 ? N : 1);
    return c > 0 ? c : 1;
}

extern "C" {

static int seghead_fwd_impl(const float *x, bool xb, const float *w, const float *bias, float *logits, int N, long V,
                            int C, int K, void *stream) {
    MVD_REQUIRE(x && w && bias && logits, "seghead_fwd: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && K > 0 && K <= KMAX, "seghead_fwd: bad shape (K<=8)");
    if (C % 32 == 0 && C >= 64 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0 && (long)N * V < (1L << 20)) {
        // few voxels, many channels: 2 / 4 / 8 lanes per voxel (>= 2 chunks of 32 channels per lane where C allows)
        const int P = C >= 256 ? 8 : (C >= 128 ? 4 : 2);
        const dim3 grid(cdiv(V * P, 256), N);
#define MVD_SH_LAUNCH(PP)                                                                                                  \
    hipLaunchKernelGGL((xb ? k_seghead_fwd_voxp<true, PP> : k_seghead_fwd_voxp<false, PP>), grid, dim3(256), 0, as_stream(stream), x, \
                       w, bias, logits, V, C, K)
        if (P == 8) MVD_SH_LAUNCH(8);
        else if (P == 4) MVD_SH_LAUNCH(4);
        else MVD_SH_LAUNCH(2);
#undef MVD_SH_LAUNCH
        return check_launch("seghead_fwd (lanes per voxel)");
    }
    if (C % 4 == 0 && (((uintptr_t)x) & 15) == 0) {
        auto kv = xb ? k_seghead_fwd_vox<true> : k_seghead_fwd_vox<false>;
        hipLaunchKernelGGL(kv, dim3(cdiv(V, 256), N), dim3(256), 0, as_stream(stream), x, w, bias, logits, V, C, K,
                           (const float *)nullptr, (const float *)nullptr, 0.f, (const float *)nullptr, (const float *)nullptr);
        return check_launch("seghead_fwd");
    }
    auto kern = xb ? k_seghead_fwd<true> : k_seghead_fwd<false>;
    hipLaunchKernelGGL(kern, dim3(cdiv(V, 64), N), dim3(256), 0, as_stream(stream), x, w, bias, logits, V, C, K);
    return check_launch("seghead_fwd");
}
int mvd_seghead_fwd(const float *x, const float *w, const float *bias, float *logits, int N, long V, int C, int K,
                    void *stream) {
    return seghead_fwd_impl(x, false, w, bias, logits, N, V, C, K, stream);
}
int mvd_seghead_fwd_bf16(const uint16_t *x, const float *w, const float *bias, float *logits, int N, long V, int C,
                         int K, void *stream) {
    return seghead_fwd_impl(reinterpret_cast<const float *>(x), true, w, bias, logits, N, V, C, K, stream);
}

static long seghead_chunk(long total, int *nblk) {
    // (a software-pipelined loop -- next quad's loads under this quad's FMAs -- was tried in round 2 and ran 1.4-2x
    // SLOWER: 96 more registers per thread; what the pass lacked on the lower levels was workgroups, not overlap)
    long nb = total / 512;
    // deep levels (8^3 ... 16^3: 1k - 8k voxels of 320 - 256 channels): 2 - 16 workgroups left the pass at 50 us for half a
    // megabyte; chunks of 64 voxels give up to 128 workgroups there
    if (nb < 128) nb = total / 64 < 128 ? total / 64 : 128;
    if (nb < 1) nb = 1;
    if (nb > 8192) nb = 8192;  // up to 32 workgroups per CU: the weight-gradient pass is a pure stream over x and dl
    long chunk = cdiv(total, nb);
    chunk = cdiv(chunk, 64) * 64;
    *nblk = (int)cdiv(total, chunk);
    return chunk;
}

size_t mvd_seghead_bwd_workspace_bytes(int N, long V, int C, int K) {
    int nblk;
    seghead_chunk((long)N * V, &nblk);
    return (size_t)nblk * (K * C + K) * sizeof(double) + 256;
}

static int seghead_bwd_impl(const float *x, bool xb, const float *w, const float *dlogits, float *dx, float *dw,
                            float *dbias, int N, long V, int C, int K, int accumulate, void *ws, size_t ws_bytes,
                            void *stream) {
    MVD_REQUIRE(x && w && dlogits && ws, "seghead_bwd: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && K > 0 && K <= KMAX, "seghead_bwd: bad shape (K<=8)");
    MVD_REQUIRE(ws_bytes >= mvd_seghead_bwd_workspace_bytes(N, V, C, K), "seghead_bwd: workspace too small");
    hipStream_t s = as_stream(stream);
    if (dx) {
        if (C % 4 == 0 && C / 4 <= 256 && V % 4 == 0 && (((uintptr_t)dx | (uintptr_t)w | (uintptr_t)dlogits) & 15) == 0) {
            const long R = 256 / (C / 4);
            long nb = cdiv(V / 4, R);
            if (nb > 8192) nb = 8192;
            hipLaunchKernelGGL(xb ? k_seghead_dx4<true> : k_seghead_dx4<false>, dim3((unsigned)nb, N), dim3(256), 0, s, dlogits, w,
                               dx, V, C, K, accumulate);
        } else {
            size_t sm = ((size_t)K * C + (size_t)K * 64) * sizeof(float);
            hipLaunchKernelGGL(xb ? k_seghead_dx<true> : k_seghead_dx<false>, dim3(cdiv(V, 64), N), dim3(256), sm, s, dlogits, w,
                               dx, V, C, K, accumulate);
        }
        if (check_launch("seghead_dx")) return 1;
    }
    if (dw) {
        MVD_REQUIRE(dbias, "seghead_bwd: dbias required with dw");
        int nblk;
        long chunk = seghead_chunk((long)N * V, &nblk);
        double *partial = reinterpret_cast<double *>(ws);
        const int CG = C / 4, R = (C % 4 == 0 && CG <= 256) ? 256 / CG : 0;
        const size_t smb = (size_t)R * (K * C + K) * sizeof(float);
        if (R >= 1 && smb <= 60 * 1024 && (((uintptr_t)x) & 15) == 0 && V % 4 == 0 && (((uintptr_t)dlogits) & 15) == 0)
            hipLaunchKernelGGL(xb ? k_seghead_dw4v<true> : k_seghead_dw4v<false>, dim3(nblk), dim3((R * CG + 63) / 64 * 64),
                               smb, s, x, dlogits, partial, N, V, C, K, chunk, (const float *)nullptr, (const float *)nullptr, 0.f,
                               (const float *)nullptr, (const float *)nullptr);
        else if (R >= 1 && smb <= 60 * 1024 && (((uintptr_t)x) & 15) == 0)
            hipLaunchKernelGGL(xb ? k_seghead_dw4<true> : k_seghead_dw4<false>, dim3(nblk), dim3((R * CG + 63) / 64 * 64), smb, s, x, dlogits, partial, N, V,
                               C, K, chunk);
        else
            hipLaunchKernelGGL(xb ? k_seghead_dw<true> : k_seghead_dw<false>, dim3(nblk), dim3(256), 0, s, x, dlogits, partial,
                               N, V, C, K, chunk);
        if (check_launch("seghead_dw")) return 1;
        // outputs [K*C] then [K]: dw and dbias are separate buffers -> two reduces over the same partials
        if (reduce_partials(partial, dw, nblk, K * C, s, K * C + K, 0)) return 1;
        if (reduce_partials(partial, dbias, nblk, K, s, K * C + K, K * C)) return 1;
    }
    return 0;
}

int mvd_seghead_bwd(const float *x, const float *w, const float *dlogits, float *dx, float *dw, float *dbias, int N,
                    long V, int C, int K, int accumulate, void *ws, size_t ws_bytes, void *stream) {
    return seghead_bwd_impl(x, false, w, dlogits, dx, dw, dbias, N, V, C, K, accumulate, ws, ws_bytes, stream);
}
int mvd_seghead_bwd_bf16(const uint16_t *x, const float *w, const float *dlogits, uint16_t *dx, float *dw, float *dbias,
                         int N, long V, int C, int K, int accumulate, void *ws, size_t ws_bytes, void *stream) {
    return seghead_bwd_impl(reinterpret_cast<const float *>(x), true, w, dlogits, reinterpret_cast<float *>(dx), dw, dbias,
                            N, V, C, K, accumulate, ws, ws_bytes, stream);
}

// ---- the seg head reading the RAW output of the last decoder conv (InstanceNorm-apply + LeakyReLU in its loaders)
int mvd_seghead_bf16_fused_ok(int N, long V, int C, int K) {
    return (N > 0 && N <= 65535 && V > 0 && V % 4 == 0 && C % 32 == 0 && C / 4 <= 256 && K > 0 && K <= KMAX &&
            (size_t)(256 / (C / 4)) * (K * C + K) * sizeof(float) <= 60 * 1024) ? 1 : 0;
}

int mvd_seghead_fwd_bf16_fused(const uint16_t *x, const float *scale, const float *shift, float slope, const float *w,
                               const float *bias, float *logits, int N, long V, int C, int K, void *stream) {
    MVD_REQUIRE(x && scale && shift && w && bias && logits, "seghead_fwd_bf16_fused: null pointer");
    MVD_REQUIRE(mvd_seghead_bf16_fused_ok(N, V, C, K) && (((uintptr_t)x) & 15) == 0,
                "seghead_fwd_bf16_fused: shape not served (query mvd_seghead_bf16_fused_ok)");
    hipLaunchKernelGGL((k_seghead_fwd_vox<true, 1>), dim3(cdiv(V, 256), N), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float *>(x), w, bias, logits, V, C, K, scale, shift, slope, (const float *)nullptr,
                       (const float *)nullptr);
    return check_launch("seghead_fwd (fused InstanceNorm + LeakyReLU loader)");
}

// dx = d(activated input) (the caller runs the InstanceNorm backward on it), dw / dbias over the re-computed activation
int mvd_seghead_bwd_bf16_fused(const uint16_t *x, const float *scale, const float *shift, float slope, const float *w,
                               const float *dlogits, uint16_t *dx, float *dw, float *dbias, int N, long V, int C, int K,
                               int accumulate, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && scale && shift && w && dlogits && ws, "seghead_bwd_bf16_fused: null pointer");
    MVD_REQUIRE(mvd_seghead_bf16_fused_ok(N, V, C, K) && (((uintptr_t)x | (uintptr_t)dlogits | (uintptr_t)w) & 15) == 0,
                "seghead_bwd_bf16_fused: shape not served (query mvd_seghead_bf16_fused_ok)");
    MVD_REQUIRE(ws_bytes >= mvd_seghead_bwd_workspace_bytes(N, V, C, K), "seghead_bwd_bf16_fused: workspace too small");
    hipStream_t s = as_stream(stream);
    if (dx) {   // does not read x: the plain kernel
        MVD_REQUIRE((((uintptr_t)dx) & 15) == 0, "seghead_bwd_bf16_fused: dx alignment");
        const long R = 256 / (C / 4);
        long nb = cdiv(V / 4, R);
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(k_seghead_dx4<true>, dim3((unsigned)nb, N), dim3(256), 0, s, dlogits, w, reinterpret_cast<float *>(dx), V,
                           C, K, accumulate);
        if (check_launch("seghead_dx")) return 1;
    }
    if (dw) {
        MVD_REQUIRE(dbias, "seghead_bwd_bf16_fused: dbias required with dw");
        int nblk;
        long chunk = seghead_chunk((long)N * V, &nblk);
        double *partial = reinterpret_cast<double *>(ws);
        const int CG = C / 4, R = 256 / CG;
        const size_t smb = (size_t)R * (K * C + K) * sizeof(float);
        hipLaunchKernelGGL((k_seghead_dw4v<true, 1>), dim3(nblk), dim3((R * CG + 63) / 64 * 64), smb, s,
                           reinterpret_cast<const float *>(x), dlogits, partial, N, V, C, K, chunk, scale, shift, slope,
                           (const float *)nullptr, (const float *)nullptr);
        if (check_launch("seghead_dw (fused loader)")) return 1;
        if (reduce_partials(partial, dw, nblk, K * C, s, K * C + K, 0)) return 1;
        if (reduce_partials(partial, dbias, nblk, K, s, K * C + K, K * C)) return 1;
    }
    return 0;
}

// fp32 twins: mean / rstd / gamma / beta instead of scale / shift (the arithmetic of the fp32 apply pass)
int mvd_seghead_fwd_fused(const float *x, const float *mean, const float *rstd, const float *gamma, const float *beta, float slope,
                          const float *w, const float *bias, float *logits, int N, long V, int C, int K, void *stream) {
    MVD_REQUIRE(x && mean && rstd && gamma && beta && w && bias && logits, "seghead_fwd_fused: null pointer");
    MVD_REQUIRE(mvd_seghead_bf16_fused_ok(N, V, C, K) && (((uintptr_t)x) & 15) == 0,
                "seghead_fwd_fused: shape not served (query mvd_seghead_bf16_fused_ok)");
    hipLaunchKernelGGL((k_seghead_fwd_vox<false, 2>), dim3(cdiv(V, 256), N), dim3(256), 0, as_stream(stream), x, w, bias, logits, V, C,
                       K, mean, rstd, slope, gamma, beta);
    return check_launch("seghead_fwd (fused InstanceNorm + LeakyReLU loader, fp32)");
}

int mvd_seghead_bwd_fused(const float *x, const float *mean, const float *rstd, const float *gamma, const float *beta, float slope,
                          const float *w, const float *dlogits, float *dx, float *dw, float *dbias, int N, long V, int C, int K,
                          int accumulate, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && mean && rstd && gamma && beta && w && dlogits && ws, "seghead_bwd_fused: null pointer");
    MVD_REQUIRE(mvd_seghead_bf16_fused_ok(N, V, C, K) && (((uintptr_t)x | (uintptr_t)dlogits | (uintptr_t)w) & 15) == 0,
                "seghead_bwd_fused: shape not served (query mvd_seghead_bf16_fused_ok)");
    MVD_REQUIRE(ws_bytes >= mvd_seghead_bwd_workspace_bytes(N, V, C, K), "seghead_bwd_fused: workspace too small");
    hipStream_t s = as_stream(stream);
    if (dx) {
        MVD_REQUIRE((((uintptr_t)dx) & 15) == 0, "seghead_bwd_fused: dx alignment");
        const long R = 256 / (C / 4);
        long nb = cdiv(V / 4, R);
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(k_seghead_dx4<false>, dim3((unsigned)nb, N), dim3(256), 0, s, dlogits, w, dx, V, C, K, accumulate);
        if (check_launch("seghead_dx")) return 1;
    }
    if (dw) {
        MVD_REQUIRE(dbias, "seghead_bwd_fused: dbias required with dw");
        int nblk;
        long chunk = seghead_chunk((long)N * V, &nblk);
        double *partial = reinterpret_cast<double *>(ws);
        const int CG = C / 4, R = 256 / CG;
        const size_t smb = (size_t)R * (K * C + K) * sizeof(float);
        hipLaunchKernelGGL((k_seghead_dw4v<false, 2>), dim3(nblk), dim3((R * CG + 63) / 64 * 64), smb, s, x, dlogits, partial, N, V, C,
                           K, chunk, mean, rstd, slope, gamma, beta);
        if (check_launch("seghead_dw (fused loader, fp32)")) return 1;
        if (reduce_partials(partial, dw, nblk, K * C, s, K * C + K, 0)) return 1;
        if (reduce_partials(partial, dbias, nblk, K, s, K * C + K, K * C)) return 1;
    }
    return 0;
}

size_t mvd_dcce_workspace_bytes(int N, long V, int K) {
    long bx = grid_for(V, cap_per_n(2048, N));
    return (size_t)bx * N * (3 * K + 1) * sizeof(double) + 256;
}

int mvd_dcce_fwd(const float *logits, const float *target, float *stats, int N, long V, int K, void *ws, size_t ws_bytes,
                 void *stream) {
    MVD_REQUIRE(logits && target && stats && ws, "dcce_fwd: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && K >= 2 && K <= KMAX, "dcce_fwd: bad shape (2<=K<=8)");
    MVD_REQUIRE(ws_bytes >= mvd_dcce_workspace_bytes(N, V, K), "dcce_fwd: workspace too small");
    long bx = grid_for(V, cap_per_n(2048, N));
    double *partial = reinterpret_cast<double *>(ws);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_dcce_fwd, dim3(bx, N), dim3(256), 0, s, logits, target, partial, N, V, K);
    if (check_launch("dcce_fwd")) return 1;
    return reduce_partials(partial, stats, (int)bx, N * (3 * K + 1), s);
}

int mvd_dcce_finalize(const float *stats, int N, const float *dstats, int Nd, float *loss, float *coef, long V, int K,
                      int batch_dice, int do_bg, float smooth, float w_ce, float w_dice, void *stream) {
    MVD_REQUIRE(stats && loss && coef && N > 0 && K >= 2 && K <= KMAX, "dcce_finalize: bad arguments");
    if (!dstats) {
        dstats = stats;
        Nd = N;
    }
    hipLaunchKernelGGL(k_dcce_finalize, dim3(1), dim3(64), 0, as_stream(stream), stats, N, dstats, Nd, loss, coef, V, K,
                       batch_dice, do_bg, smooth, w_ce, w_dice);
    return check_launch("dcce_finalize");
}

int mvd_dcce_bwd(const float *logits, const float *target, const float *coef, const float *gscale_dev, float gscale_host,
                 float *dlogits, int N, long V, int K, float w_ce, void *stream) {
    MVD_REQUIRE(logits && target && coef && dlogits, "dcce_bwd: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && K >= 2 && K <= KMAX, "dcce_bwd: bad shape");
    long bx = grid_for(V, cap_per_n(4096, N));
    hipLaunchKernelGGL(k_dcce_bwd, dim3(bx, N), dim3(256), 0, as_stream(stream), logits, target, coef, gscale_dev,
                       gscale_host, dlogits, N, V, K, w_ce);
    return check_launch("dcce_bwd");
}

int mvd_argmax_counts(const float *logits, const float *target, long long *counts, int N, long V, int K, void *stream) {
    MVD_REQUIRE(logits && target && counts, "argmax_counts: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && K >= 1 && K <= KMAX, "argmax_counts: bad shape");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(counts, 0, sizeof(long long) * K * 3, s) != hipSuccess) {
        set_error("argmax_counts: memset failed");
        return 1;
    }
    long bx = grid_for(V, cap_per_n(1024, N));
    hipLaunchKernelGGL(k_argmax_counts, dim3(bx, N), dim3(256), 0, s, logits, target,
                       reinterpret_cast<unsigned long long *>(counts), V, K);
    return check_launch("argmax_counts");
}

int mvd_softmax_select_fwd(const float *logits, float *p_sel, int N, long V, int K, int sel, void *stream) {
    MVD_REQUIRE(logits && p_sel && N > 0 && N <= 65535 && V > 0 && K >= 1 && K <= KMAX && sel >= 0 && sel < K,
                "softmax_select_fwd: bad arguments");
    long bx = grid_for(V, cap_per_n(4096, N));
    hipLaunchKernelGGL(k_softmax_select<false>, dim3(bx, N), dim3(256), 0, as_stream(stream), logits, nullptr, p_sel, V,
                       K, sel);
    return check_launch("softmax_select_fwd");
}
int mvd_softmax_select_bwd(const float *logits, const float *g, float *dlogits, int N, long V, int K, int sel,
                           void *stream) {
    MVD_REQUIRE(logits && g && dlogits && N > 0 && N <= 65535 && V > 0 && K >= 1 && K <= KMAX && sel >= 0 && sel < K,
                "softmax_select_bwd: bad arguments");
    long bx = grid_for(V, cap_per_n(4096, N));
    hipLaunchKernelGGL(k_softmax_select<true>, dim3(bx, N), dim3(256), 0, as_stream(stream), logits, g, dlogits, V, K,
                       sel);
    return check_launch("softmax_select_bwd");
}
int mvd_label_mask(const float *labels, float *mask, long n, float value, void *stream) {
    MVD_REQUIRE(labels && mask && n > 0, "label_mask: bad arguments");
    hipLaunchKernelGGL(k_label_mask, dim3(grid_for(n, 4096)), dim3(256), 0, as_stream(stream), labels, mask, n, value);
    return check_launch("label_mask");
}

size_t mvd_mse_workspace_bytes(long n) { return (size_t)grid_for(n, 4096) * sizeof(double) + 256; }
int mvd_mse_fwd(const float *a, const float *b, float *out, long n, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(a && b && out && ws && n > 0, "mse_fwd: bad arguments");
    MVD_REQUIRE(((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0, "mse_fwd: inputs must be 16-byte aligned");
    MVD_REQUIRE(ws_bytes >= mvd_mse_workspace_bytes(n), "mse_fwd: workspace too small");
    const long bx = grid_for(n, 4096);
    double *partial = reinterpret_cast<double *>(ws);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_mse_fwd, dim3(bx), dim3(256), 0, s, a, b, partial, n);
    if (check_launch("mse_fwd")) return 1;
    hipLaunchKernelGGL(k_mse_finish, dim3(1), dim3(64), 0, s, partial, out, (int)bx, n);
    return check_launch("mse_finish");
}
int mvd_mse_bwd(const float *a, const float *b, const float *g_dev, float *ga, float *gb, long n, void *stream) {
    MVD_REQUIRE(a && b && g_dev && (ga || gb) && n > 0, "mse_bwd: bad arguments");
    hipLaunchKernelGGL(k_mse_bwd, dim3(grid_for(n, 2048)), dim3(256), 0, as_stream(stream), a, b, g_dev, ga, gb, n);
    return check_launch("mse_bwd");
}

size_t mvd_kl_workspace_bytes(int N, long V) { return (size_t)grid_for((long)N * V, 2048) * sizeof(double) + 256; }

int mvd_kl_fwd(const float *ys, const float *yt, float *out, int N, int C, long V, long sn, long sc, long sv, float T,
               float eps_s, int pad, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(ys && yt && out && ws, "kl_fwd: null pointer");
    MVD_REQUIRE(N > 0 && V > 0 && C >= 1 && C <= KLCMAX && T > 0, "kl_fwd: bad shape (C<=32)");
    MVD_REQUIRE(!pad || C == 1, "kl_fwd: zero-channel padding is the C==1 branch");
    MVD_REQUIRE(ws_bytes >= mvd_kl_workspace_bytes(N, V), "kl_fwd: workspace too small");
    long bx = grid_for((long)N * V, 2048);
    double *partial = reinterpret_cast<double *>(ws);
    KlIdx ix{sn, sc, sv};
    hipStream_t s = as_stream(stream);
    if (kl_rows_ok(C, V, sn, sc, sv, pad) && (((uintptr_t)ys | (uintptr_t)yt) & 15) == 0)
        hipLaunchKernelGGL((k_kl_rows<false, false>), dim3(bx), dim3(256), 0, s, ys, yt, partial, nullptr, nullptr, nullptr,
                           1.f, N, C, V, T, eps_s);
    else
        hipLaunchKernelGGL(k_kl<false>, dim3(bx), dim3(256), 0, s, ys, yt, partial, nullptr, nullptr, nullptr, 1.f, N, C, V,
                           ix, T, eps_s, pad);
    if (check_launch("kl_fwd")) return 1;
    return reduce_partials(partial, out, (int)bx, 1, s);
}

int mvd_kl_bwd(const float *ys, const float *yt, const float *gscale_dev, float gscale_host, float *gs, float *gt, int N,
               int C, long V, long sn, long sc, long sv, float T, float eps_s, int pad, void *stream) {
    MVD_REQUIRE(ys && yt && (gs || gt), "kl_bwd: null pointer");
    MVD_REQUIRE(N > 0 && V > 0 && C >= 1 && C <= KLCMAX && T > 0, "kl_bwd: bad shape (C<=32)");
    MVD_REQUIRE(!pad || C == 1, "kl_bwd: zero-channel padding is the C==1 branch");
    long bx = grid_for((long)N * V, 4096);
    KlIdx ix{sn, sc, sv};
    if (kl_rows_ok(C, V, sn, sc, sv, pad) && (((uintptr_t)ys | (uintptr_t)yt | (uintptr_t)gs | (uintptr_t)gt) & 15) == 0)
        hipLaunchKernelGGL((k_kl_rows<true, false>), dim3(grid_for((long)N * V * (C / 4), 8192)), dim3(256), 0,
                           as_stream(stream), ys, yt, nullptr, gs, gt, gscale_dev, gscale_host, N, C, V, T, eps_s);
    else
        hipLaunchKernelGGL(k_kl<true>, dim3(bx), dim3(256), 0, as_stream(stream), ys, yt, nullptr, gs, gt, gscale_dev,
                           gscale_host, N, C, V, ix, T, eps_s, pad);
    return check_launch("kl_bwd");
}

int mvd_kl_fwd_bf16(const uint16_t *ys, const uint16_t *yt, float *out, int N, int C, long V, float T, float eps_s,
                    void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(ys && yt && out && ws, "kl_fwd_bf16: null pointer");
    MVD_REQUIRE(N > 0 && V > 0 && T > 0 && kl_rows_ok(C, V, V * (long)C, 1, C, 0),
                "kl_fwd_bf16: dense NDHWC rows with C in {4,8,16,32} only");
    MVD_REQUIRE(ws_bytes >= mvd_kl_workspace_bytes(N, V), "kl_fwd_bf16: workspace too small");
    long bx = grid_for((long)N * V, 2048);
    double *partial = reinterpret_cast<double *>(ws);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL((k_kl_rows<false, true>), dim3(bx), dim3(256), 0, s, ys, yt, partial, nullptr, nullptr, nullptr, 1.f,
                       N, C, V, T, eps_s);
    if (check_launch("kl_fwd_bf16")) return 1;
    return reduce_partials(partial, out, (int)bx, 1, s);
}

int mvd_kl_bwd_bf16(const uint16_t *ys, const uint16_t *yt, const float *gscale_dev, float gscale_host, uint16_t *gs,
                    uint16_t *gt, int N, int C, long V, float T, float eps_s, void *stream) {
    MVD_REQUIRE(ys && yt && (gs || gt), "kl_bwd_bf16: null pointer");
    MVD_REQUIRE(N > 0 && V > 0 && T > 0 && kl_rows_ok(C, V, V * (long)C, 1, C, 0),
                "kl_bwd_bf16: dense NDHWC rows with C in {4,8,16,32} only");
    hipLaunchKernelGGL((k_kl_rows<true, true>), dim3(grid_for((long)N * V * (C / 4), 8192)), dim3(256), 0, as_stream(stream),
                       ys, yt, nullptr, gs, gt, gscale_dev, gscale_host, N, C, V, T, eps_s);
    return check_launch("kl_bwd_bf16");
}
}
