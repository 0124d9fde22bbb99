// Fused InstanceNorm3d(affine) + LeakyReLU on NDHWC fp32 activations  (SURVEY K3/K4).
// HBM-bound: fwd = read x (stats) + read x + write y; bwd = read x,dy (sums) + read x,dy + write dx.
// Statistics: per-thread fp64 accumulation -> fixed-order LDS reduce -> per-block partials -> fixed-order finalize.
#include "common.h"

namespace mvd {

struct NormGeom {
    int C, CG, R, threads, nblk;
    long V, chunk;
};

static NormGeom norm_geom(int N, long V, int C) {
    NormGeom g;
    g.C = C;
    g.V = V;
    int vec = (C % 4 == 0) ? 4 : 1;
    g.CG = C / vec;
    if (g.CG > 256) {  // very wide, not on the path; fall back to scalar columns of 256
        g.CG = C;
    }
    g.R = 256 / g.CG;
    if (g.R < 1) g.R = 1;
    g.threads = ((g.R * g.CG + 63) / 64) * 64;
    if (g.threads > 1024) g.threads = 1024;
    long want = 2048 / (N > 0 ? N : 1);
    if (want < 1) want = 1;
    long rows_min = (long)g.R * 16;  // at least 16 voxels per thread
    long nblk = V / rows_min;
    if (nblk < 1) nblk = 1;
    if (nblk > want) nblk = want;
    g.nblk = (int)nblk;
    g.chunk = cdiv(V, nblk);
    return g;
}

// ---- pass 1 (fwd): partial[n][b][c][2] = (sum x, sum x^2) in fp64
template <int VEC, bool XB>
__global__ void k_in_stats(const float *__restrict__ x, double *__restrict__ partial, int C, int CG, int R, long V,
                           long chunk) {
    extern __shared__ double sm[];  // [R][C][2]
    const int n = blockIdx.y, b = blockIdx.x, nblk = gridDim.x;
    const int t = threadIdx.x;
    const int g = t % CG, r = t / CG;
    const long v0 = (long)b * chunk;
    long v1 = v0 + chunk;
    if (v1 > V) v1 = V;
    double s[VEC], ss[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) s[i] = ss[i] = 0.0;
    if (r < R) {
        const float *xp = x + ((size_t)n * V) * C + (size_t)g * VEC;
        long v = v0 + r;
        if (VEC == 4) {
            for (; v + 3L * R < v1; v += 4L * R) {  // four rows in flight per thread, accumulated in row order
                float4 q4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) q4[u] = ld4<XB>(x, ((size_t)n * V + v + (long)u * R) * C + (size_t)g * 4);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float f[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        s[i] += (double)f[i];
                        ss[i] += (double)f[i] * (double)f[i];
                    }
                }
            }
        }
        for (; v < v1; v += R) {
            if (VEC == 4) {
                float4 q = ld4<XB>(x, ((size_t)n * V + v) * C + (size_t)g * 4);
                float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    s[i] += (double)f[i];
                    ss[i] += (double)f[i] * (double)f[i];
                }
            } else {
                float f = xp[(size_t)v * C];
                s[0] += (double)f;
                ss[0] += (double)f * (double)f;
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            sm[((size_t)r * C + g * VEC + i) * 2 + 0] = s[i];
            sm[((size_t)r * C + g * VEC + i) * 2 + 1] = ss[i];
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) {
        double a = 0, q = 0;
        for (int rr = 0; rr < R; rr++) {
            a += sm[((size_t)rr * C + c) * 2 + 0];
            q += sm[((size_t)rr * C + c) * 2 + 1];
        }
        size_t o = (((size_t)n * nblk + b) * C + c) * 2;
        partial[o] = a;
        partial[o + 1] = q;
    }
}

// scale / shift (optional): the fused form scale = gamma * rstd, shift = beta - mean * scale of the bf16 path (the expression
// of k_in_scale_shift / k_in_finalize_tiles on the float-rounded mean and rstd)
__global__ void k_in_finalize(const double *__restrict__ partial, float *__restrict__ mean, float *__restrict__ rstd,
                              int C, int nblk, long V, float eps, const float *__restrict__ gamma = nullptr,
                              const float *__restrict__ beta = nullptr, float *__restrict__ scale = nullptr,
                              float *__restrict__ shift = nullptr) {
    // one wave per (n, c): lanes stride over the block partials, fixed shuffle tree (deterministic)
    const int n = blockIdx.y, c = blockIdx.x;
    double a = 0, q = 0, a1 = 0, q1 = 0;
    int b = threadIdx.x;
    for (; b + 64 < nblk; b += 128) {  // two block partials (four loads) in flight per lane, fixed order
        const size_t o = (((size_t)n * nblk + b) * C + c) * 2, o1 = (((size_t)n * nblk + b + 64) * C + c) * 2;
        const double va = partial[o], vq = partial[o + 1], wa = partial[o1], wq = partial[o1 + 1];
        a += va; q += vq; a1 += wa; q1 += wq;
    }
    for (; b < nblk; b += 64) {
        size_t o = (((size_t)n * nblk + b) * C + c) * 2;
        a += partial[o];
        q += partial[o + 1];
    }
    a = wave_sum(a + a1);
    q = wave_sum(q + q1);
    if (threadIdx.x != 0) return;
    double m = a / (double)V;
    double var = q / (double)V - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    mean[(size_t)n * C + c] = mf;
    rstd[(size_t)n * C + c] = rf;
    if (scale) {
        const float sc = gamma[c] * rf;
        scale[(size_t)n * C + c] = sc;
        shift[(size_t)n * C + c] = fmaf(-mf, sc, beta[c]);
    }
}

// conv-epilogue statistics: per-tile partials [n][tile][C][2] (fp32 sums of <= 128 values each) -> per-block fp64
// partials [n][b][C][2] in the layout k_in_finalize reads.  Thread = (channel, tile row): consecutive threads read
// consecutive (sum, sum of squares) pairs of one tile (coalesced); rows are combined through LDS in a fixed order.
__global__ void k_in_tiles_reduce(const float *__restrict__ tile, double *__restrict__ partial, int C, int CW, int R,
                                  long ntiles, long chunk) {
    extern __shared__ double sm[];  // [R][CW][2]
    const int n = blockIdx.y, b = blockIdx.x, nblk = gridDim.x;
    const int tc = threadIdx.x % CW, r = threadIdx.x / CW;
    const long t0 = (long)b * chunk;
    long t1 = t0 + chunk;
    if (t1 > ntiles) t1 = ntiles;
    for (int c0 = 0; c0 < C; c0 += CW) {
        const int c = c0 + tc;
        double a = 0, q = 0;
        if (r < R && c < C) {
            const float *tp = tile + ((size_t)n * ntiles * C + c) * 2;
            long tt = t0 + r;
            for (; tt + 3L * R < t1; tt += 4L * R) {  // four tiles in flight, added in tile order
                const float2 v0 = *reinterpret_cast<const float2 *>(tp + (size_t)tt * C * 2);
                const float2 v1 = *reinterpret_cast<const float2 *>(tp + (size_t)(tt + R) * C * 2);
                const float2 v2 = *reinterpret_cast<const float2 *>(tp + (size_t)(tt + 2L * R) * C * 2);
                const float2 v3 = *reinterpret_cast<const float2 *>(tp + (size_t)(tt + 3L * R) * C * 2);
                a += (double)v0.x; q += (double)v0.y;
                a += (double)v1.x; q += (double)v1.y;
                a += (double)v2.x; q += (double)v2.y;
                a += (double)v3.x; q += (double)v3.y;
            }
            for (; tt < t1; tt += R) {
                const float2 v = *reinterpret_cast<const float2 *>(tp + (size_t)tt * C * 2);
                a += (double)v.x;
                q += (double)v.y;
            }
        }
        __syncthreads();
        if (r < R) {
            sm[((size_t)r * CW + tc) * 2 + 0] = a;
            sm[((size_t)r * CW + tc) * 2 + 1] = q;
        }
        __syncthreads();
        if (r == 0 && c < C) {
            double sa = 0, sq = 0;
            for (int rr = 0; rr < R; rr++) {
                sa += sm[((size_t)rr * CW + tc) * 2 + 0];
                sq += sm[((size_t)rr * CW + tc) * 2 + 1];
            }
            const size_t o = (((size_t)n * nblk + b) * C + c) * 2;
            partial[o] = sa;
            partial[o + 1] = sq;
        }
    }
}

template <int VEC>
__global__ void k_in_apply(const float *__restrict__ x, const float *__restrict__ gamma,
                           const float *__restrict__ beta, const float *__restrict__ mean,
                           const float *__restrict__ rstd, float *__restrict__ y, int C, long V, float slope) {
    const int n = blockIdx.y;
    const long per_n = V * C / VEC;
    const float *xn = x + (size_t)n * V * C;
    float *yn = y + (size_t)n * V * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)((i * VEC) % C);
        if (VEC == 4) {
            float4 q = reinterpret_cast<const float4 *>(xn)[i];
            float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float xh = (f[k] - mean[(size_t)n * C + c + k]) * rstd[(size_t)n * C + c + k];
                float z = fmaf(xh, gamma[c + k], beta[c + k]);
                f[k] = z > 0.f ? z : z * slope;
            }
            reinterpret_cast<float4 *>(yn)[i] = make_float4(f[0], f[1], f[2], f[3]);
        } else {
            float xh = (xn[i] - mean[(size_t)n * C + c]) * rstd[(size_t)n * C + c];
            float z = fmaf(xh, gamma[c], beta[c]);
            yn[i] = z > 0.f ? z : z * slope;
        }
    }
}

// row-mapped apply (C % 4 == 0): thread = (4-channel group g, row r); the per-channel parameters live in registers,
// each thread streams float4 rows v = v0 + r, v0 + r + R, ... of its block's voxel chunk (no per-element index math)
template <bool XB, bool YB>
__global__ void k_in_apply_rows(const float *__restrict__ x, const float *__restrict__ gamma,
                                const float *__restrict__ beta, const float *__restrict__ mean,
                                const float *__restrict__ rstd, float *__restrict__ y, int C, int CG, int R, long V,
                                long chunk, float slope) {
    const int n = blockIdx.y, t = threadIdx.x;
    const int g = t % CG, r = t / CG;
    if (r >= R) return;
    const long v0 = (long)blockIdx.x * chunk;
    long v1 = v0 + chunk;
    if (v1 > V) v1 = V;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = g * 4 + i;
        mu[i] = mean[(size_t)n * C + c];
        rs[i] = rstd[(size_t)n * C + c];
        ga[i] = gamma[c];
        be[i] = beta[c];
    }
    const size_t base = ((size_t)n * V) * C + (size_t)g * 4;
    long v = v0 + r;
    for (; v + 3L * R < v1; v += 4L * R) {  // four independent rows in flight
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = ld4<XB>(x, base + (size_t)(v + (long)u * R) * C);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float f[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float xh = (f[i] - mu[i]) * rs[i];
                const float z = fmaf(xh, ga[i], be[i]);  // (explicit: the same rounding in every kernel that re-computes z)
                f[i] = z > 0.f ? z : z * slope;
            }
            st4<YB>(y, base + (size_t)(v + (long)u * R) * C, f[0], f[1], f[2], f[3]);
        }
    }
    for (; v < v1; v += R) {
        float4 q = ld4<XB>(x, base + (size_t)v * C);
        float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float xh = (f[i] - mu[i]) * rs[i];
            const float z = fmaf(xh, ga[i], be[i]);  // (explicit: the same rounding in every kernel that re-computes z)
            f[i] = z > 0.f ? z : z * slope;
        }
        st4<YB>(y, base + (size_t)v * C, f[0], f[1], f[2], f[3]);
    }
}

#ifndef MVD_IN_NRB
#define MVD_IN_NRB 4
#endif
constexpr int NRB = MVD_IN_NRB;  // rows in flight per thread in the backward apply pass
template <bool XB, bool YB>
__global__ void k_in_bwd_apply_rows(const float *__restrict__ x, const float *__restrict__ dy,
                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                    const float *__restrict__ mean, const float *__restrict__ rstd,
                                    const float *__restrict__ sums, float *__restrict__ dx, int C, int CG, int R, long V,
                                    long chunk, float slope) {
    const int n = blockIdx.y, t = threadIdx.x;
    const int g = t % CG, r = t / CG;
    if (r >= R) return;
    const long v0 = (long)blockIdx.x * chunk;
    long v1 = v0 + chunk;
    if (v1 > V) v1 = V;
    const float invV = 1.0f / (float)V;
    float mu[4], rs[4], ga[4], be[4], m1[4], m2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = g * 4 + i;
        const size_t nc = (size_t)n * C + c;
        mu[i] = mean[nc];
        rs[i] = rstd[nc];
        ga[i] = gamma[c];
        be[i] = beta[c];
        m1[i] = sums[nc * 2 + 0] * invV;
        m2[i] = sums[nc * 2 + 1] * invV;
    }
    const size_t base = ((size_t)n * V) * C + (size_t)g * 4;
    long v = v0 + r;
    for (; v + (NRB - 1L) * R < v1; v += (long)NRB * R) {  // NRB rows (2 NRB loads) in flight
        float4 q[NRB], e[NRB];
#pragma unroll
        for (int u = 0; u < NRB; u++) {
            q[u] = ld4<XB>(x, base + (size_t)(v + (long)u * R) * C);
            e[u] = ld4<YB>(dy, base + (size_t)(v + (long)u * R) * C);
        }
#pragma unroll
        for (int u = 0; u < NRB; u++) {
            float f[4] = {q[u].x, q[u].y, q[u].z, q[u].w}, d[4] = {e[u].x, e[u].y, e[u].z, e[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float xh = (f[i] - mu[i]) * rs[i];
                const float z = fmaf(xh, ga[i], be[i]);  // (explicit: the same rounding in every kernel that re-computes z)
                const float dz = z > 0.f ? d[i] : d[i] * slope;
                f[i] = ga[i] * rs[i] * (dz - m1[i] - xh * m2[i]);
            }
            st4<XB>(dx, base + (size_t)(v + (long)u * R) * C, f[0], f[1], f[2], f[3]);
        }
    }
    for (; v < v1; v += R) {
        float4 q = ld4<XB>(x, base + (size_t)v * C);
        float4 e = ld4<YB>(dy, base + (size_t)v * C);
        float f[4] = {q.x, q.y, q.z, q.w}, d[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float xh = (f[i] - mu[i]) * rs[i];
            const float z = fmaf(xh, ga[i], be[i]);  // (explicit: the same rounding in every kernel that re-computes z)
            const float dz = z > 0.f ? d[i] : d[i] * slope;
            f[i] = ga[i] * rs[i] * (dz - m1[i] - xh * m2[i]);
        }
        st4<XB>(dx, base + (size_t)v * C, f[0], f[1], f[2], f[3]);
    }
}

// ---- bwd pass 1: partial[n][b][c][2] = (sum dz, sum dz*xhat)
template <int VEC, bool XB, bool YB>
__global__ void k_in_bwd_stats(const float *__restrict__ x, const float *__restrict__ dy,
                               const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ mean, const float *__restrict__ rstd,
                               double *__restrict__ partial, int C, int CG, int R, long V, long chunk, float slope) {
    extern __shared__ double sm[];
    const int n = blockIdx.y, b = blockIdx.x, nblk = gridDim.x;
    const int t = threadIdx.x;
    const int g = t % CG, r = t / CG;
    const long v0 = (long)b * chunk;
    long v1 = v0 + chunk;
    if (v1 > V) v1 = V;
    double s[VEC], ss[VEC];
    float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) s[i] = ss[i] = 0.0;
    if (r < R) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            int c = g * VEC + i;
            mu[i] = mean[(size_t)n * C + c];
            rs[i] = rstd[(size_t)n * C + c];
            ga[i] = gamma[c];
            be[i] = beta[c];
        }
        const size_t base = ((size_t)n * V) * C + (size_t)g * VEC;
        long v = v0 + r;
        if (VEC == 4) {
            for (; v + R < v1; v += 2L * R) {  // two rows (four loads) in flight per thread, accumulated in row order
                const float4 q0 = ld4<XB>(x, base + (size_t)v * C), e0 = ld4<YB>(dy, base + (size_t)v * C);
                const float4 q1 = ld4<XB>(x, base + (size_t)(v + R) * C), e1 = ld4<YB>(dy, base + (size_t)(v + R) * C);
                const float fx[2][4] = {{q0.x, q0.y, q0.z, q0.w}, {q1.x, q1.y, q1.z, q1.w}};
                const float dd[2][4] = {{e0.x, e0.y, e0.z, e0.w}, {e1.x, e1.y, e1.z, e1.w}};
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float xh = (fx[u][i] - mu[i]) * rs[i];
                        const float z = fmaf(xh, ga[i], be[i]);  // (explicit: the same rounding in every kernel that re-computes z)
                        const float dz = z > 0.f ? dd[u][i] : dd[u][i] * slope;
                        s[i] += (double)dz;
                        ss[i] += (double)dz * (double)xh;
                    }
            }
        }
        for (; v < v1; v += R) {
            float f[VEC], d[VEC];
            if (VEC == 4) {
                float4 q = ld4<XB>(x, base + (size_t)v * C);
                float4 e = ld4<YB>(dy, base + (size_t)v * C);
                f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
                d[0] = e.x; d[1] = e.y; d[2] = e.z; d[3] = e.w;
            } else {
                f[0] = x[base + (size_t)v * C];
                d[0] = dy[base + (size_t)v * C];
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) {
                float xh = (f[i] - mu[i]) * rs[i];
                float z = fmaf(xh, ga[i], be[i]);
                float dz = z > 0.f ? d[i] : d[i] * slope;
                s[i] += (double)dz;
                ss[i] += (double)dz * (double)xh;
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            sm[((size_t)r * C + g * VEC + i) * 2 + 0] = s[i];
            sm[((size_t)r * C + g * VEC + i) * 2 + 1] = ss[i];
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) {
        double a = 0, q = 0;
        for (int rr = 0; rr < R; rr++) {
            a += sm[((size_t)rr * C + c) * 2 + 0];
            q += sm[((size_t)rr * C + c) * 2 + 1];
        }
        size_t o = (((size_t)n * nblk + b) * C + c) * 2;
        partial[o] = a;
        partial[o + 1] = q;
    }
}

// sums[n][c][2] (float) = (sum dz, sum dz*xhat); dgamma/dbeta over n
__global__ void k_in_bwd_finalize(const double *__restrict__ partial, float *__restrict__ sums,
                                  float *__restrict__ dgamma, float *__restrict__ dbeta, int N, int C, int nblk) {
    // one wave per channel: lanes stride over the block partials of each sample, fixed shuffle tree
    const int c = blockIdx.x;
    double tg = 0, tb = 0;
    for (int n = 0; n < N; n++) {
        double a = 0, q = 0, a1 = 0, q1 = 0;
        int b = threadIdx.x;
        for (; b + 64 < nblk; b += 128) {  // (as k_in_finalize)
            const size_t o = (((size_t)n * nblk + b) * C + c) * 2, o1 = (((size_t)n * nblk + b + 64) * C + c) * 2;
            const double va = partial[o], vq = partial[o + 1], wa = partial[o1], wq = partial[o1 + 1];
            a += va; q += vq; a1 += wa; q1 += wq;
        }
        for (; b < nblk; b += 64) {
            size_t o = (((size_t)n * nblk + b) * C + c) * 2;
            a += partial[o];
            q += partial[o + 1];
        }
        a = wave_sum(a + a1);
        q = wave_sum(q + q1);
        if (threadIdx.x == 0) {
            sums[((size_t)n * C + c) * 2 + 0] = (float)a;
            sums[((size_t)n * C + c) * 2 + 1] = (float)q;
        }
        tb += a;
        tg += q;
    }
    if (threadIdx.x == 0) {
        dgamma[c] = (float)tg;
        dbeta[c] = (float)tb;
    }
}

// conv-epilogue statistics of the bf16 z-marching conv (k_fwd16z<.., FUSE & 1>): per-workgroup partials [n][tile][C][2]
// (fp32) -> mean, rstd and the fused form scale = gamma * rstd, shift = beta - mean * scale, ONE launch: a wave per
// (n, c), lanes stride over the tiles, fp64, fixed shuffle tree (deterministic).
__global__ void k_in_finalize_tiles(const float *__restrict__ tile, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, float *__restrict__ mean, float *__restrict__ rstd,
                                    float *__restrict__ scale, float *__restrict__ shift, int C, long ntiles, long V,
                                    float eps) {
    const int n = blockIdx.y, c = blockIdx.x;
    double a = 0, q = 0;
    const float2 *tp = reinterpret_cast<const float2 *>(tile) + (size_t)n * ntiles * C + c;
    long t = threadIdx.x;
    for (; t + 192 < ntiles; t += 256) {  // four partials in flight per lane, added in tile order
        const float2 v0 = tp[(size_t)t * C], v1 = tp[(size_t)(t + 64) * C], v2 = tp[(size_t)(t + 128) * C],
                     v3 = tp[(size_t)(t + 192) * C];
        a += (double)v0.x; q += (double)v0.y;
        a += (double)v1.x; q += (double)v1.y;
        a += (double)v2.x; q += (double)v2.y;
        a += (double)v3.x; q += (double)v3.y;
    }
    for (; t < ntiles; t += 64) {
        const float2 v = tp[(size_t)t * C];
        a += (double)v.x;
        q += (double)v.y;
    }
    a = wave_sum(a);
    q = wave_sum(q);
    if (threadIdx.x != 0) return;
    const double m = a / (double)V;
    double var = q / (double)V - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
    const size_t nc = (size_t)n * C + c;
    mean[nc] = mf;
    rstd[nc] = rs;
    const float sc = gamma[c] * rs;
    scale[nc] = sc;
    shift[nc] = fmaf(-mf, sc, beta[c]);
}

// y = bf16(lrelu(fma(x, scale[n][c], shift[n][c]))): the apply pass in the form the fused conv prologue uses (k_fwd16z<.., FUSE &
// 2>), so that a tensor normalised here and one normalised in a consumer's loader agree bit for bit.  bf16 in, bf16 out.
__global__ void k_in_apply_ss16(const unsigned short *__restrict__ x, const float *__restrict__ scale,
                                const float *__restrict__ shift, unsigned short *__restrict__ y, int C, int CG, int R, long V,
                                long chunk, float slope) {
    const int n = blockIdx.y, t = threadIdx.x;
    const int g = t % CG, r = t / CG;
    if (r >= R) return;
    const long v0 = (long)blockIdx.x * chunk;
    long v1 = v0 + chunk;
    if (v1 > V) v1 = V;
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sc[i] = scale[(size_t)n * C + g * 4 + i];
        sh[i] = shift[(size_t)n * C + g * 4 + i];
    }
    const float *xf = reinterpret_cast<const float *>(x);
    float *yf = reinterpret_cast<float *>(y);
    const size_t base = ((size_t)n * V) * C + (size_t)g * 4;
    long v = v0 + r;
    for (; v + 3L * R < v1; v += 4L * R) {  // four independent rows in flight
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = ld4<true>(xf, base + (size_t)(v + (long)u * R) * C);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float f[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float z = fmaf(f[i], sc[i], sh[i]);
                f[i] = fmaxf(z, z * slope);
            }
            st4<true>(yf, base + (size_t)(v + (long)u * R) * C, f[0], f[1], f[2], f[3]);
        }
    }
    for (; v < v1; v += R) {
        float4 q = ld4<true>(xf, base + (size_t)v * C);
        float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float z = fmaf(f[i], sc[i], sh[i]);
            f[i] = fmaxf(z, z * slope);
        }
        st4<true>(yf, base + (size_t)v * C, f[0], f[1], f[2], f[3]);
    }
}

template <int VEC>
__global__ void k_in_bwd_apply(const float *__restrict__ x, const float *__restrict__ dy,
                               const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ mean, const float *__restrict__ rstd,
                               const float *__restrict__ sums, float *__restrict__ dx, int C, long V, float slope) {
    const int n = blockIdx.y;
    const long per_n = V * C / VEC;
    const size_t off = (size_t)n * V * C;
    const float invV = 1.0f / (float)V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)((i * VEC) % C);
        float f[VEC], d[VEC];
        if (VEC == 4) {
            float4 q = reinterpret_cast<const float4 *>(x + off)[i];
            float4 e = reinterpret_cast<const float4 *>(dy + off)[i];
            f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
            d[0] = e.x; d[1] = e.y; d[2] = e.z; d[3] = e.w;
        } else {
            f[0] = x[off + i];
            d[0] = dy[off + i];
        }
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            size_t nc = (size_t)n * C + c + k;
            float rs = rstd[nc];
            float xh = (f[k] - mean[nc]) * rs;
            float z = fmaf(xh, gamma[c + k], beta[c + k]);
            float dz = z > 0.f ? d[k] : d[k] * slope;
            float m1 = sums[nc * 2 + 0] * invV, m2 = sums[nc * 2 + 1] * invV;
            f[k] = gamma[c + k] * rs * (dz - m1 - xh * m2);
        }
        if (VEC == 4)
            reinterpret_cast<float4 *>(dx + off)[i] = make_float4(f[0], f[1], f[2], f[3]);
        else
            dx[off + i] = f[0];
    }
}

}  // namespace mvd

using namespace mvd;

extern "C" {

int mvd_instnorm_nblk(int N, long V, int C) { return norm_geom(N, V, C).nblk; }

size_t mvd_instnorm_workspace_bytes(int N, long V, int C) {
    NormGeom g = norm_geom(N, V, C);
    // partials (doubles) + sums [N][C][2] floats
    return (size_t)N * g.nblk * C * 2 * sizeof(double) + (size_t)N * C * 2 * sizeof(float) + 256;
}

static int in_fwd(const float *x, bool xb, const float *gamma, const float *beta, float *y, bool yb, float *mean,
                  float *rstd, int N, long V, int C, float eps, float slope, void *ws, size_t ws_bytes, void *stream,
                  const float *tile_stats = nullptr, long ntiles = 0) {
    MVD_REQUIRE(x && gamma && beta && y && mean && rstd && ws, "instnorm_fwd: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && C <= 1024, "instnorm_fwd: bad shape N=%d V=%ld C=%d", N, V, C);
    MVD_REQUIRE(ws_bytes >= mvd_instnorm_workspace_bytes(N, V, C), "instnorm_fwd: workspace too small");
    NormGeom g = norm_geom(N, V, C);
    hipStream_t s = as_stream(stream);
    double *partial = reinterpret_cast<double *>(ws);
    size_t sm = (size_t)g.R * C * 2 * sizeof(double);
    MVD_REQUIRE(sm <= 64 * 1024, "instnorm_fwd: C too large for the LDS reduce");
    const bool v4 = (C % 4 == 0) && (g.CG * 4 == C);
    MVD_REQUIRE(v4 || !(xb || yb), "instnorm_fwd: bf16 I/O needs C %% 4 == 0");
    dim3 grid(g.nblk, N);
    if (tile_stats) {
        // statistics came out of the producing conv's epilogue: no pass over x
        long nb2 = (ntiles + 63) / 64;
        if (nb2 > g.nblk) nb2 = g.nblk;  // the workspace holds N * g.nblk * C * 2 doubles
        if (nb2 > 64) nb2 = 64;
        const long chunk2 = (ntiles + nb2 - 1) / nb2;
        nb2 = (ntiles + chunk2 - 1) / chunk2;
        const int CW = C < 256 ? C : 256, R2 = 256 / CW;
        hipLaunchKernelGGL(k_in_tiles_reduce, dim3((unsigned)nb2, N), dim3(256), (size_t)R2 * CW * 2 * sizeof(double), s,
                           tile_stats, partial, C, CW, R2, ntiles, chunk2);
        if (check_launch("instnorm statistics from conv tiles")) return 1;
        hipLaunchKernelGGL(k_in_finalize, dim3(C, N), dim3(64), 0, s, partial, mean, rstd, C, (int)nb2, V, eps);
        if (check_launch("instnorm finalize")) return 1;
    } else {
        if (v4 && xb)
            hipLaunchKernelGGL((k_in_stats<4, true>), grid, dim3(g.threads), sm, s, x, partial, C, g.CG, g.R, V, g.chunk);
        else if (v4)
            hipLaunchKernelGGL((k_in_stats<4, false>), grid, dim3(g.threads), sm, s, x, partial, C, g.CG, g.R, V, g.chunk);
        else
            hipLaunchKernelGGL((k_in_stats<1, false>), grid, dim3(g.threads), sm, s, x, partial, C, g.CG, g.R, V, g.chunk);
        if (check_launch("instnorm stats")) return 1;
        // bf16 in and out: the apply pass takes the scale / shift form -- ONE arithmetic for the stand-alone pass, the apply after
        // a conv's statistics epilogue and the loader prologues (conv, weight gradient, seg head), so fused and un-fused
        // networks agree bit for bit whichever kernel produced the statistics
        float *ss = reinterpret_cast<float *>(partial + (size_t)N * g.nblk * C * 2);
        const bool ssf = v4 && xb && yb;
        hipLaunchKernelGGL(k_in_finalize, dim3(C, N), dim3(64), 0, s, partial, mean, rstd, C, g.nblk, V, eps,
                           ssf ? gamma : nullptr, ssf ? beta : nullptr, ssf ? ss : nullptr, ssf ? ss + (size_t)N * C : nullptr);
        if (check_launch("instnorm finalize")) return 1;
        if (ssf) {
            long nb2 = V / ((long)g.R * 8);
            if (nb2 < 1) nb2 = 1;
            long cap2 = 8192 / N > 0 ? 8192 / N : 1;
            if (nb2 > cap2) nb2 = cap2;
            const long chunk2 = cdiv(V, nb2);
            hipLaunchKernelGGL(k_in_apply_ss16, dim3((unsigned)cdiv(V, chunk2), N), dim3(g.threads), 0, s,
                               reinterpret_cast<const unsigned short *>(x), ss, ss + (size_t)N * C,
                               reinterpret_cast<unsigned short *>(y), C, g.CG, g.R, V, chunk2, slope);
            return check_launch("instnorm apply (scale / shift form)");
        }
    }
    long per_n = V * C / (v4 ? 4 : 1);
    long bx = cdiv(per_n, 256);
    long cap = 4096 / N > 0 ? 4096 / N : 1;
    if (bx > cap) bx = cap;
    if (v4) {
        // apply pass: more, smaller voxel chunks than the statistics pass (pure streaming, no reduction)
        long nb2 = V / ((long)g.R * 8);
        if (nb2 < 1) nb2 = 1;
        long cap2 = 8192 / N > 0 ? 8192 / N : 1;
        if (nb2 > cap2) nb2 = cap2;
        const long chunk2 = cdiv(V, nb2);
        auto kern = xb ? (yb ? k_in_apply_rows<true, true> : k_in_apply_rows<true, false>)
                       : (yb ? k_in_apply_rows<false, true> : k_in_apply_rows<false, false>);
        hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(V, chunk2), N), dim3(g.threads), 0, s, x, gamma, beta, mean, rstd, y, C,
                           g.CG, g.R, V, chunk2, slope);
    }
    else
        hipLaunchKernelGGL(k_in_apply<1>, dim3(bx, N), dim3(256), 0, s, x, gamma, beta, mean, rstd, y, C, V, slope);
    return check_launch("instnorm apply");
}

static int in_bwd(const float *x, bool xb, const float *dy, bool yb, const float *gamma, const float *beta,
                  const float *mean, const float *rstd, float *dx, float *dgamma, float *dbeta, int N, long V, int C,
                  float slope, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && dy && gamma && beta && mean && rstd && dx && dgamma && dbeta && ws, "instnorm_bwd: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && C <= 1024, "instnorm_bwd: bad shape");
    MVD_REQUIRE(ws_bytes >= mvd_instnorm_workspace_bytes(N, V, C), "instnorm_bwd: workspace too small");
    NormGeom g = norm_geom(N, V, C);
    hipStream_t s = as_stream(stream);
    double *partial = reinterpret_cast<double *>(ws);
    float *sums = reinterpret_cast<float *>(partial + (size_t)N * g.nblk * C * 2);
    size_t sm = (size_t)g.R * C * 2 * sizeof(double);
    MVD_REQUIRE(sm <= 64 * 1024, "instnorm_bwd: C too large for the LDS reduce");
    const bool v4 = (C % 4 == 0) && (g.CG * 4 == C);
    dim3 grid(g.nblk, N);
    MVD_REQUIRE(v4 || !(xb || yb), "instnorm_bwd: bf16 I/O needs C %% 4 == 0");
    if (v4) {
        auto kern = xb ? (yb ? k_in_bwd_stats<4, true, true> : k_in_bwd_stats<4, true, false>)
                       : (yb ? k_in_bwd_stats<4, false, true> : k_in_bwd_stats<4, false, false>);
        hipLaunchKernelGGL(kern, grid, dim3(g.threads), sm, s, x, dy, gamma, beta, mean, rstd, partial, C, g.CG, g.R, V,
                           g.chunk, slope);
    } else
        hipLaunchKernelGGL((k_in_bwd_stats<1, false, false>), grid, dim3(g.threads), sm, s, x, dy, gamma, beta, mean, rstd,
                           partial, C, g.CG, g.R, V, g.chunk, slope);
    if (check_launch("instnorm bwd stats")) return 1;
    hipLaunchKernelGGL(k_in_bwd_finalize, dim3(C), dim3(64), 0, s, partial, sums, dgamma, dbeta, N, C,
                       g.nblk);
    if (check_launch("instnorm bwd finalize")) return 1;
    long per_n = V * C / (v4 ? 4 : 1);
    long bx = cdiv(per_n, 256);
    long cap = 4096 / N > 0 ? 4096 / N : 1;
    if (bx > cap) bx = cap;
    if (v4) {
        long nb2 = V / ((long)g.R * 8);
        if (nb2 < 1) nb2 = 1;
        long cap2 = 8192 / N > 0 ? 8192 / N : 1;
        if (nb2 > cap2) nb2 = cap2;
        const long chunk2 = cdiv(V, nb2);
        auto kern = xb ? (yb ? k_in_bwd_apply_rows<true, true> : k_in_bwd_apply_rows<true, false>)
                       : (yb ? k_in_bwd_apply_rows<false, true> : k_in_bwd_apply_rows<false, false>);
        hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(V, chunk2), N), dim3(g.threads), 0, s, x, dy, gamma, beta, mean, rstd,
                           sums, dx, C, g.CG, g.R, V, chunk2, slope);
    }
    else
        hipLaunchKernelGGL(k_in_bwd_apply<1>, dim3(bx, N), dim3(256), 0, s, x, dy, gamma, beta, mean, rstd, sums, dx, C,
                           V, slope);
    return check_launch("instnorm bwd apply");
}

int mvd_instnorm_lrelu_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                           int N, long V, int C, float eps, float slope, void *ws, size_t ws_bytes, void *stream) {
    return in_fwd(x, false, gamma, beta, y, false, mean, rstd, N, V, C, eps, slope, ws, ws_bytes, stream);
}

int mvd_instnorm_lrelu_bwd(const float *x, const float *dy, const float *gamma, const float *beta, const float *mean,
                           const float *rstd, float *dx, float *dgamma, float *dbeta, int N, long V, int C,
                           float slope, void *ws, size_t ws_bytes, void *stream) {
    return in_bwd(x, false, dy, false, gamma, beta, mean, rstd, dx, dgamma, dbeta, N, V, C, slope, ws, ws_bytes, stream);
}

int mvd_instnorm_lrelu_fwd_prestats(const float *x, const float *tile_stats, long ntiles, const float *gamma,
                                    const float *beta, float *y, float *mean, float *rstd, int N, long V, int C, float eps,
                                    float slope, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(tile_stats && ntiles > 0, "instnorm_fwd_prestats: tile statistics required");
    return in_fwd(x, false, gamma, beta, y, false, mean, rstd, N, V, C, eps, slope, ws, ws_bytes, stream, tile_stats, ntiles);
}

int mvd_instnorm_finalize_tiles(const float *tile_stats, long ntiles, const float *gamma, const float *beta, float *mean,
                                float *rstd, float *scale, float *shift, int N, long V, int C, float eps, void *stream) {
    MVD_REQUIRE(tile_stats && gamma && beta && mean && rstd && scale && shift, "instnorm_finalize_tiles: null pointer");
    MVD_REQUIRE(ntiles > 0 && N > 0 && N <= 65535 && V > 0 && C > 0 && C <= 65535, "instnorm_finalize_tiles: bad shape");
    hipLaunchKernelGGL(k_in_finalize_tiles, dim3(C, N), dim3(64), 0, as_stream(stream), tile_stats, gamma, beta, mean, rstd,
                       scale, shift, C, ntiles, V, eps);
    return check_launch("instnorm finalize (conv tile statistics)");
}

// scale = gamma * rstd, shift = beta - mean * scale for [N][C] (the fused form of mean / rstd; same rounding as
// k_in_finalize_tiles)
__global__ void k_in_scale_shift(const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
                                 const float *__restrict__ rstd, float *__restrict__ scale, float *__restrict__ shift, int N,
                                 int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const float sc = gamma[i % C] * rstd[i];
    scale[i] = sc;
    shift[i] = fmaf(-mean[i], sc, beta[i % C]);
}

int mvd_instnorm_stats_bf16(const void *x, int x_is_bf16, const float *gamma, const float *beta, float *mean, float *rstd,
                            float *scale, float *shift, int N, long V, int C, float eps, void *ws, size_t ws_bytes,
                            void *stream) {
    MVD_REQUIRE(x && gamma && beta && mean && rstd && scale && shift && ws, "instnorm_stats: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && C % 4 == 0 && C <= 1024, "instnorm_stats: bad shape (C %% 4 == 0)");
    MVD_REQUIRE(ws_bytes >= mvd_instnorm_workspace_bytes(N, V, C), "instnorm_stats: workspace too small");
    NormGeom g = norm_geom(N, V, C);
    MVD_REQUIRE(g.CG * 4 == C, "instnorm_stats: C too wide");
    hipStream_t s = as_stream(stream);
    double *partial = reinterpret_cast<double *>(ws);
    const size_t sm = (size_t)g.R * C * 2 * sizeof(double);
    MVD_REQUIRE(sm <= 64 * 1024, "instnorm_stats: C too large for the LDS reduce");
    const float *xf = reinterpret_cast<const float *>(x);
    if (x_is_bf16)
        hipLaunchKernelGGL((k_in_stats<4, true>), dim3(g.nblk, N), dim3(g.threads), sm, s, xf, partial, C, g.CG, g.R, V, g.chunk);
    else
        hipLaunchKernelGGL((k_in_stats<4, false>), dim3(g.nblk, N), dim3(g.threads), sm, s, xf, partial, C, g.CG, g.R, V, g.chunk);
    if (check_launch("instnorm stats")) return 1;
    hipLaunchKernelGGL(k_in_finalize, dim3(C, N), dim3(64), 0, s, partial, mean, rstd, C, g.nblk, V, eps, gamma, beta, scale, shift);
    return check_launch("instnorm finalize");
}

// mean / rstd of an fp32 tensor from the fp32 conv's epilogue tiles (the statistics half of mvd_instnorm_lrelu_fwd_prestats:
// same launches, same order -- no apply pass; ops.NormActSegHeadFn in fp32)
int mvd_instnorm_stats_from_tiles(const float *tile_stats, long ntiles, float *mean, float *rstd, int N, long V, int C, float eps,
                                  void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(tile_stats && ntiles > 0 && mean && rstd && ws, "instnorm_stats_from_tiles: bad arguments");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && C <= 1024, "instnorm_stats_from_tiles: bad shape");
    MVD_REQUIRE(ws_bytes >= mvd_instnorm_workspace_bytes(N, V, C), "instnorm_stats_from_tiles: workspace too small");
    NormGeom g = norm_geom(N, V, C);
    hipStream_t s = as_stream(stream);
    double *partial = reinterpret_cast<double *>(ws);
    long nb2 = (ntiles + 63) / 64;
    if (nb2 > g.nblk) nb2 = g.nblk;
    if (nb2 > 64) nb2 = 64;
    const long chunk2 = (ntiles + nb2 - 1) / nb2;
    nb2 = (ntiles + chunk2 - 1) / chunk2;
    const int CW = C < 256 ? C : 256, R2 = 256 / CW;
    hipLaunchKernelGGL(k_in_tiles_reduce, dim3((unsigned)nb2, N), dim3(256), (size_t)R2 * CW * 2 * sizeof(double), s, tile_stats,
                       partial, C, CW, R2, ntiles, chunk2);
    if (check_launch("instnorm statistics from conv tiles")) return 1;
    hipLaunchKernelGGL(k_in_finalize, dim3(C, N), dim3(64), 0, s, partial, mean, rstd, C, (int)nb2, V, eps);
    return check_launch("instnorm finalize");
}

int mvd_instnorm_lrelu_apply_bf16(const uint16_t *x, const float *scale, const float *shift, uint16_t *y, int N, long V,
                                  int C, float slope, void *stream) {
    MVD_REQUIRE(x && scale && shift && y, "instnorm_apply_bf16: null pointer");
    MVD_REQUIRE(N > 0 && N <= 65535 && V > 0 && C > 0 && C % 4 == 0 && C <= 1024, "instnorm_apply_bf16: bad shape (C %% 4 == 0)");
    NormGeom g = norm_geom(N, V, C);
    MVD_REQUIRE(g.CG * 4 == C, "instnorm_apply_bf16: C too wide");
    long nb2 = V / ((long)g.R * 8);
    if (nb2 < 1) nb2 = 1;
    long cap2 = 8192 / N > 0 ? 8192 / N : 1;
    if (nb2 > cap2) nb2 = cap2;
    const long chunk2 = cdiv(V, nb2);
    hipLaunchKernelGGL(k_in_apply_ss16, dim3((unsigned)cdiv(V, chunk2), N), dim3(g.threads), 0, as_stream(stream), x, scale,
                       shift, y, C, g.CG, g.R, V, chunk2, slope);
    return check_launch("instnorm apply (scale / shift form)");
}

int mvd_instnorm_lrelu_fwd_bf16(const void *x, int x_is_bf16, const float *gamma, const float *beta, uint16_t *y,
                                float *mean, float *rstd, int N, long V, int C, float eps, float slope, void *ws,
                                size_t ws_bytes, void *stream) {
    return in_fwd(reinterpret_cast<const float *>(x), x_is_bf16 != 0, gamma, beta, reinterpret_cast<float *>(y), true, mean,
                  rstd, N, V, C, eps, slope, ws, ws_bytes, stream);
}

int mvd_instnorm_lrelu_bwd_bf16(const void *x, int x_is_bf16, const uint16_t *dy, const float *gamma, const float *beta,
                                const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta, int N,
                                long V, int C, float slope, void *ws, size_t ws_bytes, void *stream) {
    return in_bwd(reinterpret_cast<const float *>(x), x_is_bf16 != 0, reinterpret_cast<const float *>(dy), true, gamma, beta,
                  mean, rstd, reinterpret_cast<float *>(dx), dgamma, dbeta, N, V, C, slope, ws, ws_bytes, stream);
}
}
