// ConvTranspose3d with kernel == stride (the decoder's up-sampling, UNetDecoder.py:56-59): non-overlapping, so it is a
// plain GEMM per input voxel,
//     y[n, s*q + p, k] = bias[k] + sum_c x[n, q, c] * w[c][k][p]          (forward: [voxels x C] x [C x T*K])
//     dx[n, q, c]      = sum_p sum_k dy[n, s*q + p, k] * w[c][k][p]       (input gradient: [voxels x T*K] x [T*K x C])
// with T = s0*s1*s2 <= 8 positions.  Every activation element is used by exactly one M tile, so nothing is staged in
// LDS: a lane reads its 16 reduce channels of its voxel (64 contiguous bytes; the two lane halves share a 128-byte
// line) straight from HBM and its weight fragment (16 floats, packed layout of conv_geom.h) from L2, both one step
// ahead of the MFMAs; no barrier anywhere.  The gathered-tap engines ran these as T one-tap launches (each re-reading
// x and writing a strided eighth of y): 3 ms per step for 16 GFLOP; these kernels are bound by the HBM stream.
#include <algorithm>
#include <stdlib.h>
#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TranspGeom {
    int N, D, H, W, C, K;  // input dims, reduce / produce channels of the FORWARD op
    int s[3], T;
    long NV;               // N*D*H*W input voxels
};

// output-voxel index of input voxel v at position (0,0,0): two divisions per row, done once per lane row
__device__ inline long out_base(const TranspGeom &g, unsigned v) {
    const unsigned t1 = v / (unsigned)g.W, w = v - t1 * (unsigned)g.W;
    const unsigned t2 = t1 / (unsigned)g.H, h = t1 - t2 * (unsigned)g.H;  // t2 = n*D + d
    return (((long)t2 * g.s[0]) * (g.H * g.s[1]) + (long)h * g.s[1]) * (g.W * g.s[2]) + (long)w * g.s[2];
}

// ------------------------------------------------------------------------------------------------ forward
// grid (ceil(NV / 32), K / 32): workgroup = 32 input voxels x one 32-wide block of output channels; its four waves split
// the T positions (two each for the 2x2x2 up-sampling), so a wave carries 32 accumulator registers and five or more
// waves per SIMD hide the L2 / HBM round trips of the LDS-free operand fetches (one wave per position set with all 8
// accumulators ran at 2 waves per SIMD and 1.4 TB/s).  The x rows are re-read by the four waves from L1/L2.
__global__ __launch_bounds__(256, 4) void k_convT_fwd(const TranspGeom g, const float *__restrict__ x,
                                                      const float *__restrict__ wf, const float *__restrict__ bias,
                                                      float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const long vb = (long)blockIdx.x * 32;
    const int ppw = (g.T + 3) >> 2;            // positions per wave
    const int p0 = wave * ppw;
    if (p0 >= g.T) return;                     // whole wave (no barriers in this kernel)
    const int kb = blockIdx.y;
    const int nch = g.C >> 5;
    const long v = vb + i < g.NV ? vb + i : g.NV - 1;  // clamp: rows past the end are computed and dropped
    const float *xl = x + (size_t)v * g.C + h * 16;
    // packed weights [cc][t][h][k][16]
    const float *wl = wf + (((size_t)h * g.K + kb * 32 + i) << 4);
    const size_t wtap = (size_t)2 * g.K * 16;

    f32x16 acc[2];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.f;
    const int np = g.T - p0 < ppw ? g.T - p0 : ppw;  // 1 or 2 (ppw <= 2 since T <= 8)

    float4 a[2][4], b[2][2][4];
    auto load_b = [&](int c, float4 (&dst)[2][4]) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                dst[j][e] = *reinterpret_cast<const float4 *>(wl + ((size_t)c * g.T + p0 + (j < np ? j : 0)) * wtap + e * 4);
    };
#pragma unroll
    for (int e = 0; e < 4; e++) a[0][e] = *reinterpret_cast<const float4 *>(xl + e * 4);
    load_b(0, b[0]);
    for (int cc = 0; cc < nch; cc += 2) {  // two chunks per trip: static operand-buffer indices
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int c = cc + u;
            if (c < nch) {  // uniform
                const int cn = c + 1 < nch ? c + 1 : c;
#pragma unroll
                for (int e = 0; e < 4; e++) a[u ^ 1][e] = *reinterpret_cast<const float4 *>(xl + (size_t)cn * 32 + e * 4);
                load_b(cn, b[u ^ 1]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].x, b[u][j][e].x, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].y, b[u][j][e].y, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].z, b[u][j][e].z, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].w, b[u][j][e].w, acc[j], 0, 0, 0);
                }
            }
        }
    }
    const int k = kb * 32 + i;
    const float bv = settled(bias ? bias[k] : 0.f);
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    // output row index of THIS lane's voxel (two divisions, once); the accumulator rows a lane holds belong to other
    // lanes' voxels: fetched by a lane shuffle instead of 32 more divisions (the index math was the kernel's bottleneck)
    const int ob_own = (int)out_base(g, (unsigned)v);  // < 2^31 (host-checked)
    int poff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int p = p0 + (j < np ? j : 0);
        const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
        poff[j] = (pd * Hy + ph) * Wy + pw;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ob = __shfl(ob_own, row, 64);  // lane `row` (lower half) holds voxel vb + row
        if (vb + row < g.NV) {
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (j < np) y[(size_t)(ob + poff[j]) * g.K + k] = acc[j][r] + bv;
        }
    }
}

// Persistent form for the top level (64 -> 32 channels, 2x2x2 up-sampling): the kernel above gives a wave 32
// MFMAs per 32-channel chunk -- two chunks at the 64 -> 32 level -- between an HBM round trip for its operands and 32
// scattered stores, 352 us where the traffic (0.67 GB) and the MFMAs (17 GFLOP) each need ~0.13 ms.  Here a wave keeps its
// two positions' weights in registers for the whole launch (NCH x 32 registers), walks the 32-voxel blocks with a grid
// stride and has the next block's x rows in flight while it computes and stores the current one.
template <int NCH>
__global__ __launch_bounds__(256, 2) void k_convT_fwd_p(const TranspGeom g, const float *__restrict__ x,
                                                                       const float *__restrict__ wf,
                                                                       const float *__restrict__ bias, float *__restrict__ y,
                                                                       int nblk) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int p0 = wave * 2;  // T == 8: two positions per wave
    const int kb = blockIdx.y;
    // packed weights [cc][t][h][k][16]
    const float *wl = wf + (((size_t)h * g.K + kb * 32 + i) << 4);
    const size_t wtap = (size_t)2 * g.K * 16;
    float4 wr[NCH][2][4];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                wr[c][j][e] = *reinterpret_cast<const float4 *>(wl + ((size_t)c * g.T + p0 + j) * wtap + e * 4);
    const int k = kb * 32 + i;
    const float bv = settled(bias ? bias[k] : 0.f);
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    int poff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int p = p0 + j;
        const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
        poff[j] = (pd * Hy + ph) * Wy + pw;
    }
    float4 a[2][NCH][4];
    auto load_a = [&](int blk, float4 (&dst)[NCH][4]) {
        const long vv = (long)blk * 32 + i;
        const long v = vv < g.NV ? vv : g.NV - 1;  // clamp: rows past the end are computed and dropped
        const float *xl = x + (size_t)v * g.C + h * 16;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int e = 0; e < 4; e++) dst[c][e] = *reinterpret_cast<const float4 *>(xl + c * 32 + e * 4);
    };
    int blk = blockIdx.x;
    if (blk >= nblk) return;
    load_a(blk, a[0]);
    for (int it = 0; blk < nblk; blk += gridDim.x, it++) {
        const int nxt = blk + (int)gridDim.x < nblk ? blk + (int)gridDim.x : blk;  // (the last trip re-reads its own rows)
        auto trip = [&](float4 (&cur)[NCH][4], float4 (&nx)[NCH][4]) {
            load_a(nxt, nx);  // unconditional: in flight under this block's MFMAs and stores
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[c][e].x, wr[c][j][e].x, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[c][e].y, wr[c][j][e].y, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[c][e].z, wr[c][j][e].z, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[c][e].w, wr[c][j][e].w, acc[j], 0, 0, 0);
                }
            const long vb = (long)blk * 32;
            const long vv = vb + i;
            const int ob_own = (int)out_base(g, (unsigned)(vv < g.NV ? vv : g.NV - 1));  // < 2^31 (host-checked)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ob = __shfl(ob_own, row, 64);  // lane `row` (lower half) holds voxel vb + row
                if (vb + row < g.NV) {
#pragma unroll
                    for (int j = 0; j < 2; j++) y[(size_t)(ob + poff[j]) * g.K + k] = acc[j][r] + bv;
                }
            }
        };
        if (it & 1) trip(a[1], a[0]);
        else trip(a[0], a[1]);
    }
}

// ------------------------------------------------------------------------------------------------ input gradient
// grid (ceil(NV / 128), ceil(C / 32 / NT)): wave = 32 input voxels x NT 32-wide blocks of input channels; the reduce
// dimension runs over the T positions and the K/32 chunks of dy
template <int NT>
__global__ __launch_bounds__(256, NT == 1 ? 4 : 3) void k_convT_dgrad(const TranspGeom g, const float *__restrict__ dy,
                                                        const float *__restrict__ wb, float *__restrict__ dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const long vb = ((long)blockIdx.x * 4 + wave) * 32;
    if (vb >= g.NV) return;
    const int cb0 = blockIdx.y * NT;
    const int nkc = g.K >> 5, ncb = g.C >> 5;
    const long v = vb + i < g.NV ? vb + i : g.NV - 1;
    const long ob = out_base(g, (unsigned)v);
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    // packed weights (reduce K, produce C): [kc][t][h][c][16]
    const float *wl = wb + (((size_t)h * g.C + cb0 * 32 + i) << 4);
    const size_t wtap = (size_t)2 * g.C * 16;

    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;

    const int nsteps = g.T * nkc;  // step = (position p, chunk kc), kc fastest
    auto a_ptr = [&](int step) {
        const int p = step / nkc, kc = step - p * nkc;
        const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
        return dy + (size_t)(ob + ((long)pd * Hy + ph) * Wy + pw) * g.K + kc * 32 + h * 16;
    };
    auto b_ptr = [&](int step) {
        const int p = step / nkc, kc = step - p * nkc;
        return wl + ((size_t)kc * g.T + p) * wtap;
    };
    float4 a[2][4], b[2][NT][4];
    {
        const float *pa = a_ptr(0), *pb = b_ptr(0);
#pragma unroll
        for (int e = 0; e < 4; e++) a[0][e] = *reinterpret_cast<const float4 *>(pa + e * 4);
#pragma unroll
        for (int q = 0; q < NT; q++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                b[0][q][e] = (cb0 + q < ncb) ? *reinterpret_cast<const float4 *>(pb + (size_t)q * 512 + e * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int st = s0 + u;
            if (st < nsteps) {  // uniform
                const int sn = st + 1 < nsteps ? st + 1 : st;
                const float *pa = a_ptr(sn), *pb = b_ptr(sn);
#pragma unroll
                for (int e = 0; e < 4; e++) a[u ^ 1][e] = *reinterpret_cast<const float4 *>(pa + e * 4);
#pragma unroll
                for (int q = 0; q < NT; q++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        b[u ^ 1][q][e] = (cb0 + q < ncb) ? *reinterpret_cast<const float4 *>(pb + (size_t)q * 512 + e * 4)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int e = 0; e < 4; e++) {
#pragma unroll
                    for (int q = 0; q < NT; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].x, b[u][q][e].x, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NT; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].y, b[u][q][e].y, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NT; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].z, b[u][q][e].z, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NT; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e].w, b[u][q][e].w, acc[q], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NT; q++)
        if (cb0 + q < ncb) {
            const int c = (cb0 + q) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (vb + row < g.NV) dx[(size_t)(vb + row) * g.C + c] = acc[q][r];
            }
        }
}

// Persistent input gradient for the top level (K = 32 reduce channels, 2x2x2): a wave = one 32-channel block of dx, its
// eight positions' weights resident in registers (128 of its 256), 32-voxel groups walked with a grid stride; the dy rows
// of position p + 1 (the last position: position 0 of the wave's next group) are in flight under the 16 MFMAs of
// position p.  Same reasons as k_convT_fwd_p: the one-group-per-wave kernel spends its time in operand round trips.
__global__ __launch_bounds__(256, 2) void k_convT_dgrad_p(const TranspGeom g, const float *__restrict__ dy,
                                                         const float *__restrict__ wb, float *__restrict__ dx, int ngrp) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int ncb = g.C >> 5;
    const int cb = wave % ncb;             // host: 4 % ncb == 0
    const int gsub = wave / ncb, gper = 4 / ncb;  // voxel groups per workgroup
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    // packed weights (reduce K, produce C): [kc][t][h][c][16], kc == 0
    const float *wl = wb + (((size_t)h * g.C + cb * 32 + i) << 4);
    const size_t wtap = (size_t)2 * g.C * 16;
    float4 wr[8][4];
    int poff[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
#pragma unroll
        for (int e = 0; e < 4; e++) wr[p][e] = *reinterpret_cast<const float4 *>(wl + (size_t)p * wtap + e * 4);
        const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
        poff[p] = (pd * Hy + ph) * Wy + pw;
    }
    const int c = cb * 32 + i;
    const int gstride = (int)gridDim.x * gper;
    int grp = (int)blockIdx.x * gper + gsub;
    if (grp >= ngrp) return;
    auto row_ptr = [&](int gq) {  // this lane's dy row of group gq at position 0 (clamped past the end: computed, dropped)
        const long vv = (long)gq * 32 + i;
        const long ob = out_base(g, (unsigned)(vv < g.NV ? vv : g.NV - 1));
        return dy + (size_t)ob * g.K + h * 16;
    };
    const float *rp = row_ptr(grp);
    float4 a[2][4];
#pragma unroll
    for (int e = 0; e < 4; e++) a[0][e] = *reinterpret_cast<const float4 *>(rp + (size_t)poff[0] * g.K + e * 4);
    for (; grp < ngrp; grp += gstride) {
        const int gn = grp + gstride < ngrp ? grp + gstride : grp;  // (the last trip re-reads its own first row)
        const float *rn = row_ptr(gn);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const float *pn = p < 7 ? rp + (size_t)poff[p < 7 ? p + 1 : 0] * g.K : rn + (size_t)poff[0] * g.K;
#pragma unroll
            for (int e = 0; e < 4; e++) a[(p + 1) & 1][e] = *reinterpret_cast<const float4 *>(pn + e * 4);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p & 1][e].x, wr[p][e].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p & 1][e].y, wr[p][e].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p & 1][e].z, wr[p][e].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p & 1][e].w, wr[p][e].w, acc, 0, 0, 0);
            }
        }
        const long vb = (long)grp * 32;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (vb + row < g.NV) dx[(size_t)(vb + row) * g.C + c] = acc[r];
        }
        rp = rn;
    }
}

// ================================================================================================ bf16 twins
// Same two GEMMs on v_mfma_f32_32x32x16_bf16: a 32-channel chunk is two k-steps; lane (i, h) reads the 8 bf16 reduce
// channels s*16 + h*8 .. +7 of its voxel (16 bytes) and its weight fragment [..][s][h][k][8] of conv_bf16.hip's layout.
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));

__device__ inline bf16x8t ld_bf8(const unsigned short *p) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    return __builtin_bit_cast(bf16x8t, q);
}

__global__ __launch_bounds__(256, 4) void k_convT_fwd16(const TranspGeom g, const unsigned short *__restrict__ x,
                                                        const unsigned short *__restrict__ wf,
                                                        const float *__restrict__ bias, unsigned short *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const long vb = (long)blockIdx.x * 32;
    const int ppw = (g.T + 3) >> 2;
    const int p0 = wave * ppw;
    if (p0 >= g.T) return;
    const int kb = blockIdx.y;
    const int nch = g.C >> 5;
    const long v = vb + i < g.NV ? vb + i : g.NV - 1;
    const unsigned short *xl = x + (size_t)v * g.C + h * 8;
    // packed weights [cc][t][s][h][k][8]
    const unsigned short *wl = wf + (((size_t)h * g.K + kb * 32 + i) << 3);
    const size_t ws_ = (size_t)2 * g.K * 8;   // elements between k-steps
    const size_t wtap = 2 * ws_;              // elements between taps
    f32x16 acc[2];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.f;
    const int np = g.T - p0 < ppw ? g.T - p0 : ppw;
    bf16x8t a[2][2], b[2][2][2];  // [buffer][k-step] / [buffer][position][k-step]
    auto load_ab = [&](int c, int buf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            a[buf][s2] = ld_bf8(xl + (size_t)c * 32 + s2 * 16);
#pragma unroll
            for (int j = 0; j < 2; j++)
                b[buf][j][s2] = ld_bf8(wl + ((size_t)c * g.T + p0 + (j < np ? j : 0)) * wtap + s2 * ws_);
        }
    };
    load_ab(0, 0);
    for (int cc = 0; cc < nch; cc += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int c = cc + u;
            if (c < nch) {
                load_ab(c + 1 < nch ? c + 1 : c, u ^ 1);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u][j][s2], a[u][s2], acc[j], 0, 0, 0);  // D^T: see epilogue
            }
        }
    }
    // D^T layout (operands swapped): column (lane & 31) = this lane's own voxel, rows = output channels
    // (r & 3) + 8 * (r >> 2) + 4 * h: four 8-byte packets of 4 consecutive channels per position
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    float4 bq[4];  // the bias values first, settled (common.h): loads between the stores would serialise them
#pragma unroll
    for (int rg = 0; rg < 4; rg++)
        bq[rg] = bias ? *reinterpret_cast<const float4 *>(bias + kb * 32 + 8 * rg + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int rg = 0; rg < 4; rg++) bq[rg] = settled(bq[rg]);
    if (vb + i < g.NV) {
        const long ob = out_base(g, (unsigned)v);
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (j < np) {
                const int p = p0 + j;
                const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
                unsigned short *yo = y + (size_t)(ob + ((long)pd * Hy + ph) * Wy + pw) * g.K + kb * 32 + 4 * h;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const float4 b4 = bq[rg];
                    uint2 q;
                    q.x = (unsigned)f2bf(acc[j][rg * 4 + 0] + b4.x) | ((unsigned)f2bf(acc[j][rg * 4 + 1] + b4.y) << 16);
                    q.y = (unsigned)f2bf(acc[j][rg * 4 + 2] + b4.z) | ((unsigned)f2bf(acc[j][rg * 4 + 3] + b4.w) << 16);
                    *reinterpret_cast<uint2 *>(yo + 8 * rg) = q;
                }
            }
    }
}

// persistent form of the forward for the top level (C = 64, K = 32, 2x2x2), as k_convT_fwd_p: the one-block kernel gives a
// wave EIGHT MFMAs between its operand round trip and its stores
__global__ __launch_bounds__(256, 3) void k_convT_fwd16_p(const TranspGeom g, const unsigned short *__restrict__ x,
                                                          const unsigned short *__restrict__ wf,
                                                          const float *__restrict__ bias, unsigned short *__restrict__ y,
                                                          int nblk) {
    constexpr int NCH = 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int p0 = wave * 2;
    // packed weights [cc][t][s][h][k][8]
    const unsigned short *wl = wf + (((size_t)h * g.K + i) << 3);
    const size_t ws_ = (size_t)2 * g.K * 8, wtap = 2 * ws_;
    bf16x8t wr[NCH][2][2];  // [chunk][position][k-step]
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) wr[c][j][s2] = ld_bf8(wl + ((size_t)c * g.T + p0 + j) * wtap + s2 * ws_);
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    float4 bq[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++)
        bq[rg] = settled(bias ? *reinterpret_cast<const float4 *>(bias + 8 * rg + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f));
    long poff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int p = p0 + j;
        const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
        poff[j] = ((long)pd * Hy + ph) * Wy + pw;
    }
    bf16x8t a[2][NCH][2];
    auto load_a = [&](int blk, bf16x8t (&dst)[NCH][2]) {
        const long vv = (long)blk * 32 + i;
        const unsigned short *xl = x + (size_t)(vv < g.NV ? vv : g.NV - 1) * g.C + h * 8;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) dst[c][s2] = ld_bf8(xl + c * 32 + s2 * 16);
    };
    int blk = blockIdx.x;
    if (blk >= nblk) return;
    load_a(blk, a[0]);
    for (int it = 0; blk < nblk; blk += gridDim.x, it++) {
        const int nxt = blk + (int)gridDim.x < nblk ? blk + (int)gridDim.x : blk;
        auto trip = [&](bf16x8t (&cur)[NCH][2], bf16x8t (&nx)[NCH][2]) {
            load_a(nxt, nx);
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[c][j][s2], cur[c][s2], acc[j], 0, 0, 0);  // D^T
            const long vv = (long)blk * 32 + i;
            if (vv < g.NV) {
                const long ob = out_base(g, (unsigned)vv);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    unsigned short *yo = y + (size_t)(ob + poff[j]) * g.K + 4 * h;
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const float4 b4 = bq[rg];
                        uint2 q;
                        q.x = (unsigned)f2bf(acc[j][rg * 4 + 0] + b4.x) | ((unsigned)f2bf(acc[j][rg * 4 + 1] + b4.y) << 16);
                        q.y = (unsigned)f2bf(acc[j][rg * 4 + 2] + b4.z) | ((unsigned)f2bf(acc[j][rg * 4 + 3] + b4.w) << 16);
                        *reinterpret_cast<uint2 *>(yo + 8 * rg) = q;
                    }
                }
            }
        };
        if (it & 1) trip(a[1], a[0]);
        else trip(a[0], a[1]);
    }
}

template <int NT>
__global__ __launch_bounds__(256, 4) void k_convT_dgrad16(const TranspGeom g, const unsigned short *__restrict__ dy,
                                                          const unsigned short *__restrict__ wb,
                                                          unsigned short *__restrict__ dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const long vb = ((long)blockIdx.x * 4 + wave) * 32;
    if (vb >= g.NV) return;
    const int cb0 = blockIdx.y * NT;
    const int nkc = g.K >> 5, ncb = g.C >> 5;
    const long v = vb + i < g.NV ? vb + i : g.NV - 1;
    const long ob = out_base(g, (unsigned)v);
    const int Hy = g.H * g.s[1], Wy = g.W * g.s[2];
    // packed weights (reduce K, produce C): [kc][t][s][h][c][8]
    const unsigned short *wl = wb + (((size_t)h * g.C + cb0 * 32 + i) << 3);
    const size_t ws_ = (size_t)2 * g.C * 8, wtap = 2 * ws_;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    const int nsteps = g.T * nkc;
    bf16x8t a[2][2], b[2][NT][2];
    auto load_ab = [&](int step, int buf) {
        const int p = step / nkc, kc = step - p * nkc;
        const int pw = p % g.s[2], ph = (p / g.s[2]) % g.s[1], pd = p / (g.s[2] * g.s[1]);
        const unsigned short *pa = dy + (size_t)(ob + ((long)pd * Hy + ph) * Wy + pw) * g.K + kc * 32 + h * 8;
        const unsigned short *pb = wl + ((size_t)kc * g.T + p) * wtap;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            a[buf][s2] = ld_bf8(pa + s2 * 16);
#pragma unroll
            for (int q = 0; q < NT; q++)
                b[buf][q][s2] = ld_bf8(pb + s2 * ws_ + (size_t)(cb0 + q < ncb ? q : 0) * 256);
        }
    };
    load_ab(0, 0);
    for (int s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int st = s0 + u;
            if (st < nsteps) {
                load_ab(st + 1 < nsteps ? st + 1 : st, u ^ 1);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int q = 0; q < NT; q++)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u][q][s2], a[u][s2], acc[q], 0, 0, 0);  // D^T
            }
        }
    }
    // D^T layout: column = this lane's own voxel, rows = input channels (r & 3) + 8 * (r >> 2) + 4 * h
    if (vb + i < g.NV) {
#pragma unroll
        for (int q = 0; q < NT; q++)
            if (cb0 + q < ncb) {
                unsigned short *xo = dx + (size_t)v * g.C + (cb0 + q) * 32 + 4 * h;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(acc[q][rg * 4 + 0]) | ((unsigned)f2bf(acc[q][rg * 4 + 1]) << 16);
                    pk.y = (unsigned)f2bf(acc[q][rg * 4 + 2]) | ((unsigned)f2bf(acc[q][rg * 4 + 3]) << 16);
                    *reinterpret_cast<uint2 *>(xo + 8 * rg) = pk;
                }
            }
    }
}

static bool transp_geom(TranspGeom &g, int N, int D, int H, int W, int C, int K, const int st[3]) {
    g.N = N; g.D = D; g.H = H; g.W = W; g.C = C; g.K = K;
    for (int a = 0; a < 3; a++) g.s[a] = st[a];
    g.T = st[0] * st[1] * st[2];
    g.NV = (long)N * D * H * W;
    // 32-bit voxel arithmetic inside the kernels
    return C % 32 == 0 && K % 32 == 0 && g.T <= 8 && g.NV * g.T < (1L << 31) && g.NV > 0;
}

// return -1: shape not covered (caller uses the gathered-tap engines)
int convT_fwd_direct(const float *x, const float *wf, const float *bias, float *y, int N, int D, int H, int W, int C, int K,
                     const int st[3], hipStream_t s) {
    TranspGeom g;
    if (!transp_geom(g, N, D, H, W, C, K, st)) return -1;
    if (((uintptr_t)x | (uintptr_t)wf) & 15) return -1;
    const long bx = (g.NV + 31) / 32;
    if (bx > (1L << 30) || K / 32 > 65535) return -1;
    static const int pers = getenv("MVD_CONVT_PERSIST") ? atoi(getenv("MVD_CONVT_PERSIST")) : 1;
    if (pers && g.T == 8 && C == 64 && K == 32 && bx >= 4096) {
        // persistent, weights resident in registers (64 of the wave's 256): two workgroups per CU
        const unsigned gx = (unsigned)std::min<long>(bx, 512);
        hipLaunchKernelGGL(k_convT_fwd_p<2>, dim3(gx, 1), dim3(256), 0, s, g, x, wf, bias, y, (int)bx);
        return check_launch("convT fwd (persistent direct GEMM)");
    }
    hipLaunchKernelGGL(k_convT_fwd, dim3((unsigned)bx, K / 32), dim3(256), 0, s, g, x, wf, bias, y);
    return check_launch("convT fwd (direct GEMM)");
}

int convT_dgrad_direct(const float *dy, const float *wb, float *dx, int N, int D, int H, int W, int C, int K,
                       const int st[3], hipStream_t s) {
    TranspGeom g;
    if (!transp_geom(g, N, D, H, W, C, K, st)) return -1;
    if (((uintptr_t)dy | (uintptr_t)wb) & 15) return -1;
    const long bx = (g.NV + 127) / 128;
    if (bx > (1L << 30)) return -1;
    // one 32-channel block per wave (93 registers: five waves per SIMD hide the operand round trips) while dy is re-read
    // at most 4 times; two blocks per wave for the wide low-resolution stages
    const int ncb = C / 32;
    static const int pers = getenv("MVD_CONVT_PERSIST") ? atoi(getenv("MVD_CONVT_PERSIST")) : 1;
    const long ngrp = (g.NV + 31) / 32;
    if (pers && g.T == 8 && K == 32 && (ncb == 1 || ncb == 2 || ncb == 4) && ngrp >= 4096 && ngrp < (1L << 30)) {
        const long nb = (ngrp * ncb + 3) / 4;  // workgroups if every wave had one group
        hipLaunchKernelGGL(k_convT_dgrad_p, dim3((unsigned)std::min<long>(nb, 512)), dim3(256), 0, s, g, dy, wb, dx, (int)ngrp);
        return check_launch("convT dgrad (persistent direct GEMM)");
    }
    if (ncb > 4 && ncb % 2 == 0) {
        hipLaunchKernelGGL(k_convT_dgrad<2>, dim3((unsigned)bx, ncb / 2), dim3(256), 0, s, g, dy, wb, dx);
    } else {
        hipLaunchKernelGGL(k_convT_dgrad<1>, dim3((unsigned)bx, ncb), dim3(256), 0, s, g, dy, wb, dx);
    }
    return check_launch("convT dgrad (direct GEMM)");
}

int convT_fwd_direct16(const unsigned short *x, const unsigned short *wf, const float *bias, unsigned short *y, int N, int D,
                       int H, int W, int C, int K, const int st[3], hipStream_t s) {
    TranspGeom g;
    if (!transp_geom(g, N, D, H, W, C, K, st)) return -1;
    if (((uintptr_t)x | (uintptr_t)wf) & 15) return -1;
    const long bx = (g.NV + 31) / 32;
    if (bx > (1L << 30) || K / 32 > 65535) return -1;
    static const int pers = getenv("MVD_CONVT_PERSIST") ? atoi(getenv("MVD_CONVT_PERSIST")) : 1;
    if (pers && g.T == 8 && C == 64 && K == 32 && bx >= 4096) {
        hipLaunchKernelGGL(k_convT_fwd16_p, dim3((unsigned)std::min<long>(bx, 1024)), dim3(256), 0, s, g, x, wf, bias, y, (int)bx);
        return check_launch("convT fwd (bf16 persistent direct GEMM)");
    }
    hipLaunchKernelGGL(k_convT_fwd16, dim3((unsigned)bx, K / 32), dim3(256), 0, s, g, x, wf, bias, y);
    return check_launch("convT fwd (bf16 direct GEMM)");
}

int convT_dgrad_direct16(const unsigned short *dy, const unsigned short *wb, unsigned short *dx, int N, int D, int H, int W,
                         int C, int K, const int st[3], hipStream_t s) {
    TranspGeom g;
    if (!transp_geom(g, N, D, H, W, C, K, st)) return -1;
    if (((uintptr_t)dy | (uintptr_t)wb) & 15) return -1;
    const long bx = (g.NV + 127) / 128;
    if (bx > (1L << 30)) return -1;
    const int ncb = C / 32;
    if (ncb > 4 && ncb % 2 == 0)
        hipLaunchKernelGGL(k_convT_dgrad16<2>, dim3((unsigned)bx, ncb / 2), dim3(256), 0, s, g, dy, wb, dx);
    else
        hipLaunchKernelGGL(k_convT_dgrad16<1>, dim3((unsigned)bx, ncb), dim3(256), 0, s, g, dy, wb, dx);
    return check_launch("convT dgrad (bf16 direct GEMM)");
}

}  // namespace mvd
