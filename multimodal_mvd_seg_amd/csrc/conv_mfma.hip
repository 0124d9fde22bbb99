// fp32 MFMA implicit-GEMM conv engines for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Forward-type kernel (conv fwd, conv dgrad per parity class, convT fwd per class, convT dgrad):
//   workgroup = 4 waves = one 4 x (4*MT) x 8 tile of the iteration grid x (32*NT) output channels.
//   Loop over reduce-channel chunks of CK: stage the input halo tile [slots][CK] and the weight slice
//   [taps][h][32*NT][CK/2] into LDS (the packed weight layout of conv_geom.h makes the latter a linear copy), then
//   for every tap each wave issues MT*NT*(CK/2) MFMAs from ds_read_b128/b64 fragments (the k order inside a chunk is
//   permuted identically for A and B, so one wide LDS read feeds CK/2 k-steps).  Single LDS buffer; latency is hidden
//   by 2-3 co-resident workgroups per CU (LDS <= 75 KB each), not by in-kernel double buffering.
// Wgrad-type kernel (conv wgrad, convT wgrad):
//   workgroup = one 32(c) x 32(k) block of dW for all taps; taps are dealt round-robin to the 4 waves (<= 7 x 16
//   accumulator registers each); the GEMM K dimension is the voxel index: each workgroup walks a strided subset of
//   spatial tiles (split-K), staging the A halo [slots][32 c] and B tile [slots][32 k] in LDS, and writes fp32 partials;
//   a fixed-order second stage sums the splits in fp64 and emits the torch weight layout (deterministic, no atomics).
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FwdTile {
    int ED, EH, EW, nslots;
    int min_off[3];
    int ntd, nth, ntw;
    int K;  // produce channels of the launch (K1+K2)
    int toff[27];
    int dbg;  // timing experiments only (MVD_CONV_DBG): bit0 skip staging loads, bit1 skip the MFMA loop
};

// L32: the packed weights use the CK = 32 layout (C % 32 == 0) but this kernel still walks 8-channel sub-chunks
// {sc*4..sc*4+3} u {16+sc*4..16+sc*4+3} of each 32-chunk (used for the wide-halo stride-2 geometries only).
// Epilogue of the gather kernels: stores one 32 x 32 accumulator tile whose rows are a 4 (h) x 8 (w) voxel patch of output
// plane od.  Accumulator row r of lane half hh is voxel (r >> 2, (r & 3) + 4 * hh): everything but the 4 * hh shift along W
// is wave-uniform, so the voxel offset is scalar arithmetic and a lane adds one 64-bit pointer (the per-row index chain
// of 64-bit multiplies used to cost ~25 VALU instructions per stored value = more than the MFMAs of a few-tap tile).
__device__ __forceinline__ void store_tile32(const FwdGeom &g, const f32x16 &acc, float bv, int n, int od, int ohb, int ow0,
                                             int k, int hh, int S, int split, int Ktot, float *__restrict__ part,
                                             float *__restrict__ y1, float *__restrict__ y2) {
    const int owl = ow0 + 4 * hh;
    if (S > 1) {  // raw partial, indexed by the iteration voxel: part[split][n][o][K]
        float *pl = part + (size_t)(4 * hh) * Ktot + k;
        const size_t nb = ((size_t)split * g.N + n) * ((size_t)g.Do * g.Ho * g.Wo);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int oh = ohb + (r >> 2), owu = ow0 + (r & 3);
            const size_t uo = (nb + ((size_t)od * g.Ho + oh) * g.Wo + owu) * Ktot;
            if (oh < g.Ho && owl + (r & 3) < g.Wo) pl[uo] = acc[r];
        }
    } else if (g.K2 == 0 || g.K1 == g.K2) {
        const int Ks = g.K1;
        float *yl = (k < g.K1 ? y1 + k : y2 + (k - g.K1)) + (size_t)(4 * hh * g.so[2]) * Ks;
        const size_t pb = ((size_t)n * g.Dy + (od * g.so[0] + g.oo[0])) * g.Hy;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int oh = ohb + (r >> 2), owu = ow0 + (r & 3);
            const size_t uo = ((pb + (oh * g.so[1] + g.oo[1])) * g.Wy + (owu * g.so[2] + g.oo[2])) * Ks;
            if (oh < g.Ho && owl + (r & 3) < g.Wo) yl[uo] = acc[r] + bv;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int oh = ohb + (r >> 2), ow = owl + (r & 3);
            if (oh < g.Ho && ow < g.Wo) {
                const size_t ov = (((size_t)n * g.Dy + (od * g.so[0] + g.oo[0])) * g.Hy + (oh * g.so[1] + g.oo[1])) * g.Wy +
                                  (ow * g.so[2] + g.oo[2]);
                const float val = acc[r] + bv;
                if (k < g.K1)
                    y1[ov * g.K1 + k] = val;
                else
                    y2[ov * g.K2 + (k - g.K1)] = val;
            }
        }
    }
}

template <int CK, int NT, int MT, bool L32>
__global__ __launch_bounds__(256, 2) void k_fwd_mfma(const FwdGeom g, const FwdTile tg, const float *__restrict__ a1,
                                                     const float *__restrict__ a2, const float *__restrict__ w,
                                                     const float *__restrict__ bias, float *__restrict__ y1,
                                                     float *__restrict__ y2, const int S, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int HH = CK / 2;
    constexpr int KT = 32 * NT;
    float *Xs = lds;
    float *Ws = lds + (size_t)tg.nslots * CK;
    int *tapoff = reinterpret_cast<int *>(Ws + (size_t)g.ntaps * 2 * KT * HH);
    int *wts = tapoff + 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    const int tw_ = tile % tg.ntw, th_ = (tile / tg.ntw) % tg.nth, td_ = tile / (tg.ntw * tg.nth);
    // blockIdx.y = kb + nkb * split: skinny problems split the reduce-channel chunks over S workgroups (fixed-order
    // second stage in k_split_reduce)
    const int nkb = tg.K / (32 * NT);
    const int kb = blockIdx.y % nkb, split = blockIdx.y / nkb, n = blockIdx.z;
    const int od0 = td_ * 4, oh0 = th_ * (4 * MT), ow0 = tw_ * 8;
    const int iz0 = od0 * g.sa[0] + tg.min_off[0], iy0 = oh0 * g.sa[1] + tg.min_off[1],
              ix0 = ow0 * g.sa[2] + tg.min_off[2];
#pragma unroll
    for (int t = 0; t < 27; t++)
        if (tid == t && t < g.ntaps) {
            tapoff[t] = tg.toff[t];
            wts[t] = g.wt[t];
        }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < NT; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][q][r] = 0.f;

    int sbase[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int hy = 4 * m + (i >> 3), wx = i & 7;
        sbase[m] = ((wave * g.sa[0]) * tg.EH + hy * g.sa[1]) * tg.EW + wx * g.sa[2];
    }

    const int C = g.C1 + g.C2;
    const int nch = C / CK;  // (L32: CK == 8 sub-chunks of the 32-chunks)
    constexpr int PARTS = CK / 4;
    constexpr int ROWV = KT * HH / 4;  // float4 per (tap, h) weight row
    constexpr int SB = 8;              // staging batch (independent loads in flight per thread)
    const int nx = tg.nslots * PARTS;
    const int nw = g.ntaps * 2 * ROWV;
    const int EHW = tg.EH * tg.EW;

    const int cc_begin = (int)((long)split * nch / S), cc_end = (int)((long)(split + 1) * nch / S);
    for (int cc = cc_begin; cc < cc_end; cc++) {
        __syncthreads();
        const int c0 = L32 ? (cc >> 2) * 32 : cc * CK;
        const int sc = L32 ? (cc & 3) : 0;
        const float *src;
        int Cs, cofs;
        if (c0 < g.C1) {
            src = a1; Cs = g.C1; cofs = c0;
        } else {
            src = a2; Cs = g.C2; cofs = c0 - g.C1;
        }
        // staging in batches of SB independent 16-byte loads per thread (all issued before the first LDS write)
        for (int base = 0; base < ((tg.dbg & 1) ? 0 : nx); base += SB * 256) {
            float4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int idx = base + u * 256 + tid;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < nx) {
                    const int slot = idx / PARTS, part = idx - slot * PARTS;
                    const int ez = slot / EHW, rem = slot - ez * EHW;
                    const int ey = rem / tg.EW, ex = rem - ey * tg.EW;
                    const int id = iz0 + ez, ih = iy0 + ey, iw = ix0 + ex;
                    if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                        v[u] = *reinterpret_cast<const float4 *>(
                            src + ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * Cs + cofs +
                            (L32 ? part * 16 + sc * 4 : part * 4));
                }
            }
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int idx = base + u * 256 + tid;
                if (idx < nx) *reinterpret_cast<float4 *>(Xs + (size_t)idx * 4) = v[u];  // slot*CK + part*4 == idx*4
            }
        }
        for (int base = 0; base < ((tg.dbg & 1) ? 0 : nw); base += SB * 256) {
            float4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int idx = base + u * 256 + tid;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // (left undefined the array went to scratch memory: every load
                if (idx < nw) {                          // waited for and parked there before the next one was issued)
                    const int row = idx / ROWV, j = idx - row * ROWV;
                    const int t = row >> 1, hh = row & 1;
                    if (L32)  // HH == 4: one float4 per output channel j, strided by the 16-float k rows of the CK=32 layout
                        v[u] = *reinterpret_cast<const float4 *>(
                            w + ((((size_t)(cc >> 2) * g.T + wts[t]) * 2 + hh) * tg.K + (size_t)kb * KT + j) * 16 + sc * 4);
                    else
                        v[u] = *reinterpret_cast<const float4 *>(
                            w + ((((size_t)cc * g.T + wts[t]) * 2 + hh) * tg.K + (size_t)kb * KT) * HH + j * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < SB; u++) {
                const int idx = base + u * 256 + tid;
                if (idx < nw) *reinterpret_cast<float4 *>(Ws + (size_t)idx * 4) = v[u];  // row*KT*HH + j*4 == idx*4
            }
        }
        __syncthreads();
        for (int t = 0; t < ((tg.dbg & 2) ? 0 : g.ntaps); t++) {
            const int to = tapoff[t];
            float af[MT][HH], bf[NT][HH];
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float *pa = Xs + (size_t)(sbase[m] + to) * CK + h * HH;
                if (CK == 8) {
                    float4 q = *reinterpret_cast<const float4 *>(pa);
                    af[m][0] = q.x; af[m][1] = q.y; af[m][HH - 2] = q.z; af[m][HH - 1] = q.w;
                } else {
                    float2 q = *reinterpret_cast<const float2 *>(pa);
                    af[m][0] = q.x; af[m][1] = q.y;
                }
            }
#pragma unroll
            for (int q_ = 0; q_ < NT; q_++) {
                const float *pb = Ws + ((size_t)(t * 2 + h) * KT + q_ * 32 + i) * HH;
                if (CK == 8) {
                    float4 q = *reinterpret_cast<const float4 *>(pb);
                    bf[q_][0] = q.x; bf[q_][1] = q.y; bf[q_][HH - 2] = q.z; bf[q_][HH - 1] = q.w;
                } else {
                    float2 q = *reinterpret_cast<const float2 *>(pb);
                    bf[q_][0] = q.x; bf[q_][1] = q.y;
                }
            }
#pragma unroll
            for (int e = 0; e < HH; e++)
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int q_ = 0; q_ < NT; q_++)
                        acc[m][q_] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][e], bf[q_][e], acc[m][q_], 0, 0, 0);
        }
    }

    // epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (voxel of the 32-row M tile)
    const int od = od0 + wave;
    if (od >= g.Do) return;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q_ = 0; q_ < NT; q_++) {
            const int k = kb * KT + q_ * 32 + i;
            const float bv = settled((bias && S == 1) ? bias[k] : 0.f);
            store_tile32(g, acc[m][q_], bv, n, od, oh0 + 4 * m, ow0, k, h, S, split, tg.K, part, y1, y2);
        }
}

static const size_t LDS_LIMIT = 160 * 1024;
static const int MAX_SPLIT = 16;

// y[mapped(n,o)][k] = bias[k] + sum_s part[s][n][o][k]   (fixed order)
__global__ void k_split_reduce(const FwdGeom g, const float *__restrict__ part, const float *__restrict__ bias,
                               float *__restrict__ y1, float *__restrict__ y2, int S) {
    const int K = g.K1 + g.K2;
    const size_t per = (size_t)g.N * g.Do * g.Ho * g.Wo * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += (size_t)gridDim.x * blockDim.x) {
        float s = bias ? bias[idx % K] : 0.f;
        int j = 0;
        for (; j + 3 < S; j += 4) {  // four partials in flight, added in split order (same sum as one at a time)
            const float p0 = part[(size_t)j * per + idx], p1 = part[(size_t)(j + 1) * per + idx];
            const float p2 = part[(size_t)(j + 2) * per + idx], p3 = part[(size_t)(j + 3) * per + idx];
            s += p0;
            s += p1;
            s += p2;
            s += p3;
        }
        for (; j < S; j++) s += part[(size_t)j * per + idx];
        const int k = (int)(idx % K);
        size_t r = idx / K;
        const int ow = (int)(r % g.Wo);
        r /= g.Wo;
        const int oh = (int)(r % g.Ho);
        r /= g.Ho;
        const int od = (int)(r % g.Do);
        const int n = (int)(r / g.Do);
        const size_t ov = (((size_t)n * g.Dy + (od * g.so[0] + g.oo[0])) * g.Hy + (oh * g.so[1] + g.oo[1])) * g.Wy +
                          (ow * g.so[2] + g.oo[2]);
        if (k < g.K1)
            y1[ov * g.K1 + k] = s;
        else
            y2[ov * g.K2 + (k - g.K1)] = s;
    }
}

size_t fwd_mfma_ws(int N, long out_vox, int K) {
    // only skinny problems split: at most 64 tiles of 128 voxels per sample
    if (out_vox > 64 * 128) return 0;
    return (size_t)MAX_SPLIT * N * out_vox * K * sizeof(float) + 256;
}

// ------------------------------------------------------------------------------------------------ CK = 32 kernel
// Stride-1 gathers with C % 32 == 0 (every 3x3x3 conv fwd / dgrad of the net but the strided ones).
//   * workgroup = 4 waves = a 4 x (4*MT) x 8 voxel tile x 32*NT output channels;
//   * per 32-channel chunk the whole halo tile [slots][32 ch] -- each voxel one full 128-byte line, padded to 36
//     floats against LDS bank conflicts -- is staged ONCE (the 8-channel chunks of k_fwd_mfma re-fetch every line four
//     times and go HBM-bound);
//   * weights stream through a double-buffered LDS ring in groups of TG taps; the next group is prefetched into
//     registers while the current group's MFMAs run (one barrier per group);
//   * work items are dealt to workgroups in an XCD-contiguous order so that neighbouring tiles (shared halos) and the
//     k-blocks of one tile meet in one XCD's L2.
struct Fwd32Tile {
    int EH, EW, nslots;
    int magW, magHW;  // 16-bit reciprocal multipliers for / EW and / (EH*EW) (host-verified on the slot range)
    int min_off[3];
    int ntd, nth, ntw, nkb, S;
    int nitems;
    int K;
    int toff[27];
    int dbg;
};

template <int NT, int MT, int TG>
__global__ __launch_bounds__(256, 2) void k_fwd32(const FwdGeom g, const Fwd32Tile tg, const float *__restrict__ a1,
                                                  const float *__restrict__ a2, const float *__restrict__ w,
                                                  const float *__restrict__ bias, float *__restrict__ y1,
                                                  float *__restrict__ y2, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KT = 32 * NT;
    constexpr int XS = 36;                       // floats per halo slot (32 + 4 pad)
    constexpr int WS = 20;                       // floats per (tap, h, k) weight row (16 + 4 pad)
    constexpr int XR = MT == 2 ? 19 : 12;        // float4 per thread: 608 / 384 slots x 8
    constexpr int XB = MT == 2 ? 10 : 12;        // staging batch (loads in flight per thread)
    constexpr int WR = TG * 2 * KT * 4 / 256;    // float4 per thread per weight group
    constexpr int WBUF = TG * 2 * KT * WS;       // floats per weight buffer
    static_assert(TG * 2 * KT * 4 % 256 == 0, "weight group must be a multiple of 256 float4");
    float *Xs = lds;
    float *Wsm = lds + (size_t)XR * 32 * XS;     // halo buffer sized for XR*32 >= nslots slots

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    // XCD-contiguous item order (blocks b and b+8 share an XCD): block b takes item (b%8)*ceil(n/8) + b/8
    const int per_xcd = (tg.nitems + 7) >> 3;
    int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= tg.nitems) return;  // whole workgroup
    unsigned r_ = (unsigned)item;
    const int kb = (int)(r_ % (unsigned)tg.nkb); r_ /= (unsigned)tg.nkb;
    const int split = (int)(r_ % (unsigned)tg.S); r_ /= (unsigned)tg.S;
    const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
    const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
    const int td_ = (int)(r_ % (unsigned)tg.ntd);
    const int n = (int)(r_ / (unsigned)tg.ntd);

    const int C = g.C1 + g.C2;
    const int nch = C / 32;
    const int ngroups = (g.ntaps + TG - 1) / TG;
    const int EHW = tg.EH * tg.EW;
    const int nx = tg.nslots * 8;
    const int od0 = td_ * 4, oh0 = th_ * (4 * MT), ow0 = tw_ * 8;
    const int iz0 = od0 + tg.min_off[0], iy0 = oh0 + tg.min_off[1], ix0 = ow0 + tg.min_off[2];

    int sbase[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) sbase[m] = ((wave * tg.EH) + 4 * m + (i >> 3)) * tg.EW + (i & 7);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < NT; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][q][r] = 0.f;

    // Per-tap tables across the lanes of a wave (lane t: halo offset and packed-weight index of tap t), read back with
    // v_readlane inside the loops (round 2).  Indexed out of the kernel argument they were an s_load per tap in the MFMA
    // loop, each followed by s_waitcnt lgkmcnt(0) -- which also drains every LDS operand read in flight.
    const int lt = lane < g.ntaps ? lane : g.ntaps - 1;
    const int toff_l = tg.toff[lt < 27 ? lt : 26], wt_l = g.wt[lt < 27 ? lt : 26];
    float4 wr[WR];
    // weight group `gidx` of chunk `cc` -> registers.  float4 index inside the group: idx = u*256 + tid =
    // ((tl*2+hh)*KT + k)*4 + e4; with 256 threads the tap of slot u is a compile-time function of u, so the tap id is
    // wave-uniform (scalar load from the kernel arguments) and the per-thread part of the address is loop invariant.
    int woff[WR];
#pragma unroll
    for (int u = 0; u < WR; u++) {
        const int idx = u * 256 + tid;
        const int e4 = idx & 3, k = (idx >> 2) % KT;
        const int hh = (NT == 1) ? (tid >> 7) : (u & 1);
        woff[u] = (hh * tg.K + kb * KT + k) * 16 + e4 * 4;
    }
    auto load_w = [&](int cc, int gidx) {
#pragma unroll
        for (int u = 0; u < WR; u++) {
            const int tl = (NT == 1) ? u : (u >> 1);
            const int t = gidx * TG + tl;  // uniform
            wr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < g.ntaps && !(tg.dbg & 1)) {
                const int wt = __builtin_amdgcn_readlane(wt_l, t);
                wr[u] = *reinterpret_cast<const float4 *>(w + ((size_t)cc * g.T + wt) * 2 * tg.K * 16 + woff[u]);
            }
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < WR; u++) {
            const int idx = u * 256 + tid;
            *reinterpret_cast<float4 *>(Wsm + (size_t)buf * WBUF + (size_t)(idx >> 2) * WS + (idx & 3) * 4) = wr[u];
        }
    };

    const int cc_begin = split * nch / tg.S, cc_end = (split + 1) * nch / tg.S;
    for (int cc = cc_begin; cc < cc_end; cc++) {
        const int c0 = cc * 32;
        const float *src;
        int Cs, cofs;
        if (c0 < g.C1) {
            src = a1; Cs = g.C1; cofs = c0;
        } else {
            src = a2; Cs = g.C2; cofs = c0 - g.C1;
        }
        load_w(cc, 0);
        __syncthreads();  // B1: every wave is done with the previous chunk's LDS
        for (int base = 0; base < XR; base += XB) {
            float4 v[XB];
#pragma unroll
            for (int u = 0; u < XB; u++) {
                const int idx = (base + u) * 256 + tid;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (base + u < XR && idx < nx && !(tg.dbg & 1)) {
                    const int slot = idx >> 3;
                    const int ez = (slot * tg.magHW) >> 16, rem = slot - ez * EHW;
                    const int ey = (rem * tg.magW) >> 16, ex = rem - ey * tg.EW;
                    const int id = iz0 + ez, ih = iy0 + ey, iw = ix0 + ex;
                    if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                        v[u] = *reinterpret_cast<const float4 *>(
                            src + ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * Cs + cofs + (tid & 7) * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < XB; u++) {
                const int idx = (base + u) * 256 + tid;
                if (base + u < XR) *reinterpret_cast<float4 *>(Xs + (size_t)(idx >> 3) * XS + (idx & 7) * 4) = v[u];
            }
        }
        for (int gi = 0; gi < ngroups; gi++) {
            store_w(gi & 1);
            __syncthreads();  // B2: halo tile (gi == 0) and weight group gi visible; group gi-2's buffer is free
            if (gi + 1 < ngroups) load_w(cc, gi + 1);  // in flight during this group's MFMAs
            if (!(tg.dbg & 2)) {
                const float *wb_ = Wsm + (size_t)(gi & 1) * WBUF;
#pragma unroll
                for (int tl = 0; tl < TG; tl++) {
                    const int t = gi * TG + tl;
                    if (t < g.ntaps) {  // block-uniform
                        const int to = __builtin_amdgcn_readlane(toff_l, t);
                        float4 af[MT][4], bf[NT][4];
#pragma unroll
                        for (int m = 0; m < MT; m++) {
                            const float4 *pa = reinterpret_cast<const float4 *>(Xs + (size_t)(sbase[m] + to) * XS + h * 16);
#pragma unroll
                            for (int e = 0; e < 4; e++) af[m][e] = pa[e];
                        }
#pragma unroll
                        for (int q = 0; q < NT; q++) {
                            const float4 *pb = reinterpret_cast<const float4 *>(wb_ + ((size_t)(tl * 2 + h) * KT + q * 32 + i) * WS);
#pragma unroll
                            for (int e = 0; e < 4; e++) bf[q][e] = pb[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) {
#pragma unroll
                            for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
                                for (int m = 0; m < MT; m++)
#pragma unroll
                                    for (int q = 0; q < NT; q++) {
                                        const float avv = c4 == 0 ? af[m][e].x : (c4 == 1 ? af[m][e].y : (c4 == 2 ? af[m][e].z : af[m][e].w));
                                        const float bvv = c4 == 0 ? bf[q][e].x : (c4 == 1 ? bf[q][e].y : (c4 == 2 ? bf[q][e].z : bf[q][e].w));
                                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv, bvv, acc[m][q], 0, 0, 0);
                                    }
                        }
                    }
                }
            }
        }
    }
    // epilogue
    const int od = od0 + wave;
    if (od >= g.Do) return;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < NT; q++) {
            const int k = kb * KT + q * 32 + i;
            const float bv = settled((bias && tg.S == 1) ? bias[k] : 0.f);
            store_tile32(g, acc[m][q], bv, n, od, oh0 + 4 * m, ow0, k, h, tg.S, split, tg.K, part, y1, y2);
        }
}

// ------------------------------------------------------------------------------------------------ stride-2 CK = 32 kernel
// 3x3x3 stride-2 convolutions (the first conv of encoder stages 1..5).  A stride-2 output tile touches an input
// footprint 8x its own size and every staged voxel feeds only 27/8 taps on average, so the direct form is the right
// one (no Winograd) and the footprint has to be staged as whole 128-byte lines: the 8-channel-chunk kernel this
// replaces re-fetched every line four times and ran at 36-45 TFLOP/s.
//   * workgroup = 4 waves = a 2 x 4 x 8 output tile x 64 output channels; wave w: output plane w & 1, channel half
//     w >> 1 (one 32 x 32 accumulator tile);
//   * per 32-channel chunk the 5 x 9 x 17-slot footprint [765 slots][32 ch] is staged once (110 KB with the 36-float
//     pad: one workgroup per CU);
//   * the packed weights are read straight from L2 one tap ahead (16 floats per lane and tap), so the tap loop has
//     no barrier.
//   * eight waves: waves 4..7 take the second half of the 27 taps on the same tile (two waves per SIMD hide each
//     other's LDS / L2 waits and the single accumulator chain); their tiles are added through LDS at the end.
constexpr int SXR = 12;  // float4 per thread: 765 slots x 8 / 512
constexpr int SXB = 6;   // staging batch of the generic path (loads in flight per thread)
__global__ __launch_bounds__(512, 1) void k_fwd32s(const FwdGeom g, const Fwd32Tile tg, const float *__restrict__ a1,
                                                   const float *__restrict__ a2, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ y1,
                                                   float *__restrict__ y2) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];
    constexpr int XS = 36;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= tg.nitems) return;
    unsigned r_ = (unsigned)item;
    const int kb = (int)(r_ % (unsigned)tg.nkb); r_ /= (unsigned)tg.nkb;   // 64-channel blocks
    const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
    const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
    const int td_ = (int)(r_ % (unsigned)tg.ntd);
    const int n = (int)(r_ / (unsigned)tg.ntd);

    const int C = g.C1 + g.C2;
    const int nch = C >> 5;
    const int EHW = tg.EH * tg.EW;
    const int nx = tg.nslots * 8;
    const int od0 = td_ * 2, oh0 = th_ * 4, ow0 = tw_ * 8;
    const int iz0 = od0 * 2 + tg.min_off[0], iy0 = oh0 * 2 + tg.min_off[1], ix0 = ow0 * 2 + tg.min_off[2];
    const int dl = wave & 1, kh = (wave >> 1) & 1, tap_half = wave >> 2;
    const int tb = tap_half ? 14 : 0, nt = tap_half ? g.ntaps - 14 : 14;  // host: ntaps == 27
    const int sbase = ((2 * dl) * tg.EH + 2 * (i >> 3)) * tg.EW + 2 * (i & 7);
    const float4 *xlane = reinterpret_cast<const float4 *>(Xs + (size_t)sbase * XS + h * 16);
    const int kcol = kb * 64 + kh * 32 + i;
    // packed weights: [cc][t][h][k][16]
    const float *wlane = w + (((size_t)h * tg.K + kcol) << 4);
    const size_t wtap = (size_t)2 * tg.K * 16;

    // division-free staging for the 5 x 9 x 17 footprint (every 3x3x3 stride-2 conv): 408 threads cover three 17-slot
    // rows per pass; pass q holds plane q / 3 and rows 3 * (q % 3) + r3 (VALU work is paid in MFMA issue cycles)
    const bool fastst = tg.EH == 9 && tg.EW == 17 && tg.nslots == 765;
    const int r3 = tid / 136, rem = tid - r3 * 136;
    const int sx = rem >> 3, part = rem & 7;
    const bool st_act = tid < 408;
    const int iw = ix0 + sx;
    const bool okw = st_act && iw >= 0 && iw < g.Wi;
    bool okr[3];
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const int ih = iy0 + r3 + 3 * v;
        okr[v] = okw && ih >= 0 && ih < g.Hi;
    }
    float4 *lds_st = reinterpret_cast<float4 *>(Xs + (size_t)(r3 * 17 + sx) * XS + part * 4);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    // per-tap tables across the lanes, read back with v_readlane (see k_fwd32)
    const int lt = lane < g.ntaps ? lane : g.ntaps - 1;
    const int toff_l = tg.toff[lt < 27 ? lt : 26], wt_l = g.wt[lt < 27 ? lt : 26];

    for (int cc = 0; cc < nch; cc++) {
        const int c0 = cc * 32;
        const float *src;
        int Cs, cofs;
        if (c0 < g.C1) {
            src = a1; Cs = g.C1; cofs = c0;
        } else {
            src = a2; Cs = g.C2; cofs = c0 - g.C1;
        }
        const float *wc = wlane + (size_t)cc * g.T * wtap;
        // weights run three taps ahead of the MFMAs (4-deep register ring)
        float4 wb[4][4];
#pragma unroll
        for (int u = 0; u < 3; u++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                wb[u][e] = *reinterpret_cast<const float4 *>(wc + (size_t)__builtin_amdgcn_readlane(wt_l, tb + u) * wtap + e * 4);
        __syncthreads();
        if (fastst) {
            const unsigned off0 = (unsigned)(((iy0 + r3) * g.Wi + iw) * Cs + cofs + part * 4) << 2;  // bytes, used when ok
            const unsigned drow = (unsigned)(3 * g.Wi * Cs) << 2;
            // all 15 loads of a thread in ONE round trip, as unconditional buffer loads: a plane outside the volume gets a
            // zero-record descriptor, a lane outside it an out-of-range offset -- both read zeros.  (Predicated loads kept
            // the staging at three serialised batches of five: a branch around a load makes the compiler wait for the
            // loads before it.)
            float4 v[15];
#pragma unroll
            for (int pl = 0; pl < 5; pl++) {
                const int id = iz0 + pl;  // plane: wave-uniform
                const bool inpl = id >= 0 && id < g.Di;
                const float *plane = src + ((size_t)n * g.Di + (inpl ? id : 0)) * g.Hi * g.Wi * Cs;
                const __amdgpu_buffer_rsrc_t rp =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(plane), 0, inpl ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const unsigned o = okr[r] ? off0 + (unsigned)r * drow : 0xffffffffu;
                    v[pl * 3 + r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)o, 0, 0));
                }
            }
#pragma unroll
            for (int pq = 0; pq < 15; pq++)
                if (st_act) lds_st[((pq / 3) * 9 + 3 * (pq % 3)) * 17 * (XS / 4)] = v[pq];
        } else {
            int tid_ = tid;
            asm volatile("" : "+v"(tid_));
            for (int base = 0; base < SXR; base += SXB) {
                float4 v[SXB];
#pragma unroll
                for (int q = 0; q < SXB; q++) {
                    const int idx = (base + q) * 512 + tid_;
                    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < nx) {
                        const int slot = idx >> 3;
                        const int ez = slot / EHW, rem2 = slot - ez * EHW;
                        const int ey = (rem2 * tg.magW) >> 16, ex = rem2 - ey * tg.EW;
                        const int id = iz0 + ez, ih = iy0 + ey, iw2 = ix0 + ex;
                        if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw2 >= 0 && iw2 < g.Wi)
                            v[q] = *reinterpret_cast<const float4 *>(
                                src + ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw2) * Cs + cofs + (tid_ & 7) * 4);
                    }
                }
#pragma unroll
                for (int q = 0; q < SXB; q++) {
                    const int idx = (base + q) * 512 + tid_;
                    if (idx < nx) *reinterpret_cast<float4 *>(Xs + (size_t)(idx >> 3) * XS + (idx & 7) * 4) = v[q];
                }
            }
        }
        __syncthreads();
        // activations of tap t+1 are read from LDS into the other half of af[] before tap t's MFMAs are issued: a register
        // an in-flight MFMA reads is never the target of the next tap's ds_read
        float4 af[2][4];
        {
            const float4 *pa = xlane + (size_t)__builtin_amdgcn_readlane(toff_l, tb) * (XS / 4);
#pragma unroll
            for (int e = 0; e < 4; e++) af[0][e] = pa[e];
        }
        for (int s0 = 0; s0 < nt; s0 += 4) {  // four taps per trip: static ring indices
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int sidx = s0 + u;
                if (sidx < nt) {  // uniform
                    const int t = tb + sidx;
                    const int tn = sidx + 3 < nt ? t + 3 : tb + nt - 1;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        wb[(u + 3) & 3][e] =
                            *reinterpret_cast<const float4 *>(wc + (size_t)__builtin_amdgcn_readlane(wt_l, tn) * wtap + e * 4);
                    const float4 *pa = xlane + (size_t)__builtin_amdgcn_readlane(toff_l, sidx + 1 < nt ? t + 1 : tb + nt - 1) * (XS / 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) af[(u + 1) & 1][e] = pa[e];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][e].x, wb[u][e].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][e].y, wb[u][e].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][e].z, wb[u][e].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][e].w, wb[u][e].w, acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // waves 4..7 hand their tap half over through LDS (the halo is dead now)
    __syncthreads();
    if (tap_half) {
        float *xo = Xs + (size_t)(wave & 3) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; r++) xo[r * 64] = acc[r];
    }
    __syncthreads();
    if (tap_half) return;
    {
        const float *xi = Xs + (size_t)wave * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] += xi[r * 64];
    }
    const int od = od0 + dl;
    if (od >= g.Do) return;
    const float bv = settled(bias ? bias[kcol] : 0.f);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int oh = oh0 + (row >> 3), ow = ow0 + (row & 7);
        if (oh < g.Ho && ow < g.Wo) {
            const size_t ov = (((size_t)n * g.Dy + od) * g.Hy + oh) * g.Wy + ow;
            const float val = acc[r] + bv;
            if (kcol < g.K1)
                y1[ov * g.K1 + kcol] = val;
            else
                y2[ov * g.K2 + (kcol - g.K1)] = val;
        }
    }
}

// ------------------------------------------------------------------------------------------------ stride-2 input gradient
// dx of a 3x3x3 stride-2 pad-1 conv: dx[i] = sum over (o, t) with 2 o + t - 1 = i of dy[o] W[t].  Per axis an even i = 2 j
// meets only (t = 1, o = j); an odd i = 2 j + 1 meets (t = 0, o = j + 1) and (t = 2, o = j): eight parity classes with
// 1 .. 8 taps.  The per-class launches of the gather engine re-stage the dy tile eight times (the 1- and 2-tap classes
// for 16-32 MFMAs per stage); here ONE workgroup stages a 3 x 5 x 9 dy tile per 32-channel chunk (19 KB) and its eight
// waves each take one class (two 32-voxel M tiles: 2 x 4 x 8 positions j).  The classes are unbalanced (1 .. 8 taps), but
// the tile is small enough for several workgroups per CU, so other workgroups' waves fill the pipe.
// Weights: the packed dgrad layout [kk][t][h][c][16] straight from L2, one tap ahead.
struct Dg2Tile {
    int ntd, nth, ntw, ncb, nitems, C;
    int acc;  // dx += (the skip connection's second gradient contribution, ops._GradShare)
};
__global__ __launch_bounds__(512, 2) void k_dgrad32s(const FwdGeom g, const Dg2Tile tg, const float *__restrict__ dy,
                                                     const float *__restrict__ wb, float *__restrict__ dx) {
    __shared__ __attribute__((aligned(16))) float Xs[135 * 36];
    constexpr int XS = 36;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= tg.nitems) return;
    unsigned r_ = (unsigned)item;
    const int cb = (int)(r_ % (unsigned)tg.ncb); r_ /= (unsigned)tg.ncb;
    const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
    const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
    const int td_ = (int)(r_ % (unsigned)tg.ntd);
    const int n = (int)(r_ / (unsigned)tg.ntd);
    // here g.Di/Hi/Wi = dy grid (conv output), g.Dy/Hy/Wy = dx grid (conv input), g.C1 = K (reduce), tg.C = channels of dx
    const int K = g.C1, C = tg.C;
    const int nch = K >> 5;
    const int oz0 = td_ * 2, oy0 = th_ * 4, ox0 = tw_ * 8;
    // balance: wave w takes class w on M tile 0 (plane oz0) and the complementary class 7 - w on M tile 1 (plane oz0 + 1):
    // 2^a + 2^(3-a) = 6 or 9 tap groups per wave instead of 1 .. 8
    const int ccol = cb * 32 + i;
    const float *wlane = wb + (((size_t)h * C + ccol) << 4);
    const size_t wtap = (size_t)2 * C * 16;

    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;

    for (int kk = 0; kk < nch; kk++) {
        __syncthreads();
        {
            // the thread's (up to three) 16-byte pieces of the 3 x 5 x 9 dy tile in ONE round trip: unconditional buffer
            // loads, out-of-range offset (-> zeros) for voxels outside the volume and for pieces past the tile.  One
            // predicated load per loop trip, stored before the next was issued, cost a round trip per piece.
            const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(dy + (size_t)n * g.Di * g.Hi * g.Wi * K + kk * 32), 0, 0x7fffffff, 0x00020000);
            float4 v[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int idx = tid + u * 512;
                const int slot = idx >> 3, part = idx & 7;
                const int ez = slot / 45, rem = slot - ez * 45;
                const int ey = rem / 9, ex = rem - ey * 9;
                const int oz = oz0 + ez, oy = oy0 + ey, ox = ox0 + ex;
                const bool ok = idx < 135 * 8 && oz < g.Di && oy < g.Hi && ox < g.Wi;
                const unsigned o = ok ? (unsigned)(((oz * g.Hi + oy) * g.Wi + ox) * K + part * 4) * 4u : 0xffffffffu;
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (int)o, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int idx = tid + u * 512;
                if (idx < 135 * 8) *reinterpret_cast<float4 *>(Xs + (size_t)(idx >> 3) * XS + (idx & 7) * 4) = v[u];
            }
        }
        __syncthreads();
        const float *wc = wlane + (size_t)kk * 27 * wtap;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int q = m ? 7 - wave : wave;
            const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
            const float4 *xm = reinterpret_cast<const float4 *>(Xs + (size_t)((m * 5 + (i >> 3)) * 9 + (i & 7)) * XS + h * 16);
            // the class's taps: per axis (t = 1, d = 0) for an even coordinate, (t = 0, d = 1) and (t = 2, d = 0) for an odd
            // one; tap j of the 2^popcount(q) taps takes one bit of j per odd axis (z first).  Weights one tap ahead.
            const int ntq = 1 << (pz + py + px);
            auto tap_of = [&](int j, int &t, int &so) {
                const int bz = pz ? (j >> (py + px)) & 1 : 0, by = py ? (j >> px) & 1 : 0, bx = px ? j & 1 : 0;
                const int tz = pz ? (bz ? 2 : 0) : 1, ty = py ? (by ? 2 : 0) : 1, tx = px ? (bx ? 2 : 0) : 1;
                const int dz = (pz && !bz) ? 1 : 0, dyy = (py && !by) ? 1 : 0, dxx = (px && !bx) ? 1 : 0;
                t = (tz * 3 + ty) * 3 + tx;
                so = (dz * 5 + dyy) * 9 + dxx;
            };
            float4 wv[2][4];
            int t0, so0;
            tap_of(0, t0, so0);
#pragma unroll
            for (int e = 0; e < 4; e++) wv[0][e] = *reinterpret_cast<const float4 *>(wc + (size_t)t0 * wtap + e * 4);
            for (int j = 0; j < ntq; j += 2) {  // two taps per trip: static buffer indices (ntq is 1, 2, 4 or 8)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (j + u < ntq) {  // wave-uniform
                        int t, so, tn, son;
                        tap_of(j + u, t, so);
                        tap_of(j + u + 1 < ntq ? j + u + 1 : j + u, tn, son);
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            wv[(u + 1) & 1][e] = *reinterpret_cast<const float4 *>(wc + (size_t)tn * wtap + e * 4);
                        float4 af[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) af[e] = xm[so * (XS / 4) + e];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e].x, wv[u][e].x, acc[m], 0, 0, 0);
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e].y, wv[u][e].y, acc[m], 0, 0, 0);
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e].z, wv[u][e].z, acc[m], 0, 0, 0);
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e].w, wv[u][e].w, acc[m], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // dx voxel of accumulator row r (lane half h) of M tile m: j = (oz0 + m, oy0 + (r >> 2), ox0 + (r & 3) + 4 h), i = 2 j + p
    float *xl = dx + (size_t)(2 * 4 * h) * C + ccol;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int q = m ? 7 - wave : wave;
        const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
        const int iz = 2 * (oz0 + m) + pz;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int iy = 2 * (oy0 + (r >> 2)) + py, ixu = 2 * (ox0 + (r & 3)) + px;  // wave-uniform
            const size_t uo = ((((size_t)n * g.Dy + iz) * g.Hy + iy) * g.Wy + ixu) * C;
            if (iz < g.Dy && iy < g.Hy && ixu + 8 * h < g.Wy) xl[uo] = tg.acc ? xl[uo] + acc[m][r] : acc[m][r];
        }
    }
}

int dgrad32s(int N, int D, int H, int W, int C, int K, int Do, int Ho, int Wo, const float *dy, const float *wb, float *dx,
             hipStream_t s, int accumulate) {
    if (C % 32 || K % 32 || (((uintptr_t)dy | (uintptr_t)wb) & 15)) return -1;
    if ((long)Do * Ho * Wo * K * 4 >= (1L << 31)) return -1;  // 32-bit byte offsets inside one sample of dy (buffer loads)
    FwdGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = Do; g.Hi = Ho; g.Wi = Wo;
    g.Dy = D; g.Hy = H; g.Wy = W;
    g.C1 = K;
    Dg2Tile tg;
    tg.ntd = (Do + 1) / 2; tg.nth = (Ho + 3) / 4; tg.ntw = (Wo + 7) / 8;
    tg.ncb = C / 32;
    tg.C = C;
    tg.acc = accumulate;
    const long nitems = (long)N * tg.ntd * tg.nth * tg.ntw * tg.ncb;
    if (nitems > (1L << 30)) return -1;
    tg.nitems = (int)nitems;
    const unsigned grid = (unsigned)(((nitems + 7) / 8) * 8);
    hipLaunchKernelGGL(k_dgrad32s, dim3(grid), dim3(512), 0, s, g, tg, dy, wb, dx);
    return check_launch("conv dgrad (stride 2, fused parity classes)");
}

static int num_cus();
static int launch_fwd32s(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1,
                         float *y2, hipStream_t s) {
    const int K = g.K1 + g.K2;
    Fwd32Tile tg;
    memset(&tg, 0, sizeof(tg));
    int mn[3] = {127, 127, 127}, mx[3] = {-127, -127, -127};
    for (int t = 0; t < g.ntaps; t++)
        for (int a = 0; a < 3; a++) {
            if (g.off[t][a] < mn[a]) mn[a] = g.off[t][a];
            if (g.off[t][a] > mx[a]) mx[a] = g.off[t][a];
        }
    const int T3[3] = {2, 4, 8};
    int E[3];
    for (int a = 0; a < 3; a++) {
        E[a] = (T3[a] - 1) * 2 + (mx[a] - mn[a]) + 1;
        tg.min_off[a] = mn[a];
    }
    tg.EH = E[1]; tg.EW = E[2];
    tg.nslots = E[0] * E[1] * E[2];
    if (tg.nslots > SXR * 64 || g.ntaps != 27) return -1;
    int m = (1 << 16) / tg.EW + 1;
    for (int nn = 0; nn < tg.EH * tg.EW; nn++)
        if (((nn * m) >> 16) != nn / tg.EW) return -1;
    tg.magW = m;
    for (int t = 0; t < g.ntaps; t++)
        tg.toff[t] = ((g.off[t][0] - mn[0]) * tg.EH + (g.off[t][1] - mn[1])) * tg.EW + (g.off[t][2] - mn[2]);
    tg.ntd = (g.Do + 1) / 2;
    tg.nth = (g.Ho + 3) / 4;
    tg.ntw = (g.Wo + 7) / 8;
    tg.nkb = K / 64;
    tg.K = K;
    const long nitems = (long)g.N * tg.ntd * tg.nth * tg.ntw * tg.nkb;
    if (nitems > (1L << 30)) return -1;
    tg.nitems = (int)nitems;
    const size_t lds = (size_t)tg.nslots * 36 * sizeof(float);
    static PerDeviceFlag cfgd;
    if (!cfgd()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd32s), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_LIMIT) != hipSuccess) {
            set_error("conv fwd (stride 2): cannot raise the dynamic LDS limit");
            return 1;
        }
        cfgd() = true;
    }
    const unsigned grid = (unsigned)(((nitems + 7) / 8) * 8);
    hipLaunchKernelGGL(k_fwd32s, dim3(grid), dim3(512), lds, s, g, tg, a1, a2, w, bias, y1, y2);
    return check_launch("conv fwd (stride 2, CK=32)");
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static int split_for(const void *ws, size_t ws_bytes, long wgs, int nch, size_t out_elems, int nkb) {
    int S = 1;
    if (ws && wgs < 256 && nch >= 4) {
        long want = 512 / wgs;  // two workgroups per CU resident: never spill into a second, mostly empty round
        if (want > MAX_SPLIT) want = MAX_SPLIT;
        if (want > nch / 2) want = nch / 2;
        while (want > 1 && (size_t)want * out_elems * sizeof(float) > ws_bytes) want--;
        if (want > 1 && (long)nkb * want <= 65535) S = (int)want;
    }
    return S;
}

static int run_split_reduce(const FwdGeom &g, const float *part, const float *bias, float *y1, float *y2, int S,
                            size_t out_elems, hipStream_t s) {
    long blocks = cdiv((long)out_elems, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_split_reduce, dim3(blocks), dim3(256), 0, s, g, part, bias, y1, y2, S);
    return check_launch("conv fwd split reduce");
}

template <int NT, int MT, int TG>
static int launch_fwd32(const FwdGeom &g, Fwd32Tile &tg, const float *a1, const float *a2, const float *w,
                        const float *bias, float *y1, float *y2, void *ws, size_t ws_bytes, hipStream_t s) {
    auto kern = k_fwd32<NT, MT, TG>;
    constexpr int XR = MT == 2 ? 19 : 12;
    const size_t lds = ((size_t)XR * 32 * 36 + 2 * (size_t)TG * 2 * (32 * NT) * 20) * 4;
    static PerDeviceFlag configured;
    if (!configured()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_LIMIT) != hipSuccess) {
            set_error("conv fwd32 (mfma): cannot raise the dynamic LDS limit");
            return 1;
        }
        configured() = true;
    }
    const int K = g.K1 + g.K2;
    tg.nkb = K / (32 * NT);
    const long tiles = (long)tg.ntd * tg.nth * tg.ntw;
    const size_t out_elems = (size_t)g.N * g.Do * g.Ho * g.Wo * K;
    tg.S = split_for(ws, ws_bytes, tiles * tg.nkb * g.N, (g.C1 + g.C2) / 32, out_elems, tg.nkb);
    const long nitems = tiles * tg.nkb * tg.S * g.N;
    if (nitems > (1L << 30)) return -1;
    tg.nitems = (int)nitems;
    const long grid = ((nitems + 7) / 8) * 8;
    float *part = reinterpret_cast<float *>(ws);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, g, tg, a1, a2, w, bias, y1, y2, part);
    if (check_launch("conv fwd32 (mfma)")) return 1;
    if (tg.S > 1) return run_split_reduce(g, part, bias, y1, y2, tg.S, out_elems, s);
    return 0;
}

template <int CK, int NT, int MT, bool L32>
static int launch_fwd(const FwdGeom &g, FwdTile &tg, size_t lds, const float *a1, const float *a2, const float *w,
                      const float *bias, float *y1, float *y2, void *ws, size_t ws_bytes, hipStream_t s) {
    auto kern = k_fwd_mfma<CK, NT, MT, L32>;
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_LIMIT) != hipSuccess) {
            set_error("conv fwd (mfma): cannot raise the dynamic LDS limit");
            return 1;
        }
        configured = LDS_LIMIT;
    }
    const int K = g.K1 + g.K2;
    const int nkb = K / (32 * NT);
    const long tiles = (long)tg.ntd * tg.nth * tg.ntw;
    const size_t out_elems = (size_t)g.N * g.Do * g.Ho * g.Wo * K;
    // split the reduce chunks when the launch cannot fill the chip (8^3 / 4^3 stages with 320-640 channels)
    const int S = split_for(ws, ws_bytes, tiles * nkb * g.N, (g.C1 + g.C2) / CK, out_elems, nkb);
    dim3 grid((unsigned)tiles, nkb * S, g.N);
    float *part = reinterpret_cast<float *>(ws);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g, tg, a1, a2, w, bias, y1, y2, S, part);
    if (check_launch("conv fwd (mfma)")) return 1;
    if (S > 1) return run_split_reduce(g, part, bias, y1, y2, S, out_elems, s);
    return 0;
}

int fwd_mfma(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1, float *y2,
             void *ws, size_t ws_bytes, hipStream_t s) {
    const int C = g.C1 + g.C2, K = g.K1 + g.K2;
    const int LCK = wl_ck(C);  // chunk size of the packed weight layout
    if (LCK == 0 || g.ntaps < 1 || g.ntaps > 27) return -1;
    if (g.C1 % LCK != 0 || g.C2 % LCK != 0) return -1;          // a chunk must not straddle the two input pointers
    if (K % 32 != 0 || g.K1 % 32 != 0 || g.K2 % 32 != 0) return -1;  // nor a 32-wide N tile the two outputs
    if (g.N > 65535 || K / 32 > 65535) return -1;
    if (((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)w) & 15) return -1;
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("MVD_CONV_DBG") ? atoi(getenv("MVD_CONV_DBG")) : 0;
    int mn[3] = {127, 127, 127}, mx[3] = {-127, -127, -127};
    for (int t = 0; t < g.ntaps; t++)
        for (int a = 0; a < 3; a++) {
            if (g.off[t][a] < mn[a]) mn[a] = g.off[t][a];
            if (g.off[t][a] > mx[a]) mx[a] = g.off[t][a];
        }
    const bool unit_stride = g.sa[0] == 1 && g.sa[1] == 1 && g.sa[2] == 1;
    if (LCK == 32 && unit_stride && !(dbg & 4)) {
        // ---- CK = 32 kernel: 4x8x8 tile (one workgroup per CU) or 4x4x8 tile (two per CU)
        // a 64-wide N tile may span the two output pointers: each 32-wide half lies in one of them (K1 % 32 == 0)
        static int nt_rule = -1;  // MVD_CONV_NT: 0 = 64-wide N tile whenever K % 64 == 0, 1 (default) = narrow tile for skinny problems
        if (nt_rule < 0) nt_rule = getenv("MVD_CONV_NT") ? atoi(getenv("MVD_CONV_NT")) : 1;
        // skinny problems (8^3 / 4^3 stages): fewer than one workgroup per CU even with the channel split -- the 32-wide N
        // tile doubles the workgroup count and halves the serial MFMA chain of each
        const long wg64 = (long)g.N * ((g.Do + 3) / 4) * ((g.Ho + 3) / 4) * ((g.Wo + 7) / 8) * (K / 64);
        const int NT = (K % 64 == 0 && !(nt_rule == 1 && wg64 < 128)) ? 2 : 1;
        static int force_mt = -1;
        if (force_mt < 0) force_mt = getenv("MVD_CONV_MT") ? atoi(getenv("MVD_CONV_MT")) : 0;
        static int force_tg = -1;
        if (force_tg < 0) force_tg = getenv("MVD_CONV_TG") ? atoi(getenv("MVD_CONV_TG")) : 0;
        const int MT = force_mt ? force_mt : 1;
        Fwd32Tile t32;
        memset(&t32, 0, sizeof(t32));
        t32.dbg = dbg;
        const int T3[3] = {4, 4 * MT, 8};
        int E[3];
        for (int a = 0; a < 3; a++) {
            E[a] = (T3[a] - 1) + (mx[a] - mn[a]) + 1;
            t32.min_off[a] = mn[a];
        }
        t32.EH = E[1]; t32.EW = E[2];
        t32.nslots = E[0] * E[1] * E[2];
        auto magic = [](int d, int nmax) -> int {
            int m = (1 << 16) / d + 1;
            for (int n = 0; n < nmax; n++)
                if (((n * m) >> 16) != n / d) return -1;
            return m;
        };
        t32.magHW = magic(t32.EH * t32.EW, t32.nslots);
        t32.magW = magic(t32.EW, t32.EH * t32.EW);
        if (t32.nslots <= (MT == 2 ? 19 : 12) * 32 && t32.magHW > 0 && t32.magW > 0) {
            for (int t = 0; t < g.ntaps; t++)
                t32.toff[t] = ((g.off[t][0] - mn[0]) * t32.EH + (g.off[t][1] - mn[1])) * t32.EW + (g.off[t][2] - mn[2]);
            t32.ntd = (g.Do + 3) / 4;
            t32.nth = (g.Ho + 4 * MT - 1) / (4 * MT);
            t32.ntw = (g.Wo + 7) / 8;
            t32.K = K;
            if (MT == 2) {
                if (NT == 2) return launch_fwd32<2, 2, 3>(g, t32, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
                return launch_fwd32<1, 2, 3>(g, t32, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
            }
            if (NT == 2) return launch_fwd32<2, 1, 1>(g, t32, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
            if (force_tg == 3) return launch_fwd32<1, 1, 3>(g, t32, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
            if (force_tg == 1) return launch_fwd32<1, 1, 1>(g, t32, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
            return launch_fwd32<1, 1, 2>(g, t32, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
        }
    }
    if (LCK == 32 && g.sa[0] == 2 && g.sa[1] == 2 && g.sa[2] == 2 && g.ntaps == 27 && g.T == 27 && K % 64 == 0 &&
        g.K1 % 64 == 0 && g.so[0] == 1 && g.so[1] == 1 && g.so[2] == 1 && g.oo[0] == 0 && g.oo[1] == 0 && g.oo[2] == 0 &&
        !(dbg & 8)) {
        const long items = (long)g.N * ((g.Do + 1) / 2) * ((g.Ho + 3) / 4) * ((g.Wo + 7) / 8) * (K / 64);
        if (items >= 256) {  // enough workgroups for the chip; the small stages stay on the chunked kernel
            int r = launch_fwd32s(g, a1, a2, w, bias, y1, y2, s);
            if (r >= 0) return r;
        }
    }
    // ---- chunked kernel (8- or 4-channel chunks; 8-channel sub-chunks of the 32-layout when LCK == 32)
    const int CK = LCK == 32 ? 8 : LCK;
    FwdTile tg;
    memset(&tg, 0, sizeof(tg));
    tg.dbg = dbg;
    tg.K = K;
    // candidate tile shapes, widest first
    const int cand[3][2] = {{2, 2}, {2, 1}, {1, 1}};  // (MT, NT)
    for (int ci = 0; ci < 3; ci++) {
        const int MT = cand[ci][0], NT = cand[ci][1];
        if (K % (32 * NT) != 0) continue;
        const int T3[3] = {4, 4 * MT, 8};
        int E[3];
        for (int a = 0; a < 3; a++) {
            E[a] = (T3[a] - 1) * g.sa[a] + (mx[a] - mn[a]) + 1;
            tg.min_off[a] = mn[a];
        }
        tg.ED = E[0]; tg.EH = E[1]; tg.EW = E[2];
        tg.nslots = E[0] * E[1] * E[2];
        const size_t lds = ((size_t)tg.nslots * CK + (size_t)g.ntaps * 2 * (32 * NT) * (CK / 2)) * 4 + 64 * 4;
        // prefer configurations that leave room for >= 2 workgroups per CU; accept one per CU for the widest halos
        const bool fits2 = lds <= LDS_LIMIT / 2;
        const bool last = (ci == 2);
        if (!fits2 && !(last && lds <= LDS_LIMIT)) {
            if (!last) continue;
            return -1;
        }
        for (int t = 0; t < g.ntaps; t++)
            tg.toff[t] = ((g.off[t][0] - mn[0]) * tg.EH + (g.off[t][1] - mn[1])) * tg.EW + (g.off[t][2] - mn[2]);
        tg.ntd = (g.Do + 3) / 4;
        tg.nth = (g.Ho + 4 * MT - 1) / (4 * MT);
        tg.ntw = (g.Wo + 7) / 8;
#define MVD_FWD_DISPATCH(CKV, L32V)                                                                                      \
    {                                                                                                                    \
        if (MT == 2 && NT == 2) return launch_fwd<CKV, 2, 2, L32V>(g, tg, lds, a1, a2, w, bias, y1, y2, ws, ws_bytes, s); \
        if (MT == 2 && NT == 1) return launch_fwd<CKV, 1, 2, L32V>(g, tg, lds, a1, a2, w, bias, y1, y2, ws, ws_bytes, s); \
        return launch_fwd<CKV, 1, 1, L32V>(g, tg, lds, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);                         \
    }
        if (LCK == 32) MVD_FWD_DISPATCH(8, true)
        else if (CK == 8) MVD_FWD_DISPATCH(8, false)
        else MVD_FWD_DISPATCH(4, false)
#undef MVD_FWD_DISPATCH
    }
    return -1;
}

// ================================================================================================ wgrad
struct WgTile {
    int TD, TH, TW, lTH, lTW;
    int EAh, EAw, nslotsA, minA[3];
    int EBh, EBw, nslotsB, minB[3];
    int magAw, magAhw, magBw, magBhw;  // 16-bit reciprocal multipliers: q = (n * mag) >> 16 (exact, host-verified)
    int ntd, nth, ntw, nsplit, nkb;
    int ntiles;
    int toffA[27], toffB[27];
    int dbg;
};

// NA / NB: float4 per thread of the A halo / B tile (512 threads: LDS regions are NA*8 KiB and NB*8 KiB).  The NEXT
// tile's A and B are prefetched into registers while the current tile's MFMAs run.
// Eight waves = two groups of four on the SAME staged tile: wave & 3 picks the taps, group wave >> 2 takes every other
// pair of k-steps and writes its own split-K partial (split * 2 + group) -- two waves per SIMD cover each other's LDS
// round trips and the tile is staged once for twice the MFMAs.
// SH: 1 = every tap reads the same B slot (3x3x3 conv: B = dy), 2 = the same A slot (transposed conv), 0 = neither
template <int TPW, int NA, int NB, int SH>
__global__ __launch_bounds__(512, 1) void k_wgrad_mfma(const WgradGeom g, const WgTile tg, const float *__restrict__ a1,
                                                       const float *__restrict__ a2, const float *__restrict__ b,
                                                       float *__restrict__ partial, float *__restrict__ pbias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    float *Bs = lds + (size_t)NA * 2048;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y / tg.nkb, kb = blockIdx.y % tg.nkb;
    const int split = blockIdx.x;
    const int C = g.C1 + g.C2, K = g.K;

    int ta[TPW], tb[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        ta[j] = tb[j] = 0;  // taps beyond ntaps alias tap offset 0: their MFMAs run (branch-free) and are discarded
#pragma unroll
        for (int t = 0; t < 27; t++)
            if (t == wave + 4 * j && t < g.ntaps) {
                ta[j] = tg.toffA[t];
                tb[j] = tg.toffB[t];
            }
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    // channel block -> source pointer (a 32-block never straddles the two inputs: host checks C1 % 32 == 0 if C2 > 0)
    const int c0 = cb * 32;
    const float *asrc;
    int Cs, cofs;
    if (c0 < g.C1) {
        asrc = a1; Cs = g.C1; cofs = c0;
    } else {
        asrc = a2; Cs = g.C2; cofs = c0 - g.C1;
    }
    const int cvalid = (Cs - cofs) < 32 ? (Cs - cofs) : 32;  // partial block (e.g. the 4-channel input layer)
    const int k0 = kb * 32;
    const int kvalid = (K - k0) < 32 ? (K - k0) : 32;
    const int EAhw = tg.EAh * tg.EAw, EBhw = tg.EBh * tg.EBw;
    const int TV = tg.TD * tg.TH * tg.TW;
    const int na = tg.nslotsA * 8, nb = tg.nslotsB * 8;
    const int part = tid & 7;
    const bool aok = part * 4 < cvalid, bok = part * 4 < kvalid;

    float4 ra[NA], rb[NB];
    // Two LDS images and the next tile's loads as unconditional buffer loads with tile-independent lane offsets (the
    // form k_wgrad_wino2w12 took): one barrier per tile; no slot decode, bounds compares or 64-bit address
    // per load for interior tiles.  db = false (MVD_WGRAD_DB=0): the single image and load_tile() below.
    const bool db = !(tg.dbg & 16);
    unsigned relA[NA], relB[NB];  // byte offsets from the tile's first halo voxel (channel block and part folded in)
    int czA[NA], czB[NB];         // packed halo coordinates, -1 = this lane loads nothing for slot u
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int idx = u * 512 + tid, slot = idx >> 3;
        const int ez = (slot * tg.magAhw) >> 16, rem = slot - ez * EAhw;
        const int ey = (rem * tg.magAw) >> 16, ex = rem - ey * tg.EAw;
        relA[u] = (unsigned)(((ez * g.Hi + ey) * g.Wi + ex) * Cs + cofs + part * 4) * 4u;
        czA[u] = (aok && idx < na) ? ((ez << 16) | (ey << 8) | ex) : -1;
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int idx = u * 512 + tid, slot = idx >> 3;
        const int ez = (slot * tg.magBhw) >> 16, rem = slot - ez * EBhw;
        const int ey = (rem * tg.magBw) >> 16, ex = rem - ey * tg.EBw;
        relB[u] = (unsigned)(((ez * g.Hb + ey) * g.Wb + ex) * K + k0 + part * 4) * 4u;
        czB[u] = (bok && idx < nb) ? ((ez << 16) | (ey << 8) | ex) : -1;
    }
    const int EAd = tg.nslotsA / EAhw, EBd = tg.nslotsB / EBhw;  // halo extents along D
    auto load_tile_buf = [&](int tile) {  // tile >= ntiles: zero-record descriptors, no traffic
        const bool more = tile < tg.ntiles;
        unsigned r_ = (unsigned)(more ? tile : 0);
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * tg.TD, oh0 = th_ * tg.TH, ow0 = tw_ * tg.TW;
        const int za = od0 * g.sa[0] + tg.minA[0], ya = oh0 * g.sa[1] + tg.minA[1], xa = ow0 * g.sa[2] + tg.minA[2];
        const int zb = od0 * g.sb[0] + tg.minB[0], yb = oh0 * g.sb[1] + tg.minB[1], xb = ow0 * g.sb[2] + tg.minB[2];
        const bool intA = za >= 0 && za + EAd <= g.Di && ya >= 0 && ya + tg.EAh <= g.Hi && xa >= 0 && xa + tg.EAw <= g.Wi;
        const bool intB = zb >= 0 && zb + EBd <= g.Db && yb >= 0 && yb + tg.EBh <= g.Hb && xb >= 0 && xb + tg.EBw <= g.Wb;
        const float *baseA = asrc + ((((long)n * g.Di + za) * g.Hi + ya) * g.Wi + xa) * (long)Cs;
        const float *baseB = b + ((((long)n * g.Db + zb) * g.Hb + yb) * g.Wb + xb) * (long)K;
        const __amdgpu_buffer_rsrc_t rA =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(baseA), 0, more ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(baseB), 0, more ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < NA; u++) {
            unsigned o = czA[u] >= 0 ? relA[u] : 0xffffffffu;
            if (!intA) {  // block-uniform; VALU only inside
                const int id = za + (czA[u] >> 16), ih = ya + ((czA[u] >> 8) & 255), iw = xa + (czA[u] & 255);
                o = (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi) ? o : 0xffffffffu;
            }
            ra[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)o, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            unsigned o = czB[u] >= 0 ? relB[u] : 0xffffffffu;
            if (!intB) {
                const int id = zb + (czB[u] >> 16), ih = yb + ((czB[u] >> 8) & 255), iw = xb + (czB[u] & 255);
                o = (id >= 0 && id < g.Db && ih >= 0 && ih < g.Hb && iw >= 0 && iw < g.Wb) ? o : 0xffffffffu;
            }
            rb[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rB, (int)o, 0, 0));
        }
    };
    auto load_tile = [&](int tile) {
        unsigned r_ = (unsigned)tile;
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * tg.TD, oh0 = th_ * tg.TH, ow0 = tw_ * tg.TW;
        {
            const int z0 = od0 * g.sa[0] + tg.minA[0], y0 = oh0 * g.sa[1] + tg.minA[1], x0 = ow0 * g.sa[2] + tg.minA[2];
#pragma unroll
            for (int u = 0; u < NA; u++) {
                const int idx = u * 512 + tid;
                const int slot = idx >> 3;
                const int ez = (slot * tg.magAhw) >> 16, rem = slot - ez * EAhw;
                const int ey = (rem * tg.magAw) >> 16, ex = rem - ey * tg.EAw;
                const int id = z0 + ez, ih = y0 + ey, iw = x0 + ex;
                ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (aok && idx < na && !(tg.dbg & 1) && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi) {
                    const size_t e = ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * Cs + cofs + part * 4;
                    ra[u] = *reinterpret_cast<const float4 *>(asrc + e);
                }
            }
        }
        {
            const int z0 = od0 * g.sb[0] + tg.minB[0], y0 = oh0 * g.sb[1] + tg.minB[1], x0 = ow0 * g.sb[2] + tg.minB[2];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int idx = u * 512 + tid;
                const int slot = idx >> 3;
                const int ez = (slot * tg.magBhw) >> 16, rem = slot - ez * EBhw;
                const int ey = (rem * tg.magBw) >> 16, ex = rem - ey * tg.EBw;
                const int id = z0 + ez, ih = y0 + ey, iw = x0 + ex;
                rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bok && idx < nb && !(tg.dbg & 1) && id >= 0 && id < g.Db && ih >= 0 && ih < g.Hb && iw >= 0 && iw < g.Wb) {
                    const size_t e = ((((size_t)n * g.Db + id) * g.Hb + ih) * g.Wb + iw) * K + k0 + part * 4;
                    rb[u] = *reinterpret_cast<const float4 *>(b + e);
                }
            }
        }
    };
    constexpr int IMG = (NA + NB) * 2048;  // floats of one image pair
    auto store_tile = [&](int par) {  // unconditional: the LDS regions hold NA*512 / NB*512 float4
#pragma unroll
        for (int u = 0; u < NA; u++) *reinterpret_cast<float4 *>(As + par * IMG + (size_t)(u * 512 + tid) * 4) = ra[u];
#pragma unroll
        for (int u = 0; u < NB; u++) *reinterpret_cast<float4 *>(Bs + par * IMG + (size_t)(u * 512 + tid) * 4) = rb[u];
    };
    // operands of k-step s2 (k = voxel pair).  Byte offsets: slot*128 + lane column; the per-tap part is hoisted.
    // The lane half h is the second voxel of the pair: s2 is even and TW a power of two >= 2, so (s2 + h) only changes
    // wx by h -- that part is a per-lane constant folded into aoff / boff, and the step's slot offset is wave-uniform
    // (scalar ALU: VALU instructions are paid in MFMA issue cycles).
    int aoff[TPW], boff[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        aoff[j] = ta[j] * 128 + i * 4 + h * g.sa[2] * 128;
        boff[j] = tb[j] * 128 + i * 4 + h * g.sb[2] * 128;
    }
    const char *Ab = reinterpret_cast<const char *>(As), *Bb = reinterpret_cast<const char *>(Bs);
    constexpr int NAV = SH == 2 ? 1 : TPW, NBV = SH == 1 ? 1 : TPW;
    auto read_ops = [&](int s2, float (&av)[NAV], float (&bv)[NBV]) {
        const int v = s2 < TV ? s2 : TV - 2;  // the tail prefetch re-reads the last step (harmless); wave-uniform
        const int wx = v & (tg.TW - 1), hy = (v >> tg.lTW) & (tg.TH - 1), dz = v >> (tg.lTW + tg.lTH);
        const int sa_ = (((dz * g.sa[0]) * tg.EAh + hy * g.sa[1]) * tg.EAw + wx * g.sa[2]) * 128;
        const int sb_ = (((dz * g.sb[0]) * tg.EBh + hy * g.sb[1]) * tg.EBw + wx * g.sb[2]) * 128;
#pragma unroll
        for (int j = 0; j < NAV; j++) av[j] = *reinterpret_cast<const float *>(Ab + sa_ + aoff[j]);
#pragma unroll
        for (int j = 0; j < NBV; j++) bv[j] = *reinterpret_cast<const float *>(Bb + sb_ + boff[j]);
    };
    // bias gradient (SH == 1: every tap reads the same dy slot): the tap-group-0 wave of the c-block-0 workgroups sees
    // every dy value of its k-block exactly once per (split, group) -- per-lane sums, reduced by k_dbias_reduce
    const bool dob = SH == 1 && pbias != nullptr && cb == 0 && wave == 0;
    float bsum = 0.f;
    auto mfmas = [&](const float (&av)[NAV], const float (&bv)[NBV]) {
#pragma unroll
        for (int j = 0; j < TPW; j++)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[SH == 2 ? 0 : j], bv[SH == 1 ? 0 : j], acc[j], 0, 0, 0);
        if (SH == 1 && dob) bsum += bv[0];
    };
    // scheduling hint for one (reads of the next step | MFMAs of this step) block: one MFMA, then a few of the
    // next step's LDS reads / address VALU ops under its shadow
    auto interleave = [&]() {
#pragma unroll
        for (int j = 0; j < TPW; j++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
        }
    };

    int tile = split, par = 0;
    if (tile < tg.ntiles) load_tile(tile);
    if (db) {
        if (tile < tg.ntiles) store_tile(0);
        __syncthreads();
    }
    while (tile < tg.ntiles) {
        if (!db) {
            __syncthreads();  // every wave is done reading the previous tile
            store_tile(0);
            __syncthreads();
        }
        const int next = tile + tg.nsplit;
        if (db) load_tile_buf(next);  // in flight during this tile's MFMAs
        else if (next < tg.ntiles) load_tile(next);
        // branch-free k loop unrolled by two with ping-pong operand registers: the LDS reads of step s+1 are issued
        // under the MFMAs of step s (left alone, the scheduler sinks every read next to its MFMA and each MFMA then
        // eats a full LDS round trip)
        if (!(tg.dbg & 2)) {
            float a0[NAV], b0[NBV], a1_[NAV], b1_[NBV];
            read_ops(2 * grp, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            for (int s2 = 2 * grp; s2 < TV; s2 += 8) {  // TV is a multiple of 8 for every tile shape; group: steps 2g + 4m
                read_ops(s2 + 4, a1_, b1_);
                mfmas(a0, b0);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                read_ops(s2 + 8, a0, b0);
                mfmas(a1_, b1_);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (db) {  // the next tile goes into the other image; one barrier per tile
            if (next < tg.ntiles) store_tile(par ^ 1);
            __syncthreads();
            par ^= 1;
            Ab = reinterpret_cast<const char *>(As + par * IMG);
            Bb = reinterpret_cast<const char *>(Bs + par * IMG);
        }
        tile = next;
    }
    if (SH == 1 && dob && i < kvalid) pbias[((size_t)(split * 2 + grp) * 2 + h) * K + k0 + i] = bsum;
    // partial[split][t][c][k]; D layout: col = lane&31 -> k, row -> c
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int t = wave + 4 * j;
        if (t < g.ntaps) {
            float *po = partial + ((size_t)(split * 2 + grp) * g.ntaps + t) * C * K;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < cvalid && i < kvalid) po[(size_t)(c0 + row) * K + k0 + i] = acc[j][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ Winograd wgrad
// Weight gradient of a 3x3x3 stride-1 conv with the TRANSPOSED F(2,3) algorithm along W.  For an output pair
// (w = 2q, 2q+1), e = (dy[2q], dy[2q+1]) and the inputs d0..d3 = x[2q-1 .. 2q+2] of one (dz,dy) filter row:
//     V = B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)        E = A e = (e0, e0+e1, e0-e1, -e1)
//     M_p[c][k] += V_p[c] * E_p[k]      (p = 0..3: four rank-1 updates per pair instead of six per two voxels)
//     dW row = G^T M = (M0 + (M1+M2)/2, (M1-M2)/2, (M1+M2)/2 + M3)     (applied by k_wgrad_reduce_wino)
// The GEMM k dimension is the PAIR index (64 pairs per 2x8x8 tile); wave w owns position p = w for all nine (dz,dy)
// rows (9 accumulator tiles), so its B operand E_p is shared by its nine MFMAs of a k-step and every A operand costs
// two LDS reads and one FMA.  Staging, split-K, prefetch and the fp32-partials / fp64 fixed-order reduce are those of
// k_wgrad_mfma; 36 partial tiles per (c-block, k-block) instead of 27.
template <int NA, int NB, int NQ>
__global__ __launch_bounds__(256, 1) void k_wgrad_wino(const WgradGeom g, const WgTile tg, const float *__restrict__ a1,
                                                       const float *__restrict__ a2, const float *__restrict__ b,
                                                       float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    float *Bs = lds + (size_t)NA * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y / tg.nkb, kb = blockIdx.y % tg.nkb;
    const int split = blockIdx.x;
    const int C = g.C1 + g.C2, K = g.K;

    // position p = wave: V_p = x[ja] + sg * x[jb], E_p = ea * e0 + eb * e1
    const int ja = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int jb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const float ea = wave == 3 ? 0.f : 1.f;
    const float eb = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);

    f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    const int c0 = cb * 32;
    const float *asrc;
    int Cs, cofs;
    if (c0 < g.C1) {
        asrc = a1; Cs = g.C1; cofs = c0;
    } else {
        asrc = a2; Cs = g.C2; cofs = c0 - g.C1;
    }
    const int k0 = kb * 32;
    const int EAhw = tg.EAh * tg.EAw, EBhw = tg.EBh * tg.EBw;
    const int na = tg.nslotsA * 8, nb = tg.nslotsB * 8;
    const int part = tid & 7;

    // The slot -> (z,y,x) decode of the staging pass does not depend on the tile: it is done once, here.  Per slot a
    // thread keeps the element offset relative to the tile origin and the packed halo coordinates (for border tiles).
    int rela[NA], cza[NA], relb[NB], czb[NB];
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int idx = u * 256 + tid;
        const int slot = idx >> 3;
        const int ez = (slot * tg.magAhw) >> 16, rem = slot - ez * EAhw;
        const int ey = (rem * tg.magAw) >> 16, ex = rem - ey * tg.EAw;
        rela[u] = ((ez * g.Hi + ey) * g.Wi + ex) * Cs + cofs + part * 4;
        cza[u] = idx < na ? ((ez << 16) | (ey << 8) | ex) : -1;
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int idx = u * 256 + tid;
        const int slot = idx >> 3;
        const int ez = (slot * tg.magBhw) >> 16, rem = slot - ez * EBhw;
        const int ey = (rem * tg.magBw) >> 16, ex = rem - ey * tg.EBw;
        relb[u] = ((ez * g.Hb + ey) * g.Wb + ex) * K + k0 + part * 4;
        czb[u] = idx < nb ? ((ez << 16) | (ey << 8) | ex) : -1;
    }
    const int EAd = tg.nslotsA / EAhw, EBd = tg.nslotsB / EBhw;

    float4 ra[NA], rb[NB];
    auto load_tile = [&](int tile) {
        unsigned r_ = (unsigned)tile;
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * tg.TD, oh0 = th_ * tg.TH, ow0 = tw_ * tg.TW;
        {
            const int z0 = od0 + tg.minA[0], y0 = oh0 + tg.minA[1], x0 = ow0 + tg.minA[2];
            // tile origin (may be "before" the tensor for border tiles: only dereferenced through valid slots)
            const float *base = asrc + ((((long)n * g.Di + z0) * g.Hi + y0) * g.Wi + x0) * (long)Cs;
            const bool interior = z0 >= 0 && y0 >= 0 && x0 >= 0 && z0 + EAd <= g.Di && y0 + tg.EAh <= g.Hi &&
                                  x0 + tg.EAw <= g.Wi;  // block-uniform
            if (interior) {
#pragma unroll
                for (int u = 0; u < NA; u++) {
                    ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cza[u] >= 0) ra[u] = *reinterpret_cast<const float4 *>(base + rela[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NA; u++) {
                    const int id = z0 + (cza[u] >> 16), ih = y0 + ((cza[u] >> 8) & 255), iw = x0 + (cza[u] & 255);
                    ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cza[u] >= 0 && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                        ra[u] = *reinterpret_cast<const float4 *>(base + rela[u]);
                }
            }
        }
        {
            const float *base = b + ((((long)n * g.Db + od0) * g.Hb + oh0) * g.Wb + ow0) * (long)K;
            const bool interior = od0 + EBd <= g.Db && oh0 + tg.EBh <= g.Hb && ow0 + tg.EBw <= g.Wb;
            if (interior) {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (czb[u] >= 0) rb[u] = *reinterpret_cast<const float4 *>(base + relb[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int id = od0 + (czb[u] >> 16), ih = oh0 + ((czb[u] >> 8) & 255), iw = ow0 + (czb[u] & 255);
                    rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (czb[u] >= 0 && id < g.Db && ih < g.Hb && iw < g.Wb)
                        rb[u] = *reinterpret_cast<const float4 *>(base + relb[u]);
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NA; u++) *reinterpret_cast<float4 *>(As + (size_t)(u * 256 + tid) * 4) = ra[u];
#pragma unroll
        for (int u = 0; u < NB; u++) *reinterpret_cast<float4 *>(Bs + (size_t)(u * 256 + tid) * 4) = rb[u];
    };
    // ---- k loop.  A k-step is TWO pairs: lane half h takes the d-plane h of the (2 x TH x TW) tile, so a step is a
    // (row hy, pair column q) position; steps walk q fastest.  The transformed input V_p of halo row r = hy + gy is the
    // SAME value for the three filter rows gy that touch it, so a wave keeps a sliding window of V in registers --
    // 3 (gz) x NQ (pair columns) x a ring of 4 rows (row r lives in slot r & 3; the row loop is unrolled by four so all
    // register indices are static and nothing is ever moved).  Per step: 9 MFMAs, ONE new row piece (3 x 2
    // ds_read_b32) + (e0, e1), fetched a step ahead under the MFMAs.  Lessons measured with
    // tools/probes/mfma_probe.hip and earlier versions of this loop: (1) a single wave per SIMD cannot issue 20
    // four-byte LDS reads per nine MFMAs; (2) overwriting a VGPR that an MFMA issued just before reads as A/B stalls
    // the writer for the rest of that MFMA (111 vs 137 TFLOP/s) -- here a slot is rewritten >= NQ steps after its
    // last reader; (3) MFMAs inside divergent-looking branches make hipcc copy accumulator tiles: the loop body is
    // branch-free.
    const char *Ab = reinterpret_cast<const char *>(As), *Bb = reinterpret_cast<const char *>(Bs);
    const int rowB = tg.EAw * 128;                    // bytes between halo rows
    const int plB = tg.EAh * tg.EAw * 128;            // bytes between halo planes
    const int abase = h * plB + i * 4;                // this lane half's d-plane, lane column
    const int bbase = (h * tg.EBh * tg.EBw) * 128 + i * 4;
    auto a_addr = [&](int gz, int r, int q) { return abase + gz * plB + r * rowB + q * 256; };
    float V[3][NQ][4];
    float e0, e1;

    int tile = split;
    if (tile < tg.ntiles) load_tile(tile);
    while (tile < tg.ntiles) {
        __syncthreads();
        store_tile();
        __syncthreads();
        const int next = tile + tg.nsplit;
        if (next < tg.ntiles) load_tile(next);
        // rows 0..2 of every pair column
#pragma unroll
        for (int gz = 0; gz < 3; gz++)
#pragma unroll
            for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const int ad = a_addr(gz, r, q);
                    V[gz][q][r] = fmaf(sg, *reinterpret_cast<const float *>(Ab + ad + jb * 128),
                                       *reinterpret_cast<const float *>(Ab + ad + ja * 128));
                }
        e0 = *reinterpret_cast<const float *>(Bb + bbase);
        e1 = *reinterpret_cast<const float *>(Bb + bbase + 128);
        __builtin_amdgcn_sched_barrier(0);
        for (int hy0 = 0; hy0 < tg.TH; hy0 += 4) {  // TH is 4 or 8
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int hy = hy0 + u;
                    const float bv = fmaf(eb, e1, ea * e0);
                    // next step: (hy, q + 1) or (hy + 1, 0); the very last one re-reads an in-range row (unused)
                    const int nh = q + 1 < NQ ? hy : (hy + 1 < tg.TH ? hy + 1 : hy);
                    const int be = bbase + (nh * tg.EBw + 2 * (q + 1 < NQ ? q + 1 : 0)) * 128;
                    const int rn = hy + 3 < tg.TH + 2 ? hy + 3 : tg.TH + 1;  // new row of this column (clamped: unused)
                    float xa[3], xb[3];
#pragma unroll
                    for (int j = 0; j < 9; j++) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j / 3][q][(u + j % 3) & 3], bv, acc[j], 0, 0, 0);
                        if (j == 0) {
                            e0 = *reinterpret_cast<const float *>(Bb + be);
                            e1 = *reinterpret_cast<const float *>(Bb + be + 128);
                        } else if (j <= 3) {
                            const int ad = a_addr(j - 1, rn, q);
                            xa[j - 1] = *reinterpret_cast<const float *>(Ab + ad + ja * 128);
                            xb[j - 1] = *reinterpret_cast<const float *>(Ab + ad + jb * 128);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int gz = 0; gz < 3; gz++) V[gz][q][(u + 3) & 3] = fmaf(sg, xb[gz], xa[gz]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        tile = next;
    }
    // partial[split][g][p][c][k]; D layout: col = lane&31 -> k, row -> c
#pragma unroll
    for (int j = 0; j < 9; j++) {
        float *po = partial + (((size_t)split * 9 + j) * 4 + wave) * C * K;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            po[(size_t)(c0 + row) * K + k0 + i] = acc[j][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ Winograd wgrad, 2-D
// F(2x2,3x3)-transposed weight gradient: per filter plane gz and 2x2 output quad,
//     V = B^T P B (4x4 input patch -> 16 positions), E = A e A^T (2x2 dy quad -> 16 positions),
//     M_{gz,a,b}[c][k] += V_ab[c] * E_ab[k],   dW[gz] = G^T M[gz] G  (k_wgrad_reduce_wino2, fp64)
// -> 48 rank-1 updates per quad (4 voxels) = 12 per voxel instead of 27 (direct) or 18 (1-D).  The GEMM k index is the
// quad (32 per 2x8x8 tile, lane half h = d-plane); wave a owns position row a for the three planes and four columns
// b: 12 accumulator tiles (192 AGPRs).  R_a[col] = P[ra][col] +- P[rb][col] depends only on the patch column, and
// neighbouring quads of a row share two columns, so a wave keeps R[3 gz][ring of 4 columns] in registers and fetches
// two new columns per step (four at a row end) + the 2x2 dy quad, one step ahead; V (12) and E (4) are double
// buffered so no MFMA operand register is rewritten behind the MFMA that reads it.  The position row a is a
// template parameter (the kernel switches on the wave index once): every sign is an add/sub and the two patch rows of
// a column come from one ds_read2st64_b32.  Tile fixed to 2x8x8 voxels (halo 4 x 10 x 10 slots).
// B0 / NBW: the wave owns position columns b = B0 .. B0+NBW-1 (NBW = 4: four waves per workgroup, one per position row;
// NBW = 2: eight waves, two per row -- half the accumulators per wave, so two waves share a SIMD and cover each
// other's barrier / LDS waits).  TPB = threads per workgroup (staging loops).
// GZ0 / NGZ: the wave owns filter planes GZ0 .. GZ0+NGZ-1 (3 planes by default; the twelve-wave variant gives each wave
// one plane and all four columns: 64 accumulator registers, three waves per SIMD).
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
#ifndef MVD_WG16_DBG
#define MVD_WG16_DBG 0
#endif
#if (MVD_WG16_DBG & 64)  // diagnostic build only (tools/stamps_wgrad16.py): s_memtime stamps of one wave per tile
__device__ long long g_wg16_stamps[64 * 8];
extern "C" int mvd_debug_wg16_stamps(long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wg16_stamps), sizeof(long long) * 64 * 8) == hipSuccess ? 0 : 1;
}
#define MVD_WGS(K) { if (stamp_on && nst < 60) g_wg16_stamps[nst * 8 + (K)] = __builtin_amdgcn_s_memtime(); }
#else
#define MVD_WGS(K)
#endif
// DB: two LDS images (A 4 x 10 x 10 slots + B 2 x 8 x 8 slots, 67 584 bytes each, allocated exactly): a wave writes the
// next tile into the other image as soon as its own steps are done -- under the MFMAs of the SIMD's younger waves -- and
// the tile loop has ONE barrier per tile instead of barrier / write / barrier with the MFMA pipe idle.
template <int A, int B0, int NBW, int TPB, int NA, int NB, int GZ0 = 0, int NGZ = 3, bool DB = false>
__device__ __forceinline__ void wgrad_wino2_body(const WgradGeom &g, const WgTile &tg, const float *__restrict__ a1,
                                                 const float *__restrict__ a2, const float *__restrict__ b,
                                                 float *__restrict__ partial, float *__restrict__ pbias, float *As,
                                                 float *Bs) {
    constexpr int RA = A == 0 ? 0 : (A == 2 ? 2 : 1);
    constexpr int RB = A == 0 ? 2 : (A == 1 ? 2 : (A == 2 ? 1 : 3));
    constexpr int EAW = 10, EAH = 10, EBW = 8, EBH = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y / tg.nkb, kb = blockIdx.y % tg.nkb;
    const int split = blockIdx.x;
    const int C = g.C1 + g.C2, K = g.K;

    constexpr int NTL = NGZ * NBW;
    f32x16 acc[NTL];  // [gz - GZ0][b - B0]
#pragma unroll
    for (int j = 0; j < NTL; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    const int c0 = cb * 32;
    const float *asrc;
    int Cs, cofs;
    if (c0 < g.C1) {
        asrc = a1; Cs = g.C1; cofs = c0;
    } else {
        asrc = a2; Cs = g.C2; cofs = c0 - g.C1;
    }
    const int k0 = kb * 32;
    const int na = tg.nslotsA * 8, nb = tg.nslotsB * 8;
    const int part = tid & 7;

    int rela[NA], cza[NA], relb[NB], czb[NB];
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int idx = u * TPB + tid;
        const int slot = idx >> 3;
        const int ez = slot / (EAH * EAW), rem = slot - ez * (EAH * EAW);
        const int ey = rem / EAW, ex = rem - ey * EAW;
        rela[u] = ((ez * g.Hi + ey) * g.Wi + ex) * Cs + cofs + part * 4;
        cza[u] = idx < na ? ((ez << 16) | (ey << 8) | ex) : -1;
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int idx = u * TPB + tid;
        const int slot = idx >> 3;
        const int ez = slot / (EBH * EBW), rem = slot - ez * (EBH * EBW);
        const int ey = rem / EBW, ex = rem - ey * EBW;
        relb[u] = ((ez * g.Hb + ey) * g.Wb + ex) * K + k0 + part * 4;
        czb[u] = idx < nb ? ((ez << 16) | (ey << 8) | ex) : -1;
    }

    float4 ra[NA], rb[NB];
    auto load_tile = [&](int tile) {
        unsigned r_ = (unsigned)tile;
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * 2, oh0 = th_ * 8, ow0 = tw_ * 8;
        {
            const int z0 = od0 - 1, y0 = oh0 - 1, x0 = ow0 - 1;
            const float *base = asrc + ((((long)n * g.Di + z0) * g.Hi + y0) * g.Wi + x0) * (long)Cs;
            const bool interior = z0 >= 0 && y0 >= 0 && x0 >= 0 && z0 + 4 <= g.Di && y0 + EAH <= g.Hi && x0 + EAW <= g.Wi;
            if (interior) {
#pragma unroll
                for (int u = 0; u < NA; u++) {
                    ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cza[u] >= 0) ra[u] = *reinterpret_cast<const float4 *>(base + rela[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NA; u++) {
                    const int id = z0 + (cza[u] >> 16), ih = y0 + ((cza[u] >> 8) & 255), iw = x0 + (cza[u] & 255);
                    ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cza[u] >= 0 && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                        ra[u] = *reinterpret_cast<const float4 *>(base + rela[u]);
                }
            }
        }
        {
            const float *base = b + ((((long)n * g.Db + od0) * g.Hb + oh0) * g.Wb + ow0) * (long)K;
            const bool interior = od0 + 2 <= g.Db && oh0 + EBH <= g.Hb && ow0 + EBW <= g.Wb;
            if (interior) {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (czb[u] >= 0) rb[u] = *reinterpret_cast<const float4 *>(base + relb[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int id = od0 + (czb[u] >> 16), ih = oh0 + ((czb[u] >> 8) & 255), iw = ow0 + (czb[u] & 255);
                    rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (czb[u] >= 0 && id < g.Db && ih < g.Hb && iw < g.Wb)
                        rb[u] = *reinterpret_cast<const float4 *>(base + relb[u]);
                }
            }
        }
    };
    constexpr int DBUF = (4 * EAH * EAW + 2 * EBH * EBW) * 128;  // bytes of one image pair (DB)
    auto store_tile = [&](int par) {
        float *Ad = As + (DB ? par * (DBUF / 4) : 0), *Bd = Bs + (DB ? par * (DBUF / 4) : 0);
#pragma unroll
        for (int u = 0; u < NA; u++)
            if (!DB || (u + 1) * TPB <= 4 * EAH * EAW * 8 || u * TPB + tid < na)  // (exact allocation: no slack to write into)
                *reinterpret_cast<float4 *>(Ad + (size_t)(u * TPB + tid) * 4) = ra[u];
#pragma unroll
        for (int u = 0; u < NB; u++)
            if (!DB || (u + 1) * TPB <= 2 * EBH * EBW * 8 || u * TPB + tid < nb)
                *reinterpret_cast<float4 *>(Bd + (size_t)(u * TPB + tid) * 4) = rb[u];
    };

    const char *Ab = reinterpret_cast<const char *>(As), *Bb = reinterpret_cast<const char *>(Bs);
    const int abase = h * (EAH * EAW * 128) + i * 4;  // lane half's d-plane, lane column
    const int bbase = h * (EBH * EBW * 128) + i * 4;
    // R of patch column `col` (halo x index) of quad row hq, plane gz
    auto fetch_col = [&](int gz, int hq, int col, float &pa, float &pb) {
        const int ad = abase + ((gz * EAH + 2 * hq) * EAW + col) * 128;
        pa = *reinterpret_cast<const float *>(Ab + ad + RA * EAW * 128);
        pb = *reinterpret_cast<const float *>(Ab + ad + RB * EAW * 128);
    };
    auto rcomb = [&](float pa, float pb) { return A == 1 ? pa + pb : pa - pb; };
    auto fetch_e = [&](int hq, int wq, float (&e)[4]) {
        const int ad = bbase + ((2 * hq) * EBW + 2 * wq) * 128;
        e[0] = *reinterpret_cast<const float *>(Bb + ad);
        e[1] = *reinterpret_cast<const float *>(Bb + ad + 128);
        e[2] = *reinterpret_cast<const float *>(Bb + ad + EBW * 128);
        e[3] = *reinterpret_cast<const float *>(Bb + ad + EBW * 128 + 128);
    };
    // E row a from the dy quad: F = (A e)_a, E_b = (F A^T)_b
    auto make_E = [&](const float (&e)[4], float (&E)[4]) {
        const float f0 = A == 0 ? e[0] : (A == 1 ? e[0] + e[2] : (A == 2 ? e[0] - e[2] : -e[2]));
        const float f1 = A == 0 ? e[1] : (A == 1 ? e[1] + e[3] : (A == 2 ? e[1] - e[3] : -e[3]));
        E[0] = f0; E[1] = f0 + f1; E[2] = f0 - f1; E[3] = -f1;
    };

    float R[NGZ][4];      // [gz - GZ0][column ring: halo column x lives in slot x & 3]
    float V[2][NGZ][4];   // MFMA A operands, double buffered over the step parity
    float E[2][4];      // MFMA B operands
    // bias gradient: wave 0 of the c-block-0 workgroups sees every dy value of its k-block exactly once (the 2x2
    // quads it fetches for E): per-lane partial sums, reduced over lanes halves / splits by k_dbias_reduce
    const float bflag = (A == 0 && B0 == 0 && GZ0 == 0 && pbias != nullptr && cb == 0) ? 1.f : 0.f;
    float bsum = 0.f;
#if (MVD_WG16_DBG & 64)
    const bool stamp_on = (MVD_WG16_DBG & 128) && blockIdx.x == 100 && blockIdx.y == 0 && A == 0 && GZ0 == 0 && B0 == 0 && lane == 0;
    int nst = 0;
#endif
    int tile = split;
    if (tile < tg.ntiles) load_tile(tile);
    int par = 0;
    if (DB) {
        if (tile < tg.ntiles) store_tile(0);
        __syncthreads();
    }
    while (tile < tg.ntiles) {
        MVD_WGS(0)
        if (!DB) {
            __syncthreads();
            MVD_WGS(1)
            store_tile(0);
            MVD_WGS(2)
            __syncthreads();
        }
        MVD_WGS(3)
        const int next = tile + tg.nsplit;
        // Single image: the next tile's loads are issued here, every wave of the workgroup at the same time with the MFMA
        // pipe idle (2 k of a 20 k ticks tile, tools/stamps_wgrad_wino.py); the same predicated loads one quad row into
        // the steps measured 3-4 % SLOWER (the branches around them cost the pipelined LDS reads their counted waits).
        // DB: unconditional BUFFER loads behind the MFMAs of the first quad row (issue_load): a lane whose halo voxel is
        // outside the volume passes an out-of-range offset and gets zeros, after the last tile the descriptors have
        // zero records.
        bool nintA = true, nintB = true;
        int nz0 = 0, ny0 = 0, nx0 = 0;
        __amdgpu_buffer_rsrc_t rA, rB;
        if (DB) {
            const bool more = next < tg.ntiles;
            unsigned r_ = (unsigned)(more ? next : 0);
            const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
            const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
            const int td_ = (int)(r_ % (unsigned)tg.ntd);
            const int n = (int)(r_ / (unsigned)tg.ntd);
            nz0 = td_ * 2; ny0 = th_ * 8; nx0 = tw_ * 8;  // first output voxel; the A halo starts one voxel before it
            nintA = nz0 >= 1 && ny0 >= 1 && nx0 >= 1 && nz0 + 3 <= g.Di && ny0 - 1 + EAH <= g.Hi && nx0 - 1 + EAW <= g.Wi;
            nintB = nz0 + 2 <= g.Db && ny0 + EBH <= g.Hb && nx0 + EBW <= g.Wb;
            const float *baseA = asrc + ((((long)n * g.Di + (nz0 - 1)) * g.Hi + (ny0 - 1)) * g.Wi + (nx0 - 1)) * (long)Cs;
            const float *baseB = b + ((((long)n * g.Db + nz0) * g.Hb + ny0) * g.Wb + nx0) * (long)K;
            rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(baseA), 0, more ? 0x7fffffff : 0, 0x00020000);
            rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(baseB), 0, more ? 0x7fffffff : 0, 0x00020000);
        } else if (next < tg.ntiles) {
            load_tile(next);
        }
        auto issue_load = [&](int u) {
            if (u >= NA + NB) return;
            const bool isA = u < NA;
            const int ua = isA ? u : 0, ub = isA ? 0 : u - NA;
            const int cz = isA ? cza[ua] : czb[ub];
            unsigned o = cz >= 0 ? (unsigned)(isA ? rela[ua] : relb[ub]) * 4u : 0xffffffffu;
            if (isA ? !nintA : !nintB) {  // block-uniform; VALU only inside
                const int id = (isA ? nz0 - 1 : nz0) + (cz >> 16), ih = (isA ? ny0 - 1 : ny0) + ((cz >> 8) & 255),
                          iw = (isA ? nx0 - 1 : nx0) + (cz & 255);
                const bool ok = id >= 0 && id < (isA ? g.Di : g.Db) && ih >= 0 && ih < (isA ? g.Hi : g.Hb) && iw >= 0 &&
                                iw < (isA ? g.Wi : g.Wb);
                o = ok ? o : 0xffffffffu;
            }
            const u32x4w q4 = __builtin_amdgcn_raw_buffer_load_b128(isA ? rA : rB, (int)o, 0, 0);
            const float4 f4 = __builtin_bit_cast(float4, q4);
            if (isA) ra[ua] = f4;
            else rb[ub] = f4;
        };
        MVD_WGS(4)
        // window of the first quad: columns 0..3 of quad row 0, and its dy quad
        {
            float e[4];
            fetch_e(0, 0, e);
#pragma unroll
            for (int gz = 0; gz < NGZ; gz++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float pa, pb;
                    fetch_col(GZ0 + gz, 0, c, pa, pb);
                    R[gz][c] = rcomb(pa, pb);
                }
            make_E(e, E[0]);
            if (A == 0 && B0 == 0 && GZ0 == 0) bsum += bflag * ((e[0] + e[1]) + (e[2] + e[3]));
#pragma unroll
            for (int gz = 0; gz < NGZ; gz++) {
                V[0][gz][0] = R[gz][0] - R[gz][2];
                V[0][gz][1] = R[gz][1] + R[gz][2];
                V[0][gz][2] = R[gz][2] - R[gz][1];
                V[0][gz][3] = R[gz][1] - R[gz][3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // one quad row = four steps; WL: the next tile's loads ride behind the steps' MFMAs, two per step (DB, row 0)
        auto quad_row = [&](int hq, auto WL) {
#pragma unroll
            for (int wq = 0; wq < 4; wq++) {
                constexpr int dummy = 0;
                (void)dummy;
                const int cur = wq & 1, nxt = cur ^ 1;
                // the next step: (hq, wq + 1) needs columns 2wq+4, 2wq+5; after the row's last quad the whole window
                // (columns 0..3) of quad row hq + 1 (the tile's very last step re-reads row hq: unused)
                const int nhq = wq == 3 ? (hq < 3 ? hq + 1 : hq) : hq;
                constexpr int NC = 4;  // columns fetched at a row end; 2 otherwise
                float pa[NGZ][NC], pb[NGZ][NC], e[4];
#pragma unroll
                for (int j = 0; j < NTL; j++) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[cur][j / NBW][B0 + j % NBW], E[cur][B0 + j % NBW], acc[j], 0, 0,
                                                                  0);
                    if (j == 0) fetch_e(nhq, wq == 3 ? 0 : wq + 1, e);
                    // fetch group of this MFMA slot: 3 groups (one plane each) in a row's interior, 6 (half a plane
                    // each) at a row end; with 6 MFMAs per step the row-end groups start at slot 0
                    const int fg = (NTL >= 7 || wq != 3) ? j - 1 : j;
                    if (wq != 3) {
                        if (fg >= 0 && fg < NGZ) {  // plane fg: two new columns
                            fetch_col(GZ0 + fg, nhq, 2 * wq + 4, pa[fg][0], pb[fg][0]);
                            fetch_col(GZ0 + fg, nhq, 2 * wq + 5, pa[fg][1], pb[fg][1]);
                        }
                    } else {
                        if (fg >= 0 && fg < 2 * NGZ) {  // plane fg/2: four columns, two per slot
                            const int gz = fg >> 1, c2 = (fg & 1) * 2;
                            fetch_col(GZ0 + gz, nhq, c2, pa[gz][c2], pb[gz][c2]);
                            fetch_col(GZ0 + gz, nhq, c2 + 1, pa[gz][c2 + 1], pb[gz][c2 + 1]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // window update + operands of the next step (VALU only; the MFMAs above read V[cur] / E[cur])
                make_E(e, E[nxt]);
                if (A == 0 && B0 == 0 && GZ0 == 0) {  // the tile's very last fetch is a re-read: not counted
                    const float fl = (wq == 3 && hq == 3) ? 0.f : bflag;
                    bsum += fl * ((e[0] + e[1]) + (e[2] + e[3]));
                }
#pragma unroll
                for (int gz = 0; gz < NGZ; gz++) {
                    if (wq != 3) {
                        R[gz][(2 * wq + 4) & 3] = rcomb(pa[gz][0], pb[gz][0]);
                        R[gz][(2 * wq + 5) & 3] = rcomb(pa[gz][1], pb[gz][1]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; c++) R[gz][c] = rcomb(pa[gz][c], pb[gz][c]);
                    }
                    // next quad's patch columns are x = 2*nwq .. 2*nwq+3 with nwq = wq + 1 (or 0)
                    constexpr int dummy2 = 0;
                    (void)dummy2;
                    const int x0 = wq == 3 ? 0 : 2 * wq + 2;
                    const float r0 = R[gz][x0 & 3], r1 = R[gz][(x0 + 1) & 3], r2 = R[gz][(x0 + 2) & 3], r3 = R[gz][(x0 + 3) & 3];
                    V[nxt][gz][0] = r0 - r2;
                    V[nxt][gz][1] = r1 + r2;
                    V[nxt][gz][2] = r2 - r1;
                    V[nxt][gz][3] = r1 - r3;
                }
                if constexpr (decltype(WL)::value) {
                    issue_load(2 * wq);
                    issue_load(2 * wq + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (DB) {
            quad_row(0, std::true_type{});
#pragma unroll 1
            for (int hq = 1; hq < 4; hq++) quad_row(hq, std::false_type{});
        } else {
#pragma unroll 1
            for (int hq = 0; hq < 4; hq++) quad_row(hq, std::false_type{});
        }
        MVD_WGS(5)
        if (DB) {
            if (next < tg.ntiles) store_tile(par ^ 1);
            MVD_WGS(6)
            __syncthreads();
            MVD_WGS(7)
            par ^= 1;
            Ab = reinterpret_cast<const char *>(As) + par * DBUF;
            Bb = reinterpret_cast<const char *>(Bs) + par * DBUF;
        }
#if (MVD_WG16_DBG & 64)
        nst++;
#endif
        tile = next;
    }
    // partial[split][gz][a][b][c][k]; D layout: col = lane&31 -> k, row -> c
#pragma unroll
    for (int j = 0; j < NTL; j++) {
        float *po = partial + ((((size_t)split * 3 + GZ0 + j / NBW) * 4 + A) * 4 + B0 + j % NBW) * C * K;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            po[(size_t)(c0 + row) * K + k0 + i] = acc[j][r];
        }
    }
    if (A == 0 && B0 == 0 && GZ0 == 0 && pbias != nullptr && cb == 0) pbias[((size_t)split * 2 + h) * K + k0 + i] = bsum;
}

// dbias[k] = sum over splits and lane halves of pbias[row][k]   (fp64, fixed order).  Block = 64 channels x 16 row
// groups: group q sums rows q, q+16, ...; the 16 group sums are combined in a fixed order through LDS.
__global__ __launch_bounds__(1024) void k_dbias_reduce(const float *__restrict__ pbias, float *__restrict__ dbias, int K,
                                                       int nrows) {
    __shared__ double red[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + e;
    double s = 0;
    if (k < K) {
        int r = q;
        for (; r + 48 < nrows; r += 64) {  // four rows in flight, added in row order
            const float v0 = pbias[(size_t)r * K + k], v1 = pbias[(size_t)(r + 16) * K + k];
            const float v2 = pbias[(size_t)(r + 32) * K + k], v3 = pbias[(size_t)(r + 48) * K + k];
            s += (double)v0;
            s += (double)v1;
            s += (double)v2;
            s += (double)v3;
        }
        for (; r < nrows; r += 16) s += (double)pbias[(size_t)r * K + k];
    }
    red[q][e] = s;
    __syncthreads();
    if (q != 0 || k >= K) return;
    double t = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) t += red[j][e];
    dbias[k] = (float)t;
}

template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void k_wgrad_wino2(const WgradGeom g, const WgTile tg, const float *__restrict__ a1,
                                                        const float *__restrict__ a2, const float *__restrict__ b,
                                                        float *__restrict__ partial, float *__restrict__ pbias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    float *Bs = lds + (size_t)NA * 1024;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one code copy per position row: every wave of the workgroup runs the same trip counts, so the barriers inside
    // the copies pair up
    if (wave == 0) wgrad_wino2_body<0, 0, 4, 256, NA, NB>(g, tg, a1, a2, b, partial, pbias, As, Bs);
    else if (wave == 1) wgrad_wino2_body<1, 0, 4, 256, NA, NB>(g, tg, a1, a2, b, partial, pbias, As, Bs);
    else if (wave == 2) wgrad_wino2_body<2, 0, 4, 256, NA, NB>(g, tg, a1, a2, b, partial, pbias, As, Bs);
    else wgrad_wino2_body<3, 0, 4, 256, NA, NB>(g, tg, a1, a2, b, partial, pbias, As, Bs);
}

// eight-wave variant: wave w owns position row w >> 1 and the columns {0,1} or {2,3}; NA8 / NB8 = float4 per thread of the
// same LDS tiles staged by 512 threads
template <int NA8, int NB8>
__global__ __launch_bounds__(512, 2) void k_wgrad_wino2w8(const WgradGeom g, const WgTile tg, const float *__restrict__ a1,
                                                          const float *__restrict__ a2, const float *__restrict__ b,
                                                          float *__restrict__ partial, float *__restrict__ pbias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    float *Bs = lds + (size_t)NA8 * 2048;  // A region: NA8 float4 per thread x 512 threads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define MVD_W8(AA, BB) wgrad_wino2_body<AA, BB, 2, 512, NA8, NB8>(g, tg, a1, a2, b, partial, pbias, As, Bs)
    switch (wave) {
        case 0: MVD_W8(0, 0); break;
        case 1: MVD_W8(0, 2); break;
        case 2: MVD_W8(1, 0); break;
        case 3: MVD_W8(1, 2); break;
        case 4: MVD_W8(2, 0); break;
        case 5: MVD_W8(2, 2); break;
        case 6: MVD_W8(3, 0); break;
        default: MVD_W8(3, 2); break;
    }
#undef MVD_W8
}

// twelve-wave variant: wave w owns position row w & 3 and filter plane w >> 2 with all four columns (4 accumulator tiles):
// three waves per SIMD; NA12 / NB12 = float4 per thread of the same LDS tiles staged by 768 threads
template <int NA12, int NB12, bool DB12 = false>
__global__ __launch_bounds__(768, 1) void k_wgrad_wino2w12(const WgradGeom g, const WgTile tg, const float *__restrict__ a1,
                                                           const float *__restrict__ a2, const float *__restrict__ b,
                                                           float *__restrict__ partial, float *__restrict__ pbias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    float *Bs = lds + (DB12 ? (size_t)4 * 10 * 10 * 32 : (size_t)NA12 * 3072);  // A: exact (DB) / NA12 float4 x 768 threads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define MVD_W12(AA, GZ) wgrad_wino2_body<AA, 0, 4, 768, NA12, NB12, GZ, 1, DB12>(g, tg, a1, a2, b, partial, pbias, As, Bs)
    switch (wave) {
        case 0: MVD_W12(0, 0); break;
        case 1: MVD_W12(1, 0); break;
        case 2: MVD_W12(2, 0); break;
        case 3: MVD_W12(3, 0); break;
        case 4: MVD_W12(0, 1); break;
        case 5: MVD_W12(1, 1); break;
        case 6: MVD_W12(2, 1); break;
        case 7: MVD_W12(3, 1); break;
        case 8: MVD_W12(0, 2); break;
        case 9: MVD_W12(1, 2); break;
        case 10: MVD_W12(2, 2); break;
        default: MVD_W12(3, 2); break;
    }
#undef MVD_W12
}

// dw[k][c][gz][i][j] = sum_{a,b} G[a][i] G[b][j] (sum_split M[split][gz][a][b][c][k])   (fp64, fixed order)
// 1024 threads = 64 outputs x 4 position rows a x 4 split groups q (the small layers have only 3*C*K / 64 = 48..96
// workgroups: a quarter of the splits per thread keeps enough loads in flight to stream the ~50 MB of partials)
__global__ __launch_bounds__(1024) void k_wgrad_reduce_wino2(const float *__restrict__ partial, float *__restrict__ dw,
                                                             int C, int K, int nsplit) {
    __shared__ double redq[4][4][4][64];  // [q][a][b][e]
    __shared__ double red[4][3][64];
    const long CK = (long)C * K;
    const long per = 48 * CK;   // one split
    const long nout = 3 * CK;   // (gz, c, k)
    const int e = threadIdx.x & 63, a = (threadIdx.x >> 6) & 3, q = threadIdx.x >> 8;
    const long j = (long)blockIdx.x * 64 + e;
    double m[4] = {0, 0, 0, 0};
    if (j < nout) {
        const long gz = j / CK, ck = j - gz * CK;
        const float *src = partial + ((size_t)(gz * 4 + a) * 4) * CK + ck;
        int sp = q;
        for (; sp + 4 < nsplit; sp += 8) {  // two splits = eight loads in flight per thread, added in split order
            float v0[4], v1[4];
#pragma unroll
            for (int bq = 0; bq < 4; bq++) {
                v0[bq] = src[(size_t)sp * per + (size_t)bq * CK];
                v1[bq] = src[(size_t)(sp + 4) * per + (size_t)bq * CK];
            }
#pragma unroll
            for (int bq = 0; bq < 4; bq++) {
                m[bq] += (double)v0[bq];
                m[bq] += (double)v1[bq];
            }
        }
        for (; sp < nsplit; sp += 4) {
#pragma unroll
            for (int bq = 0; bq < 4; bq++) m[bq] += (double)src[(size_t)sp * per + (size_t)bq * CK];
        }
    }
#pragma unroll
    for (int bq = 0; bq < 4; bq++) redq[q][a][bq][e] = m[bq];
    __syncthreads();
    if (q == 0) {  // no wave leaves before the second barrier: every one of the 16 waves reaches both
#pragma unroll
        for (int bq = 0; bq < 4; bq++)
            m[bq] = ((redq[0][a][bq][e] + redq[1][a][bq][e]) + redq[2][a][bq][e]) + redq[3][a][bq][e];
        // row a of M times G: (M G)_aj
        red[a][0][e] = m[0] + 0.5 * (m[1] + m[2]);
        red[a][1][e] = 0.5 * (m[1] - m[2]);
        red[a][2][e] = 0.5 * (m[1] + m[2]) + m[3];
    }
    __syncthreads();
    if (q != 0) return;
    if (a != 0 || j >= nout) return;
    const long gz = j / CK, ck = j - gz * CK;
    const int c = (int)(ck / K), k = (int)(ck - (long)c * K);
    float *o = dw + ((size_t)k * C + c) * 27 + gz * 9;
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
        const double q0 = red[0][jj][e], q1 = red[1][jj][e], q2 = red[2][jj][e], q3 = red[3][jj][e];
        o[0 * 3 + jj] = (float)(q0 + 0.5 * (q1 + q2));
        o[1 * 3 + jj] = (float)(0.5 * (q1 - q2));
        o[2 * 3 + jj] = (float)(0.5 * (q1 + q2) + q3);
    }
}

// dw[k][c][gz][gy][0..2] = G^T (sum_split M[split][g][0..3][c][k])   (fp64 sums in a fixed order)
__global__ __launch_bounds__(256) void k_wgrad_reduce_wino(const float *__restrict__ partial, float *__restrict__ dw,
                                                           int C, int K, int nsplit) {
    __shared__ double red[4][64];
    const long per = (long)36 * C * K;   // one split
    const long nout = (long)9 * C * K;   // (g, c, k) triples; each owns 4 positions
    const int e = threadIdx.x & 63, p = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + e;
    double s0 = 0, s1 = 0;
    if (j < nout) {
        const long gi = j / ((long)C * K), ck = j - gi * (long)C * K;
        const float *src = partial + ((size_t)gi * 4 + p) * C * K + ck;
        int bsp = 0;
        for (; bsp + 1 < nsplit; bsp += 2) {
            s0 += (double)src[(size_t)bsp * per];
            s1 += (double)src[(size_t)(bsp + 1) * per];
        }
        if (bsp < nsplit) s0 += (double)src[(size_t)bsp * per];
    }
    red[p][e] = s0 + s1;
    __syncthreads();
    if (p != 0 || j >= nout) return;
    const double m0 = red[0][e], m1 = red[1][e], m2 = red[2][e], m3 = red[3][e];
    const long gi = j / ((long)C * K), ck = j - gi * (long)C * K;
    const int c = (int)(ck / K), k = (int)(ck - (long)c * K);
    float *o = dw + ((size_t)k * C + c) * 27 + gi * 3;
    o[0] = (float)(m0 + 0.5 * (m1 + m2));
    o[1] = (float)(0.5 * (m1 - m2));
    o[2] = (float)(0.5 * (m1 + m2) + m3);
}

// ------------------------------------------------------------------------------------------------ narrow-input wgrad
// C <= 8 with ntaps*C <= 128 (the 4-modality input layer: 27 taps x 4 channels = 108 rows).  Padding C to a 32-row M
// tile per tap wastes 7/8 of the MFMAs; here the GEMM M index is the (tap, channel) pair: wave w owns rows
// 32w..32w+31, one MFMA per voxel pair.  A tile in LDS is [slots][C] (16 B per voxel at C = 4), B tile [128][32 k].
template <int NAS, int NB>
__global__ __launch_bounds__(256, 2) void k_wgrad_smallc(const WgradGeom g, const WgTile tg, const float *__restrict__ a1,
                                                         const float *__restrict__ b, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                               // NAS*256 float4
    float *Bs = lds + (size_t)NAS * 1024;          // NB*256 float4
    int *toffs = reinterpret_cast<int *>(Bs + (size_t)NB * 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int kb = blockIdx.y, split = blockIdx.x;
    const int C = g.C1, K = g.K;
    const int PARTS = C / 4;
#pragma unroll
    for (int t = 0; t < 27; t++)
        if (tid == t && t < g.ntaps) toffs[t] = tg.toffA[t];
    __syncthreads();
    const int row = wave * 32 + i;
    const bool rvalid = row < g.ntaps * C;
    const int rtap = rvalid ? row / C : 0, rc = rvalid ? row % C : 0;
    // byte offset added to slot*C*4; the lane half h is the second voxel of the pair (s2 even, TW a power of two >= 2: it
    // only moves wx by one), so it is a per-lane constant here and the step's slot offsets stay wave-uniform
    const int arow = (toffs[rtap] * C + rc) * 4 + h * g.sa[2] * C * 4;
    const int k0 = kb * 32;
    const int kvalid = (K - k0) < 32 ? (K - k0) : 32;
    const int EAhw = tg.EAh * tg.EAw, EBhw = tg.EBh * tg.EBw;
    const int TV = tg.TD * tg.TH * tg.TW;
    const int na = tg.nslotsA * PARTS, nb = tg.nslotsB * 8;
    const int tb0 = tg.toffB[0] * 128 + i * 4 + h * g.sb[2] * 128;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float4 ra[NAS], rb[NB];
    auto load_tile = [&](int tile) {
        unsigned r_ = (unsigned)tile;
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * tg.TD, oh0 = th_ * tg.TH, ow0 = tw_ * tg.TW;
        {
            const int z0 = od0 * g.sa[0] + tg.minA[0], y0 = oh0 * g.sa[1] + tg.minA[1], x0 = ow0 * g.sa[2] + tg.minA[2];
#pragma unroll
            for (int u = 0; u < NAS; u++) {
                const int idx = u * 256 + tid;
                const int slot = idx / PARTS, part = idx - slot * PARTS;
                const int ez = (slot * tg.magAhw) >> 16, rem = slot - ez * EAhw;
                const int ey = (rem * tg.magAw) >> 16, ex = rem - ey * tg.EAw;
                const int id = z0 + ez, ih = y0 + ey, iw = x0 + ex;
                ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < na && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                    ra[u] = *reinterpret_cast<const float4 *>(
                        a1 + ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * C + part * 4);
            }
        }
        {
            const int z0 = od0 * g.sb[0] + tg.minB[0], y0 = oh0 * g.sb[1] + tg.minB[1], x0 = ow0 * g.sb[2] + tg.minB[2];
            const int part = tid & 7;
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int idx = u * 256 + tid;
                const int slot = idx >> 3;
                const int ez = (slot * tg.magBhw) >> 16, rem = slot - ez * EBhw;
                const int ey = (rem * tg.magBw) >> 16, ex = rem - ey * tg.EBw;
                const int id = z0 + ez, ih = y0 + ey, iw = x0 + ex;
                rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (part * 4 < kvalid && idx < nb && id >= 0 && id < g.Db && ih >= 0 && ih < g.Hb && iw >= 0 && iw < g.Wb)
                    rb[u] = *reinterpret_cast<const float4 *>(
                        b + ((((size_t)n * g.Db + id) * g.Hb + ih) * g.Wb + iw) * K + k0 + part * 4);
            }
        }
    };
    const char *Ab = reinterpret_cast<const char *>(As), *Bb = reinterpret_cast<const char *>(Bs);
    auto read_ops = [&](int s2, float &av, float &bv) {
        const int v = s2 < TV ? s2 : TV - 2;  // wave-uniform
        const int wx = v & (tg.TW - 1), hy = (v >> tg.lTW) & (tg.TH - 1), dz = v >> (tg.lTW + tg.lTH);
        const int sa_ = ((dz * g.sa[0]) * tg.EAh + hy * g.sa[1]) * tg.EAw + wx * g.sa[2];
        const int sb_ = ((dz * g.sb[0]) * tg.EBh + hy * g.sb[1]) * tg.EBw + wx * g.sb[2];
        av = *reinterpret_cast<const float *>(Ab + sa_ * C * 4 + arow);
        bv = *reinterpret_cast<const float *>(Bb + sb_ * 128 + tb0);
    };
    int tile = split;
    if (tile < tg.ntiles) load_tile(tile);
    while (tile < tg.ntiles) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NAS; u++) *reinterpret_cast<float4 *>(As + (size_t)(u * 256 + tid) * 4) = ra[u];
#pragma unroll
        for (int u = 0; u < NB; u++) *reinterpret_cast<float4 *>(Bs + (size_t)(u * 256 + tid) * 4) = rb[u];
        __syncthreads();
        const int next = tile + tg.nsplit;
        if (next < tg.ntiles) load_tile(next);
        float a0, b0, a1_, b1_;
        read_ops(0, a0, b0);
        for (int s2 = 0; s2 < TV; s2 += 4) {
            read_ops(s2 + 2, a1_, b1_);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            read_ops(s2 + 4, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b1_, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        tile = next;
    }
    // D layout: col = lane&31 -> k; row -> (tap, c) pair of this wave
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int rr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (rr < g.ntaps * C && i < kvalid) {
            const int t = rr / C, c = rr - t * C;
            partial[(((size_t)split * g.ntaps + t) * C + c) * K + k0 + i] = acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ bf16 wgrad
// bf16 A / B in HBM, bf16 MFMA (v_mfma_f32_32x32x16_bf16: 16 voxels per instruction), fp32 accumulate / partials.
// The reduce dimension of the weight gradient is the VOXEL index, but NDHWC tiles are channel-contiguous: the LDS
// images stay [slot][32 channels] (64-B rows, filled with plain 16-B stores) and the operands are fetched with gfx950's
// transposing LDS read ds_read_b64_tr_b16: per 16-lane group a 4-voxel x 16-channel block comes back channel-major,
// i.e. exactly the A (rows = reduce channels) / B (columns = produce channels) fragment of the MFMA -- two reads per
// operand per 16-voxel step.  Four consecutive 64-B rows cover all 64 banks once: the reads are conflict-free for
// unit-stride taps (2-way for the stride-2 side of strided / transposed convs).
// NA / NB: uint4 (8 channels) per thread of the A halo / B tile.  Same split-K / prefetch structure as k_wgrad_mfma.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));

__device__ inline bf16x8w tr_operand(const unsigned char *p0, const unsigned char *p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p1);
    return __builtin_bit_cast(bf16x8w, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// TRI (the plain 27-tap stride-1 gather with 8-voxel tile rows: SH == 1, TPW == 7): the three taps of an x-triple read the
// same halo row shifted by one voxel, so a wave takes two whole triples (taps 6 w .. 6 w + 5) plus one tap of the ninth
// triple (24 + w; wave 3's seventh slot stays the ones slot of the bias gradient), fetches the 12-voxel union of a
// triple with THREE transposing reads instead of six and derives the three A fragments in registers (the odd shift is
// four v_perm / v_alignbit).  10 + 1 LDS reads per 7 MFMAs instead of 16 + 1, and with 20 instead of 32 operand
// registers per step the 256-voxel tile can afford the double-buffered fetch it had to drop (its steps ran as
// table read -> address adds -> operand reads -> MFMAs, one latency chain each: MFMA pipe 40 % busy, PMC round 2).
template <int TPW, int NA, int NB, int SH, bool TRI = false>
__global__ __launch_bounds__(256, 2) void k_wgrad16(const WgradGeom g, const WgTile tg,
                                                    const unsigned short *__restrict__ a1,
                                                    const unsigned short *__restrict__ a2,
                                                    const unsigned short *__restrict__ b, float *__restrict__ partial,
                                                    float *__restrict__ pbias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    unsigned char *As = lds8;
    unsigned char *Bs = lds8 + (size_t)NA * 4096;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y / tg.nkb, kb = blockIdx.y % tg.nkb;
    const int split = blockIdx.x;
    const int C = g.C1 + g.C2, K = g.K;

    static_assert(!TRI || (TPW == 7 && SH == 1), "TRI: 27 taps over 4 waves x 7 slots, one dy fragment for all taps");
    auto tap_of = [&](int j) { return TRI ? (j < 6 ? 6 * wave + j : 24 + wave) : wave + 4 * j; };
    int ta[TPW], tb[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        ta[j] = tb[j] = 0;
#pragma unroll
        for (int t = 0; t < 27; t++)
            if (t == tap_of(j) && t < g.ntaps) {
                ta[j] = tg.toffA[t];
                tb[j] = tg.toffB[t];
            }
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    const int c0 = cb * 32;
    const unsigned short *asrc;
    int Cs, cofs;
    if (c0 < g.C1) {
        asrc = a1; Cs = g.C1; cofs = c0;
    } else {
        asrc = a2; Cs = g.C2; cofs = c0 - g.C1;
    }
    const int k0 = kb * 32;
    const int EAhw = tg.EAh * tg.EAw, EBhw = tg.EBh * tg.EBw;
    const int TV = tg.TD * tg.TH * tg.TW;
    const int na = tg.nslotsA * 4, nb = tg.nslotsB * 4;
    const int part = tid & 3;

    uint4 ra[NA], rb[NB];
    // tile-independent decode of this thread's staging slots (round 2): element offset from the tile's first halo voxel
    // and packed halo coordinates.  load_tile used to redo the slot -> (z, y, x) decode, six bounds compares and a 64-bit
    // address per 16-byte load for every tile: ~270 VALU instructions per tile and wave next to 56 MFMAs.  Interior
    // tiles (no halo voxel outside the volume; the majority) now load from scalar base + lane offset with no bounds
    // logic; border tiles test the precomputed coordinates.
    // (the packed halo coordinates the border tiles test are recomputed there: kept in registers they cost NA + NB of
    // the 256 for the whole kernel)
    unsigned relA[NA], relB[NB];  // BYTE offsets (< 2^31, host-checked tile extents): scalar base + 32-bit lane offset loads
    auto coordsA = [&](int u, int tid) {
        const int idx = u * 256 + tid;
        const int slot = idx < na ? (idx >> 2) : 0;
        const int ez = (slot * tg.magAhw) >> 16, rem = slot - ez * EAhw;
        const int ey = (rem * tg.magAw) >> 16, ex = rem - ey * tg.EAw;
        return idx < na ? ((ez << 16) | (ey << 8) | ex) : -1;
    };
    auto coordsB = [&](int u, int tid) {
        const int idx = u * 256 + tid;
        const int slot = idx < nb ? (idx >> 2) : 0;
        const int ez = (slot * tg.magBhw) >> 16, rem = slot - ez * EBhw;
        const int ey = (rem * tg.magBw) >> 16, ex = rem - ey * tg.EBw;
        return idx < nb ? ((ez << 16) | (ey << 8) | ex) : -1;
    };
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int c = coordsA(u, tid);
        relA[u] = (unsigned)((((c >> 16) * g.Hi + ((c >> 8) & 255)) * g.Wi + (c & 255)) * Cs + part * 8) * 2u;
        if (c < 0) relA[u] = part * 16;
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int c = coordsB(u, tid);
        relB[u] = (unsigned)((((c >> 16) * g.Hb + ((c >> 8) & 255)) * g.Wb + (c & 255)) * K + part * 8) * 2u;
        if (c < 0) relB[u] = part * 16;
    }
    const int EAd = (tg.nslotsA / EAhw), EBd = (tg.nslotsB / EBhw);  // halo extents along D
    // 16 bytes at wave-uniform base + 32-bit lane byte offset.  The opaque copy keeps the zero-extension next to the load,
    // which then takes the scalar-base + 32-bit-offset form; kept as 64-bit element offsets the NA + NB offsets cost
    // twice the registers and a 64-bit add per load.
    auto ld16 = [](const unsigned short *base, unsigned off) {
        asm volatile("" : "+v"(off));
        return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(base) + off);
    };
    // tile -> halo bases; true when no halo voxel of either operand lies outside its volume (block-uniform)
    int nza = 0, nya = 0, nxa = 0, nzb = 0, nyb = 0, nxb = 0;  // first halo voxel of the tile tile_bases() last decoded
    auto tile_bases = [&](int tile, const unsigned short *&baseA, const unsigned short *&baseB) {
        unsigned r_ = (unsigned)tile;
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * tg.TD, oh0 = th_ * tg.TH, ow0 = tw_ * tg.TW;
        const int za = od0 * g.sa[0] + tg.minA[0], ya = oh0 * g.sa[1] + tg.minA[1], xa = ow0 * g.sa[2] + tg.minA[2];
        const int zb = od0 * g.sb[0] + tg.minB[0], yb = oh0 * g.sb[1] + tg.minB[1], xb = ow0 * g.sb[2] + tg.minB[2];
        nza = za; nya = ya; nxa = xa; nzb = zb; nyb = yb; nxb = xb;
        baseA = asrc + ((((long)n * g.Di + za) * g.Hi + ya) * g.Wi + xa) * (long)Cs + cofs;
        baseB = b + ((((long)n * g.Db + zb) * g.Hb + yb) * g.Wb + xb) * (long)K + k0;
        return za >= 0 && za + EAd <= g.Di && ya >= 0 && ya + tg.EAh <= g.Hi && xa >= 0 && xa + tg.EAw <= g.Wi && zb >= 0 &&
               zb + EBd <= g.Db && yb >= 0 && yb + tg.EBh <= g.Hb && xb >= 0 && xb + tg.EBw <= g.Wb;
    };
    auto load_tile = [&](int tile) {
        unsigned r_ = (unsigned)tile;
        const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
        const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
        const int td_ = (int)(r_ % (unsigned)tg.ntd);
        const int n = (int)(r_ / (unsigned)tg.ntd);
        const int od0 = td_ * tg.TD, oh0 = th_ * tg.TH, ow0 = tw_ * tg.TW;
        int tid_here = tid;  // opaque copy: keeps the border tiles' coordinate decode inside this call (hoisted out of
        asm volatile("" : "+v"(tid_here));  // the tile loop it would sit in NA + NB registers again)
        {
            const int z0 = od0 * g.sa[0] + tg.minA[0], y0 = oh0 * g.sa[1] + tg.minA[1], x0 = ow0 * g.sa[2] + tg.minA[2];
            const bool interior = z0 >= 0 && z0 + EAd <= g.Di && y0 >= 0 && y0 + tg.EAh <= g.Hi && x0 >= 0 && x0 + tg.EAw <= g.Wi;
            const unsigned short *base = asrc + ((((long)n * g.Di + z0) * g.Hi + y0) * g.Wi + x0) * (long)Cs + cofs;
            if (interior) {  // block-uniform
#pragma unroll
                for (int u = 0; u < NA; u++) ra[u] = ld16(base, relA[u]);
            } else {
#pragma unroll
                for (int u = 0; u < NA; u++) {
                    const int c = coordsA(u, tid_here);
                    const int id = z0 + (c >> 16), ih = y0 + ((c >> 8) & 255), iw = x0 + (c & 255);
                    ra[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (c >= 0 && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                        ra[u] = ld16(base, relA[u]);
                }
            }
        }
        {
            const int z0 = od0 * g.sb[0] + tg.minB[0], y0 = oh0 * g.sb[1] + tg.minB[1], x0 = ow0 * g.sb[2] + tg.minB[2];
            const bool interior = z0 >= 0 && z0 + EBd <= g.Db && y0 >= 0 && y0 + tg.EBh <= g.Hb && x0 >= 0 && x0 + tg.EBw <= g.Wb;
            const unsigned short *base = b + ((((long)n * g.Db + z0) * g.Hb + y0) * g.Wb + x0) * (long)K + k0;
            if (interior) {
#pragma unroll
                for (int u = 0; u < NB; u++) rb[u] = ld16(base, relB[u]);
            } else {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int c = coordsB(u, tid_here);
                    const int id = z0 + (c >> 16), ih = y0 + ((c >> 8) & 255), iw = x0 + (c & 255);
                    rb[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (c >= 0 && id >= 0 && id < g.Db && ih >= 0 && ih < g.Hb && iw >= 0 && iw < g.Wb)
                        rb[u] = ld16(base, relB[u]);
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NA; u++) *reinterpret_cast<uint4 *>(As + (size_t)(u * 256 + tid) * 16) = ra[u];
#pragma unroll
        for (int u = 0; u < NB; u++) *reinterpret_cast<uint4 *>(Bs + (size_t)(u * 256 + tid) * 16) = rb[u];
    };
    // transposed-read lane roles: 16-lane group gg = lane >> 4 takes channels 16*(gg&1).. of the voxels of k-half h;
    // lane 4q+p of the group supplies the address of voxel row q, channel columns 4p..4p+3 (8 bytes)
    const int q4 = (lane & 15) >> 2;
    const int colb = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    int aoff[TPW], boff[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        aoff[j] = ta[j] * 64 + colb;
        boff[j] = tb[j] * 64 + colb;
    }
    constexpr int NAV = SH == 2 ? 1 : TPW, NBV = SH == 1 ? 1 : TPW;

    // Per-step operand row offsets (A rows r = 0, 1; B rows r = 0, 1) of every lane, once per workgroup, in LDS: the
    // inner loop used to recompute them -- 56 of its ~90 VALU instructions per 7 MFMAs (PMC round 2: 13 VALU per MFMA,
    // MFMA pipe 33 % busy) -- now it is one ds_read_b128 per step.
    int4 *steptab = reinterpret_cast<int4 *>(lds8 + (size_t)(NA + NB) * 4096);
    for (int e = tid; e < (TV >> 4) * 64; e += 256) {
        const int st = e >> 6, l = e & 63;
        const int hh = l >> 5, qq = (l & 15) >> 2;
        int t4[4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int v = st * 16 + 8 * hh + 4 * r + qq;
            const int wx = v & (tg.TW - 1), hy = (v >> tg.lTW) & (tg.TH - 1), dz = v >> (tg.lTW + tg.lTH);
            t4[r] = (((dz * g.sa[0]) * tg.EAh + hy * g.sa[1]) * tg.EAw + wx * g.sa[2]) * 64;
            t4[2 + r] = (((dz * g.sb[0]) * tg.EBh + hy * g.sb[1]) * tg.EBw + wx * g.sb[2]) * 64;
        }
        steptab[e] = make_int4(t4[0], t4[1], t4[2], t4[3]);
    }
    if (TRI && tid < 16) reinterpret_cast<unsigned *>(lds8 + (size_t)(NA + NB) * 4096 + (size_t)TV * 64)[tid] = 0x3f803f80u;  // bf16 ones

    // Bias gradient (SH == 1: every tap reads the same dy fragment): with 27 taps over 4 waves x 7 slots the last slot of
    // wave 3 is idle -- its A operand becomes a block of ones, so its accumulator rows are the column sums of dy over
    // the voxels of this workgroup's tiles (fp32 accumulation of bf16 values, like the weight gradient itself).  Replaces
    // a separate pass over dy (k_colsum4) per layer.
    const bool ones_slot = SH == 1 && pbias != nullptr && tap_of(TPW - 1) >= g.ntaps;
    bf16x8w ones;
#pragma unroll
    for (int e = 0; e < 8; e++) ones[e] = (__bf16)1.0f;

#if (MVD_WG16_DBG & 64)
    const bool stamp_on = !(MVD_WG16_DBG & 128) && blockIdx.x == 100 && blockIdx.y == 0 && wave == 0 && lane == 0;
    int nst = 0;
#endif
    int tile = split;
    if (tile < tg.ntiles) load_tile(tile);
    while (tile < tg.ntiles) {
        MVD_WGS(0)
        __syncthreads();
        MVD_WGS(1)
        store_tile();
        MVD_WGS(2)
        __syncthreads();
        MVD_WGS(3)
        const int next = tile + tg.nsplit;
        // TRI: the next tile's global loads are issued one or two per MFMA step inside the step loop.  Issued in one
        // burst they cost the burst's time at the L1's 64 B/clk -- 57 KB per tile and workgroup, 2-3 kilocycles of the
        // 11 a tile took with nothing else of this wave in flight.  They are BUFFER loads and unconditional: a lane
        // whose halo voxel lies outside the volume (border tiles) passes an out-of-range offset and gets zeros, and
        // after the last tile the descriptors have zero records (no traffic).  Any branch around a load makes the
        // compiler guard the following loads with s_waitcnt vmcnt(0), one memory latency per step (measured: 14 instead
        // of 6.4 kilocycles per tile).
        bool nint = true;
        __amdgpu_buffer_rsrc_t rA, rB;
        if (TRI) {
            const unsigned short *nbA = a1, *nbB = b;
            const bool more = next < tg.ntiles;
            if (more) nint = tile_bases(next, nbA, nbB);
            rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(nbA), 0, more ? 0x7fffffff : 0, 0x00020000);
            rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(nbB), 0, more ? 0x7fffffff : 0, 0x00020000);
        } else if (next < tg.ntiles) {
            load_tile(next);
        }
        MVD_WGS(4)
        // operands of step s+1 are fetched (step-table read, address adds, transposing reads) before the MFMAs of step s
        // are issued, into the other half of a double buffer: un-pipelined, every step exposed two LDS round trips
        // (table, then operands: s_waitcnt lgkmcnt(0) twice) and 16 address adds in front of its 7 MFMAs
        bf16x8w av[2][NAV], bv[2][NBV];
        auto fetch = [&](int s16, int buf) {
            const int4 t4 = steptab[(s16 >> 4) * 64 + lane];
            const int sa_[2] = {t4.x, t4.y}, sb_[2] = {t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < NAV; j++) {
                if (SH == 1 && j == TPW - 1 && ones_slot) av[buf][j] = ones;  // wave-uniform
                else av[buf][j] = tr_operand(As + sa_[0] + aoff[j], As + sa_[1] + aoff[j]);
            }
#pragma unroll
            for (int j = 0; j < NBV; j++) bv[buf][j] = tr_operand(Bs + sb_[0] + boff[j], Bs + sb_[1] + boff[j]);
        };
        constexpr bool PIPE = NA < 10;  // (the 256-voxel tile has no registers left for the second operand set)
        if constexpr (TRI) {
            // Straight-line code (the step count is a compile-time constant, no wave-dependent branch): the waits the
            // compiler places are then exact counts, i.e. the MFMAs of step s wait for the reads of step s and not
            // for the reads of step s + 1 issued just before them.  (With the run-time loop and the ones-slot branch
            // around the single-tap reads every second step sat behind s_waitcnt lgkmcnt(0): 75 instead of 32 pipe
            // cycles per MFMA and wave in the s_memtime stamps of tools/stamps_wgrad16.py.)
            // raw[buf]: the union reads of the two triples (3 x 4 voxels each), the single tap (2 x 4) and dy (2 x 4)
            constexpr int NSTEP = (NA == 10 ? 256 : 128) / 16;  // == TV / 16 (host-checked)
            s16x4 rt[2][2][3], rs[2][2], rbv[2][2];
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            auto trd = [&](const unsigned char *p_) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p_); };
            // wave 3's seventh slot (bias gradient): every lane reads the same 8 bytes of bf16 ones behind the step table
            const unsigned char *ones_row = lds8 + (size_t)(NA + NB) * 4096 + (size_t)NSTEP * 1024;
            auto fetch3 = [&](const int4 t4, int buf) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    rt[buf][q][0] = trd(As + t4.x + aoff[3 * q]);        // voxels x-1 .. x+2 of the row (tap tx = -1 base)
                    rt[buf][q][1] = trd(As + t4.y + aoff[3 * q]);        // x+3 .. x+6
                    rt[buf][q][2] = trd(As + t4.y + aoff[3 * q] + 256);  // x+7 .. x+10 (the last two are never used)
                }
                rs[buf][0] = trd(ones_slot ? ones_row : As + t4.x + aoff[6]);
                rs[buf][1] = trd(ones_slot ? ones_row : As + t4.y + aoff[6]);
                rbv[buf][0] = trd(Bs + t4.z + boff[0]);
                rbv[buf][1] = trd(Bs + t4.w + boff[0]);
            };
            auto step3 = [&](int buf) {
                const bf16x8w bfrag = __builtin_bit_cast(bf16x8w, __builtin_shufflevector(rbv[buf][0], rbv[buf][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const s16x8 a01 = __builtin_shufflevector(rt[buf][q][0], rt[buf][q][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const s16x8 a12 = __builtin_shufflevector(rt[buf][q][1], rt[buf][q][2], 0, 1, 2, 3, 4, 5, 6, 7);
                    const s16x8 f1 = __builtin_shufflevector(a01, a12, 1, 2, 3, 4, 5, 6, 7, 12);
                    const s16x8 f2 = __builtin_shufflevector(a01, a12, 2, 3, 4, 5, 6, 7, 12, 13);
                    acc[3 * q + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8w, a01), bfrag, acc[3 * q + 0], 0, 0, 0);
                    acc[3 * q + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8w, f1), bfrag, acc[3 * q + 1], 0, 0, 0);
                    acc[3 * q + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8w, f2), bfrag, acc[3 * q + 2], 0, 0, 0);
                }
                const bf16x8w sfrag = __builtin_bit_cast(bf16x8w, __builtin_shufflevector(rs[buf][0], rs[buf][1], 0, 1, 2, 3, 4, 5, 6, 7));
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag, bfrag, acc[6], 0, 0, 0);
            };
            int4 t_nxt = steptab[64 + lane];  // the table row is read two steps ahead of its MFMAs
            fetch3(steptab[lane], 0);
#pragma unroll
            for (int st = 0; st < NSTEP; st++) {
                if (st + 1 < NSTEP) {
                    const int4 t_use = t_nxt;
                    if (st + 2 < NSTEP) t_nxt = steptab[(st + 2) * 64 + lane];
                    fetch3(t_use, (st + 1) & 1);
                }
                {
                    constexpr int LPS = (NA + NB + NSTEP - 1) / NSTEP;  // loads per step
#pragma unroll
                    for (int u = st * LPS; u < (st + 1) * LPS && u < NA + NB; u++) {
                        const bool isA = u < NA;
                        const int ua = isA ? u : 0, ub = isA ? 0 : u - NA;
                        unsigned o = isA ? relA[ua] : relB[ub];
                        if (!nint) {  // block-uniform; VALU only inside
                            int tid_here = tid;
                            asm volatile("" : "+v"(tid_here));
                            const int c = isA ? coordsA(ua, tid_here) : coordsB(ub, tid_here);
                            const int id = (isA ? nza : nzb) + (c >> 16), ih = (isA ? nya : nyb) + ((c >> 8) & 255),
                                      iw = (isA ? nxa : nxb) + (c & 255);
                            const bool ok = c >= 0 && id >= 0 && id < (isA ? g.Di : g.Db) && ih >= 0 && ih < (isA ? g.Hi : g.Hb) &&
                                            iw >= 0 && iw < (isA ? g.Wi : g.Wb);
                            o = ok ? o : 0xffffffffu;
                        }
                        const u32x4w q4 = __builtin_amdgcn_raw_buffer_load_b128(isA ? rA : rB, (int)o, 0, 0);
                        if (isA) ra[ua] = make_uint4(q4[0], q4[1], q4[2], q4[3]);
                        else rb[ub] = make_uint4(q4[0], q4[1], q4[2], q4[3]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the reads to just before their use)
                step3(st & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (PIPE) {
            fetch(0, 0);
            for (int s16 = 0; s16 < TV; s16 += 32) {  // TV is a multiple of 16
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const int sc = s16 + 16 * par;
                    if (sc >= TV) break;
                    if (sc + 16 < TV) fetch(sc + 16, par ^ 1);
#pragma unroll
                    for (int j = 0; j < TPW; j++)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[par][SH == 2 ? 0 : j], bv[par][SH == 1 ? 0 : j], acc[j], 0, 0,
                                                                         0);
                }
            }
        } else {
            for (int s16 = 0; s16 < TV; s16 += 16) {
                fetch(s16, 0);
#pragma unroll
                for (int j = 0; j < TPW; j++)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][SH == 2 ? 0 : j], bv[0][SH == 1 ? 0 : j], acc[j], 0, 0, 0);
            }
        }
        MVD_WGS(5)
#if (MVD_WG16_DBG & 64)
        nst++;
#endif
        tile = next;
    }
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int t = tap_of(j);
        if (t < g.ntaps) {
            float *po = partial + ((size_t)split * g.ntaps + t) * C * K;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                po[(size_t)(c0 + row) * K + k0 + i] = acc[j][r];
            }
        }
    }
    // every row of the ones-slot accumulator holds the same sums: row 0 (lane half 0, register 0) of the c-block-0
    // workgroups goes out, one row per split
    if (ones_slot && cb == 0 && h == 0) pbias[(size_t)split * K + k0 + i] = acc[TPW - 1][0];
}

// dw[torch layout] = sum_split partial[split][t][c][k]   (fp32 partials, fp64 sum, fixed order).
// Block = 64 consecutive elements x 4 split groups: group q sums splits q, q+4, ... (4 loads in flight per thread),
// the four group sums are combined in a fixed order through LDS.
// Blocks past the weight elements (nbias_blocks of them, optional) reduce the bias-gradient rows pbias[row][K] the same
// way -- one launch instead of two per layer (the separate k_dbias_reduce launch cost ~5.5 us for 32-320 sums).
// G = split groups per block (threads = 64 G).  Round 3: with 4 groups a thread walked nsplit / 4 partials, four loads in
// flight -- 16 serial L2 round trips for the 256 partial sets of a 128^3 layer, on a grid of only 27 C K / 64 blocks (432 for
// 32 x 32 channels): 11-22 us per launch, 0.53 ms per bf16 step in 27 launches.  16 groups: four round trips.
template <int G>
__global__ __launch_bounds__(64 * G) void k_wgrad_reduce_f(WgradGeom g, const float *__restrict__ partial,
                                                           float *__restrict__ dw, int nsplit, const float *__restrict__ pbias,
                                                           float *__restrict__ dbias, int nrows, int wblocks) {
    __shared__ double red[G][64];
    const int C = g.C1 + g.C2, K = g.K;
    const long per = (long)g.ntaps * C * K;
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    auto combine = [&]() {  // fixed order: pairs, then pairs of pairs ...
        double t[G];
#pragma unroll
        for (int i = 0; i < G; i++) t[i] = red[i][e];
#pragma unroll
        for (int w = 1; w < G; w *= 2)
#pragma unroll
            for (int i = 0; i + w < G; i += 2 * w) t[i] += t[i + w];
        return t[0];
    };
    if (pbias && (int)blockIdx.x >= wblocks) {  // block-uniform
        const int k = ((int)blockIdx.x - wblocks) * 64 + e;
        double s = 0;
        if (k < K)
            for (int r = q; r < nrows; r += G) s += (double)pbias[(size_t)r * K + k];
        red[q][e] = s;
        __syncthreads();
        if (q == 0 && k < K) dbias[k] = (float)combine();
        return;
    }
    const long j = (long)blockIdx.x * 64 + e;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (j < per) {
        int b = q;
        for (; b + 3 * G < nsplit; b += 4 * G) {
            s0 += (double)partial[(size_t)b * per + j];
            s1 += (double)partial[(size_t)(b + G) * per + j];
            s2 += (double)partial[(size_t)(b + 2 * G) * per + j];
            s3 += (double)partial[(size_t)(b + 3 * G) * per + j];
        }
        for (; b < nsplit; b += G) s0 += (double)partial[(size_t)b * per + j];
    }
    red[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q != 0 || j >= per) return;
    const double s = combine();
    const int k = (int)(j % K);
    const int c = (int)((j / K) % C);
    const int t = g.wt[(int)(j / ((long)K * C))];
    size_t o = g.transposed_out ? ((size_t)c * K + k) * g.T + t : ((size_t)k * C + c) * g.T + t;
    dw[o] = (float)s;
}

static void launch_wgrad_reduce_f(unsigned blocks, hipStream_t s, const WgradGeom &g, const float *partial, float *dw, int nsplit,
                                  const float *pbias, float *dbias, int nrows, int wblocks) {
    static const int g16 = getenv("MVD_WGRAD_REDUCE_G16") ? atoi(getenv("MVD_WGRAD_REDUCE_G16")) : 1;
    if (g16 && nsplit >= 64)
        hipLaunchKernelGGL(k_wgrad_reduce_f<16>, dim3(blocks), dim3(1024), 0, s, g, partial, dw, nsplit, pbias, dbias, nrows, wblocks);
    else
        hipLaunchKernelGGL(k_wgrad_reduce_f<4>, dim3(blocks), dim3(256), 0, s, g, partial, dw, nsplit, pbias, dbias, nrows, wblocks);
}

static int wgrad_max_split(const WgradGeom &g, int per_cu = 2) {
    const int C = g.C1 + g.C2;
    const int ncb = (C + 31) / 32, nkb = (g.K + 31) / 32;
    // workgroups per CU: 1 for the fp32 kernel (register budget of its prefetch), 2 for the bf16 kernel; the
    // workspace query sizes for 2
    long ns = 256L * per_cu / ((long)ncb * nkb);
    if (ns < 1) ns = 1;
    const long per = (long)g.ntaps * C * g.K * 4;
    long cap = (256L << 20) / per;
    if (cap < 1) cap = 1;
    if (ns > cap) ns = cap;
    return (int)ns;
}

static bool wgrad_mfma_ok(const WgradGeom &g) {
    const int C = g.C1 + g.C2;
    if (g.ntaps < 1 || g.ntaps > 27) return false;
    if (g.C1 % 4 != 0 || g.C2 % 4 != 0 || g.K % 4 != 0) return false;
    if (g.C2 > 0 && g.C1 % 32 != 0) return false;
    if (C < 4 || g.K < 4) return false;
    return true;
}

size_t wgrad_mfma_ws(const WgradGeom &g) {
    if (!wgrad_mfma_ok(g)) return 0;
    // split-K partials + the bias-gradient rows (two lane halves per partial)
    return (size_t)wgrad_max_split(g) * (g.ntaps * (size_t)(g.C1 + g.C2) + 2) * g.K * sizeof(float) + 256;
}

// k_wgrad16z + the split reduce.  -1 = not that kernel's problem.  in_scale / in_shift (optional, [N][C1] fp32): the loader
// prologue -- x is the RAW output of the producing conv and the operand is lrelu(x * scale + shift) (ops.NormActConv3dFn)
int wgrad16z_run(const WgradGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *b, float *dw, void *ws,
                 size_t ws_bytes, hipStream_t s, float *dbias, int *dbias_done, const float *in_scale, const float *in_shift,
                 float slope) {
    int nsplit_z = 0;
    float *pbias_z = nullptr;
    const int C = g.C1 + g.C2;
    const int rz = wgrad16z(g, a1, a2, b, ws, ws_bytes, dbias && dbias_done, &nsplit_z, &pbias_z, s, in_scale, in_shift, slope);
    if (rz != 0) return rz;
    float *partial_z = reinterpret_cast<float *>(ws);
    const int wblocks = (int)cdiv((long)27 * C * g.K, 64);
    if (pbias_z) {
        launch_wgrad_reduce_f((unsigned)(wblocks + cdiv(g.K, 64)), s, g, partial_z, dw, nsplit_z, pbias_z, dbias,
                           nsplit_z, wblocks);
        *dbias_done = 1;
    } else {
        launch_wgrad_reduce_f((unsigned)(wblocks), s, g, partial_z, dw, nsplit_z, (const float *)nullptr,
                           (float *)nullptr, 0, 0);
    }
    return check_launch("conv wgrad reduce (bf16 z-marching)");
}

int wgrad_mfma(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws, size_t ws_bytes,
               hipStream_t s, bool bf16_in, float *dbias, int *dbias_done) {
    if (dbias_done) *dbias_done = 0;
    if (!wgrad_mfma_ok(g)) return -1;
    if (((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)b) & 15) return -1;
    if (bf16_in && (g.C1 % 32 != 0 || g.C2 % 32 != 0 || g.K % 32 != 0)) return -1;
    const int C = g.C1 + g.C2;
    if (bf16_in) {  // plain 3x3x3 stride 1 on volumes at least 32 wide: the z-marching kernel (conv_bf16w.hip)
        const int rz = wgrad16z_run(g, reinterpret_cast<const unsigned short *>(a1), reinterpret_cast<const unsigned short *>(a2),
                                    reinterpret_cast<const unsigned short *>(b), dw, ws, ws_bytes, s, dbias, dbias_done, nullptr,
                                    nullptr, 0.f);
        if (rz >= 0) return rz;
        int nsplit_s = 0;   // stride 2: k_wgrad16zs (no bias rows: dbias_done stays 0, the caller runs the column sums)
        const int rs2 = wgrad16zs(g, reinterpret_cast<const unsigned short *>(a1), reinterpret_cast<const unsigned short *>(a2),
                                  reinterpret_cast<const unsigned short *>(b), ws, ws_bytes, &nsplit_s, s);
        if (rs2 > 0) return rs2;
        if (rs2 == 0) {
            launch_wgrad_reduce_f((unsigned)cdiv((long)27 * C * g.K, 64), s, g, reinterpret_cast<float *>(ws), dw, nsplit_s,
                                  (const float *)nullptr, (float *)nullptr, 0, 0);
            return check_launch("conv wgrad reduce (bf16 z-marching, stride 2)");
        }
    }
    WgTile tg;
    memset(&tg, 0, sizeof(tg));
    tg.dbg = getenv("MVD_CONV_DBG") ? atoi(getenv("MVD_CONV_DBG")) : 0;
    int mnA[3] = {127, 127, 127}, mxA[3] = {-127, -127, -127}, mnB[3] = {127, 127, 127}, mxB[3] = {-127, -127, -127};
    for (int t = 0; t < g.ntaps; t++)
        for (int a = 0; a < 3; a++) {
            if (g.off[t][a] < mnA[a]) mnA[a] = g.off[t][a];
            if (g.off[t][a] > mxA[a]) mxA[a] = g.off[t][a];
            if (g.ob[t][a] < mnB[a]) mnB[a] = g.ob[t][a];
            if (g.ob[t][a] > mxB[a]) mxB[a] = g.ob[t][a];
        }
    // tile shapes (powers of two), largest first.  Two register/LDS configurations: (NA, NB) = (13, 4) for halo-heavy
    // A (3x3x3 convs) and (2, 16) for halo-heavy B (transposed convs); both leave room for two workgroups per CU.
    const int cand[5][3] = {{4, 8, 8}, {2, 8, 8}, {2, 4, 8}, {2, 4, 4}, {1, 4, 4}};
    // bf16, 3x3x3 stride 1 on large volumes: 256-voxel tiles (MVD_WGRAD16_BIG=0: 128) -- half the barriers and staging
    // per MFMA, a 6x10x10 halo for 256 voxels instead of 4x10x10 for 128
    static const int big16 = getenv("MVD_WGRAD16_BIG") ? atoi(getenv("MVD_WGRAD16_BIG")) : 1;
    int cfg = -1;
    for (int ci = 0; ci < 5 && cfg < 0; ci++) {
        const int T3[3] = {cand[ci][0], cand[ci][1], cand[ci][2]};
        if (ci == 0) {
            bool plain27 = bf16_in && big16 && g.ntaps == 27 && (long)g.N * g.Do * g.Ho * g.Wo >= (1L << 15);
            for (int a = 0; a < 3; a++) plain27 = plain27 && g.sa[a] == 1 && g.sb[a] == 1;
            for (int t = 1; t < g.ntaps && plain27; t++)
                for (int a = 0; a < 3; a++) plain27 = plain27 && g.ob[t][a] == g.ob[0][a];
            if (!plain27) continue;
        }
        int EA[3], EB[3];
        for (int a = 0; a < 3; a++) {
            EA[a] = (T3[a] - 1) * g.sa[a] + (mxA[a] - mnA[a]) + 1;
            EB[a] = (T3[a] - 1) * g.sb[a] + (mxB[a] - mnB[a]) + 1;
        }
        const int nA = EA[0] * EA[1] * EA[2], nB = EB[0] * EB[1] * EB[2];
        int c = -1;
        if (bf16_in) {  // k_wgrad16: 64 slots per uint4-per-thread; (NA, NB) = (10, 4) [256-voxel tile], (7, 2) or (1, 8)
            if (ci == 0) {
                if (nA <= 10 * 64 && nB <= 4 * 64) c = 2;
            } else if (nA <= 7 * 64 && nB <= 2 * 64) c = 0;
            else if (nA <= 1 * 64 && nB <= 8 * 64) c = 1;
        } else if (nA * 8 <= 13 * 256 && nB * 8 <= 4 * 256) c = 0;   // k_wgrad_mfma<.., 7, 2, ..>: 7 x 512 / 2 x 512 float4
        else if (nA * 8 <= 2 * 256 && nB * 8 <= 16 * 256) c = 1;      // k_wgrad_mfma<.., 1, 8, ..>
        if (c < 0) continue;
        cfg = c;
        tg.TD = T3[0]; tg.TH = T3[1]; tg.TW = T3[2];
        tg.lTH = T3[1] == 8 ? 3 : 2;
        tg.lTW = T3[2] == 8 ? 3 : 2;
        tg.EAh = EA[1]; tg.EAw = EA[2]; tg.nslotsA = nA;
        tg.EBh = EB[1]; tg.EBw = EB[2]; tg.nslotsB = nB;
    }
    if (cfg < 0) return -1;
    {   // 16-bit reciprocal multipliers, verified exhaustively for the slot range they are used on
        auto magic = [](int d, int nmax) -> int {
            int m = (1 << 16) / d + 1;
            for (int n = 0; n < nmax; n++)
                if (((n * m) >> 16) != n / d) return -1;
            return m;
        };
        const int EAhw = tg.EAh * tg.EAw, EBhw = tg.EBh * tg.EBw;
        const int nmaxA = bf16_in ? (cfg == 2 ? 10 : cfg == 0 ? 7 : 1) * 64 : (cfg == 0 ? 13 : 2) * 32;
        const int nmaxB = bf16_in ? (cfg == 2 ? 4 : cfg == 0 ? 2 : 8) * 64 : (cfg == 0 ? 4 : 16) * 32;
        tg.magAhw = magic(EAhw, nmaxA); tg.magAw = magic(tg.EAw, EAhw);
        tg.magBhw = magic(EBhw, nmaxB); tg.magBw = magic(tg.EBw, EBhw);
        if (tg.magAhw < 0 || tg.magAw < 0 || tg.magBhw < 0 || tg.magBw < 0) return -1;
    }
    for (int a = 0; a < 3; a++) {
        tg.minA[a] = mnA[a];
        tg.minB[a] = mnB[a];
    }
    for (int t = 0; t < g.ntaps; t++) {
        tg.toffA[t] = ((g.off[t][0] - mnA[0]) * tg.EAh + (g.off[t][1] - mnA[1])) * tg.EAw + (g.off[t][2] - mnA[2]);
        tg.toffB[t] = ((g.ob[t][0] - mnB[0]) * tg.EBh + (g.ob[t][1] - mnB[1])) * tg.EBw + (g.ob[t][2] - mnB[2]);
    }
    tg.ntd = (g.Do + tg.TD - 1) / tg.TD;
    tg.nth = (g.Ho + tg.TH - 1) / tg.TH;
    tg.ntw = (g.Wo + tg.TW - 1) / tg.TW;
    const long ntiles = (long)g.N * tg.ntd * tg.nth * tg.ntw;
    if (ntiles > (1L << 30)) return -1;
    tg.ntiles = (int)ntiles;
    const int ncb = (C + 31) / 32;
    tg.nkb = (g.K + 31) / 32;
    // fp32: one workgroup per CU (two co-resident ones measured no faster: 0.96 vs 0.94 ms on 32->64 @128^3 stride 2)
    long ns = wgrad_max_split(g, bf16_in ? 2 : 1);
    if (ns > ntiles) ns = ntiles;
    tg.nsplit = (int)ns;
    const size_t need_ws = (size_t)tg.nsplit * g.ntaps * C * g.K * sizeof(float);
    if (ws_bytes < need_ws) {
        set_error("conv wgrad (mfma): workspace too small (%zu < %zu)", ws_bytes, need_ws);
        return 1;
    }
    if ((long)ncb * tg.nkb > 65535) return -1;
    float *partial = reinterpret_cast<float *>(ws);
    const int tpw = (g.ntaps + 3) / 4;
    dim3 grid(tg.nsplit, ncb * tg.nkb);
    static const int db_mfma = getenv("MVD_WGRAD_DB") ? atoi(getenv("MVD_WGRAD_DB")) : 1;  // two LDS images (k_wgrad_mfma)
    if (!bf16_in) tg.dbg = (tg.dbg & ~16) | (db_mfma ? 0 : 16);
#define WG_LAUNCH(TPW, NA, NB, SH)                                                                                   \
    {                                                                                                              \
        auto kern = k_wgrad_mfma<TPW, NA, NB, SH>;                                                                 \
        const size_t lds = (size_t)(NA + NB) * 8192 * (db_mfma ? 2 : 1);                                           \
        static PerDeviceFlag cfgd;                                                                                  \
        if (!cfgd()) {                                                                                               \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)LDS_LIMIT) != hipSuccess) {                                               \
                set_error("conv wgrad (mfma): cannot raise the dynamic LDS limit");                                \
                return 1;                                                                                          \
            }                                                                                                      \
            cfgd() = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, g, tg, a1, a2, b, partial, (SH) == 1 ? pbias_g : nullptr); \
    }
    bool sameA = true, sameB = true;
    for (int t = 1; t < g.ntaps; t++) {
        sameA = sameA && tg.toffA[t] == tg.toffA[0];
        sameB = sameB && tg.toffB[t] == tg.toffB[0];
    }
    if (bf16_in) {
        const unsigned short *h1 = reinterpret_cast<const unsigned short *>(a1);
        const unsigned short *h2 = reinterpret_cast<const unsigned short *>(a2);
        const unsigned short *hb = reinterpret_cast<const unsigned short *>(b);
        // bias gradient from the idle 28th tap slot (k_wgrad16): 27 taps, 7 slots per wave, one dy fragment for all taps
        float *pbias16 = nullptr;
        if (dbias && dbias_done && (cfg == 0 || cfg == 2) && sameB && tpw == 7 && g.ntaps < 28 &&
            need_ws + (size_t)tg.nsplit * g.K * sizeof(float) <= ws_bytes)
            pbias16 = partial + need_ws / sizeof(float);
        // TRI: the x-triples of the plain 27-tap stride-1 gather share their halo-row reads (see k_wgrad16)
        static int tri_env = -1;
        if (tri_env < 0) tri_env = getenv("MVD_WGRAD16_TRI") ? atoi(getenv("MVD_WGRAD16_TRI")) : 1;
        bool tri = tri_env != 0 && sameB && tpw == 7 && g.ntaps == 27 && tg.TW == 8 && g.sa[2] == 1 &&
                   tg.TD * tg.TH * tg.TW == (cfg == 2 ? 256 : 128);
        for (int t = 0; t < 27 && tri; t++) tri = tg.toffA[t] == tg.toffA[t - t % 3] + t % 3;
        // (the third read of a triple runs two slots past its 10-slot halo row: inside the A image except for the very
        // last row, where it reads the first bytes of the B image -- both inside the allocation, values never used)
#define WG16T(TPW, NA, NB, SH, TRI)                                                                                          \
    hipLaunchKernelGGL((k_wgrad16<TPW, NA, NB, SH, TRI>), grid, dim3(256),                                                  \
                       (size_t)(NA + NB) * 4096 + (size_t)(tg.TD * tg.TH * tg.TW) * 64 + ((TRI) ? 64 : 0), s, g, tg, h1, h2, \
                       hb, partial,                                                                                          \
                       (TPW) == 7 && (SH) == 1 ? pbias16 : nullptr)
#define WG16(TPW, NA, NB, SH) WG16T(TPW, NA, NB, SH, false)
#define WG16_TPW(NA, NB, SH)                  \
    {                                         \
        if (tpw <= 1) WG16(1, NA, NB, SH);    \
        else if (tpw == 2) WG16(2, NA, NB, SH); \
        else if (tpw <= 4) WG16(4, NA, NB, SH); \
        else WG16(7, NA, NB, SH);             \
    }
        if (cfg == 2) {  // (chosen only for 27 taps, stride 1, one dy slot for all taps: sameB, tpw == 7)
            static PerDeviceFlag cfgd_big;
            if (!cfgd_big()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad16<7, 10, 4, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT) != hipSuccess) {
                    set_error("conv wgrad (bf16 mfma): cannot raise the dynamic LDS limit");
                    return 1;
                }
                cfgd_big() = true;
            }
            if (tri) {
                static PerDeviceFlag cfgd_tri;
                if (!cfgd_tri()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad16<7, 10, 4, 1, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT) != hipSuccess) {
                        set_error("conv wgrad (bf16 mfma): cannot raise the dynamic LDS limit");
                        return 1;
                    }
                    cfgd_tri() = true;
                }
                WG16T(7, 10, 4, 1, true);
            } else {
                WG16(7, 10, 4, 1);
            }
        } else if (cfg == 0) {
            if (sameB && tri) WG16T(7, 7, 2, 1, true);
            else if (sameB) WG16_TPW(7, 2, 1)
            else WG16_TPW(7, 2, 0)
        } else {
            if (sameA) WG16_TPW(1, 8, 2)
            else WG16_TPW(1, 8, 0)
        }
#undef WG16_TPW
#undef WG16
#undef WG16T
        if (check_launch("conv wgrad (bf16 mfma)")) return 1;
        const long per16 = (long)g.ntaps * C * g.K;
        const int wblocks = (int)cdiv(per16, 64);
        if (pbias16) {  // the bias rows ride in the same reduce launch
            launch_wgrad_reduce_f((unsigned)(wblocks + cdiv(g.K, 64)), s, g, partial, dw, tg.nsplit, pbias16,
                               dbias, tg.nsplit, wblocks);
            *dbias_done = 1;
        } else {
            launch_wgrad_reduce_f((unsigned)(wblocks), s, g, partial, dw, tg.nsplit, (const float *)nullptr,
                               (float *)nullptr, 0, 0);
        }
        return check_launch("conv wgrad reduce (bf16 mfma)");
    }
    const int wino_off = wino_mode() == 0;
    if (!bf16_in && !wino_off && cfg == 0 && g.ntaps == 27 && g.T == 27 && !g.transposed_out && g.C1 % 32 == 0 &&
        g.C2 % 32 == 0 && g.K % 32 == 0 && tg.TD == 2) {
        bool plain = true;
        for (int a = 0; a < 3; a++) plain = plain && g.sa[a] == 1 && g.sb[a] == 1;
        for (int t = 0; t < 27 && plain; t++)
            plain = g.wt[t] == t && g.off[t][0] == t / 9 - 1 && g.off[t][1] == (t / 3) % 3 - 1 && g.off[t][2] == t % 3 - 1 &&
                    g.ob[t][0] == 0 && g.ob[t][1] == 0 && g.ob[t][2] == 0;
        const size_t need_w = (size_t)tg.nsplit * 36 * C * g.K * sizeof(float);
        if (plain && need_w <= ws_bytes) {
            // Winograd F(2,3)-transposed weight gradient: 36 position tiles over pairs instead of 27 taps over voxels
            const size_t need_m2 = (size_t)tg.nsplit * 48 * C * g.K * sizeof(float);
            const size_t need_w2 = need_m2 + (size_t)tg.nsplit * 2 * g.K * sizeof(float);
            if (wino_mode() == 2 && tg.TH == 8 && tg.TW == 8 && tg.EAh == 10 && tg.EAw == 10 && tg.EBh == 8 && tg.EBw == 8 &&
                need_w2 <= ws_bytes) {
                float *pbias = (dbias && dbias_done) ? partial + need_m2 / sizeof(float) : nullptr;
                // MVD_WGRAD_W8: 12 (default) = twelve waves (one plane per wave, 3 per SIMD), 1 = eight waves, 0 = four waves
                static const int w8 = getenv("MVD_WGRAD_W8") ? atoi(getenv("MVD_WGRAD_W8")) : 12;
                // MVD_WGRAD_DB=0: single LDS image (barrier / write / barrier per tile)
                static const int db12 = getenv("MVD_WGRAD_DB") ? atoi(getenv("MVD_WGRAD_DB")) : 1;
                auto kern2 = w8 == 12 ? (db12 ? k_wgrad_wino2w12<5, 2, true> : k_wgrad_wino2w12<5, 2, false>)
                                      : (w8 ? k_wgrad_wino2w8<7, 2> : k_wgrad_wino2<13, 4>);
                static PerDeviceFlag cfgd_w2;
                if (!cfgd_w2()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)LDS_LIMIT) != hipSuccess) {
                        set_error("conv wgrad (winograd 2-D): cannot raise the dynamic LDS limit");
                        return 1;
                    }
                    cfgd_w2() = true;
                }
                hipLaunchKernelGGL(kern2, grid, dim3(w8 == 12 ? 768 : (w8 ? 512 : 256)),
                                   w8 == 12 ? (db12 ? (size_t)2 * (4 * 100 + 2 * 64) * 128 : (size_t)(5 + 2) * 12288)
                                            : (w8 ? (size_t)(7 + 2) * 8192 : (size_t)(13 + 4) * 4096), s, g,
                                   tg, a1, a2, b, partial, pbias);
                if (check_launch("conv wgrad (winograd 2-D)")) return 1;
                if (pbias) {
                    hipLaunchKernelGGL(k_dbias_reduce, dim3(cdiv(g.K, 64)), dim3(1024), 0, s, pbias, dbias, g.K, tg.nsplit * 2);
                    if (check_launch("conv wgrad dbias reduce")) return 1;
                    *dbias_done = 1;
                }
                hipLaunchKernelGGL(k_wgrad_reduce_wino2, dim3(cdiv((long)3 * C * g.K, 64)), dim3(1024), 0, s, partial, dw, C,
                                   g.K, tg.nsplit);
                return check_launch("conv wgrad reduce (winograd 2-D)");
            }
            auto kern = tg.TW == 8 ? k_wgrad_wino<13, 4, 4> : k_wgrad_wino<13, 4, 2>;
            static PerDeviceFlag cfgd_w[2];
            if (!cfgd_w[tg.TW == 8]()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_LIMIT) != hipSuccess) {
                    set_error("conv wgrad (winograd): cannot raise the dynamic LDS limit");
                    return 1;
                }
                cfgd_w[tg.TW == 8]() = true;
            }
            hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)(13 + 4) * 4096, s, g, tg, a1, a2, b, partial);
            if (check_launch("conv wgrad (winograd)")) return 1;
            hipLaunchKernelGGL(k_wgrad_reduce_wino, dim3(cdiv((long)9 * C * g.K, 64)), dim3(256), 0, s, partial, dw, C, g.K,
                               tg.nsplit);
            return check_launch("conv wgrad reduce (winograd)");
        }
    }
    if (cfg == 0 && sameB && g.C2 == 0 && C <= 8 && C % 4 == 0 && g.ntaps * C <= 128 && tg.nslotsA * (C / 4) <= 2 * 256) {
        // narrow-input layer: rows of the GEMM are (tap, channel) pairs
        long ns2 = 512 / tg.nkb;
        if (ns2 < 1) ns2 = 1;
        if (ns2 > tg.nsplit * 2) ns2 = tg.nsplit * 2;
        if (ns2 > ntiles) ns2 = ntiles;
        if ((size_t)ns2 * g.ntaps * C * g.K * sizeof(float) <= ws_bytes) tg.nsplit = (int)ns2;
        auto kern = k_wgrad_smallc<2, 4>;
        const size_t lds2 = (size_t)(2 + 4) * 4096 + 32 * 4;
        hipLaunchKernelGGL(kern, dim3(tg.nsplit, tg.nkb), dim3(256), lds2, s, g, tg, a1, b, partial);
        if (check_launch("conv wgrad (mfma, narrow input)")) return 1;
        const long per2 = (long)g.ntaps * C * g.K;
        launch_wgrad_reduce_f((unsigned)(cdiv(per2, 64)), s, g, partial, dw, tg.nsplit, (const float *)nullptr,
                           (float *)nullptr, 0, 0);
        return check_launch("conv wgrad reduce (mfma)");
    }
#define WG_TPW(NA, NB, SH)                     \
    {                                          \
        if (tpw <= 1) WG_LAUNCH(1, NA, NB, SH) \
        else if (tpw == 2) WG_LAUNCH(2, NA, NB, SH) \
        else if (tpw <= 4) WG_LAUNCH(4, NA, NB, SH) \
        else WG_LAUNCH(7, NA, NB, SH)          \
    }
    // the generic kernel runs two wave groups per workgroup, each with its own split-K partial
    if ((size_t)2 * tg.nsplit * g.ntaps * C * g.K * sizeof(float) > ws_bytes) {
        set_error("conv wgrad (mfma): workspace too small for two partials per split");
        return 1;
    }
    // bias gradient inside the kernel when every tap reads the same dy slot and the rows fit behind the partials
    const size_t need_g = (size_t)2 * tg.nsplit * g.ntaps * C * g.K * sizeof(float);
    float *pbias_g = nullptr;
    if (dbias && dbias_done && cfg == 0 && sameB && need_g + (size_t)4 * tg.nsplit * g.K * sizeof(float) <= ws_bytes)
        pbias_g = partial + need_g / sizeof(float);
    if (cfg == 0) {
        if (sameB) WG_TPW(7, 2, 1)
        else WG_TPW(7, 2, 0)
    } else {
        if (sameA) WG_TPW(1, 8, 2)
        else WG_TPW(1, 8, 0)
    }
#undef WG_TPW
#undef WG_LAUNCH
    if (check_launch("conv wgrad (mfma)")) return 1;
    const long per = (long)g.ntaps * C * g.K;
    const int wblocks = (int)cdiv(per, 64);
    if (pbias_g) {  // the bias rows ride in the same reduce launch
        launch_wgrad_reduce_f((unsigned)(wblocks + cdiv(g.K, 64)), s, g, partial, dw, 2 * tg.nsplit,
                           (const float *)pbias_g, dbias, tg.nsplit * 4, wblocks);
        *dbias_done = 1;
    } else {
        launch_wgrad_reduce_f((unsigned)(wblocks), s, g, partial, dw, 2 * tg.nsplit, (const float *)nullptr,
                           (float *)nullptr, 0, 0);
    }
    return check_launch("conv wgrad reduce (mfma)");
}

}  // namespace mvd
