// bf16 mixed-precision forward-type conv engine (BASELINE cfg 4/5: bf16 activations, fp32 master weights and
// accumulation).  v_mfma_f32_32x32x16_bf16: A[row r][k = 8h+j], B[k = 8h+j][col r] (8 bf16 = one 16-byte fragment per
// lane and k-step), fp32 accumulate, 16x the fp32 MFMA rate -- at 32..64 output channels the kernel is bound by LDS /
// L2 bandwidth, not by the matrix pipe.
//
// Same gathered-tap GEMM as conv_mfma.hip (FwdGeom), C % 32 == 0, K % 32 == 0, any gather stride:
//   * activations NDHWC bf16: a 32-channel chunk of a voxel is 64 B; halo tile in LDS [slots][32 ch] padded to 80 B;
//   * packed weights bf16 [chunk][tap][s][h][k][8] (s = 16-channel k-step, h = lane half): the B fragment of lane
//     (r, h) is one 16-byte read; tap groups stream through a double-buffered LDS ring, next group prefetched into
//     registers under the current group's MFMAs;
//   * output bf16 (fp32 bias added in the epilogue); skinny problems split the reduce chunks and go through fp32
//     partials (k_split_reduce16).
#include <stdlib.h>

#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Epilogue store widening (guide T21): after a swapped-operand 32x32 MFMA chain lane (i, h) holds, per channel group
// rg, the packed bf16 x 4 of channels 8rg + 4h .. +3 of voxel i.  One v_permlane32_swap per dword turns the pair of
// groups (k, k+1) into 16 contiguous bytes per lane: lanes 0-31 get channels 8k .. 8k+7, lanes 32-63 channels
// 8k+8 .. 8k+15 -> one 16-byte store at byte offset 16k (+16 for the upper half) instead of two 8-byte ones.  The
// 8-byte stores are issue-bound (~7 B/clk/CU), not bandwidth-bound.
__device__ inline uint4 pair_store_image(uint2 a /* group k */, uint2 b /* group k+1 */) {
    auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    return make_uint4(rx[0], ry[0], rx[1], ry[1]);
}
// img + old, element-wise on 8 packed bf16 (fp32 add, one rounding): the accumulate form of the epilogue (FwdGeom::acc)
__device__ inline uint4 add_bf16x8(uint4 a, uint4 b) {
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    unsigned r[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float lo = __uint_as_float(av[e] << 16) + __uint_as_float(bv[e] << 16);
        const float hi = __uint_as_float(av[e] & 0xffff0000u) + __uint_as_float(bv[e] & 0xffff0000u);
        r[e] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}
__device__ inline uint2 pack_bf16x4(float a, float b, float c, float d) {
    uint2 q;
    q.x = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
    q.y = (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16);
    return q;
}

__host__ __device__ inline size_t widx16(int T, int K, int t, int c, int k) {
    const int cc = c >> 5, r = c & 31, s = r >> 4, h = (r >> 3) & 1, e = r & 7;
    return ((((((size_t)cc * T + t) * 2 + s) * 2 + h) * K + k) << 3) + e;
}

// torch fp32 [K][C][T] (or transposed-conv [C][K][T]) -> wf16 (reduce C, produce K) / wb16 (reduce K, produce C)
__global__ void k_pack_weight16(const float *__restrict__ w, unsigned short *__restrict__ wf, unsigned short *__restrict__ wb,
                                int K, int C, int T, int transposed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)K * C * T;
    if (i >= total) return;
    int k = i % K;
    int c = (i / K) % C;
    int t = i / ((long)K * C);
    float v = transposed ? w[((size_t)c * K + k) * T + t] : w[((size_t)k * C + c) * T + t];
    unsigned short b = f2bf(v);
    if (wf) wf[widx16(T, K, t, c, k)] = b;
    if (wb) wb[widx16(T, C, t, k, c)] = b;
}

struct Fwd16Tile {
    int EH, EW, nslots;
    int magW, magHW;
    int min_off[3];
    int ntd, nth, ntw, nkb, S;
    int nitems;
    int K;
    int toff[27];
    signed char tdy[28];  // halo row shift of tap t (off[t][1] - min_off[1]): selects the XOR swizzle of the read
};

// XR: uint4 (8 bf16) per thread of the halo tile = ceil(nslots*4/256)
// SWZ (unit gather stride in every axis): halo slots are the bare 64-byte voxel rows with the four 16-byte parts XOR-swizzled
// by (halo y-row & 3) -- conflict-free ds_read_b128 for every tap (see k_fwd16q); the padded 80-byte slots of the
// strided layers are 3-way conflicted on unit-stride problems (PMC round 2: as many conflict cycles as access cycles).
#ifndef MVD_F16_DBG
#define MVD_F16_DBG 0
#endif
template <int NT, int MT, int TG, int XR, bool SWZ, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_fwd16(const FwdGeom g, const Fwd16Tile tg, const unsigned short *__restrict__ a1,
                                                  const unsigned short *__restrict__ a2,
                                                  const unsigned short *__restrict__ w, const float *__restrict__ bias,
                                                  unsigned short *__restrict__ y1, unsigned short *__restrict__ y2,
                                                  float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    constexpr int KT = 32 * NT;
    constexpr int TPB = NW * 64;                 // threads: wave w owns output plane od0 + w of the NW x (4 MT) x 8 tile
    constexpr int XS = SWZ ? 64 : 80;            // bytes per halo slot (64 + 16 pad when not swizzled)
    constexpr int WROW = 16;                     // bytes per (tap, s, h, k) weight fragment
    constexpr int WR = TG * 4 * KT / TPB;        // uint4 per thread per weight group (TG*2*2*KT fragments)
    constexpr int WBUF = TG * 4 * KT * WROW;     // bytes per weight buffer
    static_assert((TG * 4 * KT) % TPB == 0, "weight group must be a multiple of the thread count");
    unsigned char *Xs = lds8;
    unsigned char *Wsm = lds8 + (size_t)XR * (TPB / 4) * XS;  // halo buffer sized for XR*TPB/4 >= nslots slots
    // Per-tap tables across the lanes of a wave (round 2): lane t holds the halo byte offset of tap t and its index in
    // the packed weights; the loops read them back with v_readlane (a few cycles, result in an SGPR).  Indexed out of
    // the kernel argument they were an s_load (+ a byte load) per tap INSIDE the MFMA loop, each followed by
    // s_waitcnt lgkmcnt(0) -- SMEM returns out of order, so the wait also drained every operand read in flight.  (A
    // first fix kept the tables in LDS: in-kernel s_memtime stamps then showed the three weight loads of a group taking
    // ~900 cycles to ISSUE -- three serial LDS round trips, each in front of a 64-bit address computation.)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= tg.nitems) return;
    unsigned r_ = (unsigned)item;
    const int kb = (int)(r_ % (unsigned)tg.nkb); r_ /= (unsigned)tg.nkb;
    const int split = (int)(r_ % (unsigned)tg.S); r_ /= (unsigned)tg.S;
    const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
    const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
    const int td_ = (int)(r_ % (unsigned)tg.ntd);
    const int n = (int)(r_ / (unsigned)tg.ntd);

    const int ltc = lane < g.ntaps ? lane : g.ntaps - 1;
    const int toff_l = tg.toff[ltc < 27 ? ltc : 26] * XS;                 // (clamped to the last tap beyond ntaps)
    const int wt_l = (int)g.wt[ltc < 27 ? ltc : 26];  // (slots past the last tap load the last tap's weights: their MFMAs are skipped)
    const int C = g.C1 + g.C2;
    const int nch = C / 32;
    const int ngroups = (g.ntaps + TG - 1) / TG;
    const int EHW = tg.EH * tg.EW;
    const int nx = tg.nslots * 4;
    const int od0 = td_ * NW, oh0 = th_ * (4 * MT), ow0 = tw_ * 8;
    const int iz0 = od0 * g.sa[0] + tg.min_off[0], iy0 = oh0 * g.sa[1] + tg.min_off[1], ix0 = ow0 * g.sa[2] + tg.min_off[2];

    int sbase[MT];
#pragma unroll
    for (int m = 0; m < MT; m++)
        sbase[m] = (((wave * g.sa[0]) * tg.EH + (4 * m + (i >> 3)) * g.sa[1]) * tg.EW + (i & 7) * g.sa[2]) * XS +
                   (SWZ ? 0 : h * 16);
    const int ylq = (i >> 3) & 3;  // (row of this lane inside its M tile) & 3; M tiles start at multiples of 4 rows
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < NT; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][q][r] = 0.f;

    // weight group: fragment index f = u*256 + tid = ((tl*2 + s)*2 + hh)*KT + k ; tl is a compile-time function of u
    uint4 wrA[WR], wrB[WR];  // two groups of weights in flight (round 2): a group's loads have two groups of MFMAs to land
#pragma unroll
    for (int u = 0; u < WR; u++) wrA[u] = wrB[u] = make_uint4(0, 0, 0, 0);  // (defined on every path: stays in registers)
    unsigned woff[WR];
    constexpr int FPT = 4 * KT;  // fragments per tap
#pragma unroll
    for (int u = 0; u < WR; u++) {
        const int f = (u * TPB + tid) % FPT;
        const int k = f % KT, sh = f / KT;  // sh = s*2 + hh
        woff[u] = (unsigned)((sh * tg.K + kb * KT + k) * 16);  // BYTE offset inside a (chunk, tap) block of 4*K*8 elements
    }
    auto load_w = [&](int cc, int gidx, uint4(&dst)[WR]) {
#pragma unroll
        for (int u = 0; u < WR; u++) {
            // tap of slot u: FPT = 256 -> u; FPT = 128 -> 2u + (tid >> 7): uniform per wave either way
            const int t = __builtin_amdgcn_readfirstlane(gidx * TG + (u * TPB + tid) / FPT);  // < 32
            const int wt = __builtin_amdgcn_readlane(wt_l, t);
            // unconditional: with a branch around the load the compiler loses the vmcnt count and waits for vmcnt(0) in
            // front of every weight store -- i.e. also for the group that was requested last.  Scalar base + 32-bit lane
            // offset (the empty asm keeps the zero-extension next to the load: saddr addressing instead of a 64-bit
            // multiply-add per load -- ~30 instructions per load made the three loads of a group take ~500 cycles to issue)
            const char *wb_ = reinterpret_cast<const char *>(w + ((size_t)cc * g.T + wt) * 4 * tg.K * 8);
            asm volatile("" : "+v"(woff[u]));
            dst[u] = *reinterpret_cast<const uint4 *>(wb_ + woff[u]);
        }
    };
    auto store_w = [&](int buf, const uint4(&src)[WR]) {
#pragma unroll
        for (int u = 0; u < WR; u++)
            *reinterpret_cast<uint4 *>(Wsm + (size_t)buf * WBUF + (size_t)(u * TPB + tid) * WROW) = src[u];
    };

    const int cc_begin = split * nch / tg.S, cc_end = (split + 1) * nch / tg.S;
    // every workgroup streams the same few hundred KB of weights from L2, and workgroups launched together walk them in
    // step: the tap groups are taken in an order rotated by the work item, so that at any time the CUs of an XCD ask for
    // different lines (ablation round 2: the weight path was 64 of the 118 us this kernel needs WITHOUT its MFMAs on
    // 64 -> 64 @64^3 -- 450 MB of L2 reads per launch)
#if (MVD_F16_DBG & 64)
    long long *stamps = reinterpret_cast<long long *>(part);  // diagnostic build only: in-kernel s_memtime stamps of one wave
    int nst = 0;
#endif
    // halo of one 32-channel chunk into registers (XR x 16 bytes per thread)
    constexpr bool PREF = NW == 8;
    uint4 hv[XR];
    auto load_halo = [&](int cc) {
        const int c0 = cc * 32;
        const unsigned short *src;
        int Cs, cofs;
        if (c0 < g.C1) {
            src = a1; Cs = g.C1; cofs = c0;
        } else {
            src = a2; Cs = g.C2; cofs = c0 - g.C1;
        }
#pragma unroll
        for (int u = 0; u < XR; u++) {
            const int idx = u * TPB + tid;
            hv[u] = make_uint4(0, 0, 0, 0);
            if (idx < nx) {
                const int slot = idx >> 2;
                const int ez = (slot * tg.magHW) >> 20, rem = slot - ez * EHW;
                const int ey = (rem * tg.magW) >> 20, ex = rem - ey * tg.EW;
                const int id = iz0 + ez, ih = iy0 + ey, iw = ix0 + ex;
                if (!(MVD_F16_DBG & 16) && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                    hv[u] = *reinterpret_cast<const uint4 *>(
                        src + ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * Cs + cofs + (tid & 3) * 8);
            }
        }
    };
    const int rot = item % ngroups;
    const bool fullg = ngroups * TG == g.ntaps;  // every group has TG taps (27 taps, TG = 3): no per-slot tap tests
    for (int cc = cc_begin; cc < cc_end; cc++) {
        load_w(cc, rot, wrA);
        if (ngroups > 1) load_w(cc, (rot + 1) % ngroups, wrB);
        if (!PREF || cc == cc_begin) load_halo(cc);
        __syncthreads();  // B1: every wave is done with the previous chunk's LDS
#pragma unroll
        for (int u = 0; u < XR; u++) {
            const int idx = u * TPB + tid;
            int part = idx & 3;
            if (SWZ) {
                const int slot = idx >> 2;
                const int ez = (slot * tg.magHW) >> 20, rem = slot - ez * EHW;
                part ^= ((rem * tg.magW) >> 20) & 3;  // halo y-row of the slot
            }
            *reinterpret_cast<uint4 *>(Xs + (size_t)(idx >> 2) * XS + part * 16) = hv[u];
        }
        // PREF (one workgroup per CU: nothing else hides the fetch): the next chunk's halo travels while this chunk's 27 taps run
        if (PREF && cc + 1 < cc_end) load_halo(cc + 1);
        auto group_mfmas = [&](int gi, int wbuf) __attribute__((always_inline)) {
            const unsigned char *wb_ = Wsm + (size_t)wbuf * WBUF;
            // operand registers are double buffered over the 2*TG (tap, k-step) slots of the group: the fragments of
            // slot j+1 are read from LDS before slot j's MFMAs are issued, so a register an in-flight MFMA still reads
            // is never the target of the next ds_read (tools/probes/mfma_probe.hip: 111 -> 137 TFLOP/s effect)
            // RA = read-ahead in (tap, k-step) slots.  (Round 3: RA = 2 on the eight-wave form -- a slot is only four MFMAs per
            // wave -- measured SLOWER: 128 -> 128 @32^3 0.070 -> 0.071, 256 -> 128 0.118 -> 0.125 ms.)
            constexpr int RA = 1, NBUF = RA + 1;
            bf16x8 af[NBUF][MT], bfr[NBUF][NT];
            int to3[TG];
#pragma unroll
            for (int tl = 0; tl < TG; tl++) to3[tl] = __builtin_amdgcn_readlane(toff_l, gi * TG + tl);
            auto read_ops = [&](int j, int buf) {
                const int tl = j >> 1, s = j & 1;
                const int to = to3[tl];
                const int t = gi * TG + tl < g.ntaps ? gi * TG + tl : g.ntaps - 1;  // (SWZ only)
                const int xo = SWZ ? ((((s << 1) | h) ^ ((ylq + tg.tdy[t]) & 3)) << 4) : s * 32;
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    uint4 q = *reinterpret_cast<const uint4 *>(Xs + sbase[m] + to + xo);
                    af[buf][m] = *reinterpret_cast<bf16x8 *>(&q);
                }
#pragma unroll
                for (int q_ = 0; q_ < NT; q_++) {
                    uint4 q = *reinterpret_cast<const uint4 *>(wb_ + ((size_t)((tl * 2 + s) * 2 + h) * KT + q_ * 32 + i) * WROW);
                    bfr[buf][q_] = *reinterpret_cast<bf16x8 *>(&q);
                }
            };
#if (MVD_F16_DBG & 4)
            if (gi == 0)
#endif
            {
#pragma unroll
                for (int j = 0; j < RA && j < 2 * TG; j++) read_ops(j, j % NBUF);
            }
#pragma unroll
            for (int j = 0; j < 2 * TG; j++) {
#if (MVD_F16_DBG & 4)
                if (gi == 0)
#endif
                if (j + RA < 2 * TG) read_ops(j + RA, (j + RA) % NBUF);
#if !(MVD_F16_DBG & 2)
                if (fullg || gi * TG + (j >> 1) < g.ntaps) {  // block-uniform
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int q_ = 0; q_ < NT; q_++)
                            acc[m][q_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j % NBUF][q_], af[j % NBUF][m], acc[m][q_], 0, 0, 0);
                }
#endif
            }
        };
        // group gi: its weights (loaded two groups ago into register set gi & 1) go to LDS buffer gi & 1 -- last read by
        // group gi-2, which every wave left before the barrier of group gi-1 -- and the freed registers take group gi+2.
        // With one register set the loads of group gi+1 had only group gi's 24 MFMAs (~0.4 us) to come back from L2.
        int gr = rot, gr2 = rot + 2 >= ngroups ? rot + 2 - ngroups : rot + 2;  // rotated indices of groups gj and gj + 2
        if (gr2 >= ngroups) gr2 -= ngroups;                                     // (ngroups == 1)
#if (MVD_F16_DBG & 64)
#define MVD_STAMP(K) { if (stamp_on) { stamps[nst * 8 + (K)] = __builtin_amdgcn_s_memtime(); } }
#else
#define MVD_STAMP(K)
#endif
        // one group: PAR = its LDS buffer / register set (two named arrays: an array of arrays went to scratch memory)
#define MVD_F16_GROUP(PAR, WRX)                                                        \
        {                                                                              \
            MVD_F16_STAMP_ON                                                           \
            MVD_STAMP(0)                                                               \
            if (!(MVD_F16_DBG & 8)) store_w(PAR, WRX);                                 \
            MVD_STAMP(1)                                                               \
            if (!(MVD_F16_DBG & 1)) __syncthreads(); /* B2 */                          \
            MVD_STAMP(2)                                                               \
            if (!(MVD_F16_DBG & 8) && gj + 2 < ngroups) load_w(cc, gr2, WRX);          \
            MVD_STAMP(3)                                                               \
            group_mfmas(gr, PAR);                                                      \
            MVD_STAMP(4)                                                               \
            MVD_F16_STAMP_NEXT                                                         \
            gr = gr + 1 == ngroups ? 0 : gr + 1;                                       \
            gr2 = gr2 + 1 == ngroups ? 0 : gr2 + 1;                                    \
        }
#if (MVD_F16_DBG & 64)
#define MVD_F16_STAMP_ON const bool stamp_on = item == 64 && wave == 0 && lane == 0 && nst < 40;
#define MVD_F16_STAMP_NEXT if (stamp_on) nst++;
#else
#define MVD_F16_STAMP_ON
#define MVD_F16_STAMP_NEXT
#endif
        for (int gj = 0; gj < ngroups;) {
            MVD_F16_GROUP(0, wrA)
            if (++gj >= ngroups) break;
            MVD_F16_GROUP(1, wrB)
            ++gj;
        }
#undef MVD_F16_GROUP
#undef MVD_F16_STAMP_ON
#undef MVD_F16_STAMP_NEXT
#undef MVD_STAMP
    }
    // epilogue.  The MFMA operands are swapped (D^T = W^T X^T): column (lane & 31) = voxel of the M tile, rows = output
    // channels (r & 3) + 8 * (r >> 2) + 4 * h -- a lane holds 4 x 4 consecutive channels of ONE voxel and stores them as
    // 8-byte bf16 packets (sixteen 2-byte stores per tile cost more than the tile's MFMAs)
    const int od = od0 + wave;
    if (od >= g.Do) return;  // wave-uniform
    // the tile's bias values first, settled (common.h): loads between the stores would serialise them
    float4 bq[NT][4];
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
            bq[q][rg] = (bias && tg.S == 1) ? *reinterpret_cast<const float4 *>(bias + kb * KT + q * 32 + 8 * rg + 4 * h)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bq[q][rg] = settled(bq[q][rg]);
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int oh = oh0 + 4 * m + (i >> 3), ow = ow0 + (i & 7);
        const bool inside = oh < g.Ho && ow < g.Wo;  // (every lane takes part in the lane-half exchange below)
        const size_t o_lin = ((size_t)od * g.Ho + oh) * g.Wo + ow;
        const size_t ov = (((size_t)n * g.Dy + (od * g.so[0] + g.oo[0])) * g.Hy + (oh * g.so[1] + g.oo[1])) * g.Wy +
                          (ow * g.so[2] + g.oo[2]);
#pragma unroll
        for (int q = 0; q < NT; q++) {
            if (tg.S > 1) {
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int k = kb * KT + q * 32 + 8 * rg + 4 * h;
                    if (inside)
                        *reinterpret_cast<float4 *>(part + (((size_t)split * g.N + n) * ((size_t)g.Do * g.Ho * g.Wo) + o_lin) * tg.K + k) =
                            make_float4(acc[m][q][rg * 4 + 0], acc[m][q][rg * 4 + 1], acc[m][q][rg * 4 + 2], acc[m][q][rg * 4 + 3]);
                }
            } else {
#pragma unroll
                for (int kp = 0; kp < 2; kp++) {  // channel groups 2kp, 2kp+1: one 16-byte store per lane (T21)
                    uint2 pk[2];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int rg = 2 * kp + e;
                        const float4 b4 = bq[q][rg];
                        pk[e] = pack_bf16x4(acc[m][q][rg * 4 + 0] + b4.x, acc[m][q][rg * 4 + 1] + b4.y,
                                            acc[m][q][rg * 4 + 2] + b4.z, acc[m][q][rg * 4 + 3] + b4.w);
                    }
                    const uint4 img = pair_store_image(pk[0], pk[1]);
                    const int k = kb * KT + q * 32 + 16 * kp + 8 * h;  // first of this lane's 8 consecutive channels
                    if (inside && !(MVD_F16_DBG & 32)) {
                        if (k < g.K1) {
                            uint4 *dst = reinterpret_cast<uint4 *>(y1 + ov * g.K1 + k);
                            *dst = g.acc ? add_bf16x8(img, *dst) : img;   // (block-uniform)
                        } else
                            *reinterpret_cast<uint4 *>(y2 + ov * g.K2 + (k - g.K1)) = img;
                    }
                }
            }
        }
    }
}

// y[mapped(n,o)][k] = bf16( bias[k] + sum_s part[s][n][o][k] )
__global__ void k_split_reduce16(const FwdGeom g, const float *__restrict__ part, const float *__restrict__ bias,
                                 unsigned short *__restrict__ y1, unsigned short *__restrict__ y2, int S) {
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
            y1[ov * g.K1 + k] = f2bf(g.acc ? s + __uint_as_float((unsigned)y1[ov * g.K1 + k] << 16) : s);
        else
            y2[ov * g.K2 + (k - g.K1)] = f2bf(s);
    }
}

// ------------------------------------------------------------------------------------------------ stride-2 input gradient
// bf16 twin of k_dgrad32s (conv_mfma.hip): dx of a 3x3x3 stride-2 pad-1 conv, all eight output-parity classes from ONE
// staged dy tile.  The per-class launches of the gather engine re-read dy eight times and run 1- and 2-tap classes as whole
// launches (0.37 ms for 64 -> 32 at 128^3, whose traffic and MFMA work are ~0.07 ms each).  Workgroup = 8 waves on a
// 3 x 5 x 9 dy tile (2 x 4 x 8 positions j + the odd classes' +1 neighbours) per 32-channel reduce chunk; wave w takes
// class w on the positions of plane oz0 and class 7 - w on plane oz0 + 1 (6 or 9 tap groups per wave).  Swapped product
// D^T = W^T dy^T on v_mfma_f32_32x32x16_bf16: A = packed weights straight from L2 (one tap ahead), B = dy fragment from the
// LDS tile ([slot][64 B], 16-byte parts XORed by (slot >> 2) & 3: 2-way conflicts at worst); a lane ends with 16
// channels of ONE dx voxel and stores two 16-byte images (permlane32 exchange).  FwdGeom::acc adds into dx.
struct Dg16Tile {
    int ntd, nth, ntw, ncb, nitems, C;
    int acc;
};
__global__ __launch_bounds__(512, 2) void k_dgrad16s(const FwdGeom g, const Dg16Tile tg, const unsigned short *__restrict__ dy,
                                                     const unsigned short *__restrict__ wb, unsigned short *__restrict__ dx) {
    __shared__ __attribute__((aligned(16))) unsigned char Xs[135 * 64];
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
    // g.Di/Hi/Wi = dy grid (conv output), g.Dy/Hy/Wy = dx grid (conv input), g.C1 = K (reduce), tg.C = channels of dx
    const int K = g.C1, C = tg.C;
    const int nch = K >> 5;
    const int oz0 = td_ * 2, oy0 = th_ * 4, ox0 = tw_ * 8;
    // packed dgrad weights [chunk][tap][s][h][c][8]: lane (c = i, h) of k-step s
    const unsigned short *wlane = wb + ((size_t)h * C + cb * 32 + i) * 8;
    const size_t wstep = (size_t)2 * C * 8, wtap = 2 * wstep;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
    for (int kk = 0; kk < nch; kk++) {
        __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned short *>(dy + (size_t)n * g.Di * g.Hi * g.Wi * K + kk * 32), 0, 0x7fffffff, 0x00020000);
            uint4 v[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int idx = tid + u * 512;
                const int slot = idx >> 2, part = idx & 3;
                const int ez = slot / 45, rem = slot - ez * 45;
                const int ey = rem / 9, ex = rem - ey * 9;
                const int oz = oz0 + ez, oy = oy0 + ey, ox = ox0 + ex;
                const bool ok = idx < 135 * 4 && oz < g.Di && oy < g.Hi && ox < g.Wi;
                const unsigned o = ok ? (unsigned)(((oz * g.Hi + oy) * g.Wi + ox) * K + part * 8) * 2u : 0xffffffffu;
                v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (int)o, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int idx = tid + u * 512;
                const int slot = idx >> 2, part = idx & 3;
                if (idx < 135 * 4) *reinterpret_cast<uint4 *>(Xs + slot * 64 + ((part ^ ((slot >> 2) & 3)) << 4)) = v[u];
            }
        }
        __syncthreads();
        const unsigned short *wc = wlane + (size_t)kk * 27 * wtap;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int q = m ? 7 - wave : wave;
            const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
            const int slot0 = (m * 5 + (i >> 3)) * 9 + (i & 7);
            // the class's taps: per axis (t = 1, d = 0) for an even coordinate, (t = 0, d = 1) and (t = 2, d = 0) for an odd one
            const int ntq = 1 << (pz + py + px);
            auto tap_of = [&](int j, int &t, int &so) {
                const int bz = pz ? (j >> (py + px)) & 1 : 0, by = py ? (j >> px) & 1 : 0, bx = px ? j & 1 : 0;
                const int tz = pz ? (bz ? 2 : 0) : 1, ty = py ? (by ? 2 : 0) : 1, tx = px ? (bx ? 2 : 0) : 1;
                const int dz = (pz && !bz) ? 1 : 0, dyy = (py && !by) ? 1 : 0, dxx = (px && !bx) ? 1 : 0;
                t = (tz * 3 + ty) * 3 + tx;
                so = (dz * 5 + dyy) * 9 + dxx;
            };
            uint4 wv[2][2];
            int t0, so0;
            tap_of(0, t0, so0);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) wv[0][s2] = *reinterpret_cast<const uint4 *>(wc + (size_t)t0 * wtap + s2 * wstep);
            for (int j = 0; j < ntq; j += 2) {  // two taps per trip: static buffer indices (ntq is 1, 2, 4 or 8)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (j + u < ntq) {  // wave-uniform
                        int t, so, tn, son;
                        tap_of(j + u, t, so);
                        tap_of(j + u + 1 < ntq ? j + u + 1 : j + u, tn, son);
#pragma unroll
                        for (int s2 = 0; s2 < 2; s2++)
                            wv[(u + 1) & 1][s2] = *reinterpret_cast<const uint4 *>(wc + (size_t)tn * wtap + s2 * wstep);
                        const int slot = slot0 + so;
#pragma unroll
                        for (int s2 = 0; s2 < 2; s2++) {
                            const uint4 xq = *reinterpret_cast<const uint4 *>(Xs + slot * 64 + ((((s2 << 1) | h) ^ ((slot >> 2) & 3)) << 4));
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&wv[u][s2]),
                                                                            *reinterpret_cast<const bf16x8 *>(&xq), acc[m], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // lane (i, h) holds, per M tile m, channels (r & 3) + 8 (r >> 2) + 4 h of dx voxel 2 j + p, j = (oz0 + m, oy0 + (i >> 3),
    // ox0 + (i & 7)): two 16-byte images after the lane-half exchange
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int q = m ? 7 - wave : wave;
        const int pz = q >> 2, py = (q >> 1) & 1, px = q & 1;
        const int iz = 2 * (oz0 + m) + pz, iy = 2 * (oy0 + (i >> 3)) + py, ix = 2 * (ox0 + (i & 7)) + px;
        const bool inside = iz < g.Dy && iy < g.Hy && ix < g.Wy;
        const size_t ov = (((size_t)n * g.Dy + iz) * g.Hy + iy) * g.Wy + ix;
#pragma unroll
        for (int kp = 0; kp < 2; kp++) {
            uint2 pk[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int rg = 2 * kp + e;
                pk[e] = pack_bf16x4(acc[m][rg * 4 + 0], acc[m][rg * 4 + 1], acc[m][rg * 4 + 2], acc[m][rg * 4 + 3]);
            }
            const uint4 img = pair_store_image(pk[0], pk[1]);
            if (inside) {
                uint4 *dst = reinterpret_cast<uint4 *>(dx + ov * C + cb * 32 + 16 * kp + 8 * h);
                *dst = tg.acc ? add_bf16x8(img, *dst) : img;
            }
        }
    }
}

// (Round 3, tried: a persistent form for K <= 64 -- one workgroup per CU walking tiles, the wave's class weights resident in
// 128 registers, the next tile's dy prefetched into registers: 64 -> 32 at 128^3 0.189 -> 0.179 ms, nothing inside the step.
// The launch is bound by one memory round trip per 49 KB tile with two tiles in flight per CU either way, not by the
// weights' L2 latency; more tiles in flight do not fit beside resident weights.  Not kept.)
int dgrad16s(int N, int D, int H, int W, int C, int K, int Do, int Ho, int Wo, const unsigned short *dy, const unsigned short *wb,
             unsigned short *dx, hipStream_t s, int accumulate) {
    static const int on = getenv("MVD_DGRAD16S") ? atoi(getenv("MVD_DGRAD16S")) : 1;
    if (!on || C % 32 || K % 32 || (((uintptr_t)dy | (uintptr_t)wb | (uintptr_t)dx) & 15)) return -1;
    if ((long)Do * Ho * Wo * K * 2 >= (1L << 31)) return -1;  // 32-bit byte offsets inside one sample of dy (buffer loads)
    FwdGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = Do; g.Hi = Ho; g.Wi = Wo;
    g.Dy = D; g.Hy = H; g.Wy = W;
    g.C1 = K;
    Dg16Tile tg;
    tg.ntd = (Do + 1) / 2; tg.nth = (Ho + 3) / 4; tg.ntw = (Wo + 7) / 8;
    tg.ncb = C / 32;
    tg.C = C;
    tg.acc = accumulate;
    const long nitems = (long)N * tg.ntd * tg.nth * tg.ntw * tg.ncb;
    if (nitems > (1L << 30)) return -1;
    tg.nitems = (int)nitems;
    const unsigned grid = (unsigned)(((nitems + 7) / 8) * 8);
    hipLaunchKernelGGL(k_dgrad16s, dim3(grid), dim3(512), 0, s, g, tg, dy, wb, dx);
    return check_launch("conv dgrad bf16 (stride 2, fused parity classes)");
}

static const size_t LDS_LIMIT16 = 160 * 1024;

template <int NT, int MT, int TG, int XR, bool SWZ, int NW = 4>
static int launch_fwd16(const FwdGeom &g, Fwd16Tile &tg, const unsigned short *a1, const unsigned short *a2,
                        const unsigned short *w, const float *bias, unsigned short *y1, unsigned short *y2, void *ws,
                        size_t ws_bytes, hipStream_t s) {
    auto kern = k_fwd16<NT, MT, TG, XR, SWZ, NW>;
    const size_t lds = (size_t)XR * (NW * 16) * (SWZ ? 64 : 80) + 2 * (size_t)TG * 4 * (32 * NT) * 16 + 256;  // + tap tables
    if (lds > LDS_LIMIT16) return -1;
    static PerDeviceFlag configured;
    if (!configured()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_LIMIT16) != hipSuccess) {
            set_error("conv fwd16: cannot raise the dynamic LDS limit");
            return 1;
        }
        configured() = true;
    }
    const int K = g.K1 + g.K2;
    tg.nkb = K / (32 * NT);
    const long tiles = (long)tg.ntd * tg.nth * tg.ntw;
    const size_t out_elems = (size_t)g.N * g.Do * g.Ho * g.Wo * K;
    const long wgs = tiles * tg.nkb * g.N;
    const int nch = (g.C1 + g.C2) / 32;
    int S = 1;
    if (ws && wgs < 256 && nch >= 4) {
        long want = (512 + wgs - 1) / wgs;
        if (want > 16) want = 16;
        if (want > nch / 2) want = nch / 2;
        while (want > 1 && (size_t)want * out_elems * sizeof(float) > ws_bytes) want--;
        if (want > 1) S = (int)want;
    }
    tg.S = S;
    const long nitems = tiles * tg.nkb * tg.S * g.N;
    if (nitems > (1L << 30)) return -1;
    tg.nitems = (int)nitems;
    const long grid = ((nitems + 7) / 8) * 8;
    float *part = reinterpret_cast<float *>(ws);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, s, g, tg, a1, a2, w, bias, y1, y2, part);
    if (check_launch("conv fwd16 (bf16 mfma)")) return 1;
    if (S > 1) {
        long blocks = cdiv((long)out_elems, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_split_reduce16, dim3(blocks), dim3(256), 0, s, g, part, bias, y1, y2, S);
        return check_launch("conv fwd16 split reduce");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ persistent kernel
// 3x3x3 stride-1 layers with 32 reduce and 32 produce channels at high resolution (enc0.conv1, dec5.conv1 and their
// input gradients: 537 MB of activations, 232 GFLOP each).  A 256-voxel tile is only 108 bf16 MFMAs per wave (3.5k
// cycles) -- less than one L2 round trip -- so the one-tile-per-workgroup kernel above spends its time waiting for the
// halo.  Here one workgroup per CU walks its XCD's tiles; all 27 taps' weights (55 KB) stay in LDS for the whole walk,
// the halo of tile t+1 is fetched into registers while tile t's MFMAs run and lands in the other of two LDS halo buffers
// (one barrier per tile); B fragments are kept in registers for a filter plane (9 taps x 2 k-steps) so the LDS feeds one
// A fragment per MFMA.
constexpr int P_TPB = 512;    // 8 waves: d-plane = wave & 3, row block (4 rows) = wave >> 2; two waves per SIMD
constexpr int P_XR = 5;       // uint4 per thread: 600 slots x 4 / 512
constexpr int P_XS = 80;      // bytes per halo slot (64 + 16 pad)
constexpr int P_HALO = 600 * P_XS;
constexpr int P_WB = 27 * 2 * 2 * 32 * 16;  // bytes of the resident weights

__global__ __launch_bounds__(512, 2) void k_fwd16p(const FwdGeom g, const Fwd16Tile tg, const unsigned short *__restrict__ a1,
                                                   const unsigned short *__restrict__ w, const float *__restrict__ bias,
                                                   unsigned short *__restrict__ y1, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    unsigned char *Wsm = lds8;                 // [tap][s][h][k][16 B]
    unsigned char *Xs0 = lds8 + P_WB;          // two halo buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3;
    const int EHW = tg.EH * tg.EW;
    const int nx = tg.nslots * 4;

    // resident weights: 27 taps x 128 fragments of 16 B, in tap order (the tap -> weight index map is g.wt)
    for (int f = tid; f < 27 * 128; f += P_TPB) {
        const int t = f >> 7, r = f & 127;  // r = (s*2 + hh)*32 + k
        *reinterpret_cast<uint4 *>(Wsm + (size_t)f * 16) =
            *reinterpret_cast<const uint4 *>(w + ((size_t)g.wt[t] * 128 + r) * 8);
    }
    // tile-independent decode of this thread's staging slots
    int rel[P_XR], cz[P_XR];
#pragma unroll
    for (int u = 0; u < P_XR; u++) {
        const int idx = u * P_TPB + tid;
        const int slot = idx >> 2;
        const int ez = slot / EHW, rem = slot - ez * EHW;
        const int ey = rem / tg.EW, ex = rem - ey * tg.EW;
        rel[u] = ((ez * g.Hi + ey) * g.Wi + ex) * 32 + (idx & 3) * 8;
        cz[u] = idx < nx ? ((ez << 16) | (ey << 8) | ex) : -1;
    }
    uint4 v[P_XR];
    int n_, od0, oh0, ow0;
    auto decode = [&](int it) {
        unsigned r_ = (unsigned)(xcd * per_xcd + it);
        ow0 = (int)(r_ % (unsigned)tg.ntw) * 8; r_ /= (unsigned)tg.ntw;
        oh0 = (int)(r_ % (unsigned)tg.nth) * 8; r_ /= (unsigned)tg.nth;
        od0 = (int)(r_ % (unsigned)tg.ntd) * 4;
        n_ = (int)(r_ / (unsigned)tg.ntd);
    };
    auto valid = [&](int it) { return it < per_xcd && xcd * per_xcd + it < tg.nitems; };
    auto load_halo = [&]() {  // of the tile in (n_, od0, oh0, ow0)
        const int z0 = od0 - 1, y0 = oh0 - 1, x0 = ow0 - 1;
        const unsigned short *base = a1 + ((((long)n_ * g.Di + z0) * g.Hi + y0) * g.Wi + x0) * 32L;
#pragma unroll
        for (int u = 0; u < P_XR; u++) {
            const int id = z0 + (cz[u] >> 16), ih = y0 + ((cz[u] >> 8) & 255), iw = x0 + (cz[u] & 255);
            v[u] = make_uint4(0, 0, 0, 0);
            if (cz[u] >= 0 && !(dbg & 1) && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                v[u] = *reinterpret_cast<const uint4 *>(base + rel[u]);
        }
    };
    auto store_halo = [&](unsigned char *Xs) {
#pragma unroll
        for (int u = 0; u < P_XR; u++) {
            const int idx = u * P_TPB + tid;
            if (idx < nx) *reinterpret_cast<uint4 *>(Xs + (size_t)(idx >> 2) * P_XS + (idx & 3) * 16) = v[u];
        }
    };
    const int wd = wave & 3, wm = wave >> 2;
    const int sbase = ((wd * tg.EH + 4 * wm + (i >> 3)) * tg.EW + (i & 7)) * P_XS + h * 16;
    const unsigned char *wl = Wsm + ((size_t)h * 32 + i) * 16;  // + ((t*2 + s)*2)*32*16 = (t*2 + s) * 1024

    int it = local;
    if (!valid(it)) return;  // whole workgroup
    decode(it);
    load_halo();
    store_halo(Xs0);
    __syncthreads();  // weights + first halo visible
    int cur = 0;
    while (true) {
        const int itn = it + nlocal;
        const bool more = valid(itn);  // block-uniform
        const int cn = n_, cod0 = od0, coh0 = oh0, cow0 = ow0;
        if (more) {
            decode(itn);
            load_halo();  // in flight during this tile's MFMAs
        }
        const unsigned char *Xs = Xs0 + (size_t)cur * P_HALO;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll 1
        for (int gz = 0; gz < ((dbg & 2) ? 0 : 3); gz++) {
            bf16x8 bw[9][2];  // this filter plane's B fragments
#pragma unroll
            for (int t9 = 0; t9 < 9; t9++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    uint4 q = *reinterpret_cast<const uint4 *>(wl + (size_t)((gz * 9 + t9) * 2 + s2) * 1024);
                    bw[t9][s2] = *reinterpret_cast<bf16x8 *>(&q);
                }
            bf16x8 af[2][2];  // [buffer][k-step]
            auto read_a = [&](int t9, int buf) {
                const int to = tg.toff[gz * 9 + t9] * P_XS;
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    uint4 q = *reinterpret_cast<const uint4 *>(Xs + sbase + to + s2 * 32);
                    af[buf][s2] = *reinterpret_cast<bf16x8 *>(&q);
                }
            };
            read_a(0, 0);
#pragma unroll
            for (int t9 = 0; t9 < 9; t9++) {
                if (t9 + 1 < 9) read_a(t9 + 1, (t9 + 1) & 1);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
                        // operands swapped (D^T = W^T X^T): a lane ends up with 4 x 4 CONSECUTIVE output channels of one
                        // voxel, i.e. 8-byte bf16 stores instead of sixteen 2-byte ones (the 2-byte stores cost more than
                        // the tile's MFMAs)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[t9][s2], af[t9 & 1][s2], acc, 0, 0, 0);
            }
        }
        // epilogue of the current tile: D^T layout -- column (lane & 31) = voxel of the M tile, rows = output channels
        // (r & 3) + 8 * (r >> 2) + 4 * h
        const int od = cod0 + wd;
        if (od < g.Do && !(dbg & 4)) {
            {
                const int oh = coh0 + 4 * wm + (i >> 3), ow = cow0 + (i & 7);
                if (oh < g.Ho && ow < g.Wo) {
                    unsigned short *yo = y1 + ((((size_t)cn * g.Dy + od) * g.Hy + oh) * g.Wy + ow) * 32 + 4 * h;
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int k = 8 * rg + 4 * h;
                        float bv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (bias) {
                            const float4 b4 = *reinterpret_cast<const float4 *>(bias + k);
                            bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
                        }
                        uint2 q;
                        q.x = (unsigned)f2bf(acc[rg * 4 + 0] + bv[0]) | ((unsigned)f2bf(acc[rg * 4 + 1] + bv[1]) << 16);
                        q.y = (unsigned)f2bf(acc[rg * 4 + 2] + bv[2]) | ((unsigned)f2bf(acc[rg * 4 + 3] + bv[3]) << 16);
                        *reinterpret_cast<uint2 *>(yo + 8 * rg) = q;
                    }
                }
            }
        }
        if (!more) break;
        store_halo(Xs0 + (size_t)(cur ^ 1) * P_HALO);  // the other buffer: nobody reads it during this tile
        __syncthreads();
        cur ^= 1;
        it = itn;
    }
}


// ------------------------------------------------------------------------------------------------ k_fwd16q (round 2)
// Same shape class as k_fwd16p (3x3x3 stride 1, 32 -> 32 channels, >= 4 tiles per CU) with the two things its ablation
// blamed removed: (1) the 3-way LDS bank conflicts of the A-operand reads -- halo slots are now the bare
// 64-byte voxel rows and the four 16-byte parts of a row are XOR-swizzled with (halo y-row & 3): a ds_read_b128 lane
// group covers 4 x-runs of 4 consecutive voxels in 4 consecutive y-rows, so x & 3 picks the bank quad inside a row group
// and the swizzle separates the rows -- conflict-free for every tap (tools: brute force over layouts, round 2);
// (2) the B-operand traffic: LDS carried two reads per MFMA (A and B), i.e. it was the bound even conflict-free.  Now a
// workgroup is 4 waves (one per SIMD, 512 registers each) and every lane keeps ALL 27 taps' B fragments (its 54 16-byte
// weight fragments, 216 VGPRs) for the whole walk; a wave owns one z-plane of the 4x8x8 tile = two 32-voxel M tiles that
// share each B fragment: one LDS read per MFMA, issued a tap ahead; tap offsets are immediates.
typedef int i32x4 __attribute__((ext_vector_type(4)));
// B fragments live in the ACCUMULATOR half of the register file for the whole walk (216 + 32 accumulator registers = 248
// of 256 AGPRs) and the MFMA reads them there: with the builtin hipcc parks them in AGPRs anyway but copies each one to a
// VGPR before use (179 v_accvgpr_read + 141 v_mov per tile, PMC: 672 VALU per 108 MFMAs).  `s_nop 1`: a VALU-written
// A fragment (a compiler copy) needs two wait states before the MFMA reads it and nothing inside an asm string is
// padded; accumulate chains on the same registers need none (guide 5.7 item 2).
#define MVD_MFMA16_AB(ACC, BFRAG, AFRAG) \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "a"(BFRAG), "v"(AFRAG))
#define MVD_MFMA16_AB_NONOP(ACC, BFRAG, AFRAG) \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "a"(BFRAG), "v"(AFRAG))
constexpr int Q_TPB = 256;
constexpr int Q_XR = 10;                 // uint4 per thread: 600 slots x 4 parts / 256
constexpr int Q_HALO = 600 * 64;

struct Fwd16QTile {
    int ntd, nth, ntw, nitems;
    int wsel[27];  // weight tap index of the raster tap (dz, dy, dx) = offsets -1..1
};

// DBG (compile time; 0 in production): ablation switches for tools/bench_conv.py -- 1 no halo traffic after the first
// tile, 2 no MFMAs, 4 no output stores (results are then wrong by construction)
//
// One wave per SIMD has nobody to hide behind, so everything else is threaded through the 108-MFMA stream of a tile by
// hand and kept cheap in VALU terms (PMC, round 2: the first version spent 613 VALU instructions per tile and wave on
// address arithmetic, clamps and selects -- more than the ~540 issue slots the MFMAs leave free):
//   * halo loads of the NEXT tile at the head of the tile; interior tiles (72 % at 128^3) use scalar-base + precomputed
//     32-bit lane offsets, no bounds arithmetic at all; border tiles clamp the coordinates and zero on the way to LDS;
//   * its ten LDS writes one per tap behind taps 14..23 into the other halo buffer, a barrier behind tap 24 (counted
//     lgkmcnt: the A reads in flight are not drained), and the first two fragment groups of the next tile are already
//     fetched under taps 25/26 -- no LDS fill bubble at the tile boundary;
//   * the accumulators start from the bias and move to VGPRs when the tile is done; the conversion to bf16 and the
//     eight 8-byte stores go between the first taps of the next tile.
template <int DBG>
__global__ __launch_bounds__(256, 1) void k_fwd16q(const FwdGeom g, const Fwd16QTile tg, const unsigned short *__restrict__ a1,
                                                   const unsigned short *__restrict__ w, const float *__restrict__ bias,
                                                   unsigned short *__restrict__ y1) {
    constexpr int dbg = DBG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3;

    // resident B fragments: raster tap p, k-step s -> 16 bytes of lane (i, h)
    i32x4 bw[27][2];
#pragma unroll
    for (int p = 0; p < 27; p++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const uint4 q = *reinterpret_cast<const uint4 *>(w + ((size_t)tg.wsel[p] * 128 + (s2 * 2 + h) * 32 + i) * 8);
            bw[p][s2] = *reinterpret_cast<const i32x4 *>(&q);
        }
    // staging slots of this thread (tile independent): byte offset from the tile's first halo voxel, LDS byte offset
    unsigned rel[Q_XR];
    int wa[Q_XR];
#pragma unroll
    for (int u = 0; u < Q_XR; u++) {
        const int idx = u * Q_TPB + tid;
        const int slot = idx < 2400 ? (idx >> 2) : 599;  // slots >= 600 (last pass, tid >= 96) are loaded, never stored
        const int ez = slot / 100, rem = slot - ez * 100;
        const int ey = rem / 10, ex = rem - ey * 10;
        rel[u] = (unsigned)(((ez * g.Hi + ey) * g.Wi + ex) * 64 + (idx & 3) * 16);
        wa[u] = slot * 64 + (((idx & 3) ^ (ey & 3)) << 4);
    }
    // A-operand read addresses: M tile m (rows 4m..4m+3), halo row shift dy, k-step s (the tap offset is an immediate)
    int ra[2][3][2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int yl = 4 * m + (i >> 3);
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
                ra[m][dy][s2] = ((wave * 10 + yl) * 10 + (i & 7)) * 64 + ((((s2 << 1) | h) ^ ((yl + dy) & 3)) << 4);
    }
    // the bias in accumulator layout (register r <-> output channel (r & 3) + 8 * (r >> 2) + 4 * h): C operand of tap 0
    f32x16 biasv;
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = *reinterpret_cast<const float4 *>(bias + 8 * rg + 4 * h);
        biasv[rg * 4 + 0] = b4.x; biasv[rg * 4 + 1] = b4.y; biasv[rg * 4 + 2] = b4.z; biasv[rg * 4 + 3] = b4.w;
    }
    uint4 v[Q_XR];
    int n_, od0, oh0, ow0;
    unsigned inb = ~0u;  // bit u: slot u of this thread lies inside the volume (border tiles only)
    auto decode = [&](int it) {
        unsigned r_ = (unsigned)(xcd * per_xcd + it);
        ow0 = (int)(r_ % (unsigned)tg.ntw) * 8; r_ /= (unsigned)tg.ntw;
        oh0 = (int)(r_ % (unsigned)tg.nth) * 8; r_ /= (unsigned)tg.nth;
        od0 = (int)(r_ % (unsigned)tg.ntd) * 4;
        n_ = (int)(r_ / (unsigned)tg.ntd);
    };
    auto valid = [&](int it) { return it < per_xcd && xcd * per_xcd + it < tg.nitems; };
    auto load_halo = [&]() {
        const int z0 = od0 - 1, y0 = oh0 - 1, x0 = ow0 - 1;
        const bool interior = z0 >= 0 && z0 + 6 <= g.Di && y0 >= 0 && y0 + 10 <= g.Hi && x0 >= 0 && x0 + 10 <= g.Wi;
        if (interior) {  // block-uniform: scalar base + per-lane constant offset
            const char *tb = reinterpret_cast<const char *>(a1) + ((((long)n_ * g.Di + z0) * g.Hi + y0) * g.Wi + x0) * 64L;
            inb = ~0u;
#pragma unroll
            for (int u = 0; u < Q_XR; u++) v[u] = *reinterpret_cast<const uint4 *>(tb + rel[u]);
        } else {
            const unsigned short *base = a1 + (long)n_ * g.Di * g.Hi * g.Wi * 32L + (tid & 3) * 8;
            inb = 0;
#pragma unroll
            for (int u = 0; u < Q_XR; u++) {
                const int idx = u * Q_TPB + tid;
                const int slot = idx < 2400 ? (idx >> 2) : 599;
                const int ez = slot / 100, rem = slot - ez * 100;
                const int ey = rem / 10, ex = rem - ey * 10;
                const int id = z0 + ez, ih = y0 + ey, iw = x0 + ex;
                const bool in = id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
                inb |= in ? (1u << u) : 0u;
                const int cd = min(max(id, 0), g.Di - 1), ch = min(max(ih, 0), g.Hi - 1), cw = min(max(iw, 0), g.Wi - 1);
                v[u] = *reinterpret_cast<const uint4 *>(base + ((cd * g.Hi + ch) * g.Wi + cw) * 32);
            }
        }
    };
    auto store_halo_piece = [&](unsigned char *Xs, int u) {
        const bool in = (inb >> u) & 1u;
        uint4 q;
        q.x = in ? v[u].x : 0u; q.y = in ? v[u].y : 0u; q.z = in ? v[u].z : 0u; q.w = in ? v[u].w : 0u;
        if (u < Q_XR - 1 || tid < 96) *reinterpret_cast<uint4 *>(Xs + wa[u]) = q;  // 600 slots: the last pass is partial
    };

    int it = local;
    if (!valid(it)) return;  // whole workgroup
    decode(it);
    load_halo();
#pragma unroll
    for (int u = 0; u < Q_XR; u++) store_halo_piece(lds8, u);
    // the weight fragments are consumed by asm statements only: make their loads complete HERE, once -- otherwise the
    // wait for them lands at the loop head as `s_waitcnt vmcnt(0)` and drains the output stores every tile
#pragma unroll
    for (int p = 0; p < 27; p++) asm volatile("" : "+a"(bw[p][0]), "+a"(bw[p][1]));
    __syncthreads();
    int cur = 0;
    float pe[2][16];
    unsigned short *pyo[2] = {nullptr, nullptr};  // output address of this lane's voxel in M tile m (nullptr: outside)
    bool have_prev = false;
    auto epilogue_piece = [&](int m, int kp) {  // channel groups 2kp, 2kp+1 of M tile m: one 16-byte store per lane
        const int k = 2 * kp;
        const uint4 q = pair_store_image(
            pack_bf16x4(pe[m][k * 4 + 0], pe[m][k * 4 + 1], pe[m][k * 4 + 2], pe[m][k * 4 + 3]),
            pack_bf16x4(pe[m][k * 4 + 4], pe[m][k * 4 + 5], pe[m][k * 4 + 6], pe[m][k * 4 + 7]));
        if (pyo[m] && !(dbg & 4)) *reinterpret_cast<uint4 *>(pyo[m] + 8 * k) = q;
    };
    i32x4 af[3][2][2];  // ring of 3 tap groups x [m][k-step]: fragments are fetched TWO taps (8 MFMAs) ahead, across tiles
    auto read_a = [&](const unsigned char *Xs, int p, int buf) {
        const int dz = p / 9, dy = (p / 3) % 3, dx = p % 3;
        const int to = ((dz * 10 + dy) * 10 + dx) * 64;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const uint4 q = *reinterpret_cast<const uint4 *>(Xs + ra[m][dy][s2] + to);
                af[buf][m][s2] = *reinterpret_cast<const i32x4 *>(&q);
            }
    };
    read_a(lds8, 0, 0);
    read_a(lds8, 1, 1);
    while (true) {
        const int itn = it + nlocal;
        const bool more = valid(itn);  // block-uniform
        const int cn = n_, cod0 = od0, coh0 = oh0, cow0 = ow0;
        if (more) {
            decode(itn);
            if (!(dbg & 1)) load_halo();  // in flight during this tile's MFMAs
        }
        const unsigned char *Xs = lds8 + (size_t)cur * Q_HALO;
        unsigned char *Xn = lds8 + (size_t)(cur ^ 1) * Q_HALO;
        f32x16 acc[2] = {biasv, biasv};  // (C and D of an MFMA share one register file: the bias cannot be a VGPR C operand)
#pragma unroll
        for (int p = 0; p < 27; p++) {
            if (p + 2 < 27) read_a(Xs, p + 2, (p + 2) % 3);
            else if (more && !(dbg & 8)) read_a(Xn, p + 2 - 27, (p + 2) % 3);  // next tile's taps 0, 1 (behind the barrier of tap 24)
            if ((p == 1 || p == 3 || p == 5 || p == 7) && have_prev) epilogue_piece((p - 1) >> 2, ((p - 1) >> 1) & 1);
            if (!(dbg & 8) && p >= 14 && p <= 23 && more && !(dbg & 1)) store_halo_piece(Xn, p - 14);
            if ((dbg & 8) && p >= 17 && p <= 26 && more && !(dbg & 1)) store_halo_piece(Xn, p - 17);
            if (!(dbg & 8) && p == 24 && more) {
                // every wave's halo writes (the newest: tap 23, four LDS reads ago) done and visible; the A reads in
                // flight stay in flight
                asm volatile("s_waitcnt lgkmcnt(4)\n\ts_barrier" ::: "memory");
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int m = 0; m < 2; m++)
                    // operands swapped (D^T = W^T X^T): a lane ends with 4 x 4 consecutive output channels of one voxel
                    if (!(dbg & 2)) MVD_MFMA16_AB(acc[m], bw[p][s2], af[p % 3][m][s2]);
        }
        // an MFMA result needs 12 wait states before anything but an accumulating MFMA touches it
        asm volatile("s_nop 7\n\ts_nop 4" : "+a"(acc[0]), "+a"(acc[1]));
        {
            const int od = cod0 + wave;
#pragma unroll
            for (int m = 0; m < 2; m++) {
#pragma unroll
                for (int r = 0; r < 16; r++) pe[m][r] = acc[m][r];
                const int oh = coh0 + 4 * m + (i >> 3), ow = cow0 + (i & 7);
                pyo[m] = (od < g.Do && oh < g.Ho && ow < g.Wo)
                             ? y1 + ((((size_t)cn * g.Dy + od) * g.Hy + oh) * g.Wy + ow) * 32 + 8 * h
                             : nullptr;
            }
            have_prev = true;
        }
        if (!more) break;
        if (dbg & 8) {  // experiment: staging at the very end of the tile, plain barrier, LDS fill bubble
            __syncthreads();
            read_a(Xn, 0, 0);
            read_a(Xn, 1, 1);
        }
        cur ^= 1;
        it = itn;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) epilogue_piece(j >> 1, j & 1);  // the last tile
}

// ------------------------------------------------------------------------------------------------ k_fwd16r (round 2)
// EXPERIMENT, not the default path (MVD_FWD16R=1 selects it; measured 9 % slower than k_fwd16q, see launch_fwd16p):
// k_fwd16q with the halo moved by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, nothing for the
// wave to issue but ten 1-KB pieces per tile).  Why it was tried: the ablation of k_fwd16q (tools/bench_conv.py, MVD_FWD16Q_DBG)
// showed its phases ADDING UP -- MFMA + LDS reads 0.16 ms, halo path +0.065, stores +0.043 -- however the instructions
// were interleaved: a CU's vector-memory pipe moves ~10 B/clk, a burst of ten register loads per lane fills its queue,
// and the one wave per SIMD then sits in the VMEM issue stage instead of issuing MFMAs.  Here the pieces of tile t+2 are
// issued one every other tap of tile t (the pipe's own pace), land in the third of three halo buffers while tile t+1's
// buffer is complete, and are retired by a counted vmcnt behind tap 24 (the ten newest operations of a wave at that
// point are exactly its ten pieces of tile t+2; everything older -- tile t+1's pieces, tile t-1's stores -- is done).
// LDS image: lane-linear per piece (wave-uniform base + 16 * lane), so the XOR swizzle of the 16-byte parts is applied
// on the SOURCE address: the lane that fills part position pp of a slot fetches part pp ^ (halo row & 3) of that voxel.
// Zero padding: out-of-volume lanes fetch from a 16-byte zero page.
__device__ uint4 g_zero_page16 = {0u, 0u, 0u, 0u};
constexpr int R_BUF = 40 * 1024;  // 600 slots x 64 B rounded up to 40 pieces of 1 KB (4 waves x 10 pieces)

__global__ __launch_bounds__(256, 1) void k_fwd16r(const FwdGeom g, const Fwd16QTile tg, const unsigned short *__restrict__ a1,
                                                   const unsigned short *__restrict__ w, const float *__restrict__ bias,
                                                   unsigned short *__restrict__ y1) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds8[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    typedef const __attribute__((address_space(1))) unsigned char glb_byte;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3;

    i32x4 bw[27][2];
#pragma unroll
    for (int p = 0; p < 27; p++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const uint4 q = *reinterpret_cast<const uint4 *>(w + ((size_t)tg.wsel[p] * 128 + (s2 * 2 + h) * 32 + i) * 8);
            bw[p][s2] = *reinterpret_cast<const i32x4 *>(&q);
        }
    // piece u of this wave fills LDS bytes [(u * 4 + wave) * 1024 + 16 * lane, +16): slot = that / 64, part position pp
    unsigned rel[Q_XR];   // source byte offset from the tile's first halo voxel (interior tiles)
    int cz[Q_XR];         // halo coordinates of the slot (border tiles), -1: beyond the 600 slots
#pragma unroll
    for (int u = 0; u < Q_XR; u++) {
        const int unit = (u * 4 + wave) * 64 + lane;
        const int slot = unit >> 2, pp = unit & 3;
        const int sl = slot < 600 ? slot : 599;
        const int ez = sl / 100, rem = sl - ez * 100;
        const int ey = rem / 10, ex = rem - ey * 10;
        const int q = pp ^ (ey & 3);
        rel[u] = (unsigned)(((ez * g.Hi + ey) * g.Wi + ex) * 64 + q * 16);
        cz[u] = slot < 600 ? ((ez << 16) | (ey << 8) | ex | (q << 24)) : -1;
    }
    int ra[2][3][2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int yl = 4 * m + (i >> 3);
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
                ra[m][dy][s2] = ((wave * 10 + yl) * 10 + (i & 7)) * 64 + ((((s2 << 1) | h) ^ ((yl + dy) & 3)) << 4);
    }
    f32x16 biasv;
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = *reinterpret_cast<const float4 *>(bias + 8 * rg + 4 * h);
        biasv[rg * 4 + 0] = b4.x; biasv[rg * 4 + 1] = b4.y; biasv[rg * 4 + 2] = b4.z; biasv[rg * 4 + 3] = b4.w;
    }
    struct TileC { int n, od0, oh0, ow0; };
    auto decode = [&](int it) {
        TileC t;
        unsigned r_ = (unsigned)(xcd * per_xcd + it);
        t.ow0 = (int)(r_ % (unsigned)tg.ntw) * 8; r_ /= (unsigned)tg.ntw;
        t.oh0 = (int)(r_ % (unsigned)tg.nth) * 8; r_ /= (unsigned)tg.nth;
        t.od0 = (int)(r_ % (unsigned)tg.ntd) * 4;
        t.n = (int)(r_ / (unsigned)tg.ntd);
        return t;
    };
    auto valid = [&](int it) { return it < per_xcd && xcd * per_xcd + it < tg.nitems; };
    const glb_byte *zero_page = (glb_byte *)(const void *)&g_zero_page16;
    // one 1-KB piece of tile T's halo into halo buffer `buf`
    auto dma_piece = [&](const TileC &T, int buf, int u) {
        const int z0 = T.od0 - 1, y0 = T.oh0 - 1, x0 = T.ow0 - 1;
        const bool interior = z0 >= 0 && z0 + 6 <= g.Di && y0 >= 0 && y0 + 10 <= g.Hi && x0 >= 0 && x0 + 10 <= g.Wi;
        const glb_byte *src;
        if (interior) {  // block-uniform
            const glb_byte *tb = (glb_byte *)(const void *)a1 + ((((long)T.n * g.Di + z0) * g.Hi + y0) * g.Wi + x0) * 64L;
            src = tb + rel[u];
        } else {
            const int c = cz[u];
            const int id = z0 + ((c >> 16) & 255), ih = y0 + ((c >> 8) & 255), iw = x0 + (c & 255);
            const bool in = c >= 0 && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
            const glb_byte *vb = (glb_byte *)(const void *)a1 +
                                 ((((long)T.n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * 64L + ((c >> 24) & 3) * 16;
            src = in ? vb : zero_page;
        }
        // inline asm, not __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) drains every piece with `s_waitcnt vmcnt(0)`
        // in front of the next ds_read (it models the DMA as an LDS write), which is exactly the serialisation this
        // kernel exists to remove.  M0 = LDS byte address of the piece, saved / restored around the instruction.
        lds_byte *dst = (lds_byte *)(lds8 + (size_t)buf * R_BUF + (size_t)(u * 4 + wave) * 1024);
        const unsigned dsta = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dsta) : "memory");
    };

    int it = local;
    if (!valid(it)) return;  // whole workgroup
    TileC cur_t = decode(it);
    TileC nxt_t = cur_t, nx2_t = cur_t;
    bool more = valid(it + nlocal), more2 = valid(it + 2 * nlocal);
    if (more) nxt_t = decode(it + nlocal);
#pragma unroll
    for (int u = 0; u < Q_XR; u++) dma_piece(cur_t, 0, u);
    if (more) {
#pragma unroll
        for (int u = 0; u < Q_XR; u++) dma_piece(nxt_t, 1, u);
    }
#pragma unroll
    for (int p = 0; p < 27; p++) asm volatile("" : "+a"(bw[p][0]), "+a"(bw[p][1]));
    // tile 0's pieces are older than tile 1's ten: retire them (and everything before), then meet
    if (more) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int cur = 0;
    float pe[2][16];
    unsigned short *pyo[2] = {nullptr, nullptr};
    bool have_prev = false;
    auto epilogue_piece = [&](int m, int kp) {
        const int k = 2 * kp;
        const uint4 q = pair_store_image(
            pack_bf16x4(pe[m][k * 4 + 0], pe[m][k * 4 + 1], pe[m][k * 4 + 2], pe[m][k * 4 + 3]),
            pack_bf16x4(pe[m][k * 4 + 4], pe[m][k * 4 + 5], pe[m][k * 4 + 6], pe[m][k * 4 + 7]));
        if (pyo[m]) *reinterpret_cast<uint4 *>(pyo[m] + 8 * k) = q;
    };
    i32x4 af[3][2][2];
    auto read_a = [&](const unsigned char *Xs, int p, int buf) {
        const int dz = p / 9, dy = (p / 3) % 3, dx = p % 3;
        const int to = ((dz * 10 + dy) * 10 + dx) * 64;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const uint4 q = *reinterpret_cast<const uint4 *>(Xs + ra[m][dy][s2] + to);
                af[buf][m][s2] = *reinterpret_cast<const i32x4 *>(&q);
            }
    };
    read_a(lds8, 0, 0);
    read_a(lds8, 1, 1);
    while (true) {
        if (more2) nx2_t = decode(it + 2 * nlocal);
        const int b2 = cur >= 1 ? cur - 1 : 2;  // (cur + 2) % 3: read last during the PREVIOUS tile, free since its barrier
        const int b1 = cur == 2 ? 0 : cur + 1;  // (cur + 1) % 3
        const unsigned char *Xs = lds8 + (size_t)cur * R_BUF;
        const unsigned char *Xn = lds8 + (size_t)b1 * R_BUF;
        f32x16 acc[2] = {biasv, biasv};
#pragma unroll
        for (int p = 0; p < 27; p++) {
            if (p + 2 < 27) read_a(Xs, p + 2, (p + 2) % 3);
            else if (more) read_a(Xn, p + 2 - 27, (p + 2) % 3);  // next tile's taps 0, 1 (behind the barrier of tap 24)
            if (p < 4 && have_prev) epilogue_piece(p >> 1, p & 1);                       // stores first ...
            if (p >= 4 && p <= 22 && !(p & 1) && more2) dma_piece(nx2_t, b2, (p - 4) >> 1);  // ... then the ten pieces
            if (p == 24 && more) {
                // the ten newest vector-memory operations of this wave are its pieces of tile t+2 (none if there is no
                // such tile): everything older, i.e. the next tile's halo and the previous tile's stores, has landed
                if (more2) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int m = 0; m < 2; m++) MVD_MFMA16_AB(acc[m], bw[p][s2], af[p % 3][m][s2]);
        }
        asm volatile("s_nop 7\n\ts_nop 4" : "+a"(acc[0]), "+a"(acc[1]));
        {
            const int od = cur_t.od0 + wave;
#pragma unroll
            for (int m = 0; m < 2; m++) {
#pragma unroll
                for (int r = 0; r < 16; r++) pe[m][r] = acc[m][r];
                const int oh = cur_t.oh0 + 4 * m + (i >> 3), ow = cur_t.ow0 + (i & 7);
                pyo[m] = (od < g.Do && oh < g.Ho && ow < g.Wo)
                             ? y1 + ((((size_t)cur_t.n * g.Dy + od) * g.Hy + oh) * g.Wy + ow) * 32 + 8 * h
                             : nullptr;
            }
            have_prev = true;
        }
        if (!more) break;
        cur = b1;
        it += nlocal;
        cur_t = nxt_t;
        nxt_t = nx2_t;
        more = more2;
        more2 = valid(it + 2 * nlocal);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) epilogue_piece(j >> 1, j & 1);  // the last tile
}

// ------------------------------------------------------------------------------------------------ k_fwd16z (round 2)
// z-marching form of the 32 -> 32 channel 3x3x3 stride-1 convolution (same shape class as k_fwd16q).  What k_fwd16q's
// ablation left as the bound was the traffic a 4x8x8 tile pulls through the CU's vector-memory pipe (a 6x10x10 halo:
// 2.34 bytes fetched per byte used) with one wave per SIMD to issue it.  Here a workgroup owns an 8 x 32 (y, x) column
// and walks a chunk of z-planes:
//   * input plane z' arrives once (10 x 34 voxels: 1.33 bytes fetched per byte used, rows of 2 KB contiguous in HBM),
//     register-staged (two planes in flight: loaded three planes ahead of its MFMAs -- one plane ahead left the loaded
//     HBM latency exposed) into a ring of four LDS plane images;
//   * every A fragment read from LDS (input row r, x shift dx, k-step) feeds up to SIX MFMAs: the three dz taps -- they
//     accumulate into three different OUTPUT planes z'+1, z', z'-1, held as a ring of four accumulator sets, so no
//     partial sum ever moves -- times the one or two M tiles (output rows) of the wave that see row r at some dy:
//     24 ds_read_b128 per 108 MFMAs (k_fwd16q: 108);
//   * a wave = two M tiles (two output rows of 32 voxels), the 27 taps' B fragments resident in the accumulator half of
//     the register file as in k_fwd16q; the plane loop is unrolled four times so that ring positions are register names
//     and LDS offsets are immediates;
//   * the output plane completed by plane z' is converted between the MFMAs of plane z'+1 (fourth set), transposed
//     through a 4-KB per-wave LDS scratch and stored as four fully contiguous 1-KB wave stores; the drained set is then
//     re-initialised with the bias from an LDS table (ds_read_b128 straight into the accumulator registers).
// With one wave per SIMD nothing hides behind another wave: everything that is not an MFMA is either precomputed per
// column (the column is fixed for the life of the workgroup), an immediate, a range-checked buffer access (zeros outside
// the plane / dropped stores: no selects, no branches) or dealt out over the 24 fragment slots of a plane.
// A-operand LDS image of a plane: two sub-images (k-step 0: channels 0-15, k-step 1: 16-31) of 340 slots x 32 bytes; the
// two 16-byte halves of a slot are swapped when (slot >> 3) & 1: the 16 lanes of a ds_read_b128 group ({0-3,12-15,20-27},
// ...) read slots whose (slot mod 8) picks the 32-byte column of the 256-byte bank row, and the two lanes that share a
// column are 8 or 24 slots apart -- conflict-free for every dx; the k-step is an immediate offset.
constexpr int Z_ROWS = 10, Z_SLOTS = 34, Z_NSLOT = Z_ROWS * Z_SLOTS, Z_XR = 6, Z_PARTS = Z_NSLOT * 4;
constexpr int Z_HALF = Z_NSLOT * 32 + 64;      // +64: the two sub-images of a slot pair land on different write banks
constexpr int Z_PLANE = Z_HALF + Z_NSLOT * 32;
constexpr int Z_SCR = 4 * Z_PLANE, Z_BIAS = Z_SCR + 4 * 4096, Z_LDS = Z_BIAS + 4096;
static_assert(Z_PLANE + Z_HALF < 65536, "plane / k-step offsets must fit the 16-bit ds_read immediate");

struct Fwd16ZTile {
    int nty, ntx, nzc, zc, nitems;
    int kp, koff;  // produce channels of the packed weight tensor (its row stride), first produce channel of this launch
    int wsel[27];
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef __attribute__((address_space(3))) u32x2 lds_u2;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ inline unsigned cvt_pk_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32 (round-to-nearest-even, as f2bf)
    f32x2 v = {a, b};
    bf16x2v r = __builtin_convertvector(v, bf16x2v);
    return *reinterpret_cast<unsigned *>(&r);
}

// A fragments come straight from ds_read_b128 (no VALU-written operand): no wait states needed in front of the MFMA
#define MVD_MFMA16_ZV(ACC, BFRAG, AFRAG) \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(BFRAG), "v"(AFRAG))

template <int R>
struct ZIdx { static constexpr int value = R; };

// FUSE (round 3, the InstanceNorm fusion of the north_star's block): bit 0 = statistics epilogue -- every 16-byte chunk
// of a finished output plane that passes through the store path (8 bf16 channels of one voxel per lane, the lane's
// channel octet fixed) is also accumulated into per-lane (sum, sum of squares) registers; one cross-lane / cross-wave
// combine at the end of the kernel writes the workgroup's partial [32 channels][2] (fixed order: deterministic).
// bit 1 = input prologue -- the staged input is the RAW output of the producing conv and is normalised + activated
// between the buffer load and the LDS write: a = bf16(lrelu(fma(x, scale[n][c], shift[n][c]))), zeros where the voxel
// lies outside the volume (the zero padding applies to the activation).
template <int DBG, int FUSE = 0>
__global__ __launch_bounds__(256, 1) void k_fwd16z(const FwdGeom g, const Fwd16ZTile tg, const unsigned short *__restrict__ a1,
                                                   const unsigned short *__restrict__ w, const float *__restrict__ bias,
                                                   unsigned short *__restrict__ y1, float *__restrict__ tile_stats,
                                                   const float *__restrict__ in_scale, const float *__restrict__ in_shift,
                                                   const float slope) {
    constexpr int dbg = DBG;
    constexpr bool PRO = (FUSE & 2) != 0;  // 1: no global loads / LDS writes after the prologue, 2: no MFMAs, 4: no stores, 8: no epilogue,
                              // 16: no barrier, 32: no A-fragment reads after the prologue, 64: descriptors not updated,
                              // 128: no buffer loads / stores issued, 256: no bias re-initialisation (ablation builds;
                              // results wrong)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    // XCD-aware item order: each XCD walks a contiguous range of (n, chunk, ty, tx) -- neighbours share halo rows in its L2
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= tg.nitems) return;
    unsigned r_ = (unsigned)item;
    const int tx = (int)(r_ % (unsigned)tg.ntx); r_ /= (unsigned)tg.ntx;
    const int ty = (int)(r_ % (unsigned)tg.nty); r_ /= (unsigned)tg.nty;
    const int zchunk = (int)(r_ % (unsigned)tg.nzc);
    const int n_ = (int)(r_ / (unsigned)tg.nzc);
    const int y0 = ty * 8, x0 = tx * 32, zb = zchunk * tg.zc;
    const int ze = min(zb + tg.zc, g.Do);

    i32x4 bw[27][2];
#pragma unroll
    for (int p = 0; p < 27; p++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const uint4 q = *reinterpret_cast<const uint4 *>(
                w + ((size_t)((tg.wsel[p] * 2 + s2) * 2 + h) * tg.kp + tg.koff + i) * 8);
            bw[p][s2] = *reinterpret_cast<const i32x4 *>(&q);
        }
    // staging slots (column constants): byte offset inside an input plane -- 0xfffffff0 for parts outside the (y, x) plane:
    // the plane is read through a buffer descriptor whose range check returns zeros there (no select, no branch) -- and
    // the LDS offset
    unsigned rel[Z_XR], wa[Z_XR];
#pragma unroll
    for (int u = 0; u < Z_XR; u++) {
        const int idx = u * 256 + tid;
        const int slot = idx < Z_PARTS ? (idx >> 2) : 0;
        const int ry = slot / Z_SLOTS, sx = slot - ry * Z_SLOTS;
        const int gy = y0 - 1 + ry, gx = x0 - 1 + sx;
        const bool in = idx < Z_PARTS && gy >= 0 && gy < g.Hi && gx >= 0 && gx < g.Wi;
        const int part = idx & 3;
        rel[u] = in ? (unsigned)((gy * g.Wi + gx) * 64 + part * 16) : 0xfffffff0u;
        wa[u] = lbase + (part >> 1) * Z_HALF + slot * 32 + (((part & 1) ^ ((slot >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int u = 0; u < Z_XR; u++) asm volatile("" : "+v"(rel[u]), "+v"(wa[u]));  // one register each, no re-derivation per plane
    // A-operand read addresses: input row r of this wave, x shift dx (k-step and image parity are immediates; images 2, 3
    // lie beyond the 16-bit immediate: one v_add per read there -- a second set of 12 address registers spilled)
    unsigned ra[4][3];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int slot = (2 * wave + r) * Z_SLOTS + dx + i;
            ra[r][dx] = lbase + slot * 32 + ((h ^ ((slot >> 3) & 1)) << 4);
            asm volatile("" : "+v"(ra[r][dx]));
        }
    // the bias in accumulator layout (register r <-> output channel (r & 3) + 8 * (r >> 2) + 4 * h) lives in LDS, 16 bytes
    // per lane and register group: a drained accumulator set is re-initialised from there by four ds_read_b128 per M tile,
    // dealt out over fragment slots (no VALU; as the C operand of a first MFMA it would cost 16 arch VGPRs the kernel
    // does not have)
    unsigned bsc = lbase + Z_BIAS + lane * 16;
    asm volatile("" : "+v"(bsc));
    {
        u32x4 bq[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) b4 = *reinterpret_cast<const float4 *>(bias + tg.koff + 8 * rg + 4 * h);
            bq[rg] = u32x4{__float_as_uint(b4.x), __float_as_uint(b4.y), __float_as_uint(b4.z), __float_as_uint(b4.w)};
        }
        if (wave == 0) {
#pragma unroll
            for (int rg = 0; rg < 4; rg++) *(lds_u4 *)(bsc + rg * 1024) = bq[rg];
        }
    }
    // output transposition through the wave's 4-KB scratch: lane (i, h) holds, per M tile m and register group rg, the 4
    // channels 8 rg + 4 h .. +3 of voxel i (8 bytes packed) -> scratch [m][voxel][64 B] with the 16-byte units XORed by
    // (voxel >> 1) & 3 (ds_write_b64, 2-way conflicts: 8 LDS cycles against 6 of transfer); read back as lane L = 16-byte
    // chunk L & 3 of voxel 16 (q & 1) + L / 4 of row q >> 1 (conflict-free) = 1 KB contiguous in HBM per wave store
    const unsigned scr = lbase + Z_SCR + wave * 4096;
    unsigned wsc[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) wsc[rg] = scr + i * 64 + ((rg ^ ((i >> 1) & 3)) << 4) + h * 8;
    unsigned rsc = scr + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4);
    asm volatile("" : "+v"(rsc), "+v"(wsc[0]), "+v"(wsc[1]), "+v"(wsc[2]), "+v"(wsc[3]));
    // byte offset of this lane's chunk inside an output plane for store q (row q >> 1, voxels 16 (q & 1) ..): out of range
    // (dropped by the descriptor's range check) when the voxel lies outside the volume
    unsigned voff[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int oh = y0 + 2 * wave + (q >> 1), ow = x0 + (q & 1) * 16 + (lane >> 2);
        voff[q] = (oh < g.Ho && ow < g.Wo) ? (unsigned)((oh * g.Wy + ow) * 64 + (lane & 3) * 16) : 0xfffffff0u;
    }
    asm volatile("" : "+v"(voff[0]), "+v"(voff[1]), "+v"(voff[2]), "+v"(voff[3]));
    // FUSE: the lane's channel octet is fixed on both paths -- staging part tid & 3 (PRO), store chunk lane & 3 (ST)
    float psc[PRO ? 8 : 1], psh[PRO ? 8 : 1];
    if (PRO) {
        const float *sp = in_scale + (size_t)n_ * 32 + (tid & 3) * 8, *tp = in_shift + (size_t)n_ * 32 + (tid & 3) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            psc[e] = sp[e];
            psh[e] = tp[e];
        }
    }
    const size_t oplane = (size_t)g.Hy * g.Wy * 64;
    char *ybase = reinterpret_cast<char *>(y1) + (size_t)n_ * g.Dy * oplane;
    const char *abase = reinterpret_cast<const char *>(a1) + (size_t)n_ * g.Di * g.Hi * g.Wi * 64;
    const size_t iplane = (size_t)g.Hi * g.Wi * 64;
    const unsigned iplane32 = (unsigned)iplane, oplane32 = (unsigned)oplane;

    // plane index j <-> input plane z' = zb - 1 + j; planes 0 .. nproc-1 carry MFMAs (when inside the volume), plane
    // nproc only drains the last output plane
    const int nproc = (ze - zb) + 2;
    auto zin = [&](int j) { return zb - 1 + j; };
    auto live = [&](int j) { const int z = zin(j); return j < nproc && z >= 0 && z < g.Di; };  // block-uniform
    u32x4 v[2][Z_XR];  // two planes in flight: plane p is staged in v[p & 1] (loaded three planes ahead of its MFMAs)
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase), 0, 0, 0x00020000);
    auto set_in_plane = [&](int j) {  // descriptor of input plane zin(j) (uniform); zero records = a plane of zeros
        rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase + (size_t)max(zin(j), 0) * iplane), 0,
                                                live(j) ? (int)iplane32 : 0, 0x00020000);
    };
    auto stage_load = [&](int set, int u) { v[set][u] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)rel[u], 0, 0); };
    // PRO: normalise + activate the staged 8 channels (4 VALU per element + one conversion per pair), zeros for parts
    // outside the (y, x) plane or of a plane that does not exist (`lv`, block-uniform)
    auto prologue = [&](u32x4 q, int u, bool lv) {
        const bool ok = lv && rel[u] != 0xfffffff0u;
        unsigned d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float lo = __uint_as_float(d[e] << 16), hi = __uint_as_float(d[e] & 0xffff0000u);
            lo = __builtin_fmaf(lo, psc[2 * e], psh[2 * e]);
            hi = __builtin_fmaf(hi, psc[2 * e + 1], psh[2 * e + 1]);
            lo = fmaxf(lo, lo * slope);
            hi = fmaxf(hi, hi * slope);
            d[e] = ok ? cvt_pk_bf16(lo, hi) : 0u;
        }
        return u32x4{d[0], d[1], d[2], d[3]};
    };
    // (fused variants: slot = 64 u + tid / 4, so the swizzle bit (slot >> 3) & 1 does not depend on u and the LDS address of
    // part u is wa[0] + 2048 u -- an immediate; five registers fewer)
    auto stage_write = [&](int set, unsigned imgoff, int u, bool lv = true) {
        const u32x4 q = PRO ? prologue(v[set][u], u, lv) : v[set][u];
        const unsigned a_ = FUSE ? wa[0] + 2048u * u : wa[u];
        if (u < Z_XR - 1 || tid < Z_PARTS - (Z_XR - 1) * 256) *(lds_u4 *)(a_ + imgoff) = q;
    };
    auto load_plane = [&](int set, int j) {
        set_in_plane(j);
#pragma unroll
        for (int u = 0; u < Z_XR; u++) stage_load(set, u);
    };
    auto store_plane = [&](int set, unsigned imgoff, bool lv) {
#pragma unroll
        for (int u = 0; u < Z_XR; u++) stage_write(set, imgoff, u, lv);
    };
    i32x4 af[3];  // ring of three A fragments, fetched two fragments ahead (also across planes)
    // fragment f = (r * 3 + dx) * 2 + ks of plane image IMG
#define MVD_Z_READ_A(IMG, F, BUF)                                                                                    \
    {                                                                                                                \
        const u32x4 q_ = *(lds_u4 *)(ra[(F) / 6][((F) >> 1) % 3] + (IMG) * Z_PLANE + ((F) & 1) * Z_HALF);            \
        af[BUF] = *reinterpret_cast<const i32x4 *>(&q_);                                                             \
    }

    // (Statistics epilogue, FUSE bit 0: NOT in this kernel.  Measured in round 3 on three forms -- 16 running sums in VGPRs,
    // transient sums aliased onto the dead accumulator set with AGPR-pinned totals, private LDS slots updated with
    // ds_add_f32 -- the first two push staged input planes or weight fragments into scratch memory (the kernel runs at
    // 254 + 236 of 512 registers), the third makes the launch seven times slower (1.47 ms: LDS float atomics).  The
    // statistics epilogue lives in k_fwd16y, whose 16x16x32 tiling leaves a lane 4 channels instead of 16.)
    f32x16 S[4][2];  // accumulator ring: output plane zo lives in S[(zo - zb + 1) & 3]
    auto bias_init = [&](f32x16 &acc, int rg) {  // acc[4 rg .. 4 rg + 3] = bias
        const u32x4 q = *(lds_u4 *)(bsc + rg * 1024);
        acc[rg * 4 + 0] = __uint_as_float(q.x); acc[rg * 4 + 1] = __uint_as_float(q.y);
        acc[rg * 4 + 2] = __uint_as_float(q.z); acc[rg * 4 + 3] = __uint_as_float(q.w);
    };
    load_plane(0, 0);
    store_plane(0, 0, live(0));
    load_plane(1, 1);
    load_plane(0, 2);
#pragma unroll
    for (int p = 0; p < 27; p++) asm volatile("" : "+a"(bw[p][0]), "+a"(bw[p][1]));
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 32; q++) bias_init(S[q >> 3][(q >> 2) & 1], q & 3);
    MVD_Z_READ_A(0, 0, 0);
    MVD_Z_READ_A(0, 1, 1);

    // plane j with ring position R = j & 3: input image R, accumulator sets NEW = R (output plane z'+1: first
    // contribution, holds the bias), MID = R-1 (z'), OLD = R-2 (z'-1: complete after this plane), DRAIN = R-3 (z'-2:
    // completed by the previous plane, stored during this one, then reset to the bias: it is the next plane's NEW).
    // ONE branch-free body for every plane: a plane that does not exist (z' = -1 or D at the volume faces, and the
    // drain-only plane after the chunk) is a plane of zeros -- its descriptor has zero records, so the staging loads
    // return zeros without touching memory -- and an output plane that is not the chunk's is a descriptor with zero
    // records, so its stores are dropped.  Ablation (MVD_FWD16Z_DBG=61: MFMAs only) had shown 38.7 cycles per MFMA
    // against 32.4 for the bare loop: ~60 scalar instructions and five branches evaluating plane flags in one clump at
    // every plane boundary, and an in-order wave issues no MFMA behind them (a second, flag-free copy of the body for
    // the interior planes made the register allocator spill 160 registers).  The price: 3 of the Zc + 3 planes of a chunk
    // carry partly or wholly useless MFMAs.
    auto plane = [&](auto Rc, int j) __attribute__((always_inline)) {
        constexpr int R = decltype(Rc)::value;
        constexpr int NEW = R, MID = (R + 3) & 3, OLD = (R + 2) & 3, DRN = (R + 1) & 3, RN = (R + 1) & 3;
        __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, 0, 0x00020000);
        const bool lvn = live(j + 1);     // PRO: the plane written to LDS during this plane exists
        auto set_out_plane = [&]() {  // the plane held by DRN
            const int zo = zin(j) - 2;
            const bool st = zo >= zb && zo < ze && !(dbg & 4);
            rout = __builtin_amdgcn_make_buffer_rsrc(ybase + (size_t)max(zo, 0) * oplane, 0, st ? (int)oplane32 : 0, 0x00020000);
        };
        auto set_next_in_plane = [&]() {  // input plane j + 3
            const int z2 = zin(j + 3);
            const bool lv = j + 3 < nproc && z2 >= 0 && z2 < g.Di && !(dbg & 1);
            rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase + (size_t)max(z2, 0) * iplane), 0,
                                                    lv ? (int)iplane32 : 0, 0x00020000);
        };
        auto epi_pack = [&](int m, int rg) {  // 2 conversions + 1 ds_write_b64
            u32x2 q;
            q.x = cvt_pk_bf16(S[DRN][m][rg * 4 + 0], S[DRN][m][rg * 4 + 1]);
            q.y = cvt_pk_bf16(S[DRN][m][rg * 4 + 2], S[DRN][m][rg * 4 + 3]);
            *(lds_u2 *)(wsc[rg] + m * 2048) = q;
        };
        u32x4 ob;
        auto epi_read = [&](int q) { ob = *(lds_u4 *)(rsc + q * 1024); };
        auto epi_store = [&](int q) { __builtin_amdgcn_raw_buffer_store_b128(ob, rout, (int)voff[q], 0, 0); };
        constexpr unsigned IMGN = RN * Z_PLANE;
        // everything that is not an MFMA is dealt out over the 24 fragment slots of the plane (an MFMA leaves ~24 issue
        // cycles per 32; a clump in front of the plane is paid in full):
        //   f 1-14  one staging ds_write_b128 (plane j+1, loaded during plane j-2) in six of them
        //   f 1-4   epilogue of the drained set: 4 conversions + 2 ds_write_b64 each;  f 0, 4 the two descriptors
        //   f 3-23  one global load (plane j+3, into the registers just written out) in every fourth
        //   f 6-13  scratch read-back and one 1-KB output store, alternating
        //   f 14-21 one ds_read_b128 of the bias table into the drained set each
        //   f 17-20 barrier (plane j+1 visible), wave w in slot 17 + w;   f 22-23 first fragments of plane j+1
#pragma unroll
        for (int f = 0; f < 24; f++) {
            if (dbg & 32) {
            } else if (f + 2 < 24) MVD_Z_READ_A(R, f + 2, (f + 2) % 3)
            else MVD_Z_READ_A(RN, f + 2 - 24, (f + 2) % 3)  // behind the barrier below
            // The four waves run the same instruction stream, and one barrier per plane would keep them in lockstep: 24
            // staging loads (and 24 13-cycle LDS store transfers) in one window of the CU's single vector-memory pipe.
            // So wave w takes its barrier one fragment slot later than wave w-1 (slots 17 + w: behind its own last staging
            // write, ahead of its first read of the next image) -- after the first plane the waves run w slots apart --
            // and the staging loads sit in slots = 3 mod 4, the writes in slots = 1, 2 mod 4: no two waves load in the
            // same slot time, and the compiler can still count vmcnt (conditional loads cost it the count).
            if (!(dbg & 1)) {
                constexpr int wslot[6] = {1, 2, 6, 9, 10, 14};
#pragma unroll
                for (int u = 0; u < 6; u++)
                    if (f == wslot[u]) stage_write(RN & 1, IMGN, u, lvn);
            }
            if (f == 4 && !(dbg & 64)) set_out_plane();
            if (f == 0 && !(dbg & 64)) set_next_in_plane();
            if (f == 1) asm volatile("" : "+v"(S[DRN][0]), "+v"(S[DRN][1]));  // keeps the conversions in their slots
            if (f >= 1 && f <= 4 && !(dbg & 8)) {
                epi_pack((f - 1) >> 1, ((f - 1) & 1) * 2);
                epi_pack((f - 1) >> 1, ((f - 1) & 1) * 2 + 1);
            }
            if ((f & 3) == 3 && !(dbg & 128)) stage_load(RN & 1, f >> 2);
            if (f >= 6 && f < 14 && !(f & 1) && !(dbg & 8)) epi_read((f - 6) >> 1);
            if (f >= 6 && f < 14 && (f & 1) && !(dbg & (8 | 128))) epi_store((f - 7) >> 1);
            if (f >= 14 && f < 22 && !(dbg & 256)) bias_init(S[DRN][(f - 14) >> 2], (f - 14) & 3);  // the next plane's NEW set
            if (f >= 17 && f <= 20 && !(dbg & 16)) {
                // (this wave's plane writes retired in order before the fragment reads already consumed above)
                if (wave == f - 17) asm volatile("s_barrier" ::: "memory");
            }
            const int r = f / 6, dx = (f >> 1) % 3, ks = f & 1;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int dy = r - m;  // input row r is output row m shifted by dy - 1
                if (dy < 0 || dy > 2) continue;
                if (dbg & 2) continue;
                MVD_MFMA16_ZV(S[NEW][m], bw[0 * 9 + dy * 3 + dx][ks], af[f % 3]);
                MVD_MFMA16_ZV(S[MID][m], bw[1 * 9 + dy * 3 + dx][ks], af[f % 3]);
                MVD_MFMA16_ZV(S[OLD][m], bw[2 * 9 + dy * 3 + dx][ks], af[f % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // an MFMA result needs 12 wait states before anything but an accumulating MFMA reads it
        asm volatile("s_nop 7\n\ts_nop 4" : "+v"(S[OLD][0]), "+v"(S[OLD][1]));
    };
    for (int j = 0; j <= ((dbg & 512) ? -1 : nproc); j += 4) {
        plane(ZIdx<0>(), j);
        if (j + 1 > nproc) break;
        plane(ZIdx<1>(), j + 1);
        if (j + 2 > nproc) break;
        plane(ZIdx<2>(), j + 2);
        if (j + 3 > nproc) break;
        plane(ZIdx<3>(), j + 3);
    }
#undef MVD_Z_READ_A
}

static int num_cus16() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// -1: not this kernel's shape
static int launch_fwd16p(const FwdGeom &g, const unsigned short *a1, const unsigned short *w, const float *bias,
                         unsigned short *y1, unsigned short *y2, hipStream_t s, const Fwd16Fuse *fuse = nullptr,
                         int *stats_tiles_only = nullptr) {
    static const int off = getenv("MVD_FWD16P") ? (atoi(getenv("MVD_FWD16P")) == 0) : 0;
    if (off) return -1;
    // 32 reduce channels; 32 produce channels, or 32 + 32 into two tensors (the input gradient of a conv that read two
    // concatenated 32-channel tensors: two independent 32 -> 32 problems on the same dy; k_fwd16z only)
    const bool two_out = g.K1 == 32 && g.K2 == 32 && y2 != nullptr;
    if (g.C1 != 32 || g.C2 != 0 || g.K1 != 32 || (g.K2 != 0 && !two_out) || g.ntaps != 27 || g.T != 27 || g.acc) return -1;
    for (int a = 0; a < 3; a++)
        if (g.sa[a] != 1 || g.so[a] != 1 || g.oo[a] != 0) return -1;
    if (g.Dy != g.Do || g.Hy != g.Ho || g.Wy != g.Wo) return -1;
    int mn[3] = {127, 127, 127}, mx[3] = {-127, -127, -127};
    for (int t = 0; t < 27; t++)
        for (int a = 0; a < 3; a++) {
            if (g.off[t][a] < mn[a]) mn[a] = g.off[t][a];
            if (g.off[t][a] > mx[a]) mx[a] = g.off[t][a];
        }
    for (int a = 0; a < 3; a++)
        if (mn[a] != -1 || mx[a] != 1) return -1;
    Fwd16Tile tg;
    memset(&tg, 0, sizeof(tg));
    tg.EH = 10; tg.EW = 10; tg.nslots = 600;
    for (int t = 0; t < 27; t++)
        tg.toff[t] = ((g.off[t][0] + 1) * tg.EH + (g.off[t][1] + 1)) * tg.EW + (g.off[t][2] + 1);
    tg.ntd = (g.Do + 3) / 4;
    tg.nth = (g.Ho + 7) / 8;
    tg.ntw = (g.Wo + 7) / 8;
    const long nitems = (long)g.N * tg.ntd * tg.nth * tg.ntw;
    const long ncu = ((long)num_cus16() / 8) * 8;
    if (nitems < 4 * ncu || nitems > (1L << 30)) return -1;  // a walk of >= 4 tiles per workgroup or it does not pay
    if ((long)g.N * g.Di * g.Hi * g.Wi * 32 >= (1L << 31)) return -1;  // 32-bit element offsets inside a tile
    tg.nitems = (int)nitems;
    static const int use_q = getenv("MVD_FWD16Q") ? atoi(getenv("MVD_FWD16Q")) : 1;
    if (use_q) {
        Fwd16QTile tq;
        memset(&tq, 0, sizeof(tq));
        tq.ntd = tg.ntd; tq.nth = tg.nth; tq.ntw = tg.ntw; tq.nitems = tg.nitems;
        bool ok = true;
        for (int p = 0; p < 27 && ok; p++) {
            const int dz = p / 9 - 1, dy = (p / 3) % 3 - 1, dx = p % 3 - 1;
            int hit = -1;
            for (int t = 0; t < 27; t++)
                if (g.off[t][0] == dz && g.off[t][1] == dy && g.off[t][2] == dx) hit = t;
            if (hit < 0) ok = false;
            else tq.wsel[p] = g.wt[hit];
        }
        // measured (round 2, enc0.conv1 dgrad, bench_conv --iters 40): k_fwd16r 0.282 ms, k_fwd16q 0.258 ms -- the DMA
        // pieces cost the wave more issue time than ten register loads + ten ds_write_b128.  Selectable, off by default.
        static const int use_z = getenv("MVD_FWD16Z") ? atoi(getenv("MVD_FWD16Z")) : 1;  // default (round 2): k_fwd16z
        if (ok && use_z) {
            Fwd16ZTile tz;
            memset(&tz, 0, sizeof(tz));
            memcpy(tz.wsel, tq.wsel, sizeof(tz.wsel));
            tz.nty = (g.Ho + 7) / 8;
            tz.ntx = (g.Wo + 31) / 32;
            // z chunks: enough workgroups to fill the chip in whole rounds, chunks of >= 8 planes (2 extra input planes
            // and two partly useful MFMA planes per chunk)
            const long cols = (long)g.N * tz.nty * tz.ntx;
            int best = 1;
            double best_cost = 1e30;
            for (int nz = 1; nz <= (g.Do + 7) / 8; nz++) {
                const int zc = (g.Do + nz - 1) / nz;
                const long wgs = cols * ((g.Do + zc - 1) / zc);
                const double cost = (double)((wgs + ncu - 1) / ncu) * (zc + 2.5);
                if (cost < best_cost - 1e-9) { best_cost = cost; best = nz; }
            }
            tz.zc = (g.Do + best - 1) / best;
            tz.nzc = (g.Do + tz.zc - 1) / tz.zc;
            tz.nitems = (int)(cols * tz.nzc);
            if (stats_tiles_only) {  // query: tiles per sample of the statistics epilogue (none here); 0 = this kernel runs
                *stats_tiles_only = 0;
                return 0;
            }
            static const int dbgz = getenv("MVD_FWD16Z_DBG") ? atoi(getenv("MVD_FWD16Z_DBG")) & 1023 : 0;
            typedef void (*kz_t)(const FwdGeom, const Fwd16ZTile, const unsigned short *, const unsigned short *, const float *,
                                 unsigned short *, float *, const float *, const float *, const float);
            kz_t kfn = k_fwd16z<0>;
            switch (dbgz) {
                case 1: kfn = k_fwd16z<1>; break;
                case 2: kfn = k_fwd16z<2>; break;
                case 4: kfn = k_fwd16z<4>; break;
                case 5: kfn = k_fwd16z<5>; break;
                case 13: kfn = k_fwd16z<13>; break;
                case 29: kfn = k_fwd16z<29>; break;
                case 61: kfn = k_fwd16z<61>; break;
                case 125: kfn = k_fwd16z<125>; break;
                case 189: kfn = k_fwd16z<189>; break;
                case 317: kfn = k_fwd16z<317>; break;
                case 509: kfn = k_fwd16z<509>; break;
                case 1021: kfn = k_fwd16z<1021>; break;
                default: break;
            }
            const bool want_stats = false;  // (k_fwd16y has the statistics epilogue)
            const bool want_pro = fuse && fuse->in_scale && fuse->in_shift;
            if (want_pro && two_out) return -1;
            const int fz = (want_stats ? 1 : 0) | (want_pro ? 2 : 0);
            if (fz == 2) kfn = k_fwd16z<0, 2>;
            if (fuse && fuse->ntiles) *fuse->ntiles = want_stats ? tz.nzc * tz.nty * tz.ntx : 0;
            static PerDeviceFlag configured_z[4];
            if (!configured_z[fz]()) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z_LDS);
                if (e != hipSuccess) {
                    set_error("conv fwd16z: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
                    return 1;
                }
                configured_z[fz]() = true;
            }
            const int per_xcd = (tz.nitems + 7) / 8;
            tz.kp = g.K1 + g.K2;
            for (int q = 0; q < (two_out ? 2 : 1); q++) {
                tz.koff = 32 * q;
                hipLaunchKernelGGL(kfn, dim3((unsigned)(per_xcd * 8)), dim3(256), (size_t)Z_LDS, s, g,
                                   tz, a1, w, bias, q ? y2 : y1, want_stats ? fuse->tile_stats : nullptr, want_pro ? fuse->in_scale : nullptr,
                                   want_pro ? fuse->in_shift : nullptr, fuse ? fuse->slope : 0.f);
                if (check_launch("conv fwd16z (z-marching bf16 mfma, weights in registers)")) return 1;
            }
            return 0;
        }
        if (stats_tiles_only) { *stats_tiles_only = 0; return 1; }
        if (fuse && fuse->in_scale) return -1;   // only the z-marching kernel has the input prologue
        if (fuse && fuse->ntiles) *fuse->ntiles = 0;
        if (two_out) return -1;
        static const int use_r = getenv("MVD_FWD16R") ? atoi(getenv("MVD_FWD16R")) : 0;
        if (ok && use_r) {
            static PerDeviceFlag configured_r;
            if (!configured_r()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd16r), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_LIMIT16) != hipSuccess) {
                    set_error("conv fwd16r: cannot raise the dynamic LDS limit");
                    return 1;
                }
                configured_r() = true;
            }
            hipLaunchKernelGGL(k_fwd16r, dim3((unsigned)ncu), dim3(Q_TPB), 3 * (size_t)R_BUF, s, g, tq, a1, w, bias, y1);
            return check_launch("conv fwd16r (persistent bf16 mfma, LDS-DMA halo)");
        }
        if (ok) {
            static const int dbgq = getenv("MVD_FWD16Q_DBG") ? atoi(getenv("MVD_FWD16Q_DBG")) & 15 : 0;
            typedef void (*kq_t)(const FwdGeom, const Fwd16QTile, const unsigned short *, const unsigned short *, const float *,
                                 unsigned short *);
            static const kq_t kq[16] = {k_fwd16q<0>, k_fwd16q<1>, k_fwd16q<2>, k_fwd16q<3>, k_fwd16q<4>, k_fwd16q<5>,
                                        k_fwd16q<6>, k_fwd16q<7>, k_fwd16q<8>, k_fwd16q<9>, k_fwd16q<10>, k_fwd16q<11>,
                                        k_fwd16q<12>, k_fwd16q<13>, k_fwd16q<14>, k_fwd16q<15>};
            static PerDeviceFlag configured_q;
            if (!configured_q()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kq[dbgq]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)LDS_LIMIT16) != hipSuccess) {
                    set_error("conv fwd16q: cannot raise the dynamic LDS limit");
                    return 1;
                }
                configured_q() = true;
            }
            hipLaunchKernelGGL(kq[dbgq], dim3((unsigned)ncu), dim3(Q_TPB), 2 * (size_t)Q_HALO, s, g, tq, a1, w, bias, y1);
            return check_launch("conv fwd16q (persistent bf16 mfma, weights in registers)");
        }
    }
    if (two_out) return -1;
    if (stats_tiles_only) { *stats_tiles_only = 0; return 1; }
    if (fuse && fuse->in_scale) return -1;
    if (fuse && fuse->ntiles) *fuse->ntiles = 0;
    const size_t lds = (size_t)P_WB + 2 * (size_t)P_HALO;
    static PerDeviceFlag configured;
    if (!configured()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd16p), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_LIMIT16) != hipSuccess) {
            set_error("conv fwd16p: cannot raise the dynamic LDS limit");
            return 1;
        }
        configured() = true;
    }
    static const int dbg = getenv("MVD_FWD16P_DBG") ? atoi(getenv("MVD_FWD16P_DBG")) : 0;
    hipLaunchKernelGGL(k_fwd16p, dim3((unsigned)ncu), dim3(P_TPB), lds, s, g, tg, a1, w, bias, y1, dbg);
    return check_launch("conv fwd16p (persistent bf16 mfma)");
}

// returns 0 ok, >0 error, -1 unsupported shape
int fwd_bf16_stats_tiles(const FwdGeom &g) {
    int nt = 0;
    if ((g.C1 + g.C2) % 32 || (g.K1 + g.K2) % 32) return 0;
    if (launch_fwd16y(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (num_cus16() / 8) * 8, &nt) == 0)
        return nt;
    const int r = launch_fwd16p(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &nt);
    return r == 0 ? nt : 0;
}

int fwd_bf16_prologue_ok(const FwdGeom &g) {  // 1: the shape runs on a kernel with the InstanceNorm input prologue
    int nt = 0;
    if ((g.C1 + g.C2) % 32 || (g.K1 + g.K2) % 32 || (g.K2 != 0)) return 0;
    if ((g.C1 == 32 || g.C1 == 64) && g.C2 == 0 &&  // one producer tensor of 32 or 64 channels (k_fwd16y, one or two chunks)
        launch_fwd16y(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (num_cus16() / 8) * 8, &nt) == 0)
        return 1;
    if (g.C1 != 32 || g.C2 != 0) return 0;
    return launch_fwd16p(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &nt) == 0 ? 1 : 0;
}

int fwd_bf16(const FwdGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *w,
             const float *bias, unsigned short *y1, unsigned short *y2, void *ws, size_t ws_bytes, hipStream_t s,
             const Fwd16Fuse *fuse) {
    const int C = g.C1 + g.C2, K = g.K1 + g.K2;
    if (fuse && fuse->ntiles) *fuse->ntiles = 0;
    if (g.ntaps < 1 || g.ntaps > 27) return -1;
    if (C % 32 != 0 || g.C1 % 32 != 0 || g.C2 % 32 != 0) return -1;
    if (K % 32 != 0 || g.K1 % 32 != 0 || g.K2 % 32 != 0) return -1;
    if (((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)w) & 15) return -1;
    int mn[3] = {127, 127, 127}, mx[3] = {-127, -127, -127};
    for (int t = 0; t < g.ntaps; t++)
        for (int a = 0; a < 3; a++) {
            if (g.off[t][a] < mn[a]) mn[a] = g.off[t][a];
            if (g.off[t][a] > mx[a]) mx[a] = g.off[t][a];
        }
    {
        int r = launch_fwd16y(g, a1, a2, w, bias, y1, y2, s, fuse, (num_cus16() / 8) * 8, nullptr);
        if (r >= 0) return r;
        r = launch_fwd16p(g, a1, w, bias, y1, y2, s, fuse);
        if (r >= 0) return r;
        if (!fuse || (!fuse->in_scale && !fuse->tile_stats)) {
            r = launch_fwd16ys(g, a1, a2, w, bias, y1, y2, s, (num_cus16() / 8) * 8);
            if (r >= 0) return r;
        }
    }
    if (fuse && fuse->in_scale) return -1;  // the generic kernels have no input prologue
    const int NT = (K % 64 == 0) ? 2 : 1;
    auto magic = [](int d, int nmax) -> int {
        int m = (1 << 20) / d + 1;  // 20-bit reciprocal: n < 2048 keeps n*m inside int32
        if (nmax > 2048) return -1;
        for (int n = 0; n < nmax; n++)
            if (((n * m) >> 20) != n / d) return -1;
        return m;
    };
    // 3x3x3 stride-1 layers with K % 64 == 0 and at least one 8x8x8 tile per CU: eight waves on a 512-voxel tile -- the
    // weights of a chunk (the larger stream: 110 KB per 32 channels against a 38 KB halo) are fetched once per 512 voxels
    // instead of once per 256, the halo shrinks from 2.34 to 1.95 bytes per byte used (MVD_FWD16_W8=0: four waves)
    static const int w8 = getenv("MVD_FWD16_W8") ? atoi(getenv("MVD_FWD16_W8")) : 1;
    bool unit3 = g.ntaps == 27;
    for (int a = 0; a < 3; a++) unit3 = unit3 && g.sa[a] == 1 && mn[a] == -1 && mx[a] == 1;
    if (w8 && unit3 && NT == 2 && g.Do >= 8) {
        Fwd16Tile tg;
        memset(&tg, 0, sizeof(tg));
        tg.EH = 10; tg.EW = 10; tg.nslots = 1000;
        for (int a = 0; a < 3; a++) tg.min_off[a] = mn[a];
        tg.magHW = magic(100, 1024);
        tg.magW = magic(10, 100);
        for (int t = 0; t < g.ntaps; t++) {
            tg.toff[t] = ((g.off[t][0] - mn[0]) * tg.EH + (g.off[t][1] - mn[1])) * tg.EW + (g.off[t][2] - mn[2]);
            tg.tdy[t] = (signed char)(g.off[t][1] - mn[1]);
        }
        tg.ntd = (g.Do + 7) / 8;
        tg.nth = (g.Ho + 7) / 8;
        tg.ntw = (g.Wo + 7) / 8;
        tg.K = K;
        const long wgs8 = (long)g.N * tg.ntd * tg.nth * tg.ntw * (K / 64);
        // measured (round 2, bench_conv --iters 40): 32^3 x 128..256 channels 0.060 / 0.110 ms against 0.064 / 0.117 with four
        // waves; at 64^3 (>= 1024 workgroups) the two co-resident four-wave workgroups overlap better: 0.161 vs 0.160 forward,
        // 0.150 vs 0.142 input gradient -- so only where a CU gets at most a few tiles
        if (tg.magHW >= 0 && tg.magW >= 0 && wgs8 >= 256 && wgs8 < 1024) {
            // (round 3: eight instead of four taps per weight group -- on the idea that a group's 16 KB of weights arrive from
            // L2 later than its 32 MFMAs per wave retire -- measured no different: 0.070 vs 0.069 ms on 128 -> 128 @32^3, 37 %
            // MFMA-busy either way)
            static const int w8swz = getenv("MVD_FWD16_W8SWZ") ? atoi(getenv("MVD_FWD16_W8SWZ")) : 0;
            int r = w8swz ? launch_fwd16<2, 2, 4, 8, true, 8>(g, tg, a1, a2, w, bias, y1, y2, ws, ws_bytes, s)
                          : launch_fwd16<2, 2, 4, 8, false, 8>(g, tg, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
            if (r >= 0) return r;
        }
    }
    // (3x3x3 stride 2, tried in round 3: a 2 x 4 x 8 output tile on two waves -- 765 halo slots, two workgroups per CU so that one's
    // halo fetch overlaps the other's MFMAs -- measured SLOWER than the 4 x 4 x 8 tile below: 32 -> 64 @128^3 0.295 vs 0.229 ms,
    // 64 -> 128 @64^3 0.111 vs 0.089.  The time of these launches is the per-tap-group weight staging through LDS with its
    // barriers, not the halo: the fix is a z-marching kernel with register-resident weights and a parity-split LDS image.)
    // tile candidates: 4x8x8 (MT = 2) when its halo fits 640 slots, else 4x4x8
    for (int MT = 2; MT >= 1; MT--) {
        Fwd16Tile tg;
        memset(&tg, 0, sizeof(tg));
        const int T3[3] = {4, 4 * MT, 8};
        int E[3];
        for (int a = 0; a < 3; a++) {
            E[a] = (T3[a] - 1) * g.sa[a] + (mx[a] - mn[a]) + 1;
            tg.min_off[a] = mn[a];
        }
        tg.EH = E[1]; tg.EW = E[2];
        tg.nslots = E[0] * E[1] * E[2];
        const int xr_need = (tg.nslots * 4 + 255) / 256;
        int XR;
        if (MT == 2 && xr_need <= 10) XR = 10;
        else if (MT == 1 && xr_need <= 6) XR = 6;
        else if (MT == 1 && xr_need <= 22) XR = 22;
        else continue;
        tg.magHW = magic(tg.EH * tg.EW, XR * 64);
        tg.magW = magic(tg.EW, tg.EH * tg.EW);
        if (tg.magHW < 0 || tg.magW < 0) continue;
        for (int t = 0; t < g.ntaps; t++)
            tg.toff[t] = ((g.off[t][0] - mn[0]) * tg.EH + (g.off[t][1] - mn[1])) * tg.EW + (g.off[t][2] - mn[2]);
        tg.ntd = (g.Do + 3) / 4;
        tg.nth = (g.Ho + 4 * MT - 1) / (4 * MT);
        tg.ntw = (g.Wo + 7) / 8;
        tg.K = K;
        for (int t = 0; t < g.ntaps; t++) tg.tdy[t] = (signed char)(g.off[t][1] - mn[1]);
        // measured (round 2, tools/bench_conv.py --dtype bf16): conflict-free reads do NOT pay in this kernel -- with two
        // workgroups per CU it is bound by instruction issue, and the per-read swizzle arithmetic costs 12-15 %
        // (64->64 @64^3: 0.151 -> 0.174 ms).  Kept selectable (MVD_FWD16_SWZ=1) as the measured alternative.
        static const int swz_on = getenv("MVD_FWD16_SWZ") ? atoi(getenv("MVD_FWD16_SWZ")) : 0;
        const bool swz = swz_on && g.sa[0] == 1 && g.sa[1] == 1 && g.sa[2] == 1;
        int r = -1;
#define MVD_L16(NT_, MT_, TG_, XR_)                                                                              \
    (swz ? launch_fwd16<NT_, MT_, TG_, XR_, true>(g, tg, a1, a2, w, bias, y1, y2, ws, ws_bytes, s)             \
         : launch_fwd16<NT_, MT_, TG_, XR_, false>(g, tg, a1, a2, w, bias, y1, y2, ws, ws_bytes, s))
        if (MT == 2) {
            r = NT == 2 ? MVD_L16(2, 2, 3, 10) : MVD_L16(1, 2, 4, 10);
        } else if (XR == 6) {
            r = NT == 2 ? MVD_L16(2, 1, 3, 6) : MVD_L16(1, 1, 4, 6);
        } else {
            r = NT == 2 ? MVD_L16(2, 1, 3, 22) : MVD_L16(1, 1, 4, 22);
        }
#undef MVD_L16
        if (r >= 0) return r;
    }
    return -1;
}

int pack_weight16(const float *w, unsigned short *wf, unsigned short *wb, int K, int C, int T, int transposed,
                  hipStream_t s) {
    long total = (long)K * C * T;
    hipLaunchKernelGGL(k_pack_weight16, dim3(cdiv(total, 256)), dim3(256), 0, s, w, wf, wb, K, C, T, transposed);
    return check_launch("pack_weight16");
}

// Every bf16 weight pack of the network in ONE launch after the optimizer step (26 launches of 5-60 us per step
// otherwise).  Jobs travel in the kernel argument; a workgroup finds its job by a scalar scan over the block ranges.
// Element mapping and rounding are those of k_pack_weight16: bit-identical outputs.
struct Pack16Job {
    const float *w;
    unsigned short *wf, *wb;
    int K, C, T, transposed;
    unsigned blk_begin;
    int tiled;  // 1: 3x3x3 conv -> one workgroup per 16 k x 16 c tile (through LDS)
};
constexpr int PACK16_MAX_JOBS = 64;
struct Pack16Batch {
    int n;
    Pack16Job j[PACK16_MAX_JOBS];
};

// Tiled jobs (3x3x3 convs, C % 32 == 0 and K % 32 == 0): the workgroup reads its 16 x 16 x 27 block of the torch tensor as
// 16 contiguous 1.7 KB runs into LDS and writes both packed layouts in contiguous 256-byte runs (the element-wise form
// reads with a C*27*4-byte lane stride: 280 us for the 26 weights of the network, 4 x the time of the traffic).
__global__ __launch_bounds__(256) void k_pack16_batch(const Pack16Batch pb) {
    __shared__ float tile[16 * 432];  // [k 16][c 16][t 27]
    int q = 0;
    for (int m = 1; m < pb.n; m++)
        if (blockIdx.x >= pb.j[m].blk_begin) q = m;  // scalar scan (block ranges ascend)
    const Pack16Job &J = pb.j[q];
    const unsigned lb = blockIdx.x - J.blk_begin;
    const int tid = threadIdx.x;
    if (!J.tiled) {
        const long i = (long)lb * 256 + tid;
        const long total = (long)J.K * J.C * J.T;
        if (i >= total) return;
        const int k = (int)(i % J.K);
        const int c = (int)((i / J.K) % J.C);
        const int t = (int)(i / ((long)J.K * J.C));
        const float v = J.transposed ? J.w[((size_t)c * J.K + k) * J.T + t] : J.w[((size_t)k * J.C + c) * J.T + t];
        const unsigned short b = f2bf(v);
        if (J.wf) J.wf[widx16(J.T, J.K, t, c, k)] = b;
        if (J.wb) J.wb[widx16(J.T, J.C, t, k, c)] = b;
        return;
    }
    const int K = J.K, C = J.C;
    const int nct = C >> 4;
    const int k0 = (int)(lb / (unsigned)nct) * 16, c0 = (int)(lb % (unsigned)nct) * 16;
    for (int idx = tid; idx < 16 * 432; idx += 256) {
        const int r = idx / 432, o = idx - r * 432;
        tile[idx] = J.w[((size_t)(k0 + r) * C + c0) * 27 + o];
    }
    __syncthreads();
    // thread = (8-channel half hh, row 0..15, element e 0..7) of the destination: 256 consecutive bytes per half and tap
    const int hh = tid >> 7, row = (tid >> 3) & 15, e = tid & 7;
    if (J.wf) {  // reduce channel c = c0 + 8 hh + e, produce channel k = k0 + row
        unsigned short *dst = J.wf + widx16(27, K, 0, c0 + 8 * hh + e, k0 + row);
        const float *src = tile + row * 432 + (8 * hh + e) * 27;
        for (int t = 0; t < 27; t++) dst[(size_t)t * 4 * K * 8] = f2bf(src[t]);  // widx16: tap stride = 2 * 2 * K * 8
    }
    if (J.wb) {  // reduce channel k = k0 + 8 hh + e, produce channel c = c0 + row
        unsigned short *dst = J.wb + widx16(27, C, 0, k0 + 8 * hh + e, c0 + row);
        const float *src = tile + (8 * hh + e) * 432 + row * 27;
        for (int t = 0; t < 27; t++) dst[(size_t)t * 4 * C * 8] = f2bf(src[t]);
    }
}

int pack_weights16_batch(int n, const float *const *w, unsigned short *const *wf, unsigned short *const *wb, const int *K,
                         const int *C, const int *T, const int *transposed, hipStream_t s) {
    int done = 0;
    while (done < n) {
        Pack16Batch pb;
        memset(&pb, 0, sizeof(pb));
        unsigned blocks = 0;
        int m = 0;
        for (; m < PACK16_MAX_JOBS && done + m < n; m++) {
            const int q = done + m;
            Pack16Job &J = pb.j[m];
            J.w = w[q]; J.wf = wf[q]; J.wb = wb[q];
            J.K = K[q]; J.C = C[q]; J.T = T[q]; J.transposed = transposed[q];
            J.blk_begin = blocks;
            J.tiled = (J.T == 27 && !J.transposed && J.K % 32 == 0 && J.C % 32 == 0) ? 1 : 0;
            blocks += J.tiled ? (unsigned)((J.K / 16) * (J.C / 16)) : (unsigned)cdiv((long)J.K * J.C * J.T, 256);
        }
        pb.n = m;
        if (blocks > 0) hipLaunchKernelGGL(k_pack16_batch, dim3(blocks), dim3(256), 0, s, pb);
        if (check_launch("pack_weights16_batch")) return 1;
        done += m;
    }
    return 0;
}

}  // namespace mvd
