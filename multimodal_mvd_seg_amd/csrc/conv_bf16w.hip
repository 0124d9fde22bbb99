// k_wgrad16z (round 3): z-marching weight gradient of the plain 3x3x3 stride-1 convolution in bf16 -- the largest kernel
// family of the bf16 train step (nnUNetTrainer.py:888-925 backward; the convs of get_network_from_plans.py:41-44 at the
// 128^3 ... 32^3 stages).
//
// Same product as k_wgrad16<7, 10, 4, 1, true> (conv_mfma.hip): dw[tap][c][k] = sum over voxels x[voxel + tap][c] * dy[voxel][k]
// on v_mfma_f32_32x32x16_bf16 (16 voxels per instruction), operands fetched from channel-contiguous LDS rows with the
// transposing read ds_read_b64_tr_b16, a wave = two x-triples + one tap of the ninth triple (wave 3's seventh slot
// multiplies a block of ones: the bias gradient), split-K partials reduced by k_wgrad_reduce_f.  What changes is the data
// movement.  The tiled kernel stages a 6 x 10 x 10 halo for a 4 x 8 x 8 tile (2.34 x the tile's voxels; PMC round 3: 0.83 GB
// fetched for 0.54 GB), decodes tile coordinates, reads a per-step offset table and adds an address per transposing read:
// 6.6 vector instructions per MFMA, bound by the vector issue port (PMC rounds 2 and 3).  Here a workgroup owns a column
// of 8 x 32 output voxels and marches along z:
//   * an input plane (10 x 34 halo slots of 64 B) is loaded ONCE and serves the output planes z-1, z, z+1 from a ring of
//     four LDS images (1.33 x the voxels, the in-plane halo only); dy planes alternate between two images;
//   * the column is fixed, so every in-plane bound is decided once per column: a lane's buffer-load offset is either its
//     element's offset or out of range (the descriptor returns zeros), and planes outside the volume / the chunk are
//     zero-record descriptors -- no bounds logic and no address arithmetic per plane beyond one scalar base;
//   * the 16 steps of a plane are straight-line code: every LDS read is base register + immediate (row pitch 34 slots,
//     a step is half an output row); the three fragments of an x-triple come from FOUR reads (slots +0..3, +4..7 for tap
//     x-1; +2..5, +6..9 for tap x+1: both land in aligned register quads) and four v_alignbit for the odd shift;
//     12 reads + 8 vector instructions per 7 MFMAs;
//   * global loads are issued two per step in the middle of a plane and written to LDS a whole plane later (first version:
//     written ten steps later -- every plane then waited ~1.3 plane times on HBM latency); two barriers per plane (one bare
//     s_barrier before the writes: the images they overwrite were read by the previous plane; one after them).  One workgroup per CU (143 KB of LDS), so the schedule is explicit: operands of step s+1 are fetched before
//     the MFMAs of step s.
// Eligibility (host): 27 taps in raster order, stride 1, pad 1, channels in blocks of 32, W >= 32.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8z __attribute__((ext_vector_type(8)));
typedef unsigned u32x4z __attribute__((ext_vector_type(4)));
typedef unsigned u32x2z __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4w lds_s16x4w;

constexpr int WZ_TH = 8, WZ_TW = 32;
constexpr int WZ_SROW = WZ_TW + 2;                  // halo slots per row (= LDS row pitch in slots)
constexpr int WZ_ROWB = WZ_SROW * 64;               // 2176 B
constexpr int WZ_PLANE = (WZ_TH + 2) * WZ_ROWB;     // 21760 B per x plane image
constexpr int WZ_NRING = 4;
constexpr int WZ_BPLANE = WZ_TH * WZ_TW * 64;       // 16384 B per dy plane image
constexpr int WZ_BOFF = WZ_NRING * WZ_PLANE;        // 87040
constexpr int WZ_ONES = WZ_BOFF + 2 * WZ_BPLANE;    // 119808
// the ones region covers every (base + immediate) the single-tap reads of a plane can form
constexpr int WZ_ONES_BYTES = (WZ_TH - 1) * WZ_ROWB + (16 + 4 + 11 + 1) * 64;
constexpr int WZ_LDS = WZ_ONES + WZ_ONES_BYTES;
constexpr int WZ_APARTS = (WZ_TH + 2) * WZ_SROW * 4;  // 1360 16-byte parts per x plane
static_assert(WZ_LDS <= 160 * 1024, "k_wgrad16z: LDS budget");
static_assert((WZ_TH - 1) * WZ_ROWB + 16 * 64 + 384 < 65536, "ds_read immediates");

struct WgZTile {
    int nty, ntx, nzc, zc;   // column tiles, z chunks per column, planes per chunk
    int nunits, nsplit, nkb;
};

__device__ inline s16x4w wz_trd(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w *)addr);
}

// Staging loads and their waits are inline asm: the compiler's s_waitcnt pass merges the counter states of the prologue and
// of both unrolled planes at the loop head and ends up waiting for vmcnt(0) -- for the loads issued a few steps earlier -- in
// front of the dy writes.  Issue order is fixed (set S: NA x parts, then NB dy parts; two sets in flight), so the exact
// counts are known: x parts of the older set done <=> at most 2 (NA + NB) - NA loads outstanding, all of it <=> NA + NB.
__device__ inline void wz_bload(u32x4z &r, unsigned voff, u32x4z rsrc) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc));
}
__device__ inline u32x4z wz_rsrc(const char *base, bool ok) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    u32x4z r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, ok ? 0x7fffffffu : 0u, 0x00020000u};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]);
    r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]);
    return r;
}
template <int N, int CNT>
__device__ inline void wz_wait(u32x4z (&r)[CNT]) {  // s_waitcnt vmcnt(N); the staged registers are only valid behind it
    static_assert(CNT == 2 || CNT == 3 || CNT == 4 || CNT == 6 || CNT == 10, "staging set sizes");
    if constexpr (CNT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N));
    if constexpr (CNT == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) : "n"(N));
    if constexpr (CNT == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
    if constexpr (CNT == 6)
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(N));
    if constexpr (CNT == 10)
        asm volatile("s_waitcnt vmcnt(%10)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])
                     : "n"(N));
}

// NG = wave groups of the workgroup (4 waves each).  NG = 2: eight waves, two per SIMD -- group g takes the output rows
// 4 g .. 4 g + 3 of every plane (8 of its 16 steps) with its own accumulators, which are added through LDS at the end.
// Ablation (round 3, 32 -> 32 @128^3, kernel time under rocprofv3 with GRBM_GUI_ACTIVE): full kernel 193 us = 340 k cycles per
// XCD at 1.76 GHz; without the staging loads / LDS writes 146 us = 295 k at 2.02 GHz; MFMAs + shifts only 135 us = 291 k at
// 2.16 GHz; everything but the MFMAs 97 us at 2.51 GHz -- MFMA stream plus HBM stream run into the power limit, so the
// parts "add up" in time although they overlap in cycles.  NG = 2 hides the operand reads (plane loop without staging
// 0.142 vs 0.161 ms) but not that.
// PRO: loader prologue (ops.NormActConv3dFn backward): x is the RAW bf16 output of the producing conv; the operand of the
// product is a = bf16(lrelu(fma(x, scale[n][c], shift[n][c]))) -- the arithmetic of k_in_apply_ss16 / the forward loader
// prologue of k_fwd16y, applied to the staged 16-byte parts in registers (a thread's parts are always the channel octet
// tid & 3: eight scale / shift registers), one part per step ten steps ahead of its LDS write; voxels outside the volume
// stay zero (the zero padding applies to the activation).  The activated tensor is never read: it need not exist.
template <int DBG, int NG, bool PRO>
__global__ __launch_bounds__(256 * NG, 1) void k_wgrad16z(const WgradGeom g, const WgZTile tg, const unsigned short *__restrict__ a1,
                                                          const unsigned short *__restrict__ a2,
                                                          const unsigned short *__restrict__ b, float *__restrict__ partial,
                                                          float *__restrict__ pbias, const float *__restrict__ in_scale,
                                                          const float *__restrict__ in_shift, const float slope) {
    static_assert(!PRO || NG == 1, "the loader prologue is scheduled for the four-wave form");
    constexpr int NT = 256 * NG;                     // threads
    constexpr int NA = (WZ_APARTS + NT - 1) / NT;    // staging loads per thread: x plane (6 / 3)
    constexpr int NB = WZ_TH * WZ_TW * 4 / NT;       // dy plane (4 / 2)
    constexpr int SP = 16 / NG;                      // steps per plane and wave
    constexpr int ST_BAR1 = NG == 1 ? 3 : 0, ST_WA = ST_BAR1 + 1, ST_WB = ST_BAR1 + 2, ST_BAR2 = NG == 1 ? 7 : 3;
    constexpr int ST_LD0 = NG == 1 ? 6 : 3;          // one staging load per step from here
    static_assert(ST_LD0 + NA + NB <= SP && ST_WB < ST_LD0 + (NG == 1 ? 1 : 1), "staging schedule");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y / tg.nkb, kb = blockIdx.y % tg.nkb;
    const int split = blockIdx.x;
    const int C = g.C1 + g.C2, K = g.K;
    const int c0 = cb * 32, k0 = kb * 32;
    const unsigned short *asrc;
    int Cs, cofs;
    if (c0 < g.C1) {
        asrc = a1; Cs = g.C1; cofs = c0;
    } else {
        asrc = a2; Cs = g.C2; cofs = c0 - g.C1;
    }
    const int D = g.Do, H = g.Ho, W = g.Wo;

    f32x16w acc[7];
#pragma unroll
    for (int j = 0; j < 7; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    // bf16 ones for wave 3's seventh slot (the bias gradient): written once, never overwritten
    for (int e = tid; e < WZ_ONES_BYTES / 4; e += NT) reinterpret_cast<unsigned *>(lds8 + WZ_ONES)[e] = 0x3f803f80u;

    // transposing-read lane roles (as k_wgrad16): the 16-lane group lane >> 4 takes channels 16 * (group & 1) .. of the voxels
    // of k-half h; lane 4 q + p of the group supplies the address of voxel row q, channel columns 4 p .. 4 p + 3 (8 bytes)
    const int q4 = (lane & 15) >> 2;
    const int colb = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const unsigned lane_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds8 + (unsigned)((8 * h + q4) * 64 + colb);
    // the wave group's first output row
    const unsigned growA = (unsigned)(grp * (WZ_TH / NG) * WZ_ROWB), growB = (unsigned)(grp * (WZ_TH / NG) * WZ_TW * 64);
    // triples 2 w, 2 w + 1 (dz = T / 3, dy = T % 3), single tap 24 + w of the ninth triple (dz = dy = 2, dx = w)
    const int T0 = 2 * wave, T1 = 2 * wave + 1;
    const int dzq0 = T0 / 3, dzq1 = T1 / 3;
    const unsigned bq0 = lane_base + growA + (unsigned)((T0 % 3) * WZ_ROWB);
    const unsigned bq1 = lane_base + growA + (unsigned)((T1 % 3) * WZ_ROWB);
    const bool ones_slot = wave == 3;
    const unsigned bs_ = ones_slot ? lane_base + (unsigned)WZ_ONES : lane_base + growA + (unsigned)(2 * WZ_ROWB + wave * 64);
    const unsigned bB = lane_base + growB + (unsigned)WZ_BOFF;

    // staging roles: x plane part idx = u * NT + tid -> (halo row, halo column, 16-byte part), LDS image linear in idx;
    // dy plane part likewise
    int hrA[NA], hcA[NA];
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int idx = u * NT + tid;
        const int slot = idx >> 2;
        const int hr = slot / WZ_SROW, hc = slot - hr * WZ_SROW;
        hrA[u] = idx < WZ_APARTS ? hr : -100;
        hcA[u] = hc;
    }

    for (int unit = split; unit < tg.nunits; unit += tg.nsplit) {
        unsigned r_ = (unsigned)unit;
        const int zc_ = (int)(r_ % (unsigned)tg.nzc); r_ /= (unsigned)tg.nzc;
        const int tx_ = (int)(r_ % (unsigned)tg.ntx); r_ /= (unsigned)tg.ntx;
        const int ty_ = (int)(r_ % (unsigned)tg.nty);
        const int n = (int)(r_ / (unsigned)tg.nty);
        const int y0 = ty_ * WZ_TH, x0 = tx_ * WZ_TW;
        const int z0 = zc_ * tg.zc;
        const int nz = min(tg.zc, D - z0);
        // per-lane element offsets from the column's corner; out of range where the halo voxel lies outside the plane
        unsigned voA[NA], voB[NB];
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int gy = y0 - 1 + hrA[u], gx = x0 - 1 + hcA[u];
            const bool ok = hrA[u] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            voA[u] = ok ? (unsigned)(((hrA[u] * W + hcA[u]) * Cs) * 2 + ((u * NT + tid) & 3) * 16) : 0x80000000u;
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int idx = u * NT + tid;
            const int slot = idx >> 2, r = slot >> 5, c = slot & 31;
            const bool ok = y0 + r < H && x0 + c < W;
            voB[u] = ok ? (unsigned)(((r * W + c) * K) * 2 + (idx & 3) * 16) : 0x80000000u;
        }
        float psc[PRO ? 8 : 1], psh[PRO ? 8 : 1];
        if (PRO) {
            const float *sp = in_scale + (size_t)n * g.C1 + c0 + (tid & 3) * 8, *tp = in_shift + (size_t)n * g.C1 + c0 + (tid & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                psc[e] = sp[e];
                psh[e] = tp[e];
            }
        }
        auto xform = [&](u32x4z &q, const bool ok) {  // (same expression as k_fwd16y's prologue: bit-identical activations)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float lo = __uint_as_float(q[e] << 16), hi = __uint_as_float(q[e] & 0xffff0000u);
                lo = __builtin_fmaf(lo, psc[2 * e], psh[2 * e]);
                hi = __builtin_fmaf(hi, psc[2 * e + 1], psh[2 * e + 1]);
                lo = fmaxf(lo, lo * slope);
                hi = fmaxf(hi, hi * slope);
                f32x2w v2 = {lo, hi};
                bf16x2w r2 = __builtin_convertvector(v2, bf16x2w);
                q[e] = ok ? *reinterpret_cast<unsigned *>(&r2) : 0u;
            }
        };
        auto plane_ok = [&](int z) { return z >= 0 && z < D; };
        // corner addresses of plane 0 of this sample (the corner itself may lie outside the volume: only valid lanes load)
        const long planeA = (long)H * W * Cs * 2, planeB = (long)H * W * K * 2;
        const char *cornerA = reinterpret_cast<const char *>(asrc) + ((long)n * D * planeA + ((long)(y0 - 1) * W + (x0 - 1)) * Cs * 2 + cofs * 2);
        const char *cornerB = reinterpret_cast<const char *>(b) + ((long)n * D * planeB + ((long)y0 * W + x0) * K * 2 + k0 * 2);
        auto rsrcA = [&](int z) {  // x plane z; planes outside the volume are zero-record descriptors (zeros, no traffic)
            const bool ok = z >= 0 && z < D;
            return wz_rsrc(cornerA + (ok ? (long)z * planeA : 0), ok);
        };
        auto rsrcB = [&](int z, bool want) {
            const bool ok = want && z >= 0 && z < D;
            return wz_rsrc(cornerB + (ok ? (long)z * planeB : 0), ok);
        };
        // staged planes: set p & 1 is written to LDS by plane p and re-loaded by it for plane p + 2 (two planes in flight)
        u32x4z ra[2][NA], rb[2][NB];
        auto write_A = [&](int slot, const u32x4z (&r)[NA]) {
#pragma unroll
            for (int u = 0; u < NA; u++)
                if (u < NA - 1 || tid < WZ_APARTS - (NA - 1) * NT)
                    *reinterpret_cast<u32x4z *>(lds8 + slot * WZ_PLANE + (u * NT + tid) * 16) = r[u];
        };
        auto write_B = [&](int buf, const u32x4z (&r)[NB]) {
#pragma unroll
            for (int u = 0; u < NB; u++) *reinterpret_cast<u32x4z *>(lds8 + WZ_BOFF + buf * WZ_BPLANE + (u * NT + tid) * 16) = r[u];
        };
        auto load_A = [&](u32x4z (&r)[NA], const u32x4z d) {
#pragma unroll
            for (int u = 0; u < NA; u++) wz_bload(r[u], voA[u], d);
        };
        auto load_B = [&](u32x4z (&r)[NB], const u32x4z d) {
#pragma unroll
            for (int u = 0; u < NB; u++) wz_bload(r[u], voB[u], d);
        };
        // ---- prologue: x planes z0 - 1, z0, z0 + 1 -> ring slots 0, 1, 2; dy plane z0 -> image 0 (all loads in flight
        // together); sets 0 / 1 = x planes z0 + 2 / z0 + 3 and dy planes z0 + 1 / z0 + 2, written by planes 0 / 1
        __syncthreads();  // (the previous column's reads are done; no staging load is outstanding)
        {
            u32x4z r0[NA];
            load_A(r0, rsrcA(z0 - 1));
            load_A(ra[0], rsrcA(z0));
            load_A(ra[1], rsrcA(z0 + 1));
            load_B(rb[0], rsrcB(z0, true));
            wz_wait<2 * NA + NB>(r0);
            if (PRO) {
#pragma unroll
                for (int u = 0; u < NA; u++) xform(r0[u], voA[u] != 0x80000000u && plane_ok(z0 - 1));
            }
            write_A(0, r0);
            wz_wait<NA + NB>(ra[0]);
            if (PRO) {
#pragma unroll
                for (int u = 0; u < NA; u++) xform(ra[0][u], voA[u] != 0x80000000u && plane_ok(z0));
            }
            write_A(1, ra[0]);
            wz_wait<0>(ra[1]);
            wz_wait<0>(rb[0]);
            if (PRO) {
#pragma unroll
                for (int u = 0; u < NA; u++) xform(ra[1][u], voA[u] != 0x80000000u && plane_ok(z0 + 1));
            }
            write_A(2, ra[1]);
            write_B(0, rb[0]);
            load_A(ra[0], rsrcA(2 <= nz ? z0 + 2 : -1));
            load_B(rb[0], rsrcB(z0 + 1, 1 < nz));
            load_A(ra[1], rsrcA(3 <= nz ? z0 + 3 : -1));
            load_B(rb[1], rsrcB(z0 + 2, 2 < nz));
            if (PRO) {  // set 0 is written by plane 0 before any step could transform it (set 1: steps 10 .. 15 of plane 0)
                wz_wait<NA + NB>(ra[0]);
#pragma unroll
                for (int u = 0; u < NA; u++) xform(ra[0][u], voA[u] != 0x80000000u && 2 <= nz && plane_ok(z0 + 2));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // operands: [buf][triple][read], single tap [buf][read], dy [buf][read]
        s16x4w rt[2][2][4], rs[2][2], rv[2][2];
        // (read order = the order the next step's MFMAs need them: dy and triple 0 first, the single tap last -- the first
        // MFMA of a step waits for four reads issued seven gaps earlier, not for the two issued one gap earlier)
        auto fetch = [&](unsigned aq0, unsigned aq1, unsigned as, unsigned ab, int imm, int immB, int buf) {
            rv[buf][0] = wz_trd(ab + immB);
            rv[buf][1] = wz_trd(ab + immB + 256);
            rt[buf][0][0] = wz_trd(aq0 + imm);
            rt[buf][0][1] = wz_trd(aq0 + imm + 256);
            rt[buf][0][2] = wz_trd(aq0 + imm + 128);
            rt[buf][0][3] = wz_trd(aq0 + imm + 384);
            rt[buf][1][0] = wz_trd(aq1 + imm);
            rt[buf][1][1] = wz_trd(aq1 + imm + 256);
            rt[buf][1][2] = wz_trd(aq1 + imm + 128);
            rt[buf][1][3] = wz_trd(aq1 + imm + 384);
            rs[buf][0] = wz_trd(as + imm);
            rs[buf][1] = wz_trd(as + imm + 256);
        };
        // MFMA order of a step: the four fragments that come straight from the reads first, the single tap, then the two
        // odd-shift fragments -- their eight v_alignbit fill the gaps of the first MFMAs instead of standing in front of them
        auto mfmas = [&](int buf) {
            const bf16x8z bfrag = __builtin_bit_cast(bf16x8z, __builtin_shufflevector(rv[buf][0], rv[buf][1], 0, 1, 2, 3, 4, 5, 6, 7));
            s16x8w f0[2], f1[2], f2[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                f0[q] = __builtin_shufflevector(rt[buf][q][0], rt[buf][q][1], 0, 1, 2, 3, 4, 5, 6, 7);   // voxels x-1 .. x+6
                f2[q] = __builtin_shufflevector(rt[buf][q][2], rt[buf][q][3], 0, 1, 2, 3, 4, 5, 6, 7);   // x+1 .. x+8
                f1[q] = __builtin_shufflevector(f0[q], f2[q], 1, 2, 3, 4, 5, 6, 7, 14);                  // x .. x+7
            }
            const bf16x8z sfrag = __builtin_bit_cast(bf16x8z, __builtin_shufflevector(rs[buf][0], rs[buf][1], 0, 1, 2, 3, 4, 5, 6, 7));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f0[0]), bfrag, acc[0], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f2[0]), bfrag, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f0[1]), bfrag, acc[3], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f2[1]), bfrag, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag, bfrag, acc[6], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f1[0]), bfrag, acc[1], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f1[1]), bfrag, acc[4], 0, 0, 0);
        };
        auto ring_of = [&](int p, int dz) { return (unsigned)(((p + dz) & 3) * WZ_PLANE); };
        unsigned aq0 = bq0 + ring_of(0, dzq0), aq1 = bq1 + ring_of(0, dzq1);
        unsigned as = ones_slot ? bs_ : bs_ + ring_of(0, 2);
        unsigned ab = bB;
        fetch(aq0, aq1, as, ab, 0, 0, 0);
        auto plane = [&](const int p, auto SETC) {
            constexpr int SET = decltype(SETC)::value;
            const int z = z0 + p;
            // Registers ra / rb[SET] hold x plane z + 2 and dy plane z + 1 (loaded two planes ago): written behind the first
            // barrier of this plane (the images they replace -- x plane z - 2, dy plane z - 1 -- were read by plane p - 1),
            // then re-loaded with x plane z + 4 (the halo plane of the chunk's last output plane included) and dy plane z + 3.
            const u32x4z rA = rsrcA(p + 4 <= nz ? z + 4 : -1);
            const u32x4z rB = rsrcB(z + 3, p + 3 < nz);
            const unsigned naq0 = bq0 + ring_of(p + 1, dzq0), naq1 = bq1 + ring_of(p + 1, dzq1);
            const unsigned nas = ones_slot ? bs_ : bs_ + ring_of(p + 1, 2);
            const unsigned nab = bB + (unsigned)(((p + 1) & 1) * WZ_BPLANE);
#pragma unroll
            for (int st = 0; st < SP; st++) {
                if (DBG & 1) {
                } else if (st < SP - 1) {
                    const int imm = ((st + 1) >> 1) * WZ_ROWB + ((st + 1) & 1) * 16 * 64;
                    const int immB = (((st + 1) >> 1) * WZ_TW + ((st + 1) & 1) * 16) * 64;
                    fetch(aq0, aq1, as, ab, imm, immB, (st + 1) & 1);
                } else {
                    fetch(naq0, naq1, nas, nab, 0, 0, 0);  // step 0 of the next plane (behind this plane's second barrier)
                }
                if (DBG & (2 | 32)) {
                } else if (st >= ST_LD0 && st < ST_LD0 + NA) wz_bload(ra[SET][st - ST_LD0], voA[st - ST_LD0], rA);
                else if (st >= ST_LD0 + NA && st < ST_LD0 + NA + NB) wz_bload(rb[SET][st - ST_LD0 - NA], voB[st - ST_LD0 - NA], rB);
                constexpr int ST_X0 = ST_LD0 + NB;  // PRO: part u of the OTHER set (x plane z + 3, loaded one plane ago) at step
                if (PRO && st >= ST_X0 && st < ST_X0 + NA) {  // ST_X0 + u: every younger load is one of this plane's
                    // the other set is complete once no more loads are outstanding than this plane has issued so far
                    constexpr int Y0 = ST_X0 - ST_LD0 + 1;
                    const int u_ = st - ST_X0;
                    if (u_ == 0) wz_wait<Y0>(ra[1 - SET]);
                    else if (u_ == 1) wz_wait<Y0 + 1>(ra[1 - SET]);
                    else if (u_ == 2) wz_wait<Y0 + 2>(ra[1 - SET]);
                    else if (u_ == 3) wz_wait<Y0 + 3>(ra[1 - SET]);
                    else if (u_ == 4) wz_wait<Y0 + 4>(ra[1 - SET]);
                    else wz_wait<Y0 + 5>(ra[1 - SET]);
                    xform(ra[1 - SET][st - ST_X0], voA[st - ST_X0] != 0x80000000u && p + 3 <= nz && plane_ok(z + 3));
                }
                if (!(DBG & 8)) mfmas(st & 1);
                if (st == ST_WA && !(DBG & (2 | 64))) {
                    wz_wait<2 * (NA + NB) - NA>(ra[SET]);  // (this set's loads are the older half of those outstanding)
                    write_A((p + 3) & 3, ra[SET]);
                }
                if (st == ST_WB && !(DBG & (2 | 64))) {
                    wz_wait<NA + NB>(rb[SET]);
                    write_B((p + 1) & 1, rb[SET]);
                }
                // the step as one pipeline: per MFMA gap two transposing reads of the next step's operands and two of this
                // step's v_alignbit (NG = 1: what is not placed inside a gap is paid in full)
                if (!(DBG & 16)) {
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (j < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        if (j >= 1 && j < 5) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        if (PRO && st >= ST_X0 && st < ST_X0 + NA) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        if ((st == ST_WA && j < NA) || (st == ST_WB && j < NB)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st == ST_BAR1 && !(DBG & 4)) asm volatile("s_barrier" ::: "memory");  // every wave has finished the previous plane's reads
                if (st == ST_BAR2 && !(DBG & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            aq0 = naq0; aq1 = naq1; as = nas; ab = nab;
        };
        for (int rep = 0; rep < ((DBG & 128) ? 2 : 1); rep++)  // (timing ablation: the plane loop twice = fixed cost + 2 x loop)
        for (int p = 0; p < nz; p += 2) {
            plane(p, std::integral_constant<int, 0>());
            if (p + 1 < nz) plane(p + 1, std::integral_constant<int, 1>());
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last planes' zero-record loads)
    }
    if constexpr (NG == 2) {  // group 1's sums join group 0's through LDS ([register][thread]: conflict-free)
        __syncthreads();
        float *lf = reinterpret_cast<float *>(lds8);
        static_assert(7 * 16 * 256 * 4 <= WZ_LDS, "accumulator exchange fits the LDS allocation");
        if (grp == 1) {
#pragma unroll
            for (int j = 0; j < 7; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) lf[(j * 16 + r) * 256 + (tid & 255)] = acc[j][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int j = 0; j < 7; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][r] += lf[(j * 16 + r) * 256 + tid];
    }
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const int t = j < 6 ? 6 * wave + j : 24 + wave;
        if (t < 27) {
            float *po = partial + ((size_t)split * 27 + t) * C * K;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                po[(size_t)(c0 + row) * K + k0 + i] = acc[j][r];
            }
        }
    }
    // every row of the ones-slot accumulator holds the same column sums of dy: row 0 of the c-block-0 workgroups goes out
    if (ones_slot && pbias != nullptr && cb == 0 && h == 0) pbias[(size_t)split * K + k0 + i] = acc[6][0];
}

// ------------------------------------------------------------------------------------------------ stride 2
// k_wgrad16zs (round 3): the same z-marching weight gradient for 3x3x3 STRIDE-2 convs (first conv of encoder stages 1, 2):
// dw[tap][c][k] = sum over OUTPUT voxels o of x[2 o + tap - 1][c] dy[o][k].  The tiled kernel (k_wgrad16<7,7,2,1,false>) needed
// 0.198 ms for 32 -> 64 at 128^3 -> 64^3, whose traffic is worth 0.08 ms.  Differences to k_wgrad16z:
//   * column = 4 x 32 OUTPUT voxels; an output plane o needs the input planes 2 o - 1, 2 o, 2 o + 1 (9 rows x 65 columns each):
//     a ring of exactly three LDS images -- plane 2 o + 1 stays for o + 1, the other two are replaced IN PLACE behind the
//     plane's first barrier by the two planes that travelled in registers while plane o was multiplied;
//   * the x image is split by the parity of the input column, [input row 9][parity 2][34 slots][64 B] (as k_fwd16ys): tap
//     dx reads parity dx & 1 at unit stride; dx = 0 and dx = 2 of a (dz, dy) pair share three reads of the even line (the
//     odd shift is four v_alignbit), dx = 1 takes two of the odd line: 5 reads + 4 shifts per triple;
//   * a workgroup multiplies one 32-channel block of x with TWO 32-channel blocks of dy (14 accumulators per wave): x is
//     read once for K = 64; no ones slot -- the bias gradient of these layers takes the column-sum pass.
constexpr int WS_TH = 4, WS_TW = 32, WS_ROWS = 2 * WS_TH + 1, WS_COLS = 2 * WS_TW + 1;
constexpr int WS_LP = WS_TW + 2;                          // slots per (input row, parity) line: 33 used
constexpr int WS_LPB = WS_LP * 64;                        // 2176 B
constexpr int WS_PLANE = WS_ROWS * 2 * WS_LPB;            // 39168 B per x plane image
constexpr int WS_DYB = WS_TH * WS_TW * 64;                // 8192 B per (dy plane, 32-channel block)
constexpr int WS_BOFF = 3 * WS_PLANE;                     // 117504
constexpr int WS_LDS = WS_BOFF + 2 * WS_DYB;              // one dy plane, two channel blocks
constexpr int WS_APARTS = WS_ROWS * WS_COLS * 4;          // 2340 16-byte parts per x plane
constexpr int WS_NA = (WS_APARTS + 255) / 256;            // 10
constexpr int WS_NB = 2 * WS_TH * WS_TW * 4 / 256;        // 4
static_assert(WS_LDS <= 160 * 1024, "k_wgrad16zs: LDS budget");
static_assert((WS_ROWS * 2) * WS_LPB + 16 * 64 + 512 < 65536, "ds_read immediates");

__global__ __launch_bounds__(256, 1) void k_wgrad16zs(const WgradGeom g, const WgZTile tg, const unsigned short *__restrict__ a1,
                                                      const unsigned short *__restrict__ b, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y / tg.nkb, kb2 = blockIdx.y % tg.nkb;   // 32 reduce channels x 64 produce channels
    const int split = blockIdx.x;
    const int C = g.C1, K = g.K;
    const int c0 = cb * 32, k0 = kb2 * 64;
    const int Do = g.Do, Ho = g.Ho, Wo = g.Wo, Di = g.Di, Hi = g.Hi, Wi = g.Wi;

    f32x16w acc[2][7];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int j = 0; j < 7; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[q][j][r] = 0.f;

    const int q4 = (lane & 15) >> 2;
    const int colb = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const unsigned lane_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds8 + (unsigned)((8 * h + q4) * 64 + colb);
    // triples 2 w, 2 w + 1 over (dz, dy) = (T / 3, T % 3); single tap dx = w of the ninth pair (dz = dy = 2), waves 0 .. 2
    const int T0 = 2 * wave, T1 = 2 * wave + 1;
    const int dzq0 = T0 / 3, dzq1 = T1 / 3;
    const unsigned bq0 = lane_base + (unsigned)((T0 % 3) * 2 * WS_LPB);   // input row 2 r + dy: dy rows of two lines each
    const unsigned bq1 = lane_base + (unsigned)((T1 % 3) * 2 * WS_LPB);
    const bool has_single = wave < 3;
    const unsigned bs_ = lane_base + (unsigned)(2 * 2 * WS_LPB + (wave & 1) * WS_LPB + (wave >> 1) * 64);
    const unsigned bB = lane_base + (unsigned)WS_BOFF;

    // staging roles: x plane part idx = u * 256 + tid -> (input row, input column, part); LDS address by column parity
    int rrA[WS_NA], ccA[WS_NA];
    unsigned ldsA[WS_NA];
#pragma unroll
    for (int u = 0; u < WS_NA; u++) {
        const int idx = u * 256 + tid;
        const int slot = idx >> 2, part = idx & 3;
        const int R = slot / WS_COLS, X = slot - R * WS_COLS;
        rrA[u] = idx < WS_APARTS ? R : -100;
        ccA[u] = X;
        ldsA[u] = (unsigned)((R * 2 + (X & 1)) * WS_LPB + (X >> 1) * 64 + part * 16);
    }

    for (int unit = split; unit < tg.nunits; unit += tg.nsplit) {
        unsigned r_ = (unsigned)unit;
        const int zc_ = (int)(r_ % (unsigned)tg.nzc); r_ /= (unsigned)tg.nzc;
        const int tx_ = (int)(r_ % (unsigned)tg.ntx); r_ /= (unsigned)tg.ntx;
        const int ty_ = (int)(r_ % (unsigned)tg.nty);
        const int n = (int)(r_ / (unsigned)tg.nty);
        const int y0 = ty_ * WS_TH, x0 = tx_ * WS_TW;     // output coordinates
        const int z0 = zc_ * tg.zc;
        const int nz = min(tg.zc, Do - z0);
        unsigned voA[WS_NA], voB[WS_NB];
#pragma unroll
        for (int u = 0; u < WS_NA; u++) {
            const int gy = 2 * y0 - 1 + rrA[u], gx = 2 * x0 - 1 + ccA[u];
            const bool ok = rrA[u] >= 0 && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
            voA[u] = ok ? (unsigned)(((rrA[u] * Wi + ccA[u]) * C) * 2 + ((u * 256 + tid) & 3) * 16) : 0x80000000u;
        }
#pragma unroll
        for (int u = 0; u < WS_NB; u++) {   // dy plane: idx -> (channel block q, voxel, part)
            const int idx = u * 256 + tid;
            const int qb = idx >> 9, v = (idx >> 2) & 127, r = v >> 5, c = v & 31;
            const bool ok = y0 + r < Ho && x0 + c < Wo;
            voB[u] = ok ? (unsigned)(((r * Wo + c) * K + qb * 32) * 2 + (idx & 3) * 16) : 0x80000000u;
        }
        const long planeA = (long)Hi * Wi * C * 2, planeB = (long)Ho * Wo * K * 2;
        const char *cornerA = reinterpret_cast<const char *>(a1) + ((long)n * Di * planeA + ((long)(2 * y0 - 1) * Wi + (2 * x0 - 1)) * C * 2 + c0 * 2);
        const char *cornerB = reinterpret_cast<const char *>(b) + ((long)n * Do * planeB + ((long)y0 * Wo + x0) * K * 2 + k0 * 2);
        auto rsrcA = [&](int z, bool want) {  // input plane z
            const bool ok = want && z >= 0 && z < Di;
            return wz_rsrc(cornerA + (ok ? (long)z * planeA : 0), ok);
        };
        auto rsrcB = [&](int z, bool want) {  // output (dy) plane z
            const bool ok = want && z >= 0 && z < Do;
            return wz_rsrc(cornerB + (ok ? (long)z * planeB : 0), ok);
        };
        u32x4z ra[2][WS_NA], rb[WS_NB];
        auto write_A = [&](int slot, const u32x4z (&r)[WS_NA]) {
#pragma unroll
            for (int u = 0; u < WS_NA; u++)
                if (u < WS_NA - 1 || tid < WS_APARTS - (WS_NA - 1) * 256)
                    *reinterpret_cast<u32x4z *>(lds8 + slot * WS_PLANE + ldsA[u]) = r[u];
        };
        auto write_B = [&]() {
#pragma unroll
            for (int u = 0; u < WS_NB; u++) *reinterpret_cast<u32x4z *>(lds8 + WS_BOFF + (u * 256 + tid) * 16) = rb[u];
        };
        auto load_A = [&](u32x4z (&r)[WS_NA], const u32x4z d) {
#pragma unroll
            for (int u = 0; u < WS_NA; u++) wz_bload(r[u], voA[u], d);
        };
        auto load_B = [&](const u32x4z d) {
#pragma unroll
            for (int u = 0; u < WS_NB; u++) wz_bload(rb[u], voB[u], d);
        };
        auto slot_of = [&](int zin) { return ((zin % 3) + 3) % 3; };   // ring slot of input plane zin (any sign)
        // ---- prologue: input plane 2 z0 - 1 into its slot; planes 2 z0, 2 z0 + 1 and dy plane z0 into the registers
        __syncthreads();
        load_A(ra[0], rsrcA(2 * z0 - 1, true));
        wz_wait<0>(ra[0]);
        write_A(slot_of(2 * z0 - 1), ra[0]);
        load_A(ra[0], rsrcA(2 * z0, true));
        load_A(ra[1], rsrcA(2 * z0 + 1, true));
        load_B(rsrcB(z0, true));

        s16x4w rt[2][2][5], rs[2][2], rv[2][2][2];   // [buf][triple][read], single [buf][read], dy [buf][block][read]
        for (int p = 0; p < nz; p++) {
            const int o = z0 + p;
            asm volatile("s_barrier" ::: "memory");   // every wave has finished the previous plane's reads
            wz_wait<WS_NA + WS_NB>(ra[0]);
            write_A(slot_of(2 * o), ra[0]);
            wz_wait<WS_NB>(ra[1]);
            write_A(slot_of(2 * o + 1), ra[1]);
            wz_wait<0>(rb);
            write_B();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the next output plane's inputs travel while this one is multiplied
            load_A(ra[0], rsrcA(2 * o + 2, p + 1 < nz));
            load_A(ra[1], rsrcA(2 * o + 3, p + 1 < nz));
            load_B(rsrcB(o + 1, p + 1 < nz));
            const unsigned aq0 = bq0 + (unsigned)(slot_of(2 * o - 1 + dzq0) * WS_PLANE);
            const unsigned aq1 = bq1 + (unsigned)(slot_of(2 * o - 1 + dzq1) * WS_PLANE);
            const unsigned as = bs_ + (unsigned)(slot_of(2 * o + 1) * WS_PLANE);
            auto fetch = [&](int st, int buf) {
                const int r = st >> 1, sx = st & 1;
                const int imm = (2 * r) * 2 * WS_LPB + sx * 16 * 64;         // input row 2 r (+ dy in the base), even line
                const int immB = (r * WS_TW + sx * 16) * 64;
                rv[buf][0][0] = wz_trd(bB + immB);
                rv[buf][0][1] = wz_trd(bB + immB + 256);
                rv[buf][1][0] = wz_trd(bB + immB + WS_DYB);
                rv[buf][1][1] = wz_trd(bB + immB + WS_DYB + 256);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const unsigned a = q ? aq1 : aq0;
                    rt[buf][q][0] = wz_trd(a + imm);                  // even line: columns c .. c + 3   (dx = 0: elements 0 .. 7)
                    rt[buf][q][1] = wz_trd(a + imm + 256);            //            c + 4 .. c + 7
                    rt[buf][q][2] = wz_trd(a + imm + 512);            //            c + 8 .. c + 11     (dx = 2: elements 1 .. 8)
                    rt[buf][q][3] = wz_trd(a + imm + WS_LPB);         // odd line (dx = 1)
                    rt[buf][q][4] = wz_trd(a + imm + WS_LPB + 256);
                }
                if (has_single) {   // wave-uniform
                    rs[buf][0] = wz_trd(as + imm);
                    rs[buf][1] = wz_trd(as + imm + 256);
                }
            };
            auto mfmas = [&](int buf) {
                bf16x8z bf[2];
#pragma unroll
                for (int q = 0; q < 2; q++)
                    bf[q] = __builtin_bit_cast(bf16x8z, __builtin_shufflevector(rv[buf][q][0], rv[buf][q][1], 0, 1, 2, 3, 4, 5, 6, 7));
                s16x8w f0[2], f1[2], f2[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    f0[q] = __builtin_shufflevector(rt[buf][q][0], rt[buf][q][1], 0, 1, 2, 3, 4, 5, 6, 7);   // dx = 0
                    f1[q] = __builtin_shufflevector(rt[buf][q][3], rt[buf][q][4], 0, 1, 2, 3, 4, 5, 6, 7);   // dx = 1
                    const s16x8w t12 = __builtin_shufflevector(rt[buf][q][1], rt[buf][q][2], 0, 1, 2, 3, 4, 5, 6, 7);
                    f2[q] = __builtin_shufflevector(f0[q], t12, 1, 2, 3, 4, 5, 6, 7, 12);                     // dx = 2
                }
#pragma unroll
                for (int kq = 0; kq < 2; kq++) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        acc[kq][3 * q + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f0[q]), bf[kq], acc[kq][3 * q + 0], 0, 0, 0);
                        acc[kq][3 * q + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f1[q]), bf[kq], acc[kq][3 * q + 1], 0, 0, 0);
                    }
                }
                if (has_single) {
                    const bf16x8z sfrag = __builtin_bit_cast(bf16x8z, __builtin_shufflevector(rs[buf][0], rs[buf][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int kq = 0; kq < 2; kq++)
                        acc[kq][6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag, bf[kq], acc[kq][6], 0, 0, 0);
                }
#pragma unroll
                for (int kq = 0; kq < 2; kq++)
#pragma unroll
                    for (int q = 0; q < 2; q++)
                        acc[kq][3 * q + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8z, f2[q]), bf[kq], acc[kq][3 * q + 2], 0, 0, 0);
            };
            fetch(0, 0);
#pragma unroll
            for (int st = 0; st < 2 * WS_TH; st++) {
                if (st + 1 < 2 * WS_TH) fetch(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(st & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last plane's zero-record loads)
    }
#pragma unroll
    for (int kq = 0; kq < 2; kq++)
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int t = j < 6 ? 6 * wave + j : 24 + wave;
            if (t < 27) {
                float *po = partial + ((size_t)split * 27 + t) * C * K;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    po[(size_t)(c0 + row) * K + k0 + 32 * kq + i] = acc[kq][j][r];
                }
            }
        }
}

static int &wgrad16z_mode() {
    static int v = getenv("MVD_WGRAD16Z") ? atoi(getenv("MVD_WGRAD16Z")) : 1;
    return v;
}
void wgrad16z_enable(int on) { wgrad16z_mode() = on; }

// Returns -1 when the problem is not this kernel's (the caller falls through to k_wgrad16), 0 after a launch (the partials
// [nsplit][27][C][K] and, when pbias_out is set, the bias rows [nsplit][K] are in ws), > 0 on error.
static bool wgrad16z_shape_ok(const WgradGeom &g) {
    const int C = g.C1 + g.C2;
    if (g.ntaps != 27 || g.T != 27 || g.C1 % 32 != 0 || g.C2 % 32 != 0 || g.K % 32 != 0 || C < 32) return false;
    if (g.Di != g.Do || g.Hi != g.Ho || g.Wi != g.Wo || g.Db != g.Do || g.Hb != g.Ho || g.Wb != g.Wo) return false;
    for (int a = 0; a < 3; a++)
        if (g.sa[a] != 1 || g.sb[a] != 1) return false;
    for (int t = 0; t < 27; t++)
        if (g.off[t][0] != t / 9 - 1 || g.off[t][1] != (t / 3) % 3 - 1 || g.off[t][2] != t % 3 - 1 || g.ob[t][0] != 0 ||
            g.ob[t][1] != 0 || g.ob[t][2] != 0)
            return false;
    if (g.Wo < 32 || g.Ho < 8 || g.Do < 8) return false;
    // 32-bit lane offsets inside a plane, 64-bit scalar plane bases
    if ((long)g.Ho * g.Wo * (g.C1 > g.C2 ? g.C1 : g.C2) * 2 >= (1L << 31) || (long)g.Ho * g.Wo * g.K * 2 >= (1L << 31)) return false;
    return true;
}

bool wgrad16z_prologue_ok(const WgradGeom &g) { return wgrad16z_shape_ok(g) && g.C2 == 0; }

int wgrad16z(const WgradGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *b, void *ws,
             size_t ws_bytes, bool want_bias, int *nsplit_out, float **pbias_out, hipStream_t s, const float *in_scale,
             const float *in_shift, float slope) {
    const bool pro = in_scale != nullptr && in_shift != nullptr;
    if (!wgrad16z_mode() && !pro) return -1;
    if (!wgrad16z_shape_ok(g) || (pro && g.C2 != 0)) return -1;
    const int C = g.C1 + g.C2;
    WgZTile tg;
    tg.nty = (g.Ho + WZ_TH - 1) / WZ_TH;
    tg.ntx = (g.Wo + WZ_TW - 1) / WZ_TW;
    const int ncb = C / 32;
    tg.nkb = g.K / 32;
    const long blocks = (long)ncb * tg.nkb;
    if (blocks > 65535) return -1;
    const long columns = (long)g.N * tg.nty * tg.ntx;
    const long per_round = blocks >= 256 ? 1 : 256 / blocks;
    // z chunks per column: the count that minimises rounds x (planes per chunk + the pipeline fill of a chunk)
    long best_cost = -1;
    int best_nzc = 1;
    for (int nzc = 1; nzc <= 32 && nzc * 8 <= g.Do; nzc++) {
        const int zc = (g.Do + nzc - 1) / nzc;
        const int nzc_eff = (g.Do + zc - 1) / zc;
        const long units = columns * nzc_eff;
        const long rounds = (units + per_round - 1) / per_round;
        const long cost = rounds * (zc + 4);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_nzc = nzc_eff;
        }
    }
    tg.zc = (g.Do + best_nzc - 1) / best_nzc;
    tg.nzc = (g.Do + tg.zc - 1) / tg.zc;
    const long units = columns * tg.nzc;
    if (units > (1L << 30)) return -1;
    tg.nunits = (int)units;
    const long rounds = (units + per_round - 1) / per_round;
    tg.nsplit = (int)((units + rounds - 1) / rounds);
    const size_t need = (size_t)tg.nsplit * 27 * C * g.K * sizeof(float);
    const size_t need_b = want_bias ? (size_t)tg.nsplit * g.K * sizeof(float) : 0;
    if (need + need_b > ws_bytes) return -1;
    float *partial = reinterpret_cast<float *>(ws);
    float *pbias = want_bias ? partial + need / sizeof(float) : nullptr;
    typedef void (*kfn_t)(const WgradGeom, const WgZTile, const unsigned short *, const unsigned short *, const unsigned short *,
                          float *, float *, const float *, const float *, float);
    // wave groups: 1 (default) = four waves, 2 = eight.  Measured equal inside the step (11.27-11.41 vs 11.39-11.48 ms): with
    // the staging traffic on, the kernel runs against the power limit (1.76 GHz; 2.16 GHz for the MFMA stream alone), where
    // the cycles the second wave per SIMD saves (340 k -> 300 k per XCD without staging) do not turn into time
    static const int ng_env = getenv("MVD_WGRAD16Z_NG") ? (atoi(getenv("MVD_WGRAD16Z_NG")) == 2 ? 2 : 1) : 1;
    const int ng = pro ? 1 : ng_env;
    kfn_t kfn = pro ? k_wgrad16z<0, 1, true> : ng == 2 ? k_wgrad16z<0, 2, false> : k_wgrad16z<0, 1, false>;
#ifdef MVD_WG16Z_ABLATE
    static const int dbg = getenv("MVD_WG16Z_DBG") ? atoi(getenv("MVD_WG16Z_DBG")) : 0;  // timing ablation only: results are wrong
#define WZ_DBG(V) if (dbg == V && !pro) kfn = ng == 2 ? k_wgrad16z<V, 2, false> : k_wgrad16z<V, 1, false>;
    WZ_DBG(1) WZ_DBG(2) WZ_DBG(3) WZ_DBG(4) WZ_DBG(8) WZ_DBG(16) WZ_DBG(32) WZ_DBG(64) WZ_DBG(128) WZ_DBG(129) WZ_DBG(130) WZ_DBG(131) WZ_DBG(160) WZ_DBG(192) WZ_DBG(136) WZ_DBG(132)
#undef WZ_DBG
#endif
    static PerDeviceFlag cfgd, cfgd_pro;
    PerDeviceFlag &cf = pro ? cfgd_pro : cfgd;
    if (!cf()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, WZ_LDS) != hipSuccess) {
            set_error("conv wgrad (bf16 z-marching): cannot raise the dynamic LDS limit");
            return 1;
        }
        cf() = true;
    }
    hipLaunchKernelGGL(kfn, dim3(tg.nsplit, (unsigned)blocks), dim3(256 * ng), WZ_LDS, s, g, tg, a1, a2, b, partial, pbias, in_scale, in_shift,
                       slope);
    if (check_launch("conv wgrad (bf16 z-marching)")) return 1;
    *nsplit_out = tg.nsplit;
    *pbias_out = pbias;
    return 0;
}


// stride-2 twin: -1 when the problem is not this kernel's, 0 after a launch (partials [nsplit][27][C][K] in ws; no bias rows)
int wgrad16zs(const WgradGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *b, void *ws,
              size_t ws_bytes, int *nsplit_out, hipStream_t s) {
    static const int on = getenv("MVD_WGRAD16ZS") ? atoi(getenv("MVD_WGRAD16ZS")) : 1;
    if (!on || !wgrad16z_mode()) return -1;
    if (g.ntaps != 27 || g.T != 27 || g.C1 % 32 != 0 || g.C2 != 0 || a2 != nullptr || g.K % 64 != 0) return -1;
    for (int a = 0; a < 3; a++)
        if (g.sa[a] != 2 || g.sb[a] != 1) return -1;
    for (int t = 0; t < 27; t++)
        if (g.off[t][0] != t / 9 - 1 || g.off[t][1] != (t / 3) % 3 - 1 || g.off[t][2] != t % 3 - 1 || g.ob[t][0] != 0 ||
            g.ob[t][1] != 0 || g.ob[t][2] != 0)
            return -1;
    if (g.Db != g.Do || g.Hb != g.Ho || g.Wb != g.Wo) return -1;
    if (g.Do != (g.Di - 1) / 2 + 1 || g.Ho != (g.Hi - 1) / 2 + 1 || g.Wo != (g.Wi - 1) / 2 + 1) return -1;
    if (g.Wo < 32 || g.Ho < 4 || g.Do < 4) return -1;
    if ((long)g.Hi * g.Wi * g.C1 * 2 >= (1L << 31) || (long)g.Ho * g.Wo * g.K * 2 >= (1L << 31)) return -1;
    WgZTile tg;
    tg.nty = (g.Ho + WS_TH - 1) / WS_TH;
    tg.ntx = (g.Wo + WS_TW - 1) / WS_TW;
    const int ncb = g.C1 / 32;
    tg.nkb = g.K / 64;
    const long blocks = (long)ncb * tg.nkb;
    if (blocks > 65535) return -1;
    const long columns = (long)g.N * tg.nty * tg.ntx;
    const long per_round = blocks >= 256 ? 1 : 256 / blocks;
    long best_cost = -1;
    int best_nzc = 1;
    for (int nzc = 1; nzc <= 32 && nzc * 4 <= g.Do; nzc++) {
        const int zc = (g.Do + nzc - 1) / nzc;
        const int nzc_eff = (g.Do + zc - 1) / zc;
        const long units = columns * nzc_eff;
        const long rounds = (units + per_round - 1) / per_round;
        const long cost = rounds * (2 * zc + 3);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_nzc = nzc_eff;
        }
    }
    tg.zc = (g.Do + best_nzc - 1) / best_nzc;
    tg.nzc = (g.Do + tg.zc - 1) / tg.zc;
    const long units = columns * tg.nzc;
    if (units > (1L << 30) || units * blocks < 128) return -1;   // (small problems: the tiled kernel)
    tg.nunits = (int)units;
    const long rounds = (units + per_round - 1) / per_round;
    tg.nsplit = (int)((units + rounds - 1) / rounds);
    const size_t need = (size_t)tg.nsplit * 27 * g.C1 * g.K * sizeof(float);
    if (need > ws_bytes) return -1;
    static PerDeviceFlag cfgd;
    if (!cfgd()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad16zs), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS) !=
            hipSuccess) {
            set_error("conv wgrad (bf16 z-marching, stride 2): cannot raise the dynamic LDS limit");
            return 1;
        }
        cfgd() = true;
    }
    hipLaunchKernelGGL(k_wgrad16zs, dim3(tg.nsplit, (unsigned)blocks), dim3(256), WS_LDS, s, g, tg, a1, b, reinterpret_cast<float *>(ws));
    if (check_launch("conv wgrad (bf16 z-marching, stride 2)")) return 1;
    *nsplit_out = tg.nsplit;
    return 0;
}

}  // namespace mvd
