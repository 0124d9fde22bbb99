// k_fwd16y (round 3): z-marching 3x3x3 stride-1 convolution with 32 PRODUCE channels and 32 or 64 REDUCE channels at the
// patch resolution, on v_mfma_f32_16x16x32_bf16 -- the forward of the north_star's fused block (get_network_from_plans.py:
// 41-44) including the first decoder conv of the top stage (UNetDecoder.py:61-65: 2 x skip -> skip, read through two
// pointers = the eliminated torch.cat), its input gradients, and the InstanceNorm pieces of the block:
//   FUSE bit 0: per-workgroup (sum, sum of squares) of the fp32 accumulators per output channel (statistics epilogue);
//   FUSE bit 1: InstanceNorm-apply + LeakyReLU of the PRODUCING block on the staged input (loader prologue, NCH = 1).
//
// Why a second z-marching kernel.  k_fwd16z (32x32x16 tiles, one wave = 2 output rows x all 32 channels) keeps the 27
// taps' weights in 216 accumulator registers per wave and runs at 254 + 236 of the 512 registers: there is no room for
// 16 running sums (measured: they push staged planes / weight fragments into scratch memory; LDS float atomics instead
// made the launch 7x slower), and 64 reduce channels (110 KB of weights) do not fit at all.  Here the workgroup's four
// waves are 2 row groups x 2 halves of the OUTPUT channels, on 16x16x32 tiles: a wave owns 4 output rows x 32 voxels x
// 16 channels, so
//   * its weights are 27 taps x 4 registers per 32 reduce channels: 108 (NCH = 1) or 216 (NCH = 2) accumulator registers;
//   * one MFMA contracts all 32 channels of a chunk (K = 32): a B fragment is ONE 16-byte part of a voxel's 64-byte
//     row, the LDS image is the plain [slot][64 B] plane (parts XORed by 2 * ((x >> 2) & 1): conflict-free ds_read_b128 at
//     every x shift; row pitch 40 slots so that the swizzle does not depend on the row and rows are immediates);
//   * a lane ends with 4 consecutive channels of ONE voxel per tile: the statistics are 8 running registers, and two
//     v_permlane16_swap per tile pair turn the packed bf16 into 16-byte store images -- no LDS transposition;
//   * everything else follows k_fwd16z: input plane z' arrives once and feeds the three output planes z'+1, z', z'-1 held
//     in a ring of four accumulator sets; planes that do not exist are zero-record buffer descriptors; one barrier per
//     plane (two LDS images suffice: plane j+1 is written while plane j is read, behind the barrier that ended plane j-1).
// MFMA count per plane and wave: 216 x NCH of 16 cycles = the 108 x 32 cycles of k_fwd16z at NCH = 1.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x4y __attribute__((ext_vector_type(4)));
typedef int i32x4y __attribute__((ext_vector_type(4)));
typedef unsigned u32x4y __attribute__((ext_vector_type(4)));
typedef float f32x2y __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2y __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4y lds_u4y;

constexpr int Y_ROWP = 40, Y_SROW = 34;
// KQ = 16-channel output groups of the workgroup's four waves: 2 (32 produce channels: 2 row groups x 2 halves, an 8 x 32
// column) or 4 (64 produce channels: ONE row group x 4 quarters, a 4 x 32 column)
constexpr int y_rows(int KQ) { return 4 * (4 / KQ) + 2; }                       // halo rows of a plane image
constexpr int y_parts(int KQ) { return y_rows(KQ) * Y_SROW * 4; }               // 16-byte parts of a chunk's plane
constexpr int y_xr(int KQ) { return (y_parts(KQ) + 255) / 256; }                // staging rounds per chunk
constexpr int y_chunk(int KQ) { return y_rows(KQ) * Y_ROWP * 64; }              // bytes of one 32-channel plane image

struct Fwd16YTile {
    int nty, ntx, nzc, zc, nitems;
    int kp, koff;   // produce channels of the packed weight tensor (its row stride), first produce channel of this launch
    int vstride;    // bytes per voxel of the source tensor(s)
    int wsel[27];
};

__device__ inline unsigned cvt_pk_bf16y(float a, float b) {  // one v_cvt_pk_bf16_f32 (round-to-nearest-even)
    f32x2y v = {a, b};
    bf16x2y r = __builtin_convertvector(v, bf16x2y);
    return *reinterpret_cast<unsigned *>(&r);
}

#define MVD_MFMA16Y(ACC, WFRAG, XFRAG) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WFRAG), "v"(XFRAG))

template <int R>
struct YIdx { static constexpr int value = R; };

// compile-time slot loop: the plane body holds 36-72 fragment slots x up to 9 MFMAs -- beyond the size to which
// `#pragma unroll` unrolls (the remaining loop indexed the weight / staging arrays at run time: scratch memory)
template <class F, int... I>
__device__ __forceinline__ void y_slots_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void y_slots(F &&f) {
    y_slots_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

template <int NCH, int FUSE, int KQ = 2>
__global__ __launch_bounds__(256, 1) void k_fwd16y(const FwdGeom g, const Fwd16YTile tg, const unsigned short *__restrict__ a1,
                                                   const unsigned short *__restrict__ a2, const unsigned short *__restrict__ w,
                                                   const float *__restrict__ bias, unsigned short *__restrict__ y1,
                                                   float *__restrict__ tile_stats, const float *__restrict__ in_scale,
                                                   const float *__restrict__ in_shift, const float slope) {
    constexpr bool ST = (FUSE & 1) != 0, PRO = (FUSE & 2) != 0;
    // (PRO with NCH = 2: the producer is ONE 64-channel tensor, chunk c = its channels 32 c .. 32 c + 31 -- host-checked)
    constexpr int RG = 4 / KQ;                                    // row groups of four output rows
    constexpr int Y_PARTS = y_parts(KQ), Y_XR = y_xr(KQ), Y_CHUNK = y_chunk(KQ), OB = 32 * KQ;  // OB: bytes per output voxel
    constexpr int IMG = NCH * Y_CHUNK;   // bytes of a plane image (all chunks)
    constexpr int NS = 36 * NCH;         // fragment slots per plane: (chunk, input row r, x half, dx)
    constexpr int NSET = NCH == 1 ? 2 : 1;  // planes in flight in registers (NCH = 2: a plane takes twice as long)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = KQ == 2 ? wave >> 1 : 0, kh = KQ == 2 ? wave & 1 : wave;  // row group (output rows 4 rg .. + 3), 16-channel group
    const int n16 = lane & 15, kq = lane >> 4;
    // XCD-aware item order: each XCD walks a contiguous range of (n, chunk, ty, tx) -- neighbours share halo rows in its L2
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= tg.nitems) return;
    unsigned r_ = (unsigned)item;
    const int tx = (int)(r_ % (unsigned)tg.ntx); r_ /= (unsigned)tg.ntx;
    const int ty = (int)(r_ % (unsigned)tg.nty); r_ /= (unsigned)tg.nty;
    const int zchunk = (int)(r_ % (unsigned)tg.nzc);
    const int n_ = (int)(r_ / (unsigned)tg.nzc);
    const int y0 = ty * (4 * RG), x0 = tx * 32, zb = zchunk * tg.zc;
    const int ze = min(zb + tg.zc, g.Do);

    // weights: A operand of the swapped product D^T = W^T X^T -- lane (m = n16, kq) holds reduce channels 8 kq .. 8 kq + 7
    // of produce channel koff + 16 kh + m (the packed layout [chunk][tap][kq][k][8] of mvd_pack_weight_bf16)
    i32x4y bw[NCH][27];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int p = 0; p < 27; p++) {
            const uint4 q = *reinterpret_cast<const uint4 *>(
                w + ((size_t)((c * g.T + tg.wsel[p]) * 4 + kq) * tg.kp + tg.koff + 16 * kh + n16) * 8);
            bw[c][p] = *reinterpret_cast<const i32x4y *>(&q);
        }
    // staging slots (column constants, shared by the chunks): byte offset inside a source plane -- 0xfffffff0 outside the
    // (y, x) plane: the buffer descriptor's range check returns zeros there -- and the LDS offset inside a chunk image
    unsigned rel[Y_XR], wa[Y_XR];
#pragma unroll
    for (int u = 0; u < Y_XR; u++) {
        const int idx = u * 256 + tid;
        const bool valid = idx < Y_PARTS;
        const int slot = valid ? (idx >> 2) : 0;
        const int ry = slot / Y_SROW, sx = slot - ry * Y_SROW;
        const int gy = y0 - 1 + ry, gx = x0 - 1 + sx;
        const bool in = valid && gy >= 0 && gy < g.Hi && gx >= 0 && gx < g.Wi;
        const int part = idx & 3;
        rel[u] = in ? (unsigned)((gy * g.Wi + gx) * tg.vstride + part * 16) : 0xfffffff0u;
        wa[u] = lbase + (ry * Y_ROWP + sx) * 64 + ((part ^ (((sx >> 2) & 1) << 1)) << 4);
    }
#pragma unroll
    for (int u = 0; u < Y_XR; u++) asm volatile("" : "+v"(rel[u]), "+v"(wa[u]));
    // B-operand read addresses: x shift dx of this lane's voxel n16 (input row, x half, chunk and image are immediates;
    // NCH = 2: the second image lies beyond the 16-bit immediate)
    unsigned rb[NCH][3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
        const int sx = dx + n16;
        rb[0][dx] = lbase + ((4 * rg) * Y_ROWP + sx) * 64 + ((kq ^ (((sx >> 2) & 1) << 1)) << 4);
        asm volatile("" : "+v"(rb[0][dx]));
        if (NCH == 2) {
            rb[NCH - 1][dx] = rb[0][dx] + IMG;
            asm volatile("" : "+v"(rb[NCH - 1][dx]));
        }
    }
    // output: after the lane-row exchange lane (n16, g = kq) stores 16 bytes = channels 8 (g >> 1) .. + 7 of this wave's 16
    // of voxel (row m, x = 16 (g & 1) + n16); out of range (dropped by the descriptor) outside the volume
    unsigned voff[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int oh = y0 + 4 * rg + m, ow = x0 + (kq & 1) * 16 + n16;
        voff[m] = (oh < g.Ho && ow < g.Wo) ? (unsigned)((oh * g.Wy + ow) * OB + kh * 32 + (kq >> 1) * 16) : 0xfffffff0u;
    }
    asm volatile("" : "+v"(voff[0]), "+v"(voff[1]), "+v"(voff[2]), "+v"(voff[3]));
    float bq[4];  // bias of this lane's 4 channels (D rows 4 kq + e of the wave's 16)
#pragma unroll
    for (int e = 0; e < 4; e++) bq[e] = bias ? bias[tg.koff + 16 * kh + 4 * kq + e] : 0.f;
    float psc[PRO ? NCH : 1][PRO ? 8 : 1], psh[PRO ? NCH : 1][PRO ? 8 : 1];  // PRO: scale / shift of the staged channel octets
    if (PRO) {                                                                // (part tid & 3 of every chunk)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const float *sp = in_scale + (size_t)n_ * 32 * NCH + c * 32 + (tid & 3) * 8;
            const float *tp = in_shift + (size_t)n_ * 32 * NCH + c * 32 + (tid & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                psc[c][e] = sp[e];
                psh[c][e] = tp[e];
            }
        }
    }
    float ssum[ST ? 4 : 1], ssq[ST ? 4 : 1];  // ST: running sums of this lane's 4 channels over its voxels
#pragma unroll
    for (int e = 0; e < (ST ? 4 : 1); e++) ssum[e] = ssq[e] = 0.f;

    const size_t oplane = (size_t)g.Hy * g.Wy * OB;
    char *ybase = reinterpret_cast<char *>(y1) + (size_t)n_ * g.Dy * oplane;
    const size_t iplane = (size_t)g.Hi * g.Wi * tg.vstride;
    const char *abase[NCH];
    abase[0] = reinterpret_cast<const char *>(a1) + (size_t)n_ * g.Di * iplane;
    if (NCH == 2) abase[NCH - 1] = reinterpret_cast<const char *>(a2) + (size_t)n_ * g.Di * iplane;
    const unsigned iplane32 = (unsigned)iplane, oplane32 = (unsigned)oplane;

    // plane index j <-> input plane z' = zb - 1 + j; planes 0 .. nproc-1 carry useful MFMAs, plane nproc only drains
    const int nproc = (ze - zb) + 2;
    auto zin = [&](int j) { return zb - 1 + j; };
    auto live = [&](int j) { const int z = zin(j); return j < nproc && z >= 0 && z < g.Di; };  // block-uniform
    u32x4y v[NSET][NCH][Y_XR];
    __amdgpu_buffer_rsrc_t rin[NCH];
    auto set_in_plane = [&](int j) {  // descriptors of input plane zin(j) (uniform); zero records = a plane of zeros
#pragma unroll
        for (int c = 0; c < NCH; c++)
            rin[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase[c] + (size_t)max(zin(j), 0) * iplane), 0,
                                                       live(j) ? (int)iplane32 : 0, 0x00020000);
    };
    auto stage_load = [&](int set, int c, int u) { v[set][c][u] = __builtin_amdgcn_raw_buffer_load_b128(rin[c], (int)rel[u], 0, 0); };
    auto prologue = [&](u32x4y q, int u, bool lv, int c) {
        const bool ok = lv && rel[u] != 0xfffffff0u;
        unsigned d[4] = {q.x, q.y, q.z, q.w};
        const int cp = PRO ? c : 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float lo = __uint_as_float(d[e] << 16), hi = __uint_as_float(d[e] & 0xffff0000u);
            lo = __builtin_fmaf(lo, psc[cp][2 * e], psh[cp][2 * e]);
            hi = __builtin_fmaf(hi, psc[cp][2 * e + 1], psh[cp][2 * e + 1]);
            lo = fmaxf(lo, lo * slope);
            hi = fmaxf(hi, hi * slope);
            d[e] = ok ? cvt_pk_bf16y(lo, hi) : 0u;
        }
        return u32x4y{d[0], d[1], d[2], d[3]};
    };
    auto stage_write = [&](int set, int c, unsigned imgoff, int u, bool lv) {
        const u32x4y q = PRO ? prologue(v[set][c][u], u, lv, c) : v[set][c][u];
        if (u < Y_XR - 1 || tid < Y_PARTS - (Y_XR - 1) * 256) *(lds_u4y *)(wa[u] + imgoff + c * Y_CHUNK) = q;
    };
    auto load_plane = [&](int set, int j) {
        set_in_plane(j);
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int u = 0; u < Y_XR; u++) stage_load(set, c, u);
    };
    // prologue of the march: plane 0 in image 0, the next plane(s) in flight
    load_plane(0, 0);
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < Y_XR; u++) stage_write(0, c, 0, u, live(0));
    load_plane(NSET == 2 ? 1 : 0, 1);
    if (NSET == 2) load_plane(0, 2);
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int p = 0; p < 27; p++) asm volatile("" : "+a"(bw[c][p]));

    f32x4y S[4][4][2];  // accumulator ring: output plane zo lives in S[(zo - zb + 1) & 3]; [output row m][x half]
#pragma unroll
    for (int q = 0; q < 32; q++) {
        S[q >> 3][(q >> 1) & 3][q & 1] = f32x4y{bq[0], bq[1], bq[2], bq[3]};
        asm volatile("" : "+v"(S[q >> 3][(q >> 1) & 3][q & 1]));  // (materialised here, see epilogue_pair)
    }
    __syncthreads();
    i32x4y af[4];  // ring of four B fragments, fetched three slots ahead (also across planes): the edge rows' slots hold
    //               only three MFMAs (48 cycles), two of them do not cover an LDS round trip
    // slot s = (chunk c, f): f = (r * 2 + xh) * 3 + dx of plane image IMGI
#define MVD_Y_READ(IMGI, SLOT, BUF)                                                                                       \
    {                                                                                                                     \
        const int c_ = (SLOT) / 36, f_ = (SLOT) % 36;                                                                     \
        const int off_ = (f_ / 6) * (Y_ROWP * 64) + ((f_ / 3) & 1) * 1024 + c_ * Y_CHUNK + (NCH == 1 ? (IMGI) * IMG : 0);     \
        const u32x4y q_ = *(lds_u4y *)(rb[NCH == 1 ? 0 : (IMGI)][f_ % 3] + off_);                                        \
        af[BUF] = *reinterpret_cast<const i32x4y *>(&q_);                                                                \
    }
    MVD_Y_READ(0, 0, 0);
    MVD_Y_READ(0, 1, 1);
    MVD_Y_READ(0, 2, 2);
    asm volatile("s_nop 4");  // (the accumulator sets were written by VALU moves)

    // plane j, ring position R = j & 3 (image R & 1): sets NEW = R (output plane z'+1, holds the bias), MID = R-1 (z'),
    // OLD = R-2 (z'-1: complete after this plane), DRN = R-3 (z'-2: completed by the previous plane, converted, stored and
    // reset to the bias during this one: it is the next plane's NEW).  One branch-free body for every plane, as in k_fwd16z.
    auto plane = [&](auto Rc, int j) __attribute__((always_inline)) {
        constexpr int R = decltype(Rc)::value;
        constexpr int NEW = R, MID = (R + 3) & 3, OLD = (R + 2) & 3, DRN = (R + 1) & 3;
        constexpr int IC = R & 1, IN_ = IC ^ 1;   // image read by this plane / image written during it
        __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, 0, 0x00020000);
        const int zo = zin(j) - 2;
        const bool pst = zo >= zb && zo < ze;       // the drained plane is one of the chunk's
        const bool lvn = live(j + 1);               // the plane written to LDS during this one exists (PRO)
        auto epilogue_pair = [&](int m) {           // tiles (m, x half 0) and (m, x half 1) of the drained set
            f32x4y t0 = S[DRN][m][0], t1 = S[DRN][m][1];
            if (ST) {
                if (pst) {  // uniform
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        ssum[e] += t0[e];
                        ssq[e] = __builtin_fmaf(t0[e], t0[e], ssq[e]);
                        ssum[e] += t1[e];
                        ssq[e] = __builtin_fmaf(t1[e], t1[e], ssq[e]);
                    }
                }
            }
            unsigned ax = cvt_pk_bf16y(t0[0], t0[1]), ay = cvt_pk_bf16y(t0[2], t0[3]);
            unsigned bx = cvt_pk_bf16y(t1[0], t1[1]), by = cvt_pk_bf16y(t1[2], t1[3]);
            // odd 16-lane rows of the first operand <-> even rows of the second: afterwards rows 0 / 2 hold channels
            // 0-7 / 8-15 of tile 0's voxel, rows 1 / 3 the same of tile 1's (16 contiguous bytes per lane)
            auto rx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
            auto ry = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
            const u32x4y img = {rx[0], ry[0], rx[1], ry[1]};
            __builtin_amdgcn_raw_buffer_store_b128(img, rout, (int)voff[m], 0, 0);
            S[DRN][m][0] = f32x4y{bq[0], bq[1], bq[2], bq[3]};   // the next plane's NEW set
            S[DRN][m][1] = f32x4y{bq[0], bq[1], bq[2], bq[3]};
            // materialised HERE: left alone the compiler sinks these moves down to the first MFMA that accumulates into the
            // set, and a VALU write needs two wait states before an MFMA reads it -- which the hazard recogniser does not
            // insert in front of inline asm (sporadically stale accumulators: found by the run-to-run comparison)
            asm volatile("" : "+v"(S[DRN][m][0]), "+v"(S[DRN][m][1]));
        };
        y_slots<NS>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            // B fragment three slots ahead (the last three: the next plane's first, behind the barrier below)
            if (s + 3 < NS) MVD_Y_READ(IC, s + 3, (s + 3) % 4)
            else MVD_Y_READ(IN_, s + 3 - NS, (s + 3) % 4)
            if (s == 0) {  // descriptors: the drained output plane (zero records: stores dropped) ...
                rout = __builtin_amdgcn_make_buffer_rsrc(ybase + (size_t)max(zo, 0) * oplane, 0, pst ? (int)oplane32 : 0, 0x00020000);
            }
            if (s == 1) set_in_plane(j + (NSET == 2 ? 3 : 2));  // ... and the input plane the loads below fetch
            // Non-MFMA work goes where the slots are long (an MFMA leaves 8 of its 16 cycles to other issue: rows r = 2, 3
            // have nine MFMAs per slot, r = 0, 5 three).  NCH = 1: plane j+1 from the registers into the other image in the
            // r = 1 slots, the drained set (conversion + lane-row exchange + store + bias reset, one tile pair per slot) in
            // the r = 2 slots, the loads of plane j+3 into the freed registers in the r = 3 slots.
            if (NCH == 1) {
                if (s >= 6 && s < 6 + Y_XR) stage_write((R + 1) & 1, 0, IN_ * IMG, s - 6, lvn);
                if (s >= 12 && s < 16) epilogue_pair(s - 12);
                if (s >= 18 && s < 18 + Y_XR) stage_load((R + 1) & 1, 0, s - 18);
            } else {  // 72 slots: writes 6..17, drained set 18..21, loads of plane j+2 in the second chunk's long slots
                if (s >= 6 && s < 6 + 2 * Y_XR) stage_write(0, (s - 6) / Y_XR, IN_ * IMG, (s - 6) % Y_XR, lvn);
                if (s >= 18 && s < 22) epilogue_pair(s - 18);
                if (s >= 42 && s < 42 + 2 * Y_XR) stage_load(0, (s - 42) / Y_XR, (s - 42) % Y_XR);
            }
            // plane j+1 complete in LDS (this wave's writes retired in order before the fragment reads it has consumed since),
            // and no read of this plane's image is issued behind the barrier (the read above was its last): the next plane
            // may overwrite it
            if (s == NS - 4) asm volatile("s_barrier" ::: "memory");
            constexpr int c = s / 36, f = s % 36, r = f / 6, xh = (f / 3) & 1, dx = f % 3;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int dy = r - m;  // input row r is output row m shifted by dy - 1
                if (dy < 0 || dy > 2) continue;  // (folded: m, r are constants)
                MVD_MFMA16Y(S[NEW][m][xh], bw[c][0 * 9 + dy * 3 + dx], af[s % 4]);
                MVD_MFMA16Y(S[MID][m][xh], bw[c][1 * 9 + dy * 3 + dx], af[s % 4]);
                MVD_MFMA16Y(S[OLD][m][xh], bw[c][2 * 9 + dy * 3 + dx], af[s % 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // an MFMA result needs wait states before anything but an accumulating MFMA reads it (the next plane converts OLD)
        asm volatile("s_nop 7\n\ts_nop 4" : "+v"(S[OLD][0][0]), "+v"(S[OLD][0][1]));
    };
    for (int j = 0; j <= nproc; j += 4) {
        plane(YIdx<0>(), j);
        if (j + 1 > nproc) break;
        plane(YIdx<1>(), j + 1);
        if (j + 2 > nproc) break;
        plane(YIdx<2>(), j + 2);
        if (j + 3 > nproc) break;
        plane(YIdx<3>(), j + 3);
    }
#undef MVD_Y_READ
    if (ST) {
        // voxels of a channel quad sit in the 16 lanes of a row: xor-butterfly over lane bits 0..3, wave totals through
        // LDS (all plane images are dead), the two row groups added in order
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], off, 64);
                ssq[e] += __shfl_xor(ssq[e], off, 64);
            }
        __syncthreads();
        float *wtot = reinterpret_cast<float *>(lds8);  // [wave][16 channels][2]
        if (n16 == 0) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                wtot[(wave * 16 + 4 * kq + e) * 2 + 0] = ssum[e];
                wtot[(wave * 16 + 4 * kq + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        if (tid < 32 * KQ) {   // [16 KQ channels][2]
            const int c = tid >> 1, which = tid & 1, khc = c >> 4, cl = c & 15;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < RG; r++) a += wtot[((r * KQ + khc) * 16 + cl) * 2 + which];   // wave = rg * KQ + kh (KQ = 2) / kh
            const int tile = (zchunk * tg.nty + ty) * tg.ntx + tx;
            tile_stats[(((size_t)n_ * ((size_t)tg.nzc * tg.nty * tg.ntx) + tile) * tg.kp + tg.koff) * 2 + tid] = a;
        }
    }
}

static int g_fwd16y = -1;  // -1: from MVD_FWD16Y (default on)
int fwd16y_enabled() {
    if (g_fwd16y < 0) g_fwd16y = getenv("MVD_FWD16Y") ? (atoi(getenv("MVD_FWD16Y")) != 0) : 1;
    return g_fwd16y;
}
void fwd16y_enable(int on) { g_fwd16y = on ? 1 : 0; }

// host side: returns -1 when the shape is not this kernel's; *stats_tiles (query mode, no launch) = tiles per sample
int launch_fwd16y(const FwdGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *w,
                  const float *bias, unsigned short *y1, unsigned short *y2, hipStream_t s, const Fwd16Fuse *fuse, int ncu,
                  int *stats_tiles_only) {
    if (!fwd16y_enabled()) return -1;
    const int KT = g.K1;                                   // produce channels per launch: 32 (KQ = 2) or 64 (KQ = 4)
    if (KT != 32 && KT != 64) return -1;
    const int KQ = KT / 16, rows = 4 * (4 / KQ);
    const bool two_out = g.K2 == KT && y2 != nullptr;
    if (stats_tiles_only && g.K2 != 0) return -1;
    if ((g.K2 != 0 && !two_out) || g.ntaps != 27 || g.T != 27 || g.acc) return -1;
    int nch, vstride;
    if (g.C1 == 32 && g.C2 == 0) { nch = 1; vstride = 64; }
    else if (g.C1 == 32 && g.C2 == 32) { nch = 2; vstride = 64; }
    else if (g.C1 == 64 && g.C2 == 0) { nch = 2; vstride = 128; }
    else return -1;
    for (int a = 0; a < 3; a++)
        if (g.sa[a] != 1 || g.so[a] != 1 || g.oo[a] != 0) return -1;
    if (g.Dy != g.Do || g.Hy != g.Ho || g.Wy != g.Wo || g.Di != g.Do || g.Hi != g.Ho || g.Wi != g.Wo) return -1;
    if ((long)g.Hi * g.Wi * vstride >= (1L << 31) || (long)g.Hy * g.Wy * 2 * KT >= (1L << 31)) return -1;
    Fwd16YTile tz;
    memset(&tz, 0, sizeof(tz));
    for (int p = 0; p < 27; p++) {
        const int dz = p / 9 - 1, dy = (p / 3) % 3 - 1, dx = p % 3 - 1;
        int hit = -1;
        for (int t = 0; t < 27; t++)
            if (g.off[t][0] == dz && g.off[t][1] == dy && g.off[t][2] == dx) hit = t;
        if (hit < 0) return -1;
        tz.wsel[p] = g.wt[hit];
    }
    tz.nty = (g.Ho + rows - 1) / rows;
    tz.ntx = (g.Wo + 31) / 32;
    // large volumes only (>= 4 4x8x8 tiles per CU, the threshold of the other weights-resident kernels)
    if ((long)g.N * ((g.Do + 3) / 4) * ((g.Ho + 7) / 8) * ((g.Wo + 7) / 8) < 4L * ncu) return -1;
    const long cols = (long)g.N * tz.nty * tz.ntx;
    int best = 1;
    double best_cost = 1e30;
    for (int nz = 1; nz <= (g.Do + 7) / 8; nz++) {  // z chunks: whole rounds of the chip, >= 8 planes per chunk
        const int zc = (g.Do + nz - 1) / nz;
        const long wgs = cols * ((g.Do + zc - 1) / zc);
        const double cost = (double)((wgs + ncu - 1) / ncu) * (zc + 2.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = nz; }
    }
    tz.zc = (g.Do + best - 1) / best;
    tz.nzc = (g.Do + tz.zc - 1) / tz.zc;
    tz.nitems = (int)(cols * tz.nzc);
    tz.vstride = vstride;
    const bool stats_ok = !two_out && g.Ho % rows == 0 && g.Wo % 32 == 0;
    if (stats_tiles_only) {
        *stats_tiles_only = stats_ok ? tz.nzc * tz.nty * tz.ntx : 0;
        return 0;
    }
    const bool want_stats = fuse && fuse->tile_stats && stats_ok;
    const bool want_pro = fuse && fuse->in_scale && fuse->in_shift;
    if (want_pro && (two_out || (nch == 2 && vstride != 128))) return -1;  // one producer tensor of 32 or 64 channels
    if (fuse && fuse->ntiles) *fuse->ntiles = want_stats ? tz.nzc * tz.nty * tz.ntx : 0;
    typedef void (*ky_t)(const FwdGeom, const Fwd16YTile, const unsigned short *, const unsigned short *, const unsigned short *,
                         const float *, unsigned short *, float *, const float *, const float *, const float);
    static const ky_t kern[2][2][4] = {{{k_fwd16y<1, 0, 2>, k_fwd16y<1, 1, 2>, k_fwd16y<1, 2, 2>, k_fwd16y<1, 3, 2>},
                                        {k_fwd16y<2, 0, 2>, k_fwd16y<2, 1, 2>, k_fwd16y<2, 2, 2>, k_fwd16y<2, 3, 2>}},
                                       {{k_fwd16y<1, 0, 4>, k_fwd16y<1, 1, 4>, k_fwd16y<1, 2, 4>, k_fwd16y<1, 3, 4>},
                                        {k_fwd16y<2, 0, 4>, k_fwd16y<2, 1, 4>, k_fwd16y<2, 2, 4>, k_fwd16y<2, 3, 4>}}};
    const int fz = (want_stats ? 1 : 0) | (want_pro ? 2 : 0), ki = KQ == 2 ? 0 : 1;
    const ky_t kfn = kern[ki][nch - 1][fz];
    const size_t lds = 2 * (size_t)nch * (KQ == 2 ? y_chunk(2) : y_chunk(4));
    static PerDeviceFlag configured[2][2][4];
    if (!configured[ki][nch - 1][fz]()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(2 * 2 * y_chunk(2)));
        if (e != hipSuccess) {
            set_error("conv fwd16y: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            return 1;
        }
        configured[ki][nch - 1][fz]() = true;
    }
    const int per_xcd = (tz.nitems + 7) / 8;
    tz.kp = g.K1 + g.K2;
    const unsigned short *c1 = vstride == 128 ? a1 + 32 : a2;  // second chunk: the other tensor, or channels 32.. of the one
    for (int q = 0; q < (two_out ? 2 : 1); q++) {
        tz.koff = KT * q;
        hipLaunchKernelGGL(kfn, dim3((unsigned)(per_xcd * 8)), dim3(256), lds, s, g, tz, a1, c1, w, bias, q ? y2 : y1,
                           want_stats ? fuse->tile_stats : nullptr, want_pro ? fuse->in_scale : nullptr,
                           want_pro ? fuse->in_shift : nullptr, fuse ? fuse->slope : 0.f);
        if (check_launch("conv fwd16y (z-marching bf16 mfma 16x16x32, weights in registers)")) return 1;
    }
    return 0;
}

}  // namespace mvd
