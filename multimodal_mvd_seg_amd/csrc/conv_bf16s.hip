// k_fwd16ys (round 3): z-marching 3x3x3 STRIDE-2 convolution, 32 reduce -> 64 produce channels, bf16 -- the first conv of
// encoder stage 1 (get_network_from_plans.py:41-44 with stride 2: PlainConvEncoder's strided first block), forward.
//
// The generic kernel (k_fwd16<2,1,3,22>) gives this layer one workgroup per CU whose phases never overlap: an 88 KB halo
// fetch, nine weight groups through LDS with a barrier each, 108 MFMAs per wave, the store -- ~14 us per 128-voxel tile,
// 0.23 ms for 32 -> 64 at 128^3 -> 64^3, whose traffic (0.27 GB in, 0.07 GB out) is worth 0.07 ms.  This is k_fwd16y's
// structure bent to stride 2:
//   * a workgroup owns a column of 4 x 32 OUTPUT voxels and marches along z over the INPUT planes; wave w owns output
//     channels 16 w .. 16 w + 15 for the whole tile: its 27 x 4 weight registers stay in the accumulator file;
//   * an odd input plane 2 q + 1 feeds output planes q (dz = 2) and q + 1 (dz = 0), an even one 2 q feeds q (dz = 1): two
//     accumulator sets, 144 / 72 MFMAs (v_mfma_f32_16x16x32_bf16) per plane and wave; a finished set is converted,
//     exchanged across lane rows, stored and reset during the following even plane;
//   * stride 2 along x would make the lanes of a B-fragment read 128 bytes apart (half of the banks): the LDS image is split
//     by the PARITY of the input column -- [input row 9][parity 2][40 slots][64 B] -- so that tap dx reads parity dx & 1 at
//     unit stride, shifted by dx >> 1, with k_fwd16y's part swizzle (conflict-free ds_read_b128);
//   * input rows are immediates: output row m, tap dy reads input row 2 m + dy; planes outside the volume / chunk are
//     zero-record descriptors, columns / rows outside it out-of-range offsets (zeros), as everywhere.
// Two LDS images (92 KB), one barrier per plane, two staged planes in flight in registers.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef int i32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4s lds_u4s;

constexpr int S2_ROWP = 40;                       // slots per (input row, parity) line
constexpr int S2_ROWS = 9, S2_COLS = 65;          // input footprint of a 4 x 32 output tile
constexpr int S2_IMG = S2_ROWS * 2 * S2_ROWP * 64;  // 46080 bytes per plane image
constexpr int S2_PARTS = S2_ROWS * S2_COLS * 4;   // 2340 16-byte parts per plane
constexpr int S2_XR = (S2_PARTS + 255) / 256;     // 10 staging rounds
constexpr int S2_NS = S2_ROWS * 2 * 3;            // 54 fragment slots per plane: (input row R, x half, dx)

struct Fwd16STile {
    int nty, ntx, nzc, zc, nitems;
    int kp;        // produce channels of the packed weight tensor (row stride)
    int wsel[27];
};

__device__ inline unsigned cvt_pk_bf16s(float a, float b) {
    f32x2s v = {a, b};
    bf16x2s r = __builtin_convertvector(v, bf16x2s);
    return *reinterpret_cast<unsigned *>(&r);
}

#define MVD_MFMA16S(ACC, WFRAG, XFRAG) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WFRAG), "v"(XFRAG))

template <int R>
struct SIdx { static constexpr int value = R; };

template <class F, int... I>
__device__ __forceinline__ void s_slots_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void s_slots(F &&f) {
    s_slots_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

__global__ __launch_bounds__(256, 1) void k_fwd16ys(const FwdGeom g, const Fwd16STile tg, const unsigned short *__restrict__ a1,
                                                    const unsigned short *__restrict__ w, const float *__restrict__ bias,
                                                    unsigned short *__restrict__ y1) {
    constexpr int OB = 128;  // bytes per output voxel (64 channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int kh = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = 16-channel group
    const int n16 = lane & 15, kq = lane >> 4;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= tg.nitems) return;
    unsigned r_ = (unsigned)item;
    const int tx = (int)(r_ % (unsigned)tg.ntx); r_ /= (unsigned)tg.ntx;
    const int ty = (int)(r_ % (unsigned)tg.nty); r_ /= (unsigned)tg.nty;
    const int zchunk = (int)(r_ % (unsigned)tg.nzc);
    const int n_ = (int)(r_ / (unsigned)tg.nzc);
    const int y0 = ty * 4, x0 = tx * 32, zb = zchunk * tg.zc;   // output coordinates
    const int ze = min(zb + tg.zc, g.Do);

    // weights: A operand of D^T = W^T X^T -- lane (m = n16, kq) holds reduce channels 8 kq .. + 7 of produce channel 16 kh + m
    i32x4s bw[27];
#pragma unroll
    for (int p = 0; p < 27; p++) {
        const uint4 q = *reinterpret_cast<const uint4 *>(w + ((size_t)(tg.wsel[p] * 4 + kq) * tg.kp + 16 * kh + n16) * 8);
        bw[p] = *reinterpret_cast<const i32x4s *>(&q);
    }
    // staging slots (column constants): byte offset inside a source plane (0xfffffff0 outside it: zeros) and LDS offset
    unsigned rel[S2_XR], wa[S2_XR];
#pragma unroll
    for (int u = 0; u < S2_XR; u++) {
        const int idx = u * 256 + tid;
        const bool valid = idx < S2_PARTS;
        const int slot = valid ? (idx >> 2) : 0;
        const int R = slot / S2_COLS, X = slot - R * S2_COLS;
        const int gy = 2 * y0 - 1 + R, gx = 2 * x0 - 1 + X;
        const bool in = valid && gy >= 0 && gy < g.Hi && gx >= 0 && gx < g.Wi;
        const int part = idx & 3, xi = X >> 1;
        rel[u] = in ? (unsigned)((gy * g.Wi + gx) * 64 + part * 16) : 0xfffffff0u;
        wa[u] = lbase + ((R * 2 + (X & 1)) * S2_ROWP + xi) * 64 + ((part ^ (((xi >> 2) & 1) << 1)) << 4);
    }
#pragma unroll
    for (int u = 0; u < S2_XR; u++) asm volatile("" : "+v"(rel[u]), "+v"(wa[u]));
    // B-operand read addresses: shift sh = dx >> 1 of this lane's voxel n16 (input row, parity, x half and image: immediates)
    unsigned rb[2];
#pragma unroll
    for (int sh = 0; sh < 2; sh++) {
        const int sx = sh + n16;
        rb[sh] = lbase + sx * 64 + ((kq ^ (((sx >> 2) & 1) << 1)) << 4);
        asm volatile("" : "+v"(rb[sh]));
    }
    // output: after the lane-row exchange lane (n16, g = kq) stores 16 bytes = channels 8 (g >> 1) .. + 7 of this wave's 16 of
    // voxel (row m, x = 16 (g & 1) + n16); out of range (dropped by the descriptor) outside the volume
    unsigned voff[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int oh = y0 + m, ow = x0 + (kq & 1) * 16 + n16;
        voff[m] = (oh < g.Ho && ow < g.Wo) ? (unsigned)((oh * g.Wy + ow) * OB + kh * 32 + (kq >> 1) * 16) : 0xfffffff0u;
    }
    asm volatile("" : "+v"(voff[0]), "+v"(voff[1]), "+v"(voff[2]), "+v"(voff[3]));
    float bq[4];
#pragma unroll
    for (int e = 0; e < 4; e++) bq[e] = bias ? bias[16 * kh + 4 * kq + e] : 0.f;

    const size_t oplane = (size_t)g.Hy * g.Wy * OB;
    char *ybase = reinterpret_cast<char *>(y1) + (size_t)n_ * g.Dy * oplane;
    const size_t iplane = (size_t)g.Hi * g.Wi * 64;
    const char *abase = reinterpret_cast<const char *>(a1) + (size_t)n_ * g.Di * iplane;
    const unsigned iplane32 = (unsigned)iplane, oplane32 = (unsigned)oplane;

    // plane index j <-> input plane z' = 2 zb - 1 + j; planes 0 .. 2 nzo carry MFMAs of the chunk, plane 2 nzo + 1 drains
    const int nzo = ze - zb, nproc = 2 * nzo + 1;
    auto zin = [&](int j) { return 2 * zb - 1 + j; };
    auto live = [&](int j) { const int z = zin(j); return j <= 2 * nzo && z >= 0 && z < g.Di; };  // block-uniform
    u32x4s v[2][S2_XR];
    __amdgpu_buffer_rsrc_t rin;
    auto set_in_plane = [&](int j) {
        rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase + (size_t)max(zin(j), 0) * iplane), 0,
                                                live(j) ? (int)iplane32 : 0, 0x00020000);
    };
    auto stage_load = [&](int set, int u) { v[set][u] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)rel[u], 0, 0); };
    auto stage_write = [&](int set, unsigned imgoff, int u) {
        if (u < S2_XR - 1 || tid < S2_PARTS - (S2_XR - 1) * 256) *(lds_u4s *)(wa[u] + imgoff) = v[set][u];
    };
    auto load_plane = [&](int set, int j) {
        set_in_plane(j);
#pragma unroll
        for (int u = 0; u < S2_XR; u++) stage_load(set, u);
    };
    load_plane(0, 0);
#pragma unroll
    for (int u = 0; u < S2_XR; u++) stage_write(0, 0, u);
    load_plane(1, 1);
    load_plane(0, 2);
#pragma unroll
    for (int p = 0; p < 27; p++) asm volatile("" : "+a"(bw[p]));

    f32x4s S[2][4][2];  // accumulator sets: output plane o (relative to zb) lives in S[o & 1]; [output row m][x half]
#pragma unroll
    for (int q = 0; q < 16; q++) {
        S[q >> 3][(q >> 1) & 3][q & 1] = f32x4s{bq[0], bq[1], bq[2], bq[3]};
        asm volatile("" : "+v"(S[q >> 3][(q >> 1) & 3][q & 1]));
    }
    __syncthreads();
    i32x4s af[4];  // ring of four B fragments, fetched three slots ahead (also across planes)
    // slot s: R = s / 6 (input row), xh = (s / 3) & 1, dx = s % 3
#define MVD_S_READ(IMGI, SLOT, BUF)                                                                                      \
    {                                                                                                                    \
        constexpr int R_ = (SLOT) / 6, xh_ = ((SLOT) / 3) & 1, dx_ = (SLOT) % 3;                                       \
        constexpr int off_ = ((R_ * 2 + (dx_ & 1)) * S2_ROWP) * 64 + xh_ * 1024 + (IMGI) * S2_IMG;                       \
        const u32x4s q_ = *(lds_u4s *)(rb[dx_ >> 1] + off_);                                                            \
        af[BUF] = *reinterpret_cast<const i32x4s *>(&q_);                                                               \
    }
    MVD_S_READ(0, 0, 0);
    MVD_S_READ(0, 1, 1);
    MVD_S_READ(0, 2, 2);
    asm volatile("s_nop 4");

    // plane j, P = j & 3.  Even j (odd input plane 2 q + 1): set OS (output q) takes dz = 2, set NW (output q + 1) dz = 0.
    // Odd j (even input plane 2 q): set CU (output q) takes dz = 1, the other set (output q - 1, complete since the previous
    // plane) is converted, stored and reset to the bias.
    auto plane = [&](auto Pc, int j) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr bool EVENJ = (P & 1) == 0;
        constexpr int NW = (P >> 1) & 1, OS = NW ^ 1;   // even j: newer / older set;  odd j: CU = NW, drained = OS
        constexpr int IC = P & 1, IN_ = IC ^ 1;         // image read by this plane / image written during it
        constexpr int RO = (P & 1) * 2;                 // ring offset of this plane's slot 0
        __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, 0, 0x00020000);
        const int zo = zb + (j - 3) / 2;                // (odd j) the drained output plane
        const bool pst = !EVENJ && j >= 3 && zo < ze;
        auto epilogue_pair = [&](int m) {
            f32x4s t0 = S[OS][m][0], t1 = S[OS][m][1];
            unsigned ax = cvt_pk_bf16s(t0[0], t0[1]), ay = cvt_pk_bf16s(t0[2], t0[3]);
            unsigned bx = cvt_pk_bf16s(t1[0], t1[1]), by = cvt_pk_bf16s(t1[2], t1[3]);
            auto rx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
            auto ry = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
            const u32x4s img = {rx[0], ry[0], rx[1], ry[1]};
            __builtin_amdgcn_raw_buffer_store_b128(img, rout, (int)voff[m], 0, 0);
            S[OS][m][0] = f32x4s{bq[0], bq[1], bq[2], bq[3]};
            S[OS][m][1] = f32x4s{bq[0], bq[1], bq[2], bq[3]};
            // (pinned where written: a VALU write needs two wait states before an inline-asm MFMA reads it)
            asm volatile("" : "+v"(S[OS][m][0]), "+v"(S[OS][m][1]));
        };
        s_slots<S2_NS>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            // (54 slots per plane: the fragment ring advances by 54 % 4 = 2 per plane -- RO continues the ring index)
            if (s + 3 < S2_NS) MVD_S_READ(IC, s + 3, (s + 3 + RO) % 4)
            else MVD_S_READ(IN_, s + 3 - S2_NS, (s + 3 + RO) % 4)
            if (s == 0 && !EVENJ)
                rout = __builtin_amdgcn_make_buffer_rsrc(ybase + (size_t)max(zo, 0) * oplane, 0, pst ? (int)oplane32 : 0, 0x00020000);
            if (s == 1) set_in_plane(j + 3);
            // plane j + 1 from the registers into the other image (slots 6 ..), the drained set in slots 18 .. 21 (odd j), the
            // loads of plane j + 3 into the freed registers (slots 24 ..)
            if (s >= 6 && s < 6 + S2_XR) stage_write((P + 1) & 1, IN_ * S2_IMG, s - 6);
            if (!EVENJ && s >= 18 && s < 22) epilogue_pair(s - 18);
            if (s >= 24 && s < 24 + S2_XR) stage_load((P + 1) & 1, s - 24);
            if (s == S2_NS - 4) asm volatile("s_barrier" ::: "memory");
            constexpr int R = s / 6, xh = (s / 3) & 1, dx = s % 3;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int dy = R - 2 * m;  // input row R = 2 m + dy
                if (dy < 0 || dy > 2) continue;
                if (EVENJ) {
                    MVD_MFMA16S(S[OS][m][xh], bw[2 * 9 + dy * 3 + dx], af[(s + RO) % 4]);
                    MVD_MFMA16S(S[NW][m][xh], bw[0 * 9 + dy * 3 + dx], af[(s + RO) % 4]);
                } else {
                    MVD_MFMA16S(S[NW][m][xh], bw[1 * 9 + dy * 3 + dx], af[(s + RO) % 4]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // an MFMA result needs wait states before anything but an accumulating MFMA reads it (the next plane may convert a set)
        asm volatile("s_nop 7\n\ts_nop 4" : "+v"(S[0][0][0]), "+v"(S[1][0][0]));
    };
    for (int j = 0; j <= nproc; j += 4) {
        plane(SIdx<0>(), j);
        if (j + 1 > nproc) break;
        plane(SIdx<1>(), j + 1);
        if (j + 2 > nproc) break;
        plane(SIdx<2>(), j + 2);
        if (j + 3 > nproc) break;
        plane(SIdx<3>(), j + 3);
    }
#undef MVD_S_READ
}

static int &fwd16ys_mode() {
    static int v = getenv("MVD_FWD16YS") ? atoi(getenv("MVD_FWD16YS")) : 1;
    return v;
}

// host side: -1 when the shape is not this kernel's
int launch_fwd16ys(const FwdGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *w, const float *bias,
                   unsigned short *y1, unsigned short *y2, hipStream_t s, int ncu) {
    if (!fwd16ys_mode() || !fwd16y_enabled() || g.ntaps != 27 || g.acc) return -1;  // (mvd_set_bf16_zmarch_kernel(0) switches both off)
    if (g.C1 != 32 || g.C2 != 0 || g.K1 != 64 || g.K2 != 0 || a2 != nullptr || y2 != nullptr) return -1;
    for (int a = 0; a < 3; a++)
        if (g.sa[a] != 2 || g.so[a] != 1 || g.oo[a] != 0) return -1;
    if (g.Dy != g.Do || g.Hy != g.Ho || g.Wy != g.Wo) return -1;
    if (g.Do != (g.Di - 1) / 2 + 1 || g.Ho != (g.Hi - 1) / 2 + 1 || g.Wo != (g.Wi - 1) / 2 + 1) return -1;
    if ((long)g.Hi * g.Wi * 64 >= (1L << 31) || (long)g.Hy * g.Wy * 128 >= (1L << 31)) return -1;
    Fwd16STile tz;
    memset(&tz, 0, sizeof(tz));
    for (int p = 0; p < 27; p++) {
        const int dz = p / 9 - 1, dy = (p / 3) % 3 - 1, dx = p % 3 - 1;
        int hit = -1;
        for (int t = 0; t < 27; t++)
            if (g.off[t][0] == dz && g.off[t][1] == dy && g.off[t][2] == dx) hit = t;
        if (hit < 0) return -1;
        tz.wsel[p] = g.wt[hit];
    }
    tz.nty = (g.Ho + 3) / 4;
    tz.ntx = (g.Wo + 31) / 32;
    // large volumes only: at least two workgroups' worth of columns x planes per CU
    if ((long)g.N * g.Do * tz.nty * tz.ntx < 8L * ncu) return -1;
    const long cols = (long)g.N * tz.nty * tz.ntx;
    int best = 1;
    double best_cost = 1e30;
    for (int nz = 1; nz <= (g.Do + 3) / 4; nz++) {  // z chunks: whole rounds of the chip, >= 4 output planes per chunk
        const int zc = (g.Do + nz - 1) / nz;
        const long wgs = cols * ((g.Do + zc - 1) / zc);
        const double cost = (double)((wgs + ncu - 1) / ncu) * (2 * zc + 4.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = nz; }
    }
    tz.zc = (g.Do + best - 1) / best;
    tz.nzc = (g.Do + tz.zc - 1) / tz.zc;
    tz.nitems = (int)(cols * tz.nzc);
    tz.kp = g.K1 + g.K2;
    static PerDeviceFlag configured;
    if (!configured()) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd16ys), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 2 * S2_IMG);
        if (e != hipSuccess) {
            set_error("conv fwd16ys: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            return 1;
        }
        configured() = true;
    }
    const int per_xcd = (tz.nitems + 7) / 8;
    hipLaunchKernelGGL(k_fwd16ys, dim3((unsigned)(per_xcd * 8)), dim3(256), 2 * S2_IMG, s, g, tz, a1, w, bias, y1);
    return check_launch("conv fwd16ys (z-marching stride-2 bf16 mfma 16x16x32, weights in registers)");
}

}  // namespace mvd
