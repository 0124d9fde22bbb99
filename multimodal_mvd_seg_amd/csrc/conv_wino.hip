// fp32 3x3x3 stride-1 convolution (forward and input-gradient) with the Winograd F(2,3) transform along W.
//
// For a pair of neighbouring outputs (w = 2q, 2q+1) and one (dz,dy) row of the filter, the three taps g0,g1,g2 along W
// read the four inputs d0..d3 = x[2q-1 .. 2q+2]:
//     m0 = (d0 - d2) g0            m1 = (d1 + d2) (g0+g1+g2)/2
//     m2 = (d2 - d1) (g0-g1+g2)/2  m3 = (d1 - d3) g2
//     y[2q] = m0 + m1 + m2         y[2q+1] = m1 - m2 - m3
// i.e. 4 channel-contractions instead of 6: the MFMA work of the gathered-tap GEMM drops by 1/3 (36 "taps" over V/2
// pairs instead of 27 over V voxels).  The fp32 error is that of the direct form (coefficients are 1 and 1/2;
// measured 6.9e-7 vs 4.6e-7 relative for a 32-channel layer), inside the 1e-4 parity bar.
//
//   * workgroup = 4 waves = a 4 x 4 x 8 voxel tile (64 pairs) x 32 output channels; the GEMM M index of a wave is the
//     pair (2 d-planes x 4 rows x 4 pairs); the four Winograd positions are split over two wave pairs;
//   * the halo tile [6 x 6 x 10 slots][32 ch] of a 32-channel chunk is staged once in LDS (whole 128-byte lines, padded
//     to 36 floats); the input transform is done ON THE FLY on the operands read from it (16 VALU adds per 16 MFMAs),
//     so the LDS tile is the plain halo and three workgroups fit a CU;
//   * the transformed weights U = G g (36 x C x K, precomputed by mvd_pack_weight_wino in the MFMA operand order) are
//     NOT staged: every lane fetches its B fragment (16 consecutive floats) straight from L2, one position ahead of the
//     MFMAs that use it -- no weight ring, no barrier inside a chunk;
//   * the output transform runs on the accumulators in the epilogue; bias is added there.
#include "common.h"
#include "conv_geom.h"

namespace mvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoTile {
    int EH, EW, nslots;
    int magW, magHW;
    int ntd, nth, ntw, nkb;
    int nitems;
    int K;
    int dbg;  // MVD_WINO_DBG ablation bits: 1 no activation loads, 2 no MFMAs, 4 no weight prefetch loads
};

// U layout: [cc][g = (oz+1)*3 + (oy+1)][p][e4][h][k][4], reduce channel c = cc*32 + h*16 + e (e = e4*4 + c4): the
// e4-th 16-byte load of a wave (lanes (h, k)) is two contiguous 512-byte runs
__host__ __device__ inline size_t uidx(int K, int cc, int g, int p, int h, int k, int e) {
    return (((((((size_t)cc * 9 + g) * 4 + p) * 4 + (e >> 2)) * 2 + h) * K + k) << 2) + (e & 3);
}

// w: torch Conv3d weight [K][C][3][3][3].  uf: conv forward (reduce C, produce K); ub: input gradient (reduce K,
// produce C, taps mirrored: the tap that reads dy at offset o is w[.., 1-o]).
__global__ void k_pack_wino(const float *__restrict__ w, float *__restrict__ uf, float *__restrict__ ub, int K, int C) {
    const long total = (long)9 * 4 * C * K;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // idx enumerates (g, p, c, k) with k fastest
    const int k = (int)(idx % K);
    long r = idx / K;
    const int c = (int)(r % C);
    r /= C;
    const int p = (int)(r & 3);
    const int g = (int)(r >> 2);
    const int gz = g / 3, gy = g % 3;
    const float *wp = w + (((size_t)k * C + c) * 27);
    if (uf) {
        const float g0 = wp[(gz * 3 + gy) * 3 + 0], g1 = wp[(gz * 3 + gy) * 3 + 1], g2 = wp[(gz * 3 + gy) * 3 + 2];
        const float u = p == 0 ? g0 : (p == 1 ? (g0 + g1 + g2) * 0.5f : (p == 2 ? (g0 - g1 + g2) * 0.5f : g2));
        uf[uidx(K, c >> 5, g, p, (c >> 4) & 1, k, c & 15)] = u;
    }
    if (ub) {
        const int mz = 2 - gz, my = 2 - gy;
        const float g0 = wp[(mz * 3 + my) * 3 + 2], g1 = wp[(mz * 3 + my) * 3 + 1], g2 = wp[(mz * 3 + my) * 3 + 0];
        const float u = p == 0 ? g0 : (p == 1 ? (g0 + g1 + g2) * 0.5f : (p == 2 ? (g0 - g1 + g2) * 0.5f : g2));
        ub[uidx(C, k >> 5, g, p, (k >> 4) & 1, c, k & 15)] = u;
    }
}

int pack_weight_wino2(const float *w, float *uf, float *ub, int K, int C, hipStream_t s);

// MVD_WINO: 0 = direct engines only, 1 = F(2,3) along W, 2 (default) = F(2x2,3x3) over H and W for the forward-type
// passes.  The weight gradient uses the F(2,3)-transposed kernel in modes 1 and 2.
int wino_mode() {
    static int m = -1;
    if (m < 0) m = getenv("MVD_WINO") ? atoi(getenv("MVD_WINO")) : 2;
    return m;
}
size_t wino_weight_elems(int C, int K) { return (size_t)(wino_mode() == 2 ? 48 : 36) * C * K; }

int pack_weight_wino(const float *w, float *uf, float *ub, int K, int C, hipStream_t s) {
    if (wino_mode() == 2) return pack_weight_wino2(w, uf, ub, K, C, s);
    const long total = (long)36 * C * K;
    hipLaunchKernelGGL(k_pack_wino, dim3(cdiv(total, 256)), dim3(256), 0, s, w, uf, ub, K, C);
    return check_launch("pack_weight_wino");
}

constexpr int WXS = 36;   // floats per halo slot (32 + 4 pad)
constexpr int WXR = 12;   // float4 per thread: 360 slots x 8 / 256 threads = 11.25
#ifdef MVD_WINO_ABLATE            // ablation build: MVD_WINO_DBG bit 0 skips the halo loads, bit 2 the weight loads
constexpr bool kAblate = true;
#else
constexpr bool kAblate = false;
#endif
constexpr int WXB = 6;    // staging batch (loads in flight per thread)

// Workgroup = 4 waves on one 4 x 4 x 8 voxel tile (64 pairs) x 32 output channels.  Wave w: M half = w & 1 (d-planes
// 2*(w&1), +1: 32 pairs), position pair ph = w >> 1: waves 0,1 accumulate the Winograd positions p = 0,1 (inputs
// d0,d1,d2), waves 2,3 the positions p = 2,3 (inputs d1,d2,d3) -- 288 MFMAs each, perfectly balanced, 32 accumulator
// registers per lane, so three workgroups (12 waves) share a CU.  The output transform needs one accumulator tile of
// the partner wave (w ^ 2): y[2q] = (m0 + m1) + m2 is finished by waves 0,1, y[2q+1] = (m1 - m2) - m3 by waves 2,3;
// the tiles cross through the (by then free) halo buffer.
__global__ __launch_bounds__(256, 3) void k_fwd_wino(const FwdGeom g, const WinoTile tg, const float *__restrict__ a1,
                                                     const float *__restrict__ a2, const float *__restrict__ u,
                                                     const float *__restrict__ bias, float *__restrict__ y1,
                                                     float *__restrict__ y2) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave & 1, ph = wave >> 1;
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= tg.nitems) return;  // whole workgroup
    unsigned r_ = (unsigned)item;
    const int kb = (int)(r_ % (unsigned)tg.nkb); r_ /= (unsigned)tg.nkb;
    const int tw_ = (int)(r_ % (unsigned)tg.ntw); r_ /= (unsigned)tg.ntw;
    const int th_ = (int)(r_ % (unsigned)tg.nth); r_ /= (unsigned)tg.nth;
    const int td_ = (int)(r_ % (unsigned)tg.ntd);
    const int n = (int)(r_ / (unsigned)tg.ntd);

    const int C = g.C1 + g.C2;
    const int nch = C >> 5;
    const int EHW = tg.EH * tg.EW;
    const int nx = tg.nslots * 8;
    const int od0 = td_ * 4, oh0 = th_ * 4, ow0 = tw_ * 8;
    const int iz0 = od0 - 1, iy0 = oh0 - 1, ix0 = ow0 - 1;

    // pair i of this wave: d-plane 2*mh + (i >> 4), row (i >> 2) & 3, pair i & 3 of the row; first slot = d0 + ph
    const int sbase = ((2 * mh + (i >> 4)) * tg.EH + ((i >> 2) & 3)) * tg.EW + 2 * (i & 3) + ph;
    const float *xlane = Xs + (size_t)sbase * WXS + h * 16;
    // B fragment of (chunk cc, group gi, position p): four 16-byte quarters of lane (h, k = kb*32 + i)
    const size_t ustep = (size_t)2 * tg.K * 16;  // floats between consecutive (group, position) blocks
    const size_t uq = (size_t)2 * tg.K * 4;      // floats between the four quarters of a fragment
    const float *ulane = u + (((size_t)h * tg.K + kb * 32 + i) << 2) + (size_t)(2 * ph) * ustep;

    f32x16 acc[2];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.f;

    for (int cc = 0; cc < nch; cc++) {
        const int c0 = cc * 32;
        const float *src;
        int Cs, cofs;
        if (c0 < g.C1) {
            src = a1; Cs = g.C1; cofs = c0;
        } else {
            src = a2; Cs = g.C2; cofs = c0 - g.C1;
        }
        const float *uc = ulane + (size_t)cc * 36 * ustep;
        float4 wb[2][4];  // this wave's two positions of the current group; refilled one position ahead
#pragma unroll
        for (int e = 0; e < 4; e++) wb[0][e] = *reinterpret_cast<const float4 *>(uc + e * uq);
        __syncthreads();  // every wave is done with the previous chunk's halo
        // the slot -> voxel index math is recomputed per chunk on purpose: hoisted out of the chunk loop it would sit
        // in ~30 registers across the MFMA loop (and spill to scratch = real HBM writes)
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        for (int base = 0; base < WXR; base += WXB) {
            float4 v[WXB];
#pragma unroll
            for (int q = 0; q < WXB; q++) {
                const int idx = (base + q) * 256 + tid_;
                v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < nx && !(kAblate && (tg.dbg & 1))) {
                    const int slot = idx >> 3;
                    const int ez = (slot * tg.magHW) >> 16, rem = slot - ez * EHW;
                    const int ey = (rem * tg.magW) >> 16, ex = rem - ey * tg.EW;
                    const int id = iz0 + ez, ih = iy0 + ey, iw = ix0 + ex;
                    if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                        v[q] = *reinterpret_cast<const float4 *>(
                            src + ((((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * Cs + cofs + (tid_ & 7) * 4);
                }
            }
#pragma unroll
            for (int q = 0; q < WXB; q++) {
                const int idx = (base + q) * 256 + tid_;
                if (idx < nx) *reinterpret_cast<float4 *>(Xs + (size_t)(idx >> 3) * WXS + (idx & 7) * 4) = v[q];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int gi = 0; gi < 9; gi++) {
            const int gz = gi / 3, gy = gi - gz * 3;
            const float4 *px = reinterpret_cast<const float4 *>(xlane + (size_t)((gz * tg.EH + gy) * tg.EW) * WXS);
            float4 s0[4], s1[4], s2[4];  // ph = 0: d0, d1, d2;  ph = 1: d1, d2, d3
            float4 vv[2][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s0[e] = px[e];
                s1[e] = px[(WXS / 4) + e];
                s2[e] = px[2 * (WXS / 4) + e];
            }
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {
                // next position's weights: (gi, 1) after (gi, 0), (gi + 1, 0) after (gi, 1); the chunk's last step
                // re-reads its own block (harmless)
                const int nxt = pp == 0 ? gi * 4 + 1 : (gi < 8 ? gi * 4 + 4 : gi * 4 + 1);
                if (!(kAblate && (tg.dbg & 4))) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        wb[pp ^ 1][e] = *reinterpret_cast<const float4 *>(uc + (size_t)nxt * ustep + e * uq);
                }
                // (vv is double-buffered over pp: rewriting an operand register right behind the MFMA that reads it stalls)
                if (ph == 0) {  // wave-uniform: p0 = d0 - d2, p1 = d1 + d2
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        vv[pp][e] = pp == 0 ? make_float4(s0[e].x - s2[e].x, s0[e].y - s2[e].y, s0[e].z - s2[e].z, s0[e].w - s2[e].w)
                                        : make_float4(s1[e].x + s2[e].x, s1[e].y + s2[e].y, s1[e].z + s2[e].z, s1[e].w + s2[e].w);
                } else {        // p2 = d2 - d1, p3 = d1 - d3
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        vv[pp][e] = pp == 0 ? make_float4(s1[e].x - s0[e].x, s1[e].y - s0[e].y, s1[e].z - s0[e].z, s1[e].w - s0[e].w)
                                        : make_float4(s0[e].x - s2[e].x, s0[e].y - s2[e].y, s0[e].z - s2[e].z, s0[e].w - s2[e].w);
                }
                if (!(kAblate && (tg.dbg & 2))) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float4 bq = wb[pp][e];
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[pp][e].x, bq.x, acc[pp], 0, 0, 0);
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[pp][e].y, bq.y, acc[pp], 0, 0, 0);
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[pp][e].z, bq.z, acc[pp], 0, 0, 0);
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[pp][e].w, bq.w, acc[pp], 0, 0, 0);
                    }
                }
            }
        }
    }
    // output transform: waves 0,1 hold (m0, m1) and finish y[2q] = (m0 + m1) + m2; waves 2,3 hold (m2, m3) and finish
    // y[2q+1] = (m1 - m2) - m3.  Each wave hands one tile to its partner through the halo buffer.
    __syncthreads();  // all MFMA operand reads of the halo are done
    {
        float *xo = Xs + (size_t)wave * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; r++) xo[r * 64] = ph == 0 ? acc[1][r] : acc[0][r];  // m1 or m2
    }
    __syncthreads();
    const float *xi = Xs + (size_t)(wave ^ 2) * 1024 + lane;
    const int k = kb * 32 + i;
    const float bv = settled(bias ? bias[k] : 0.f);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int pi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int od = od0 + 2 * mh + (pi >> 4), oh = oh0 + ((pi >> 2) & 3), ow = ow0 + 2 * (pi & 3) + ph;
        const float other = xi[r * 64];
        const float val = ph == 0 ? (acc[0][r] + acc[1][r]) + other + bv : (other - acc[0][r]) - acc[1][r] + bv;
        if (od < g.Do && oh < g.Ho && ow < g.Wo) {
            const size_t ov = (((size_t)n * g.Dy + od) * g.Hy + oh) * g.Wy + ow;
            if (k < g.K1)
                y1[ov * g.K1 + k] = val;
            else
                y2[ov * g.K2 + (k - g.K1)] = val;
        }
    }
}

// ================================================================================================ F(2x2, 3x3) over H and W
// Two-dimensional variant: for each filter plane dz, a 2x2 output quad reads a 4x4 input patch;
//     V = B^T d B (16 positions), M_ab[k] += V_ab[c] U_ab[c][k] with U = G g G^T, y = A^T M A
// -> 3 x 16 channel contractions per 4 outputs = 12 per output instead of 27 (direct) or 18 (F(2,3) along W only).
// Workgroup = 4 waves on a 4 x 4 x 8 voxel tile = 32 quads = ONE 32-row M tile; wave a owns row a of the 4x4 position
// grid (4 accumulator tiles).  Per step (plane gz, 4-channel quarter e): 8 ds_read_b128 (two patch rows x four
// columns), 32 VALU (row transform R = P[ra] +- P[rb], column transform V_b), 16 MFMAs on four independent chains,
// and the four 16-byte weight quarters of the next step fetched from L2.  The output transform runs the column half
// in registers (t0 = M_a0 + M_a1 + M_a2, t1 = M_a1 - M_a2 - M_a3), crosses the row half through the freed halo buffer
// and wave (yr, yc) writes output voxel (yr, yc) of every quad.
// U2 layout: [cc][gz][a][e4][b][h][k][4], reduce channel c = cc*32 + h*16 + e4*4 + c4
__host__ __device__ inline size_t u2idx(int K, int cc, int gz, int a, int b, int h, int k, int e) {
    return ((((((((size_t)cc * 3 + gz) * 4 + a) * 4 + (e >> 2)) * 4 + b) * 2 + h) * K + k) << 2) + (e & 3);
}

__device__ __forceinline__ void pack_wino2_item(const float *__restrict__ w, float *__restrict__ uf,
                                                float *__restrict__ ub, int K, int C, long idx) {
    const int k = (int)(idx % K);
    long r = idx / K;
    const int c = (int)(r % C);
    r /= C;
    const int b = (int)(r & 3), a = (int)((r >> 2) & 3), gz = (int)(r >> 4);
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const float *wp = w + (((size_t)k * C + c) * 27);
    if (uf) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) s += G[a][i] * G[b][j] * wp[(gz * 3 + i) * 3 + j];
        uf[u2idx(K, c >> 5, gz, a, b, (c >> 4) & 1, k, c & 15)] = s;
    }
    if (ub) {  // input gradient: reduce K, produce C, filter mirrored in all three axes
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) s += G[a][i] * G[b][j] * wp[((2 - gz) * 3 + (2 - i)) * 3 + (2 - j)];
        ub[u2idx(C, k >> 5, gz, a, b, (k >> 4) & 1, c, k & 15)] = s;
    }
}

__global__ void k_pack_wino2(const float *__restrict__ w, float *__restrict__ uf, float *__restrict__ ub, int K, int C) {
    const long total = (long)3 * 16 * C * K;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    pack_wino2_item(w, uf, ub, K, C, idx);
}

// ---------------------------------------------------------------------------------------------- one launch for all layers
// Every conv weight of the network re-packed in ONE launch after the optimizer step (the per-layer pack kernels are
// launch-bound: 39 launches of 17-26 us per step).  Jobs travel in the kernel argument; a workgroup finds its job by a
// scalar scan over the (at most PACK_MAX_JOBS) block ranges.
struct PackJob {
    const float *w;
    float *wf, *wb, *uf, *ub;
    int K, C, T, transposed;
    unsigned blk_begin;
    int tiled;  // 1: 3x3x3 conv with C % 32 == 0 and K % 32 == 0 -> one workgroup per 16 k x 16 c tile (through LDS)
};
constexpr int PACK_MAX_JOBS = 48;
struct PackBatch {
    int n;
    PackJob j[PACK_MAX_JOBS];
};

// Tiled jobs: the workgroup reads its 16 x 16 x 27 block of the torch tensor as 16 contiguous 1.7 KB runs into LDS and
// writes every packed layout in contiguous 256-byte .. 1 KB runs (the element-wise kernels read with a C*27*4-byte
// lane stride).  Arithmetic and summation order are those of k_pack_weight / pack_wino2_item: bit-identical outputs.
__global__ __launch_bounds__(256) void k_pack_batch(const PackBatch pb) {
    __shared__ float tile[16 * 432];  // [k 16][c 16][t 27]
    int jid = 0;
    for (int q = 1; q < pb.n; q++)
        if (blockIdx.x >= pb.j[q].blk_begin) jid = q;
    const PackJob &J = pb.j[jid];
    const unsigned lb = blockIdx.x - J.blk_begin;
    const int tid = threadIdx.x;
    if (!J.tiled) {
        const long i = (long)lb * 256 + tid;
        if (i >= (long)J.K * J.C * J.T) return;
        const int k = (int)(i % J.K), c = (int)((i / J.K) % J.C), t = (int)(i / ((long)J.K * J.C));  // wf order [t][c][k]
        const float v = J.transposed ? J.w[((size_t)c * J.K + k) * J.T + t] : J.w[((size_t)k * J.C + c) * J.T + t];
        if (J.wf) J.wf[widx(wl_ck(J.C), J.T, J.C, J.K, t, c, k)] = v;
        if (J.wb) J.wb[widx(wl_ck(J.K), J.T, J.K, J.C, t, k, c)] = v;
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
    const int hi = tid >> 4, lo = tid & 15;
    if (J.wf) {  // [cc][t][h][k][16 e], c = c0 + lo, k = k0 + hi
        const int cc = c0 >> 5, h = (c0 >> 4) & 1;
        for (int t = 0; t < 27; t++)
            J.wf[((((size_t)cc * 27 + t) * 2 + h) * K + k0 + hi) * 16 + lo] = tile[hi * 432 + lo * 27 + t];
    }
    if (J.wb) {  // [kk][t][hk][c][16 ek], k = k0 + lo, c = c0 + hi
        const int kk = k0 >> 5, hk = (k0 >> 4) & 1;
        for (int t = 0; t < 27; t++)
            J.wb[((((size_t)kk * 27 + t) * 2 + hk) * C + c0 + hi) * 16 + lo] = tile[lo * 432 + hi * 27 + t];
    }
    if (J.uf || J.ub) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const int q4 = tid >> 6, mid = (tid >> 2) & 15, l4 = tid & 3;
        // uf: c = c0 + q4 * 4 + l4, k = k0 + mid;  ub: k = k0 + q4 * 4 + l4, c = c0 + mid
        const float *wu = tile + mid * 432 + (q4 * 4 + l4) * 27;
        const float *wv = tile + (q4 * 4 + l4) * 432 + mid * 27;
        for (int pos = 0; pos < 48; pos++) {
            const int b = pos & 3, a_ = (pos >> 2) & 3, gz = pos >> 4;
            if (J.uf) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) s += G[a_][i] * G[b][j] * wu[(gz * 3 + i) * 3 + j];
                const int c = c0 + q4 * 4 + l4;
                J.uf[u2idx(K, c >> 5, gz, a_, b, (c >> 4) & 1, k0 + mid, c & 15)] = s;
            }
            if (J.ub) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) s += G[a_][i] * G[b][j] * wv[((2 - gz) * 3 + (2 - i)) * 3 + (2 - j)];
                const int k = k0 + q4 * 4 + l4;
                J.ub[u2idx(C, k >> 5, gz, a_, b, (k >> 4) & 1, c0 + mid, k & 15)] = s;
            }
        }
    }
}

// jobs: host arrays of n entries.  uf / ub entries may be null (no Winograd pack for that layer).
int pack_weights_batch(int n, const float *const *w, float *const *wf, float *const *wb, float *const *uf, float *const *ub,
                       const int *K, const int *C, const int *T, const int *transposed, hipStream_t s) {
    int done = 0;
    while (done < n) {
        PackBatch pb;
        memset(&pb, 0, sizeof(pb));
        unsigned blocks = 0;
        int m = 0;
        for (; m < PACK_MAX_JOBS && done + m < n; m++) {
            const int q = done + m;
            PackJob &J = pb.j[m];
            J.w = w[q]; J.wf = wf[q]; J.wb = wb[q]; J.uf = uf[q]; J.ub = ub[q];
            J.K = K[q]; J.C = C[q]; J.T = T[q]; J.transposed = transposed[q];
            J.blk_begin = blocks;
            J.tiled = (J.T == 27 && !J.transposed && J.K % 32 == 0 && J.C % 32 == 0) ? 1 : 0;
            blocks += J.tiled ? (unsigned)((J.K / 16) * (J.C / 16)) : (unsigned)cdiv((long)J.K * J.C * J.T, 256);
        }
        pb.n = m;
        if (blocks > 0) hipLaunchKernelGGL(k_pack_batch, dim3(blocks), dim3(256), 0, s, pb);
        if (check_launch("pack_weights_batch")) return 1;
        done += m;
    }
    return 0;
}

int pack_weight_wino2(const float *w, float *uf, float *ub, int K, int C, hipStream_t s) {
    const long total = (long)48 * C * K;
    hipLaunchKernelGGL(k_pack_wino2, dim3(cdiv(total, 256)), dim3(256), 0, s, w, uf, ub, K, C);
    return check_launch("pack_weight_wino2");
}

__device__ inline float4 f4_axpy(float s, const float4 b, const float4 a) {  // a + s*b
    return make_float4(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z), fmaf(s, b.w, a.w));
}
__device__ inline float4 f4_sub(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ inline float4 f4_add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Packed fp32 VALU through inline asm.  On gfx950 every fp32 VALU instruction takes ~3 cycles away from the fp32 MFMA
// pipe of its SIMD (tools/probes/valu_mix_probe.hip: they do not overlap, not even across waves), so the input transform
// is written with v_pk_fma_f32 / v_pk_add_f32 on the register pairs ds_read_b128 delivers -- 16 instead of 32 VALU
// instructions per 16 MFMAs.  Plain <2 x float> arithmetic does not survive: the backend's pre-emit peephole unpacks
// packed F32 instructions it finds behind an MFMA.
// HAZARD: a VALU write needs 2 wait states before an MFMA reads the register as SrcA/B, and the compiler's hazard
// recognizer does not look inside inline asm -- hence the trailing s_nop 1 of every block whose results feed MFMAs.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// One half (two of the four channels of a float4) of the F(2x2,3x3) input transform of a step:
//   R_c = a_c + sg * b_c (c = 0..3: the four patch columns);  V0 = R0 - R2, V1 = R1 + R2, V2 = R2 - R1, V3 = R1 - R3
// in place: a0 -> V0, a2 -> V2, a3 -> V3; V1 is returned (a1 is consumed as scratch for R1).
__device__ __forceinline__ v2f wino2_input_transform(v2f sg, v2f &a0, v2f a1, v2f &a2, v2f &a3, v2f b0, v2f b1, v2f b2,
                                                     v2f b3) {
    v2f t;
    asm("v_pk_fma_f32 %0, %5, %6, %0\n\t"
        "v_pk_fma_f32 %1, %5, %7, %1\n\t"
        "v_pk_fma_f32 %2, %5, %8, %2\n\t"
        "v_pk_fma_f32 %3, %5, %9, %3\n\t"
        "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"  // V0 = R0 - R2
        "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"  // V3 = R1 - R3
        "v_pk_add_f32 %4, %1, %2\n\t"                            // V1 = R1 + R2
        "v_pk_add_f32 %2, %2, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"  // V2 = R2 - R1
        "s_nop 1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t)
        : "v"(sg), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    return t;
}

#ifndef MVD_W2B
#define MVD_W2B 12
#endif
constexpr int W2B = MVD_W2B;  // staging loads in flight per thread before the LDS stores
constexpr int W2EH = 6, W2EW = 10, W2EHW = 60;  // halo of the 4 x 4 x 8 tile: 6 x 6 x 10 slots
__device__ __forceinline__ constexpr int w2_coff(int c) { return (c >> 1) + 5 * (c & 1); }  // slot offset of patch column c

#ifndef MVD_WINO_DBG
#define MVD_WINO_DBG 0
#endif
#if (MVD_WINO_DBG & 64)  // diagnostic build only (tools/stamps_wino.py): s_memtime stamps of one wave per chunk
__device__ long long g_wino_stamps[64 * 8];
extern "C" int mvd_debug_wino_stamps(long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_stamps), sizeof(long long) * 64 * 8) == hipSuccess ? 0 : 1;
}
#define MVD_WS(K) { if (stamp_on && nst < 60) g_wino_stamps[nst * 8 + (K)] = __builtin_amdgcn_s_memtime(); }
#else
#define MVD_WS(K)
#endif
__global__ __launch_bounds__(256, 3) void k_fwd_wino2(const FwdGeom g, const WinoTile tg, const float *__restrict__ a1,
                                                      const float *__restrict__ a2, const float *__restrict__ u,
                                                      const float *__restrict__ bias, float *__restrict__ y1,
                                                      float *__restrict__ y2, float *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int per_xcd = (tg.nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= tg.nitems) return;  // whole workgroup
    // the divisions run on the VALU (float reciprocal); readfirstlane moves the wave-uniform results back to scalar
    // registers so that every address derived from them is scalar arithmetic
    unsigned r_ = (unsigned)item;
    const int kb = __builtin_amdgcn_readfirstlane((int)(r_ % (unsigned)tg.nkb)); r_ /= (unsigned)tg.nkb;
    const int tw_ = __builtin_amdgcn_readfirstlane((int)(r_ % (unsigned)tg.ntw)); r_ /= (unsigned)tg.ntw;
    const int th_ = __builtin_amdgcn_readfirstlane((int)(r_ % (unsigned)tg.nth)); r_ /= (unsigned)tg.nth;
    const int td_ = __builtin_amdgcn_readfirstlane((int)(r_ % (unsigned)tg.ntd));
    const int n = __builtin_amdgcn_readfirstlane((int)(r_ / (unsigned)tg.ntd));

    const int C = g.C1 + g.C2;
    const int nch = C >> 5;
    const int od0 = td_ * 4, oh0 = th_ * 4, ow0 = tw_ * 8;
    const int iz0 = od0 - 1, iy0 = oh0 - 1, ix0 = ow0 - 1;

    // position row a = wave: R = P[ra] + sg * P[rb]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const v2f sg2 = {sg, sg};
    // LDS layout: each 10-slot halo row holds the even x first (x -> (x >> 1) + 5 * (x & 1)), and quad i = (plane i & 3,
    // column (i >> 2) & 3, quad row i >> 4).  With the 144-byte slot stride this is the assignment for which all eight
    // ds_read_b128 of a step are bank-conflict free (4.5 instead of 12.9 LDS cycles per read: tools/probes/
    // lds_pattern_probe.hip sweeps every bit assignment).  The patch columns c = 0..3 of a quad (x = 2 col + c) sit at
    // slot offsets 0, 5, 1, 6 from the quad's origin.
    const int sbase = ((i & 3) * W2EH + 2 * (i >> 4)) * W2EW + ((i >> 2) & 3);
    const v4f *xa4 = reinterpret_cast<const v4f *>(Xs + (size_t)(sbase + ra * W2EW) * WXS + h * 16);
    const v4f *xb4 = reinterpret_cast<const v4f *>(Xs + (size_t)(sbase + rb * W2EW) * WXS + h * 16);
    // weights: step (cc, gz, e) -> block ((cc*3 + gz)*4 + a)*4 + e of 4 quarters (b) x [h][k][4]; the block address is
    // wave-uniform (scalar registers), the lane adds a 32-bit offset
    const unsigned uq = 2u * tg.K * 4;  // floats between the b quarters
    const unsigned ustep = 4 * uq;      // floats between consecutive e steps
    const unsigned ulane = ((unsigned)(h * tg.K + kb * 32 + i)) << 4;  // BYTES: scalar base + 32-bit lane offset loads
    const float *uwave = u + (size_t)wave * 4 * ustep;

    // halo staging without divisions: 240 threads cover three (z, y) rows of 10 slots x 8 float4 per pass; pass q holds
    // rows 3q .. 3q+2, i.e. plane q >> 1 and y = r3 + 3 * (q & 1).  Everything lane-dependent is computed once per item.
    const bool st_act = tid < 240;
    const int r3 = tid / 80, rem = tid - r3 * 80;
    const int sx = rem >> 3, part = rem & 7;
    const int iw = ix0 + sx, ihA = iy0 + r3, ihB = ihA + 3;
    const bool okw = st_act && iw >= 0 && iw < g.Wi;
    const bool okA = okw && ihA >= 0 && ihA < g.Hi, okB = okw && ihB >= 0 && ihB < g.Hi;
    v4f *lds_st = reinterpret_cast<v4f *>(Xs + (size_t)(r3 * W2EW + (sx >> 1) + 5 * (sx & 1)) * WXS + part * 4);

    const int k = kb * 32 + i;
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[b][r] = 0.f;

#if (MVD_WINO_DBG & 64)
    const bool stamp_on = item == 300 && wave == 0 && lane == 0;
    int nst = 0;
#endif
    for (int cc = 0; cc < nch; cc++) {
        MVD_WS(0)
        const int c0 = cc * 32;
        const float *src;
        int Cs, cofs;
        if (c0 < g.C1) {
            src = a1; Cs = g.C1; cofs = c0;
        } else {
            src = a2; Cs = g.C2; cofs = c0 - g.C1;
        }
        const float *uc = uwave + (size_t)cc * 3 * 16 * ustep;  // 3 planes x 4 position rows x 4 steps
        float4 wb[2][4];
#pragma unroll
        for (int b = 0; b < 4; b++)
            wb[0][b] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(uc + b * uq) + ulane);
        MVD_WS(1)
        __syncthreads();  // every wave is done with the previous chunk's halo
        MVD_WS(2)
        {
            // byte offsets inside a (n, z) plane; only used when okA / okB (host checks Hi * Wi * Cs * 4 < 2^31)
            const unsigned offA = (unsigned)((ihA * g.Wi + iw) * Cs + cofs + part * 4) << 2;
            const unsigned offB = offA + ((unsigned)(3 * g.Wi * Cs) << 2);
#pragma unroll
            for (int base = 0; base < WXR; base += W2B) {
                v4f v[W2B];
#pragma unroll
                for (int q = 0; q < W2B; q++) {
                    const int pq = base + q, id = iz0 + (pq >> 1);  // plane: wave-uniform
                    const float *plane = src + ((size_t)n * g.Di + id) * g.Hi * g.Wi * Cs;
                    const bool ok = ((pq & 1) ? okB : okA) && id >= 0 && id < g.Di && !(kAblate && (tg.dbg & 1));
                    v[q] = v4f{0.f, 0.f, 0.f, 0.f};
                    if (ok) {
                        unsigned o = (pq & 1) ? offB : offA;
                        asm("" : "+v"(o));  // scalar plane base + 32-bit lane offset (see the weight loads)
                        v[q] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(plane) + o);
                    }
                }
#pragma unroll
                for (int q = 0; q < W2B; q++) {
                    const int pq = base + q;
                    if (st_act) lds_st[((pq >> 1) * W2EHW + 3 * (pq & 1) * W2EW) * (WXS / 4)] = v[q];
                }
            }
        }
        MVD_WS(3)
        __syncthreads();
        MVD_WS(4)
        // 12 steps (plane gz, channel quarter e).  The patch of step s + 1 is read from LDS before the MFMAs of step s
        // are issued, so its latency hides behind them: the P[ra] rows go to a second register set (parity e & 1), the
        // P[rb] rows back into the registers the transform of step s has just released.
        v4f pa[2][4], pb[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            pa[0][c] = xa4[w2_coff(c) * (WXS / 4)];
            pb[c] = xb4[w2_coff(c) * (WXS / 4)];
        }
#pragma unroll 1
        for (int gz = 0; gz < 3; gz++) {
            const int po = gz * W2EHW * (WXS / 4);  // float4 offset of the plane
#pragma unroll
            for (int e = 0; e < 4; e++) {
                {   // next step's weights: (gz, e + 1), (gz + 1, 0); the chunk's last step re-reads its own
                    const int nxt = (gz == 2 && e == 3) ? gz * 16 + e : (e == 3 ? (gz + 1) * 16 : gz * 16 + e + 1);
                    if (!(kAblate && (tg.dbg & 4))) {
                        const float *un = uc + (size_t)nxt * ustep;  // wave-uniform
                        // opaque 32-bit copy: the zero-extension stays in this block, so the loads select the
                        // scalar-base + 32-bit-offset form (no 64-bit VALU add per load)
                        unsigned ul = ulane;
                        asm("" : "+v"(ul));
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            wb[(e + 1) & 1][b] =
                                *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(un + b * uq) + ul);
                    }
                }
                // next step's patch (the last step of the chunk reads plane 3 = a valid, unused halo plane)
                const int pn = (e == 3) ? po + W2EHW * (WXS / 4) : po + e + 1;
#pragma unroll
                for (int c = 0; c < 4; c++) pa[(e + 1) & 1][c] = xa4[pn + w2_coff(c) * (WXS / 4)];
                __builtin_amdgcn_sched_barrier(0);
                v2f Vl[4], Vh[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    Vl[c] = pa[e & 1][c].xy;
                    Vh[c] = pa[e & 1][c].zw;
                }
                Vl[1] = wino2_input_transform(sg2, Vl[0], Vl[1], Vl[2], Vl[3], pb[0].xy, pb[1].xy, pb[2].xy, pb[3].xy);
                Vh[1] = wino2_input_transform(sg2, Vh[0], Vh[1], Vh[2], Vh[3], pb[0].zw, pb[1].zw, pb[2].zw, pb[3].zw);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; c++) pb[c] = xb4[pn + w2_coff(c) * (WXS / 4)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vl[b].x, wb[e & 1][b].x, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vl[b].y, wb[e & 1][b].y, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vh[b].x, wb[e & 1][b].z, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vh[b].y, wb[e & 1][b].w, acc[b], 0, 0, 0);
            }
        }
        MVD_WS(5)
#if (MVD_WINO_DBG & 64)
        nst++;
#endif
    }
    // the bias, settled before the first store (common.h): the stores of the tile then pipeline instead of running as
    // sixteen serialised write round trips
    const float bv = settled(bias ? bias[k] : 0.f);
    // output transform.  Column half in registers, row half across the four waves through LDS.
    f32x16 t0, t1;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        t0[r] = (acc[0][r] + acc[1][r]) + acc[2][r];
        t1[r] = (acc[1][r] - acc[2][r]) - acc[3][r];
    }
    MVD_WS(0)
    __syncthreads();  // all MFMA operand reads of the halo are done
    MVD_WS(1)
    {
        float *xo = Xs + (size_t)wave * 2048 + lane;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            xo[r * 64] = t0[r];
            xo[1024 + r * 64] = t1[r];
        }
    }
    MVD_WS(2)
    __syncthreads();
    MVD_WS(3)
    // wave (yr, yc) = (wave >> 1, wave & 1) finishes output voxel (yr, yc) of each quad from column tile t_yc of
    // position rows {0,1,2} (yr = 0: sum) or {1,2,3} (yr = 1: t[1] - t[2] - t[3]).  Accumulator row r of lane half h is
    // quad q = (r & 3) + 8 * (r >> 2) + 4 * h = (plane r & 3, column 2 * ((r >> 2) & 1) + h, quad row r >> 3): plane,
    // quad row and the column's upper bit are wave-uniform, the lane half moves the output by two voxels along W.
    const int yr = wave >> 1, yc = wave & 1;
    const float *xt = Xs + (size_t)yc * 1024 + lane;
    const int owl = ow0 + yc + 2 * h;                          // + 4 * ((r >> 2) & 1)
    const bool okw0 = owl < g.Wo, okw1 = owl + 4 < g.Wo;
    float ssum = 0.f, ssq = 0.f;  // InstanceNorm statistics of this tile (optional epilogue)
    // all 48 exchange reads first (unconditional: one LDS latency instead of sixteen behind the store predicates)
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float ta = xt[(size_t)(yr + 0) * 2048 + r * 64], tb = xt[(size_t)(yr + 1) * 2048 + r * 64],
                    tc = xt[(size_t)(yr + 2) * 2048 + r * 64];
        vals[r] = (yr == 0 ? (ta + tb) + tc : (ta - tb) - tc) + bv;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g.K2 == 0 || g.K1 == g.K2) {
        // one voxel stride for every lane: per-lane column pointer (once) + a scalar voxel offset per r
        const int Ks = g.K1;
        float *ylane = (k < g.K1 ? y1 + k : y2 + (k - g.K1)) + (size_t)(2 * h) * Ks;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float val = vals[r];
            const int od = od0 + (r & 3), oh = oh0 + 2 * (r >> 3) + yr;   // wave-uniform
            const int owu = ow0 + yc + 4 * ((r >> 2) & 1);                // wave-uniform part of ow
            const size_t uo = ((((size_t)n * g.Dy + od) * g.Hy + oh) * g.Wy + owu) * Ks;
            if (od < g.Do && oh < g.Ho && (((r >> 2) & 1) ? okw1 : okw0) && !(kAblate && (tg.dbg & 8) && val != 12345.f)) {
                ylane[uo] = val;
                ssum += val;
                ssq += val * val;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float val = vals[r];
            const int od = od0 + (r & 3), oh = oh0 + 2 * (r >> 3) + yr;
            const int ow = owl + 4 * ((r >> 2) & 1);
            if (od < g.Do && oh < g.Ho && ow < g.Wo) {
                const size_t ov = (((size_t)n * g.Dy + od) * g.Hy + oh) * g.Wy + ow;
                if (k < g.K1)
                    y1[ov * g.K1 + k] = val;
                else
                    y2[ov * g.K2 + (k - g.K1)] = val;
            }
        }
    }
    MVD_WS(6)
    if (stats != nullptr) {  // block-uniform
        // sum x, sum x^2 over the tile's (up to 128) voxels per output channel, in a fixed order: lane halves, then waves
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        float *st = Xs + 8192;  // behind the 4 x 2048-float exchange area
        if (h == 0) {
            st[(wave * 32 + i) * 2 + 0] = ssum;
            st[(wave * 32 + i) * 2 + 1] = ssq;
        }
        __syncthreads();
        if (wave == 0 && h == 0) {
            const float a = (st[i * 2] + st[(32 + i) * 2]) + (st[(64 + i) * 2] + st[(96 + i) * 2]);
            const float q = (st[i * 2 + 1] + st[(32 + i) * 2 + 1]) + (st[(64 + i) * 2 + 1] + st[(96 + i) * 2 + 1]);
            const int tile = (td_ * tg.nth + th_) * tg.ntw + tw_;
            float *o = stats + (((size_t)n * (tg.ntd * tg.nth * tg.ntw) + tile) * tg.K + k) * 2;
            o[0] = a;
            o[1] = q;
        }
    }
}

// returns -1 when the problem is not a plain 3x3x3 stride-1 gather with 32-multiple channels (caller falls back)
int fwd_wino(const FwdGeom &g, const float *a1, const float *a2, const float *u, const float *bias, float *y1, float *y2,
             hipStream_t s, float *stats, int *stats_done) {
    if (stats_done) *stats_done = 0;
    const int C = g.C1 + g.C2, K = g.K1 + g.K2;
    if (!u || g.ntaps != 27 || g.T != 27) return -1;
    if (C % 32 || g.C1 % 32 || g.C2 % 32 || K % 32 || g.K1 % 32 || g.K2 % 32) return -1;
    for (int a = 0; a < 3; a++)
        if (g.sa[a] != 1 || g.so[a] != 1 || g.oo[a] != 0) return -1;
    if (g.Dy != g.Do || g.Hy != g.Ho || g.Wy != g.Wo) return -1;
    if (((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)u) & 15) return -1;
    // the 27 taps must be the full 3x3x3 stencil, either in filter order (forward) or mirrored (input gradient); the
    // caller passes the matching U (uf / ub of mvd_pack_weight_wino)
    bool plain = true, mirrored = true;
    for (int t = 0; t < 27; t++) {
        const int oz = g.off[t][0], oy = g.off[t][1], ox = g.off[t][2];
        if (oz < -1 || oz > 1 || oy < -1 || oy > 1 || ox < -1 || ox > 1) return -1;
        const int pos = ((oz + 1) * 3 + (oy + 1)) * 3 + (ox + 1);
        if (g.wt[t] != pos) plain = false;
        if (g.wt[t] != 26 - pos) mirrored = false;
    }
    if (!plain && !mirrored) return -1;
    WinoTile tg;
    memset(&tg, 0, sizeof(tg));
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("MVD_WINO_DBG") ? atoi(getenv("MVD_WINO_DBG")) : 0;
    tg.dbg = dbg;
    tg.EH = 6; tg.EW = 10; tg.nslots = 360;
    auto magic = [](int d, int nmax) -> int {
        int m = (1 << 16) / d + 1;
        for (int n = 0; n < nmax; n++)
            if (((n * m) >> 16) != n / d) return -1;
        return m;
    };
    tg.magHW = magic(tg.EH * tg.EW, WXR * 32);
    tg.magW = magic(tg.EW, tg.EH * tg.EW);
    if (tg.magHW < 0 || tg.magW < 0) return -1;
    tg.ntd = (g.Do + 3) / 4;
    tg.nth = (g.Ho + 3) / 4;
    tg.ntw = (g.Wo + 7) / 8;
    tg.nkb = K / 32;
    tg.K = K;
    const long nitems = (long)g.N * tg.ntd * tg.nth * tg.ntw * tg.nkb;
    if (nitems > (1L << 30)) return -1;
    tg.nitems = (int)nitems;
    const size_t lds = (size_t)tg.nslots * WXS * sizeof(float);
    const unsigned grid = (unsigned)(((nitems + 7) / 8) * 8);
    if (wino_mode() == 2) {
        float *st = (stats && stats_done && g.K2 == 0) ? stats : nullptr;
        hipLaunchKernelGGL(k_fwd_wino2, dim3(grid), dim3(256), lds, s, g, tg, a1, a2, u, bias, y1, y2, st);
        if (st) *stats_done = 1;
    } else
        hipLaunchKernelGGL(k_fwd_wino, dim3(grid), dim3(256), lds, s, g, tg, a1, a2, u, bias, y1, y2);
    return check_launch("conv fwd (winograd)");
}

}  // namespace mvd
