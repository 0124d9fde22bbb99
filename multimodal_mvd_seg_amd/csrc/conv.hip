// Conv3d / ConvTranspose3d entry points: geometry set-up, engine dispatch (MFMA implicit GEMM when the shape allows,
// scalar gather kernels otherwise) and the scalar engines themselves.
#include "common.h"
#include "conv_geom.h"

namespace mvd {

static int g_engine_mode = 0;  // 0 auto, 1 scalar only
// Winograd F(2,3) engine switches (debug): MVD_WINO=0 disables it, MVD_WINO_MIN overrides the minimum number of
// 128-voxel x 32-channel work items below which the direct engines (which can split the reduction) are used
#define g_wino_off (wino_mode() == 0)
// MVD_TRANSP=0: transposed convs through the gathered-tap engines (debug / A-B)
static const int g_transp_off = getenv("MVD_TRANSP") ? (atoi(getenv("MVD_TRANSP")) == 0) : 0;
static long g_wino_min_items = getenv("MVD_WINO_MIN") ? atol(getenv("MVD_WINO_MIN")) : 256;

// =============================================================================================== scalar forward-type
// one thread per (n, o, k); k fastest so weight reads and output writes are coalesced and A is a wave broadcast
__global__ void k_fwd_scalar(FwdGeom g, const float *__restrict__ a1, const float *__restrict__ a2,
                             const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ y1,
                             float *__restrict__ y2) {
    const int K = g.K1 + g.K2, C = g.C1 + g.C2;
    const int CK = wl_ck(C);
    const long total = (long)g.N * g.Do * g.Ho * g.Wo * K;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int k = (int)(idx % K);
        long r = idx / K;
        int ow = (int)(r % g.Wo);
        r /= g.Wo;
        int oh = (int)(r % g.Ho);
        r /= g.Ho;
        int od = (int)(r % g.Do);
        int n = (int)(r / g.Do);
        float acc = bias ? bias[k] : 0.f;
        for (int t = 0; t < g.ntaps; t++) {
            int id = od * g.sa[0] + g.off[t][0], ih = oh * g.sa[1] + g.off[t][1], iw = ow * g.sa[2] + g.off[t][2];
            if (id < 0 || id >= g.Di || ih < 0 || ih >= g.Hi || iw < 0 || iw >= g.Wi) continue;
            size_t vox = (((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw;
            const int tw = g.wt[t];
            const float *p1 = a1 + vox * g.C1;
            for (int c = 0; c < g.C1; c++) acc += p1[c] * w[widx(CK, g.T, C, K, tw, c, k)];
            if (g.C2) {
                const float *p2 = a2 + vox * g.C2;
                for (int c = 0; c < g.C2; c++) acc += p2[c] * w[widx(CK, g.T, C, K, tw, g.C1 + c, k)];
            }
        }
        size_t ov = (((size_t)n * g.Dy + (od * g.so[0] + g.oo[0])) * g.Hy + (oh * g.so[1] + g.oo[1])) * g.Wy +
                    (ow * g.so[2] + g.oo[2]);
        if (k < g.K1)
            y1[ov * g.K1 + k] = acc;
        else
            y2[ov * g.K2 + (k - g.K1)] = acc;
    }
}

int fwd_scalar(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1,
               float *y2, hipStream_t s) {
    long total = (long)g.N * g.Do * g.Ho * g.Wo * (g.K1 + g.K2);
    if (total <= 0) return 0;
    long blocks = cdiv(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_fwd_scalar, dim3(blocks), dim3(256), 0, s, g, a1, a2, w, bias, y1, y2);
    return check_launch("conv fwd (scalar)");
}

// =============================================================================================== scalar wgrad-type
// grid (ceil(ntaps*C*K/256), nsplit): each thread owns one (t,c,k) and a slice of the voxels; fp64 partials,
// fixed-order second stage.
__global__ void k_wgrad_scalar(WgradGeom g, const float *__restrict__ a1, const float *__restrict__ a2,
                               const float *__restrict__ b, double *__restrict__ partial, long chunk) {
    const int C = g.C1 + g.C2, K = g.K;
    const long per = (long)g.ntaps * C * K;
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= per) return;
    const int k = (int)(j % K);
    const int c = (int)((j / K) % C);
    const int t = (int)(j / ((long)K * C));
    const long total = (long)g.N * g.Do * g.Ho * g.Wo;
    const long v0 = (long)blockIdx.y * chunk;
    long v1 = v0 + chunk;
    if (v1 > total) v1 = total;
    double acc = 0.0;
    for (long v = v0; v < v1; v++) {
        long r = v;
        int ow = (int)(r % g.Wo);
        r /= g.Wo;
        int oh = (int)(r % g.Ho);
        r /= g.Ho;
        int od = (int)(r % g.Do);
        int n = (int)(r / g.Do);
        int id = od * g.sa[0] + g.off[t][0], ih = oh * g.sa[1] + g.off[t][1], iw = ow * g.sa[2] + g.off[t][2];
        if (id < 0 || id >= g.Di || ih < 0 || ih >= g.Hi || iw < 0 || iw >= g.Wi) continue;
        int bd = od * g.sb[0] + g.ob[t][0], bh = oh * g.sb[1] + g.ob[t][1], bw = ow * g.sb[2] + g.ob[t][2];
        if (bd < 0 || bd >= g.Db || bh < 0 || bh >= g.Hb || bw < 0 || bw >= g.Wb) continue;
        size_t va = (((size_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw;
        size_t vb = (((size_t)n * g.Db + bd) * g.Hb + bh) * g.Wb + bw;
        float av = c < g.C1 ? a1[va * g.C1 + c] : a2[va * g.C2 + (c - g.C1)];
        acc += (double)(av * b[vb * K + k]);
    }
    partial[(size_t)blockIdx.y * per + j] = acc;
}

// dw[torch layout] = sum_split partial[split][t][c][k]
__global__ void k_wgrad_reduce_d(WgradGeom g, const double *__restrict__ partial, float *__restrict__ dw, int nsplit) {
    const int C = g.C1 + g.C2, K = g.K;
    const long per = (long)g.ntaps * C * K;
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= per) return;
    double s = 0;
    for (int b = 0; b < nsplit; b++) s += partial[(size_t)b * per + j];
    const int k = (int)(j % K);
    const int c = (int)((j / K) % C);
    const int t = g.wt[(int)(j / ((long)K * C))];
    size_t o = g.transposed_out ? ((size_t)c * K + k) * g.T + t : ((size_t)k * C + c) * g.T + t;
    dw[o] = (float)s;
}

static int wgrad_scalar_split(const WgradGeom &g, long *chunk) {
    long total = (long)g.N * g.Do * g.Ho * g.Wo;
    long per = (long)g.ntaps * (g.C1 + g.C2) * g.K;
    long nsplit = cdiv(total, 512);
    long cap = (64L << 20) / (per * 8);  // keep the partial buffer <= 64 MiB
    if (cap < 1) cap = 1;
    if (nsplit > cap) nsplit = cap;
    if (nsplit > 256) nsplit = 256;
    if (nsplit < 1) nsplit = 1;
    *chunk = cdiv(total, nsplit);
    return (int)cdiv(total, *chunk);
}

size_t wgrad_scalar_ws(const WgradGeom &g) {
    long chunk;
    int ns = wgrad_scalar_split(g, &chunk);
    return (size_t)ns * g.ntaps * (g.C1 + g.C2) * g.K * sizeof(double) + 256;
}

int wgrad_scalar(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws,
                 size_t ws_bytes, hipStream_t s) {
    long chunk;
    int ns = wgrad_scalar_split(g, &chunk);
    MVD_REQUIRE(ws_bytes >= wgrad_scalar_ws(g), "conv wgrad (scalar): workspace too small");
    long per = (long)g.ntaps * (g.C1 + g.C2) * g.K;
    double *partial = reinterpret_cast<double *>(ws);
    hipLaunchKernelGGL(k_wgrad_scalar, dim3(cdiv(per, 256), ns), dim3(256), 0, s, g, a1, a2, b, partial, chunk);
    if (check_launch("conv wgrad (scalar)")) return 1;
    hipLaunchKernelGGL(k_wgrad_reduce_d, dim3(cdiv(per, 256)), dim3(256), 0, s, g, partial, dw, ns);
    return check_launch("conv wgrad reduce");
}

// =============================================================================================== bias gradient
// dbias[k] = sum over rows of dy[rows][K]
__global__ void k_colsum(const float *__restrict__ x, double *__restrict__ partial, long rows, int K, long chunk) {
    __shared__ double sm[256];
    const int t = threadIdx.x;
    const long r0 = (long)blockIdx.x * chunk;
    long r1 = r0 + chunk;
    if (r1 > rows) r1 = rows;
    if (K <= 256) {
        const int R = 256 / K;  // rows in flight; lanes walk a row contiguously
        const int k = t % K, r = t / K;
        double acc = 0;
        if (r < R)
            for (long i = r0 + r; i < r1; i += R) acc += (double)x[(size_t)i * K + k];
        sm[t] = (r < R) ? acc : 0.0;
        __syncthreads();
        if (t < K) {
            double s = 0;
            for (int rr = 0; rr < R; rr++) s += sm[rr * K + t];
            partial[(size_t)blockIdx.x * K + t] = s;
        }
    } else {
        for (int kk = t; kk < K; kk += 256) {
            double acc = 0;
            for (long i = r0; i < r1; i++) acc += (double)x[(size_t)i * K + kk];
            partial[(size_t)blockIdx.x * K + kk] = acc;
        }
    }
}

// K % 4 == 0, K <= 1024: float4 loads, several rows in flight per thread (HBM-bound streaming pass)
template <bool BF>
__global__ void k_colsum4(const float *__restrict__ x, double *__restrict__ partial, long rows, int K, long chunk) {
    extern __shared__ double sm4[];  // [R][K]
    const int t = threadIdx.x;
    const int KG = K / 4;
    const int R = blockDim.x / KG;
    const int g = t % KG, r = t / KG;
    const long r0 = (long)blockIdx.x * chunk;
    long r1 = r0 + chunk;
    if (r1 > rows) r1 = rows;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (r < R) {
        auto ld = [&](long row) -> float4 {
            if (BF) {
                const uint2 q = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(x) + (size_t)row * K + g * 4);
                return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                                   __uint_as_float(q.y & 0xffff0000u));
            }
            return *reinterpret_cast<const float4 *>(x + (size_t)row * K + g * 4);
        };
        long i = r0 + r;
        for (; i + 3L * R < r1; i += 4L * R) {
            float4 q0 = ld(i);
            float4 q1 = ld(i + R);
            float4 q2 = ld(i + 2L * R);
            float4 q3 = ld(i + 3L * R);
            a0 += (double)((q0.x + q1.x) + (q2.x + q3.x));
            a1 += (double)((q0.y + q1.y) + (q2.y + q3.y));
            a2 += (double)((q0.z + q1.z) + (q2.z + q3.z));
            a3 += (double)((q0.w + q1.w) + (q2.w + q3.w));
        }
        for (; i < r1; i += R) {
            float4 q = ld(i);
            a0 += (double)q.x; a1 += (double)q.y; a2 += (double)q.z; a3 += (double)q.w;
        }
        double *o = sm4 + (size_t)r * K + g * 4;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
    __syncthreads();
    for (int k = t; k < K; k += blockDim.x) {
        double s = 0;
        for (int rr = 0; rr < R; rr++) s += sm4[(size_t)rr * K + k];
        partial[(size_t)blockIdx.x * K + k] = s;
    }
}

static int colsum_blocks(long rows, long *chunk) {
    long nb = rows / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    *chunk = cdiv(rows, nb);
    return (int)cdiv(rows, *chunk);
}
static size_t colsum_ws(long rows, int K) {
    long chunk;
    return (size_t)colsum_blocks(rows, &chunk) * K * sizeof(double) + 256;
}
static int colsum(const float *x, float *out, long rows, int K, void *ws, hipStream_t s, bool bf = false) {
    long chunk;
    int nb = colsum_blocks(rows, &chunk);
    double *partial = reinterpret_cast<double *>(ws);
    if (K % 4 == 0 && K <= 1024 && (((uintptr_t)x) & 15) == 0) {
        const int KG = K / 4;
        const int R = 256 / KG > 0 ? 256 / KG : 1;
        const int threads = (R * KG + 63) / 64 * 64;
        if (bf)
            hipLaunchKernelGGL(k_colsum4<true>, dim3(nb), dim3(threads), (size_t)R * K * sizeof(double), s, x, partial, rows,
                               K, chunk);
        else
            hipLaunchKernelGGL(k_colsum4<false>, dim3(nb), dim3(threads), (size_t)R * K * sizeof(double), s, x, partial, rows,
                               K, chunk);
    } else if (bf) {
        set_error("bias grad (bf16): K %% 4 != 0 is not supported");
        return 2;
    } else {
        hipLaunchKernelGGL(k_colsum, dim3(nb), dim3(256), 0, s, x, partial, rows, K, chunk);
    }
    if (check_launch("bias grad")) return 1;
    return reduce_partials(partial, out, nb, K, s);
}

// =============================================================================================== geometry builders
static inline int out_dim(int in, int k, int s) { return (in + 2 * ((k - 1) / 2) - k) / s + 1; }

static int check_ks(const int ks[3], const int st[3], const char *who) {
    for (int a = 0; a < 3; a++) {
        if (!(ks[a] == 1 || ks[a] == 3)) {
            set_error("%s: kernel size must be 1 or 3 per axis (got %d)", who, ks[a]);
            return 1;
        }
        if (!(st[a] == 1 || st[a] == 2)) {
            set_error("%s: stride must be 1 or 2 per axis (got %d)", who, st[a]);
            return 1;
        }
    }
    return 0;
}

static void conv_fwd_geom(FwdGeom &g, int N, int D, int H, int W, int C1, int C2, int K, const int ks[3],
                          const int st[3]) {
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = D; g.Hi = H; g.Wi = W;
    g.Do = out_dim(D, ks[0], st[0]); g.Ho = out_dim(H, ks[1], st[1]); g.Wo = out_dim(W, ks[2], st[2]);
    g.Dy = g.Do; g.Hy = g.Ho; g.Wy = g.Wo;
    g.C1 = C1; g.C2 = C2; g.K1 = K; g.K2 = 0;
    int t = 0;
    for (int a = 0; a < ks[0]; a++)
        for (int b = 0; b < ks[1]; b++)
            for (int c = 0; c < ks[2]; c++) {
                g.off[t][0] = a - (ks[0] - 1) / 2;
                g.off[t][1] = b - (ks[1] - 1) / 2;
                g.off[t][2] = c - (ks[2] - 1) / 2;
                g.wt[t] = t;
                t++;
            }
    g.ntaps = g.T = t;
    for (int a = 0; a < 3; a++) {
        g.sa[a] = st[a];
        g.so[a] = 1;
        g.oo[a] = 0;
    }
}

// u (optional): Winograd-domain weights of mvd_pack_weight_wino for this pass (uf for the forward, ub for the input
// gradient); used for plain 3x3x3 stride-1 problems with enough tiles to fill the chip, the direct engines otherwise
static int run_fwd(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1,
                   float *y2, void *ws, size_t ws_bytes, hipStream_t s, const float *u = nullptr, float *stats = nullptr,
                   int *stats_done = nullptr) {
    if (stats_done) *stats_done = 0;
    if (g_engine_mode == 0 && u && !g_wino_off) {
        const long tiles = (long)g.N * ((g.Do + 3) / 4) * ((g.Ho + 3) / 4) * ((g.Wo + 7) / 8) * ((g.K1 + g.K2) / 32);
        if (tiles >= g_wino_min_items) {
            int r = fwd_wino(g, a1, a2, u, bias, y1, y2, s, stats, stats_done);
            if (r >= 0) return r;
            if (stats_done) *stats_done = 0;
        }
    }
    if (g_engine_mode == 0) {
        int r = fwd_mfma(g, a1, a2, w, bias, y1, y2, ws, ws_bytes, s);
        if (r >= 0) return r;
    }
    return fwd_scalar(g, a1, a2, w, bias, y1, y2, s);
}

static int run_wgrad(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws,
                     size_t ws_bytes, hipStream_t s, float *dbias = nullptr, int *dbias_done = nullptr) {
    if (dbias_done) *dbias_done = 0;
    if (g_engine_mode == 0) {
        int r = wgrad_mfma(g, a1, a2, b, dw, ws, ws_bytes, s, false, dbias, dbias_done);
        if (r >= 0) return r;
        if (dbias_done) *dbias_done = 0;
    }
    return wgrad_scalar(g, a1, a2, b, dw, ws, ws_bytes, s);
}

static int run_fwd16(const FwdGeom &g, const uint16_t *a1, const uint16_t *a2, const uint16_t *w, const float *bias,
                     uint16_t *y1, uint16_t *y2, void *ws, size_t ws_bytes, hipStream_t s, const Fwd16Fuse *fuse = nullptr) {
    int r = fwd_bf16(g, a1, a2, w, bias, y1, y2, ws, ws_bytes, s, fuse);
    if (r < 0 && fuse && fuse->in_scale) {
        set_error("bf16 conv engine: the fused InstanceNorm input prologue needs the z-marching kernel (3x3x3 stride 1, 32 -> 32 "
                  "channels, >= 4 tiles per CU); C=%d+%d K=%d+%d", g.C1, g.C2, g.K1, g.K2);
        return 3;
    }
    if (r < 0) {
        set_error("bf16 conv engine: unsupported shape (needs C %% 32 == 0 and K %% 32 == 0; C=%d+%d K=%d+%d)", g.C1, g.C2,
                  g.K1, g.K2);
        return 3;
    }
    return r;
}

static int run_wgrad16(const WgradGeom &g, const uint16_t *a1, const uint16_t *a2, const uint16_t *b, float *dw, void *ws,
                       size_t ws_bytes, hipStream_t s, float *dbias = nullptr, int *dbias_done = nullptr) {
    int r = wgrad_mfma(g, reinterpret_cast<const float *>(a1), reinterpret_cast<const float *>(a2),
                       reinterpret_cast<const float *>(b), dw, ws, ws_bytes, s, true, dbias, dbias_done);
    if (r < 0) {
        set_error("bf16 wgrad: unsupported shape (needs C %% 32 == 0 and K %% 32 == 0; C=%d+%d K=%d)", g.C1, g.C2, g.K);
        return 3;
    }
    return r;
}

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

}  // namespace mvd

using namespace mvd;

extern "C" {

int mvd_set_conv_engine(int mode) {
    g_engine_mode = mode ? 1 : 0;
    return 0;
}

int mvd_set_wino_min_items(long n) {
    g_wino_min_items = n < 0 ? 256 : n;
    return 0;
}

size_t mvd_conv_fwd_workspace_bytes(int N, long out_voxels, int K) { return fwd_mfma_ws(N, out_voxels, K); }

static int conv3d_fwd_impl(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *uf,
                           const float *bias, float *y, int N, int D, int H, int W, int K, const int ksize[3],
                           const int stride[3], void *ws, size_t ws_bytes, void *stream, float *stats = nullptr,
                           int *stats_done = nullptr) {
    MVD_REQUIRE(x1 && wf && y && C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "conv3d_fwd: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_fwd: bad shape");
    if (check_ks(ksize, stride, "conv3d_fwd")) return 2;
    FwdGeom g;
    conv_fwd_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    return run_fwd(g, x1, x2, wf, bias, y, nullptr, ws, ws_bytes, as_stream(stream), uf, stats, stats_done);
}

int mvd_conv3d_fwd(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *bias, float *y, int N,
                   int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                   void *stream) {
    return conv3d_fwd_impl(x1, C1, x2, C2, wf, nullptr, bias, y, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream);
}

int mvd_conv3d_fwd_wino(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *uf,
                        const float *bias, float *y, int N, int D, int H, int W, int K, const int ksize[3],
                        const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    return conv3d_fwd_impl(x1, C1, x2, C2, wf, uf, bias, y, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream);
}

/* 1 when mvd_conv3d_fwd_wino / mvd_conv3d_dgrad_wino would run the Winograd kernel for this conv (the caller can skip
 * packing uf/ub otherwise).  Cin/Cout: reduce / produce channel counts of the pass (swap them for the dgrad). */
int mvd_conv_wino_applicable(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3], const int stride[3]) {
    if (g_engine_mode != 0 || g_wino_off) return 0;
    for (int a = 0; a < 3; a++)
        if (ksize[a] != 3 || stride[a] != 1) return 0;
    if (C1 % 32 || C2 % 32 || K % 32 || C1 <= 0 || K <= 0) return 0;
    const long tiles_f = (long)N * ((D + 3) / 4) * ((H + 3) / 4) * ((W + 7) / 8) * (K / 32);
    const long tiles_b = (long)N * ((D + 3) / 4) * ((H + 3) / 4) * ((W + 7) / 8) * ((C1 + C2) / 32);
    return (tiles_f >= g_wino_min_items ? 1 : 0) | (tiles_b >= g_wino_min_items ? 2 : 0);
}

size_t mvd_wino_weight_elems(int C, int K) { return wino_weight_elems(C, K); }
int mvd_wino_mode(void) { return wino_mode(); }

/* InstanceNorm statistics epilogue: tiles per sample of the partial-statistics buffer [N][tiles][K][2] floats */
size_t mvd_conv_stats_tiles(int D, int H, int W) { return (size_t)((D + 3) / 4) * ((H + 3) / 4) * ((W + 7) / 8); }

int mvd_conv3d_fwd_wino_stats(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *uf,
                              const float *bias, float *y, float *stats, int *stats_done, int N, int D, int H, int W,
                              int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(stats_done, "conv3d_fwd_wino_stats: stats_done is required");
    return conv3d_fwd_impl(x1, C1, x2, C2, wf, uf, bias, y, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream, stats,
                           stats_done);
}

int mvd_pack_weights_batch(int n, const float *const *w, float *const *wf, float *const *wb, float *const *uf,
                           float *const *ub, const int *K, const int *C, const int *T, const int *transposed,
                           void *stream) {
    MVD_REQUIRE(n > 0 && w && wf && wb && uf && ub && K && C && T && transposed, "pack_weights_batch: null table");
    for (int q = 0; q < n; q++) {
        MVD_REQUIRE(w[q] && (wf[q] || wb[q] || uf[q] || ub[q]), "pack_weights_batch: job without source or destination");
        MVD_REQUIRE(K[q] > 0 && C[q] > 0 && T[q] > 0 && T[q] <= MVD_MAX_TAPS, "pack_weights_batch: bad K / C / T");
        if (uf[q] || ub[q]) {
            MVD_REQUIRE(wino_mode() == 2, "pack_weights_batch: Winograd tables need MVD_WINO=2 (F(2x2,3x3) layout)");
            MVD_REQUIRE(T[q] == 27 && !transposed[q] && K[q] % 32 == 0 && C[q] % 32 == 0,
                        "pack_weights_batch: Winograd tables need a 3x3x3 conv with C %% 32 == 0 and K %% 32 == 0");
        }
    }
    return pack_weights_batch(n, w, wf, wb, uf, ub, K, C, T, transposed, as_stream(stream));
}

int mvd_pack_weight_wino(const float *w, float *uf, float *ub, int K, int C, void *stream) {
    MVD_REQUIRE(w && (uf || ub) && K > 0 && C > 0, "pack_weight_wino: bad arguments");
    MVD_REQUIRE(K % 32 == 0 && C % 32 == 0, "pack_weight_wino: needs C %% 32 == 0 and K %% 32 == 0");
    return pack_weight_wino(w, uf, ub, K, C, as_stream(stream));
}

static int conv3d_dgrad_impl(const float *dy, const float *wb, const float *ub, float *dx1, int C1, float *dx2, int C2,
                             int N, int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws,
                             size_t ws_bytes, void *stream);

int mvd_conv3d_dgrad(const float *dy, const float *wb, float *dx1, int C1, float *dx2, int C2, int N, int D, int H, int W,
                     int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    return conv3d_dgrad_impl(dy, wb, nullptr, dx1, C1, dx2, C2, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream);
}

int mvd_conv3d_dgrad_wino(const float *dy, const float *wb, const float *ub, float *dx1, int C1, float *dx2, int C2, int N,
                          int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                          void *stream) {
    return conv3d_dgrad_impl(dy, wb, ub, dx1, C1, dx2, C2, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream);
}

static int conv3d_dgrad_impl(const float *dy, const float *wb, const float *ub, float *dx1, int C1, float *dx2, int C2,
                             int N, int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws,
                             size_t ws_bytes, void *stream) {
    MVD_REQUIRE(dy && wb && dx1 && C1 > 0 && C2 >= 0 && (C2 == 0 || dx2), "conv3d_dgrad: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_dgrad: bad shape");
    if (check_ks(ksize, stride, "conv3d_dgrad")) return 2;
    const int dims[3] = {D, H, W};
    int od[3], pad[3];
    for (int a = 0; a < 3; a++) {
        od[a] = out_dim(dims[a], ksize[a], stride[a]);
        pad[a] = (ksize[a] - 1) / 2;
    }
    // 3x3x3 stride-2 conv with 32-multiple channels and a single input: all eight parity classes in one kernel
    static int dg2 = -1;
    if (dg2 < 0) dg2 = getenv("MVD_DGRAD2") ? atoi(getenv("MVD_DGRAD2")) : 1;
    if (dg2 && g_engine_mode == 0 && C2 == 0 && ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3 && stride[0] == 2 &&
        stride[1] == 2 && stride[2] == 2 && ((long)N * od[0] * od[1] * od[2] >= 4096 || g_wino_min_items <= 1)) {
        int r = dgrad32s(N, D, H, W, C1, K, od[0], od[1], od[2], dy, wb, dx1, as_stream(stream));
        if (r >= 0) return r;
    }
    // one launch per output-parity class (1 class per stride-1 axis, 2 per stride-2 axis)
    for (int pd = 0; pd < stride[0]; pd++)
        for (int ph = 0; ph < stride[1]; ph++)
            for (int pw = 0; pw < stride[2]; pw++) {
                const int p[3] = {pd, ph, pw};
                FwdGeom g;
                memset(&g, 0, sizeof(g));
                g.N = N;
                g.Di = od[0]; g.Hi = od[1]; g.Wi = od[2];
                g.Dy = D; g.Hy = H; g.Wy = W;
                int grid[3];
                bool empty = false;
                for (int a = 0; a < 3; a++) {
                    grid[a] = (dims[a] - p[a] + stride[a] - 1) / stride[a];
                    if (grid[a] <= 0) empty = true;
                    g.sa[a] = 1;
                    g.so[a] = stride[a];
                    g.oo[a] = p[a];
                }
                if (empty) continue;
                g.Do = grid[0]; g.Ho = grid[1]; g.Wo = grid[2];
                g.C1 = K; g.C2 = 0; g.K1 = C1; g.K2 = C2;
                int nt = 0;
                for (int ta = 0; ta < ksize[0]; ta++)
                    for (int tb = 0; tb < ksize[1]; tb++)
                        for (int tc = 0; tc < ksize[2]; tc++) {
                            const int t3[3] = {ta, tb, tc};
                            int off[3];
                            bool ok = true;
                            for (int a = 0; a < 3; a++) {
                                int num = p[a] + pad[a] - t3[a];  // dy index q: q*s + t - pad = o*s + p
                                if (num % stride[a] != 0) { ok = false; break; }
                                off[a] = num / stride[a];
                            }
                            if (!ok) continue;
                            for (int a = 0; a < 3; a++) g.off[nt][a] = (int8_t)off[a];
                            g.wt[nt] = (int8_t)((ta * ksize[1] + tb) * ksize[2] + tc);
                            nt++;
                        }
                g.ntaps = nt;
                g.T = ksize[0] * ksize[1] * ksize[2];
                if (nt == 0) continue;
                int r = run_fwd(g, dy, nullptr, wb, nullptr, dx1, dx2, ws, ws_bytes, as_stream(stream), ub);
                if (r) return r;
            }
    return 0;
}

static void conv_wgrad_geom(WgradGeom &g, int N, int D, int H, int W, int C1, int C2, int K, const int ks[3],
                            const int st[3]) {
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = D; g.Hi = H; g.Wi = W;
    g.Do = out_dim(D, ks[0], st[0]); g.Ho = out_dim(H, ks[1], st[1]); g.Wo = out_dim(W, ks[2], st[2]);
    g.Db = g.Do; g.Hb = g.Ho; g.Wb = g.Wo;
    g.C1 = C1; g.C2 = C2; g.K = K;
    int t = 0;
    for (int a = 0; a < ks[0]; a++)
        for (int b = 0; b < ks[1]; b++)
            for (int c = 0; c < ks[2]; c++) {
                g.off[t][0] = a - (ks[0] - 1) / 2;
                g.off[t][1] = b - (ks[1] - 1) / 2;
                g.off[t][2] = c - (ks[2] - 1) / 2;
                g.wt[t] = t;
                t++;
            }
    g.ntaps = g.T = t;
    for (int a = 0; a < 3; a++) {
        g.sa[a] = st[a];
        g.sb[a] = 1;
    }
    g.transposed_out = 0;
}

size_t mvd_conv3d_wgrad_workspace_bytes(int C, int K, int T, int N, int Do, int Ho, int Wo) {
    WgradGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Do = Do; g.Ho = Ho; g.Wo = Wo; g.C1 = C; g.C2 = 0; g.K = K; g.ntaps = g.T = T;
    size_t a = max_sz(wgrad_scalar_ws(g), wgrad_mfma_ws(g));
    return max_sz(a, colsum_ws((long)N * Do * Ho * Wo, K));
}

int mvd_conv3d_wgrad(const float *x1, int C1, const float *x2, int C2, const float *dy, float *dw, float *dbias, int N,
                     int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                     void *stream) {
    MVD_REQUIRE(x1 && dy && dw && ws && C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "conv3d_wgrad: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_wgrad: bad shape");
    if (check_ks(ksize, stride, "conv3d_wgrad")) return 2;
    WgradGeom g;
    conv_wgrad_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    MVD_REQUIRE(ws_bytes >= mvd_conv3d_wgrad_workspace_bytes(C1 + C2, K, g.T, N, g.Do, g.Ho, g.Wo),
                "conv3d_wgrad: workspace too small");
    hipStream_t s = as_stream(stream);
    int dbias_done = 0;
    int r = run_wgrad(g, x1, x2, dy, dw, ws, ws_bytes, s, dbias, &dbias_done);
    if (r) return r;
    if (dbias && !dbias_done) return colsum(dy, dbias, (long)N * g.Do * g.Ho * g.Wo, K, ws, s);  // ws is free again
    return 0;
}

// ------------------------------------------------------------------------------------------------ ConvTranspose3d k == s
int mvd_convT3d_fwd(const float *x, const float *wf, const float *bias, float *y, int N, int D, int H, int W, int C, int K,
                    const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && wf && y && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && K > 0, "convT3d_fwd: bad arguments");
    for (int a = 0; a < 3; a++) MVD_REQUIRE(stride[a] == 1 || stride[a] == 2, "convT3d_fwd: stride must be 1 or 2");
    if (g_engine_mode == 0 && !g_transp_off) {
        int r = convT_fwd_direct(x, wf, bias, y, N, D, H, W, C, K, stride, as_stream(stream));
        if (r >= 0) return r;
    }
    for (int pd = 0; pd < stride[0]; pd++)
        for (int ph = 0; ph < stride[1]; ph++)
            for (int pw = 0; pw < stride[2]; pw++) {
                FwdGeom g;
                memset(&g, 0, sizeof(g));
                g.N = N;
                g.Di = g.Do = D; g.Hi = g.Ho = H; g.Wi = g.Wo = W;
                g.Dy = D * stride[0]; g.Hy = H * stride[1]; g.Wy = W * stride[2];
                g.C1 = C; g.K1 = K;
                g.ntaps = 1;
                g.T = stride[0] * stride[1] * stride[2];
                g.wt[0] = (int8_t)((pd * stride[1] + ph) * stride[2] + pw);
                const int p[3] = {pd, ph, pw};
                for (int a = 0; a < 3; a++) {
                    g.sa[a] = 1;
                    g.so[a] = stride[a];
                    g.oo[a] = p[a];
                }
                int r = run_fwd(g, x, nullptr, wf, bias, y, nullptr, ws, ws_bytes, as_stream(stream));
                if (r) return r;
            }
    return 0;
}

int mvd_convT3d_dgrad(const float *dy, const float *wb, float *dx, int N, int D, int H, int W, int C, int K,
                      const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(dy && wb && dx && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && K > 0, "convT3d_dgrad: bad arguments");
    for (int a = 0; a < 3; a++) MVD_REQUIRE(stride[a] == 1 || stride[a] == 2, "convT3d_dgrad: stride must be 1 or 2");
    if (g_engine_mode == 0 && !g_transp_off) {
        int r = convT_dgrad_direct(dy, wb, dx, N, D, H, W, C, K, stride, as_stream(stream));
        if (r >= 0) return r;
    }
    FwdGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = D * stride[0]; g.Hi = H * stride[1]; g.Wi = W * stride[2];
    g.Do = g.Dy = D; g.Ho = g.Hy = H; g.Wo = g.Wy = W;
    g.C1 = K; g.K1 = C;
    int t = 0;
    for (int pd = 0; pd < stride[0]; pd++)
        for (int ph = 0; ph < stride[1]; ph++)
            for (int pw = 0; pw < stride[2]; pw++) {
                g.off[t][0] = pd; g.off[t][1] = ph; g.off[t][2] = pw;
                g.wt[t] = t;
                t++;
            }
    g.ntaps = g.T = t;
    for (int a = 0; a < 3; a++) {
        g.sa[a] = stride[a];
        g.so[a] = 1;
        g.oo[a] = 0;
    }
    return run_fwd(g, dy, nullptr, wb, nullptr, dx, nullptr, ws, ws_bytes, as_stream(stream));
}

static void convT_wgrad_geom(WgradGeom &g, int N, int D, int H, int W, int C, int K, const int st[3]) {
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = g.Do = D; g.Hi = g.Ho = H; g.Wi = g.Wo = W;
    g.Db = D * st[0]; g.Hb = H * st[1]; g.Wb = W * st[2];
    g.C1 = C; g.C2 = 0; g.K = K;
    int t = 0;
    for (int pd = 0; pd < st[0]; pd++)
        for (int ph = 0; ph < st[1]; ph++)
            for (int pw = 0; pw < st[2]; pw++) {
                g.ob[t][0] = pd; g.ob[t][1] = ph; g.ob[t][2] = pw;
                g.wt[t] = t;
                t++;
            }
    g.ntaps = g.T = t;
    for (int a = 0; a < 3; a++) {
        g.sa[a] = 1;
        g.sb[a] = st[a];
    }
    g.transposed_out = 1;
}

size_t mvd_convT3d_wgrad_workspace_bytes(int C, int K, int T, int N, int D, int H, int W) {
    WgradGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Do = D; g.Ho = H; g.Wo = W; g.C1 = C; g.K = K; g.ntaps = g.T = T;
    size_t a = max_sz(wgrad_scalar_ws(g), wgrad_mfma_ws(g));
    return max_sz(a, colsum_ws((long)N * D * H * W * T, K));
}

int mvd_convT3d_wgrad(const float *x, const float *dy, float *dw, float *dbias, int N, int D, int H, int W, int C, int K,
                      const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && dy && dw && ws && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && K > 0, "convT3d_wgrad: bad arguments");
    for (int a = 0; a < 3; a++) MVD_REQUIRE(stride[a] == 1 || stride[a] == 2, "convT3d_wgrad: stride must be 1 or 2");
    WgradGeom g;
    convT_wgrad_geom(g, N, D, H, W, C, K, stride);
    MVD_REQUIRE(ws_bytes >= mvd_convT3d_wgrad_workspace_bytes(C, K, g.T, N, D, H, W), "convT3d_wgrad: workspace too small");
    hipStream_t s = as_stream(stream);
    if (dbias) {
        int r = colsum(dy, dbias, (long)N * g.Db * g.Hb * g.Wb, K, ws, s);
        if (r) return r;
    }
    return run_wgrad(g, x, nullptr, dy, dw, ws, ws_bytes, s);
}

// ------------------------------------------------------------------------------------------------ bf16 twins
int mvd_pack_weight_bf16(const float *w, uint16_t *wf, uint16_t *wb, int K, int C, int T, int transposed, void *stream) {
    MVD_REQUIRE(w && (wf || wb) && K > 0 && C > 0 && T > 0 && T <= MVD_MAX_TAPS, "pack_weight_bf16: bad arguments");
    MVD_REQUIRE(!wf || C % 32 == 0, "pack_weight_bf16: wf needs C %% 32 == 0");
    MVD_REQUIRE(!wb || K % 32 == 0, "pack_weight_bf16: wb needs K %% 32 == 0");
    return pack_weight16(w, wf, wb, K, C, T, transposed, as_stream(stream));
}

int mvd_pack_weights_bf16_batch(int n, const float *const *w, uint16_t *const *wf, uint16_t *const *wb, const int *K,
                                const int *C, const int *T, const int *transposed, void *stream) {
    MVD_REQUIRE(n > 0 && w && wf && wb && K && C && T && transposed, "pack_weights_bf16_batch: null table");
    for (int q = 0; q < n; q++) {
        MVD_REQUIRE(w[q] && (wf[q] || wb[q]), "pack_weights_bf16_batch: job without source or destination");
        MVD_REQUIRE(K[q] > 0 && C[q] > 0 && T[q] > 0 && T[q] <= MVD_MAX_TAPS, "pack_weights_bf16_batch: bad K / C / T");
        MVD_REQUIRE(!wf[q] || C[q] % 32 == 0, "pack_weights_bf16_batch: wf needs C %% 32 == 0");
        MVD_REQUIRE(!wb[q] || K[q] % 32 == 0, "pack_weights_bf16_batch: wb needs K %% 32 == 0");
    }
    return pack_weights16_batch(n, w, reinterpret_cast<unsigned short *const *>(wf), reinterpret_cast<unsigned short *const *>(wb),
                                K, C, T, transposed, as_stream(stream));
}

int mvd_conv3d_fwd_bf16(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const uint16_t *wf, const float *bias, uint16_t *y, int N,
                   int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                   void *stream) {
    MVD_REQUIRE(x1 && wf && y && C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "conv3d_fwd_bf16: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_fwd_bf16: bad shape");
    if (check_ks(ksize, stride, "conv3d_fwd_bf16")) return 2;
    FwdGeom g;
    conv_fwd_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    return run_fwd16(g, x1, x2, wf, bias, y, nullptr, ws, ws_bytes, as_stream(stream));
}

int mvd_set_bf16_zmarch_kernel(int which) {
    if (which != 0 && which != 1) {
        set_error("set_bf16_zmarch_kernel: 0 = k_fwd16z (32x32x16 tiles), 1 = k_fwd16y (16x16x32 tiles, InstanceNorm fusion)");
        return 2;
    }
    fwd16y_enable(which);
    return 0;
}

int mvd_set_bf16_wgrad_kernel(int which) {
    if (which != 0 && which != 1) {
        set_error("set_bf16_wgrad_kernel: 0 = k_wgrad16 (tiled), 1 = k_wgrad16z (z-marching)");
        return 2;
    }
    wgrad16z_enable(which);
    return 0;
}

int mvd_conv3d_fwd_bf16_stats_tiles(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3],
                                    const int stride[3]) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || K <= 0 || check_ks(ksize, stride, "conv3d_fwd_bf16_stats_tiles"))
        return 0;
    FwdGeom g;
    conv_fwd_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    return fwd_bf16_stats_tiles(g);
}

int mvd_conv3d_fwd_bf16_prologue_ok(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3],
                                    const int stride[3]) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || K <= 0 || check_ks(ksize, stride, "conv3d_fwd_bf16_prologue_ok"))
        return 0;
    FwdGeom g;
    conv_fwd_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    return fwd_bf16_prologue_ok(g);
}

int mvd_conv3d_fwd_bf16_fused(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const uint16_t *wf, const float *bias,
                              uint16_t *y, int N, int D, int H, int W, int K, const int ksize[3], const int stride[3],
                              const float *in_scale, const float *in_shift, float slope, float *tile_stats, int *ntiles_out,
                              void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x1 && wf && y && C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "conv3d_fwd_bf16_fused: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_fwd_bf16_fused: bad shape");
    MVD_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3d_fwd_bf16_fused: scale and shift come together");
    MVD_REQUIRE(!tile_stats || ntiles_out, "conv3d_fwd_bf16_fused: tile_stats needs ntiles_out");
    if (check_ks(ksize, stride, "conv3d_fwd_bf16_fused")) return 2;
    FwdGeom g;
    conv_fwd_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    Fwd16Fuse f;
    f.tile_stats = tile_stats;
    f.ntiles = ntiles_out;
    f.in_scale = in_scale;
    f.in_shift = in_shift;
    f.slope = slope;
    if (ntiles_out) *ntiles_out = 0;
    return run_fwd16(g, x1, x2, wf, bias, y, nullptr, ws, ws_bytes, as_stream(stream), &f);
}

static int dgrad_bf16_impl(const uint16_t *dy, const uint16_t *wb, uint16_t *dx1, int C1, uint16_t *dx2, int C2, int N, int D, int H,
                           int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream,
                           int acc) {
    MVD_REQUIRE(dy && wb && dx1 && C1 > 0 && C2 >= 0 && (C2 == 0 || dx2), "conv3d_dgrad_bf16: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_dgrad_bf16: bad shape");
    if (check_ks(ksize, stride, "conv3d_dgrad_bf16")) return 2;
    const int dims[3] = {D, H, W};
    int od[3], pad[3];
    for (int a = 0; a < 3; a++) {
        od[a] = out_dim(dims[a], ksize[a], stride[a]);
        pad[a] = (ksize[a] - 1) / 2;
    }
    // 3x3x3 stride-2 conv with 32-multiple channels and a single input: all eight parity classes in one kernel
    if (g_engine_mode == 0 && C2 == 0 && ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3 && stride[0] == 2 && stride[1] == 2 &&
        stride[2] == 2 && (long)N * od[0] * od[1] * od[2] >= 512) {
        int r = dgrad16s(N, D, H, W, C1, K, od[0], od[1], od[2], dy, wb, dx1, as_stream(stream), acc);
        if (r >= 0) return r;
    }
    // one launch per output-parity class (1 class per stride-1 axis, 2 per stride-2 axis)
    for (int pd = 0; pd < stride[0]; pd++)
        for (int ph = 0; ph < stride[1]; ph++)
            for (int pw = 0; pw < stride[2]; pw++) {
                const int p[3] = {pd, ph, pw};
                FwdGeom g;
                memset(&g, 0, sizeof(g));
                g.N = N;
                g.Di = od[0]; g.Hi = od[1]; g.Wi = od[2];
                g.Dy = D; g.Hy = H; g.Wy = W;
                int grid[3];
                bool empty = false;
                for (int a = 0; a < 3; a++) {
                    grid[a] = (dims[a] - p[a] + stride[a] - 1) / stride[a];
                    if (grid[a] <= 0) empty = true;
                    g.sa[a] = 1;
                    g.so[a] = stride[a];
                    g.oo[a] = p[a];
                }
                if (empty) continue;
                g.Do = grid[0]; g.Ho = grid[1]; g.Wo = grid[2];
                g.C1 = K; g.C2 = 0; g.K1 = C1; g.K2 = C2;
                int nt = 0;
                for (int ta = 0; ta < ksize[0]; ta++)
                    for (int tb = 0; tb < ksize[1]; tb++)
                        for (int tc = 0; tc < ksize[2]; tc++) {
                            const int t3[3] = {ta, tb, tc};
                            int off[3];
                            bool ok = true;
                            for (int a = 0; a < 3; a++) {
                                int num = p[a] + pad[a] - t3[a];  // dy index q: q*s + t - pad = o*s + p
                                if (num % stride[a] != 0) { ok = false; break; }
                                off[a] = num / stride[a];
                            }
                            if (!ok) continue;
                            for (int a = 0; a < 3; a++) g.off[nt][a] = (int8_t)off[a];
                            g.wt[nt] = (int8_t)((ta * ksize[1] + tb) * ksize[2] + tc);
                            nt++;
                        }
                g.ntaps = nt;
                g.T = ksize[0] * ksize[1] * ksize[2];
                g.acc = (int8_t)acc;
                if (nt == 0) continue;
                int r = run_fwd16(g, dy, nullptr, wb, nullptr, dx1, dx2, ws, ws_bytes, as_stream(stream));
                if (r) return r;
            }
    return 0;
}

int mvd_conv3d_dgrad_bf16(const uint16_t *dy, const uint16_t *wb, uint16_t *dx1, int C1, uint16_t *dx2, int C2, int N, int D, int H, int W,
                     int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    return dgrad_bf16_impl(dy, wb, dx1, C1, dx2, C2, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream, 0);
}

static bool strided3(const int ksize[3], const int stride[3]) {
    return ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3 && (stride[0] == 2 || stride[1] == 2 || stride[2] == 2);
}

int mvd_conv3d_dgrad_acc_ok(int is_bf16, int N, int D, int H, int W, int C1, int K, const int ksize[3], const int stride[3]) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C1 <= 0 || K <= 0 || check_ks(ksize, stride, "conv3d_dgrad_acc_ok")) return 0;
    if (C1 % 32 || K % 32 || g_engine_mode != 0) return 0;
    if (is_bf16) return strided3(ksize, stride) ? 1 : 0;   // every parity class runs on the generic bf16 kernel
    if (!(ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3 && stride[0] == 2 && stride[1] == 2 && stride[2] == 2)) return 0;
    const long items = (long)N * out_dim(D, 3, 2) * out_dim(H, 3, 2) * out_dim(W, 3, 2);
    static int dg2 = -1;
    if (dg2 < 0) dg2 = getenv("MVD_DGRAD2") ? atoi(getenv("MVD_DGRAD2")) : 1;
    return (dg2 && (items >= 4096 || g_wino_min_items <= 1) && (long)out_dim(D, 3, 2) * out_dim(H, 3, 2) * out_dim(W, 3, 2) * K * 4 < (1L << 31)) ? 1 : 0;
}

int mvd_conv3d_dgrad_bf16_acc(const uint16_t *dy, const uint16_t *wb, uint16_t *dx1, int C1, int N, int D, int H, int W, int K,
                              const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(mvd_conv3d_dgrad_acc_ok(1, N, D, H, W, C1, K, ksize, stride), "conv3d_dgrad_bf16_acc: no accumulating kernel for this "
                "shape (ask mvd_conv3d_dgrad_acc_ok first)");
    return dgrad_bf16_impl(dy, wb, dx1, C1, nullptr, 0, N, D, H, W, K, ksize, stride, ws, ws_bytes, stream, 1);
}

int mvd_conv3d_dgrad_acc(const float *dy, const float *wb, float *dx1, int C1, int N, int D, int H, int W, int K,
                         const int ksize[3], const int stride[3], void *stream) {
    MVD_REQUIRE(dy && wb && dx1, "conv3d_dgrad_acc: null pointer");
    MVD_REQUIRE(mvd_conv3d_dgrad_acc_ok(0, N, D, H, W, C1, K, ksize, stride), "conv3d_dgrad_acc: no accumulating kernel for this shape "
                "(ask mvd_conv3d_dgrad_acc_ok first)");
    int r = dgrad32s(N, D, H, W, C1, K, out_dim(D, 3, 2), out_dim(H, 3, 2), out_dim(W, 3, 2), dy, wb, dx1, as_stream(stream), 1);
    if (r < 0) {
        set_error("conv3d_dgrad_acc: the fused stride-2 kernel refused the shape");
        return 3;
    }
    return r;
}

int mvd_convT3d_fwd_bf16(const uint16_t *x, const uint16_t *wf, const float *bias, uint16_t *y, int N, int D, int H, int W, int C, int K,
                    const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && wf && y && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && K > 0, "convT3d_fwd_bf16: bad arguments");
    for (int a = 0; a < 3; a++) MVD_REQUIRE(stride[a] == 1 || stride[a] == 2, "convT3d_fwd_bf16: stride must be 1 or 2");
    if (!g_transp_off) {
        int r = convT_fwd_direct16(x, wf, bias, y, N, D, H, W, C, K, stride, as_stream(stream));
        if (r >= 0) return r;
    }
    for (int pd = 0; pd < stride[0]; pd++)
        for (int ph = 0; ph < stride[1]; ph++)
            for (int pw = 0; pw < stride[2]; pw++) {
                FwdGeom g;
                memset(&g, 0, sizeof(g));
                g.N = N;
                g.Di = g.Do = D; g.Hi = g.Ho = H; g.Wi = g.Wo = W;
                g.Dy = D * stride[0]; g.Hy = H * stride[1]; g.Wy = W * stride[2];
                g.C1 = C; g.K1 = K;
                g.ntaps = 1;
                g.T = stride[0] * stride[1] * stride[2];
                g.wt[0] = (int8_t)((pd * stride[1] + ph) * stride[2] + pw);
                const int p[3] = {pd, ph, pw};
                for (int a = 0; a < 3; a++) {
                    g.sa[a] = 1;
                    g.so[a] = stride[a];
                    g.oo[a] = p[a];
                }
                int r = run_fwd16(g, x, nullptr, wf, bias, y, nullptr, ws, ws_bytes, as_stream(stream));
                if (r) return r;
            }
    return 0;
}

int mvd_convT3d_dgrad_bf16(const uint16_t *dy, const uint16_t *wb, uint16_t *dx, int N, int D, int H, int W, int C, int K,
                      const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(dy && wb && dx && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && K > 0, "convT3d_dgrad_bf16: bad arguments");
    for (int a = 0; a < 3; a++) MVD_REQUIRE(stride[a] == 1 || stride[a] == 2, "convT3d_dgrad_bf16: stride must be 1 or 2");
    if (!g_transp_off) {
        int r = convT_dgrad_direct16(dy, wb, dx, N, D, H, W, C, K, stride, as_stream(stream));
        if (r >= 0) return r;
    }
    FwdGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N;
    g.Di = D * stride[0]; g.Hi = H * stride[1]; g.Wi = W * stride[2];
    g.Do = g.Dy = D; g.Ho = g.Hy = H; g.Wo = g.Wy = W;
    g.C1 = K; g.K1 = C;
    int t = 0;
    for (int pd = 0; pd < stride[0]; pd++)
        for (int ph = 0; ph < stride[1]; ph++)
            for (int pw = 0; pw < stride[2]; pw++) {
                g.off[t][0] = pd; g.off[t][1] = ph; g.off[t][2] = pw;
                g.wt[t] = t;
                t++;
            }
    g.ntaps = g.T = t;
    for (int a = 0; a < 3; a++) {
        g.sa[a] = stride[a];
        g.so[a] = 1;
        g.oo[a] = 0;
    }
    return run_fwd16(g, dy, nullptr, wb, nullptr, dx, nullptr, ws, ws_bytes, as_stream(stream));
}

int mvd_conv3d_wgrad_bf16(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const uint16_t *dy, float *dw, float *dbias, int N,
                     int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                     void *stream) {
    MVD_REQUIRE(x1 && dy && dw && ws && C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "conv3d_wgrad_bf16: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_wgrad_bf16: bad shape");
    if (check_ks(ksize, stride, "conv3d_wgrad_bf16")) return 2;
    WgradGeom g;
    conv_wgrad_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    MVD_REQUIRE(ws_bytes >= mvd_conv3d_wgrad_workspace_bytes(C1 + C2, K, g.T, N, g.Do, g.Ho, g.Wo),
                "conv3d_wgrad_bf16: workspace too small");
    hipStream_t s = as_stream(stream);
    // the 27-tap kernel produces the bias gradient from its idle tap slot; the other shapes take the column-sum pass
    int dbias_done = 0;
    int r = run_wgrad16(g, x1, x2, dy, dw, ws, ws_bytes, s, dbias, &dbias_done);
    if (r) return r;
    if (dbias && !dbias_done) return colsum(reinterpret_cast<const float *>(dy), dbias, (long)N * g.Do * g.Ho * g.Wo, K, ws, s, true);
    return 0;
}

int mvd_conv3d_wgrad_bf16_prologue_ok(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3], const int stride[3]) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || K <= 0 || check_ks(ksize, stride, "conv3d_wgrad_bf16_prologue_ok"))
        return 0;
    WgradGeom g;
    conv_wgrad_geom(g, N, D, H, W, C1, C2, K, ksize, stride);
    return wgrad16z_prologue_ok(g) ? 1 : 0;
}

int mvd_conv3d_wgrad_bf16_fused(const uint16_t *x1, int C1, const uint16_t *dy, float *dw, float *dbias, int N, int D, int H, int W,
                                int K, const int ksize[3], const int stride[3], const float *in_scale, const float *in_shift,
                                float slope, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x1 && dy && dw && ws && in_scale && in_shift && C1 > 0, "conv3d_wgrad_bf16_fused: null pointer / bad channels");
    MVD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && K > 0, "conv3d_wgrad_bf16_fused: bad shape");
    if (check_ks(ksize, stride, "conv3d_wgrad_bf16_fused")) return 2;
    WgradGeom g;
    conv_wgrad_geom(g, N, D, H, W, C1, 0, K, ksize, stride);
    MVD_REQUIRE(wgrad16z_prologue_ok(g), "conv3d_wgrad_bf16_fused: shape not served by the z-marching kernel (query _prologue_ok)");
    MVD_REQUIRE(ws_bytes >= mvd_conv3d_wgrad_workspace_bytes(C1, K, g.T, N, g.Do, g.Ho, g.Wo),
                "conv3d_wgrad_bf16_fused: workspace too small");
    hipStream_t s = as_stream(stream);
    int dbias_done = 0;
    const int r = wgrad16z_run(g, x1, nullptr, dy, dw, ws, ws_bytes, s, dbias, &dbias_done, in_scale, in_shift, slope);
    if (r < 0) {
        set_error("conv3d_wgrad_bf16_fused: the z-marching kernel refused the launch (workspace?)");
        return 2;
    }
    if (r) return r;
    if (dbias && !dbias_done) return colsum(reinterpret_cast<const float *>(dy), dbias, (long)N * g.Do * g.Ho * g.Wo, K, ws, s, true);
    return 0;
}

int mvd_convT3d_wgrad_bf16(const uint16_t *x, const uint16_t *dy, float *dw, float *dbias, int N, int D, int H, int W, int C, int K,
                      const int stride[3], void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(x && dy && dw && ws && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && K > 0, "convT3d_wgrad_bf16: bad arguments");
    for (int a = 0; a < 3; a++) MVD_REQUIRE(stride[a] == 1 || stride[a] == 2, "convT3d_wgrad_bf16: stride must be 1 or 2");
    WgradGeom g;
    convT_wgrad_geom(g, N, D, H, W, C, K, stride);
    MVD_REQUIRE(ws_bytes >= mvd_convT3d_wgrad_workspace_bytes(C, K, g.T, N, D, H, W), "convT3d_wgrad_bf16: workspace too small");
    hipStream_t s = as_stream(stream);
    if (dbias) {
        int r = colsum(reinterpret_cast<const float *>(dy), dbias, (long)N * g.Db * g.Hb * g.Wb, K, ws, s, true);
        if (r) return r;
    }
    return run_wgrad16(g, x, nullptr, dy, dw, ws, ws_bytes, s);
}
}
