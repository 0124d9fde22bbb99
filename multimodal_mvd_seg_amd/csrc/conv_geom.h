// Generic "gathered-tap GEMM" problem descriptors shared by the scalar and the MFMA conv engines.
//
// Forward-type problem (conv fwd, conv dgrad, convT fwd, convT dgrad all reduce to it):
//   Y[n, o*so + oo, k] = bias[k] + sum_{t < ntaps} sum_{c < C1+C2} A[n, o*sa + off[t], c] * W[wt[t]][c][k]
// with o over the iteration grid (Do,Ho,Wo), A = channel concat of a1 [.,C1] and a2 [.,C2] with spatial dims
// (Di,Hi,Wi) (reads outside are zero), Y = channel split into y1 [.,K1] and y2 [.,K2] with spatial dims (Dy,Hy,Wy).
//
// Wgrad-type problem (conv wgrad, convT wgrad):
//   dW[wt[t]][c][k] = sum_n sum_o A[n, o*sa + off[t], c] * B[n, o*sb + ob[t], k]
// B [.,K] with spatial dims (Db,Hb,Wb) (reads outside are zero).
#pragma once
#include <stdint.h>

namespace mvd {

// Packed weight layout shared by mvd_pack_weight, the scalar engine and the MFMA engine.  For a reduce dimension
// C (multiple of 4) the tensor is stored in chunks of CK = 8 (or 4) reduce-channels, each chunk tap-major, and inside
// a (chunk, tap) as [h][k][e] with c = cc*CK + h*(CK/2) + e: exactly the image one workgroup stages into LDS and
// the order the 32x32x2 fp32 MFMA consumes (lane half h takes CK/2 consecutive k-steps from one ds_read_b128/b64).
// C % 4 != 0 falls back to plain [t][c][k] (scalar engine only).
// CK = 32 whenever possible: a 32-channel chunk of an NDHWC voxel is one whole 128-byte line, so the staging pass of
// the MFMA kernels touches every input line exactly once (8-channel chunks re-fetched each line 4 times from HBM).
__host__ __device__ inline int wl_ck(int C) {
    return (C % 32 == 0) ? 32 : ((C % 8 == 0) ? 8 : ((C % 4 == 0) ? 4 : 0));
}
__host__ __device__ inline size_t widx(int CK, int T, int C, int K, int t, int c, int k) {
    if (CK == 0) return ((size_t)t * C + c) * K + k;
    const int hh = CK / 2;
    const int cc = c / CK, r = c % CK;
    return ((((size_t)cc * T + t) * 2 + r / hh) * K + k) * hh + (r % hh);
}

struct FwdGeom {
    int N;
    int T;  // taps of the full weight tensor (packed-layout stride)
    int Di, Hi, Wi;
    int Do, Ho, Wo;
    int Dy, Hy, Wy;
    int C1, C2, K1, K2;
    int ntaps;
    int sa[3], so[3], oo[3];
    int8_t off[27][3];
    int8_t wt[27];
    int8_t acc;  // 1: y1 += result (the gradient of a tensor with a second consumer, ops._GradShare); engines that support it:
                 // k_fwd16 / k_split_reduce16 (bf16 generic), k_dgrad32s (fp32); the others refuse (-1)
};

struct WgradGeom {
    int N;
    int Di, Hi, Wi;  // A dims
    int Db, Hb, Wb;  // B dims
    int Do, Ho, Wo;  // iteration grid
    int C1, C2, K;
    int ntaps;       // taps of this launch
    int T;           // taps of the full weight tensor (layout stride)
    int sa[3], sb[3];
    int8_t off[27][3], ob[27][3];
    int8_t wt[27];
    int transposed_out;  // 0: dw[k][c][T] (Conv3d), 1: dw[c][k][T] (ConvTranspose3d)
};

// engines (return 0 ok, >0 error, -1 = shape not supported by this engine)
int fwd_scalar(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1,
               float *y2, hipStream_t s);
// ws/ws_bytes: optional scratch for the split-reduce path of skinny problems (few tiles, many reduce channels)
size_t fwd_mfma_ws(int N, long out_vox, int K);
int fwd_mfma(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1, float *y2,
             void *ws, size_t ws_bytes, hipStream_t s);
size_t wgrad_scalar_ws(const WgradGeom &g);
int wgrad_scalar(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws,
                 size_t ws_bytes, hipStream_t s);
size_t wgrad_mfma_ws(const WgradGeom &g);
// dbias / dbias_done (optional): engines that see every dy value anyway (the 2-D Winograd kernel) also produce the bias
// gradient and set *dbias_done = 1; otherwise the caller runs the column-sum kernel
int wgrad_mfma(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws, size_t ws_bytes,
               hipStream_t s, bool bf16_in = false, float *dbias = nullptr, int *dbias_done = nullptr);

// fp32 Winograd F(2,3)-along-W engine for plain 3x3x3 stride-1 problems (conv_wino.hip); u = Winograd-domain weights
// stats / stats_done (optional): the F(2x2,3x3) kernel can also emit per-tile (sum y, sum y^2) per output channel,
// [n][tile][K][2] floats with tile = 4x4x8-voxel tiles in raster order (the InstanceNorm statistics epilogue)
int fwd_wino(const FwdGeom &g, const float *a1, const float *a2, const float *u, const float *bias, float *y1, float *y2,
             hipStream_t s, float *stats = nullptr, int *stats_done = nullptr);
int pack_weight_wino(const float *w, float *uf, float *ub, int K, int C, hipStream_t s);
int dgrad32s(int N, int D, int H, int W, int C, int K, int Do, int Ho, int Wo, const float *dy, const float *wb, float *dx,
             hipStream_t s, int accumulate = 0);
// bf16 twin (conv_bf16.hip): all eight parity classes of a 3x3x3 stride-2 conv's input gradient from one staged dy tile
int dgrad16s(int N, int D, int H, int W, int C, int K, int Do, int Ho, int Wo, const unsigned short *dy, const unsigned short *wb,
             unsigned short *dx, hipStream_t s, int accumulate = 0);
// bf16 z-marching stride-2 forward conv 32 -> 64 channels (conv_bf16s.hip): -1 = not this kernel's shape
int launch_fwd16ys(const FwdGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *w, const float *bias,
                   unsigned short *y1, unsigned short *y2, hipStream_t s, int ncu);
// bf16 z-marching weight gradient of the plain 3x3x3 stride-1 conv (conv_bf16w.hip): -1 = not this kernel's problem, 0 = the
// partials [nsplit][27][C][K] (and, if asked, bias rows [nsplit][K] at *pbias_out) are in ws
int wgrad16z(const WgradGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *b, void *ws,
             size_t ws_bytes, bool want_bias, int *nsplit_out, float **pbias_out, hipStream_t s, const float *in_scale = nullptr,
             const float *in_shift = nullptr, float slope = 0.f);
// the same + the split reduce into dw / dbias (conv_mfma.hip)
int wgrad16z_run(const WgradGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *b, float *dw, void *ws,
                 size_t ws_bytes, hipStream_t s, float *dbias, int *dbias_done, const float *in_scale, const float *in_shift,
                 float slope);
bool wgrad16z_prologue_ok(const WgradGeom &g);
// stride-2 twin (k_wgrad16zs): partials only, the bias gradient stays with the caller's column-sum pass
int wgrad16zs(const WgradGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *b, void *ws,
              size_t ws_bytes, int *nsplit_out, hipStream_t s);
void wgrad16z_enable(int on);
int pack_weights_batch(int n, const float *const *w, float *const *wf, float *const *wb, float *const *uf, float *const *ub,
                       const int *K, const int *C, const int *T, const int *transposed, hipStream_t s);
int wino_mode();                          // MVD_WINO: 0 off, 1 F(2,3) along W, 2 F(2x2,3x3) (default)
size_t wino_weight_elems(int C, int K);   // floats of one uf / ub buffer in the active mode

// ConvTranspose3d (kernel == stride) as LDS-free GEMMs (conv_transp.hip); -1 = shape not covered
int convT_fwd_direct(const float *x, const float *wf, const float *bias, float *y, int N, int D, int H, int W, int C, int K,
                     const int st[3], hipStream_t s);
int convT_dgrad_direct(const float *dy, const float *wb, float *dx, int N, int D, int H, int W, int C, int K,
                       const int st[3], hipStream_t s);

int convT_fwd_direct16(const unsigned short *x, const unsigned short *wf, const float *bias, unsigned short *y, int N, int D,
                       int H, int W, int C, int K, const int st[3], hipStream_t s);
int convT_dgrad_direct16(const unsigned short *dy, const unsigned short *wb, unsigned short *dx, int N, int D, int H, int W,
                         int C, int K, const int st[3], hipStream_t s);

// bf16 forward-type engine (conv_bf16.hip): bf16 activations / packed weights, fp32 accumulate, bf16 output
// Optional InstanceNorm fusion of the z-marching kernel k_fwd16z (get_network_from_plans.py:41-44: conv -> InstanceNorm3d
// -> LeakyReLU): `tile_stats` != null asks for the per-workgroup (sum, sum of squares) of the bf16-rounded OUTPUT per
// channel ([N][ntiles][K][2] floats; *ntiles is set to the tiles per sample, or to 0 when the kernel that ran cannot emit
// them); `in_scale` / `in_shift` != null ([N][C1] floats) make the kernel normalise + activate its INPUT while staging it:
// a = bf16(lrelu(fma(x, scale, shift))) -- the InstanceNorm-apply of the producing block folded into this consumer's
// loader (zero padding applies to a, i.e. halo voxels outside the volume stay 0).  `required` = fail (-1) instead of
// running un-fused when the shape does not take the z-marching kernel.
struct Fwd16Fuse {
    float *tile_stats = nullptr;
    int *ntiles = nullptr;
    const float *in_scale = nullptr, *in_shift = nullptr;
    float slope = 0.01f;
};
int fwd_bf16(const FwdGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *w,
             const float *bias, unsigned short *y1, unsigned short *y2, void *ws, size_t ws_bytes, hipStream_t s,
             const Fwd16Fuse *fuse = nullptr);
int fwd16y_enabled();
void fwd16y_enable(int on);
// k_fwd16y (conv_bf16y.hip): -1 = not this kernel's shape; stats_tiles_only != null = query (no launch)
int launch_fwd16y(const FwdGeom &g, const unsigned short *a1, const unsigned short *a2, const unsigned short *w,
                  const float *bias, unsigned short *y1, unsigned short *y2, hipStream_t s, const Fwd16Fuse *fuse, int ncu,
                  int *stats_tiles_only);
// tiles per sample the z-marching kernel would emit statistics for (0: this shape does not run on it)
int fwd_bf16_stats_tiles(const FwdGeom &g);
// 1 when the shape runs on a kernel that has the InstanceNorm input prologue (Fwd16Fuse::in_scale / in_shift)
int fwd_bf16_prologue_ok(const FwdGeom &g);
int pack_weight16(const float *w, unsigned short *wf, unsigned short *wb, int K, int C, int T, int transposed,
                  hipStream_t s);
int pack_weights16_batch(int n, const float *const *w, unsigned short *const *wf, unsigned short *const *wb, const int *K,
                         const int *C, const int *T, const int *transposed, hipStream_t s);

}  // namespace mvd
