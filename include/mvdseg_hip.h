/* mvdseg_hip.h -- C ABI of libmvdseg_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the one
 * data-parallel hot path of JaronTu/Multimodal_MVD_Seg: the nnU-Net-v2 3d_fullres PlainConvUNet train step
 * (SURVEY.md section 8).  The reference has NO FFI on this path (it calls torch.nn -> ATen -> cuDNN/MIOpen);
 * every entry point below cites the reference call site whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller (the PyTorch caching
 *    allocator on the Python side).  The library never allocates or frees device memory and keeps no pointer
 *    across calls.  `stream` is a hipStream_t passed as void* (the caller's current stream).
 *  - return 0 on success, non-zero otherwise; no exception crosses the ABI; mvd_last_error() returns the
 *    thread-local message of the last failing call.
 *  - activations are NDHWC fp32 ("channels last 3d"): x[n][d][h][w][c].  Logits, targets and the 1-channel maps
 *    of the topology losses are planar NCDHW fp32.  Weights cross the boundary in torch layout
 *    (Conv3d [K][C][kd][kh][kw], ConvTranspose3d [C][K][kd][kh][kw]); mvd_pack_* builds the tap-major shadow
 *    copies the kernels stream.
 *  - every reduction is a fixed-order two-stage reduce (no float atomics): results are run-to-run bit-identical.
 *  - workspaces: caller passes `ws` / `ws_bytes`; mvd_*_workspace_bytes() returns the requirement.
 */
#ifndef MVDSEG_HIP_H
#define MVDSEG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVD_VERSION 100 /* 0.1.0 */
#define MVD_MAX_TAPS 27

int mvd_version(void);
const char *mvd_last_error(void);
/* 1 if the build contains the MFMA implicit-GEMM engine (it always does; kept for host-side asserts) */
int mvd_has_mfma(void);
/* force the scalar gather kernels (debug / cross-check): 0 = auto (MFMA when shapes allow), 1 = scalar only */
int mvd_set_conv_engine(int mode);
/* debug/test hook: minimum number of 128-voxel x 32-channel work items for the Winograd conv kernel (n < 0: default) */
int mvd_set_wino_min_items(long n);

/* ------------------------------------------------------------------------------------------------------------
 * Weight packing.  torch Conv3d weight [K][C][T] (T = kd*kh*kw taps, row-major) ->
 *   wf[T][C][K]  (forward: reduce over C, produce K)      wb[T][K][C]  (dgrad: reduce over K, produce C)
 * `transposed` != 0: source is a ConvTranspose3d weight [C][K][T] (same two outputs).
 * Replaces nothing in the reference (torch keeps one layout); it is the price of the tap-major layout. */
int mvd_pack_weight(const float *w, float *wf, float *wb, int K, int C, int T, int transposed, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Conv3d, kernel k[a] in {1,3}, padding (k-1)/2, stride s[a] in {1,2}, dilation 1 (K1/K5/K6 of SURVEY 2.4).
 * Replaces torch.nn.Conv3d inside every ConvDropoutNormReLU (get_network_from_plans.py:39-45,
 * UNetDecoder.py:61-65) and the decoder's torch.cat((x, skip), 1) (UNetDecoder.py:107): the input is the channel
 * concatenation of x1 [N,D,H,W,C1] and x2 [N,D,H,W,C2] (x2 may be NULL with C2 = 0).
 *   y[n,o,k] = bias[k] + sum_t sum_c x[n, o*s + t - pad, c] * wf[t][c][k],   y: [N,Do,Ho,Wo,K]
 * Do = (D + 2*pad - k)/s + 1. */
/* ws (optional, may be NULL): scratch for the split-reduce path the engine takes on skinny problems (few output
 * voxels, hundreds of reduce channels: the 8^3 / 4^3 stages); size from mvd_conv_fwd_workspace_bytes(N, output voxels
 * per sample, output channels).  Same for dgrad / convT fwd / convT dgrad (output = the tensor being produced). */
size_t mvd_conv_fwd_workspace_bytes(int N, long out_voxels, int K);
int mvd_conv3d_fwd(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *bias, float *y,
                   int N, int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws,
                   size_t ws_bytes, void *stream);
/* dgrad: dx = conv_transpose(dy, w); written as dx1 [.,C1] and dx2 [.,C2] (channel split of the concat). */
int mvd_conv3d_dgrad(const float *dy, const float *wb, float *dx1, int C1, float *dx2, int C2, int N, int D, int H,
                     int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream);
/* wgrad: dw in TORCH layout [K][C1+C2][T], dbias [K] (may be NULL).  Fixed-order split reduction through `ws`. */
/* Winograd variants of the two entries above for 3x3x3 stride-1 convs with C % 32 == 0, K % 32 == 0 (every conv of
 * the network but the input layer and the strided ones): F(2x2,3x3) over H and W -- 4/9 of the multiply-adds of
 * the direct form (env MVD_WINO=1 selects F(2,3) along W only: 2/3) -- fp32 throughout, same results within fp32
 * round-off (tests: <= 1e-5 relative to fp64).
 * mvd_pack_weight_wino: torch weight [K][C][3][3][3] -> uf (forward) / ub (input gradient), mvd_wino_weight_elems(C, K)
 * floats each, in the MFMA operand order of the active mode.  The *_wino conv entries take the direct
 * packed weights as well and fall back to the direct engines (same results) for shapes the Winograd kernel does
 * not cover (strides, 1x1x1, fewer than 256 work items, uf/ub == NULL). */
/* bit 0: the forward would use the Winograd kernel, bit 1: the input gradient would (0: skip packing uf/ub) */
int mvd_conv_wino_applicable(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3], const int stride[3]);
/* floats in one uf / ub buffer (48*C*K for the default F(2x2,3x3) mode, 36*C*K for F(2,3) along W) */
size_t mvd_wino_weight_elems(int C, int K);
/* 0 = direct engines only, 1 = F(2,3) along W, 2 = F(2x2,3x3) over H and W (default; env MVD_WINO) */
int mvd_wino_mode(void);
int mvd_pack_weight_wino(const float *w, float *uf, float *ub, int K, int C, void *stream);
/* Every conv weight of a network re-packed in ONE launch (the per-layer pack entries are launch-bound; called by the
 * fused optimizer right after the SGD update).  Host tables of n jobs: w[q] the torch-layout weight, wf/wb the
 * mvd_pack_weight outputs, uf/ub the mvd_pack_weight_wino outputs (null entries are skipped; uf/ub need MVD_WINO=2). */
int mvd_pack_weights_batch(int n, const float *const *w, float *const *wf, float *const *wb, float *const *uf,
                           float *const *ub, const int *K, const int *C, const int *T, const int *transposed,
                           void *stream);
int mvd_conv3d_fwd_wino(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *uf,
                        const float *bias, float *y, int N, int D, int H, int W, int K, const int ksize[3],
                        const int stride[3], void *ws, size_t ws_bytes, void *stream);
/* InstanceNorm statistics epilogue ("optional sum x / sum x^2 epilogue"): as mvd_conv3d_fwd_wino; when the
 * Winograd kernel runs it also writes per-tile (sum y, sum y^2) per output channel to stats
 * [N][mvd_conv_stats_tiles(D,H,W)][K][2] floats and sets *stats_done = 1 (0: not produced, run the plain norm). */
size_t mvd_conv_stats_tiles(int D, int H, int W);
int mvd_conv3d_fwd_wino_stats(const float *x1, int C1, const float *x2, int C2, const float *wf, const float *uf,
                              const float *bias, float *y, float *stats, int *stats_done, int N, int D, int H, int W,
                              int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream);
int mvd_conv3d_dgrad_wino(const float *dy, const float *wb, const float *ub, float *dx1, int C1, float *dx2, int C2, int N,
                          int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                          void *stream);
size_t mvd_conv3d_wgrad_workspace_bytes(int C, int K, int T, int N, int Do, int Ho, int Wo);
int mvd_conv3d_wgrad(const float *x1, int C1, const float *x2, int C2, const float *dy, float *dw, float *dbias,
                     int N, int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws,
                     size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * ConvTranspose3d with kernel == stride (non-overlapping; K2).  Replaces the decoder's transpconvs
 * (UNetDecoder.py:56-59,106).  x: [N,D,H,W,C] -> y: [N,D*s0,H*s1,W*s2,K];  wf[T][C][K], T = s0*s1*s2.
 *   y[n, q*s + p, k] = bias[k] + sum_c x[n,q,c] * w[c][k][p] */
int mvd_convT3d_fwd(const float *x, const float *wf, const float *bias, float *y, int N, int D, int H, int W, int C,
                    int K, const int stride[3], void *ws, size_t ws_bytes, void *stream);
int mvd_convT3d_dgrad(const float *dy, const float *wb, float *dx, int N, int D, int H, int W, int C, int K,
                      const int stride[3], void *ws, size_t ws_bytes, void *stream);
size_t mvd_convT3d_wgrad_workspace_bytes(int C, int K, int T, int N, int D, int H, int W);
/* dw in TORCH layout [C][K][T], dbias [K] */
int mvd_convT3d_wgrad(const float *x, const float *dy, float *dw, float *dbias, int N, int D, int H, int W, int C,
                      int K, const int stride[3], void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * bf16 mixed precision (BASELINE cfg 4/5: the reference's autocast path, nnUNetTrainer.py:906): activations and
 * packed weights are bf16 (uint16_t storage), accumulation and bias fp32, outputs bf16; master weights and all
 * weight gradients stay fp32.  C % 32 == 0 and K % 32 == 0 (the 4-modality input layer: mvd_pad_channels_bf16 first).
 * mvd_pack_weight_bf16: torch fp32 weight -> wf16 (reduce C, produce K) / wb16 (reduce K, produce C), layout
 * [chunk32][tap][kstep][lane half][out channel][8]. */
int mvd_pack_weight_bf16(const float *w, uint16_t *wf, uint16_t *wb, int K, int C, int T, int transposed, void *stream);
/* Every bf16 pack of a network in ONE launch (called by the fused optimizer after its update, like
 * mvd_pack_weights_batch for the fp32 layouts): host tables of n jobs, outputs bit-identical to mvd_pack_weight_bf16. */
int mvd_pack_weights_bf16_batch(int n, const float *const *w, uint16_t *const *wf, uint16_t *const *wb, const int *K,
                                const int *C, const int *T, const int *transposed, void *stream);
int mvd_conv3d_fwd_bf16(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const uint16_t *wf, const float *bias,
                        uint16_t *y, int N, int D, int H, int W, int K, const int ksize[3], const int stride[3],
                        void *ws, size_t ws_bytes, void *stream);
int mvd_conv3d_dgrad_bf16(const uint16_t *dy, const uint16_t *wb, uint16_t *dx1, int C1, uint16_t *dx2, int C2, int N,
                          int D, int H, int W, int K, const int ksize[3], const int stride[3], void *ws, size_t ws_bytes,
                          void *stream);
int mvd_convT3d_fwd_bf16(const uint16_t *x, const uint16_t *wf, const float *bias, uint16_t *y, int N, int D, int H, int W,
                         int C, int K, const int stride[3], void *ws, size_t ws_bytes, void *stream);
int mvd_convT3d_dgrad_bf16(const uint16_t *dy, const uint16_t *wb, uint16_t *dx, int N, int D, int H, int W, int C, int K,
                           const int stride[3], void *ws, size_t ws_bytes, void *stream);
/* weight gradients from bf16 activations / bf16 dy: operands widened to fp32 in registers, fp32 MFMA, fp32 partials,
 * fp64 fixed-order reduce -> fp32 dw (torch layout) and fp32 dbias.  Workspace sizes: the fp32 queries above. */
int mvd_conv3d_wgrad_bf16(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const uint16_t *dy, float *dw,
                          float *dbias, int N, int D, int H, int W, int K, const int ksize[3], const int stride[3],
                          void *ws, size_t ws_bytes, void *stream);
int mvd_convT3d_wgrad_bf16(const uint16_t *x, const uint16_t *dy, float *dw, float *dbias, int N, int D, int H, int W,
                           int C, int K, const int stride[3], void *ws, size_t ws_bytes, void *stream);
/* The fused block of the north_star in bf16 -- Conv3d 3x3x3 -> InstanceNorm3d(affine) -> LeakyReLU
 * (get_network_from_plans.py:41-44; block composition UNetDecoder.py:61-65 / PlainConvEncoder) -- on the z-marching
 * conv kernel (3x3x3, stride 1, 32 -> 32 channels, large volumes):
 *   mvd_conv3d_fwd_bf16_stats_tiles: tiles per sample the conv would emit statistics for (0: this shape runs on a kernel
 *     without the epilogue; the caller then uses mvd_instnorm_lrelu_fwd_bf16).
 *   mvd_conv3d_fwd_bf16_fused: mvd_conv3d_fwd_bf16 plus
 *     - tile_stats != NULL: per-workgroup (sum, sum of squares) of the bf16-rounded output per channel,
 *       [N][*ntiles_out][K][2] floats (*ntiles_out = 0 on return: the kernel that ran has no epilogue);
 *     - in_scale / in_shift != NULL ([N][C1] floats): the INPUT is the raw output of the producing conv and is
 *       normalised + activated while it is staged, a = bf16(lrelu(fma(x, scale, shift))) with zero padding applied to a
 *       -- the producing block's InstanceNorm-apply + LeakyReLU folded into this consumer's loader, so the activated
 *       tensor is never written to HBM.  Error 3 when the shape does not take the z-marching kernel.
 *   mvd_instnorm_finalize_tiles: tile statistics -> mean, rstd (biased variance, eps inside the root), scale = gamma*rstd,
 *     shift = beta - mean*scale; one launch, fp64, fixed order.
 *   mvd_instnorm_lrelu_apply_bf16: y = bf16(lrelu(fma(x, scale, shift))) -- the same arithmetic as the fused prologue, for
 *     the consumers that have none.  The backward pass is mvd_instnorm_lrelu_bwd_bf16 on (x, mean, rstd). */
int mvd_conv3d_fwd_bf16_stats_tiles(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3],
                                    const int stride[3]);
/* Gradient of a tensor with TWO consumers (a skip connection: the next encoder stage's strided conv and the decoder conv
 * that reads it as its second pointer, UNetDecoder.py:106-108): the consumer whose backward runs second adds its input
 * gradient into the buffer the first one wrote, inside its own kernel, instead of autograd's separate elementwise add.
 *   mvd_conv3d_dgrad_acc_ok: 1 when an accumulating kernel exists for the conv (3x3x3, stride 2, 32-multiple channels).
 *   mvd_conv3d_dgrad_acc / _bf16_acc: dx1 += input gradient (fp32: a + b as torch's add; bf16: fp32 add of the two bf16
 *   values, one rounding, as torch's bf16 add). */
int mvd_conv3d_dgrad_acc_ok(int is_bf16, int N, int D, int H, int W, int C1, int K, const int ksize[3], const int stride[3]);
int mvd_conv3d_dgrad_acc(const float *dy, const float *wb, float *dx1, int C1, int N, int D, int H, int W, int K,
                         const int ksize[3], const int stride[3], void *stream);
int mvd_conv3d_dgrad_bf16_acc(const uint16_t *dy, const uint16_t *wb, uint16_t *dx1, int C1, int N, int D, int H, int W, int K,
                              const int ksize[3], const int stride[3], void *ws, size_t ws_bytes, void *stream);
/* which z-marching kernel serves the 3x3x3 stride-1 convs with 32 produce channels at the patch resolution: 1 (default,
 * MVD_FWD16Y) = k_fwd16y (16x16x32 tiles; 32 or 64 reduce channels; statistics epilogue and loader prologue), 0 =
 * k_fwd16z (32x32x16 tiles, 32 reduce channels, loader prologue only).  A/B and cross-check switch. */
int mvd_set_bf16_zmarch_kernel(int which);
/* which kernel computes the bf16 weight gradient of the plain 3x3x3 stride-1 convs on volumes at least 32 wide
 * (nnUNetTrainer.py:888-925 backward): 1 (default, MVD_WGRAD16Z) = k_wgrad16z (z-marching column, csrc/conv_bf16w.hip),
 * 0 = k_wgrad16 (4x8x8 tiles).  A/B and cross-check switch. */
int mvd_set_bf16_wgrad_kernel(int which);
/* Weight gradient with the loader prologue (ops.NormActConv3dFn.backward; get_network_from_plans.py:41-44 fused block, the
 * backward of nnUNetTrainer.py:888-925): x1 is the RAW bf16 output of the producing conv, the operand of the product is
 * lrelu(x1 * in_scale[n][c] + in_shift[n][c]) rounded to bf16 -- bit-identical to mvd_conv3d_wgrad_bf16 over the tensor
 * mvd_instnorm_lrelu_apply_bf16 would write, which therefore need not exist.  _prologue_ok: 1 when the shape is served. */
int mvd_conv3d_wgrad_bf16_prologue_ok(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3], const int stride[3]);
int mvd_conv3d_wgrad_bf16_fused(const uint16_t *x1, int C1, const uint16_t *dy, float *dw, float *dbias, int N, int D, int H, int W,
                                int K, const int ksize[3], const int stride[3], const float *in_scale, const float *in_shift,
                                float slope, void *ws, size_t ws_bytes, void *stream);
/* 1 when mvd_conv3d_fwd_bf16_fused accepts in_scale / in_shift for this shape (a kernel with the loader prologue runs it) */
int mvd_conv3d_fwd_bf16_prologue_ok(int N, int D, int H, int W, int C1, int C2, int K, const int ksize[3],
                                    const int stride[3]);
int mvd_conv3d_fwd_bf16_fused(const uint16_t *x1, int C1, const uint16_t *x2, int C2, const uint16_t *wf, const float *bias,
                              uint16_t *y, int N, int D, int H, int W, int K, const int ksize[3], const int stride[3],
                              const float *in_scale, const float *in_shift, float slope, float *tile_stats, int *ntiles_out,
                              void *ws, size_t ws_bytes, void *stream);
int mvd_instnorm_finalize_tiles(const float *tile_stats, long ntiles, const float *gamma, const float *beta, float *mean,
                                float *rstd, float *scale, float *shift, int N, long V, int C, float eps, void *stream);
int mvd_instnorm_lrelu_apply_bf16(const uint16_t *x, const float *scale, const float *shift, uint16_t *y, int N, long V,
                                  int C, float slope, void *stream);
/* statistics pass alone (for a conv whose kernel has no epilogue): mean, rstd and scale / shift of x (fp32 or bf16 NDHWC);
 * workspace: mvd_instnorm_workspace_bytes */
int mvd_instnorm_stats_bf16(const void *x, int x_is_bf16, const float *gamma, const float *beta, float *mean, float *rstd,
                            float *scale, float *shift, int N, long V, int C, float eps, void *ws, size_t ws_bytes,
                            void *stream);
/* InstanceNorm+LeakyReLU with bf16 output (statistics and arithmetic fp32/fp64 as in the fp32 entry points).
 * x is fp32 (x_is_bf16 == 0: the first, fp32, conv of the network) or bf16; y and dy are bf16; dx has x's type.
 * Workspace: mvd_instnorm_workspace_bytes.  C % 4 == 0. */
int mvd_instnorm_lrelu_fwd_bf16(const void *x, int x_is_bf16, const float *gamma, const float *beta, uint16_t *y,
                                float *mean, float *rstd, int N, long V, int C, float eps, float slope, void *ws,
                                size_t ws_bytes, void *stream);
int mvd_instnorm_lrelu_bwd_bf16(const void *x, int x_is_bf16, const uint16_t *dy, const float *gamma, const float *beta,
                                const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta, int N,
                                long V, int C, float slope, void *ws, size_t ws_bytes, void *stream);
/* seg head on bf16 activations: fp32 weights, fp32 planar logits (the loss kernels stay fp32), bf16 dx, fp32 dw/dbias */
int mvd_seghead_fwd_bf16(const uint16_t *x, const float *w, const float *bias, float *logits, int N, long V, int C,
                         int K, void *stream);
int mvd_seghead_bwd_bf16(const uint16_t *x, const float *w, const float *dlogits, uint16_t *dx, float *dw, float *dbias,
                         int N, long V, int C, int K, int accumulate, void *ws, size_t ws_bytes, void *stream);
/* The seg head of the LAST decoder stage reading the RAW bf16 output of that stage's last conv (UNetDecoder.py:110 after
 * get_network_from_plans.py:41-44): a = bf16(lrelu(x * scale[n][c] + shift[n][c])) is applied in the loaders, the activated
 * tensor is not materialised.  _bwd: dx = d a (the caller runs the InstanceNorm backward on it), dw / dbias over the
 * re-computed a.  Bit-identical to mvd_seghead_*_bf16 over the tensor mvd_instnorm_lrelu_apply_bf16 would write. */
int mvd_seghead_bf16_fused_ok(int N, long V, int C, int K);
/* fp32 twins (mean / rstd / gamma / beta: the arithmetic of the fp32 apply pass); same eligibility query */
int mvd_seghead_fwd_fused(const float *x, const float *mean, const float *rstd, const float *gamma, const float *beta, float slope,
                          const float *w, const float *bias, float *logits, int N, long V, int C, int K, void *stream);
int mvd_seghead_bwd_fused(const float *x, const float *mean, const float *rstd, const float *gamma, const float *beta, float slope,
                          const float *w, const float *dlogits, float *dx, float *dw, float *dbias, int N, long V, int C, int K,
                          int accumulate, void *ws, size_t ws_bytes, void *stream);
/* mean / rstd from the fp32 conv epilogue's tile statistics without the apply pass (the first half of
 * mvd_instnorm_lrelu_fwd_prestats) */
int mvd_instnorm_stats_from_tiles(const float *tile_stats, long ntiles, float *mean, float *rstd, int N, long V, int C, float eps,
                                  void *ws, size_t ws_bytes, void *stream);
int mvd_seghead_fwd_bf16_fused(const uint16_t *x, const float *scale, const float *shift, float slope, const float *w,
                               const float *bias, float *logits, int N, long V, int C, int K, void *stream);
int mvd_seghead_bwd_bf16_fused(const uint16_t *x, const float *scale, const float *shift, float slope, const float *w,
                               const float *dlogits, uint16_t *dx, float *dw, float *dbias, int N, long V, int C, int K,
                               int accumulate, void *ws, size_t ws_bytes, void *stream);
/* flat conversions (round-to-nearest-even) for tensors that cross the precision boundary (distillation features) */
int mvd_cast_f32_to_bf16(const float *src, uint16_t *dst, long n, void *stream);
int mvd_cast_bf16_to_f32(const uint16_t *src, float *dst, long n, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * InstanceNorm3d(eps, affine) + LeakyReLU(slope), fused (K3/K4).  Replaces nn.InstanceNorm3d + nn.LeakyReLU of
 * every ConvDropoutNormReLU (get_network_from_plans.py:41-44).  x,y: [N,V,C] NDHWC with V = D*H*W.
 * Biased variance over V, no running stats.  mean/rstd [N][C] are outputs (saved for backward).
 * ws: 2*N*nblk*C doubles (nblk from mvd_instnorm_nblk). */
int mvd_instnorm_nblk(int N, long V, int C);
size_t mvd_instnorm_workspace_bytes(int N, long V, int C);
int mvd_instnorm_lrelu_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                           float *rstd, int N, long V, int C, float eps, float slope, void *ws, size_t ws_bytes,
                           void *stream);
/* dx [N,V,C]; dgamma/dbeta [C] (overwritten).  Recomputes z = xhat*gamma+beta from x for the LeakyReLU mask. */
/* forward with the statistics taken from the producing conv's epilogue (mvd_conv3d_fwd_wino_stats): no pass over x
 * for mean / variance; everything else as mvd_instnorm_lrelu_fwd */
int mvd_instnorm_lrelu_fwd_prestats(const float *x, const float *tile_stats, long ntiles, const float *gamma,
                                    const float *beta, float *y, float *mean, float *rstd, int N, long V, int C, float eps,
                                    float slope, void *ws, size_t ws_bytes, void *stream);
int mvd_instnorm_lrelu_bwd(const float *x, const float *dy, const float *gamma, const float *beta,
                           const float *mean, const float *rstd, float *dx, float *dgamma, float *dbeta, int N,
                           long V, int C, float slope, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * 1x1x1 segmentation head (K5): x NDHWC [N,V,C] -> logits planar [N,K,V].  Replaces decoder.seg_layers[s]
 * (UNetDecoder.py:70,110).  w: torch layout [K][C], bias [K]. */
int mvd_seghead_fwd(const float *x, const float *w, const float *bias, float *logits, int N, long V, int C, int K,
                    void *stream);
/* dx [N,V,C] (overwritten, or accumulated into when accumulate != 0); dw [K][C], dbias [K] */
size_t mvd_seghead_bwd_workspace_bytes(int N, long V, int C, int K);
int mvd_seghead_bwd(const float *x, const float *w, const float *dlogits, float *dx, float *dw, float *dbias, int N,
                    long V, int C, int K, int accumulate, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused softmax + cross-entropy + soft-Dice statistics (K7).  Replaces DC_and_CE_loss
 * (nnUNetTrainer.py:359-361: RobustCrossEntropyLoss + MemoryEfficientSoftDiceLoss{smooth 1e-5, do_bg False}).
 * logits planar [N,K,V]; target float labels [N,V] (robust_ce_loss.py:12-16 casts float->long).
 * fwd: stats[N][3K+1]: per sample (sum p*y, sum p, sum y) for k = 0..K-1, then the sample's sum_vox -log p_target
 * (fp64 accumulation, fixed order, rounded once to fp32).  This is also the tensor that is all-gathered for
 * DDP batch-dice (ddp_allgather.py:25-48, collective C2).
 * mvd_dcce_finalize does the scalar composition (dc = (2I+s)/clip(G+P+s,1e-8); mean over (n, k>=kstart);
 * batch_dice sums over n first) and emits the per-(n,k) Dice gradient coefficients used by bwd.  The Dice part may
 * be evaluated on a different sample set (`dstats`, Nd samples: the all-gathered stats) than the CE part. */
size_t mvd_dcce_workspace_bytes(int N, long V, int K);
int mvd_dcce_fwd(const float *logits, const float *target, float *stats /*[N][3K+1]*/, int N, long V, int K, void *ws,
                 size_t ws_bytes, void *stream);
/* loss[0] = w_ce*CE + w_dice*(-mean dc); loss[1] = CE; loss[2] = -mean dc;
 * coef[Nd][K][2] = w_dice * (dLdice/dI, dLdice/dP) (zero for k < kstart). */
int mvd_dcce_finalize(const float *stats, int N, const float *dstats, int Nd, float *loss, float *coef, long V, int K,
                      int batch_dice, int do_bg, float smooth, float w_ce, float w_dice, void *stream);
/* dlogits[n,k,v] = g * ( w_ce/(N*V) * (p_k - y_k) + p_k * (q_k - sum_j p_j q_j) ),  q_k = coefI[n,k]*y_k + coefP[n,k],
 * g = gscale_host * (gscale_dev ? gscale_dev[0] : 1) */
int mvd_dcce_bwd(const float *logits, const float *target, const float *coef /*[N][K][2]*/, const float *gscale_dev,
                 float gscale_host, float *dlogits, int N, long V, int K, float w_ce, void *stream);
/* p_sel[n,v] = softmax(logits[n,:,v])[sel] (MVDTrainer.py:907 takes channel 2 of the prediction for the topology
 * term; the build feeds the probability to soft-clDice).  bwd: dlogits[n,k,v] = g[n,v] * p_sel * ((k==sel) - p_k). */
int mvd_softmax_select_fwd(const float *logits, float *p_sel, int N, long V, int K, int sel, void *stream);
int mvd_softmax_select_bwd(const float *logits, const float *g, float *dlogits, int N, long V, int K, int sel,
                           void *stream);
/* mask[i] = (labels[i] == value) ? 1.f : 0.f  (one-hot channel `value` of the target, MVDTrainer.py:904-908) */
int mvd_label_mask(const float *labels, float *mask, long n, float value, void *stream);
/* validation: argmax -> one-hot -> tp/fp/fn per class over (n, v) (nnUNetTrainer.py:973-990).  counts[K][3] i64 */
int mvd_argmax_counts(const float *logits, const float *target, long long *counts, int N, long V, int K,
                      void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * KL distillation (K8).  Replaces distill_kl / l2_loss(channel_wise=True) (other_loss.py:51-64, :67-76):
 *   loss = T^2/(N*Ceff*V) * sum p_t * (log p_t - log_softmax(y_s/T + eps_s)),  p_t = softmax(y_t/T) over channels.
 * Element (n,c,v) lives at n*sn + c*sc + v*sv (planar: sc = V, sv = 1; NDHWC: sc = 1, sv = C).
 * pad_zero_channel != 0 implements the C == 1 branch (:55-57): a zero logit channel is appended (Ceff = 2).
 * out[0] = loss.  bwd writes g_s, g_t (either may be NULL) = gscale[0] * dloss/dy. */
size_t mvd_kl_workspace_bytes(int N, long V);
int mvd_kl_fwd(const float *ys, const float *yt, float *out, int N, int C, long V, long sn, long sc, long sv,
               float T, float eps_s, int pad_zero_channel, void *ws, size_t ws_bytes, void *stream);
int mvd_kl_bwd(const float *ys, const float *yt, const float *gscale_dev, float gscale_host, float *gs, float *gt,
               int N, int C, long V, long sn, long sc, long sv, float T, float eps_s, int pad_zero_channel,
               void *stream);
/* bf16 feature maps (mixed precision): dense NDHWC rows [N*V][C], C in {4,8,16,32}, no zero-channel padding; fp32
 * arithmetic, bf16 gradients.  Workspace: mvd_kl_workspace_bytes. */
int mvd_kl_fwd_bf16(const uint16_t *ys, const uint16_t *yt, float *out, int N, int C, long V, float T, float eps_s,
                    void *ws, size_t ws_bytes, void *stream);
int mvd_kl_bwd_bf16(const uint16_t *ys, const uint16_t *yt, const float *gscale_dev, float gscale_host, uint16_t *gs,
                    uint16_t *gt, int N, int C, long V, float T, float eps_s, void *stream);

/* Plain feature MSE: l2_loss(input, target, channel_wise=False) = mean(|a - b|^2) over all n elements
 * (nnunetv2/training/loss/other_loss.py:77-78).  Elementwise: any dense layout, both tensors in the same one.
 * out[0] = loss; bwd writes ga = g[0] * 2 (a - b) / n and gb = -ga (either may be NULL). */
size_t mvd_mse_workspace_bytes(long n);
int mvd_mse_fwd(const float *a, const float *b, float *out, long n, void *ws, size_t ws_bytes, void *stream);
int mvd_mse_bwd(const float *a, const float *b, const float *g_dev, float *ga, float *gb, long n, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Soft skeleton primitives (K9).  Replace soft_erode / soft_dilate of soft_skeleton.py:6-22 on planar volumes
 * [NC, D, H, W].  Bit-exact vs torch CPU (min/max only).  bwd reproduces autograd's routing: max_pool3d sends the
 * gradient to the FIRST maximum in scan order; torch.min(a,b) splits ties 1/2-1/2 (so the three axis pools of
 * erode get 1/4,1/4,1/2 on a triple tie). */
/* `code` (optional in fwd, required by bwd): per-voxel routing record written by fwd (uint16 for erode: arg-min
 * position per axis + tie pattern; uint8 for dilate: arg-max position 0..26 in the 3x3x3 window). */
int mvd_soft_erode_fwd(const float *x, float *y, uint16_t *code, int NC, int D, int H, int W, void *stream);
int mvd_soft_erode_bwd(const uint16_t *code, const float *dy, float *dx, int NC, int D, int H, int W, void *stream);
int mvd_soft_dilate_fwd(const float *x, float *y, uint8_t *code, int NC, int D, int H, int W, void *stream);
int mvd_soft_dilate_bwd(const uint8_t *code, const float *dy, float *dx, int NC, int D, int H, int W, void *stream);
/* skeleton update (soft_skeleton.py:31,35-36): delta = relu(img - opened); init: skel = delta;
 * else skel_out = skel + relu(delta - skel*delta) (no FMA contraction).  n = element count. */
int mvd_skel_update_fwd(const float *img, const float *opened, const float *skel_in, float *skel_out, long n,
                        int init, void *stream);
/* grads: d_img, d_opened (= -d_img masked), d_skel_in (NULL when init) from d_skel_out */
int mvd_skel_update_bwd(const float *img, const float *opened, const float *skel_in, const float *d_skel_out,
                        float *d_img, float *d_opened, float *d_skel_in, long n, int init, void *stream);
/* One whole soft_skel step (soft_skeleton.py:30-31 when init != 0, :33-36 otherwise) in one launch, LDS-tiled:
 *   init:  o = dilate(erode(img));                     skel_out = relu(img - o)
 *   else:  e1 = erode(img); o = dilate(erode(e1));     skel_out = skel_in + relu(d - skel_in * d), d = relu(e1 - o)
 * Writes e1 (the next iteration's img; NULL when init), opened = o (saved for the backward of the update) and, when the
 * pointers are non-NULL, the routing codes of the three stencils in the format of mvd_soft_erode_fwd /
 * mvd_soft_dilate_fwd, so mvd_soft_erode_bwd / mvd_soft_dilate_bwd / mvd_skel_update_bwd back-propagate it.  Bit-exact
 * with the primitive-per-launch chain. */
int mvd_skel_iter_fwd(const float *img, const float *skel_in, float *e1, float *opened, float *skel_out, uint16_t *c_e1,
                      uint16_t *c_e2, uint8_t *c_o, int NC, int D, int H, int W, int init, void *stream);
/* sums for soft-clDice: out[0] = sum a*b, out[1] = sum a  (fixed-order) */
int mvd_dot_sum(const float *a, const float *b, float *out, long n, void *ws, size_t ws_bytes, void *stream);
size_t mvd_dot_sum_workspace_bytes(long n);

/* soft-clDice scalar composition (clDice_metric.py:7-36 formula on soft skeletons):
 * sums = (sum skel_p*tgt, sum skel_p, sum skel_t*pred, sum skel_t); tprec = (s0+smooth)/(s1+smooth),
 * tsens = (s2+smooth)/(s3+smooth); out[0] = 1 - 2*tprec*tsens/(tprec+tsens); out[1..4] = d out[0] / d sums[0..3]. */
int mvd_cldice_combine(const float *sums, float *out, float smooth, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Connected components on a voxel grid (K12; integer, bit-exact).  Replaces the H0 / connected-component step the
 * reference runs on the CPU through the vendored TopologyLayer C++ (hom.cpp:51-69 restricted to vertices+edges
 * == union-find).  mask: uint8 [D,H,W]; labels int32 [D,H,W]: 0 = background, else 1 + smallest linear index of
 * the component (canonical).  conn in {6, 14, 26}.  count[0] = number of components. */
int mvd_cc_label(const uint8_t *mask, int32_t *labels, int32_t *count, int D, int H, int W, int conn, void *stream);
/* mask[v] = (f[v] > thr) (or >= when ge != 0) */
int mvd_threshold_mask(const float *f, uint8_t *mask, long n, float thr, int ge, void *stream);

/* H0 persistence: birth / death pairing of the connected components of the sub- (sublevel != 0) or super-level sets
 * of a scalar field f [D,H,W] on the grid graph with conn in {6, 14 (Freudenthal), 26}.  Replaces, for maxdim 0, the
 * CPU persistence of the vendored TopologyLayer C++: lower-star extension complex.cpp:136-146, filtration order
 * complex.cpp:182-196, reduction hom.cpp:51-69 (== union-find with the elder rule on vertices + edges), one bar per
 * vertex hom.cpp:155-185, called from functional/sublevel.py:21-49 / nn/levelset.py:137-163 after a D2H copy.
 *   device (mvd_h0_sorted_edges): keys_sorted[e] = all D*H*W*n_off 64-bit edge keys, ascending:
 *       (order-preserving bits of max(g[u], g[v])) << 32 | (v * n_off + k), g = f or -f; edges that leave the grid
 *       carry ~0 and sort last; the first mvd_h0_num_edges() keys are the filtration order of the 1-cells.
 *   host (mvd_h0_pair_host, plain C++): one elder-rule union-find sweep over those keys copied to host memory.
 *       death[v] / death_vertex[v] describe the bar born at vertex v (birth value f[v]): the value at which it dies
 *       (+inf, resp. -inf for super-level, for the essential bar of each component) and the critical (arg-max) vertex
 *       of the killing edge (-1 for essential bars) -- the `backprop_lookup` of hom.cpp:178-183.  Returns the number
 *       of essential bars, < 0 on error.
 * Integer / comparison work: bit-exact against oracle/cc_oracle.c, multiset-exact against the reference C++. */
long mvd_h0_num_edges(int D, int H, int W, int conn);
size_t mvd_h0_workspace_bytes(int D, int H, int W, int conn);
int mvd_h0_sorted_edges(const float *f, uint64_t *keys_sorted, int D, int H, int W, int conn, int sublevel, void *ws,
                        size_t ws_bytes, void *stream);
long mvd_h0_pair_host(const float *f_host, const uint64_t *keys_host, long n_edges, int D, int H, int W, int conn,
                      int sublevel, float *death, int64_t *death_vertex);

/* Connected-component post-processing of a predicted segmentation.  Replaces, on device,
 * remove_all_but_largest_component_from_segmentation (nnunetv2/postprocessing/remove_connected_components.py:22-34):
 *   mask = union over label_set of (seg == l)            -> mvd_seg_label_mask  (region_or_label_to_mask, :27-30)
 *   cc   = mvd_cc_label(mask, conn = 26)                  (skimage.measure.label full connectivity inside acvl_utils)
 *   kept = the `keep` (1 or 2) largest components         -> mvd_cc_keep_largest (remove_all_but_two_largest_component, :31)
 *   out  = seg, with mask & ~kept set to background       -> mvd_seg_remove_components (:32-33)
 * All integer; bit-exact against oracle/postproc_oracle.py.  Ties in component size go to the component with the
 * smaller canonical label (first in scan order).  label_set is a HOST array of 1..16 labels.  kept: int32[4] on the
 * device = {label0, label1, size0, size1} (0 = none).  workspace: mvd_cc_keep_workspace_bytes(n) device bytes. */
int mvd_seg_label_mask(const int32_t *seg, uint8_t *mask, long n, const int32_t *label_set, int nlabels, void *stream);
size_t mvd_cc_keep_workspace_bytes(long n);
int mvd_cc_keep_largest(const int32_t *cc_labels, long n, int keep, int32_t *kept, void *workspace, void *stream);
int mvd_seg_remove_components(const int32_t *seg, const int32_t *cc_labels, const int32_t *kept, int32_t *out, long n,
                              int background, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * SGD(momentum, nesterov, weight decay) with global-norm clipping (K10).  Replaces
 * torch.nn.utils.clip_grad_norm_(params, 12) + torch.optim.SGD.step (nnUNetTrainer.py:473-477, :918-924) on flat
 * parameter / gradient / momentum buffers of n floats.
 *   sumsq: out[0] = sum g^2 (fixed-order).  step: g *= grad_scale (1/world_size under data parallelism: the mean
 *   DDP takes after its SUM all-reduce, nnUNetTrainer.py:220-222; 1 otherwise); g *= min(1, max_norm /
 *   (grad_scale*sqrt(sumsq)+1e-6)) (read from device memory: no host sync); g += wd*p; buf = first ? g : mom*buf + g;
 *   g = g + mom*buf (nesterov); p -= lr*g */
size_t mvd_sumsq_workspace_bytes(long n);
int mvd_grad_sumsq(const float *g, float *out, long n, void *ws, size_t ws_bytes, void *stream);
int mvd_sgd_nesterov_step(float *p, const float *g, float *buf, const float *sumsq, long n, float lr, float momentum,
                          float weight_decay, float max_norm, float grad_scale, int first_step, void *stream);
/* the same step with its scalars read from a DEVICE array hyper[6] = {lr, momentum, weight_decay, max_norm, grad_scale,
 * first_step (0/1)}: a hipGraph-captured train step (nnUNetTrainer.py:888-925 replayed as one graph launch) follows the
 * PolyLR schedule (polylr.py:16-20, stepped per epoch :880) without re-capture.  sumsq must be given (max_norm <= 0 in
 * hyper[3] switches clipping off). */
int mvd_sgd_nesterov_step_dev(float *p, const float *g, float *buf, const float *sumsq, long n, const float *hyper,
                              void *stream);

/* misc elementwise helpers used by the host glue (all fixed-order / exact) */
int mvd_nchw_to_ndhwc(const float *src, float *dst, int N, int C, long V, void *stream);
int mvd_ndhwc_to_nchw(const float *src, float *dst, int N, int C, long V, void *stream);
/* bf16 mixed precision, narrow network input (4 modalities): fp32 [N][C][V] (src_ndhwc = 0) or [N][V][C] -> bf16
 * [N][V][Cpad] with zero channels C .. Cpad-1 (C <= 8, Cpad = 32), so that the first conv runs on the 32-channel bf16
 * engines (get_network_from_plans.py:41-44 under the autocast of nnUNetTrainer.py:906: bf16 operands, fp32 accumulate). */
int mvd_pad_channels_bf16(const float *src, uint16_t *dst, int N, int C, int Cpad, long V, int src_ndhwc, void *stream);
int mvd_axpy(float *y, const float *x, float a, long n, void *stream); /* y += a*x */

/* ------------------------------------------------------------------------------------------------------------
 * Sliding-window inference helpers.  Planar fp32 [C][D][H][W].
 * mvd_flip_add: dst (+)= flip(src) on the axes of mask (bit 0: D, 1: H, 2: W) -- the mirror test-time augmentation of
 *   predict_from_raw_data.py:562-588 (torch.flip of the input and of each prediction, summed);
 * mvd_sw_accumulate: logits[:, tile] += tile_logits * scale * gauss, npred[tile] += gauss (gauss NULL: weight 1)
 *   -- predict_from_raw_data.py:706-707;  mvd_sw_normalize: logits /= npred (:709). */
int mvd_flip_add(const float *src, float *dst, int C, int D, int H, int W, int mask, int accumulate, void *stream);
int mvd_sw_accumulate(const float *tile, const float *gauss, float scale, float *logits, float *npred, int K, int pd, int ph,
                      int pw, int D, int H, int W, int oz, int oy, int ox, void *stream);
int mvd_sw_normalize(float *logits, const float *npred, int K, long V, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * On-device training feed (the deterministic part).  Replaces, for case volumes resident in HBM, the
 * crop + constant pad of nnUNetDataLoader3D.generate_train_batch (dataloading/data_loader_3d.py:31-46: data padded
 * with 0, seg with -1), MirrorTransform (nnUNetTrainer.py:738-739), RemoveLabelTransform(-1, 0) (:745) and
 * DownsampleSegForDSTransform2 order 0 (deep_supervision_donwsampling.py:27-55) + NumpyToTensor 'float' (:768).
 * Planar layouts: vol/seg [C][D][H][W] of one case; out [C][pd][ph][pw] float32 (the caller offsets `out` to batch
 * row j).  (lbz, lby, lbx) = bbox_lbs of get_bbox (base_data_loader.py:56-139; may be negative / overhang).
 * flip_mask bit 0/1/2 mirrors D/H/W of the padded patch.  Bit-exact against oracle/feed_oracle.py. */
int mvd_feed_crop_pad_f32(const float *vol, float *out, int C, int D, int H, int W, int pd, int ph, int pw, int lbz,
                          int lby, int lbx, int flip_mask, float pad, void *stream);
/* int16 segmentation -> float32 target; after padding with `pad`, voxels equal to replace_from become replace_to
 * when `replace` != 0 */
int mvd_feed_crop_pad_seg_i16(const int16_t *seg, float *out, int C, int D, int H, int W, int pd, int ph, int pw, int lbz,
                              int lby, int lbx, int flip_mask, int pad, int replace, int replace_from, int replace_to,
                              void *stream);
/* order-0 resize [BC][D][H][W] -> [BC][d][h][w]: source index floor((2o+1)n/(2m)) per axis (pixel-centre aligned
 * nearest neighbour == skimage.transform.resize(order=0) == scipy.ndimage.zoom(order=0, grid_mode=True)) */
int mvd_feed_downsample_seg(const float *in, float *out, long BC, int D, int H, int W, int d, int h, int w, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVDSEG_HIP_H */
