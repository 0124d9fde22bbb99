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

struct FwdGeom {
    int N;
    int Di, Hi, Wi;
    int Do, Ho, Wo;
    int Dy, Hy, Wy;
    int C1, C2, K1, K2;
    int ntaps;
    int sa[3], so[3], oo[3];
    int8_t off[27][3];
    int8_t wt[27];
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
int fwd_mfma(const FwdGeom &g, const float *a1, const float *a2, const float *w, const float *bias, float *y1, float *y2,
             hipStream_t s);
size_t wgrad_scalar_ws(const WgradGeom &g);
int wgrad_scalar(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws,
                 size_t ws_bytes, hipStream_t s);
size_t wgrad_mfma_ws(const WgradGeom &g);
int wgrad_mfma(const WgradGeom &g, const float *a1, const float *a2, const float *b, float *dw, void *ws, size_t ws_bytes,
               hipStream_t s);

}  // namespace mvd
