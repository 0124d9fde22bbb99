// fp32 MFMA implicit-GEMM conv engines (placeholder until the kernels land: returns "unsupported").
#include "common.h"
#include "conv_geom.h"

namespace mvd {
int fwd_mfma(const FwdGeom &, const float *, const float *, const float *, const float *, float *, float *, hipStream_t) {
    return -1;
}
size_t wgrad_mfma_ws(const WgradGeom &) { return 0; }
int wgrad_mfma(const WgradGeom &, const float *, const float *, const float *, float *, void *, size_t, hipStream_t) {
    return -1;
}
}  // namespace mvd
