// On-device training feed (deterministic part): crop + constant pad of a resident case volume to the
// patch (data_loader_3d.py:31-46), mirroring (MirrorTransform, nnUNetTrainer.py:738-739), RemoveLabelTransform(-1, 0)
// (:745) and nearest-neighbour down-sampling of the target to the deep-supervision scales
// (deep_supervision_donwsampling.py:27-55).  Byte / index work only: bit-exact against oracle/feed_oracle.py.
// Planar [C][D][H][W]; every output element has one writer; HBM-bound gathers.
#include "common.h"

namespace mvd {

// out[c][z][y][x] = P[c][fz][fy][fx] with P = pad(vol[:, lb : lb + patch]) and f = index mirrored on the axes of
// flip_mask (bit 0: D, 1: H, 2: W).  After padding, values equal to rep_from become rep_to (do_rep).
template <typename TI>
__global__ void k_feed_crop_pad(const TI *__restrict__ vol, float *__restrict__ out, int C, int D, int H, int W, int pd,
                                int ph, int pw, int lbz, int lby, int lbx, int flip_mask, float pad, int do_rep,
                                float rep_from, float rep_to) {
    const long total = (long)C * pd * ph * pw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int x = (int)(r % pw); r /= pw;
        const int y = (int)(r % ph); r /= ph;
        const int z = (int)(r % pd);
        const int c = (int)(r / pd);
        const int sz = lbz + ((flip_mask & 1) ? pd - 1 - z : z);
        const int sy = lby + ((flip_mask & 2) ? ph - 1 - y : y);
        const int sx = lbx + ((flip_mask & 4) ? pw - 1 - x : x);
        float v = pad;
        if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W)
            v = (float)vol[(((size_t)c * D + sz) * H + sy) * W + sx];
        if (do_rep && v == rep_from) v = rep_to;
        out[idx] = v;
    }
}

// order-0 resize with pixel-centre alignment: source index = floor((2 o + 1) n / (2 m)) (== skimage.transform.resize
// order 0 == scipy.ndimage.zoom(order=0, mode='nearest', grid_mode=True))
__device__ __forceinline__ int nn_index(int o, int n, int m) {
    const long i = ((long)(2 * o + 1) * n) / (2L * m);
    return i < n ? (int)i : n - 1;
}

__global__ void k_feed_downsample_seg(const float *__restrict__ in, float *__restrict__ out, long BC, int D, int H, int W,
                                      int d, int h, int w) {
    const long total = BC * d * h * w;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h); r /= h;
        const int z = (int)(r % d);
        const long bc = r / d;
        out[idx] = in[((bc * D + nn_index(z, D, d)) * H + nn_index(y, H, h)) * W + nn_index(x, W, w)];
    }
}

}  // namespace mvd

using namespace mvd;

static inline unsigned feed_grid(long n) {
    long b = cdiv(n, 256);
    if (b > 16384) b = 16384;
    return (unsigned)(b < 1 ? 1 : b);
}

static bool feed_shape_ok(int C, int D, int H, int W, int pd, int ph, int pw, int lbz, int lby, int lbx) {
    if (C <= 0 || D <= 0 || H <= 0 || W <= 0 || pd <= 0 || ph <= 0 || pw <= 0) return false;
    // the box may hang over the volume on every side, but must stay within +-2^30 so the index math cannot wrap
    const int lim = 1 << 30;
    return lbz > -lim && lby > -lim && lbx > -lim && lbz < lim && lby < lim && lbx < lim && pd < lim && ph < lim && pw < lim;
}

extern "C" {

int mvd_feed_crop_pad_f32(const float *vol, float *out, int C, int D, int H, int W, int pd, int ph, int pw, int lbz,
                          int lby, int lbx, int flip_mask, float pad, void *stream) {
    MVD_REQUIRE(vol && out, "feed_crop_pad_f32: null pointer");
    MVD_REQUIRE(feed_shape_ok(C, D, H, W, pd, ph, pw, lbz, lby, lbx), "feed_crop_pad_f32: bad shape");
    MVD_REQUIRE(flip_mask >= 0 && flip_mask < 8, "feed_crop_pad_f32: flip_mask is a 3-bit axis mask");
    const long total = (long)C * pd * ph * pw;
    hipLaunchKernelGGL(k_feed_crop_pad<float>, dim3(feed_grid(total)), dim3(256), 0, as_stream(stream), vol, out, C, D, H,
                       W, pd, ph, pw, lbz, lby, lbx, flip_mask, pad, 0, 0.f, 0.f);
    return check_launch("feed_crop_pad_f32");
}

int mvd_feed_crop_pad_seg_i16(const int16_t *seg, float *out, int C, int D, int H, int W, int pd, int ph, int pw, int lbz,
                              int lby, int lbx, int flip_mask, int pad, int replace, int replace_from, int replace_to,
                              void *stream) {
    MVD_REQUIRE(seg && out, "feed_crop_pad_seg_i16: null pointer");
    MVD_REQUIRE(feed_shape_ok(C, D, H, W, pd, ph, pw, lbz, lby, lbx), "feed_crop_pad_seg_i16: bad shape");
    MVD_REQUIRE(flip_mask >= 0 && flip_mask < 8, "feed_crop_pad_seg_i16: flip_mask is a 3-bit axis mask");
    MVD_REQUIRE(pad >= -32768 && pad <= 32767, "feed_crop_pad_seg_i16: pad must fit int16");
    const long total = (long)C * pd * ph * pw;
    hipLaunchKernelGGL(k_feed_crop_pad<int16_t>, dim3(feed_grid(total)), dim3(256), 0, as_stream(stream), seg, out, C, D,
                       H, W, pd, ph, pw, lbz, lby, lbx, flip_mask, (float)pad, replace ? 1 : 0, (float)replace_from,
                       (float)replace_to);
    return check_launch("feed_crop_pad_seg_i16");
}

int mvd_feed_downsample_seg(const float *in, float *out, long BC, int D, int H, int W, int d, int h, int w, void *stream) {
    MVD_REQUIRE(in && out, "feed_downsample_seg: null pointer");
    MVD_REQUIRE(BC > 0 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0, "feed_downsample_seg: bad shape");
    MVD_REQUIRE(d <= D && h <= H && w <= W, "feed_downsample_seg: output must not be larger than the input");
    const long total = BC * d * h * w;
    hipLaunchKernelGGL(k_feed_downsample_seg, dim3(feed_grid(total)), dim3(256), 0, as_stream(stream), in, out, BC, D, H, W,
                       d, h, w);
    return check_launch("feed_downsample_seg");
}
}
