// Error plumbing, version, layout helpers and the shared fixed-order second-stage reducer.
#include <stdarg.h>

#include "common.h"
#include "conv_geom.h"

namespace mvd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void k_reduce_partials(const double *__restrict__ partials, float *__restrict__ out, int nblk, int stride,
                                  int offset) {
    // one wave per output value, lanes stride over blocks, fixed shuffle tree -> deterministic
    int j = blockIdx.x;
    // four loads in flight per lane (a lane's trips were one dependent round trip each: up to 128 of them for the
    // seg-head weight gradient), combined in a fixed order
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int b = threadIdx.x;
    for (; b + 192 < nblk; b += 256) {
        const double v0 = partials[(size_t)b * stride + offset + j], v1 = partials[(size_t)(b + 64) * stride + offset + j];
        const double v2 = partials[(size_t)(b + 128) * stride + offset + j], v3 = partials[(size_t)(b + 192) * stride + offset + j];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; b < nblk; b += 64) s0 += partials[(size_t)b * stride + offset + j];
    double s = (s0 + s1) + (s2 + s3);
    s = wave_sum(s);
    if (threadIdx.x == 0) out[j] = (float)s;
}

int reduce_partials(const double *partials, float *out, int nblk, int nv, hipStream_t s, int stride, int offset) {
    if (stride <= 0) stride = nv;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nv), dim3(64), 0, s, partials, out, nblk, stride, offset);
    return check_launch("reduce_partials");
}

// ---------------------------------------------------------------- layout transposes  [N][C][V] <-> [N][V][C]
// 32 voxels x C tile through LDS so both sides are coalesced.
__global__ void k_nchw_to_ndhwc(const float *__restrict__ src, float *__restrict__ dst, int C, long V) {
    __shared__ float tile[32][33];
    long v0 = (long)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    int n = blockIdx.z;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r;
        long v = v0 + tx;
        tile[r][tx] = (c < C && v < V) ? src[((size_t)n * C + c) * V + v] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        long v = v0 + r;
        int c = c0 + tx;
        if (c < C && v < V) dst[((size_t)n * V + v) * C + c] = tile[tx][r];
    }
}
// C == 4 (the network input): a thread gathers its voxel from the four planes (coalesced per plane) and writes one float4
__global__ void k_nchw_to_ndhwc_c4(const float *__restrict__ src, float *__restrict__ dst, long V) {
    const int n = blockIdx.y;
    const float *s = src + (size_t)n * 4 * V;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x)
        *reinterpret_cast<float4 *>(dst + ((size_t)n * V + v) * 4) = make_float4(s[v], s[V + v], s[2 * V + v], s[3 * V + v]);
}
// fp32 [N][C][V] (planar) or [N][V][C] (ndhwc != 0), C <= 8 -> bf16 [N][V][Cpad] with zero channels C .. Cpad-1: four lanes per
// voxel, lane part p writes the 16-byte piece p (a wave instruction stores 1 KB contiguous); only part 0 reads
template <int CPAD>
__global__ void k_pad_channels_bf16(const float *__restrict__ src, unsigned short *__restrict__ dst, int C, long V, int ndhwc) {
    const int n = blockIdx.y;
    constexpr int PARTS = CPAD / 8;
    const long total = V * PARTS;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = i / PARTS;
        const int part = (int)(i - v * PARTS);
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (part == 0) {
            float f[8];
#pragma unroll
            for (int c = 0; c < 8; c++)
                f[c] = c < C ? (ndhwc ? src[((size_t)n * V + v) * C + c] : src[((size_t)n * C + c) * V + v]) : 0.f;
            q.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
            q.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
            q.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
            q.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
        }
        *reinterpret_cast<uint4 *>(dst + ((size_t)n * V + v) * CPAD + part * 8) = q;
    }
}
__global__ void k_ndhwc_to_nchw(const float *__restrict__ src, float *__restrict__ dst, int C, long V) {
    __shared__ float tile[32][33];
    long v0 = (long)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    int n = blockIdx.z;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        long v = v0 + r;
        int c = c0 + tx;
        tile[r][tx] = (c < C && v < V) ? src[((size_t)n * V + v) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r;
        long v = v0 + tx;
        if (c < C && v < V) dst[((size_t)n * C + c) * V + v] = tile[tx][r];
    }
}

__global__ void k_axpy(float *__restrict__ y, const float *__restrict__ x, float a, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = y[i] + a * x[i];
}

// flat fp32 <-> bf16 conversion, 4 elements per thread (n % 4 tail handled scalar)
template <bool TO_BF>
__global__ void k_cast(const void *__restrict__ src, void *__restrict__ dst, long n) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 q = ld4<!TO_BF>(src, (size_t)i * 4);
        st4<TO_BF>(dst, (size_t)i * 4, q.x, q.y, q.z, q.w);
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) st1<TO_BF>(dst, (size_t)i, ld1<!TO_BF>(src, (size_t)i));
}

// torch [K][C][T] (or transposed-conv [C][K][T]) -> wf / wb in the packed layout of conv_geom.h (widx)
__global__ void k_pack_weight(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wb, int K,
                              int C, int T, int transposed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)K * C * T;
    if (i >= total) return;
    // i indexes wf: [t][c][k]
    int k = i % K;
    int c = (i / K) % C;
    int t = i / ((long)K * C);
    float v = transposed ? w[((size_t)c * K + k) * T + t] : w[((size_t)k * C + c) * T + t];
    if (wf) wf[widx(wl_ck(C), T, C, K, t, c, k)] = v;  // reduce over C, produce K
    if (wb) wb[widx(wl_ck(K), T, K, C, t, k, c)] = v;  // reduce over K, produce C
}

}  // namespace mvd

using namespace mvd;

extern "C" {

int mvd_version(void) { return MVD_VERSION; }
const char *mvd_last_error(void) { return g_err; }
int mvd_has_mfma(void) { return 1; }

int mvd_pack_weight(const float *w, float *wf, float *wb, int K, int C, int T, int transposed, void *stream) {
    MVD_REQUIRE(w && (wf || wb) && K > 0 && C > 0 && T > 0 && T <= MVD_MAX_TAPS, "pack_weight: bad arguments");
    long total = (long)K * C * T;
    hipLaunchKernelGGL(k_pack_weight, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, wf, wb, K, C, T,
                       transposed);
    return check_launch("pack_weight");
}

int mvd_nchw_to_ndhwc(const float *src, float *dst, int N, int C, long V, void *stream) {
    MVD_REQUIRE(src && dst && N > 0 && C > 0 && V > 0 && N <= 65535, "nchw_to_ndhwc: bad arguments");
    if (C == 4 && (((uintptr_t)dst) & 15) == 0) {
        long bx = cdiv(V, 256);
        if (bx > 8192) bx = 8192;
        hipLaunchKernelGGL(k_nchw_to_ndhwc_c4, dim3((unsigned)bx, N), dim3(256), 0, as_stream(stream), src, dst, V);
        return check_launch("nchw_to_ndhwc");
    }
    dim3 g(cdiv(V, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(k_nchw_to_ndhwc, g, dim3(256), 0, as_stream(stream), src, dst, C, V);
    return check_launch("nchw_to_ndhwc");
}
int mvd_pad_channels_bf16(const float *src, uint16_t *dst, int N, int C, int Cpad, long V, int src_ndhwc, void *stream) {
    MVD_REQUIRE(src && dst && N > 0 && N <= 65535 && V > 0, "pad_channels_bf16: bad arguments");
    MVD_REQUIRE(C >= 1 && C <= 8 && Cpad == 32, "pad_channels_bf16: C must be 1..8 and Cpad 32");
    MVD_REQUIRE((((uintptr_t)dst) & 15) == 0, "pad_channels_bf16: dst must be 16-byte aligned");
    long bx = cdiv(V * 4, 256);
    if (bx > 16384) bx = 16384;
    hipLaunchKernelGGL(k_pad_channels_bf16<32>, dim3((unsigned)bx, N), dim3(256), 0, as_stream(stream), src,
                       reinterpret_cast<unsigned short *>(dst), C, V, src_ndhwc);
    return check_launch("pad_channels_bf16");
}
int mvd_ndhwc_to_nchw(const float *src, float *dst, int N, int C, long V, void *stream) {
    MVD_REQUIRE(src && dst && N > 0 && C > 0 && V > 0 && N <= 65535, "ndhwc_to_nchw: bad arguments");
    dim3 g(cdiv(V, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(k_ndhwc_to_nchw, g, dim3(256), 0, as_stream(stream), src, dst, C, V);
    return check_launch("ndhwc_to_nchw");
}
int mvd_axpy(float *y, const float *x, float a, long n, void *stream) {
    MVD_REQUIRE(y && x && n > 0, "axpy: bad arguments");
    long blocks = cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_axpy, dim3(blocks), dim3(256), 0, as_stream(stream), y, x, a, n);
    return check_launch("axpy");
}
int mvd_cast_f32_to_bf16(const float *src, uint16_t *dst, long n, void *stream) {
    MVD_REQUIRE(src && dst && n > 0, "cast_f32_to_bf16: bad arguments");
    long blocks = cdiv(cdiv(n, 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_cast<true>, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, n);
    return check_launch("cast_f32_to_bf16");
}
int mvd_cast_bf16_to_f32(const uint16_t *src, float *dst, long n, void *stream) {
    MVD_REQUIRE(src && dst && n > 0, "cast_bf16_to_f32: bad arguments");
    long blocks = cdiv(cdiv(n, 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_cast<false>, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, n);
    return check_launch("cast_bf16_to_f32");
}
}
