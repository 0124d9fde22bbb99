// Internal helpers shared by the kernel translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mvdseg_hip.h"

namespace mvd {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

#define MVD_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mvd::set_error(__VA_ARGS__);  \
            return 2;                     \
        }                                 \
    } while (0)

// Every entry point converts its stream argument first; that is also where a stale (sticky, already reported or
// benign) error of an earlier HIP call in this thread -- e.g. one of PyTorch's own probes -- is cleared, so that
// check_launch() only ever reports the status of OUR launch.
inline hipStream_t as_stream(void *s) {
    (void)hipGetLastError();
    return reinterpret_cast<hipStream_t>(s);
}

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// "dynamic LDS limit raised for this kernel" flags are per DEVICE (hipFuncSetAttribute acts on the current device's copy
// of the code object; ADVICE r2: a process-wide flag left a second GPU of the same process without the attribute).
struct PerDeviceFlag {
    unsigned char done[64] = {};
    bool &operator()() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        return reinterpret_cast<bool &>(done[dev]);
    }
};

// A value loaded from memory, made "arrived" HERE: the empty volatile asm is a use the compiler must wait for, once and
// unconditionally.  Without it a bias loaded at the top of an epilogue has its first use inside the predicated store
// blocks, every such block gets an s_waitcnt vmcnt(0), and that also waits for the previous block's STORE: the stores
// of a tile run as serialised write round trips (k_fwd_wino2: 23 of a tile's 85 kilocycles, tools/stamps_wino.py).
__device__ __forceinline__ float settled(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float4 settled(float4 v) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return v;
}

// bf16 storage helpers (uint16_t bit patterns).  f2bf: v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserved.
__device__ inline unsigned short f2bf(float f) {
    __bf16 h = (__bf16)f;
    return *reinterpret_cast<unsigned short *>(&h);
}
// 4 consecutive channels at element index e of a float (BF = false) or bf16 (BF = true) tensor, as fp32
template <bool BF>
__device__ inline float4 ld4(const void *p, size_t e) {
    if (BF) {
        const uint2 q = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(p) + e);
        return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                           __uint_as_float(q.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + e);
}
template <bool BF>
__device__ inline void st4(void *p, size_t e, float a, float b, float c, float d) {
    if (BF) {
        uint2 q;
        q.x = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
        q.y = (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16);
        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(p) + e) = q;
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + e) = make_float4(a, b, c, d);
    }
}

template <bool BF>
__device__ inline float ld1(const void *p, size_t e) {
    if (BF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(p)[e] << 16);
    return reinterpret_cast<const float *>(p)[e];
}
template <bool BF>
__device__ inline void st1(void *p, size_t e, float v) {
    if (BF) reinterpret_cast<unsigned short *>(p)[e] = f2bf(v);
    else reinterpret_cast<float *>(p)[e] = v;
}

// 64-lane wavefront reductions (CDNA wave = 64)
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum of NV doubles per thread; result valid in thread 0.  blockDim.x multiple of 64, <= 1024.
template <int NV>
__device__ inline void block_sum(double (&v)[NV], double *smem /* >= NV*16 doubles */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) smem[i * 16 + wid] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            double s = 0;
            for (int w = 0; w < nw; w++) s += smem[i * 16 + w];
            v[i] = s;
        }
    }
}

// generic fixed-order second stage: out[j] = sum_b partial[b*stride + offset + j]  (double -> float), j < nv
int reduce_partials(const double *partials, float *out, int nblk, int nv, hipStream_t s, int stride = 0, int offset = 0);

}  // namespace mvd
