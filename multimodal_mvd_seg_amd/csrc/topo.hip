// Soft-skeleton primitives (erode / dilate / skeleton update) on planar volumes and the integer
// connected-component labelling (K12).  Compiled with -ffp-contract=off: `delta - skel*delta` must round the product
// (torch does mul then sub) for bit-exact forward parity.
#include "common.h"

namespace mvd {

// ------------------------------------------------------------------------------------------------ erode
// code bits: [1:0] argmin pos along D (0,1,2 = d-1,d,d+1), [3:2] along H, [5:4] along W,
//            [7:6] p1 vs p2 (0: p1<p2, 1: equal, 2: p1>p2), [9:8] min(p1,p2) vs p3 (same encoding)
__device__ inline float axis_min(const float *p, long stride, int pos, int len, int &arg) {
    // first minimum in scan order (-F.max_pool3d(-x) keeps the first maximum, soft_skeleton.py:12-14)
    float best = INFINITY;
    arg = 1;
    bool have = false;
#pragma unroll
    for (int o = -1; o <= 1; o++) {
        int q = pos + o;
        if (q < 0 || q >= len) continue;
        float v = p[(long)o * stride];
        if (!have || v < best) {
            best = v;
            arg = o + 1;
            have = true;
        }
    }
    return best;
}

__global__ void k_erode_fwd(const float *__restrict__ x, float *__restrict__ y, uint16_t *__restrict__ code, long total,
                            int D, int H, int W) {
    const long HW = (long)H * W, DHW = (long)D * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i % DHW;
        int d = (int)(r / HW), h = (int)((r % HW) / W), w = (int)(r % W);
        int a1, a2, a3;
        float p1 = axis_min(x + i, HW, d, D, a1);
        float p2 = axis_min(x + i, W, h, H, a2);
        float p3 = axis_min(x + i, 1, w, W, a3);
        float m12 = fminf(p1, p2);
        y[i] = fminf(m12, p3);
        if (code) {
            int c12 = p1 < p2 ? 0 : (p1 == p2 ? 1 : 2);
            int c3 = m12 < p3 ? 0 : (m12 == p3 ? 1 : 2);
            code[i] = (uint16_t)(a1 | (a2 << 2) | (a3 << 4) | (c12 << 6) | (c3 << 8));
        }
    }
}

__device__ inline void erode_shares(uint16_t c, float s[3]) {
    int c12 = (c >> 6) & 3, c3 = (c >> 8) & 3;
    float sm = c3 == 0 ? 1.f : (c3 == 1 ? 0.5f : 0.f);  // share of min(p1,p2)
    s[2] = 1.f - sm;
    float s1 = c12 == 0 ? 1.f : (c12 == 1 ? 0.5f : 0.f);
    s[0] = s1 * sm;
    s[1] = (1.f - s1) * sm;
}

__global__ void k_erode_bwd(const uint16_t *__restrict__ code, const float *__restrict__ dy, float *__restrict__ dx,
                            long total, int D, int H, int W) {
    const long HW = (long)H * W, DHW = (long)D * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i % DHW;
        int pos[3] = {(int)(r / HW), (int)((r % HW) / W), (int)(r % W)};
        const int len[3] = {D, H, W};
        const long st[3] = {HW, (long)W, 1};
        float g = 0.f;
        // fixed accumulation order: centre, then axis D (-,+), H (-,+), W (-,+)
        {
            uint16_t c = code[i];
            float s[3];
            erode_shares(c, s);
            float gy = dy[i];
#pragma unroll
            for (int a = 0; a < 3; a++)
                if (((c >> (2 * a)) & 3) == 1) g += s[a] * gy;
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int o = -1; o <= 1; o += 2) {
                const int q = pos[a] + o;  // output voxel at offset o along axis a; this voxel sits at -o in its window
                const bool ok = q >= 0 && q < len[a];
                const long j = ok ? i + (long)o * st[a] : i;  // (unconditional loads, selected afterwards: see k_dilate_bwd)
                const uint16_t c = code[j];
                const float gy = dy[j];
                float s[3];
                erode_shares(c, s);
                g = (ok && (int)((c >> (2 * a)) & 3) == (1 - o)) ? g + s[a] * gy : g;
            }
        }
        dx[i] = g;
    }
}

// ------------------------------------------------------------------------------------------------ dilate (3x3x3 max)
__global__ void k_dilate_fwd(const float *__restrict__ x, float *__restrict__ y, uint8_t *__restrict__ code, long total,
                             int D, int H, int W) {
    const long HW = (long)H * W, DHW = (long)D * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i % DHW;
        int d = (int)(r / HW), h = (int)((r % HW) / W), w = (int)(r % W);
        float best = -INFINITY;
        int arg = 13;
        bool have = false;
        for (int a = -1; a <= 1; a++) {
            if (d + a < 0 || d + a >= D) continue;
            for (int b = -1; b <= 1; b++) {
                if (h + b < 0 || h + b >= H) continue;
#pragma unroll
                for (int c = -1; c <= 1; c++) {
                    if (w + c < 0 || w + c >= W) continue;
                    float v = x[i + a * HW + b * W + c];
                    if (!have || v > best) {  // first maximum in scan order (max_pool3d CPU kernel)
                        best = v;
                        arg = (a + 1) * 9 + (b + 1) * 3 + (c + 1);
                        have = true;
                    }
                }
            }
        }
        y[i] = best;
        if (code) code[i] = (uint8_t)arg;
    }
}

__global__ void k_dilate_bwd(const uint8_t *__restrict__ code, const float *__restrict__ dy, float *__restrict__ dx,
                             long total, int D, int H, int W) {
    const long HW = (long)H * W, DHW = (long)D * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i % DHW;
        int d = (int)(r / HW), h = (int)((r % HW) / W), w = (int)(r % W);
        // All 27 (code, dy) pairs are loaded unconditionally (a neighbour outside the volume reads this voxel and is
        // ignored) and accumulated by selects in the same order: with `continue` / `if` around the loads the compiler
        // waited for each one before issuing the next -- 54 dependent round trips per voxel, 107 us for a 17-MB volume.
        float g = 0.f;
#pragma unroll
        for (int a = -1; a <= 1; a++)
#pragma unroll
            for (int b = -1; b <= 1; b++)
#pragma unroll
                for (int c = -1; c <= 1; c++) {
                    const bool ok = d + a >= 0 && d + a < D && h + b >= 0 && h + b < H && w + c >= 0 && w + c < W;
                    const long j = ok ? i + a * HW + b * W + c : i;  // output voxel; this voxel is at (-a,-b,-c) in its window
                    const int want = (1 - a) * 9 + (1 - b) * 3 + (1 - c);
                    const int cj = (int)code[j];
                    const float gy = dy[j];
                    g = (ok && cj == want) ? g + gy : g;
                }
        dx[i] = g;
    }
}

// ------------------------------------------------------------------------------------------------ skeleton update
__global__ void k_skel_update_fwd(const float *__restrict__ img, const float *__restrict__ opened,
                                  const float *__restrict__ skel_in, float *__restrict__ skel_out, long n, int init) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float delta = fmaxf(img[i] - opened[i], 0.f);
        if (init) {
            skel_out[i] = delta;
        } else {
            float s = skel_in[i];
            float prod = s * delta;
            skel_out[i] = s + fmaxf(delta - prod, 0.f);
        }
    }
}

__global__ void k_skel_update_bwd(const float *__restrict__ img, const float *__restrict__ opened,
                                  const float *__restrict__ skel_in, const float *__restrict__ go,
                                  float *__restrict__ d_img, float *__restrict__ d_opened, float *__restrict__ d_skel,
                                  long n, int init) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float diff = img[i] - opened[i];
        float delta = fmaxf(diff, 0.f);
        float g = go[i];
        float gdelta;
        if (init) {
            gdelta = g;
        } else {
            float s = skel_in[i];
            float prod = s * delta;
            float r = delta - prod;
            float gr = r > 0.f ? g : 0.f;     // relu'(r)
            gdelta = gr - gr * s;            // d r/d delta = 1 - s  (autograd: gr + (-gr)*s)
            d_skel[i] = g + (-gr) * delta;   // d/dskel of skel + relu(delta - skel*delta)
        }
        float gi = diff > 0.f ? gdelta : 0.f;
        d_img[i] = gi;
        d_opened[i] = -gi;
    }
}

// ------------------------------------------------------------------------------------------------ fused skeleton iteration
// One iteration of soft_skel (soft_skeleton.py:33-36) in ONE launch:
//     e1 = erode(img);  e2 = erode(e1);  o = dilate(e2);  skel' = skel + relu(relu(e1 - o) - skel * relu(e1 - o))
// (INIT: the step in front of the loop, :30-31: e1 := img, skel' = relu(img - o)).  The primitive-per-launch chain read and
// wrote the 16.8 MB volume 11 times per iteration through four instruction-bound kernels (27 bounds-checked global
// loads and three 64-bit divisions per voxel: k_dilate_fwd ran at 7 % of HBM); here a workgroup stages an 8x8x32 tile
// with a halo of 3 in LDS once and walks the three stencils in LDS (halo 2 -> 1 -> 0).  Same arithmetic, same scan order
// of the arg-min / arg-max, same routing codes: forward stays bit-exact and the existing backward kernels consume the
// codes unchanged.  Volume borders: a neighbour outside the volume is skipped exactly as in the per-primitive kernels
// (global coordinates, not tile coordinates, decide).
constexpr int SK_TZ = 8, SK_TY = 8, SK_TX = 32;
constexpr int SK_AZ = SK_TZ + 6, SK_AY = SK_TY + 6, SK_AX = SK_TX + 6;  // img, halo 3
constexpr int SK_BZ = SK_TZ + 4, SK_BY = SK_TY + 4, SK_BX = SK_TX + 4;  // e1, halo 2
constexpr int SK_CZ = SK_TZ + 2, SK_CY = SK_TY + 2, SK_CX = SK_TX + 2;  // e2, halo 1

// min over a 3-window whose out-of-volume entries hold +inf (so they are never selected): first minimum in scan order
__device__ inline float axis_min3(float lo, float mid, float hi, int &arg) {
    float best = lo;
    arg = 0;
    if (mid < best) { best = mid; arg = 1; }
    if (hi < best) { best = hi; arg = 2; }
    return best;
}
__device__ inline float erode_padded(const float *p, int sz, int sy, uint16_t &code) {
    int a1, a2, a3;
    const float c = p[0];
    const float p1 = axis_min3(p[-sz], c, p[sz], a1);
    const float p2 = axis_min3(p[-sy], c, p[sy], a2);
    const float p3 = axis_min3(p[-1], c, p[1], a3);
    const float m12 = fminf(p1, p2);
    const int c12 = p1 < p2 ? 0 : (p1 == p2 ? 1 : 2);
    const int c3 = m12 < p3 ? 0 : (m12 == p3 ? 1 : 2);
    code = (uint16_t)(a1 | (a2 << 2) | (a3 << 4) | (c12 << 6) | (c3 << 8));
    return fminf(m12, p3);
}

// Out-of-volume cells of the LDS images hold +inf in the erode inputs (A, B) and -inf in the dilate input (C): a
// neighbour outside the volume can then never be the (first) minimum / maximum, which is exactly what skipping it does
// for finite data -- and the stencil loops carry no bounds logic (the first version, with per-neighbour coordinate
// tests, was slower than the four separate launches it replaced).  Every stage pads its own output again (one
// coordinate test per cell): +inf for e1, -inf for e2.
template <bool INIT>
__global__ __launch_bounds__(256) void k_skel_iter_fwd(const float *__restrict__ img, const float *__restrict__ skel_in,
                                                       float *__restrict__ e1_out, float *__restrict__ opened_out,
                                                       float *__restrict__ skel_out, uint16_t *__restrict__ c_e1,
                                                       uint16_t *__restrict__ c_e2, uint8_t *__restrict__ c_o, int D, int H,
                                                       int W, int ntz, int nty, int ntx) {
    __shared__ float A[SK_AZ * SK_AY * SK_AX];
    __shared__ float B[INIT ? 1 : SK_BZ * SK_BY * SK_BX];
    __shared__ float Cc[SK_CZ * SK_CY * SK_CX];
    const int tid = threadIdx.x;
    unsigned b = blockIdx.x;
    const int tx = (int)(b % (unsigned)ntx); b /= (unsigned)ntx;
    const int ty = (int)(b % (unsigned)nty); b /= (unsigned)nty;
    const int tz = (int)(b % (unsigned)ntz);
    const long nc = (long)(b / (unsigned)ntz);
    const int z0 = tz * SK_TZ, y0 = ty * SK_TY, x0 = tx * SK_TX;
    const long HW = (long)H * W;
    const long vbase = nc * D * HW;
    const float *src = img + vbase;
    // stage A: the image with a halo of 3
    // (unconditional loads from clamped coordinates, four in flight per thread, padded by a select afterwards: a predicate
    // around the load made every trip of this loop a memory round trip of its own)
    {
        constexpr int NA_ = SK_AZ * SK_AY * SK_AX;
        for (int base = 0; base < NA_; base += 4 * 256) {
            float v[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = base + u * 256 + tid;
                const int ii = idx < NA_ ? idx : 0;
                const int lx = ii % SK_AX, ly = (ii / SK_AX) % SK_AY, lz = ii / (SK_AX * SK_AY);
                const int gz = z0 + lz - 3, gy = y0 + ly - 3, gx = x0 + lx - 3;
                in[u] = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int cz = min(max(gz, 0), D - 1), cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
                v[u] = src[(long)cz * HW + (long)cy * W + cx];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = base + u * 256 + tid;
                if (idx < NA_) A[idx] = in[u] ? v[u] : INFINITY;
            }
        }
    }
    __syncthreads();
    // stage B: e1 = erode(img) on the tile + halo 2
    if (!INIT) {
        for (int idx = tid; idx < SK_BZ * SK_BY * SK_BX; idx += 256) {
            const int lx = idx % SK_BX, ly = (idx / SK_BX) % SK_BY, lz = idx / (SK_BX * SK_BY);
            uint16_t code;
            const float v = erode_padded(A + ((lz + 1) * SK_AY + (ly + 1)) * SK_AX + (lx + 1), SK_AY * SK_AX, SK_AX, code);
            const int gz = z0 + lz - 2, gy = y0 + ly - 2, gx = x0 + lx - 2;
            const bool inside = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
            B[idx] = inside ? v : INFINITY;  // (a cell just outside the volume has finite neighbours: pad it again)
            if (inside && lz >= 2 && lz < 2 + SK_TZ && ly >= 2 && ly < 2 + SK_TY && lx >= 2 && lx < 2 + SK_TX) {
                const long o = vbase + (long)gz * HW + (long)gy * W + gx;
                e1_out[o] = v;
                if (c_e1) c_e1[o] = code;
            }
        }
        __syncthreads();
    }
    // stage C: e2 = erode(e1) on the tile + halo 1 (-inf outside the volume: it feeds the max)
    for (int idx = tid; idx < SK_CZ * SK_CY * SK_CX; idx += 256) {
        const int lx = idx % SK_CX, ly = (idx / SK_CX) % SK_CY, lz = idx / (SK_CX * SK_CY);
        const int gz = z0 + lz - 1, gy = y0 + ly - 1, gx = x0 + lx - 1;
        uint16_t code;
        float v;
        if (INIT)
            v = erode_padded(A + ((lz + 2) * SK_AY + (ly + 2)) * SK_AX + (lx + 2), SK_AY * SK_AX, SK_AX, code);
        else
            v = erode_padded(B + ((lz + 1) * SK_BY + (ly + 1)) * SK_BX + (lx + 1), SK_BY * SK_BX, SK_BX, code);
        const bool inside = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        if (c_e2 && inside && lz >= 1 && lz <= SK_TZ && ly >= 1 && ly <= SK_TY && lx >= 1 && lx <= SK_TX)
            c_e2[vbase + (long)gz * HW + (long)gy * W + gx] = code;
        Cc[idx] = inside ? v : -INFINITY;
    }
    __syncthreads();
    // stage D + E: o = dilate(e2), skeleton update, on the tile
#pragma unroll
    for (int u = 0; u < SK_TZ * SK_TY * SK_TX / 256; u++) {
        const int j = u * 256 + tid;
        const int lx = j % SK_TX, ly = (j / SK_TX) % SK_TY, lz = j / (SK_TX * SK_TY);
        const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
        if (gz >= D || gy >= H || gx >= W) continue;
        const float *q = Cc + ((lz + 1) * SK_CY + (ly + 1)) * SK_CX + (lx + 1);
        float best = -INFINITY;
        int arg = 13;  // (a volume of one voxel: every neighbour is -inf and the centre wins below)
#pragma unroll
        for (int a = -1; a <= 1; a++)
#pragma unroll
            for (int bb = -1; bb <= 1; bb++)
#pragma unroll
                for (int c = -1; c <= 1; c++) {
                    const float v = q[(a * SK_CY + bb) * SK_CX + c];
                    if (v > best) {  // first maximum in scan order (max_pool3d)
                        best = v;
                        arg = (a + 1) * 9 + (bb + 1) * 3 + (c + 1);
                    }
                }
        const long o = vbase + (long)gz * HW + (long)gy * W + gx;
        opened_out[o] = best;
        if (c_o) c_o[o] = (uint8_t)arg;
        const float e1 = INIT ? A[((lz + 3) * SK_AY + (ly + 3)) * SK_AX + (lx + 3)]
                              : B[((lz + 2) * SK_BY + (ly + 2)) * SK_BX + (lx + 2)];
        const float delta = fmaxf(e1 - best, 0.f);
        if (INIT) {
            skel_out[o] = delta;
        } else {
            const float sk = skel_in[o];
            const float prod = sk * delta;
            skel_out[o] = sk + fmaxf(delta - prod, 0.f);
        }
    }
}

__global__ void k_dot_sum(const float *__restrict__ a, const float *__restrict__ b, double *__restrict__ partial,
                          long n) {
    __shared__ double red[2 * 16];
    double acc[2] = {0.0, 0.0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x = a[i];
        acc[0] += (double)x * (double)b[i];
        acc[1] += (double)x;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * 2 + 0] = acc[0];
        partial[(size_t)blockIdx.x * 2 + 1] = acc[1];
    }
}

__global__ void k_cldice_combine(const float *__restrict__ s, float *__restrict__ out, float smooth) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float d1 = s[1] + smooth, d3 = s[3] + smooth;
    float a = (s[0] + smooth) / d1, b = (s[2] + smooth) / d3;
    float ab = a + b;
    out[0] = 1.0f - 2.0f * (a * b) / ab;
    float dfa = 2.0f * b * b / (ab * ab), dfb = 2.0f * a * a / (ab * ab);
    out[1] = -dfa / d1;
    out[2] = dfa * a / d1;
    out[3] = -dfb / d3;
    out[4] = dfb * b / d3;
}

// ------------------------------------------------------------------------------------------------ connected components
__global__ void k_threshold(const float *__restrict__ f, uint8_t *__restrict__ m, long n, float thr, int ge) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        m[i] = ge ? (f[i] >= thr) : (f[i] > thr);
}

__global__ void k_cc_init(const uint8_t *__restrict__ mask, int32_t *__restrict__ parent, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        parent[i] = mask[i] ? (int32_t)i : -1;
}

__device__ inline int32_t cc_find(int32_t *parent, int32_t x) {
    // parent[] only ever decreases (atomicMin), so the walk terminates at a root
    int32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__device__ inline void cc_union(int32_t *parent, int32_t a, int32_t b) {
    // link the larger root under the smaller one; retry when another wave re-rooted `a` meanwhile
    for (int guard = 0; guard < (1 << 30); guard++) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) {
            int32_t t = a;
            a = b;
            b = t;
        }
        int32_t old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;  // a was no longer a root: continue from what it pointed to
    }
}

struct CcOffsets {
    int n;
    int off[13][3];
};

__global__ void k_cc_merge(const uint8_t *__restrict__ mask, int32_t *__restrict__ parent, int D, int H, int W,
                           CcOffsets offs) {
    const long HW = (long)H * W, total = (long)D * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        int d = (int)(i / HW), h = (int)((i % HW) / W), w = (int)(i % W);
        for (int k = 0; k < offs.n; k++) {
            int dd = d + offs.off[k][0], hh = h + offs.off[k][1], ww = w + offs.off[k][2];
            if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            long j = ((long)dd * H + hh) * W + ww;
            if (mask[j]) cc_union(parent, (int32_t)i, (int32_t)j);
        }
    }
}

__global__ void k_cc_flatten(int32_t *__restrict__ parent, int32_t *__restrict__ count, long n) {
    // every root is final after k_cc_merge completed (kernel boundary); labels = 1 + root index
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t p = parent[i];
        if (p < 0) continue;
        if (p == (int32_t)i) atomicAdd(count, 1);
    }
}
// in-place path compression: non-roots are rewritten to their root index, which is still a valid ancestor for any
// concurrent reader; roots (parent[r] == r) are never rewritten.
__global__ void k_cc_compress(int32_t *parent, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p < 0) continue;
        int32_t x = (int32_t)i;
        while (p != x) {
            x = p;
            p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&parent[i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// after the kernel boundary every entry holds its root (or -1): label = root + 1, background 0
__global__ void k_cc_finish(int32_t *labels, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        labels[i] = labels[i] + 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Connected-component post-processing (remove_connected_components.py:22-34): label-set mask, component
// sizes, the `keep` largest components, relabel of everything else in the mask to the background label.
struct LabelSet {
    int n;
    int32_t v[16];
};

__global__ void k_seg_label_mask(const int32_t *__restrict__ seg, uint8_t *__restrict__ mask, long n, LabelSet ls) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t s = seg[i];
        bool in = false;
        for (int k = 0; k < ls.n; k++) in |= (s == ls.v[k]);
        mask[i] = in ? 1 : 0;
    }
}

// sizes[root] += 1 for every voxel of the component whose canonical label is root + 1.  Each thread walks RUN
// consecutive voxels and issues one integer atomic per run of equal labels (components are spatially coherent, so
// a large component does not serialise n atomics on one address).
template <int RUN>
__global__ void k_cc_sizes(const int32_t *__restrict__ labels, int32_t *__restrict__ sizes, long n) {
    long nchunks = (n + RUN - 1) / RUN;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long)gridDim.x * blockDim.x) {
        long i0 = c * RUN, i1 = i0 + RUN < n ? i0 + RUN : n;
        int32_t cur = 0, cnt = 0;
        for (long i = i0; i < i1; i++) {
            int32_t l = labels[i];
            if (l != cur) {
                if (cur > 0) atomicAdd(&sizes[cur - 1], cnt);
                cur = l;
                cnt = 0;
            }
            cnt++;
        }
        if (cur > 0) atomicAdd(&sizes[cur - 1], cnt);
    }
}

// key = size << 32 | ~root: larger size wins, ties go to the smaller canonical label.  size 0 == no component.
__device__ __forceinline__ void top2_push(unsigned long long &a1, unsigned long long &a2, unsigned long long k) {
    if (k > a1) {
        a2 = a1;
        a1 = k;
    } else if (k > a2) {
        a2 = k;
    }
}
__device__ __forceinline__ void top2_block(unsigned long long &a1, unsigned long long &a2, unsigned long long *sm) {
    // sm: 2 * blockDim.x keys
    const int t = threadIdx.x;
    sm[2 * t] = a1;
    sm[2 * t + 1] = a2;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) {
            unsigned long long b1 = sm[2 * (t + s)], b2 = sm[2 * (t + s) + 1];
            top2_push(a1, a2, b1);
            top2_push(a1, a2, b2);
            sm[2 * t] = a1;
            sm[2 * t + 1] = a2;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_cc_top2_part(const int32_t *__restrict__ sizes, long n,
                                                      unsigned long long *__restrict__ cand) {
    __shared__ unsigned long long sm[512];
    unsigned long long a1 = 0, a2 = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t sz = sizes[i];
        if (sz > 0) top2_push(a1, a2, ((unsigned long long)(uint32_t)sz << 32) | (uint32_t)(~(uint32_t)i));
    }
    top2_block(a1, a2, sm);
    if (threadIdx.x == 0) {
        cand[2 * blockIdx.x] = a1;
        cand[2 * blockIdx.x + 1] = a2;
    }
}
__global__ void __launch_bounds__(256) k_cc_top2_final(const unsigned long long *__restrict__ cand, int ncand, int keep,
                                                       int32_t *__restrict__ kept) {
    __shared__ unsigned long long sm[512];
    unsigned long long a1 = 0, a2 = 0;
    for (int i = threadIdx.x; i < ncand; i += blockDim.x) top2_push(a1, a2, cand[i]);
    top2_block(a1, a2, sm);
    if (threadIdx.x == 0) {
        // kept[0..1] = canonical labels (0 = none); kept[2..3] = their sizes
        kept[0] = (a1 >> 32) ? (int32_t)(~(uint32_t)a1) + 1 : 0;
        kept[1] = (keep > 1 && (a2 >> 32)) ? (int32_t)(~(uint32_t)a2) + 1 : 0;
        kept[2] = (int32_t)(a1 >> 32);
        kept[3] = keep > 1 ? (int32_t)(a2 >> 32) : 0;
    }
}

__global__ void k_seg_remove_components(const int32_t *__restrict__ seg, const int32_t *__restrict__ cc,
                                        const int32_t *__restrict__ kept, int32_t *__restrict__ out, long n,
                                        int32_t background) {
    const int32_t k0 = kept[0], k1 = kept[1];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t l = cc[i];
        out[i] = (l > 0 && l != k0 && l != k1) ? background : seg[i];
    }
}

}  // namespace mvd

using namespace mvd;

static inline long ew_grid(long n) {
    long b = cdiv(n, 256);
    return b > 4096 ? 4096 : (b < 1 ? 1 : b);
}

static int fill_offsets(int conn, CcOffsets *o) {
    o->n = 0;
    if (conn == 6) {
        int t[3][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
        for (int i = 0; i < 3; i++) memcpy(o->off[o->n++], t[i], sizeof(int) * 3);
    } else if (conn == 14) {
        int t[7][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
        for (int i = 0; i < 7; i++) memcpy(o->off[o->n++], t[i], sizeof(int) * 3);
    } else if (conn == 26) {
        for (int dz = 0; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (dz == 0 && (dy < 0 || (dy == 0 && dx <= 0))) continue;
                    o->off[o->n][0] = dz;
                    o->off[o->n][1] = dy;
                    o->off[o->n][2] = dx;
                    o->n++;
                }
    } else {
        return 1;
    }
    return 0;
}

extern "C" {

#define VOL_CHECK(name)                                                                                        \
    MVD_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && (long)NC * D * H * W < (1L << 40), name ": bad volume shape")

int mvd_soft_erode_fwd(const float *x, float *y, uint16_t *code, int NC, int D, int H, int W, void *stream) {
    MVD_REQUIRE(x && y, "soft_erode_fwd: null pointer");
    VOL_CHECK("soft_erode_fwd");
    long total = (long)NC * D * H * W;
    hipLaunchKernelGGL(k_erode_fwd, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), x, y, code, total, D, H, W);
    return check_launch("soft_erode_fwd");
}
int mvd_soft_erode_bwd(const uint16_t *code, const float *dy, float *dx, int NC, int D, int H, int W, void *stream) {
    MVD_REQUIRE(code && dy && dx, "soft_erode_bwd: null pointer");
    VOL_CHECK("soft_erode_bwd");
    long total = (long)NC * D * H * W;
    hipLaunchKernelGGL(k_erode_bwd, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), code, dy, dx, total, D, H, W);
    return check_launch("soft_erode_bwd");
}
int mvd_soft_dilate_fwd(const float *x, float *y, uint8_t *code, int NC, int D, int H, int W, void *stream) {
    MVD_REQUIRE(x && y, "soft_dilate_fwd: null pointer");
    VOL_CHECK("soft_dilate_fwd");
    long total = (long)NC * D * H * W;
    hipLaunchKernelGGL(k_dilate_fwd, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), x, y, code, total, D, H, W);
    return check_launch("soft_dilate_fwd");
}
int mvd_soft_dilate_bwd(const uint8_t *code, const float *dy, float *dx, int NC, int D, int H, int W, void *stream) {
    MVD_REQUIRE(code && dy && dx, "soft_dilate_bwd: null pointer");
    VOL_CHECK("soft_dilate_bwd");
    long total = (long)NC * D * H * W;
    hipLaunchKernelGGL(k_dilate_bwd, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), code, dy, dx, total, D, H,
                       W);
    return check_launch("soft_dilate_bwd");
}

int mvd_skel_update_fwd(const float *img, const float *opened, const float *skel_in, float *skel_out, long n, int init,
                        void *stream) {
    MVD_REQUIRE(img && opened && skel_out && (init || skel_in) && n > 0, "skel_update_fwd: bad arguments");
    hipLaunchKernelGGL(k_skel_update_fwd, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), img, opened, skel_in,
                       skel_out, n, init);
    return check_launch("skel_update_fwd");
}
int mvd_skel_update_bwd(const float *img, const float *opened, const float *skel_in, const float *d_skel_out,
                        float *d_img, float *d_opened, float *d_skel_in, long n, int init, void *stream) {
    MVD_REQUIRE(img && opened && d_skel_out && d_img && d_opened && (init || (skel_in && d_skel_in)) && n > 0,
                "skel_update_bwd: bad arguments");
    hipLaunchKernelGGL(k_skel_update_bwd, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), img, opened, skel_in,
                       d_skel_out, d_img, d_opened, d_skel_in, n, init);
    return check_launch("skel_update_bwd");
}

int mvd_skel_iter_fwd(const float *img, const float *skel_in, float *e1, float *opened, float *skel_out, uint16_t *c_e1,
                      uint16_t *c_e2, uint8_t *c_o, int NC, int D, int H, int W, int init, void *stream) {
    MVD_REQUIRE(img && opened && skel_out, "skel_iter_fwd: null pointer");
    MVD_REQUIRE(init || (skel_in && e1), "skel_iter_fwd: an iteration needs skel_in and an e1 output");
    VOL_CHECK("skel_iter_fwd");
    const int ntz = (D + SK_TZ - 1) / SK_TZ, nty = (H + SK_TY - 1) / SK_TY, ntx = (W + SK_TX - 1) / SK_TX;
    const long nb = (long)NC * ntz * nty * ntx;
    MVD_REQUIRE(nb < (1L << 31), "skel_iter_fwd: volume too large");
    hipStream_t s = as_stream(stream);
    if (init)
        hipLaunchKernelGGL(k_skel_iter_fwd<true>, dim3((unsigned)nb), dim3(256), 0, s, img, skel_in, e1, opened, skel_out, c_e1,
                           c_e2, c_o, D, H, W, ntz, nty, ntx);
    else
        hipLaunchKernelGGL(k_skel_iter_fwd<false>, dim3((unsigned)nb), dim3(256), 0, s, img, skel_in, e1, opened, skel_out,
                           c_e1, c_e2, c_o, D, H, W, ntz, nty, ntx);
    return check_launch("skel_iter_fwd");
}

size_t mvd_dot_sum_workspace_bytes(long n) {
    long b = cdiv(n, 256);
    if (b > 1024) b = 1024;
    return (size_t)b * 2 * sizeof(double) + 256;
}
int mvd_dot_sum(const float *a, const float *b, float *out, long n, void *ws, size_t ws_bytes, void *stream) {
    MVD_REQUIRE(a && b && out && ws && n > 0, "dot_sum: bad arguments");
    MVD_REQUIRE(ws_bytes >= mvd_dot_sum_workspace_bytes(n), "dot_sum: workspace too small");
    long bx = cdiv(n, 256);
    if (bx > 1024) bx = 1024;
    double *partial = reinterpret_cast<double *>(ws);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_dot_sum, dim3(bx), dim3(256), 0, s, a, b, partial, n);
    if (check_launch("dot_sum")) return 1;
    return reduce_partials(partial, out, (int)bx, 2, s);
}

int mvd_cldice_combine(const float *sums, float *out, float smooth, void *stream) {
    MVD_REQUIRE(sums && out, "cldice_combine: null pointer");
    hipLaunchKernelGGL(k_cldice_combine, dim3(1), dim3(64), 0, as_stream(stream), sums, out, smooth);
    return check_launch("cldice_combine");
}

int mvd_threshold_mask(const float *f, uint8_t *mask, long n, float thr, int ge, void *stream) {
    MVD_REQUIRE(f && mask && n > 0, "threshold_mask: bad arguments");
    hipLaunchKernelGGL(k_threshold, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), f, mask, n, thr, ge);
    return check_launch("threshold_mask");
}

int mvd_cc_label(const uint8_t *mask, int32_t *labels, int32_t *count, int D, int H, int W, int conn, void *stream) {
    MVD_REQUIRE(mask && labels && count, "cc_label: null pointer");
    MVD_REQUIRE(D > 0 && H > 0 && W > 0 && (long)D * H * W < 2147483647L, "cc_label: bad grid (int32 labels)");
    CcOffsets offs;
    MVD_REQUIRE(fill_offsets(conn, &offs) == 0, "cc_label: conn must be 6, 14 or 26");
    long n = (long)D * H * W;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(count, 0, sizeof(int32_t), s) != hipSuccess) {
        set_error("cc_label: memset failed");
        return 1;
    }
    // `labels` doubles as the parent forest until the last two passes
    hipLaunchKernelGGL(k_cc_init, dim3(ew_grid(n)), dim3(256), 0, s, mask, labels, n);
    hipLaunchKernelGGL(k_cc_merge, dim3(ew_grid(n)), dim3(256), 0, s, mask, labels, D, H, W, offs);
    hipLaunchKernelGGL(k_cc_flatten, dim3(ew_grid(n)), dim3(256), 0, s, labels, count, n);
    if (check_launch("cc_label merge")) return 1;
    hipLaunchKernelGGL(k_cc_compress, dim3(ew_grid(n)), dim3(256), 0, s, labels, n);
    hipLaunchKernelGGL(k_cc_finish, dim3(ew_grid(n)), dim3(256), 0, s, labels, n);
    return check_launch("cc_label relabel");
}

int mvd_seg_label_mask(const int32_t *seg, uint8_t *mask, long n, const int32_t *label_set, int nlabels, void *stream) {
    MVD_REQUIRE(seg && mask && label_set, "seg_label_mask: null pointer");
    MVD_REQUIRE(n > 0 && nlabels >= 1 && nlabels <= 16, "seg_label_mask: 1..16 labels");
    LabelSet ls;
    ls.n = nlabels;
    for (int i = 0; i < 16; i++) ls.v[i] = i < nlabels ? label_set[i] : 0;  // host array, passed by value
    hipLaunchKernelGGL(k_seg_label_mask, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), seg, mask, n, ls);
    return check_launch("seg_label_mask");
}

size_t mvd_cc_keep_workspace_bytes(long n) { return (size_t)n * sizeof(int32_t) + 2 * 1024 * sizeof(uint64_t); }

int mvd_cc_keep_largest(const int32_t *cc_labels, long n, int keep, int32_t *kept, void *workspace, void *stream) {
    MVD_REQUIRE(cc_labels && kept && workspace, "cc_keep_largest: null pointer");
    MVD_REQUIRE(n > 0 && n < 2147483647L && (keep == 1 || keep == 2), "cc_keep_largest: keep must be 1 or 2");
    hipStream_t s = as_stream(stream);
    unsigned long long *cand = (unsigned long long *)workspace;  // 2 * 1024 keys, then n sizes
    int32_t *sizes = (int32_t *)(cand + 2 * 1024);
    if (hipMemsetAsync(sizes, 0, (size_t)n * sizeof(int32_t), s) != hipSuccess) {
        set_error("cc_keep_largest: memset failed");
        return 1;
    }
    constexpr int RUN = 16;
    hipLaunchKernelGGL(k_cc_sizes<RUN>, dim3(ew_grid(cdiv(n, RUN))), dim3(256), 0, s, cc_labels, sizes, n);
    long nb = cdiv(n, 256);
    int blocks = (int)(nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(k_cc_top2_part, dim3(blocks), dim3(256), 0, s, sizes, n, cand);
    hipLaunchKernelGGL(k_cc_top2_final, dim3(1), dim3(256), 0, s, cand, 2 * blocks, keep, kept);
    return check_launch("cc_keep_largest");
}

int mvd_seg_remove_components(const int32_t *seg, const int32_t *cc_labels, const int32_t *kept, int32_t *out, long n,
                              int background, void *stream) {
    MVD_REQUIRE(seg && cc_labels && kept && out, "seg_remove_components: null pointer");
    MVD_REQUIRE(n > 0, "seg_remove_components: empty");
    hipLaunchKernelGGL(k_seg_remove_components, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), seg, cc_labels, kept,
                       out, n, (int32_t)background);
    return check_launch("seg_remove_components");
}
}
