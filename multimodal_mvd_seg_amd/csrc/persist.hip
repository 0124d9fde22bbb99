// H0 persistence (birth / death pairing) of a scalar field on a voxel grid -- SURVEY K12 / a-11.
//
// Replaces, for vertices + edges (maxdim 0), what the reference computes on the CPU through its vendored TopologyLayer
// C++ (nnunetv2/training/topologylayer/functional/persistence/): complex.cpp:136-146 (lower-star extension: a cell's
// value is the max of its vertex values, its critical vertex the arg-max), complex.cpp:182-196 (filtration order =
// sort by (value, dimension)), hom.cpp:51-69 (column reduction: restricted to vertices and edges it IS union-find with
// the elder rule), hom.cpp:155-185 (one bar per vertex, essential bars die at +inf).  The reference builds the boundary
// matrix in a std::map and reduces columns one by one; here
//   device: one key per grid edge = (order-preserving bits of max(g[u], g[v]))<<32 | edge index, all edges of the
//           D x H x W grid at once (k_h0_edge_keys), then ONE 64-bit radix sort (hipCUB / rocPRIM) = the filtration
//           order of the 1-cells; vertices need no sort: "older" is a comparison of (value, index);
//   host:   a single union-find sweep over the sorted edges with the elder rule (mvd_h0_pair_host) -- the inherently
//           sequential part, O(E alpha(V)) integer work on an 8 MB parent array.
// Integer / comparison work only: results are bit-exact against oracle/cc_oracle.c (same tie-breaking: vertices by
// (value, linear index), edges by (value, enumeration index)) and, as multisets of (birth, death), against the
// reference's own C++ (tests/golden/persistence_grid.json).
#include "common.h"

#include <hipcub/hipcub.hpp>

#include <math.h>

namespace mvd {

struct H0Offsets {
    int n;
    int off[13][3];
};

static int h0_fill_offsets(int conn, H0Offsets *o) {
    // half-neighbourhood (each undirected edge once), same enumeration as oracle/cc_oracle.c::neighbour_offsets
    o->n = 0;
    if (conn == 6) {
        const int t[3][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
        for (int i = 0; i < 3; i++) memcpy(o->off[o->n++], t[i], sizeof(int) * 3);
    } else if (conn == 14) {
        const int t[7][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
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

// monotone map float -> uint32 (a < b  <=>  key(a) < key(b)); -0.0 is canonicalised to +0.0 by the caller
__host__ __device__ inline uint32_t f2ord(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float v;
    memcpy(&v, &u, 4);
    return v;
}
__host__ __device__ inline float h0_sign(float v, int sublevel) {
    float g = sublevel ? v : -v;  // super-level persistence = sub-level persistence of -f (nn/levelset.py:150-160)
    return g == 0.f ? 0.f : g;
}

// one thread per (vertex, half-neighbour): key = ord(max(g[v], g[u])) << 32 | (v * n_off + k); edges that leave the
// grid get the all-ones key and sort to the end (their number is known on the host analytically)
__global__ void k_h0_edge_keys(const float *__restrict__ f, uint64_t *__restrict__ keys, int D, int H, int W,
                               const H0Offsets o, int sublevel) {
    const long N = (long)D * H * W, total = N * o.n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long v = e / o.n;
        const int k = (int)(e - v * o.n);
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long)W * H));
        const int zz = z + o.off[k][0], yy = y + o.off[k][1], xx = x + o.off[k][2];
        uint64_t key = ~0ull;
        if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const long u = ((long)zz * H + yy) * W + xx;
            const float a = h0_sign(f[v], sublevel), b = h0_sign(f[u], sublevel);
            key = ((uint64_t)f2ord(a < b ? b : a) << 32) | (uint64_t)e;
        }
        keys[e] = key;
    }
}

static long h0_valid_edges(int D, int H, int W, const H0Offsets &o) {
    long n = 0;
    for (int k = 0; k < o.n; k++) {
        const long a = D - abs(o.off[k][0]), b = H - abs(o.off[k][1]), c = W - abs(o.off[k][2]);
        if (a > 0 && b > 0 && c > 0) n += a * b * c;
    }
    return n;
}

static inline int32_t uf_find(int32_t *p, int32_t x) {
    while (p[x] != x) {
        p[x] = p[p[x]];
        x = p[x];
    }
    return x;
}

}  // namespace mvd

using namespace mvd;

extern "C" {

long mvd_h0_num_edges(int D, int H, int W, int conn) {
    H0Offsets o;
    if (D <= 0 || H <= 0 || W <= 0 || h0_fill_offsets(conn, &o)) return -1;
    return h0_valid_edges(D, H, W, o);
}

size_t mvd_h0_workspace_bytes(int D, int H, int W, int conn) {
    H0Offsets o;
    if (D <= 0 || H <= 0 || W <= 0 || h0_fill_offsets(conn, &o)) return 0;
    const long total = (long)D * H * W * o.n;
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, temp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)total, 0, 64);
    return (size_t)total * sizeof(uint64_t) + temp + 512;
}

int mvd_h0_sorted_edges(const float *f, uint64_t *keys_sorted, int D, int H, int W, int conn, int sublevel, void *ws,
                        size_t ws_bytes, void *stream) {
    H0Offsets o;
    MVD_REQUIRE(f && keys_sorted && ws, "h0_sorted_edges: null pointer");
    MVD_REQUIRE(D > 0 && H > 0 && W > 0 && h0_fill_offsets(conn, &o) == 0, "h0_sorted_edges: conn must be 6, 14 or 26");
    const long total = (long)D * H * W * o.n;
    MVD_REQUIRE(total < (1L << 31), "h0_sorted_edges: grid too large (edge index must fit 31 bits)");
    MVD_REQUIRE(ws_bytes >= mvd_h0_workspace_bytes(D, H, W, conn), "h0_sorted_edges: workspace too small");
    hipStream_t s = as_stream(stream);
    uint64_t *keys = reinterpret_cast<uint64_t *>(ws);
    void *temp = reinterpret_cast<char *>(ws) + ((size_t)total * sizeof(uint64_t) + 255) / 256 * 256;
    size_t temp_bytes = ws_bytes - (((size_t)total * sizeof(uint64_t) + 255) / 256 * 256);
    long bx = cdiv(total, 256);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(k_h0_edge_keys, dim3(bx), dim3(256), 0, s, f, keys, D, H, W, o, sublevel);
    if (check_launch("h0_edge_keys")) return 1;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, keys_sorted, (int)total, 0, 64, s);
    if (e != hipSuccess) {
        set_error("h0_sorted_edges: radix sort: %s", hipGetErrorString(e));
        return 1;
    }
    return check_launch("h0_sort");
}

/* Host half (plain C++, no device access): elder-rule union-find over the sorted edge keys.
 * f_host: the field (as given, not sign-flipped); keys_host: the first n_edges = mvd_h0_num_edges() sorted keys.
 * death[v] = value at which the bar born at vertex v dies (+inf / -inf for the essential bar of each component under
 * sub-/super-level filtration, i.e. death = -(+inf) after undoing the sign flip), death_vertex[v] = critical (arg-max)
 * vertex of the killing edge or -1.  birth of the bar of vertex v is f_host[v].  Returns the number of essential bars. */
long mvd_h0_pair_host(const float *f_host, const uint64_t *keys_host, long n_edges, int D, int H, int W, int conn,
                      int sublevel, float *death, int64_t *death_vertex) {
    H0Offsets o;
    if (!f_host || !keys_host || !death || !death_vertex || D <= 0 || H <= 0 || W <= 0 || h0_fill_offsets(conn, &o)) {
        set_error("h0_pair_host: bad arguments");
        return -1;
    }
    const long N = (long)D * H * W;
    if (N >= (1L << 31) || n_edges != h0_valid_edges(D, H, W, o)) {
        set_error("h0_pair_host: n_edges does not match the grid (%ld)", n_edges);
        return -1;
    }
    int32_t *p = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    if (!p) {
        set_error("h0_pair_host: out of host memory");
        return -1;
    }
    const float inf = sublevel ? INFINITY : -INFINITY;
    for (long i = 0; i < N; i++) {
        p[i] = (int32_t)i;
        death[i] = inf;
        death_vertex[i] = -1;
    }
    long doff[13];
    for (int k = 0; k < o.n; k++) doff[k] = ((long)o.off[k][0] * H + o.off[k][1]) * W + o.off[k][2];
    long alive = N;
    for (long e = 0; e < n_edges && alive > 1; e++) {
        const uint64_t key = keys_host[e];
        const uint32_t idx = (uint32_t)key;
        const long v = idx / (uint32_t)o.n;
        const long u = v + doff[idx - (uint32_t)v * (uint32_t)o.n];
        const int32_t a = uf_find(p, (int32_t)v), b = uf_find(p, (int32_t)u);
        if (a == b) continue;  // the edge closes a 1-cycle: not an H0 event
        const float ga = h0_sign(f_host[a], sublevel), gb = h0_sign(f_host[b], sublevel);
        const bool a_younger = ga > gb || (ga == gb && a > b);  // filtration rank of a vertex = (value, index)
        const int32_t young = a_younger ? a : b, old = a_younger ? b : a;
        const float gv = h0_sign(f_host[v], sublevel), gu = h0_sign(f_host[u], sublevel);
        const float val = gv < gu ? gu : gv;                     // == ord2f(key >> 32)
        death[young] = sublevel ? val : -val;
        death_vertex[young] = gv < gu ? u : v;                   // complex.cpp:141-145: the arg-max vertex of the edge
        p[young] = old;
        alive--;
    }
    long ness = 0;
    for (long i = 0; i < N; i++) ness += isinf(death[i]) ? 1 : 0;
    free(p);
    return ness;
}
}
