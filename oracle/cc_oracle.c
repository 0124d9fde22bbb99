/* TEST INFRASTRUCTURE -- plain-C CPU restatement of the integer connected-component / H0 pairing step.
 *
 * What it follows in the reference (nnUNet/nnunetv2/training/topologylayer/functional/persistence/):
 *   complex.cpp:136-146  lower-star extension: a cell's filtration value = max of its vertex values, its
 *                        critical vertex = the arg-max vertex
 *   complex.cpp:182-196  filtration order = sort by (value, dimension): vertices before edges on ties
 *   hom.cpp:51-69        column reduction of the sorted boundary matrix.  Restricted to vertices + edges
 *                        (MAXDIM = 0) the reduction is exactly union-find with the elder rule: an edge
 *                        whose endpoints lie in different components kills the YOUNGER component (the one
 *                        whose oldest vertex comes later in the filtration); its death value is the edge value
 *   hom.cpp:155-185      one bar per vertex (complex.cpp:129-131 numPairs(0) = #vertices), essential bars die
 *                        at +inf
 * Pinned against the reference's own C++ (compiled to oracle/_ref by oracle/build_ref.py) in
 * tests/test_oracle_vs_reference.py and through tests/golden/persistence_*.json.
 * std::sort leaves the order of equal (value, dim) cells unspecified, so diagrams are compared as multisets of
 * (birth, death); this file breaks ties by cell index (a valid filtration order).
 *
 * Grid graphs: vertices = voxels of a D x H x W grid (row-major, W fastest); edges by `conn`:
 *   6  -> axis neighbours; 14 -> Freudenthal triangulation edges (axis + (0,1,1),(1,0,1),(1,1,0),(1,1,1)
 *   positive diagonals -- for D == 1 this is levelset.py:64-93 init_freudenthal_2d); 26 -> full 3x3x3.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int neighbour_offsets(int conn, int off[13][3]) {
    /* half-neighbourhood (each undirected edge once), lexicographically positive offsets */
    int n = 0;
    if (conn == 6) {
        int o[3][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
        for (int i = 0; i < 3; i++) { memcpy(off[n++], o[i], sizeof(int) * 3); }
    } else if (conn == 14) {
        int o[7][3] = {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
        for (int i = 0; i < 7; i++) { memcpy(off[n++], o[i], sizeof(int) * 3); }
    } else if (conn == 26) {
        for (int dz = 0; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (dz == 0 && (dy < 0 || (dy == 0 && dx <= 0))) continue;
                    off[n][0] = dz; off[n][1] = dy; off[n][2] = dx; n++;
                }
    }
    return n;
}

static int64_t uf_find(int64_t *p, int64_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

/* Connected components of a binary mask.  labels[v] = 1 + smallest linear index of v's component (0 for
 * background) -- a canonical labelling, so it can be compared bit-for-bit.  Returns the component count. */
int64_t mvd_oracle_cc_label(const uint8_t *mask, int D, int H, int W, int conn, int32_t *labels) {
    int off[13][3];
    int no = neighbour_offsets(conn, off);
    int64_t N = (int64_t)D * H * W;
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * N);
    for (int64_t i = 0; i < N; i++) p[i] = i;
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int64_t v = ((int64_t)z * H + y) * W + x;
                if (!mask[v]) continue;
                for (int k = 0; k < no; k++) {
                    int zz = z + off[k][0], yy = y + off[k][1], xx = x + off[k][2];
                    if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    int64_t u = ((int64_t)zz * H + yy) * W + xx;
                    if (!mask[u]) continue;
                    int64_t a = uf_find(p, v), b = uf_find(p, u);
                    if (a == b) continue;
                    if (a < b) p[b] = a; else p[a] = b; /* root = smallest index */
                }
            }
    int64_t count = 0;
    for (int64_t i = 0; i < N; i++) {
        if (!mask[i]) { labels[i] = 0; continue; }
        int64_t r = uf_find(p, i);
        labels[i] = (int32_t)(r + 1);
        if (r == i) count++;
    }
    free(p);
    return count;
}

typedef struct { float val; int64_t a, b; int64_t idx; } edge_t;
typedef struct { float val; int64_t idx; } vert_t;

static int cmp_edge(const void *x, const void *y) {
    const edge_t *a = (const edge_t *)x, *b = (const edge_t *)y;
    if (a->val < b->val) return -1;
    if (a->val > b->val) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}
static int cmp_vert(const void *x, const void *y) {
    const vert_t *a = (const vert_t *)x, *b = (const vert_t *)y;
    if (a->val < b->val) return -1;
    if (a->val > b->val) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* H0 persistence of the SUB-level lower-star filtration of f on the grid graph.
 * Outputs, per vertex v (bar born at v): death[v] (+inf for the essential class of each component's oldest
 * vertex) and death_vertex[v] = critical (arg-max) vertex of the killing edge, or -1.
 * birth value of the bar is f[v].  Returns the number of essential (infinite) bars.
 * For super-level persistence call it on -f (nn/levelset.py does the same and negates the diagram back). */
int64_t mvd_oracle_h0_persistence(const float *f, int D, int H, int W, int conn, float *death,
                                  int64_t *death_vertex) {
    int off[13][3];
    int no = neighbour_offsets(conn, off);
    int64_t N = (int64_t)D * H * W;
    /* filtration rank of vertices: sort by (value, index) */
    vert_t *vs = (vert_t *)malloc(sizeof(vert_t) * N);
    int64_t *rank = (int64_t *)malloc(sizeof(int64_t) * N);
    for (int64_t i = 0; i < N; i++) { vs[i].val = f[i]; vs[i].idx = i; }
    qsort(vs, N, sizeof(vert_t), cmp_vert);
    for (int64_t i = 0; i < N; i++) rank[vs[i].idx] = i;
    edge_t *es = (edge_t *)malloc(sizeof(edge_t) * (size_t)N * no);
    int64_t ne = 0;
    for (int z = 0; z < D; z++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int64_t v = ((int64_t)z * H + y) * W + x;
                for (int k = 0; k < no; k++) {
                    int zz = z + off[k][0], yy = y + off[k][1], xx = x + off[k][2];
                    if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    int64_t u = ((int64_t)zz * H + yy) * W + xx;
                    es[ne].a = v; es[ne].b = u;
                    es[ne].val = f[v] < f[u] ? f[u] : f[v]; /* complex.cpp:141-142 */
                    es[ne].idx = ne;
                    ne++;
                }
            }
    qsort(es, ne, sizeof(edge_t), cmp_edge);
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * N); /* root = oldest (lowest-rank) vertex */
    for (int64_t i = 0; i < N; i++) { p[i] = i; death[i] = INFINITY; death_vertex[i] = -1; }
    for (int64_t e = 0; e < ne; e++) {
        int64_t a = uf_find(p, es[e].a), b = uf_find(p, es[e].b);
        if (a == b) continue;               /* edge creates a 1-cycle: not an H0 event */
        int64_t young = rank[a] > rank[b] ? a : b, old = rank[a] > rank[b] ? b : a;
        death[young] = es[e].val;           /* elder rule: hom.cpp:51-69 with pivot = younger vertex */
        death_vertex[young] = (f[es[e].a] < f[es[e].b]) ? es[e].b : es[e].a;
        p[young] = old;
    }
    int64_t ninf = 0;
    for (int64_t i = 0; i < N; i++) ninf += isinf(death[i]) ? 1 : 0;
    free(vs); free(rank); free(es); free(p);
    return ninf;
}
