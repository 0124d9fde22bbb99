// Micro-probe: ds_read_b128 throughput for the lane -> address patterns of the Winograd forward kernel (pure LDS loop,
// 12 waves per CU).  build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/lds_pattern_probe tools/probes/lds_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 3) void k(float *out, const int *addr, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int j = threadIdx.x; j < 12800; j += 256) lds[j] = (float)j;
    __syncthreads();
    const unsigned a = (unsigned)addr[threadIdx.x & 63];
    v4f s0 = {}, s1 = {}, s2 = {}, s3 = {};
    for (int it = 0; it < iters; it++) {
        v4f r0, r1, r2, r3;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:144\n\tds_read_b128 %2, %4 offset:288\n\t"
                     "ds_read_b128 %3, %4 offset:432\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a) : "memory");
        s0 += r0; s1 += r1; s2 += r2; s3 += r3;
    }
    s0 += s1 + s2 + s3;
    out[blockIdx.x * 256 + threadIdx.x] = s0.x + s0.y + s0.z + s0.w;
}

static float run_table(const int *h, float *out, int *daddr);
static void run(const char *name, int (*f)(int), float *out, int *daddr) {
    int h[64];
    for (int l = 0; l < 64; l++) h[l] = f(l);
    hipMemcpy(daddr, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000, blocks = 768;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 51200, 0, out, daddr, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 51200, 0, out, daddr, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per CU: 12 waves x iters x 4 reads
    const double reads = 12.0 * iters * 4;
    printf("%-46s %.3f ms  %.2f cycles(2.4GHz)/ds_read_b128 per CU  (%.0f B/clk/CU)\n", name, ms, ms * 1e-3 * 2.4e9 / reads,
           1024.0 * reads / (ms * 1e-3 * 2.4e9));
}

static int contiguous(int l) { return l * 16; }
static int kernel_now(int l) {  // k_fwd_wino2: quad i = l & 31: plane i>>3, row (i>>2)&1, col i&3; slot stride 144 B; h -> +64 B
    int i = l & 31, h = l >> 5;
    return (((i >> 3) * 6 + 2 * ((i >> 2) & 1)) * 10 + 2 * (i & 3)) * 144 + h * 64;
}
static int remap_planes(int l) {  // col i&3, plane 2*bit2 + bit4, row bit3
    int i = l & 31, h = l >> 5;
    int plane = ((i >> 1) & 2) | ((i >> 4) & 1), row = (i >> 3) & 1;
    return ((plane * 6 + 2 * row) * 10 + 2 * (i & 3)) * 144 + h * 64;
}
static int evenodd_rows(int l) {  // x-permuted rows (even x first), col i&3, plane (i>>2)&3, row i>>4
    int i = l & 31, h = l >> 5;
    return (((i >> 2) & 3) * 60 + 2 * (i >> 4) * 10 + (i & 3)) * 144 + h * 64;
}
static int stride144(int l) { return (l & 31) * 144 + (l >> 5) * 64; }
static int stride288(int l) { return (l & 31) * 288 + (l >> 5) * 64; }
static int stride272(int l) { return (l & 31) * 272 + (l >> 5) * 64; }   // 68-float slots
static int stride160(int l) { return (l & 31) * 160 + (l >> 5) * 64; }   // 40-float slots

static float run_table(const int *h, float *out, int *daddr) {
    hipMemcpy(daddr, h, 64 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 4000, blocks = 768;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 51200, 0, out, daddr, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 51200, 0, out, daddr, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (12.0f * iters * 4);
}

// every assignment of the five quad-index bits to (col: 2 bits, plane: 2 bits, row: 1 bit) x layout x plane stride
static void sweep(float *out, int *daddr) {
    int perm[5] = {0, 1, 2, 3, 4};  // role of lane bit b: 0,1 = col bits, 2,3 = plane bits, 4 = row bit
    int best_n = 0;
    do {
        if (perm[0] > perm[1] || perm[2] > perm[3]) continue;  // col0<col1, plane0<plane1 are interchangeable pairs... keep all orders distinct anyway
    } while (0);
    int roles[5];
    for (int code = 0; code < 3125; code++) {
        int c = code, cnt[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < 5; b++) { roles[b] = c % 5; c /= 5; cnt[roles[b]]++; }
        bool ok = true;
        for (int r = 0; r < 5; r++) if (cnt[r] != 1) ok = false;
        if (!ok) continue;
        for (int layout = 0; layout < 2; layout++)
            for (int PS = 60; PS <= 62; PS++) {
                int h[64];
                for (int l = 0; l < 64; l++) {
                    int i = l & 31, hh = l >> 5, col = 0, plane = 0, row = 0;
                    for (int b = 0; b < 5; b++) {
                        int bit = (i >> b) & 1;
                        if (roles[b] == 0) col |= bit;
                        if (roles[b] == 1) col |= bit << 1;
                        if (roles[b] == 2) plane |= bit;
                        if (roles[b] == 3) plane |= bit << 1;
                        if (roles[b] == 4) row |= bit;
                    }
                    int slot = plane * PS + 2 * row * 10 + (layout ? col : 2 * col);
                    h[l] = slot * 144 + hh * 64;
                }
                float cyc = run_table(h, out, daddr);
                if (cyc < 6.5f) {
                    printf("cycles %.2f layout %s PS %d roles(bit0..4) %d%d%d%d%d\n", cyc, layout ? "evenodd" : "ident", PS, roles[0],
                           roles[1], roles[2], roles[3], roles[4]);
                    best_n++;
                }
            }
    }
    printf("sweep done: %d patterns under 6.5 cycles\n", best_n);
}

int main() {
    float *out; int *daddr;
    hipMalloc(&out, 1 << 22); hipMalloc(&daddr, 256);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run("contiguous 16 B per lane", contiguous, out, daddr);
    run("k_fwd_wino2 pattern (slot stride 144 B)", kernel_now, out, daddr);
    run("remapped quads (planes p, p+2 per 8 lanes)", remap_planes, out, daddr);
    run("even/odd x layout, 4 planes x 4 cols per 16", evenodd_rows, out, daddr);
    run("lane stride 144 B", stride144, out, daddr);
    run("lane stride 288 B", stride288, out, daddr);
    run("lane stride 272 B", stride272, out, daddr);
    run("lane stride 160 B", stride160, out, daddr);
    sweep(out, daddr);
    return 0;
}
