// What does the memory system deliver for the K > 64 solver's access pattern -- whole rows of betaT[V][K] (K doubles each) picked by
// word id, gathered tile by tile (16 rows per tile, lane l loads components l and l + 64 of every row) by persistent single-wave
// workgroups?  Throughput against the rows a CU keeps in flight (workgroups per CU x tiles in flight per wave), for beta = 40 MB
// (config 4: V = 50 000, K = 100: far beyond the 4 MB L2 of an XCD, inside the 256 MB Infinity Cache) and 4 MB (config 2).
//   hipcc --offload-arch=gfx950 -O3 row_gather.hip -o row_gather && ./row_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

template <int DEPTH>   // tiles of 16 rows in flight per wave
__global__ __launch_bounds__(64) void gather(const double *bT, const int *ids, int K, int tiles_per_wg, double *out) {
    const int lane = threadIdx.x;
    const int *my = ids + (size_t)blockIdx.x * tiles_per_wg * 16;
    double pre[DEPTH][32];
    double acc = 0.0;
    auto fetch = [&](int t, int slot) __attribute__((always_inline)) {
        const int id = lane < 16 ? my[t * 16 + lane] : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double *row = bT + (size_t)__builtin_amdgcn_readlane(id, j) * K;
            pre[slot][2 * j] = lane < K ? row[lane] : 0.0;
            pre[slot][2 * j + 1] = lane + 64 < K ? row[lane + 64] : 0.0;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, d);
    for (int t = 0; t < tiles_per_wg; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += pre[d][j];
            if (t + d + DEPTH < tiles_per_wg) fetch(t + d + DEPTH, d);
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;
}

template <int DEPTH>
static void run(const double *bT, const int *ids, int K, int wg_per_cu, int tiles_per_wg, double *out, const char *what) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<DEPTH>), dim3(grid), dim3(64), 0, 0, bT, ids, K, tiles_per_wg, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((gather<DEPTH>), dim3(grid), dim3(64), 0, 0, bT, ids, K, tiles_per_wg, out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * tiles_per_wg * 16 * K * 8;
    printf("%-10s %2d workgroups per CU x %d tile(s) in flight = %4d rows per CU: %7.3f ms  %6.2f TB/s  (%.0f cycles per tile and wave at 2.1 GHz)\n",
           what, wg_per_cu, DEPTH, wg_per_cu * DEPTH * 16, ms, bytes / ms * 1e-9, ms * 1e-3 * 2.1e9 / (tiles_per_wg));
}

int main() {
    struct Cfg { int V, K; const char *name; } cfgs[2] = {{50000, 100, "40 MB"}, {10000, 50, "4 MB"}};
    for (const Cfg &c : cfgs) {
        std::vector<double> h((size_t)c.V * c.K, 1.0);
        double *bT; (void)hipMalloc(&bT, h.size() * 8);
        (void)hipMemcpy(bT, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        const int max_wg = 256 * 16, tiles = 64;
        // word ids: Zipf-like (the synthetic corpus' shape does not matter much: uniform ids give the same picture within 10 %)
        std::vector<int> ids((size_t)max_wg * tiles * 16);
        srand(1);
        for (auto &v : ids) { const double u = (rand() + 1.0) / (RAND_MAX + 2.0); v = (int)(c.V * u * u) % c.V; }
        int *dids; (void)hipMalloc(&dids, ids.size() * 4);
        (void)hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
        double *out; (void)hipMalloc(&out, (size_t)max_wg * 64 * 8);
        for (int w : {4, 8, 16}) run<1>(bT, dids, c.K, w, tiles, out, c.name);
        for (int w : {4, 8}) run<2>(bT, dids, c.K, w, tiles, out, c.name);
        run<4>(bT, dids, c.K, 4, tiles, out, c.name);
        (void)hipFree(bT); (void)hipFree(dids); (void)hipFree(out);
    }
    return 0;
}
