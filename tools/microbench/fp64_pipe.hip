// Do v_mfma_f64_16x16x4_f64 and the fp64 vector instructions of ANOTHER wave on the same SIMD run side by side, or do they share
// one pipe?  One workgroup of eight waves per CU (waves w and w + 4 land on the same SIMD): waves 0..3 issue back-to-back
// independent matrix-core instructions, waves 4..7 back-to-back independent v_fma_f64 -- each role alone, then both together.
// Side by side: T(both) ~ max(T(mfma), T(fma)); one pipe: T(both) ~ T(mfma) + T(fma).  Also: two MFMA waves on a SIMD, two
// FMA waves on a SIMD (each role's own scaling), and fp32 FMAs beside the fp64 matrix instructions.
//   hipcc --offload-arch=gfx950 -O3 fp64_pipe.hip -o fp64_pipe && ./fp64_pipe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

// role bits per wave group (waves 0..3 = group 0, waves 4..7 = group 1): 0 idle, 1 mfma f64, 2 fma f64, 3 fma f32
template <int R0, int R1>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int iters) {
    const int wv = threadIdx.x >> 6, grp = wv >> 2, lane = threadIdx.x & 63;
    const int role = grp == 0 ? R0 : R1;
    const long long t0 = __builtin_readcyclecounter();
    double res = 0.0;
    if (role == 1) {
        v4d a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = (v4d){0.0, 0.0, 0.0, 0.0};
        const double x = 1.0 + lane * 1e-9, y = 1.0 - lane * 1e-9;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) res += a[q][0] + a[q][1] + a[q][2] + a[q][3];
    } else if (role == 2) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = lane + q;
        const double x = 1.0 + lane * 1e-9, y = 1e-9;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = fma(a[q], x, y);   // 128 v_fma_f64 per iteration = 512 issue cycles = 8 MFMAs' 64
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) res += a[q];
    } else if (role == 3) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = lane + q;
        const float x = 1.0f + lane * 1e-7f, y = 1e-7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = fmaf(a[q], x, y);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) res += a[q];
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[(size_t)blockIdx.x * 8 + wv] = t1 - t0;
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = res;
}

template <int R0, int R1>
static void run(const char *name, double *out, long long *cyc, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R0, R1>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R0, R1>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms   cycles per iteration: group 0 (waves 0-3) %7.1f   group 1 (waves 4-7) %7.1f\n", name, ms,
           R0 ? (double)h[0] / iters : 0.0, R1 ? (double)h[4] / iters : 0.0);
}

int main() {
    const int grid = 256, iters = 20000;
    double *out; long long *cyc;
    hipMalloc(&out, (size_t)grid * 512 * sizeof(double));
    hipMalloc(&cyc, (size_t)grid * 8 * sizeof(long long));
    printf("per iteration and wave: 8 x v_mfma_f64_16x16x4_f64, or 128 x v_fma_f64 / v_fma_f32\n");
    run<1, 0>("mfma f64 alone (one wave per SIMD)", out, cyc, grid, iters);
    run<2, 0>("fma f64 alone", out, cyc, grid, iters);
    run<3, 0>("fma f32 alone", out, cyc, grid, iters);
    run<1, 1>("mfma f64 + mfma f64 on each SIMD", out, cyc, grid, iters);
    run<2, 2>("fma f64 + fma f64 on each SIMD", out, cyc, grid, iters);
    run<1, 2>("mfma f64 + fma f64 on each SIMD", out, cyc, grid, iters);
    run<1, 3>("mfma f64 + fma f32 on each SIMD", out, cyc, grid, iters);
    return 0;
}
