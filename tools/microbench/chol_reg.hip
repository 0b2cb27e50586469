// Register-resident Cholesky + triangular inverse of one n x n SPD matrix per wavefront, n a compile-time constant
// (49 = K - 1 at BASELINE's K = 50): lane i holds row i in registers, the code is one straight line (static register
// indices, no LDS, no branches but the pivot test), broadcasts by v_readlane with a constant lane.
//   Cholesky  right-looking: column j scaled by 1 / sqrt(pivot), then row[k] -= l_i * l_k for k > j
//   inverse   X = L^-1, lane c = column c: x_i = -(sum_{c<=l<i} L[i][l] x_l) / L[i][i], L[i][l] read from lane i
// Measures shader-clock cycles per matrix with W single-wave workgroups per CU, against the LDS form's
// 65 k (Cholesky) + 26 k (inverse) cycles of post_kernel<3,1,false> (profiles/r02 solver_prof).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chol_reg.hip -o chol_reg && ./chol_reg
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

constexpr int N = 49;

__device__ __forceinline__ double bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double &s, double &r) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    const double rr = fma(-g, g, d);
    g = fma(rr, h, g);
    e = fma(-h, g, 0.5);
    h = fma(h, e, h);
    s = g; r = h + h;
}

template <int MODE>   // 1: Cholesky only, 3: + inverse
__global__ __launch_bounds__(64, 2) void k(const double *A, double *Lout, double *Xout, long long *cyc, int nmat) {
    const int lane = threadIdx.x;
    long long tc = 0, ti = 0;
    for (int m = blockIdx.x; m < nmat; m += gridDim.x) {
        const double *a = A + (size_t)m * N * N;
        double row[N];
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = lane < N ? a[(size_t)lane * N + j] : (j == lane ? 1.0 : 0.0);
        const long long t0 = __builtin_readcyclecounter();
        bool ok = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double d = bcast(row[j], j);
            if (!(d > 0.0)) { ok = false; break; }
            double s, r;
            sqrt_and_rsqrt(d, s, r);
            const double l = (lane == j) ? s : row[j] * r;
            row[j] = l;
#pragma unroll
            for (int kk = j + 1; kk < N; ++kk) row[kk] = fma(-l, bcast(l, kk), row[kk]);
        }
        const long long t1 = __builtin_readcyclecounter();
        tc += t1 - t0;
        if (Lout && lane < N) {
#pragma unroll
            for (int j = 0; j < N; ++j) Lout[(size_t)m * N * N + (size_t)lane * N + j] = (j <= lane && ok) ? row[j] : 0.0;
        }
        if (MODE & 2) {
            double x[N];
            const int c = lane;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double rd = 1.0 / bcast(row[i], i);   // one division per row, uniform
                double t0a = 0.0, t1a = 0.0;
#pragma unroll
                for (int l = 0; l < i; ++l) {
                    const double lil = bcast(row[l], i);      // L[i][l] from lane i
                    if (l & 1) t1a = fma(lil, x[l], t1a); else t0a = fma(lil, x[l], t0a);
                }
                x[i] = (i == c) ? rd : (i > c ? -(t0a + t1a) * rd : 0.0);
            }
            const long long t2 = __builtin_readcyclecounter();
            ti += t2 - t1;
            if (Xout && lane < N) {
#pragma unroll
                for (int i = 0; i < N; ++i) Xout[(size_t)m * N * N + (size_t)i * N + lane] = x[i];
            }
        }
    }
    if (lane == 0) { cyc[2 * blockIdx.x] = tc; cyc[2 * blockIdx.x + 1] = ti; }
}

int main() {
    const int nmat = 2048 * 8, per_cu = 8, cus = 256;
    std::vector<double> hA((size_t)nmat * N * N);
    srand(1);
    for (int m = 0; m < nmat; ++m) {
        std::vector<double> B(N * N);
        for (auto &v : B) v = (rand() / (double)RAND_MAX) - 0.5;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double t = (i == j) ? 3.0 : 0.0;
                for (int q = 0; q < N; ++q) t += B[i * N + q] * B[j * N + q];
                hA[(size_t)m * N * N + i * N + j] = t;
            }
    }
    double *dA, *dL, *dX;
    long long *dc;
    hipMalloc(&dA, hA.size() * 8); hipMalloc(&dL, hA.size() * 8); hipMalloc(&dX, hA.size() * 8);
    const int grid = per_cu * cus;
    hipMalloc(&dc, grid * 16);
    hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, dA, dL, dX, dc, nmat);
        hipDeviceSynchronize();
    }
    std::vector<long long> hc(grid * 2);
    hipMemcpy(hc.data(), dc, grid * 16, hipMemcpyDeviceToHost);
    double tc = 0, ti = 0;
    for (int b = 0; b < grid; ++b) { tc += hc[2 * b]; ti += hc[2 * b + 1]; }
    printf("register-resident n=%d, %d workgroups/CU: Cholesky %.0f cycles/matrix, inverse %.0f cycles/matrix\n", N, per_cu, tc / nmat, ti / nmat);
    std::vector<double> hL(hA.size()), hX(hA.size());
    hipMemcpy(hL.data(), dL, hL.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hX.data(), dX, hX.size() * 8, hipMemcpyDeviceToHost);
    double eL = 0, eX = 0;
    for (int m = 0; m < 64; ++m) {
        const double *a = &hA[(size_t)m * N * N], *L = &hL[(size_t)m * N * N], *X = &hX[(size_t)m * N * N];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j <= i; ++j) {
                double t = 0, u = 0;
                for (int q = 0; q <= j; ++q) t += L[i * N + q] * L[j * N + q];
                eL = fmax(eL, fabs(t - a[i * N + j]));
                for (int q = j; q <= i; ++q) u += L[i * N + q] * X[q * N + j];   // L X = I
                eX = fmax(eX, fabs(u - (i == j ? 1.0 : 0.0)));
            }
    }
    printf("max |L L^T - A| = %.3e, max |L X - I| = %.3e\n", eL, eX);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, dA, (double *)nullptr, (double *)nullptr, dc, nmat);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d matrices (Cholesky + inverse): %.3f ms -> %.2f ms per 100k matrices\n", nmat, ms, ms * 100000.0 / nmat);
    return 0;
}
