// stm_post_common.h -- what the post-solve kernels (stm_post.h for K <= 64, stm_post_big2.h for 64 < K <= 112, stm_post_any.h beyond) share:
// the parameter block, the Cholesky pivot tolerance, the fixed-order reductions of the replicated nu accumulators and of
// the per-document bounds (reference src/modules/stm.py:582, 592).
#pragma once
#include <type_traits>
#include "stm_wave.h"

namespace stm {

struct PostParams {
    int64_t N;
    int K, n, V;
    const int64_t *indptr;
    const int32_t *indices;
    const double *counts;
    const int32_t *aspect;
    const double *betaT;   // [A][V][K]
    const double *mu;      // [N][n]
    const double *eta;     // [N][n]
    const double *siginv;  // [n][n]
    int siginv_diag;
    double sigmaentropy;
    double *theta;         // [N][K]
    double *bound;         // [N]
    double *beta_ssT;      // [A][V][K], pre-zeroed, atomically accumulated
    double *sigma_part;    // [nrep][n][n] replicated accumulators of nu (pre-zeroed, atomics)
    int nrep;
    int64_t first;         // this launch covers order[first .. first + count)
    int64_t count;
    const int32_t *order;
    const int64_t *tick;   // optional [N][2]: ticket -> {indptr[doc], doc | Nd << 32} (the header in one scalar load; nullable)
    int32_t *pd_path;
    int32_t *err_flag;
    double *hess_out, *chol_out, *nu_out;  // optional [N][n][n] dumps (nullable)
    int debug_flags;       // timing experiments only: 1 skip phi atomics, 2 skip b b^T, 4 skip nu, 8 skip Cholesky;
                           // 16 (tests): NaN into the whole LDS allocation before every document
    int lds_doubles;       // size of the dynamic LDS allocation
    double *a_scratch;     // post_any_kernel: per-workgroup HBM scratch (A, L, b, sqrt(c))
    int64_t phi_doc;       // document whose phi is dumped (-1: none)
    double *phi_out;       // [K][Nd(phi_doc)]
    long long *prof;       // optional [N][PROF_SLOTS] (shared with the solver's): [32..39] post-kernel phase cycles
    double *rw;            // post_kernel (K <= 64): [nnz] r_dw of stm_betass.h, word-major ...
    const int32_t *wm_slot; // ... at wm_slot[CSR position]
    int nd_max;            // post_any_kernel: words of the longest document (sizes the per-workgroup scratch behind a_scratch)
};
constexpr int K_LIMIT = 512;   // topics: eight vector components per lane in the solver's general form (stm_solver.h, VPL = 8)

// A Cholesky pivot that is only the rounding left over from cancelling the diagonal entry counts as failed (as in the
// oracle): make_pd can leave an exactly singular matrix (n = 2: always when both diagonals are raised), and the sign of such a
// pivot -- like the sign of the smallest eigenvalue the reference tests, stm.py:1017 -- hangs on the last bit of the input.
constexpr double PIVOT_TOL = 32.0 * 2.220446049250313e-16;
constexpr int PT = 64;    // topics padded to 64 (K <= 64 in this kernel)
constexpr int TW = 16;    // words per tile
constexpr int TLD = 18;   // leading dimension of T: MFMA fragment reads are conflict-free

typedef double v4d __attribute__((ext_vector_type(4)));

// LDS hand-off between lanes of the (single) wave of a workgroup: the LDS executes a wave's operations in
// order, so only the compiler must not reorder them -- unlike __syncthreads() this does not drain the
// global loads that are deliberately kept in flight across the hand-off
#define STM_POST_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)


// out = sum over the replicated / per-block partial copies, in a fixed order: 64 slots per block, the copies
// split over four 64-thread groups, four independent partial sums per thread, then a fixed combine.
// gridDim.y > 1: block row y sums the copies [y * chunk, (y + 1) * chunk) into out + y * nn (the first of two stages:
// one thread walking thousands of copies is a chain of dependent memory round trips)
__global__ __launch_bounds__(256) void reduce_sigma_kernel(const double *part, int nblocks, int nn, double *out, int chunk) {
    __shared__ double sh[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + tx;
    const int lo = blockIdx.y * chunk, hi = lo + chunk < nblocks ? lo + chunk : nblocks;
    const int cnt = hi > lo ? hi - lo : 0;
    const int per = (cnt + 3) >> 2;
    const int b0 = lo + ty * per, b1 = b0 + per < hi ? b0 + per : hi;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (q < nn) {
        int b = b0;
        for (; b + 3 < b1; b += 4) {
            const double a = part[(size_t)b * nn + q], c = part[(size_t)(b + 1) * nn + q];
            const double d = part[(size_t)(b + 2) * nn + q], e = part[(size_t)(b + 3) * nn + q];
            t0 += a; t1 += c; t2 += d; t3 += e;
        }
        for (; b < b1; ++b) t0 += part[(size_t)b * nn + q];
    }
    sh[ty][tx] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (ty == 0 && q < nn) out[(size_t)blockIdx.y * nn + q] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}

// per-block sums of contiguous chunks of the per-document bounds (first stage of reduce_bound_kernel)
__global__ __launch_bounds__(256) void bound_partial_kernel(const double *bound, int64_t N, double *part) {
    __shared__ double sh[256];
    const int64_t chunk = (N + gridDim.x - 1) / gridDim.x;
    const int64_t d0 = (int64_t)blockIdx.x * chunk, d1 = d0 + chunk < N ? d0 + chunk : N;
    double t = 0.0;
    for (int64_t i = d0 + threadIdx.x; i < d1; i += 256) t += bound[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// bound = np.sum(calculated_bounds) (stm.py:592): one block, fixed tree => deterministic.  out[1] = this rank's device
// error flag (1.0 per failing rank after the all-reduce of the packed buffer: every rank learns of a failure anywhere in
// the same collective), out[2] = the flag's value on this rank
__global__ __launch_bounds__(1024) void reduce_bound_kernel(const double *bound, int64_t N, double *out, const int32_t *err) {
    __shared__ double sh[1024];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 1024) t += bound[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = sh[0];
        out[1] = (err && *err) ? 1.0 : 0.0;
    }
}


// sigma_ss[i][j] = sigma_ss[j][i] for the 16 x 16 blocks below the block diagonal (the post kernels add nu's upper block triangle only)
__global__ void mirror_blocks_kernel(double *a, int n) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * n) return;
    const int i = q / n, j = q % n;
    if ((i >> 4) > (j >> 4)) a[q] = a[(size_t)j * n + i];
}

// sigma_ss from the accumulator-tile layout the K <= 64 post kernel sums nu in (stm_post.h): tile (b, bj), b <= bj, at
// bj (bj + 1) / 2 + b holds element (16 b + fq + 4 r, 16 bj + fr) at [r][lane = 16 fq + fr]; cells below the diagonal
// take their mirror image (nu is symmetric, and only the upper block triangle is accumulated); rem: n = 16 NB + 1 and the
// last column sits in a slot of its own (post_kernel<NB, 1, ...>)
__global__ void untile_sigma_kernel(const double *tiles, int n, double *out, int rem) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * n) return;
    int i = q / n, j = q % n;
    if (i > j) { const int t = i; i = j; j = t; }
    if (rem && j == n - 1) {   // the column beyond the full blocks: its own slot behind their tiles, entry i
        const int nb = (n - 1) >> 4;
        out[q] = tiles[(size_t)(nb * (nb + 1) / 2) * 256 + i];
        return;
    }
    const int b = i >> 4, bj = j >> 4, il = i & 15, fr = j & 15;
    out[q] = tiles[((size_t)(bj * (bj + 1) / 2 + b) * 4 + (il >> 2)) * 64 + (il & 3) * 16 + fr];
}

}  // namespace stm
