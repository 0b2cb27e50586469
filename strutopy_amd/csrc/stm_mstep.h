// stm_mstep.h -- device side of the M-step (reference src/modules/stm.py:622-747) that closes
// one EM iteration without moving eta / mu / beta off the GPU, plus the RCCL binding.
//
//   moments_kernel     local regression moments for update_mu (stm.py:678-706)
//   set_mu_kernel      mu = X @ gamma^T (stm.py:706) or the CTM column mean (stm.py:651)
//   covariance_kernel  (eta - mu)^T (eta - mu) (stm.py:723)
//   beta kernels       beta = beta_ss / rowsum (stm.py:741-745), word-major on the device
//
// All of these are thin HBM-bound streaming kernels over N x (K-1) or V x K doubles; the
// tiny dense solves (p x p regression, (K-1)^2 Cholesky of Sigma) stay on the host, in the
// Python mirror of the reference's M-step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stm {

// per-block partials of [ N | sum_x (p) | sum_eta (n) | XtX (p*p) | Xt_eta (p*n) ]
// one thread per output slot, docs strided over blocks; reduced by reduce_sigma_kernel
__global__ __launch_bounds__(256) void moments_kernel(const double *X, const double *eta, int64_t N,
                                                      int p, int n, double *part, int L) {
    const int64_t chunk = (N + gridDim.x - 1) / gridDim.x;
    const int64_t d0 = (int64_t)blockIdx.x * chunk;
    const int64_t d1 = d0 + chunk < N ? d0 + chunk : N;
    // sixteen documents in flight per thread (eight independent partial sums, fixed order): the loop is a chain of memory round trips
    auto sum4 = [&](auto term) {
        double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int64_t d = d0;
        for (; d + 15 < d1; d += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = term(d + u);
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u & 7] += v[u];
        }
        for (; d < d1; ++d) t[0] += term(d);
        return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    };
    for (int slot = threadIdx.x; slot < L; slot += blockDim.x) {
        double t = 0.0;
        int s = slot;
        if (s == 0) {
            t = (double)(d1 > d0 ? d1 - d0 : 0);
        } else if ((s -= 1) < p) {
            t = sum4([&](int64_t d) { return X[d * p + s]; });
        } else if ((s -= p) < n) {
            t = sum4([&](int64_t d) { return eta[d * n + s]; });
        } else if ((s -= n) < p * p) {
            const int a = s / p, b = s % p;
            t = sum4([&](int64_t d) { return X[d * p + a] * X[d * p + b]; });
        } else {
            s -= p * p;
            const int a = s / n, i = s % n;
            t = sum4([&](int64_t d) { return X[d * p + a] * eta[d * n + i]; });
        }
        part[(size_t)blockIdx.x * L + slot] = t;
    }
}

// mu[d][i] = sum_q X[d][q] * gamma[i][q]   (gamma [(n)][p]; stm.py:703-706: no intercept)
// or mu[d][i] = mean_eta[i] when X == nullptr (CTM, stm.py:651)
__global__ void set_mu_kernel(const double *X, const double *gamma, const double *mean_eta,
                              int64_t N, int p, int n, double *mu) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N * n) return;
    const int64_t d = q / n;
    const int i = (int)(q % n);
    if (!X) { mu[q] = mean_eta[i]; return; }
    double t = 0.0;
    for (int a = 0; a < p; ++a) t += X[d * p + a] * gamma[(size_t)i * p + a];
    mu[q] = t;
}

// per-block partials of cov[i][j] = sum_d (eta-mu)[d][i] (eta-mu)[d][j] (mu == nullptr: eta^T eta); any n.
// 256 threads as 16 x 16: thread (ty, tx) owns the 4 x 4 cells rows 64 blockIdx.z + 4 ty .., columns 64 blockIdx.y + 4 tx ..
// (grid y = z = ceil(n / 64)) -- eight doubles read from the LDS tile per document and sixteen FMAs, where one column x
// sixteen rows read seventeen; every cell still adds its documents in order.  The tile holds the block's 64 row-side
// components (first half) and its 64 column-side components (second half) of 32 documents.
__global__ __launch_bounds__(256) void covariance_kernel(const double *eta, const double *mu, int64_t N,
                                                         int n, double *part) {
    constexpr int TD = 32;  // documents per LDS tile
    __shared__ __attribute__((aligned(16))) double diff[TD][130];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int jc = 4 * tx + 64 * blockIdx.y, ib = 4 * ty + 64 * blockIdx.z;
    const int ia = 4 * ty, ja = 64 + 4 * tx;   // where this thread's components sit in a tile row
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    const int64_t chunk = (N + gridDim.x - 1) / gridDim.x;
    const int64_t d0 = (int64_t)blockIdx.x * chunk;
    const int64_t d1 = d0 + chunk < N ? d0 + chunk : N;
    for (int64_t base = d0; base < d1; base += TD) {
        const int cnt = (int)((d1 - base) < TD ? (d1 - base) : TD);
        {   // the tile: all of a thread's sixteen loads in flight, then its stores (a rolled loop is sixteen round trips in sequence)
            double v[TD * 128 / 256];
#pragma unroll
            for (int it = 0; it < TD * 128 / 256; ++it) {
                const int q = threadIdx.x + 256 * it, dd = q >> 7, c = q & 127;
                const int i = (c < 64 ? 64 * (int)blockIdx.z : 64 * (int)blockIdx.y - 64) + c;   // the component this tile cell holds
                const bool in = dd < cnt && i < n;
                const int64_t at = in ? (base + dd) * n + i : 0;
                const double e = eta[at], m = mu ? mu[at] : 0.0;
                v[it] = in ? (mu ? e - m : e) : 0.0;
            }
#pragma unroll
            for (int it = 0; it < TD * 128 / 256; ++it) {
                const int q = threadIdx.x + 256 * it;
                diff[q >> 7][q & 127] = v[it];
            }
        }
        __syncthreads();
        for (int dd = 0; dd < cnt; ++dd) {
            const double2 *row = reinterpret_cast<const double2 *>(&diff[dd][0]);
            const double2 a01 = row[ia >> 1], a23 = row[(ia >> 1) + 1], b01 = row[ja >> 1], b23 = row[(ja >> 1) + 1];
            const double a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = ib + u, j = jc + v;
            if (i < n && j < n) part[(size_t)blockIdx.x * n * n + (size_t)i * n + j] = acc[u][v];
        }
}

// column sums of the word-major beta_ss: part[block][k] = sum over the block's words of bssT[v][k]
__global__ __launch_bounds__(256) void beta_rowsum_kernel(const double *bssT, int V, int K, double *part) {
    const int kl = threadIdx.x & 63, sub = threadIdx.x >> 6;  // 4 word lanes x 64 topics (x blockIdx.y)
    const int k = kl + 64 * blockIdx.y;
    __shared__ double sh[4][64];
    const int chunk = (V + gridDim.x - 1) / gridDim.x;
    const int v0 = blockIdx.x * chunk, v1 = v0 + chunk < V ? v0 + chunk : V;
    double t = 0.0;
    if (k < K)
        for (int v = v0 + sub; v < v1; v += 4) t += bssT[(size_t)v * K + k];
    sh[sub][kl] = t;
    __syncthreads();
    if (sub == 0 && k < K) part[(size_t)blockIdx.x * K + k] = (sh[0][kl] + sh[1][kl]) + (sh[2][kl] + sh[3][kl]);
}
// betaT[v][k] = bssT[v][k] / rowsum[k] where rowsum != 0 else 0 (2-D beta, stm.py:741-745)
__global__ void beta_normalise_kernel(const double *bssT, const double *rowsum, int64_t VK, int K, double *betaT) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= VK) return;
    const double rs = rowsum[q % K];
    betaT[q] = (rs != 0.0) ? bssT[q] / rs : 0.0;
}
// 3-D beta: the reference's np.sum(beta_ss, axis=1) runs over TOPICS, so every (level, word)
// column is normalised by its own sum over k (stm.py:741 with a 3-D array)
__global__ void beta_normalise_topics_kernel(const double *bssT, int64_t AV, int K, double *betaT) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= AV) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += bssT[r * K + k];
    for (int k = 0; k < K; ++k) betaT[r * K + k] = (s != 0.0) ? bssT[r * K + k] / s : 0.0;
}

// Held-out per-word log-likelihood of one document per wavefront (reference src/modules/heldout.py:88-97):
//   doc_ll[d] = sum_w c_w log(theta_d . beta[:, w]) / sum_w c_w      (lane = word, topics in order)
__global__ __launch_bounds__(64) void heldout_kernel(const int64_t *indptr, const int32_t *indices, const double *counts,
                                                     const double *betaT, const double *theta, int64_t N, int K,
                                                     double *doc_ll) {
    const int64_t d = blockIdx.x;
    if (d >= N) return;
    const int lane = threadIdx.x;
    const int64_t p0 = indptr[d];
    const int nd = (int)(indptr[d + 1] - p0);
    const double *th = theta + d * K;
    double num = 0.0, den = 0.0;
    for (int v = lane; v < nd; v += 64) {
        const double *row = betaT + (size_t)indices[p0 + v] * K;
        double dot = 0.0;
        for (int k = 0; k < K; ++k) dot += th[k] * row[k];
        const double c = counts[p0 + v];
        num += c * log(dot);
        den += c;
    }
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o); den += __shfl_xor(den, o); }
    if (lane == 0) doc_ll[d] = num / den;
}

}  // namespace stm

// RCCL binding (resolved lazily with dlopen so a single-GPU run never loads librccl)
void stm_mstep_comm_destroy(void *comm);
