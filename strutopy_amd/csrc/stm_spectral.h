// stm_spectral.h -- device side of the spectral initialisation (reference src/modules/stm.py:30-296,
// the init_type src/05_train.py:92 uses).
//
//   gram_kernel        Q = Htilde^T Htilde - diag(Hhat) (stm.py:122-157) as a dense Vk x Vk matrix, one workgroup
//                      per word: the word's row is accumulated in LDS over the documents that contain it, in
//                      ascending document order (the order scipy's sparse product adds them in) -- no atomics,
//                      run-to-run identical
//   fastAnchor pieces  (stm.py:160-226) column sums of squares, first-maximum search, row scaling,
//                      Q @ Q[m]^T, rank-one projection off every row outside `basis`
//   anchor_project     q_i = M y_i for every word i (M = anchor rows; stm.py:239, :266-270): the inputs of the per-word QP
//   nnls_kernel        the per-word QPs themselves (stm.py:271-285), one thread per word
// All HBM-bound streaming passes over the 8 Vk^2-byte matrix (200 MB at maxV = 5000).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stm_wave.h"

namespace stm {

// one workgroup per word a (row of Q); dynamic LDS: row[Vk]
__global__ __launch_bounds__(256) void gram_kernel(const int64_t *doc_ptr, const int32_t *doc_word, const double *doc_h,
                                                   const int64_t *word_ptr, const int32_t *word_doc, const double *word_h,
                                                   const double *hhat, int Vk, double *Q, int32_t *err_flag) {
    extern __shared__ double grow[];
    const int a = blockIdx.x;
    for (int c = threadIdx.x; c < Vk; c += blockDim.x) grow[c] = 0.0;
    __syncthreads();
    const int64_t w0 = word_ptr[a], w1 = word_ptr[a + 1];
    for (int64_t e = w0; e < w1; ++e) {          // documents containing word a, ascending
        const int d = word_doc[e];
        const double ha = word_h[e];
        const int64_t p0 = doc_ptr[d], p1 = doc_ptr[d + 1];
        for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x)   // words of a document are unique: no two threads share a cell
            grow[doc_word[p]] += ha * doc_h[p];
        __syncthreads();                          // the next document may touch the same cells from other threads
    }
    if (threadIdx.x == 0) grow[a] -= hhat[a];
    __syncthreads();
    // assert np.all(Q.sum(axis=1) > 0) (stm.py:152-154): fixed-order block sum
    __shared__ double part[256];
    double t = 0.0;
    for (int c = threadIdx.x; c < Vk; c += blockDim.x) {
        const double v = grow[c];
        Q[(size_t)a * Vk + c] = v;
        t += v;
    }
    part[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && !(part[0] > 0.0)) atomicMax(err_flag, 1);
}

// ---- gram on the handle's RESIDENT corpus (document shard): no host preparation of scaled orientations.
// document length over the kept terms -> div[d] = n_d (n_d - 1) (stm.py:135-141), one wave per document
__global__ __launch_bounds__(64) void gram_docdiv_kernel(const int64_t *indptr, const int32_t *indices, const double *counts,
                                                         const int32_t *pos, int64_t N, double *div) {
    const int64_t d = blockIdx.x;
    if (d >= N) return;
    double t = 0.0;
    for (int64_t p = indptr[d] + threadIdx.x; p < indptr[d + 1]; p += 64) t += pos[indices[p]] >= 0 ? counts[p] : 0.0;   // counts are integers: exact in any order
    t = wave_sum(t);
    if (threadIdx.x == 0) div[d] = t * (t - 1.0);
}
// per CSR entry: kept-term id (or -1) and h = count / sqrt(n_d (n_d - 1)) (stm.py:142-146), one wave per document
__global__ __launch_bounds__(64) void gram_scale_kernel(const int64_t *indptr, const int32_t *indices, const double *counts,
                                                        const int32_t *pos, const double *div, int64_t N, int32_t *ent_j, double *ent_h) {
    const int64_t d = blockIdx.x;
    if (d >= N) return;
    const double sq = sqrt(div[d]);
    for (int64_t p = indptr[d] + threadIdx.x; p < indptr[d + 1]; p += 64) {
        ent_j[p] = pos[indices[p]];
        ent_h[p] = counts[p] / sq;
    }
}
// gram_kernel over the resident CSR: word_q[e] is the CSR position of (document word_doc[e], kept term a), documents
// ascending within a term.  Hhat[a] = sum_d count / (n_d (n_d - 1)) is summed along the way (thread 0, in document order).
__global__ __launch_bounds__(256) void gram_resident_kernel(const int64_t *indptr, const int32_t *ent_j, const double *ent_h,
                                                            const double *counts, const double *div, const int64_t *word_ptr,
                                                            const int32_t *word_doc, const int32_t *word_q, int Vk, double *Q) {
    extern __shared__ double grow[];
    const int a = blockIdx.x;
    for (int c = threadIdx.x; c < Vk; c += blockDim.x) grow[c] = 0.0;
    __syncthreads();
    const int64_t w0 = word_ptr[a], w1 = word_ptr[a + 1];
    double hh = 0.0;
    for (int64_t e = w0; e < w1; ++e) {
        const int d = word_doc[e];
        const int q = word_q[e];
        const double ha = ent_h[q];
        if (threadIdx.x == 0) hh += counts[q] / div[d];
        const int64_t p0 = indptr[d], p1 = indptr[d + 1];
        for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
            const int j = ent_j[p];
            if (j >= 0) grow[j] += ha * ent_h[p];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) grow[a] -= hh;
    __syncthreads();
    for (int c = threadIdx.x; c < Vk; c += blockDim.x) Q[(size_t)a * Vk + c] = grow[c];
}
// assert np.all(Q.sum(axis=1) > 0) (stm.py:152-154) on the complete (all-reduced) matrix: one block per row, fixed-order sum
__global__ __launch_bounds__(256) void gram_rowsum_check_kernel(const double *Q, int Vk, int32_t *err_flag) {
    __shared__ double part[256];
    const int a = blockIdx.x;
    double t = 0.0;
    for (int c = threadIdx.x; c < Vk; c += blockDim.x) t += Q[(size_t)a * Vk + c];
    part[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && !(part[0] > 0.0)) atomicMax(err_flag, 1);
}

// partial column sums of squares: part[blockIdx.y][c] = sum over the block's rows of Q[r][c]^2
__global__ __launch_bounds__(256) void colsq_kernel(const double *Q, int Vk, double *part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int chunk = (Vk + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * chunk, r1 = r0 + chunk < Vk ? r0 + chunk : Vk;
    if (c >= Vk) return;
    double t0 = 0.0, t1 = 0.0;
    int r = r0;
    for (; r + 1 < r1; r += 2) {
        const double a = Q[(size_t)r * Vk + c], b = Q[(size_t)(r + 1) * Vk + c];
        t0 += a * a; t1 += b * b;
    }
    if (r < r1) { const double a = Q[(size_t)r * Vk + c]; t0 += a * a; }
    part[(size_t)blockIdx.y * Vk + c] = t0 + t1;
}

// row_squared_sum[:, basis] = 0 (stm.py:222; `basis` still holds zeros for the anchors not chosen yet, so entry 0 goes
// too), then the FIRST index of the maximum (np.argmax) and 1 / sqrt(max) (stm.py:181-184).  One block.
__global__ __launch_bounds__(1024) void anchor_pick_kernel(double *rss, int Vk, const int32_t *basis, int nbasis, int zero_first,
                                                           int32_t *pick, double *normalizer) {
    __shared__ double bv[1024];
    __shared__ int bi[1024];
    if (zero_first) {
        for (int q = threadIdx.x; q < nbasis; q += 1024) rss[basis[q]] = 0.0;
        __syncthreads();
    }
    double best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = threadIdx.x; c < Vk; c += 1024) {
        const double v = rss[c];
        if (v > best || (v == best && c < arg)) { best = v; arg = c; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = arg;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = bv[threadIdx.x + o];
            const int c = bi[threadIdx.x + o];
            if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && c < bi[threadIdx.x])) { bv[threadIdx.x] = v; bi[threadIdx.x] = c; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *pick = bi[0]; *normalizer = 1.0 / sqrt(bv[0]); }
}

// Q[m] = Q[m] * normalizer (stm.py:185); m and the factor are read from device memory
__global__ void scale_row_kernel(double *Q, int Vk, const int32_t *pick, const double *normalizer) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Vk) Q[(size_t)(*pick) * Vk + c] *= *normalizer;
}

// inner[r] = Q[r] . Q[m] (stm.py:188-193): one wavefront per row, fixed order
__global__ __launch_bounds__(64) void row_dot_kernel(const double *Q, int Vk, const int32_t *pick, double *inner) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const double *a = Q + (size_t)r * Vk, *b = Q + (size_t)(*pick) * Vk;
    double t0 = 0.0, t1 = 0.0;
    int c = lane;
    for (; c + 64 < Vk; c += 128) { t0 = fma(a[c], b[c], t0); t1 = fma(a[c + 64], b[c + 64], t1); }
    if (c < Vk) t0 = fma(a[c], b[c], t0);
    const double t = wave_sum(t0 + t1);
    if (lane == 0) inner[r] = t;
}

// Q[r] -= inner[r] * Q[m] for every row r outside `basis` (stm.py:205-217); skip[r] != 0 marks the excluded rows
__global__ __launch_bounds__(256) void project_off_kernel(double *Q, int Vk, const int32_t *pick, const double *inner, const uint8_t *skip) {
    const int r = blockIdx.y;
    if (skip[r]) return;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Vk) return;
    Q[(size_t)r * Vk + c] -= inner[r] * Q[(size_t)(*pick) * Vk + c];
}

// q[i][k] = Q[i] . M[k], M[k] = Q[anchor[k]] (stm.py:239, 266-270): one wavefront per word i, its row held in registers
// in chunks of 64 * RCH columns
__global__ __launch_bounds__(64) void anchor_project_kernel(const double *Q, int Vk, const int32_t *anchor, int K, double *q) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double *a = Q + (size_t)i * Vk;
    for (int k = 0; k < K; ++k) {
        const double *b = Q + (size_t)anchor[k] * Vk;
        double t0 = 0.0, t1 = 0.0;
        int c = lane;
        for (; c + 64 < Vk; c += 128) { t0 = fma(a[c], b[c], t0); t1 = fma(a[c + 64], b[c + 64], t1); }
        if (c < Vk) t0 = fma(a[c], b[c], t0);
        const double t = wave_sum(t0 + t1);
        if (lane == 0) q[(size_t)i * K + k] = t;
    }
}

// recover_l2's per-term QP (stm.py:257-285):  x = argmin 1/2 x^T P x + q_i^T x  s.t. x <= 0,  weights_i = -x, i.e. the
// non-negative least-squares fit  w = argmin_{w >= 0} 1/2 w^T P w - q_i^T w  (P = M M^T positive definite: the minimiser is unique,
// whatever solver finds it -- the reference calls quadprog).  Lawson-Hanson active set on the normal equations, one THREAD per
// term (5000 small independent problems; scalar code, per-thread state in scratch, the passive-set Cholesky factor in a private
// slice of global memory).  Anchor terms get their one-hot row (stm.py:261-264).
// A column whose passive-set factorisation fails (numerically dependent) or that is dropped again by a zero-length step is
// BANNED until w next changes -- it would otherwise be re-picked and re-factorised until the iteration cap.  Every term ends
// with a check of the QP's KKT conditions on its result; *nbad counts the terms that fail it (iteration cap, or a violated
// dual left behind by a ban): the reference's quadprog raises on a P that is not positive definite, stm_spectral_weights
// reports these the same way instead of returning weights that are not the minimiser.
constexpr int NNLS_KMAX = 512;   // (= K_LIMIT, stm_post_common.h)
template <int KMAX>   // 128 or NNLS_KMAX: the per-thread vectors live in scratch memory, sized at compile time
__global__ __launch_bounds__(64) void nnls_kernel(const double *q, const int32_t *anchor, int K, int Vk, double *fac /* [Vk][K][K] */,
                                                   double *weights /* [Vk][K] */, int32_t *nbad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Vk) return;
    double *w_out = weights + (size_t)i * K;
    for (int k = 0; k < K; ++k) w_out[k] = 0.0;
    for (int k = 0; k < K; ++k)
        if (anchor[k] == i) {              // vec[np.where(anchor == i)] = 1
            w_out[k] = 1.0;
            return;
        }
    const double *qi = q + (size_t)i * K;
    // P = M M^T = the rows `anchor` of q (q[a][k] = Q[a] . Q[anchor[k]])
    auto Pm = [&](int a, int b) -> double { return q[(size_t)anchor[a] * K + b]; };
    double w[KMAX], z[KMAX], rhs[KMAX];
    int idx[KMAX];
    bool inS[KMAX], banned[KMAX];
    double qmax = 0.0;
    for (int k = 0; k < K; ++k) { w[k] = 0.0; inS[k] = false; banned[k] = false; qmax = fmax(qmax, fabs(qi[k])); }
    const double tol = 1e-13 * fmax(qmax, 1e-300) * K;
    double *L = fac + (size_t)i * K * K;    // lower triangular factor of P_SS, row-major with leading dimension K
    int it = 0;
    for (; it < 3 * K; ++it) {
        // dual = q - P w over the free terms; the most violated one joins the passive set
        int jbest = -1;
        double dbest = tol;
        for (int j = 0; j < K; ++j) {
            if (inS[j] || banned[j]) continue;
            double d = qi[j];
            for (int k = 0; k < K; ++k)
                if (inS[k]) d -= Pm(j, k) * w[k];
            if (d > dbest) { dbest = d; jbest = j; }
        }
        if (jbest < 0) break;
        inS[jbest] = true;
        for (int inner = 0; inner < 3 * K; ++inner) {
            int s = 0;
            for (int k = 0; k < K; ++k)
                if (inS[k]) idx[s++] = k;
            // Cholesky of P_SS, then L y = q_S, L^T z = y
            bool ok = true;
            for (int a = 0; a < s && ok; ++a) {
                for (int b = 0; b <= a; ++b) {
                    double t = Pm(idx[a], idx[b]);
                    for (int c = 0; c < b; ++c) t -= L[a * K + c] * L[b * K + c];
                    if (a == b) {
                        if (!(t > 0.0)) { ok = false; break; }
                        L[a * K + a] = sqrt(t);
                    } else {
                        L[a * K + b] = t / L[b * K + b];
                    }
                }
            }
            if (!ok) { inS[jbest] = false; w[jbest] = 0.0; banned[jbest] = true; break; }   // numerically dependent column: leave it out
            for (int a = 0; a < s; ++a) {
                double t = qi[idx[a]];
                for (int c = 0; c < a; ++c) t -= L[a * K + c] * rhs[c];
                rhs[a] = t / L[a * K + a];
            }
            for (int a = s - 1; a >= 0; --a) {
                double t = rhs[a];
                for (int c = a + 1; c < s; ++c) t -= L[c * K + a] * z[c];
                z[a] = t / L[a * K + a];
            }
            bool allpos = true;
            for (int a = 0; a < s; ++a) allpos = allpos && (z[a] > 0.0);
            if (allpos) {
                for (int a = 0; a < s; ++a) w[idx[a]] = z[a];
                for (int k = 0; k < K; ++k) banned[k] = false;   // w moved: every column is a candidate again
                break;
            }
            double alpha = 1.0;
            for (int a = 0; a < s; ++a)
                if (!(z[a] > 0.0)) {
                    const double wa = w[idx[a]];
                    const double r = wa / (wa - z[a]);
                    if (r < alpha) alpha = r;
                }
            for (int a = 0; a < s; ++a) {
                const int k = idx[a];
                w[k] += alpha * (z[a] - w[k]);
                if (!(w[k] > tol * 1e-3)) { w[k] = 0.0; inS[k] = false; }
            }
            if (alpha > 0.0) { for (int k = 0; k < K; ++k) banned[k] = false; }
            else if (!inS[jbest]) { banned[jbest] = true; break; }   // a zero-length step threw the new column out again
        }
    }
    for (int k = 0; k < K; ++k) w_out[k] = w[k];
    // KKT of  min 1/2 w'Pw - q'w, w >= 0:  w >= 0 (by construction), dual d = q - P w <= 0 where w = 0, d = 0 where w > 0
    bool bad = it >= 3 * K;
    const double ktol = 1e-8 * fmax(qmax, 1e-300) * K;
    for (int j = 0; j < K; ++j) {
        double d = qi[j];
        for (int k = 0; k < K; ++k)
            if (inS[k]) d -= Pm(j, k) * w[k];
        bad = bad || (inS[j] ? fabs(d) > ktol : d > ktol);
    }
    if (bad && nbad) atomicAdd(nbad, 1);
}

}  // namespace stm
