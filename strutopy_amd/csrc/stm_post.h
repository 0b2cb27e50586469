// stm_post.h -- per-document theta / Hessian / Cholesky / nu / bound / phi: one wavefront per doc.
//
// Replaces, per document, reference src/modules/stm.py:547-588:
//   theta (547-549), hessian (986-1026, incl. make_pd 964-984 and the +1e-5 branch),
//   decompose_hessian (1031-1050), lower_bound (1068-1101), optimize_nu (1052-1066),
//   update_z (1103-1118) and the sigma_ss / beta_ss accumulation (582-588).
//
// Layout: words of the document are processed in tiles of 64 (lane = word): the lane reads
// its word's contiguous K-vector from betaT[A][V][K], forms b = a*sqrt(c)/colsum(a) and
// phi, scatters phi into beta_ss (word-major, fp64 HW atomics) and parks b in an LDS tile
// bt[64 topics][65] (word contiguous: conflict-free writes; the odd leading dimension keeps
// the strided block reads at <= 2-way).  The K x K contraction b b^T is then accumulated
// from LDS with an 8x8 register block per lane (lane grid 8x8 -> 64x64 outputs), so each
// LDS value feeds 8 FMAs.  The (K-1)^2 matrix then lives in ONE padded LDS array (leading dimension 65 =>
// row- and column-wise lane access are both bank-conflict-free): Cholesky overwrites the
// strict lower triangle with L, the untouched upper triangle still holds A for the
// make_pd fallbacks, and U^{-1} later overwrites the upper triangle.
#pragma once
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
    int64_t first;         // this launch covers order[first .. first + gridDim.x)
    const int32_t *order;
    int32_t *pd_path;
    int32_t *err_flag;
    double *hess_out, *chol_out, *nu_out;  // optional [N][n][n] dumps (nullable)
    int debug_flags;       // timing experiments only: 1 skip phi atomics, 2 skip b b^T, 4 skip nu, 8 skip Cholesky
    int64_t phi_doc;       // document whose phi is dumped (-1: none)
    double *phi_out;       // [K][Nd(phi_doc)]
};

constexpr int PT = 64;        // topics padded to 64 (K <= 64 in this kernel)
constexpr int MLD = 65;       // leading dimension of the LDS matrix

__global__ __launch_bounds__(64) void post_kernel(PostParams P) {
    // bt tile [PT][MLD] and the n x n matrix [64][MLD] share one LDS region (33 KiB)
    __shared__ __attribute__((aligned(16))) double smem[64 * MLD];
    __shared__ double sex[PT];   // exp(eta~)            (unshifted, stm.py:1000,1088,1114)
    __shared__ double sth[PT];   // stable_softmax(eta~) (stm.py:998,1083)
    __shared__ double ssq[WAVE]; // sqrt(count) of the tile's words
    __shared__ double sdv[PT];   // eta - mu broadcast (dense siginv only)
    __shared__ double srow[PT];  // rowsum(c') per topic
    double *bt = smem;
    double *M = smem;
    const int lane = threadIdx.x;
    const int K = P.K, n = P.n;
    const double *S = P.siginv;
    double *sig_acc = P.sigma_part + (size_t)(blockIdx.x % P.nrep) * (size_t)n * n;

    {
        const int64_t ticket = P.first + blockIdx.x;
        if (ticket >= P.N) return;
        const int64_t doc = P.order ? (int64_t)P.order[ticket] : ticket;
        const int64_t p0 = P.indptr[doc];
        const int Nd = (int)(P.indptr[doc + 1] - p0);
        const int ntile = (Nd + WAVE - 1) / WAVE;
        const int asp = P.aspect ? P.aspect[doc] : 0;
        const double *bT = P.betaT + (size_t)asp * (size_t)P.V * K;
        double *bssT = P.beta_ssT + (size_t)asp * (size_t)P.V * K;

        // ---- eta~, theta (unshifted softmax, stm.py:547-549), stable softmax, exp(eta~)
        const bool isn = lane < n, isk = lane < K;
        const double eta_i = isn ? P.eta[doc * n + lane] : 0.0;  // lane K-1 holds the appended 0
        const double mu_i = isn ? P.mu[doc * n + lane] : 0.0;
        const double ex = isk ? exp(eta_i) : 0.0;
        const double sumex = wave_sum(ex);
        if (isk) P.theta[doc * K + lane] = ex / sumex;
        const double m = wave_nanmax(isk ? eta_i : -INFINITY);
        const double es = isk ? exp(eta_i - m) : 0.0;
        const double ssum = wave_sum(es);
        const double ths = es / ssum;
        if (lane < PT) { sex[lane] = ex; sth[lane] = isk ? ths : 0.0; }
        __syncthreads();

        double csum = 0.0, ll = 0.0, rowc = 0.0;
        bool bad = false;
        double acc[8][8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;
        const int br = lane & 7, bc = lane >> 3;
        // topic rows K..63 of the tile stay zero for the whole document
        for (int k = K; k < PT; ++k) bt[(size_t)k * MLD + lane] = 0.0;

        for (int tile = 0; tile < ntile; ++tile) {
            const int v = tile * WAVE + lane;
            double *bcol = bt + lane;  // bt[k][lane]
            if (v < Nd) {
                const int idx = P.indices[p0 + v];
                const double c = P.counts[p0 + v];
                const double *row = bT + (size_t)idx * K;
                double Ssum = 0.0, lls = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double a = row[k] * sex[k];
                    Ssum += a;                 // np.sum(a, 0)
                    lls += sth[k] * a;         // theta @ (beta * exp(eta~)), stm.py:1088-1094
                }
                const double sq = sqrt(c);
                const double w = sq / Ssum;    // update_z: sqrt(c) / colsum, stm.py:1115
                ll += log(lls) * c;
                csum += c;
                double *bss = bssT + (size_t)idx * K;
                for (int k = 0; k < K; ++k) {
                    const double a = row[k] * sex[k];
                    bcol[(size_t)k * MLD] = a * sq / Ssum;   // hessian b, stm.py:1001
                    const double phi = a * w * sq;  // stm.py:1115-1116
                    bad |= !(phi >= 0.0);
                    if (!(P.debug_flags & 1)) unsafeAtomicAdd(bss + k, phi);  // beta_ss[:, idx] += phi, stm.py:588
                    if (P.phi_out && doc == P.phi_doc) P.phi_out[(size_t)k * Nd + v] = phi;
                }
                ssq[lane] = sq;
            } else {
                for (int k = 0; k < K; ++k) bcol[(size_t)k * MLD] = 0.0;
                ssq[lane] = 0.0;
            }
            __syncthreads();
            // rowsum(c') with c' = b * sqrt(c), stm.py:1002,1011
            if (isk)
                for (int vv = 0; vv < WAVE; ++vv) rowc += bt[(size_t)lane * MLD + vv] * ssq[vv];
            // b b^T, 8x8 register block per lane
            if (!(P.debug_flags & 2))
            for (int vv = 0; vv < WAVE; ++vv) {
                const double *rp = bt + (size_t)(8 * br) * MLD + vv;
                const double *cp = bt + (size_t)(8 * bc) * MLD + vv;
                double ra[8], ca[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) { ra[a] = rp[(size_t)a * MLD]; ca[a] = cp[(size_t)a * MLD]; }
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int b = 0; b < 8; ++b) acc[a][b] = fma(ra[a], ca[b], acc[a][b]);
            }
            __syncthreads();
        }
        if (wave_any(bad)) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */);
        const double Ndoc = (double)(long long)wave_sum(csum);
        ll = wave_sum(ll);

        // ---- assemble H = b b^T - N theta theta^T, diag += -rowsum(c') + N theta, [:-1,:-1] + siginv
        // (theta/rowc of row i live in lane i: fetch via LDS)
        if (lane < PT) srow[lane] = rowc;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int i = 8 * br + a;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int j = 8 * bc + b;
                if (i < n && j < n) {
                    double h = acc[a][b] - Ndoc * (sth[i] * sth[j]);
                    if (i == j) h = h - srow[i] + Ndoc * sth[i];
                    M[(size_t)i * MLD + j] = h + S[(size_t)i * n + j];
                }
            }
        }
        __syncthreads();

        // ---- PD handling.  diagA: current diagonal of A (lane i); off-diagonals of A are read
        // from the upper triangle of M, which Cholesky never writes.
        double diagA = isn ? M[(size_t)lane * MLD + lane] : 1.0;
        double Ldiag = 1.0;
        auto cholesky = [&]() -> bool {  // np.linalg.cholesky; L strictly-lower into M, diag in Ldiag
            bool ok = true;
            for (int j = 0; j < n; ++j) {
                double t = 0.0;
                if (isn && lane >= j) {
                    t = (lane == j) ? diagA : M[(size_t)j * MLD + lane];
                    for (int l = 0; l < j; ++l)
                        t -= M[(size_t)lane * MLD + l] * M[(size_t)j * MLD + l];
                }
                const double d = lane_bcast(t, j);
                if (!(d > 0.0)) { ok = false; break; }
                const double ljj = sqrt(d);
                if (lane == j) Ldiag = ljj;
                if (isn && lane > j) M[(size_t)lane * MLD + j] = t / ljj;
                __syncthreads();
            }
            __syncthreads();
            return ok;
        };
        auto make_pd = [&]() {  // stm.py:964-984
            if (isn) {
                double mag = 0.0;
                for (int j = 0; j < n; ++j) {
                    const double aij = (j == lane) ? diagA
                                     : (j > lane ? M[(size_t)lane * MLD + j] : M[(size_t)j * MLD + lane]);
                    mag += fabs(aij);
                }
                mag -= fabs(diagA);
                if (diagA < mag) diagA = mag;
            }
        };
        auto dump = [&](double *base, bool lower_L) {
            if (!base) return;
            double *o = base + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j) {
                    double val;
                    if (lower_L) val = (j == lane) ? Ldiag : (j < lane ? M[(size_t)lane * MLD + j] : 0.0);
                    else val = (j == lane) ? diagA
                             : (j > lane ? M[(size_t)lane * MLD + j] : M[(size_t)j * MLD + lane]);
                    o[(size_t)lane * n + j] = val;
                }
        };
        int path = 0;
        bool upper = false, fail = false;
        bool ok = (P.debug_flags & 8) ? true : cholesky();                 // PD test of hessian(), stm.py:1017 (as Cholesky success)
        if (!ok) {
            make_pd(); path = 1;              // stm.py:1019
            ok = cholesky();                  // stm.py:1020
            if (!ok) {
                if (isn) diagA += 1e-5;       // stm.py:1021
                path = 2;
                dump(P.hess_out, false);
                ok = cholesky();              // decompose_hessian, stm.py:1040
                if (!ok) {
                    make_pd();                // stm.py:1043
                    ok = cholesky();
                    if (!ok) {                // stm.py:1046-1048: scipy cholesky (UPPER) of make_pd(H)+1e-5 I
                        make_pd();
                        const double keep = diagA;
                        if (isn) diagA += 1e-5;
                        ok = cholesky();
                        diagA = keep;
                        upper = true;
                        if (!ok) fail = true;
                    }
                }
            } else dump(P.hess_out, false);
        } else dump(P.hess_out, false);
        if (P.pd_path) P.pd_path[doc] = path;
        if (fail) {
            atomicMax(P.err_flag, 3 /* STM_ERR_LINALG */);
            return;
        }
        if (P.chol_out) {
            double *o = P.chol_out + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j) {
                    double val = (j == lane) ? Ldiag : (j < lane ? M[(size_t)lane * MLD + j] : 0.0);
                    if (upper) o[(size_t)j * n + lane] = val;  // the reference holds the upper factor here
                    else o[(size_t)lane * n + j] = val;
                }
        }

        // ---- bound (stm.py:1068-1101)
        const double det = wave_sum(isn ? log(Ldiag) : 0.0);
        double q = 0.0;
        {
            const double d = eta_i - mu_i;
            if (P.siginv_diag) {
                if (isn) q = (d * S[(size_t)lane * n + lane]) * d;
            } else {
                if (isn) sdv[lane] = d;
                __syncthreads();
                if (isn) {
                    double t = 0.0;
                    for (int j = 0; j < n; ++j) t += sdv[j] * S[(size_t)j * n + lane];
                    q = t * d;
                }
            }
        }
        q = wave_sum(q);
        P.bound[doc] = ll + (-det) - 0.5 * q - P.sigmaentropy;  // uniform store

        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066), accumulated into sigma_ss
        const double Rdiag = 1.0 / Ldiag;
        if (P.debug_flags & 4) return;
        if (!upper) {
            // R = U^{-1}, U = L^T: column c in lane c, rows from the bottom up; R overwrites the upper triangle
            for (int i = n - 2; i >= 0; --i) {
                double t = 0.0;
                if (isn && lane > i) {
                    for (int l = i + 1; l < n; ++l) {
                        const double rlc = (l == lane) ? Rdiag : (l < lane ? M[(size_t)l * MLD + lane] : 0.0);
                        t -= M[(size_t)l * MLD + i] * rlc;
                    }
                }
                const double lii = lane_bcast(Ldiag, i);
                __syncthreads();
                if (isn && lane > i) M[(size_t)i * MLD + lane] = t / lii;
                __syncthreads();
            }
        }
        for (int i = 0; i < n; ++i) {
            double t = 0.0;
            const double rii = lane_bcast(Rdiag, i);
            if (isn) {
                if (upper) t = (lane == i) ? Rdiag * Rdiag : 0.0;
                else {
                    const int l0 = i > lane ? i : lane;
                    for (int l = l0; l < n; ++l) {
                        const double ril = (l == i) ? rii : M[(size_t)i * MLD + l];
                        const double rjl = (l == lane) ? Rdiag : M[(size_t)lane * MLD + l];
                        t += ril * rjl;
                    }
                }
                unsafeAtomicAdd(sig_acc + (size_t)i * n + lane, t);  // sigma_ss += nu, stm.py:582
                if (P.nu_out) P.nu_out[(size_t)doc * n * n + (size_t)i * n + lane] = t;
            }
        }
        __syncthreads();
    }
}

// out = sum over the replicated / per-block partial copies (fixed order)
__global__ void reduce_sigma_kernel(const double *part, int nblocks, int nn, double *out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nn) return;
    double t = 0.0;
    for (int b = 0; b < nblocks; ++b) t += part[(size_t)b * nn + q];
    out[q] = t;
}

// bound = np.sum(calculated_bounds) (stm.py:592): one block, fixed tree => deterministic
__global__ __launch_bounds__(1024) void reduce_bound_kernel(const double *bound, int64_t N, double *out) {
    __shared__ double sh[1024];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 1024) t += bound[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

}  // namespace stm
