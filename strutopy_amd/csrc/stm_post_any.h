// stm_post_any.h -- the post-solve step for ANY number of topics (the product path for K > 112; STM_POST_ANY=1 selects it for every K
// as a second implementation the tests hold against the matrix-core kernels).
//
// Same arithmetic as the other post kernels (reference src/modules/stm.py:547-588: theta, hessian + make_pd ladder,
// decompose_hessian, lower_bound, optimize_nu, update_z, the sigma_ss / beta_ss accumulation), written for generality instead of
// speed: one workgroup of 256 threads per document (persistent over a strided set of documents), every matrix in a
// per-workgroup HBM scratch -- A (the Hessian with its fixes, n x n), L (n x n; A's storage takes R = inv(L^T) afterwards), b
// (N_d x K, word-major) and sqrt(c) -- plain loops in the order the reference's expressions imply (sums over words and over
// topics ascending), a left-looking column Cholesky (two barriers per column), back substitution with one thread per column.
// WM (K <= 128): phi leaves as the scalar r_dw per (document, word) for the word-major beta_ss pass (stm_betass.h) and nu is summed into
// the workgroup's own n x n slab by plain read-modify-write -- no atomics, run-to-run identical like the matrix-core kernels.  Beyond 128
// topics (the pass holds two topics per lane) phi goes to beta_ss and nu to replicated accumulators with fp64 atomics: two runs agree to
// rounding, not bit for bit.  The reference takes any K (stm.py:311-329); K <= 112 has the matrix-core kernels.
#pragma once
#include "stm_post_common.h"

namespace stm {

constexpr int ANY_BS = 256;

// doubles of HBM scratch per workgroup: A | L | b | sqrt(c)
__host__ __device__ inline size_t post_any_scratch(int K, int nd_max) {
    const size_t n = (size_t)(K - 1);
    return 2 * n * n + (size_t)nd_max * (size_t)(K + 1) + 8;
}
// doubles of dynamic LDS: exp(eta~) | stable_softmax(eta~) | eta - mu | rowsum(c') | the reduction tree | scalars
__host__ __device__ inline size_t post_any_lds_doubles(int K) { return 4 * (size_t)K + ANY_BS + 8; }

template <bool WM>
__global__ __launch_bounds__(ANY_BS) void post_any_kernel(PostParams P) {
    extern __shared__ __attribute__((aligned(16))) double any_lds[];
    const int tid = threadIdx.x;
    const int K = P.K, n = P.n;
    double *ex = any_lds, *ths = ex + K, *dvec = ths + K, *rowc = dvec + K, *red = rowc + K, *sc = red + ANY_BS;
    const size_t per_wg = post_any_scratch(K, P.nd_max);
    double *A = P.a_scratch + (size_t)blockIdx.x * per_wg, *L = A + (size_t)n * n, *bm = L + (size_t)n * n;
    double *sqv = bm + (size_t)P.nd_max * K;
    const double *S = P.siginv;
    double *sig_acc = P.sigma_part + (size_t)(WM ? blockIdx.x : blockIdx.x % P.nrep) * (size_t)n * n;
    auto nu_add = [&](double *cell, double t) __attribute__((always_inline)) {   // (WM: the slab is this workgroup's, a cell one thread's)
        if constexpr (WM) *cell += t; else unsafeAtomicAdd(cell, t);
    };

    // fixed-order tree over the workgroup (every thread gets the total)
    auto block_sum = [&](double v) -> double {
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int s = ANY_BS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        return red[0];
    };
    auto block_nanmax = [&](double v) -> double {
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        for (int s = ANY_BS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = nanmax(red[tid], red[tid + s]);
            __syncthreads();
        }
        return red[0];
    };
    auto block_any = [&](bool b) -> bool { return block_sum(b ? 1.0 : 0.0) != 0.0; };

    for (int64_t tk = blockIdx.x; tk < P.count; tk += gridDim.x) {
        __syncthreads();
        const int64_t ticket = P.first + tk;
        const int64_t doc = P.order ? (int64_t)P.order[ticket] : ticket;
        const int64_t p0 = P.indptr[doc];
        const int Nd = (int)(P.indptr[doc + 1] - p0);
        const int asp = P.aspect ? P.aspect[doc] : 0;
        const double *bT = P.betaT + (size_t)asp * (size_t)P.V * K;
        double *bssT = P.beta_ssT + (size_t)asp * (size_t)P.V * K;
        const bool dump_phi = P.phi_out && doc == P.phi_doc;

        // ---- eta~ = [eta, 0]; theta (unshifted softmax, stm.py:547-549); stable softmax (stm.py:905-909); exp(eta~)
        double se = 0.0, mx = -INFINITY;
        for (int k = tid; k < K; k += ANY_BS) {
            const double e = k < n ? P.eta[doc * n + k] : 0.0;
            const double x = exp(e);
            ex[k] = x;
            dvec[k] = k < n ? e - P.mu[doc * n + k] : 0.0;
            se += x;
            mx = nanmax(mx, e);
        }
        const double sumex = block_sum(se);
        const double m = block_nanmax(mx);
        double ss = 0.0;
        for (int k = tid; k < K; k += ANY_BS) {
            P.theta[doc * K + k] = ex[k] / sumex;
            const double e = k < n ? P.eta[doc * n + k] : 0.0;
            const double s = exp(e - m);
            ths[k] = s;
            ss += s;
        }
        const double ssum = block_sum(ss);
        for (int k = tid; k < K; k += ANY_BS) ths[k] = ths[k] / ssum;
        __syncthreads();

        // ---- one thread per word: column sum, theta @ a (stm.py:1088-1095), b (stm.py:1001), phi (stm.py:1103-1118)
        double csum = 0.0, ll = 0.0;
        bool bad = false;
        for (int v = tid; v < Nd; v += ANY_BS) {
            const int idx = P.indices[p0 + v];
            const double c = P.counts[p0 + v];
            const double *row = bT + (size_t)idx * K;
            double Sw = 0.0, Lw = 0.0;
            for (int k = 0; k < K; ++k) {
                const double a = row[k] * ex[k];
                Sw += a;
                Lw += ths[k] * a;
            }
            const double sq = sqrt(c), w = sq / Sw;
            if constexpr (WM) P.rw[P.wm_slot[p0 + v]] = (w * sq) * sumex;     // r_dw (stm_betass.h): phi = beta theta r
            ll += log(Lw) * c;
            csum += c;
            sqv[v] = sq;
            bad |= !(Sw > 0.0 && Sw < INFINITY);
            double *bv = bm + (size_t)v * K;
            for (int k = 0; k < K; ++k) {
                const double a = row[k] * ex[k];
                bv[k] = a * sq / Sw;                                  // hessian's b
                const double ph = a * w * sq;                        // update_z's phi
                if (!WM && !(P.debug_flags & 1)) unsafeAtomicAdd(bssT + (size_t)idx * K + k, ph);
                if (dump_phi) P.phi_out[(size_t)k * Nd + v] = ph;
            }
        }
        const double Ndoc = (double)(long long)block_sum(csum);       // int(np.sum(word_count)), stm.py:1002
        ll = block_sum(ll);
        // rowsum(c'), stm.py:1002,1011: lane = topic, words ascending
        bool neg = false;
        for (int k = tid; k < K; k += ANY_BS) {
            double t = 0.0;
            for (int v = 0; v < Nd; ++v) t += bm[(size_t)v * K + k] * sqv[v];
            rowc[k] = t;
            neg |= !(t >= 0.0);
        }
        if (block_any(bad || neg)) { if (tid == 0) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */); }   // stm.py:1117

        // ---- H = b b^T - N theta theta^T - diag(rowsum) + N diag(theta) + siginv (stm.py:1001-1013), both triangles
        for (int q = tid; q < n * n; q += ANY_BS) {
            const int i = q / n, j = q - i * n;
            if (i > j) continue;
            double t = 0.0;
            for (int v = 0; v < Nd; ++v) t += bm[(size_t)v * K + i] * bm[(size_t)v * K + j];
            double h = t - Ndoc * (ths[i] * ths[j]);
            if (i == j) h = h - rowc[i] + Ndoc * ths[i];
            h += S[(size_t)i * n + j];
            A[(size_t)i * n + j] = h;
            A[(size_t)j * n + i] = h;
        }
        __syncthreads();

        // np.linalg.cholesky(A) -> L (lower, zeros above), left-looking by columns: a thread owns rows tid, tid + 256, ...
        auto cholesky = [&]() -> bool {
            for (int q = tid; q < n * n; q += ANY_BS) L[q] = 0.0;
            __syncthreads();
            for (int j = 0; j < n; ++j) {
                const double *lj = L + (size_t)j * n;
                for (int i = j + tid; i < n; i += ANY_BS) {
                    double t = A[(size_t)i * n + j];
                    const double *li = L + (size_t)i * n;
                    for (int l = 0; l < j; ++l) t -= li[l] * lj[l];
                    L[(size_t)i * n + j] = t;                      // (unscaled until the pivot is known)
                    if (i == j) {                                  // tid == 0
                        const bool ok = t > PIVOT_TOL * A[(size_t)j * n + j];   // see PIVOT_TOL
                        sc[0] = ok ? sqrt(t) : 0.0;
                        sc[1] = ok ? 1.0 : 0.0;
                    }
                }
                __syncthreads();
                const double ljj = sc[0];
                if (sc[1] == 0.0) { __syncthreads(); return false; }
                for (int i = j + tid; i < n; i += ANY_BS) L[(size_t)i * n + j] = (i == j) ? ljj : L[(size_t)i * n + j] / ljj;
                __syncthreads();
            }
            return true;
        };
        auto make_pd = [&]() {   // stm.py:964-984
            for (int i = tid; i < n; i += ANY_BS) {
                const double dv = A[(size_t)i * n + i];
                double mag = 0.0;
                for (int j = 0; j < n; ++j) mag += fabs(A[(size_t)i * n + j]);
                mag -= fabs(dv);
                if (dv < mag) A[(size_t)i * n + i] = mag;
            }
            __syncthreads();
        };
        auto add_eps = [&]() {
            for (int i = tid; i < n; i += ANY_BS) A[(size_t)i * n + i] += 1e-5;
            __syncthreads();
        };
        auto dump_hess = [&]() {
            double *o = P.hess_out + (size_t)doc * n * n;
            for (int q = tid; q < n * n; q += ANY_BS) o[q] = A[q];
        };
        // the reference's PD ladder, one Cholesky site (see post_big2_kernel): 0 hessian()'s PD test, 1 after make_pd, 2 + 1e-5 and
        // decompose_hessian's np.linalg.cholesky, 3 after make_pd, 4 scipy's UPPER factor of make_pd(H) + 1e-5 I
        int path = 0;
        bool upper = false, fail = false;
        for (int attempt = 0;; ++attempt) {
            if (attempt == 1) { make_pd(); path = 1; }
            else if (attempt == 2) { add_eps(); path = 2; }
            else if (attempt == 3) make_pd();
            else if (attempt == 4) {
                make_pd();
                // (the + 1e-5 goes to a copy in the reference; A is not read again here, and L's diagonal is what the bound uses)
                add_eps();
            }
            if (P.hess_out && attempt <= 2) dump_hess();
            const bool ok = cholesky();
            if (attempt == 4) { upper = true; fail = !ok; break; }
            if (ok) break;
        }
        if (P.pd_path && tid == 0) P.pd_path[doc] = path;
        if (fail) {
            if (tid == 0) atomicMax(P.err_flag, 3 /* STM_ERR_LINALG */);
            continue;
        }
        if (P.chol_out) {
            double *o = P.chol_out + (size_t)doc * n * n;
            for (int q = tid; q < n * n; q += ANY_BS) {
                const int i = q / n, j = q - i * n;
                o[q] = upper ? L[(size_t)j * n + i] : L[q];     // the reference holds the upper factor on the last rung
            }
        }

        // ---- bound (stm.py:1068-1101)
        {
            double dt = 0.0, qd = 0.0;
            for (int i = tid; i < n; i += ANY_BS) {
                dt += log(L[(size_t)i * n + i]);
                double t = 0.0;
                for (int j = 0; j < n; ++j) t += dvec[j] * S[(size_t)j * n + i];
                qd += t * dvec[i];
            }
            const double det = block_sum(dt), quad = block_sum(qd);
            if (tid == 0) P.bound[doc] = ll + (-det) - 0.5 * quad - P.sigmaentropy;
        }

        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066): R = inv(L^T), upper, into A's storage; one thread per column
        double *R = A;
        if (!(P.debug_flags & 4)) {
            double *nu_doc = P.nu_out ? P.nu_out + (size_t)doc * n * n : nullptr;
            if (upper) {   // triu(U^T) = diag(U): nu = diag(1 / L_ii^2)
                for (int i = tid; i < n; i += ANY_BS) {
                    const double r = 1.0 / L[(size_t)i * n + i];
                    nu_add(sig_acc + (size_t)i * n + i, r * r);
                    if (nu_doc)
                        for (int j = 0; j < n; ++j) nu_doc[(size_t)i * n + j] = (j == i) ? r * r : 0.0;
                }
            } else {
                __syncthreads();
                for (int c = tid; c < n; c += ANY_BS) {
                    for (int i = c; i >= 0; --i) {
                        double t = (i == c) ? 1.0 : 0.0;
                        for (int l = i + 1; l <= c; ++l) t -= L[(size_t)l * n + i] * R[(size_t)l * n + c];
                        R[(size_t)i * n + c] = t / L[(size_t)i * n + i];
                    }
                }
                __syncthreads();
                for (int q = tid; q < n * n; q += ANY_BS) {
                    const int i = q / n, j = q - i * n;
                    if (i > j) continue;
                    double t = 0.0;
                    const double *ri = R + (size_t)i * n, *rj = R + (size_t)j * n;
                    for (int l = j; l < n; ++l) t += ri[l] * rj[l];
                    nu_add(sig_acc + (size_t)i * n + j, t);
                    if (i != j) nu_add(sig_acc + (size_t)j * n + i, t);
                    if (nu_doc) { nu_doc[(size_t)i * n + j] = t; nu_doc[(size_t)j * n + i] = t; }
                }
            }
        }
    }
}

}  // namespace stm
