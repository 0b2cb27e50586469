// stm_post_big.h -- the post-solve step (stm_post.h) for 64 < K <= 128 topics.
//
// Same arithmetic, in the same order, as post_kernel (reference src/modules/stm.py:547-588: theta,
// hessian + make_pd ladder, decompose_hessian, lower_bound, optimize_nu, update_z, accumulation), but
// a lane owns TWO topics / matrix rows (lane and lane + 64) and everything runs on the VALU: the
// (K-1)^2 matrix (up to 127 x 127 doubles = 129 KB) takes most of the LDS, so there is one workgroup
// per CU and no room for the MFMA accumulator sets of the K <= 64 kernel.  This is the coverage path
// for BASELINE config 4 (K = 100); it is not tuned.
#pragma once
#include "stm_post.h"

namespace stm {

constexpr int BT = 128;   // topics padded to 128

inline size_t post_big_lds_doubles(int n, int MLD) { return (size_t)n * MLD + (size_t)BT * TLD + 4 * BT + 4 * TW; }

__global__ __launch_bounds__(64) void post_big_kernel(PostParams P) {
    extern __shared__ __attribute__((aligned(16))) double big_lds[];
    const int lane = threadIdx.x;
    const int K = P.K, n = P.n, MLD = P.MLD;
    double *M = big_lds;                        // [n][MLD]: H, then A (upper) / L (lower) / R (upper)
    double *T = M + (size_t)n * MLD;            // [BT][TLD] word tile, topic-major
    double *sex = T + (size_t)BT * TLD;         // exp(eta~)
    double *sth = sex + BT;                     // stable_softmax(eta~)
    double *sdv = sth + BT;                     // eta - mu (dense siginv only)
    double *srd = sdv + BT;                     // 1 / diag(L)
    double *wpar = srd + BT;                    // per word of the tile: { sqrt(c), S, 1/S, sqrt(c)/S }
    const double *S = P.siginv;
    double *sig_acc = P.sigma_part + (size_t)(blockIdx.x % P.nrep) * (size_t)n * n;
    const int fr = lane & 15, fq = lane >> 4;
    const int k0 = lane, k1 = lane + WAVE;      // this lane's two topics / rows

    for (int64_t tk = blockIdx.x; tk < P.count; tk += gridDim.x) {
        const int64_t ticket = P.first + tk;
        const int64_t doc = P.order ? (int64_t)P.order[ticket] : ticket;
        const int64_t p0 = P.indptr[doc];
        const int Nd = (int)(P.indptr[doc + 1] - p0);
        const int asp = P.aspect ? P.aspect[doc] : 0;
        const double *bT = P.betaT + (size_t)asp * (size_t)P.V * K;
        double *bssT = P.beta_ssT + (size_t)asp * (size_t)P.V * K;
        const bool dump_phi = P.phi_out && doc == P.phi_doc;

        // ---- eta~, theta (unshifted softmax, stm.py:547-549), stable softmax, exp(eta~)
        double etav[2], muv[2], exv[2], thsv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = lane + WAVE * r;
            etav[r] = (k < n) ? P.eta[doc * n + k] : 0.0;   // topic K-1 holds the appended 0
            muv[r] = (k < n) ? P.mu[doc * n + k] : 0.0;
            exv[r] = (k < K) ? exp(etav[r]) : 0.0;
        }
        const double sumex = wave_sum(exv[0] + exv[1]);
        double mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = lane + WAVE * r;
            if (k < K) { P.theta[doc * K + k] = exv[r] / sumex; mloc = nanmax(mloc, etav[r]); }
        }
        const double m = wave_nanmax(mloc);
        double esv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) esv[r] = (lane + WAVE * r < K) ? exp(etav[r] - m) : 0.0;
        const double ssum = wave_sum(esv[0] + esv[1]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = lane + WAVE * r;
            thsv[r] = esv[r] / ssum;
            sex[k] = exv[r];
            sth[k] = (k < K) ? thsv[r] : 0.0;
        }
        for (int q = lane; q < n * MLD; q += WAVE) M[q] = 0.0;
        for (int q = lane; q < BT * TLD; q += WAVE) T[q] = 0.0;
        __syncthreads();

        double csum = 0.0, ll = 0.0, rowc[2] = {0.0, 0.0};
        bool bad = false;
        const int kc = (K + 3) >> 2;  // topics per quarter in the per-word sums (<= 32)
        for (int t0 = 0; t0 < Nd; t0 += TW) {
            const int nw = Nd - t0 < TW ? Nd - t0 : TW;
            const int my_idx = (lane < nw) ? P.indices[p0 + t0 + lane] : 0;
            const double my_c = (lane < nw) ? P.counts[p0 + t0 + lane] : 0.0;
            // -- 1. gather: coalesced beta rows, transposed into T[topic][word]
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + WAVE * r;
                for (int j = 0; j < TW; ++j) {
                    const int idx = __builtin_amdgcn_readlane(my_idx, j);
                    T[(size_t)k * TLD + j] = (k < K && j < nw) ? bT[(size_t)idx * K + k] : 0.0;
                }
            }
            __syncthreads();
            // -- 2. per-word sums, lane = (word fr, topic quarter fq)
            {
                double Sp = 0.0, Lp = 0.0;
                const int kb = fq * kc;
                for (int kk = 0; kk < kc; ++kk) {
                    const int k = kb + kk;
                    const double a = T[(size_t)k * TLD + fr] * sex[k];
                    Sp += a;              // np.sum(a, 0)
                    Lp += sth[k] * a;     // theta @ (beta * exp(eta~)), stm.py:1088-1094
                }
                Sp += __shfl_xor(Sp, 16); Sp += __shfl_xor(Sp, 32);
                Lp += __shfl_xor(Lp, 16); Lp += __shfl_xor(Lp, 32);
                if (lane < nw) {
                    const double c = my_c, sq = sqrt(c);
                    ll += log_pos(Lp) * c;
                    csum += c;
                    double *wp = wpar + 4 * lane;
                    wp[0] = sq; wp[1] = Sp; wp[2] = 1.0 / Sp; wp[3] = sq / Sp;   // update_z: sqrt(c) / colsum, stm.py:1115
                }
            }
            __syncthreads();
            // -- 3. scatter phi, rowsum(c'), T <- b (lane = topic)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + WAVE * r;
                if (k < K) {
                    double *trow = T + (size_t)k * TLD;
                    for (int j = 0; j < nw; ++j) {
                        const int idx = __builtin_amdgcn_readlane(my_idx, j);
                        const double sq = wpar[4 * j], Sj = wpar[4 * j + 1], rj = wpar[4 * j + 2], wj = wpar[4 * j + 3];
                        const double a = trow[j] * exv[r];
                        const double num = a * sq;            // b = a*sqrt(c)/S (stm.py:1001)
                        const double q0 = num * rj;
                        const double b = fma(fma(-q0, Sj, num), rj, q0);
                        const double phi = a * wj * sq;       // stm.py:1115-1116
                        bad |= !(phi >= 0.0);
                        rowc[r] += b * sq;                    // rowsum(c'), stm.py:1002,1011
                        trow[j] = b;
                        unsafeAtomicAdd(bssT + (size_t)idx * K + k, phi);   // stm.py:588
                        if (dump_phi) P.phi_out[(size_t)k * Nd + t0 + j] = phi;
                    }
                }
            }
            __syncthreads();
            // -- 4. H[:, j] += b b^T restricted to the tile, lane = column j (two columns per lane)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int j = lane + WAVE * r;
                if (j < n) {
                    double own[TW];
#pragma unroll
                    for (int w = 0; w < TW; ++w) own[w] = T[(size_t)j * TLD + w];
                    for (int i = 0; i < n; ++i) {
                        const double2 *ti = reinterpret_cast<const double2 *>(T + (size_t)i * TLD);
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int w = 0; w < TW; w += 2) {
                            const double2 v = ti[w >> 1];
                            s0 = fma(v.x, own[w], s0);
                            s1 = fma(v.y, own[w + 1], s1);
                        }
                        M[(size_t)i * MLD + j] += s0 + s1;
                    }
                }
            }
            __syncthreads();
        }
        if (wave_any(bad)) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */);
        const double Ndoc = (double)(long long)wave_sum(csum);
        ll = wave_sum(ll);

        // ---- H = b b^T - N theta theta^T, diag += -rowsum(c') + N theta, [:-1,:-1] + siginv (lane = row)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = lane + WAVE * r;
            if (i < n) {
                double *mi = M + (size_t)i * MLD;
                const double thi = sth[i];
                for (int j = 0; j < n; ++j) {
                    double h = mi[j] - Ndoc * (thi * sth[j]);
                    if (j == i) h = h - rowc[r] + Ndoc * thi;
                    const double sij = (P.siginv_diag && j != i) ? 0.0 : S[(size_t)i * n + j];
                    mi[j] = h + sij;
                }
            }
        }
        __syncthreads();

        // ---- PD ladder around one Cholesky (the upper triangle keeps A, L goes to the strict lower triangle)
        double diagA[2], Ldiag[2] = {1.0, 1.0};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = lane + WAVE * r;
            diagA[r] = (i < n) ? M[(size_t)i * MLD + i] : 1.0;
        }
        auto make_pd = [&]() __attribute__((always_inline)) {  // stm.py:964-984
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n) {
                    double mag = 0.0;
                    for (int j = 0; j < n; ++j)
                        mag += (j == i) ? 0.0 : fabs(j > i ? M[(size_t)i * MLD + j] : M[(size_t)j * MLD + i]);
                    if (diagA[r] < mag) diagA[r] = mag;
                }
            }
        };
        auto dump = [&](double *base) __attribute__((always_inline)) {
            if (!base) return;
            double *o = base + (size_t)doc * n * n;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n)
                    for (int j = 0; j < n; ++j)
                        o[(size_t)i * n + j] = (j == i) ? diagA[r] : (j > i ? M[(size_t)i * MLD + j] : M[(size_t)j * MLD + i]);
            }
        };
        int path = 0;
        bool upper = false, fail = false;
        double keep[2] = {0.0, 0.0};
        for (int attempt = 0;; ++attempt) {
            if (attempt == 2) dump(P.hess_out);
            bool ok = true;
            for (int j = 0; j < n; ++j) {
                double t[2] = {0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i = lane + WAVE * r;
                    if (i < n && i >= j) {
                        const double *ri = M + (size_t)i * MLD, *rj = M + (size_t)j * MLD;
                        double a0 = 0.0, a1 = 0.0;
                        int l = 0;
                        for (; l + 1 < j; l += 2) {
                            a0 = fma(ri[l], rj[l], a0);
                            a1 = fma(ri[l + 1], rj[l + 1], a1);
                        }
                        if (l < j) a0 = fma(ri[l], rj[l], a0);
                        t[r] = ((i == j) ? diagA[r] : rj[i]) - (a0 + a1);
                    }
                }
                const double d = lane_bcast((j >> 6) ? t[1] : t[0], j & 63);
                if (!(d > 0.0)) { ok = false; break; }
                const double ljj = sqrt(d), rjj = 1.0 / ljj;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i = lane + WAVE * r;
                    if (i == j) Ldiag[r] = ljj;
                    if (i < n && i > j) M[(size_t)i * MLD + j] = t[r] * rjj;
                }
                __syncthreads();
            }
            __syncthreads();
            if (attempt == 4) { diagA[0] = keep[0]; diagA[1] = keep[1]; upper = true; fail = !ok; break; }
            if (ok) {
                if (attempt < 2) dump(P.hess_out);
                break;
            }
            if (attempt == 0) { make_pd(); path = 1; }
            else if (attempt == 1) { diagA[0] += 1e-5; diagA[1] += 1e-5; path = 2; }
            else if (attempt == 2) { make_pd(); }
            else { make_pd(); keep[0] = diagA[0]; keep[1] = diagA[1]; diagA[0] += 1e-5; diagA[1] += 1e-5; }
        }
        if (P.pd_path) P.pd_path[doc] = path;
        if (fail) {
            atomicMax(P.err_flag, 3 /* STM_ERR_LINALG */);
            continue;
        }
        if (P.chol_out) {
            double *o = P.chol_out + (size_t)doc * n * n;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n)
                    for (int j = 0; j < n; ++j) {
                        const double val = (j == i) ? Ldiag[r] : (j < i ? M[(size_t)i * MLD + j] : 0.0);
                        if (upper) o[(size_t)j * n + i] = val;
                        else o[(size_t)i * n + j] = val;
                    }
            }
        }

        // ---- bound (stm.py:1068-1101)
        double dl = 0.0, q = 0.0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (lane + WAVE * r < n) dl += log(Ldiag[r]);
        const double det = wave_sum(dl);
        if (P.siginv_diag) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n) { const double d = etav[r] - muv[r]; q += (d * S[(size_t)i * n + i]) * d; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (lane + WAVE * r < n) sdv[lane + WAVE * r] = etav[r] - muv[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n) {
                    double t = 0.0;
                    for (int j = 0; j < n; ++j) t += sdv[j] * S[(size_t)j * n + i];
                    q += t * sdv[i];
                }
            }
        }
        q = wave_sum(q);
        P.bound[doc] = ll + (-det) - 0.5 * q - P.sigmaentropy;

        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066): R = L^-T into the upper triangle
        double Rdiag[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            Rdiag[r] = 1.0 / Ldiag[r];
            srd[lane + WAVE * r] = (lane + WAVE * r < n) ? Rdiag[r] : 0.0;
        }
        __syncthreads();
        if (!upper) {
            for (int i = n - 2; i >= 0; --i) {
                double t[2] = {0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int c = lane + WAVE * r;
                    if (c < n && c > i) {
                        double a0 = M[(size_t)c * MLD + i] * Rdiag[r], a1 = 0.0;   // the l == c term
                        int l = i + 1;
                        for (; l + 1 < n - 1; l += 2) {
                            a0 = fma(M[(size_t)l * MLD + i], (l < c) ? M[(size_t)l * MLD + c] : 0.0, a0);
                            a1 = fma(M[(size_t)(l + 1) * MLD + i], (l + 1 < c) ? M[(size_t)(l + 1) * MLD + c] : 0.0, a1);
                        }
                        if (l < n - 1) a0 = fma(M[(size_t)l * MLD + i], (l < c) ? M[(size_t)l * MLD + c] : 0.0, a0);
                        t[r] = -(a0 + a1);
                    }
                }
                const double rii = srd[i];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int c = lane + WAVE * r;
                    if (c < n && c > i) M[(size_t)i * MLD + c] = t[r] * rii;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // nu[i][j] = sum_{l >= max(i,j)} R[i][l] R[j][l], lane = column j; sigma_ss += nu (stm.py:582)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + WAVE * r;
            if (j < n) {
                for (int i = 0; i < n; ++i) {
                    const int l0 = i > j ? i : j;
                    double v = 0.0;
                    if (upper) {
                        v = (i == j) ? Rdiag[r] * Rdiag[r] : 0.0;
                    } else {
                        for (int l = l0; l < n; ++l) {
                            const double ril = (l == i) ? srd[i] : M[(size_t)i * MLD + l];
                            const double rjl = (l == j) ? Rdiag[r] : M[(size_t)j * MLD + l];
                            v = fma(ril, rjl, v);
                        }
                    }
                    unsafeAtomicAdd(sig_acc + (size_t)i * n + j, v);
                    if (P.nu_out) P.nu_out[(size_t)doc * n * n + (size_t)i * n + j] = v;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace stm
