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

inline size_t post_big_lds_doubles(int n, int MLD) { return (((size_t)n * MLD + 1) & ~(size_t)1) + (size_t)BT * TLD + 4 * BT + 4 * TW; }

__global__ __launch_bounds__(64) void post_big_kernel(PostParams P) {
    extern __shared__ __attribute__((aligned(16))) double big_lds[];
    const int lane = threadIdx.x;
    const int K = P.K, n = P.n, MLD = P.MLD;
    double *M = big_lds;                        // [n][MLD]: H, then A (upper) / L (lower) / R (upper)
    double *T = M + (((size_t)n * MLD + 1) & ~(size_t)1);   // [BT][TLD] word tile, topic-major (16-byte aligned rows)
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
        long long tp[8];
        tp[0] = P.prof ? (long long)__builtin_readcyclecounter() : 0;

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

        if (P.prof) tp[1] = (long long)__builtin_readcyclecounter();
        double csum = 0.0, ll = 0.0, rowc[2] = {0.0, 0.0};
        bool bad = false;
        const int kc = (K + 3) >> 2;  // topics per quarter in the per-word sums (<= 32)
        long long tq[5] = {0, 0, 0, 0, 0};
        for (int t0 = 0; t0 < Nd; t0 += TW) {
            const int nw = Nd - t0 < TW ? Nd - t0 : TW;
            const int my_idx = (lane < nw) ? P.indices[p0 + t0 + lane] : 0;
            const double my_c = (lane < nw) ? P.counts[p0 + t0 + lane] : 0.0;
            long long c0 = P.prof ? (long long)__builtin_readcyclecounter() : 0;
            // -- 1. gather: coalesced beta rows, transposed into T[topic][word]
            {   // all 32 loads first (words beyond the document carry id 0, lanes beyond K read topic 0), then the stores
                double gv[2][TW];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < TW; ++j) {
                        const int idx = __builtin_amdgcn_readlane(my_idx, j);
                        gv[r][j] = bT[(size_t)idx * K + (lane + WAVE * r < K ? lane + WAVE * r : 0)];
                    }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = lane + WAVE * r;
                    double2 *row = reinterpret_cast<double2 *>(T + (size_t)k * TLD);
#pragma unroll
                    for (int j = 0; j < TW; j += 2)
                        row[j >> 1] = make_double2((k < K && j < nw) ? gv[r][j] : 0.0, (k < K && j + 1 < nw) ? gv[r][j + 1] : 0.0);
                }
            }
            __syncthreads();
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[0] += c1 - c0; c0 = c1; }
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
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[1] += c1 - c0; c0 = c1; }
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
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[2] += c1 - c0; c0 = c1; }
            // -- 4. H += b b^T restricted to the tile, block by block on the matrix cores (upper block triangle; there are no
            // registers for 28-36 accumulator tiles, so every 16 x 16 product goes straight into the LDS matrix, mirrored)
            {
                const int nblk = (n + 15) >> 4;
#pragma unroll 1
                for (int bi = 0; bi < nblk; ++bi) {
                    const int ra = bi * 16 + fr;                 // rows of T beyond n are topic K-1 / zeros: masked at the store
                    double fa[TW / 4];
#pragma unroll
                    for (int sk = 0; sk < TW / 4; ++sk) fa[sk] = T[(size_t)ra * TLD + sk * 4 + fq];
#pragma unroll 1
                    for (int bj = bi; bj < nblk; bj += 2) {      // two blocks at a time: their MFMA chains interleave, and the
                        const bool two = bj + 1 < nblk;          // old values of M are fetched while the products are formed
                        const int rb0 = bj * 16 + fr, rb1 = (two ? bj + 1 : bj) * 16 + fr;
                        double fb0[TW / 4], fb1[TW / 4], m0[4], m1[4], t0[4], t1[4];
#pragma unroll
                        for (int sk = 0; sk < TW / 4; ++sk) { fb0[sk] = T[(size_t)rb0 * TLD + sk * 4 + fq]; fb1[sk] = T[(size_t)rb1 * TLD + sk * 4 + fq]; }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = bi * 16 + fq + 4 * r, ic = i < n ? i : n - 1;
                            const int j0 = rb0 < n ? rb0 : n - 1, j1 = rb1 < n ? rb1 : n - 1;
                            m0[r] = M[(size_t)ic * MLD + j0]; m1[r] = M[(size_t)ic * MLD + j1];
                            t0[r] = M[(size_t)j0 * MLD + ic]; t1[r] = M[(size_t)j1 * MLD + ic];
                        }
                        v4d a0 = (v4d){0.0, 0.0, 0.0, 0.0}, a1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int sk = 0; sk < TW / 4; ++sk) {
                            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[sk], fb0[sk], a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[sk], fb1[sk], a1, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = bi * 16 + fq + 4 * r;
                            if (i < n && rb0 < n) {
                                M[(size_t)i * MLD + rb0] = m0[r] + a0[r];
                                if (bi != bj) M[(size_t)rb0 * MLD + i] = t0[r] + a0[r];
                            }
                            if (two && i < n && rb1 < n) {
                                M[(size_t)i * MLD + rb1] = m1[r] + a1[r];
                                M[(size_t)rb1 * MLD + i] = t1[r] + a1[r];
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[3] += c1 - c0; c0 = c1; }
        }
        if (P.prof && lane == 0) for (int q = 0; q < 4; ++q) P.prof[doc * 40 + 24 + q] = tq[q];
        if (P.prof) tp[2] = (long long)__builtin_readcyclecounter();
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

        if (P.prof) tp[3] = (long long)__builtin_readcyclecounter();
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
                {   // both row sets in one loop: the broadcast row j feeds two FMAs, twelve LDS reads in flight
                    const double *r0 = M + (size_t)(k0 < n ? k0 : n - 1) * MLD, *r1 = M + (size_t)(k1 < n ? k1 : n - 1) * MLD;
                    const double *rj = M + (size_t)j * MLD;
                    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
                    int l = 0;
                    for (; l + 3 < j; l += 4) {
                        double x[4], y[4], pj[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { x[q] = r0[l + q]; y[q] = r1[l + q]; pj[q] = rj[l + q]; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) { a[q] = fma(x[q], pj[q], a[q]); b[q] = fma(y[q], pj[q], b[q]); }
                    }
                    for (; l < j; ++l) { const double pj = rj[l]; a[0] = fma(r0[l], pj, a[0]); b[0] = fma(r1[l], pj, b[0]); }
                    if (k0 < n && k0 >= j) t[0] = ((k0 == j) ? diagA[0] : rj[k0]) - ((a[0] + a[1]) + (a[2] + a[3]));
                    if (k1 < n && k1 >= j) t[1] = ((k1 == j) ? diagA[1] : rj[k1]) - ((b[0] + b[1]) + (b[2] + b[3]));
                }
                const double d = lane_bcast((j >> 6) ? t[1] : t[0], j & 63);
                if (!(d > PIVOT_TOL * lane_bcast((j >> 6) ? diagA[1] : diagA[0], j & 63))) { ok = false; break; }
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

        if (P.prof) tp[4] = (long long)__builtin_readcyclecounter();
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

        if (P.prof) tp[5] = (long long)__builtin_readcyclecounter();
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
                {   // both columns in one loop over the rows l of R below i (L[l][i] is a broadcast read)
                    const int c0 = k0 < n ? k0 : n - 1, c1 = k1 < n ? k1 : n - 1;
                    const double *ci = M + i, *p0 = M + c0, *p1 = M + c1;
                    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
                    a[0] = M[(size_t)c0 * MLD + i] * Rdiag[0];   // the l == c terms
                    b[0] = M[(size_t)c1 * MLD + i] * Rdiag[1];
                    int l = i + 1;
                    for (; l + 3 < n - 1; l += 4) {
                        double u[4], x[4], y[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            u[q] = ci[(size_t)(l + q) * MLD]; x[q] = p0[(size_t)(l + q) * MLD]; y[q] = p1[(size_t)(l + q) * MLD];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            a[q] = fma(u[q], (l + q < c0) ? x[q] : 0.0, a[q]);
                            b[q] = fma(u[q], (l + q < c1) ? y[q] : 0.0, b[q]);
                        }
                    }
                    for (; l < n - 1; ++l) {
                        const double u = ci[(size_t)l * MLD];
                        a[0] = fma(u, (l < c0) ? p0[(size_t)l * MLD] : 0.0, a[0]);
                        b[0] = fma(u, (l < c1) ? p1[(size_t)l * MLD] : 0.0, b[0]);
                    }
                    if (k0 < n && k0 > i) t[0] = -((a[0] + a[1]) + (a[2] + a[3]));
                    if (k1 < n && k1 > i) t[1] = -((b[0] + b[1]) + (b[2] + b[3]));
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
        if (P.prof) tp[6] = (long long)__builtin_readcyclecounter();
        // nu[i][j] = sum_{l >= max(i,j)} R[i][l] R[j][l], lane = columns j0, j1; sigma_ss += nu (stm.py:582).
        // The diagonal of M is free by now (A's and L's diagonals live in registers): R[x][x] goes there so the
        // loop has no special cases; row i of R is a broadcast read, rows j0 / j1 are the lane's own.
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (lane + WAVE * r < n) M[(size_t)(lane + WAVE * r) * MLD + lane + WAVE * r] = Rdiag[r];
        __syncthreads();
        {
            const int j0 = k0 < n ? k0 : n - 1, j1 = k1 < n ? k1 : n - 1;
            const double *q0 = M + (size_t)j0 * MLD, *q1 = M + (size_t)j1 * MLD;
            for (int i = 0; i < n; ++i) {
                double v0, v1;
                if (upper) {
                    v0 = (k0 == i) ? Rdiag[0] * Rdiag[0] : 0.0;
                    v1 = (k1 == i) ? Rdiag[1] * Rdiag[1] : 0.0;
                } else {
                    const double *qi = M + (size_t)i * MLD;
                    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
                    int l = i;
                    for (; l + 3 < n; l += 4) {
                        double u[4], x[4], y[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { u[q] = qi[l + q]; x[q] = q0[l + q]; y[q] = q1[l + q]; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            a[q] = fma(u[q], (l + q >= j0) ? x[q] : 0.0, a[q]);
                            b[q] = fma(u[q], (l + q >= j1) ? y[q] : 0.0, b[q]);
                        }
                    }
                    for (; l < n; ++l) {
                        const double u = qi[l];
                        a[0] = fma(u, (l >= j0) ? q0[l] : 0.0, a[0]);
                        b[0] = fma(u, (l >= j1) ? q1[l] : 0.0, b[0]);
                    }
                    v0 = (a[0] + a[1]) + (a[2] + a[3]);
                    v1 = (b[0] + b[1]) + (b[2] + b[3]);
                }
                if (k0 < n) {
                    unsafeAtomicAdd(sig_acc + (size_t)i * n + k0, v0);
                    if (P.nu_out) P.nu_out[(size_t)doc * n * n + (size_t)i * n + k0] = v0;
                }
                if (k1 < n) {
                    unsafeAtomicAdd(sig_acc + (size_t)i * n + k1, v1);
                    if (P.nu_out) P.nu_out[(size_t)doc * n * n + (size_t)i * n + k1] = v1;
                }
            }
        }
        if (P.prof && lane == 0) {
            tp[7] = (long long)__builtin_readcyclecounter();
            for (int q = 0; q < 7; ++q) P.prof[doc * 40 + 32 + q] = tp[q + 1] - tp[q];
        }
        __syncthreads();
    }
}

}  // namespace stm
