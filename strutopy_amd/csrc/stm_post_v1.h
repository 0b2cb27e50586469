// stm_post_v1.h -- the round-1/2 post kernel (square LDS matrix, register-staged topic-major word tile), kept for A/B
// runs against stm_post.h while the restructured kernel is brought up (STM_POST_IMPL=1 selects it).
#pragma once
#include "stm_post_common.h"

namespace stm {

#ifndef STM_POST_WPE
#define STM_POST_WPE 2   // waves per SIMD the post kernel is register-budgeted for
#endif

// Leading dimension of the LDS matrix: rows start 16-byte aligned and are read two doubles at a time
// (ds_read_b128); MLD = 2 * odd makes the 16-lane groups of such a lane-strided read hit distinct bank quads
// (tools/microbench/lds_read.hip: 1.5x the throughput of ds_read_b64 at an odd stride).
inline int post_v1_mld(int n) {
    int m = n + (n & 1);
    if ((m & 3) != 2) m += 2;
    return m;
}
// entries of the two per-topic LDS vectors: the per-word sums read topics [0, 4 * ceil(K / 4))
__host__ __device__ inline int post_v1_vec_len(int K) { return 4 * ((K + 3) >> 2); }
// doubles of dynamic LDS: region 0 = max(T + per-word pack, M), then two per-topic vectors whose contents
// change with the phase (20.4 KB at K = 50: eight workgroups per CU)
inline size_t post_v1_lds_doubles(int n, int MLD, int K) {
    const size_t t = (size_t)PT * TLD + 4 * TW, m = (size_t)n * MLD;
    return (t > m ? t : m) + 2 * (size_t)post_v1_vec_len(K);
}

// NB 16 x 16 blocks cover the (K-1)^2 matrix on the matrix cores; REM == 1: n = 16 NB + 1 exactly (K = 50:
// 49 = 3 * 16 + 1), and the one row / column beyond the blocks is carried on the VALU (one double per
// lane) instead of padding to NB + 1 blocks -- 6 accumulator tiles instead of 10, twice (b b^T and nu).
template <int NB, int REM, bool DUMP>
__global__ __launch_bounds__(64, STM_POST_WPE) void post_kernel_v1(PostParams P) {
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int R0 = 16 * NB;   // index of the remainder row (REM == 1)
    extern __shared__ __attribute__((aligned(16))) double post_lds[];
    int lane = threadIdx.x;
    const int K = P.K, n = P.n, MLD = P.MLD;
    double *T = post_lds;  // [PT][TLD]
    double *M = post_lds;  // [n][MLD] (after the word loop)
    double *wpar = post_lds + (size_t)PT * TLD;  // word-tile phase only, behind T: per word { sqrt(c), S, 1/S, sqrt(c)/S }
    double *vec = post_lds + ((size_t)PT * TLD + 4 * TW > (size_t)n * MLD ? (size_t)PT * TLD + 4 * TW : (size_t)n * MLD);
    double *sex = vec;            // word tiles: exp(eta~) (unshifted, stm.py:1000,1088,1114) ...
    double *srd = vec;            // ... after the factorisation: 1 / diag(L)
    const int KV = post_v1_vec_len(K);
    double *sth = vec + KV;       // word tiles + assembly: stable_softmax(eta~) (stm.py:998,1083) ...
    double *sdv = vec + KV;       // ... bound: eta - mu broadcast (dense siginv only)
    const double *S = P.siginv;
    double *sig_acc = P.sigma_part + (size_t)(blockIdx.x % P.nrep) * (size_t)n * n;
    bool isn = lane < n, isk = lane < K;
    int fr = lane & 15, fq = lane >> 4;  // MFMA fragment coordinates
    // The lane id is re-read behind an opaque move at the start of every phase: otherwise the lane-dependent LDS
    // addresses of the unrolled tile code are hoisted out of the document loop as invariants and live in scratch memory.
    auto relane = [&]() __attribute__((always_inline)) {
        int l = threadIdx.x;
        asm volatile("" : "+v"(l));
        lane = l; isn = l < n; isk = l < K; fr = l & 15; fq = l >> 4;
    };

    v4d acc_nu[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc_nu[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    double nu_rem = 0.0;   // REM: running sum of nu[lane][R0]

    for (int64_t tk = blockIdx.x; tk < P.count; tk += gridDim.x) {
        relane();
        if (P.debug_flags & 16) {   // nothing may depend on what an earlier document or kernel left in the LDS
            STM_POST_SYNC();
            for (int q = lane; q < P.lds_doubles; q += WAVE) post_lds[q] = __builtin_nan("");
            STM_POST_SYNC();
        }
        const int64_t ticket = P.first + tk;
        // the document header through the scalar cache (uniform, constant for the kernel's lifetime)
        const int64_t doc = P.order ? (int64_t)scalar_load(P.order + ticket) : ticket;
        const int64_t p0 = scalar_load(P.indptr + doc);
        const int Nd = (int)(scalar_load(P.indptr + doc + 1) - p0);
        const int asp = P.aspect ? scalar_load(P.aspect + doc) : 0;
        const double *bT = P.betaT + (size_t)asp * (size_t)P.V * K;
        double *bssT = P.beta_ssT + (size_t)asp * (size_t)P.V * K;
        const bool dump_phi = P.phi_out && doc == P.phi_doc;
        long long tp[8];
        tp[0] = P.prof ? (long long)__builtin_readcyclecounter() : 0;

        // ---- eta~, theta (unshifted softmax, stm.py:547-549), stable softmax, exp(eta~)
        const double eta_i = isn ? P.eta[doc * n + lane] : 0.0;  // lane K-1 holds the appended 0
        const double mu_i = isn ? P.mu[doc * n + lane] : 0.0;
        const double ex = isk ? exp(eta_i) : 0.0;
        const double sumex = wave_sum(ex);
        if (isk) P.theta[doc * K + lane] = ex / sumex;
        const double m = wave_nanmax(isk ? eta_i : -INFINITY);
        const double es = isk ? exp(eta_i - m) : 0.0;
        const double ssum = wave_sum(es);
        const double ths = es / ssum;
        STM_POST_SYNC();  // the previous document's readers of M / vec are done
        if (lane < KV) {
            sex[lane] = ex;
            sth[lane] = isk ? ths : 0.0;
        }
        // topic rows K..63 of T stay zero for the whole document
        for (int q = lane; q < PT * TLD; q += WAVE) T[q] = 0.0;
        STM_POST_SYNC();

        if (P.prof) tp[1] = (long long)__builtin_readcyclecounter();
        relane();
        double csum = 0.0, ll = 0.0, rowc = 0.0;
        bool bad = false;
        v4d acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
        double hrem = 0.0;   // REM: (b b^T)[lane][R0]
        const int kc = (K + 3) >> 2;  // topics per quarter in step 2
        long long tq[4] = {0, 0, 0, 0};

        // Software pipeline over the tiles: while tile t is reduced / scattered / multiplied, the beta rows
        // of tile t+1 are already in flight (into the registers the LDS transpose of tile t has just
        // released) and the word ids / counts of tile t+2 are being fetched.
        auto load_ids = [&](int t0, int &idx, double &c) __attribute__((always_inline)) {
            const bool in = t0 + lane < Nd && lane < TW;
            idx = in ? P.indices[p0 + t0 + lane] : 0;
            c = in ? P.counts[p0 + t0 + lane] : 0.0;
        };
        double g[TW];
        auto load_rows = [&](int t0, int idx_l) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int idx = __builtin_amdgcn_readlane(idx_l, j);
                g[j] = (isk && t0 + j < Nd) ? bT[(size_t)idx * K + lane] : 0.0;
            }
        };
        int my_idx, idx1;
        double my_c, c1;
        load_ids(0, my_idx, my_c);
        load_ids(TW, idx1, c1);
        load_rows(0, my_idx);
        for (int t0 = 0; t0 < Nd; t0 += TW) {
            const int nw = Nd - t0 < TW ? Nd - t0 : TW;
            long long c0 = P.prof ? (long long)__builtin_readcyclecounter() : 0;
            // -- 1. the tile's 16 coalesced rows (issued one tile ago), transposed into T[topic][word]
            if (isk) {
                double2 *row = reinterpret_cast<double2 *>(T + (size_t)lane * TLD);
#pragma unroll
                for (int j = 0; j < TW; j += 2) row[j >> 1] = make_double2(g[j], g[j + 1]);
            }
            int idx2;
            double c2;
            if (t0 + TW < Nd) load_rows(t0 + TW, idx1);
            load_ids(t0 + 2 * TW, idx2, c2);
            STM_POST_SYNC();
            if (P.prof) { const long long c1 = __builtin_readcyclecounter(); tq[0] += c1 - c0; c0 = c1; }
            // -- 2. per-word sums, lane = (word fr, topic quarter fq)
            {
                double Sp = 0.0, Lp = 0.0;
                const int k0 = fq * kc;
#pragma unroll 4
                for (int kk = 0; kk < kc; ++kk) {
                    const int k = k0 + kk;
                    const double a = T[(size_t)k * TLD + fr] * sex[k];
                    Sp += a;              // np.sum(a, 0)
                    Lp += sth[k] * a;     // theta @ (beta * exp(eta~)), stm.py:1088-1094
                }
                Sp += __shfl_xor(Sp, 16); Sp += __shfl_xor(Sp, 32);
                Lp += __shfl_xor(Lp, 16); Lp += __shfl_xor(Lp, 32);
                if (lane < TW) {   // quarter 0 owns the word; words beyond the document get { 0, 0 }
                    double wq = 0.0, sq = 0.0;
                    if (lane < nw) {
                        const double c = my_c;
                        sq = sqrt(c);
                        ll += log_pos(Lp) * c;
                        csum += c;
                        wq = sq / Sp;     // sqrt(c) / colsum: update_z, stm.py:1115, and the factor of b, stm.py:1001
                    }
                    *reinterpret_cast<double2 *>(wpar + 2 * lane) = make_double2(wq, sq);
                }
            }
            STM_POST_SYNC();
            if (P.prof) { const long long c1 = __builtin_readcyclecounter(); tq[1] += c1 - c0; c0 = c1; }
            // -- 3. scatter phi, rowsum(c'), T <- b (lane = topic)
            if (isk) {
                double *trow = T + (size_t)lane * TLD;
                // four words per round, their LDS traffic in 16-byte pieces; columns beyond the document hold
                // zeros and get zeros back.  b = a * (sqrt(c) / S) serves both the Hessian (stm.py:1001, which
                // divides a * sqrt(c) by S: <= 1.5 ulp apart) and phi = b * sqrt(c) (stm.py:1115-1116, this order);
                // rowsum(c') of stm.py:1002,1011 is the row sum of that same product.
                auto round4 = [&](int j0, auto fullc) __attribute__((always_inline)) {
                    double2 *tp2 = reinterpret_cast<double2 *>(trow + j0);
                    const double2 ta = tp2[0], tb = tp2[1];
                    const double2 *wp2 = reinterpret_cast<const double2 *>(wpar + 2 * j0);
                    const double2 w0 = wp2[0], w1 = wp2[1], w2 = wp2[2], w3 = wp2[3];   // { sqrt(c) / S, sqrt(c) }
                    const double b0 = (ta.x * ex) * w0.x, b1 = (ta.y * ex) * w1.x;
                    const double b2 = (tb.x * ex) * w2.x, b3 = (tb.y * ex) * w3.x;
                    const double ph[4] = {b0 * w0.y, b1 * w1.y, b2 * w2.y, b3 * w3.y};
                    tp2[0] = make_double2(b0, b1);
                    tp2[1] = make_double2(b2, b3);
                    rowc += ph[0]; rowc += ph[1]; rowc += ph[2]; rowc += ph[3];
                    bad |= !(ph[0] >= 0.0) | !(ph[1] >= 0.0) | !(ph[2] >= 0.0) | !(ph[3] >= 0.0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        if (decltype(fullc)::value || j < nw) {   // uniform
                            const int idx = __builtin_amdgcn_readlane(my_idx, j);
                            if (!(P.debug_flags & 1)) unsafeAtomicAdd(bssT + (size_t)idx * K + lane, ph[u]);  // stm.py:588
                            if (dump_phi) P.phi_out[(size_t)lane * Nd + t0 + j] = ph[u];
                        }
                    }
                };
                if (nw == TW) {   // a full tile: the four rounds in one straight line, their LDS reads in flight together
#pragma unroll
                    for (int j0 = 0; j0 < TW; j0 += 4) round4(j0, std::true_type{});
                } else {
                    for (int j0 = 0; j0 < nw; j0 += 4) round4(j0, std::false_type{});
                }
            }
            STM_POST_SYNC();
            if (P.prof) { const long long c1 = __builtin_readcyclecounter(); tq[2] += c1 - c0; c0 = c1; }
            // -- 4. b b^T on the matrix cores, upper block triangle
            if (!(P.debug_flags & 2)) {
#pragma unroll
                for (int s = 0; s < TW / 4; ++s) {
                    double f[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) f[b] = T[(size_t)(b * 16 + fr) * TLD + s * 4 + fq];
                    int t = 0;
#pragma unroll
                    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                        for (int bj = bi; bj < NB; ++bj, ++t)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc[t], 0, 0, 0);
                }
                if (REM && isn) {
                    const double2 *own = reinterpret_cast<const double2 *>(T + (size_t)lane * TLD);
                    const double2 *rem = reinterpret_cast<const double2 *>(T + (size_t)R0 * TLD);
                    double h0 = 0.0, h1 = 0.0;
#pragma unroll
                    for (int w = 0; w < TW / 2; ++w) {
                        const double2 a = own[w], b = rem[w];
                        h0 = fma(a.x, b.x, h0);
                        h1 = fma(a.y, b.y, h1);
                    }
                    hrem += h0 + h1;
                }
            }
            STM_POST_SYNC();
            if (P.prof) { const long long c1t = __builtin_readcyclecounter(); tq[3] += c1t - c0; }
            my_idx = idx1; my_c = c1; idx1 = idx2; c1 = c2;
        }
        if (P.prof && lane == 0) for (int q = 0; q < 4; ++q) P.prof[doc * 40 + 24 + q] = tq[q];
        if (P.prof) tp[2] = (long long)__builtin_readcyclecounter();
        relane();
        if (wave_any(bad)) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */);
        const double Ndoc = (double)(long long)wave_sum(csum);
        ll = wave_sum(ll);

        // ---- assemble H = b b^T - N theta theta^T, diag += -rowsum(c') + N theta, [:-1,:-1] + siginv:
        // the MFMA tiles go to LDS raw, then lane i finishes row i (keeps the 4*NT tile elements
        // from being in flight at once)
        {
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NB; ++bj, ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = bi * 16 + fq + 4 * r, j = bj * 16 + fr;
                        if (i < n && j < n) {
                            M[(size_t)i * MLD + j] = acc[t][r];
                            if (bi != bj) M[(size_t)j * MLD + i] = acc[t][r];
                        }
                    }
            if (REM && isn) {
                M[(size_t)lane * MLD + R0] = hrem;
                M[(size_t)R0 * MLD + lane] = hrem;
            }
        }
        STM_POST_SYNC();
        if (isn) {
            double *mi = M + (size_t)lane * MLD;
            const double thi = sth[lane];
            if (P.siginv_diag) {   // what stm.py:501 produces: only the diagonal of siginv is non-zero
                const double sii = S[(size_t)lane * n + lane];
                double2 *mi2 = reinterpret_cast<double2 *>(mi);
                const double2 *th2 = reinterpret_cast<const double2 *>(sth);
                int j = 0;
#pragma unroll 2
                for (; j + 1 < n; j += 2) {
                    const double2 mv = mi2[j >> 1], tv = th2[j >> 1];
                    double h0 = mv.x - Ndoc * (thi * tv.x), h1 = mv.y - Ndoc * (thi * tv.y);
                    if (j == lane) h0 = (h0 - rowc + Ndoc * thi) + sii;
                    if (j + 1 == lane) h1 = (h1 - rowc + Ndoc * thi) + sii;
                    mi2[j >> 1] = make_double2(h0, h1);
                }
                if (j < n) {
                    double h = mi[j] - Ndoc * (thi * sth[j]);
                    if (j == lane) h = (h - rowc + Ndoc * thi) + sii;
                    mi[j] = h;
                }
            } else {
                for (int j = 0; j < n; ++j) {
                    double h = mi[j] - Ndoc * (thi * sth[j]);
                    if (j == lane) h = h - rowc + Ndoc * thi;
                    mi[j] = h + S[(size_t)lane * n + j];
                }
            }
        }
        STM_POST_SYNC();

        if (P.prof) tp[3] = (long long)__builtin_readcyclecounter();
        relane();
        // ---- PD handling.  diagA: current diagonal of A (lane i); off-diagonals of A are read
        // from the upper triangle of M, which Cholesky never writes.
        double diagA = isn ? M[(size_t)lane * MLD + lane] : 1.0;
        double Ldiag = 1.0;
        // np.linalg.cholesky; L strictly-lower into M, diagonal in Ldiag.  TWO columns per step: the dot
        // products of columns j and j+1 against the finished columns share the loads of the lane's own row
        // (3 LDS reads per 2 FMAs), column j+1's last term uses L[:, j] straight from registers
        // (L[j+1][j] by v_readlane), and the serial per-column tail (pivot broadcast, sqrt, reciprocal,
        // LDS hand-off) is paid once per pair.
        auto cholesky = [&]() -> bool {
            // a pivot never exceeds its diagonal entry (what is subtracted from it are squares, in floating point too): an
            // entry <= 0 (or NaN) fails some pivot test for certain, and the attempt is decided without factorising
            if (wave_any(isn && !(diagA > 0.0))) return false;
            bool ok = true;
            int j = 0;
            for (; j + 1 < n; j += 2) {
                double tA = 0.0, tB = 0.0;
                if (isn && lane >= j) {
                    const double *ri = M + (size_t)lane * MLD, *rj = M + (size_t)j * MLD, *rk = rj + MLD;
                    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
                    int l = 0;
                    for (; l + 7 < j; l += 8) {   // twelve 16-byte LDS reads in flight per round (rows are 16-byte aligned)
                        double2 xv[4], pv[4], qv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            xv[u] = *reinterpret_cast<const double2 *>(ri + l + 2 * u);
                            pv[u] = *reinterpret_cast<const double2 *>(rj + l + 2 * u);
                            qv[u] = *reinterpret_cast<const double2 *>(rk + l + 2 * u);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            a0 = fma(xv[u].x, pv[u].x, a0); b0 = fma(xv[u].x, qv[u].x, b0);
                            a1 = fma(xv[u].y, pv[u].y, a1); b1 = fma(xv[u].y, qv[u].y, b1);
                        }
                    }
                    for (; l + 3 < j; l += 4) {   // six 16-byte LDS reads in flight per round
                        const double2 xa = *reinterpret_cast<const double2 *>(ri + l), xb = *reinterpret_cast<const double2 *>(ri + l + 2);
                        const double2 pa = *reinterpret_cast<const double2 *>(rj + l), pb = *reinterpret_cast<const double2 *>(rj + l + 2);
                        const double2 qa = *reinterpret_cast<const double2 *>(rk + l), qb = *reinterpret_cast<const double2 *>(rk + l + 2);
                        a0 = fma(xa.x, pa.x, a0); b0 = fma(xa.x, qa.x, b0);
                        a1 = fma(xa.y, pa.y, a1); b1 = fma(xa.y, qa.y, b1);
                        a0 = fma(xb.x, pb.x, a0); b0 = fma(xb.x, qb.x, b0);
                        a1 = fma(xb.y, pb.y, a1); b1 = fma(xb.y, qb.y, b1);
                    }
                    for (; l + 1 < j; l += 2) {
                        const double x0 = ri[l], x1 = ri[l + 1];
                        a0 = fma(x0, rj[l], a0);
                        b0 = fma(x0, rk[l], b0);
                        a1 = fma(x1, rj[l + 1], a1);
                        b1 = fma(x1, rk[l + 1], b1);
                    }
                    if (l < j) {
                        const double x0 = ri[l];
                        a0 = fma(x0, rj[l], a0);
                        b0 = fma(x0, rk[l], b0);
                    }
                    tA = ((lane == j) ? diagA : rj[lane]) - (a0 + a1);               // A[lane][j] - ...
                    if (lane > j) tB = ((lane == j + 1) ? diagA : rk[lane]) - (b0 + b1);
                }
                const double dA = lane_bcast(tA, j);
                if (!(dA > PIVOT_TOL * lane_bcast(diagA, j))) { ok = false; break; }   // see PIVOT_TOL
                double ljj, rjj;                // LAPACK dpotf2 scales the column by the reciprocal as well
                sqrt_and_rsqrt(dA, ljj, rjj);
                const double lA = (isn && lane > j) ? tA * rjj : 0.0;             // L[lane][j]
                tB -= lA * lane_bcast(lA, j + 1);                                 // ... - L[lane][j] L[j+1][j]
                const double dB = lane_bcast(tB, j + 1);
                if (lane == j) Ldiag = ljj;
                if (isn && lane > j) M[(size_t)lane * MLD + j] = lA;
                if (!(dB > PIVOT_TOL * lane_bcast(diagA, j + 1))) { ok = false; break; }
                double lkk, rkk;
                sqrt_and_rsqrt(dB, lkk, rkk);
                if (lane == j + 1) Ldiag = lkk;
                if (isn && lane > j + 1) M[(size_t)lane * MLD + j + 1] = tB * rkk;
                STM_POST_SYNC();
            }
            if (ok && j < n) {   // odd n: the last column on its own
                double t = 0.0;
                if (lane == j) {
                    const double *ri = M + (size_t)lane * MLD;
                    double a0 = 0.0;
                    for (int l = 0; l < j; ++l) a0 = fma(ri[l], ri[l], a0);
                    t = diagA - a0;
                }
                const double d = lane_bcast(t, j);
                if (!(d > PIVOT_TOL * lane_bcast(diagA, j))) ok = false;
                else if (lane == j) Ldiag = sqrt(d);
            }
            STM_POST_SYNC();
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
        // One Cholesky site for every stage of the reference's PD ladder (five inlined copies put
        // the fallback ones on cold paths, where the register allocator parks its spill reloads):
        //   0 hessian(): PD test as Cholesky success (stm.py:1017)   1 after make_pd (stm.py:1019-1020)
        //   2 +1e-5 (stm.py:1021), decompose_hessian's np.linalg.cholesky (stm.py:1040)
        //   3 after make_pd (stm.py:1043)   4 scipy cholesky (UPPER) of make_pd(H) + 1e-5 I (stm.py:1046-1048)
        int path = 0;
        bool upper = false, fail = false;
        double keep = 0.0;
        for (int attempt = 0;; ++attempt) {
            if (DUMP && attempt == 2) dump(P.hess_out, false);
            const bool ok = (attempt == 0 && (P.debug_flags & 8)) ? true : cholesky();
            if (attempt == 4) { diagA = keep; upper = true; fail = !ok; break; }
            if (ok) {
                if (DUMP && attempt < 2) dump(P.hess_out, false);
                break;
            }
            if (attempt == 0) { make_pd(); path = 1; }
            else if (attempt == 1) { if (isn) diagA += 1e-5; path = 2; }
            else if (attempt == 2) { make_pd(); }
            else { make_pd(); keep = diagA; if (isn) diagA += 1e-5; }
        }
        if (P.pd_path) P.pd_path[doc] = path;
        if (fail) {
            atomicMax(P.err_flag, 3 /* STM_ERR_LINALG */);
            continue;
        }
        if (DUMP && P.chol_out) {
            double *o = P.chol_out + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j) {
                    double val = (j == lane) ? Ldiag : (j < lane ? M[(size_t)lane * MLD + j] : 0.0);
                    if (upper) o[(size_t)j * n + lane] = val;  // the reference holds the upper factor here
                    else o[(size_t)lane * n + j] = val;
                }
        }

        if (P.prof) tp[4] = (long long)__builtin_readcyclecounter();
        // ---- bound (stm.py:1068-1101)
        const double det = wave_sum(isn ? log(Ldiag) : 0.0);
        double q = 0.0;
        {
            const double d = eta_i - mu_i;
            if (P.siginv_diag) {
                if (isn) q = (d * S[(size_t)lane * n + lane]) * d;
            } else {
                if (isn) sdv[lane] = d;
                STM_POST_SYNC();
                if (isn) {
                    double t = 0.0;
                    for (int j = 0; j < n; ++j) t += sdv[j] * S[(size_t)j * n + lane];
                    q = t * d;
                }
            }
        }
        q = wave_sum(q);
        P.bound[doc] = ll + (-det) - 0.5 * q - P.sigmaentropy;  // uniform store
        if (P.debug_flags & 4) continue;

        if (P.prof) tp[5] = (long long)__builtin_readcyclecounter();
        relane();
        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066)
        const double Rdiag = 1.0 / Ldiag;
        if (isn) srd[lane] = Rdiag;
        STM_POST_SYNC();
        long long ti[4] = {0, 0, 0, 0};
        if (P.prof) ti[0] = (long long)__builtin_readcyclecounter();
        if (!upper) {
            // X = L^-1 (so that nu = X^T X), blocked by 16 and IN PLACE of L: lower triangle and diagonal of M.
            // (I) the diagonal blocks, all at once, lane = (block, column c): X[i][c] = -(sum_{c<=l<i} L[i][l] X[l][c]) / L[i][i]
            //     row by row and in place; a lane only ever reads back its own column, so the steps need no hand-off.
            {
                const int c = lane & 15, base = lane & ~15;
                const int rows = n - base < 16 ? n - base : 16;    // rows of this lane's block (<= 0: no block)
                const int rb = base < n ? base : 0;                 // lanes beyond the matrix shadow block 0 (nothing is stored)
                const int rlast = (rows > 0 ? rows : 16) - 1;
                double *xc = M + (size_t)rb * MLD + (base < n ? lane : c);   // X[rb + l][column]: the lane's own column, in place
                if (base < n && c < rows) xc[(size_t)c * MLD] = srd[lane];  // X[c][c] = 1 / L[c][c]  (M's diagonal is free)
                // every step fetches its whole row of L (broadcast per block) and the whole column of X in one batch --
                // one LDS round trip per step -- and masks the terms outside [c, i)
#pragma unroll 1
                for (int i = 1; i < 16; ++i) {
                    const int ir = i < rlast ? i : rlast;           // clamped: reads stay inside the matrix
                    const double *lrow = M + (size_t)(rb + ir) * MLD + rb;
                    double lv[16], xv[16];
#pragma unroll
                    for (int l = 0; l < 16; l += 2) {
                        const double2 t = *reinterpret_cast<const double2 *>(lrow + l);
                        lv[l] = t.x; lv[l + 1] = t.y;
                        xv[l] = xc[(size_t)(l < rlast ? l : rlast) * MLD];
                        xv[l + 1] = xc[(size_t)(l + 1 < rlast ? l + 1 : rlast) * MLD];
                    }
                    const double rd = srd[rb + ir];
                    double t0 = 0.0, t1 = 0.0;
#pragma unroll
                    for (int l = 0; l < 16; l += 2) {
                        // both factors are selected: the row of L runs into columns nobody ever wrote (0 x NaN is NaN)
                        const bool m0 = l >= c && l < i, m1 = l + 1 >= c && l + 1 < i;
                        t0 = fma(m0 ? lv[l] : 0.0, m0 ? xv[l] : 0.0, t0);
                        t1 = fma(m1 ? lv[l + 1] : 0.0, m1 ? xv[l + 1] : 0.0, t1);
                    }
                    // the row-i reads of every lane precede this store in the instruction stream; later steps read rows > i of L
                    if (base < n && i > c && i < rows) xc[(size_t)i * MLD] = -(t0 + t1) * rd;
                }
            }
            STM_POST_SYNC();
            if (P.prof) ti[1] = (long long)__builtin_readcyclecounter();
            // (II) X_ij = -X_ii (sum_{j<=k<i} L_ik X_kj) on the matrix cores, block columns left to right, block rows
            //      top down (X_ij takes the place of L_ij, which no later product reads).  The inner sum comes out
            //      of the MFMA in exactly the register layout its B operand wants, so it never visits the LDS.
            {
                // runtime loops on purpose (the kernel lives at its VGPR budget): four fragment pairs in flight per step;
                // loads are unconditional on clamped rows, masks are applied to the loaded values
                const int nm1 = n - 1;
#pragma unroll 1
                for (int bj = 0; bj + 1 < NB; ++bj) {
#pragma unroll 1
                    for (int bi = bj + 1; bi < NB; ++bi) {
                        const int ar = bi * 16 + fr, arc = ar < n ? ar : nm1;
                        const double *arow = M + (size_t)arc * MLD;           // row of L_i* / X_ii for the A operands
                        const int bc = bj * 16 + fr;
                        v4d sacc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                        for (int k = bj; k < bi; ++k) {
                            double av[4], bv[4];
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) {
                                const int kk = k * 16 + 4 * sk + fq;           // < 16 (NB - 1) <= n: full blocks only
                                av[sk] = arow[kk];                              // L_ik[fr][4 sk + fq]
                                bv[sk] = M[(size_t)kk * MLD + bc];              // X_kj[4 sk + fq][fr]
                            }
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) {
                                const int kk = k * 16 + 4 * sk + fq;
                                const double a = (ar < n) ? av[sk] : 0.0;
                                const double bb = (k > bj || bc <= kk) ? bv[sk] : 0.0;   // the diagonal block of X is lower triangular
                                sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, sacc, 0, 0, 0);
                            }
                        }
                        double xv[4];
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int ac = bi * 16 + 4 * sk + fq;
                            xv[sk] = arow[ac < n ? ac : nm1];                   // X_ii[fr][4 sk + fq]
                        }
                        v4d dacc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int ac = bi * 16 + 4 * sk + fq;
                            const double a = (ac <= ar && ar < n) ? xv[sk] : 0.0;
                            dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sacc[sk], dacc, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = bi * 16 + fq + 4 * r;
                            if (row < n) M[(size_t)row * MLD + bc] = -dacc[r];
                        }
                        STM_POST_SYNC();
                    }
                }
            }
            if (P.prof) ti[2] = (long long)__builtin_readcyclecounter();
            if (REM) {   // (III) the row beyond the blocks: X[R0][j] = -X[R0][R0] sum_{j<=k<R0} L[R0][k] X[k][j], lane = column j
                double t[4] = {0.0, 0.0, 0.0, 0.0};
                const double *lr = M + (size_t)R0 * MLD, *xc = M + (lane < R0 ? lane : 0);
                for (int k = 0; k < R0; k += 8) {
                    double lv[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { lv[u] = lr[k + u]; xv[u] = xc[(size_t)(k + u) * MLD]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u & 3] = fma(lv[u], (k + u >= lane) ? xv[u] : 0.0, t[u & 3]);
                }
                const double xr = -((t[0] + t[1]) + (t[2] + t[3])) * srd[R0];
                if (lane < R0) M[(size_t)R0 * MLD + lane] = xr;   // after every lane's reads of row R0 (one instruction stream)
            }
        }
        STM_POST_SYNC();
        if (P.prof) tp[6] = (long long)__builtin_readcyclecounter();
        relane();
        if (P.prof && lane == 0 && !upper) { P.prof[doc * 40 + 28] = ti[1] - ti[0]; P.prof[doc * 40 + 29] = ti[2] - ti[1]; P.prof[doc * 40 + 30] = tp[6] - ti[2]; P.prof[doc * 40 + 31] = ti[0] - tp[5]; }
        // nu = R R^T = X^T X on the matrix cores, accumulated straight into the workgroup's running sum
        // (sigma_ss += nu, stm.py:582); fragment R[b*16 + fr][s4 + fq] = X[s4 + fq][b*16 + fr], zero below the diagonal
        v4d nud[DUMP ? NT : 1];
        if (DUMP) {
#pragma unroll
            for (int t = 0; t < NT; ++t) { nud[t] = acc_nu[t]; acc_nu[t] = (v4d){0.0, 0.0, 0.0, 0.0}; }
        }
        for (int s4 = 0; s4 < n; s4 += 4) {
            const int col = s4 + fq;
            double f[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
                double v = 0.0;
                if (row < n && col < n) {   // R[row][col] = X[col][row], X lower triangular with its diagonal in M
                    if (upper) v = (col == row) ? srd[row] : 0.0;
                    else if (col >= row) v = M[(size_t)col * MLD + row];
                }
                f[b] = v;
            }
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NB; ++bj, ++t)
                    acc_nu[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc_nu[t], 0, 0, 0);
        }
        if (REM) {   // nu[i][R0] = R[i][R0] R[R0][R0]: R is upper triangular and R0 is its last row
            double v = 0.0;
            if (lane == R0) v = Rdiag * Rdiag;
            else if (isn && !upper) v = M[(size_t)R0 * MLD + lane] * srd[R0];   // X[R0][lane] = R[lane][R0]
            nu_rem += v;
            if (DUMP && P.nu_out && isn) {
                P.nu_out[(size_t)doc * n * n + (size_t)lane * n + R0] = v;
                P.nu_out[(size_t)doc * n * n + (size_t)R0 * n + lane] = v;
            }
        }
        if (DUMP) {  // parity-test build: per-document nu
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NB; ++bj, ++t) {
                    if (P.nu_out) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = bi * 16 + fq + 4 * r, j = bj * 16 + fr;
                            if (i < n && j < n) {
                                P.nu_out[(size_t)doc * n * n + (size_t)i * n + j] = acc_nu[t][r];
                                P.nu_out[(size_t)doc * n * n + (size_t)j * n + i] = acc_nu[t][r];
                            }
                        }
                    }
                    acc_nu[t] += nud[t];
                }
        }
        if (P.prof && lane == 0) {
            tp[7] = (long long)__builtin_readcyclecounter();
            for (int q = 0; q < 7; ++q) P.prof[doc * 40 + 32 + q] = tp[q + 1] - tp[q];
        }
    }

    // ---- one flush of the workgroup's nu sum into its replica of sigma_ss
    {
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = bi; bj < NB; ++bj, ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = bi * 16 + fq + 4 * r, j = bj * 16 + fr;
                    if (i < n && j < n) {
                        unsafeAtomicAdd(sig_acc + (size_t)i * n + j, acc_nu[t][r]);
                        if (bi != bj) unsafeAtomicAdd(sig_acc + (size_t)j * n + i, acc_nu[t][r]);
                    }
                }
        if (REM && isn) {
            unsafeAtomicAdd(sig_acc + (size_t)lane * n + R0, nu_rem);
            if (lane != R0) unsafeAtomicAdd(sig_acc + (size_t)R0 * n + lane, nu_rem);
        }
    }
}

}  // namespace stm
