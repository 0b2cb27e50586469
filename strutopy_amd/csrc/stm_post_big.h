// stm_post_big.h -- the post-solve step (stm_post.h) for 64 < K <= 128 topics.
//
// Same arithmetic as post_kernel (reference src/modules/stm.py:547-588: theta, hessian + make_pd ladder,
// decompose_hessian, lower_bound, optimize_nu, update_z, accumulation), but a lane owns TWO topics / matrix
// rows (lane and lane + 64), one wave per workgroup.  That wave has its SIMD's whole 512-entry register file
// (the b b^T accumulator tiles stay in registers for the whole document), and the LDS holds only the row-packed
// lower triangle (L, then X = L^-1) so that three workgroups share a CU at K = 100; A (upper triangle) lives in
// a per-workgroup HBM scratch.  Blocked Cholesky (matrix-core block column updates, register panels), blocked
// inverse and nu = X^T X as in stm_post.h.  BASELINE config 4 (K = 100) runs here; see DESIGN.md 4.2b.
#pragma once
#include <type_traits>
#include "stm_post_common.h"

namespace stm {

constexpr int BT = 128;   // topics padded to 128

inline int post_big_mld(int n) { return 16 * ((n + 15) / 16) + 1; }   // (PostParams.MLD is not used by this kernel)
// The LDS matrix is the LOWER triangle only, row-packed (row i: i + 1 cells).  The blocked code still addresses whole
// 16 x 16 diagonal blocks: what it reads above the diagonal belongs to the following rows and is selected away, what it
// would write there goes to a per-lane dump cell behind the matrix.  A itself (upper triangle, read only by the block
// column updates of the Cholesky, make_pd and the dumps) lives in a per-workgroup HBM scratch.  45 KB instead of 124 KB
// at K = 100: three workgroups per CU.  Only the n real rows exist: row indices are clamped on reads, masked on writes.
__host__ __device__ inline int post_big_row(int i) { return (i * (i + 1)) >> 1; }
__host__ __device__ inline int post_big_nv(int n) { const int v = ((n + 1 + 7) >> 3) << 3; return v < BT ? v : BT; }   // per-topic LDS vector: K rounded up to 8
__host__ __device__ inline int post_big_tri(int n) { return post_big_row(n) + 16 + 1; }   // + slack (block reads past the last row) + one dump cell
// region 0: the matrix, or -- during the word loop -- the tile, the per-word pack and exp(eta~)
__host__ __device__ inline int post_big_reg0(int n) {
    const int tri = post_big_tri(n), tile = BT * TLD + 4 * TW + post_big_nv(n);
    return ((tri > tile ? tri : tile) + 1) & ~1;
}
// + ONE per-topic vector whose content changes with the phase (theta -> diagonal of A -> eta - mu -> 1 / diag(L)):
// 40.6 KB at K = 100, i.e. FOUR workgroups per CU -- one per SIMD (three with the 44.8 KB of separate vectors)
inline size_t post_big_lds_doubles(int n) { return (size_t)post_big_reg0(n) + post_big_nv(n); }

template <int NB>   // NB = ceil((K-1) / 16) block rows: 4 .. 8
__global__ __launch_bounds__(64) void post_big_kernel(PostParams P) {
    extern __shared__ __attribute__((aligned(16))) double big_lds[];
    int lane = threadIdx.x;
    const int K = P.K, n = P.n;
    constexpr int MROWS = 16 * NB;
    const int MDUMP = post_big_row(n) + 16, REG0 = post_big_reg0(n), NV = post_big_nv(n);
    double *M = big_lds;                        // row-packed lower triangle (post_big_row): L, then X = L^-1
    double *T = big_lds;                        // [BT][TLD] word tile, topic-major (16-byte aligned rows); word loop only
    double *wpar = big_lds + BT * TLD;          // word loop: per word of the tile { sqrt(c), S, 1/S, sqrt(c)/S }, behind the tile
    double *sex = wpar + 4 * TW;                // word loop: exp(eta~), behind the pack (all inside region 0: M is not live then)
    double *vec = big_lds + REG0;               // the one per-topic vector outside region 0
    double *Ag = P.a_scratch + (size_t)blockIdx.x * (size_t)n * n;   // A, upper triangle (row-major n x n), HBM scratch
    auto RS = [](int i) __attribute__((always_inline)) { return post_big_row(i); };
    double *sth = vec;                          // word loop + assembly: stable_softmax(eta~)
    double *sdv = vec;                          // PD ladder: current diagonal of A; bound: eta - mu (dense siginv only)
    double *srd = vec;                          // inverse: 1 / diag(L)
    const double *S = P.siginv;
    double *sig_acc = P.sigma_part + (size_t)(blockIdx.x % P.nrep) * (size_t)n * n;
    constexpr int NT = NB * (NB + 1) / 2;
    int fr = lane & 15, fq = lane >> 4;
    int k0 = lane, k1 = lane + WAVE;            // this lane's two topics / rows
    // The lane id is re-read behind an opaque move at the start of every phase: otherwise each lane-dependent LDS address
    // of the unrolled tile code is hoisted out of the document loop as a loop invariant, and lives in scratch memory.
    auto relane = [&]() __attribute__((always_inline)) {
        int l = threadIdx.x;
        asm volatile("" : "+v"(l));
        lane = l; fr = l & 15; fq = l >> 4; k0 = l; k1 = l + WAVE;
    };

    for (int64_t tk = blockIdx.x; tk < P.count; tk += gridDim.x) {
        relane();
        if (P.debug_flags & 16) {   // nothing may depend on what an earlier document or kernel left in the LDS
            __syncthreads();
            for (int q = lane; q < P.lds_doubles; q += WAVE) big_lds[q] = __builtin_nan("");
            __syncthreads();
        }
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
            if (k < NV) {
                sex[k] = exv[r];
                sth[k] = (k < K) ? thsv[r] : 0.0;
            }
        }
        for (int q = lane; q < BT * TLD; q += WAVE) T[q] = 0.0;
        __syncthreads();

        if (P.prof) tp[1] = (long long)__builtin_readcyclecounter();
        relane();
        double csum = 0.0, ll = 0.0, rowc[2] = {0.0, 0.0};
        bool bad = false;
        const int kc = (K + 3) >> 2;  // topics per quarter in the per-word sums (<= 32)
        long long tq[5] = {0, 0, 0, 0, 0};
        v4d hacc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) hacc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
        // the word ids / counts of a tile are fetched two tiles ahead, its beta rows one tile ahead (32 loads in flight
        // across the previous tile's sums, scatter and matrix-core steps)
        auto load_ids = [&](int t0, int &idx, double &c) __attribute__((always_inline)) {
            const bool in = t0 + lane < Nd && lane < TW;
            idx = in ? P.indices[p0 + t0 + lane] : 0;
            c = in ? P.counts[p0 + t0 + lane] : 0.0;
        };
        double gv[2][TW];
        auto load_rows = [&](int idx_lane) __attribute__((always_inline)) {   // words beyond the document carry id 0, lanes beyond K read topic 0
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int idx = __builtin_amdgcn_readlane(idx_lane, j);
                    gv[r][j] = bT[(size_t)idx * K + (lane + WAVE * r < K ? lane + WAVE * r : 0)];
                }
        };
        int my_idx, idx1;
        double my_c, c1n;
        load_ids(0, my_idx, my_c);
        load_ids(TW, idx1, c1n);
        constexpr bool PREFETCH = NB < 8;   // at NB = 8 the 36 accumulator tiles leave no room for a tile of rows in flight
        if (PREFETCH) load_rows(my_idx);
        for (int t0 = 0; t0 < Nd; t0 += TW) {
            relane();
            const int nw = Nd - t0 < TW ? Nd - t0 : TW;
            long long c0 = P.prof ? (long long)__builtin_readcyclecounter() : 0;
            if (!PREFETCH) load_rows(my_idx);
            // -- 1. the tile's beta rows (issued one tile ago), transposed into T[topic][word]
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + WAVE * r;
                double2 *row = reinterpret_cast<double2 *>(T + (size_t)k * TLD);
#pragma unroll
                for (int j = 0; j < TW; j += 2)
                    row[j >> 1] = make_double2((k < K && j < nw) ? gv[r][j] : 0.0, (k < K && j + 1 < nw) ? gv[r][j + 1] : 0.0);
            }
            int idx2;
            double c2n;
            if (PREFETCH && t0 + TW < Nd) load_rows(idx1);
            load_ids(t0 + 2 * TW, idx2, c2n);
            __syncthreads();
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[0] += c1 - c0; c0 = c1; }
            // -- 2. per-word sums, lane = (word fr, topic quarter fq)
            {
                double Sp = 0.0, Lp = 0.0;
                const int kb = fq * kc;
                for (int kk = 0; kk < kc; kk += 8) {   // eight topics' operands in flight; the sums keep their order
                    double tv[8], ev[8], sv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = kb + kk + u < NV ? kb + kk + u : NV - 1;   // masked below when beyond the quarter
                        tv[u] = T[(size_t)k * TLD + fr]; ev[u] = sex[k]; sv[u] = sth[k];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double a = (kk + u < kc) ? tv[u] * ev[u] : 0.0;
                        Sp += a;              // np.sum(a, 0)
                        Lp += sv[u] * a;      // theta @ (beta * exp(eta~)), stm.py:1088-1094
                    }
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
            // -- 3. scatter phi, rowsum(c'), T <- b (lane = topics k0, k1; four words x two topics in flight)
            {
                const bool v0 = k0 < K, v1 = k1 < K;
                double *tr0 = T + (size_t)k0 * TLD, *tr1 = T + (size_t)k1 * TLD;
                for (int j0 = 0; j0 < nw; j0 += 4) {
                    double sqv[4], Sv[4], rv[4], wv[4], a0[4], a1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double2 w01 = *reinterpret_cast<const double2 *>(wpar + 4 * (j0 + u));
                        const double2 w23 = *reinterpret_cast<const double2 *>(wpar + 4 * (j0 + u) + 2);
                        sqv[u] = w01.x; Sv[u] = w01.y; rv[u] = w23.x; wv[u] = w23.y;
                        a0[u] = tr0[j0 + u] * exv[0]; a1[u] = tr1[j0 + u] * exv[1];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        const bool val = j < nw;      // the words beyond the tile leave zeros in T and add nothing
                        const int idx = __builtin_amdgcn_readlane(my_idx, j & (TW - 1));
                        const double n0 = a0[u] * sqv[u], n1 = a1[u] * sqv[u];      // b = a*sqrt(c)/S (stm.py:1001)
                        const double q0 = n0 * rv[u], q1 = n1 * rv[u];
                        const double b0 = fma(fma(-q0, Sv[u], n0), rv[u], q0), b1 = fma(fma(-q1, Sv[u], n1), rv[u], q1);
                        const double ph0 = a0[u] * wv[u] * sqv[u], ph1 = a1[u] * wv[u] * sqv[u];   // stm.py:1115-1116
                        bad |= val && ((v0 && !(ph0 >= 0.0)) || (v1 && !(ph1 >= 0.0)));
                        rowc[0] += (val && v0) ? b0 * sqv[u] : 0.0;                 // rowsum(c'), stm.py:1002,1011
                        rowc[1] += (val && v1) ? b1 * sqv[u] : 0.0;
                        tr0[j] = (val && v0) ? b0 : 0.0;
                        tr1[j] = (val && v1) ? b1 : 0.0;
                        if (val && v0) unsafeAtomicAdd(bssT + (size_t)idx * K + k0, ph0);   // stm.py:588
                        if (val && v1) unsafeAtomicAdd(bssT + (size_t)idx * K + k1, ph1);
                        if (dump_phi && val) {
                            if (v0) P.phi_out[(size_t)k0 * Nd + t0 + j] = ph0;
                            if (v1) P.phi_out[(size_t)k1 * Nd + t0 + j] = ph1;
                        }
                    }
                }
            }
            __syncthreads();
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[2] += c1 - c0; c0 = c1; }
            // -- 4. H += b b^T restricted to the tile on the matrix cores: the NB (NB + 1) / 2 accumulator tiles of the upper
            // block triangle stay in registers for the whole document (one wave per SIMD owns the full 512-register file)
            {
                double f[NB][TW / 4];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int sk = 0; sk < TW / 4; ++sk) f[b][sk] = T[(size_t)(b * 16 + fr) * TLD + sk * 4 + fq];   // rows beyond n: masked at the store
                int t = 0;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj, ++t)
#pragma unroll
                        for (int sk = 0; sk < TW / 4; ++sk)
                            hacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi][sk], f[bj][sk], hacc[t], 0, 0, 0);
            }
            __syncthreads();
            if (P.prof) { long long c1 = __builtin_readcyclecounter(); tq[3] += c1 - c0; c0 = c1; }
            my_idx = idx1; my_c = c1n; idx1 = idx2; c1n = c2n;
        }
        if (P.prof && lane == 0) for (int q = 0; q < 4; ++q) P.prof[doc * PROF_SLOTS + 24 + q] = tq[q];
        if (P.prof) tp[2] = (long long)__builtin_readcyclecounter();
        if (wave_any(bad)) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */);
        const double Ndoc = (double)(long long)wave_sum(csum);
        ll = wave_sum(ll);

        // ---- H = b b^T - N theta theta^T, diag += -rowsum(c') + N theta, [:-1,:-1] + siginv, formed on the accumulator
        // tiles and written to the HBM scratch (upper triangle: all that make_pd, the Cholesky and the dumps read);
        // the diagonal also goes to sdv
        // (the diagonal's -rowsum(c') + N theta + siginv_ii is applied by the lane that owns the row, below: it holds rowsum(c'))
        __syncthreads();
        {
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NB; ++bj, ++t) {
                    const int j = bj * 16 + fr, jc = j < n ? j : n - 1;
                    const double thj = sth[jc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = bi * 16 + fq + 4 * r, ic = i < n ? i : n - 1;
                        const double thi = sth[ic];
                        double h = hacc[t][r] - Ndoc * (thi * thj);
                        const bool dg = bi == bj && i == j;
                        const double sij = (dg || P.siginv_diag) ? 0.0 : S[(size_t)ic * n + jc];
                        const double v = dg ? h : h + sij;     // the diagonal stays raw here
                        if (j < n && (bi != bj || (i <= j))) Ag[(size_t)i * n + j] = v;
                    }
                }
        }
        __syncthreads();

        if (P.prof) tp[3] = (long long)__builtin_readcyclecounter();
        relane();
        // ---- PD ladder around one Cholesky (the upper triangle keeps A, L goes to the strict lower triangle)
        double diagA[2], Ldiag[2] = {1.0, 1.0};
        // the diagonal, by the lane that owns the row: (h_ii - rowsum(c')_i) + N theta_i, + siginv_ii (stm.py:1003-1013), in that order
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = lane + WAVE * r;
            diagA[r] = 1.0;
            if (i < n) {
                const double h = Ag[(size_t)i * n + i];
                diagA[r] = ((h - rowc[r]) + Ndoc * thsv[r]) + S[(size_t)i * n + i];
                Ag[(size_t)i * n + i] = diagA[r];
            }
        }
        __syncthreads();   // every reader of theta (sth) is done: the vector now carries the diagonal
        auto make_pd = [&]() __attribute__((always_inline)) {  // stm.py:964-984
            // A comes from the HBM scratch: both rows of the lane and eight columns per round in flight, sums in column order
            const int i0 = k0 < n ? k0 : n - 1, i1 = k1 < n ? k1 : n - 1;
            double mag0 = 0.0, mag1 = 0.0;
            for (int j0 = 0; j0 < n; j0 += 8) {
                double v0[8], v1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u < n ? j0 + u : n - 1;
                    v0[u] = Ag[j > i0 ? (size_t)i0 * n + j : (size_t)j * n + i0];
                    v1[u] = Ag[j > i1 ? (size_t)i1 * n + j : (size_t)j * n + i1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    mag0 += (j == i0 || j >= n) ? 0.0 : fabs(v0[u]);
                    mag1 += (j == i1 || j >= n) ? 0.0 : fabs(v1[u]);
                }
            }
            if (k0 < n && diagA[0] < mag0) diagA[0] = mag0;
            if (k1 < n && diagA[1] < mag1) diagA[1] = mag1;
        };
        auto dump = [&](double *base) __attribute__((always_inline)) {
            if (!base) return;
            double *o = base + (size_t)doc * n * n;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n)
                    for (int j = 0; j < n; ++j)
                        o[(size_t)i * n + j] = (j == i) ? diagA[r] : (j > i ? Ag[(size_t)i * n + j] : Ag[(size_t)j * n + i]);
            }
        };
        constexpr int nblk = NB;
        const int nm1 = n - 1;
        long long tc[3] = {0, 0, 0};
        int path = 0;
        bool upper = false, fail = false;
        double keep[2] = {0.0, 0.0};
        for (int attempt = 0;; ++attempt) {
            if (attempt == 2) dump(P.hess_out);
            bool ok = true;
            // blocked left-looking Cholesky, panels of 16 columns: (a) the block column minus the products of the
            // finished panels on the matrix cores, A read from the upper triangle (transposed) and the current diagonal
            // from sdv; (b) the panel itself with a lane's two rows in registers
            __syncthreads();
            sdv[lane] = diagA[0];
            if (lane + WAVE < NV) sdv[lane + WAVE] = diagA[1];
            __syncthreads();
            // every pivot that passes lies in (32 eps, 1] x its diagonal entry (up to rounding): the range test of
            // sqrt_and_rsqrt can be made on the diagonal, once
            const bool fast_sqrt = !wave_any((k0 < n && !(diagA[0] > 1e-260 && diagA[0] < 1e270)) || (k1 < n && !(diagA[1] > 1e-260 && diagA[1] < 1e270)));
            // A pivot never exceeds its diagonal entry (the products subtracted from it are squares, in floating point too),
            // so a diagonal entry <= 0 (or NaN) fails some pivot test for certain: the outcome of this attempt is known
            // without factorising (at random init most K = 100 documents leave the first rung this way).
            if (wave_any((k0 < n && !(diagA[0] > 0.0)) || (k1 < n && !(diagA[1] > 0.0)))) ok = false;
#pragma unroll 1
            for (int p = 0; p < nblk && ok; ++p) {
                const int J0 = 16 * p;
                long long cc0 = P.prof ? (long long)__builtin_readcyclecounter() : 0;
                const int bc = J0 + fr, bcc = bc < n ? bc : nm1;
                const double *brow = M + RS(bcc);                        // row of L_p* for the B operands (L_pk^T)
                const double *acol = Ag + (size_t)bcc * n;               // A[i][bc] = A[bc][i], i >= bc: row bc of the upper triangle
#pragma unroll 1
                for (int bi = p; bi < nblk; bi += 2) {
                    const bool two = bi + 1 < nblk;
                    const int ar0 = bi * 16 + fr, ar1 = (two ? bi + 1 : bi) * 16 + fr;
                    const double *arow0 = M + RS(ar0 < n ? ar0 : nm1), *arow1 = M + RS(ar1 < n ? ar1 : nm1);
                    double old0[4], old1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // A[i][j] = M[j][i] (upper), diagonal from sdv
                        const int i0 = bi * 16 + fq + 4 * r, i1 = i0 + 16;
                        const int i0c = i0 < n ? i0 : nm1, i1c = i1 < n ? i1 : nm1;
                        old0[r] = (i0c == bcc) ? sdv[bcc] : acol[i0c >= bcc ? i0c : bcc];
                        old1[r] = acol[i1c >= bcc ? i1c : bcc];
                    }
                    v4d a0 = (v4d){0.0, 0.0, 0.0, 0.0}, a1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                    for (int k = 0; k < p; ++k) {
                        double av0[4], av1[4], bv[4];
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int kk = k * 16 + 4 * sk + fq;
                            av0[sk] = arow0[kk]; av1[sk] = arow1[kk]; bv[sk] = brow[kk];
                        }
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[sk], bv[sk], a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[sk], bv[sk], a1, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i0 = bi * 16 + fq + 4 * r, i1 = i0 + 16;
                        // no masks: rows / columns beyond n are padding nobody reads, and what would land above the
                        // diagonal (where A lives) goes to the padding column instead
                        M[(i0 >= bc && i0 < n) ? RS(i0) + bc : MDUMP] = old0[r] - a0[r];
                        if (two) M[i1 < n ? RS(i1) + bc : MDUMP] = old1[r] - a1[r];
                    }
                }
                __syncthreads();
                if (P.prof) { long long c1 = __builtin_readcyclecounter(); tc[0] += c1 - cc0; cc0 = c1; }
                // (b) the panel: rows k0, k1 of the block column in registers; column j's finished entries of row J come
                //     from the lane that owns row J (all of a panel's rows sit in one register set: J >> 6 == p >> 2)
                double w[2][16];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i = lane + WAVE * r, ic = i < n ? i : nm1;
                    const double *wr = M + RS(ic >= J0 ? ic : J0);   // rows above the panel shadow its first row (never stored)
#pragma unroll
                    for (int c = 0; c < 16; ++c) w[r][c] = wr[J0 + c];
                }
                // Right-looking inside the panel and free of branches, so that the updates of the later columns fill the
                // latency of the square root chain: a failed pivot (or a column beyond n) only raises a flag, and whatever
                // the remaining steps compute from it is never stored.
                const bool hi = (p >> 2) != 0;
                const int ol0 = J0 & 63;
                auto panel = [&](auto fastc) __attribute__((always_inline)) {
                    bool bad = false;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int J = J0 + j;
                        const double d = lane_bcast(hi ? w[1][j] : w[0][j], ol0 + j);
                        const double dA = lane_bcast(hi ? diagA[1] : diagA[0], ol0 + j);
                        bad |= (J < n) && !(d > PIVOT_TOL * dA);
                        double ljj, rjj;
                        if constexpr (decltype(fastc)::value) {   // sqrt_and_rsqrt without its range test (made on diagA, once)
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
                            ljj = g; rjj = h + h;
                        } else {
                            ljj = sqrt(d); rjj = 1.0 / ljj;
                        }
                        if (k0 == J) Ldiag[0] = ljj;
                        if (k1 == J) Ldiag[1] = ljj;
                        w[0][j] *= rjj; w[1][j] *= rjj;
#pragma unroll
                        for (int c = j + 1; c < 16; ++c) {
                            const double x = lane_bcast(hi ? w[1][j] : w[0][j], ol0 + c);   // L[J0 + c][J]
                            w[0][c] = fma(-w[0][j], x, w[0][c]);
                            w[1][c] = fma(-w[1][j], x, w[1][c]);
                        }
                    }
                    return bad;
                };
                if (P.prof) { long long c1 = __builtin_readcyclecounter(); tc[1] += c1 - cc0; cc0 = c1; }
                if (fast_sqrt ? panel(std::true_type{}) : panel(std::false_type{})) ok = false;
                if (P.prof) { long long c1 = __builtin_readcyclecounter(); tc[2] += c1 - cc0; cc0 = c1; }
                if (!ok) break;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i = lane + WAVE * r;
#pragma unroll
                    for (int c = 0; c < 16; ++c)   // rows on or above the diagonal write to the padding column instead
                        M[(i < n && i > J0 + c) ? RS(i) + J0 + c : MDUMP] = w[r][c];
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
                        const double val = (j == i) ? Ldiag[r] : (j < i ? M[RS(i) + j] : 0.0);
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
        relane();
        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066): X = L^-1 blocked by 16 and IN PLACE of L
        // (lower triangle and diagonal of M), then nu = X^T X -- the scheme of post_kernel with run-time block counts
        double Rdiag[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            Rdiag[r] = 1.0 / Ldiag[r];
            if (lane + WAVE * r < NV) srd[lane + WAVE * r] = (lane + WAVE * r < n) ? Rdiag[r] : 0.0;
        }
        __syncthreads();
        long long ti[3] = {0, 0, 0};
        if (P.prof) ti[0] = (long long)__builtin_readcyclecounter();
        if (!upper) {
            // (I) the diagonal blocks, four per pass, lane = (block, column c):
            //     X[i][c] = -(sum_{c<=l<i} L[i][l] X[l][c]) / L[i][i], row by row; a lane reads back only its own column
#pragma unroll 1
            for (int pass = 0; pass * WAVE < n; ++pass) {
                const int gl = lane + WAVE * pass;
                const int c = lane & 15, base = gl & ~15;
                const int rows = n - base < 16 ? n - base : 16;    // rows of this lane's block (<= 0: no block)
                const int rb = base < n ? base : 0;                 // lanes beyond the matrix shadow block 0 (nothing is stored)
                const int rlast = (rows > 0 ? rows : 16) - 1;
                const int col = base < n ? gl : c;                  // X[rb + l][col] = M[RS(rb + l) + col]
                if (base < n && c < rows) M[RS(rb + c) + col] = srd[gl];   // X[c][c] = 1 / L[c][c]  (M's diagonal is free)
#pragma unroll 1
                for (int i = 1; i < 16; ++i) {
                    const int ir = i < rlast ? i : rlast;           // clamped: reads stay inside the matrix
                    const double *lrow = M + RS(rb + ir) + rb;
                    double lv[16], xv[16];
#pragma unroll
                    for (int l = 0; l < 16; ++l) {
                        lv[l] = lrow[l];
                        xv[l] = M[RS(rb + (l < rlast ? l : rlast)) + col];
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
                    if (base < n && i > c && i < rows) M[RS(rb + i) + col] = -(t0 + t1) * rd;
                }
            }
            __syncthreads();
            if (P.prof) ti[1] = (long long)__builtin_readcyclecounter();
            // (II) X_ij = -X_ii (sum_{j<=k<i} L_ik X_kj) on the matrix cores, block columns left to right, block rows top
            //      down (X_ij takes the place of L_ij, which no later product reads); the inner sum leaves the MFMA in the
            //      register layout its B operand wants
#pragma unroll 1
            for (int bj = 0; bj + 1 < nblk; ++bj) {
#pragma unroll 1
                for (int bi = bj + 1; bi < nblk; ++bi) {
                    const int ar = bi * 16 + fr, arc = ar < n ? ar : nm1;
                    const double *arow = M + RS(arc);                     // row of L_i* / X_ii for the A operands
                    const int bc = bj * 16 + fr;
                    v4d sacc0 = (v4d){0.0, 0.0, 0.0, 0.0}, sacc1 = (v4d){0.0, 0.0, 0.0, 0.0};
                    int k = bj;
#pragma unroll 1
                    for (; k + 1 < bi; k += 2) {   // two k blocks in flight, separate accumulators (summed below)
                        double av[8], bv[8];
#pragma unroll
                        for (int sk = 0; sk < 8; ++sk) {
                            const int kk = k * 16 + 4 * sk + fq;           // < 16 (nblk - 1) <= n: full blocks only
                            av[sk] = arow[kk];                              // L_ik[fr][4 sk + fq]
                            bv[sk] = M[RS(kk) + bc];                        // X_kj[4 sk + fq][fr]
                        }
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int kk = k * 16 + 4 * sk + fq;
                            const double a = (ar < n) ? av[sk] : 0.0, a2 = (ar < n) ? av[sk + 4] : 0.0;
                            const double bb = (k > bj || bc <= kk) ? bv[sk] : 0.0;   // the diagonal block of X is lower triangular
                            sacc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, sacc0, 0, 0, 0);
                            sacc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bv[sk + 4], sacc1, 0, 0, 0);
                        }
                    }
                    if (k < bi) {
                        double av[4], bv[4];
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int kk = k * 16 + 4 * sk + fq;
                            av[sk] = arow[kk];
                            bv[sk] = M[RS(kk) + bc];
                        }
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int kk = k * 16 + 4 * sk + fq;
                            const double a = (ar < n) ? av[sk] : 0.0;
                            const double bb = (k > bj || bc <= kk) ? bv[sk] : 0.0;
                            sacc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, sacc0, 0, 0, 0);
                        }
                    }
                    const v4d sacc = sacc0 + sacc1;
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
                        if (row < n) M[RS(row) + bc] = -dacc[r];
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (P.prof) tp[6] = (long long)__builtin_readcyclecounter();
        relane();
        if (P.prof && lane == 0 && !upper) { P.prof[doc * PROF_SLOTS + 28] = ti[1] - ti[0]; P.prof[doc * PROF_SLOTS + 29] = tp[6] - ti[1]; }
        if (P.prof && lane == 0) { P.prof[doc * PROF_SLOTS + 30] = tc[0]; P.prof[doc * PROF_SLOTS + 31] = tc[2]; P.prof[doc * PROF_SLOTS + 23] = tc[1]; }
        // nu = R R^T = X^T X (sigma_ss += nu, stm.py:582), one block column bj of output tiles (bi <= bj) at a time on the
        // matrix cores: nu[i][j] = sum_{l >= 16 bj} X[l][i] X[l][j]; fragment X[s4 + fq][b*16 + fr], zero above the diagonal
        double *nu_doc = P.nu_out ? P.nu_out + (size_t)doc * n * n : nullptr;
        if (upper) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + WAVE * r;
                if (i < n) {
                    const double v = Rdiag[r] * Rdiag[r];
                    unsafeAtomicAdd(sig_acc + (size_t)i * n + i, v);
                    if (nu_doc)
                        for (int j = 0; j < n; ++j) nu_doc[(size_t)i * n + j] = (j == i) ? v : 0.0;
                }
            }
        } else {
#pragma unroll 1
            for (int bj = 0; bj < nblk; ++bj) {
                const int rj = bj * 16 + fr, rjc = rj < n ? rj : nm1;
                v4d acc[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                for (int s4 = bj * 16; s4 < n; s4 += 4) {
                    const int col = s4 + fq, colc = col < n ? col : nm1;
                    const double *xr = M + RS(colc);
                    double f[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) f[b] = xr[b < bj ? b * 16 + fr : rjc];   // blocks beyond bj repeat block bj (unused)
                    const double fb = (col < n && rj < n && col >= rj) ? f[7] : 0.0;   // f[7] is always block bj's own fragment
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        if (b > bj) break;
                        const double fa = (b == bj) ? fb : ((col < n) ? f[b] : 0.0);
                        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[b], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (b > bj) break;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = b * 16 + fq + 4 * r, j = rj;
                        if (i < n && j < n) {
                            unsafeAtomicAdd(sig_acc + (size_t)i * n + j, acc[b][r]);
                            // blocks below the block diagonal are mirrored once, after the replicas are summed (mirror_blocks_kernel)
                            if (nu_doc) {
                                nu_doc[(size_t)i * n + j] = acc[b][r];
                                nu_doc[(size_t)j * n + i] = acc[b][r];
                            }
                        }
                    }
                }
            }
        }
        if (P.prof && lane == 0) {
            tp[7] = (long long)__builtin_readcyclecounter();
            for (int q = 0; q < 7; ++q) P.prof[doc * PROF_SLOTS + 32 + q] = tp[q + 1] - tp[q];
        }
        __syncthreads();
    }
}

}  // namespace stm
