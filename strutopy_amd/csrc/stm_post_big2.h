// stm_post_big2.h -- the post-solve step for 64 < K <= 112 topics: TWO wavefronts per document.
//
// Same arithmetic as post_kernel (stm_post.h; reference src/modules/stm.py:547-588: theta, hessian + make_pd ladder,
// decompose_hessian, lower_bound, optimize_nu, update_z, the sigma_ss / beta_ss accumulation) and the same data path:
// word tiles fetched betaT -> LDS by global_load_lds_dwordx4, r_dw stored for the word-major beta_ss pass
// (stm_betass.h) instead of phi atomics, the (K-1)^2 matrix as a row-packed lower triangle in LDS that a failed rung of
// the PD ladder re-assembles from the live b b^T accumulators, nu summed into the workgroup's own slab (no atomics:
// run-to-run identical results).  What differs from K <= 64 is who does the work.  A K = 100 matrix takes 40 KB of LDS,
// so four documents fit a CU whatever the kernel does; with one wave per document (stm_post_big.h, rounds 1-3) that is
// one wave per SIMD holding 28 accumulator tiles in 224 + working registers, and nothing hides the latency of its
// serial chains.  Here a workgroup is two waves (<= 256 registers each: two waves per SIMD, eight per CU) that share
// the document:
//   * lane l of wave w owns topic / matrix row 64 w + l wherever the K <= 64 kernel says "lane = topic / row";
//   * the accumulator tiles of b b^T, the block rows of the Cholesky's block-column updates, the diagonal blocks and
//     the block columns of the inverse and of nu = X^T X are dealt out to the two waves (each SIMD has its own matrix
//     core), the per-word sums split the tile's words (lane = (word, eighth of the topics));
//   * the Cholesky panel of 16 columns runs on BOTH waves without any exchange inside it: the wave that does not own
//     the panel's diagonal block carries a shadow copy of those 16 rows in its lanes 48..63 (n - 64 <= 48) and repeats
//     the pivot chain on it -- the same instructions on the same numbers -- while its own rows are updated;
//   * the off-diagonal blocks of X = L^-1 form a dependency grid (block (r, c) needs (r-1, c) and overwrites what
//     (r, c-1) read): its anti-diagonals are independent and are dealt out two blocks at a time.
// The waves meet at workgroup barriers that wait for the LDS only (the tile fetch of the next tile stays in flight).
// K > 112 (n - 64 > 48: no room for the shadow rows) stays with post_big_kernel.
#pragma once
#include "stm_post.h"

namespace stm {

// LDS hand-off between the two waves of a workgroup: every LDS operation of this wave has completed, then the barrier.
// (__syncthreads() would also drain the global loads that are deliberately kept in flight across it.)
#define STM_WG_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// LDS map (doubles).  Region 0 is the two word tiles + the per-word / per-topic vectors during the word loop and the
// packed matrix afterwards; the exchange slots of the two waves and the Cholesky panels' column broadcasts sit behind it.
struct Post2Lds {
    int tile;    // doubles per tile buffer: 16 rows of PITCH = 2 PC doubles
    int wpar;    // per word { sqrt(c) / S, sqrt(c) }
    int sex;     // exp(eta~) per topic, zeros from K on (2 PC entries)
    int eth;     // exp(eta~) * stable_softmax(eta~)
    int sth;     // stable_softmax(eta~) (128 entries)
    int sdv;     // eta - mu (dense siginv only; 128 entries)
    int mdump;   // two cells behind the matrix (+ 16 of slack) that masked stores go to
    int xch;     // 16 exchange slots
    int cb;      // 2 x 16: a panel's column broadcast, one per wave
    int total;
};
__host__ __device__ inline Post2Lds post2_lds_map(int K, int PC) {
    Post2Lds L;
    const int n = K - 1;
    L.tile = TW * 2 * PC;
    L.wpar = 2 * L.tile;
    L.sex = L.wpar + 2 * TW;
    L.eth = L.sex + 2 * PC;
    L.sth = L.eth + 2 * PC;
    L.sdv = L.sth + 128;
    const int tile_part = L.sdv + 128;
    L.mdump = tri_row(n) + 16;
    const int mat_part = L.mdump + 2;
    L.xch = ((tile_part > mat_part ? tile_part : mat_part) + 1) & ~1;
    L.cb = L.xch + 16;
    L.total = L.cb + 32;
    return L;
}
// tile row pitch in 16-byte chunks: >= ceil(K / 2) and = 8 (mod 16) -- the per-word sums (ds_read_b128, lane = (word,
// eighth), chunk e + 8 kk) and the MFMA fragment reads (ds_read_b64, pitch = 16 mod 32 doubles) are then conflict-free
__host__ __device__ inline int post2_pc(int K) { return K <= 80 ? 40 : 56; }
__host__ __device__ inline bool post2_serves(int K) { return K > 64 && K <= 112; }

// X_FLAG: four slots -- (attempt parity, wave) -- so that a wave which returns early from attempt a and enters attempt a + 1
// without a barrier in between (a clean matrix: no assemble) never rewrites a flag its partner has not read yet
enum { X_CSUM = 0, X_LL = 2, X_Q = 4, X_DET = 6, X_FLAG = 8, X_BAD = 12 };

// NWV = 2 or 4 waves per document.  Four (round 5): waves 2 and 3 own no topics / matrix rows -- they take their share of everything that is
// dealt out by tile, block row or block column (b b^T, the block-column updates, the inverse's block columns, nu) and wait at the barriers
// through the rest; every element is still produced by the same operations in the same order, so the results are the two-wave kernel's bit
// for bit.  With <= 128 registers a wave of every resident document sits on every SIMD (four documents per CU either way: the LDS).
#ifndef STM_PB2_W4_OCC
#define STM_PB2_W4_OCC 3   // waves per SIMD the four-wave form is compiled for: 3 -- <= 168 registers (64 spilled), three documents per CU: 21.8 ms at config 4's
                           // share; 4 -- <= 128 registers, four documents per CU, but 536 spilled registers: 30.5 ms (the two-wave form: 19.9 ms)
#endif
template <int W0, int NWV, typename F>
__device__ __forceinline__ void by_wave(int wv, F &&f) {   // f(integral_constant<int, wv>), wv known at compile time inside
    if constexpr (W0 + 1 == NWV) f(std::integral_constant<int, W0>{});
    else { if (wv == W0) f(std::integral_constant<int, W0>{}); else by_wave<W0 + 1, NWV>(wv, f); }
}

#ifndef STM_PB2_W2_OCC
#define STM_PB2_W2_OCC 2   // waves per SIMD the two-wave form is compiled for (1: 512 registers, no spills -- a diagnostic: the LDS keeps four documents per CU)
#endif
template <int NB, int PC, bool DBG, int NWV = 2>
__global__ __launch_bounds__(64 * NWV, NWV == 2 ? STM_PB2_W2_OCC : STM_PB2_W4_OCC) void post_big2_kernel(PostParams P) {
    static_assert(NWV == 2 || NWV == 4, "two or four waves per document");
    constexpr int NT = NB * (NB + 1) / 2, NTW = (NT + NWV - 1) / NWV;   // accumulator tiles: all / per wave (tile t belongs to wave t % NWV)
    constexpr int PITCH = 2 * PC, QP = PC / 8;
    constexpr int NQW = TW * PC / 128;    // LDS-DMA instructions per wave and tile
    constexpr int TILE = TW * PITCH;
    static_assert(PC % 16 == 8 && PITCH >= 16 * NB && TW * PC % 128 == 0, "tile pitch");
    extern __shared__ __attribute__((aligned(16))) double post2_lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int lane = threadIdx.x & 63, gl = threadIdx.x;
    const int K = P.K, n = P.n, nm1 = n - 1;
    const Post2Lds LM = post2_lds_map(K, PC);
    const int CP = (K + 1) >> 1, MDUMP = LM.mdump;
    double *M = post2_lds;
    double *wpar = post2_lds + LM.wpar, *sex = post2_lds + LM.sex, *eth = post2_lds + LM.eth;
    double *sth = post2_lds + LM.sth, *sdv = post2_lds + LM.sdv;
    double *xch = post2_lds + LM.xch;
    double *cb = post2_lds + LM.cb + 16 * (wv & 1);
    const bool rowwave = NWV == 2 || wv < 2;   // (uniform) the waves whose lanes own topics / matrix rows
    const double *S = P.siginv;
    const bool sdiag = P.siginv_diag != 0;
    // this workgroup's own sum of nu, accumulator-tile layout (tile (b, bj), b <= bj, at bj (bj + 1) / 2 + b)
    double *sig_acc = P.sigma_part + (size_t)blockIdx.x * (size_t)NT * 4 * WAVE;
    bool isn = gl < n, isk = gl < K;
    int fr = lane & 15, fq = lane >> 4;
    auto relane = [&]() __attribute__((always_inline)) {
        int l = threadIdx.x;
        asm volatile("" : "+v"(l));
        gl = l; lane = l & 63; isn = l < n; isk = l < K; fr = l & 15; fq = (l >> 4) & 3;
    };
    auto RS = [](int i) __attribute__((always_inline)) { return tri_row(i); };
    const unsigned K8 = 8u * (unsigned)K;

    for (int64_t tk = blockIdx.x; tk < P.count; tk += gridDim.x) {
        relane();
        STM_WG_SYNC();   // the previous document's readers of the LDS are done
        if (DBG && (P.debug_flags & 16)) {   // nothing may depend on what an earlier document or kernel left in the LDS
            for (int q = gl; q < P.lds_doubles; q += NWV * WAVE) post2_lds[q] = __builtin_nan("");
            STM_WG_SYNC();
        }
        const int64_t ticket = P.first + tk;
        int64_t doc, p0;
        int Nd;
        if (P.tick) {
            const int64_t t0 = scalar_load(P.tick + 2 * ticket), t1 = scalar_load(P.tick + 2 * ticket + 1);
            p0 = t0; doc = t1 & 0xffffffffLL; Nd = (int)(t1 >> 32);
        } else {
            doc = P.order ? (int64_t)scalar_load(P.order + ticket) : ticket;
            p0 = scalar_load(P.indptr + doc);
            Nd = (int)(scalar_load(P.indptr + doc + 1) - p0);
        }
        const int asp = P.aspect ? scalar_load(P.aspect + doc) : 0;
        const double *bT = P.betaT + (size_t)asp * (size_t)P.V * K;
        long long tp[8];
        if (DBG) tp[0] = P.prof ? (long long)__builtin_readcyclecounter() : 0;

        // word ids (lane w < 16: word t0 + w, both waves) and counts / word-major slots (lane 8 w' + e of wave wv: word
        // t0 + 8 wv + w'); lanes beyond the document repeat its last word (a valid row for the fetch)
        auto load_ids = [&](int t0, int &idx, double &c, int &slot) __attribute__((always_inline)) {
            idx = 0; c = 0.0; slot = 0;
            if (rowwave) {
                const int wi = t0 + lane, wc = t0 + 8 * wv + (lane >> 3), last = Nd - 1;
                idx = P.indices[p0 + (wi < last ? wi : last)];
                c = P.counts[p0 + (wc < last ? wc : last)];
                slot = P.wm_slot[p0 + (wc < last ? wc : last)];
            }
        };
        // the 16 rows of a tile, betaT -> LDS: chunk c = 64 q + lane (16 bytes) is chunk c mod PC of word c / PC; this wave
        // issues the instructions q = wv, wv + 2, ...  Chunks beyond the topics of a row repeat its last one: finite, and
        // the sums meet them with zeros.
        auto tile_fetch = [&](int idxv, int buf) __attribute__((always_inline)) {
            if (!rowwave) return;
            unsigned off[NQW];
            int id[NQW];
#pragma unroll
            for (int i = 0; i < NQW; ++i) {
                const int c = 64 * (2 * i + wv) + lane;
                int w = (int)(((unsigned)c * (65536u / PC + 1u)) >> 16);      // c / PC for c < 1024
                int o = c - (int)__umul24((unsigned)w, (unsigned)PC);
                o = o < CP ? o : CP - 1;
                id[i] = __builtin_amdgcn_ds_bpermute(4 * w, idxv);
                off[i] = 16u * (unsigned)o;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NQW; ++i) off[i] += __umul24((unsigned)id[i], K8);
            const unsigned lds0 = lds_addr(post2_lds + buf * TILE) + 1024u * (unsigned)wv;
            unsigned keep;
#pragma unroll
            for (int i = 0; i < NQW; ++i)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off[i]), "s"(lds0 + 2048u * i), "s"(bT) : "memory");
        };

        int idx0, idx1, sl0, sl1;
        double c0, c1;
        load_ids(0, idx0, c0, sl0);
        load_ids(TW, idx1, c1, sl1);

        // ---- eta~, theta (unshifted softmax, stm.py:547-549), stable softmax, exp(eta~).  Every wave-wide sum runs over
        // all K topics on BOTH waves (a lane evaluates its own topic and its partner's in the other wave): the same
        // additions in the same order, so the two waves hold the same bits without an exchange.
        const int go = gl ^ 64;
        const double eta_i = isn ? P.eta[doc * n + gl] : 0.0;   // topic K-1 holds the appended 0
        const double eta_o = go < n ? P.eta[doc * n + go] : 0.0;
        const double mu_i = isn ? P.mu[doc * n + gl] : 0.0;
        const double ex = isk ? exp(eta_i) : 0.0, ex_o = go < K ? exp(eta_o) : 0.0;
        const double sumex = wave_sum(wv == 0 ? ex + ex_o : ex_o + ex);
        if (isk) P.theta[doc * K + gl] = ex / sumex;
        const double m = wave_nanmax(nanmax(isk ? eta_i : -INFINITY, go < K ? eta_o : -INFINITY));
        const double es = isk ? exp(eta_i - m) : 0.0, es_o = go < K ? exp(eta_o - m) : 0.0;
        const double ssum = wave_sum(wv == 0 ? es + es_o : es_o + es);
        const double ths = es / ssum;
        if (gl < 2 * PC) {
            sex[gl] = ex;                          // zeros from K on
            eth[gl] = isk ? ex * ths : 0.0;        // theta . (beta * exp(eta~)) = sum_k beta_k (exp(eta~)_k theta_k), stm.py:1088-1094
        }
        if (rowwave) {
            sth[gl] = isk ? ths : 0.0;
            sdv[gl] = isn ? eta_i - mu_i : 0.0;
        }
        tile_fetch(idx0, 0);
        STM_WG_SYNC();
        // (eta - mu)^T siginv (eta - mu) (stm.py:1098-1099): nothing later depends on it, and region 0 is free for its vector now
        {
            double q = 0.0;
            const double d = eta_i - mu_i;
            if (sdiag) {
                if (isn) q = (d * S[(size_t)gl * n + gl]) * d;
            } else if (isn) {   // column gl of siginv from the L2, eight rows in flight; the sum keeps its order
                double t = 0.0;
                int j = 0;
                for (; j + 7 < n; j += 8) {
                    double sv8[8], dv8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sv8[u] = S[(size_t)(j + u) * n + gl]; dv8[u] = sdv[j + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) t += dv8[u] * sv8[u];
                }
                for (; j < n; ++j) t += sdv[j] * S[(size_t)j * n + gl];
                q = t * d;
            }
            q = wave_sum(q);
            if (lane == 0 && rowwave) xch[X_Q + wv] = q;
        }

        if (DBG && P.prof) tp[1] = (long long)__builtin_readcyclecounter();
        double csum = 0.0, ll = 0.0, rowc = 0.0;
        double Lst = 1.0, cst = 0.0;   // a (word, tile) pair's theta @ a and count waiting for the next batched logarithm
        bool sbad = false;
        v4d acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
        long long tq[4] = {0, 0, 0, 0};

        const bool dump_phi = P.phi_out && doc == P.phi_doc;
        for (int t0 = 0, buf = 0; t0 < Nd; t0 += TW, buf ^= 1) {
            const int nw = Nd - t0 < TW ? Nd - t0 : TW;
            long long cy0 = (DBG && P.prof) ? (long long)__builtin_readcyclecounter() : 0;
            relane();
            wait_vmem();        // this wave's share of the tile (fetched a whole tile ago) and its word ids have landed
            STM_WG_SYNC();      // ... and the other wave's; the other buffer's readers finished a tile ago
            double *T = post2_lds + buf * TILE;
            if (t0 + TW < Nd) tile_fetch(idx1, buf ^ 1);
            int idx2, sl2;
            double c2;
            load_ids(t0 + 2 * TW, idx2, c2, sl2);
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[0] += cy - cy0; cy0 = cy; }
            // -- 1. per-word sums, lane = (word 8 wv + (lane >> 3), eighth lane & 7 of the row: chunks e, e + 8, ...)
            if (rowwave) {
                const int w = 8 * wv + (lane >> 3), e = lane & 7;
                const double2 *T2 = reinterpret_cast<const double2 *>(T) + w * PC + e;
                const double2 *E2 = reinterpret_cast<const double2 *>(sex) + e;
                const double2 *H2 = reinterpret_cast<const double2 *>(eth) + e;
                double sx = 0.0, sy = 0.0, lx = 0.0, ly = 0.0;
#pragma unroll
                for (int kk = 0; kk < QP; ++kk) {
                    const double2 t = T2[8 * kk], ev = E2[8 * kk], h = H2[8 * kk];
                    sx = fma(t.x, ev.x, sx); sy = fma(t.y, ev.y, sy);    // np.sum(a, 0), a = beta * exp(eta~)
                    lx = fma(t.x, h.x, lx); ly = fma(t.y, h.y, ly);      // theta @ a
                }
                double Sw = sx + sy, Lw = lx + ly;
                Sw += dpp_move<DPP_XOR1>(Sw); Sw += dpp_move<DPP_XOR2>(Sw); Sw += dpp_move<DPP_HALF_MIRROR>(Sw);
                Lw += dpp_move<DPP_XOR1>(Lw); Lw += dpp_move<DPP_XOR2>(Lw); Lw += dpp_move<DPP_HALF_MIRROR>(Lw);
                const bool valid = w < nw, own = e == 0;
                const double c = valid ? c0 : 0.0, sq = sqrt(c);
                const double wq = valid ? sq / Sw : 0.0;   // sqrt(c) / colsum: update_z, stm.py:1115, and the factor of b, stm.py:1001
                // c * log(theta @ a) (stm.py:1095): the eight lanes of a word all hold its Lw -- lane e keeps the one of every
                // eighth tile, and the logarithm is taken once per eight tiles over 64 distinct (word, tile) pairs
                {
                    const bool mine = ((t0 / TW) & 7) == e;
                    Lst = mine ? (valid ? Lw : 1.0) : Lst;
                    cst = mine ? c : cst;
                    if (((t0 / TW) & 7) == 7) {   // uniform
                        ll += cst * log_pos(Lst);
                        Lst = 1.0; cst = 0.0;
                    }
                }
                csum += own ? c : 0.0;
                if (own) *reinterpret_cast<double2 *>(wpar + 2 * w) = make_double2(wq, sq);
                // phi = beta * theta * r (stm_betass.h): r = exp-sum * c / S, in update_z's association (sqrt(c) / S) * sqrt(c).
                // assert np.all(phi >= 0) (stm.py:1117) fails exactly when a column sum is 0 (0 * inf), infinite or NaN.
                if (valid && own) P.rw[sl0] = (wq * sq) * sumex;
                sbad |= valid && !(Sw > 0.0 && Sw < INFINITY);
            }
            STM_WG_SYNC();
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[1] += cy - cy0; cy0 = cy; }
            // -- 2. rowsum(c'), T <- b (lane = topic gl): b = a * (sqrt(c) / S) (stm.py:1001), rowsum(c') of stm.py:1002,1011
            if (isk) {
                double *tc = T + gl;
                const double2 *wp2 = reinterpret_cast<const double2 *>(wpar);
#pragma unroll
                for (int w0 = 0; w0 < TW; w0 += 4) {
                    double t[4];
                    double2 wp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { t[u] = tc[(w0 + u) * PITCH]; wp[u] = wp2[w0 + u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double b = (t[u] * ex) * wp[u].x;
                        const double ph = b * wp[u].y;
                        tc[(w0 + u) * PITCH] = b;
                        rowc += ph;
                    }
                }
            }
            STM_WG_SYNC();
            if (dump_phi && isk)   // the reference keeps the last document's phi (stm.py:1116)
                for (int w = 0; w < nw; ++w) P.phi_out[(size_t)gl * Nd + t0 + w] = T[w * PITCH + gl] * wpar[2 * w + 1];
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[2] += cy - cy0; cy0 = cy; }
            // -- 3. b b^T on the matrix cores, upper block triangle; tile t of the row-major enumeration belongs to wave t & 1
            {
                // (the fragments of four words at a time, the next four in flight behind the matrix-core work: all sixteen
                // words' at once do not fit next to this wave's accumulator tiles)
                const double *tr = T + fq * PITCH + fr;
                auto tiles = [&](auto wc) __attribute__((always_inline)) {
                    constexpr int W = decltype(wc)::value;
                    double f[2][NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) f[0][b] = tr[16 * b];
#pragma unroll
                    for (int s = 0; s < TW / 4; ++s) {
                        if (4 * s >= nw) break;   // uniform: groups of four words beyond the document are rows of zeros
                        if (s + 1 < TW / 4) {
#pragma unroll
                            for (int b = 0; b < NB; ++b) f[(s + 1) & 1][b] = tr[4 * (s + 1) * PITCH + 16 * b];
                        }
                        int t = 0;
#pragma unroll
                        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                            for (int bj = bi; bj < NB; ++bj, ++t)
                                if (t % NWV == W) acc[t / NWV] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s & 1][bi], f[s & 1][bj], acc[t / NWV], 0, 0, 0);
                    }
                };
                by_wave<0, NWV>(wv, tiles);
            }
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[3] += cy - cy0; }
            idx0 = idx1; c0 = c1; sl0 = sl1; idx1 = idx2; c1 = c2; sl1 = sl2;
        }
        ll += cst * log_pos(Lst);   // the pairs of the last, incomplete batch of tiles (lanes without one: 0 * log(1))
        if (DBG && P.prof && gl == 0) for (int q = 0; q < 4; ++q) P.prof[doc * PROF_SLOTS + 24 + q] = tq[q];
        if (DBG && P.prof) tp[2] = (long long)__builtin_readcyclecounter();
        relane();
        if (wave_any(sbad || (isk && !(rowc >= 0.0)))) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */);   // stm.py:1117
        {
            const double cs = wave_sum(csum), lls = wave_sum(ll);
            if (lane == 0 && rowwave) { xch[X_CSUM + wv] = cs; xch[X_LL + wv] = lls; }
        }
        STM_WG_SYNC();
        const double Ndoc = (double)(long long)(xch[X_CSUM] + xch[X_CSUM + 1]);
        ll = xch[X_LL] + xch[X_LL + 1];

        // ---- b b^T - N theta theta^T on the accumulator tiles, once (stm.py:1003-1006); the assembly below adds siginv
        {
            auto fold = [&](auto wc) __attribute__((always_inline)) {
                constexpr int W = decltype(wc)::value;
                int t = 0;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj, ++t)
                        if (t % NWV == W) {
                            const double thj = sth[bj * 16 + fr];
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[t / NWV][r] = acc[t / NWV][r] - Ndoc * (sth[bi * 16 + fq + 4 * r] * thj);
                        }
            };
            by_wave<0, NWV>(wv, fold);
        }
        STM_WG_SYNC();   // the tiles and the vectors of region 0 are done with: the matrix takes their place

        // ---- H (+ siginv off the diagonal) from the accumulator tiles into the packed lower triangle: element (i, j), i <= j,
        // of the upper block triangle is stored as M[j][i].  The diagonal cells get the raw b b^T - N theta^2; the lane that
        // owns row i turns it into diagA below.  Run again (same registers, same operations, same bits) when a failed
        // factorisation has eaten the matrix.
        auto assemble = [&]() __attribute__((always_inline)) {
            auto part = [&](auto wc) __attribute__((always_inline)) {
                constexpr int W = decltype(wc)::value;
                int t = 0;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj, ++t)
                        if (t % NWV == W) {
                            const int j = bj * 16 + fr, jc = j < n ? j : nm1;
                            const int rsj = RS(jc);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = bi * 16 + fq + 4 * r, ic = i < n ? i : nm1;
                                double h = acc[t / NWV][r];
                                if (!sdiag && i != j) h += S[(size_t)ic * n + jc];
                                const bool st = j < n && (bi != bj || i <= j);
                                M[st ? rsj + i : MDUMP] = h;
                            }
                        }
            };
            by_wave<0, NWV>(wv, part);
        };

        double diagA = 1.0, Ldiag = 1.0;
        bool clean = false;
        int chol_calls = 0;
        long long tcc[4] = {0, 0, 0, 0};
        // np.linalg.cholesky, blocked by 16 columns and in place of the packed triangle.  Per panel: (a) the block column
        // minus the products of the finished panels on the matrix cores, block rows dealt out to the two waves; (b) the
        // panel with a lane holding its row's 16 entries in registers, right-looking and free of branches on the data.
        auto cholesky = [&]() __attribute__((always_inline)) -> bool {
            // a pivot never exceeds its diagonal entry: an entry <= 0 (or NaN) fails some pivot test for certain, and the
            // attempt is decided without factorising.  (The diagonal cells of a clean M are not read again: diagA is.)
            const bool neg = wave_any(isn && !(diagA > 0.0));
            const bool slow = wave_any(isn && !(diagA > 1e-260 && diagA < 1e270));   // (see sqrt_and_rsqrt_pivot)
            double *flag = xch + X_FLAG + 2 * (chol_calls & 1);
            ++chol_calls;
            if (lane == 0 && rowwave) flag[wv] = (neg ? 1.0 : 0.0) + (slow ? 2.0 : 0.0);
            if (isn) M[RS(gl) + gl] = diagA;
            STM_WG_SYNC();
            const double f0 = flag[0], f1 = flag[1];
            if (f0 == 1.0 || f0 == 3.0 || f1 == 1.0 || f1 == 3.0) return false;
            const bool fast = f0 == 0.0 && f1 == 0.0;
            clean = false;
            bool ok = true;
            __builtin_amdgcn_s_setprio(2);   // the factorisation is the document's longest dependent stretch (updates -> panel -> updates ...)
#pragma unroll 1
            for (int p = 0; p < NB && ok; ++p) {
                const int J0 = 16 * p;
                relane();
                long long cq = (DBG && P.prof) ? (long long)__builtin_readcyclecounter() : 0;
                if (p > 0) {
                    const int bc = J0 + fr, bcc = bc < n ? bc : nm1;
                    const double *brow = M + RS(bcc);                       // row of L_p* for the B operands (L_pk^T)
#pragma unroll 1
                    for (int bi = p + wv; bi < NB; bi += NWV) {
                        const int ar = bi * 16 + fr, arc = ar < n ? ar : nm1;
                        const double *arow = M + RS(arc);
                        int dst[4];
                        double old[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = bi * 16 + fq + 4 * r;
                            dst[r] = (i >= bc && i < n) ? RS(i) + bc : MDUMP;   // lower triangle incl. the diagonal (bc <= i < n)
                            old[r] = M[dst[r]];
                        }
                        v4d a = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                        for (int k = 0; k < p; ++k) {
                            double av[4], bv[4];
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) {
                                const int kk = k * 16 + 4 * sk + fq;
                                av[sk] = arow[kk]; bv[sk] = brow[kk];
                            }
#pragma unroll
                            for (int sk = 0; sk < 4; ++sk) a = __builtin_amdgcn_mfma_f64_16x16x4f64(av[sk], bv[sk], a, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) M[dst[r]] = old[r] - a[r];
                    }
                    STM_WG_SYNC();
                }
                if (DBG && P.prof) { const long long c1 = __builtin_readcyclecounter(); tcc[0] += c1 - cq; cq = c1; }
                // (b) the wave that owns rows J0 .. J0 + 15 (wave J0 >> 6) and, when that is wave 0, wave 1 as well -- with a
                // shadow of those rows in its lanes 48 .. 63.  Rows above the panel shadow its first row, rows beyond n the
                // last one (never stored).
                const int wp = J0 >> 6;
                if (wv >= wp && rowwave) {
                    const bool owner = wv == wp;
                    const int pl0 = owner ? (J0 & 63) : 48;                       // lane of the panel's first row
                    const bool shadow = !owner && lane >= 48;
                    const int row = shadow ? J0 + lane - 48 : gl;                // the row this lane carries
                    const int ic = row < J0 ? J0 : (row < n ? row : nm1);
                    const double2 *wr = reinterpret_cast<const double2 *>(M + RS(ic) + J0);
                    double w[16];
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2) { const double2 t = wr[c2]; w[2 * c2] = t.x; w[2 * c2 + 1] = t.y; }
                    // Column J's finished entries reach the other lanes through a 16-entry LDS column of this wave's own
                    // (one store by the diagonal block's lanes, then broadcast reads), one column behind: the update with
                    // column J-1 is applied while column J's pivot chain runs.  That chain does not wait for it: the lane that
                    // owns row J computes its fully updated diagonal entry w[j] - L[J][J-1]^2 from its own registers.
                    bool badl = false;
                    if (DBG && P.prof) { pin(w[0]); pin(w[15]); const long long c1 = __builtin_readcyclecounter(); tcc[1] += c1 - cq; cq = c1; }
                    const bool indiag = (unsigned)(lane - pl0) < 16u;
                    double *cbw = cb + (indiag ? lane - pl0 : 0);
                    __builtin_amdgcn_s_setprio(3);   // the pivot chain: ahead of the SIMD's other wave (another document's)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (J0 + j < n) {   // uniform
                            const int pl = pl0 + j;
                            const double tmp = j > 0 ? fma(-w[j > 0 ? j - 1 : 0], w[j > 0 ? j - 1 : 0], w[j]) : w[0];
                            const double d = lane_bcast(tmp, pl);
                            badl |= owner && (lane == pl) && !(tmp > PIVOT_TOL * diagA);   // see PIVOT_TOL
                            double ljj, rjj;                        // LAPACK dpotf2 scales the column by the reciprocal as well
                            if (fast) sqrt_and_rsqrt_pivot(d, ljj, rjj); else sqrt_and_rsqrt(d, ljj, rjj);   // (uniform)
                            if (owner && lane == pl) Ldiag = ljj;
                            if (j > 0) {   // the update with column J - 1 (stored at the end of the previous step), four broadcasts at a time
#pragma unroll
                                for (int c0 = j; c0 < 16; c0 += 4) {
                                    double xs[4];
#pragma unroll
                                    for (int u = 0; u < 4; ++u) if (c0 + u < 16) xs[u] = cb[c0 + u];   // L[J0 + c][J - 1]
#pragma unroll
                                    for (int u = 0; u < 4; ++u) if (c0 + u < 16) w[c0 + u] = fma(-w[j > 0 ? j - 1 : 0], xs[u], w[c0 + u]);
                                }
                            }
                            w[j] *= rjj;
                            if (j < 15) {
                                STM_POST_SYNC();
                                if (indiag) *cbw = w[j];
                                STM_POST_SYNC();
                            }
                        }
                    }
                    __builtin_amdgcn_s_setprio(2);
                    if (owner) {
                        const bool bad = wave_any(badl);
                        if (lane == 0) xch[X_BAD] = bad ? 1.0 : 0.0;
                    }
                    if (DBG && P.prof) { pin(w[15]); const long long c1 = __builtin_readcyclecounter(); tcc[2] += c1 - cq; cq = c1; }
                    // pairs (c, c + 1) with the first cell strictly below the diagonal; the second one is then at most the
                    // diagonal cell, which is free (it takes X's diagonal later).  A failed panel stores rubbish: the ladder
                    // re-assembles.
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2) {
                        const int c = 2 * c2;
                        const bool st = !shadow && row < n && row > J0 + c;
                        *reinterpret_cast<double2 *>(M + (st ? RS(row) + J0 + c : MDUMP)) = make_double2(w[c], w[c + 1]);
                    }
                }
                STM_WG_SYNC();
                if (xch[X_BAD] != 0.0) ok = false;
                if (DBG && P.prof) { const long long c1 = __builtin_readcyclecounter(); tcc[3] += c1 - cq; }
            }
            __builtin_amdgcn_s_setprio(0);
            return ok;
        };
        auto make_pd = [&]() __attribute__((always_inline)) {  // stm.py:964-984; M holds A (clean)
            if (isn) {
                double mag = 0.0;
                const double *ri = M + RS(gl);
                for (int j0 = 0; j0 < n; j0 += 4) {
                    double lo[4], up[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u < n ? j0 + u : nm1;
                        lo[u] = ri[j < gl ? j : 0]; up[u] = M[RS(j) + (j > gl ? gl : 0)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        const double aij = (j == gl) ? diagA : (j < gl ? lo[u] : up[u]);
                        mag += j < n ? fabs(aij) : 0.0;
                    }
                }
                mag -= fabs(diagA);
                if (diagA < mag) diagA = mag;
            }
        };
        auto dump_hess = [&]() {
            double *o = P.hess_out + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j)
                    o[(size_t)gl * n + j] = (j == gl) ? diagA : (j < gl ? M[RS(gl) + j] : M[RS(j) + gl]);
        };

        // One assembly site and one Cholesky site for every stage of the reference's PD ladder:
        //   0 hessian(): PD test as Cholesky success (stm.py:1017)   1 after make_pd (stm.py:1019-1020)
        //   2 +1e-5 (stm.py:1021), decompose_hessian's np.linalg.cholesky (stm.py:1040)
        //   3 after make_pd (stm.py:1043)   4 scipy cholesky (UPPER) of make_pd(H) + 1e-5 I (stm.py:1046-1048)
        int path = 0;
        bool upper = false, fail = false;
        double keep = 0.0;
        for (int attempt = 0;; ++attempt) {
            relane();
            if (!clean) {
                assemble();
                STM_WG_SYNC();
                clean = true;
                if (attempt == 0 && isn) {
                    const double sii = S[(size_t)gl * n + gl];
                    diagA = ((M[RS(gl) + gl] - rowc) + Ndoc * ths) + sii;   // stm.py:1003-1013, in this order
                }
                if (DBG && P.prof && attempt == 0) tp[3] = (long long)__builtin_readcyclecounter();
            }
            // what the previous, failed attempt asks for
            if (attempt == 1) { make_pd(); path = 1; }
            else if (attempt == 2) { if (isn) diagA += 1e-5; path = 2; }
            else if (attempt == 3) { make_pd(); }
            else if (attempt == 4) { make_pd(); keep = diagA; if (isn) diagA += 1e-5; }
            if (DBG && P.hess_out && attempt <= 2) dump_hess();
            // (make_pd and the dump read M, the Cholesky starts by writing its diagonal: the cells it writes are read as
            // don't-care operands of selects only, and the first block-column update comes after a barrier)
            const bool ok = cholesky();
            if (attempt == 4) { diagA = keep; upper = true; fail = !ok; break; }
            if (ok) break;
        }
        if (P.pd_path && gl == 0) P.pd_path[doc] = path;
        if (fail) {
            if (gl == 0) atomicMax(P.err_flag, 3 /* STM_ERR_LINALG */);
            continue;
        }
        if (DBG && P.chol_out) {
            double *o = P.chol_out + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j) {
                    const double val = (j == gl) ? Ldiag : (j < gl ? M[RS(gl) + j] : 0.0);
                    if (upper) o[(size_t)j * n + gl] = val;  // the reference holds the upper factor here
                    else o[(size_t)gl * n + j] = val;
                }
        }

        if (DBG && P.prof) tp[4] = (long long)__builtin_readcyclecounter();
        // ---- bound (stm.py:1068-1101); X's diagonal 1 / L_ii goes to the triangle's free diagonal cells
        {
            const double det = wave_sum(isn ? log(Ldiag) : 0.0);
            if (lane == 0 && rowwave) xch[X_DET + wv] = det;
        }
        const double Rdiag = 1.0 / Ldiag;
        if (isn) M[RS(gl) + gl] = Rdiag;
        STM_WG_SYNC();
        if (gl == 0) {
            const double det = xch[X_DET] + xch[X_DET + 1], q = xch[X_Q] + xch[X_Q + 1];
            P.bound[doc] = ll + (-det) - 0.5 * q - P.sigmaentropy;
        }

        if (DBG && P.prof) tp[5] = (long long)__builtin_readcyclecounter();
        relane();
        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066)
        long long ti[3] = {0, 0, 0};
        if (DBG && P.prof) ti[0] = (long long)__builtin_readcyclecounter();
        if (!upper) {
            // X = L^-1 (so that nu = X^T X), blocked by 16 and IN PLACE of L.
            // (I) all diagonal blocks at once, lane = (block gl >> 4, column c), the column in registers:
            //     x[i] = X[i][c] = -(sum_{l<i} L[i][l] x[l]) / L[i][i]   (x[l] = 0 above the diagonal, x[c] = 1 / L[c][c]);
            //     the rows of L are independent of x, so their loads run ahead of the substitution chain, and every store
            //     comes after every load (one instruction stream per wave, and the two waves own different blocks).
            __builtin_amdgcn_s_setprio(2);   // substitution chains, then dependent MFMA chains: ahead of the SIMD's other wave
            if (rowwave) {
                const int c = gl & 15, rb = gl & ~15;
                const bool has = rb < n;
                const int rbc = has ? rb : 0;                         // lanes beyond the matrix shadow block 0 (nothing is stored)
                const int rows = n - rbc < 16 ? n - rbc : 16;
                // row rbc + i of the block starts at RS(rbc + i) + rbc = RS(rbc) + rbc + i rbc + RS(i) (rbc is even); its
                // entry i is the diagonal cell (1 / L_ii); rows beyond the matrix shadow the block's last row
                const int base0 = RS(rbc) + rbc, lastoff = base0 + (rows - 1) * rbc + RS(rows - 1);
                auto row_of = [&](int i) __attribute__((always_inline)) {
                    return reinterpret_cast<const double2 *>(M + (i < rows ? base0 + i * rbc + tri_row(i) : lastoff));
                };
                double x[16];
                double2 buf[2][8];   // the row after next is fetched while a row is consumed
                {
                    const double2 r0 = row_of(0)[0];
                    x[0] = (c == 0) ? r0.x : -0.0;
                }
                { const double2 *r1 = row_of(1); buf[1][0] = r1[0]; }
#pragma unroll
                for (int i = 1; i < 16; ++i) {
                    if (i + 1 < 16) {
                        const double2 *rn = row_of(i + 1);
#pragma unroll
                        for (int l2 = 0; 2 * l2 < i + 2; ++l2) buf[(i + 1) & 1][l2] = rn[l2];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double t0 = 0.0, t1 = 0.0;
#pragma unroll
                    for (int l2 = 0; 2 * l2 < i; ++l2) {
                        const double2 lv = buf[i & 1][l2];
                        t0 = fma(lv.x, x[2 * l2], t0);
                        if (2 * l2 + 1 < i) t1 = fma(lv.y, x[2 * l2 + 1], t1);
                    }
                    const double2 rdp = buf[i & 1][i >> 1];
                    const double rd = (i & 1) ? rdp.y : rdp.x;          // the row's diagonal cell
                    x[i] = (i == c) ? rd : -(t0 + t1) * rd;
                    pin(x[i]);     // the substitution step stays between the two fetches (instruction selection would sink it)
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool st = has && i >= c && i < rows;
                    M[st ? base0 + i * rbc + tri_row(i) + c : MDUMP] = x[i];
                }
            }
            STM_WG_SYNC();
            if (DBG && P.prof) ti[1] = (long long)__builtin_readcyclecounter();
            // (II) X_rc = -X_rr (sum_{c<=k<r} L_rk X_kc) on the matrix cores.  The block columns of X are independent of each
            //      other given L -- column c reads L_rk, k >= c, and its own X_kc -- so two columns are formed at a time, one
            //      per wave, top down, and kept in REGISTERS: a finished X_kc leaves the MFMA in exactly the register layout
            //      the B operand of the rows below wants, so it never visits the LDS, and nothing of L is overwritten while
            //      either wave still reads it.  Both waves then store their column over L's (barrier before and after).
            //      Three rounds at NB = 7; the longer column of a pair alternates between the waves.
            {
                v4d xcol[NB];
#pragma unroll 1
                for (int c0 = 0; c0 + 1 < NB; c0 += NWV) {
                    const int c = c0 + (NWV == 2 ? (wv ^ ((c0 >> 1) & 1)) : wv);
                    const int bc = c * 16 + fr;
                    if (c + 1 < NB) {
                        // k outermost: step k first closes X_kc (its sum is complete: the rows above are done), then adds L_rk X_kc to the
                        // sums of ALL rows r > k -- their fragment loads go out together, one LDS round trip per k instead of one per
                        // (r, k) (21 of them for column 0 at NB = 7, each in front of four dependent matrix-core instructions).  Every
                        // row's sum still receives its terms in the order k = c, c + 1, ... (four products each): the same bits.
                        // xcol[r] holds row r's running sum until step r turns it into X_rc.
#pragma unroll
                        for (int r = 0; r < NB; ++r) xcol[r] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int k = 0; k < NB; ++k) {
                            if (k >= c) {   // uniform
                                double bop[4];                                  // B operand of this step: X_kc[4 sk + fq][fr]
                                if (k == c) {   // X_cc: the lower-triangular diagonal block, from the LDS
#pragma unroll
                                    for (int sk = 0; sk < 4; ++sk) {
                                        const int kk = k * 16 + 4 * sk + fq;
                                        const double bv = M[RS(kk) + bc];
                                        bop[sk] = (bc <= kk) ? bv : 0.0;
                                    }
                                } else {        // X_kc = -X_kk (sum): the sum left the matrix cores in the B operand's layout
                                    const int ar = k * 16 + fr, arc = ar < n ? ar : nm1;
                                    const double *arow = M + RS(arc);
                                    double xv[4];
#pragma unroll
                                    for (int sk = 0; sk < 4; ++sk) {
                                        const int ac = k * 16 + 4 * sk + fq;
                                        xv[sk] = arow[ac < arc ? ac : arc];                 // X_kk[fr][4 sk + fq], at most the diagonal cell
                                    }
                                    v4d dacc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                                    for (int sk = 0; sk < 4; ++sk) {
                                        const int ac = k * 16 + 4 * sk + fq;
                                        dacc = __builtin_amdgcn_mfma_f64_16x16x4f64((ac <= ar && ar < n) ? xv[sk] : 0.0, xcol[k][sk], dacc, 0, 0, 0);
                                    }
                                    xcol[k] = -dacc;
#pragma unroll
                                    for (int sk = 0; sk < 4; ++sk) bop[sk] = xcol[k][sk];
                                }
                                if (k + 1 < NB) {
                                    double av[NB][4];
#pragma unroll
                                    for (int r = k + 1; r < NB; ++r) {
                                        const int ar = r * 16 + fr, arc = ar < n ? ar : nm1;
                                        const double *arow = M + RS(arc);                 // row of L_r* for the A operands
#pragma unroll
                                        for (int sk = 0; sk < 4; ++sk) av[r][sk] = arow[k * 16 + 4 * sk + fq];   // L_rk[fr][4 sk + fq] (k < NB - 1: full blocks)
                                    }
#pragma unroll
                                    for (int r = k + 1; r < NB; ++r) {
                                        const int ar = r * 16 + fr;
#pragma unroll
                                        for (int sk = 0; sk < 4; ++sk)
                                            xcol[r] = __builtin_amdgcn_mfma_f64_16x16x4f64((ar < n) ? av[r][sk] : 0.0, bop[sk], xcol[r], 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    STM_WG_SYNC();   // both columns are complete: nobody reads L's columns c0, c0 + 1 below the diagonal blocks any more
                    if (c + 1 < NB) {
#pragma unroll
                        for (int r = 1; r < NB; ++r) {
                            if (r > c) {   // uniform
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int row = r * 16 + fq + 4 * q;
                                    M[row < n ? RS(row) + bc : MDUMP] = xcol[r][q];
                                }
                            }
                        }
                    }
                    STM_WG_SYNC();
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (DBG && P.prof) ti[2] = (long long)__builtin_readcyclecounter();
        }
        if (DBG && P.prof) tp[6] = (long long)__builtin_readcyclecounter();
        relane();
        if (DBG && P.prof && gl == 0 && !upper) { P.prof[doc * PROF_SLOTS + 28] = ti[1] - ti[0]; P.prof[doc * PROF_SLOTS + 29] = ti[2] - ti[1]; }
        if (DBG && P.prof && gl == 0) { P.prof[doc * PROF_SLOTS + 30] = tcc[0]; P.prof[doc * PROF_SLOTS + 31] = tcc[2]; P.prof[doc * PROF_SLOTS + 23] = tcc[1] + tcc[3]; }
        // nu = R R^T = X^T X (sigma_ss += nu, stm.py:582), one block column bj of output tiles (b <= bj) at a time on the
        // matrix cores, block columns dealt out to the waves: nu[i][j] = sum_{l >= 16 bj} X[l][i] X[l][j] (X is lower
        // triangular); fragment X[s4 + fq][b * 16 + fr], zero above the diagonal.  The workgroup's running sum lives in a slab
        // of its own in the accumulators' register layout, added to with no-return atomics (a cell has ONE writer: its order of
        // additions is the order of the workgroup's documents).  Cells beyond n are exact zeros.
        double *nu_doc = (DBG && P.nu_out) ? P.nu_out + (size_t)doc * n * n : nullptr;
        if (upper) {   // nu = diag(1 / L_ii^2): element (i, i) sits in tile (b, b) at r = ((i & 15) - fq) / 4, lane = (fq, fr = i & 15)
            for (int bb = wv; bb < NB; bb += NWV) {
                const int i = bb * 16 + fr, r = (fr - fq) >> 2;
                if (((fr - fq) & 3) == 0 && fr >= fq && i < n) {
                    const double rdi = M[RS(i) + i];
                    const double v = rdi * rdi;
                    sig_acc[((size_t)(bb * (bb + 1) / 2 + bb) * 4 + r) * WAVE + lane] += v;
                    if (DBG && nu_doc)
                        for (int j = 0; j < n; ++j) nu_doc[(size_t)i * n + j] = (j == i) ? v : 0.0;
                }
            }
        } else {
            // Row quads outermost, like b b^T: the fragments X[s4 + fq][b 16 + fr], b <= R, of a quad in block row R are read ONCE and serve
            // every tile (b, bj), b <= bj <= R -- (R + 1)(R + 2) / 2 products per R + 1 loads.  (One block column at a time, as in the
            // K <= 64 kernel, read the fragments again for every block column: 640 loads and 91 load-then-wait steps per document
            // at NB = 7 for 252 products.)  Tile (b, bj) is slot bj (bj + 1) / 2 + b of the slab and belongs to wave slot & 1; it still
            // receives its products in the order s4 = 16 bj, 16 bj + 4, ...: the same bits.
            auto nu_tiles = [&](auto wc) __attribute__((always_inline)) {
                constexpr int W = decltype(wc)::value;
                v4d an[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) an[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int R = 0; R < NB; ++R) {
                    const int rjR = R * 16 + fr;
#pragma unroll 1
                    for (int s4 = 16 * R; s4 < 16 * R + 16 && s4 < n; s4 += 4) {
                        const int col = s4 + fq, colc = col < n ? col : nm1;
                        const double *xr = M + RS(colc);
                        double g[R + 1];
#pragma unroll
                        for (int bb = 0; bb < R; ++bb) g[bb] = xr[bb * 16 + fr];
                        g[R] = xr[rjR < n ? rjR : nm1];
#pragma unroll
                        for (int bb = 0; bb < R; ++bb) g[bb] = (col < n) ? g[bb] : 0.0;
                        g[R] = (col < n && rjR < n && col >= rjR) ? g[R] : 0.0;      // X is lower triangular
#pragma unroll
                        for (int bj = 0; bj <= R; ++bj)
#pragma unroll
                            for (int bb = 0; bb <= bj; ++bb) {
                                if ((bj * (bj + 1) / 2 + bb) % NWV == W)
                                    an[(bj * (bj + 1) / 2 + bb) / NWV] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[bb], g[bj], an[(bj * (bj + 1) / 2 + bb) / NWV], 0, 0, 0);
                            }
                    }
                }
#pragma unroll
                for (int bj = 0; bj < NB; ++bj)
#pragma unroll
                    for (int bb = 0; bb <= bj; ++bb) {
                        const int t = bj * (bj + 1) / 2 + bb;
                        if (t % NWV != W) continue;
                        double *slab = sig_acc + (size_t)t * 4 * WAVE + lane;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // fire-and-forget: the cell belongs to this wave alone (no contention, program order from one document
                            // to the next), so the sum is the same every run -- and no old value has to be fetched and held
                            unsafeAtomicAdd(slab + r * WAVE, an[t / NWV][r]);
                            if (DBG && nu_doc) {
                                const int i = bb * 16 + fq + 4 * r, j = bj * 16 + fr;
                                if (i < n && j < n) {
                                    nu_doc[(size_t)i * n + j] = an[t / NWV][r];
                                    nu_doc[(size_t)j * n + i] = an[t / NWV][r];
                                }
                            }
                        }
                    }
            };
            by_wave<0, NWV>(wv, nu_tiles);
        }
        if (DBG && P.prof && gl == 0) {
            tp[7] = (long long)__builtin_readcyclecounter();
            for (int q2 = 0; q2 < 7; ++q2) P.prof[doc * PROF_SLOTS + 32 + q2] = tp[q2 + 1] - tp[q2];
        }
    }
}

}  // namespace stm
