// stm_post.h -- per-document theta / Hessian / Cholesky / nu / bound / phi after the solve, K <= 64.
//
// Replaces, per document, reference src/modules/stm.py:547-588:
//   theta (547-549), hessian (986-1026, incl. make_pd 964-984 and the +1e-5 branch),
//   decompose_hessian (1031-1050), lower_bound (1068-1101), optimize_nu (1052-1066),
//   update_z (1103-1118) and the sigma_ss / beta_ss accumulation (582-588).
//
// One wavefront per workgroup, PERSISTENT over a strided set of documents, and built to be resident ELEVEN to a CU
// (three waves per SIMD by registers: 156 VGPRs, no scratch; 14 KB of LDS at K = 50 with the double-buffered word tile) --
// every phase of a document is a chain of dependent instructions, and what hides one wave's latencies is the other waves
// of its SIMD.  The kernel issues NO atomics.
//
// Words are processed in tiles of 16, word-major in LDS (T[word][topic], exactly the layout of betaT's rows), two tile
// buffers:
//   0. fetch    the tile's 16 beta rows go from betaT[A][V][K] STRAIGHT INTO LDS (global_load_lds_dwordx4: lane l of
//               instruction q moves 16-byte chunk 64 q + l of the tile) -- no staging registers, no transposition; the
//               fetch of tile t+1 is issued at the top of tile t's round, the word ids / counts / slots of tile t+2 with it
//   1. sums     lane = (word, quarter of the topics): colsum S_w and theta.(beta*exp(eta)) per word; the four quarters sit
//               in one DPP quad; log / sqrt / division once per word.  r_dw = (sum_k exp(eta~)_k) c / S_w -- all that
//               beta_ss needs of this (document, word), see stm_betass.h -- is stored to the entry's word-major slot
//   2. scatter  lane = topic: T is overwritten with b = a*sqrt(c)/S (phi itself is only formed for the one document whose
//               phi the reference keeps, stm.py:1116)
//   3. b b^T    fp64 MFMA (v_mfma_f64_16x16x4_f64): the K x K contraction, upper block triangle only, accumulated in
//               registers over all tiles of the document
// The (K-1)^2 matrix then lives in LDS as its LOWER TRIANGLE ONLY, row-packed (rows padded to an even length: 16-byte
// rows), aliasing the tiles: the blocked Cholesky (block-column updates on the matrix cores, 16-column panels factorised
// right-looking in registers, the pivot column broadcast through a 64-entry LDS vector) overwrites it with L, the blocked
// in-place inverse with X = L^-1, and nu = X^T X = H^-1 is a Gram product again (matrix cores, only the tiles and the
// k-range the triangular shape leaves), added per document into the workgroup's OWN slab of sigma_ss partials in the
// accumulator-tile layout (plain read-modify-write, prefetched; reduce_sigma_kernel adds the slabs in a fixed order, then
// untile_sigma_kernel lays the matrix out).  A failed rung of the reference's PD ladder finds A again by re-running the
// assembly from the b b^T accumulators, which stay in registers until the ladder is through.
#pragma once
#include "stm_post_common.h"

// Ablation builds (tools/ablate.sh, never shipped): -DSTM_ABLATE=<bits> removes one phase of post_kernel at compile time -- the results are
// garbage, the PD ladder is forced to accept and no error is raised -- to measure what each phase costs the kernel UNDER CONTENTION
// (eleven waves per CU: a phase's own cycle count says how long it took, not what removing it would buy).
//   1 per-word sums   2 b b^T (fragments + matrix cores)   4 lane = topic pass (row sums, remainder row)   8 factorisation
//   16 inverse   32 nu   64 the tile fetches   128 assembly   256 nothing (the baseline with the ladder forced)
#ifndef STM_ABLATE
#define STM_ABLATE 0
#endif

namespace stm {

// start of row i of the row-packed lower triangle (row i: i + 1 cells, padded to an even count)
__host__ __device__ inline int tri_row(int i) { return 2 * ((i + 1) >> 1) * ((i >> 1) + 1); }

// LDS map (doubles).  Region 0 is the word tile + its per-word / per-topic vectors during the word loop and the packed
// matrix afterwards; one 64-entry per-topic vector (theta -> eta - mu -> 1 / diag L) lives behind it.  The tile's row
// pitch is a property of the instantiation: NB blocks serve K <= 16 NB + 2 topics, i.e. rows of at most 8 NB + 1 16-byte
// chunks -- every tile offset is an immediate (K = 50: NB = 3, 25 chunks, exactly betaT's row).
struct PostLds {
    int pitch;   // tile row pitch in doubles: 16 NB + 2
    int qp;      // 16-byte chunks per quarter of a row in the per-word sums: ceil((8 NB + 1) / 4)
    int tile;    // doubles per tile buffer (two of them: the fetch of tile t+1 is issued before tile t is touched)
    int zp;      // zeros behind the second tile (the last quarter reads a few chunks past the last row)
    int wpar;    // per word { sqrt(c) / S, sqrt(c) }
    int sex;     // exp(eta~) per topic, zeros from K on (8 qp entries)
    int eth;     // exp(eta~) * stable_softmax(eta~) per topic, zeros from K on
    int mdump;   // two cells behind the matrix (+ 16 of slack) that masked stores go to
    int vec;     // the per-topic vector of the factor phases, behind the matrix
    int total;
};
__host__ __device__ constexpr int post_pitch(int NB) { return 16 * NB + 2; }
__host__ __device__ inline PostLds post_lds_map(int K, int NB) {
    PostLds L;
    const int n = K - 1;
    L.pitch = post_pitch(NB);
    L.qp = (8 * NB + 1 + 3) / 4;
    L.tile = TW * L.pitch + 8;          // + 8: what the last quarter of the first buffer's last row reads past it must be finite too
    L.zp = 2 * L.tile - 8;
    L.wpar = 2 * L.tile;
    L.sex = L.wpar + 2 * TW;
    L.eth = L.sex + 8 * L.qp;
    const int tile_part = L.eth + 8 * L.qp;
    L.mdump = tri_row(n) + 16;
    L.vec = L.mdump + 2;
    const int mat_part = L.vec + 64 + 64;   // + the Cholesky panel's column broadcast
    L.total = ((tile_part > mat_part ? tile_part : mat_part) + 1) & ~1;
    return L;
}

// NB 16 x 16 blocks cover the b b^T accumulation on the matrix cores; REM == 1: n = 16 NB + 1 exactly (K = 50:
// 49 = 3 * 16 + 1), and the one row / column beyond the blocks is carried on the VALU (one double per lane) instead of
// padding to NB + 1 blocks -- 6 accumulator tiles instead of 10.  The factorisation, the inverse and nu work on
// NBC = ceil(n / 16) blocks either way (a one-row block costs them four matrix-core passes).
// WPE: waves per SIMD the registers are budgeted for.  DBG: per-document dumps, cycle counters, LDS poisoning.
template <int NB, int REM, int WPE, bool DBG>
__global__ __launch_bounds__(64, WPE) void post_kernel(PostParams P) {
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int R0 = 16 * NB;       // index of the remainder row (REM == 1)
    constexpr int NBC = NB + REM;
    constexpr int PITCH = post_pitch(NB), PC = PITCH / 2;   // tile row pitch in doubles / 16-byte chunks
    constexpr int QP = (PC + 3) / 4;                         // chunks per quarter row in the per-word sums
    constexpr int RPI = 64 / PC;                             // whole tile rows per LDS-DMA instruction (PC <= 33)
    constexpr int NQ = (TW + RPI - 1) / RPI;                 // LDS-DMA instructions per tile
    extern __shared__ __attribute__((aligned(16))) double post_lds[];
    int lane = threadIdx.x;
    const int K = P.K, n = P.n, nm1 = n - 1;
    const PostLds LM = post_lds_map(K, NB);
    const int CP = (K + 1) >> 1, MDUMP = LM.mdump;   // CP: chunks of a row that hold topics
    constexpr int TILE = TW * PITCH + 8;   // doubles per tile buffer (post_lds_map)
    double *M = post_lds;            // row-packed lower triangle (after the word loop)
    double *wpar = post_lds + LM.wpar, *sex = post_lds + LM.sex, *eth = post_lds + LM.eth;
    double *vec = post_lds + LM.vec;
    double *cb = vec + 64;           // the Cholesky panel's column broadcast
    double *sth = vec;               // word loop .. PD ladder: stable_softmax(eta~) (stm.py:998,1083)
    double *sdv = vec;               // bound: eta - mu broadcast (dense siginv only)
    double *srd = vec;               // inverse: 1 / diag(L)
    const double *S = P.siginv;
    const bool sdiag = P.siginv_diag != 0;
    // this workgroup's own sum of nu, tile layout: NBV (NBV + 1) / 2 tiles of the full blocks' upper triangle and, with REM, one
    // more 256-double slot whose first 64 entries hold the last column (nu[i][R0] in entry i)
    constexpr int NBV = REM ? NB : NBC, NU_TILES = NBV * (NBV + 1) / 2 + REM;
    double *sig_acc = P.sigma_part + (size_t)blockIdx.x * (size_t)NU_TILES * 4 * WAVE;
    bool isn = lane < n, isk = lane < K;
    int fr = lane & 15, fq = lane >> 4;  // MFMA fragment coordinates
    // The lane id is re-read behind an opaque move at the start of every phase: otherwise the lane-dependent LDS
    // addresses of the unrolled code are hoisted out of the document loop as invariants and live in scratch memory.
    auto relane = [&]() __attribute__((always_inline)) {
        int l = threadIdx.x;
        asm volatile("" : "+v"(l));
        lane = l; isn = l < n; isk = l < K; fr = l & 15; fq = l >> 4;
    };
    auto RS = [](int i) __attribute__((always_inline)) { return tri_row(i); };
    const unsigned K8 = 8u * (unsigned)K;

    for (int64_t tk = blockIdx.x; tk < P.count; tk += gridDim.x) {
        relane();
        if (DBG && (P.debug_flags & 16)) {   // nothing may depend on what an earlier document or kernel left in the LDS
            STM_POST_SYNC();
            for (int q = lane; q < P.lds_doubles; q += WAVE) post_lds[q] = __builtin_nan("");
            STM_POST_SYNC();
        }
        const int64_t ticket = P.first + tk;
        // the document header through the scalar cache (uniform, constant for the kernel's lifetime)
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

        // word ids (lane w < 16: word t0 + w) and counts (lane 4 w + q: word t0 + w) of a tile; lanes beyond the document
        // repeat its last word (a valid row for the fetch)
        auto load_ids = [&](int t0, int &idx, double &c, int &slot) __attribute__((always_inline)) {
            const int wi = t0 + lane, wc = t0 + (lane >> 2), last = Nd - 1;
            idx = P.indices[p0 + (wi < last ? wi : last)];
            c = P.counts[p0 + (wc < last ? wc : last)];   // masked where it is used: a select here would wait for the load at once
            slot = P.wm_slot[p0 + (wc < last ? wc : last)];   // where this word's r goes (stm_betass.h)
        };
        // the 16 rows of a tile, betaT -> LDS.  One instruction moves RPI WHOLE rows (lane l: 16-byte chunk l mod PC of the instruction's
        // row l / PC; the lanes beyond RPI rows are masked off), so a lane's (row, chunk) is the same for every instruction and every
        // tile -- the index arithmetic of a tile is one multiply-add per instruction (rounds 3-4 packed 64 chunks into every
        // instruction: one instruction fewer per tile at K = 50, eight vector instructions more per instruction).  Chunks beyond the
        // topics of a row (K below this instantiation's maximum) repeat its last one: finite, and the sums meet them with zeros.
        // All word ids first (one round trip through the crossbar), then the fetches back to back.
        auto tile_fetch = [&](int idxv, int buf) __attribute__((always_inline)) {
            unsigned off[NQ];
            int id[NQ];
            const int fr_ = (int)(((unsigned)lane * (65536u / PC + 1u)) >> 16);     // lane / PC (PC <= 33)
            int fo = lane - (int)__umul24((unsigned)fr_, (unsigned)PC);
            fo = fo < CP ? fo : CP - 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) id[q] = __builtin_amdgcn_ds_bpermute(4 * (q * RPI + fr_), idxv);   // (rows >= 16: masked lanes)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) off[q] = __umul24((unsigned)id[q], K8) + 16u * (unsigned)fo;
            // Hand-written, so that the compiler does not know these loads write the LDS: it would make every LDS read
            // behind them wait for ALL memory operations in flight (it cannot tell the two tile buffers apart) -- the wait
            // that matters is the counted one at the top of the tile loop.  M0 carries the LDS destination and exec the lanes that
            // hold a chunk (both saved and restored: the compiler owns them).
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(post_lds + buf * TILE);
            unsigned keep;
            unsigned long long ex;
#pragma unroll
            for (int q = 0; q < ((STM_ABLATE & 64) ? 0 : NQ); ++q) {
                constexpr int full = RPI * PC;
                const int rows = TW - q * RPI < RPI ? TW - q * RPI : RPI;      // (compile-time after unrolling)
                const unsigned long long mask = rows * PC >= 64 ? ~0ull : ((1ull << (rows * PC)) - 1ull);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %5\n\t"
                             "global_load_lds_dwordx4 %2, %4\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep), "=&s"(ex) : "v"(off[q]), "s"(lds0 + 16u * (unsigned)(full * q)), "s"(bT), "s"(mask) : "memory");
            }
        };

        int idx0, idx1, sl0, sl1;
        double c0, c1;
        load_ids(0, idx0, c0, sl0);
        load_ids(TW, idx1, c1, sl1);

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
        STM_POST_SYNC();  // the previous document's readers of region 0 / vec are done (one wave: the LDS works in order)
        wait_lds();
        if (lane < 8 * QP) {
            sex[lane] = ex;                         // zeros from K on
            eth[lane] = isk ? ex * ths : 0.0;       // theta . (beta * exp(eta~)) = sum_k beta_k (exp(eta~)_k theta_k), stm.py:1088-1094
        }
        if (8 * QP > WAVE && lane + WAVE < 8 * QP) { sex[lane + WAVE] = 0.0; eth[lane + WAVE] = 0.0; }
        if (lane < 16) post_lds[(lane >> 3) * TILE + TW * PITCH + (lane & 7)] = 0.0;   // the pads behind both tile buffers
        tile_fetch(idx0, 0);

        if (DBG && P.prof) tp[1] = (long long)__builtin_readcyclecounter();
        double csum = 0.0, ll = 0.0, rowc = 0.0;
        double Lst = 1.0, cst = 0.0;   // a (word, tile) pair's theta @ a and count waiting for the next batched logarithm
        bool sbad = false;
        v4d acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
        double hrem = 0.0;   // REM: (b b^T)[lane][R0]
        long long tq[4] = {0, 0, 0, 0};

        const bool dump_phi = P.phi_out && doc == P.phi_doc;
        for (int t0 = 0, buf = 0; t0 < Nd; t0 += TW, buf ^= 1) {
            const int nw = Nd - t0 < TW ? Nd - t0 : TW;
            long long cy0 = (DBG && P.prof) ? (long long)__builtin_readcyclecounter() : 0;
            relane();
            // The tile (fetched a whole tile ago) and the word ids have landed; nothing slow is in flight behind them -- this
            // kernel issues no atomics.
            wait_vmem();
            STM_POST_SYNC();
            double *T = post_lds + buf * TILE;
            if (t0 + TW < Nd) tile_fetch(idx1, buf ^ 1);   // the other buffer's readers finished a tile ago
            int idx2, sl2;
            double c2;
            load_ids(t0 + 2 * TW, idx2, c2, sl2);         // plain loads: the compiler waits for them where they are first read
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[0] += cy - cy0; cy0 = cy; }
            // -- 1. per-word sums, lane = (word lane >> 2, quarter lane & 3 of the row; zeros of sex / eth from K on)
            {
                const int w = lane >> 2, q = lane & 3;
                const double2 *T2 = reinterpret_cast<const double2 *>(T) + w * PC + q * QP;
                const double2 *E2 = reinterpret_cast<const double2 *>(sex) + q * QP;
                const double2 *H2 = reinterpret_cast<const double2 *>(eth) + q * QP;
                double sx = (STM_ABLATE & 1) ? 1.0 : 0.0, sy = 0.0, lx = sx, ly = 0.0;
#pragma unroll
                for (int kk = 0; kk < ((STM_ABLATE & 1) ? 0 : QP); ++kk) {
                    const double2 t = T2[kk], e = E2[kk], h = H2[kk];
                    sx = fma(t.x, e.x, sx); sy = fma(t.y, e.y, sy);    // np.sum(a, 0), a = beta * exp(eta~)
                    lx = fma(t.x, h.x, lx); ly = fma(t.y, h.y, ly);    // theta @ a
                }
                double Sw = sx + sy, Lw = lx + ly;
                Sw += dpp_move<DPP_XOR1>(Sw); Sw += dpp_move<DPP_XOR2>(Sw);
                Lw += dpp_move<DPP_XOR1>(Lw); Lw += dpp_move<DPP_XOR2>(Lw);
                const bool valid = w < nw, own = q == 0;
                const double c = valid ? c0 : 0.0, sq = sqrt(c);
                const double wq = valid ? sq / Sw : 0.0;   // sqrt(c) / colsum: update_z, stm.py:1115, and the factor of b, stm.py:1001
                // c * log(theta @ a) (stm.py:1095): the quad's four lanes all hold the word's Lw -- lane q keeps the one of every
                // fourth tile, and the logarithm is taken once per four tiles over 64 distinct (word, tile) pairs
                {
                    const bool mine = ((t0 / TW) & 3) == q;     // t0 / TW: the tile's number
                    Lst = mine ? (valid ? Lw : 1.0) : Lst;
                    cst = mine ? c : cst;
                    if (((t0 / TW) & 3) == 3) {   // uniform
                        ll += cst * log_pos(Lst);
                        Lst = 1.0; cst = 0.0;
                    }
                }
                csum += own ? c : 0.0;
                if (own) *reinterpret_cast<double2 *>(wpar + 2 * w) = make_double2(wq, sq);
                // REM: b[w][R0] = (beta exp(eta~)) sqrt(c) / S of the remainder row's topic, once per word here instead of once per
                // word and LANE in the lane = topic pass below (the same two multiplications).  It goes into the 8-double pads behind
                // the two tile buffers (words 0..7 / 8..15), which only have to be finite (the sums meet them with zeros).
                if (REM) {
                    const double rbw = (T[w * PITCH + R0] * sex[R0]) * wq;
                    if (own) post_lds[(w >> 3) * TILE + TW * PITCH + (w & 7)] = rbw;
                }
                // phi = beta * theta * r (stm_betass.h): r = exp-sum * c / S, in update_z's association (sqrt(c) / S) * sqrt(c).
                // assert np.all(phi >= 0) (stm.py:1117) fails exactly when a column sum is 0 (0 * inf), infinite or NaN.
                if (valid && own) P.rw[sl0] = (wq * sq) * sumex;
                sbad |= valid && !(Sw > 0.0 && Sw < INFINITY);
            }
            STM_POST_SYNC();
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[1] += cy - cy0; cy0 = cy; }
            // -- 2. rowsum(c') and b b^T.  b = a * (sqrt(c) / S) = (beta * exp(eta~)) * (sqrt(c) / S) serves both the Hessian
            // (stm.py:1001, which divides a * sqrt(c) by S: <= 1.5 ulp apart) and phi = b * sqrt(c) (stm.py:1115-1116, this
            // order); rowsum(c') of stm.py:1002,1011 is the row sum of that same product.  b is never written back to the
            // tile: each reader scales the beta it reads (the same two multiplications, the same bits) -- the matrix cores'
            // fragments by their topic's exp(eta~) and their word's sqrt(c) / S, the lane = topic pass (row sums in word
            // order, the remainder row's products) by its own.  Words beyond the document carry sqrt(c) / S = 0: b = 0.
            // (phi itself goes to beta_ss in stm_betass.h's word-major pass.)
            {
                const double2 *wp2 = reinterpret_cast<const double2 *>(wpar);
                const double *tr = T + fq * PITCH + fr;
                const double *tc = T + (lane < PITCH ? lane : 0);
                double exf[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) exf[b] = sex[16 * b + fr];
                double h0 = 0.0, h1 = 0.0;
                // Four words per step: the step's LDS reads first (fragments and the lane = topic pass's cells), then its six
                // matrix-core instructions, then the pass's arithmetic in their shadow.  Steps beyond the document's last word
                // are rows of zeros: skipped (they would add +0 everywhere).
#pragma unroll
                for (int s = 0; s < TW / 4; ++s) {
                    if (4 * s >= nw) break;   // uniform
                    double fl[NB], t[4], rb4[4];
                    double2 wp[4];
#pragma unroll
                    for (int b = 0; b < NB; ++b) fl[b] = tr[4 * s * PITCH + 16 * b];
                    const double wqs = wpar[2 * (4 * s + fq)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        t[u] = tc[(4 * s + u) * PITCH]; wp[u] = wp2[4 * s + u];
                        if (REM) rb4[u] = post_lds[((4 * s + u) >> 3) * TILE + TW * PITCH + ((4 * s + u) & 7)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(STM_ABLATE & 2)) {
                        double f[NB];
#pragma unroll
                        for (int b = 0; b < NB; ++b) f[b] = (fl[b] * exf[b]) * wqs;
                        int tt = 0;
#pragma unroll
                        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                            for (int bj = bi; bj < NB; ++bj, ++tt)
                                acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc[tt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // lane = topic (lanes from K on: exp(eta~) = 0 against finite cells)
                    if (!(STM_ABLATE & 4)) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double b = (t[u] * ex) * wp[u].x;
                            rowc += b * wp[u].y;
                            if (REM) {
                                if (u & 1) h1 = fma(b, rb4[u], h1); else h0 = fma(b, rb4[u], h0);
                            }
                        }
                    }
                    // the sums are pinned here, or the step's arithmetic sinks below the later steps' loads (registers)
                    asm volatile("" : "+v"(rowc), "+v"(h0), "+v"(h1));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (REM) hrem += h0 + h1;
                if (dump_phi && isk)   // the reference keeps the last document's phi (stm.py:1116)
                    for (int w = 0; w < nw; ++w)
                        P.phi_out[(size_t)lane * Nd + t0 + w] = ((T[w * PITCH + lane] * ex) * wpar[2 * w]) * wpar[2 * w + 1];
            }
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[2] += cy - cy0; cy0 = cy; }
            STM_POST_SYNC();
            wait_lds();         // every read of this buffer has returned before the tile after next is fetched into it
            if (DBG && P.prof) { const long long cy = __builtin_readcyclecounter(); tq[3] += cy - cy0; }
            idx0 = idx1; c0 = c1; sl0 = sl1; idx1 = idx2; c1 = c2; sl1 = sl2;
        }
        ll += cst * log_pos(Lst);   // the pairs of the last, incomplete batch of tiles (lanes without one: 0 * log(1))
        // From here to the end of the document the wave runs dependent chains (assembly -> ladder -> inverse -> nu): it goes ahead of
        // the SIMD's other waves, whose word loops have independent work to fill the gaps (priority 1; 2-3 inside the factorisation)
        __builtin_amdgcn_s_setprio(1);
        STM_POST_SYNC();
        sth[lane] = isk ? ths : 0.0;   // inside region 0, behind the matrix: the tiles are done
        if (DBG && P.prof && lane == 0) for (int q = 0; q < 4; ++q) P.prof[doc * PROF_SLOTS + 24 + q] = tq[q];
        if (DBG && P.prof) tp[2] = (long long)__builtin_readcyclecounter();
        relane();
        if (!STM_ABLATE && wave_any(sbad || (isk && !(rowc >= 0.0)))) atomicMax(P.err_flag, 7 /* STM_ERR_PHI */);   // stm.py:1117
        const double Ndoc = (double)(long long)wave_sum(csum);
        ll = wave_sum(ll);

        // ---- H = b b^T - N theta theta^T (+ siginv off the diagonal), from the accumulator tiles into the packed lower
        // triangle: element (i, j), i <= j, of the upper block triangle is stored as M[j][i].  The diagonal cells get the raw
        // b b^T - N theta^2; the lane that owns row i turns it into diagA below.  Run again (same registers, same
        // operations, same bits) when a failed factorisation has eaten the matrix.
        auto assemble = [&]() __attribute__((always_inline)) {
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NB; ++bj, ++t) {
                    const int j = bj * 16 + fr, jc = j < n ? j : nm1;
                    const double thj = sth[jc];
                    const int rsj = RS(jc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = bi * 16 + fq + 4 * r, ic = i < n ? i : nm1;
                        double h = acc[t][r] - Ndoc * (sth[ic] * thj);
                        if (!sdiag && i != j) h += S[(size_t)ic * n + jc];
                        const bool st = j < n && (bi != bj || i <= j);
                        M[st ? rsj + i : MDUMP] = h;
                    }
                }
            if (REM) {
                const int ic = isn ? lane : nm1;
                double h = hrem - Ndoc * (sth[ic] * sth[R0]);
                if (!sdiag && lane != R0) h += S[(size_t)R0 * n + ic];
                M[isn ? RS(R0) + lane : MDUMP] = h;
            }
        };

        double diagA = 1.0, Ldiag = 1.0;
        bool clean = false;
        long long tcc[4] = {0, 0, 0, 0};
        // np.linalg.cholesky, blocked by 16 columns and in place of the packed triangle.  Per panel: (a) the block column
        // minus the products of the finished panels on the matrix cores; (b) the panel with lane i holding row i's 16
        // entries in registers, right-looking and free of branches on the data -- a failed pivot only raises a flag, and
        // what the remaining columns compute from it is never stored -- so the updates of the later columns fill the
        // latency of the pivot's rsq + Goldschmidt chain; finished entries of row J come from lane J by v_readlane.
        auto cholesky = [&]() __attribute__((always_inline)) -> bool {
            // a pivot never exceeds its diagonal entry (what is subtracted from it are squares, in floating point too): an
            // entry <= 0 (or NaN) fails some pivot test for certain, and the attempt is decided without factorising
            if (!STM_ABLATE && wave_any(isn && !(diagA > 0.0))) return false;
            // a pivot never exceeds its diagonal entry and a pivot that passes is above 32 eps of it: with the diagonal in
            // the normal range every accepted pivot is, and the per-pivot range test of sqrt_and_rsqrt can go
            const bool fast = !wave_any(isn && !(diagA > 1e-260 && diagA < 1e270));
            clean = false;
            if (isn) M[RS(lane) + lane] = diagA;
            STM_POST_SYNC();
            bool ok = true;
            __builtin_amdgcn_s_setprio(2);   // the factorisation (updates -> panel -> updates ...) is the document's longest dependent stretch
            constexpr int NBP = REM ? NB : NBC;   // REM: the one column beyond the full blocks is a single pivot, below
#pragma unroll 1
            for (int p = 0; p < ((STM_ABLATE & 8) ? 0 : NBP) && ok; ++p) {
                const int J0 = 16 * p;
                relane();
                long long cq = (DBG && P.prof) ? (long long)__builtin_readcyclecounter() : 0;
                if (p > 0) {
                    const int bc = J0 + fr, bcc = bc < n ? bc : nm1;
                    const double *brow = M + RS(bcc);                       // row of L_p* for the B operands (L_pk^T)
#pragma unroll 1
                    for (int bi = p; bi < NBC; ++bi) {
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
                    STM_POST_SYNC();
                }
                if (DBG && P.prof) { const long long c1 = __builtin_readcyclecounter(); tcc[0] += c1 - cq; cq = c1; }
                // (b) rows above the panel shadow its first row, rows beyond n the last one (never stored)
                const int ic = lane < J0 ? J0 : (isn ? lane : nm1);
                const double2 *wr = reinterpret_cast<const double2 *>(M + RS(ic) + J0);
                double w[16];
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) { const double2 t = wr[c2]; w[2 * c2] = t.x; w[2 * c2 + 1] = t.y; }
                // Column J's finished entries reach the other lanes through a 64-entry LDS column (one store, then broadcast
                // reads two at a time -- no v_readlane pair per update), one column behind: the update with column J-1 is
                // applied while column J's pivot chain runs.  That chain does not wait for it: lane J owns row J, so its
                // fully updated diagonal entry is w[j] - L[J][J-1]^2 from its own registers (the very fma the update
                // below performs for that lane).
                bool badl = false;
                if (DBG && P.prof) { pin(w[0]); pin(w[15]); const long long c1 = __builtin_readcyclecounter(); tcc[1] += c1 - cq; cq = c1; }
                __builtin_amdgcn_s_setprio(3);   // the pivot chain is the longest serial stretch of the kernel: ahead of the SIMD's other waves
                double *cbw = cb + lane;
                const double *cbr = cb + J0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (J0 + j < n) {   // uniform
                        const int J = J0 + j;
                        const double tmp = j > 0 ? fma(-w[j > 0 ? j - 1 : 0], w[j > 0 ? j - 1 : 0], w[j]) : w[0];
                        const double d = lane_bcast(tmp, J);
                        badl |= (lane == J) && !(tmp > PIVOT_TOL * diagA);   // see PIVOT_TOL
                        double ljj, rjj;                        // LAPACK dpotf2 scales the column by the reciprocal as well
                        if (fast) sqrt_and_rsqrt_pivot(d, ljj, rjj); else sqrt_and_rsqrt(d, ljj, rjj);   // (uniform)
                        if (lane == J) Ldiag = ljj;
                        if (j > 0) {   // the update with column J - 1 (stored at the end of the previous step), four broadcasts at a time
#pragma unroll
                            for (int c0 = j; c0 < 16; c0 += 4) {
                                double xs[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u) if (c0 + u < 16) xs[u] = cbr[c0 + u];   // L[J0 + c][J - 1]
#pragma unroll
                                for (int u = 0; u < 4; ++u) if (c0 + u < 16) w[c0 + u] = fma(-w[j > 0 ? j - 1 : 0], xs[u], w[c0 + u]);
                            }
                        }
                        w[j] *= rjj;
                        if (j < 15) {
                            STM_POST_SYNC();
                            *cbw = w[j];
                            STM_POST_SYNC();
                        }
                    }
                }
                const bool bad = wave_any(badl);
                __builtin_amdgcn_s_setprio(2);
                if (DBG && P.prof) { pin(w[15]); const long long c1 = __builtin_readcyclecounter(); tcc[2] += c1 - cq; cq = c1; }
                if (bad && !STM_ABLATE) { ok = false; break; }
                // pairs (c, c + 1) with the first cell strictly below the diagonal; the second one is then at most the
                // diagonal cell, which is free (it takes X's diagonal later)
#pragma unroll
                for (int c2 = 0; c2 < 8; ++c2) {
                    const int c = 2 * c2;
                    const bool st = isn && lane > J0 + c;
                    *reinterpret_cast<double2 *>(M + (st ? RS(lane) + J0 + c : MDUMP)) = make_double2(w[c], w[c + 1]);
                }
                STM_POST_SYNC();
                if (DBG && P.prof) { const long long c1 = __builtin_readcyclecounter(); tcc[3] += c1 - cq; }
            }
            if (REM && ok && !(STM_ABLATE & 8)) {   // the last pivot: A[R0][R0] - sum_k L[R0][k]^2 (row R0 is final: every panel stored its part of it)
                relane();
                const double l = lane < R0 ? M[RS(R0) + lane] : 0.0;
                const double d = M[RS(R0) + R0] - wave_sum(l * l);
                double ljj, rjj;
                sqrt_and_rsqrt(d, ljj, rjj);
                if (lane == R0) Ldiag = ljj;
                if (wave_any(lane == R0 && !(d > PIVOT_TOL * diagA))) ok = false;
            }
            __builtin_amdgcn_s_setprio(1);
            if (STM_ABLATE) return true;
            return ok;
        };
        auto make_pd = [&]() __attribute__((always_inline)) {  // stm.py:964-984; M holds A (clean)
            if (isn) {
                double mag = 0.0;
                const double *ri = M + RS(lane);
                for (int j = 0; j < n; ++j) {
                    const double lo = ri[j < lane ? j : 0], up = M[RS(j) + (j > lane ? lane : 0)];
                    const double aij = (j == lane) ? diagA : (j < lane ? lo : up);
                    mag += fabs(aij);
                }
                mag -= fabs(diagA);
                if (diagA < mag) diagA = mag;
            }
        };
        auto dump_hess = [&]() {
            double *o = P.hess_out + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j)
                    o[(size_t)lane * n + j] = (j == lane) ? diagA : (j < lane ? M[RS(lane) + j] : M[RS(j) + lane]);
        };

        // One assembly site and one Cholesky site for every stage of the reference's PD ladder:
        //   0 hessian(): PD test as Cholesky success (stm.py:1017)   1 after make_pd (stm.py:1019-1020)
        //   2 +1e-5 (stm.py:1021), decompose_hessian's np.linalg.cholesky (stm.py:1040)
        //   3 after make_pd (stm.py:1043)   4 scipy cholesky (UPPER) of make_pd(H) + 1e-5 I (stm.py:1046-1048)
        int path = 0;
        bool upper = false, fail = false;
        double keep = 0.0;
        STM_POST_SYNC();
        for (int attempt = 0;; ++attempt) {
            relane();
            if (!clean) {
                if (!(STM_ABLATE & 128)) assemble();
                STM_POST_SYNC();
                clean = true;
                if (attempt == 0 && isn) {
                    const double sii = S[(size_t)lane * n + lane];
                    diagA = ((M[RS(lane) + lane] - rowc) + Ndoc * ths) + sii;   // stm.py:1003-1013, in this order
                }
                if (DBG && P.prof && attempt == 0) tp[3] = (long long)__builtin_readcyclecounter();
            }
            // what the previous, failed attempt asks for
            if (attempt == 1) { make_pd(); path = 1; }
            else if (attempt == 2) { if (isn) diagA += 1e-5; path = 2; }
            else if (attempt == 3) { make_pd(); }
            else if (attempt == 4) { make_pd(); keep = diagA; if (isn) diagA += 1e-5; }
            if (DBG && P.hess_out && attempt <= 2) dump_hess();
            const bool ok = cholesky();
            if (attempt == 4) { diagA = keep; upper = true; fail = !ok; break; }
            if (ok) break;
        }
        if (P.pd_path) P.pd_path[doc] = path;
        if (fail) {
            atomicMax(P.err_flag, 3 /* STM_ERR_LINALG */);
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        if (DBG && P.chol_out) {
            double *o = P.chol_out + (size_t)doc * n * n;
            if (isn)
                for (int j = 0; j < n; ++j) {
                    const double val = (j == lane) ? Ldiag : (j < lane ? M[RS(lane) + j] : 0.0);
                    if (upper) o[(size_t)j * n + lane] = val;  // the reference holds the upper factor here
                    else o[(size_t)lane * n + j] = val;
                }
        }

        if (DBG && P.prof) tp[4] = (long long)__builtin_readcyclecounter();
        // ---- bound (stm.py:1068-1101)
        const double det = wave_sum(isn ? log(Ldiag) : 0.0);
        double q = 0.0;
        {
            const double d = eta_i - mu_i;
            if (sdiag) {
                if (isn) q = (d * S[(size_t)lane * n + lane]) * d;
            } else {
                STM_POST_SYNC();
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

        if (DBG && P.prof) tp[5] = (long long)__builtin_readcyclecounter();
        relane();
        // ---- nu = inv(triu(L^T)) inv(triu(L^T))^T (stm.py:1052-1066)
        const double Rdiag = 1.0 / Ldiag;
        STM_POST_SYNC();
        srd[lane] = isn ? Rdiag : 0.0;
        STM_POST_SYNC();
        long long ti[3] = {0, 0, 0};
        if (DBG && P.prof) ti[0] = (long long)__builtin_readcyclecounter();
        if (!upper && !(STM_ABLATE & 16)) {
            // X = L^-1 (so that nu = X^T X), blocked by 16 and IN PLACE of L (diagonal in the triangle's free diagonal cells).
            // (I) all diagonal blocks at once, lane = (block, column c), the column in registers:
            //     x[i] = X[i][c] = -(sum_{l<i} L[i][l] x[l]) / L[i][i]   (x[l] = 0 above the diagonal, x[c] = 1 / L[c][c]);
            //     the rows of L are independent of x, so their loads run ahead of the substitution chain, and every store
            //     comes after every load (one instruction stream, the LDS works in order).
            __builtin_amdgcn_s_setprio(2);   // a 15-step substitution chain
            {
                const int c = lane & 15, rb = lane & ~15;
                const bool has = rb < n;
                const int rbc = has ? rb : 0;                         // lanes beyond the matrix shadow block 0 (nothing is stored)
                const int rows = n - rbc < 16 ? n - rbc : 16;
                const double2 *rd2 = reinterpret_cast<const double2 *>(srd + rbc);
                // row rbc + i of the block starts at RS(rbc + i) + rbc = RS(rbc) + rbc + i rbc + RS(i) (rbc is even);
                // rows beyond the matrix shadow the block's last row (their x is never stored)
                const int base0 = RS(rbc) + rbc, lastoff = base0 + (rows - 1) * rbc + RS(rows - 1);
                auto row_of = [&](int i) __attribute__((always_inline)) {
                    return reinterpret_cast<const double2 *>(M + (i < rows ? base0 + i * rbc + tri_row(i) : lastoff));
                };
                double x[16];
                double2 buf[2][8];   // the row after next is fetched while a row is consumed
                x[0] = (c == 0) ? rd2[0].x : -0.0;
                { const double2 *r1 = row_of(1); buf[1][0] = r1[0]; }
#pragma unroll
                for (int i = 1; i < 16; ++i) {
                    if (i + 1 < 16) {
                        const double2 *rn = row_of(i + 1);
#pragma unroll
                        for (int l2 = 0; 2 * l2 < i + 1; ++l2) buf[(i + 1) & 1][l2] = rn[l2];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double t0 = 0.0, t1 = 0.0;
#pragma unroll
                    for (int l2 = 0; 2 * l2 < i; ++l2) {
                        const double2 lv = buf[i & 1][l2];
                        t0 = fma(lv.x, x[2 * l2], t0);
                        if (2 * l2 + 1 < i) t1 = fma(lv.y, x[2 * l2 + 1], t1);
                    }
                    const double2 rdp = rd2[i >> 1];
                    const double rd = (i & 1) ? rdp.y : rdp.x;
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
            __builtin_amdgcn_s_setprio(1);   // (the rest of the document -- inverse, nu -- stays ahead of the other waves' word loops)
            STM_POST_SYNC();
            if (DBG && P.prof) ti[1] = (long long)__builtin_readcyclecounter();
            // (II) X_ij = -X_ii (sum_{j<=k<i} L_ik X_kj) on the matrix cores, block columns left to right, block rows
            //      top down (X_ij takes the place of L_ij, which no later product reads).  The inner sum comes out
            //      of the MFMA in exactly the register layout its B operand wants, so it never visits the LDS.
            constexpr int NBI = REM ? NB : NBC;   // REM: the one row beyond the full blocks is done on the VALU, below
#pragma unroll 1
            for (int bj = 0; bj + 1 < NBI; ++bj) {
#pragma unroll 1
                for (int bi = bj + 1; bi < NBI; ++bi) {
                    const int ar = bi * 16 + fr, arc = ar < n ? ar : nm1;
                    const double *arow = M + RS(arc);                     // row of L_i* / X_ii for the A operands
                    const int bc = bj * 16 + fr;
                    v4d sacc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                    for (int k = bj; k < bi; ++k) {
                        double av[4], bv[4];
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const int kk = k * 16 + 4 * sk + fq;           // < 16 (NBC - 1) <= n: full blocks only
                            av[sk] = arow[kk];                              // L_ik[fr][4 sk + fq]
                            bv[sk] = M[RS(kk) + bc];                        // X_kj[4 sk + fq][fr]
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
                        xv[sk] = arow[ac < arc ? ac : arc];                 // X_ii[fr][4 sk + fq], at most the diagonal cell
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
                        M[row < n ? RS(row) + bc : MDUMP] = -dacc[r];
                    }
                    STM_POST_SYNC();
                }
            }
            if (DBG && P.prof) ti[2] = (long long)__builtin_readcyclecounter();
            if (REM) {   // (III) the row beyond the blocks: X[R0][j] = -X[R0][R0] sum_{j<=k<R0} L[R0][k] X[k][j], lane = column j
                double t[4] = {0.0, 0.0, 0.0, 0.0};
                const double2 *lr = reinterpret_cast<const double2 *>(M + RS(R0));
                const int jc = lane < R0 ? lane : 0;
#pragma unroll 1
                for (int k = 0; k < R0; k += 8) {
                    double2 lv[4];
                    double xv[8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) lv[u] = lr[(k >> 1) + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = M[RS(k + u) + jc];   // above the diagonal: past the end of row k + u, selected away
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double l = (u & 1) ? lv[u >> 1].y : lv[u >> 1].x;
                        t[u & 3] = fma(l, (k + u >= lane) ? xv[u] : 0.0, t[u & 3]);
                    }
                }
                const double xr = -((t[0] + t[1]) + (t[2] + t[3])) * srd[R0];
                STM_POST_SYNC();
                M[lane < R0 ? RS(R0) + lane : MDUMP] = xr;   // after every lane's reads of row R0 (one instruction stream)
            }
        }
        STM_POST_SYNC();
        if (DBG && P.prof) tp[6] = (long long)__builtin_readcyclecounter();
        relane();
        if (DBG && P.prof && lane == 0 && !upper) { P.prof[doc * PROF_SLOTS + 28] = ti[1] - ti[0]; P.prof[doc * PROF_SLOTS + 29] = ti[2] - ti[1]; P.prof[doc * PROF_SLOTS + 30] = tp[6] - ti[2]; P.prof[doc * PROF_SLOTS + 31] = ti[0] - tp[5]; }
        // nu = R R^T = X^T X (sigma_ss += nu, stm.py:582), one block column bj of output tiles (bi <= bj) at a time on the
        // matrix cores: nu[i][j] = sum_{l >= 16 bj} X[l][i] X[l][j] (X is lower triangular: rows above block bj add nothing);
        // fragment X[s4 + fq][b*16 + fr], zero above the diagonal.  The workgroup's running sum lives in a slab of its own in
        // the accumulators' register layout (tile (b, bj) at bj (bj + 1) / 2 + b, [r][lane]: every access one 512-byte
        // run); its old values are fetched before the matrix-core loop they are added behind -- plain loads and stores,
        // nobody else touches the slab.  Cells beyond n are exact zeros.  untile_sigma_kernel undoes the layout.
        double *nu_doc = (DBG && P.nu_out) ? P.nu_out + (size_t)doc * n * n : nullptr;
        if (upper) {   // nu = diag(1 / L_ii^2): element (i, i) sits in tile (b, b) at r = ((i & 15) - fq) / 4, lane = (fq, fr = i & 15)
            if (REM && lane == R0) {
                const double v = srd[R0] * srd[R0];
                sig_acc[(size_t)(NU_TILES - 1) * 4 * WAVE + R0] += v;
                if (DBG && nu_doc)
                    for (int j = 0; j < n; ++j) nu_doc[(size_t)R0 * n + j] = (j == R0) ? v : 0.0;
            }
            for (int bb = 0; bb < NBV; ++bb) {
                const int i = bb * 16 + fr, r = (fr - fq) >> 2;
                if (((fr - fq) & 3) == 0 && fr >= fq && i < n) {
                    const double v = srd[i] * srd[i];
                    sig_acc[((size_t)(bb * (bb + 1) / 2 + bb) * 4 + r) * WAVE + lane] += v;
                    if (DBG && nu_doc)
                        for (int j = 0; j < n; ++j) nu_doc[(size_t)i * n + j] = (j == i) ? v : 0.0;
                }
            }
        } else if (!(STM_ABLATE & 32)) {
            if (REM) {   // the last column on the VALU: nu[i][R0] = X[R0][i] X[R0][R0] (row R0 is X's only row with an entry there)
                const int ic = isn ? lane : nm1;
                double *cell = sig_acc + (size_t)(NU_TILES - 1) * 4 * WAVE + lane;
                const double oldv = *cell;
                const double v = isn ? M[RS(R0) + ic] * M[RS(R0) + R0] : 0.0;
                *cell = oldv + v;
                if (DBG && nu_doc && isn) { nu_doc[(size_t)lane * n + R0] = v; nu_doc[(size_t)R0 * n + lane] = v; }
            }
            // Row quads outermost, like b b^T: the fragments X[s4 + fq][b 16 + fr] of a quad in block row R are read ONCE and serve every
            // tile (b, bj), b <= bj <= min(R, NBV - 1) (one block column at a time read them again for every block column: 27 load-then-
            // wait steps per document at K = 50 instead of 13).  Tile (b, bj) sits at slot bj (bj + 1) / 2 + b and still receives its
            // products in the order s4 = 16 bj, 16 bj + 4, ...: the same bits.
            constexpr int NVT = NBV * (NBV + 1) / 2;
            v4d an[NVT], old[NVT];
#pragma unroll
            for (int t = 0; t < NVT; ++t) {
                an[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) old[t][r] = sig_acc[(size_t)(t * 4 + r) * WAVE + lane];
            }
#pragma unroll
            for (int R = 0; R < NBC; ++R) {
                const int RB = R < NBV ? R : NBV - 1;      // last block column this block row feeds (compile-time after unrolling)
                const int rjR = RB * 16 + fr;
#pragma unroll 1
                for (int s4 = 16 * R; s4 < 16 * R + 16 && s4 < n; s4 += 4) {
                    const int col = s4 + fq, colc = col < n ? col : nm1;
                    const double *xr = M + RS(colc);
                    double g[NBV];
#pragma unroll
                    for (int bb = 0; bb < NBV; ++bb)
                        if (bb <= RB) g[bb] = xr[bb * 16 + fr < n ? bb * 16 + fr : nm1];
#pragma unroll
                    for (int bb = 0; bb < NBV; ++bb)
                        if (bb <= RB) g[bb] = (col < n) ? g[bb] : 0.0;
                    if (R < NBV) g[RB] = (rjR < n && col >= rjR) ? g[RB] : 0.0;   // the diagonal block row: X is lower triangular
#pragma unroll
                    for (int bj = 0; bj < NBV; ++bj)
#pragma unroll
                        for (int bb = 0; bb <= bj; ++bb)
                            if (bj <= RB)
                                an[bj * (bj + 1) / 2 + bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[bb], g[bj], an[bj * (bj + 1) / 2 + bb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int bj = 0; bj < NBV; ++bj)
#pragma unroll
                for (int bb = 0; bb <= bj; ++bb) {
                    const int t = bj * (bj + 1) / 2 + bb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sig_acc[(size_t)(t * 4 + r) * WAVE + lane] = old[t][r] + an[t][r];
                        if (DBG && nu_doc) {
                            const int i = bb * 16 + fq + 4 * r, j = bj * 16 + fr;
                            if (i < n && j < n) {
                                nu_doc[(size_t)i * n + j] = an[t][r];
                                nu_doc[(size_t)j * n + i] = an[t][r];
                            }
                        }
                    }
                }
        }
        __builtin_amdgcn_s_setprio(0);
        if (DBG && P.prof && lane == 0 && (P.debug_flags & 32)) for (int q2 = 0; q2 < 4; ++q2) P.prof[doc * PROF_SLOTS + 28 + q2] = tcc[q2];
        if (DBG && P.prof && lane == 0) {
            tp[7] = (long long)__builtin_readcyclecounter();
            for (int q2 = 0; q2 < 7; ++q2) P.prof[doc * PROF_SLOTS + 32 + q2] = tp[q2 + 1] - tp[q2];
        }
    }
}

}  // namespace stm
