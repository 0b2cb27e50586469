// stm_solver.h -- per-document variational optimisation of eta: one wavefront per document.
//
// Replaces, per document, reference src/modules/stm.py:532-546 (get_beta gather,
// optimize_eta) -- i.e. scipy.optimize.minimize(f, x0=eta, jac=df, method="BFGS") with
// the objective f (stm.py:920-944) and the gradient df (stm.py:946-958) exactly as the
// reference defines them, and scipy's _minimize_bfgs / DCSRCH / Wolfe2 / zoom control
// flow restated as ONE wave-uniform state machine with a single
// evaluation site, so the K x Nd contraction is instantiated once.
//
// Data layout: beta is held word-major in HBM, betaT[A][V][K], so a word's K-vector is one
// contiguous 8K-byte run.  The document's K x Nd block beta_d is gathered ONCE and then stays
// on chip for the ~64 objective evaluations of the solve (re-streaming it from L2/HBM per
// evaluation was the v1 bottleneck: 0.5 TB of slab traffic per E-step at 100k documents):
//   * words 0..63   -> registers of the lane that owns the word (breg[KREG], KREG >= K)
//   * words 64..Nd-1 -> a private LDS slab slab[word][KP], KP = K rounded up to 4j+2: a lane
//     streams its word's row with ds_read_b128 at immediate offsets, and the row stride of
//     8j+4 dwords makes every 16-lane group of a wave hit 16 distinct 4-bank slots (no
//     conflicts); sized per launch from the longest document of the launch -- documents are
//     launched longest-first in groups of equal LDS occupancy (see stm_api.hip)
//   * documents too long for the 160 KiB LDS use the GLOBAL_SLAB variant (slab in HBM/L2).
// exp(eta~ - m) is broadcast through a 64-entry LDS vector.  The BFGS inverse-Hessian
// estimate lives in a private global slab (n x n, touched only nit times).  Lane i holds
// component i of every length-(K-1) vector; all line-search scalars are wave-uniform.
#pragma once
#include <type_traits>
#include "stm_wave.h"

#ifndef STM_EVAL_GROUP
#define STM_EVAL_GROUP 8   // topics per scheduling group of the evaluation's register pass
#endif
#ifndef STM_LATER_PASS
#define STM_LATER_PASS 0   // 1: searches mom_k0 <= k <= mom_k1 start with a moment pass of their own (measured in round 5: a wash at C2 --
#endif                     // the failing third search is already cut after ONE evaluation -- and the extra code costs the rest 2 %)
#ifndef STM_LSE_LATE
#define STM_LSE_LATE 1     // (bit 0: plain evaluations, bit 1: the fused first one -- that one spills) two-wave form: wave 0 takes the log-sum-exp's logarithm together with its word's (log_pos2) behind barrier (1)
#endif
#ifndef STM_FUSE_SLAB
#define STM_FUSE_SLAB 7    // bit 0: data_F, bit 1: words_F3, bit 2: moments_words take the register word and the slab word side by side
#endif
#ifndef STM_FUSE_PIPE
#define STM_FUSE_PIPE 4    // pairs of topics per software-pipelined group of the side-by-side evaluation (0: not pipelined)
#endif
#ifndef STM_FUSE_PIPE3
#define STM_FUSE_PIPE3 2   // the same for the first evaluation's three-sum pass (twelve chains, four reads per pair of topics)
#endif
#ifndef STM_REG_PIPE
#define STM_REG_PIPE 4     // wave 0's register-word pass of a plain evaluation, pipelined the same way (pairs of topics per group)
#endif
#ifndef STM_REG_PIPE3
#define STM_REG_PIPE3 3    // ... and of the first evaluation's three-sum pass
#endif
#ifndef STM_QUAD_W0
#define STM_QUAD_W0 1      // two-wave form: the prior's quadratic form on wave 0 (it waits for wave 1's df at barrier (1) otherwise), its wave sum
#endif                     // together with the log-sum-exp's
#ifndef STM_G0_PIPE
#define STM_G0_PIPE 1      // DMA set-up: g0's lane = topic accumulation reads the next four words' operands ahead
#endif
#ifndef STM_SM_BATCH
#define STM_SM_BATCH 1     // state machine: independent wave reductions of a state run as one batch (wave_reduce_n: same sums, same bits)
#endif
#ifndef STM_DCSTEP_ILP
#define STM_DCSTEP_ILP 0   // dcstep: the independent divisions of a round at once (div_n).  Bit-identical and -0.5 % in the one-document-per-workgroup
                           // kernel; in the persistent form its extra live values tip the allocator over (51 instead of 17 spilled VGPRs, +15 %): off
#endif
#ifndef STM_PREFETCH
#define STM_PREFETCH 1     // persistent two-wave form: wave 1 touches the next document's CSR / eta / mu lines while it waits for the first evaluation
#endif
#ifndef STM_C0_LDS
#define STM_C0_LDS 1       // DMA form: the register word's count parked in the LDS (see its use)
#endif
#ifndef STM_FUSE_GROUP
#define STM_FUSE_GROUP 4   // topics per scheduling group of the side-by-side three-sum passes (twelve chains)
#endif
#ifndef STM_MOM_GROUP
#define STM_MOM_GROUP 8    // ... of the moment pass (three broadcast vectors, six sums)
#endif

// the per-document cycle counters (tools/solver_prof.py) exist in the -DSTM_TESTING build only: in the product kernel their clock reads and
// running totals are compiled out (31 spilled SGPRs less in the two-wave K = 50 form)
#ifdef STM_TESTING
#define STM_PROF(P) ((P).prof)
#else
#define STM_PROF(P) ((long long *)nullptr)
#endif

namespace stm {

// LDS hand-off between lanes of ONE wave: the LDS executes a wave's operations in order, so only
// the compiler has to be kept from reordering them (a workgroup barrier here would also have to
// be matched by the other wave of a two-wave workgroup)
#define STM_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// same, when the hand-off also goes through global memory (the BFGS matrix slab): wait for the stores
#define STM_WAVE_SYNC_MEM() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

struct SolverParams {
    int64_t N;
    int K, n, V;
    const int64_t *indptr;
    const int32_t *indices;
    const double *counts;
    const int32_t *aspect;  // nullable
    const double *betaT;    // [A][V][K]
    const double *colsum;   // [A][V] sum_k beta[k][w] in topic order (np.sum(beta_doc_kv, axis=0), stm.py:954): a property of the word,
                            // computed once per EM iteration (stm_api.hip: beta_colsum_kernel) instead of once per document
    const double *mu;       // [N][n]
    double *eta;            // [N][n] in/out
    const double *siginv;   // [n][n]
    int siginv_diag;        // 1: off-diagonals are exactly zero (what stm.py:501 produces)
    double sig_bound;       // upper bound on the largest eigenvalue of siginv (max absolute row sum)
    double *slab_beta;      // GLOBAL_SLAB variant only: [grid][(K+2)][ld]
    double *slab_H;         // [grid][n][n]
    int ld;                 // slab capacity in words (held outside registers), per launch
    int KP;                 // slab row length: K rounded up to 4j+2 (16-byte rows, conflict-free ds_read_b128)
    int64_t first;          // this launch covers order[first .. first + count)
    int64_t count;          // tickets of this launch (gridDim.x of them start as a workgroup's first document)
    int32_t *ticket_ctr;    // persistent form (two-wave kernels): the launch's ticket counter, zero at launch -- a workgroup takes documents
                            // gridDim.x + atomicAdd(ticket_ctr, 1) until none are left (nullptr: one document per workgroup, count == gridDim.x)
    const int32_t *order;   // optional processing order (nullable)
    const int64_t *tick;    // optional [N][2]: ticket -> {indptr[doc], doc | Nd << 32} of that order (nullable: the header is read through order / indptr)
    int32_t *status, *nit, *nfev, *njev;
    int32_t *err_flag;
    int lds_doubles;        // dynamic LDS of this launch, in doubles (debug bit3 poisons it)
    int zrow;               // DMA form: index of the all-zero row behind the last row of betaT (A * V)
    int mom_k0, mom_k1;     // searches mom_k0 <= k <= mom_k1 (besides the first one, k = 0) start with a moment pass of their own (S_OUTER_TOP);
                            // an outcome-neutral choice -- the pass only decides earlier what the search would find out (mom_k1 < mom_k0: none)
    int debug_flags;        // bit0: skip the BFGS loop (bring-up aid); bit1: no line-search cuts, bit2: no reuse of DCSRCH's first
                            // evaluation by wolfe2 (every evaluation scipy makes is made: A/B check of the shortcuts); bit4: no moment
                            // pass in front of the first search (the cuts of rounds 1-2 only)
    long long *prof;        // optional [N][PROF_SLOTS] shader-clock totals per document: [0] init, [1] evaluations, [2] state machine,
                            // [3] BFGS update, [8+st] cycles in state st (low 40 bits) and visits of it (bits 40..)
};

// doubles by which the two waves' DMA staging areas (32 rows of 2 ((KREG / 2) | 1) doubles each) reach beyond the BFGS matrix' n^2
__host__ __device__ inline int solver_dma_stage_extra(int kreg, int n) {
    const int need = 2 * 32 * 2 * ((kreg / 2) | 1);
    return need > n * n ? need - n * n : 0;
}

enum : int {
    S_INIT_DONE = 0, S_OUTER_TOP, S_W1_START, S_W1_ITER, S_W2_START, S_W2_FIRST, S_W2_TOP,
    S_W2_GOT_G, S_W2_GOT_F, S_ZOOM_TOP, S_ZOOM_GOT_F, S_ZOOM_GOT_G, S_MOMENTS, S_ACCEPT,
    S_ACCEPT2, S_FINISH
};

// scipy/optimize/_dcsrch.py:502-728 dcstep (wave-uniform scalars).  State in and out BY VALUE and
// the final interval update as selects: with reference parameters the conditional swaps of
// (stx, fx, dx) / (sty, fy, dy) became an indexed stack array, i.e. scratch-memory round trips in
// the middle of every line-search step.
struct DcStep {
    double stx, fx, dx, sty, fy, dy, stp;
    bool brackt;
};
__device__ __forceinline__ DcStep dcstep(DcStep S, double fp, double dp, double stpmin, double stpmax) {
    const double stx = S.stx, fx = S.fx, dx = S.dx, sty = S.sty, fy = S.fy, dy = S.dy, stp = S.stp;
    bool brackt = S.brackt;
    const double sgnd = np_sign(dp) * np_sign(dx);
    double stpf, stpc, stpq, theta, s, gamma, p, q, r;
#if STM_DCSTEP_ILP
    // The three cases that interpolate between stx and stp share their first two rounds of divisions, and within a round the
    // quotients do not depend on each other: div_n takes a round at once (every quotient keeps the bits of its own `/`; the one
    // a case does not use is computed for nothing -- the chain is latency-bound, its issue slots are free).
    if (fp > fx || sgnd < 0.0 || fabs(dp) < fabs(dx)) {
        const double A = fx - fp, B = stp - stx;
        const double n1[3] = {3.0 * A, A, dp}, d1[3] = {B, B, dp - dx};
        double q1[3];
        div_n(n1, d1, q1);                       // 3 (fx - fp) / (stp - stx); (fx - fp) / (stp - stx) [case 1's stpq]; dp / (dp - dx) [cases 2, 3]
        theta = q1[0] + dx + dp;
        s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        const double n2[4] = {theta, dx, dp, dx}, d2[4] = {s, s, s, q1[1] + dx};
        double q2[4];
        div_n(n2, d2, q2);                       // theta / s, dx / s, dp / s; dx / ((fx - fp) / (stp - stx) + dx) [case 1's stpq]
        const double rad = q2[0] * q2[0] - q2[1] * q2[2];
        if (fp > fx) {
            gamma = s * sqrt(rad);
            if (stp < stx) gamma *= -1;
            p = (gamma - dx) + theta;
            q = ((gamma - dx) + gamma) + dp;
            r = p / q;
            stpc = stx + r * (stp - stx);
            stpq = stx + (q2[3] * 0.5) * (stp - stx);     // (x / 2.0 == x * 0.5 exactly)
            if (fabs(stpc - stx) <= fabs(stpq - stx)) stpf = stpc;
            else stpf = stpc + (stpq - stpc) * 0.5;
            brackt = true;
        } else if (sgnd < 0.0) {
            gamma = s * sqrt(rad);
            if (stp > stx) gamma *= -1;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + dx;
            r = p / q;
            stpc = stp + r * (stx - stp);
            stpq = stp + q1[2] * (stx - stp);
            if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            brackt = true;
        } else {
            gamma = s * sqrt(py_max2(0.0, rad));
            if (stp > stx) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = (gamma + (dx - dp)) + gamma;
            r = p / q;
            if (r < 0 && gamma != 0) stpc = stp + r * (stx - stp);
            else if (stp > stx) stpc = stpmax;
            else stpc = stpmin;
            stpq = stp + q1[2] * (stx - stp);
            if (brackt) {
                if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
                else stpf = stpq;
                if (stp > stx) stpf = py_min2(stp + 0.66 * (sty - stp), stpf);
                else stpf = py_max2(stp + 0.66 * (sty - stp), stpf);
            } else {
                if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
                else stpf = stpq;
                stpf = np_clip(stpf, stpmin, stpmax);
            }
        }
    } else {
        if (brackt) {
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            s = py_max3(fabs(theta), fabs(dy), fabs(dp));
            const double n2[3] = {theta, dy, dp}, d2[3] = {s, s, s};
            double q2[3];
            div_n(n2, d2, q2);
            gamma = s * sqrt(q2[0] * q2[0] - q2[1] * q2[2]);
            if (stp > sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + dy;
            r = p / q;
            stpc = stp + r * (sty - stp);
            stpf = stpc;
        } else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
#else
    if (fp > fx) {
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma *= -1;
        p = (gamma - dx) + theta;
        q = ((gamma - dx) + gamma) + dp;
        r = p / q;
        stpc = stx + r * (stp - stx);
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        if (fabs(stpc - stx) <= fabs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma *= -1;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + dx;
        r = p / q;
        stpc = stp + r * (stx - stp);
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = true;
    } else if (fabs(dp) < fabs(dx)) {
        theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
        s = py_max3(fabs(theta), fabs(dx), fabs(dp));
        gamma = s * sqrt(py_max2(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (dx - dp)) + gamma;
        r = p / q;
        if (r < 0 && gamma != 0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            if (stp > stx) stpf = py_min2(stp + 0.66 * (sty - stp), stpf);
            else stpf = py_max2(stp + 0.66 * (sty - stp), stpf);
        } else {
            if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            stpf = np_clip(stpf, stpmin, stpmax);
        }
    } else {
        if (brackt) {
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            s = py_max3(fabs(theta), fabs(dy), fabs(dp));
            gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + dy;
            r = p / q;
            stpc = stp + r * (sty - stp);
            stpf = stpc;
        } else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
#endif
    // update the interval which contains a minimizer
    const bool hi = fp > fx, neg = sgnd < 0;
    DcStep R;
    R.sty = hi ? stp : (neg ? stx : sty);
    R.fy = hi ? fp : (neg ? fx : fy);
    R.dy = hi ? dp : (neg ? dx : dy);
    R.stx = hi ? stx : stp;
    R.fx = hi ? fx : fp;
    R.dx = hi ? dx : dp;
    R.stp = stpf;
    R.brackt = brackt;
    return R;
}

// scipy/optimize/_linesearch.py:477-508 _cubicmin; false == None
__device__ __forceinline__ bool cubicmin(double a, double fa, double fpa, double b, double fb,
                                         double c, double fc, double &xmin) {
    const double C = fpa, db = b - a, dc = c - a;
    const double t = db * dc;
    const double denom = (t * t) * (db - dc);
    const double d00 = dc * dc, d01 = -(db * db), d10 = -(dc * dc * dc), d11 = db * db * db;
    const double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
    double A = d00 * v0 + d01 * v1, B = d10 * v0 + d11 * v1;
    if (!finite_d(denom) || !finite_d(A) || !finite_d(B) || denom == 0.0) return false;
    A /= denom; B /= denom;
    const double radical = B * B - 3 * A * C;
    if (!finite_d(A) || !finite_d(B) || !finite_d(radical) || radical < 0) return false;
    const double den2 = 3 * A;
    if (den2 == 0.0 || !finite_d(den2)) return false;
    const double x = a + (-B + sqrt(radical)) / den2;
    if (!finite_d(x)) return false;
    xmin = x;
    return true;
}
// scipy/optimize/_linesearch.py:511-529 _quadmin
__device__ __forceinline__ bool quadmin(double a, double fa, double fpa, double b, double fb,
                                        double &xmin) {
    const double D = fa, C = fpa, db = b - a * 1.0;
    const double den = db * db;
    const double num = fb - D - C * db;
    if (den == 0.0 || !finite_d(den) || !finite_d(num)) return false;
    const double B = num / den;
    const double den2 = 2.0 * B;
    if (den2 == 0.0 || !finite_d(den2)) return false;
    const double x = a - C / den2;
    if (!finite_d(x)) return false;
    xmin = x;
    return true;
}

// NW = 1: one wavefront per document.
// NW = 2: two wavefronts per document (VPL = 1, KREG > 0).  Wave 0 runs the solver state machine
// and evaluates words 0..63 (registers); wave 1 is an evaluation server that holds words 64..127 in ITS
// registers and the words beyond 128 in the (small) LDS slab and computes, per request, the gradient df
// and the prior's quadratic form (while wave 0 takes max / exp) and then its share of the data term --
// the independent pieces of one objective evaluation run on two SIMD slots at once (three barriers
// per evaluation; everything the solver's control flow sees still passes through wave 0).
// DIRECT = 1 (VPL = 2, K > 64, where beta -- 40 MB at config 4 -- is far beyond what a CU can keep of a document):
// no private copy of beta_d at all.  Every pass over the document (column sums + g0 once, then every objective
// evaluation) re-gathers the rows from betaT itself, 16 words at a time through one LDS tile: a row is one coalesced
// run, rows are shared by all documents and stay in the L2 / Infinity Cache, whereas the per-document HBM slab of the
// GLOBAL_SLAB form (120 KB per document at K = 100) is streamed from HBM once per evaluation.
// DMA = true (two-wave form, K == KREG, K / 2 odd or padded to it): the rows of beta_d reach the registers through the LDS.
// One lane per word, 25 loads of 16 bytes from its own row, touches 64 cache lines per instruction and uses an eighth of
// each; here a row is one coalesced run moved by the LDS-DMA path into a staging area (the BFGS matrix's, unused until the
// first update), read back as ds_read_b128 by the lane that owns the word -- and while the rows are there, g0 is summed with
// lane = topic (two LDS reads, a multiply and an add per word) instead of 50 cross-lane reductions.
template <int VPL, int KREG, bool GLOBAL_SLAB, int NW = 1, int DIRECT = 0, bool DMA = false>
__global__ __launch_bounds__(64 * NW, (VPL > 2 ? 1 : VPL == 2 ? 2 : KREG > 50 ? 1 : KREG > 0 ? 2 : 4)) void solver_kernel(SolverParams P_arg) {
    constexpr int KMAX = 64 * VPL;
    constexpr int VREG = (KREG > 0) ? WAVE * NW : 0;  // words held in registers
    constexpr int KR = (KREG > 0) ? KREG : 2;
    static_assert(KREG % 2 == 0 && KREG <= KMAX, "KREG must be even and <= 64*VPL");
    static_assert(NW == 1 || (NW == 2 && VPL == 1 && KREG > 0 && !GLOBAL_SLAB), "two-wave form: VPL = 1, registers + LDS");
    static_assert(!DIRECT || (VPL == 2 && KREG == 0 && !GLOBAL_SLAB && NW == 1), "direct gather: the K > 64 one-wave form");
    static_assert(!DMA || (NW == 2 && VPL == 1 && KREG > 0 && !GLOBAL_SLAB && !DIRECT), "LDS-staged gather: the two-wave form");
    constexpr int TWS = 16;   // DIRECT: words per LDS tile
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];  // slab[ld][KP] | crow[ld] | wrow[ld] | H[n][n] (NW = 2)
    __shared__ __attribute__((aligned(16))) double se[KMAX + 2];  // exp(eta~ - m), broadcast to every lane
    __shared__ __attribute__((aligned(16))) double sv[KMAX + 2];  // vector broadcast (matvec operand / s; exp(eta~ - m) p~ of the moment pass)
    __shared__ __attribute__((aligned(16))) double sw[KMAX + 2];  // vector broadcast (w = H y; exp(eta~ - m) p~^2 of the moment pass)
    __shared__ double svb[NW == 2 ? KMAX + 1 : 1];  // wave 1's private broadcast vector
    // wave 0 <-> wave 1 mailbox (NW = 2)
    __shared__ double xch_xt[NW == 2 ? WAVE : 1];   // trial point
    __shared__ double xch_gv[NW == 2 ? WAVE : 1];   // df at the trial point / partial g0
    __shared__ double xch_res[8];                   // [0] data-term share, [1] quadratic form, [2] m, [4..5] word-count shares, [6..7] rho, cc of a BFGS update
    __shared__ int xch_cmd[4];
    __shared__ double ss[64];                       // the scalar state of the line searches (struct ud; wave 0's, Ndoc both waves')
    enum : int { U_old_fval, U_old_old_fval, U_gnorm, U_phi0, U_old_phi0, U_derphi0, U_Lb, U_Lv, U_prange, U_stx, U_fx, U_gx, U_sty, U_fy, U_gy, U_stmin, U_stmax, U_width, U_width1, U_finit, U_ginit, U_gtest, U_w1_a1, U_w1_f1, U_alpha0, U_alpha1, U_phi_a0, U_phi_a1, U_derphi_a0, U_a_lo, U_a_hi, U_phi_lo, U_phi_hi, U_derphi_lo, U_phi_rec, U_a_rec, U_a_j, U_acc_alpha, U_acc_f, U_alpha, U_fval, U_dval, U_cache_f, U_Ndoc, U_sig_lmax, U_mvar0, U_mD1, U_mD2, U_mg0p, U_mqx, U_COUNT };
    static_assert(U_COUNT <= 64, "scalar state");                      // [0] request bits (1 f, 2 g, 4 exit, 8 BFGS update (16: from the identity)), [1..2] bad-beta flags

    // One workgroup per document for the one-wave forms (no work-queue loop: a single-wave workgroup has no hardware barrier, so
    // cross-lane hand-offs through LDS must not straddle a loop back edge).  The two-wave form is PERSISTENT (round 5): measured
    // with tools/slot_gaps.py, the chip held 900 of its 1024 document slots busy -- between the end of a workgroup and the first
    // instruction of the next one in its place lie ~11 k cycles (4.7 us: wave termination, LDS / register release, dispatch and
    // state initialisation of a two-wave, 256-VGPR, 40 KB-LDS workgroup), 12 % of a document.  A workgroup now takes the next
    // document off a ticket counter itself: its first ticket is its block index, the following ones gridDim.x + atomicAdd(counter);
    // the add is issued by wave 1 at the START of a document (its return overlaps the index loads) and parked in the LDS, so the
    // hand-over at the end of a document is one workgroup barrier.  Tickets are handed out in order, i.e. longest documents first,
    // whoever is free next -- what the hardware dispatcher did.
    // The document loop must not look like one to the optimiser's hoisting passes: what does not change from document to document
    // (lane-derived addresses, flags and pointers of the parameter block) would otherwise be computed in front of the loop and
    // kept live through every document -- 190 spilled VGPRs and 220 spilled SGPRs in a kernel that sits at 256 registers.  So
    // the thread index and the address of the parameter block (the kernel-argument segment) are re-read behind an opaque move at
    // every document, and everything derived from them is a document's own.
#ifndef STM_SOLVER_PERSIST_FORM
#define STM_SOLVER_PERSIST_FORM 1
#endif
    // (The K > 64 direct-gather form, one wave per workgroup, was measured persistent too -- 242 VGPRs, no scratch, bit-identical: 8.79 ->
    // 9.15 ms at config 4's share.  The dispatch of a one-wave workgroup is cheap; the loop is not.  It stays one workgroup per document.)
    constexpr bool PERSIST = NW == 2 && STM_SOLVER_PERSIST_FORM;
    const bool persist = PERSIST && P_arg.ticket_ctr != nullptr;
    // (two slots, alternating: wave 1 may be a document ahead of wave 0's read of the ticket -- never two, there are barriers in between)
    __shared__ int tk_slot[2];
    __shared__ int pf_dump[PERSIST && STM_PREFETCH ? WAVE : 1];   // where the next document's prefetch lands (never read)
    int tk_par = 0;
    auto next_doc = [&]() __attribute__((always_inline)) -> int {
        __syncthreads();      // both waves are done with this document's LDS; the next ticket has long been parked in it
        const int t = uni(tk_slot[tk_par]);
        tk_par ^= 1;
        return t;
    };
    for (int tk = blockIdx.x;; tk = next_doc()) {
        if ((int64_t)tk >= P_arg.count) return;   // (uniform over the workgroup)
        // (the parameter block is the kernel's one argument: offset 0 of the kernel-argument segment)
        using ParamT = std::conditional_t<PERSIST, const __attribute__((address_space(4))) SolverParams, const SolverParams>;
        ParamT *Pk;
        unsigned tid = threadIdx.x;
        if constexpr (PERSIST) {
            Pk = (ParamT *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(Pk), "+v"(tid));
        } else Pk = &P_arg;
        ParamT &P = *Pk;
        const int lane = tid & (WAVE - 1);
        const int wv = NW == 2 ? uni((int)(tid >> 6)) : 0;   // wave-uniform, and known to the compiler as such (scalar branches)
        const int K = P.K, n = P.n, ld = P.ld, KP = P.KP;
        double *slab = GLOBAL_SLAB ? P.slab_beta + (size_t)blockIdx.x * (size_t)(KP + 2) * ld : dyn_lds;
        // DIRECT: dyn_lds = tile[TWS][KP] | crow[ld] | wrow[ld] | sidx[ld] (int32)
        double *crow = slab + (size_t)KP * (DIRECT ? TWS : ld);  // counts of the slab words
        // the HBM slab is topic-major (slab[k][word]: a wave's read of one topic for 64 words is one coalesced run), the LDS
        // slab word-major (slab[word][KP]: a lane streams its own row with ds_read_b128)
        auto SI = [&](int vv, int k) __attribute__((always_inline)) -> size_t { return GLOBAL_SLAB ? (size_t)k * ld + vv : (size_t)vv * KP + k; };
        double *wrow = crow + ld;               // counts / colsum(beta_d)
        int32_t *sidx = reinterpret_cast<int32_t *>(wrow + ld);   // DIRECT: word ids of the document
        // BFGS inverse-Hessian estimate (n x n): in LDS behind the slab for the two-wave form (its slab is
        // small), in a private global slab otherwise
        double *Hs = (NW == 2) ? dyn_lds + (size_t)(KP + 2) * ld : P.slab_H + (size_t)blockIdx.x * (size_t)n * n;
        const double *S = P.siginv;
        const bool sdiag = P.siginv_diag != 0;
        const int64_t ticket = P.first + tk;
        int tk_next = 0;
        if (PERSIST && persist && tid == WAVE * (NW - 1)) tk_next = atomicAdd(P.ticket_ctr, 1) + (int)gridDim.x;
        if (tid < 64) ss[tid] = 0.0;   // (read only after the document's first workgroup barrier)
        if (P.debug_flags & 8) {   // tests: nothing may depend on what an earlier workgroup left in the LDS
            for (int q = tid; q < P.lds_doubles; q += WAVE * NW) dyn_lds[q] = __builtin_nan("");
            for (int q = tid; q < KMAX + 1; q += WAVE * NW) { sv[q] = __builtin_nan(""); sw[q] = __builtin_nan(""); se[q] = __builtin_nan(""); }
            __syncthreads();
        }
        const long long t_begin = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;   // set-up (gather, g0) counts as init
        // the document's header: one 16-byte scalar load (order -> indptr would be two dependent round trips before the first index load)
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
        const int NdL = (Nd > VREG && wv == NW - 1) ? Nd - VREG : 0;  // words in the slab (<= ld): the last wave's
        if (STM_PROF(P) && tid == 0) STM_PROF(P)[doc * PROF_SLOTS + 39] = (long long)wall_clock64();   // (100 MHz, the same on every XCD: the XCDs' shader clocks are not aligned; tools/slot_gaps.py)
        const int asp = P.aspect ? scalar_load(P.aspect + doc) : 0;
        const double *bT = P.betaT + (size_t)asp * (size_t)P.V * K;

        // lane vectors first: their loads fly while the beta rows are gathered
        double x[VPL], g[VPL], p[VPL], xt[VPL], gv[VPL], mu[VPL], sd[VPL], g0[VPL];
#pragma unroll
        for (int r = 0; r < VPL; ++r) {
            const int i = lane + WAVE * r;
            const bool act = i < n;
            x[r] = act ? P.eta[doc * n + i] : 0.0;
            mu[r] = act ? P.mu[doc * n + i] : 0.0;
            sd[r] = act ? S[(size_t)i * n + i] : 0.0;
            g[r] = 0.0; p[r] = 0.0; xt[r] = 0.0; gv[r] = 0.0; g0[r] = 0.0;
        }

        // ---- gather beta_d (stm.py:614-617), lane = word; assert beta >= 0 (stm.py:534)
        double csum = 0.0;
        bool bad = false;
        double breg[KR];   // beta_d[:, word] of word `wv*64 + lane` (KREG > 0)
        if constexpr (PERSIST) {   // (the DMA form fills them under a lane mask: without a definition of their own at the top of a document the
#pragma unroll                     // registers would count as live around the document loop's back edge -- 100 registers the set-up phase cannot use)
            for (int k = 0; k < KR; ++k) breg[k] = 0.0;
        }
        double c0 = 0.0, w0 = 0.0;
        const int wreg = wv * WAVE + lane;  // this lane's register-resident word
        constexpr bool COOP = !GLOBAL_SLAB && VPL == 1;   // LDS slab: its rows are gathered lane = topic, one coalesced run per word
        constexpr int SB = 24;                             // slab rows fetched together with the register-resident rows (all of them at ~150 words)
        const bool act = KREG > 0 && wreg < Nd;
        // every index first (one memory round trip), then every row that fits in flight (a second one)
        const int idx_reg = act ? P.indices[p0 + wreg] : 0;
        const int idx_slab = (COOP && lane < NdL) ? P.indices[p0 + VREG + lane] : 0;
        const double cnt_slab = (COOP && lane < NdL) ? P.counts[p0 + VREG + lane] : 0.0;
        double cs0 = 1.0;
        const double *csv = P.colsum + (size_t)asp * (size_t)P.V;
        if (act) { c0 = P.counts[p0 + wreg]; cs0 = csv[idx_reg]; }
        const double cs_slab = (COOP && lane < NdL) ? csv[idx_slab] : 1.0;
        long long t_g1 = 0;
        double csum_all = 0.0;
        STM_UD(ss, Ndoc);
        if constexpr (DMA) {
            // The host selects this form only when K == KREG, KP == 2 * CHP, and every row (and the zero row behind the
            // last one, P.zrow) is within 4 GiB of P.betaT.
            constexpr int CH = KREG / 2, CHP = CH | 1, RPI = WAVE / CHP, PITCH = 2 * CHP;   // 16-byte pieces per row (odd: conflict-free b128 reads), rows per fetch, doubles per staged row
            // Staging rows per wave and phase: 32 -- TWO round trips to the L2 for the wave's 64 register rows (three with the 24
            // rows that fit a half of the BFGS matrix' area; a round trip is ~3 k cycles under load and nothing overlaps it).  The
            // two waves' staging areas reach DMA_STAGE_EXTRA doubles beyond that area: the host sizes the dynamic LDS for it.
            constexpr int RW = 32;
            static_assert(RW >= 8 && RW % RPI == 0 && WAVE % 8 == 0, "staging area");
            constexpr unsigned long long FULL = RPI * CHP >= 64 ? ~0ull : ((1ull << (RPI * CHP)) - 1ull);
            const unsigned K8 = 8u * (unsigned)K;
            const int rr = lane / CHP, cc = lane - rr * CHP;
            const unsigned coff = 16u * (unsigned)(cc < CH ? cc : CH - 1);                 // (the padding piece repeats the row's last one)
            const unsigned arow = (unsigned)asp * (unsigned)P.V;
            // byte offset of the row of this lane's register word / slab word; words the document does not have: the zero row
            const unsigned off_reg = (act ? arow + (unsigned)idx_reg : (unsigned)P.zrow) * K8;
            const unsigned off_slab = (lane < NdL ? arow + (unsigned)idx_slab : (unsigned)P.zrow) * K8;
            double *stg = Hs + wv * (RW * PITCH);
            const unsigned stg_lds = lds_addr(stg), slab_lds = lds_addr(slab);
            // the rows of the words held by lanes first .. first + RPI - 1 of `offv`, packed, to LDS address dst
            auto fetch = [&](unsigned offv, int first, unsigned dst, unsigned long long mask) __attribute__((always_inline)) {
                const unsigned o = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (first + rr), (int)offv) + coff;
                lds_dma16(P.betaT, o, dst, mask);
            };
            // c / colsum of this wave's register words (sv / sw are free until the first evaluation); the choice is made
            // opaque, or every read through it becomes two reads and a select
            __attribute__((address_space(3))) double *wq = (__attribute__((address_space(3))) double *)(wv == 0 ? sv : sw);
            asm volatile("" : "+v"(wq));
            const int nsl = NdL < WAVE ? NdL : WAVE;  // slab words that pair up with this wave's register words in the sums
            const int kk = lane < K ? lane : K - 1;   // lane = topic
            double Qc = 0.0, Oc = 0.0, Sc = 0.0, Pc = 0.0, T = 0.0;   // the reduction tree's pending sums of 4 / 8 / 16 / 32 words
#pragma unroll
            for (int ph = 0; ph * RW < WAVE; ++ph) {
                const int r0 = ph * RW, nr = (WAVE - r0 < RW) ? WAVE - r0 : RW;
#pragma unroll
                for (int i = 0; i < nr; i += RPI) fetch(off_reg, r0 + i, stg_lds + (unsigned)(i * PITCH * 8), FULL);
                if (ph == 0) {
                    // the slab rows straight to their place (lane-packed rows: pitch KP = PITCH)
                    for (int j = 0; j < nsl; j += RPI) {
                        const int rows = nsl - j < RPI ? nsl - j : RPI;
                        fetch(off_slab, j, slab_lds + (unsigned)(j * PITCH * 8), rows == RPI ? FULL : ((1ull << (rows * CHP)) - 1ull));
                    }
                    for (int v0 = WAVE; v0 < NdL; v0 += WAVE) {   // documents beyond 192 words (rare)
                        const int cnt = NdL - v0 < WAVE ? NdL - v0 : WAVE;
                        const unsigned o2 = (lane < cnt ? arow + (unsigned)P.indices[p0 + VREG + v0 + lane] : (unsigned)P.zrow) * K8;
                        for (int j = 0; j < cnt; j += RPI) {
                            const int rows = cnt - j < RPI ? cnt - j : RPI;
                            fetch(o2, j, slab_lds + (unsigned)((v0 + j) * PITCH * 8), rows == RPI ? FULL : ((1ull << (rows * CHP)) - 1ull));
                        }
                    }
                }
                wait_vmem();
                STM_WAVE_SYNC();
                if (ph == 0) {
                    // assert beta >= 0 (stm.py:534): a row with a negative or NaN entry carries a NaN column sum (beta_colsum_kernel)
                    if (act) { w0 = c0 / cs0; csum += c0; bad |= !(cs0 >= 0.0); }
                    wq[lane] = w0;
                    const double wsl = (lane < NdL) ? cnt_slab / cs_slab : 0.0;
                    if (wv == 1) svb[lane] = wsl;
                    for (int vv = lane; vv < NdL; vv += WAVE) {
                        const double c = vv < WAVE ? cnt_slab : P.counts[p0 + VREG + vv];
                        const double colsum = vv < WAVE ? cs_slab : csv[P.indices[p0 + VREG + vv]];
                        bad |= !(colsum >= 0.0);
                        crow[vv] = c;
                        wrow[vv] = vv < WAVE ? wsl : c / colsum;
                        csum += c;
                    }
                    STM_WAVE_SYNC();
                }
                if (lane >= r0 && lane < r0 + nr) {
                    const double2 *row2 = reinterpret_cast<const double2 *>(stg + (lane - r0) * PITCH);
#pragma unroll
                    for (int k = 0; k < KR; k += 2) {
                        const double2 t = row2[k >> 1];
                        breg[k] = t.x;
                        breg[k + 1] = t.y;
                    }
                }
                // g0 = beta_d @ (c / colsum) (stm.py:954), lane = topic: the leaves and the balanced tree of wave_sum() over the
                // wave's 64 words -- the same additions in the same pairing as the cross-lane form, so the same bits.  Four
                // words at a time (the scheduler would otherwise hoist a whole phase's LDS reads above beta_d's registers).
#if STM_G0_PIPE
                // software-pipelined: the operands of the next four words are read while the current four are multiplied and added (each
                // step otherwise waits out its own round trip to the LDS, sixteen steps per wave)
                double pa[2][4], pw[2][4], ps[2][4], pb[2][4];
                auto g0_load = [&](int j0, int b) __attribute__((always_inline)) {
                    const int wi0 = r0 + j0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { pa[b][u] = stg[(j0 + u) * PITCH + kk]; pw[b][u] = wq[wi0 + u]; }
                    if (wv == 1 && wi0 < NdL) {   // uniform: these words pair up with slab words wi0 .. (beyond the last: weight 0)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sr = wi0 + u < nsl ? wi0 + u : nsl - 1;
                            ps[b][u] = slab[(size_t)sr * KP + kk]; pb[b][u] = svb[wi0 + u];
                        }
                    }
                };
                g0_load(0, 0);
#pragma unroll
                for (int j0 = 0; j0 < nr; j0 += 4) {
                    const int wi0 = r0 + j0, qi = wi0 >> 2, b = (j0 >> 2) & 1;
                    if (j0 + 4 < nr) g0_load(j0 + 4, b ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    double v[4];
                    if (wv == 1 && wi0 < NdL) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { v[u] = pa[b][u] * pw[b][u]; v[u] = v[u] + ps[b][u] * pb[b][u]; }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = pa[b][u] * pw[b][u];
                    }
#else
#pragma unroll
                for (int j0 = 0; j0 < nr; j0 += 4) {
                    const int wi0 = r0 + j0, qi = wi0 >> 2;
                    double v[4];
                    __builtin_amdgcn_sched_barrier(0);
                    if (wv == 1 && wi0 < NdL) {   // uniform: these words pair up with slab words wi0 .. (beyond the last: weight 0)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sr = wi0 + u < nsl ? wi0 + u : nsl - 1;
                            const double sl = slab[(size_t)sr * KP + kk];
                            v[u] = stg[(j0 + u) * PITCH + kk] * wq[wi0 + u];
                            v[u] = v[u] + sl * svb[wi0 + u];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = stg[(j0 + u) * PITCH + kk] * wq[wi0 + u];
                    }
#endif
                    const double Q = (v[0] + v[1]) + (v[2] + v[3]);
                    if ((qi & 1) == 0) Qc = Q;
                    else {
                        const double O = Qc + Q;
                        if ((qi & 2) == 0) Oc = O;
                        else {
                            const double S = Oc + O;
                            if ((qi & 4) == 0) Sc = S;
                            else {
                                const double Pn = Sc + S;
                                if ((qi & 8) == 0) Pc = Pn; else T = Pc + Pn;
                            }
                        }
                    }
#if STM_G0_PIPE
                    asm volatile("" : "+v"(Qc), "+v"(Oc), "+v"(Sc), "+v"(Pc), "+v"(T));
#endif
                }
                wait_lds();   // every read of the staging area has returned before the next rows are fetched into it
                STM_WAVE_SYNC();
            }
            g0[0] = (lane < n) ? T : 0.0;
            if (NdL > WAVE) {   // slab words beyond the first 64 (uniform; rare)
                for (int k = 0; k < n; ++k) {
                    double t = 0.0;
                    for (int vv = lane + WAVE; vv < NdL; vv += WAVE) t += slab[SI(vv, k)] * wrow[vv];
                    t = wave_sum(t);
                    if (lane == k) g0[0] += t;
                }
            }
            t_g1 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
            if (STM_PROF(P) && wv == 1 && lane == 0) STM_PROF(P)[doc * PROF_SLOTS + 7] = t_g1 - t_begin;
        }
        if constexpr (!DMA) {
        if (KREG > 0) {
            // unconditional loads of KREG doubles from the row start (the buffer is padded; rows are 16-byte aligned for even K),
            // zeros beyond K by selection: no branch per topic
            const double *row = bT + (size_t)idx_reg * K;
            if ((K & 1) == 0) {
                const double2 *row2 = reinterpret_cast<const double2 *>(row);
#pragma unroll
                for (int k = 0; k < KR; k += 2) {
                    const double2 t = row2[k >> 1];
                    breg[k] = (act && k < K) ? t.x : 0.0;
                    breg[k + 1] = (act && k + 1 < K) ? t.y : 0.0;
                }
            } else {
                double tmp[KR];
#pragma unroll
                for (int k = 0; k < KR; ++k) tmp[k] = row[k];
#pragma unroll
                for (int k = 0; k < KR; ++k) breg[k] = (act && k < K) ? tmp[k] : 0.0;
            }
        }
        double bv0[COOP ? SB : 1];
        if (COOP && NdL > 0) {   // one block of SB loads (unused slots fetch word 0's row), issued behind the register rows
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = __builtin_amdgcn_readlane(idx_slab, u);
                bv0[u] = bT[(size_t)idx * K + (lane < K ? lane : 0)];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KREG > 0) {
#pragma unroll
            for (int k = 0; k < KR; ++k) bad |= !(breg[k] >= 0.0);
            if (act) {
                w0 = c0 / cs0;
                csum += c0;
            }
        }
        if (COOP) {
#pragma unroll
            for (int u = 0; u < SB; ++u)
                if (u < NdL) {   // uniform
                    bad |= !(bv0[u] >= 0.0);
                    if (lane < K) slab[(size_t)u * KP + lane] = bv0[u];
                }
            // the rest of the slab, eight rows in flight
            for (int v0 = 0; v0 < NdL; v0 += WAVE) {
                const int cnt = NdL - v0 < WAVE ? NdL - v0 : WAVE;
                const int my_idx = v0 == 0 ? idx_slab : (lane < cnt ? P.indices[p0 + VREG + v0 + lane] : 0);
                for (int j0 = (v0 == 0 ? SB : 0); j0 < cnt; j0 += 8) {
                    double bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = __builtin_amdgcn_readlane(my_idx, (j0 + u) & (WAVE - 1));
                        bv[u] = (lane < K && j0 + u < cnt) ? bT[(size_t)idx * K + lane] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < cnt) {   // uniform
                            bad |= !(bv[u] >= 0.0);
                            if (lane < K) slab[(size_t)(v0 + j0 + u) * KP + lane] = bv[u];
                        }
                }
            }
            // the word counts now (the other wave waits for their sum); the column sums of the slab rows after the exchange
            for (int vv = lane; vv < NdL; vv += WAVE) csum += vv < WAVE ? cnt_slab : P.counts[p0 + VREG + vv];
        } else if constexpr (DIRECT) {
            for (int vv = lane; vv < NdL; vv += WAVE) {
                sidx[vv] = P.indices[p0 + vv];
                const double c = P.counts[p0 + vv];
                crow[vv] = c;
                csum += c;
            }
        } else
        for (int vv = lane; vv < NdL; vv += WAVE) {
            const int idx = P.indices[p0 + VREG + vv];
            const double c = P.counts[p0 + VREG + vv];
            const double *row = bT + (size_t)idx * K;
            double colsum = 0.0;
            int k = 0;
            for (; k + 7 < K; k += 8) {   // eight loads of the lane's row in flight; the column sum keeps its order
                double b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = row[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    bad |= !(b[u] >= 0.0);
                    slab[SI(vv, k + u)] = b[u];
                    colsum += b[u];
                }
            }
            for (; k < K; ++k) {
                const double b = row[k];
                bad |= !(b >= 0.0);
                slab[SI(vv, k)] = b;
                colsum += b;
            }
            for (k = K; k < KP; ++k) slab[SI(vv, k)] = 0.0;
            crow[vv] = c;
            wrow[vv] = c / colsum;
            csum += c;
        }
        t_g1 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
        if (STM_PROF(P) && NW == 2 && wv == 1 && lane == 0) STM_PROF(P)[doc * PROF_SLOTS + 7] = t_g1 - t_begin;   // wave 1: register rows + slab
        }   // !DMA
        // se[k] stays 0 for k >= K (the register pass is unrolled to KREG)
        if (wv == 0) {
            for (int i = lane; i < KMAX + 2; i += WAVE) se[i] = 0.0;
            if (lane < 2) { sv[KMAX + lane] = 0.0; sw[KMAX + lane] = 0.0; }   // the moment pass reads all three up to KP
        }
        csum_all = wave_sum(csum);
        bool bad_all = wave_any(bad);
        if (PERSIST && persist && tid == WAVE * (NW - 1)) tk_slot[tk_par] = tk_next;   // (the rows' waits have covered the add's return)
        if (NW == 2) {
            if (lane == 0) { xch_res[4 + wv] = csum_all; xch_cmd[1 + wv] = bad_all ? 1 : 0; }
            __syncthreads();
            csum_all = xch_res[4] + xch_res[5];
            bad_all = (xch_cmd[1] | xch_cmd[2]) != 0;
        } else {
            __syncthreads();  // slab stores -> visible to the whole wave
        }
        if (bad_all) {
            // (a persistent workgroup drops the tickets it would still have taken: STM_ERR_BETA is fatal to the E-step -- the host raises
            // the reference's assert, stm.py:534 -- so eta / status / nit of the documents left untouched are never looked at)
            atomicMax(P.err_flag, 2 /* STM_ERR_BETA */);
            return;
        }
        Ndoc = (double)(long long)csum_all;  // int(np.sum(word_count)), stm.py:933
#if STM_C0_LDS
        // DMA form: the count of the lane's register word waits in the LDS, behind the BFGS matrix, where the staging rows of the set-up
        // were (both waves are past them: the barrier above) -- an evaluation uses it once, at its very end, and two registers held
        // through the whole document for that were what the allocator spilled (read back from scratch in front of every wave sum)
        // (its address is formed anew at every use: held in a register it was spilled in c0's place)
        if constexpr (DMA) Hs[(size_t)n * n + tid] = c0;
        auto c0_lds = [&]() __attribute__((always_inline)) -> double {
            unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // the lane, from the hardware (two instructions;
            asm volatile("" : "+v"(l));                                                        //  the thread index itself sits in scratch by now)
            return Hs[(size_t)n * n + WAVE * wv + l];
        };
#define STM_C0 (DMA ? c0_lds() : c0)
#else
#define STM_C0 c0
#endif
        if (!DMA && COOP && NdL > 0) {
            // lane = word again for the column sums of the slab rows (same order of additions as the per-lane gather loop);
            // only this wave reads crow / wrow before the next hand-off
            STM_WAVE_SYNC();
            for (int vv = lane; vv < NdL; vv += WAVE) {
                const double c = vv < WAVE ? cnt_slab : P.counts[p0 + VREG + vv];
                const double colsum = vv < WAVE ? cs_slab : csv[P.indices[p0 + VREG + vv]];
                double *dst = slab + (size_t)vv * KP;
                for (int k = K; k < KP; ++k) dst[k] = 0.0;
                crow[vv] = c;
                wrow[vv] = c / colsum;
            }
            STM_WAVE_SYNC();
        }

        const long long t_g2 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
        // ---- DIRECT: one LDS tile of TWS rows, fetched from betaT for every pass over the document, one tile ahead of its use.
        // Lane l loads components 2 l and 2 l + 1 of a row: ONE 16-byte load per row and lane (K <= 128: the whole row in one
        // instruction, scalar base + 32-bit lane offset -- a level of beta stays below 4 GiB, stm_set_topics) and one 16-byte LDS
        // store (rounds 2-3: components l and l + 64, two masked loads and up to three stores per row -- the sweep spent 2.5 k cycles
        // per tile issuing them).  The fetch is one straight line: no select behind a load (it would wait for it, row by row) and no
        // branch per row (the wait-count bookkeeping gives up across them): rows beyond the document repeat its last word and are
        // zeroed when the tile is STORED, the lanes beyond a row never load (their registers keep the zeros a sweep starts with),
        // and an odd K's last lane -- which reads 8 bytes into the next row (or the padding behind betaT) -- drops its second
        // component at the store as well.
        double2 pre[DIRECT ? TWS : 1];
        if constexpr (PERSIST && DIRECT) {   // (loaded under a lane mask: defined here, or the registers stay live around the document loop's back edge -- see breg)
#pragma unroll
            for (int j = 0; j < TWS; ++j) pre[j] = make_double2(0.0, 0.0);
        }
        const bool has0 = 2 * lane < K, has1 = 2 * lane + 1 < K;
        auto tile_fetch = [&](int t0) __attribute__((always_inline)) {
            if constexpr (DIRECT) {
                if (t0 == 0) {   // a sweep starts (uniform): what the lanes beyond a row store for the whole sweep
#pragma unroll
                    for (int j = 0; j < TWS; ++j) pre[j] = make_double2(0.0, 0.0);
                }
                const int wi = t0 + lane < NdL ? t0 + lane : NdL - 1;
                const int my = (lane < TWS && NdL > 0) ? sidx[wi] : 0;   // (an empty document fetches row 0 and never looks at it)
                const unsigned l16 = 16u * (unsigned)lane, K8 = 8u * (unsigned)K;
                const char *base = reinterpret_cast<const char *>(bT);
                if (has0) {
#pragma unroll
                    for (int j = 0; j < TWS; ++j) {
                        const unsigned o = (unsigned)__builtin_amdgcn_readlane(my, j) * K8 + l16;
                        double2 v;
                        __builtin_memcpy(&v, base + o, sizeof(v));      // (8-byte aligned for odd K: global_load_dwordx4 takes it)
                        pre[j] = v;
                    }
                }
            }
        };
        auto tile_store = [&](int nw) __attribute__((always_inline)) {   // pre -> tile[j][2 lane], tile[j][2 lane + 1]; zeros beyond K and beyond the document
            if constexpr (DIRECT) {
                const bool odd = K & 1;   // uniform
#pragma unroll
                for (int j = 0; j < TWS; ++j) {
                    const bool in = j < nw;   // uniform
                    pre[j] = make_double2(in ? pre[j].x : 0.0, (in && (!odd || has1)) ? pre[j].y : 0.0);
                }
                if (2 * lane < KP) {                                     // (KP is even)
#pragma unroll
                    for (int j = 0; j < TWS; ++j) *reinterpret_cast<double2 *>(slab + (size_t)j * KP + 2 * lane) = pre[j];
                }
                // K = 127, 128 (KP = 130): the padding of row `lane`, beyond what 64 lanes cover
                if (KP > 2 * WAVE && lane < TWS) *reinterpret_cast<double2 *>(slab + (size_t)lane * KP + 2 * WAVE) = make_double2(0.0, 0.0);
            }
        };
        // g0 / v of the sweeps below: acc[r] += tile[w][lane + 64 r] * wts[t0 + w] over the tile's words in order, four words' LDS reads in
        // flight together and no branch on the lane (lanes without a second topic repeat their first one; their sums are never used).
        // Words beyond the document are rows of zeros in the tile and get the weight 0 (wts is not read beyond the document).
        auto tile_axpy = [&](const double *wts, int t0, int nw, double (&acc)[VPL]) __attribute__((always_inline)) {
            if constexpr (DIRECT) {
                const int c1 = (lane + WAVE < KP) ? lane + WAVE : lane;
#pragma unroll
                for (int w0 = 0; w0 < TWS; w0 += 4) {
                    if (w0 >= nw) break;   // uniform
                    double wq[4], b0[4], b1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int w = w0 + u;
                        const double t = wts[t0 + (w < nw ? w : nw - 1)];
                        wq[u] = w < nw ? t : 0.0;
                        b0[u] = slab[(size_t)w * KP + lane];
                        b1[u] = slab[(size_t)w * KP + c1];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[0] = fma(b0[u], wq[u], acc[0]);
                        acc[1] = fma(b1[u], wq[u], acc[1]);
                    }
                }
            }
        };
        // lane = (word lane & 15, quarter lane >> 4 of the row): the lane's share of sum_k f(k) over the row, the four
        // quarters combined by two exchanges -- every lane of a word ends up with the word's total
        const int dw = lane & 15, dq = lane >> 4;
        const int kq2 = ((KP >> 1) + 3) >> 2;                       // 16-byte pieces per quarter
        // DIRECT: ONE sweep over the document's rows for g0, f(x0)'s data term and the moment test's vector (after eval_DF below)
        const bool fuse0 = DIRECT && !(P.debug_flags & (1 | 2 | 16));
        if constexpr (DIRECT) {
            if (NdL > 0) STM_WAVE_SYNC();                            // sidx / crow visible to the wave
            // c / colsum(beta_d) for every word up front (the column sum is a property of the word, P.colsum): inside the tile
            // loop each tile paid a dependent colsum load and an IEEE division before its first FMA
#pragma unroll 4
            for (int vv = lane; vv < NdL; vv += WAVE) {
                const double cs = csv[sidx[vv]];
                bad |= !(cs >= 0.0);     // assert beta >= 0 (stm.py:534): NaN column sum <=> the row has a negative or NaN entry
                wrow[vv] = crow[vv] / cs;
            }
            if (NdL > 0) STM_WAVE_SYNC();
            double g0a[VPL];
#pragma unroll
            for (int r = 0; r < VPL; ++r) g0a[r] = 0.0;
            if (!fuse0) tile_fetch(0);   // (fuse0: g0 comes out of the one sweep that also serves f(x0) and the moment test, below)
            for (int t0 = 0; !fuse0 && t0 < NdL; t0 += TWS) {
                const int nw = NdL - t0 < TWS ? NdL - t0 : TWS;
                tile_store(nw);
                STM_WAVE_SYNC();
                if (t0 + TWS < NdL) tile_fetch(t0 + TWS);
                tile_axpy(wrow, t0, nw, g0a);   // g0 += beta_d[:, tile] @ (c / colsum), lane = topic
                STM_WAVE_SYNC();
            }
#pragma unroll
            for (int r = 0; r < VPL; ++r) g0[r] = (lane + WAVE * r < n) ? g0a[r] : 0.0;
            if (wave_any(bad)) {   // "Some entries of beta are negative or nan." (stm.py:534)
                atomicMax(P.err_flag, 2 /* STM_ERR_BETA */);
                return;
            }
        }
        // g0 = beta_d @ (c / colsum(beta_d)) -- the eta-independent data term of df (stm.py:954)
        auto g0_slab = [&](int k) __attribute__((always_inline)) -> double {
            double t = 0.0;
            for (int vv = lane; vv < NdL; vv += WAVE) t += slab[SI(vv, k)] * wrow[vv];
            return t;
        };
        auto g0_put = [&](int k, double t) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < VPL; ++r)
                if (k == lane + WAVE * r) g0[r] = t;
        };
        if constexpr (DIRECT || DMA) {
            // done above (DIRECT: tile by tile; DMA: from the staged rows)
        } else if (KREG > 0 && VPL == 1) {
            // wave_sum()'s additions in wave_sum()'s order, but the four row totals of every topic are combined for all
            // topics at once: after the intra-row steps lane (row r, position c) keeps the row-r total of topic 16 q + c
            // in acc[q]; two cross-row exchanges per q finish the sums, and lane k picks topic k.
            double accq[4] = {0.0, 0.0, 0.0, 0.0};
            const int c16 = lane & 15;
            // straight-line code for all KREG topics (rows K.. of the registers are zeros, topic K-1 is computed and dropped),
            // so that the KREG reduction chains interleave; three forms of the slab term: none (this wave owns no slab
            // words), one word per lane (at most 64 slab words), the general strided loop
            const double wl = (lane < NdL) ? wrow[lane] : 0.0;      // the lane's slab word among the first 64
            const int srow_w = (lane < NdL) ? lane : 0;
            auto chains = [&](auto mode) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    double v = breg[k] * w0;
                    if (decltype(mode)::value == 1) v = v + ((lane < NdL) ? slab[SI(srow_w, k)] * wl : 0.0);
                    if (decltype(mode)::value == 2) v = v + g0_slab(k);
                    v += dpp_move<DPP_XOR1>(v);
                    v += dpp_move<DPP_XOR2>(v);
                    v += dpp_move<DPP_HALF_MIRROR>(v);
                    v += dpp_move<DPP_MIRROR>(v);
                    if ((k & 15) == c16) accq[k >> 4] = v;
                }
            };
            // ONE instantiation of the 50 chains (each mode is ~7 KB of straight-line code, and the kernel has to share a
            // 64 KB instruction cache with the waves of two CUs): wave 0 owns no slab words (mode 0); the last wave adds its
            // first 64 slab words inside the chains (mode 1) and any further ones (documents beyond 192 words) in the
            // rolled loop below
            if (NW == 2 && wv == 0) chains(std::integral_constant<int, 0>());
            else chains(std::integral_constant<int, 1>());
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 16 < KR) {
                    double a = accq[q];
                    a += __shfl_xor(a, 16);       // rows (0,1) and (2,3): lane_bcast(v, 0) + lane_bcast(v, 16) ...
                    a += __shfl_xor(a, 32);       // ... + (lane_bcast(v, 32) + lane_bcast(v, 48))
                    accq[q] = a;
                }
            const int q = lane >> 4;
            g0[0] = (lane < n) ? (q == 0 ? accq[0] : q == 1 ? accq[1] : q == 2 ? accq[2] : accq[3]) : 0.0;
            if (NdL > WAVE)   // slab words beyond the first 64 (uniform; rare): one rolled loop over the topics
                for (int k = 0; k < n; ++k) {
                    double t = 0.0;
                    for (int vv = lane + WAVE; vv < NdL; vv += WAVE) t += slab[SI(vv, k)] * wrow[vv];
                    t = wave_sum(t);
                    if (lane == k) g0[0] += t;
                }
        } else if (KREG > 0) {
#pragma unroll
            for (int k = 0; k < KR; ++k)
                if (k < n) g0_put(k, wave_sum(breg[k] * w0 + g0_slab(k)));
        } else {
            for (int k = 0; k < n; k += 8) {   // eight topics' reductions interleave (each is wave_sum(), unchanged)
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = 0.0;
                for (int vv = lane; vv < NdL; vv += WAVE) {
                    const double wq = wrow[vv];
                    double b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) b[u] = slab[SI(vv, k + u < KP ? k + u : k)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] += b[u] * wq;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = wave_sum(t[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k + u < n) g0_put(k + u, t[u]);
            }
        }

        const long long t_g3 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
        if (STM_PROF(P) && lane == 0 && wv == 0) { STM_PROF(P)[doc * PROF_SLOTS + 4] = t_g1 - t_begin; STM_PROF(P)[doc * PROF_SLOTS + 5] = t_g2 - t_g1; STM_PROF(P)[doc * PROF_SLOTS + 6] = t_g3 - t_g2; }
        long long t_init = 0, t_eval = 0, t_sm = 0, t_upd = 0;
        int nfev = 0, njev = 0;
        double *svx = (NW == 2 && wv == 1) ? svb : sv;  // this wave's private broadcast vector

        // ---- pieces of f(eta), stm.py:920-944
        // m = max(eta~), exp(eta~ - m) -> se[], multiplicity of the maximum, sum of the other exponentials
        auto head_F = [&](double &m_out, int &icnt_out, double &ssum_out, double &e_out) __attribute__((always_inline)) {
            double mloc = 0.0;  // the appended 0 of eta~
#pragma unroll
            for (int r = 0; r < VPL; ++r)
                if (lane + WAVE * r < n) mloc = nanmax(mloc, xt[r]);
            const double m = wave_nanmax(mloc);
            int icnt = 0;
            double ssum = 0.0;
#pragma unroll
            for (int r = 0; r < VPL; ++r) {
                const int i = lane + WAVE * r;
                const double val = (i < n) ? xt[r] : 0.0;
                const double e = exp(val - m);
                if (i < K) se[i] = e;
                if (r == 0) e_out = (i < K) ? e : 0.0;
                const bool ismax = (i < K) && (val == m);
                icnt += __popcll(__ballot(ismax));
                ssum += (i < K && !ismax) ? e : 0.0;
            }
            m_out = m; icnt_out = icnt; ssum_out = ssum;
        };
        // scipy.special.logsumexp: log1p(s/m) + log(m) + a_max (m = multiplicity of the maximum)
        auto lse_F = [&](double m, int icnt, double ssum_lane) __attribute__((always_inline)) -> double {
            const double ssum = wave_sum(ssum_lane);
            if (icnt == 1) return log1p_pos(ssum) + m;  // s/1, + log(1)
            const double cnt = (double)icnt;
            return log1p(ssum != 0.0 ? ssum / cnt : ssum) + log(cnt) + m;
        };
        auto lse_tail = [&](double m, int icnt, double ssum) __attribute__((always_inline)) -> double {   // lse_F behind its wave sum
            if (icnt == 1) return log1p_pos(ssum) + m;
            const double cnt = (double)icnt;
            return log1p(ssum != 0.0 ? ssum / cnt : ssum) + log(cnt) + m;
        };
        // exp(eta~ - m) @ beta_d[:, word] for the lane's register word: data_F's two chains
        auto reg_dot = [&]() __attribute__((always_inline)) -> double {
            const double2 *se2 = reinterpret_cast<const double2 *>(se);
            double s0 = 0.0, s1 = 0.0;
#if STM_REG_PIPE
            constexpr int GI = STM_REG_PIPE, NI = KR / 2, NG = (NI + GI - 1) / GI;   // software-pipelined groups (see data_F)
            double2 eb[2][GI];
#pragma unroll
            for (int i = 0; i < GI; ++i) eb[0][i] = se2[i < NI ? i : 0];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < GI; ++i) eb[(g + 1) & 1][i] = se2[(g + 1) * GI + i < NI ? (g + 1) * GI + i : NI - 1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GI; ++i) {
                    const int q = g * GI + i;
                    if (q < NI) { s0 = fma(eb[g & 1][i].x, breg[2 * q], s0); s1 = fma(eb[g & 1][i].y, breg[2 * q + 1], s1); }
                }
                asm volatile("" : "+v"(s0), "+v"(s1));
            }
#else
#pragma unroll
            for (int k = 0; k < KR; k += 2) {
                if (k % STM_EVAL_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                const double2 e = se2[k / 2];
                s0 = fma(e.x, breg[k], s0);
                s1 = fma(e.y, breg[k + 1], s1);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            return s0 + s1;
        };
        // c . (m + log(exp(eta~ - m) @ beta_d)) restricted to this wave's words, per lane.
        // (e_lane: exp(eta~ - m) of topic `lane`, 0 beyond K -- kept for broadcast experiments.)
        auto data_F = [&](double m, double e_lane, const int NdL) __attribute__((always_inline)) -> double {  // NdL shadows: slab words to cover
            double part = 0.0;
            const double2 *se2 = reinterpret_cast<const double2 *>(se);
            if constexpr (DMA && (STM_FUSE_SLAB & 1)) {
                // The wave that owns the slab (at most one slab word per lane: documents up to 192 words): the lane's register word and
                // its slab word TOGETHER -- the same four FMA chains and the same two logarithms as below, one after the other there,
                // side by side here (they share the broadcast reads of exp(eta~ - m), and each chain fills the other's latencies; this
                // wave is the one the other waits for at the evaluation's last barrier).  K == KREG == KP in this form.
                if (NdL > 0 && NdL <= WAVE) {   // uniform
                    const int ia = lane < NdL ? lane : NdL - 1;
                    const double2 *ra = reinterpret_cast<const double2 *>(slab + (size_t)ia * KP);
                    double s0 = 0.0, s1 = 0.0, a0 = 0.0, a1 = 0.0;
#if STM_FUSE_PIPE
                    // software-pipelined: a group's LDS reads are in flight while the group before it is multiplied (left to itself every
                    // group waits out its own round trip to the LDS -- thirteen of them with four topics per group)
                    constexpr int GI = STM_FUSE_PIPE, NI = KR / 2, NG = (NI + GI - 1) / GI;
                    double2 eb[2][GI], rb[2][GI];
#pragma unroll
                    for (int i = 0; i < GI; ++i) { eb[0][i] = se2[i < NI ? i : 0]; rb[0][i] = ra[i < NI ? i : 0]; }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g + 1 < NG) {
#pragma unroll
                            for (int i = 0; i < GI; ++i) {
                                const int q = (g + 1) * GI + i < NI ? (g + 1) * GI + i : NI - 1;
                                eb[(g + 1) & 1][i] = se2[q]; rb[(g + 1) & 1][i] = ra[q];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < GI; ++i) {
                            const int q = g * GI + i;
                            if (q < NI) {
                                const double2 e = eb[g & 1][i], ba = rb[g & 1][i];
                                s0 = fma(e.x, breg[2 * q], s0);
                                s1 = fma(e.y, breg[2 * q + 1], s1);
                                a0 = fma(e.x, ba.x, a0);
                                a1 = fma(e.y, ba.y, a1);
                            }
                        }
                        asm volatile("" : "+v"(s0), "+v"(s1), "+v"(a0), "+v"(a1));
                    }
#else
#pragma unroll
                    for (int k = 0; k < KR; k += 2) {
                        if (k % STM_FUSE_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                        const double2 e = se2[k / 2], ba = ra[k / 2];
                        s0 = fma(e.x, breg[k], s0);
                        s1 = fma(e.y, breg[k + 1], s1);
                        a0 = fma(e.x, ba.x, a0);
                        a1 = fma(e.y, ba.y, a1);
                        // (the scheduler otherwise runs the register word's chains first and parks the slab rows in scratch)
                        if (k % STM_FUSE_GROUP == STM_FUSE_GROUP - 2 || k + 2 >= KR) asm volatile("" : "+v"(s0), "+v"(s1), "+v"(a0), "+v"(a1));
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    const double sums[2] = {s0 + s1, a0 + a1};
                    double lg[2];
#ifdef STM_NO_LOG2
                    lg[0] = log_pos(sums[0]); lg[1] = log_pos(sums[1]);
#else
                    log_pos2(sums, lg);
#endif
                    part = (wreg < Nd) ? STM_C0 * (m + lg[0]) : 0.0;
                    part += (lane < NdL) ? crow[ia] * (m + lg[1]) : 0.0;
                    return part;
                }
            }
            if (KREG > 0) {  // register-resident word: beta_d column in registers, two FMA chains
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < KR; k += 2) {
                    // bound the number of se values in flight (the scheduler would otherwise
                    // hoist all KREG broadcast reads and spill beta_d out of the VGPRs)
                    if (k % STM_EVAL_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                    const double2 e = se2[k / 2];
                    s0 = fma(e.x, breg[k], s0);
                    s1 = fma(e.y, breg[k + 1], s1);
                }
                __builtin_amdgcn_sched_barrier(0);
                const double lg = m + log_pos(s0 + s1);
                part = (wreg < Nd) ? STM_C0 * lg : 0.0;
            }
            if constexpr (DIRECT) {   // re-gather tile by tile; lane = (word, quarter of the topics)
                tile_fetch(0);
                const int k0 = dq * kq2, k1 = (dq + 1) * kq2 < (KP >> 1) ? (dq + 1) * kq2 : (KP >> 1);
                for (int t0 = 0; t0 < NdL; t0 += TWS) {
                    const int nw = NdL - t0 < TWS ? NdL - t0 : TWS;
                    tile_store(nw);
                    STM_WAVE_SYNC();
                    if (t0 + TWS < NdL) tile_fetch(t0 + TWS);
                    const double2 *tr = reinterpret_cast<const double2 *>(slab + (size_t)dw * KP);
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll 4
                    for (int kk = k0; kk < k1; ++kk) {
                        const double2 e = se2[kk], b = tr[kk];
                        a0 = fma(e.x, b.x, a0);
                        a1 = fma(e.y, b.y, a1);
                    }
                    double sdot = a0 + a1;
                    sdot += __shfl_xor(sdot, 16);
                    sdot += __shfl_xor(sdot, 32);
                    const double lg = m + log_pos(sdot);
                    part += (dq == 0 && dw < nw) ? crow[t0 + dw] * lg : 0.0;
                    STM_WAVE_SYNC();
                }
                return part;
            }
            // slab words: every lane streams its word's row (ds_read_b128), two 64-word tiles per
            // sweep so the broadcast se pair is read once for both
            const int kp2 = KP >> 1;
            for (int vb = 0; vb < NdL; vb += 2 * WAVE) {
                const int va = vb + lane, vc = va + WAVE;
                const bool two = vb + WAVE < NdL;  // uniform
                const int ia = va < NdL ? va : NdL - 1, ic = vc < NdL ? vc : NdL - 1;
                double a0 = 0.0, a1 = 0.0;
                if constexpr (GLOBAL_SLAB) {   // topic-major HBM slab: the same sums in the same order, coalesced loads
                    const double *pa = slab + ia, *pc = slab + ic;
                    double c0s = 0.0, c1s = 0.0;
#pragma unroll 4
                    for (int kk = 0; kk < kp2; ++kk) {
                        const double2 e = se2[kk];
                        const double bax = pa[(size_t)(2 * kk) * ld], bay = pa[(size_t)(2 * kk + 1) * ld];
                        const double bcx = pc[(size_t)(2 * kk) * ld], bcy = pc[(size_t)(2 * kk + 1) * ld];
                        a0 = fma(e.x, bax, a0);
                        a1 = fma(e.y, bay, a1);
                        c0s = fma(e.x, bcx, c0s);
                        c1s = fma(e.y, bcy, c1s);
                    }
                    const double la = m + log_pos(a0 + a1), lc = m + log_pos(c0s + c1s);
                    part += (va < NdL) ? crow[ia] * la : 0.0;
                    part += (two && vc < NdL) ? crow[ic] * lc : 0.0;
                    continue;
                }
                const double2 *ra = reinterpret_cast<const double2 *>(slab + (size_t)ia * KP);
                const double2 *rc = reinterpret_cast<const double2 *>(slab + (size_t)ic * KP);
                if (two) {
                    double c0s = 0.0, c1s = 0.0;
#pragma unroll 5
                    for (int kk = 0; kk < kp2; ++kk) {
                        const double2 e = se2[kk], ba = ra[kk], bc = rc[kk];
                        a0 = fma(e.x, ba.x, a0);
                        a1 = fma(e.y, ba.y, a1);
                        c0s = fma(e.x, bc.x, c0s);
                        c1s = fma(e.y, bc.y, c1s);
                    }
                    const double la = m + log_pos(a0 + a1), lc = m + log_pos(c0s + c1s);
                    part += (va < NdL) ? crow[ia] * la : 0.0;
                    part += (vc < NdL) ? crow[ic] * lc : 0.0;
                } else {
#pragma unroll 5
                    for (int kk = 0; kk < kp2; ++kk) {
                        const double2 e = se2[kk], ba = ra[kk];
                        a0 = fma(e.x, ba.x, a0);
                        a1 = fma(e.y, ba.y, a1);
                    }
                    const double la = m + log_pos(a0 + a1);
                    part += (va < NdL) ? crow[ia] * la : 0.0;
                }
            }
            return part;
        };
        // d^T siginv d, wave-summed; dv(r) = the lane's component lane + 64 r of d
        auto quad_of = [&](auto dv, double *svx) __attribute__((always_inline)) -> double {   // svx: the wave's broadcast vector (dense siginv)
            double q = 0.0;
            if (sdiag) {
#pragma unroll
                for (int r = 0; r < VPL; ++r)
                    if (lane + WAVE * r < n) {
                        const double d = dv(r);
                        q += (d * sd[r]) * d;
                    }
            } else {
#pragma unroll
                for (int r = 0; r < VPL; ++r)
                    if (lane + WAVE * r < n) svx[lane + WAVE * r] = dv(r);
                STM_WAVE_SYNC();
#pragma unroll
                for (int r = 0; r < VPL; ++r) {
                    const int i = lane + WAVE * r;
                    if (i < n) {
                        double t = 0.0;
                        for (int j = 0; j < n; ++j) t += svx[j] * S[(size_t)j * n + i];
                        q += t * svx[i];
                    }
                }
                STM_WAVE_SYNC();
            }
            return wave_sum(q);
        };
        // (eta - mu)^T siginv (eta - mu)
        auto quad_F = [&]() __attribute__((always_inline)) -> double {
            return quad_of([&](int r) __attribute__((always_inline)) -> double { return xt[r] - mu[r]; }, svx);
        };
        // Moment pass of the two-wave form (see S_OUTER_TOP): with e = exp(eta~ - m) in se[], e p~ in sv[] and e p~^2 in sw[],
        // the mean and the variance of p~ under q_w(k) ~ beta_d[k, w] e_k for each of this wave's words, summed with the word
        // counts: d1 = sum_w c_w E_{q_w}[p~] (the derivative of f's data term along p), d2 = sum_w c_w Var_{q_w}(p~) (its
        // second derivative).  Per lane; the caller wave-sums.  Feeds a sufficient condition only -- nothing scipy computes.
        constexpr bool MOM = !GLOBAL_SLAB;   // forms with a moment pass (the HBM-slab fallback keeps the cuts of rounds 1-2)
        // mean and variance from the three sums (reciprocal by two Newton steps: ~1 ulp; a sufficient condition's input)
        auto mv = [&](double s0, double s1, double s2, double &m1, double &var) __attribute__((always_inline)) {
            double ri = __builtin_amdgcn_rcp(s0);
            ri = fma(fma(-s0, ri, 1.0), ri, ri);
            ri = fma(fma(-s0, ri, 1.0), ri, ri);
            m1 = s1 * ri;
            var = s2 * ri - m1 * m1;
        };
        // Two-wave form: the FIRST evaluation and the moment pass in one sweep over beta_d (p = -df(x0) is known before f's data
        // term is: df needs no beta_d).  The sums of exp(eta~ - m) beta_d are data_F's, chain for chain, so f(x0) has the bits of
        // the plain evaluation; the two extra sums per word ride on the same register / LDS reads.
        // pair / xu / lu: one more logarithm, lu = log_pos(xu), taken together with the register word's (wave 0: the log-sum-exp's)
        auto words_F3 = [&](double m, const int NdL, double &part, double &d1, double &d2, const bool pair, const double xu, double &lu) __attribute__((always_inline)) {
            part = 0.0; d1 = 0.0; d2 = 0.0;
            if constexpr (NW == 2) {
                const double2 *se2 = reinterpret_cast<const double2 *>(se);
                const double2 *sv2 = reinterpret_cast<const double2 *>(sv);
                const double2 *sw2 = reinterpret_cast<const double2 *>(sw);
                if constexpr (DMA && (STM_FUSE_SLAB & 2)) {
                    if (NdL > 0 && NdL <= WAVE) {   // uniform: register word and slab word side by side (see data_F)
                        const double cw = (wreg < Nd) ? P.counts[p0 + wreg] : 0.0;
                        const int ia = lane < NdL ? lane : NdL - 1;
                        const double2 *ra = reinterpret_cast<const double2 *>(slab + (size_t)ia * KP);
                        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, q0 = 0.0, q1 = 0.0;
                        double A0 = 0.0, A1 = 0.0, B0 = 0.0, B1 = 0.0, Q0 = 0.0, Q1 = 0.0;
#if STM_FUSE_PIPE3
                        // software-pipelined like data_F's side-by-side loop: a group's four reads per pair of topics fly while the group before is multiplied
                        constexpr int GI = STM_FUSE_PIPE3, NI = KR / 2, NG = (NI + GI - 1) / GI;
                        double2 eb[2][GI], ub[2][GI], vb[2][GI], rb[2][GI];
#pragma unroll
                        for (int i = 0; i < GI; ++i) { const int q = i < NI ? i : 0; eb[0][i] = se2[q]; ub[0][i] = sv2[q]; vb[0][i] = sw2[q]; rb[0][i] = ra[q]; }
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            if (g + 1 < NG) {
#pragma unroll
                                for (int i = 0; i < GI; ++i) {
                                    const int q = (g + 1) * GI + i < NI ? (g + 1) * GI + i : NI - 1;
                                    eb[(g + 1) & 1][i] = se2[q]; ub[(g + 1) & 1][i] = sv2[q]; vb[(g + 1) & 1][i] = sw2[q]; rb[(g + 1) & 1][i] = ra[q];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < GI; ++i) {
                                const int q = g * GI + i;
                                if (q < NI) {
                                    const double2 e = eb[g & 1][i], u = ub[g & 1][i], v = vb[g & 1][i], ba = rb[g & 1][i];
                                    a0 = fma(e.x, breg[2 * q], a0); a1 = fma(e.y, breg[2 * q + 1], a1);
                                    b0 = fma(u.x, breg[2 * q], b0); b1 = fma(u.y, breg[2 * q + 1], b1);
                                    q0 = fma(v.x, breg[2 * q], q0); q1 = fma(v.y, breg[2 * q + 1], q1);
                                    A0 = fma(e.x, ba.x, A0); A1 = fma(e.y, ba.y, A1);
                                    B0 = fma(u.x, ba.x, B0); B1 = fma(u.y, ba.y, B1);
                                    Q0 = fma(v.x, ba.x, Q0); Q1 = fma(v.y, ba.y, Q1);
                                }
                            }
                            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(q0), "+v"(q1), "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1), "+v"(Q0), "+v"(Q1));
                        }
#else
#pragma unroll
                        for (int k = 0; k < KR; k += 2) {
                            if (k % STM_FUSE_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                            const double2 e = se2[k / 2], u = sv2[k / 2], v = sw2[k / 2], ba = ra[k / 2];
                            a0 = fma(e.x, breg[k], a0); a1 = fma(e.y, breg[k + 1], a1);
                            b0 = fma(u.x, breg[k], b0); b1 = fma(u.y, breg[k + 1], b1);
                            q0 = fma(v.x, breg[k], q0); q1 = fma(v.y, breg[k + 1], q1);
                            A0 = fma(e.x, ba.x, A0); A1 = fma(e.y, ba.y, A1);
                            B0 = fma(u.x, ba.x, B0); B1 = fma(u.y, ba.y, B1);
                            Q0 = fma(v.x, ba.x, Q0); Q1 = fma(v.y, ba.y, Q1);
                            if (k % STM_FUSE_GROUP == STM_FUSE_GROUP - 2 || k + 2 >= KR)   // all twelve chains advance together (see data_F)
                                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(q0), "+v"(q1), "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1), "+v"(Q0), "+v"(Q1));
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        const double sums[2] = {a0 + a1, A0 + A1};
                        double lg[2], m1, var, M1, VAR;
                        log_pos2(sums, lg);
                        mv(sums[0], b0 + b1, q0 + q1, m1, var);
                        mv(sums[1], B0 + B1, Q0 + Q1, M1, VAR);
                        const bool in = wreg < Nd, ins = lane < NdL;
                        part = in ? cw * (m + lg[0]) : 0.0;
                        d1 = in ? cw * m1 : 0.0;
                        d2 = in ? cw * var : 0.0;
                        part += ins ? crow[ia] * (m + lg[1]) : 0.0;
                        d1 += ins ? crow[ia] * M1 : 0.0;
                        d2 += ins ? crow[ia] * VAR : 0.0;
                        return;
                    }
                }
                {
                    const double cw = (wreg < Nd) ? P.counts[p0 + wreg] : 0.0;   // (= c0; see moments_words)
                    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, q0 = 0.0, q1 = 0.0;
#if STM_REG_PIPE3
                    constexpr int GI = STM_REG_PIPE3, NI = KR / 2, NG = (NI + GI - 1) / GI;   // software-pipelined groups (see data_F)
                    double2 eb[2][GI], ub[2][GI], vb[2][GI];
#pragma unroll
                    for (int i = 0; i < GI; ++i) { const int q = i < NI ? i : 0; eb[0][i] = se2[q]; ub[0][i] = sv2[q]; vb[0][i] = sw2[q]; }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g + 1 < NG) {
#pragma unroll
                            for (int i = 0; i < GI; ++i) {
                                const int q = (g + 1) * GI + i < NI ? (g + 1) * GI + i : NI - 1;
                                eb[(g + 1) & 1][i] = se2[q]; ub[(g + 1) & 1][i] = sv2[q]; vb[(g + 1) & 1][i] = sw2[q];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < GI; ++i) {
                            const int q = g * GI + i;
                            if (q < NI) {
                                const double2 e = eb[g & 1][i], u = ub[g & 1][i], v = vb[g & 1][i];
                                a0 = fma(e.x, breg[2 * q], a0); a1 = fma(e.y, breg[2 * q + 1], a1);
                                b0 = fma(u.x, breg[2 * q], b0); b1 = fma(u.y, breg[2 * q + 1], b1);
                                q0 = fma(v.x, breg[2 * q], q0); q1 = fma(v.y, breg[2 * q + 1], q1);
                            }
                        }
                        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(q0), "+v"(q1));
                    }
#else
#pragma unroll
                    for (int k = 0; k < KR; k += 2) {
                        if (k % STM_MOM_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                        const double2 e = se2[k / 2], u = sv2[k / 2], v = sw2[k / 2];
                        a0 = fma(e.x, breg[k], a0); a1 = fma(e.y, breg[k + 1], a1);
                        b0 = fma(u.x, breg[k], b0); b1 = fma(u.y, breg[k + 1], b1);
                        q0 = fma(v.x, breg[k], q0); q1 = fma(v.y, breg[k + 1], q1);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    const double s0 = a0 + a1;
                    double lg;
                    if (pair) {   // (uniform)
                        const double sums[2] = {s0, xu};
                        double l2[2];
                        log_pos2(sums, l2);
                        lg = m + l2[0]; lu = l2[1];
                    } else lg = m + log_pos(s0);
                    double m1, var;
                    mv(s0, b0 + b1, q0 + q1, m1, var);
                    const bool in = wreg < Nd;
                    part = in ? cw * lg : 0.0;
                    d1 = in ? cw * m1 : 0.0;
                    d2 = in ? cw * var : 0.0;
                }
                const int kp2 = KP >> 1;
                for (int vb = 0; vb < NdL; vb += WAVE) {
                    const int va = vb + lane, ia = va < NdL ? va : NdL - 1;
                    const double2 *ra = reinterpret_cast<const double2 *>(slab + (size_t)ia * KP);
                    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll 5
                    for (int kk = 0; kk < kp2; ++kk) {
                        const double2 e = se2[kk], u = sv2[kk], v = sw2[kk], ba = ra[kk];
                        a0 = fma(e.x, ba.x, a0); a1 = fma(e.y, ba.y, a1);
                        b0 = fma(u.x, ba.x, b0); b1 = fma(u.y, ba.y, b1);
                        q0 = fma(v.x, ba.x, q0); q1 = fma(v.y, ba.y, q1);
                    }
                    const double s0 = a0 + a1, la = m + log_pos(s0);
                    double m1, var;
                    mv(s0, b0 + b1, q0 + q1, m1, var);
                    const bool in = va < NdL;
                    part += in ? crow[ia] * la : 0.0;
                    d1 += in ? crow[ia] * m1 : 0.0;
                    d2 += in ? crow[ia] * var : 0.0;
                }
            }
        };
        auto moments_words = [&](double &d1, double &d2, const int NdL) __attribute__((always_inline)) {
            d1 = 0.0; d2 = 0.0;
            if constexpr (MOM && (NW == 1 || STM_LATER_PASS)) {
                const double2 *se2 = reinterpret_cast<const double2 *>(se);
                const double2 *sv2 = reinterpret_cast<const double2 *>(sv);
                const double2 *sw2 = reinterpret_cast<const double2 *>(sw);
                if constexpr (DMA && (STM_FUSE_SLAB & 4)) {
                    if (NdL > 0 && NdL <= WAVE) {   // uniform: register word and slab word side by side (see data_F)
                        const double cw = (wreg < Nd) ? P.counts[p0 + wreg] : 0.0;
                        const int ia = lane < NdL ? lane : NdL - 1;
                        const double2 *ra = reinterpret_cast<const double2 *>(slab + (size_t)ia * KP);
                        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, q0 = 0.0, q1 = 0.0;
                        double A0 = 0.0, A1 = 0.0, B0 = 0.0, B1 = 0.0, Q0 = 0.0, Q1 = 0.0;
#pragma unroll
                        for (int k = 0; k < KR; k += 2) {
                            if (k % STM_FUSE_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                            const double2 e = se2[k / 2], u = sv2[k / 2], v = sw2[k / 2], ba = ra[k / 2];
                            a0 = fma(e.x, breg[k], a0); a1 = fma(e.y, breg[k + 1], a1);
                            b0 = fma(u.x, breg[k], b0); b1 = fma(u.y, breg[k + 1], b1);
                            q0 = fma(v.x, breg[k], q0); q1 = fma(v.y, breg[k + 1], q1);
                            A0 = fma(e.x, ba.x, A0); A1 = fma(e.y, ba.y, A1);
                            B0 = fma(u.x, ba.x, B0); B1 = fma(u.y, ba.y, B1);
                            Q0 = fma(v.x, ba.x, Q0); Q1 = fma(v.y, ba.y, Q1);
                            if (k % STM_FUSE_GROUP == STM_FUSE_GROUP - 2 || k + 2 >= KR)   // all twelve chains advance together (see data_F)
                                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(q0), "+v"(q1), "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1), "+v"(Q0), "+v"(Q1));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        double m1, var, M1, VAR;
                        mv(a0 + a1, b0 + b1, q0 + q1, m1, var);
                        mv(A0 + A1, B0 + B1, Q0 + Q1, M1, VAR);
                        const bool in = wreg < Nd, ins = lane < NdL;
                        d1 = in ? cw * m1 : 0.0;
                        d2 = in ? cw * var : 0.0;
                        d1 += ins ? crow[ia] * M1 : 0.0;
                        d2 += ins ? crow[ia] * VAR : 0.0;
                        return;
                    }
                }
                if constexpr (KREG > 0) {
                    // the word's count again from memory (L2): a second use of c0 at this site costs 225 spilled registers
                    const double cw = (wreg < Nd) ? P.counts[p0 + wreg] : 0.0;
                    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll
                    for (int k = 0; k < KR; k += 2) {
                        if (k % STM_MOM_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
                        const double2 e = se2[k / 2], u = sv2[k / 2], v = sw2[k / 2];
                        a0 = fma(e.x, breg[k], a0); a1 = fma(e.y, breg[k + 1], a1);
                        b0 = fma(u.x, breg[k], b0); b1 = fma(u.y, breg[k + 1], b1);
                        q0 = fma(v.x, breg[k], q0); q1 = fma(v.y, breg[k + 1], q1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double m1, var;
                    mv(a0 + a1, b0 + b1, q0 + q1, m1, var);
                    const bool in = wreg < Nd;
                    d1 = in ? cw * m1 : 0.0;
                    d2 = in ? cw * var : 0.0;
                }
                if constexpr (!DIRECT) {   // (DIRECT: the fused set-up sweep below delivers the sums)
                    const int kp2 = KP >> 1;
                    for (int vb = 0; vb < NdL; vb += WAVE) {
                        const int va = vb + lane, ia = va < NdL ? va : NdL - 1;
                        const double2 *ra = reinterpret_cast<const double2 *>(slab + (size_t)ia * KP);
                        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll 5
                        for (int kk = 0; kk < kp2; ++kk) {
                            const double2 e = se2[kk], u = sv2[kk], v = sw2[kk], ba = ra[kk];
                            a0 = fma(e.x, ba.x, a0); a1 = fma(e.y, ba.y, a1);
                            b0 = fma(u.x, ba.x, b0); b1 = fma(u.y, ba.y, b1);
                            q0 = fma(v.x, ba.x, q0); q1 = fma(v.y, ba.y, q1);
                        }
                        double m1, var;
                        mv(a0 + a1, b0 + b1, q0 + q1, m1, var);
                        const bool in = va < NdL;
                        d1 += in ? crow[ia] * m1 : 0.0;
                        d2 += in ? crow[ia] * var : 0.0;
                    }
                }
            }
        };
        // the whole of f on one wave (NW = 1)
        auto eval_F = [&]() __attribute__((always_inline)) -> double {
            double m, ssum, e_lane = 0.0;
            int icnt;
            head_F(m, icnt, ssum, e_lane);
            STM_WAVE_SYNC();
            double lse = lse_F(m, icnt, ssum);
            double part = wave_sum(data_F(m, e_lane, NdL));
            double q = quad_F();
            STM_WAVE_SYNC();
            return 0.5 * q - (part - Ndoc * lse);
        };

        // df(eta): stm.py:946-958 (data term g0 has no eta dependence)
        auto eval_DF = [&]() __attribute__((always_inline)) {
            double ex[VPL];
            double sl = 0.0;
#pragma unroll
            for (int r = 0; r < VPL; ++r) {
                ex[r] = (lane + WAVE * r < n) ? exp(xt[r]) : 0.0;
                sl += ex[r];
            }
            const double sumexp = wave_sum(sl) + 1.0;  // + exp(0)
            const double scale = Ndoc / sumexp;
            if (sdiag) {
#pragma unroll
                for (int r = 0; r < VPL; ++r)
                    gv[r] = (lane + WAVE * r < n) ? sd[r] * (xt[r] - mu[r]) - (g0[r] - scale * ex[r]) : 0.0;
            } else {
#pragma unroll
                for (int r = 0; r < VPL; ++r)
                    if (lane + WAVE * r < n) svx[lane + WAVE * r] = xt[r] - mu[r];
                STM_WAVE_SYNC();
#pragma unroll
                for (int r = 0; r < VPL; ++r) {
                    const int i = lane + WAVE * r;
                    double t = 0.0;
                    if (i < n)
                        for (int j = 0; j < n; ++j) t += S[(size_t)i * n + j] * svx[j];
                    gv[r] = (i < n) ? t - (g0[r] - scale * ex[r]) : 0.0;
                }
                STM_WAVE_SYNC();
            }
        };

        // ---- DIRECT (K > 64), fused set-up sweep.  Every pass over the document re-gathers its rows (~80 k cycles at K = 100), and a
        // document whose first search is dead needed three of them: g0, f(x0), the moments.  All three come out of one:
        //   (1) g0 += beta_d[:, tile] (c / colsum)                       lane = topic           (as the set-up sweep it replaces)
        //   (2) S_w = sum_k beta_d[k, w] exp(eta~_k - m), c_w log S_w     lane = (word, quarter) (data_F's sums, chain for chain)
        //   (3) v += beta_d[:, tile] (c / S)                              lane = topic
        // v_k exp(eta~_k - m) = dD/d eta~_k, the data term of the TRUE gradient of f at x0.  The moment test (S_OUTER_TOP) needs
        // D1 = D'(0) = v . p~ -- exact -- and an upper bound of D2 = sum_w c_w Var_{q_w}(p~): a variance is at most the mean square
        // about ANY constant a, so D2 <= sum_w c_w E_{q_w}[(p~ - a)^2] = sum_k v_k (p~_k - a)^2 (a = D1 / N_d).  That is 1.3-2x the
        // exact D2 on traced data and proves 92-97 % of what the exact one proves -- without the second sweep p would have to wait for.
        double part0 = 0.0, vdat[VPL];
        bool have0 = false;
#pragma unroll
        for (int r = 0; r < VPL; ++r) vdat[r] = 0.0;
        if constexpr (DIRECT) {
            if (fuse0) {
#pragma unroll
                for (int r = 0; r < VPL; ++r) xt[r] = x[r];
                double m0, ssum0, e_l = 0.0;
                int icnt0;
                head_F(m0, icnt0, ssum0, e_l);     // exp(eta~ - m) -> se[] (the first evaluation recomputes it: no sweep)
                STM_WAVE_SYNC();
                double g0a[VPL], va[VPL], part = 0.0;
#pragma unroll
                for (int r = 0; r < VPL; ++r) { g0a[r] = 0.0; va[r] = 0.0; }
                const double2 *se2 = reinterpret_cast<const double2 *>(se);
                const int k0 = dq * kq2, k1 = (dq + 1) * kq2 < (KP >> 1) ? (dq + 1) * kq2 : (KP >> 1);
#ifdef STM_SWEEP_PROF   // a build for tools/solver_prof.py: wait + store / fetch issue / (1) / (2) / (3) into profile slots 40-44
                long long tsw[5] = {0, 0, 0, 0, 0};
#define STM_SWEEP_MARK(q, ...) if (STM_PROF(P)) { __VA_ARGS__; const long long cy = __builtin_readcyclecounter(); tsw[q] += cy - cy0; cy0 = cy; }
#else
#define STM_SWEEP_MARK(q, ...)
#endif
                tile_fetch(0);
                for (int t0 = 0; t0 < NdL; t0 += TWS) {
                    const int nw = NdL - t0 < TWS ? NdL - t0 : TWS;
#ifdef STM_SWEEP_PROF
                    long long cy0 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
#endif
                    tile_store(nw);
                    STM_WAVE_SYNC();
                    STM_SWEEP_MARK(0, wait_lds())
                    if (t0 + TWS < NdL) tile_fetch(t0 + TWS);
                    STM_SWEEP_MARK(1, (void)0)
                    tile_axpy(wrow, t0, nw, g0a);    // (1)
                    STM_SWEEP_MARK(2, pin(g0a[0]))
                    const double2 *tr = reinterpret_cast<const double2 *>(slab + (size_t)dw * KP);   // (2)
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll 4
                    for (int kk = k0; kk < k1; ++kk) {
                        const double2 e = se2[kk], b = tr[kk];
                        a0 = fma(e.x, b.x, a0);
                        a1 = fma(e.y, b.y, a1);
                    }
                    double sdot = a0 + a1;
                    sdot += __shfl_xor(sdot, 16);
                    sdot += __shfl_xor(sdot, 32);
                    const double lg = m0 + log_pos(sdot);
                    const bool mine = dq == 0 && dw < nw;
                    const double cw = crow[t0 + (dw < nw ? dw : 0)];
                    part += mine ? cw * lg : 0.0;
                    STM_WAVE_SYNC();                 // (1)'s reads of wrow are done
                    if (mine) wrow[t0 + dw] = cw / sdot;
                    STM_WAVE_SYNC();
                    STM_SWEEP_MARK(3, wait_lds())
                    tile_axpy(wrow, t0, nw, va);     // (3)
                    STM_WAVE_SYNC();
                    STM_SWEEP_MARK(4, pin(va[0]))
                }
#ifdef STM_SWEEP_PROF
                if (STM_PROF(P) && lane == 0) for (int q = 0; q < 5; ++q) STM_PROF(P)[doc * PROF_SLOTS + 40 + q] = tsw[q];
#endif
#pragma unroll
                for (int r = 0; r < VPL; ++r) {
                    const int i = lane + WAVE * r;
                    g0[r] = (i < n) ? g0a[r] : 0.0;
                    vdat[r] = (i < K) ? va[r] * se[i] : 0.0;
                }
                part0 = wave_sum(part);
                have0 = true;
            }
        }
        // the first evaluation of f after that sweep: everything but the data term
        auto eval_F0 = [&]() __attribute__((always_inline)) -> double {
            double m, ssum, e_lane = 0.0;
            int icnt;
            head_F(m, icnt, ssum, e_lane);
            STM_WAVE_SYNC();
            const double lse = lse_F(m, icnt, ssum);
            const double q = quad_F();
            STM_WAVE_SYNC();
            return 0.5 * q - (part0 - Ndoc * lse);
        };

        // rows [i0, i1) of column j of the BFGS update  H <- H - rho (s w^T + w s^T) + cc s s^T  (see S_ACCEPT2).
        // Rows are independent, but the compiler cannot prove that a later row's load does not alias an earlier
        // row's store: fetch four rows, then store four (LDS latency paid once per batch).
        auto bfgs_rows = [&](int j, int i0, int i1, bool ident, double rhok, double cc, double sj, double wj) __attribute__((always_inline)) {
            double *hp = Hs + j;
            int i = i0;
            for (; i + 3 < i1; i += 4) {
                double h[4], si[4], wi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[q] = ident ? (i + q == j ? 1.0 : 0.0) : hp[(size_t)(i + q) * n];
                    si[q] = sv[i + q]; wi[q] = sw[i + q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    hp[(size_t)(i + q) * n] = h[q] - rhok * (si[q] * wj + wi[q] * sj) + cc * (si[q] * sj);
            }
            for (; i < i1; ++i) {
                const double h = ident ? (i == j ? 1.0 : 0.0) : hp[(size_t)i * n];
                const double si = sv[i], wi = sw[i];
                hp[(size_t)i * n] = h - rhok * (si * wj + wi * sj) + cc * (si * sj);
            }
        };
        const int nh = NW == 2 ? ((n / 2 + 3) & ~3) : n;   // two-wave form: wave 0 updates rows [0, nh), wave 1 the rest

        // ---- two-wave form: wave 1 serves evaluation requests until wave 0 posts "exit"
        if (NW == 2) {
            // df needs the data term g0 of ALL words: wave 0 hands its share to wave 1
            if (wv == 0 && lane < n) xch_gv[lane] = g0[0];
            __syncthreads();
            if (wv == 1) {
                if (lane < n) g0[0] += xch_gv[lane];
                bool pf_pending = PERSIST && STM_PREFETCH;
                auto prefetch_next = [&]() __attribute__((always_inline)) {
                  if constexpr (PERSIST && STM_PREFETCH) {
                    // The next document's header is known (its ticket came back during the set-up) and this wave has nothing to do while wave 0
                    // digests the first evaluation: touch the lines that document's set-up reads first -- word ids, counts, eta, mu: cold, each
                    // read once per E-step and otherwise two dependent trips to the HBM at the head of every document -- so that they wait
                    // in the L2.  One instruction, a line per lane: lanes 0..15 ids, 16..47 counts, 48..55 eta, 56..63 mu.
                    const int tkn = uni(tk_slot[tk_par]);   // (parked there during the set-up)
                    if (persist && P.tick && (int64_t)tkn < P.count) {
                        const int64_t tn = P.first + tkn;
                        const int64_t h0 = scalar_load(P.tick + 2 * tn), h1 = scalar_load(P.tick + 2 * tn + 1);
                        const int64_t docn = h1 & 0xffffffffLL;
                        const int Ndn = (int)(h1 >> 32);
                        const int li = lane < 16 ? lane : lane < 48 ? lane - 16 : (lane - 48) & 7;
                        const char *base = lane < 16 ? reinterpret_cast<const char *>(P.indices + h0)
                                         : lane < 48 ? reinterpret_cast<const char *>(P.counts + h0)
                                         : lane < 56 ? reinterpret_cast<const char *>(P.eta + docn * n) : reinterpret_cast<const char *>(P.mu + docn * n);
                        const int len = lane < 16 ? 4 * Ndn : lane < 48 ? 8 * Ndn : 8 * n;
                        // (the first line may start anywhere inside a 128-byte line: one more line covers the tail)
                        const int lead = (int)(reinterpret_cast<uintptr_t>(base) & 127);
                        if (128 * li < len + lead) {
                            const int off = 128 * li - lead;
                            l2_prefetch_line(base + (off > 0 ? off : 0), lds_addr(pf_dump));
                        }
                    }
                  }
                  pf_pending = false;
                };
                for (;;) {
                    __syncthreads();  // (0) request posted: xch_cmd[0] and the trial point
                    const int cmd = uni(xch_cmd[0]);
                    if (cmd & 4) break;
                    if (cmd & 8) {   // its half of a BFGS matrix update (s, w in sv / sw; rho, cc in the mailbox)
                        if (lane < n) bfgs_rows(lane, nh < n ? nh : n, n, (cmd & 16) != 0, uni(xch_res[6]), uni(xch_res[7]), sv[lane], sw[lane]);
                        __syncthreads();  // (u) update complete
                        continue;
                    }
                    if (STM_LATER_PASS && (cmd & 32)) {  // the moment pass of a later search (k >= 1): e, e p~, e p~^2 in se / sv / sw, p in the mailbox
                        const double pl = (lane < n) ? xch_xt[lane] : 0.0;
                        double d1, d2;
                        moments_words(d1, d2, NdL);
                        d1 = wave_sum(d1); d2 = wave_sum(d2);
                        const double g0p = wave_sum(g0[0] * pl);
                        if (lane == 0) { xch_res[3] = g0p; xch_res[5] = d1; xch_res[6] = d2; }
                        __syncthreads();  // (2) results posted
                        continue;
                    }
                    if (cmd & 64) {  // the first evaluation, fused with the moment pass along p = -df(x0) (words_F3)
                        xt[0] = (lane < n) ? xch_xt[lane] : 0.0;
                        const double q = STM_QUAD_W0 ? 0.0 : quad_F();
                        eval_DF();
                        if (lane < n) xch_gv[lane] = gv[0];
                        __syncthreads();  // (1) df posted; exp(eta~ - m) in se[], m in xch_res[2]
                        const double pl = -gv[0];   // (0 beyond n)
                        __syncthreads();  // (1b) exp(eta~ - m) p~ and exp(eta~ - m) p~^2 in sv[] / sw[]
                        double part, d1, d2, lu_unused;
                        words_F3(uni(xch_res[2]), NdL, part, d1, d2, false, 0.0, lu_unused);
                        part = wave_sum(part); d1 = wave_sum(d1); d2 = wave_sum(d2);
                        const double g0p = wave_sum(g0[0] * pl);
                        if (lane == 0) { xch_res[0] = part; xch_res[1] = q; xch_res[3] = g0p; xch_res[5] = d1; xch_res[6] = d2; }
                        __syncthreads();  // (2) results posted
                        if (pf_pending) prefetch_next();
                        continue;
                    }
                    xt[0] = (lane < n) ? xch_xt[lane] : 0.0;
                    // while wave 0 works on max / exp(eta~ - m): the pieces that do not need them
                    double q = 0.0;
                    if ((cmd & 1) && !STM_QUAD_W0) q = quad_F();
                    if (cmd & 2) {
                        eval_DF();
                        if (lane < n) xch_gv[lane] = gv[0];
                    }
                    __syncthreads();  // (1) exp(eta~ - m) in se[], m in xch_res[2]
                    if (cmd & 1) {
                        const double part = wave_sum(data_F(uni(xch_res[2]), se[lane], NdL));  // words 64..127 + the slab
                        if (lane == 0) { xch_res[0] = part; xch_res[1] = q; }
                    }
                    __syncthreads();  // (2) results posted
                    if (pf_pending) prefetch_next();
                }
                if (persist) continue;   // -> next_doc()
                return;
            }
        }
        // wave 0 of the two-wave form: f and/or df at xt, shared with wave 1
        auto eval_split = [&](bool do_f, bool do_g, double &f_out) __attribute__((always_inline)) {
            double m = 0.0, ssum = 0.0, e_lane = 0.0;
            int icnt = 1;
            // -DSTM_EVAL_PROF: where one evaluation's cycles go on wave 0 (profile slots 40..44, accumulated in memory).  Not in
            // the shipped build: the clock reads cost ten spilled SGPRs around every evaluation.
#ifdef STM_EVAL_PROF
            long long tc = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
            auto lap = [&](int slot) __attribute__((always_inline)) {
                if (STM_PROF(P)) {
                    const long long now = (long long)__builtin_readcyclecounter();
                    if (lane == 0) STM_PROF(P)[doc * PROF_SLOTS + slot] += now - tc;
                    tc = now;
                }
            };
#else
            auto lap = [](int) __attribute__((always_inline)) {};
#endif
            if (lane < n) xch_xt[lane] = xt[0];
            if (lane == 0) xch_cmd[0] = (do_f ? 1 : 0) | (do_g ? 2 : 0);
            __syncthreads();   // (0) wave 1 starts on df and the quadratic form ...
            lap(40);
            double lse = 0.0, part = 0.0, stot = 0.0, q0 = 0.0;
            constexpr bool LATE = (STM_LSE_LATE & 1) && KREG > 0;
            if (do_f) {        // ... while this wave takes max / exp(eta~ - m) and the log-sum-exp
                head_F(m, icnt, ssum, e_lane);
                if (lane == 0) xch_res[2] = m;
                if (STM_QUAD_W0 && sdiag && LATE) {   // (xt - mu)^T siginv (xt - mu): quad_F's terms and sum, beside the log-sum-exp's sum
                    const double d = xt[0] - mu[0];
                    double two[2] = {ssum, (lane < n) ? (d * sd[0]) * d : 0.0};
                    wave_sum_n(two);
                    stot = two[0]; q0 = two[1];
                } else {
                    if (STM_QUAD_W0) q0 = quad_F();
                    if (LATE) stot = wave_sum(ssum); else lse = lse_F(m, icnt, ssum);
                }
            }
            lap(41);
            __syncthreads();   // (1)
            lap(42);
            if (do_f) {
                if (LATE && icnt == 1) {
                    // lse_F's log1p_pos(s) = log_pos(u) + (s - (u - 1)) / u, u = 1 + s: its logarithm and the word's, the evaluation's two
                    // long chains on this wave, run side by side (log_pos2: log_pos's operations, its bits) -- and behind the barrier,
                    // where the other wave has its slab words to do, instead of in front of it
                    const double u = 1.0 + stot, cr = stot - (u - 1.0);
                    const double sums[2] = {reg_dot(), u};
                    double lg[2];
                    log_pos2(sums, lg);
                    part = wave_sum((wreg < Nd) ? STM_C0 * (m + lg[0]) : 0.0);
                    lse = ((u == INFINITY) ? lg[1] : lg[1] + cr * __builtin_amdgcn_rcp(u)) + m;
                } else {
                    if (LATE) lse = lse_tail(m, icnt, stot);
                    part = wave_sum(data_F(m, e_lane, 0));   // words 0..63 (wave 1 owns the slab)
                }
            }
            lap(43);
            __syncthreads();   // (2)
            lap(44);
            if (do_f) {
                const double part_all = part + uni(xch_res[0]);
                const double q = STM_QUAD_W0 ? q0 : uni(xch_res[1]);
                f_out = 0.5 * q - (part_all - Ndoc * lse);
            }
            if (do_g) gv[0] = (lane < n) ? xch_gv[lane] : 0.0;
        };

        // wave 0 of the two-wave form: f, df and the moments D1, D2, g0 . p, p^T siginv p at x0 (p = -df(x0): the first direction)
        auto eval_first = [&](double &f_out, double &D1_out, double &D2_out, double &g0p_out, double &qx_out) __attribute__((always_inline)) {
            double m = 0.0, ssum = 0.0, e_lane = 0.0;
            int icnt = 1;
            if (lane < n) xch_xt[lane] = xt[0];
            if (lane == 0) xch_cmd[0] = 1 | 2 | 64;
            __syncthreads();   // (0) wave 1: quadratic form and df
            head_F(m, icnt, ssum, e_lane);
            if (lane == 0) xch_res[2] = m;
            constexpr bool LATE = (STM_LSE_LATE & 2) != 0;
            double lse = 0.0, stot = 0.0;
            if (LATE) stot = wave_sum(ssum); else lse = lse_F(m, icnt, ssum);
            const double qw0 = STM_QUAD_W0 ? quad_F() : 0.0;
            __syncthreads();   // (1)
            gv[0] = (lane < n) ? xch_gv[lane] : 0.0;
            {
                const double pl = -gv[0], ep = e_lane * pl;   // e_lane: exp(eta~ - m) of topic `lane` (0 beyond K), p~ = 0 from n on
                sv[lane] = ep; sw[lane] = ep * pl;
            }
            __syncthreads();   // (1b)
            double part, d1, d2;
            {   // the log-sum-exp's logarithm beside the word's (see eval_split); ONE instance of the three-sum loop either way
                const bool pair = LATE && icnt == 1;
                const double u = 1.0 + stot, cr = stot - (u - 1.0);
                double lu = 0.0;
                if (LATE && !pair) lse = lse_tail(m, icnt, stot);
                words_F3(m, 0, part, d1, d2, pair, u, lu);   // words 0..63 (wave 1 owns the slab)
                if (pair) lse = ((u == INFINITY) ? lu : lu + cr * __builtin_amdgcn_rcp(u)) + m;
            }
            part = wave_sum(part); d1 = wave_sum(d1); d2 = wave_sum(d2);
            // p^T siginv p here: wave 1 is still on the slab words (its private vector svb is free from barrier (1) on)
            qx_out = quad_of([&](int r) __attribute__((always_inline)) -> double { return -gv[r]; }, svb);
            __syncthreads();   // (2)
            const double part_all = part + uni(xch_res[0]);
            const double q = STM_QUAD_W0 ? qw0 : uni(xch_res[1]);
            f_out = 0.5 * q - (part_all - Ndoc * lse);
            D1_out = d1 + uni(xch_res[5]); D2_out = py_max2(0.0, d2 + uni(xch_res[6])); g0p_out = uni(xch_res[3]);
        };

        auto dot = [&](const double (&a)[VPL], const double (&b)[VPL]) __attribute__((always_inline)) -> double {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < VPL; ++r) t += a[r] * b[r];
            return wave_sum(t);
        };
        auto dot_lane = [&](const double (&a)[VPL], const double (&b)[VPL]) __attribute__((always_inline)) -> double {   // dot()'s per-lane term
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < VPL; ++r) t += a[r] * b[r];
            return t;
        };
        auto maxabs_lane = [&](const double (&a)[VPL]) __attribute__((always_inline)) -> double {   // maxabs()'s
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < VPL; ++r) t = nanmax(t, fabs(a[r]));
            return t;
        };
        auto maxabs = [&](const double (&a)[VPL]) __attribute__((always_inline)) -> double {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < VPL; ++r) t = nanmax(t, fabs(a[r]));
            return wave_nanmax(t);
        };
        // out_i = sum_j H[j][i] * vec_j  (H symmetric; row j is contiguous across lanes)
        auto matvecH = [&](const double (&vec)[VPL], double (&out)[VPL]) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < VPL; ++r)
                if (lane + WAVE * r < n) sv[lane + WAVE * r] = vec[r];
            STM_WAVE_SYNC_MEM();
            if constexpr (VPL == 2) {   // H lives in HBM here: both rows of the lane in one loop, sixteen loads in flight
                const int i0 = lane, i1 = lane + WAVE < n ? lane + WAVE : n - 1;
                double t0 = 0.0, t1 = 0.0;
                int j = 0;
                for (; j + 7 < n; j += 8) {
                    double h0[8], h1[8], vv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { h0[q] = Hs[(size_t)(j + q) * n + i0]; h1[q] = Hs[(size_t)(j + q) * n + i1]; vv[q] = sv[j + q]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) { t0 += h0[q] * vv[q]; t1 += h1[q] * vv[q]; }
                }
                for (; j < n; ++j) { const double v = sv[j]; t0 += Hs[(size_t)j * n + i0] * v; t1 += Hs[(size_t)j * n + i1] * v; }
                out[0] = t0;
                out[VPL - 1] = (lane + WAVE < n) ? t1 : 0.0;
            } else
#pragma unroll
            for (int r = 0; r < VPL; ++r) {
                const int i = lane + WAVE * r;
                double t = 0.0;
                if (i < n) {   // the sum stays in order (H g must round like scipy's dot up to reassociation-free parts)
                    int j = 0;
                    for (; j + 7 < n; j += 8) {   // sixteen LDS reads in flight, the additions stay in order
                        double hv[8], vv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { hv[q] = Hs[(size_t)(j + q) * n + i]; vv[q] = sv[j + q]; }
#pragma unroll
                        for (int q = 0; q < 8; ++q) t += hv[q] * vv[q];
                    }
                    for (; j < n; ++j) t += Hs[(size_t)j * n + i] * sv[j];
                }
                out[r] = t;
            }
            STM_WAVE_SYNC();
        };

        // ---- scipy _minimize_bfgs state (optimize/_optimize.py:1328-1502)
        const double gtol = 1e-5, c1 = 1e-4, c2 = 0.9, amax = 1e100, amin = 1e-100, xtol = 1e-14;
        // The curvature test |phi'(s)| <= c2 |phi'(0)| is out of reach on [0, s] while s U <= CURV |phi'(0)| (U >= phi''): then
        // |phi'(s)| >= 0.901 |phi'(0)|, against the 0.9 |phi'(0)| the test needs -- a margin of 0.1 % of |phi'(0)|, where the rounding of
        // phi' = df . p is ~1e-13 of it.  (Rounds 1-3 used 0.05: late EM iterations showed searches whose bracket closes on a
        // minimiser of f at 0.5-0.9 of the reach and never gets below half of it -- ~50 evaluations each until DCSRCH gives up.)
        const double CURV = 0.099;
        const int maxiter = n * 200;
        STM_UD(ss, old_fval); STM_UD(ss, old_old_fval); STM_UD(ss, gnorm);
        int k = 0, status = 0;
        bool H_ident = true;
        // K > 64 (the BFGS matrix in HBM): the rank-two update hands the next direction's H g over -- it passes every new H[i][j] through
        // a register anyway, and S_OUTER_TOP's product would be a third trip over the 78 KB (matvecH's additions, in matvecH's order)
        double Hg[VPL];
        bool have_Hg = false;
#pragma unroll
        for (int r = 0; r < VPL; ++r) Hg[r] = 0.0;
        // line-search shared
        STM_UD(ss, phi0); STM_UD(ss, old_phi0); STM_UD(ss, derphi0); STM_UD(ss, Lb); STM_UD(ss, Lv); STM_UD(ss, prange);
        STM_UD(ss, sig_lmax);
        sig_lmax = P.sig_bound;
        // DCSRCH state (optimize/_dcsrch.py)
        STM_UD(ss, stx); STM_UD(ss, fx); STM_UD(ss, gx); STM_UD(ss, sty); STM_UD(ss, fy); STM_UD(ss, gy); STM_UD(ss, stmin); STM_UD(ss, stmax); STM_UD(ss, width); STM_UD(ss, width1); STM_UD(ss, finit); STM_UD(ss, ginit); STM_UD(ss, gtest);
        int stage = 1, w1_calls = 0;
        STM_UD(ss, w1_a1); STM_UD(ss, w1_f1);          // DCSRCH's first trial step and its function value (wolfe2 starts at the same step)
        bool w1_have = false;
        bool brackt = false;
        // wolfe2 / zoom state (optimize/_linesearch.py)
        STM_UD(ss, alpha0); STM_UD(ss, alpha1); STM_UD(ss, phi_a0); STM_UD(ss, phi_a1); STM_UD(ss, derphi_a0);
        int w2_i = 0;
        STM_UD(ss, a_lo); STM_UD(ss, a_hi); STM_UD(ss, phi_lo); STM_UD(ss, phi_hi); STM_UD(ss, derphi_lo); STM_UD(ss, phi_rec); STM_UD(ss, a_rec); STM_UD(ss, a_j);
        int zi = 0;
        STM_UD(ss, acc_alpha); STM_UD(ss, acc_f);
        bool acc_have_g = false;
        // evaluation request / result + scipy ScalarFunction's last-x cache
        STM_UD(ss, alpha); STM_UD(ss, fval); STM_UD(ss, dval); STM_UD(ss, cache_f);
        bool want_eval = true, need_f = true, need_g = true;
        STM_UD(ss, mvar0); STM_UD(ss, mD1); STM_UD(ss, mD2); STM_UD(ss, mg0p); STM_UD(ss, mqx);   // moment pass (S_MOMENTS)
        bool want_mom = false;
        bool have_x = false, f_ok = false, g_ok = false;
        int st = S_INIT_DONE;
        // Second outcome-preserving test for a failing search, on the objective itself.  f along the ray has
        //   f''(s) = p^T siginv p + N_d Var_{theta_s}(p~) - sum_w c_w Var_{q_w,s}(p~) <= U   on [0, b]
        // (U as in S_OUTER_TOP; the data term only lowers it).  With f(0) and a rejected f(b) known, the chord
        // bound f(s) >= f(0) + s (f(b) - f(0)) / b - U s (b - s) / 2 shows that the sufficient-decrease test
        // f(s) <= f(0) + c1 s phi'(0), which DCSRCH and _zoom both require of an accepted step, fails on all of
        // (0, b) when (f(b) - f(0)) / b + c1 |phi'(0)| > U b / 2.  Below s_lo = CURV |phi'(0)| / U the curvature test
        // is out of reach (first cut); above it the violation s * margin must dwarf the rounding of f.
        auto armijo_dead = [&](double b, double phi_b) __attribute__((always_inline)) -> bool {
            const double tr = b * prange;
            const double U = (tr <= 1.0) ? py_min2((double)Lb, Lv * (1.0 + tr + tr * tr)) : (double)Lb;
            const double slope0 = -derphi0;
            // margin = (phi_b - phi0) / b + c1 slope0 - U b / 2 and s_lo = CURV slope0 / U, both multiplied through by b > 0 and U > 0:
            // two IEEE divisions less on the state machine's dependency chain (the inequalities are sufficient conditions with a
            // 1e-9 margin, so their last-bit rounding is immaterial)
            const double margin_b = (phi_b - phi0) + b * (c1 * slope0 - 0.5 * U * b);
            return slope0 > 0.0 && b > 0.0 && U > 0.0 && margin_b > 0.0 &&
                   CURV * slope0 * margin_b >= 1e-9 * py_max2(1.0, fabs((double)phi0)) * (U * b);
        };

        if (P.debug_flags & 1) st = S_FINISH;
        const bool cuts = !(P.debug_flags & 2), reuse = !(P.debug_flags & 4), mproof = !(P.debug_flags & 16);
        long guard = 0;
        if (STM_PROF(P)) t_init = (long long)__builtin_readcyclecounter() - t_begin;
        while (st != S_FINISH) {
            if (++guard > 400000L) { status = 1000 + st; break; }
            const long long tq0 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
            if (MOM && NW == 1 && want_mom) {   // one-wave forms: the moment pass requested by S_OUTER_TOP (operands in se / sv / sw)
                double d1, d2;
                moments_words(d1, d2, NdL);
                mD1 = wave_sum(d1); mD2 = py_max2(0.0, wave_sum(d2)); mg0p = dot(g0, p);   // (mqx: S_OUTER_TOP)
                want_mom = false;
            }
            if (STM_LATER_PASS && MOM && NW == 2 && want_mom) {   // two-wave form, a search after the first one: both waves sweep their words (no logarithm, no f)
                if (lane < n) xch_xt[lane] = p[0];
                if (lane == 0) xch_cmd[0] = 32;
                __syncthreads();   // (0)
                double d1, d2;
                moments_words(d1, d2, 0);   // words 0..63 (wave 1 owns the slab and the complete g0)
                d1 = wave_sum(d1); d2 = wave_sum(d2);
                __syncthreads();   // (2)
                mD1 = d1 + uni(xch_res[5]); mD2 = py_max2(0.0, d2 + uni(xch_res[6])); mg0p = uni(xch_res[3]);
                want_mom = false;
            }
            if (want_eval) {
                // xk + alpha * pk (separate multiply and add, like numpy)
                double xn[VPL];
                bool same = have_x;
#pragma unroll
                for (int r = 0; r < VPL; ++r) {
                    xn[r] = x[r] + alpha * p[r];
                    same = same && (xn[r] == xt[r]);
                }
                same = wave_all(same);
                if (!same) {
#pragma unroll
                    for (int r = 0; r < VPL; ++r) xt[r] = xn[r];
                    have_x = true; f_ok = false; g_ok = false;
                }
                if (NW == 2 && st == S_INIT_DONE && cuts && mproof) {   // the first evaluation carries the moment pass
                    double fnew = 0.0, D1 = 0.0, D2 = 0.0, g0p = 0.0, qx = 0.0;
                    eval_first(fnew, D1, D2, g0p, qx);
                    mD1 = D1; mD2 = D2; mg0p = g0p; mqx = qx;
                    cache_f = fnew; f_ok = true; ++nfev; g_ok = true; ++njev;
                    fval = cache_f; dval = 0.0;
                } else if (NW == 2) {
                    const bool do_f = need_f && !f_ok, do_g = need_g && !g_ok;
                    if (do_f || do_g) {
                        double fnew = 0.0;
                        eval_split(do_f, do_g, fnew);
                        if (do_f) { cache_f = fnew; f_ok = true; ++nfev; }
                        if (do_g) { g_ok = true; ++njev; }
                    }
                    if (need_f) fval = cache_f;
                    if (need_g) dval = dot(gv, p);
                } else {
                    if (need_f) {
                        if (!f_ok) { cache_f = (DIRECT && have0 && st == S_INIT_DONE) ? eval_F0() : eval_F(); f_ok = true; ++nfev; }
                        fval = cache_f;
                    }
                    if (need_g) {
                        if (!g_ok) { eval_DF(); g_ok = true; ++njev; }
                        dval = dot(gv, p);
                    }
                }
                want_eval = false;
            }
            const long long tq1 = STM_PROF(P) ? (long long)__builtin_readcyclecounter() : 0;
            const bool was_upd = st == S_ACCEPT2;
            const int st_in = st;
            switch (st) {
            case S_INIT_DONE: {
                old_fval = fval;
#pragma unroll
                for (int r = 0; r < VPL; ++r) g[r] = gv[r];
                if (STM_SM_BATCH) {
                    double rs[1] = {dot_lane(g, g)}, rm[1] = {maxabs_lane(g)};
                    wave_reduce_n(rs, rm);
                    old_old_fval = old_fval + sqrt(rs[0]) / 2;
                    gnorm = rm[0];
                } else {
                    old_old_fval = old_fval + sqrt(dot(g, g)) / 2;
                    gnorm = maxabs(g);
                }
                st = S_OUTER_TOP;
            } [[fallthrough]];   // states that follow each other without an evaluation share one trip through the loop
            case S_OUTER_TOP: {
                if (!(gnorm > gtol && k < maxiter)) { st = S_FINISH; break; }
                if (H_ident) {
#pragma unroll
                    for (int r = 0; r < VPL; ++r) p[r] = -g[r];
                } else if (NW == 1 && VPL == 2 && have_Hg) {
#pragma unroll
                    for (int r = 0; r < VPL; ++r) p[r] = -Hg[r];
                    have_Hg = false;
                } else {
                    double t[VPL];
                    matvecH(g, t);
#pragma unroll
                    for (int r = 0; r < VPL; ++r) p[r] = -t[r];
                }
                if (!STM_SM_BATCH) derphi0 = dot(g, p);
                {   // Lipschitz bounds of phi'(s) = df(x + s p).p along this direction (see S_W1_ITER):
                    //   phi''(s) = p^T [siginv + N_d (diag(theta_s) - theta_s theta_s^T)] p = p^T siginv p + N_d Var_{theta_s}([p, 0])
                    // (a) Var <= (max [p, 0] - min [p, 0])^2 / 4 for every theta                       -> Lb
                    // (b) theta_s(i) <= theta_0(i) exp(s range) (|d log theta_s(i) / ds| <= range), hence
                    //     Var_{theta_s} <= E_{theta_s}[(p~ - c)^2] <= exp(s range) Var_{theta_0},  c = E_{theta_0} p~   -> Lv (1 + t + t^2), t = s range <= 1
                    double hi = 0.0, lo = 0.0, mx = 0.0;
#pragma unroll
                    for (int r = 0; r < VPL; ++r) {
                        hi = nanmax(hi, p[r]); lo = nanmax(lo, -p[r]);
                        if (lane + WAVE * r < n) mx = nanmax(mx, x[r]);
                    }
                    double range, pp, m;
                    if (STM_SM_BATCH) {   // g . p, p . p and the three maxima in one batch of reductions
                        double rs[2] = {dot_lane(g, p), dot_lane(p, p)}, rm[3] = {hi, lo, mx};
                        wave_reduce_n(rs, rm);
                        derphi0 = rs[0]; pp = rs[1]; range = rm[0] + rm[1]; m = rm[2];
                    } else {
                        range = wave_nanmax(hi) + wave_nanmax(lo);
                        pp = dot(p, p);
                        m = wave_nanmax(mx);   // max of [x, 0]
                    }
                    Lb = sig_lmax * pp + Ndoc * (0.25 * (range * range));
                    prange = range;
                    double e[VPL], z = 0.0, e1 = 0.0;
#pragma unroll
                    for (int r = 0; r < VPL; ++r) {
                        e[r] = (lane + WAVE * r < n) ? exp(x[r] - m) : 0.0;
                        z += e[r]; e1 += e[r] * p[r];
                    }
                    const double eK = exp(-m);          // the appended topic: p~ = 0
                    double psp = 0.0;
#pragma unroll
                    for (int r = 0; r < VPL; ++r)
                        if (sdiag && lane + WAVE * r < n) psp += (p[r] * sd[r]) * p[r];
                    double Z, c, quad;
                    if (STM_SM_BATCH) {
                        double rs[3] = {z, e1, psp};
                        wave_sum_n(rs);
                        Z = rs[0] + eK; c = rs[1] / Z; quad = sdiag ? rs[2] : sig_lmax * pp;
                    } else {
                        Z = wave_sum(z) + eK; c = wave_sum(e1) / Z;
                        quad = sdiag ? wave_sum(psp) : sig_lmax * pp;
                    }
                    double v2 = 0.0;
#pragma unroll
                    for (int r = 0; r < VPL; ++r) {
                        const double dv = p[r] - c;
                        v2 += e[r] * (dv * dv);
                    }
                    const double var0 = (wave_sum(v2) + eK * (c * c)) / Z;
                    Lv = (quad + Ndoc * var0) * (1.0 + 1e-9);
                    // Third outcome-preserving test, before the first search spends an evaluation (k = 0: p = -df(x0), x0 the warm
                    // start left by the previous EM iteration).  The reference's df is not the gradient of f, and from EM iteration 4 on
                    // more than half of the documents start where f increases (or dips by 1e-4 and then increases) along p: every trial
                    // step is rejected, DCSRCH and zoom bisect towards 0 until they give up -> status 2, x unchanged.  Along the ray
                    //   f(s) = Q(s) + N_d lse(eta~ + s p~) - D(s),  D(s) = sum_w c_w log sum_k beta_d[k, w] exp(eta~_k + s p~_k),
                    // Q is exactly quadratic, and the tilted distributions obey theta_s(i) >= e^{-s r} theta_0(i) and
                    // q_{w,s}(i) <= e^{s r} q_{w,0}(i), r = range of p~ = [p, 0]; a variance is the minimum over a of E[(p~ - a)^2], hence
                    //   Var_{theta_s}(p~) >= e^{-s r} var0,   D''(s) = sum_w c_w Var_{q_{w,s}}(p~) <= e^{s r} D2,   D2 = D''(0),
                    //   f'(s) - c1 phi'(0) >= a0 + s p^T siginv p + N_d var0 (1 - e^{-s r}) / r - D2 (e^{s r} - 1) / r =: h(s),
                    //   a0 = f'(0) + c1 |phi'(0)| with the TRUE slope f'(0) = phi'(0) + g0 . p - D1, D1 = D'(0) (df's data term is the constant g0),
                    // so f(s) - [f(0) + c1 s phi'(0)] >= H(s) = int_0^s h.  h is concave (h'' <= 0).
                    // (1) For s < s_x = CURV |phi'(0)| / U' (U' >= phi'' on [0, s_x], Lb / Lv as above) the curvature test
                    //     |phi'(s)| <= 0.9 |phi'(0)| is out of reach: |phi'(s)| >= 0.901 |phi'(0)| (the dot product's rounding is 1e-13 of it).
                    // (2) If h(s_x) > 0, H rises and then at most falls on [s_x, b], so H >= min(H(s_x), H(b)) there; when that exceeds the
                    //     rounding of f (1e-9 max(1, |f|), as in armijo_dead) the sufficient-decrease test fails on all of [s_x, b].
                    // b = the first trial step of DCSRCH and of wolfe2 (the same number): it is rejected by (2), which brackets [0, b], and
                    // every later step of either search lies in (0, b) (S_W1_ITER's comment).  DCSRCH and _zoom accept a step only when
                    // both tests hold -- nowhere on (0, b].  Every quantity is bounded in the safe direction (polynomial bounds of
                    // the exponentials where their differences would cancel) and carries a 1e-9 relative allowance; NaNs fail the comparisons.
                    // D1 and D2 cost one pass over beta_d with three sums per word instead of one (moments_words) and no logarithm.
                    // Searches after the first one (mom_k0 <= k <= mom_k1): the same test with a pass of its own -- from EM iteration 5 on the
                    // documents that move at all take two steps and then a third search that cannot succeed (the fixed point of the
                    // reference's df is not a minimiser of f), which scipy -- and rounds 1-4 here -- only find out evaluation by evaluation.
                    // The argument above does not depend on k: x is the current iterate, p = -H g, b the search's first trial step.
                    const bool later = STM_LATER_PASS && MOM && !DIRECT && k >= P.mom_k0 && k <= P.mom_k1 && k > 0;
                    if (MOM && cuts && mproof && (k == 0 || later) && derphi0 < 0.0 && range > 0.0) {
                        mvar0 = var0;
                        want_mom = true;
                        if (DIRECT && have0) {   // the sums came with the set-up sweep
                            double t1 = 0.0;
#pragma unroll
                            for (int r = 0; r < VPL; ++r) t1 += vdat[r] * ((lane + WAVE * r < n) ? p[r] : 0.0);
                            const double D1 = wave_sum(t1), am = D1 / Ndoc;
                            double t2 = 0.0;
#pragma unroll
                            for (int r = 0; r < VPL; ++r) {
                                const double dv = ((lane + WAVE * r < n) ? p[r] : 0.0) - am;
                                t2 += vdat[r] * (dv * dv);
                            }
                            mD1 = D1; mD2 = py_max2(0.0, wave_sum(t2)) * (1.0 + 1e-9); mg0p = dot(g0, p);
                            mqx = quad_of([&](int r) __attribute__((always_inline)) -> double { return p[r]; }, sv);
                        } else if (NW == 1 || k > 0) {
                            // one-wave forms, and the later searches of the two-wave form: the pass runs at the loop's evaluation site (like
                            // an evaluation, nothing of this block is live across it); its operands go through the LDS.  (Two-wave form,
                            // k = 0: done with the first evaluation.)
                            mqx = quad_of([&](int r) __attribute__((always_inline)) -> double { return p[r]; }, sv);   // (before sv is an operand)
#pragma unroll
                            for (int r = 0; r < VPL; ++r) {
                                const int i = lane + WAVE * r;
                                const double el = (i < n) ? e[r] : (i == n ? eK : 0.0), pl = (i < n) ? p[r] : 0.0;
                                se[i] = el; sv[i] = el * pl; sw[i] = (el * pl) * pl;
                            }
                            STM_WAVE_SYNC();
                        }
                    }
                }
                phi0 = old_fval;
                old_phi0 = old_old_fval;
                st = S_W1_START;
                if (MOM && want_mom) {
                    st = S_MOMENTS;
                    if ((NW == 1 && !(DIRECT && have0)) || (STM_LATER_PASS && NW == 2 && k > 0)) break;      // through the loop top for the pass
                    want_mom = false;        // two-wave form, first search: straight to the verdict
                }
            } [[fallthrough]];
            case S_MOMENTS:
            case S_W1_START: {  // scalar_search_wolfe1 + DCSRCH START
                if (MOM && st == S_MOMENTS) {   // the verdict of the moment pass (see S_OUTER_TOP)
                    double b = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
                    if (b < 0) b = 1.0;
                    // D2 must be an UPPER bound of D''(0).  The word variances of the LDS / register forms are s2 / s0 - m1^2 on the raw
                    // p~: where p~ is nearly constant over the topics a word loads on, the cancellation leaves an absolute error of
                    // ~1e-16 N_d max|p~|^2 of either sign, which the relative allowances below do not cover -- max|p~| <= range (p~
                    // contains a 0), so 64 ulp of N_d range^2 does.  (DIRECT bounds D2 by a mean square about a constant: no cancellation.)
                    const double range = prange, D2 = mD2 + (DIRECT ? 0.0 : 1.5e-14 * (double)Ndoc * (double)prange * (double)prange), qx = mqx, g0p = mg0p, D1 = mD1;
                    const double slope0 = -derphi0, nv = Ndoc * mvar0;
                    const double a0 = ((derphi0 + g0p) - D1) + c1 * slope0, a0tol = 1e-9 * (slope0 + fabs(g0p) + fabs(D1));
                    bool dead = false;
                    // h <= a0 + s (p^T siginv p + N_d var0): two out of five documents leave here (f falls along p)
                    if (finite_d(b) && b > 0.0 && qx >= 0.0 && a0 + b * (qx + nv) > 0.0) {
                        const double s0 = CURV * slope0 / Lv, t0 = s0 * range;
                        const double Ux = (t0 <= 1.0) ? py_min2((double)Lb, Lv * (1.0 + t0 + t0 * t0)) : (double)Lb;
                        const double sx = py_min2(CURV * slope0 / Ux, b);
                        const double ir = 1.0 / range, fm = 1e-9 * py_max2(1.0, fabs((double)phi0));
                        // lower bounds of h(s) and H(s) (reciprocals instead of quotients: inside the 1e-9 allowance)
                        auto hH = [&](double sq, double &h_out, double &H_out) __attribute__((always_inline)) {
                            const double t = sq * range, et = exp(t), eti = 1.0 / et;
                            const bool small = t < 0.05;
                            const double A = small ? t - 0.5 * t * t : 1.0 - eti;                                   // <= 1 - e^-t
                            const double B = small ? t + 0.5 * t * t * et : et - 1.0;                             // >= e^t - 1
                            const double C = small ? t * t * (0.5 - t * (1.0 / 6.0)) : (t - 1.0) + eti;           // <= t - 1 + e^-t
                            const double E = small ? t * t * (0.5 + t * (1.0 / 6.0) * et) : (et - 1.0) - t;       // >= e^t - 1 - t
                            const double up1 = nv * (A * ir), dn1 = D2 * (B * ir);
                            const double up2 = nv * ((C * ir) * ir), dn2 = D2 * ((E * ir) * ir);
                            h_out = (a0 + qx * sq + up1 - dn1) - (a0tol + 1e-9 * (qx * sq + up1 + dn1));
                            H_out = (a0 * sq + 0.5 * qx * sq * sq + up2 - dn2) - (a0tol * sq + 1e-9 * (0.5 * qx * sq * sq + up2 + dn2));
                        };
                        double hx, Hx, hb, Hb;
                        hH(b, hb, Hb);
                        if (Ux > 0.0 && sx > 0.0 && Hb >= fm) {   // (b itself has to be rejected: that is what brackets)
                            if (sx >= b) dead = true;
                            else {
                                hH(sx, hx, Hx);
                                dead = hx > 0.0 && Hx >= fm;
                            }
                        }
                    }
                    if (dead) { status = 2; st = S_FINISH; break; }
                    st = S_W1_START;
                }
                double a1;
                w1_have = false;
                if (derphi0 != 0) {
                    a1 = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
                    if (a1 < 0) a1 = 1.0;
                } else a1 = 1.0;
                if (a1 < amin || a1 > amax || derphi0 >= 0) { st = S_W2_START; break; }  // ERROR
                brackt = false; stage = 1; finit = phi0; ginit = derphi0; gtest = c1 * ginit;
                width = amax - amin; width1 = width / 0.5;
                stx = 0.0; fx = finit; gx = ginit; sty = 0.0; fy = finit; gy = ginit;
                stmin = 0; stmax = a1 + 4.0 * a1;
                w1_calls = 1;
                if (!finite_d(a1)) { st = S_W2_START; break; }
                alpha = a1; need_f = true; need_g = true; want_eval = true;
                st = S_W1_ITER;
            } break;
            case S_W1_ITER: {  // DCSRCH._iterate with (stp, f, g) = (alpha, fval, dval)
#ifdef STM_SM_PROF   // a build for tools/solver_prof.py: where one DCSRCH step's cycles go (profile slots 40..43: tests / dcstep / interval + clip / cuts)
                long long smc = (long long)__builtin_readcyclecounter();
#define STM_SM_LAP(q) if (STM_PROF(P)) { const long long now_ = (long long)__builtin_readcyclecounter(); if (lane == 0) STM_PROF(P)[doc * PROF_SLOTS + 40 + (q)] += now_ - smc; smc = now_; }
#else
#define STM_SM_LAP(q)
#endif
                ++w1_calls;
                double stp = alpha;
                const double f = fval, gd = dval;
                if (w1_calls == 2) { w1_a1 = stp; w1_f1 = f; w1_have = true; }
                const double ftest = finit + stp * gtest;
                if (stage == 1 && f <= ftest && gd >= 0) stage = 2;
                int task = 0;  // 0 FG, 1 CONVERGENCE, 2 WARNING
                if (brackt && (stp <= stmin || stp >= stmax)) task = 2;
                if (brackt && stmax - stmin <= xtol * stmax) task = 2;
                if (stp == amax && f <= ftest && gd <= gtest) task = 2;
                if (stp == amin && (f > ftest || gd >= gtest)) task = 2;
                if (f <= ftest && fabs(gd) <= c2 * -ginit) task = 1;
                if (task == 1) {
                    acc_alpha = stp; acc_f = f; acc_have_g = true;
                    st = S_ACCEPT;
                    break;
                }
                if (task == 2) { st = S_W2_START; break; }
                STM_SM_LAP(0)
                {
                    // modified function (psi) in stage 1, plain one otherwise: selects, not branches
                    const bool mod = (stage == 1 && f <= fx && f > ftest);
                    const double gt = mod ? (double)gtest : 0.0;
                    DcStep in;
                    in.stx = stx; in.sty = sty; in.stp = stp; in.brackt = brackt;
                    in.fx = mod ? fx - stx * gt : (double)fx;
                    in.fy = mod ? fy - sty * gt : (double)fy;
                    in.dx = mod ? gx - gt : (double)gx;
                    in.dy = mod ? gy - gt : (double)gy;
                    const double fm = mod ? f - stp * gt : f, gm = mod ? gd - gt : gd;
                    const DcStep out = dcstep(in, fm, gm, stmin, stmax);
                    stx = out.stx; sty = out.sty; stp = out.stp; brackt = out.brackt;
                    fx = mod ? out.fx + out.stx * gt : out.fx;
                    fy = mod ? out.fy + out.sty * gt : out.fy;
                    gx = mod ? out.dx + gt : out.dx;
                    gy = mod ? out.dy + gt : out.dy;
                }
                STM_SM_LAP(1)
                if (brackt) {
                    if (fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
                    width1 = width;
                    width = fabs(sty - stx);
                }
                if (brackt) {
                    stmin = py_min2(stx, sty);
                    stmax = py_max2(stx, sty);
                } else {
                    stmin = stp + 1.1 * (stp - stx);
                    stmax = stp + 4.0 * (stp - stx);
                }
                stp = np_clip(stp, amin, amax);
                if ((brackt && (stp <= stmin || stp >= stmax)) ||
                    (brackt && stmax - stmin <= xtol * stmax))
                    stp = stx;
                if (!finite_d(stp) || w1_calls >= 100) { st = S_W2_START; break; }
                STM_SM_LAP(2)
                // Outcome-preserving shortcut for the tail of a failing search.  Once the minimiser is
                // bracketed every later trial step lies in [0, smax], smax = max(stx, sty) (the interval
                // only shrinks; an out-of-range step is replaced by stx).  phi'(s) = df(x + s p).p has
                // 0 <= phi'(s) - phi'(0) <= s Lb: the Jacobian of df is siginv + N_d (diag(theta) - theta theta^T),
                // whose quadratic form in p is p^T siginv p plus N_d times the variance of [p, 0] under theta,
                // at most a quarter of its squared range whatever theta is (Lb is set in S_OUTER_TOP).
                // DCSRCH reports convergence only if |phi'(s)| <= 0.9 |phi'(0)|, impossible while
                // smax Lb < 0.1 |phi'(0)| (tested as smax Ls <= CURV |phi'(0)|, see CURV).  What
                // remains is ~60 evaluations inside rounding noise that can only end in a WARNING or
                // the 100-call cap, i.e. alpha = None and the hand-over to wolfe2, whose start does
                // not depend on DCSRCH's final state.
                if (brackt && cuts) {
                    const double smax = py_max2(stx, sty), tr = smax * prange;
                    const double Ls = (tr <= 1.0) ? py_min2((double)Lb, Lv * (1.0 + tr + tr * tr)) : (double)Lb;
                    if (smax * Ls <= CURV * -derphi0) { st = S_W2_START; break; }
                    if (stx == 0.0 && sty > 0.0 && armijo_dead(sty, fy)) {
                        // Rejected at DCSRCH's FIRST step: wolfe2 starts at the same step (the same expression of phi0, old_phi0,
                        // derphi0), reuses f there (S_W2_START), finds it above the sufficient-decrease line (armijo_dead implies
                        // that) and calls _zoom(0, alpha1), where S_ZOOM_TOP applies this very test to the same numbers -> status 2.
                        if (reuse && w1_have && w1_calls == 2) { ++nfev; status = 2; st = S_FINISH; break; }
                        st = S_W2_START; break;
                    }
                }
                alpha = stp; need_f = true; need_g = true; want_eval = true;
                st = S_W1_ITER;
                STM_SM_LAP(3)
            } break;
            case S_W2_GOT_G: {
                const double derphi_a1 = dval;
                if (fabs(derphi_a1) <= -c2 * derphi0) {
                    acc_alpha = alpha1; acc_f = phi_a1; acc_have_g = true;
                    st = S_ACCEPT;
                    break;
                }
                if (derphi_a1 >= 0) {
                    a_lo = alpha1; a_hi = alpha0; phi_lo = phi_a1; phi_hi = phi_a0; derphi_lo = derphi_a1;
                    zi = 0; phi_rec = phi0; a_rec = 0;
                    st = S_ZOOM_TOP;
                    break;
                }
                const double alpha2 = py_min2(2 * alpha1, amax);
                alpha0 = alpha1; alpha1 = alpha2; phi_a0 = phi_a1; derphi_a0 = derphi_a1;
                alpha = alpha1; need_f = true; need_g = false; want_eval = true;
                st = S_W2_GOT_F;
            } break;
            case S_W2_GOT_F: {
                phi_a1 = fval;
                ++w2_i;
                st = S_W2_TOP;
            } break;
            case S_W2_START: {  // scalar_search_wolfe2 (optimize/_linesearch.py:341-474)
                alpha0 = 0;
                if (derphi0 != 0) alpha1 = py_min2(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
                else alpha1 = 1.0;
                if (alpha1 < 0) alpha1 = 1.0;
                alpha1 = py_min2(alpha1, amax);
                st = S_W2_FIRST;
                if (!(reuse && w1_have && alpha1 == w1_a1)) {
                    alpha = alpha1; need_f = true; need_g = false; want_eval = true;
                    break;
                }
                // phi(alpha1) was DCSRCH's first evaluation (same x_k, p_k and step): scipy evaluates it
                // again and gets the same number
                fval = w1_f1; ++nfev;
            } [[fallthrough]];
            case S_W2_FIRST: {
                phi_a1 = fval; phi_a0 = phi0; derphi_a0 = derphi0; w2_i = 0;
                st = S_W2_TOP;
            } [[fallthrough]];
            case S_W2_TOP: {
                if (w2_i >= 10) {  // bracketing loop exhausted: alpha returned, gradient None
                    acc_alpha = alpha1; acc_f = phi_a1; acc_have_g = false;
                    st = S_ACCEPT;
                    break;
                }
                if (alpha1 == 0 || alpha0 > amax) { status = 2; st = S_FINISH; break; }
                if (!(phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && w2_i > 0))) {
                    alpha = alpha1; need_f = false; need_g = true; want_eval = true;
                    st = S_W2_GOT_G;
                    break;
                }
                a_lo = alpha0; a_hi = alpha1; phi_lo = phi_a0; phi_hi = phi_a1; derphi_lo = derphi_a0;
                zi = 0; phi_rec = phi0; a_rec = 0;
                st = S_ZOOM_TOP;
            } [[fallthrough]];
            case S_ZOOM_TOP: {  // _zoom (optimize/_linesearch.py:532-621)
                // same bound as in S_W1_ITER: every later a_j lies between a_lo and a_hi, and zoom accepts
                // only when |phi'(a_j)| <= 0.9 |phi'(0)|; if that is out of reach the remaining iterations
                // can only exhaust maxiter = 10 -> _LineSearchError -> status 2 with x unchanged
                if (cuts && a_lo >= 0 && a_hi >= 0) {
                    const double smax = py_max2(a_lo, a_hi), tr = smax * prange;
                    const double Ls = (tr <= 1.0) ? py_min2((double)Lb, Lv * (1.0 + tr + tr * tr)) : (double)Lb;
                    if (smax * Ls <= CURV * -derphi0) { status = 2; st = S_FINISH; break; }
                    if (a_lo == 0.0 && a_hi > 0.0 && armijo_dead(a_hi, phi_hi)) { status = 2; st = S_FINISH; break; }
                }
                const double dalpha = a_hi - a_lo;
                double a, b;
                if (dalpha < 0) { a = a_hi; b = a_lo; } else { a = a_lo; b = a_hi; }
                double cchk = 0, aj = a_j;
                bool have = false;
                if (zi > 0) {
                    cchk = 0.2 * dalpha;
                    have = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, aj);
                }
                if (zi == 0 || !have || aj > b - cchk || aj < a + cchk) {
                    const double qchk = 0.1 * dalpha;
                    have = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, aj);
                    if (!have || aj > b - qchk || aj < a + qchk) aj = a_lo + 0.5 * dalpha;
                }
                a_j = aj;
                alpha = a_j; need_f = true; need_g = false; want_eval = true;
                st = S_ZOOM_GOT_F;
            } break;
            case S_ZOOM_GOT_F: {
                const double phi_aj = fval;
                if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
                    phi_rec = phi_hi; a_rec = a_hi; a_hi = a_j; phi_hi = phi_aj;
                    ++zi;
                    if (zi > 10) { status = 2; st = S_FINISH; break; }
                    st = S_ZOOM_TOP;
                    break;
                }
                alpha = a_j; need_f = false; need_g = true; want_eval = true;
                st = S_ZOOM_GOT_G;
            } break;
            case S_ZOOM_GOT_G: {
                const double derphi_aj = dval;
                const double phi_aj = fval;  // value at a_j from S_ZOOM_GOT_F (unchanged)
                if (fabs(derphi_aj) <= -c2 * derphi0) {
                    acc_alpha = a_j; acc_f = phi_aj; acc_have_g = true;
                    st = S_ACCEPT;
                    break;
                }
                if (derphi_aj * (a_hi - a_lo) >= 0) {
                    phi_rec = phi_hi; a_rec = a_hi; a_hi = a_lo; phi_hi = phi_lo;
                } else {
                    phi_rec = phi_lo; a_rec = a_lo;
                }
                a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
                ++zi;
                if (zi > 10) { status = 2; st = S_FINISH; break; }
                st = S_ZOOM_TOP;
            } break;
            case S_ACCEPT: {
                if (!acc_have_g) {  // gfkp1 is None -> myfprime(xkp1)
                    alpha = acc_alpha; need_f = false; need_g = true; want_eval = true;
                }
                st = S_ACCEPT2;
            } break;
            case S_ACCEPT2: {
                double s[VPL], y[VPL];
#pragma unroll
                for (int r = 0; r < VPL; ++r) {
                    s[r] = acc_alpha * p[r];
                    x[r] = x[r] + s[r];
                    y[r] = gv[r] - g[r];
                    g[r] = gv[r];
                }
                ++k;
                old_old_fval = phi0;
                old_fval = acc_f;
                double pp_acc, rhok_inv;
                if (STM_SM_BATCH) {   // max |g|, p . p and y . s in one batch of reductions
                    double rs[2] = {dot_lane(p, p), dot_lane(y, s)}, rm[1] = {maxabs_lane(g)};
                    wave_reduce_n(rs, rm);
                    pp_acc = rs[0]; rhok_inv = rs[1]; gnorm = rm[0];
                } else {
                    gnorm = maxabs(g);
                    pp_acc = 0.0; rhok_inv = 0.0;
                }
                if (gnorm <= gtol) { st = S_FINISH; break; }
                if (!STM_SM_BATCH) pp_acc = dot(p, p);
                if (acc_alpha * sqrt(pp_acc) <= 0.0) { st = S_FINISH; break; }  // xrtol = 0
                if (!finite_d(old_fval)) { status = 2; st = S_FINISH; break; }
                if (!STM_SM_BATCH) rhok_inv = dot(y, s);
                const double rhok = (rhok_inv == 0.0) ? 1000.0 : 1.0 / rhok_inv;
                // H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T, expanded (H symmetric):
                //   H - rho (s w^T + w s^T) + (rho^2 y^T w + rho) s s^T,  w = H y
                double w[VPL];
                if (H_ident) {
#pragma unroll
                    for (int r = 0; r < VPL; ++r) w[r] = y[r];
                } else matvecH(y, w);
                const double yHy = dot(y, w);
                const double cc = rhok * rhok * yHy + rhok;
#pragma unroll
                for (int r = 0; r < VPL; ++r)
                    if (lane + WAVE * r < n) {
                        sv[lane + WAVE * r] = s[r]; sw[lane + WAVE * r] = w[r];
                        if (NW == 1 && VPL == 2) se[lane + WAVE * r] = g[r];   // (free until the next evaluation's head_F rewrites it)
                    }
                if (NW == 2) {
                    if (lane == 0) { xch_res[6] = rhok; xch_res[7] = cc; xch_cmd[0] = 8 | (H_ident ? 16 : 0); }
                    __syncthreads();   // (0) wave 1 takes rows [nh, n)
                    if (lane < n) bfgs_rows(lane, 0, nh < n ? nh : n, H_ident, rhok, cc, s[0], w[0]);
                    __syncthreads();   // (u)
                } else {
                    STM_WAVE_SYNC();
                    if constexpr (VPL == 2) {   // H in HBM: both columns of the lane, eight rows (sixteen loads) per round
                        const int j0 = lane, j1 = lane + WAVE < n ? lane + WAVE : n - 1;
                        const bool v1 = lane + WAVE < n;
                        const double s0 = s[0], s1 = s[VPL - 1], w0 = w[0], w1 = w[VPL - 1];
                        double tg0 = 0.0, tg1 = 0.0;   // (H_new g)[j0], [j1]: sum over the rows i in order, as matvecH adds them
                        for (int i = 0; i < n; i += 8) {
                            double h0[8], h1[8], si[8], wi[8], gi[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int ic = i + q < n ? i + q : n - 1;
                                h0[q] = H_ident ? (ic == j0 ? 1.0 : 0.0) : Hs[(size_t)ic * n + j0];
                                h1[q] = H_ident ? (ic == j1 ? 1.0 : 0.0) : Hs[(size_t)ic * n + j1];
                                si[q] = sv[ic]; wi[q] = sw[ic]; gi[q] = se[ic];
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (i + q < n) {   // uniform
                                    const double n0 = h0[q] - rhok * (si[q] * w0 + wi[q] * s0) + cc * (si[q] * s0);
                                    const double n1 = h1[q] - rhok * (si[q] * w1 + wi[q] * s1) + cc * (si[q] * s1);
                                    Hs[(size_t)(i + q) * n + j0] = n0;
                                    if (v1) Hs[(size_t)(i + q) * n + j1] = n1;
                                    tg0 += n0 * gi[q]; tg1 += n1 * gi[q];
                                }
                        }
                        Hg[0] = tg0; Hg[VPL - 1] = v1 ? tg1 : 0.0;
                        have_Hg = true;
                    } else
#pragma unroll
                    for (int r = 0; r < VPL; ++r) {
                        const int j = lane + WAVE * r;
                        if (j < n) bfgs_rows(j, 0, n, H_ident, rhok, cc, s[r], w[r]);
                    }
                    STM_WAVE_SYNC_MEM();
                }
                H_ident = false;
                st = S_OUTER_TOP;
            } break;
            default: st = S_FINISH; break;
            }
            if (STM_PROF(P)) {
                const long long tq2 = (long long)__builtin_readcyclecounter();
                t_eval += tq1 - tq0;
                if (was_upd) t_upd += tq2 - tq1; else t_sm += tq2 - tq1;
                if (lane == 0) STM_PROF(P)[doc * PROF_SLOTS + 8 + st_in] += (tq2 - tq1) + (1LL << 40);   // cycles in the low 40 bits, visits above (slots 24.. are the post kernels')
            }
        }
        if (NW == 2) {  // release the evaluation server
            if (lane == 0) xch_cmd[0] = 4;
            __syncthreads();
        }
        if (status == 0) {
            if (k >= maxiter) status = 1;
            else {
                bool anynan = (gnorm != gnorm) || (old_fval != old_fval);
#pragma unroll
                for (int r = 0; r < VPL; ++r) anynan |= (x[r] != x[r]);
                if (wave_any(anynan)) status = 3;
            }
        }
#pragma unroll
        for (int r = 0; r < VPL; ++r) {
            const int i = lane + WAVE * r;
            if (i < n) P.eta[doc * n + i] = x[r];
        }
        // uniform stores (every lane writes the same word)
        if (STM_PROF(P) && lane == 0) {
            STM_PROF(P)[doc * PROF_SLOTS + 0] = t_init; STM_PROF(P)[doc * PROF_SLOTS + 1] = t_eval; STM_PROF(P)[doc * PROF_SLOTS + 2] = t_sm; STM_PROF(P)[doc * PROF_SLOTS + 3] = t_upd;
            // absolute begin / end of the document on the shader clock and the 100 MHz wall clock at its end (tools/slot_gaps.py: how many
            // documents the chip really has in flight -- what the dispatch of one workgroup per document costs; [39] the wall clock at its begin)
            STM_PROF(P)[doc * PROF_SLOTS + 45] = t_begin; STM_PROF(P)[doc * PROF_SLOTS + 46] = (long long)__builtin_readcyclecounter(); STM_PROF(P)[doc * PROF_SLOTS + 47] = (long long)wall_clock64();
        }
        if (P.status) P.status[doc] = status;
        if (P.nit) P.nit[doc] = k;
        if (P.nfev) P.nfev[doc] = nfev;
        if (P.njev) P.njev[doc] = njev;
        if (!persist) return;
    }
}

}  // namespace stm
