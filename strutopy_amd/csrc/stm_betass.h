// stm_betass.h -- beta_ss without atomics: the word-major pass over phi (K <= 64 path).
//
// Reference src/modules/stm.py:584-588 adds every document's phi (K x N_d, update_z stm.py:1103-1118) into the columns
// beta_ss[:, words of the document].  Doing that from the per-document kernel means one atomic 8K-byte row update per
// (document, word): 15 M of them per E-step at BASELINE configs[1], each a read-modify-write of lines that the 4 MB L2 of
// an XCD cannot keep (beta and beta_ss are 4 MB each) -- a floor of ~5 ms whatever the kernel around them does.
// phi factorises:   phi[k, (d, w)] = beta[v][k] * exp(eta~_d)[k] * c_dw / S_dw = beta[v][k] * theta_d[k] * r_dw,
//                   r_dw = sum_k exp(eta~_d)[k] * c_dw / S_dw   (S_dw = the column sum the post kernel has anyway),
// so the post kernel only stores the scalar r_dw per (document, word) -- scattered to the entry's slot in word-major order --
// and this pass computes, for every (level, word) row,
//   beta_ss[a][v][:] = beta[a][v][:] * sum_{(d, w): word v, level a} theta_d[:] * r_dw
// as one gather of theta rows per entry: a sparse (words x documents) times dense (documents x K) product over the corpus in
// word-major order (built once per corpus and K: a counting sort).  No atomics, every sum in a fixed order: beta_ss is
// run-to-run identical.
// The gather is what costs: 15 M rows of 8K bytes out of a theta that is ten times an XCD's L2.  So the documents are cut
// into G groups small enough for L2 (BETASS_GROUP_BYTES of theta), the grid is ordered group-major -- at any time the
// whole chip gathers from one or two groups -- and a wave writes the partial sum of its (row, group) cells; a second kernel
// adds the G partials of a row in order and multiplies by beta.
#pragma once
#include "stm_wave.h"

namespace stm {

struct BetaSsParams {
    int K, G;                 // topics; document groups
    int64_t R;                // rows = levels * V
    const int32_t *cptr;      // [R * G + 1] first entry of (row, group) in the word-major arrays
    const int32_t *wm_doc;    // word-major: document of the entry
    const double *rw;         // [nnz] r_dw, WORD-major (the post kernel scatters it there through its copy of the slot map)
    const double *theta;      // [N][K]
    const double *betaT;      // [A][V][K]
    double *part;             // [G][R][K] partial sums
    double *beta_ssT;         // [A][V][K]
};

constexpr int BETASS_GROUP_BYTES = 2 << 20;   // theta bytes per document group (0.5 / 1 / 2 / 4 MB measured at configs[1]: 4.25 / 4.14 / 4.11 / 4.29 ms post + pass; STM_BETASS_GROUP_KB overrides)
constexpr int BETASS_ROWS = 4;                // rows per wave (2 / 4 / 8 / 16 rows and 4 / 8 / 12 / 16 loads in flight measured; 4 rows, 8 loads)

// One wave per (ROWS rows, group), lane = topic; blockIdx is group-major.  A cell's (document, r) pairs are fetched
// lane-parallel -- the next cell's while the current one is consumed -- and handed out with v_readlane, DEPTH theta rows in flight.
// TPL: topics per lane (2 for 64 < K <= 128: topics lane and lane + 64).
template <int DEPTH, int ROWS, int TPL = 1>
__global__ __launch_bounds__(256) void beta_ss_part_kernel(BetaSsParams P) {
    const int lane = threadIdx.x & 63;
    const int K = P.K, G = P.G;
    const int64_t R = P.R, wpg = (R + ROWS - 1) / ROWS;     // waves per group
    const int64_t bpg = (wpg + 3) / 4;                                      // blocks per group
    const int g = (int)(blockIdx.x / bpg);
    const int64_t wv = (blockIdx.x % bpg) * 4 + (threadIdx.x >> 6);
    if (wv >= wpg) return;
    const int64_t r0 = wv * ROWS, r1 = r0 + ROWS < R ? r0 + ROWS : R;
    const int kl = lane < K ? lane : 0, kl1 = lane + WAVE < K ? lane + WAVE : 0;
    const double *th0 = P.theta + kl, *th1 = P.theta + kl1;
    // lane q <= rows: the cell boundaries cp[(r0 + q) * G + g] and, in the next lane block, their ends
    const int nr = (int)(r1 - r0);
    const int lo_l = lane < nr ? P.cptr[(r0 + lane) * G + g] : 0, hi_l = lane < nr ? P.cptr[(r0 + lane) * G + g + 1] : 0;
    auto fetch = [&](int e0, int e1, int &d, double &r) __attribute__((always_inline)) {   // entries [e0, min(e1, e0 + 64))
        const bool in = e0 + lane < e1;
        const int el = in ? e0 + lane : 0;                 // lanes beyond the batch read entry 0 and carry r = 0
        d = P.wm_doc[el];
        const double rv = P.rw[el];
        r = in ? rv : 0.0;
    };
    int q = 0;
    int e0 = __builtin_amdgcn_readlane(lo_l, 0), e1 = __builtin_amdgcn_readlane(hi_l, 0);
    int dl, dn = 0;
    double rl, rn = 0.0;
    fetch(e0, e1, dl, rl);
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, acc1[4] = {0.0, 0.0, 0.0, 0.0};
    while (q < nr) {
        const int cnt = e1 - e0 < WAVE ? e1 - e0 : WAVE;
        // the batch after this one: the rest of the cell, or the next row's cell
        int nq = q, n0 = e0 + WAVE, n1 = e1;
        if (n0 >= e1) {
            nq = q + 1;
            if (nq < nr) { n0 = __builtin_amdgcn_readlane(lo_l, nq); n1 = __builtin_amdgcn_readlane(hi_l, nq); }
        }
        if (nq < nr) fetch(n0, n1, dn, rn);
        for (int u = 0; u < cnt; u += DEPTH) {   // DEPTH loads all the same: the entries beyond cnt carry r = 0
            double th[DEPTH], tg[DEPTH], r[DEPTH];
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                const int ut = (u + t) & (WAVE - 1);
                const int d = __builtin_amdgcn_readlane(dl, ut);
                th[t] = th0[(size_t)d * K];
                if (TPL == 2) tg[t] = th1[(size_t)d * K];
                r[t] = lane_bcast(rl, ut);
            }
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                acc[t & 3] = fma(th[t], r[t], acc[t & 3]);
                if (TPL == 2) acc1[t & 3] = fma(tg[t], r[t], acc1[t & 3]);
            }
        }
        if (nq != q) {   // the cell is complete
            if (lane < K) P.part[((size_t)g * R + (size_t)(r0 + q)) * K + lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            if (TPL == 2 && lane + WAVE < K) P.part[((size_t)g * R + (size_t)(r0 + q)) * K + lane + WAVE] = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
            acc1[0] = acc1[1] = acc1[2] = acc1[3] = 0.0;
        }
        q = nq; e0 = n0; e1 = n1; dl = dn; rl = rn;
    }
}

// The same pass for even K: two theta rows per load instruction.  A lane owns one 16-byte piece of a row (two topics), lanes
// 0 .. K/2-1 serve the even entries of a batch and lanes K/2 .. K-1 the odd ones: half the memory instructions (each moves 2 x 8K
// bytes), half the FMAs' instruction count, twice the rows in flight per wave for the same DEPTH -- the pass is bound by how many
// gathers the CU keeps in the air.  A cell's sum is (even entries) + (odd entries), each in four round-robin partial sums.
// RPI = 1 (64 < K <= 128, even): a row is more than 32 pieces, one theta row per load instruction (lanes 0 .. K/2-1).
template <int DEPTH, int ROWS, int RPI = 2>
__global__ __launch_bounds__(256) void beta_ss_part2_kernel(BetaSsParams P) {
    const int lane = threadIdx.x & 63;
    const int K = P.K, G = P.G, CH = K >> 1;
    const int64_t R = P.R, wpg = (R + ROWS - 1) / ROWS;
    const int64_t bpg = (wpg + 3) / 4;
    const int g = (int)(blockIdx.x / bpg);
    const int64_t wv = (blockIdx.x % bpg) * 4 + (threadIdx.x >> 6);
    if (wv >= wpg) return;
    const int64_t r0 = wv * ROWS, r1 = r0 + ROWS < R ? r0 + ROWS : R;
    const int rr = (RPI == 2 && lane >= CH) ? 1 : 0, cc = lane - rr * CH;
    const bool act = lane < RPI * CH;
    const unsigned coff = 16u * (unsigned)(act ? cc : 0);
    const char *th0 = reinterpret_cast<const char *>(P.theta);
    const unsigned K8 = 8u * (unsigned)K;
    const int nr = (int)(r1 - r0);
    const int lo_l = lane < nr ? P.cptr[(r0 + lane) * G + g] : 0, hi_l = lane < nr ? P.cptr[(r0 + lane) * G + g + 1] : 0;
    auto fetch = [&](int e0, int e1, int &d, double &r) __attribute__((always_inline)) {   // entries [e0, min(e1, e0 + 64))
        const bool in = e0 + lane < e1;
        const int el = in ? e0 + lane : 0;                 // lanes beyond the batch read entry 0 and carry r = 0
        d = P.wm_doc[el];
        const double rv = P.rw[el];
        r = in ? rv : 0.0;
    };
    int q = 0;
    int e0 = __builtin_amdgcn_readlane(lo_l, 0), e1 = __builtin_amdgcn_readlane(hi_l, 0);
    int dl, dn = 0;
    double rl, rn = 0.0;
    fetch(e0, e1, dl, rl);
    double ax[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0};
    while (q < nr) {
        const int cnt = e1 - e0 < WAVE ? e1 - e0 : WAVE;
        int nq = q, n0 = e0 + WAVE, n1 = e1;
        if (n0 >= e1) {
            nq = q + 1;
            if (nq < nr) { n0 = __builtin_amdgcn_readlane(lo_l, nq); n1 = __builtin_amdgcn_readlane(hi_l, nq); }
        }
        if (nq < nr) fetch(n0, n1, dn, rn);
        const unsigned doff = (unsigned)dl * K8;             // (N K 8 < 4 GiB: checked by the host)
        const int rlo = __double2loint(rl), rhi = __double2hiint(rl);
        for (int u = 0; u < cnt; u += RPI * DEPTH) {        // DEPTH loads = RPI DEPTH entries; the entries beyond cnt carry r = 0
            double2 th[DEPTH];
            double r[DEPTH];
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                const int src = 4 * ((u + RPI * t + rr) & (WAVE - 1));
                const unsigned o = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)doff) + coff;
                th[t] = *reinterpret_cast<const double2 *>(th0 + o);
                r[t] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, rhi), __builtin_amdgcn_ds_bpermute(src, rlo));
            }
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                ax[t & 3] = fma(th[t].x, r[t], ax[t & 3]);
                ay[t & 3] = fma(th[t].y, r[t], ay[t & 3]);
            }
        }
        if (nq != q) {   // the cell is complete: even-entry lanes add their odd-entry partners' sums
            double sx = (ax[0] + ax[1]) + (ax[2] + ax[3]), sy = (ay[0] + ay[1]) + (ay[2] + ay[3]);
            if (RPI == 2) {
                const int partner = 4 * (lane + CH < WAVE ? lane + CH : lane);
                sx += __hiloint2double(__builtin_amdgcn_ds_bpermute(partner, __double2hiint(sx)), __builtin_amdgcn_ds_bpermute(partner, __double2loint(sx)));
                sy += __hiloint2double(__builtin_amdgcn_ds_bpermute(partner, __double2hiint(sy)), __builtin_amdgcn_ds_bpermute(partner, __double2loint(sy)));
            }
            if (lane < CH)
                *reinterpret_cast<double2 *>(P.part + ((size_t)g * R + (size_t)(r0 + q)) * K + 2 * lane) = make_double2(sx, sy);
#pragma unroll
            for (int t = 0; t < 4; ++t) { ax[t] = 0.0; ay[t] = 0.0; }
        }
        q = nq; e0 = n0; e1 = n1; dl = dn; rl = rn;
    }
}

// beta_ss[row][k] = beta[row][k] * sum_g part[g][row][k], groups in ascending order
__global__ __launch_bounds__(256) void beta_ss_reduce_kernel(BetaSsParams P) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, RK = P.R * P.K;
    if (q >= RK) return;
    double t = 0.0;
    for (int g = 0; g < P.G; ++g) t += P.part[(size_t)g * RK + q];
    P.beta_ssT[q] = P.betaT[q] * t;
}

}  // namespace stm
