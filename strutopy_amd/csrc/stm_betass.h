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

constexpr int BETASS_GROUP_BYTES = 1 << 20;   // theta bytes per document group
constexpr int BETASS_ROWS = 4;                // rows per wave (4 / 8 / 16 and 8 / 16 loads in flight measured: 0.56 - 0.73 ms at configs[1])

// One wave per (ROWS rows, group), lane = topic; blockIdx is group-major.  A cell's (document, r) pairs are fetched
// lane-parallel -- the next cell's while the current one is consumed -- and handed out with v_readlane, DEPTH theta rows in flight.
template <int DEPTH, int ROWS>
__global__ __launch_bounds__(256) void beta_ss_part_kernel(BetaSsParams P) {
    const int lane = threadIdx.x & 63;
    const int K = P.K, G = P.G;
    const int64_t R = P.R, wpg = (R + ROWS - 1) / ROWS;     // waves per group
    const int64_t bpg = (wpg + 3) / 4;                                      // blocks per group
    const int g = (int)(blockIdx.x / bpg);
    const int64_t wv = (blockIdx.x % bpg) * 4 + (threadIdx.x >> 6);
    if (wv >= wpg) return;
    const int64_t r0 = wv * ROWS, r1 = r0 + ROWS < R ? r0 + ROWS : R;
    const int kl = lane < K ? lane : 0;
    const double *th0 = P.theta + kl;
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
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
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
            double th[DEPTH], r[DEPTH];
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                const int ut = (u + t) & (WAVE - 1);
                const int d = __builtin_amdgcn_readlane(dl, ut);
                th[t] = th0[(size_t)d * K];
                r[t] = lane_bcast(rl, ut);
            }
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) acc[t & 3] = fma(th[t], r[t], acc[t & 3]);
        }
        if (nq != q) {   // the cell is complete
            if (lane < K) P.part[((size_t)g * R + (size_t)(r0 + q)) * K + lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
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
