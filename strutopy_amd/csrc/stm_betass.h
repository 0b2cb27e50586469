// stm_betass.h -- beta_ss without atomics: the word-major pass over phi (K <= 64 path).
//
// Reference src/modules/stm.py:584-588 adds every document's phi (K x N_d, update_z stm.py:1103-1118) into the columns
// beta_ss[:, words of the document].  Doing that from the per-document kernel means one atomic 8K-byte row update per
// (document, word): 15 M of them per E-step at BASELINE configs[1], each a read-modify-write of lines that the 4 MB L2 of
// an XCD cannot keep (beta and beta_ss are 4 MB each) -- a floor of ~5 ms whatever the kernel around them does.
// phi factorises:   phi[k, (d, w)] = beta[v][k] * exp(eta~_d)[k] * c_dw / S_dw = beta[v][k] * theta_d[k] * r_dw,
//                   r_dw = sum_k exp(eta~_d)[k] * c_dw / S_dw   (S_dw = the column sum the post kernel has anyway),
// so the post kernel only stores the scalar r_dw per (document, word) and this kernel computes, for every (level, word) row,
//   beta_ss[a][v][:] = beta[a][v][:] * sum_{(d, w): word v, level a} theta_d[:] * r_dw
// as one gather of theta rows per entry -- a sparse (words x documents) times dense (documents x K) product over the
// corpus in word-major order (built once per corpus by stm_set_corpus: a counting sort).  No atomics, and the entries of a
// row are added in ascending document order: beta_ss is run-to-run identical.  Rows longer than SEG entries (the
// most frequent words of a real vocabulary) are cut into segments whose partial sums are added atomically.
#pragma once
#include "stm_wave.h"

namespace stm {

struct BetaSsParams {
    int K;
    int64_t nseg;
    const int32_t *seg_row;   // (level * V + word) of the segment
    const int32_t *seg_lo;    // entries [seg_lo, seg_hi) of the word-major arrays
    const int32_t *seg_hi;
    const uint8_t *seg_multi; // 1: the row has more than one segment (atomic add instead of a store)
    const int32_t *wm_doc;    // word-major: document of the entry
    const int32_t *wm_pos;    // word-major: position of the entry in the document-major (CSR) arrays
    const double *rw;         // [nnz] r_dw, document-major (post kernel)
    const double *theta;      // [N][K]
    const double *betaT;      // [A][V][K]
    double *beta_ssT;         // [A][V][K] (pre-zeroed)
};

constexpr int BETASS_SEG = 4096;

// one wave per segment, lane = topic; 64 entries per batch: their (document, r) pairs are fetched lane-parallel and handed
// out with v_readlane, eight theta rows in flight
__global__ __launch_bounds__(256) void beta_ss_kernel(BetaSsParams P) {
    const int lane = threadIdx.x & 63;
    const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= P.nseg) return;
    const int K = P.K;
    const int row = P.seg_row[seg], lo = P.seg_lo[seg], hi = P.seg_hi[seg];
    const int kl = lane < K ? lane : 0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int e0 = lo; e0 < hi; e0 += WAVE) {
        const int cnt = hi - e0 < WAVE ? hi - e0 : WAVE;
        const int el = e0 + (lane < cnt ? lane : 0);
        const int dl = P.wm_doc[el];
        const double rl = lane < cnt ? P.rw[P.wm_pos[el]] : 0.0;
        int u = 0;
        for (; u + 7 < cnt; u += 8) {
            double th[8], r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int d = __builtin_amdgcn_readlane(dl, u + q);
                th[q] = P.theta[(size_t)d * K + kl];
                r[q] = lane_bcast(rl, u + q);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q & 3] = fma(th[q], r[q], acc[q & 3]);
        }
        for (; u < cnt; ++u) {
            const int d = __builtin_amdgcn_readlane(dl, u);
            acc[u & 3] = fma(P.theta[(size_t)d * K + kl], lane_bcast(rl, u), acc[u & 3]);
        }
    }
    if (lane < K) {
        const double v = P.betaT[(size_t)row * K + lane] * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (P.seg_multi[seg]) unsafeAtomicAdd(P.beta_ssT + (size_t)row * K + lane, v);
        else P.beta_ssT[(size_t)row * K + lane] = v;
    }
}

}  // namespace stm
