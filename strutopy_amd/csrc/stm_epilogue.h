// stm_epilogue.h -- what follows the post kernel in one EM iteration, and what precedes the solver in the next, in few launches.
//
// Round 5 ran twelve dispatches of 4-8 us between the post kernel and the host's read-back (reduce x2, untile, bound x2, moments,
// reduce x2, covariance, reduce x2, copy-out), seven between the read-back and the next solver (copy, set_mu, rowsum, reduce x2,
// normalise, colsum) and three in front of the solver (copy, two fills).  Here the same arithmetic IN THE SAME ORDER -- every sum
// below adds its terms exactly as the kernel it replaces did, so the fit stays bit for bit where it was (tools/bitcmp.py) -- runs as
//
//   estep_head_kernel      siginv from the pinned staging area, the error flag + ticket counters and the nu slabs zeroed
//   epilogue_a_kernel      roles by block range: first-stage sum of the nu slabs | per-block sums of the bounds |
//                          regression moments (stm.py:678-706) | eta^T eta (stm.py:723)                          -- all independent
//   epilogue_b_kernel      sigma_ss laid out from the summed tiles | the bound (stm.py:592) + error slot |
//                          both stages of the moment / covariance reductions in one block per sixteen slots
//   (copy_out_err_kernel   the packed head to the pinned read-back area: behind the all-reduce when there is a communicator)
//   mstep_tail_a_kernel    mu = X gamma^T (stm.py:706; gamma read from the pinned staging area into the LDS) | word-major column
//                          sums of beta_ss
//   mstep_tail_b_kernel    beta = beta_ss / rowsum (stm.py:741-745) and the per-word column sums of the new beta the solver divides by
//
// No workgroup of a launch reads what another one of the SAME launch wrote: the launch boundary is the hand-off everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stm_mstep.h"
#include "stm_post_common.h"

namespace stm {

constexpr int EPI_RED_Y = 32;        // rows of a two-stage reduction (reduce_copies in stm_api.hip)
constexpr int EPI_BOUND_BLOCKS = 128;
constexpr int EPI_COV_BLOCKS = 1024;

// What reduce_sigma_kernel's block row `y` leaves for slot q: the copies [lo, hi) split over four groups, four partial sums
// per group, a fixed combine -- here by ONE thread, the four groups in lockstep (sixteen loads in flight per step).
__device__ __forceinline__ double reduce_row_value(const double *part, size_t nn, int q, int lo, int hi) {
    const int cnt = hi > lo ? hi - lo : 0;
    const int per = (cnt + 3) >> 2;
    double t[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) t[g][u] = 0.0;
    for (int s = 0; s < per; s += 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int b = lo + g * per + s;
            const int b1 = (lo + g * per + per) < hi ? (lo + g * per + per) : hi;
            if (b + 3 < b1) {
                const double a0 = part[(size_t)b * nn + q], a1 = part[(size_t)(b + 1) * nn + q];
                const double a2 = part[(size_t)(b + 2) * nn + q], a3 = part[(size_t)(b + 3) * nn + q];
                t[g][0] += a0; t[g][1] += a1; t[g][2] += a2; t[g][3] += a3;
            } else {
                for (; b < b1; ++b) t[g][0] += part[(size_t)b * nn + q];
            }
        }
    }
    double sh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) sh[g] = (t[g][0] + t[g][1]) + (t[g][2] + t[g][3]);
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// reduce_copies(part, nblocks, nn, out) of stm_api.hip for the sixteen slots 16 bx .. by one 256-thread block: single stage below
// 4 EPI_RED_Y copies, else EPI_RED_Y row sums of ceil(nblocks / EPI_RED_Y) copies each (thread = slot x row pair) and their sum.
__device__ __forceinline__ void reduce_copies_block16(const double *part, int nblocks, int nn, double *out, int bx, double *lds /* [EPI_RED_Y][16] */) {
    const int tx = threadIdx.x & 15, yl = threadIdx.x >> 4;
    const int q = bx * 16 + tx;
    if (nblocks < 4 * EPI_RED_Y) {
        if (yl == 0 && q < nn) out[q] = reduce_row_value(part, (size_t)nn, q, 0, nblocks);
        return;
    }
    const int chunk = (nblocks + EPI_RED_Y - 1) / EPI_RED_Y;
    if (q < nn) {
#pragma unroll
        for (int u = 0; u < EPI_RED_Y / 16; ++u) {
            const int y = yl + 16 * u;
            const int lo = y * chunk, hi = lo + chunk < nblocks ? lo + chunk : nblocks;
            lds[y * 16 + tx] = reduce_row_value(part, (size_t)nn, q, lo, hi);
        }
    }
    __syncthreads();
    if (yl == 0 && q < nn) out[q] = reduce_row_value(lds, 16, tx, 0, EPI_RED_Y);
}

struct EpiParams {
    // nu: the per-workgroup slabs (or atomically filled replicas) -> sigma_ss
    const double *sig_part;
    int sig_copies, sig_nn, sig_chunk, sig_rows;   // rows: 1 (fewer than 4 EPI_RED_Y copies: one stage) or EPI_RED_Y
    double *sig_red;                               // [rows][sig_nn]
    int n, sig_layout;                             // 0: n x n, lower block triangle mirrored; 1: accumulator tiles; 2: tiles, last column in a slot of its own
    double *sigma_ss;
    // bound
    const double *bound;
    int64_t N;
    double *bound_part, *scal;
    const int32_t *err;
    // regression moments and eta^T eta (nb_mom == 0: E-step only)
    const double *X, *eta;
    int p, Lr;
    double *mom_part, *mom_out, *cov_part, *cov_out;
    int cov_g;                                     // ceil(n / 64)
    // blocks per role
    int nb_cov, nb_mom, nb_sig, nb_bound;          // stage A
    int nb_sig2, nb_mom2, nb_cov2;                 // stage B (+ one block for the bound)
};

// ---- the kernels of stm_mstep.h / stm_post_common.h as block bodies with explicit block coordinates ------------------------------
// moments_kernel's block bx of gx -- the same sums, eight threads to a slot: the kernel's thread keeps eight partial sums t[0..7] over
// sixteen documents at a time (t[j] takes documents d0 + 8 m + j in order, t[0] the remainder behind the last full sixteen) and combines
// them in a fixed tree; here thread (slot, j) IS t[j] -- twelve dependent additions instead of ninety-eight at 100k documents -- and
// the tree runs over the LDS.  Same operands, same order, same bits.
__device__ __forceinline__ void moments_block(const double *X, const double *eta, int64_t N, int p, int n, double *part, int L, int bx, int gx,
                                              double *sh /* [8][32] */) {
    const int64_t chunk = (N + gx - 1) / gx;
    const int64_t d0 = (int64_t)bx * chunk;
    const int64_t d1 = d0 + chunk < N ? d0 + chunk : N;
    const int64_t len = d1 > d0 ? d1 - d0 : 0;
    const int64_t full = len / 16 * 16;               // documents in full groups of sixteen
    const int sl = threadIdx.x & 31, j = threadIdx.x >> 5;
    auto strided = [&](auto term) {
        double t = 0.0;
        int64_t m = 0;
        for (; m + 3 < full / 8; m += 4) {             // four of the thread's documents in flight
            const double v0 = term(d0 + 8 * m + j), v1 = term(d0 + 8 * (m + 1) + j), v2 = term(d0 + 8 * (m + 2) + j), v3 = term(d0 + 8 * (m + 3) + j);
            t += v0; t += v1; t += v2; t += v3;
        }
        for (; m < full / 8; ++m) t += term(d0 + 8 * m + j);
        if (j == 0)
            for (int64_t d = d0 + full; d < d1; ++d) t += term(d);
        return t;
    };
    for (int s0 = 0; s0 < L; s0 += 32) {
        const int slot = s0 + sl;
        double t = 0.0;
        if (slot < L) {
            int s = slot;
            if (s == 0) {
                t = j == 0 ? (double)len : 0.0;
            } else if ((s -= 1) < p) {
                t = strided([&](int64_t d) { return X[d * p + s]; });
            } else if ((s -= p) < n) {
                t = strided([&](int64_t d) { return eta[d * n + s]; });
            } else if ((s -= n) < p * p) {
                const int a = s / p, b = s % p;
                t = strided([&](int64_t d) { return X[d * p + a] * X[d * p + b]; });
            } else {
                s -= p * p;
                const int a = s / n, i = s % n;
                t = strided([&](int64_t d) { return X[d * p + a] * eta[d * n + i]; });
            }
        }
        __syncthreads();
        sh[j * 32 + sl] = t;
        __syncthreads();
        if (j == 0 && slot < L) {
            const double r = slot == 0 ? sh[sl]
                                       : ((sh[sl] + sh[32 + sl]) + (sh[64 + sl] + sh[96 + sl])) + ((sh[128 + sl] + sh[160 + sl]) + (sh[192 + sl] + sh[224 + sl]));
            part[(size_t)bx * L + slot] = r;
        }
    }
}

constexpr int COV_TD = 32;   // documents per LDS tile of the covariance block
// covariance_kernel's block (bx, by, bz).  ONE: n <= 64 -- the row-side and the column-side components of a tile are the same 64, held
// once (half the loads, half the LDS: nine blocks per CU instead of four); the sums are the same, term for term.
template <bool ONE>
__device__ __forceinline__ void covariance_block(const double *eta, const double *mu, int64_t N, int n, double *part, int bx, int by, int bz, int gx,
                                                 double *tile /* [COV_TD][ONE ? 66 : 130] */) {
    constexpr int W = ONE ? 64 : 128, LDW = W + 2, PER = COV_TD * W / 256;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int jc = 4 * tx + 64 * by, ib = 4 * ty + 64 * bz;
    const int ia = 4 * ty, ja = (ONE ? 0 : 64) + 4 * tx;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    const int64_t chunk = (N + gx - 1) / gx;
    const int64_t d0 = (int64_t)bx * chunk;
    const int64_t d1 = d0 + chunk < N ? d0 + chunk : N;
    // (the next tile's global loads are issued behind the current tile's LDS stores and land while it is multiplied: a block's four
    // tiles at 100k documents were four exposed round trips to memory)
    double v[PER];
    auto fetch = [&](int64_t base) __attribute__((always_inline)) {
        const int cnt = (int)((d1 - base) < COV_TD ? (d1 - base) : COV_TD);
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int q = threadIdx.x + 256 * it, dd = q / W, c = q & (W - 1);
            const int i = ONE ? c : (c < 64 ? 64 * bz : 64 * by - 64) + c;
            const bool in = dd < cnt && i < n;
            const int64_t at = in ? (base + dd) * n + i : 0;
            const double e = eta[at], m = mu ? mu[at] : 0.0;
            v[it] = in ? (mu ? e - m : e) : 0.0;
        }
    };
    if (d0 < d1) fetch(d0);
    for (int64_t base = d0; base < d1; base += COV_TD) {
        const int cnt = (int)((d1 - base) < COV_TD ? (d1 - base) : COV_TD);
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int q = threadIdx.x + 256 * it;
            tile[(q / W) * LDW + (q & (W - 1))] = v[it];
        }
        __syncthreads();
        if (base + COV_TD < d1) fetch(base + COV_TD);
        for (int dd = 0; dd < cnt; ++dd) {
            const double2 *row = reinterpret_cast<const double2 *>(tile + dd * LDW);
            const double2 a01 = row[ia >> 1], a23 = row[(ia >> 1) + 1], b01 = row[ja >> 1], b23 = row[(ja >> 1) + 1];
            const double a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) acc[u][vv] = fma(a[u], b[vv], acc[u][vv]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = ib + u, j = jc + v;
            if (i < n && j < n) part[(size_t)bx * n * n + (size_t)i * n + j] = acc[u][v];
        }
}

// reduce_sigma_kernel's block (bx, by)
__device__ __forceinline__ void reduce_stage1_block(const double *part, int nblocks, int nn, double *out, int chunk, int bx, int by, double *sh /* [4][64] */) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int q = bx * 64 + tx;
    const int lo = by * chunk, hi = lo + chunk < nblocks ? lo + chunk : nblocks;
    const int cnt = hi > lo ? hi - lo : 0;
    const int per = (cnt + 3) >> 2;
    const int b0 = lo + ty * per, b1 = b0 + per < hi ? b0 + per : hi;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (q < nn) {
        int b = b0;
        for (; b + 3 < b1; b += 4) {
            const double a = part[(size_t)b * nn + q], c = part[(size_t)(b + 1) * nn + q];
            const double d = part[(size_t)(b + 2) * nn + q], e = part[(size_t)(b + 3) * nn + q];
            t0 += a; t1 += c; t2 += d; t3 += e;
        }
        for (; b < b1; ++b) t0 += part[(size_t)b * nn + q];
    }
    sh[ty * 64 + tx] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (ty == 0 && q < nn) out[(size_t)by * nn + q] = (sh[tx] + sh[64 + tx]) + (sh[128 + tx] + sh[192 + tx]);
}

__device__ __forceinline__ void bound_partial_block(const double *bound, int64_t N, double *part, int bx, int gx, double *sh /* [256] */) {
    const int64_t chunk = (N + gx - 1) / gx;
    const int64_t d0 = (int64_t)bx * chunk, d1 = d0 + chunk < N ? d0 + chunk : N;
    double t = 0.0;
    for (int64_t i = d0 + threadIdx.x; i < d1; i += 256) t += bound[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[bx] = sh[0];
}

// ---- stage A: everything that reads what the post kernel left and nothing else -------------------------------------------------------
template <bool ONE>   // ONE: n <= 64 (covariance_block)
__global__ __launch_bounds__(256) void epilogue_a_kernel(EpiParams ep) {
    __shared__ __attribute__((aligned(16))) double smem[COV_TD * (ONE ? 66 : 130)];
    int id = blockIdx.x;
    if (id < ep.nb_cov) {       // the longest blocks first
        const int bx = id % EPI_COV_BLOCKS, r = id / EPI_COV_BLOCKS;
        covariance_block<ONE>(ep.eta, nullptr, ep.N, ep.n, ep.cov_part, bx, r % ep.cov_g, r / ep.cov_g, EPI_COV_BLOCKS, smem);
        return;
    }
    id -= ep.nb_cov;
    if (id < ep.nb_mom) {
        moments_block(ep.X, ep.eta, ep.N, ep.p, ep.n, ep.mom_part, ep.Lr, id, ep.nb_mom, smem);
        return;
    }
    id -= ep.nb_mom;
    if (id < ep.nb_sig) {
        const int gx = (ep.sig_nn + 63) / 64;
        reduce_stage1_block(ep.sig_part, ep.sig_copies, ep.sig_nn, ep.sig_red, ep.sig_chunk, id % gx, id / gx, smem);
        return;
    }
    id -= ep.nb_sig;
    bound_partial_block(ep.bound, ep.N, ep.bound_part, id, ep.nb_bound, smem);
}

// ---- stage B: the second stages ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void epilogue_b_kernel(EpiParams ep) {
    __shared__ double smem[EPI_RED_Y * 16];
    int id = blockIdx.x;
    if (id < ep.nb_cov2) {
        reduce_copies_block16(ep.cov_part, EPI_COV_BLOCKS, ep.n * ep.n, ep.cov_out, id, smem);
        return;
    }
    id -= ep.nb_cov2;
    if (id < ep.nb_mom2) {
        reduce_copies_block16(ep.mom_part, ep.nb_mom, ep.Lr, ep.mom_out, id, smem);
        return;
    }
    id -= ep.nb_mom2;
    if (id < ep.nb_sig2) {     // sigma_ss[i][j] from the slot that holds it (untile_sigma_kernel / mirror_blocks_kernel), summed over the rows
        const int n = ep.n, q = id * 256 + (int)threadIdx.x;
        if (q >= n * n) return;
        int i = q / n, j = q % n;
        size_t src;
        if (ep.sig_layout == 0) {
            src = ((i >> 4) > (j >> 4)) ? (size_t)j * n + i : (size_t)q;
        } else {
            if (i > j) { const int t = i; i = j; j = t; }
            if (ep.sig_layout == 2 && j == n - 1) {
                const int nb = (n - 1) >> 4;
                src = (size_t)(nb * (nb + 1) / 2) * 256 + i;
            } else {
                const int b = i >> 4, bj = j >> 4, il = i & 15, fr = j & 15;
                src = ((size_t)(bj * (bj + 1) / 2 + b) * 4 + (il >> 2)) * 64 + (il & 3) * 16 + fr;
            }
        }
        ep.sigma_ss[q] = ep.sig_rows == 1 ? ep.sig_red[src] : reduce_row_value(ep.sig_red, (size_t)ep.sig_nn, (int)src, 0, EPI_RED_Y);
        return;
    }
    // bound = np.sum(calculated_bounds) (stm.py:592): reduce_bound_kernel's tree over the EPI_BOUND_BLOCKS block sums (its upper levels add zeros)
    double *sh = smem;
    sh[threadIdx.x] = (int)threadIdx.x < ep.nb_bound ? 0.0 + ep.bound_part[threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ep.scal[0] = sh[0];
        ep.scal[1] = (ep.err && *ep.err) ? 1.0 : 0.0;
    }
}

// ---- in front of the solver: siginv out of the pinned staging area, flag + ticket counters and the nu slabs zeroed ---------------------------
__global__ __launch_bounds__(256) void estep_head_kernel(const double *sig_src, double *siginv, int n2, int32_t *err, int nerr,
                                                         double *slabs, size_t nslab, double *bss, size_t nbss) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    if (sig_src)
        for (size_t q = tid; q < (size_t)n2; q += nt) siginv[q] = sig_src[q];
    for (size_t q = tid; q < (size_t)nerr; q += nt) err[q] = 0;
    double2 *s2 = reinterpret_cast<double2 *>(slabs);   // (hipMalloc'ed: 256-byte aligned)
    const double2 z = {0.0, 0.0};
    for (size_t q = tid; q < nslab / 2; q += nt) s2[q] = z;
    if ((nslab & 1) && tid == 0) slabs[nslab - 1] = 0.0;
    double2 *b2 = reinterpret_cast<double2 *>(bss);
    if (((uintptr_t)bss & 15) == 0) {
        for (size_t q = tid; q < nbss / 2; q += nt) b2[q] = z;
        if ((nbss & 1) && tid == 0) bss[nbss - 1] = 0.0;
    } else {
        for (size_t q = tid; q < nbss; q += nt) bss[q] = 0.0;
    }
}

// ---- M-step tail -------------------------------------------------------------------------------------------------------------------------------
struct TailParams {
    // mu
    const double *X, *coef;      // coef: gamma [n][p] (regression) or mean_eta [n] (X == nullptr); pinned host or device memory
    int64_t N;
    int p, n, coef_len, coef_in_lds;
    double *mu;
    int nb_mu;
    // beta
    const double *bssT;
    int V, K;
    double *rs_part;             // [128][K]
    int nb_rs_x, nb_rs;          // rowsum role: nb_rs_x x ceil(K / 64) blocks
    double *betaT, *colsum;
    int wpb;                     // word rows per block of stage B
};

__global__ __launch_bounds__(256) void mstep_tail_a_kernel(TailParams tp) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    int id = blockIdx.x;
    if (id < tp.nb_rs) {   // beta_rowsum_kernel's block (id % nb_rs_x, id / nb_rs_x)
        const int bx = id % tp.nb_rs_x, by = id / tp.nb_rs_x;
        const int kl = threadIdx.x & 63, sub = threadIdx.x >> 6;
        const int k = kl + 64 * by;
        const int chunk = (tp.V + tp.nb_rs_x - 1) / tp.nb_rs_x;
        const int v0 = bx * chunk, v1 = v0 + chunk < tp.V ? v0 + chunk : tp.V;
        double t = 0.0;
        if (k < tp.K)
            for (int v = v0 + sub; v < v1; v += 4) t += tp.bssT[(size_t)v * tp.K + k];
        dyn[sub * 64 + kl] = t;
        __syncthreads();
        if (sub == 0 && k < tp.K) tp.rs_part[(size_t)bx * tp.K + k] = (dyn[kl] + dyn[64 + kl]) + (dyn[128 + kl] + dyn[192 + kl]);
        return;
    }
    id -= tp.nb_rs;
    // set_mu_kernel, grid-strided, the coefficients staged once per block (one PCIe read of n p doubles instead of a copy dispatch)
    const double *coef = tp.coef;
    if (tp.coef_in_lds) {
        for (int q = threadIdx.x; q < tp.coef_len; q += 256) dyn[q] = tp.coef[q];
        __syncthreads();
        coef = dyn;
    }
    const int64_t tot = tp.N * tp.n;
    const int n = tp.n, p = tp.p;
    // four elements per thread and round: their covariate loads fly together (one element at a time is a chain of dependent round
    // trips to memory, nine of them per thread at 100k documents)
    auto four = [&](const int64_t (&q)[4], const int64_t (&d)[4], const int (&i)[4], int cnt) __attribute__((always_inline)) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        if (!tp.X) {
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = coef[i[u]];
        } else {
            for (int a = 0; a < p; ++a) {
                double xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = tp.X[d[u] * p + a];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] += xv[u] * coef[(size_t)i[u] * p + a];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < cnt) tp.mu[q[u]] = t[u];
    };
    const int64_t step = (int64_t)tp.nb_mu * 256;
    const bool small = tot < ((int64_t)1 << 32);   // (a 64-bit division per element is most of the arithmetic otherwise)
    for (int64_t q0 = (int64_t)id * 256 + threadIdx.x; q0 < tot; q0 += 4 * step) {
        int64_t q[4], d[4];
        int i[4], cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t qq = q0 + u * step;
            const bool in = qq < tot;
            cnt += in ? 1 : 0;
            q[u] = in ? qq : q0;                  // (beyond the end: the first element again, computed and not stored)
            if (small) { const unsigned du = (unsigned)q[u] / (unsigned)n; d[u] = du; i[u] = (int)((unsigned)q[u] - du * (unsigned)n); }
            else { d[u] = q[u] / n; i[u] = (int)(q[u] % n); }
        }
        four(q, d, i, cnt);
    }
}

// beta[v][k] = beta_ss[v][k] / rowsum[k] (0 where the sum is 0; stm.py:741-745) for wpb word rows per block, and colsum[v] as
// beta_colsum_kernel forms it.  rowsum[k]: reduce_copies over the 128 block sums of stage A, both of its stages redone per block
// (K x 128 L2 hits) in sixty-four-topic pieces.
__global__ __launch_bounds__(256) void mstep_tail_b_kernel(TailParams tp) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int K = tp.K;
    double *rs = dyn;                       // [K]
    double *s1 = dyn + ((K + 1) & ~1);      // [EPI_RED_Y][64]
    double *rows = s1 + EPI_RED_Y * 64;     // [wpb][K]
    const int nb = tp.nb_rs_x;              // copies
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int kc = K - k0 < 64 ? K - k0 : 64;
        if (nb < 4 * EPI_RED_Y) {
            if ((int)threadIdx.x < kc) rs[k0 + threadIdx.x] = reduce_row_value(tp.rs_part, (size_t)K, k0 + (int)threadIdx.x, 0, nb);
        } else {
            const int chunk = (nb + EPI_RED_Y - 1) / EPI_RED_Y;
            double val[EPI_RED_Y * 64 / 256];   // (every row sum's loads in flight at once: the rounds are memory round trips)
#pragma unroll
            for (int u = 0; u < EPI_RED_Y * 64 / 256; ++u) {
                const int e = threadIdx.x + 256 * u;
                const int y = e / kc, kk = e % kc;
                const int lo = y * chunk, hi = lo + chunk < nb ? lo + chunk : nb;
                val[u] = e < EPI_RED_Y * kc ? reduce_row_value(tp.rs_part, (size_t)K, k0 + kk, lo, hi) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < EPI_RED_Y * 64 / 256; ++u) {
                const int e = threadIdx.x + 256 * u;
                if (e < EPI_RED_Y * kc) s1[(e / kc) * 64 + e % kc] = val[u];
            }
            __syncthreads();
            if ((int)threadIdx.x < kc) rs[k0 + threadIdx.x] = reduce_row_value(s1, 64, (int)threadIdx.x, 0, EPI_RED_Y);
        }
        __syncthreads();
    }
    const int v0 = blockIdx.x * tp.wpb, v1 = v0 + tp.wpb < tp.V ? v0 + tp.wpb : tp.V;
    const int cnt = (v1 - v0) * K;
    const size_t base = (size_t)v0 * K;
    for (int e0 = threadIdx.x; e0 < cnt; e0 += 4 * 256) {   // (four loads in flight: one at a time is a chain of round trips to memory)
        double bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[u] = tp.bssT[base + (e0 + 256 * u < cnt ? e0 + 256 * u : e0)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < cnt) {
                const double r = rs[e % K];
                const double v = (r != 0.0) ? bv[u] / r : 0.0;
                tp.betaT[base + e] = v;
                rows[e] = v;
            }
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < v1 - v0; w += 256) {
        double t = 0.0;
        bool bad = false;
        for (int k = 0; k < K; ++k) {
            const double v = rows[w * K + k];
            bad |= !(v >= 0.0);
            t += v;
        }
        tp.colsum[v0 + w] = bad ? __builtin_nan("") : t;
    }
}

// 3-D beta (stm.py:741 with a 3-D array: the sum runs over topics) + the column sums of the result
__global__ void beta_normalise_topics_colsum_kernel(const double *bssT, int64_t AV, int K, double *betaT, double *colsum) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= AV) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += bssT[r * K + k];
    double t = 0.0;
    bool bad = false;
    for (int k = 0; k < K; ++k) {
        const double v = (s != 0.0) ? bssT[r * K + k] / s : 0.0;
        betaT[r * K + k] = v;
        bad |= !(v >= 0.0);
        t += v;
    }
    colsum[r] = bad ? __builtin_nan("") : t;
}

}  // namespace stm
