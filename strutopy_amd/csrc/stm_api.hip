// stm_api.hip -- C-ABI (include/stm_estep.h) over the gfx950 E-step kernels.
// Host side: device buffers, the single HIP stream, launches, HIP-event timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/stm_estep.h"
#include "stm_mstep.h"
#include "stm_betass.h"
#include "stm_post.h"
#include "stm_post_big2.h"
#include "stm_post_any.h"
#include "stm_solver.h"
#include "stm_epilogue.h"

namespace {

thread_local std::string g_err;

// doubles reserved in the packed all-reduce buffer for the M-step moments
// [ n_docs | sum_x (p) | sum_eta (n) | XtX (p*p) | Xt_eta (p*n) | eta^T eta (n*n) ]: sized for p <= 8 covariate
// columns when the model is set up, grown by stm_put_covariates when X is wider
constexpr size_t moments_len(int p, int n) { return 1 + (size_t)p + n + (size_t)p * p + (size_t)p * n + (size_t)n * n; }
constexpr size_t round64(size_t x) { return (x + 63) / 64 * 64; }
constexpr int MOM_BLOCKS = stm::EPI_COV_BLOCKS;   // partial-sum blocks of the moment / covariance passes (4 per CU)
// ... of the regression moments: fewer when X is wide (Lr ~ p^2 doubles per block; at most 128 MB of partials)
inline int mom_blocks(size_t Lr) { return (int)std::max<size_t>(16, std::min<size_t>(MOM_BLOCKS, ((size_t)1 << 24) / std::max<size_t>(Lr, 1))); }

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError(); /* the runtime's sticky error must not fail the NEXT call's hipGetLastError() check */ \
            return fail(STM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
        }                                                                                   \
    } while (0)

template <class T>
int dalloc(T **p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) return fail(STM_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    return STM_OK;
}
// grow-only allocation for buffers reused on every EM iteration (no free / malloc in the loop)
template <class T>
int ensure(T **p, size_t *cur, size_t count) {
    if (*p && *cur >= count) return STM_OK;
    if (int rc = dalloc(p, count)) return rc;
    *cur = count;
    return STM_OK;
}
template <class T>
void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

// [A][R][C] -> [A][C][R]
__global__ void transpose_kernel(const double *in, double *out, int R, int C) {
    __shared__ double tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * C;
    int c = blockIdx.x * 32 + threadIdx.x, r = blockIdx.y * 32 + threadIdx.y;
    for (int dy = 0; dy < 32; dy += 8)
        if (r + dy < R && c < C) tile[threadIdx.y + dy][threadIdx.x] = in[base + (size_t)(r + dy) * C + c];
    __syncthreads();
    int oc = blockIdx.y * 32 + threadIdx.x, orow = blockIdx.x * 32 + threadIdx.y;
    for (int dy = 0; dy < 32; dy += 8)
        if (orow + dy < C && oc < R) out[base + (size_t)(orow + dy) * R + oc] = tile[threadIdx.x][threadIdx.y + dy];
}

// colsum[a][w] = sum_k betaT[a][w][k], added in topic order (what np.sum(beta_doc_kv, axis=0) of stm.py:954 gives for the word's column).
// A row with an entry that is negative or NaN gets NaN: the E-step's `assert np.all(beta_doc_kv >= 0)` (stm.py:534) fails for
// exactly the documents that contain such a word, and the solver forms that test from the column sums it loads anyway
// (!(colsum >= 0)) instead of comparing all K entries of every gathered row again in every document.
__global__ void beta_colsum_kernel(const double *betaT, int64_t AV, int K, double *colsum) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= AV) return;
    const double *row = betaT + r * K;
    double t = 0.0;
    bool bad = false;
    for (int k = 0; k < K; ++k) {
        const double v = row[k];
        bad |= !(v >= 0.0);
        t += v;
    }
    colsum[r] = bad ? __builtin_nan("") : t;
}

// small device -> pinned-host copy done by the GPU itself (no DMA engine round trip)
__global__ void copy_out_kernel(const double *src, double *dst, size_t cnt) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < cnt) dst[q] = src[q];
}
// ... and the device error flag with it (a 4-byte hipMemcpyAsync of its own is a blit kernel and a gap on the stream: 11 us per EM iteration)
__global__ void copy_out_err_kernel(const double *src, double *dst, size_t cnt, const int32_t *err, int32_t *err_dst) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < cnt) dst[q] = src[q];
    if (q == 0) *err_dst = *err;
}

// Every STM_* environment switch the library knows, read ONCE, in stm_create (an inherited environment cannot change kernels under
// a live handle; INTEGRATION.md lists them).  The A/B switches select between data paths that give the same results; the debug
// switches (dumps, cycle counters, poisoned LDS, partial E-steps, the fault injector) exist only in a build with -DSTM_TESTING
// (strutopy_amd/libstm_hip_testing.so, what the tests and tools/ load when they need them) -- there they can also be changed on a
// live handle with stm_debug_set.
struct stm_switches {
    int solver_mode = 0;             // STM_SOLVER_MODE: 0 auto (two waves per document, registers + LDS), 3 one wave, 1 LDS only, 2 global slab only
    int solver_dma = 1;              // STM_SOLVER_DMA: K = 50 / 64 set-up through the LDS-DMA path
    int solver_max_docs_per_cu = 16; // STM_SOLVER_MAX_DOCS_PER_CU
    int solver_k100_direct = 1;      // STM_SOLVER_K100_DIRECT: K > 64 re-gathers beta rows per pass (0: per-document slab)
    int solver_persist = 1;          // STM_SOLVER_PERSIST: the two-wave solver's workgroups take documents off a ticket counter
    int slab_budget_mb = 24576;      // STM_SLAB_BUDGET_MB
    int betass_group_kb = stm::BETASS_GROUP_BYTES >> 10;   // STM_BETASS_GROUP_KB: theta bytes per document group of the beta_ss pass
    int betass_two = 1;              // STM_BETASS_TWO: two theta rows per load in that pass (even K)
    int post_any = 0;                // STM_POST_ANY: every K through post_any_kernel
    int post_big2 = 1;               // STM_POST_BIG2: 64 < K <= 112 through post_big2_kernel
    int post_big2_waves = 2;         // STM_POST_BIG2_WAVES: 2 | 4
    int post_rem = 1;                // STM_POST_REM
    int post_max_wg_per_cu = 0;      // STM_POST_MAX_WG_PER_CU (0: as many as LDS and registers allow)
    int sigma_replicas = 256;        // STM_SIGMA_REPLICAS (atomically filled nu replicas: K > 112)
    int mom_k0 = 2, mom_k1 = 4;      // STM_MOM_K0 / STM_MOM_K1: later searches that start with a moment pass (stm_solver.h)
    // -DSTM_TESTING only
    int debug_prof = 0, debug_dump = 0, debug_stage = 3, debug_flags = 0, post_debug = 0, debug_fail_plan = 0;
};
static stm_switches read_switches() {
    stm_switches w;
    w.solver_mode = env_int("STM_SOLVER_MODE", w.solver_mode);
    w.solver_dma = env_int("STM_SOLVER_DMA", w.solver_dma);
    w.solver_max_docs_per_cu = env_int("STM_SOLVER_MAX_DOCS_PER_CU", w.solver_max_docs_per_cu);
    w.solver_k100_direct = env_int("STM_SOLVER_K100_DIRECT", w.solver_k100_direct);
    w.solver_persist = env_int("STM_SOLVER_PERSIST", w.solver_persist);
    w.slab_budget_mb = env_int("STM_SLAB_BUDGET_MB", w.slab_budget_mb);
    w.betass_group_kb = env_int("STM_BETASS_GROUP_KB", w.betass_group_kb);
    w.betass_two = env_int("STM_BETASS_TWO", w.betass_two);
    w.post_any = env_int("STM_POST_ANY", w.post_any);
    w.post_big2 = env_int("STM_POST_BIG2", w.post_big2);
    w.post_big2_waves = env_int("STM_POST_BIG2_WAVES", w.post_big2_waves);
    w.post_rem = env_int("STM_POST_REM", w.post_rem);
    w.post_max_wg_per_cu = env_int("STM_POST_MAX_WG_PER_CU", w.post_max_wg_per_cu);
    w.sigma_replicas = env_int("STM_SIGMA_REPLICAS", w.sigma_replicas);
    w.mom_k0 = env_int("STM_MOM_K0", w.mom_k0);
    w.mom_k1 = env_int("STM_MOM_K1", w.mom_k1);
#ifdef STM_TESTING
    w.debug_prof = env_int("STM_DEBUG_PROF", 0);
    w.debug_dump = env_int("STM_DEBUG_DUMP", 0);
    w.debug_stage = env_int("STM_DEBUG_STAGE", 3);
    w.debug_flags = env_int("STM_DEBUG_FLAGS", 0);
    w.post_debug = env_int("STM_POST_DEBUG", 0);
    w.debug_fail_plan = env_int("STM_DEBUG_FAIL_PLAN", 0);
#endif
    return w;
}

}  // namespace

struct stm_handle {
    stm_switches sw;
    int device = 0;
    hipStream_t stream = nullptr;
    int cu = 0;
    std::string name;
    size_t hbm = 0;
    // corpus
    int64_t N = 0, nnz = 0;
    int V = 0, A = 1, maxNd = 0;
    std::vector<int64_t> h_indptr;
    std::vector<int32_t> h_len_sorted;   // document lengths in processing order (longest first)
    int nd_max = 1;                      // words of the longest document
    int64_t *d_indptr = nullptr;
    int32_t *d_indices = nullptr, *d_aspect = nullptr, *d_order = nullptr;
    int64_t *d_tick = nullptr;           // per ticket of the longest-first order: {indptr[doc], doc | Nd << 32} (one scalar load instead of two dependent ones)
    double *d_counts = nullptr;
    // the corpus in word-major order (stm_betass.h): entries sorted by (level, word), ascending document within a row
    int32_t *d_wm_doc = nullptr, *d_wm_pos = nullptr, *d_cptr = nullptr;
    double *d_bss_part = nullptr;   // [G][A * V][K] partial sums of the word-major pass
    int G = 1;                      // document groups of the word-major pass
    std::vector<int32_t> h_indices, h_aspect;   // kept for the word-major build (stm_set_topics fixes the chunk size)
    double *d_rw = nullptr;     // [nnz] r_dw written by the post kernel
    size_t sigma_part_len = 0;
    // model
    int K = 0, n = 0;
    double *d_betaT = nullptr, *d_tmpKV = nullptr, *d_colsum = nullptr;
    double *d_beta_ssT = nullptr, *d_sigma_ss = nullptr, *d_scal = nullptr;  // views into d_pack
    double *d_eta = nullptr, *d_mu = nullptr, *d_theta = nullptr, *d_bound = nullptr;
    double *d_siginv = nullptr, *d_sigma_part = nullptr;
    int32_t *d_status = nullptr, *d_nit = nullptr, *d_nfev = nullptr, *d_njev = nullptr, *d_pd = nullptr;
    int *d_counters = nullptr;
    int32_t *d_err = nullptr;
    double *d_slab_beta = nullptr, *d_slab_H = nullptr;
    size_t slab_beta_len = 0;
    int chunk = 0, nrep = 256;   // documents per launch, replicated nu accumulators
    // solver launch plan: runs of the longest-first order with equal LDS occupancy
    struct Group { int64_t first, count; int ld; size_t lds_bytes; bool global; int64_t resident = 0; };   // resident: workgroups the chip holds at once (two-wave form: the persistent launch's grid)
    std::vector<Group> groups;
    int kreg = 0;                // register-resident topic count of the solver instantiation (0: none)
    int nw = 1;                  // wavefronts per document in the solver
    int KP = 0;                  // slab row length (doubles)
    int vpl = 1;                 // vector components per lane in the solver (2 for 64 < K <= 128)
    double *d_red = nullptr;     // first-stage sums of the two-stage reductions + the bound's per-block sums
    size_t red_len = 0;
    bool dma = false;            // two-wave solver with LDS-staged row gather (K == KREG = 50 or 64)
    bool direct = false;         // K > 64: rows re-gathered from betaT per pass instead of a per-document slab
    bool any = false;            // K > 112 (or STM_POST_ANY=1): the general post kernel (stm_post_any.h)
    bool wm = false;             // phi through r_dw + the word-major beta_ss pass and nu in per-workgroup slabs (run-to-run identical): every K <= 128;
                                 // beyond, post_any_kernel adds both with fp64 atomics
    bool big2 = false;           // 64 < K <= 112: the two-waves-per-document post kernel (stm_post_big2.h) + the word-major beta_ss pass
    // optional dumps
    double *d_phi = nullptr;
    int64_t phi_doc = -1;
    double *d_hess = nullptr, *d_chol = nullptr, *d_nu = nullptr;
    long long *d_prof = nullptr;
    // M-step
    int p = 0;
    double *d_X = nullptr, *d_mom = nullptr, *d_gamma = nullptr, *d_cov = nullptr;
    size_t mom_len = 0, cov_len = 0, gamma_len = 0, phi_len = 0;
    // comm
    void *comm = nullptr;
    int rank = 0, nranks = 1;
    double *d_pack = nullptr, *d_extra = nullptr, *d_small = nullptr;
    size_t extra_cap = 0, small_len = 0;
    double *d_ascratch = nullptr; size_t ascratch_len = 0;   // post_any_kernel's per-workgroup A, L, b
    bool exchange_single = false;   // stm_comm_set_exchange: ONE all-reduce of the whole packed buffer per EM iteration instead of two
    size_t pack_len = 0;
    void *spectral = nullptr;   // spectral-initialisation workspace (stm_spectral_api.inc)
    // timing
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // the word-major beta_ss pass (K <= 64) is the one piece of an E-step that nothing needs before the beta update: the fused
    // iteration enqueues it BEHIND the read-back, so that the host's M-step algebra runs while the GPU is still busy with it
    hipEvent_t ev_b[4] = {nullptr, nullptr, nullptr, nullptr}, ev_back = nullptr;   // around the pass (two pairs, used in turn: the last COMPLETED pass is what gets timed); read-back complete
    int bss_pair = 0;
    bool bss_pair_used = false;
    bool bss_deferred = false;
    bool last_deferred = false;   // the last E-step's pass was enqueued behind ev[3] (its time is added to 'estep' by stm_last_kernel_ms)
    float ms_bss = 0;
    float ms[3] = {0, 0, 0};
    bool beta_set = false;
    // pinned staging for the small per-iteration transfers (siginv in; moments, covariance, sigma_ss out):
    // a copy to / from fresh pageable memory makes the runtime pin the caller's pages on the fly (ms)
    void *stage = nullptr;
    static constexpr size_t STAGE_BYTES = 1 << 20;
    // pinned regions of the single-synchronisation EM iteration (stm_em_begin / stm_em_finish), so that nothing enqueued
    // there shares a staging area with a transfer still in flight: [0, 1M) read-back, [1M, 1.5M) siginv, [1.5M, 2M) gamma
    // (grow-only: sized from n and p when first needed -- a one-hot X with hundreds of columns reads back megabytes)
    void *stage_back = nullptr, *stage_sig = nullptr, *stage_gam = nullptr;
    size_t stage_back_cap = 0, stage_sig_cap = 0, stage_gam_cap = 0;
};

void stm_spectral_destroy(void *p);

// grow-only pinned host buffer (the stream is drained before an old one is released)
static int ensure_pinned(stm_handle *h, void **p, size_t *cap, size_t bytes) {
    if (*p && *cap >= bytes) return STM_OK;
    if (*p) { HIP_TRY(hipStreamSynchronize(h->stream)); (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = std::max<size_t>(bytes, 4096);
    HIP_TRY(hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return STM_OK;
}

static int use_device(stm_handle *h) {
    HIP_TRY(hipSetDevice(h->device));
    return STM_OK;
}

// out[nn] = sum of the nblocks copies part[b][nn], fixed order; many copies: in two stages of RED_Y block rows
constexpr int RED_Y = stm::EPI_RED_Y, BOUND_BLOCKS = stm::EPI_BOUND_BLOCKS;
static int reduce_copies(stm_handle *h, const double *part, int nblocks, int nn, double *out) {
    const unsigned gx = (unsigned)((nn + 63) / 64);
    if (nblocks < 4 * RED_Y) {
        hipLaunchKernelGGL(stm::reduce_sigma_kernel, dim3(gx), dim3(256), 0, h->stream, part, nblocks, nn, out, nblocks);
    } else {
        if (int rc = ensure(&h->d_red, &h->red_len, (size_t)RED_Y * nn + BOUND_BLOCKS)) return rc;
        const int chunk = (nblocks + RED_Y - 1) / RED_Y;
        hipLaunchKernelGGL(stm::reduce_sigma_kernel, dim3(gx, RED_Y), dim3(256), 0, h->stream, part, nblocks, nn, h->d_red + BOUND_BLOCKS, chunk);
        hipLaunchKernelGGL(stm::reduce_sigma_kernel, dim3(gx), dim3(256), 0, h->stream, (const double *)(h->d_red + BOUND_BLOCKS), RED_Y, nn, out, RED_Y);
    }
    HIP_TRY(hipGetLastError());
    return STM_OK;
}


using SolverFn = void (*)(stm::SolverParams);

#ifndef POST_WPE
#define POST_WPE 3   // waves per SIMD the K <= 64 post kernel is register-budgeted for (twelve single-wave workgroups per CU)
#endif

// solver instantiations: KREG topics of the register-resident words (0: none), LDS or global slab,
// one or two wavefronts per document
static SolverFn solver_fn(int kreg, bool global_slab, int nw = 1, int vpl = 1, bool direct = false, bool dma = false) {
    if (dma && kreg == 50) return stm::solver_kernel<1, 50, false, 2, 0, true>;   // K == 50 / 64: rows staged through the LDS
    if (dma && kreg == 64) return stm::solver_kernel<1, 64, false, 2, 0, true>;
    if (vpl == 2 && direct) return stm::solver_kernel<2, 0, false, 1, 1>;   // 64 < K <= 128, rows re-gathered per pass
    if (vpl == 2) return global_slab ? stm::solver_kernel<2, 0, true> : stm::solver_kernel<2, 0, false>;   // 64 < K <= 128
    if (vpl == 4) return stm::solver_kernel<4, 0, true>;   // 128 < K <= 256: the general form (slab and BFGS matrix in HBM)
    if (vpl == 8) return stm::solver_kernel<8, 0, true>;   // 256 < K <= 512
    if (global_slab) return stm::solver_kernel<1, 0, true>;
    if (nw == 2) {
        switch (kreg) {
        case 16: return stm::solver_kernel<1, 16, false, 2>;
        case 32: return stm::solver_kernel<1, 32, false, 2>;
        case 50: return stm::solver_kernel<1, 50, false, 2>;
        default: return stm::solver_kernel<1, 64, false, 2>;
        }
    }
    switch (kreg) {
    case 16: return stm::solver_kernel<1, 16, false>;
    case 32: return stm::solver_kernel<1, 32, false>;
    case 50: return stm::solver_kernel<1, 50, false>;
    case 64: return stm::solver_kernel<1, 64, false>;
    default: return stm::solver_kernel<1, 0, false>;
    }
}

// slab row length: K rounded up to 4j+2 doubles (see stm_solver.h)
static int slab_row(int K) { return ((std::max(K, 2) - 2 + 3) / 4) * 4 + 2; }

constexpr size_t LDS_PER_CU = 160 * 1024;
constexpr int64_t GENERAL_DOCS_PER_LAUNCH = 4096;   // documents per launch of the solver's general forms (K > 128)
constexpr int SOLVER_TICKETS = 63;     // ticket counters of the persistent solver launches of one E-step (behind the error flag in d_err)
constexpr size_t LDS_STATIC = 4096;   // static LDS of the solver kernel (se, sv, sw, mailbox, scalar state) when the runtime cannot be asked

// Cut the longest-first document order into launches of equal LDS occupancy.
static int plan_solver(stm_handle *h) {
    const int K = h->K;
    // STM_SOLVER_MODE: 0 auto (two waves per document, registers + LDS), 3 one wave (registers + LDS),
    // 1 one wave, LDS only, 2 one wave, global slab only (v1 data path)
    const int mode = h->sw.solver_mode;
    h->kreg = 0;
    h->nw = 1;
    if (mode == 0 || mode == 3) h->kreg = K <= 16 ? 16 : K <= 32 ? 32 : K <= 50 ? 50 : 64;
    if (mode == 0) h->nw = 2;
    h->vpl = K > 256 ? 8 : K > 128 ? 4 : K > 64 ? 2 : 1;
    if (h->vpl >= 2) { h->kreg = 0; h->nw = 1; }   // two (four, eight) vector components per lane: beta_d in LDS / HBM only
    // rows through the LDS-DMA path: needs K == KREG (packed rows of K doubles are the slab's rows) and 32-bit row offsets
    h->dma = h->nw == 2 && K == h->kreg && (K == 50 || K == 64) && slab_row(K) == 2 * ((K / 2) | 1) &&
             ((size_t)h->A * h->V * K + 64) * sizeof(double) < ((size_t)1 << 32) && h->sw.solver_dma != 0;
    const int vreg = h->kreg > 0 ? 64 * h->nw : 0;
    const int cmax = std::max(1, h->sw.solver_max_docs_per_cu);
    const int KP = slab_row(h->kreg > 0 ? std::max(h->kreg, K) : K);
    h->KP = KP;
    const size_t h_lds = h->nw == 2 ? ((size_t)h->n * h->n + (h->dma ? (size_t)stm::solver_dma_stage_extra(h->kreg, h->n) : 0)) * sizeof(double) : 0;  // BFGS matrix in LDS (two-wave form) + what the DMA staging needs beyond it
    // K > 64 (STM_SOLVER_K100 = "direct" unless set to 0): no copy of beta_d, one 16-word LDS tile re-gathered from betaT per pass
    h->direct = h->vpl == 2 && mode == 0 && h->sw.solver_k100_direct != 0;
    auto lds_of = [&](int nd) {
        if (h->direct) return (size_t)16 * KP * sizeof(double) + (size_t)nd * (2 * sizeof(double) + sizeof(int32_t)) + 16;
        return (size_t)(KP + 2) * (size_t)std::max(0, nd - vreg) * sizeof(double) + h_lds;
    };
    // the kernel's own static LDS (what a workgroup needs is static + dynamic: a constant here went stale when the scalar
    // line-search state moved into LDS, and a document just below the limit then failed in hipFuncSetAttribute)
    size_t lds_static = LDS_STATIC;
    {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, (const void *)solver_fn(h->kreg, false, h->nw, h->vpl, h->direct, h->dma)) == hipSuccess) lds_static = fa.sharedSizeBytes;
        else (void)hipGetLastError();
    }
    auto per_cu = [&](int nd) -> int {
        const size_t b = ((lds_of(nd) + lds_static + 511) / 512) * 512;
        if (mode == 2 || b > LDS_PER_CU || h->vpl > 2) return 0;   // (K > 128: the general form, slab in HBM)
        if (h->direct) return (int)std::min<size_t>((size_t)cmax, LDS_PER_CU / b);
        // K > 64: one wave per document and nothing to hide its latencies but other documents -- below four
        // documents per CU the HBM slab (occupancy bound by registers only) wins (C4: 93 / 25 / 43 -> 54 / 24 / 27 ms)
        if (h->vpl == 2 && mode == 0 && LDS_PER_CU / b < 4) return 0;
        return (int)std::min<size_t>((size_t)cmax, LDS_PER_CU / b);
    };
    h->groups.clear();
    size_t max_dyn = 0, glob_len = 0;
    const int64_t N = h->N;
    const size_t budget = (size_t)h->sw.slab_budget_mb << 20;
    for (int64_t i = 0; i < N;) {
        const int nd = h->h_len_sorted[(size_t)i];
        const int c = per_cu(nd);
        int64_t j = i + 1;
        while (j < N && per_cu(h->h_len_sorted[(size_t)j]) == c) ++j;
        stm_handle::Group g{i, j - i, 0, 0, c == 0};
        if (c == 0) {
            g.ld = (nd + 63) / 64 * 64;
            const size_t per_doc = (size_t)(KP + 2) * g.ld;
            size_t docs = std::min<size_t>((size_t)(j - i), std::max<size_t>(1, budget / (per_doc * sizeof(double))));
            if (h->vpl > 2) docs = std::min<size_t>(docs, GENERAL_DOCS_PER_LAUNCH);   // (K > 128: see stm_set_topics)
            glob_len = std::max(glob_len, docs * per_doc);
        } else {
            g.ld = std::max(0, nd - vreg);
            g.lds_bytes = lds_of(nd);
            max_dyn = std::max(max_dyn, g.lds_bytes);
        }
        h->groups.push_back(g);
        i = j;
    }
    if (max_dyn > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)solver_fn(h->kreg, false, h->nw, h->vpl, h->direct, h->dma),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_dyn);
        if (e != hipSuccess) (void)hipGetLastError();
        if (e != hipSuccess) return fail(STM_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    }
    if (h->nw == 2)   // persistent launches: what the chip keeps resident of each group's workgroups (registers and this group's LDS)
        for (auto &gr : h->groups) {
            if (gr.global) continue;
            int per = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void *)solver_fn(h->kreg, false, h->nw, h->vpl, h->direct, h->dma), 64 * h->nw, gr.lds_bytes) != hipSuccess) { (void)hipGetLastError(); per = 0; }
            gr.resident = (int64_t)std::max(per, 0) * std::max(h->cu, 1);
        }
    h->slab_beta_len = glob_len;
    dfree(h->d_slab_beta);
    if (glob_len) if (int rc = dalloc(&h->d_slab_beta, glob_len)) return rc;
    return STM_OK;
}

// The corpus in word-major order for stm_betass.h: entries sorted by (level * V + word, document group), documents ascending.
static int build_word_major(stm_handle *h) {
    const int64_t N = h->N, nnz = h->nnz;
    const size_t R = (size_t)h->A * (size_t)h->V;
    const int64_t gdocs = std::max<int64_t>(1, (int64_t)h->sw.betass_group_kb * 1024 / ((int64_t)h->K * 8));
    int64_t G = std::max<int64_t>(1, (N + gdocs - 1) / gdocs);
    G = std::min<int64_t>(G, 64);
    G = std::min<int64_t>(G, std::max<int64_t>(1, ((int64_t)256 << 20) / (int64_t)std::max<size_t>(R * (size_t)h->K, 1)));   // <= 2 GB of partial sums
    const int64_t gd = (N + G - 1) / G;   // documents per group
    h->G = (int)G;
    const int64_t *indptr = h->h_indptr.data();
    const int32_t *indices = h->h_indices.data();
    const bool asp = !h->h_aspect.empty();
    std::vector<int32_t> cptr(R * (size_t)G + 1, 0);
    for (int64_t d = 0; d < N; ++d) {
        const size_t base = (asp ? (size_t)h->h_aspect[(size_t)d] : 0) * (size_t)h->V, g = (size_t)(d / gd);
        for (int64_t q = indptr[d]; q < indptr[d + 1]; ++q) ++cptr[(base + (size_t)indices[q]) * (size_t)G + g + 1];
    }
    for (size_t r = 0; r + 1 < cptr.size(); ++r) cptr[r + 1] += cptr[r];
    std::vector<int32_t> wm_doc((size_t)nnz), wm_slot((size_t)nnz), fill(cptr.begin(), cptr.end() - 1);
    for (int64_t d = 0; d < N; ++d) {
        const size_t base = (asp ? (size_t)h->h_aspect[(size_t)d] : 0) * (size_t)h->V, g = (size_t)(d / gd);
        for (int64_t q = indptr[d]; q < indptr[d + 1]; ++q) {
            const int32_t slot = fill[(base + (size_t)indices[q]) * (size_t)G + g]++;
            wm_doc[(size_t)slot] = (int32_t)d;
            wm_slot[(size_t)q] = slot;     // CSR position -> word-major slot (the post kernel scatters r there)
        }
    }
    if (int rc = dalloc(&h->d_wm_doc, (size_t)nnz)) return rc;
    if (int rc = dalloc(&h->d_wm_pos, (size_t)nnz)) return rc;
    if (int rc = dalloc(&h->d_cptr, cptr.size())) return rc;
    if (int rc = dalloc(&h->d_bss_part, (size_t)G * R * (size_t)h->K)) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_cptr, cptr.data(), sizeof(int32_t) * cptr.size(), hipMemcpyHostToDevice, h->stream));
    if (nnz) {
        HIP_TRY(hipMemcpyAsync(h->d_wm_doc, wm_doc.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_wm_pos, wm_slot.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));   // the host vectors go out of scope
    return STM_OK;
}

extern "C" {

const char *stm_last_error(void) { return g_err.c_str(); }

int stm_device_count(int *count) {
    if (!count) return fail(STM_ERR_INVALID, "stm_device_count: count is NULL");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    *count = (e == hipSuccess && cnt > 0) ? cnt : 0;
    if (*count == 0) return fail(STM_ERR_NO_DEVICE, "no HIP device available (the E-step has no CPU fallback)");
    return STM_OK;
}

int stm_create(stm_handle **out, int device_ordinal) {
    if (!out) return fail(STM_ERR_INVALID, "stm_create: out is NULL");
    *out = nullptr;
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(STM_ERR_NO_DEVICE, "no HIP device available (the E-step has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= cnt) return fail(STM_ERR_INVALID, "bad device ordinal");
    stm_handle *h = new stm_handle();
    h->sw = read_switches();
    h->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess) { delete h; return fail(STM_ERR_HIP, "hipSetDevice failed"); }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device_ordinal) != hipSuccess) { delete h; return fail(STM_ERR_HIP, "hipGetDeviceProperties failed"); }
    h->cu = pr.multiProcessorCount;
    h->name = std::string(pr.gcnArchName) + " " + pr.name;
    h->hbm = pr.totalGlobalMem;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(STM_ERR_HIP, "hipStreamCreate failed"); }
    for (auto &ev : h->ev)
        if (hipEventCreate(&ev) != hipSuccess) { delete h; return fail(STM_ERR_HIP, "hipEventCreate failed"); }
    for (auto &ev : h->ev_b)
        if (hipEventCreate(&ev) != hipSuccess) { delete h; return fail(STM_ERR_HIP, "hipEventCreate failed"); }
    if (hipEventCreate(&h->ev_back) != hipSuccess) { delete h; return fail(STM_ERR_HIP, "hipEventCreate failed"); }
    if (hipHostMalloc(&h->stage, stm_handle::STAGE_BYTES, hipHostMallocDefault) != hipSuccess) h->stage = nullptr;
    *out = h;
    return STM_OK;
}

void stm_destroy(stm_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    stm_mstep_comm_destroy(h->comm);
    stm_spectral_destroy(h->spectral);
    dfree(h->d_indptr); dfree(h->d_indices); dfree(h->d_aspect); dfree(h->d_order); dfree(h->d_tick); dfree(h->d_counts);
    dfree(h->d_wm_doc); dfree(h->d_wm_pos); dfree(h->d_rw); dfree(h->d_cptr); dfree(h->d_bss_part);
    dfree(h->d_red); dfree(h->d_betaT); dfree(h->d_tmpKV); dfree(h->d_colsum); dfree(h->d_eta); dfree(h->d_mu);
    dfree(h->d_theta); dfree(h->d_bound); dfree(h->d_siginv); dfree(h->d_sigma_part);
    dfree(h->d_status); dfree(h->d_nit); dfree(h->d_nfev); dfree(h->d_njev); dfree(h->d_pd);
    dfree(h->d_counters); dfree(h->d_err); dfree(h->d_slab_beta); dfree(h->d_slab_H); dfree(h->d_phi);
    dfree(h->d_hess); dfree(h->d_chol); dfree(h->d_nu); dfree(h->d_prof);
    dfree(h->d_X); dfree(h->d_mom); dfree(h->d_gamma); dfree(h->d_cov); dfree(h->d_pack); dfree(h->d_ascratch); dfree(h->d_small);
    if (h->stage) (void)hipHostFree(h->stage);
    if (h->stage_back) (void)hipHostFree(h->stage_back);
    if (h->stage_sig) (void)hipHostFree(h->stage_sig);
    if (h->stage_gam) (void)hipHostFree(h->stage_gam);
    for (auto &ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : h->ev_b) if (ev) (void)hipEventDestroy(ev);
    if (h->ev_back) (void)hipEventDestroy(h->ev_back);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int stm_device_info(stm_handle *h, char *name_out, int name_len, int *cu_count, int64_t *hbm_bytes) {
    if (!h) return fail(STM_ERR_INVALID, "null handle");
    if (name_out && name_len > 0) {
        strncpy(name_out, h->name.c_str(), (size_t)name_len - 1);
        name_out[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = h->cu;
    if (hbm_bytes) *hbm_bytes = (int64_t)h->hbm;
    return STM_OK;
}

int stm_set_corpus(stm_handle *h, int64_t N, int32_t V, const int64_t *indptr, const int32_t *indices,
                   const double *counts, const int32_t *aspect, int32_t A) {
    if (!h || N < 0 || V < 1 || !indptr) return fail(STM_ERR_INVALID, "stm_set_corpus: bad arguments");
    if (N >= (int64_t)1 << 31) return fail(STM_ERR_INVALID, "stm_set_corpus: N must be < 2^31 per GPU shard");
    if (A < 1) A = 1;
    if (A > 1 && !aspect) return fail(STM_ERR_INVALID, "stm_set_corpus: A > 1 needs aspect[]");
    if (int rc = use_device(h)) return rc;
    const int64_t nnz = indptr[N] - indptr[0];
    if (indptr[0] != 0 || nnz < 0) return fail(STM_ERR_INVALID, "stm_set_corpus: indptr must start at 0 and be monotone");
    if (nnz >= (int64_t)1 << 31) return fail(STM_ERR_INVALID, "stm_set_corpus: nnz must be < 2^31 per GPU shard (32-bit word-major slots)");
    if (nnz > 0 && (!indices || !counts)) return fail(STM_ERR_INVALID, "stm_set_corpus: indices/counts are NULL");
    int maxNd = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t nd = indptr[i + 1] - indptr[i];
        if (nd < 1) return fail(STM_ERR_INVALID, "stm_set_corpus: empty document (the reference indexes doc_array[:, 0], stm.py:523)");
        maxNd = std::max<int64_t>(maxNd, nd);
    }
    for (int64_t q = 0; q < nnz; ++q)
        if (indices[q] < 0 || indices[q] >= V) return fail(STM_ERR_INVALID, "stm_set_corpus: word id out of range");
    {   // A word id may appear once per document (gensim's doc2bow, what the reference is fed, guarantees it).  The reference
        // itself would add a repeated id's phi column once (beta_ss[:, idx] += phi with a repeated idx keeps the last write,
        // stm.py:588) while counting it twice everywhere else; the kernels let no two lanes share a word's cells.  Rejected.
        std::vector<int32_t> seen((size_t)V, -1);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t q = indptr[i]; q < indptr[i + 1]; ++q) {
                if (seen[(size_t)indices[q]] == (int32_t)i)
                    return fail(STM_ERR_INVALID, "stm_set_corpus: a document holds the same word id twice (merge the counts first)");
                seen[(size_t)indices[q]] = (int32_t)i;
            }
    }
    if (aspect)
        for (int64_t i = 0; i < N; ++i)
            if (aspect[i] < 0 || aspect[i] >= A) return fail(STM_ERR_INVALID, "stm_set_corpus: aspect out of range");
    h->N = N; h->V = V; h->A = A; h->nnz = nnz; h->maxNd = maxNd;
    h->h_indptr.assign(indptr, indptr + N + 1);
    if (int rc = dalloc(&h->d_indptr, (size_t)N + 1)) return rc;
    if (int rc = dalloc(&h->d_indices, (size_t)nnz)) return rc;
    if (int rc = dalloc(&h->d_counts, (size_t)nnz)) return rc;
    if (int rc = dalloc(&h->d_order, (size_t)N)) return rc;
    if (int rc = dalloc(&h->d_tick, 2 * (size_t)N)) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_indptr, indptr, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice, h->stream));
    if (nnz) {
        HIP_TRY(hipMemcpyAsync(h->d_indices, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_counts, counts, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
    }
    dfree(h->d_aspect);
    if (aspect && A > 1) {
        if (int rc = dalloc(&h->d_aspect, (size_t)N)) return rc;
        HIP_TRY(hipMemcpyAsync(h->d_aspect, aspect, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
    }
    // longest documents first: the work queue then ends on short ones (smaller tail)
    std::vector<int32_t> order((size_t)N);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return (indptr[a + 1] - indptr[a]) > (indptr[b + 1] - indptr[b]);
    });
    if (N) HIP_TRY(hipMemcpyAsync(h->d_order, order.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
    {
        std::vector<int64_t> tick(2 * (size_t)N);
        for (int64_t i = 0; i < N; ++i) {
            const int64_t d = order[(size_t)i];
            tick[2 * (size_t)i] = indptr[d];
            tick[2 * (size_t)i + 1] = d | ((indptr[d + 1] - indptr[d]) << 32);
        }
        if (N) HIP_TRY(hipMemcpyAsync(h->d_tick, tick.data(), sizeof(int64_t) * 2 * (size_t)N, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));   // (tick is a local)
    }
    h->h_len_sorted.resize((size_t)N);
    for (int64_t i = 0; i < N; ++i) h->h_len_sorted[(size_t)i] = (int32_t)(indptr[order[i] + 1] - indptr[order[i]]);
    h->nd_max = 1;
    for (int64_t i = 0; i < N; ++i) h->nd_max = std::max(h->nd_max, (int)h->h_len_sorted[(size_t)i]);
    // word-major order (stm_betass.h): a counting sort of the CSR positions by (level, word, document chunk); documents
    // ascend within a row.  The chunk size is fixed when K is known (stm_set_topics); until then the entries are kept
    // on the host.
    h->h_indices.assign(indices, indices + nnz);
    if (aspect && A > 1) h->h_aspect.assign(aspect, aspect + N); else h->h_aspect.clear();
    if (int rc = dalloc(&h->d_rw, (size_t)nnz)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->K = 0;
    return STM_OK;
}

int stm_set_topics(stm_handle *h, int32_t K) {
    if (!h || h->V == 0) return fail(STM_ERR_INVALID, "stm_set_topics: set the corpus first");
    if (K < 2) return fail(STM_ERR_INVALID, "stm_set_topics: K must be >= 2");
    if (K > stm::K_LIMIT) return fail(STM_ERR_INVALID, "stm_set_topics: K > " + std::to_string(stm::K_LIMIT) + " is not supported by this build");
    if (int rc = use_device(h)) return rc;
    if ((size_t)K * h->V * sizeof(double) >= ((size_t)1 << 32)) return fail(STM_ERR_INVALID, "stm_set_topics: one level of beta must stay below 4 GiB (32-bit row offsets)");
    h->K = K; h->n = K - 1;
    const size_t N = (size_t)h->N, n = (size_t)h->n, KV = (size_t)h->A * K * h->V;
    // + 160: the kernels read whole 16-byte pieces / KREG <= 64 doubles from a row start, masked beyond K, and what they mask must be
    // finite; the first K <= 128 of them are the ZERO ROW (index A V) that tile fetches of words a document does not have are pointed at
    if (int rc = dalloc(&h->d_betaT, KV + 160)) return rc;
    HIP_TRY(hipMemsetAsync(h->d_betaT + KV, 0, sizeof(double) * 160, h->stream));
    // one packed buffer [ scalars(8) | sigma_ss | moments | beta_ss ] so a single all-reduce covers it
    h->extra_cap = round64(moments_len(8, h->n));
    h->pack_len = 8 + n * n + h->extra_cap + KV;
    if (int rc = dalloc(&h->d_pack, h->pack_len)) return rc;
    HIP_TRY(hipMemsetAsync(h->d_pack, 0, sizeof(double) * h->pack_len, h->stream));
    h->d_scal = h->d_pack;
    h->d_sigma_ss = h->d_pack + 8;
    h->d_extra = h->d_pack + 8 + n * n;
    h->d_beta_ssT = h->d_extra + h->extra_cap;
    h->p = 0;
    dfree(h->d_X);
    if (int rc = dalloc(&h->d_tmpKV, KV)) return rc;
    if (int rc = dalloc(&h->d_colsum, (size_t)h->A * h->V)) return rc;
    if (int rc = dalloc(&h->d_eta, N * n)) return rc;
    if (int rc = dalloc(&h->d_mu, N * n)) return rc;
    if (int rc = dalloc(&h->d_theta, N * K)) return rc;
    if (int rc = dalloc(&h->d_bound, N)) return rc;
    if (int rc = dalloc(&h->d_siginv, n * n)) return rc;
    if (int rc = dalloc(&h->d_status, N)) return rc;
    if (int rc = dalloc(&h->d_nit, N)) return rc;
    if (int rc = dalloc(&h->d_nfev, N)) return rc;
    if (int rc = dalloc(&h->d_njev, N)) return rc;
    if (int rc = dalloc(&h->d_pd, N)) return rc;
    if (int rc = dalloc(&h->d_counters, 8)) return rc;
    if (int rc = dalloc(&h->d_err, 1 + SOLVER_TICKETS)) return rc;   // [0] the error flag, [1..] the ticket counters of the persistent solver launches
    h->big2 = h->sw.post_any == 0 && stm::post2_serves(K) && h->sw.post_big2 != 0;
    h->any = (K > stm::PT && !h->big2) || h->sw.post_any != 0;   // (112 < K <= 128 ran a one-wave matrix-core kernel of its own until round 5: occupancy 1, scratch, atomics)
    h->wm = K <= 128;
    if (h->wm) if (int rc = build_word_major(h)) return rc;   // stm_betass.h
    // one block (one wave) per document.  The solver keeps beta_d on chip (64 words in registers,
    // the rest in LDS); launches are cut so every launch has one LDS size / occupancy class.
    if (int rc = plan_solver(h)) return rc;
    const size_t budget = (size_t)h->sw.slab_budget_mb << 20;
    h->chunk = (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(h->N, 1), (int64_t)(budget / std::max<size_t>(n * n * sizeof(double), 8))));
    // K > 128 (the general forms): every document of a launch holds its BFGS matrix (2 MB at K = 512) and its copy of beta_d in HBM -- a
    // bounded number of workgroups per launch (sixteen per CU) instead of whatever the 24 GB budget allows; more launches, not more memory
    if (h->vpl > 2) h->chunk = (int)std::min<int64_t>(h->chunk, GENERAL_DOCS_PER_LAUNCH);
    h->nrep = h->sw.sigma_replicas;
    {   // documents that keep their BFGS matrix in the global slab: all of them for the one-wave forms,
        // only the too-long-for-LDS groups for the two-wave form
        int64_t need = 0;
        for (const auto &gr : h->groups)
            if (h->nw == 1 || gr.global) need = std::max<int64_t>(need, std::min<int64_t>(gr.count, h->chunk));
        if (int rc = dalloc(&h->d_slab_H, (size_t)std::max<int64_t>(need, 1) * n * n)) return rc;
    }
    h->sigma_part_len = 0;
    if (int rc = ensure(&h->d_sigma_part, &h->sigma_part_len, (size_t)h->nrep * n * n)) return rc;
    HIP_TRY(hipMemsetAsync(h->d_eta, 0, sizeof(double) * std::max<size_t>(N * n, 1), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_mu, 0, sizeof(double) * std::max<size_t>(N * n, 1), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_theta, 0, sizeof(double) * std::max<size_t>(N * K, 1), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->beta_set = false;
    dfree(h->d_hess); dfree(h->d_chol); dfree(h->d_nu);
    dfree(h->d_prof);
    if (h->sw.debug_prof) {
        if (int rc = dalloc(&h->d_prof, N * stm::PROF_SLOTS)) return rc;
        HIP_TRY(hipMemset(h->d_prof, 0, sizeof(long long) * N * stm::PROF_SLOTS));
    }
    if (h->sw.debug_dump) {
        if (int rc = dalloc(&h->d_hess, N * n * n)) return rc;
        if (int rc = dalloc(&h->d_chol, N * n * n)) return rc;
        if (int rc = dalloc(&h->d_nu, N * n * n)) return rc;
    }
    return STM_OK;
}

// after every change of betaT: the per-word column sums the solver divides the counts by
static int refresh_colsum(stm_handle *h) {
    const int64_t AV = (int64_t)h->A * h->V;
    hipLaunchKernelGGL(beta_colsum_kernel, dim3((unsigned)((AV + 255) / 256)), dim3(256), 0, h->stream, (const double *)h->d_betaT, AV, h->K, h->d_colsum);
    HIP_TRY(hipGetLastError());
    return STM_OK;
}

static int transpose3(stm_handle *h, const double *in, double *out, int R, int C) {
    dim3 blk(32, 8), grd((C + 31) / 32, (R + 31) / 32, h->A);
    hipLaunchKernelGGL(transpose_kernel, grd, blk, 0, h->stream, in, out, R, C);
    HIP_TRY(hipGetLastError());
    return STM_OK;
}

#define NEED_MODEL(h)                                                                   \
    if (!(h) || (h)->K == 0) return fail(STM_ERR_INVALID, "call stm_set_corpus and stm_set_topics first"); \
    if (int rc_ = use_device(h)) return rc_;

int stm_put_beta(stm_handle *h, const double *beta) {
    NEED_MODEL(h);
    if (!beta) return fail(STM_ERR_INVALID, "beta is NULL");
    const size_t KV = (size_t)h->A * h->K * h->V;
    HIP_TRY(hipMemcpyAsync(h->d_tmpKV, beta, sizeof(double) * KV, hipMemcpyHostToDevice, h->stream));
    if (int rc = transpose3(h, h->d_tmpKV, h->d_betaT, h->K, h->V)) return rc;
    if (int rc = refresh_colsum(h)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->beta_set = true;
    return STM_OK;
}
int stm_get_beta(stm_handle *h, double *beta) {
    NEED_MODEL(h);
    const size_t KV = (size_t)h->A * h->K * h->V;
    if (int rc = transpose3(h, h->d_betaT, h->d_tmpKV, h->V, h->K)) return rc;
    HIP_TRY(hipMemcpyAsync(beta, h->d_tmpKV, sizeof(double) * KV, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}
int stm_get_beta_ss(stm_handle *h, double *beta_ss) {
    NEED_MODEL(h);
    const size_t KV = (size_t)h->A * h->K * h->V;
    if (int rc = transpose3(h, h->d_beta_ssT, h->d_tmpKV, h->V, h->K)) return rc;
    HIP_TRY(hipMemcpyAsync(beta_ss, h->d_tmpKV, sizeof(double) * KV, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}

static int put_vec(stm_handle *h, double *dst, const double *src, size_t cnt) {
    if (!src) return fail(STM_ERR_INVALID, "source pointer is NULL");
    const size_t bytes = sizeof(double) * cnt;
    if (cnt && h->stage && bytes <= stm_handle::STAGE_BYTES) {
        memcpy(h->stage, src, bytes);
        HIP_TRY(hipMemcpyAsync(dst, h->stage, bytes, hipMemcpyHostToDevice, h->stream));
    } else if (cnt) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}
static int get_vec(stm_handle *h, double *dst, const double *src, size_t cnt) {
    if (!dst) return fail(STM_ERR_INVALID, "destination pointer is NULL");
    const size_t bytes = sizeof(double) * cnt;
    if (cnt && h->stage && bytes <= stm_handle::STAGE_BYTES) {
        // the GPU writes into the (device-mapped) pinned staging buffer itself
        hipLaunchKernelGGL(copy_out_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, src, (double *)h->stage, cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(dst, h->stage, bytes);
        return STM_OK;
    }
    if (cnt) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}
int stm_put_eta(stm_handle *h, const double *eta) { NEED_MODEL(h); return put_vec(h, h->d_eta, eta, (size_t)h->N * h->n); }
int stm_put_mu(stm_handle *h, const double *mu) { NEED_MODEL(h); return put_vec(h, h->d_mu, mu, (size_t)h->N * h->n); }
int stm_get_eta(stm_handle *h, double *eta) { NEED_MODEL(h); return get_vec(h, eta, h->d_eta, (size_t)h->N * h->n); }
int stm_get_mu(stm_handle *h, double *mu) { NEED_MODEL(h); return get_vec(h, mu, h->d_mu, (size_t)h->N * h->n); }
int stm_get_theta(stm_handle *h, double *theta) { NEED_MODEL(h); return get_vec(h, theta, h->d_theta, (size_t)h->N * h->K); }
int stm_get_sigma_ss(stm_handle *h, double *s) { NEED_MODEL(h); return get_vec(h, s, h->d_sigma_ss, (size_t)h->n * h->n); }
int stm_put_sigma_ss(stm_handle *h, const double *s) { NEED_MODEL(h); return put_vec(h, h->d_sigma_ss, s, (size_t)h->n * h->n); }
int stm_put_beta_ss(stm_handle *h, const double *beta_ss) {
    NEED_MODEL(h);
    if (!beta_ss) return fail(STM_ERR_INVALID, "beta_ss is NULL");
    const size_t KV = (size_t)h->A * h->K * h->V;
    HIP_TRY(hipMemcpyAsync(h->d_tmpKV, beta_ss, sizeof(double) * KV, hipMemcpyHostToDevice, h->stream));
    if (int rc = transpose3(h, h->d_tmpKV, h->d_beta_ssT, h->K, h->V)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}
int stm_get_bound_docs(stm_handle *h, double *b) { NEED_MODEL(h); return get_vec(h, b, h->d_bound, (size_t)h->N); }

int stm_get_diagnostics(stm_handle *h, int32_t *status, int32_t *nit, int32_t *nfev, int32_t *njev, int32_t *pd_path) {
    NEED_MODEL(h);
    const size_t bytes = sizeof(int32_t) * (size_t)h->N;
    if (status) HIP_TRY(hipMemcpyAsync(status, h->d_status, bytes, hipMemcpyDeviceToHost, h->stream));
    if (nit) HIP_TRY(hipMemcpyAsync(nit, h->d_nit, bytes, hipMemcpyDeviceToHost, h->stream));
    if (nfev) HIP_TRY(hipMemcpyAsync(nfev, h->d_nfev, bytes, hipMemcpyDeviceToHost, h->stream));
    if (njev) HIP_TRY(hipMemcpyAsync(njev, h->d_njev, bytes, hipMemcpyDeviceToHost, h->stream));
    if (pd_path) HIP_TRY(hipMemcpyAsync(pd_path, h->d_pd, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}

}  // extern "C"

// Everything of one E-step enqueued on the handle's stream, no wait.  em_stage: siginv goes through the EM iteration's own
// pinned region (the caller guarantees the previous use of it has completed).
// beta_ss from the r_dw the post kernel left behind (stm_betass.h), timed by its own pair of events
static int bss_enqueue(stm_handle *h) {
    h->bss_deferred = false;
    stm::BetaSsParams bp{};
    const int K = h->K;
    bp.K = K; bp.G = h->G; bp.R = (int64_t)h->A * h->V; bp.cptr = h->d_cptr; bp.wm_doc = h->d_wm_doc; bp.rw = h->d_rw;
    bp.theta = h->d_theta; bp.betaT = h->d_betaT; bp.part = h->d_bss_part; bp.beta_ssT = h->d_beta_ssT;
    const int64_t wpg = (bp.R + stm::BETASS_ROWS - 1) / stm::BETASS_ROWS, bpg = (wpg + 3) / 4;
    h->bss_pair ^= 1; h->bss_pair_used = true;
    HIP_TRY(hipEventRecord(h->ev_b[2 * h->bss_pair], h->stream));
    // even K: two rows per load instruction (needs 16-byte aligned rows and N K 8 < 4 GiB for its 32-bit offsets)
    const bool two = (K % 2 == 0) && (size_t)h->N * K * 8 < ((size_t)1 << 32) && h->sw.betass_two != 0;
    if (two && K <= 64) hipLaunchKernelGGL((stm::beta_ss_part2_kernel<8, stm::BETASS_ROWS, 2>), dim3((unsigned)(bpg * h->G)), dim3(256), 0, h->stream, bp);
    else if (two) hipLaunchKernelGGL((stm::beta_ss_part2_kernel<8, stm::BETASS_ROWS, 1>), dim3((unsigned)(bpg * h->G)), dim3(256), 0, h->stream, bp);
    else if (K <= 64) hipLaunchKernelGGL((stm::beta_ss_part_kernel<8, stm::BETASS_ROWS, 1>), dim3((unsigned)(bpg * h->G)), dim3(256), 0, h->stream, bp);
    else hipLaunchKernelGGL((stm::beta_ss_part_kernel<8, stm::BETASS_ROWS, 2>), dim3((unsigned)(bpg * h->G)), dim3(256), 0, h->stream, bp);
    hipLaunchKernelGGL(stm::beta_ss_reduce_kernel, dim3((unsigned)((bp.R * K + 255) / 256)), dim3(256), 0, h->stream, bp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_b[2 * h->bss_pair + 1], h->stream));
    return STM_OK;
}
// the pass's time, whenever its events have completed (never waits)
static void bss_time(stm_handle *h) {
    for (int q = 0; q < 2; ++q) {   // this E-step's pass if it is through, else the one before
        const int pr = q == 0 ? h->bss_pair : h->bss_pair ^ 1;
        float ms = 0;
        if (hipEventQuery(h->ev_b[2 * pr + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->ev_b[2 * pr], h->ev_b[2 * pr + 1]) == hipSuccess) { h->ms_bss = ms; return; }
        (void)hipGetLastError();
    }
    // (no pass has completed yet -- the first fused iteration: 0 is reported rather than waiting for it here)
}

// What an E-step's launches need that can FAIL on the host -- sizes, device and pinned allocations, function attributes, the
// occupancy query -- decided BEFORE anything is enqueued: a rank of a multi-GPU fit that returned early between its first launch
// and the all-reduce would leave its peers waiting in the collective (stm_em_begin plans first and, on failure, still joins them).
using PostFn = void (*)(stm::PostParams);
struct EstepPlan {
    int dbg_stage = 3, post_debug = 0;
    bool run_post = false, rem_used = false;
    PostFn pfn = nullptr;
    unsigned wg_threads = 64;
    size_t lds = 0, slab = 0;
    int64_t grid = 0;
    int nrep = 0;
};
static int estep_plan(stm_handle *h, bool em_stage, EstepPlan &pl) {
    const int n = h->n, K = h->K;
    const int dbg_stage = h->sw.debug_stage;  // 0: no kernels, 1: solver only, 3: all
#ifdef STM_TESTING
    if (h->sw.debug_fail_plan) return fail(STM_ERR_HIP, "STM_DEBUG_FAIL_PLAN: simulated allocation failure in the E-step's plan");   // (the fault injector of tests/test_gpu_round2.py; not in the product build)
#endif
    if (em_stage) if (int rc = ensure_pinned(h, &h->stage_sig, &h->stage_sig_cap, sizeof(double) * (size_t)n * n)) return rc;
    // last document's phi is what the reference leaves in self.phi (stm.py:1116)
    h->phi_doc = h->N - 1;
    if (h->N > 0) {
        const size_t nd = (size_t)(h->h_indptr[h->N] - h->h_indptr[h->N - 1]);
        if (int rc = ensure(&h->d_phi, &h->phi_len, (size_t)K * nd)) return rc;
    }
    const bool run_post = (dbg_stage & 2) && h->N > 0;
    const int post_debug = h->sw.post_debug;
    PostFn pfn = nullptr;
    unsigned wg_threads = 64;
    size_t lds = 0, slab = (size_t)n * n;
    int64_t grid = 0;
    int nrep = h->nrep;
    bool rem_used = false;   // K <= 64 post kernel instantiated with REM = 1 (decides the layout of its nu slabs)
    if (run_post) {
        // persistent workgroups: as many as the LDS / register budget keeps resident.  The matrix is n x n (n = K - 1):
        // 16 x 16 MFMA blocks, and when n is one past a multiple of 16 (K = 50: 49 = 3 * 16 + 1) the last row / column
        // of b b^T rides on the VALU instead of a padded block (post_kernel)
        const bool rem = n > 16 && n % 16 == 1 && h->sw.post_rem;
        const int nb = rem ? n / 16 : (n + 15) / 16;
        // the DBG = true instantiations (per-document dumps, cycle counters, LDS poisoning) exist in the -DSTM_TESTING build only
#ifdef STM_TESTING
        const bool dbg = h->d_nu != nullptr || h->d_prof != nullptr || post_debug != 0;
#define STM_DBG_PICK(KERNEL_DBG, KERNEL) (dbg ? (KERNEL_DBG) : (KERNEL))
#else
#define STM_DBG_PICK(KERNEL_DBG, KERNEL) (KERNEL)
#endif
        PostFn pf;
        if (rem) {
            pf = STM_DBG_PICK((nb == 1 ? stm::post_kernel<1, 1, POST_WPE, true> : nb == 2 ? stm::post_kernel<2, 1, POST_WPE, true> : stm::post_kernel<3, 1, POST_WPE, true>),
                              (nb == 1 ? stm::post_kernel<1, 1, POST_WPE, false> : nb == 2 ? stm::post_kernel<2, 1, POST_WPE, false> : stm::post_kernel<3, 1, POST_WPE, false>));
        } else {
            pf = STM_DBG_PICK((nb <= 1 ? stm::post_kernel<1, 0, POST_WPE, true> : nb == 2 ? stm::post_kernel<2, 0, POST_WPE, true>
                               : nb == 3 ? stm::post_kernel<3, 0, POST_WPE, true> : stm::post_kernel<4, 0, 2, true>),
                              (nb <= 1 ? stm::post_kernel<1, 0, POST_WPE, false> : nb == 2 ? stm::post_kernel<2, 0, POST_WPE, false>
                               : nb == 3 ? stm::post_kernel<3, 0, POST_WPE, false> : stm::post_kernel<4, 0, 2, false>));
        }
        const bool big2 = h->big2;                  // two waves per document (stm_post_big2.h)
        const bool any = h->any;                    // any K: one workgroup per document, everything in HBM scratch (stm_post_any.h)
        rem_used = rem && K <= stm::PT && !any;
        const int nbb = (n + 15) / 16;
        PostFn pf2 = nullptr;
        const int pc2 = stm::post2_pc(K);
        const int nwv2 = h->sw.post_big2_waves == 4 ? 4 : 2;   // waves per document (stm_post_big2.h)
        if (big2) {
#define STM_PB2(NBV, PCV) (nwv2 == 4 ? STM_DBG_PICK((stm::post_big2_kernel<NBV, PCV, true, 4>), (stm::post_big2_kernel<NBV, PCV, false, 4>)) \
                                     : STM_DBG_PICK((stm::post_big2_kernel<NBV, PCV, true, 2>), (stm::post_big2_kernel<NBV, PCV, false, 2>)))
            if (pc2 == 40) pf2 = nbb <= 4 ? STM_PB2(4, 40) : STM_PB2(5, 40);
            else pf2 = nbb <= 5 ? STM_PB2(5, 56) : nbb == 6 ? STM_PB2(6, 56) : STM_PB2(7, 56);
#undef STM_PB2
#undef STM_DBG_PICK
        }
        pfn = any ? (h->wm ? (PostFn)stm::post_any_kernel<true> : (PostFn)stm::post_any_kernel<false>) : big2 ? pf2 : pf;
        wg_threads = any ? (unsigned)stm::ANY_BS : big2 ? 64u * (unsigned)nwv2 : 64u;
        lds = (any ? stm::post_any_lds_doubles(K) : big2 ? (size_t)stm::post2_lds_map(K, pc2).total : (size_t)stm::post_lds_map(K, nb).total) * sizeof(double);
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)pfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)pfn, (int)wg_threads, lds));
        per_cu = std::max(1, std::min(per_cu, h->sw.post_max_wg_per_cu > 0 ? h->sw.post_max_wg_per_cu : (any ? 4 : 16)));
        grid = std::min<int64_t>(h->N, (int64_t)per_cu * std::max(h->cu, 1));
        if (any)    // ... its per-workgroup HBM scratch (A, L, b: 4 MB + N_d K doubles at K = 512) stays below 4 GB
            grid = std::max<int64_t>(1, std::min<int64_t>(grid, (int64_t)(((size_t)4 << 30) / (stm::post_any_scratch(K, h->nd_max) * sizeof(double)))));
        if (any) {  // A, L, b and sqrt(c) of the document a workgroup is on (sized for the longest document)
            if (int rc = ensure(&h->d_ascratch, &h->ascratch_len, (size_t)grid * stm::post_any_scratch(K, h->nd_max))) return rc;
        }
        // nu is summed per workgroup in a slab of its own (plain read-modify-write, nrep = grid) or, beyond 128 topics, atomically into
        // nrep replicas (post_any_kernel<false>); the epilogue adds them in a fixed order
        nrep = h->wm ? (int)grid : h->nrep;
        const int nbc = (n + 15) / 16;
        // accumulator-tile layout; with REM (post_kernel) the last column has a slot of its own instead of a block column of tiles
        slab = any ? (size_t)n * n : (rem_used ? (size_t)((nbc - 1) * nbc / 2 + 1) * 256 : (size_t)(nbc * (nbc + 1) / 2) * 256);
        if (int rc = ensure(&h->d_sigma_part, &h->sigma_part_len, (size_t)nrep * slab + slab)) return rc;   // + one slab: the reduced tiles
    }
    // the first stage of the two-stage reductions + the bound's block sums (reduce_copies grows it otherwise)
    if (int rc = ensure(&h->d_red, &h->red_len, (size_t)RED_Y * std::max(slab, (size_t)n * n) + BOUND_BLOCKS)) return rc;

    pl.dbg_stage = dbg_stage; pl.post_debug = post_debug; pl.run_post = run_post; pl.rem_used = rem_used;
    pl.pfn = pfn; pl.wg_threads = wg_threads; pl.lds = lds; pl.slab = slab; pl.grid = grid; pl.nrep = nrep;
    return STM_OK;
}

static int estep_enqueue(stm_handle *h, const double *siginv, double sigmaentropy, bool em_stage, bool defer_bss = false, const EstepPlan *ready = nullptr,
                         bool with_moments = false) {
    NEED_MODEL(h);
    if (!h->beta_set) return fail(STM_ERR_INVALID, "stm_estep: beta has not been set");
    if (!siginv) return fail(STM_ERR_INVALID, "stm_estep: siginv is NULL");
    const int n = h->n, K = h->K;
    int diag = 1;
    for (int i = 0; i < n && diag; ++i)
        for (int j = 0; j < n; ++j)
            if (i != j && siginv[(size_t)i * n + j] != 0.0) { diag = 0; break; }
    double sig_bound = 0.0;   // max absolute row sum >= largest eigenvalue (siginv is symmetric)
    for (int i = 0; i < n; ++i) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) r += fabs(siginv[(size_t)i * n + j]);
        sig_bound = std::max(sig_bound, r);
    }
    const size_t KV = (size_t)h->A * K * h->V;
    const bool wm = h->wm;    // phi through r_dw + the word-major pass (beyond 128 topics post_any_kernel adds it atomically)
    // ---- (A) the plan (estep_plan: every fallible host-side step), unless the caller made it already
    EstepPlan pl_own;
    if (!ready) { if (int rc = estep_plan(h, em_stage, pl_own)) return rc; }
    const EstepPlan &pl = ready ? *ready : pl_own;
    const int post_debug = pl.post_debug;
    const bool run_post = pl.run_post, rem_used = pl.rem_used;
    const PostFn pfn = pl.pfn;
    const unsigned wg_threads = pl.wg_threads;
    const size_t lds = pl.lds, slab = pl.slab;
    const int64_t grid = pl.grid;
    const int nrep = pl.nrep;
    const int dbg_stage = pl.dbg_stage;

    // ---- (B) the E-step, enqueued on the handle's stream
    // one launch in front of the solver (stm_epilogue.h): siginv out of a pinned staging area, the error flag + ticket counters and
    // the nu slabs zeroed (they are zeroed HERE so that nothing stands between the solver and the post kernel)
    {
        const double *sig_src = nullptr;
        if (em_stage) {
            sig_src = (const double *)h->stage_sig;
        } else if (h->stage && sizeof(double) * (size_t)n * n <= stm_handle::STAGE_BYTES / 2) {   // second half of the staging buffer
            sig_src = (const double *)((char *)h->stage + stm_handle::STAGE_BYTES / 2);
        }
        if (sig_src) memcpy((void *)sig_src, siginv, sizeof(double) * (size_t)n * n);
        else HIP_TRY(hipMemcpyAsync(h->d_siginv, siginv, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, h->stream));
        const size_t nslab = run_post ? (size_t)nrep * slab : 0;
        const size_t nbss = (!wm || h->nnz == 0) ? KV : 0;   // (the word-major pass writes every cell)
        const size_t work = std::max<size_t>((nslab + nbss) / 2, (size_t)n * n);
        const unsigned hb = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (work + 1023) / 1024));
        hipLaunchKernelGGL(stm::estep_head_kernel, dim3(hb), dim3(256), 0, h->stream, sig_src, h->d_siginv, n * n, h->d_err, 1 + SOLVER_TICKETS,
                           h->d_sigma_part, nslab, h->d_beta_ssT, nbss);
        HIP_TRY(hipGetLastError());
    }

    stm::SolverParams sp{};
    sp.N = h->N; sp.K = K; sp.n = n; sp.V = h->V; sp.KP = h->KP; sp.zrow = (int)((int64_t)h->A * h->V);
    sp.indptr = h->d_indptr; sp.indices = h->d_indices; sp.counts = h->d_counts; sp.aspect = h->d_aspect;
    sp.betaT = h->d_betaT; sp.colsum = h->d_colsum; sp.mu = h->d_mu; sp.eta = h->d_eta; sp.siginv = h->d_siginv; sp.siginv_diag = diag; sp.sig_bound = sig_bound;
    sp.slab_beta = h->d_slab_beta; sp.slab_H = h->d_slab_H;
    sp.order = h->d_order; sp.tick = h->d_tick; sp.status = h->d_status; sp.nit = h->d_nit; sp.nfev = h->d_nfev; sp.njev = h->d_njev;
    sp.err_flag = h->d_err;
    sp.debug_flags = h->sw.debug_flags;
    sp.mom_k0 = h->sw.mom_k0; sp.mom_k1 = h->sw.mom_k1;   // later searches that start with a moment pass (stm_solver.h)
    sp.prof = h->d_prof;

    stm::PostParams pp{};
    pp.N = h->N; pp.K = K; pp.n = n; pp.V = h->V;
    pp.indptr = h->d_indptr; pp.indices = h->d_indices; pp.counts = h->d_counts; pp.aspect = h->d_aspect;
    pp.betaT = h->d_betaT; pp.mu = h->d_mu; pp.eta = h->d_eta; pp.siginv = h->d_siginv; pp.siginv_diag = diag;
    pp.sigmaentropy = sigmaentropy; pp.theta = h->d_theta; pp.bound = h->d_bound; pp.beta_ssT = h->d_beta_ssT;
    pp.sigma_part = h->d_sigma_part; pp.nrep = nrep; pp.order = h->d_order; pp.tick = h->d_tick;
    pp.pd_path = h->d_pd; pp.err_flag = h->d_err;
    pp.hess_out = h->d_hess; pp.chol_out = h->d_chol; pp.nu_out = h->d_nu;
    pp.phi_doc = h->phi_doc; pp.phi_out = h->d_phi;
    pp.debug_flags = post_debug;
    pp.prof = h->d_prof;
    pp.lds_doubles = (int)(lds / sizeof(double));
    pp.a_scratch = h->d_ascratch;
    pp.nd_max = h->nd_max;
    pp.first = 0; pp.count = h->N;
    pp.rw = h->d_rw; pp.wm_slot = h->d_wm_pos;

    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    int n_launch = 0;
    const bool solver_persist = h->sw.solver_persist != 0;
    if (dbg_stage & 1)
        for (const auto &gr : h->groups) {
            const SolverFn fn = gr.global ? solver_fn(0, true, 1, h->vpl) : solver_fn(h->kreg, false, h->nw, h->vpl, h->direct, h->dma);
            const unsigned bdim = gr.global ? 64u : 64u * (unsigned)h->nw;
            sp.ld = gr.ld;
            sp.lds_doubles = (int)(gr.lds_bytes / sizeof(double));
            int64_t step = h->chunk;
            if (gr.global)
                step = std::min<int64_t>(step, std::max<int64_t>(1, (int64_t)(h->slab_beta_len / ((size_t)(sp.KP + 2) * (size_t)std::max(gr.ld, 1)))));
            for (int64_t off = 0; off < gr.count; off += step) {
                sp.first = gr.first + off;
                unsigned g = (unsigned)std::min<int64_t>(step, gr.count - off);
                sp.count = g;
                sp.ticket_ctr = nullptr;
                // two-wave form: persistent workgroups, as many as the chip keeps resident, each taking documents off the launch's
                // ticket counter (stm_solver.h; a launch beyond the counters the E-step zeroes runs one workgroup per document)
                if (!gr.global && h->nw == 2 && gr.resident > 0 && n_launch < SOLVER_TICKETS && solver_persist) {
                    sp.ticket_ctr = h->d_err + 1 + n_launch;
                    g = (unsigned)std::min<int64_t>(g, gr.resident);
                }
                ++n_launch;
                hipLaunchKernelGGL(fn, dim3(g), dim3(bdim), gr.lds_bytes, h->stream, sp);
                HIP_TRY(hipGetLastError());
            }
        }
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    h->bss_deferred = false; h->last_deferred = false;
    if (run_post) {
        hipLaunchKernelGGL(pfn, dim3((unsigned)grid), dim3(wg_threads), lds, h->stream, pp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->ev[2], h->stream));
        if (wm && h->nnz > 0) {
            h->last_deferred = defer_bss;
            if (defer_bss) h->bss_deferred = true;     // the caller enqueues it behind its read-back
            else if (int rc = bss_enqueue(h)) return rc;
        }
    } else {
        HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    }
    // what follows the post kernel, in two launches (stm_epilogue.h): the nu slabs summed in their fixed order and laid out as
    // sigma_ss, the bound, and -- for the resident EM iteration -- the regression moments and eta^T eta with their reductions
    {
        stm::EpiParams ep{};
        ep.sig_part = h->d_sigma_part; ep.sig_copies = nrep; ep.sig_nn = (int)slab;
        ep.sig_rows = nrep < 4 * RED_Y ? 1 : RED_Y;
        ep.sig_chunk = ep.sig_rows == 1 ? nrep : (nrep + RED_Y - 1) / RED_Y;
        ep.sig_red = h->d_red + BOUND_BLOCKS;
        ep.n = n; ep.sig_layout = slab == (size_t)n * n ? 0 : (rem_used ? 2 : 1);
        ep.sigma_ss = h->d_sigma_ss;
        ep.bound = h->d_bound; ep.N = h->N; ep.bound_part = h->d_red; ep.scal = h->d_scal; ep.err = h->d_err;
        ep.nb_sig = (int)((slab + 63) / 64) * ep.sig_rows; ep.nb_bound = BOUND_BLOCKS;
        ep.nb_sig2 = (n * n + 255) / 256;
        if (with_moments) {
            const int p = h->d_X ? h->p : 0;
            ep.X = h->d_X; ep.eta = h->d_eta; ep.p = p; ep.Lr = 1 + p + n + p * p + p * n;
            ep.mom_part = h->d_mom; ep.mom_out = h->d_extra;
            ep.cov_part = h->d_cov + (size_t)n * n; ep.cov_out = h->d_extra + ep.Lr;
            ep.cov_g = (n + 63) / 64;
            ep.nb_mom = mom_blocks((size_t)ep.Lr); ep.nb_cov = MOM_BLOCKS * ep.cov_g * ep.cov_g;
            ep.nb_mom2 = (ep.Lr + 15) / 16; ep.nb_cov2 = (n * n + 15) / 16;
        }
        hipLaunchKernelGGL(n <= 64 ? stm::epilogue_a_kernel<true> : stm::epilogue_a_kernel<false>, dim3((unsigned)(ep.nb_cov + ep.nb_mom + ep.nb_sig + ep.nb_bound)), dim3(256), 0, h->stream, ep);
        hipLaunchKernelGGL(stm::epilogue_b_kernel, dim3((unsigned)(ep.nb_cov2 + ep.nb_mom2 + ep.nb_sig2 + 1)), dim3(256), 0, h->stream, ep);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    return STM_OK;
}

// after the stream has been waited for: kernel times and the device error flag of the last E-step
static int estep_check(stm_handle *h, int32_t err) {
    HIP_TRY(hipEventElapsedTime(&h->ms[0], h->ev[0], h->ev[1]));
    HIP_TRY(hipEventElapsedTime(&h->ms[1], h->ev[1], h->ev[2]));   // post kernel; the beta_ss pass is added by stm_last_kernel_ms
    bss_time(h);
    HIP_TRY(hipEventElapsedTime(&h->ms[2], h->ev[0], h->ev[3]));
    if (err == STM_ERR_BETA) return fail(STM_ERR_BETA, "Some entries of beta are negative or nan.");
    if (err == STM_ERR_PHI) return fail(STM_ERR_PHI, "Some values of phi are zero or nan.");
    if (err == STM_ERR_LINALG) return fail(STM_ERR_LINALG, "Cholesky decomposition of the Hessian failed after every fallback");
    if (err) return fail(STM_ERR_INVALID, "device error flag " + std::to_string(err));
    return STM_OK;
}

extern "C" {

int stm_estep(stm_handle *h, const double *siginv, double sigmaentropy, double *bound_total) {
    if (int rc = estep_enqueue(h, siginv, sigmaentropy, false)) return rc;
    double tot = 0.0;
    int32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&tot, h->d_scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&err, h->d_err, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bound_total) *bound_total = tot;
    return estep_check(h, err);
}

int stm_get_phi(stm_handle *h, int64_t doc, double *phi) {
    NEED_MODEL(h);
    if (doc != h->phi_doc || !h->d_phi) return fail(STM_ERR_INVALID, "stm_get_phi: only the last document's phi is kept (stm.py:1116)");
    const size_t nd = (size_t)(h->h_indptr[doc + 1] - h->h_indptr[doc]);
    return get_vec(h, phi, h->d_phi, (size_t)h->K * nd);
}

// Tests and tools: a debug switch on a live handle (name = the environment variable's).  A product build has none.
int stm_debug_set(stm_handle *h, const char *name, int value) {
    if (!h || !name) return fail(STM_ERR_INVALID, "stm_debug_set: null argument");
#ifdef STM_TESTING
    const std::string k(name);
    if (k == "STM_DEBUG_PROF") h->sw.debug_prof = value;            // (both: before stm_set_topics)
    else if (k == "STM_DEBUG_DUMP") h->sw.debug_dump = value;
    else if (k == "STM_DEBUG_STAGE") h->sw.debug_stage = value;
    else if (k == "STM_DEBUG_FLAGS") h->sw.debug_flags = value;
    else if (k == "STM_POST_DEBUG") h->sw.post_debug = value;
    else if (k == "STM_DEBUG_FAIL_PLAN") h->sw.debug_fail_plan = value;
    else return fail(STM_ERR_INVALID, "stm_debug_set: unknown switch " + k);
    return STM_OK;
#else
    (void)value;
    return fail(STM_ERR_INVALID, std::string("stm_debug_set(") + name + "): this library was built without -DSTM_TESTING (debug switches live in libstm_hip_testing.so)");
#endif
}
int stm_is_testing_build(void) {
#ifdef STM_TESTING
    return 1;
#else
    return 0;
#endif
}

int stm_debug_get_prof(stm_handle *h, long long *out) {
    NEED_MODEL(h);
    if (!h->d_prof) return fail(STM_ERR_INVALID, "cycle counters are off (a -DSTM_TESTING build with STM_DEBUG_PROF=1 / stm_debug_set before stm_set_topics)");
    HIP_TRY(hipMemcpy(out, h->d_prof, sizeof(long long) * (size_t)h->N * stm::PROF_SLOTS, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(h->d_prof, 0, sizeof(long long) * (size_t)h->N * stm::PROF_SLOTS));
    return STM_OK;
}

int stm_debug_get_mats(stm_handle *h, double *hess, double *chol, double *nu) {
    NEED_MODEL(h);
    if (!h->d_hess) return fail(STM_ERR_INVALID, "matrix dumps are off (a -DSTM_TESTING build with STM_DEBUG_DUMP=1 / stm_debug_set before stm_set_topics)");
    const size_t cnt = (size_t)h->N * h->n * h->n;
    if (hess) if (int rc = get_vec(h, hess, h->d_hess, cnt)) return rc;
    if (chol) if (int rc = get_vec(h, chol, h->d_chol, cnt)) return rc;
    if (nu) if (int rc = get_vec(h, nu, h->d_nu, cnt)) return rc;
    return STM_OK;
}

int stm_last_kernel_ms(stm_handle *h, float *ms3) {
    if (!h || !ms3) return fail(STM_ERR_INVALID, "null argument");
    // "post" = the post kernel + the beta_ss pass; "estep" = first to last kernel of the E-step.  A deferred pass (fused
    // iteration) runs behind the E-step's last event and may still be in flight: the last COMPLETED pass stands in for it
    // (its time does not vary from one iteration to the next), 0 before any has completed.
    bss_time(h);
    const float pass = h->wm ? h->ms_bss : 0.0f;
    ms3[0] = h->ms[0]; ms3[1] = h->ms[1] + pass; ms3[2] = h->ms[2] + (h->last_deferred ? pass : 0.0f);
    return STM_OK;
}
int stm_last_pass_ms(stm_handle *h, float *ms) {
    if (!h || !ms) return fail(STM_ERR_INVALID, "null argument");
    bss_time(h);
    *ms = h->wm ? h->ms_bss : 0.0f;
    return STM_OK;
}
int stm_synchronize(stm_handle *h) {
    if (!h) return fail(STM_ERR_INVALID, "null handle");
    if (int rc = use_device(h)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return STM_OK;
}

int stm_estep_host(const stm_estep_args *a, int device_ordinal) {
    if (!a) return fail(STM_ERR_INVALID, "null args");
    stm_handle *h = nullptr;
    int rc = stm_create(&h, device_ordinal);
    if (rc) return rc;
    auto done = [&](int code) { std::string keep = g_err; stm_destroy(h); g_err = keep; return code; };
    if ((rc = stm_set_corpus(h, a->N, a->V, a->indptr, a->indices, a->counts, a->aspect, a->A))) return done(rc);
    if ((rc = stm_set_topics(h, a->K))) return done(rc);
    if ((rc = stm_put_beta(h, a->beta))) return done(rc);
    if ((rc = stm_put_mu(h, a->mu))) return done(rc);
    if ((rc = stm_put_eta(h, a->eta))) return done(rc);
    double tot = 0.0;
    if ((rc = stm_estep(h, a->siginv, a->sigmaentropy, &tot))) return done(rc);
    if (a->bound_total) *a->bound_total = tot;
    if ((rc = stm_get_eta(h, a->eta))) return done(rc);
    if (a->theta && (rc = stm_get_theta(h, a->theta))) return done(rc);
    if (a->bound && (rc = stm_get_bound_docs(h, a->bound))) return done(rc);
    if (a->sigma_ss && (rc = stm_get_sigma_ss(h, a->sigma_ss))) return done(rc);
    if (a->beta_ss && (rc = stm_get_beta_ss(h, a->beta_ss))) return done(rc);
    if ((rc = stm_get_diagnostics(h, a->status, a->nit, a->nfev, a->njev, a->pd_path))) return done(rc);
    return done(STM_OK);
}

}  // extern "C"

#include "stm_mstep_api.inc"
#include "stm_spectral_api.inc"
