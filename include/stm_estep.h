/*
 * stm_estep.h -- C-ABI of the MI355X-native STM E-step library (libstm_hip.so).
 *
 * The reference (mkrcke/strutopy) is pure Python and has NO FFI / plugin
 * interface: the boundary this library sits behind is the method surface of
 * its `STM` class (reference src/modules/stm.py:310).  Each entry point below
 * names the reference code it replaces; strutopy_amd/stm.py binds them with
 * ctypes and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; device memory is owned by
 *    the handle, host buffers by the caller (C-contiguous, fp64 unless noted);
 *    the library never keeps a host pointer past the call.
 *  - every function returns 0 on success, a STM_ERR_* code otherwise;
 *    stm_last_error() returns a thread-local message for the last failure.
 *  - one handle = one GPU = one HIP stream; handles share no mutable state,
 *    so one process per GPU (or one thread per handle) is safe.
 *  - matrices use the reference's layouts: beta [A][K][V], eta/mu [N][K-1],
 *    theta [N][K], sigma/siginv/sigma_ss [(K-1)][(K-1)].
 */
#ifndef STM_ESTEP_H
#define STM_ESTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STM_OK 0
#define STM_ERR_INVALID 1     /* bad argument / call order */
#define STM_ERR_BETA 2        /* "Some entries of beta are negative or nan." (stm.py:534) */
#define STM_ERR_LINALG 3      /* Cholesky failed after every fallback (stm.py:1040-1048) */
#define STM_ERR_HIP 4         /* HIP runtime error */
#define STM_ERR_NO_DEVICE 5   /* no usable GPU */
#define STM_ERR_COMM 6        /* RCCL error */
#define STM_ERR_PHI 7         /* "Some values of phi are zero or nan." (stm.py:1117) */

typedef struct stm_handle stm_handle;

/* ---- lifetime --------------------------------------------------------- */
/* number of usable GPUs (0 and STM_ERR_NO_DEVICE when there is none) */
int stm_device_count(int *count);
int stm_create(stm_handle **out, int device_ordinal);
void stm_destroy(stm_handle *h);
const char *stm_last_error(void);
/* "gfx950 ...": name + CU count of the device behind the handle */
int stm_device_info(stm_handle *h, char *name_out, int name_len, int *cu_count, int64_t *hbm_bytes);

/* ---- corpus and model state (replaces STM.__init__ state, stm.py:366-399) */
/* documents as CSR (the packed form of the reference's list-of-(id,count) BoW,
 * stm.py:522-533): indptr[N+1], indices[nnz] (unique within a document,
 * 0 <= id < V), counts[nnz] (integers stored as fp64), aspect[N] (level of the
 * content covariate per document, stm.py:527-528; NULL when A == 1).
 * Limits per handle (one GPU's shard), checked before the handle is touched: N < 2^31, nnz < 2^31 (32-bit word-major
 * slots), no empty document, no word id twice in one document (STM_ERR_INVALID: the reference would count such an id's phi
 * column once in beta_ss and twice everywhere else, stm.py:588; gensim's doc2bow never produces one).  Besides the CSR arrays the handle keeps the corpus in word-major order for the atomics-free
 * beta_ss pass (8 bytes per entry on the device + a host copy of indices[]) -- INTEGRATION.md lists the memory. */
int stm_set_corpus(stm_handle *h, int64_t N, int32_t V, const int64_t *indptr,
                   const int32_t *indices, const double *counts, const int32_t *aspect, int32_t A);
/* allocate K-dependent state; eta = 0, mu = 0 like stm.py:457,467.  2 <= K <= 512 (K <= 64: one topic per
 * lane and the matrix-core post kernel; 64 < K <= 112: two topics per lane in the solver, two wavefronts per document
 * in the post step; 112 < K <= 128: two topics per lane in the solver, the general post step (stm_post_any.h), still without
 * atomics; 128 < K <= 512: the general forms -- four / eight vector components per lane in the solver with the slab and the BFGS
 * matrix in HBM, the general post step with fp64 atomics: correct, not tuned), K * V * 8 < 2^32 per level of beta (32-bit row
 * offsets): STM_ERR_INVALID beyond */
int stm_set_topics(stm_handle *h, int32_t K);
int stm_put_beta(stm_handle *h, const double *beta /* [A][K][V] */);
int stm_put_eta(stm_handle *h, const double *eta /* [N][K-1] */);
int stm_put_mu(stm_handle *h, const double *mu /* [N][K-1] */);
int stm_get_beta(stm_handle *h, double *beta);
int stm_get_eta(stm_handle *h, double *eta);
int stm_get_mu(stm_handle *h, double *mu);
int stm_get_theta(stm_handle *h, double *theta /* [N][K] */);

/* ---- the hot path: STM.E_step (stm.py:489-597) ------------------------ */
/* siginv / sigmaentropy are the reference preamble's values (stm.py:499-501),
 * computed by the caller with the reference's own expression.  Runs the
 * per-document BFGS (stm.py:917-962 + scipy BFGS), theta (547-549), Hessian
 * (986-1026), Cholesky (1031-1050), bound (1068-1101), nu (1052-1066), phi
 * (1103-1118) and accumulates sigma_ss / beta_ss (582-588) on the GPU.
 * eta is updated in place on the device; *bound_total = sum of bounds (592). */
int stm_estep(stm_handle *h, const double *siginv, double sigmaentropy, double *bound_total);
int stm_get_sigma_ss(stm_handle *h, double *sigma_ss /* [(K-1)^2] */);
int stm_get_beta_ss(stm_handle *h, double *beta_ss /* [A][K][V] */);
int stm_get_bound_docs(stm_handle *h, double *bound /* [N] */);
/* overwrite the device-resident sufficient statistics (host-side reduction fallback) */
int stm_put_sigma_ss(stm_handle *h, const double *sigma_ss /* [(K-1)^2] */);
int stm_put_beta_ss(stm_handle *h, const double *beta_ss /* [A][K][V] */);
/* per-document solver diagnostics of the last E-step (any pointer may be NULL):
 * scipy OptimizeResult.status / .nit, evaluation counts, PD-fix path
 * (0 none, 1 make_pd, 2 make_pd + 1e-5; stm.py:1017-1021) */
int stm_get_diagnostics(stm_handle *h, int32_t *status, int32_t *nit, int32_t *nfev, int32_t *njev,
                        int32_t *pd_path);
/* phi (K x Nd) of one document as left in self.phi by stm.py:1116 */
int stm_get_phi(stm_handle *h, int64_t doc, double *phi);

/* One-shot form over host buffers (upload, E-step, download): what a
 * reference-side `STM.E_step` stub would call.  Same field meaning as above. */
typedef struct stm_estep_args {
    int64_t N;
    int32_t K, V, A;
    const int64_t *indptr;
    const int32_t *indices;
    const double *counts;
    const int32_t *aspect;  /* nullable */
    const double *beta;     /* [A][K][V] */
    const double *mu;       /* [N][K-1] */
    double *eta;            /* [N][K-1] in/out */
    const double *siginv;   /* [(K-1)^2] */
    double sigmaentropy;
    double *theta;          /* [N][K] out */
    double *bound;          /* [N] out, nullable */
    double *sigma_ss;       /* out */
    double *beta_ss;        /* out */
    double *bound_total;    /* out */
    int32_t *status, *nit, *nfev, *njev, *pd_path; /* out, nullable */
} stm_estep_args;
int stm_estep_host(const stm_estep_args *args, int device_ordinal);

/* ---- M-step on the device (stm.py:622-747, next-row f-1) -------------- */
/* prevalence covariates X [N][p] as used by update_mu (stm.py:661-671, already
 * one-hot encoded by the host when the reference would encode them); any p (the moment region grows) */
int stm_put_covariates(stm_handle *h, const double *X, int32_t p);
/* local (this shard's) moments for update_mu (stm.py:678-706) and update_sigma (stm.py:723), written into the
 * moment region of the packed sufficient-statistic buffer and, when out != NULL, copied to the host:
 *   [ n_docs | sum_x (p) | sum_eta (K-1) | XtX (p*p) | Xt_eta (p*(K-1)) | eta^T eta ((K-1)^2) ]
 * (p = 0 without covariates: the CTM branch).  With gamma from the centred moments,
 * (eta - X gamma^T)^T (eta - X gamma^T) = eta^T eta - gamma Xt_eta - (gamma Xt_eta)^T + gamma XtX gamma^T,
 * so document shards exchange everything the M-step needs in one exchange per EM iteration (stm_em_begin sends it as two
 * all-reduces: [scalars | sigma_ss | moments] in front of the host's read-back, beta_ss behind it). */
int stm_mstep_moments(stm_handle *h, double *out, int64_t out_len);
/* mu_d = x_d @ gamma^T (stm.py:706; gamma [(K-1)][p]) or, when gamma == NULL,
 * mu_d = mean_eta (CTM branch, stm.py:651; mean_eta [(K-1)]) */
int stm_mstep_set_mu(stm_handle *h, const double *gamma, const double *mean_eta);
/* covariance = (eta - mu)^T (eta - mu) of this shard (stm.py:723) */
int stm_mstep_covariance(stm_handle *h, double *cov /* [(K-1)^2] */);
/* beta = beta_ss / rowsum over words (stm.py:741-745; for 3-D beta_ss the
 * reference's axis=1 sum runs over topics -- reproduced) from the (reduced)
 * beta_ss held on the device */
int stm_mstep_update_beta(stm_handle *h);

/* One EM iteration on resident state with ONE host wait (what STM.expectation_maximization runs):
 *   stm_em_begin   enqueues the E-step (as stm_estep), the moments (as stm_mstep_moments) and, when a communicator is
 *                  attached, the all-reduce of [scalars | sigma_ss | moments] (the head of the packed buffer); waits once;
 *                  returns the (reduced) bound, sigma_ss and moments, and reports the E-step's errors like stm_estep.  Every
 *                  fallible host-side step (sizes, allocations, function attributes) comes before anything is enqueued, and
 *                  a rank's device error flag travels in slot 1 of the packed scalars: with a communicator EVERY rank returns
 *                  an error in the iteration in which any rank's E-step failed.  The word-major beta_ss pass (K <= 112) and,
 *                  with a communicator, the all-reduce of beta_ss are enqueued BEHIND the read-back and may still be running
 *                  when the call returns (the host's M-step algebra overlaps them); whatever touches beta_ss next on the
 *                  handle (stm_em_finish, stm_mstep_update_beta, stm_get_beta_ss, ...) is ordered behind them.  Every rank
 *                  enqueues the same two collectives in the same order.
 *   stm_em_finish  enqueues mu (regression on gamma, or the constant mean_eta when gamma == NULL) and beta from the
 *                  (reduced) beta_ss -- stm_mstep_set_mu + stm_mstep_update_beta without their waits; whatever is called
 *                  next on the handle is ordered behind them. */
int stm_em_begin(stm_handle *h, const double *siginv, double sigmaentropy, double *bound_total,
                 double *sigma_ss /* [(K-1)^2] */, double *moments, int64_t moments_cap);
int stm_em_finish(stm_handle *h, const double *gamma /* [(K-1)][p], nullable */, const double *mean_eta /* [(K-1)] */);

/* ---- held-out likelihood (next-row f-3) -------------------------------- */
/* eval_heldout(heldout, theta, beta) of src/modules/heldout.py:88-97 against the beta resident on the
 * handle: doc_ll[d] = sum_w c_w log(theta_d . beta[:, w]) / sum_w c_w for the documents given as CSR
 * (word ids < V of the handle).  theta [N][K] is a host array; NULL uses the resident theta of the last
 * E-step (then N must be the corpus' N).  The caller averages doc_ll (np.mean, heldout.py:97). */
int stm_eval_heldout(stm_handle *h, int64_t N, const int64_t *indptr, const int32_t *indices,
                     const double *counts, const double *theta, double *doc_ll);

/* ---- spectral initialisation (next-row f-4; reference src/modules/stm.py:30-296) ---- */
/* gram (stm.py:122-157): Q = Htilde^T Htilde - diag(Hhat) over the Vk kept terms, dense and resident on the
 * handle.  The caller passes the kept part of the document-term matrix in both orientations -- per document
 * (doc_ptr[N+1], doc_word, doc_h) and per term (word_ptr[Vk+1], word_doc ascending, word_h) -- with
 * h = count / sqrt(n_d (n_d - 1)), n_d the document's length over the kept terms, and hhat[w] = sum_d
 * count_dw / (n_d (n_d - 1)).  STM_ERR_BETA mirrors the reference's assert on the row sums (stm.py:152-154).
 * As imported with scikit-learn >= 1.x the reference's row normalisation (stm.py:156) acts on a discarded
 * copy: Q stays unnormalised, here too. */
int stm_spectral_gram(stm_handle *h, int64_t N, int32_t Vk, const int64_t *doc_ptr, const int32_t *doc_word,
                      const double *doc_h, const int64_t *word_ptr, const int32_t *word_doc, const double *word_h,
                      const double *hhat);
/* The same matrix from the handle's RESIDENT corpus (stm_set_corpus), restricted to the Vk distinct word ids keep[]
 * (stm.py:50-58: the maxV most frequent terms): document lengths over the kept terms, the scaling and both
 * orientations are formed by the library (a host counting sort of the CSR positions + device passes), nothing is
 * prepared in NumPy.  The corpus may be one shard of a document-sharded fit -- gram is a sum over documents
 * (stm.py:122-157): pass flags = 1 (no row-sum assert on the shard's own matrix), sum the shards' matrices with
 * stm_spectral_allreduce (RCCL, in place on the device; a no-op without a communicator) or stm_spectral_get_q /
 * stm_spectral_put_q (host reduction), then stm_spectral_check runs the reference's assert (stm.py:152-154) on the
 * complete matrix.  Every rank then finds the same anchors and weights. */
int stm_spectral_gram_resident(stm_handle *h, int32_t Vk, const int32_t *keep, int32_t flags);
int stm_spectral_allreduce(stm_handle *h);
int stm_spectral_put_q(stm_handle *h, const double *Q /* [Vk][Vk] */);
int stm_spectral_check(stm_handle *h);
/* rows of the matrix fastAnchor's caller holds: gram's result, after stm_spectral_anchors with the first
 * anchor's row rescaled (stm.py:185 modifies the caller's matrix in its first round) */
int stm_spectral_get_q(stm_handle *h, const int32_t *rows, int32_t nrows, double *out /* [nrows][Vk] */);
/* fastAnchor (stm.py:160-226): K greedy anchor terms (indices into the kept terms) */
int stm_spectral_anchors(stm_handle *h, int32_t K, int32_t *anchor /* [K] out */);
/* the per-word QP inputs of recover_l2 (stm.py:239-270): q[i][k] = Q[i] . Q[anchor[k]]; P = q[anchor] */
int stm_spectral_project(stm_handle *h, int32_t K, const int32_t *anchor, double *q_out /* [Vk][K] */);
/* recover_l2's per-term QPs on the device (stm.py:257-285): weights[i] = -argmin_{x <= 0} 1/2 x'Px + q_i'x with P = q[anchor],
 * one-hot rows for the anchor terms; the strictly convex QP is solved as the non-negative least-squares fit it is
 * (Lawson-Hanson active set), one thread per term.  K <= 512. */
int stm_spectral_weights(stm_handle *h, int32_t K, const int32_t *anchor, double *weights_out /* [Vk][K] */);
int stm_spectral_release(stm_handle *h);

/* ---- multi-GPU: one RCCL all-reduce of the sufficient statistics ------- */
/* rank 0 calls stm_comm_unique_id and ships the 128 bytes to the other ranks
 * (any side channel); then every rank calls stm_comm_init. */
int stm_comm_unique_id(void *out128);
int stm_comm_init(stm_handle *h, const void *uid128, int rank, int nranks);
/* what RCCL itself says about the handle's communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): bench
 * provenance for multi-GPU lines.  nranks = 0 when the handle has no communicator. */
int stm_comm_info(stm_handle *h, int32_t *nranks, int32_t *rank, int32_t *device);
/* How stm_em_begin exchanges the sufficient statistics when the handle has a communicator: single = 0 (default, "split") -- two
 * all-reduces per EM iteration, [scalars | sigma_ss | moments] in front of the host's read-back and beta_ss (K V doubles) behind it,
 * overlapping the host's M-step algebra; single = 1 -- ONE all-reduce of the whole packed buffer (the exchange BASELINE.json's
 * north_star names), the beta_ss pass in front of it.  Same sums either way; every rank must choose the same. */
int stm_comm_set_exchange(stm_handle *h, int32_t single);
/* sum over ranks, in place on the device, of the packed buffer
 * [ bound | sigma_ss | moments (as left by stm_mstep_moments) | beta_ss ];
 * the first moments_len doubles of the reduced moment region are copied to `moments` (nullable when 0).
 * Without a communicator (one GPU) nothing is reduced and the local values are returned. */
int stm_allreduce_suffstats(stm_handle *h, double *bound_total, double *moments, int64_t moments_len);
/* sum over ranks of a host vector of any length through its own device buffer (the covariance fallback) */
int stm_allreduce_small(stm_handle *h, double *buf, int64_t len);

/* ---- timing / profiling ---------------------------------------------- */
/* HIP-event time (ms) of the kernels of the last stm_estep on the handle's
 * stream: [0] solver kernel, [1] post kernel, [2] whole E-step */
int stm_last_kernel_ms(stm_handle *h, float *ms3);
/* ... and of the word-major beta_ss pass alone (part of [1] above; 0 beyond 128 topics, where phi is added atomically) */
int stm_last_pass_ms(stm_handle *h, float *ms);
int stm_synchronize(stm_handle *h);

#ifdef __cplusplus
}
#endif
#endif
