"""Spectral initialisation of beta -- drop-in for `spectral_init` of the reference (src/modules/stm.py:30-85, the
`init_type="spectral"` that src/05_train.py:92 and src/03_fit_reference_model.py use).

Same statements as the reference, with the V x V work on the GPU behind the C-ABI (include/stm_estep.h):
  word probabilities, the maxV most frequent terms      host (np.argsort(-wprob), stm.py:50-58)
  gram: Q = Htilde^T Htilde - diag(Hhat)                 stm_spectral_gram     (stm.py:122-157)
  fastAnchor: greedy anchor terms                        stm_spectral_anchors  (stm.py:160-226)
  recover_l2: q_i = M y_i for every term, P = M M^T,      stm_spectral_weights  (stm.py:239, 257-285)
              per-term QP  min 1/2 x'Px + q'x, x <= 0
  beta[:, keep], + 0.001 / V, / total sum                 host (stm.py:78-83)

The reference hands the per-term QP to qpsolvers/quadprog.  It is the strictly convex non-negative least-squares
problem min || M^T w - y ||, w >= 0 (w = -x), whose minimiser does not depend on the solver; it is solved on the device
by a Lawson-Hanson active set, one thread per term (the tests hold it against quadprog's Goldfarb-Idnani method as restated
in oracle/spectral_oracle.py and against the QP's KKT conditions).  Quirks of the reference that are kept: Q is NOT row-normalised
(sklearn's normalize(copy=False) works on a discarded CSR copy of the CSC product), fastAnchor uses column sums of
squares and never projects row 0, the first anchor's row of the caller's Q is rescaled, no sum-to-one constraint,
and the final division by the TOTAL sum leaves every row of beta summing to 1 / K.
"""
import numpy as np

from .corpus import pack_bow


def kept_terms(corpus, maxV=5000, comm=None):
    """wprob and keep of stm.py:50-58.  create_dtm's csr_matrix has max(word id) + 1 columns.  With a multi-rank `comm`
    the corpus is this rank's shard: the term totals (integers, exact in fp64 in any order) are summed over the shards, so
    every rank -- like a single process on the whole corpus -- ranks the same numbers."""
    vmax = int(corpus.indices.max()) + 1
    if comm is not None and comm.size > 1:
        vmax = int(comm.allreduce_host(np.array([float(vmax)]), op="max")[0])
    tot = np.bincount(corpus.indices, weights=corpus.counts, minlength=vmax)
    if comm is not None and comm.size > 1:
        tot = comm.allreduce_host(tot)
    wprob = tot / np.sum(tot)
    keep = np.argsort(-1 * wprob)[:maxV]
    return wprob[keep], keep


def gram_inputs(corpus, keep):
    """The kept columns of the document-term matrix in both orientations, scaled as gram() scales them (stm.py:135-146)."""
    vmax = max(int(corpus.indices.max()), int(np.max(keep))) + 1     # (a shard need not contain every kept term)
    pos = np.full(vmax, -1, dtype=np.int64)
    pos[keep] = np.arange(len(keep))
    col = pos[corpus.indices]
    sel = col >= 0
    doc = np.repeat(np.arange(corpus.N, dtype=np.int64), np.diff(corpus.indptr))[sel]
    col, cnt = col[sel], corpus.counts[sel]
    wc = np.bincount(doc, weights=cnt, minlength=corpus.N)              # word_counts = dtm.sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        div = wc * (wc - 1)
        h = cnt / np.sqrt(div)[doc]
        hhat = np.bincount(col, weights=cnt / div[doc], minlength=len(keep))
    doc_ptr = np.zeros(corpus.N + 1, dtype=np.int64)
    np.cumsum(np.bincount(doc, minlength=corpus.N), out=doc_ptr[1:])
    order = np.argsort(col, kind="stable")                               # term-major, documents ascending within a term
    word_ptr = np.zeros(len(keep) + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=len(keep)), out=word_ptr[1:])
    return dict(doc_ptr=doc_ptr, doc_word=col.astype(np.int32), doc_h=np.ascontiguousarray(h),
                word_ptr=word_ptr, word_doc=doc[order].astype(np.int32), word_h=np.ascontiguousarray(h[order]), hhat=hhat)


def spectral_init(corpus, K, V, maxV=5000, verbose=True, engine=None, details=None, comm=None, resident=False):
    """Drop-in for spectral_init(corpus, K, V, maxV) (stm.py:30-85); `corpus` is the BoW list or a PackedCorpus,
    `engine` a strutopy_amd.engine.HipEstepEngine (one is created on GPU 0 when omitted).

    resident: the engine already holds `corpus` (STM.__init__ has called set_corpus) -- gram then runs on the resident CSR
    (stm_spectral_gram_resident: no NumPy preparation of the two scaled orientations).  comm: `corpus` is this rank's
    shard of a document-sharded fit; gram is a sum over documents, so the shards' matrices are summed (one all-reduce of
    Vk^2 doubles) and every rank finds the same anchors and the same beta."""
    corpus = pack_bow(corpus)
    own = engine is None
    if own:
        from .engine import HipEstepEngine
        engine = HipEstepEngine(0)
    sharded = comm is not None and comm.size > 1
    try:
        wprob, keep = kept_terms(corpus, maxV, comm)
        if verbose:
            print("Create gram matrix...")
        # a caller's engine keeps the corpus it holds (its aspect / A, longest-first order and word-major index belong to the
        # caller's model): only an engine created here, or one the caller says already holds `corpus`, runs the resident gram
        if hasattr(engine, "spectral_gram_resident") and (own or resident):
            err = ""
            try:
                if not resident:
                    engine.set_corpus(corpus.indptr, corpus.indices, corpus.counts, max(int(V), int(corpus.indices.max()) + 1))
                engine.spectral_gram_resident(keep, check=not sharded)
            except Exception as e:
                if not sharded:
                    raise
                err = f"{type(e).__name__}: {e}"
            if sharded:
                # a rank that failed must not leave its peers waiting in the Vk^2 all-reduce: one status word first
                errs = [e for e in comm.allgather(err) if e]
                if errs:
                    raise RuntimeError("spectral initialisation failed on a rank: " + errs[0])
                comm.spectral_reduce(engine)
                engine.spectral_check()
        else:
            if sharded:
                raise NotImplementedError("a sharded spectral initialisation needs the engine's resident gram (resident=True)")
            engine.spectral_gram(corpus.N, len(keep), gram_inputs(corpus, keep))
        if verbose:
            print("Find anchor words...")
        anchor = engine.spectral_anchors(K)
        if verbose:
            print("Recover values for beta")
        weights = engine.spectral_weights(anchor)              # q_i = M y_i and the per-term QPs, on the device
        if details is not None:
            details.update(wprob=wprob, keep=keep, anchor=anchor.astype(np.float64), weights=weights)
        engine.spectral_release()
    finally:
        if own:
            engine.close()
    A = weights.T * wprob                                                # p(w|z) = p(z|w) p(w), stm.py:290
    A = A.T / np.sum(A, axis=1)
    assert np.any(A > 0), "Negative probabilities for some words."
    assert np.any(A < 1), "Word probabilities larger than one."
    beta = A.T
    beta_new = np.zeros(K * V).reshape(K, V)
    beta_new[:, keep] = beta
    beta_new = beta_new + 0.001 / V
    if details is not None:
        details["beta_kept"] = beta
    return beta_new / np.sum(beta_new)
