"""Held-out likelihood by document completion -- drop-in for the two helpers of the reference's
src/modules/heldout.py that its training script uses (src/05_train.py:120): ``cut_in_half`` (:70-85)
and ``eval_heldout`` (:88-97).  The per-word gather-dot runs on the GPU (stm_eval_heldout); there is
no CPU fallback.
"""
import numpy as np

from .corpus import PackedCorpus, pack_bow


def cut_in_half(doc_set):
    """heldout.py:70-85: every other (word, count) pair, starting at index 0 / at index 1.

    Accepts the reference's array/list of BoW documents (returns two object arrays like the reference)
    or a PackedCorpus (returns two PackedCorpus).
    """
    if isinstance(doc_set, PackedCorpus):
        ip, ix, c = doc_set.indptr, doc_set.indices, doc_set.counts
        halves = []
        for start in (0, 1):
            pos = np.concatenate([np.arange(ip[i] + start, ip[i + 1], 2) for i in range(doc_set.N)]) if doc_set.N else np.zeros(0, np.int64)
            lens = np.array([len(range(int(ip[i]) + start, int(ip[i + 1]), 2)) for i in range(doc_set.N)], dtype=np.int64)
            indptr = np.zeros(doc_set.N + 1, dtype=np.int64)
            np.cumsum(lens, out=indptr[1:])
            halves.append(PackedCorpus(indptr, ix[pos].copy(), c[pos].copy(), doc_set.V))
        return halves[0], halves[1]
    first_half = np.zeros(len(doc_set), dtype=np.ndarray)
    second_half = np.zeros(len(doc_set), dtype=np.ndarray)
    for doc in range(len(doc_set)):
        first_half[doc] = doc_set[doc][0::2]
        second_half[doc] = doc_set[doc][1::2]
    return first_half, second_half


def eval_heldout(heldout, theta, beta, device=0, engine=None):
    """heldout.py:88-97: mean over documents of sum_w c_w log(theta_d @ beta[:, w]) / sum_w c_w."""
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    if beta.ndim != 2:
        raise ValueError("eval_heldout indexes a 2-D beta (heldout.py:93)")
    K, V = beta.shape
    docs = pack_bow(list(heldout) if not isinstance(heldout, PackedCorpus) else heldout, V=V)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    if theta.shape != (docs.N, K):
        raise ValueError(f"theta has shape {theta.shape}, expected {(docs.N, K)}")
    own = engine is None
    if own:
        from .engine import HipEstepEngine  # raises without a GPU / the built library
        engine = HipEstepEngine(device)
        engine.set_corpus(docs.indptr, docs.indices, docs.counts, V)
        engine.set_topics(K)
        engine.put_beta(beta)
    try:
        doc_ll = engine.eval_heldout(docs.indptr, docs.indices, docs.counts, theta)
    finally:
        if own:
            engine.close()
    return np.mean(doc_ll)
