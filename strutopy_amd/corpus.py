"""Corpus packing and the streaming synthetic-corpus generator.

* ``pack_bow`` turns the reference's document format -- a list of lists of
  ``(word_id, count)`` tuples (gensim BoW; reference src/modules/stm.py:331-332,
  unpacked per document per iteration at stm.py:522-533) -- into one CSR triple
  that is uploaded to HBM once.
* ``synthetic_corpus`` follows the data-generating process of the reference's
  ``CorpusCreation`` (src/modules/generate_docs.py: beta_k ~ Dir(0.05) :180,
  gamma ~ N(m, 0.001 I) :194-200, x_d in {0,1}^level :207-212,
  eta_d ~ N(x_d gamma^T, 0.001 I) :221-228, theta = softmax([eta, 0]) :268-271,
  words ~ Multinomial(n_words, theta beta) :297-302, unused terms dropped and
  ids re-assigned in order of first appearance :304-316) without materialising
  the dense N x V probability matrix (generate_docs.py:297), so it scales to the
  100k / 1M document configurations.
"""
import os
from dataclasses import dataclass
from itertools import chain

import numpy as np


@dataclass
class PackedCorpus:
    """CSR form of a bag-of-words corpus."""
    indptr: np.ndarray   # int64 [N+1]
    indices: np.ndarray  # int32 [nnz], unique within a document
    counts: np.ndarray   # float64 [nnz]
    V: int

    @property
    def N(self):
        return len(self.indptr) - 1

    @property
    def nnz(self):
        return int(self.indptr[-1])

    def __len__(self):
        return self.N

    def slice(self, lo, hi):
        """Documents [lo, hi) as an independent PackedCorpus (same vocabulary)."""
        a, b = int(self.indptr[lo]), int(self.indptr[hi])
        return PackedCorpus((self.indptr[lo:hi + 1] - a).astype(np.int64), self.indices[a:b].copy(),
                            self.counts[a:b].copy(), self.V)

    def to_bow(self):
        """Back to the reference's list-of-lists-of-tuples format."""
        docs = []
        for i in range(self.N):
            sl = slice(self.indptr[i], self.indptr[i + 1])
            docs.append(list(zip(self.indices[sl].tolist(), self.counts[sl].astype(np.int64).tolist())))
        return docs

    def word_counts(self):
        """Corpus-wide word totals (what stm.py:485-486 derives from create_dtm)."""
        return np.bincount(self.indices, weights=self.counts, minlength=self.V)


def _packbow_module():
    """The C walker of the BoW lists (csrc/packbow.c, built in-tree by __graft_entry__.build); None when it is not built."""
    try:
        from . import _packbow
        return _packbow
    except ImportError:
        return None


def pack_bow(documents, V=None, merge_duplicates=False):
    """The corpus in any of the forms a caller may hold -> PackedCorpus, once (the reference rebuilds
    np.array(documents[i]) for every document in every EM iteration, stm.py:522-533):

    * list[list[(word_id, count)]] -- the reference's BoW lists (stm.py:311): one pass in C (csrc/packbow.c; an
      iterator-based NumPy path when the helper is not built) -- no per-document Python statement or NumPy call
    * a PackedCorpus, or the CSR triple (indptr, indices, counts) itself: validated, no copy of already-typed arrays
    * a scipy.sparse matrix (documents x terms, what create_dtm builds, stm.py:87-119)
    * a path to a MatrixMarket file (the format of the shipped src/artifacts/wiki_data/BoW_corpus.mm)

    A word id may appear once per document (gensim's doc2bow, what the reference is fed, guarantees it): a repeated id raises
    ValueError naming the document -- a deviation from the reference, which accepts it and then counts that word's phi column once in
    beta_ss and twice everywhere else (stm.py:588).  merge_duplicates=True sums the counts of a repeated id instead
    (what the scipy.sparse path always does).
    """
    if isinstance(documents, PackedCorpus):     # (checked once: the product's own objects carry the mark)
        if getattr(documents, "_checked_ok", False):
            return documents
        return _checked(_merged(documents) if merge_duplicates else documents, V)
    if isinstance(documents, (str, os.PathLike)):
        c = read_mm(documents)
        if V is not None and c.V < V:
            c = PackedCorpus(c.indptr, c.indices, c.counts, int(V))
        return _checked(c, V)
    if hasattr(documents, "tocsr") and hasattr(documents, "shape"):      # scipy.sparse, documents x terms
        m = documents.tocsr(copy=True)     # tocsr() of a CSR matrix is the caller's object: never canonicalise that in place
        m.sum_duplicates(); m.sort_indices()
        return _checked(PackedCorpus(m.indptr.astype(np.int64), m.indices.astype(np.int32), np.asarray(m.data, dtype=np.float64),
                                     int(m.shape[1] if V is None else max(V, m.shape[1]))), V)
    if isinstance(documents, (tuple, list)) and len(documents) == 3 and isinstance(documents[0], np.ndarray) and np.ndim(documents[0]) == 1 \
            and isinstance(documents[1], np.ndarray) and isinstance(documents[2], np.ndarray):
        indptr, indices, counts = documents
        vmax = int(np.max(indices)) + 1 if len(indices) else 0
        c = PackedCorpus(np.ascontiguousarray(indptr, dtype=np.int64), np.ascontiguousarray(indices, dtype=np.int32),
                         np.ascontiguousarray(counts, dtype=np.float64), int(vmax if V is None else V))
        return _checked(_merged(c) if merge_duplicates else c, V)
    N = len(documents)
    pb = _packbow_module()
    if pb is not None:
        lens = np.zeros(N, dtype=np.int64)
        nnz = int(pb.lengths(documents, lens))
        indptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(lens, out=indptr[1:])
        indices = np.empty(nnz, dtype=np.int32)
        counts = np.empty(nnz, dtype=np.float64)
        vmax = int(pb.fill(documents, indices, counts)) + 1
    else:
        lens = np.fromiter(map(len, documents), dtype=np.int64, count=N)
        if N and lens.min() < 1:
            raise IndexError("empty document: the reference indexes doc_array[:, 0] (stm.py:523)")
        indptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(lens, out=indptr[1:])
        nnz = int(indptr[-1])
        try:
            flat = np.fromiter(chain.from_iterable(chain.from_iterable(documents)), dtype=np.float64, count=2 * nnz)
        except ValueError as e:   # an entry that is not a (word_id, count) pair
            raise IndexError(f"documents must hold (word_id, count) pairs (stm.py:522-526): {e}") from None
        indices = flat[0::2].astype(np.int32)
        counts = np.ascontiguousarray(flat[1::2])
        if nnz and (indices.min() < 0 or np.any(indices != flat[0::2])):
            raise IndexError("word ids must be non-negative integers below 2^31")
        vmax = int(indices.max()) + 1 if nnz else 0
    if V is None:
        V = vmax
    elif vmax > V:
        raise IndexError(f"word id {vmax - 1} is out of range for a dictionary of length {V}")
    c = PackedCorpus(indptr, indices, counts, int(V))
    return _unique_words(_merged(c) if merge_duplicates else c)


def _merged(c):
    """The counts of a word id that a document holds more than once, summed (the document's ids then ascend); documents without
    a repeated id keep their order."""
    if len(c.indices) < 2 or (c.N and np.min(np.diff(c.indptr)) < 1):
        return c
    doc = np.repeat(np.arange(c.N, dtype=np.int64), np.diff(c.indptr))
    key = doc * np.int64(max(c.V, 1)) + c.indices
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.concatenate([[True], np.diff(ks) != 0])
    if first.all():
        return c
    dup_docs = np.unique(doc[order][~first])
    touched = np.isin(doc, dup_docs)
    keep = np.flatnonzero(~touched)                         # entries of untouched documents, in place
    o2 = order[np.isin(doc[order], dup_docs)]               # entries of the documents with repeats, sorted by (doc, id)
    k2 = key[o2]
    f2 = np.concatenate([[True], np.diff(k2) != 0])
    starts = np.flatnonzero(f2)
    m_idx, m_cnt, m_doc = c.indices[o2][starts], np.add.reduceat(c.counts[o2], starts), doc[o2][starts]
    all_doc = np.concatenate([doc[keep], m_doc])
    pos = np.concatenate([keep, o2[starts]])                # original positions keep the order inside untouched documents
    fin = np.lexsort((pos, all_doc))
    indices = np.concatenate([c.indices[keep], m_idx])[fin].astype(np.int32)
    counts = np.concatenate([c.counts[keep], m_cnt])[fin]
    indptr = np.zeros(c.N + 1, dtype=np.int64)
    np.cumsum(np.bincount(all_doc, minlength=c.N), out=indptr[1:])
    return PackedCorpus(indptr, indices, np.ascontiguousarray(counts, dtype=np.float64), c.V)


def _unique_words(c):
    """A word id may appear once per document -- what gensim's doc2bow produces and stm_set_corpus requires (the kernels let
    no two lanes share a word's cells; the reference itself would count a repeated id's phi column once in beta_ss and twice
    everywhere else, stm.py:588).  Rejected here, with the document named; scipy.sparse input has its duplicates summed."""
    if c.N and np.min(np.diff(c.indptr)) < 1:      # (the indexing below takes indptr - 1)
        raise IndexError("empty document: the reference indexes doc_array[:, 0] (stm.py:523)")
    if len(c.indices) < 2:
        c._checked_ok = True
        return c
    inner = np.ones(len(c.indices) - 1, dtype=bool)
    inner[np.asarray(c.indptr[1:-1], dtype=np.int64) - 1] = False       # pairs that straddle two documents
    if not np.all(np.diff(c.indices)[inner] > 0):                       # (rows in ascending order are unique at once)
        doc = np.repeat(np.arange(c.N, dtype=np.int64), np.diff(c.indptr))
        key = np.sort(doc * np.int64(max(c.V, 1)) + c.indices)
        dup = np.flatnonzero(np.diff(key) == 0)
        if len(dup):
            d, w = divmod(int(key[dup[0]]), max(c.V, 1))
            raise ValueError(f"document {d} holds word id {w} more than once: merge the counts first "
                             "(pack_bow(..., merge_duplicates=True); gensim's doc2bow never produces this)")
    c._checked_ok = True
    return c


def _checked(c, V):
    """A CSR corpus handed over as arrays: the checks pack_bow makes on the BoW lists."""
    if len(c.indptr) < 1 or c.indptr[0] != 0 or (c.N and np.min(np.diff(c.indptr)) < 1):
        raise IndexError("empty document: the reference indexes doc_array[:, 0] (stm.py:523)")
    if len(c.indices) != int(c.indptr[-1]) or len(c.counts) != len(c.indices):
        raise IndexError("indptr, indices and counts disagree")
    if len(c.indices) and (int(c.indices.min()) < 0 or int(c.indices.max()) >= c.V):
        raise IndexError(f"word id {int(c.indices.max())} is out of range for a dictionary of length {c.V}")
    if V is not None and c.V > V:
        if len(c.indices) and int(c.indices.max()) >= V:
            raise IndexError(f"word id {int(c.indices.max())} is out of range for a dictionary of length {V}")
        c = PackedCorpus(c.indptr, c.indices, c.counts, int(V))
    return _unique_words(c)


def read_mm(path):
    """MatrixMarket coordinate file (documents x terms, 1-based, as gensim's MmCorpus writes the
    reference's src/artifacts/wiki_data/BoW_corpus.mm) -> PackedCorpus, without going through Python
    lists of tuples (the reference rebuilds np.array(documents[i]) per document per iteration, stm.py:522).
    Entries of one document keep their file order when the file is already grouped by document (gensim writes
    it that way); otherwise they are grouped stably."""
    with open(path, "rb") as fh:
        head = fh.readline()
        if not head.startswith(b"%%MatrixMarket matrix coordinate"):
            raise ValueError("not a MatrixMarket coordinate file")
        line = fh.readline()
        while line.startswith(b"%"):
            line = fh.readline()
        n_docs, n_terms, nnz = (int(t) for t in line.split())
        try:   # the C tokenizer of pandas: ~20x np.loadtxt
            import pandas as pd
            data = pd.read_csv(fh, sep=r"\s+", header=None, comment="%", dtype=np.float64, engine="c").to_numpy()
        except ImportError:
            data = np.loadtxt(fh, ndmin=2)
    if data.shape[0] != nnz or (nnz and data.shape[1] != 3):
        raise ValueError(f"expected {nnz} entries of (doc, term, value), found an array of shape {data.shape}")
    doc = data[:, 0].astype(np.int64) - 1
    term = data[:, 1].astype(np.int64) - 1
    val = data[:, 2]
    if nnz and (doc.min() < 0 or doc.max() >= n_docs or term.min() < 0 or term.max() >= n_terms):
        raise ValueError("entry outside the declared matrix shape")
    if nnz and np.any(np.diff(doc) < 0):
        order = np.argsort(doc, kind="stable")
        doc, term, val = doc[order], term[order], val[order]
    lens = np.bincount(doc, minlength=n_docs)
    if n_docs and lens.min() < 1:
        raise IndexError("empty document: the reference indexes doc_array[:, 0] (stm.py:523)")
    indptr = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr[1:])
    return PackedCorpus(indptr, term.astype(np.int32), np.ascontiguousarray(val, dtype=np.float64), int(n_terms))


@dataclass
class SyntheticCorpus:
    corpus: PackedCorpus
    X: np.ndarray          # [N, level] 0/1 prevalence covariates (the generator's `metadata`)
    beta_true: np.ndarray  # [K, V_requested]
    gamma_true: np.ndarray
    V_requested: int


def synthetic_corpus(n_docs, V, K, n_words=150, level=1, seed=12345, remove_terms=True, chunk=20000):
    """STM data-generating process of the reference's CorpusCreation, streamed.

    A multinomial draw of ``n_words`` from the mixture ``theta_d @ beta`` equals
    ``n_words`` i.i.d. draws of (topic ~ theta_d, word ~ beta_topic); sampling the
    pairs directly avoids the dense N x V matrix.
    """
    rng = np.random.default_rng(seed)
    beta = rng.dirichlet(np.repeat(0.05, V), size=K)
    mean = rng.standard_normal(level)
    mean = rng.multivariate_normal(mean, np.diag(np.full(level, 0.001)))
    gamma = rng.multivariate_normal(mean, np.diag(np.full(level, 0.001)), K - 1)  # (K-1) x level
    X = rng.integers(0, 2, size=(n_docs, level))
    cdf_beta = np.cumsum(beta, axis=1)
    cdf_beta[:, -1] = 1.0
    all_idx, all_cnt, lens = [], [], []
    for lo in range(0, n_docs, chunk):
        hi = min(n_docs, lo + chunk)
        m = hi - lo
        eta = X[lo:hi] @ gamma.T + rng.normal(0.0, np.sqrt(0.001), size=(m, K - 1))
        eta_ = np.concatenate([eta, np.zeros((m, 1))], axis=1)
        eta_ -= eta_.max(axis=1, keepdims=True)
        theta = np.exp(eta_)
        theta /= theta.sum(axis=1, keepdims=True)
        cdf_theta = np.cumsum(theta, axis=1)
        cdf_theta[:, -1] = 1.0
        u = rng.random((m, n_words))
        # topic of every token: one flat searchsorted over row-offset CDFs
        rows = np.arange(m, dtype=np.float64)[:, None]
        z = np.searchsorted((cdf_theta + rows).ravel(), (u + rows).ravel(), side="right").reshape(m, n_words)
        z -= (np.arange(m) * K)[:, None]
        np.clip(z, 0, K - 1, out=z)
        u2 = rng.random((m, n_words))
        w = np.empty((m, n_words), dtype=np.int64)
        for k in range(K):
            sel = z == k
            if sel.any():
                w[sel] = np.searchsorted(cdf_beta[k], u2[sel], side="right")
        np.clip(w, 0, V - 1, out=w)
        w.sort(axis=1)
        # run-length encode every row
        new = np.ones((m, n_words), dtype=bool)
        new[:, 1:] = w[:, 1:] != w[:, :-1]
        flat_new = new.ravel()
        starts = np.flatnonzero(flat_new)
        ends = np.append(starts[1:], m * n_words)
        all_idx.append(w.ravel()[starts])
        all_cnt.append((ends - starts).astype(np.float64))
        lens.append(new.sum(axis=1))
    idx = np.concatenate(all_idx)
    cnt = np.concatenate(all_cnt)
    lens = np.concatenate(lens)
    indptr = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr[1:])
    V_eff = V
    if remove_terms:
        # generate_docs.py:304-316: ids are re-assigned in order of first appearance
        first = np.full(V, np.iinfo(np.int64).max, dtype=np.int64)
        np.minimum.at(first, idx, np.arange(len(idx)))
        used = np.flatnonzero(first != np.iinfo(np.int64).max)
        order = used[np.argsort(first[used], kind="stable")]
        remap = np.full(V, -1, dtype=np.int64)
        remap[order] = np.arange(len(order))
        idx = remap[idx]
        V_eff = len(order)
    corpus = PackedCorpus(indptr, idx.astype(np.int32), cnt, int(V_eff))
    return SyntheticCorpus(corpus, X.astype(np.float64), beta, gamma, V)
