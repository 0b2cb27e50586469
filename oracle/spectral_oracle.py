"""CPU restatement (NumPy, dense) of the reference's spectral initialisation -- TEST INFRASTRUCTURE.

Follows reference src/modules/stm.py:30-296 AS IMPORTED IN THIS IMAGE (scipy 1.15.3, scikit-learn 1.7.2):
  spectral_init :30-85   word probabilities, the maxV most frequent terms (np.argsort(-wprob)), gram, fastAnchor,
                         recover_l2, beta embedded into K x V, + 0.001 / V, divided by the TOTAL sum (:81-83)
  gram          :122-157 Q = Htilde^T Htilde - diag(Hhat), Htilde = dtm / sqrt(n_d (n_d - 1)),
                         Hhat = colsum(dtm / (n_d (n_d - 1))).  The product is a CSC matrix, and
                         sklearn.preprocessing.normalize(Q, copy=False) converts a CSC input to CSR -- a copy --
                         whose result :156 discards: the matrix the reference goes on with is NOT row-normalised.
  fastAnchor    :160-226 greedy Gram-Schmidt: COLUMN sums of squares (:174, :221), the chosen row scaled by
                         1 / sqrt(max) (:185; in the first round this also rescales the caller's Q, which is still
                         the same sparse object), projection removed from every row except those listed in
                         `basis` -- a float vector initialised with zeros, so row 0 is always excluded (:216) and
                         column 0 can only be chosen first (:222)
  recover_l2    :229-296 per word the QP  min 1/2 x^T P x + q^T x  s.t. x <= 0  (P = M M^T, q = M y, M = anchor rows of
                         Q), weights = -x, i.e. the non-negative least-squares fit  min || M^T w - y ||, w >= 0
                         (no sum-to-one constraint, :244-246 is commented out); anchors get a one-hot row (:261-264).
                         The reference calls qpsolvers.solve_qp(solver="quadprog"), absent from this image: the QP is
                         strictly convex, so its minimiser does not depend on the solver; scipy.optimize.nnls solves it
                         here (PARITY UNPINNED for this one step: no reference output can be generated offline).

Only tests/, __graft_entry__.smoke() and tools/ import this module.
"""
import numpy as np


def word_probabilities(indptr, indices, counts, maxV=5000):
    """wprob and `keep` of stm.py:50-58 (V = largest word id + 1, like create_dtm's csr_matrix)."""
    vmax = int(indices.max()) + 1
    tot = np.bincount(indices, weights=counts, minlength=vmax)
    wprob = tot / np.sum(tot)
    keep = np.argsort(-1 * wprob)[:maxV]
    return wprob[keep], keep


def gram(indptr, indices, counts, keep):
    """stm.py:122-157 on the kept columns, dense."""
    vmax = int(indices.max()) + 1
    pos = np.full(vmax, -1, dtype=np.int64)
    pos[keep] = np.arange(len(keep))
    N, Vk = len(indptr) - 1, len(keep)
    D = np.zeros((N, Vk))
    doc = np.repeat(np.arange(N), np.diff(indptr))
    sel = pos[indices] >= 0
    D[doc[sel], pos[indices[sel]]] = counts[sel]
    wc = D.sum(axis=1)
    div = wc * (wc - 1)
    Ht = D / np.sqrt(div)[:, None]
    Q = Ht.T @ Ht - np.diag((D / div[:, None]).sum(axis=0))
    assert np.all(Q.sum(axis=1) > 0), "Encountered zeroes in Q row sums, can not normalize."
    return Q


def fast_anchor(Q, K):
    """stm.py:160-226.  Returns (anchor indices, the caller-visible Q = input with the first anchor row rescaled)."""
    Q = np.array(Q, dtype=np.float64, copy=True)
    rss = np.sum(Q * Q, axis=0)
    basis = np.zeros(K)
    Q_caller = None
    for i in range(K):
        maxind = int(np.argmax(rss))
        basis[i] = maxind
        normalizer = 1 / np.sqrt(rss[maxind])
        Q[maxind] = Q[maxind] * normalizer
        if i == 0:
            Q_caller = Q.copy()
        inner = Q @ Q[maxind]
        project = np.outer(inner, Q[maxind])
        project[np.intp(basis)] = 0
        Q = Q - project
        rss = np.sum(Q * Q, axis=0)
        rss[np.intp(basis)] = 0
    return basis, Q_caller


def recover_l2(Q, anchor, wprob):
    """stm.py:229-296 with scipy.optimize.nnls in place of quadprog (see the module docstring)."""
    from scipy.optimize import nnls
    anchor = np.intp(anchor)
    M = Q[anchor]
    P = M @ M.T
    R = np.linalg.cholesky(P).T                      # P = R^T R
    Vk, K = Q.shape[0], len(anchor)
    weights = np.zeros((Vk, K))
    for i in range(Vk):
        hit = np.where(anchor == i)[0]
        if len(hit):
            weights[i, hit] = 1
        else:
            q = M @ Q[i]
            weights[i] = nnls(R, np.linalg.solve(R.T, q))[0]
    A = weights.T * wprob
    A = A.T / np.sum(A, axis=1)
    return A.T


def spectral_init(indptr, indices, counts, K, V, maxV=5000):
    wprob, keep = word_probabilities(indptr, indices, counts, maxV)
    Q = gram(indptr, indices, counts, keep)
    anchor, Qc = fast_anchor(Q, K)
    beta = recover_l2(Qc, anchor, wprob)
    beta_new = np.zeros(K * V).reshape(K, V)
    beta_new[:, keep] = beta
    beta_new = beta_new + 0.001 / V
    return beta_new / np.sum(beta_new), dict(wprob=wprob, keep=keep, Q=Q, anchor=anchor, Q_caller=Qc, beta_kept=beta)
