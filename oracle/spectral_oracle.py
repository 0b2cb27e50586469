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
                         The reference calls qpsolvers.solve_qp(solver="quadprog") -- a third-party dependency that is
                         absent from this image (pyproject.toml: qpsolvers[quadprog]; uv.lock pins quadprog 0.1.13).
                         quadprog is R's solve.QP: the dual active-set method of Goldfarb & Idnani, "A numerically
                         stable dual method for solving strictly convex quadratic programs", Mathematical Programming
                         27 (1983) 1-33.  `solve_qp_goldfarb_idnani` below restates that published algorithm (steps
                         0-2 of its section 3) and `recover_l2` runs it; tests/test_spectral.py also holds it against
                         the non-negative least-squares form on both fixtures' inputs.

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


def solve_qp_goldfarb_idnani(P, q, G, h, max_iter=None, Pinv=None):
    """min 1/2 x'Px + q'x  s.t.  G x <= h, P symmetric positive definite: the dual active-set method of Goldfarb & Idnani
    (1983), the algorithm behind quadprog / R's solve.QP that the reference reaches through qpsolvers (stm.py:271-281).

    In the paper's notation the constraints are n_j' x >= b_j (here n_j = -G[j], b_j = -h[j]); N holds the normals of the
    active set A, u >= 0 its multipliers, and with N* = (N' P^-1 N)^-1 N' P^-1 and H = P^-1 (I - N N*):
      step 0  x = -P^-1 q (the unconstrained minimum), A empty
      step 1  pick a violated constraint p (quadprog: the largest violation relative to the length of its normal);
              none -> x is optimal
      step 2  z = H n_p (primal step direction), r = N* n_p (dual step direction);
              t1 = min { u_k / r_k : r_k > 0 } (keeps u >= 0; k leaves A), t2 = -s_p / (z' n_p) (makes p active), t = min;
              t infinite: infeasible.  z = 0: dual step only (u -= t r, u_p += t), drop k, repeat step 2.
              Otherwise x += t z, u -= t r, u_p += t; full step (t = t2): p joins A, back to step 1; partial step:
              drop k, repeat step 2.
    Dense NumPy with the operators rebuilt from their definitions at every step (n <= 128 here); `Pinv`: P^-1 when the
    caller solves many QPs with the same P (recover_l2: one per term)."""
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).ravel()
    Nall = -np.asarray(G, dtype=np.float64).T          # column j = n_j
    b = -np.asarray(h, dtype=np.float64).ravel()
    m = Nall.shape[1]
    if Pinv is None:
        L = np.linalg.cholesky(P)
        Pinv = np.linalg.solve(L.T, np.linalg.solve(L, np.eye(len(q))))

    def psolve(v):                                      # P^-1 v
        return Pinv @ v

    norms = np.sqrt(np.sum(Nall * Nall, axis=0))
    x = -psolve(q)
    A, u = [], np.zeros(0)
    eps = np.finfo(float).eps
    it, max_iter = 0, (max_iter or 50 * (m + len(q)) + 50)
    while True:
        s = Nall.T @ x - b
        viol = np.where(np.isin(np.arange(m), A), 0.0, s / norms)
        p = int(np.argmin(viol))
        if viol[p] >= -1e3 * eps * max(1.0, float(np.abs(x).max())):
            return x
        n_p, u_p = Nall[:, p], 0.0
        while True:
            it += 1
            if it > max_iter:
                raise RuntimeError("Goldfarb-Idnani: iteration limit")
            if A:
                NA = Nall[:, A]
                W = psolve(NA)                                         # P^-1 N
                r = np.linalg.solve(NA.T @ W, W.T @ n_p)               # N* n_p
                z = psolve(n_p) - W @ r                                # H n_p
            else:
                r, z = np.zeros(0), psolve(n_p)
            t1, k = np.inf, -1
            for idx in range(len(A)):
                if r[idx] > 0 and u[idx] / r[idx] < t1:
                    t1, k = u[idx] / r[idx], idx
            zn = float(z @ n_p)
            s_p = float(n_p @ x - b[p])
            t2 = -s_p / zn if zn > eps * max(1.0, float(n_p @ psolve(n_p))) else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                raise ValueError("constraints are inconsistent, no solution")   # quadprog's message
            if not np.isfinite(t2):                                            # step in the dual space only
                u = u - t * r
                u_p += t
                A.pop(k); u = np.delete(u, k)
                continue
            x = x + t * z
            u = u - t * r
            u_p += t
            if t == t2:                                                        # full step: p becomes active
                A.append(p); u = np.append(u, u_p)
                break
            A.pop(k); u = np.delete(u, k)                                      # partial step: k leaves, p stays violated


def solve_qp(P, q, G=None, h=None, solver="quadprog", verbose=False, **kwargs):
    """The one qpsolvers entry point the reference uses (stm.py:271), on the restated Goldfarb-Idnani method."""
    n = len(np.asarray(q).ravel())
    if G is None:
        G, h = np.zeros((0, n)), np.zeros(0)
    return solve_qp_goldfarb_idnani(P, q, G, h)


def nnls_weights(P, q):
    """The same minimiser in its non-negative least-squares form (w = -x = argmin_{w >= 0} || R w - R^-T q ||, P = R'R):
    what the device solver (stm_spectral.h, Lawson-Hanson) computes; kept here as the cross-check of the two forms."""
    from scipy.optimize import nnls
    R = np.linalg.cholesky(0.5 * (P + P.T)).T
    return nnls(R, np.linalg.solve(R.T, q))[0]


def recover_l2_weights(Q, anchor, rows=None):
    """stm.py:239-285: P = M M', and per term i the QP min 1/2 x'Px + (M y_i)'x, x <= 0, weights[i] = -x; one-hot rows
    for the anchor terms (stm.py:261-264).  `rows`: only these terms (the others stay zero)."""
    anchor = np.intp(anchor)
    M = Q[anchor]
    P = M @ M.T
    Vk, K = Q.shape[0], len(anchor)
    G, h = np.eye(K), np.zeros(K)
    L = np.linalg.cholesky(P)
    Pinv = np.linalg.solve(L.T, np.linalg.solve(L, np.eye(K)))
    Pinv = 0.5 * (Pinv + Pinv.T)
    weights = np.zeros((Vk, K))
    for i in (range(Vk) if rows is None else rows):
        hit = np.where(anchor == i)[0]
        if len(hit):
            weights[i, hit] = 1
        else:
            weights[i] = -solve_qp_goldfarb_idnani(P, M @ Q[i], G, h, Pinv=Pinv)
    return weights


def recover_l2(Q, anchor, wprob):
    """stm.py:229-296."""
    weights = recover_l2_weights(Q, anchor)
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
