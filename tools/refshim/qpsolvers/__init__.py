"""Stand-in for `qpsolvers` (absent from this image; tools/make_golden.py only).

`solve_qp` is called by the reference in exactly one place, recover_l2 of the spectral initialisation
(src/modules/stm.py:271), with solver="quadprog", a dense positive definite P = M M^T, G = I and h = 0:
    min 1/2 x^T P x + q^T x   s.t.  x <= 0.
That QP is strictly convex, so its minimiser is unique and independent of the solver; this stand-in returns it
via scipy.optimize.nnls (x = -w, w = argmin_{w >= 0} || R w - R^-T q ||, P = R^T R).  Contains no reference code.
Goldens that depend on it are labelled as such (tests/golden/spectral_*.npz, key `qp_solver`).
"""
import numpy as np


def solve_qp(P, q, G=None, h=None, solver=None, verbose=False, **kwargs):
    from scipy.optimize import nnls
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).ravel()
    if G is None or h is None or not np.array_equal(np.asarray(G), np.eye(len(q))) or np.any(np.asarray(h) != 0):
        raise NotImplementedError("stand-in solve_qp: only the form recover_l2 uses (G = I, h = 0)")
    R = np.linalg.cholesky(P).T
    w, _ = nnls(R, np.linalg.solve(R.T, q))
    return -w
