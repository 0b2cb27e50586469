#!/usr/bin/env python3
"""How many of the Hessians that fail hessian()'s PD test (stm.py:1017) are already given away by a 2 x 2 principal minor?
(A cheap certain-failure test would let the PD ladder skip the factorisation whose outcome is known.)
usage: pd_minor_probe.py [docs] [V] [K] [EM iterations] [sample]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
ND, VV, KK, ITS, SAMPLE = (int(a) for a in (sys.argv[1:6] + ["4000", "50000", "100", "4", "600"][len(sys.argv) - 1:]))
syn = synthetic_corpus(ND, VV, KK, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=KK, X=syn.X, kappa_interactions=False, max_em_iter=ITS, sigma_prior=0,
        convergence_threshold=1e-12, init_type="random")
c = syn.corpus
for it in range(ITS):
    beta = m._engine.get_beta() if hasattr(m._engine, "get_beta") else m.beta
    m._em_iteration_resident()
    siginv = np.array(m.siginv)     # (the E-step's: the preamble sets it, the M-step changes sigma only)
    eta = m._engine.get_eta()
    pd = m.solver_diagnostics()["pd_path"]
    if beta.ndim == 3:
        beta = beta[0]
    n = KK - 1
    rng = np.random.default_rng(it)
    docs = rng.choice(ND, size=min(SAMPLE, ND), replace=False)
    fails = caught = caught_margin = 0
    first_fail = []
    for d in docs:
        lo, hi = c.indptr[d], c.indptr[d + 1]
        idx, cnt = c.indices[lo:hi], c.counts[lo:hi]
        e = np.append(eta[d], 0.0)
        ex = np.exp(e)
        th = np.exp(e - e.max()); th /= th.sum()
        a = beta[:, idx] * ex[:, None]
        S = a.sum(0)
        b = a * np.sqrt(cnt) / S
        H = b @ b.T - cnt.sum() * np.outer(th, th)
        H[np.diag_indices(KK)] += -(b * np.sqrt(cnt)).sum(1) + cnt.sum() * th
        H = H[:n, :n] + (siginv if siginv is not None else 0.0)
        try:
            np.linalg.cholesky(H)
            ok = True
        except np.linalg.LinAlgError:
            ok = False
        if not ok:
            fails += 1
            dg = np.diag(H)
            M2 = H * H - np.outer(dg, dg)
            np.fill_diagonal(M2, -1.0)
            if (dg <= 0).any() or (M2 >= 0).any():
                caught += 1
            if (dg <= 0).any() or (H * H >= np.outer(dg, dg) * (1 + 1e-9))[~np.eye(n, dtype=bool)].any():
                caught_margin += 1
            # the column at which a left-looking factorisation fails
            L = np.zeros_like(H)
            for j in range(n):
                t = H[j, j] - L[j, :j] @ L[j, :j]
                if not t > 0:
                    first_fail.append(j); break
                L[j, j] = np.sqrt(t)
                L[j + 1:, j] = (H[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    ff = np.array(first_fail) if first_fail else np.array([0])
    print(f"EM iteration {it}: device pd_path counts {np.bincount(pd, minlength=3)}; sample of {len(docs)}: {fails} fail the PD test, {caught} of them have a"
          f" non-positive diagonal entry or 2 x 2 minor ({caught_margin} with a 1e-9 margin); failing column: mean {ff.mean():.1f}, p10 {np.percentile(ff, 10):.0f}, p50 {np.percentile(ff, 50):.0f}, p90 {np.percentile(ff, 90):.0f} of {n}")
