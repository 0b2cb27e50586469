#!/usr/bin/env python3
"""Randomised check of the device-resident EM iteration (moments / mu / covariance / beta kernels, stm_mstep.h) against the
host-NumPy M-step that mirrors the reference statement by statement (STM.E_step + STM.M_step): two EM iterations each way on
random corpora, covariates, model types and regression modes.   python tools/fuzz_em.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import PackedCorpus

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for case in range(n_cases):
    K = int(rng.choice([2, 3, 5, 10, 17, 33, 50, 64, 65, 81, 100, 112, 113, 120, 128]))
    V = int(rng.integers(max(K, 60), 1500)); N = int(rng.integers(8, 300))
    lens = rng.integers(1, min(V, 120) + 1, size=N)
    docs = [np.sort(rng.choice(V, int(L), replace=False)) for L in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 6, size=len(indices)).astype(np.float64)
    c = PackedCorpus(indptr, indices, counts, V)
    model_type = str(rng.choice(["STM", "STM", "CTM"]))
    mode = str(rng.choice(["ols", "ols", "ridge", "lasso"]))
    p = int(rng.integers(1, 4))
    X = rng.integers(0, 2, size=(N, p)).astype(np.float64)
    content = bool(rng.random() < 0.25)
    A = int(rng.integers(2, 4)) if content else None
    kw = dict(documents=c, dictionary=None, content=content, K=K, X=X, kappa_interactions=content, max_em_iter=2, sigma_prior=float(rng.choice([0, 0, 0.3])),
              convergence_threshold=1e-12, init_type="random", model_type=model_type, mode=mode)
    if content:
        kw.update(beta_index=rng.integers(0, A, size=N), A=A)
    out = []
    tag = f"case {case}: K={K} V={V} N={N} p={p} {model_type} {mode} content={content}"
    try:
        for resident in (True, False):
            m = STM(**kw)
            m.expectation_maximization(saving=False, resident=resident)
            out.append((np.array(m.last_bounds), np.array(m.beta, dtype=float).copy(), m.sigma.copy(), m.mu.copy(), m.eta.copy()))
            m.close()
    except Exception as e:
        print(tag, "EXCEPTION", repr(e)); bad += 1; continue
    a, b = out
    msgs = []
    if not np.allclose(a[0], b[0], rtol=1e-8): msgs.append(f"bounds {a[0]} vs {b[0]}")
    for nm, i, rt, at in (("beta", 1, 1e-6, 1e-12), ("sigma", 2, 1e-6, 1e-8), ("mu", 3, 1e-6, 1e-8), ("eta", 4, 1e-5, 1e-6)):
        if not np.allclose(a[i], b[i], rtol=rt, atol=at): msgs.append(f"{nm} {np.max(np.abs(a[i] - b[i])):.2e}")
    print(tag, "OK" if not msgs else "MISMATCH: " + "; ".join(msgs), flush=True)
    bad += bool(msgs)
print(f"{n_cases - bad} of {n_cases} cases agree")
