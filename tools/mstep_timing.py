#!/usr/bin/env python3
"""Wall-clock of the pieces of one device-resident EM iteration (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
syn = synthetic_corpus(100000, 10000, 50, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=3,
        sigma_prior=0, convergence_threshold=1e-9, init_type="random")
eng = m._engine
orig = {}
acc = {}
for name in ("estep", "moments", "allreduce_suffstats", "set_mu_regression", "covariance", "get_sigma_ss", "update_beta", "put_covariates"):
    f = getattr(eng, name)
    def wrap(f=f, name=name):
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append(1e3 * (time.perf_counter() - t)); return r
        return g
    setattr(eng, name, wrap())
beta0 = m.beta.copy()
for rep in range(2):
    m.beta = beta0; m.init_mu(); m.init_eta(); m.init_sigma()
    for nm in ("beta", "eta", "mu"): m._push(nm)
    eng.synchronize()
    for it in range(3):
        acc.clear()
        t = time.perf_counter(); m._em_iteration_resident(); tot = 1e3 * (time.perf_counter() - t)
        print(f"rep{rep} it{it}: total {tot:.2f} ms  " + "  ".join(f"{k} {sum(v):.2f}" for k, v in acc.items()))
