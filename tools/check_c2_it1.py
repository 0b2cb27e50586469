#!/usr/bin/env python3
"""How many documents of the full-size reference golden (tests/golden/c2_full.npz) take a different scipy status / nit /
PD path in EM iteration 1 (whose inputs differ from the reference's by the rounding of one M-step) -- a tighter look at
what tests/test_gpu_parity.py::test_full_size_against_the_reference_itself bounds by 1e-4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c2_full.npz"))
syn = synthetic_corpus(int(g["n_docs"]), int(g["V_requested"]), int(g["K"]), n_words=int(g["n_words"]), seed=int(g["seed"]))
m = STM(documents=syn.corpus, dictionary=None, content=False, K=int(g["K"]), X=syn.X, kappa_interactions=False, max_em_iter=2,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
for it in range(2):
    beta_ss, sigma_ss = m.E_step()
    d = m.solver_diagnostics()
    p = f"it{it}_"
    print(f"it{it}: bound rel err {abs(m.bound - float(g[p + 'bound'])) / abs(float(g[p + 'bound'])):.2e}; mismatching documents: "
          + ", ".join(f"{k} {int(np.sum(d[k] != g[p + k]))}" for k in ("status", "nit", "pd_path"))
          + f"; eta sample max abs diff {np.max(np.abs(m.eta[g['sample_docs']] - g[p + 'eta_sample'])):.2e}; nfev mean {d['nfev'].mean():.2f}")
    m.M_step(beta_ss, sigma_ss)
