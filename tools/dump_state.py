#!/usr/bin/env python3
"""The complete input state of one late E-step of the bench corpus (first documents only) for offline analysis of the solver's
line searches (what scipy does there, which cuts would apply):  python tools/dump_state.py <em_iteration> [docs]  ->  gpurun_out/state_it<N>.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
IT = int(sys.argv[1]); ND = int(sys.argv[2]) if len(sys.argv) > 2 else 600
syn = synthetic_corpus(100000, 10000, 50, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=IT + 2,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
for it in range(IT):
    m._em_iteration_resident()
eng, c = m._engine, syn.corpus
m._preamble()
beta, eta, mu = eng.get_beta(), eng.get_eta(), eng.get_mu()
m._em_iteration_resident()
d = m.solver_diagnostics()
end = int(c.indptr[ND])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"state_it{IT}.npz"), beta=beta, eta=eta[:ND], mu=mu[:ND], siginv_used=m.siginv,
                    indptr=c.indptr[:ND + 1], indices=c.indices[:end], counts=c.counts[:end], nit=d["nit"][:ND], nfev=d["nfev"][:ND],
                    status=d["status"][:ND])
print("saved", IT, "nit hist", np.bincount(d["nit"])[:5], "nfev mean", d["nfev"].mean())
