import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
syn = synthetic_corpus(100000, 10000, 50, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=50,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
for it in range(46):
    m._em_iteration_resident()
    if it % 5 == 0 or it > 40:
        d = m.solver_diagnostics()
        print(f"it{it}: nfev {d['nfev'].mean():.2f} njev {d['njev'].mean():.2f} nit mean {d['nit'].mean():.3f} hist {np.bincount(d['nit'])[:6]} nfev hist {np.bincount(d['nfev'])[:30:3]} sigma diag min/max {np.diag(m.sigma).min():.4f}/{np.diag(m.sigma).max():.4f} solver {m.timings[-1]['kernels']['solver']:.2f}", flush=True)
