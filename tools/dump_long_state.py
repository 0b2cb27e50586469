#!/usr/bin/env python3
"""Input of tools/make_golden.py long_c5: the state of a config-5-shaped fit (content covariate, A = 2 levels of beta, K = 50, V = 10k; the
4000-document corpus of tests/test_gpu_round2.py's long-run test) at the first EM iteration whose E-step ran in the long-run regime
(mean scipy nit >= 8), for the first ND documents:  python tools/dump_long_state.py [ND]  ->  gpurun_out/c5_long_state.npz
(beta, sigma, eta / mu / CSR / aspect of the documents: data -- the imported reference is teacher-forced on it in the build container)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
ND = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, V, K, A = 4000, 10_000, 50, 2
syn = synthetic_corpus(N, V, K, n_words=150, seed=12345)
c = syn.corpus
aspect = np.random.default_rng(777).integers(0, A, size=N).astype(np.int32)
m = STM(documents=c, dictionary=None, content=True, K=K, X=syn.X, kappa_interactions=True, A=A, beta_index=aspect,
        max_em_iter=200, sigma_prior=0, convergence_threshold=1e-12, init_type="random")
its, mean_nit = 0, 0.0
while its < 120 and mean_nit < 8.0:
    m._em_iteration_resident()
    its += 1
    if its >= 10:
        mean_nit = float(m.solver_diagnostics()["nit"].mean())
assert mean_nit >= 8.0, (its, mean_nit)
# the state the NEXT E-step starts from
beta, mu, eta, sigma = m.beta.copy(), m.mu.copy(), m.eta.copy(), m.sigma.copy()
m._preamble()
siginv, sigent = m.siginv.copy(), float(m.sigmaentropy)
m._estep_device()
d = m.solver_diagnostics()
end = int(c.indptr[ND])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "c5_long_state.npz"), beta=beta, sigma=sigma, siginv=siginv, sigmaentropy=np.float64(sigent),
                    eta=eta[:ND], mu=mu[:ND], indptr=c.indptr[:ND + 1], indices=c.indices[:end], counts=c.counts[:end], aspect=aspect[:ND],
                    X=np.asarray(syn.X)[:ND], K=np.int32(K), V=np.int32(c.V), A=np.int32(A), em_iteration=np.int32(its),
                    gpu_nit=d["nit"][:ND], gpu_status=d["status"][:ND], gpu_eta=m.eta[:ND])
print("saved: EM iteration", its, "mean nit (all docs)", d["nit"].mean(), "first", ND, "docs:", d["nit"][:ND].mean(), "V", c.V)
