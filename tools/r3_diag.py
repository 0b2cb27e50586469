"""sigma_ss of EM iteration 0 at configs[1] against c2_full.npz, for the kernel selected by STM_POST_IMPL."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tests.test_gpu_parity import load_golden, _c2_corpus, _rel
from strutopy_amd import STM
g = load_golden("c2_full")
syn = _c2_corpus(g)
K = int(g["K"])
m = STM(documents=syn.corpus, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False,
        max_em_iter=2, sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
for it in range(2):
    p = f"it{it}_"
    beta_ss, sigma_ss = m.E_step()
    d = m.solver_diagnostics()
    print("impl", os.environ.get("STM_POST_IMPL", "0"), "it", it, "sigma_ss rel", _rel(sigma_ss, g[p + "sigma_ss"]),
          "bound rel", abs(m.bound - float(g[p + "bound"])) / abs(float(g[p + "bound"])),
          "pd mism", int(np.sum(d["pd_path"] != g[p + "pd_path"])), "asym", float(np.max(np.abs(sigma_ss - sigma_ss.T))),
          "kernels", m.timings[-1]["kernels"] if hasattr(m, "timings") and m.timings else None)
    m.M_step(beta_ss, sigma_ss)
