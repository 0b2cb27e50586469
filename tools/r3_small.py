"""mid-size bring-up check: several documents per workgroup, against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import stm_oracle
from strutopy_amd.engine import estep_host
from strutopy_amd.corpus import synthetic_corpus
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
V = 10000
syn = synthetic_corpus(N, V, K, n_words=150, seed=5)
c = syn.corpus
rng = np.random.default_rng(0)
beta = rng.gamma(0.1, 1, size=(K, c.V)); beta /= beta.sum(axis=1)[:, None]
n = K - 1
mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
siginv, sigent = stm_oracle.preamble(np.eye(n) * 20.0)
d = estep_host(c.indptr, c.indices, c.counts, beta, mu, eta, siginv, sigent)
print("gpu done", flush=True)
o = stm_oracle.estep(c.indptr, c.indices, c.counts, beta, mu, eta, siginv, sigent, nthreads=0)
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))
print("N", N, "bound rel", abs(d["bound"] - o["bound"]) / abs(o["bound"]), "sigma_ss", rel(d["sigma_ss"], o["sigma_ss"]), "beta_ss", rel(d["beta_ss"], o["beta_ss"]),
      "status eq", np.array_equal(d["status"], o["status"]), "pd eq", np.array_equal(d["pd_path"], o["pd_path"]), "theta", rel(d["theta"], o["theta"]))
