import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import stm_oracle
from strutopy_amd.engine import estep_host
stm_oracle.build()
for K in (3, 17, 50, 64, 65, 80, 100, 128):
  for mode in (0, 1, 2):
    rng = np.random.default_rng(K * 10 + mode)
    V, N = 700, 50
    docs = [np.sort(rng.choice(V, int(rng.integers(1, 120)), replace=False)) for _ in range(N)]
    indptr = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 6, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    sig = (1e6, 1e3, 50.0)[mode]
    siginv, sigent = stm_oracle.preamble(np.eye(n) * sig)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    try:
        o = stm_oracle.estep(*args, nthreads=0)
    except Exception as e:
        print(K, mode, "oracle raises", repr(e)[:100]); 
        try: estep_host(*args); print("  but GPU did not raise")
        except Exception as e2: print("  GPU raises", repr(e2)[:100])
        continue
    d = estep_host(*args)
    ok = all(np.array_equal(d[k], o[k]) for k in ("status", "nit", "pd_path"))
    rel = np.max(np.abs(d["sigma_ss"] - o["sigma_ss"])) / np.max(np.abs(o["sigma_ss"]))
    relb = abs(d["bound"] - o["bound"]) / abs(o["bound"])
    print(K, mode, "pd_path", np.bincount(o["pd_path"], minlength=3), "agree", ok, "sigma rel", rel, "bound rel", relb)
