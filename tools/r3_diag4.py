"""find the documents whose nu differs between the new and the v1 post kernel (EM iteration 0 state, chunks of 10k docs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["STM_DEBUG_DUMP"] = "1"
import numpy as np
from strutopy_amd.engine import HipEstepEngine
from strutopy_amd.corpus import synthetic_corpus, PackedCorpus
from strutopy_amd import STM
N, V, K = 100000, 10000, 50
n = K - 1
syn = synthetic_corpus(N, V, K, n_words=150, seed=12345)
os.environ.pop("STM_DEBUG_DUMP")
m = STM(documents=syn.corpus, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False, max_em_iter=3,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
beta, mu, eta = m.beta.copy(), m.mu.copy(), m.eta.copy()
m._preamble() if hasattr(m, "_preamble") else None
bss, sss = m.E_step()
siginv, sigent = m.siginv.copy(), float(m.sigmaentropy)
os.environ["STM_DEBUG_DUMP"] = "1"
c = syn.corpus
CH = 10000
for s in range(0, N, CH):
    ip = c.indptr[s:s + CH + 1] - c.indptr[s]
    ind = c.indices[c.indptr[s]:c.indptr[s + CH]]; cnt = c.counts[c.indptr[s]:c.indptr[s + CH]]
    out = {}
    for impl in ("1", "0"):
        os.environ["STM_POST_IMPL"] = impl
        e = HipEstepEngine(0)
        e.set_corpus(ip, ind, cnt, c.V); e.set_topics(K)
        e.put_beta(beta); e.put_mu(mu[s:s + CH]); e.put_eta(eta[s:s + CH])
        e.estep(siginv, sigent)
        out[impl] = e.debug_mats() + (e.diagnostics()["pd_path"] if hasattr(e, "diagnostics") else None,)
        e.close()
    for nm, k in (("hess", 0), ("chol", 1), ("nu", 2)):
        a = out["1"][k].reshape(CH, -1); b = out["0"][k].reshape(CH, -1)
        r = np.max(np.abs(a - b), axis=1) / np.max(np.abs(a), axis=1)
        w = np.argsort(r)[-3:][::-1]
        print(f"chunk {s}: {nm} new-vs-v1 median {np.median(r):.1e} max {r.max():.2e} worst docs {[(int(s + i), float('%.1e' % r[i]), int(ip[i + 1] - ip[i])) for i in w]}")
    d = out["0"][2].sum(0) - out["1"][2].sum(0)
    print(f"   chunk sum diff max {np.abs(d).max():.2e}")
