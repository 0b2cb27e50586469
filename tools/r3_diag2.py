"""Per-document H / L / nu of the post kernel against the oracle's (same inputs), K = 50: where does rounding differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["STM_DEBUG_DUMP"] = "1"
import numpy as np
from oracle import stm_oracle
from strutopy_amd.engine import HipEstepEngine
from strutopy_amd.corpus import synthetic_corpus
from strutopy_amd import STM
N, V, K = 3000, 10000, 50
syn = synthetic_corpus(N, V, K, n_words=150, seed=7)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False, max_em_iter=3,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
c = syn.corpus
for it in range(3):
    beta, mu, eta, siginv, sigent = m.beta.copy(), m.mu.copy(), m.eta.copy(), None, None
    bss, sss = m.E_step()
    siginv, sigent = m.siginv.copy(), float(m.sigmaentropy)
    e = HipEstepEngine(0)
    e.set_corpus(c.indptr, c.indices, c.counts, c.V); e.set_topics(K)
    e.put_beta(beta); e.put_mu(mu); e.put_eta(eta)
    e.estep(siginv, sigent)
    hess, chol, nu = e.debug_mats()
    eta_out = e.get_eta()
    e.close()
    o = stm_oracle.estep(c.indptr, c.indices, c.counts, beta, mu, eta, siginv, sigent, dump_mats=True, nthreads=0)
    def per_doc(a, b):
        a = a.reshape(N, -1); b = b.reshape(N, -1)
        return np.max(np.abs(a - b), axis=1) / np.max(np.abs(b), axis=1)
    for nm, a, b in (("hess", hess, o["hess"]), ("chol", chol, o["chol"]), ("nu", nu, o["nu"])):
        r = per_doc(a, b)
        print(f"impl {os.environ.get('STM_POST_IMPL','0')} it{it} {nm:5s} median {np.median(r):.2e} p99 {np.percentile(r,99):.2e} max {r.max():.2e} argmax {r.argmax()}")
    # conditioning of the worst nu document
    w = per_doc(nu, o["nu"]).argmax()
    H = o["hess"].reshape(N, K - 1, K - 1)[w]
    ev = np.linalg.eigvalsh(H)
    print("   worst doc cond", ev.max() / ev.min(), "sum nu rel", np.max(np.abs(nu.sum(0) - o["nu"].sum(0))) / np.max(np.abs(o["nu"].sum(0))))
    m.M_step(bss, sss)
