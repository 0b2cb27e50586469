"""new vs v1 post kernel on identical resident state: structure of the sigma_ss difference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
syn = synthetic_corpus(N, 10000, 50, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=3,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
for it in range(3):
    eta_in = m.eta.copy()
    res = {}
    for impl in ("1", "0", "0"):
        os.environ["STM_POST_IMPL"] = impl
        m._engine.put_eta(eta_in) if hasattr(m._engine, "put_eta") else None
        bss, sss = m.E_step()
        res.setdefault(impl, []).append((sss.copy(), bss.copy(), m.bound))
    a, b, b2 = res["1"][0][0], res["0"][0][0], res["0"][1][0]
    d = b - a
    print(f"it{it}: max|new-v1| {np.abs(d).max():.3e} rel {np.abs(d).max()/np.abs(a).max():.3e}; new run-to-run {np.abs(b-b2).max():.3e}; "
          f"diag diff mean {np.diag(d).mean():.3e} offdiag mean {(d.sum()-np.trace(d))/(49*48):.3e}; bss rel {np.abs(res['1'][0][1]-res['0'][0][1]).max()/np.abs(res['1'][0][1]).max():.2e}; bound {res['1'][0][2]!r} {res['0'][0][2]!r}")
    blk = np.array([[np.abs(d[16*i:16*i+16, 16*j:16*j+16]).max() for j in range(4)] for i in range(4)])
    print(np.array2string(blk, precision=2))
    m.M_step(bss, sss)
