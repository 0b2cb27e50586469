import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import reference_beta0
from strutopy_amd.engine import HipEstepEngine
from strutopy_amd.corpus import synthetic_corpus
N = int(sys.argv[1])
syn = synthetic_corpus(N, 10000, 50, n_words=150, seed=12345)
c = syn.corpus
K, n = 50, 49
beta = reference_beta0(K, c.V)
e = HipEstepEngine(0)
e.set_corpus(c.indptr, c.indices, c.counts, c.V)
e.set_topics(K)
e.put_beta(beta)
siginv, sigent = np.eye(n) / 20.0, float(n * 0.5 * np.log(20.0))
print("estep...", flush=True)
bound = e.estep(siginv, sigent)
print("N", N, "bound", bound, "pd", np.bincount(e.get_diagnostics()["pd_path"], minlength=3), flush=True)
bss = e.get_beta_ss()
print("beta_ss colsum rel", float(np.max(np.abs(bss.sum(axis=0) - c.word_counts())) / c.word_counts().max()))
e.close()
wc = c.word_counts()
err = np.abs(bss.sum(axis=0) - wc)
bad = np.nonzero(err > 1e-9 * wc.max())[0]
print("bad words", len(bad), bad[:10], "err", err[bad[:5]], "wc", wc[bad[:5]])
if len(bad):
    # entries of the first bad word
    v = bad[0]
    ent = np.nonzero(c.indices == v)[0]
    print("word", v, "entries", len(ent), "first/last pos", ent[:3], ent[-3:], "max pos", ent.max(), "nnz", len(c.indices))
