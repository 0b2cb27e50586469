import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def P(*a): print(*a, flush=True)
from strutopy_amd import engine as E
from oracle import stm_oracle as O
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(G, "toy_ctm.npz"))
ndoc = int(os.environ.get("NDOC", "4"))
indptr = g["indptr"][:ndoc+1]; nnz = int(indptr[-1])
P("create"); e = E.HipEstepEngine(0); P(e.device_info())
e.set_corpus(indptr, g["indices"][:nnz], g["counts"][:nnz], int(g["V"])); P("corpus ok")
e.set_topics(int(g["K"])); P("topics ok")
e.put_beta(g["beta0"]); P("beta ok", np.abs(e.get_beta()-g["beta0"]).max())
e.put_mu(g["it0_mu_in"][:ndoc]); e.put_eta(g["it0_eta_in"][:ndoc]); P("state ok")
t=time.time(); b = e.estep(g["it0_siginv"], float(g["it0_sigmaentropy"])); P("estep ok", b, time.time()-t, e.kernel_ms())
P("eta", e.get_eta()); P("golden", g["it0_eta"][:ndoc]); P(e.get_diagnostics())
P("bound", e.get_bound_docs(), g["it0_bound_doc"][:ndoc])
