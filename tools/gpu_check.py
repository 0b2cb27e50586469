#!/usr/bin/env python3
"""Ad-hoc GPU parity/timing probe (development aid; the real checks live in tests/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import stm_oracle as O
from strutopy_amd import engine as E

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

def beta0(K, V):
    np.random.seed(123456)
    b = np.random.gamma(0.1, 1, V * K).reshape(K, V)
    return b / b.sum(1)[:, None]

def cmp(name, its):
    g = np.load(os.path.join(G, name + ".npz"))
    K, V = int(g["K"]), int(g["V"])
    beta = g["beta0"] if "beta0" in g else beta0(K, V)
    aspect = g["aspect"] if "aspect" in g else None
    for it in range(its):
        p = f"it{it}_"
        args = (g["indptr"], g["indices"], g["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"], float(g[p + "sigmaentropy"]))
        o = O.estep(*args, aspect=aspect, nthreads=0)
        t = time.time(); d = E.estep_host(*args, aspect=aspect); dt = time.time() - t
        rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
        print(f"{name} it{it}: bound rel {abs(d['bound']-o['bound'])/abs(o['bound']):.2e} (vs golden {abs(d['bound']-float(g[p+'bound']))/abs(float(g[p+'bound'])):.2e})"
              f" eta {np.abs(d['eta']-o['eta']).max():.2e} bound_doc {np.max(np.abs(d['bound_doc']-o['bound_doc'])/np.abs(o['bound_doc'])):.2e}"
              f" theta {np.abs(d['theta']-o['theta']).max():.2e} sigma_ss {rel(d['sigma_ss'],o['sigma_ss']):.2e} beta_ss {rel(d['beta_ss'],o['beta_ss']):.2e}"
              f" status_eq {np.mean(d['status']==o['status']):.4f} nit_eq {np.mean(d['nit']==o['nit']):.4f} pd_eq {np.mean(d['pd_path']==o['pd_path']):.4f}"
              f" nfev {d['nfev'].mean():.1f}/{o['nfev'].mean():.1f} wall {dt:.2f}s", flush=True)
        rs = o["beta_ss"].sum(-1 if beta.ndim == 2 else 1, keepdims=True)
        beta = np.divide(o["beta_ss"], rs, out=np.zeros_like(o["beta_ss"]), where=rs != 0)

if __name__ == "__main__":
    which = sys.argv[1:] or ["toy_ctm", "edge", "content_a2", "c1_k10", "k50_v10k", "wiki_k50"]
    its = dict(toy_ctm=2, edge=2, content_a2=2, c1_k10=3, k50_v10k=2, wiki_k50=2)
    for w in which:
        try:
            cmp(w, its[w])
        except Exception as e:
            print(w, "FAILED:", repr(e), flush=True)
