#!/usr/bin/env python3
"""Where the HOST's time goes in one resident EM iteration (what the GPU waits for at small shards):  python tools/host_gaps.py [docs] [iterations]
Times, per EM iteration (mean over the later ones): the em_begin call (enqueue + the one wait), the M-step algebra up to em_finish, em_finish itself,
the rest of the iteration (Sigma, timings), and the preamble of the next one -- against the iteration's wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
ND = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
ITS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
syn = synthetic_corpus(ND, 10000, 50, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=ITS, sigma_prior=0,
        convergence_threshold=1e-12, init_type="random")
eng = m._engine
T = {k: [] for k in ("preamble", "em_begin", "algebra", "em_finish", "rest", "total")}
marks = {}
ob, of, op = eng.em_begin, eng.em_finish, m._preamble
def em_begin(*a):
    marks["b0"] = time.perf_counter(); r = ob(*a); marks["b1"] = time.perf_counter(); return r
def em_finish(**k):
    marks["f0"] = time.perf_counter(); r = of(**k); marks["f1"] = time.perf_counter(); return r
def preamble():
    marks["p0"] = time.perf_counter(); op(); marks["p1"] = time.perf_counter()
eng.em_begin, eng.em_finish, m._preamble = em_begin, em_finish, preamble
for it in range(ITS):
    t0 = time.perf_counter()
    m._em_iteration_resident()
    t1 = time.perf_counter()
    T["preamble"].append(marks["p1"] - marks["p0"]); T["em_begin"].append(marks["b1"] - marks["b0"]); T["algebra"].append(marks["f0"] - marks["b1"])
    T["em_finish"].append(marks["f1"] - marks["f0"]); T["rest"].append(t1 - marks["f1"]); T["total"].append(t1 - t0)
eng.synchronize()
lo = ITS // 3
print(f"{ND} documents, EM iterations {lo}..{ITS - 1}, microseconds per iteration:")
for k, v in T.items():
    print(f"   {k:10s} {1e6 * np.mean(v[lo:]):8.1f}")
print("   kernels   ", {k: round(1e3 * float(np.mean([t['kernels'][k] for t in m.timings[lo:]])), 1) for k in ("solver", "post", "pass", "estep")})
