#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot C-ABI entry point stm_estep_host (host buffers in, host buffers
out: handle creation, corpus / beta / mu / eta upload, E-step, download of eta, theta, bounds and the
sufficient statistics) at BASELINE configs[1].  DESIGN.md section 6 quotes this next to bench.py's resident rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd.corpus import synthetic_corpus
from strutopy_amd.engine import estep_host

N, V, K = 100_000, 10_000, 50
c = synthetic_corpus(N, V, K, n_words=150, seed=12345).corpus
rs = np.random.RandomState(123456)
beta = rs.gamma(0.1, 1, c.V * K).reshape(K, c.V); beta /= beta.sum(axis=1)[:, None]
n = K - 1
z = np.zeros((N, n))
siginv, sigent = np.eye(n) / 20.0, float(n * 0.5 * np.log(20.0))
for rep in range(3):
    t = time.perf_counter()
    o = estep_host(c.indptr, c.indices, c.counts, beta, z, z, siginv, sigent)
    dt = time.perf_counter() - t
    print(f"rep {rep}: stm_estep_host {dt * 1e3:.1f} ms  -> {N / dt:,.0f} docs/s (PCIe + setup inclusive), bound {o['bound']!r}")
