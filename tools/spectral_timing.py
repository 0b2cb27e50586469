#!/usr/bin/env python3
"""Seconds of the spectral initialisation (stm.py:30-296) on a configs[1]-shaped corpus, by stage."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd.corpus import synthetic_corpus
from strutopy_amd.engine import HipEstepEngine
from strutopy_amd.spectral import kept_terms, spectral_init
N, V, K = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 10000, 50)))
c = synthetic_corpus(N, V, K, n_words=150, seed=12345).corpus
e = HipEstepEngine(0)
t = time.time(); e.set_corpus(c.indptr, c.indices, c.counts, c.V); print(f"set_corpus {time.time() - t:.3f} s")
for rep in range(2):
    t0 = time.time(); wprob, keep = kept_terms(c, 5000); t1 = time.time()
    e.spectral_gram_resident(keep); t2 = time.time()
    a = e.spectral_anchors(K); t3 = time.time()
    w = e.spectral_weights(a); t4 = time.time()
    e.spectral_release()
    print(f"kept_terms {t1 - t0:.3f}  gram (resident) {t2 - t1:.3f}  anchors {t3 - t2:.3f}  weights {t4 - t3:.3f}  total {t4 - t0:.3f} s")
t = time.time(); beta = spectral_init(c, K, c.V, verbose=False, engine=e, resident=True); print(f"spectral_init end to end {time.time() - t:.3f} s")
e.close()
