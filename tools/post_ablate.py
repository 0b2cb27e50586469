#!/usr/bin/env python3
"""One E-step of the bench corpus with parts of the post kernel disabled (STM_POST_DEBUG bits: 1 phi atomics,
2 b b^T MFMA, 4 nu, 8 Cholesky) -- timing experiment only, the results of such runs are wrong."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
syn = synthetic_corpus(100000, 10000, 50, n_words=150, seed=12345)
for flags in (0, 1, 2, 3, 4, 8):
    os.environ["STM_POST_DEBUG"] = str(flags)
    m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=1,
            sigma_prior=0, convergence_threshold=1e-9, init_type="random")
    m._estep_device(); m._estep_device()
    print("flags", flags, m._engine.kernel_ms())
    m.close()
