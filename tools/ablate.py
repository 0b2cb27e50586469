#!/usr/bin/env python3
"""Post kernel time of one library build: E-steps at EM iteration 0's state, repeated without an M-step (the post kernel's inputs do not
depend on its own outputs, so ablated builds -- tools/ablate.sh -- can be timed on valid inputs).  STM_LIB_PATH selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM
from strutopy_amd.corpus import synthetic_corpus
ND, VV, KK = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 10000, 50)))
syn = synthetic_corpus(ND, VV, KK, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=KK, X=syn.X, kappa_interactions=False, max_em_iter=1,
        sigma_prior=0, convergence_threshold=1e-9, init_type="random")
ms = []
for it in range(6):
    m._estep_device()
    ms.append(m._engine.kernel_ms())
print(os.environ.get("STM_LIB_PATH", "default").split("/")[-1], " post ms:", " ".join(f"{k['post']:.3f}" for k in ms),
      " solver ms:", " ".join(f"{k['solver']:.3f}" for k in ms[1:3]), flush=True)
