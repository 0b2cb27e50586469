#!/usr/bin/env python3
"""Per-document solver state visits at one EM iteration (STM_DEBUG_PROF) for the first documents:  python tools/dump_prof.py <it> [docs] -> gpurun_out/prof_it<N>.npz"""
import ctypes as C, os, sys
os.environ["STM_DEBUG_PROF"] = "1"
os.environ.setdefault("STM_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "strutopy_amd", "libstm_hip_testing.so"))   # debug switches: the -DSTM_TESTING build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from strutopy_amd import STM, _lib
from strutopy_amd.corpus import synthetic_corpus
IT = int(sys.argv[1]); ND = int(sys.argv[2]) if len(sys.argv) > 2 else 600
syn = synthetic_corpus(100000, 10000, 50, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=IT + 2,
        sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
out = np.zeros((m.N, 48), dtype=np.int64)
for it in range(IT + 1):
    if it == IT:   # reading the counters clears them
        _lib.check(_lib.lib().stm_debug_get_prof(m._engine._h, out.ctypes.data_as(C.POINTER(C.c_longlong))))
    m._em_iteration_resident()
_lib.check(_lib.lib().stm_debug_get_prof(m._engine._h, out.ctypes.data_as(C.POINTER(C.c_longlong))))
d = m.solver_diagnostics()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"prof_it{IT}.npz"), visits=(out[:ND, 8:24] >> 40), cycles=(out[:ND, 8:24] & ((1 << 40) - 1)),
                    nit=d["nit"][:ND], nfev=d["nfev"][:ND], status=d["status"][:ND])
print("saved", IT)
