#!/usr/bin/env python3
"""How many documents does the solver really keep in flight?  (STM_DEBUG_PROF=1; profile slots 45-47: a document's absolute begin /
end on the shader clock and the 100 MHz wall clock at its end.)  Sum of the documents' in-kernel spans / the launch's span = mean
concurrency; against the 4 documents x 256 CUs the registers allow, the rest is what the dispatch of one workgroup per document
(and the launch's tail) costs.     python tools/slot_gaps.py [docs V K [iteration]]"""
import ctypes as C, os, sys
os.environ["STM_DEBUG_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM, _lib
from strutopy_amd.corpus import synthetic_corpus
ND, VV, KK = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 10000, 50)))
IT = int(sys.argv[4]) if len(sys.argv) > 4 else 7
syn = synthetic_corpus(ND, VV, KK, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=KK, X=syn.X, kappa_interactions=False, max_em_iter=IT + 1, sigma_prior=0,
        convergence_threshold=1e-12, init_type="random")
for it in range(IT + 1):
    m._em_iteration_resident()
out = np.zeros((m.N, 48), dtype=np.int64)
_lib.check(_lib.lib().stm_debug_get_prof(m._engine._h, out.ctypes.data_as(C.POINTER(C.c_longlong))))
b, e, wb, we = out[:, 45], out[:, 46], out[:, 39], out[:, 47]
span = we.max() - wb.min()                       # 100 MHz ticks
inside = (we - wb).sum()
clk = (e - b).sum() / max(inside, 1) * 0.1       # GHz
print(f"EM iteration {IT}: solver kernel {m.timings[-1]['kernels']['solver']:.3f} ms (event), span of the documents {span / 1e5:.3f} ms (wall clock); shader clock {clk:.3f} GHz")
print(f"   cycles per document inside the kernel {(e - b).mean():.0f} = {(we - wb).mean() / 100:.2f} us; mean documents in flight {inside / span:.1f} (of {4 * 256})")
ev = np.concatenate([np.stack([wb, np.ones_like(wb)], 1), np.stack([we, -np.ones_like(we)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
t = ev[:, 0] - wb.min()
for lo, hi in ((0.0, 0.05), (0.05, 0.5), (0.5, 0.95), (0.95, 1.0)):
    sel = (t >= lo * span) & (t < hi * span)
    dt = np.diff(np.append(t[sel], min(hi * span, t.max())))
    print(f"   {lo:.2f}-{hi:.2f} of the span: {np.sum(conc[sel] * dt) / max(dt.sum(), 1):.1f} in flight")
