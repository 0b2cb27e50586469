#!/usr/bin/env python3
"""How many documents does the solver really keep in flight?  (STM_DEBUG_PROF=1; profile slots 45-47: a document's absolute begin /
end on the shader clock and the 100 MHz wall clock at its end.)  Sum of the documents' in-kernel spans / the launch's span = mean
concurrency; against the 4 documents x 256 CUs the registers allow, the rest is what the dispatch of one workgroup per document
(and the launch's tail) costs.     python tools/slot_gaps.py [docs V K [iteration]]"""
import ctypes as C, os, sys
os.environ["STM_DEBUG_PROF"] = "1"
os.environ.setdefault("STM_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "strutopy_amd", "libstm_hip_testing.so"))   # debug switches: the -DSTM_TESTING build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM, _lib
from strutopy_amd.corpus import synthetic_corpus
ND, VV, KK = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 10000, 50)))
IT = int(sys.argv[4]) if len(sys.argv) > 4 else 7
syn = synthetic_corpus(ND, VV, KK, n_words=150, seed=12345)
m = STM(documents=syn.corpus, dictionary=None, content=False, K=KK, X=syn.X, kappa_interactions=False, max_em_iter=IT + 1, sigma_prior=0,
        convergence_threshold=1e-12, init_type="random")
prev = None
for it in range(IT + 1):
    if it == IT and it:
        prev = m.solver_diagnostics()["nfev"].copy()
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
dur = (we - wb) / 100.0   # us
print("   document durations, us: " + ", ".join(f"p{q} {np.percentile(dur, q):.1f}" for q in (1, 10, 50, 90, 99, 99.9)) + f", max {dur.max():.1f}")
late = we > wb.min() + 0.95 * span
print(f"   documents that end in the last 5 % of the span: {late.sum()}, mean duration {dur[late].mean():.1f} us, the 20 longest {np.sort(dur[late])[-20:].round(0).tolist()}")
nf = m.solver_diagnostics()["nfev"]
print(f"   nfev: mean {nf.mean():.2f}, p99 {np.percentile(nf, 99):.0f}, max {nf.max()}; documents with nfev >= 12: {(nf >= 12).sum()}, >= 20: {(nf >= 20).sum()}")
if prev is not None:
    heavy_prev, heavy = prev >= 12, nf >= 12
    print(f"   predictability: of the {heavy.sum()} documents with nfev >= 12 now, {(heavy & heavy_prev).sum()} had nfev >= 12 in the iteration before ({heavy_prev.sum()} had it then); "
          f"corr(nfev, previous nfev) = {np.corrcoef(nf, prev)[0, 1]:.3f}; mean duration of the heavy ones {dur[heavy].mean():.1f} us")
    for name, cls in (("nfev <= 2 before", prev <= 2), ("nfev > 2 before", prev > 2)):
        print(f"      {name}: {cls.sum()} documents, now mean {dur[cls].mean():.1f} us, {(dur[cls] > 30).mean() * 100:.0f} % longer than 30 us")
