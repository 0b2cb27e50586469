#!/usr/bin/env python3
"""Per-document shader-clock breakdown of the solver kernel (STM_DEBUG_PROF=1): init / evaluations /
state machine / BFGS update, averaged over a 20k-document corpus at EM iteration 0 and 1."""
import ctypes as C, os, sys
os.environ["STM_DEBUG_PROF"] = "1"
os.environ.setdefault("STM_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "strutopy_amd", "libstm_hip_testing.so"))   # debug switches: the -DSTM_TESTING build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd import STM, _lib
from strutopy_amd.corpus import synthetic_corpus
ND, VV, KK = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (20000, 10000, 50)))   # docs V K [iterations [first printed]]
ITS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
FIRST = int(sys.argv[5]) if len(sys.argv) > 5 else 0
syn = synthetic_corpus(ND, VV, KK, n_words=150, seed=12345)
LEVELS = int(os.environ.get("PROF_LEVELS", "1"))   # 2: config 5's content covariate (document level ~ uniform, as bench.py)
if LEVELS > 1:
    aspect = np.random.default_rng(777).integers(0, LEVELS, size=ND).astype(np.int32)
    m = STM(documents=syn.corpus, dictionary=None, content=True, K=KK, X=syn.X, kappa_interactions=True, A=LEVELS, beta_index=aspect,
            max_em_iter=ITS, sigma_prior=0, convergence_threshold=1e-9, init_type="random")
else:
    m = STM(documents=syn.corpus, dictionary=None, content=False, K=KK, X=syn.X, kappa_interactions=False, max_em_iter=ITS,
            sigma_prior=0, convergence_threshold=1e-9, init_type="random")
for it in range(ITS):
    m._em_iteration_resident()
    out = np.zeros((m.N, 48), dtype=np.int64)
    _lib.check(_lib.lib().stm_debug_get_prof(m._engine._h, out.ctypes.data_as(C.POINTER(C.c_longlong))))
    if it < FIRST:
        continue
    d = m.solver_diagnostics()
    tot = out[:, :4].sum(1)
    print(f"it{it}: cycles/doc init {out[:,0].mean():.0f} eval {out[:,1].mean():.0f} sm {out[:,2].mean():.0f} upd {out[:,3].mean():.0f} total {tot.mean():.0f}"
          f" | nfev {d['nfev'].mean():.1f} njev {d['njev'].mean():.1f} nit {d['nit'].mean():.2f} | per eval {out[:,1].mean()/d['nfev'].mean():.0f} sm/eval {out[:,2].mean()/d['nfev'].mean():.0f}"
          f" kernel {m.timings[-1]['kernels']}")
    if "pd_path" in d:
        print("   pd_path counts", np.bincount(d["pd_path"], minlength=3))
        for pth in np.unique(d["pd_path"]):   # what a rung of the PD ladder costs: the post kernel's phases by path
            sel = d["pd_path"] == pth
            print(f"      path {pth}: {sel.sum()} documents, post cycles/doc: assembly {out[sel, 34].mean():.0f}, ladder {out[sel, 35].mean():.0f}, total {out[sel, 32:39].sum(1).mean():.0f}")
    print("   solver set-up cycles/doc (wave 0): gather %.0f, word-count exchange %.0f, lane vectors + g0 %.0f; wave 1 gather incl. slab %.0f" % tuple(out[:, 4:8].mean(0)))
    if KK > 64:   # the K > 64 solver's fused set-up sweep (stm_solver.h, DIRECT; zeros unless the library was built with -DSTM_SWEEP_PROF)
        print("   K > 64 set-up sweep cycles/doc: wait for the tile + store %.0f, next tile's fetch issue %.0f, g0 %.0f, per-word sums %.0f, v %.0f" % tuple(out[:, 40:45].mean(0)))
    else:
        print("   one evaluation on wave 0, cycles/doc: post + barrier 0 %.0f, max/exp %.0f, barrier 1 %.0f, lse + data term %.0f, barrier 2 %.0f" % tuple(out[:, 40:45].mean(0)))
        if os.environ.get("STM_SM_PROF_LIB"):   # a library built with -DSTM_SM_PROF: the same slots hold one DCSRCH step's pieces instead
            print("   DCSRCH steps (S_W1_ITER), cycles/doc: tests %.0f, dcstep %.0f, interval + clip %.0f, cuts + request %.0f" % tuple(out[:, 40:44].mean(0)))
    names = ["INIT_DONE", "OUTER_TOP", "W1_START", "W1_ITER", "W2_START", "W2_FIRST", "W2_TOP", "W2_GOT_G", "W2_GOT_F",
             "ZOOM_TOP", "ZOOM_GOT_F", "ZOOM_GOT_G", "MOMENTS", "ACCEPT", "ACCEPT2", "FINISH"]
    pn = ["prologue", "word tiles", "H assembly", "Cholesky ladder", "bound", "inverse", "nu"]
    print("   post kernel cycles/doc: " + ", ".join(f"{pn[q]} {out[:, 32 + q].mean():.0f}" for q in range(7)) + f", total {out[:, 32:39].sum(1).mean():.0f}")
    if True:
        if KK > 64:
            print("   post tile phases cycles/doc: gather %.0f sums %.0f scatter %.0f H-acc %.0f" % tuple(out[:, 24:28].mean(0)))
        else:   # K <= 64: no T <- b phase since round 4 (stm_post.h)
            print("   post tile phases cycles/doc: fetch wait %.0f, sums %.0f, b b^T + lane = topic pass %.0f, round end %.0f" % tuple(out[:, 24:28].mean(0)))
    print("   post inverse phases cycles/doc: diag blocks %.0f, MFMA blocks %.0f, remainder row %.0f, pre %.0f" % tuple(out[:, 28:32].mean(0)))
    if KK > 64:   # post_any_kernel reuses the last two slots (and slot 23) for its Cholesky
        print("   post_big Cholesky cycles/doc: block column updates %.0f, panel loads %.0f, panels %.0f" % (out[:, 30].mean(), out[:, 23].mean(), out[:, 31].mean()))
    for i, nm in enumerate(names):
        c, v = (out[:, 8 + i] & ((1 << 40) - 1)).mean(), (out[:, 8 + i] >> 40).mean()
        if v > 0:
            print(f"      {nm:12s} visits/doc {v:6.2f}  cycles/visit {c / v:8.0f}  cycles/doc {c:9.0f}")
