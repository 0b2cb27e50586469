#!/usr/bin/env python3
"""What scipy's line searches do on late E-steps, and what the solver's outcome-preserving tests would decide there -- on the CPU,
with scipy itself (scalar_search_wolfe1 / scalar_search_wolfe2 driven by a BFGS loop like _minimize_bfgs), every evaluation logged.

    python tools/trace_linesearch.py golden:it8 [docs]                # tests/golden/k50_late.npz, the reference's EM iteration 8
    python tools/trace_linesearch.py gpurun_out/state_it40.npz [docs]  # a state written by tools/dump_state.py on the GPU box

Per (search index k, success) it prints how many searches there are, how many evaluations scipy spends, and how many of them the
moment test of stm_solver.h (S_MOMENTS; restated in tests/test_moment_pass_math.py for k = 0) proves dead -- at the start of the
search and after three DCSRCH evaluations (the general interval [s_x, max(b, 5 s_x, L)], profiles/HISTORY.md 4.1).  A proof for a search
that SUCCEEDS would be a soundness bug: the tool exits with status 1 if it ever sees one.  This is the evidence behind the
percentages quoted in profiles/HISTORY.md; it is not part of the test suite (a minute per 400 documents)."""
import os, sys, warnings
import numpy as np
from scipy.optimize._linesearch import scalar_search_wolfe1, scalar_search_wolfe2
from scipy.special import logsumexp

warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C1, CURV = 1e-4, 0.099
src = sys.argv[1]
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if src.startswith("golden:"):
    g = np.load(os.path.join(ROOT, "tests", "golden", "k50_late.npz"))
    p_ = src.split(":")[1] + "_"
    beta, eta_in, mu_all, siginv = g[p_ + "beta_in"], g[p_ + "eta_in"], g[p_ + "mu_in"], g[p_ + "siginv"]
    indptr, indices, counts, ref_nit = g["indptr"], g["indices"], g["counts"], g[p_ + "nit"]
else:
    g = np.load(src)
    beta, eta_in, mu_all, siginv = g["beta"], g["eta"], g["mu"], g["siginv_used"]
    indptr, indices, counts, ref_nit = g["indptr"], g["indices"], g["counts"], g["nit"]   # nit: the HIP solver's
ND = min(ND, len(indptr) - 1)
K = beta.shape[0]
n = K - 1
sig_lmax = np.abs(siginv).sum(axis=1).max()
isdiag = np.count_nonzero(siginv - np.diag(np.diag(siginv))) == 0


def verdict(x, p, bd, c, N, phi0, old_phi0, derphi0, L):
    """The moment test for the search from x along p (any k); L = the largest step evaluated so far."""
    pt, xt = np.append(p, 0.0), np.append(x, 0.0)
    r = pt.max() - pt.min()
    if not (derphi0 < 0 and r > 0):
        return False
    e = np.exp(xt - xt.max()); th = e / e.sum()
    cm = th @ pt; var0 = th @ (pt - cm) ** 2
    qx = p @ siginv @ p; pp = p @ p
    Lb = sig_lmax * pp + N * 0.25 * r * r
    Lv = ((qx if isdiag else sig_lmax * pp) + N * var0) * (1 + 1e-9)
    S0, S1, S2 = e @ bd, (e * pt) @ bd, (e * pt * pt) @ bd
    m1 = S1 / S0; D1 = c @ m1; D2 = max(0.0, c @ (S2 / S0 - m1 * m1))
    g0p = (bd @ (c / bd.sum(0)))[:-1] @ p
    slope0 = -derphi0
    b = min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0)
    if b < 0:
        b = 1.0
    nv = N * var0
    a0 = ((derphi0 + g0p) - D1) + C1 * slope0; a0tol = 1e-9 * (slope0 + abs(g0p) + abs(D1))
    s0 = CURV * slope0 / Lv; t0 = s0 * r
    Ux = min(Lb, Lv * (1 + t0 + t0 * t0)) if t0 <= 1 else Lb
    sx = CURV * slope0 / Ux; ir = 1 / r

    def hH(sq):
        t = sq * r; et = np.exp(t); eti = 1 / et; small = t < 0.05
        A = t - 0.5 * t * t if small else 1 - eti
        B = t + 0.5 * t * t * et if small else et - 1
        C = t * t * (0.5 - t / 6) if small else (t - 1) + eti
        E = t * t * (0.5 + t / 6 * et) if small else (et - 1) - t
        up1, dn1, up2, dn2 = nv * (A * ir), D2 * (B * ir), nv * ((C * ir) * ir), D2 * ((E * ir) * ir)
        return ((a0 + qx * sq + up1 - dn1) - (a0tol + 1e-9 * (qx * sq + up1 + dn1)),
                (a0 * sq + 0.5 * qx * sq * sq + up2 - dn2) - (a0tol * sq + 1e-9 * (0.5 * qx * sq * sq + up2 + dn2)))

    fm = 1e-9 * max(1, abs(phi0))
    if not (np.isfinite(b) and b > 0 and qx >= 0 and Ux > 0 and sx > 0):
        return False
    hx, Hx = hH(sx)
    low_ok = hx > 0 and Hx >= fm
    if L <= b and (b <= sx or low_ok) and hH(b)[1] >= fm:          # the first step is rejected: that brackets [0, b]
        return True
    if low_ok and b >= sx / 512:                                    # steps below s_x may pass the sufficient-decrease test
        return hH(max(b, 5 * sx, L) * (1 + 1e-6))[1] >= fm
    return False


stat, unsound = {}, 0
for doc in range(ND):
    w = indices[indptr[doc]:indptr[doc + 1]]; c = counts[indptr[doc]:indptr[doc + 1]]
    bd, m, N = beta[:, w], mu_all[doc], float(int(c.sum()))
    g0 = bd @ (c / bd.sum(0))

    def f(e):
        e = np.append(e, 0.0)
        return 0.5 * (e[:-1] - m) @ siginv @ (e[:-1] - m) - (np.dot(c, e.max() + np.log(np.exp(e - e.max()) @ bd)) - N * logsumexp(e))

    def df(e):
        e = np.append(e, 0.0)
        return siginv @ (e[:-1] - m) - (g0 - (N / np.sum(np.exp(e))) * np.exp(e))[:-1]

    x = eta_in[doc].copy(); H = np.eye(n); fk = f(x); gk = df(x); old_old = fk + np.linalg.norm(gk) / 2; k = 0
    while np.abs(gk).max() > 1e-5 and k < n * 200:
        p = -H @ gk
        ev = []
        phi = lambda a: (ev.append(("f", a)), f(x + a * p))[1]
        dphi = lambda a: df(x + a * p) @ p
        a, f1, _ = scalar_search_wolfe1(phi, dphi, fk, old_old, gk @ p, amin=1e-100, amax=1e100)
        w1 = [s for _, s in ev]
        if a is None:
            a, f1, _, _ = scalar_search_wolfe2(phi, dphi, fk, old_old, gk @ p, amax=1e100)
        ok = a is not None
        st = stat.setdefault((min(k, 2), ok), [0, 0, 0, 0, 0])
        st[0] += 1; st[1] += len(ev)
        v0 = verdict(x, p, bd, c, N, fk, old_old, gk @ p, 0.0)
        st[2] += v0
        v3 = False
        if len(w1) >= 3:
            v3 = verdict(x, p, bd, c, N, fk, old_old, gk @ p, max(w1[:3])); st[3] += 1; st[4] += v3
        unsound += ok and (v0 or v3)
        if not ok:
            break
        xn = x + a * p; gn = df(xn); s = xn - x; y = gn - gk
        old_old, fk, x, gk, k = fk, f1, xn, gn, k + 1
        rho_inv = y @ s; rho = 1000.0 if rho_inv == 0 else 1 / rho_inv
        A1 = np.eye(n) - s[:, None] * y[None, :] * rho; A2 = np.eye(n) - y[:, None] * s[None, :] * rho
        H = A1 @ (H @ A2) + rho * s[:, None] * s[None, :]
    if k != ref_nit[doc]:
        print(f"document {doc}: nit {k} here, {int(ref_nit[doc])} in the file")
for key in sorted(stat):
    t = stat[key]
    print(f"{src}: search {key[0]}{'+' if key[0] == 2 else ''} {'succeeds' if key[1] else 'fails   '}: {t[0]:4d} searches, {t[1] / t[0]:5.1f} f-evaluations each (scipy, uncut);"
          f" proven dead at the start: {t[2]:4d}; of the {t[3]} that reach three DCSRCH evaluations: {t[4]}")
print("soundness:", "OK (no successful search is ever 'proven' dead)" if not unsound else f"VIOLATED in {unsound} searches")
sys.exit(1 if unsound else 0)
