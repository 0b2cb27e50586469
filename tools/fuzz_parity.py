#!/usr/bin/env python3
"""Randomised parity sweep of the HIP E-step against the CPU oracle (not part of the test suite: a wider net than the fixed
cases of tests/test_gpu_parity.py).  Random K, V, document lengths, counts, eta / mu scales, diagonal or dense siginv, one or
several beta levels; scipy status / nit / PD path must agree exactly, values to the tolerances of DESIGN.md section 7.

    python tools/fuzz_parity.py [n_cases] [seed] [long|warm]  # long: vocabularies up to 9000 and one document using most of it
                                                              # warm: a SECOND E-step per case, started at the first one's eta with beta and
                                                              # mu perturbed like an M-step would -- the regime of EM iterations >= 1, where the
                                                              # first line search mostly cannot succeed and the solver's moment pass decides
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import stm_oracle
from strutopy_amd.engine import estep_host

stm_oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
LONG = len(sys.argv) > 3 and sys.argv[3] == "long"
WARM = len(sys.argv) > 3 and sys.argv[3] == "warm"
fired = [0, 0]
bad = 0
for case in range(n_cases):
    K = int(rng.choice([2, 3, 5, 10, 16, 17, 18, 33, 34, 49, 50, 51, 64, 65, 66, 70, 80, 81, 97, 100, 112, 113, 128, 129, 150, 200, 257]))   # (65..80, 81..112: the two tile pitches of post_big2_kernel; 113+: post_any_kernel; 129+: the general forms, stm_post_any.h)
    V = int(rng.integers(3000, 9000)) if LONG else int(rng.integers(max(K, 40), 4000))
    N = int(rng.integers(1, 80))
    maxlen = int(rng.choice([3, 20, 70, 140, 200, 400, min(V, 1500)]))
    lens = rng.integers(1, min(V, maxlen) + 1, size=N)
    if LONG:
        lens[int(rng.integers(0, N))] = int(rng.integers(V // 2, V))   # beyond the LDS for most K: the global-slab solver variant
    docs = [np.sort(rng.choice(V, int(L), replace=False)) for L in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, int(rng.choice([2, 5, 50, 1000])), size=len(indices)).astype(np.float64)
    A = int(rng.choice([1, 1, 1, 2, 3]))
    conc = float(rng.choice([0.02, 0.1, 1.0]))
    beta = rng.gamma(conc, 1, size=(A, K, V)) + 1e-300
    beta /= beta.sum(axis=2)[:, :, None]
    aspect = rng.integers(0, A, size=N).astype(np.int32) if A > 1 else None
    if A == 1:
        beta = beta[0]
    n = K - 1
    scale = float(rng.choice([0.0, 0.05, 0.5, 2.0]))
    mu = rng.normal(0, scale, size=(N, n)); eta = rng.normal(0, scale, size=(N, n))
    if rng.random() < 0.7:
        sigma = np.eye(n) * float(rng.choice([0.5, 5.0, 20.0, 200.0]))
        if rng.random() < 0.5:
            Bm = rng.normal(size=(n, n)) * 0.3; sigma = sigma + Bm @ Bm.T
        siginv, sigent = stm_oracle.preamble(sigma)           # what stm.py:499-501 produces (diagonal)
    else:
        Bm = rng.normal(size=(n, n)); sigma = Bm @ Bm.T + np.eye(n) * n
        siginv, sigent = np.linalg.inv(sigma), float(0.5 * np.linalg.slogdet(sigma)[1])   # a dense siginv
    if WARM:   # the second E-step of an EM run: warm start, parameters moved a little
        o1 = stm_oracle.estep(indptr, indices, counts, beta, mu, eta, siginv, sigent, aspect=aspect, nthreads=0)
        eta = o1["eta"]
        step = float(rng.choice([0.01, 0.05, 0.2]))
        beta = beta * np.exp(step * rng.standard_normal(beta.shape))
        beta /= beta.sum(axis=-1, keepdims=True)
        mu = mu + step * rng.standard_normal(mu.shape)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    tag = f"case {case}: K={K} V={V} N={N} maxlen={maxlen} A={A} scale={scale} dense={not np.allclose(siginv, np.diag(np.diag(siginv)))}"
    try:
        d = estep_host(*args, aspect=aspect)
        o = stm_oracle.estep(*args, aspect=aspect, nthreads=0)
    except Exception as e:   # both sides should raise alike; report and go on
        print(tag, "EXCEPTION", repr(e)); bad += 1; continue
    msgs = []
    if WARM:   # how often the moment pass decided: documents that stay put with one evaluation on the device
        still = o["nit"] == 0
        fired[0] += int(np.sum(still & (d["nfev"] <= 2))); fired[1] += int(np.sum(still))
    for k in ("status", "nit", "pd_path"):
        if not np.array_equal(d[k], o[k]):
            msgs.append(f"{k} differs in {int(np.sum(d[k] != o[k]))} documents")
    rel = lambda a, b: float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))
    de = float(np.max(np.abs(d["eta"] - o["eta"])))
    db = float(np.max(np.abs(d["bound_doc"] - o["bound_doc"]) / np.maximum(np.abs(o["bound_doc"]), 1.0)))
    ds, dbs = rel(d["sigma_ss"], o["sigma_ss"]), rel(d["beta_ss"], o["beta_ss"])
    note = ""
    if de > 1e-7 or db > 1e-8 or ds > 1e-7 or dbs > 1e-7:
        # long BFGS runs amplify rounding: measure the oracle's own sensitivity to a 1e-13 perturbation of eta0
        # (oracle vs scipy itself differs by as much on such documents) before calling it a mismatch
        o2 = stm_oracle.estep(indptr, indices, counts, beta, mu, eta * (1.0 + 1e-13 * rng.standard_normal(eta.shape)) + 1e-15, siginv, sigent,
                              aspect=aspect, nthreads=0)
        fl_e = float(np.max(np.abs(o2["eta"] - o["eta"])))
        fl_b = float(np.max(np.abs(o2["bound_doc"] - o["bound_doc"]) / np.maximum(np.abs(o["bound_doc"]), 1.0)))
        fl_s = rel(o2["sigma_ss"], o["sigma_ss"])
        note = f" [values off: eta {de:.1e} bound {db:.1e} sigma_ss {ds:.1e} beta_ss {dbs:.1e}; oracle's own sensitivity: eta {fl_e:.1e} bound {fl_b:.1e} sigma_ss {fl_s:.1e}]"
        if de > max(1e-7, 100 * fl_e): msgs.append(f"eta {de:.2e}")
        if db > max(1e-8, 100 * fl_b): msgs.append(f"bound {db:.2e}")
        if ds > max(1e-7, 100 * fl_s): msgs.append(f"sigma_ss {ds:.2e}")
        if dbs > max(1e-7, 100 * max(fl_e, fl_s)): msgs.append(f"beta_ss {dbs:.2e}")
    print(tag, ("OK" if not msgs else "MISMATCH: " + "; ".join(msgs)) + note, f"(nit max {int(o['nit'].max())}, status {np.bincount(o['status'], minlength=3).tolist()})", flush=True)
    bad += bool(msgs)
    if msgs:   # keep the inputs of a disagreeing case for a closer look
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{case}.npz"), indptr=indptr, indices=indices, counts=counts, beta=beta,
                            mu=mu, eta=eta, siginv=siginv, sigent=sigent, aspect=aspect if aspect is not None else np.zeros(0, np.int32),
                            d_pd=d["pd_path"], o_pd=o["pd_path"], d_bound=d["bound_doc"], o_bound=o["bound_doc"], d_sigma_ss=d["sigma_ss"],
                            o_sigma_ss=o["sigma_ss"], d_eta=d["eta"], o_eta=o["eta"])
if WARM:
    print(f"warm starts: {fired[1]} documents did not move, {fired[0]} of them finished within two evaluations (moment pass)")
print(f"{n_cases - bad} of {n_cases} cases agree")
sys.exit(1 if bad else 0)
