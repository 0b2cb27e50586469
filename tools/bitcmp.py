#!/usr/bin/env python3
"""Does a new build of the library leave the fit bit for bit where the old one did?

    python tools/bitcmp.py <lib A> <lib B> [<lib C> ...] [docs] [iterations] [K] [V]      (every library against the first)

Runs the bench corpus' resident EM iterations once per library (each in its own process: STM_LIB_PATH is read at load time) and
compares, per EM iteration, sha256(eta), the ELBO, and the solver's status / nit / nfev per document.  What a change is allowed to
move: a re-scheduling of independent work (ILP) -- nothing; a new outcome-preserving cut -- nfev only.  Prints one line per
iteration and exits 1 when eta, status or nit differ anywhere."""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    nd, its, K, V = (int(a) for a in sys.argv[2:6])
    syn = synthetic_corpus(nd, V, K, n_words=150, seed=12345)
    m = STM(documents=syn.corpus, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False, max_em_iter=its,
            sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    out = []
    for it in range(its):
        m._em_iteration_resident()
        d = m.solver_diagnostics()
        eta = m._engine.get_eta()
        out.append(dict(it=it, eta=hashlib.sha256(np.ascontiguousarray(eta).tobytes()).hexdigest(), bound=float(m.bound).hex(),
                        status=hashlib.sha256(d["status"].tobytes()).hexdigest(), nit=hashlib.sha256(d["nit"].tobytes()).hexdigest(),
                        nfev=float(d["nfev"].mean()), nit_mean=float(d["nit"].mean()), solver_ms=m.timings[-1]["kernels"]["solver"]))
    print("BITCMP " + json.dumps(out))


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child()
    sys.exit(0)
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
nd, its, K, V = (rest + ["100000", "12", "50", "10000"][len(rest):])[:4]
res = []
for lib in libs:
    env = dict(os.environ, STM_LIB_PATH=os.path.abspath(lib))
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", nd, its, K, V], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("BITCMP ")]
    if not line:
        print(o.stdout[-2000:], o.stderr[-2000:])
        sys.exit(2)
    res.append(json.loads(line[0][7:]))
bad = False
for lib, r in zip(libs[1:], res[1:]):
    print(f"== {libs[0]} -> {lib}")
    for a, b in zip(res[0], r):
        same = {k: a[k] == b[k] for k in ("eta", "bound", "status", "nit")}
        bad |= not (same["eta"] and same["status"] and same["nit"])
        if len(libs) == 2 or not all(same.values()):
            print(f"it {a['it']:2d}: eta {'same' if same['eta'] else 'DIFFERS'}  ELBO {'same' if same['bound'] else 'differs'}  status {'same' if same['status'] else 'DIFFERS'}"
                  f"  nit {'same' if same['nit'] else 'DIFFERS'} ({a['nit_mean']:.3f})  nfev {a['nfev']:.3f} -> {b['nfev']:.3f}  solver {a['solver_ms']:.3f} -> {b['solver_ms']:.3f} ms")
    n = len(r)
    print(f"   all {n} iterations: eta / status / nit {'identical' if not bad else 'DIFFER'}; nfev mean {sum(x['nfev'] for x in res[0]) / n:.3f} -> {sum(x['nfev'] for x in r) / n:.3f}")
print("solver ms, mean over the iterations (all / from iteration 5 on):")
for lib, r in zip(libs, res):
    late = [x["solver_ms"] for x in r[5:]] or [float("nan")]
    print(f"   {os.path.basename(lib):28s} {sum(x['solver_ms'] for x in r) / len(r):.3f} / {sum(late) / len(late):.3f}")
sys.exit(1 if bad else 0)
