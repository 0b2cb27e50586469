#!/usr/bin/env python3
"""BASELINE configs[3] AS A CORPUS on one GPU: 1M synthetic documents, V = 50k, K = 100 (nnz ~ 150 M, ~ 10 GB resident).

    python tools/c4_corpus_shards.py [docs V K shards]          (defaults 1000000 50000 100 8)

What the 8-GPU run of that configuration does is: cut the corpus with dist.shard_bounds(indptr, 8), run the E-step of each shard
on its own GPU, add the shards' sufficient statistics (one all-reduce).  Everything but the all-reduce can be verified on ONE GPU:
  (1) the whole corpus' E-step (two of them: the cold start and the warm second one) -> bound, sigma_ss, beta_ss, eta, status / nit
  (2) the eight shards one after another, each through its own handle exactly as a rank would hold it
  (3) asserts: every document's eta / status / nit / bound is the SAME BITS in its shard as in the whole corpus (documents are
      independent given beta, mu, siginv: stm.py:519-588); the shards' bound / sigma_ss / beta_ss sums equal the whole corpus'
      to 1e-12 relative (they differ by the order of the fixed-order reductions only); nnz balance within 1 %.
Prints one JSON line (kept as profiles/r05_c4_corpus_shards.json).  Exits 1 on a failed assert."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from strutopy_amd.corpus import synthetic_corpus
from strutopy_amd.dist import shard_bounds
from strutopy_amd.engine import HipEstepEngine


def reference_beta0(K, V):   # the reference's random init (stm.py:361,425-429): numpy legacy RNG seeded with 123456
    rs = np.random.RandomState(123456)
    b = rs.gamma(0.1, 1, V * K).reshape(K, V)
    return b / b.sum(axis=1)[:, None]


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def run(c, beta, K, siginv, sigent, steps=2):
    e = HipEstepEngine(0)
    t = time.perf_counter()
    e.set_corpus(c.indptr, c.indices, c.counts, c.V)
    e.set_topics(K)
    e.put_beta(beta)
    ingest = time.perf_counter() - t
    out = []
    for _ in range(steps):
        t = time.perf_counter()
        bound = e.estep(siginv, sigent)
        wall = time.perf_counter() - t
        d = e.get_diagnostics()
        out.append(dict(bound=bound, sigma_ss=e.get_sigma_ss(), beta_ss=e.get_beta_ss(), eta=e.get_eta(), bd=e.get_bound_docs(),
                        status=d["status"].copy(), nit=d["nit"].copy(), pd=d["pd_path"].copy(), ms=e.kernel_ms(), wall=wall))
    e.close()
    return out, ingest


def main():
    N, V, K, W = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (1_000_000, 50_000, 100, 8)))
    n = K - 1
    t = time.perf_counter()
    syn = synthetic_corpus(N, V, K, n_words=150, seed=12345)
    c = syn.corpus
    gen_s = time.perf_counter() - t
    beta = reference_beta0(K, c.V)
    siginv, sigent = np.eye(n) / 20.0, float(n * 0.5 * np.log(20.0))
    whole, ingest = run(c, beta, K, siginv, sigent)
    cuts = shard_bounds(c.indptr, W)
    nnz = [int(c.indptr[hi] - c.indptr[lo]) for lo, hi in cuts]
    res = {"workload": f"{N} synthetic docs x 150 words, V={V} (effective {c.V}), K={K}: BASELINE configs[3] as ONE corpus on one GPU, cut into {W} nnz-balanced shards",
           "docs": N, "nnz": int(c.nnz), "corpus_gen_s": gen_s, "ingest_s_whole": ingest, "shards": [dict(lo=lo, hi=hi, nnz=z) for (lo, hi), z in zip(cuts, nnz)],
           "nnz_imbalance": max(nnz) / (sum(nnz) / W) - 1.0, "estep": []}
    acc = [dict(bound=0.0, sigma_ss=np.zeros((n, n)), beta_ss=np.zeros((K, c.V)), ms=[]) for _ in whole]
    same = [dict(eta=True, status=True, nit=True, pd=True, bd=True) for _ in whole]
    for lo, hi in cuts:
        sh, _ = run(c.slice(lo, hi), beta, K, siginv, sigent)
        for i, (w, s) in enumerate(zip(whole, sh)):
            acc[i]["bound"] += s["bound"]; acc[i]["sigma_ss"] += s["sigma_ss"]; acc[i]["beta_ss"] += s["beta_ss"]; acc[i]["ms"].append(s["ms"])
            same[i]["eta"] &= np.array_equal(w["eta"][lo:hi], s["eta"]); same[i]["bd"] &= np.array_equal(w["bd"][lo:hi], s["bd"])
            for k in ("status", "nit", "pd"):
                same[i][k] &= np.array_equal(w[k][lo:hi], s[k])
    ok = res["nnz_imbalance"] <= 0.01
    for i, w in enumerate(whole):
        a = acc[i]
        r = dict(estep=i, whole_kernel_ms=w["ms"], whole_wall_ms=1e3 * w["wall"], docs_per_s_whole=N / (w["ms"]["estep"] * 1e-3),
                 shard_kernel_ms_max=max(m["estep"] for m in a["ms"]),
                 bound_whole=w["bound"], bound_rel=abs(a["bound"] - w["bound"]) / abs(w["bound"]),
                 sigma_ss_rel=rel(a["sigma_ss"], w["sigma_ss"]), beta_ss_rel=rel(a["beta_ss"], w["beta_ss"]),
                 beta_ss_colsum_rel=rel(w["beta_ss"].sum(axis=0), c.word_counts()),
                 per_document_bits_equal=same[i], mean_nit=float(w["nit"].mean()), status2_share=float((w["status"] == 2).mean()))
        ok &= r["bound_rel"] <= 1e-12 and r["sigma_ss_rel"] <= 1e-12 and r["beta_ss_rel"] <= 1e-12 and all(same[i].values()) and r["beta_ss_colsum_rel"] <= 1e-11
        res["estep"].append(r)
    res["ok"] = bool(ok)
    print(json.dumps(res))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
