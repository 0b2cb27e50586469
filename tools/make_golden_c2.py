#!/usr/bin/env python3
"""Golden vectors of the REFERENCE at BASELINE.json's full single-GPU size (configs[1]: 100k synthetic
documents x 150 words, V=10k, K=50): two EM iterations of the imported reference, E-steps run on
document shards in worker processes (documents are independent inside E_step, stm.py:519-588), the
M-step (stm.py:622-747) on one full-size reference object.

Runs ONLY in the build container (needs /root/reference).  The corpus is the build's own streaming
generator (strutopy_amd.corpus.synthetic_corpus, seed 12345 -- the one bench.py and the full-size GPU
tests use), so the .npz stores only checksums of the inputs plus the reference's outputs:

    python tools/make_golden_c2.py [n_docs] [workers]      # ~15 min on 8 cores at n_docs = 100000

Output: tests/golden/c2_full.npz
"""
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

K, V_REQ, N_WORDS, SEED = 50, 10_000, 150, 12345
_G = {}


def _docs_of(corpus, lo, hi):
    ip, ix, ct = corpus.indptr, corpus.indices, corpus.counts
    return [[(int(w), int(c)) for w, c in zip(ix[ip[d]:ip[d + 1]], ct[ip[d]:ip[d + 1]])] for d in range(lo, hi)]


def _shard(job):
    """Reference E_step on documents [lo, hi) from the given global state."""
    import make_golden as mg   # imports the reference (with the gensim / qpsolvers stand-ins on sys.path)
    lo, hi, beta, mu, eta, sigma = job
    corpus, X = _G["corpus"], _G["X"]
    docs = _docs_of(corpus, lo, hi)
    m = mg.make_model(docs, {i: str(i) for i in range(corpus.V)}, K, X[lo:hi], max_em_iter=1)
    m.beta = beta.copy(); m.mu = mu[lo:hi].copy(); m.eta = eta[lo:hi].copy(); m.sigma = sigma.copy()
    m._rec_reset()
    beta_ss, sigma_ss = m.E_step()
    r = m.rec
    return dict(lo=lo, hi=hi, beta_ss=beta_ss, sigma_ss=sigma_ss, bound=float(m.bound), eta=m.eta.copy(),
                theta=m.theta.copy(), bound_doc=np.asarray(r["bound"]), status=np.asarray(r["status"], np.int8),
                nit=np.asarray(r["nit"], np.int16), nfev=np.asarray(r["nfev"], np.int32),
                pd_path=np.asarray(r["pd_path"], np.int8), siginv=np.asarray(m.siginv),
                sigmaentropy=float(m.sigmaentropy))


def main():
    from strutopy_amd.corpus import synthetic_corpus
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    syn = synthetic_corpus(n_docs, V_REQ, K, n_words=N_WORDS, seed=SEED)
    c = syn.corpus
    _G["corpus"], _G["X"] = c, syn.X
    V, n = c.V, K - 1
    import make_golden as mg
    rs = np.random.RandomState(123456)                     # the reference's init, stm.py:361,425-429
    beta = rs.gamma(0.1, 1, V * K).reshape(K, V); beta = beta / beta.sum(axis=1)[:, None]
    mu, eta, sigma = np.zeros((n_docs, n)), np.zeros((n_docs, n)), np.eye(n) * 20.0   # stm.py:366-399
    cols = np.linspace(0, V - 1, 64).astype(np.int64)
    rows = np.linspace(0, n_docs - 1, min(n_docs, 500)).astype(np.int64)
    out = dict(n_docs=np.int64(n_docs), K=np.int32(K), V=np.int32(V), V_requested=np.int32(V_REQ), n_words=np.int32(N_WORDS),
               seed=np.int64(SEED), nnz=np.int64(c.indptr[-1]), sample_cols=cols, sample_docs=rows,
               checksum_indices=np.int64(np.sum(c.indices.astype(np.int64) * (np.arange(len(c.indices)) % 9973))),
               checksum_counts=np.float64(np.sum(c.counts * (np.arange(len(c.counts)) % 9973))),
               checksum_X=np.float64(syn.X.sum()), beta0_cols=beta[:, cols].copy())
    # the full-size reference object for the M-step (its constructor repeats the seeded beta init)
    full = mg.make_model(_docs_of(c, 0, min(n_docs, 64)), {i: str(i) for i in range(V)}, K, syn.X[:min(n_docs, 64)], max_em_iter=2)
    assert np.array_equal(full.beta, beta), "the seeded random init of the reference moved"
    full.N = n_docs; full.X = syn.X; full.documents = None
    step = (n_docs + 4 * workers - 1) // (4 * workers)
    with mp.get_context("fork").Pool(workers) as pool:
        for it in range(2):
            t = time.time()
            jobs = [(lo, min(n_docs, lo + step), beta, mu, eta, sigma) for lo in range(0, n_docs, step)]
            res = sorted(pool.map(_shard, jobs, chunksize=1), key=lambda r: r["lo"])
            cat = lambda k: np.concatenate([r[k] for r in res])  # noqa: E731
            beta_ss = sum(r["beta_ss"] for r in res); sigma_ss = sum(r["sigma_ss"] for r in res)
            bound_doc = cat("bound_doc")
            eta, theta = cat("eta"), cat("theta")
            p = f"it{it}_"
            out[p + "bound"] = np.float64(np.sum(bound_doc))        # np.sum(list) of stm.py:592 (pairwise)
            out[p + "bound_shard_sum"] = np.float64(sum(r["bound"] for r in res))
            out[p + "bound_doc_sample"] = bound_doc[rows]
            out[p + "status"], out[p + "nit"], out[p + "pd_path"] = cat("status"), cat("nit"), cat("pd_path")
            out[p + "nfev_sum"] = np.int64(cat("nfev").sum())
            out[p + "eta_sample"], out[p + "theta_sample"] = eta[rows], theta[rows]
            out[p + "eta_colsum"], out[p + "theta_colsum"] = eta.sum(axis=0), theta.sum(axis=0)
            out[p + "sigma_ss"] = sigma_ss
            out[p + "beta_ss_rowsum"], out[p + "beta_ss_colsum"] = beta_ss.sum(axis=1), beta_ss.sum(axis=0)
            out[p + "beta_ss_cols"] = beta_ss[:, cols].copy()
            out[p + "siginv"], out[p + "sigmaentropy"] = res[0]["siginv"], np.float64(res[0]["sigmaentropy"])
            # M-step on the full state (stm.py:622-634)
            full.eta, full.mu, full.sigma, full.beta = eta.copy(), mu.copy(), sigma.copy(), beta.copy()
            full.M_step(beta_ss, sigma_ss)
            mu, sigma, beta = np.asarray(full.mu).copy(), np.asarray(full.sigma).copy(), np.asarray(full.beta).copy()
            out[p + "gamma"] = np.asarray(full.gamma).copy()
            out[p + "sigma_out"] = sigma
            out[p + "beta_out_cols"] = beta[:, cols].copy()
            out[p + "mu_sample"] = mu[rows]
            print(f"it{it}: bound={out[p + 'bound']!r} ({time.time() - t:.0f}s) status2={np.mean(out[p + 'status'] == 2):.3f} "
                  f"nit_mean={out[p + 'nit'].mean():.2f} pd_path={np.bincount(out[p + 'pd_path'], minlength=3)}", flush=True)
    mg.save("c2_full" if n_docs == 100_000 else f"c2_full_{n_docs}", **out)


if __name__ == "__main__":
    main()
