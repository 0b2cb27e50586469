#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference.

Runs ONLY in the build container (needs /root/reference; the GPU box has no
reference).  The reference's modules are imported unmodified from
/root/reference/src with two stand-in packages on sys.path (tools/refshim:
gensim, qpsolvers -- absent from this image).  Nothing
of the reference is copied: the outputs are data (inputs + expected outputs).

    python tools/make_golden.py [case ...]

Every .npz records numpy/scipy/sklearn versions: the reference delegates its
solver to scipy (stm.py:960), so goldens are "the reference as imported here".
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import sklearn  # noqa: E402

import modules.stm as ref_stm  # noqa: E402
from modules.generate_docs import CorpusCreation  # noqa: E402
from modules.stm import STM  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
VERS = dict(numpy=np.__version__, scipy=scipy.__version__, sklearn=sklearn.__version__)


class RecSTM(STM):
    """Reference STM with per-document recording wrapped around its own methods."""

    def _rec_reset(self):
        self.rec = dict(status=[], nit=[], nfev=[], njev=[], fun=[], pd_path=[], bound=[],
                        hess=[], chol=[], nu=[])
        self._keep_mats = getattr(self, "_keep_mats", False)

    def optimize_eta(self, eta, mu, word_count, beta_doc):
        res = super().optimize_eta(eta, mu, word_count, beta_doc)
        r = self.rec
        r["status"].append(res.status); r["nit"].append(res.nit)
        r["nfev"].append(res.nfev); r["njev"].append(res.njev); r["fun"].append(res.fun)
        return res

    def make_pd(self, M):
        self._make_pd_calls = getattr(self, "_make_pd_calls", 0) + 1
        return super().make_pd(M)

    def decompose_hessian(self, hess, approx):
        L = super().decompose_hessian(hess, approx)
        if self._keep_mats:
            self.rec["chol"].append(L.copy())
        return L

    def optimize_nu(self, L):
        nu = super().optimize_nu(L)
        if self._keep_mats:
            self.rec["nu"].append(nu.copy())
        return nu

    def lower_bound(self, L, mu, word_count, beta_doc_kv, eta):
        b = super().lower_bound(L, mu, word_count, beta_doc_kv, eta)
        self.rec["bound"].append(float(b))
        return b


# The +1e-5 branch is detected by patching np.fill_diagonal only while inside hessian():
_orig_fill = np.fill_diagonal


def _install_plus_detector(model):
    model._last_plus = False

    def hessian(eta, word_count, beta_doc_kv):
        model._last_plus = False
        calls = {"n": 0}

        def fill(a, val, wrap=False):
            calls["n"] += 1
            return _orig_fill(a, val, wrap)

        np.fill_diagonal = fill
        try:
            # fill_diagonal is called once for the Hessian diagonal, once per make_pd,
            # and once more for the +1e-5 branch.
            STM_hess = STM.hessian
            model._make_pd_calls = 0
            f = STM_hess(model, eta, word_count, beta_doc_kv)
        finally:
            np.fill_diagonal = _orig_fill
        path = 0
        if model._make_pd_calls:
            path = 2 if calls["n"] >= 3 else 1
        model.rec["pd_path"].append(path)
        if model._keep_mats:
            model.rec["hess"].append(f.copy())
        return f

    model.hessian = hessian


def docs_to_csr(docs):
    indptr = np.zeros(len(docs) + 1, dtype=np.int64)
    idx, cnt = [], []
    for i, d in enumerate(docs):
        for w, c in d:
            idx.append(int(w)); cnt.append(float(c))
        indptr[i + 1] = len(idx)
    return indptr, np.asarray(idx, dtype=np.int32), np.asarray(cnt, dtype=np.float64)


def run_em(model, n_iter, keep_beta_ss="full", sample_cols=None):
    """Run n_iter EM iterations on a RecSTM, recording state around every step."""
    out = {}
    for it in range(n_iter):
        model._rec_reset()
        eta0 = model.eta.copy()
        mu0 = model.mu.copy()
        sigma0 = model.sigma.copy()
        t = time.time()
        beta_ss, sigma_ss = model.E_step()
        te = time.time() - t
        p = f"it{it}_"
        out[p + "eta_in"] = eta0
        out[p + "mu_in"] = mu0
        out[p + "sigma_in"] = sigma0
        out[p + "siginv"] = np.asarray(model.siginv)
        out[p + "sigmaentropy"] = np.float64(model.sigmaentropy)
        out[p + "eta"] = model.eta.copy()
        out[p + "theta"] = model.theta.copy()
        out[p + "bound"] = np.float64(model.bound)
        out[p + "bound_doc"] = np.asarray(model.rec["bound"])
        for k in ("status", "nit", "nfev", "njev", "pd_path"):
            out[p + k] = np.asarray(model.rec[k], dtype=np.int32)
        out[p + "fun"] = np.asarray(model.rec["fun"], dtype=np.float64)
        out[p + "sigma_ss"] = sigma_ss.copy()
        if keep_beta_ss == "full":
            out[p + "beta_ss"] = beta_ss.copy()
        else:
            out[p + "beta_ss_rowsum"] = beta_ss.sum(axis=-1)
            out[p + "beta_ss_colsum"] = beta_ss.sum(axis=-2)
            out[p + "beta_ss_cols"] = beta_ss[..., sample_cols].copy()
        out[p + "phi_last"] = np.asarray(model.phi).copy()
        out[p + "estep_seconds"] = np.float64(te)
        if model._keep_mats:
            out[p + "hess"] = np.asarray(model.rec["hess"])
            out[p + "chol"] = np.asarray(model.rec["chol"])
            out[p + "nu"] = np.asarray(model.rec["nu"])
        model.M_step(beta_ss, sigma_ss)
        out[p + "mu_out"] = model.mu.copy()
        out[p + "sigma_out"] = model.sigma.copy()
        if keep_beta_ss == "full":
            out[p + "beta_out"] = np.asarray(model.beta).copy()
        else:
            out[p + "beta_out_cols"] = np.asarray(model.beta)[..., sample_cols].copy()
        if hasattr(model, "gamma"):
            out[p + "gamma"] = np.asarray(model.gamma).copy()
        print(f"    it{it}: bound={model.bound!r} estep={te:.1f}s "
              f"status2={np.mean(np.asarray(model.rec['status']) == 2):.3f} "
              f"nit_mean={np.mean(model.rec['nit']):.2f} pd_path={np.bincount(model.rec['pd_path'], minlength=3)}")
    return out


def make_model(docs, dictionary, K, X, model_type="STM", content=False, interactions=False,
               beta_index=None, A=None, max_em_iter=3, keep_mats=False, mode="ols", sigma_prior=0):
    m = RecSTM(documents=docs, dictionary=dictionary, content=content, K=K, X=X,
               kappa_interactions=interactions, max_em_iter=max_em_iter, sigma_prior=sigma_prior,
               convergence_threshold=1e-5, init_type="random", model_type=model_type,
               beta_index=beta_index, A=A, mode=mode)
    m._keep_mats = keep_mats
    _install_plus_detector(m)
    return m


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    arrs["versions"] = np.asarray(json.dumps(VERS))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


# --------------------------------------------------------------------------
def case_toy_ctm():
    """tests/test_integration.py:14-68 of the reference (K=3, CTM, 2 EM iterations)."""
    np.random.seed(42)
    K, V, N, n_words, level = 3, 200, 50, 50, 1
    gamma = np.random.multivariate_normal(np.random.standard_normal(level),
                                          np.diag(np.full(level, 0.001)), K - 1)
    corpus = CorpusCreation(n_topics=K, n_docs=N, n_words=n_words, V=V, level=level, dgp="STM",
                            gamma=gamma)
    corpus.generate_documents(remove_terms=True)
    corpus.split_corpus(proportion=0.8)
    docs = corpus.train_docs
    np.random.seed(42)
    X = corpus.metadata[:len(docs)]
    m = make_model(docs, corpus.dictionary, K, X, model_type="CTM", max_em_iter=2, keep_mats=True)
    beta0 = m.beta.copy()
    out = run_em(m, 2)
    indptr, idx, cnt = docs_to_csr(docs)
    save("toy_ctm", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(X, dtype=np.float64),
         K=np.int32(K), V=np.int32(len(corpus.dictionary)), beta0=beta0,
         final_bound=np.float64(m.last_bounds[-1]), **out)


def _synthetic(K, V, n_docs, seed):
    np.random.seed(seed)
    c = CorpusCreation(n_topics=K, n_docs=n_docs, n_words=150, V=V, level=1, dgp="STM")
    c.generate_documents(remove_terms=True)
    return c


def case_c1_k10():
    """BASELINE config 0 shape: src/04 recipe, 1k docs x 150 words, V=2k, K=10, prevalence only."""
    K = 10
    c = _synthetic(K, 2000, 1000, 12345)
    docs = c.documents
    m = make_model(docs, c.dictionary, K, c.metadata, max_em_iter=3)
    beta0 = m.beta.copy()
    out = run_em(m, 3)
    indptr, idx, cnt = docs_to_csr(docs)
    save("c1_k10", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(c.metadata, dtype=np.float64),
         K=np.int32(K), V=np.int32(len(c.dictionary)), beta0=beta0, **out)


def case_k50_v10k():
    """BASELINE config 1 shape at 300 documents: K=50, V=10k, 150 words/doc."""
    K = 50
    c = _synthetic(K, 10000, 300, 2024)
    docs = c.documents
    m = make_model(docs, c.dictionary, K, c.metadata, max_em_iter=2)
    V = len(c.dictionary)
    cols = np.linspace(0, V - 1, 64).astype(np.int64)
    beta0_cols = m.beta[:, cols].copy()
    out = run_em(m, 2, keep_beta_ss="summary", sample_cols=cols)
    indptr, idx, cnt = docs_to_csr(docs)
    save("k50_v10k", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(c.metadata, dtype=np.float64),
         K=np.int32(K), V=np.int32(V), sample_cols=cols, beta0_cols=beta0_cols, **out)


def _wiki_case(K, name, teacher_forced_beta=False, corpus_ref=None):
    """src/03_fit_reference_model.py:40-74 config on the shipped wiki BoW corpus; EM its 0-1.
    ELBO[0] is pinned by the shipped src/artifacts/reference_model/<K>/lower_bound.pickle."""
    import pickle

    import pandas as pd
    from scipy.io import mmread
    art = os.path.join(REF, "src", "artifacts")
    M = mmread(os.path.join(art, "wiki_data", "BoW_corpus.mm")).tocsr()
    M.sort_indices()
    docs = []
    for i in range(M.shape[0]):
        sl = slice(M.indptr[i], M.indptr[i + 1])
        docs.append([(int(w), float(v)) for w, v in zip(M.indices[sl], M.data[sl])])
    data = pd.read_csv(os.path.join(art, "wiki_data", "corpus_preproc.csv"))
    xmat = np.array(data.loc[:, ["statistics"]])
    dictionary = {i: str(i) for i in range(M.shape[1])}
    shipped = pickle.load(open(os.path.join(art, "reference_model", str(K), "lower_bound.pickle"), "rb"))
    np.random.seed(12345)
    m = make_model(docs, dictionary, K, xmat, max_em_iter=25)
    V = M.shape[1]
    cols = np.linspace(0, V - 1, 64).astype(np.int64)
    if teacher_forced_beta:   # the beta entering EM iteration 1 in full (the HIP path is teacher-forced there)
        beta_in = {}
        orig = m.M_step

        def m_step(beta_ss, sigma_ss):
            orig(beta_ss, sigma_ss)
            beta_in[len(beta_in) + 1] = np.asarray(m.beta).copy()
        m.M_step = m_step
    out = run_em(m, 2, keep_beta_ss="summary", sample_cols=cols)
    if teacher_forced_beta:
        out["it1_beta_in"] = beta_in[1]
    print("    shipped ELBO[0:2] =", shipped[0], shipped[1])
    if corpus_ref:   # the corpus lives in another fixture (same .mm file)
        corpus = dict(corpus=np.asarray(corpus_ref))
    else:
        indptr, idx, cnt = docs_to_csr(docs)
        corpus = dict(indptr=indptr, indices=idx, counts=cnt)
    save(name, X=np.asarray(xmat, dtype=np.float64),
         K=np.int32(K), V=np.int32(V), sample_cols=cols,
         shipped_lower_bound=np.asarray(shipped, dtype=np.float64), **corpus, **out)


def case_wiki_k50():
    _wiki_case(50, "wiki_k50")


def case_wiki_k70():
    """The second of the two numbers the reference ships for this path: K = 70 on the wiki corpus
    (src/artifacts/reference_model/70/lower_bound.pickle[0] = -868098.47); takes the K > 64 kernels on real data."""
    _wiki_case(70, "wiki_k70", teacher_forced_beta=True, corpus_ref="wiki_k50")


def case_content_a2():
    """BASELINE config 4 shape, toy size: content covariate with A=2 levels, 3-D beta."""
    K, A = 5, 2
    c = _synthetic(K, 300, 120, 777)
    docs = c.documents
    rng = np.random.default_rng(5)
    bidx = rng.integers(0, A, size=len(docs))
    m = make_model(docs, c.dictionary, K, c.metadata, content=True, interactions=True,
                   beta_index=bidx, A=A, max_em_iter=2)
    beta0 = m.beta.copy()
    out = run_em(m, 2)
    indptr, idx, cnt = docs_to_csr(docs)
    save("content_a2", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(c.metadata, dtype=np.float64),
         K=np.int32(K), V=np.int32(len(c.dictionary)), A=np.int32(A), aspect=bidx.astype(np.int32),
         beta0=beta0, **out)


def case_edge():
    """Hand-made ragged corpus + non-trivial state (random mu/eta start, dense Sigma)."""
    rng = np.random.default_rng(99)
    K, V = 6, 300
    docs = []
    docs.append([(5, 1)])                                   # single word, count 1
    docs.append([(17, 50)])                                 # single word, heavy count
    docs.append([(int(w), 1) for w in range(0, 300)])       # every word once (Nd = V = 300)
    docs.append([(int(w), int(rng.integers(1, 4))) for w in sorted(rng.choice(V, 200, replace=False))])
    docs.append([(3, 1000), (4, 1)])                        # huge count
    docs.append([(int(w), int(rng.integers(1, 30))) for w in sorted(rng.choice(V, 65, replace=False))])
    docs.append([(int(w), 1) for w in sorted(rng.choice(V, 64, replace=False))])
    docs.append([(int(w), 2) for w in sorted(rng.choice(V, 63, replace=False))])
    for _ in range(40):
        nd = int(rng.integers(2, 130))
        docs.append([(int(w), int(rng.integers(1, 6))) for w in sorted(rng.choice(V, nd, replace=False))])
    X = rng.integers(0, 2, size=(len(docs), 1))
    dictionary = {i: str(i) for i in range(V)}
    m = make_model(docs, dictionary, K, X, max_em_iter=2, keep_mats=True)
    n = K - 1
    m.mu = rng.normal(0, 0.5, size=(len(docs), n))
    m.eta = rng.normal(0, 0.5, size=(len(docs), n))
    Bm = rng.normal(size=(n, n))
    m.sigma = Bm @ Bm.T + 0.5 * np.eye(n)
    beta0 = m.beta.copy()
    out = run_em(m, 2)
    indptr, idx, cnt = docs_to_csr(docs)
    save("edge", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(X, dtype=np.float64),
         K=np.int32(K), V=np.int32(V), beta0=beta0, **out)


def case_functions():
    """Per-function vectors: f/df at random points (closures of stm.py:920-958 captured by
    intercepting the scipy call), make_pd / decompose_hessian / optimize_nu on crafted matrices."""
    rng = np.random.default_rng(7)
    K = 8
    c = _synthetic(K, 400, 12, 31337)
    docs = c.documents
    m = make_model(docs, c.dictionary, K, c.metadata, max_em_iter=1)
    n = K - 1
    Bm = rng.normal(size=(n, n))
    m.sigma = Bm @ Bm.T + np.eye(n)
    sigobj = np.linalg.cholesky(m.sigma)
    m.siginv = np.linalg.inv(sigobj).T * np.linalg.inv(sigobj)   # stm.py:501 (same expression)
    m.sigmaentropy = np.sum(np.log(np.diag(sigobj)))
    captured = {}

    def fake_minimize(f, x0, args=(), jac=None, method=None):
        captured["f"], captured["df"], captured["args"] = f, jac, args
        raise StopIteration

    real = ref_stm.optimize.minimize
    fvals, gvals, etas, mus, bfgs_x, bfgs_status, bfgs_nit, bfgs_fun = [], [], [], [], [], [], [], []
    # also a DENSE siginv variant (what a non-reference caller could hand in)
    dense = np.linalg.inv(m.sigma)
    fvals_d, gvals_d, bfgs_x_d, bfgs_status_d, bfgs_nit_d = [], [], [], [], []
    for i, d in enumerate(docs):
        arr = np.array(d)
        idx, cnt = arr[:, 0], arr[:, 1]
        bd = m.get_beta(idx, None)
        mu = rng.normal(0, 0.3, n)
        mus.append(mu)
        pts = rng.normal(0, 1.0, size=(4, n))
        pts[0] = 0.0
        etas.append(pts)
        for dense_flag in (False, True):
            keep = m.siginv
            if dense_flag:
                m.siginv = dense
            ref_stm.optimize.minimize = fake_minimize
            try:
                m.optimize_eta(pts[1], mu, cnt, bd)
            except StopIteration:
                pass
            finally:
                ref_stm.optimize.minimize = real
            f, df, args = captured["f"], captured["df"], captured["args"]
            fv = [float(f(p, *args)) for p in pts]
            gv = [np.asarray(df(p, *args)) for p in pts]
            res = STM.optimize_eta(m, pts[1].copy(), mu, cnt, bd)
            if dense_flag:
                fvals_d.append(fv); gvals_d.append(gv); bfgs_x_d.append(res.x)
                bfgs_status_d.append(res.status); bfgs_nit_d.append(res.nit)
            else:
                fvals.append(fv); gvals.append(gv); bfgs_x.append(res.x)
                bfgs_status.append(res.status); bfgs_nit.append(res.nit); bfgs_fun.append(res.fun)
            m.siginv = keep
    # crafted matrices for make_pd / decompose / nu
    mats, names = [], []
    Bm = rng.normal(size=(n, n)); mats.append(Bm @ Bm.T + n * np.eye(n)); names.append("pd")
    S = rng.normal(size=(n, n)); S = S + S.T; mats.append(S); names.append("indefinite")
    Z = np.array([[1.0, -1.0], [-1.0, 1.0]]); Zp = np.zeros((n, n)); Zp[:2, :2] = Z
    Zp[2:, 2:] = np.eye(n - 2); mats.append(Zp); names.append("singular_dd")
    S2 = -np.abs(S); np.fill_diagonal(S2, -3.0); mats.append(S2); names.append("negdiag")
    mk, Ls, nus, mk_in = [], [], [], []
    for Mx in mats:
        a = Mx.copy()
        mk_in.append(Mx.copy())
        mk.append(STM.make_pd(m, a.copy()))
        h = Mx.copy()
        Lx = STM.decompose_hessian(m, h, approx=None)
        Ls.append(Lx); nus.append(STM.optimize_nu(m, Lx))
    indptr, idx, cnt = docs_to_csr(docs)
    save("functions", indptr=indptr, indices=idx, counts=cnt, K=np.int32(K),
         V=np.int32(len(c.dictionary)), beta0=m.beta.copy(), siginv=np.asarray(m.siginv),
         siginv_dense=dense, mus=np.asarray(mus), etas=np.asarray(etas),
         fvals=np.asarray(fvals), gvals=np.asarray(gvals), bfgs_x=np.asarray(bfgs_x),
         bfgs_status=np.asarray(bfgs_status), bfgs_nit=np.asarray(bfgs_nit), bfgs_fun=np.asarray(bfgs_fun),
         fvals_dense=np.asarray(fvals_d), gvals_dense=np.asarray(gvals_d),
         bfgs_x_dense=np.asarray(bfgs_x_d), bfgs_status_dense=np.asarray(bfgs_status_d),
         bfgs_nit_dense=np.asarray(bfgs_nit_d),
         mats=np.asarray(mk_in), mat_names=np.asarray(names), make_pd=np.asarray(mk),
         chol=np.asarray(Ls), nu=np.asarray(nus))


def case_heldout():
    """Held-out likelihood by document completion (src/modules/heldout.py:70-97, caller src/05_train.py:120):
    the C1-shaped corpus cut in half, scored with theta / beta of the fitted c1_k10 golden model."""
    from modules.heldout import cut_in_half, eval_heldout
    g = np.load(os.path.join(OUT, "c1_k10.npz"))
    indptr, idx, cnt = g["indptr"], g["indices"], g["counts"]
    docs = np.zeros(len(indptr) - 1, dtype=np.ndarray)
    for i in range(len(docs)):
        sl = slice(indptr[i], indptr[i + 1])
        docs[i] = list(zip(idx[sl].tolist(), cnt[sl].astype(np.int64).tolist()))
    first, second = cut_in_half(docs)
    theta, beta = g["it2_theta"], g["it2_beta_out"]
    mean = eval_heldout(second, theta, beta)
    per_doc = np.array([eval_heldout([second[i]], theta[i:i + 1], beta) for i in range(len(second))])
    i1, x1, c1 = docs_to_csr(list(first))
    i2, x2, c2 = docs_to_csr(list(second))
    save("heldout", K=np.int32(g["K"]), V=np.int32(g["V"]), theta=theta, beta=beta,
         first_indptr=i1, first_indices=x1, first_counts=c1,
         second_indptr=i2, second_indices=x2, second_counts=c2,
         mean=np.float64(mean), per_doc=per_doc)


def case_k50_late():
    """BASELINE config 1 shape at 2000 documents, LATER EM iterations: from iteration 4 on about half of the
    documents take BFGS steps that move (successful line searches), which iterations 0-1 of the other K=50
    goldens never exercise.  Nine EM iterations of the reference; for iterations 3, 4, 5 and 8 the complete
    E-step input state (beta, eta, mu, siginv) and per-document outputs are kept, so that oracle and HIP path
    can be TEACHER-FORCED (the reference's free-running trajectory is chaotic)."""
    K, keep = 50, (3, 4, 5, 8)
    c = _synthetic(K, 10000, 2000, 4242)
    docs = c.documents
    m = make_model(docs, c.dictionary, K, c.metadata, max_em_iter=9)
    V = len(c.dictionary)
    out = {}
    for it in range(9):
        m._rec_reset()
        p = f"it{it}_"
        if it in keep:
            out[p + "beta_in"] = np.asarray(m.beta).copy()
            out[p + "eta_in"] = m.eta.copy()
            out[p + "mu_in"] = m.mu.copy()
        t = time.time()
        beta_ss, sigma_ss = m.E_step()
        te = time.time() - t
        out[p + "bound"] = np.float64(m.bound)
        out[p + "nit_mean"] = np.float64(np.mean(m.rec["nit"]))
        out[p + "nfev_mean"] = np.float64(np.mean(m.rec["nfev"]))
        if it in keep:
            out[p + "siginv"] = np.asarray(m.siginv)
            out[p + "sigmaentropy"] = np.float64(m.sigmaentropy)
            out[p + "eta"] = m.eta.copy()
            out[p + "theta"] = m.theta.copy()
            out[p + "bound_doc"] = np.asarray(m.rec["bound"])
            for k in ("status", "nit", "nfev", "njev", "pd_path"):
                out[p + k] = np.asarray(m.rec[k], dtype=np.int32)
            out[p + "sigma_ss"] = sigma_ss.copy()
            out[p + "beta_ss_rowsum"] = beta_ss.sum(axis=-1)
            out[p + "beta_ss_colsum"] = beta_ss.sum(axis=-2)
        m.M_step(beta_ss, sigma_ss)
        print(f"    it{it}: bound={m.bound!r} estep={te:.1f}s status2={np.mean(np.asarray(m.rec['status']) == 2):.3f} "
              f"nit_mean={np.mean(m.rec['nit']):.3f} nfev_mean={np.mean(m.rec['nfev']):.1f} "
              f"pd_path={np.bincount(m.rec['pd_path'], minlength=3)}")
    indptr, idx, cnt = docs_to_csr(docs)
    save("k50_late", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(c.metadata, dtype=np.float64),
         K=np.int32(K), V=np.int32(V), kept=np.asarray(keep, dtype=np.int32), **out)


def _spectral_parts(docs, K, V, maxV=5000):
    """The reference's spectral_init (stm.py:30-85), statement by statement, keeping the intermediate results."""
    dtm = ref_stm.create_dtm(corpus=docs)
    wprob = np.sum(dtm, axis=0)
    wprob = wprob / np.sum(wprob)
    wprob = np.array(wprob).flatten()
    keep = np.argsort(-1 * wprob)[:maxV]
    dtm = dtm[:, keep]
    wprob = wprob[keep]
    Q = ref_stm.gram(dtm)
    Q_gram = Q.toarray()
    anchor = ref_stm.fastAnchor(Q, K, verbose=False)
    Q_caller = Q.toarray()                       # fastAnchor rescales one row of the caller's matrix
    beta = ref_stm.recover_l2(Q, anchor, wprob)   # qpsolvers.solve_qp -> tools/refshim -> the restated Goldfarb-Idnani method
    full = ref_stm.spectral_init(docs, K, V, maxV=maxV, verbose=False)
    return dict(wprob=wprob, keep=keep, Q_gram=Q_gram, anchor=np.asarray(anchor), Q_caller=Q_caller,
                beta_kept=np.asarray(beta), beta=np.asarray(full))


def case_spectral_c1():
    """Spectral initialisation (stm.py:30-296; the init src/05_train.py:92 uses) on the C1-shaped corpus, K = 10."""
    g = np.load(os.path.join(OUT, "c1_k10.npz"))
    indptr, idx, cnt = g["indptr"], g["indices"], g["counts"]
    docs = [list(zip(idx[indptr[i]:indptr[i + 1]].tolist(), cnt[indptr[i]:indptr[i + 1]].astype(np.int64).tolist()))
            for i in range(len(indptr) - 1)]
    K, V = 10, int(g["V"])
    r = _spectral_parts(docs, K, V)
    rows = np.linspace(0, V - 1, 48).astype(np.int64)
    save("spectral_c1", K=np.int32(K), V=np.int32(V), corpus=np.asarray("c1_k10"), wprob=r["wprob"], keep=r["keep"],
         anchor=r["anchor"], sample_rows=rows, Q_gram_rows=r["Q_gram"][rows], Q_gram_rowsum=r["Q_gram"].sum(axis=1),
         Q_gram_colsq=(r["Q_gram"] ** 2).sum(axis=0), Q_caller_anchor_rows=r["Q_caller"][np.intp(r["anchor"])],
         beta_kept=r["beta_kept"], beta=r["beta"], qp_solver=np.asarray("Goldfarb-Idnani dual active set (quadprog's algorithm) as restated in oracle/spectral_oracle.py, through tools/refshim/qpsolvers"))


def case_spectral_wiki():
    """Spectral initialisation on the shipped wiki corpus (V = 13852 > maxV = 5000: the frequency cut is exercised),
    K = 50 as in src/03_fit_reference_model.py."""
    g = np.load(os.path.join(OUT, "wiki_k50.npz"))
    indptr, idx, cnt = g["indptr"], g["indices"], g["counts"]
    docs = [list(zip(idx[indptr[i]:indptr[i + 1]].tolist(), cnt[indptr[i]:indptr[i + 1]].tolist()))
            for i in range(len(indptr) - 1)]
    K, V = 50, int(g["V"])
    r = _spectral_parts(docs, K, V)
    Vk = len(r["keep"])
    rows = np.linspace(0, Vk - 1, 24).astype(np.int64)
    cols = np.linspace(0, V - 1, 64).astype(np.int64)
    save("spectral_wiki", K=np.int32(K), V=np.int32(V), corpus=np.asarray("wiki_k50"), wprob=r["wprob"], keep=r["keep"],
         anchor=r["anchor"], sample_rows=rows, Q_gram_rows=r["Q_gram"][rows], Q_gram_rowsum=r["Q_gram"].sum(axis=1),
         Q_gram_colsq=(r["Q_gram"] ** 2).sum(axis=0), beta_rowsum=r["beta"].sum(axis=1), beta_colsum=r["beta"].sum(axis=0),
         sample_cols=cols, beta_cols=r["beta"][:, cols], beta_kept_anchor_cols=r["beta_kept"][:, np.intp(r["anchor"])],
         qp_solver=np.asarray("Goldfarb-Idnani dual active set (quadprog's algorithm) as restated in oracle/spectral_oracle.py, through tools/refshim/qpsolvers"))


def case_k100_v5k():
    """BASELINE config 3's K = 100 against the reference itself (the other K > 64 checks are oracle-only): 400 documents,
    V = 5000, EM iterations 0-2 with the input state of each (beta of iterations 1-2 stored: teacher-forced)."""
    K = 100
    c = _synthetic(K, 5000, 400, 100100)
    docs = c.documents
    m = make_model(docs, c.dictionary, K, c.metadata, max_em_iter=3)
    out = {}
    for it in range(3):
        m._rec_reset()
        p = f"it{it}_"
        if it > 0:
            out[p + "beta_in"] = np.asarray(m.beta).copy()
        out[p + "eta_in"], out[p + "mu_in"] = m.eta.copy(), m.mu.copy()
        beta_ss, sigma_ss = m.E_step()
        out[p + "siginv"], out[p + "sigmaentropy"] = np.asarray(m.siginv), np.float64(m.sigmaentropy)
        out[p + "eta"], out[p + "theta"] = m.eta.copy(), m.theta.copy()
        out[p + "bound"], out[p + "bound_doc"] = np.float64(m.bound), np.asarray(m.rec["bound"])
        for k in ("status", "nit", "nfev", "njev", "pd_path"):
            out[p + k] = np.asarray(m.rec[k], dtype=np.int32)
        out[p + "sigma_ss"] = sigma_ss.copy()
        out[p + "beta_ss_rowsum"], out[p + "beta_ss_colsum"] = beta_ss.sum(axis=-1), beta_ss.sum(axis=-2)
        m.M_step(beta_ss, sigma_ss)
        print(f"    it{it}: bound={m.bound!r} nit_mean={np.mean(m.rec['nit']):.3f} pd_path={np.bincount(m.rec['pd_path'], minlength=3)}")
    indptr, idx, cnt = docs_to_csr(docs)
    save("k100_v5k", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(c.metadata, dtype=np.float64),
         K=np.int32(K), V=np.int32(len(c.dictionary)), **out)


def case_content_k50():
    """BASELINE config 4's shape (K = 50, content covariate with A = 2 levels of beta) against the reference: 300 documents,
    V = 4000, two EM iterations incl. the reference's axis-1 normalisation of the 3-D beta_ss (stm.py:741)."""
    K, A = 50, 2
    c = _synthetic(K, 4000, 300, 50502)
    docs = c.documents
    bidx = np.random.default_rng(11).integers(0, A, size=len(docs))
    m = make_model(docs, c.dictionary, K, c.metadata, content=True, interactions=True, beta_index=bidx, A=A, max_em_iter=2)
    out = {}
    for it in range(2):
        m._rec_reset()
        p = f"it{it}_"
        if it > 0:
            out[p + "beta_in"] = np.asarray(m.beta).copy()
        out[p + "eta_in"], out[p + "mu_in"] = m.eta.copy(), m.mu.copy()
        beta_ss, sigma_ss = m.E_step()
        out[p + "siginv"], out[p + "sigmaentropy"] = np.asarray(m.siginv), np.float64(m.sigmaentropy)
        out[p + "eta"] = m.eta.copy()
        out[p + "bound"], out[p + "bound_doc"] = np.float64(m.bound), np.asarray(m.rec["bound"])
        for k in ("status", "nit", "pd_path"):
            out[p + k] = np.asarray(m.rec[k], dtype=np.int32)
        out[p + "sigma_ss"] = sigma_ss.copy()
        out[p + "beta_ss_rowsum"], out[p + "beta_ss_colsum"] = beta_ss.sum(axis=-1), beta_ss.sum(axis=-2)
        m.M_step(beta_ss, sigma_ss)
        out[p + "sigma_out"] = m.sigma.copy()
        out[p + "beta_out_colsum"] = np.asarray(m.beta).sum(axis=-2)
        out[p + "beta_out_rowsum"] = np.asarray(m.beta).sum(axis=-1)
        print(f"    it{it}: bound={m.bound!r} nit_mean={np.mean(m.rec['nit']):.3f}")
    indptr, idx, cnt = docs_to_csr(docs)
    save("content_k50", indptr=indptr, indices=idx, counts=cnt, X=np.asarray(c.metadata, dtype=np.float64), K=np.int32(K),
         V=np.int32(len(c.dictionary)), A=np.int32(A), aspect=bidx.astype(np.int32), **out)


def case_mstep_modes():
    """The M-step branches beside OLS (stm.py:678-689 mode="ridge" / "lasso", stm.py:721-728 sigma_prior > 0): 300
    documents of the C1 shape (K = 10, V = 2000) with a THREE-level prevalence covariate (one-hot encoded by update_mu,
    stm.py:665-667), two EM iterations of the reference per configuration from the same beta0."""
    K = 10
    c = _synthetic(K, 2000, 300, 4242)
    docs = c.documents
    N = len(docs)
    X = (np.asarray(c.metadata, dtype=np.float64).reshape(N, -1)[:, 0].astype(np.int64) + np.arange(N) % 2).astype(np.float64)
    indptr, idx, cnt = docs_to_csr(docs)
    out = dict(indptr=indptr, indices=idx, counts=cnt, X=X, K=np.int32(K), V=np.int32(len(c.dictionary)))
    for tag, mode, sp in (("ridge", "ridge", 0), ("lasso", "lasso", 0), ("sp05", "ols", 0.5)):
        m = make_model(docs, c.dictionary, K, X, max_em_iter=2, mode=mode, sigma_prior=sp)
        if "beta0" not in out:
            out["beta0"] = m.beta.copy()
        assert np.array_equal(out["beta0"], m.beta)
        r = run_em(m, 2)
        for k, v in r.items():
            if k.split("_", 1)[1] in ("bound", "gamma", "mu_out", "sigma_out", "beta_out", "sigma_ss", "status", "nit", "pd_path", "eta"):
                out[f"{tag}_{k}"] = v
    save("mstep_modes", **out)


def case_long_c5():
    """The LONG-RUN regime, pinned to the reference: config 5's shape (K = 50, content covariate, A = 2
    levels of beta, V = 10k) at the first EM iteration of a device fit whose documents take about ten BFGS iterations each
    (tools/dump_long_state.py -> gpurun_out/c5_long_state.npz: beta, sigma, eta, mu and the CSR of the first 300 documents -- data).
    The imported reference is TEACHER-FORCED on that state for ONE E-step; per document its status / nit / nfev / eta / bound are the
    golden.  The oracle's run on the same state is compared here and the counts of documents where it differs from the reference are
    stored as metadata (`ref_vs_oracle_*`): ten BFGS iterations amplify a last-bit difference into one accepted step more or less
    (DESIGN.md sections 2 and 7; profiles/HISTORY.md section 9), so the tests' bar for the HIP path is the oracle's own distance from the reference, not zero.
    Without the state file the inputs are taken from the existing golden (re-generation on a new scipy)."""
    src = os.path.join(REPO, "gpurun_out", "c5_long_state.npz")
    if not os.path.exists(src):
        src = os.path.join(OUT, "c5_long.npz")
    st = np.load(src)
    K, V, A = int(st["K"]), int(st["V"]), int(st["A"])
    indptr, idx, cnt = st["indptr"], st["indices"], st["counts"]
    N = len(indptr) - 1
    docs = [[(int(idx[q]), int(cnt[q])) for q in range(indptr[d], indptr[d + 1])] for d in range(N)]
    from gensim.corpora.dictionary import Dictionary
    dictionary = Dictionary({i: str(i) for i in range(V)})
    m = make_model(docs, dictionary, K, st["X"], content=True, interactions=True, beta_index=st["aspect"], A=A, max_em_iter=1)
    assert np.asarray(m.beta).shape == st["beta"].shape, (np.asarray(m.beta).shape, st["beta"].shape)
    m.beta, m.sigma = st["beta"].copy(), st["sigma"].copy()
    m.eta, m.mu = st["eta"].copy(), st["mu"].copy()
    m._rec_reset()
    t = time.time()
    beta_ss, sigma_ss = m.E_step()
    te = time.time() - t
    assert np.allclose(np.asarray(m.siginv), st["siginv"], rtol=1e-12, atol=0) and np.isclose(float(m.sigmaentropy), float(st["sigmaentropy"]), rtol=1e-12)
    r = {k: np.asarray(m.rec[k], dtype=np.int32) for k in ("status", "nit", "nfev", "njev", "pd_path")}
    print(f"    reference: {te:.1f}s, mean nit {r['nit'].mean():.2f}, mean nfev {r['nfev'].mean():.1f}, status {np.bincount(r['status'])}, pd_path {np.bincount(r['pd_path'], minlength=3)}")
    # the oracle on the same state: how far a faithful restatement lands from scipy itself in this regime (the tests' bar)
    sys.path.insert(0, REPO)
    from oracle import stm_oracle
    o = stm_oracle.estep(indptr, idx, cnt, st["beta"], st["mu"], st["eta"], np.asarray(m.siginv), float(m.sigmaentropy), aspect=st["aspect"].astype(np.int32), nthreads=1)
    d_nit = int(np.sum(o["nit"] != r["nit"])); d_status = int(np.sum(o["status"] != r["status"]))
    d_eta = float(np.max(np.abs(o["eta"] - m.eta)))
    same = (o["nit"] == r["nit"]) & (o["status"] == r["status"])
    d_eta_same = float(np.max(np.abs(o["eta"] - m.eta)[same])) if same.any() else 0.0
    d_bound = float(abs(o["bound"] - m.bound) / abs(m.bound))
    print(f"    oracle vs reference: nit differs in {d_nit} / {N} documents, status in {d_status}; eta max abs {d_eta:.3e} (documents with equal nit / status: {d_eta_same:.3e}); ELBO rel {d_bound:.3e}")
    # ... and the reference against itself from a start moved by a relative 1e-13: its own sensitivity there
    m2 = make_model(docs, dictionary, K, st["X"], content=True, interactions=True, beta_index=st["aspect"], A=A, max_em_iter=1)
    m2.beta, m2.sigma, m2.eta, m2.mu = st["beta"].copy(), st["sigma"].copy(), st["eta"] * (1.0 + 1e-13), st["mu"].copy()
    m2._rec_reset()
    m2.E_step()
    s_nit = int(np.sum(np.asarray(m2.rec["nit"]) != r["nit"])); s_status = int(np.sum(np.asarray(m2.rec["status"]) != r["status"]))
    s_eta = float(np.max(np.abs(m2.eta - m.eta)))
    print(f"    reference vs reference(eta (1 + 1e-13)): nit differs in {s_nit}, status in {s_status}; eta max abs {s_eta:.3e}")
    meta = dict(ref_self_nit=np.int32(s_nit), ref_self_status=np.int32(s_status), ref_self_eta=np.float64(s_eta))
    for k in ("gpu_nit", "gpu_status", "gpu_eta", "em_iteration"):
        if k in st.files:
            meta[k] = st[k]
    save("c5_long", indptr=indptr, indices=idx, counts=cnt, aspect=st["aspect"].astype(np.int32), X=st["X"], K=np.int32(K), V=np.int32(V), A=np.int32(A),
         beta=st["beta"], sigma=st["sigma"], eta=st["eta"], mu=st["mu"], siginv=np.asarray(m.siginv), sigmaentropy=np.float64(m.sigmaentropy),
         out_eta=m.eta.copy(), out_theta=m.theta.copy(), out_bound=np.float64(m.bound), out_bound_doc=np.asarray(m.rec["bound"]),
         out_sigma_ss=sigma_ss.copy(), out_beta_ss_rowsum=beta_ss.sum(axis=-1), out_beta_ss_colsum=beta_ss.sum(axis=-2),
         **{"out_" + k: v for k, v in r.items()},
         ref_vs_oracle_nit=np.int32(d_nit), ref_vs_oracle_status=np.int32(d_status), ref_vs_oracle_eta=np.float64(d_eta),
         ref_vs_oracle_eta_same_path=np.float64(d_eta_same), ref_vs_oracle_bound_rel=np.float64(d_bound), **meta)


CASES = dict(long_c5=case_long_c5, mstep_modes=case_mstep_modes, toy_ctm=case_toy_ctm, heldout=case_heldout, functions=case_functions, edge=case_edge,
             content_a2=case_content_a2, c1_k10=case_c1_k10, k50_v10k=case_k50_v10k,
             wiki_k50=case_wiki_k50, wiki_k70=case_wiki_k70, k50_late=case_k50_late, spectral_c1=case_spectral_c1, k100_v5k=case_k100_v5k, content_k50=case_content_k50,
             spectral_wiki=case_spectral_wiki)

if __name__ == "__main__":
    import logging
    import warnings
    warnings.filterwarnings("ignore")
    logging.disable(logging.CRITICAL)
    which = sys.argv[1:] or list(CASES)
    for name in which:
        print(f"[{name}]")
        t = time.time()
        CASES[name]()
        print(f"  done in {time.time() - t:.1f}s")
