"""Host-side logic of strutopy_amd (no GPU): corpus packing, the synthetic generator, the STM
mirror class' state plumbing / M-step / EM driver -- driven through the `engine=` test hook
with tests/_oracle_engine.OracleEngine standing in for the HIP engine -- against the golden
vectors produced by the reference (tools/make_golden.py)."""
import os
import pickle
import re

import numpy as np
import pytest

from _oracle_engine import OracleEngine
from conftest import ROOT, load_golden

from strutopy_amd.corpus import PackedCorpus, pack_bow, synthetic_corpus
from strutopy_amd.dist import shard_bounds
from strutopy_amd.stm import STM, encode_covariates


def _corpus(g):
    return PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))


def _model(g, model_type, max_em_iter, **kw):
    return STM(documents=_corpus(g), dictionary=None, content=kw.pop("content", False), K=int(g["K"]),
               X=g["X"][:, 0], kappa_interactions=kw.pop("kappa_interactions", False), max_em_iter=max_em_iter,
               sigma_prior=0, convergence_threshold=1e-5, init_type="random", model_type=model_type,
               engine=OracleEngine(), **kw)


# ----------------------------------------------------------------------------- corpus
def test_pack_bow_roundtrip_and_errors():
    docs = [[(3, 2), (7, 1)], [(0, 5)], [(1, 1), (2, 1), (9, 4)]]
    c = pack_bow(docs, V=10)
    assert c.N == 3 and c.nnz == 6 and c.V == 10
    assert c.indptr.tolist() == [0, 2, 3, 6] and c.indices.dtype == np.int32 and c.counts.dtype == np.float64
    assert c.to_bow() == docs
    assert c.word_counts().tolist() == [5, 1, 1, 2, 0, 0, 0, 1, 0, 4]
    assert pack_bow(c) is c
    s = c.slice(1, 3)
    assert s.to_bow() == docs[1:]
    with pytest.raises(IndexError):   # the reference indexes doc_array[:, 0] (stm.py:523)
        pack_bow([[(1, 1)], []])
    with pytest.raises(IndexError):   # beta[:, idx] with idx >= V (stm.py:617)
        pack_bow([[(11, 1)]], V=10)


def test_synthetic_corpus_follows_the_dgp():
    a = synthetic_corpus(300, 500, 6, n_words=40, seed=5)
    b = synthetic_corpus(300, 500, 6, n_words=40, seed=5)
    c = a.corpus
    assert np.array_equal(c.indices, b.corpus.indices) and np.array_equal(c.counts, b.corpus.counts)
    assert c.N == 300 and a.X.shape == (300, 1) and set(np.unique(a.X)) <= {0.0, 1.0}
    per_doc = np.add.reduceat(c.counts, c.indptr[:-1])
    assert np.all(per_doc == 40)                        # Multinomial(n_words, .), generate_docs.py:302
    for i in range(c.N):                                 # unique ids within a document
        ids = c.indices[c.indptr[i]:c.indptr[i + 1]]
        assert len(np.unique(ids)) == len(ids)
    assert c.indices.min() == 0 and c.indices.max() == c.V - 1 and c.V <= 500
    assert len(np.unique(c.indices)) == c.V               # unused terms dropped (generate_docs.py:304-316)
    assert np.allclose(a.beta_true.sum(axis=1), 1.0)
    keep = synthetic_corpus(50, 500, 6, n_words=40, seed=5, remove_terms=False)
    assert keep.corpus.V == 500


def test_shard_bounds_cover_and_balance():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 200, size=1000)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    for world in (1, 2, 3, 8):
        b = shard_bounds(indptr, world)
        assert b[0][0] == 0 and b[-1][1] == 1000 and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        nnz = [indptr[hi] - indptr[lo] for lo, hi in b]
        assert max(nnz) - min(nnz) <= 2 * lens.max()


def test_encode_covariates_matches_update_mu_preparation():
    x = np.array([0, 1, 1, 0])
    assert np.array_equal(encode_covariates(x), x[:, None].astype(float))       # already 0/1: kept (stm.py:665)
    cat = np.array([2, 0, 1, 2])
    enc = encode_covariates(cat)                                                 # one-hot, sorted categories
    assert np.array_equal(enc, np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]))
    assert encode_covariates(None) is None


def test_cut_in_half_matches_reference_halves():
    from strutopy_amd.heldout import cut_in_half
    g = load_golden("heldout")
    full = _corpus(load_golden("c1_k10"))
    first, second = cut_in_half(full)                                   # heldout.py:70-85 on the packed form
    assert np.array_equal(first.indptr, g["first_indptr"]) and np.array_equal(first.indices, g["first_indices"])
    assert np.array_equal(second.indptr, g["second_indptr"]) and np.array_equal(second.counts, g["second_counts"])
    docs = np.array(full.slice(0, 5).to_bow(), dtype=object)
    a, b = cut_in_half(docs)                                            # and on the reference's list form
    assert list(a[0]) == list(docs[0][0::2]) and list(b[3]) == list(docs[3][1::2])


def test_read_mm_roundtrip(tmp_path):
    import scipy.io
    import scipy.sparse
    from strutopy_amd.corpus import read_mm
    c = synthetic_corpus(40, 120, 4, n_words=25, seed=9).corpus
    m = scipy.sparse.csr_matrix((c.counts, c.indices, c.indptr), shape=(c.N, c.V))
    path = str(tmp_path / "bow.mtx")
    scipy.io.mmwrite(path, m.astype(np.int64))
    r = read_mm(path)
    assert r.N == c.N and r.V == c.V and np.array_equal(r.indptr, c.indptr)
    for i in range(c.N):   # same (word, count) sets per document
        sl = slice(c.indptr[i], c.indptr[i + 1])
        assert dict(zip(r.indices[sl].tolist(), r.counts[sl].tolist())) == dict(zip(c.indices[sl].tolist(), c.counts[sl].tolist()))


def test_read_mm_gensim_layout_and_the_shipped_wiki_corpus(tmp_path):
    """gensim's MmCorpus writes a space-padded size line and entries grouped by document; the reference ships
    src/artifacts/wiki_data/BoW_corpus.mm in that layout, and tests/golden/wiki_k50.npz holds its CSR."""
    from strutopy_amd.corpus import read_mm
    path = tmp_path / "bow.mm"
    path.write_text("%%MatrixMarket matrix coordinate real general\n3 6 5                                  \n"
                    "1 2 1\n1 5 3\n2 1 2\n3 6 1\n3 3 4\n")
    c = read_mm(str(path))
    assert c.N == 3 and c.V == 6 and c.indptr.tolist() == [0, 2, 3, 5]
    assert c.indices.tolist() == [1, 4, 0, 5, 2] and c.counts.tolist() == [1, 3, 2, 1, 4]   # file order kept within a document
    path.write_text("%%MatrixMarket matrix coordinate real general\n% a comment\n2 4 3\n2 1 1\n1 4 2\n2 3 5\n")
    c = read_mm(str(path))                                           # not grouped by document: grouped stably
    assert c.indptr.tolist() == [0, 1, 3] and c.indices.tolist() == [3, 0, 2] and c.counts.tolist() == [2, 1, 5]
    path.write_text("%%MatrixMarket matrix coordinate real general\n2 4 1\n1 4 2\n")
    with pytest.raises(IndexError):                                   # document 2 is empty (stm.py:523)
        read_mm(str(path))
    wiki = "/root/reference/src/artifacts/wiki_data/BoW_corpus.mm"   # only in the build container
    if os.path.exists(wiki):
        g = load_golden("wiki_k50")
        w = read_mm(wiki)
        assert w.V == int(g["V"]) and np.array_equal(w.indptr, g["indptr"])
        assert np.array_equal(w.indices, g["indices"]) and np.array_equal(w.counts, g["counts"])


def test_pack_bow_accepts_what_the_reference_accepts():
    """np.array(documents[i]) (stm.py:522) takes tuples, lists or arrays of (id, count), integer or float counts."""
    docs = [[(3, 2.0), (7, 1.0)], [[0, 5]], np.array([[1, 1], [2, 1], [9, 4]])]
    c = pack_bow(docs, V=10)
    assert c.indptr.tolist() == [0, 2, 3, 6] and c.indices.tolist() == [3, 7, 0, 1, 2, 9]
    assert c.counts.tolist() == [2, 1, 5, 1, 1, 4]
    big = synthetic_corpus(2000, 800, 5, n_words=60, seed=2).corpus
    again = pack_bow(big.to_bow(), V=big.V)
    assert np.array_equal(again.indptr, big.indptr) and np.array_equal(again.indices, big.indices)
    assert np.array_equal(again.counts, big.counts)
    with pytest.raises(IndexError):
        pack_bow([[(1.5, 1)]], V=10)
    with pytest.raises(IndexError):
        pack_bow([[(-1, 1)]], V=10)


def test_pack_bow_c_walker_and_iterator_path_agree(monkeypatch):
    """csrc/packbow.c (built by __graft_entry__.build) and the iterator-based NumPy path: same arrays, same errors."""
    import strutopy_amd.corpus as cm
    if cm._packbow_module() is None:
        pytest.skip("strutopy_amd/_packbow is not built")
    big = synthetic_corpus(3000, 900, 5, n_words=80, seed=4).corpus
    bow = big.to_bow()
    bow[5] = [list(p) for p in bow[5]]                     # a document of lists
    bow[6] = np.array(bow[6], dtype=np.float64)            # ... and one as an array, like np.array(documents[i]) takes them
    fast = pack_bow(bow, V=big.V)
    monkeypatch.setattr(cm, "_packbow_module", lambda: None)
    slow = pack_bow(bow, V=big.V)
    for a in ("indptr", "indices", "counts"):
        assert np.array_equal(getattr(fast, a), getattr(slow, a)) and np.array_equal(getattr(fast, a), getattr(big, a))
    monkeypatch.undo()
    for bad in ([[(1, 1)], []], [[(1.5, 1)]], [[(-1, 1)]], [[(1, 1, 1)]], [[3]], [[(2 ** 31, 1)]]):
        with pytest.raises(IndexError):
            pack_bow(bad, V=10)
    with pytest.raises(IndexError):
        pack_bow([[(11, 1)]], V=10)


def test_corpus_as_csr_triple_sparse_matrix_or_matrix_market_path(tmp_path):
    """The forms a caller may already hold: the CSR triple, create_dtm's scipy matrix (stm.py:87-119) and a
    MatrixMarket file go into STM(...) as they are -- no detour through lists of tuples."""
    import scipy.sparse as sp
    from scipy.io import mmwrite
    from _oracle_engine import OracleEngine
    from strutopy_amd.stm import STM
    big = synthetic_corpus(400, 300, 4, n_words=40, seed=9)
    m = sp.csr_matrix((big.corpus.counts, big.corpus.indices, big.corpus.indptr), shape=(big.corpus.N, big.corpus.V))
    m.sort_indices()                                       # (a scipy matrix is taken in canonical form: word ids ascending)
    c = PackedCorpus(m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64), big.corpus.V)
    path = tmp_path / "corpus.mm"
    with open(path, "wb") as fh:
        mmwrite(fh, m.tocoo())
    for form in ((c.indptr, c.indices, c.counts), m, str(path), c.to_bow()):
        p = pack_bow(form, V=c.V)
        assert np.array_equal(p.indptr, c.indptr) and np.array_equal(p.indices, c.indices) and np.array_equal(p.counts, c.counts) and p.V == c.V
        model = STM(documents=form, dictionary={i: str(i) for i in range(c.V)}, content=False, K=4, X=big.X, kappa_interactions=False,
                    max_em_iter=1, sigma_prior=0, convergence_threshold=1e-5, init_type="random", engine=OracleEngine())
        assert model.N == c.N and model.V == c.V
    with pytest.raises(IndexError):
        pack_bow((np.array([0, 1, 1]), np.array([2], dtype=np.int32), np.array([1.0])))         # an empty document
    with pytest.raises(IndexError):
        pack_bow((np.array([0, 1]), np.array([12], dtype=np.int32), np.array([1.0])), V=10)     # word id beyond the dictionary
    # a word id twice in one document: the reference would count its phi column once in beta_ss and twice
    # everywhere else (stm.py:588), the kernels let no two lanes share a word -- rejected, whatever the order of the row;
    # the same id in two documents (also across the boundary) is what a corpus is
    dup = (np.array([0, 2, 5]), np.array([3, 1, 4, 1, 4], dtype=np.int32), np.ones(5))
    with pytest.raises(ValueError, match="document 1 holds word id 4"):
        pack_bow(dup, V=10)
    with pytest.raises(ValueError, match="more than once"):
        pack_bow([[(3, 1), (1, 2)], [(4, 1), (1, 1), (4, 2)]])
    ok = pack_bow((np.array([0, 2, 5]), np.array([3, 1, 1, 4, 3], dtype=np.int32), np.ones(5)), V=10)
    assert ok.N == 2
    # a scipy matrix has its duplicates summed (what csr_matrix means by them) -- on a COPY: the caller's matrix keeps its entries
    raw = sp.csr_matrix((np.ones(5), np.array([3, 1, 4, 1, 4]), np.array([0, 2, 5])), shape=(2, 10))
    summed = pack_bow(raw)
    assert summed.indices.tolist() == [1, 3, 1, 4] and summed.counts.tolist() == [1.0, 1.0, 1.0, 2.0]
    assert raw.indices.tolist() == [3, 1, 4, 1, 4] and raw.nnz == 5


# ----------------------------------------------------------------------------- STM mirror
def test_constructor_state_matches_reference_init():
    g = load_golden("toy_ctm")
    m = _model(g, "CTM", 2)
    K, V, N = int(g["K"]), int(g["V"]), len(g["indptr"]) - 1
    assert m.beta.shape == (K, V) and m.theta.shape == (N, K) and m.eta.shape == (N, K - 1)
    assert m.sigma.shape == (K - 1, K - 1) and np.array_equal(m.sigma, 20 * np.eye(K - 1))   # stm.py:460-461
    assert np.array_equal(m.beta, g["beta0"])             # legacy-RNG random init, stm.py:361,425-429
    assert not m.eta.any() and not m.mu.any() and not m.theta.any()
    assert np.array_equal(m.wcounts, _corpus(g).word_counts())


@pytest.mark.parametrize("resident", [False, True])
def test_toy_pipeline_reproduces_reference_trace(resident):
    """The reference's own integration pipeline (tests/test_integration.py:14-68): CTM, 2 EM its."""
    g = load_golden("toy_ctm")
    m = _model(g, "CTM", 2)
    m.expectation_maximization(saving=False, resident=resident)
    assert len(m.last_bounds) == 2
    assert m.last_bounds[0] == pytest.approx(float(g["it0_bound"]), rel=1e-10)
    assert m.bound == pytest.approx(float(g["final_bound"]), rel=1e-9)
    assert np.allclose(m.mu, g["it1_mu_out"], rtol=0, atol=1e-7)
    assert np.allclose(m.sigma, g["it1_sigma_out"], rtol=1e-7, atol=1e-10)
    assert np.allclose(m.beta, g["it1_beta_out"], rtol=1e-7, atol=1e-12)
    assert np.allclose(m.theta.sum(axis=1), 1.0, atol=1e-4) and np.allclose(m.beta.sum(axis=1), 1.0, atol=1e-4)


def test_estep_mstep_surface_one_iteration():
    """E_step() -> (beta_ss, sigma_ss); M_step(beta_ss, sigma_ss) -- the reference call pattern (stm.py:861-863)."""
    g = load_golden("c1_k10")
    m = _model(g, "STM", 3)
    beta_ss, sigma_ss = m.E_step()
    assert beta_ss.shape == m.beta.shape and sigma_ss.shape == m.sigma.shape
    assert m.bound == pytest.approx(float(g["it0_bound"]), rel=1e-10) and m.last_bounds == [m.bound]
    assert np.allclose(beta_ss, g["it0_beta_ss"], rtol=1e-8, atol=1e-12)
    assert np.allclose(m.eta, g["it0_eta"], atol=1e-7) and np.allclose(m.theta, g["it0_theta"], atol=1e-7)
    assert np.allclose(m.phi, g["it0_phi_last"], rtol=1e-8)              # self.phi = last document's phi
    assert np.allclose(m.siginv, g["it0_siginv"]) and m.sigmaentropy == pytest.approx(float(g["it0_sigmaentropy"]))
    m.M_step(beta_ss, sigma_ss)
    assert np.allclose(m.gamma, g["it0_gamma"], rtol=1e-6, atol=1e-9)    # coef_ only, intercept dropped (stm.py:703)
    assert np.allclose(m.mu, g["it0_mu_out"], atol=1e-8)
    assert np.allclose(m.sigma, g["it0_sigma_out"], rtol=1e-7, atol=1e-10)
    assert np.allclose(m.beta, g["it0_beta_out"], rtol=1e-8, atol=1e-14)
    d = m.solver_diagnostics()
    assert np.array_equal(d["status"], g["it0_status"]) and np.array_equal(d["nit"], g["it0_nit"])


def test_resident_em_three_iterations_stm_ols():
    g = load_golden("c1_k10")
    m = _model(g, "STM", 3)
    m.fit(saving=False)   # north_star's name for expectation_maximization
    assert len(m.last_bounds) == 3
    for it in range(3):
        assert m.last_bounds[it] == pytest.approx(float(g[f"it{it}_bound"]), rel=1e-8)
    # three chained iterations: the 1e-9-level eta differences of A.5 compound (measured: gamma/mu 3e-8,
    # sigma 5e-9, beta 3e-10 absolute, run-to-run varying with the OpenMP accumulation order)
    assert np.allclose(m.gamma, g["it2_gamma"], rtol=1e-5, atol=1e-6)
    assert np.allclose(m.mu, g["it2_mu_out"], atol=1e-6)
    assert np.allclose(m.sigma, g["it2_sigma_out"], rtol=1e-6, atol=1e-7)
    assert np.allclose(m.beta, g["it2_beta_out"], rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("resident", [False, True])
def test_content_covariate_levels(resident):
    """kappa_interactions + content: per-level beta (stm.py:430-431,614-615) and the reference's
    axis=1 normalisation of the 3-D beta_ss (stm.py:741)."""
    g = load_golden("content_a2")
    m = _model(g, "STM", 2, content=True, kappa_interactions=True, A=int(g["A"]), beta_index=g["aspect"])
    assert m.beta.shape == (2, int(g["K"]), int(g["V"])) and np.array_equal(m.beta, g["beta0"])
    m.expectation_maximization(saving=False, resident=resident)
    assert m.last_bounds[0] == pytest.approx(float(g["it0_bound"]), rel=1e-10)
    assert m.last_bounds[1] == pytest.approx(float(g["it1_bound"]), rel=1e-8)
    assert np.allclose(m.beta, g["it1_beta_out"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("tag,mode,sp", __import__("_mstep_modes").CONFIGS)
@pytest.mark.parametrize("resident", [False, True])
def test_mstep_branches_against_the_reference(tag, mode, sp, resident):
    """mode="ridge" / "lasso" and sigma_prior > 0 (stm.py:678-689, 721-728) against the reference's own two EM iterations;
    resident=True takes the moment-based ridge solve / lasso coordinate descent / shrinkage of the device loop."""
    import _mstep_modes
    _mstep_modes.run(load_golden("mstep_modes"), tag, mode, sp, resident, engine=OracleEngine())


def test_covariance_from_moments_and_its_guard():
    """One all-reduce: (eta - mu)^T (eta - mu) expanded in the reduced moments equals the direct product; when the
    expansion would cancel more than four digits the resident loop takes the exact (second all-reduce) form."""
    g = load_golden("c1_k10")
    m = _model(g, "STM", 2)
    rng = np.random.default_rng(0)
    N, n, p = 500, 9, 3
    X = rng.integers(0, 2, size=(N, p)).astype(float)
    eta = rng.normal(0, 1, size=(N, n)) + X @ rng.normal(size=(p, n))
    m.gamma = rng.normal(size=(n, p))
    cov, ratio = m._covariance_from_moments(eta.T @ eta, N, eta.sum(0), X.T @ X, X.T @ eta)
    d = eta - X @ m.gamma.T
    assert np.allclose(cov, d.T @ d, rtol=1e-11, atol=1e-9) and np.allclose(cov, cov.T, rtol=0, atol=1e-10) and 0 < ratio
    cov, ratio = m._covariance_from_moments(eta.T @ eta, N, eta.sum(0))          # CTM: mu = column mean
    d = eta - eta.mean(0)
    assert np.allclose(cov, d.T @ d, rtol=1e-11, atol=1e-9)
    far = 1e4 + 1e-3 * rng.normal(size=(N, n))                                   # eta^T eta ~ 1e10, covariance ~ 1e-4
    _, ratio = m._covariance_from_moments(far.T @ far, N, far.sum(0))
    assert ratio < 1e-4
    # both forms through the resident loop
    out = {}
    for form in ("moments", "exact"):
        mm = _model(g, "STM", 2)
        mm.cov_exchange = form
        mm.expectation_maximization(saving=False)
        assert mm.cov_exchanges == [form, form]
        out[form] = (mm.sigma.copy(), np.array(mm.last_bounds))
    assert np.allclose(out["moments"][0], out["exact"][0], rtol=1e-9, atol=1e-12)
    assert np.allclose(out["moments"][1], out["exact"][1], rtol=1e-12)


def test_unknown_mode_falls_back_to_ols_like_the_reference(capsys):
    g = load_golden("c1_k10")
    a, b = _model(g, "STM", 1, mode="foo"), _model(g, "STM", 1)
    a.expectation_maximization(saving=False)          # resident path; stm.py:696-700 prints a notice and uses OLS
    b.expectation_maximization(saving=False)
    assert "default 'ols'" in capsys.readouterr().out
    assert np.allclose(a.gamma, b.gamma, rtol=1e-12) and np.allclose(a.sigma, b.sigma, rtol=1e-12)
    assert a.phi is not None and a.phi.shape[0] == int(g["K"])   # self.phi is populated after a resident iteration too


def test_em_driver_convergence_and_caps():
    g = load_golden("toy_ctm")
    m = _model(g, "CTM", 50)
    m.convergence_threshold = 1.0          # |new-old|/|old| < 1 at the second iteration (stm.py:891-893)
    m.expectation_maximization(saving=False)
    assert len(m.last_bounds) == 2
    assert m.EM_is_converged(0) is False    # needs two bounds (stm.py:885)
    assert m.max_its_reached(49) and not m.max_its_reached(3)


def test_save_model_layout(tmp_path):
    g = load_golden("c1_k10")
    m = _model(g, "STM", 1)
    m.expectation_maximization(saving=True, output_dir=str(tmp_path))
    for f in ("beta_hat", "theta_hat", "sigma_hat", "eta_hat", "mu_hat", "X", "gamma_hat"):   # stm.py:1123-1141
        assert os.path.exists(tmp_path / (f + ".npy")), f
    with open(tmp_path / "lower_bound.pickle", "rb") as fh:
        assert pickle.load(fh) == m.last_bounds
    assert np.load(tmp_path / "theta_hat.npy").shape == (m.N, m.K)


def test_error_behaviour_matches_reference():
    g = load_golden("toy_ctm")
    with pytest.raises(ValueError):       # neither of the reference's two init types
        STM(documents=_corpus(g), dictionary=None, content=False, K=3, X=g["X"][:, 0], kappa_interactions=False,
            max_em_iter=1, sigma_prior=0, convergence_threshold=1e-5, init_type="kmeans", engine=OracleEngine())
    with pytest.raises(ValueError):       # stm.py:389-390
        STM(documents=_corpus(g), dictionary=None, content=False, K=0, X=None, kappa_interactions=False,
            max_em_iter=1, sigma_prior=0, convergence_threshold=1e-5, init_type="random", engine=OracleEngine())
    m = _model(g, "CTM", 1)
    m.sigma = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(np.linalg.LinAlgError):   # np.linalg.cholesky(self.sigma), stm.py:499
        m.E_step()
    m = _model(g, "CTM", 1)
    b = m.beta.copy(); b[0, int(g["indices"][0])] = -0.5; m.beta = b
    with pytest.raises(AssertionError):          # stm.py:534
        m.E_step()
    with pytest.raises(AssertionError):          # stm.py:720
        _model(g, "CTM", 1).update_sigma(np.zeros((2, 2)), sigprior=2)


def test_accepts_reference_bow_lists_and_dictionary():
    g = load_golden("toy_ctm")
    docs = _corpus(g).to_bow()
    dictionary = {i: str(i) for i in range(int(g["V"]))}
    m = STM(documents=docs, dictionary=dictionary, content=False, K=3, X=g["X"][:, 0], kappa_interactions=False,
            max_em_iter=2, sigma_prior=0, convergence_threshold=1e-5, init_type="random", model_type="CTM",
            engine=OracleEngine())
    m.expectation_maximization(saving=False)
    assert m.bound == pytest.approx(float(g["final_bound"]), rel=1e-9)


# ----------------------------------------------------------------------------- product hygiene
def test_product_never_touches_the_oracle():
    """The shipped package must not import / load anything under oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "strutopy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libstm_oracle" not in txt and "stm_oracle" not in txt, f


def test_lasso_from_moments_is_sklearns_lasso():
    """mode="lasso" (stm.py:677-681) from the centred moments alone: strutopy_amd.stm.lasso_from_moments (coordinate descent on the
    Gram matrix -- sklearn's enet_coordinate_descent_gram restated, all targets side by side) against sklearn.linear_model.Lasso
    (alpha = 1, fit_intercept = True: what the reference calls) on one-hot and dense designs, shrunk-to-zero and dense solutions."""
    import sklearn.linear_model as lm
    from strutopy_amd.stm import lasso_from_moments
    rng = np.random.default_rng(0)
    for trial in range(6):
        N, n = 400 + 50 * trial, 9
        if trial < 3:
            X = np.eye(3)[rng.integers(0, 3, N)]                       # a three-level covariate, one-hot (singular centred Gram matrix)
        else:
            X = rng.normal(size=(N, 5)) * 3.0
        G = rng.normal(size=(X.shape[1], n)) * (4.0 if trial % 2 else 1.2)
        eta = X @ G + rng.normal(size=(N, n))
        want = lm.Lasso(alpha=1, fit_intercept=True).fit(X, eta).coef_
        Xc, yc = X - X.mean(0), eta - eta.mean(0)
        got = lasso_from_moments(Xc.T @ Xc, Xc.T @ yc, np.sum(yc * yc, axis=0), N)
        assert got.shape == want.shape and np.allclose(got, want, rtol=1e-9, atol=1e-10), trial
        # ... and from the raw sums a sharded fit all-reduces (what _em_iteration_resident forms them from)
        xb, eb = X.mean(0), eta.mean(0)
        got2 = lasso_from_moments(X.T @ X - N * np.outer(xb, xb), X.T @ eta - N * np.outer(xb, eb), np.sum(eta * eta, axis=0) - N * eb * eb, N)
        assert np.allclose(got2, want, rtol=1e-7, atol=1e-9), trial
    assert np.count_nonzero(want) > 0
