"""Parity of the HIP E-step (through the C-ABI of include/stm_estep.h) on a real MI355X.

Compared against (i) the CPU oracle on the same inputs, (ii) the golden vectors produced by the
reference, and (iii) at BASELINE.json's full size (100k documents, V=10k, K=50) size-independent
properties.  fp64 tolerances: scipy status / nit / PD path exact,
eta <= 1e-7 abs, per-document bound <= 1e-8 rel, total ELBO <= 1e-9 rel (north_star asks 1e-6).
"""
import os

import numpy as np
import pytest

from conftest import load_golden, reference_beta0

pytestmark = pytest.mark.gpu

CASES = [("toy_ctm", 2), ("edge", 2), ("content_a2", 2), ("c1_k10", 3), ("k50_v10k", 2), ("wiki_k50", 2)]


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def _norm_beta(bss):
    rs = bss.sum(axis=1)[:, None]
    return np.divide(bss, rs, out=np.zeros_like(bss), where=rs != 0)


def _check(d, o, tag):
    assert np.array_equal(d["status"], o["status"]), f"{tag}: scipy status differs"
    assert np.array_equal(d["nit"], o["nit"]), f"{tag}: BFGS iteration counts differ"
    assert np.array_equal(d["pd_path"], o["pd_path"]), f"{tag}: PD-fix path differs"
    assert np.max(np.abs(d["eta"] - o["eta"])) <= 1e-7, tag
    assert np.max(np.abs(d["theta"] - o["theta"])) <= 1e-7, tag
    assert np.max(np.abs(d["bound_doc"] - o["bound_doc"]) / np.abs(o["bound_doc"])) <= 1e-8, tag
    assert abs(d["bound"] - o["bound"]) <= 1e-9 * abs(o["bound"]), tag
    assert _rel(d["sigma_ss"], o["sigma_ss"]) <= 1e-7, tag
    assert _rel(d["beta_ss"], o["beta_ss"]) <= 1e-7, tag


def test_native_library_is_what_runs():
    from strutopy_amd import _lib
    from strutopy_amd.engine import HipEstepEngine
    e = HipEstepEngine(0)
    info = e.device_info()
    e.close()
    assert "gfx950" in info["name"] and info["cu"] >= 200
    assert any(os.path.basename(_lib.LIB_PATH) in line for line in open("/proc/self/maps"))


@pytest.mark.parametrize("name,its", CASES)
def test_estep_matches_oracle_and_reference(oracle, name, its):
    from strutopy_amd.engine import estep_host
    g = load_golden(name)
    aspect = g["aspect"] if "aspect" in g.files else None
    beta = g["beta0"] if "beta0" in g.files else reference_beta0(int(g["K"]), int(g["V"]))
    for it in range(its):
        p = f"it{it}_"
        args = (g["indptr"], g["indices"], g["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                float(g[p + "sigmaentropy"]))
        o = oracle.estep(*args, aspect=aspect, nthreads=0)
        d = estep_host(*args, aspect=aspect)
        _check(d, o, f"{name} it{it}")
        # and directly against what the reference produced
        assert np.array_equal(d["status"], g[p + "status"]) and np.array_equal(d["nit"], g[p + "nit"])
        assert np.max(np.abs(d["eta"] - g[p + "eta"])) <= 1e-7
        assert abs(d["bound"] - float(g[p + "bound"])) <= 1e-9 * abs(float(g[p + "bound"]))
        assert _rel(d["sigma_ss"], g[p + "sigma_ss"]) <= 1e-7
        if p + "beta_ss" in g.files:
            assert _rel(d["beta_ss"], g[p + "beta_ss"]) <= 1e-7
        else:
            assert _rel(d["beta_ss"][:, g["sample_cols"]], g[p + "beta_ss_cols"]) <= 1e-7
            assert _rel(d["beta_ss"].sum(axis=0), g[p + "beta_ss_colsum"]) <= 1e-7
        beta = g[p + "beta_out"] if p + "beta_out" in g.files else _norm_beta(o["beta_ss"])


def test_wiki_known_answer_shipped_by_reference():
    from strutopy_amd.engine import estep_host
    g = load_golden("wiki_k50")
    shipped = float(g["shipped_lower_bound"][0])   # -855111.02, reference_model/50/lower_bound.pickle
    d = estep_host(g["indptr"], g["indices"], g["counts"], reference_beta0(50, int(g["V"])), g["it0_mu_in"],
                   g["it0_eta_in"], g["it0_siginv"], float(g["it0_sigmaentropy"]))
    assert abs(d["bound"] - shipped) <= 1e-9 * abs(shipped)


def test_hessian_cholesky_nu_per_document(monkeypatch):
    from strutopy_amd.engine import HipEstepEngine
    for name in ("toy_ctm", "edge"):
        g = load_golden(name)
        e = HipEstepEngine(0, debug={"STM_DEBUG_DUMP": 1})   # (the -DSTM_TESTING build: the product library has no debug switches)
        e.set_corpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
        e.set_topics(int(g["K"]))
        e.put_beta(g["beta0"]); e.put_mu(g["it0_mu_in"]); e.put_eta(g["it0_eta_in"])
        e.estep(g["it0_siginv"], float(g["it0_sigmaentropy"]))
        hess, chol, nu = e.debug_mats()
        phi = e.get_phi_last()
        e.close()
        assert _rel(hess, g["it0_hess"]) <= 1e-7, name
        assert _rel(chol, g["it0_chol"]) <= 1e-7, name
        assert _rel(nu, g["it0_nu"]) <= 1e-6, name
        assert _rel(phi, g["it0_phi_last"]) <= 1e-7, name


@pytest.mark.parametrize("K,nd_max", [(2, 40), (17, 700), (18, 60), (33, 100), (34, 100), (49, 130), (51, 70), (64, 90), (65, 120), (80, 100), (81, 100), (100, 300),
                                      (112, 120), (113, 90), (128, 90)])
def test_shapes_at_the_limits(oracle, K, nd_max):
    """smallest / largest K of this build (K <= 64: one topic per lane + MFMA post kernel; 64 < K <= 112: two topics per lane
    in the solver, two waves per document in the post step -- 80 | 81 is where its tile pitch changes, 112 | 113 where the
    general post_any_kernel takes over, in its atomics-free form up to K = 128) and documents longer than one 64-word tile.  18 | 33 | 34 | 49 | 51: every block
    count of post_kernel with and without its remainder row (K = 16 NB + 2), i.e. every rows-per-instruction of the tile fetch."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(K)
    V, N = 900, 70
    docs = [np.sort(rng.choice(V, int(rng.integers(1, nd_max + 1)), replace=False)) for _ in range(N)]
    docs[0] = np.arange(nd_max)
    indptr = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 6, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    Bm = rng.normal(size=(n, n)); sigma = Bm @ Bm.T + np.eye(n)
    siginv, sigent = oracle.preamble(sigma)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    _check(estep_host(*args), oracle.estep(*args, nthreads=0), f"K={K}")
    dense = np.linalg.inv(sigma)          # a caller-supplied dense siginv takes the general path
    args = (indptr, indices, counts, beta, mu, eta, dense, sigent)
    _check(estep_host(*args), oracle.estep(*args, nthreads=0), f"K={K} dense")


def _random_case(rng, K, V, N, nd_max, dense):
    docs = [np.sort(rng.choice(V, int(rng.integers(1, nd_max + 1)), replace=False)) for _ in range(N)]
    docs[0] = np.arange(nd_max)
    indptr = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 6, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    Bm = rng.normal(size=(n, n)); sigma = Bm @ Bm.T + np.eye(n)
    return indptr, indices, counts, beta, mu, eta, sigma


@pytest.mark.parametrize("K,nd_max", [(129, 90), (160, 300), (256, 70), (257, 40), (300, 50), (512, 30)])
def test_more_than_128_topics(oracle, K, nd_max):
    """K > 128 (the reference takes any K, stm.py:311-329): the solver's general form with four / eight vector components per
    lane (slab and BFGS matrix in HBM) and post_any_kernel (stm_post_any.h) -- 129 | 256 | 257 are the edges of the two, 512 the
    largest K the headers advertise (its per-document HBM state is bounded per launch, not by the corpus)."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(K)
    indptr, indices, counts, beta, mu, eta, sigma = _random_case(rng, K, 900, 40, nd_max, False)
    siginv, sigent = oracle.preamble(sigma)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    _check(estep_host(*args), oracle.estep(*args, nthreads=0), f"K={K}")
    if K <= 160:
        args = (indptr, indices, counts, beta, mu, eta, np.linalg.inv(sigma), sigent)
        _check(estep_host(*args), oracle.estep(*args, nthreads=0), f"K={K} dense")


@pytest.mark.parametrize("K,nd_max", [(2, 40), (3, 30), (18, 300), (50, 130), (100, 200), (128, 90)])
def test_general_post_kernel_is_a_second_implementation(oracle, monkeypatch, K, nd_max):
    """STM_POST_ANY=1 routes every K through post_any_kernel (plain loops in HBM scratch, no matrix cores, no LDS matrix): it must
    agree with the oracle wherever the matrix-core kernels do, and with them (same solver, so eta is the same bits)."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(500 + K)
    indptr, indices, counts, beta, mu, eta, sigma = _random_case(rng, K, 700, 50, nd_max, False)
    siginv, sigent = oracle.preamble(sigma)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    fast = estep_host(*args)
    monkeypatch.setenv("STM_POST_ANY", "1")
    slow = estep_host(*args)
    _check(slow, oracle.estep(*args, nthreads=0), f"K={K} general post kernel")
    assert np.array_equal(slow["eta"], fast["eta"]) and np.array_equal(slow["pd_path"], fast["pd_path"])
    assert np.max(np.abs(slow["bound_doc"] - fast["bound_doc"]) / np.abs(fast["bound_doc"])) <= 1e-10
    assert _rel(slow["sigma_ss"], fast["sigma_ss"]) <= 1e-9 and _rel(slow["beta_ss"], fast["beta_ss"]) <= 1e-10


def test_general_post_kernel_matrices_per_document(monkeypatch):
    """H, L and nu of post_any_kernel against the reference's own (toy_ctm / edge goldens: PD ladder paths included)."""
    from strutopy_amd.engine import HipEstepEngine
    monkeypatch.setenv("STM_POST_ANY", "1")
    for name in ("toy_ctm", "edge"):
        g = load_golden(name)
        e = HipEstepEngine(0, debug={"STM_DEBUG_DUMP": 1})
        e.set_corpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
        e.set_topics(int(g["K"]))
        e.put_beta(g["beta0"]); e.put_mu(g["it0_mu_in"]); e.put_eta(g["it0_eta_in"])
        e.estep(g["it0_siginv"], float(g["it0_sigmaentropy"]))
        hess, chol, nu = e.debug_mats()
        phi = e.get_phi_last()
        e.close()
        assert _rel(hess, g["it0_hess"]) <= 1e-7, name
        assert _rel(chol, g["it0_chol"]) <= 1e-7, name
        assert _rel(nu, g["it0_nu"]) <= 1e-6, name
        assert _rel(phi, g["it0_phi_last"]) <= 1e-7, name


@pytest.mark.parametrize("K,nd_max", [(17, 70), (34, 50), (49, 70), (50, 90), (64, 90), (100, 60), (128, 40)])
def test_post_kernels_ignore_stale_lds(oracle, monkeypatch, K, nd_max):
    """STM_POST_DEBUG=16 fills the post kernel's LDS with NaN before every document (STM_DEBUG_FLAGS=8: the solver's, per
    workgroup): a masked term that multiplies
    an unwritten cell by zero (instead of selecting it away) turns sigma_ss into NaN."""
    from strutopy_amd.engine import estep_host
    monkeypatch.setenv("STM_POST_DEBUG", "16")
    monkeypatch.setenv("STM_DEBUG_FLAGS", "8")     # the solver's LDS too (per workgroup)
    rng = np.random.default_rng(1000 + K)
    V, N = 600, 40
    docs = [np.sort(rng.choice(V, int(rng.integers(1, nd_max + 1)), replace=False)) for _ in range(N)]
    indptr = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 6, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    siginv, sigent = oracle.preamble(np.eye(n) * 20.0)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    _check(estep_host(*args, testing=True), oracle.estep(*args, nthreads=0), f"K={K} poisoned LDS")


@pytest.mark.parametrize("big2", ["1", "0"])
def test_k100_two_topics_per_lane(oracle, monkeypatch, big2):
    """BASELINE config 4's K = 100: the two-topics-per-lane solver and the post step -- post_big2_kernel (two waves per
    document; STM_POST_BIG2=1, the default) and, with STM_POST_BIG2=0, the general post_any_kernel in its atomics-free form --
    per-document Hessian / Cholesky / nu against the oracle, and the device M-step at n = 99 (resident EM iterations
    against the host-NumPy M-step)."""
    monkeypatch.setenv("STM_POST_BIG2", big2)
    from strutopy_amd import STM
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.engine import HipEstepEngine
    rng = np.random.default_rng(100)
    K, V, N = 100, 2500, 48
    lens = rng.integers(1, 260, size=N)
    docs = [np.sort(rng.choice(V, int(L), replace=False)) for L in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 5, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    siginv, sigent = oracle.preamble(np.eye(n) * 20.0)
    e = HipEstepEngine(0, debug={"STM_DEBUG_DUMP": 1})
    e.set_corpus(indptr, indices, counts, V)
    e.set_topics(K)
    e.put_beta(beta); e.put_mu(mu); e.put_eta(eta)
    e.estep(siginv, sigent)
    hess, chol, nu = e.debug_mats()
    phi = e.get_phi_last()
    e.close()
    o = oracle.estep(indptr, indices, counts, beta, mu, eta, siginv, sigent, dump_mats=True, nthreads=0)
    assert _rel(hess, o["hess"]) <= 1e-9
    assert _rel(chol, o["chol"]) <= 1e-9
    assert _rel(nu, o["nu"]) <= 1e-8
    assert _rel(phi, o["phi_last"]) <= 1e-8
    c = PackedCorpus(indptr, indices, counts, V)
    X = rng.integers(0, 2, size=(N, 2)).astype(np.float64)   # 0/1 columns are used as they are (stm.py:656-671)
    out = []
    for resident in (True, False):
        m = STM(documents=c, dictionary=None, content=False, K=K, X=X, kappa_interactions=False, max_em_iter=2,
                sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
        m.expectation_maximization(saving=False, resident=resident)
        out.append((np.array(m.last_bounds), m.beta.copy(), m.sigma.copy(), m.gamma.copy()))
        m.close()
    assert np.allclose(out[0][0], out[1][0], rtol=1e-9)
    assert np.allclose(out[0][1], out[1][1], rtol=1e-6, atol=1e-12)
    assert np.allclose(out[0][2], out[1][2], rtol=1e-6, atol=1e-9)
    assert np.allclose(out[0][3], out[1][3], rtol=1e-6, atol=1e-8)


def test_content_levels_at_k50(oracle):
    """Per-level beta (content covariate, A = 3 levels) at the headline K = 50: the aspect selects the beta the rows are
    gathered from and the beta_ss slice phi is added to (stm.py:527-528, 584-588), through the 3 x 3-block + remainder-row
    post kernel and the two-wave solver, documents on both sides of the 128-word register limit."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(350)
    K, V, N, A = 50, 1500, 90, 3
    lens = rng.integers(1, 230, size=N)
    docs = [np.sort(rng.choice(V, int(L), replace=False)) for L in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 4, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(A, K, V)); beta /= beta.sum(axis=2)[:, :, None]
    aspect = rng.integers(0, A, size=N).astype(np.int32)
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    siginv, sigent = oracle.preamble(np.eye(n) * 20.0)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    _check(estep_host(*args, aspect=aspect), oracle.estep(*args, aspect=aspect, nthreads=0), "A=3, K=50")


@pytest.mark.parametrize("mode", [3, 1, 2])
def test_solver_variants_agree_with_the_oracle(oracle, monkeypatch, mode):
    """STM_SOLVER_MODE picks the other data paths of the solver (3: one wave per document, beta_d in registers + LDS; 1: LDS
    slab only; 2: the global slab long documents fall back to): same scipy trajectory, same results."""
    from strutopy_amd.engine import estep_host
    monkeypatch.setenv("STM_SOLVER_MODE", str(mode))
    g = load_golden("c1_k10")
    args = (g["indptr"], g["indices"], g["counts"], g["beta0"], g["it0_mu_in"], g["it0_eta_in"], g["it0_siginv"],
            float(g["it0_sigmaentropy"]))
    d = estep_host(*args)
    assert np.array_equal(d["status"], g["it0_status"]) and np.array_equal(d["nit"], g["it0_nit"])
    assert np.max(np.abs(d["eta"] - g["it0_eta"])) <= 1e-7
    assert abs(d["bound_doc"].sum() - float(g["it0_bound"])) <= 1e-9 * abs(float(g["it0_bound"]))
    g = load_golden("k50_v10k")
    beta = reference_beta0(int(g["K"]), int(g["V"]))
    args = (g["indptr"], g["indices"], g["counts"], beta, g["it0_mu_in"], g["it0_eta_in"], g["it0_siginv"],
            float(g["it0_sigmaentropy"]))
    d = estep_host(*args)
    assert np.array_equal(d["status"], g["it0_status"]) and np.array_equal(d["nit"], g["it0_nit"])
    assert np.max(np.abs(d["eta"] - g["it0_eta"])) <= 1e-7


def test_exactly_singular_hessian_after_make_pd(oracle):
    """K = 3: make_pd turns the 2 x 2 Hessian of some documents into [[|o|, o], [o, |o|]], exactly singular; the sign of the second
    pivot (like the sign of the smallest eigenvalue the reference tests, stm.py:1017) is rounding noise there.  Oracle and kernels
    count a pivot within 32 ulp of the cancelled diagonal as failed (profiles/HISTORY.md section 9), so both take the + 1e-5 branch and nu
    stays finite.  Inputs: a case found by tools/fuzz_parity.py (58 documents, three beta levels, counts up to 1000)."""
    from strutopy_amd.engine import estep_host
    g = load_golden("k3_singular_inputs")
    args = (g["indptr"], g["indices"], g["counts"], g["beta"], g["mu"], g["eta"], g["siginv"], float(g["sigent"]))
    o = oracle.estep(*args, aspect=g["aspect"], nthreads=0)
    d = estep_host(*args, aspect=g["aspect"])
    assert np.bincount(o["pd_path"], minlength=3)[2] >= 2          # the singular documents are in there
    _check(d, o, "K=3 singular")
    assert np.isfinite(d["sigma_ss"]).all() and np.max(np.abs(d["sigma_ss"])) < 1e7


@pytest.mark.parametrize("K", [17, 64, 100, 128])
def test_pd_ladder_beyond_the_first_rung(oracle, K):
    """A wide prior (sigma = 1e3 I) leaves the Hessian of random-init documents indefinite: the first Cholesky fails and
    make_pd (stm.py:964-984, 1019-1020) decides -- the blocked factorisations of both post kernels restart from A in the
    upper triangle and the ladder's current diagonal."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(10 * K + 1)
    V, N = 700, 50
    docs = [np.sort(rng.choice(V, int(rng.integers(1, 120)), replace=False)) for _ in range(N)]
    indptr = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 6, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    mu = rng.normal(0, 0.3, size=(N, n)); eta = rng.normal(0, 0.3, size=(N, n))
    siginv, sigent = oracle.preamble(np.eye(n) * 1e3)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    o = oracle.estep(*args, nthreads=0)
    assert np.count_nonzero(o["pd_path"]) >= 5
    _check(estep_host(*args), o, f"K={K} ladder")


def test_documents_longer_than_the_lds(oracle):
    """A document whose K x Nd block cannot live in the 160 KB LDS takes the global-slab solver variant;
    shorter ones in the same corpus stay on chip."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(5)
    K, V = 40, 6000
    lens = [5200, 3, 150, 4100, 64, 65, 129, 128, 2000, 90]
    docs = [np.sort(rng.choice(V, L, replace=False)) for L in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 4, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    n = K - 1
    N = len(lens)
    mu = rng.normal(0, 0.2, size=(N, n)); eta = rng.normal(0, 0.2, size=(N, n))
    siginv, sigent = oracle.preamble(np.eye(n) * 3.0)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    _check(estep_host(*args), oracle.estep(*args, nthreads=0), "long documents")


@pytest.mark.gpu
@pytest.mark.parametrize("K", [18, 50])
def test_documents_at_the_edge_of_the_lds(oracle, K):
    """Document lengths that walk across the point where static + dynamic LDS of the two-wave solver reach 160 KB: the plan
    must count the kernel's real static LDS (asked from the runtime) -- with a stale constant a document a few words below
    the limit was planned for the LDS form and failed in hipFuncSetAttribute (found by tools/fuzz_parity.py)."""
    from strutopy_amd.engine import estep_host
    rng = np.random.default_rng(K)
    V = 3000
    n = K - 1
    kreg = 32 if K <= 32 else 50
    kp = ((kreg - 2 + 3) // 4) * 4 + 2                       # slab row length (stm_api.hip: slab_row)
    edge = 128 + (160 * 1024 - n * n * 8 - 3600) // ((kp + 2) * 8)   # words at which the workgroup's LDS is about 160 KB
    lens = [edge + d for d in (-3, -2, -1, 0, 1, 2)] + [100, 130]
    docs = [np.sort(rng.choice(V, L, replace=False)) for L in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate(docs).astype(np.int32)
    counts = rng.integers(1, 4, size=len(indices)).astype(np.float64)
    beta = rng.gamma(0.1, 1, size=(K, V)); beta /= beta.sum(axis=1)[:, None]
    N = len(lens)
    mu = rng.normal(0, 0.1, size=(N, n)); eta = rng.normal(0, 0.1, size=(N, n))
    siginv, sigent = oracle.preamble(np.eye(n) * 2.0)
    args = (indptr, indices, counts, beta, mu, eta, siginv, sigent)
    _check(estep_host(*args), oracle.estep(*args, nthreads=0), "documents at the LDS limit")


def test_invalid_inputs_raise_like_the_reference():
    from strutopy_amd.engine import HipEstepEngine, estep_host
    g = load_golden("toy_ctm")
    args = [g["indptr"], g["indices"], g["counts"], g["beta0"].copy(), g["it0_mu_in"], g["it0_eta_in"],
            g["it0_siginv"], float(g["it0_sigmaentropy"])]
    args[3][1, int(g["indices"][3])] = -1e-9
    with pytest.raises(AssertionError):      # "Some entries of beta are negative or nan." stm.py:534
        estep_host(*args)
    args[3][1, int(g["indices"][3])] = np.nan
    with pytest.raises(AssertionError):
        estep_host(*args)
    e = HipEstepEngine(0)
    with pytest.raises(ValueError):          # empty document: the reference indexes doc_array[:, 0]
        e.set_corpus(np.array([0, 2, 2]), np.array([1, 2]), np.array([1.0, 1.0]), 5)
    with pytest.raises(ValueError):          # word id outside the dictionary
        e.set_corpus(np.array([0, 2]), np.array([1, 7]), np.array([1.0, 1.0]), 5)
    with pytest.raises(ValueError, match="same word id twice"):      # stm_set_corpus's own check (the C-ABI may be called directly)
        e.set_corpus(np.array([0, 2, 5]), np.array([3, 1, 4, 1, 4]), np.ones(5), 5)
    with pytest.raises(ValueError):          # call order
        e.set_topics(4)
    e.set_corpus(np.array([0, 2, 5]), np.array([3, 1, 1, 4, 3]), np.ones(5), 5)   # the same id in two documents is a corpus
    e.set_topics(4)
    e.close()


def test_stm_class_reproduces_reference_traces(oracle):
    from strutopy_amd import STM
    from strutopy_amd.corpus import PackedCorpus
    g = load_golden("toy_ctm")
    c = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    for resident in (True, False):
        m = STM(documents=c, dictionary=None, content=False, K=3, X=g["X"][:, 0], kappa_interactions=False,
                max_em_iter=2, sigma_prior=0, convergence_threshold=1e-5, init_type="random", model_type="CTM")
        m.expectation_maximization(saving=False, resident=resident)
        assert m.bound == pytest.approx(float(g["final_bound"]), rel=1e-9)    # tests/test_integration.py pipeline
        assert np.allclose(m.beta, g["it1_beta_out"], rtol=1e-7, atol=1e-12)
        assert np.allclose(m.sigma, g["it1_sigma_out"], rtol=1e-7, atol=1e-10)
        m.close()
    g = load_golden("c1_k10")
    c = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    m = STM(documents=c, dictionary=None, content=False, K=10, X=g["X"][:, 0], kappa_interactions=False,
            max_em_iter=3, sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")
    beta_ss, sigma_ss = m.E_step()
    assert np.allclose(beta_ss, g["it0_beta_ss"], rtol=1e-7, atol=1e-12)
    assert np.allclose(sigma_ss, g["it0_sigma_ss"], rtol=1e-7)
    assert np.allclose(m.phi, g["it0_phi_last"], rtol=1e-7)
    m.M_step(beta_ss, sigma_ss)
    assert np.allclose(m.gamma, g["it0_gamma"], rtol=1e-6, atol=1e-9)
    assert np.allclose(m.beta, g["it0_beta_out"], rtol=1e-7, atol=1e-14)
    m.last_bounds = []
    m.beta = g["beta0"]; m.init_mu(); m.init_eta(); m.init_sigma()
    m.expectation_maximization(saving=False)            # device-resident E+M iterations
    for it in range(3):
        assert m.last_bounds[it] == pytest.approx(float(g[f"it{it}_bound"]), rel=1e-8)
    assert np.allclose(m.sigma, g["it2_sigma_out"], rtol=1e-6, atol=1e-7)
    assert np.allclose(m.beta, g["it2_beta_out"], rtol=1e-5, atol=1e-10)
    assert np.allclose(m.theta.sum(axis=1), 1.0, atol=1e-12)
    m.close()


def test_eval_heldout_matches_reference(oracle):
    """Document-completion held-out likelihood (src/modules/heldout.py:88-97, used by src/05_train.py:120)."""
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.heldout import eval_heldout
    g = load_golden("heldout")
    held = PackedCorpus(g["second_indptr"], g["second_indices"], g["second_counts"], int(g["V"]))
    assert eval_heldout(held, g["theta"], g["beta"]) == pytest.approx(float(g["mean"]), rel=1e-12)
    assert eval_heldout(held.to_bow(), g["theta"], g["beta"]) == pytest.approx(float(g["mean"]), rel=1e-12)
    from strutopy_amd.engine import HipEstepEngine
    e = HipEstepEngine(0)
    e.set_corpus(held.indptr, held.indices, held.counts, held.V)
    e.set_topics(int(g["K"]))
    e.put_beta(g["beta"])
    per_doc = e.eval_heldout(held.indptr, held.indices, held.counts, g["theta"])
    e.close()
    assert np.allclose(per_doc, g["per_doc"], rtol=1e-12)
    assert np.allclose(per_doc, oracle.eval_heldout_docs(held.indptr, held.indices, held.counts, g["theta"], g["beta"]), rtol=1e-12)
    with pytest.raises(ValueError):
        eval_heldout(held, g["theta"][:5], g["beta"])


def test_rccl_all_reduce_path_single_rank():
    """The RCCL binding (dlopen, ncclCommInitRank, ncclAllReduce on the packed device buffer) with a
    one-rank communicator: the reduction is the identity, so every statistic must come back unchanged."""
    from strutopy_amd.engine import HipEstepEngine
    g = load_golden("c1_k10")
    e = HipEstepEngine(0)
    e.set_corpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    e.set_topics(int(g["K"]))
    e.put_beta(g["beta0"]); e.put_mu(g["it0_mu_in"]); e.put_eta(g["it0_eta_in"])
    bound = e.estep(g["it0_siginv"], float(g["it0_sigmaentropy"]))
    beta_ss, sigma_ss = e.get_beta_ss(), e.get_sigma_ss()
    e.comm_init(e.comm_unique_id(), 0, 1)
    e.put_covariates(g["X"][:, :1])
    mom = e.moments(1)                    # left in the moment region of the packed buffer: [N | sx | se | XtX | Xte | ete]
    n = int(g["K"]) - 1
    eta, X = e.get_eta(), g["X"][:, :1]
    want = np.concatenate([[len(eta)], X.sum(0), eta.sum(0), (X.T @ X).ravel(), (X.T @ eta).ravel(), (eta.T @ eta).ravel()])
    assert mom.shape == (1 + 1 + n + 1 + n + n * n,) and np.allclose(mom, want, rtol=1e-12, atol=1e-9)
    b2, mom2 = e.allreduce_suffstats(mom)
    assert b2 == bound and np.array_equal(mom2, mom)
    assert np.array_equal(e.get_beta_ss(), beta_ss) and np.array_equal(e.get_sigma_ss(), sigma_ss)
    small = np.linspace(-1, 1, 81).reshape(9, 9)
    assert np.array_equal(e.allreduce_small(small), small)
    big = np.linspace(-1, 1, 127 * 127)   # any length: its own device buffer (K >= 66 did not fit)
    assert np.array_equal(e.allreduce_small(big), big)
    e.close()


@pytest.fixture(scope="module")
def full_size():
    """BASELINE.json configs[1]: 100k synthetic documents x 150 words, V=10k, K=50."""
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    syn = synthetic_corpus(100_000, 10_000, 50, n_words=150, seed=12345)
    c = syn.corpus
    K, n = 50, 49
    beta = reference_beta0(K, c.V)
    e = HipEstepEngine(0)
    e.set_corpus(c.indptr, c.indices, c.counts, c.V)
    e.set_topics(K)
    e.put_beta(beta)
    siginv, sigent = np.eye(n) / 20.0, float(n * 0.5 * np.log(20.0))
    bound = e.estep(siginv, sigent)
    out = dict(corpus=c, beta=beta, siginv=siginv, sigent=sigent, bound=bound, eta=e.get_eta(), theta=e.get_theta(),
               beta_ss=e.get_beta_ss(), sigma_ss=e.get_sigma_ss(), bound_doc=e.get_bound_docs(),
               diag=e.get_diagnostics(), engine=e)
    yield out
    e.close()


def test_full_size_invariants(full_size):
    f = full_size
    c = f["corpus"]
    # phi columns sum to the word count (stm.py:1115-1116)  =>  beta_ss column sums = corpus word totals
    assert _rel(f["beta_ss"].sum(axis=0), c.word_counts()) <= 1e-11
    assert abs(f["beta_ss"].sum() - c.counts.sum()) <= 1e-11 * c.counts.sum()
    assert f["beta_ss"].min() >= 0
    assert np.allclose(f["theta"].sum(axis=1), 1.0, atol=1e-12) and f["theta"].min() > 0
    assert np.allclose(f["theta"][:, :-1] / f["theta"][:, -1:], np.exp(f["eta"]), rtol=1e-12)   # softmax([eta, 0])
    s = f["sigma_ss"]
    assert np.allclose(s, s.T, rtol=1e-12) and np.linalg.eigvalsh(s).min() > 0                 # sum of H^-1, all PD
    assert np.isfinite(f["bound_doc"]).all() and f["bound"] == pytest.approx(f["bound_doc"].sum(), rel=1e-12)
    assert set(np.unique(f["diag"]["status"])) <= {0, 2} and f["diag"]["nit"].max() < 200 * 49
    assert set(np.unique(f["diag"]["pd_path"])) <= {0, 1, 2}


def test_full_size_sample_against_oracle(oracle, full_size):
    f = full_size
    S = 3000
    sub = f["corpus"].slice(0, S)
    z = np.zeros((S, 49))
    o = oracle.estep(sub.indptr, sub.indices, sub.counts, f["beta"], z, z, f["siginv"], f["sigent"], nthreads=0)
    assert np.array_equal(f["diag"]["status"][:S], o["status"]) and np.array_equal(f["diag"]["nit"][:S], o["nit"])
    assert np.array_equal(f["diag"]["pd_path"][:S], o["pd_path"])
    assert np.max(np.abs(f["eta"][:S] - o["eta"])) <= 1e-7
    assert np.max(np.abs(f["bound_doc"][:S] - o["bound_doc"]) / np.abs(o["bound_doc"])) <= 1e-8
    assert abs(f["bound_doc"][:S].sum() - o["bound"]) <= 1e-9 * abs(o["bound"])


def test_full_size_repeatable_and_shard_linear(full_size):
    """Re-running from the same inputs repeats eta exactly (no cross-document coupling); the
    sufficient statistics of two document shards add up to the full ones (what the all-reduce relies on)."""
    from strutopy_amd.engine import HipEstepEngine
    f = full_size
    e = f["engine"]
    e.put_eta(np.zeros_like(f["eta"]))
    b2 = e.estep(f["siginv"], f["sigent"])
    assert np.array_equal(e.get_eta(), f["eta"]) and np.array_equal(e.get_bound_docs(), f["bound_doc"])
    assert b2 == f["bound"]
    assert _rel(e.get_beta_ss(), f["beta_ss"]) <= 1e-12          # fp64 atomics: order-dependent rounding only
    c = f["corpus"]
    half = c.N // 2
    acc_b, acc_s, acc_bound = np.zeros_like(f["beta_ss"]), np.zeros_like(f["sigma_ss"]), 0.0
    for lo, hi in ((0, half), (half, c.N)):
        sh = c.slice(lo, hi)
        es = HipEstepEngine(0)
        es.set_corpus(sh.indptr, sh.indices, sh.counts, c.V)
        es.set_topics(50)
        es.put_beta(f["beta"])
        acc_bound += es.estep(f["siginv"], f["sigent"])
        acc_b += es.get_beta_ss(); acc_s += es.get_sigma_ss()
        assert np.array_equal(es.get_eta(), f["eta"][lo:hi])
        es.close()
    assert _rel(acc_b, f["beta_ss"]) <= 1e-12 and _rel(acc_s, f["sigma_ss"]) <= 1e-12
    assert acc_bound == pytest.approx(f["bound"], rel=1e-13)


def _c2_corpus(g):
    """The corpus tools/make_golden_c2.py ran the reference on, regenerated and checked against its checksums."""
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(int(g["n_docs"]), int(g["V_requested"]), int(g["K"]), n_words=int(g["n_words"]), seed=int(g["seed"]))
    c = syn.corpus
    assert c.V == int(g["V"]) and int(c.indptr[-1]) == int(g["nnz"])
    assert int(np.sum(c.indices.astype(np.int64) * (np.arange(len(c.indices)) % 9973))) == int(g["checksum_indices"])
    assert float(np.sum(c.counts * (np.arange(len(c.counts)) % 9973))) == float(g["checksum_counts"])
    return syn


def test_full_size_against_the_reference_itself():
    """BASELINE.json configs[1] end to end against the imported reference (tests/golden/c2_full.npz, made by
    tools/make_golden_c2.py): two EM iterations over all 100k documents -- ELBO trace, every document's scipy
    status / nit / PD path, sufficient statistics and the M-step results, through both the reference's own
    call pattern (E_step -> M_step on the host) and the device-resident loop."""
    from strutopy_amd import STM
    g = load_golden("c2_full")
    syn = _c2_corpus(g)
    K, rows, cols = int(g["K"]), g["sample_docs"], g["sample_cols"]

    def model():
        return STM(documents=syn.corpus, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False,
                   max_em_iter=2, sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM")

    m = model()
    assert np.array_equal(m.beta[:, cols], g["beta0_cols"])
    for it in range(2):
        p = f"it{it}_"
        beta_ss, sigma_ss = m.E_step()
        d = m.solver_diagnostics()
        assert m.bound == pytest.approx(float(g[p + "bound"]), rel=1e-10)       # north_star asks 1e-6
        if it == 0:   # identical inputs: every document takes the reference's path
            for k in ("status", "nit", "pd_path"):
                assert np.array_equal(d[k], g[p + k]), k
        else:         # inputs differ by the rounding of one M-step: allow a handful of documents to flip
            for k in ("status", "nit", "pd_path"):
                assert np.mean(d[k] != g[p + k]) <= 1e-4, k
        assert np.max(np.abs(m.eta[rows] - g[p + "eta_sample"])) <= 1e-7
        assert np.max(np.abs(m.theta[rows] - g[p + "theta_sample"])) <= 1e-7
        assert np.allclose(m.eta.sum(axis=0), g[p + "eta_colsum"], rtol=1e-7, atol=1e-6)
        assert np.allclose(m.theta.sum(axis=0), g[p + "theta_colsum"], rtol=1e-9)
        # it 0: identical inputs.  it 1: the inputs carry the rounding of iteration 0's sigma_ss through one M-step, and
        # that rounding is dominated by ONE document (62648: its make_pd'ed Hessian is nearly singular, nu entries ~2e5,
        # and any two factorisation orders -- LAPACK's, the oracle's, this kernel's -- differ by ~3e-10 of that); the
        # solver amplifies the 3e-10 difference in sigma to 4e-8 in the next sigma_ss (<= 1e-7)
        assert _rel(sigma_ss, g[p + "sigma_ss"]) <= (1e-9 if it == 0 else 1e-7)
        assert _rel(beta_ss.sum(axis=1), g[p + "beta_ss_rowsum"]) <= 1e-9
        assert _rel(beta_ss.sum(axis=0), g[p + "beta_ss_colsum"]) <= 1e-9
        assert _rel(beta_ss[:, cols], g[p + "beta_ss_cols"]) <= 1e-7
        assert np.allclose(m.siginv, g[p + "siginv"], rtol=1e-9, atol=1e-14)
        m.M_step(beta_ss, sigma_ss)
        assert np.allclose(m.gamma, g[p + "gamma"], rtol=1e-6, atol=1e-9)
        loose = 1.0 if it == 0 else 10.0      # it 1 inherits the amplified rounding of that one document (above)
        assert np.allclose(m.sigma, g[p + "sigma_out"], rtol=1e-7 * loose, atol=1e-9 * loose)
        assert np.allclose(m.beta[:, cols], g[p + "beta_out_cols"], rtol=1e-7 * loose, atol=1e-14)
        assert np.allclose(m.mu[rows], g[p + "mu_sample"], rtol=1e-6, atol=1e-9)
    m.close()
    m = model()
    m.expectation_maximization(saving=False)                                   # device-resident E + M
    assert m.last_bounds[0] == pytest.approx(float(g["it0_bound"]), rel=1e-10)
    assert m.last_bounds[1] == pytest.approx(float(g["it1_bound"]), rel=1e-9)
    assert np.allclose(m.sigma, g["it1_sigma_out"], rtol=1e-6, atol=1e-8)
    assert np.allclose(m.beta[:, cols], g["it1_beta_out_cols"], rtol=1e-6, atol=1e-13)
    assert np.max(np.abs(m.eta[rows] - g["it1_eta_sample"])) <= 1e-6
    m.close()
