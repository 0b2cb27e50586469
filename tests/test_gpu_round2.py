"""Round-2 parity cases on a real MI355X (all through the C-ABI): the device M-step for per-level beta, later EM
iterations at K=50 (teacher-forced against the reference), BASELINE config 4's per-GPU share, the multi-rank
paths (host reduction between two processes on one GPU, bench.py's own launcher, a one-rank RCCL communicator at
K=100), and the ctypes stub of INTEGRATION.md executed against an object with the reference's attributes."""
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, load_golden, reference_beta0

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def _corpus(g):
    from strutopy_amd.corpus import PackedCorpus
    return PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))


# ------------------------------------------------------------------ M-step branches beside OLS
@pytest.mark.parametrize("tag,mode,sp", __import__("_mstep_modes").CONFIGS)
@pytest.mark.parametrize("resident", [False, True])
def test_mstep_branches_against_the_reference_on_the_gpu(tag, mode, sp, resident):
    """mode="ridge" / "lasso" and sigma_prior = 0.5 through the HIP path: host M-step and the resident loop (ridge: the
    moment-based solve on the reduced statistics; lasso: coordinate descent on the centred Gram matrix) against the reference's run."""
    import _mstep_modes
    _mstep_modes.run(load_golden("mstep_modes"), tag, mode, sp, resident)


# ------------------------------------------------------------------ wide prevalence design (p ~ 300)
def test_resident_em_with_three_hundred_covariate_columns():
    """A prevalence design with 300 columns (what one-hot encoding a category with hundreds of levels gives): the fused
    iteration reads back 8 + n^2 + (1 + p + n + p^2 + p n) doubles -- far beyond one megabyte -- and the regression
    moments' per-block partials are p^2 doubles each.  Nothing in stm_em_begin may depend on a fixed-size region: the
    resident loop must agree with the host M-step (sklearn on the full eta, stm.py:696-723) on the same fit."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(1500, 3000, 10, n_words=60, seed=11)
    X = np.random.default_rng(2).integers(0, 2, size=(1500, 300)).astype(np.float64)   # 0/1: kept as given (stm.py:662)
    out = {}
    for resident in (True, False):
        m = STM(documents=syn.corpus, dictionary=None, content=False, K=10, X=X, kappa_interactions=False, max_em_iter=2,
                sigma_prior=0, convergence_threshold=1e-12, init_type="random", mode="ridge")
        assert m._Xenc.shape[1] == 300
        m.expectation_maximization(saving=False, resident=resident)
        out[resident] = (np.array(m.last_bounds), m.beta.copy(), m.sigma.copy(), m.gamma.copy())
        if resident:
            assert m.cov_exchanges and len(m.timings) == 2
        m.close()
    a, b = out[True], out[False]
    assert np.allclose(a[0], b[0], rtol=1e-9)
    assert np.allclose(a[1], b[1], rtol=1e-6, atol=1e-12)
    assert np.allclose(a[2], b[2], rtol=1e-6, atol=1e-9)
    assert np.allclose(a[3], b[3], rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------ C5: content covariate, device M-step for A > 1
def test_resident_em_with_content_levels_against_reference():
    """content=True + kappa_interactions=True, A=2: the resident loop runs beta_normalise_topics_kernel (the
    reference's np.sum(beta_ss, axis=1) over TOPICS for a 3-D beta_ss, stm.py:741) -- against the reference's
    own two EM iterations (tests/golden/content_a2.npz)."""
    from strutopy_amd import STM
    g = load_golden("content_a2")
    out = {}
    for resident in (True, False):
        m = STM(documents=_corpus(g), dictionary=None, content=True, K=int(g["K"]), X=g["X"][:, 0], kappa_interactions=True,
                A=int(g["A"]), beta_index=g["aspect"], max_em_iter=2, sigma_prior=0, convergence_threshold=1e-12,
                init_type="random", model_type="STM")
        assert m.beta.shape == (2, int(g["K"]), int(g["V"])) and np.array_equal(m.beta, g["beta0"])
        m.expectation_maximization(saving=False, resident=resident)
        assert m.last_bounds[0] == pytest.approx(float(g["it0_bound"]), rel=1e-10)
        assert m.last_bounds[1] == pytest.approx(float(g["it1_bound"]), rel=1e-8)
        assert np.allclose(m.beta, g["it1_beta_out"], rtol=1e-6, atol=1e-12)
        assert np.allclose(m.sigma, g["it1_sigma_out"], rtol=1e-6, atol=1e-9)
        used = g["it1_beta_out"].sum(axis=1) > 0
        assert np.allclose(m.beta.sum(axis=1)[used], 1.0, atol=1e-9)
        out[resident] = (m.beta.copy(), m.sigma.copy(), m.gamma.copy())
        m.close()
    for a, b in zip(out[True], out[False]):
        assert np.allclose(a, b, rtol=1e-7, atol=1e-11)


def test_content_levels_full_iteration_at_k50(oracle):
    """One full resident EM iteration at C5's shape per GPU share scaled down (K=50, V=10k, A=2, 3000 documents):
    device M-step vs the NumPy statement of stm.py:741-745 on the device's own beta_ss."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(3000, 10_000, 50, n_words=150, seed=5)
    aspect = np.random.default_rng(1).integers(0, 2, size=3000).astype(np.int32)
    m = STM(documents=syn.corpus, dictionary=None, content=True, K=50, X=syn.X, kappa_interactions=True, A=2,
            beta_index=aspect, max_em_iter=1, sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    beta0, c = m.beta.copy(), syn.corpus
    m._em_iteration_resident()
    beta_ss = m._engine.get_beta_ss()
    o = oracle.estep(c.indptr, c.indices, c.counts, beta0, np.zeros((c.N, 49)), np.zeros((c.N, 49)), m.siginv,
                     float(m.sigmaentropy), aspect=aspect, nthreads=0)
    assert m.bound == pytest.approx(o["bound"], rel=1e-10)
    assert _rel(beta_ss, o["beta_ss"]) <= 1e-7
    rs = beta_ss.sum(axis=1)[:, None]                                  # axis=1 of (A, K, V): over topics
    want = np.divide(beta_ss, rs, out=np.zeros_like(beta_ss), where=rs != 0)
    assert np.allclose(m.beta, want, rtol=1e-12, atol=0)
    m.close()


# ------------------------------------------------------------------ K=50, later EM iterations, teacher-forced
@pytest.mark.parametrize("flags", ["0", "16", "6"])
def test_k50_later_iterations_teacher_forced(oracle, monkeypatch, flags):
    """tests/golden/k50_late.npz: the reference's EM iterations 3, 4, 5 and 8 at K=50 / V=10k (2000 documents) with the
    complete input state of each E-step.  From iteration 4 on about half of the documents take BFGS steps that move; the
    HIP solver must follow scipy exactly there.  STM_DEBUG_FLAGS=6 disables the outcome-preserving line-search cuts,
    16 only the moment pass in front of the first search (round 3): all three must agree with the reference AND
    with each other (the cuts change nothing)."""
    from strutopy_amd.engine import estep_host
    monkeypatch.setenv("STM_DEBUG_FLAGS", flags)     # (read by the -DSTM_TESTING build only: flags 0 runs through the product library)
    g = load_golden("k50_late")
    for it in g["kept"]:
        p = f"it{int(it)}_"
        args = (g["indptr"], g["indices"], g["counts"], g[p + "beta_in"], g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                float(g[p + "sigmaentropy"]))
        d = estep_host(*args, testing=flags != "0")
        for k in ("status", "nit", "pd_path"):
            assert np.array_equal(d[k], g[p + k]), f"it{it}: {k} differs from the reference in {np.sum(d[k] != g[p + k])} documents"
        assert np.max(np.abs(d["eta"] - g[p + "eta"])) <= 1e-7
        assert np.max(np.abs(d["bound_doc"] - g[p + "bound_doc"]) / np.abs(g[p + "bound_doc"])) <= 1e-8
        assert abs(d["bound"] - float(g[p + "bound"])) <= 1e-10 * abs(float(g[p + "bound"]))
        assert _rel(d["sigma_ss"], g[p + "sigma_ss"]) <= 1e-8
        assert _rel(d["beta_ss"].sum(axis=1), g[p + "beta_ss_rowsum"]) <= 1e-9
        assert _rel(d["beta_ss"].sum(axis=0), g[p + "beta_ss_colsum"]) <= 1e-9
        if int(it) >= 4:
            assert np.mean(g[p + "nit"] > 0) > 0.2          # the regime this case exists for
        if flags == "0":   # and the oracle, evaluation by evaluation (nfev is informational between GPU and oracle)
            o = oracle.estep(*args, nthreads=0)
            assert np.array_equal(o["status"], g[p + "status"]) and np.array_equal(o["nit"], g[p + "nit"])
            assert np.max(np.abs(d["eta"] - o["eta"])) <= 1e-7


def test_moment_pass_proves_failing_first_searches(monkeypatch):
    """The moment pass (stm_solver.h, S_MOMENTS) must actually fire: at the reference's EM iteration 8 (k50_late) 44 % of
    the documents end in a first line search that cannot succeed; with the pass they finish after one evaluation and one
    pass over beta_d, without it after 5-6 evaluations.  Same status / nit / eta bit for bit (x is left unchanged either way)."""
    from strutopy_amd.engine import estep_host
    g = load_golden("k50_late")
    p = "it8_"
    args = (g["indptr"], g["indices"], g["counts"], g[p + "beta_in"], g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
            float(g[p + "sigmaentropy"]))
    out = {}
    for flags in ("0", "16"):
        monkeypatch.setenv("STM_DEBUG_FLAGS", flags)
        out[flags] = estep_host(*args, testing=flags != "0")
    a, b = out["0"], out["16"]
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["nit"], b["nit"])
    assert np.array_equal(a["eta"], b["eta"])
    still = a["nit"] == 0
    assert still.mean() > 0.3
    # documents that do not move: one evaluation (plus the reused / skipped ones scipy would count) against 5-6
    assert a["nfev"][still].mean() <= 0.45 * b["nfev"][still].mean()
    assert np.array_equal(a["nfev"][~still], b["nfev"][~still]) or a["nfev"][~still].mean() <= b["nfev"][~still].mean()



def _against_reference(d, g, p, tag):
    for k in ("status", "nit", "pd_path"):
        assert np.array_equal(d[k], g[p + k]), f"{tag}: {k} differs in {np.sum(d[k] != g[p + k])} documents"
    assert np.max(np.abs(d["eta"] - g[p + "eta"])) <= 1e-7, tag
    assert np.max(np.abs(d["bound_doc"] - g[p + "bound_doc"]) / np.abs(g[p + "bound_doc"])) <= 1e-8, tag
    assert abs(d["bound"] - float(g[p + "bound"])) <= 1e-10 * abs(float(g[p + "bound"])), tag
    assert _rel(d["sigma_ss"], g[p + "sigma_ss"]) <= 1e-8, tag
    assert _rel(d["beta_ss"].sum(axis=-1), g[p + "beta_ss_rowsum"]) <= 1e-9, tag
    assert _rel(d["beta_ss"].sum(axis=-2), g[p + "beta_ss_colsum"]) <= 1e-9, tag


def test_k100_kernels_against_the_reference_itself():
    """K = 100 (two topics per lane: solver_kernel<2,0,false,1,1> re-gathering beta rows per pass; post_big2_kernel<7,56>: two waves per document) against
    three EM iterations of the REFERENCE (tests/golden/k100_v5k.npz), teacher-forced."""
    from strutopy_amd.engine import estep_host
    g = load_golden("k100_v5k")
    for it in range(3):
        p = f"it{it}_"
        beta = reference_beta0(100, int(g["V"])) if it == 0 else g[p + "beta_in"]
        d = estep_host(g["indptr"], g["indices"], g["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                       float(g[p + "sigmaentropy"]))
        _against_reference(d, g, p, f"k100 it{it}")


def test_wiki_k70_known_answer_and_teacher_forced_iteration():
    """wiki corpus at K = 70 (real data, N_d ~ 60, the K > 64 kernels): EM iteration 0 against the number the reference ships
    (src/artifacts/reference_model/70/lower_bound.pickle[0] = -868098.47) and both iterations against the reference run of
    tests/golden/wiki_k70.npz, iteration 1 teacher-forced with the reference's beta."""
    from strutopy_amd.engine import estep_host
    g = load_golden("wiki_k70")
    c = load_golden(str(g["corpus"]))
    shipped = float(g["shipped_lower_bound"][0])
    for it, beta in ((0, reference_beta0(70, int(g["V"]))), (1, g["it1_beta_in"])):
        p = f"it{it}_"
        d = estep_host(c["indptr"], c["indices"], c["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                       float(g[p + "sigmaentropy"]))
        _against_reference(d, g, p, f"wiki k70 it{it}")
        assert _rel(d["beta_ss"][:, g["sample_cols"]], g[p + "beta_ss_cols"]) <= 1e-8
        assert np.max(np.abs(d["theta"] - g[p + "theta"])) <= 1e-7
        if it == 0:
            assert abs(d["bound"] - shipped) <= 1e-9 * abs(shipped)


@pytest.mark.parametrize("K", [70, 100, 113, 128])
def test_k_above_64_statistics_are_run_to_run_identical(K):
    """64 < K <= 112 (post_big2_kernel) and 112 < K <= 128 (post_any_kernel<WM>; the one-wave kernel of rounds 1-5 with its atomics is
    gone): no atomics on the data path -- r_dw + the word-major beta_ss pass, nu in per-workgroup slabs, fixed-order reductions --
    so beta_ss, sigma_ss and the bound of two E-steps on the same state agree bit for bit."""
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    c = synthetic_corpus(3000 if K <= 112 else 1200, 4000, K, n_words=120, seed=K).corpus
    beta = reference_beta0(K, c.V)
    n = K - 1
    rng = np.random.default_rng(K)
    eta0 = rng.normal(0, 0.2, size=(c.N, n))
    siginv = np.eye(n) / 20.0
    sigent = 0.5 * n * np.log(20.0)
    runs = []
    for _ in range(2):
        e = HipEstepEngine(0)
        e.set_corpus(c.indptr, c.indices, c.counts, c.V)
        e.set_topics(K)
        e.put_beta(beta); e.put_mu(np.zeros((c.N, n))); e.put_eta(eta0)
        bound = e.estep(siginv, sigent)
        runs.append((bound, e.get_beta_ss(), e.get_sigma_ss(), e.get_eta()))
        e.close()
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("K", [70, 100])
def test_k_above_64_four_waves_per_document_equal_two(K, monkeypatch):
    """post_big2_kernel<..., NWV = 4> (STM_POST_BIG2_WAVES=4): waves 2 and 3 only take tiles, block rows and block columns off the two
    row-owning waves -- every element keeps its operations and their order, so at an equal number of workgroups (the nu slabs
    are per workgroup: three documents per CU for both) the statistics agree with the two-wave form bit for bit."""
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    c = synthetic_corpus(3000 if K <= 112 else 1200, 4000, K, n_words=120, seed=K).corpus
    beta = reference_beta0(K, c.V)
    n = K - 1
    rng = np.random.default_rng(K)
    eta0 = rng.normal(0, 0.2, size=(c.N, n))
    siginv = np.eye(n) / 20.0 + 0.001          # dense: the assembly adds its off-diagonals
    sigent = 0.5 * n * np.log(20.0)
    monkeypatch.setenv("STM_POST_MAX_WG_PER_CU", "3")
    runs = []
    for waves in ("2", "4"):
        monkeypatch.setenv("STM_POST_BIG2_WAVES", waves)
        e = HipEstepEngine(0)
        e.set_corpus(c.indptr, c.indices, c.counts, c.V)
        e.set_topics(K)
        e.put_beta(beta); e.put_mu(np.zeros((c.N, n))); e.put_eta(eta0)
        bound = e.estep(siginv, sigent)
        runs.append((bound, e.get_beta_ss(), e.get_sigma_ss(), e.get_eta(), e.get_theta(), e.get_bound_docs()))
        e.close()
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("K", [130, 200])
def test_more_than_128_topics_device_mstep_and_resident_em(K):
    """K > 128 through the device M-step: the covariance kernel's 64 x 64 blocks beyond the second (n = 129: a third block row of
    one component), the moments, and a resident EM fit whose first bound equals the plain E-step's."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    syn = synthetic_corpus(300, 600, K, n_words=60, seed=K)
    c = syn.corpus
    n = K - 1
    rng = np.random.default_rng(K)
    eta = rng.normal(0, 0.5, size=(c.N, n)); mu = rng.normal(0, 0.5, size=(c.N, n))
    e = HipEstepEngine(0)
    e.set_corpus(c.indptr, c.indices, c.counts, c.V)
    e.set_topics(K)
    e.put_eta(eta); e.put_mu(mu)
    cov = e.covariance()
    e.close()
    want = (eta - mu).T @ (eta - mu)
    assert np.max(np.abs(cov - want)) <= 1e-11 * np.max(np.abs(want))
    m = STM(documents=c, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False, max_em_iter=3,
            sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    beta0, eta0, mu0 = m.beta.copy(), m.eta.copy(), m.mu.copy()
    m.expectation_maximization(saving=False)
    assert len(m.last_bounds) == 3 and np.all(np.isfinite(m.last_bounds))
    m2 = STM(documents=c, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False, max_em_iter=3,
             sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    assert np.array_equal(m2.beta, beta0) and np.array_equal(m2.eta, eta0) and np.array_equal(m2.mu, mu0)
    m2.E_step()
    assert m2.bound == pytest.approx(m.last_bounds[0], rel=1e-10)
    m.close(); m2.close()


def test_content_covariate_at_k50_against_the_reference_itself():
    """BASELINE config 4's shape (K = 50, A = 2) against the reference: E-steps teacher-forced, then the resident loop's
    device M-step for per-level beta (beta_normalise_topics_kernel) against the reference's M-step results."""
    from strutopy_amd import STM
    from strutopy_amd.engine import estep_host
    g = load_golden("content_k50")
    b0 = np.repeat(reference_beta0(50, int(g["V"]))[None], 2, axis=0)
    for it in range(2):
        p = f"it{it}_"
        beta = b0 if it == 0 else g[p + "beta_in"]
        d = estep_host(g["indptr"], g["indices"], g["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                       float(g[p + "sigmaentropy"]), aspect=g["aspect"])
        _against_reference(d, g, p, f"content k50 it{it}")
    m = STM(documents=_corpus(g), dictionary=None, content=True, K=50, X=g["X"][:, 0], kappa_interactions=True, A=2,
            beta_index=g["aspect"], max_em_iter=2, sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    m.expectation_maximization(saving=False)
    assert m.last_bounds[0] == pytest.approx(float(g["it0_bound"]), rel=1e-10)
    assert m.last_bounds[1] == pytest.approx(float(g["it1_bound"]), rel=1e-8)
    assert np.allclose(m.sigma, g["it1_sigma_out"], rtol=1e-6, atol=1e-9)
    assert np.allclose(m.beta.sum(axis=-2), g["it1_beta_out_colsum"], rtol=1e-7, atol=1e-12)
    assert np.allclose(m.beta.sum(axis=-1), g["it1_beta_out_rowsum"], rtol=1e-6, atol=1e-12)
    m.close()


def test_k100_rejects_bad_beta_like_the_reference():
    """assert np.all(beta_doc_kv >= 0) (stm.py:534) in the K > 64 solver, which checks the rows while it re-gathers them."""
    from strutopy_amd.engine import estep_host
    g = load_golden("k100_v5k")
    beta = reference_beta0(100, int(g["V"])).copy()
    args = [g["indptr"], g["indices"], g["counts"], beta, g["it0_mu_in"], g["it0_eta_in"], g["it0_siginv"], float(g["it0_sigmaentropy"])]
    w = int(g["indices"][int(g["indptr"][7]) + 3])          # a word of document 7
    for bad, k in ((-1e-12, 99), (np.nan, 3), (-0.5, 70)):
        args[3] = beta.copy()
        args[3][k, w] = bad
        with pytest.raises(AssertionError):
            estep_host(*args)
    args[3] = beta
    estep_host(*args)                                        # and the clean input still runs


# ------------------------------------------------------------------ C4's per-GPU share
def test_config4_share_invariants_and_oracle_sample(oracle):
    """BASELINE configs[3] per GPU: 125k documents, V=50k, K=100 (beta = 40 MB, beyond the L2): size-independent
    invariants on everything, the oracle on a 500-document sample."""
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    syn = synthetic_corpus(125_000, 50_000, 100, n_words=150, seed=12345)
    c, K, n = syn.corpus, 100, 99
    beta = reference_beta0(K, c.V)
    e = HipEstepEngine(0)
    e.set_corpus(c.indptr, c.indices, c.counts, c.V)
    e.set_topics(K)
    e.put_beta(beta)
    siginv, sigent = np.eye(n) / 20.0, float(n * 0.5 * np.log(20.0))
    bound = e.estep(siginv, sigent)
    eta, theta, beta_ss, sigma_ss, bd, diag = e.get_eta(), e.get_theta(), e.get_beta_ss(), e.get_sigma_ss(), e.get_bound_docs(), e.get_diagnostics()
    assert _rel(beta_ss.sum(axis=0), c.word_counts()) <= 1e-11           # phi columns sum to the word counts
    assert abs(beta_ss.sum() - c.counts.sum()) <= 1e-11 * c.counts.sum() and beta_ss.min() >= 0
    assert np.allclose(theta.sum(axis=1), 1.0, atol=1e-12) and theta.min() > 0
    assert np.allclose(theta[:, :-1] / theta[:, -1:], np.exp(eta), rtol=1e-12)
    assert np.allclose(sigma_ss, sigma_ss.T, rtol=1e-12) and np.linalg.eigvalsh(sigma_ss).min() > 0
    assert np.isfinite(bd).all() and bound == pytest.approx(bd.sum(), rel=1e-12)
    assert set(np.unique(diag["status"])) <= {0, 2} and set(np.unique(diag["pd_path"])) <= {0, 1, 2}
    S = 500
    sub = c.slice(0, S)
    z = np.zeros((S, n))
    o = oracle.estep(sub.indptr, sub.indices, sub.counts, beta, z, z, siginv, sigent, nthreads=0)
    for k in ("status", "nit", "pd_path"):
        assert np.array_equal(diag[k][:S], o[k]), k
    assert np.max(np.abs(eta[:S] - o["eta"])) <= 1e-7
    assert np.max(np.abs(bd[:S] - o["bound_doc"]) / np.abs(o["bound_doc"])) <= 1e-8
    # second E-step from the moved state (eta of the first, same beta): the trajectories of later iterations
    b2 = e.estep(siginv, sigent)
    o2 = oracle.estep(sub.indptr, sub.indices, sub.counts, beta, z, eta[:S], siginv, sigent, nthreads=0)
    d2 = e.get_diagnostics()
    assert np.array_equal(d2["status"][:S], o2["status"]) and np.array_equal(d2["nit"][:S], o2["nit"])
    assert np.max(np.abs(e.get_eta()[:S] - o2["eta"])) <= 1e-7 and np.isfinite(b2)
    e.close()


def test_config5_full_size_invariants_oracle_sample_and_mstep(oracle):
    """BASELINE configs[4] at its full size: 100k documents, V=10k, K=50 with a content covariate (A = 2 levels of beta,
    stm.py:527-530, 584-588, 741).  Size-independent invariants on everything (per-level column sums of beta_ss = per-level
    word counts), the oracle on a 1000-document sample, and one full resident iteration's beta against the NumPy
    statement of the reference's axis-1 normalisation on the device's own beta_ss."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    N, V, K, A = 100_000, 10_000, 50, 2
    syn = synthetic_corpus(N, V, K, n_words=150, seed=12345)
    c, n = syn.corpus, K - 1
    aspect = np.random.default_rng(5).integers(0, A, size=N).astype(np.int32)
    m = STM(documents=c, dictionary=None, content=True, K=K, X=syn.X, kappa_interactions=True, A=A, beta_index=aspect,
            max_em_iter=1, sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    beta0 = m.beta.copy()
    assert beta0.shape == (A, K, c.V)
    m._em_iteration_resident()
    e = m._engine
    eta, theta, beta_ss, bd, diag = e.get_eta(), e.get_theta(), e.get_beta_ss(), e.get_bound_docs(), e.get_diagnostics()
    doc = np.repeat(np.arange(N), np.diff(c.indptr))
    for a in range(A):                                     # phi columns sum to the word count, per level
        sel = aspect[doc] == a
        wc = np.bincount(c.indices[sel], weights=c.counts[sel], minlength=c.V)
        assert _rel(beta_ss[a].sum(axis=0), wc) <= 1e-11, a
    assert abs(beta_ss.sum() - c.counts.sum()) <= 1e-11 * c.counts.sum() and beta_ss.min() >= 0
    assert np.allclose(theta.sum(axis=1), 1.0, atol=1e-12) and theta.min() > 0
    assert np.allclose(theta[:, :-1] / theta[:, -1:], np.exp(eta), rtol=1e-12)
    assert np.isfinite(bd).all() and m.bound == pytest.approx(bd.sum(), rel=1e-12)
    assert set(np.unique(diag["status"])) <= {0, 2} and set(np.unique(diag["pd_path"])) <= {0, 1, 2}
    S = 1000
    sub = c.slice(0, S)
    z = np.zeros((S, n))
    o = oracle.estep(sub.indptr, sub.indices, sub.counts, beta0, z, z, m.siginv, float(m.sigmaentropy), aspect=aspect[:S], nthreads=0)
    for k in ("status", "nit", "pd_path"):
        assert np.array_equal(diag[k][:S], o[k]), k
    assert np.max(np.abs(eta[:S] - o["eta"])) <= 1e-7
    assert np.max(np.abs(bd[:S] - o["bound_doc"]) / np.abs(o["bound_doc"])) <= 1e-8
    rs = beta_ss.sum(axis=1)[:, None]                      # axis=1 of (A, K, V): over topics (stm.py:741)
    want = np.divide(beta_ss, rs, out=np.zeros_like(beta_ss), where=rs != 0)
    assert np.allclose(m.beta, want, rtol=1e-12, atol=0)
    m.close()


def test_a_rank_that_fails_before_its_first_launch_still_joins_the_collectives(monkeypatch):
    """stm_em_begin with a communicator: a rank whose host-side planning fails (an allocation, a function attribute) has
    enqueued nothing yet -- it must still enter the iteration's two all-reduces, with the error slot set, or its peers wait
    forever.  One-rank RCCL communicator: the failing call returns its own error promptly, the collectives were entered (the
    next iteration runs on the same communicator and gives the trace of an undisturbed fit)."""
    from strutopy_amd import STM, dist as sdist
    from strutopy_amd._lib import StmError
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(600, 1500, 20, n_words=80, seed=21)

    from strutopy_amd.engine import HipEstepEngine

    def model(testing=False):   # (the fault injector exists in the -DSTM_TESTING build only)
        return STM(documents=syn.corpus, dictionary=None, content=False, K=20, X=syn.X, kappa_interactions=False, max_em_iter=3,
                   sigma_prior=0, convergence_threshold=1e-12, init_type="random", comm=sdist.RcclComm(sdist.TcpGroup(0, 1)),
                   engine=HipEstepEngine(0, testing=testing))
    ref = model()
    ref.expectation_maximization(saving=False)
    m = model(testing=True)
    m._engine.debug_set("STM_DEBUG_FAIL_PLAN", 1)
    with pytest.raises(StmError, match="STM_DEBUG_FAIL_PLAN"):
        m._em_iteration_resident()
    m._engine.debug_set("STM_DEBUG_FAIL_PLAN", 0)
    with pytest.raises(ValueError, match="STM_TESTING"):      # ... and the product library refuses debug switches altogether
        ref._engine.debug_set("STM_DEBUG_FAIL_PLAN", 1)
    m.last_bounds = []
    m.expectation_maximization(saving=False)
    assert np.allclose(m.last_bounds, ref.last_bounds, rtol=1e-12)
    ref.close(); m.close()


def test_late_em_iteration_40_teacher_forced_against_the_oracle(oracle):
    """Beyond EM iteration ~30 the successful line searches get longer (10 -> 18 evaluations per document): 2000
    documents of the configs[1] shape driven to EM iteration 40 on the device, then that iteration's E-step from the
    device's own state through the HIP path and through the oracle -- every scipy status / nit / PD path equal."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(2000, 10_000, 50, n_words=150, seed=77)
    c = syn.corpus
    m = STM(documents=c, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=41,
            sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    for _ in range(40):
        m._em_iteration_resident()
    beta, mu, eta = m.beta.copy(), m.mu.copy(), m.eta.copy()
    m._em_iteration_resident()                              # EM iteration 40 on the device
    d = m.solver_diagnostics()
    siginv, sigent = m.siginv.copy(), float(m.sigmaentropy)
    o = oracle.estep(c.indptr, c.indices, c.counts, beta, mu, eta, siginv, sigent, nthreads=0)
    for k in ("status", "nit", "pd_path"):
        assert np.array_equal(d[k], o[k]), f"{k}: {int(np.sum(d[k] != o[k]))} documents differ"
    assert d["nit"].mean() > 0.5 and o["nfev"].mean() >= d["nfev"].mean()   # the regime the test is about: steps that move
    assert np.max(np.abs(m.eta - o["eta"])) <= 1e-7
    assert m.bound == pytest.approx(o["bound"], rel=1e-10)
    m.close()


# ------------------------------------------------------------------ multi-rank paths
def test_k100_resident_em_through_a_one_rank_rccl_communicator():
    """at K >= 66 the (K-1)^2 covariance did not fit the old fixed-size all-reduce region.
    K=100 with a real (one-rank) RCCL communicator, both covariance forms, and a covariate wide enough (41 one-hot
    columns) that the moment region of the packed buffer has to grow -- against the host-NumPy M-step."""
    from strutopy_amd import STM, dist as sdist
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(400, 3000, 100, n_words=120, seed=8)
    Xcat = np.random.default_rng(2).integers(0, 41, size=400)            # 41 categories -> one-hot (stm.py:665-671)
    res = {}
    for tag, X, cov in (("bin-moments", syn.X, "moments"), ("bin-exact", syn.X, "exact"), ("wide-moments", Xcat, "moments"),
                        ("wide-exact", Xcat, "exact"), ("bin-host", syn.X, None), ("wide-host", Xcat, None)):
        comm = sdist.RcclComm(sdist.TcpGroup(0, 1)) if cov else None
        m = STM(documents=syn.corpus, dictionary=None, content=False, K=100, X=X, kappa_interactions=False, max_em_iter=2,
                sigma_prior=0, convergence_threshold=1e-12, init_type="random", comm=comm)
        if cov:
            assert m._engine._h and m.comm.kind == "rccl"
            m.cov_exchange = cov
            m.expectation_maximization(saving=False)
            assert m.cov_exchanges == [cov, cov]
        else:
            m.expectation_maximization(saving=False, resident=False)
        res[tag] = (np.array(m.last_bounds), m.sigma.copy(), m.gamma.copy(), m.beta.copy())
        m.close()
    for kind in ("bin", "wide"):
        ref = res[kind + "-host"]
        for form in ("moments", "exact"):
            got = res[f"{kind}-{form}"]
            assert np.allclose(got[0], ref[0], rtol=1e-9), (kind, form)
            assert np.allclose(got[1], ref[1], rtol=1e-6, atol=1e-9), (kind, form)
            assert np.allclose(got[3], ref[3], rtol=1e-6, atol=1e-12), (kind, form)
        assert res[kind + "-moments"][2].shape == ref[2].shape
    assert np.allclose(res["bin-moments"][2], res["bin-host"][2], rtol=1e-6, atol=1e-8)


def test_single_and_split_exchange_give_the_same_fit_through_a_one_rank_rccl_communicator():
    """STM(exchange="single"): ONE ncclAllReduce of the whole packed buffer [bound | sigma_ss | moments | beta_ss] per EM iteration (the
    exchange BASELINE.json's north_star names) instead of the default two ("split": beta_ss behind the host's read-back).  Same sums:
    with a real one-rank communicator both forms -- and a switch from one to the other in the middle of a fit -- give the trace and
    the parameters of the fit without a communicator, bit for bit."""
    from strutopy_amd import STM, dist as sdist
    from strutopy_amd.corpus import synthetic_corpus
    syn = synthetic_corpus(1500, 3000, 50, n_words=100, seed=5)
    res = {}
    for tag in ("none", "split", "single", "switch"):
        comm = None if tag == "none" else sdist.RcclComm(sdist.TcpGroup(0, 1))
        m = STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=4,
                sigma_prior=0, convergence_threshold=1e-12, init_type="random", comm=comm, exchange="single" if tag == "single" else "split")
        if tag == "switch":
            for it in range(4):
                m.exchange = "single" if it % 2 else "split"
                m._em_iteration_resident()
        else:
            m.expectation_maximization(saving=False)
        if comm is not None:
            assert m.comm.kind == "rccl" and m._engine.comm_info()["nranks"] == 1
        res[tag] = (np.array(m.last_bounds), m.sigma.copy(), m.beta.copy(), m.eta.copy())
        m.close()
    with pytest.raises(ValueError):
        STM(documents=syn.corpus, dictionary=None, content=False, K=50, X=syn.X, kappa_interactions=False, max_em_iter=1, sigma_prior=0,
            convergence_threshold=1e-12, init_type="random", exchange="both")
    for tag in ("split", "single", "switch"):
        for a, b in zip(res[tag], res["none"]):
            assert np.array_equal(a, b), tag


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RANK_SCRIPT = r'''
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
from strutopy_amd import STM, dist as sdist
from strutopy_amd.corpus import PackedCorpus
g = np.load(os.path.join(%(root)r, "tests", "golden", "c1_k10.npz"))
full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
group = sdist.init_from_env(timeout=120)
comm = sdist.HostComm(group)
lo, hi = sdist.shard_bounds(full.indptr, group.size)[group.rank]
m = STM(documents=full.slice(lo, hi), dictionary=None, content=False, K=10, X=g["X"][lo:hi, 0], kappa_interactions=False,
        max_em_iter=3, sigma_prior=0, convergence_threshold=1e-12, init_type="random", comm=comm, device=0)
assert type(m._engine).__name__ == "HipEstepEngine" and m.N_total == full.N
m.expectation_maximization(saving=False)
np.savez(%(out)r + str(group.rank) + ".npz", lo=lo, hi=hi, bounds=np.array(m.last_bounds), sigma=m.sigma, beta=m.beta, eta=m.eta, mu=m.mu,
         gamma=m.gamma, maps=np.array("libstm_hip.so" in open("/proc/self/maps").read()))
comm.barrier()
m.close()
'''


def test_two_ranks_on_one_gpu_reproduce_the_single_process_fit(tmp_path):
    """Two processes, each with its own stm_handle on GPU 0 and a contiguous document shard, exchanging the packed
    sufficient statistics through the host group (TcpGroup: RCCL refuses two ranks on one device): exercises
    stm_put_sigma_ss / stm_put_beta_ss, the sharded resident loop and the M-step from reduced moments on the HIP engine."""
    from strutopy_amd import STM
    g = load_golden("c1_k10")
    port = _free_port()
    script = _RANK_SCRIPT % dict(root=ROOT, out=str(tmp_path / "rank"))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", STM_RDZV_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    ref = STM(documents=_corpus(g), dictionary=None, content=False, K=10, X=g["X"][:, 0], kappa_interactions=False,
              max_em_iter=3, sigma_prior=0, convergence_threshold=1e-12, init_type="random")
    ref.expectation_maximization(saving=False)
    res = [np.load(str(tmp_path / f"rank{r}.npz")) for r in range(2)]
    assert int(res[0]["lo"]) == 0 and int(res[0]["hi"]) == int(res[1]["lo"]) and int(res[1]["hi"]) == ref.N
    for r in res:
        lo, hi = int(r["lo"]), int(r["hi"])
        assert bool(r["maps"])
        # free-running EM amplifies the 1e-16 differences of the two summation orders (the reference itself moves by 1e-9 at EM
        # iteration 2 under a 1e-15 perturbation): tight at iterations 0-1, 1e-8 at iteration 2
        assert np.allclose(r["bounds"][:2], ref.last_bounds[:2], rtol=1e-10) and np.allclose(r["bounds"], ref.last_bounds, rtol=1e-8)
        assert np.allclose(r["sigma"], ref.sigma, rtol=1e-5, atol=1e-8)
        assert np.allclose(r["beta"], ref.beta, rtol=1e-5, atol=1e-11)
        assert np.allclose(r["gamma"], ref.gamma, rtol=1e-5, atol=1e-7)
        assert np.allclose(r["eta"], ref.eta[lo:hi], atol=1e-6) and np.allclose(r["mu"], ref.mu[lo:hi], atol=1e-6)
    assert np.array_equal(res[0]["sigma"], res[1]["sigma"]) and np.array_equal(res[0]["beta"], res[1]["beta"])
    for it in range(3):
        assert res[0]["bounds"][it] == pytest.approx(float(g[f"it{it}_bound"]), rel=1e-8)
    ref.close()


def test_bench_launches_its_own_ranks_and_strong_scaling_shards_one_corpus():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (device wrap-around on a one-GPU box, host
    reduction because RCCL refuses duplicate devices) and reports n_gpus from the communicator; `--scaling strong`
    shards ONE corpus, so its ELBO trace equals the single-rank one."""
    def run(*extra):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "3000", "--vocab", "2000", "--topics", "20", "--steps", "2",
               "--warmup", "1", "--cpu-sample", "0", *extra]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "STM_RDZV_PORT")}
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])
    one = run("--gpus", "1", "--scaling", "strong")
    two = run("--gpus", "2", "--scaling", "strong")
    weak = run("--gpus", "2")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and weak["n_gpus"] == 2
    assert two["scaling"] == "strong" and weak["scaling"] == "weak"
    assert two["config"]["docs_total"] == 3000 and weak["config"]["docs_total"] == 6000
    assert np.allclose(two["elbo_trace"], one["elbo_trace"], rtol=1e-9)
    assert weak["strong_scaling"]["docs_total"] == 3000
    assert np.allclose(weak["strong_scaling"]["elbo_trace"], one["elbo_trace"], rtol=1e-9)
    assert two["config"]["allreduces_per_iteration"] == 1
    assert "roofline" in one and "roofline" not in two
    # provenance: two ranks on the one GPU of this box are not two GPUs, and the line says so
    assert one["config"]["devices_distinct"] and not two["config"]["devices_distinct"]
    assert two["config"]["device_ordinals"] == [0, 0] and two["config"]["rccl_comm_count"] == [0] and two["config"]["allreduce"] == "tcp-host"
    for b in (one, two, weak):
        assert b["value"] == pytest.approx(b["config"]["docs_total"] * b["steps"] / (b["ms_per_step"] * 1e-3 * b["steps"]), rel=1e-6)
    # --gpus 1 --allreduce rccl: a real one-rank communicator, so the with-communicator iteration (small all-reduce, read-back,
    # beta_ss pass + its all-reduce) is what runs -- same trace as without one
    rc = run("--gpus", "1", "--scaling", "strong", "--allreduce", "rccl")
    assert rc["config"]["allreduce"] == "rccl" and rc["config"]["rccl_comm_count"] == [1] and rc["config"]["allreduces_per_iteration"] == 2
    assert np.allclose(rc["elbo_trace"], one["elbo_trace"], rtol=1e-12)


def test_bench_under_torch_distributed_run_like_the_driver():
    """The driver's own launch of the multi-GPU bench: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- ranks from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, no STM_RDZV_PORT: the product's TCP group
    finds its peers through the rendezvous file keyed by MASTER_ADDR / MASTER_PORT.  Two ranks on the one GPU of this box (host reduction:
    RCCL refuses duplicate devices); ONE JSON line, from rank 0, with the whole job's documents."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--docs", "3000", "--vocab", "2000", "--topics", "20", "--cpu-sample", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "STM_RDZV_PORT", "STM_RDZV_SECRET")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["docs_total"] == 6000 and d["steps"] == 2
    assert d["value"] == pytest.approx(6000 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)       # whole-job throughput
    assert len(d["elbo_trace"]) == 2 and np.all(np.isfinite(d["elbo_trace"]))
    assert d["strong_scaling"]["docs_total"] == 3000


def test_bench_line_exchange_and_counter_provenance():
    """`bench.py --gpus 1 --allreduce rccl --exchange single|split`: the line says which exchange ran (one all-reduce of the packed buffer per
    EM iteration, or two) through a real one-rank communicator, with the same ELBO trace; `roofline.traffic` / `.compute` are measured by the
    run itself (rocprofv3 --pmc child passes) unless switched off, and the line says which it was."""
    def run(*extra):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "4000", "--vocab", "3000", "--topics", "50", "--steps", "2", "--warmup", "1",
               "--cpu-sample", "0", "--late-sample", "0", *extra]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "STM_RDZV_PORT")}
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    split = run("--gpus", "1", "--allreduce", "rccl", "--live-traffic", "off")
    single = run("--gpus", "1", "--allreduce", "rccl", "--exchange", "single", "--live-traffic", "off")
    assert split["config"]["allreduce"] == "rccl" and split["config"]["rccl_comm_count"] == [1]
    assert split["config"]["exchange"] == "split" and split["config"]["allreduces_per_iteration"] == 2
    assert single["config"]["exchange"] == "single" and single["config"]["allreduces_per_iteration"] == 1
    assert single["elbo_trace"] == split["elbo_trace"]
    assert split["roofline"]["traffic_source"] is None or split["roofline"]["traffic_source"].startswith("profiles/")   # (no committed entry for this workload)
    import shutil
    live = run("--gpus", "1")
    r = live["roofline"]
    assert r["dominant_single_kernel"]["kernel"] in ("stm::solver_kernel", "stm::post_kernel") and 0 < r["dominant_single_kernel"]["frac"] < 1
    if shutil.which("rocprofv3"):
        assert r["traffic_source"].startswith("live:") and r["compute_source"].startswith("live:")
        assert r["traffic"] > 0 and all(k["traffic"] > 0 and k["compute"]["fp64_pipe_busy"] > 0 for k in r["kernels"].values())


def test_explicit_rccl_on_duplicate_devices_fails_cleanly():
    """`--allreduce rccl` asked for by name must not fall back silently: two ranks on this box's single GPU make RCCL refuse
    (duplicate device) -- or, if it cannot even be loaded, say so -- and the launcher exits non-zero with the reason,
    promptly (no rank left hanging in a collective).  The only N > 1 behaviour of RCCL that one GPU can show."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "1000", "--vocab", "1000", "--topics", "10", "--steps", "1",
           "--warmup", "0", "--cpu-sample", "0", "--gpus", "2", "--allreduce", "rccl"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "STM_RDZV_PORT")}
    env.update(NCCL_DEBUG="WARN")
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0, r.stdout[-500:]
    assert time.time() - t < 240
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no bench line from a failed set-up
    assert re.search(r"ncclCommInitRank|RCCL|rccl|duplicate", r.stderr), r.stderr[-2000:]


# ------------------------------------------------------------------ the reference-side stub of INTEGRATION.md, executed
def test_integration_stub_runs_against_a_reference_shaped_object():
    """The ctypes stub INTEGRATION.md hands to a reference maintainer (section 2) is executed as written -- only the
    library path is made absolute -- bound to a minimal object carrying the attributes the reference's E_step reads
    (stm.py:489-597), and must reproduce the reference's own E-step results (tests/golden/c1_k10.npz)."""
    from strutopy_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class _Args(C.Structure)" in b)
    stub = stub.replace('C.CDLL("libstm_hip.so")', f"C.CDLL({_lib.LIB_PATH!r})")
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    g = load_golden("c1_k10")

    class RefShaped:      # what STM.__init__ of the reference leaves on self (stm.py:366-399, 412-476)
        pass
    m = RefShaped()
    m.documents = _corpus(g).to_bow()
    m.N, m.K, m.V = len(m.documents), int(g["K"]), int(g["V"])
    m.beta, m.mu, m.eta = g["beta0"].copy(), g["it0_mu_in"].copy(), g["it0_eta_in"].copy()
    m.sigma = g["it0_sigma_in"].copy()
    m.betaindex, m.last_bounds = None, []
    beta_ss, sigma_ss = ns["E_step"](m)
    assert m.bound == pytest.approx(float(g["it0_bound"]), rel=1e-10) and m.last_bounds == [m.bound]
    assert np.allclose(beta_ss, g["it0_beta_ss"], rtol=1e-7, atol=1e-12)
    assert np.allclose(sigma_ss, g["it0_sigma_ss"], rtol=1e-7)
    assert np.max(np.abs(m.eta - g["it0_eta"])) <= 1e-7 and np.max(np.abs(m.theta - g["it0_theta"])) <= 1e-7
    assert np.allclose(m.siginv, g["it0_siginv"])
    # content covariate through the same stub (A = 2 levels, stm.py:527-528)
    g = load_golden("content_a2")
    m = RefShaped()
    m.documents = _corpus(g).to_bow()
    m.N, m.K, m.V = len(m.documents), int(g["K"]), int(g["V"])
    m.beta, m.mu, m.eta, m.sigma = g["beta0"].copy(), g["it0_mu_in"].copy(), g["it0_eta_in"].copy(), g["it0_sigma_in"].copy()
    m.betaindex, m.last_bounds = g["aspect"], []
    beta_ss, _ = ns["E_step"](m)
    assert m.bound == pytest.approx(float(g["it0_bound"]), rel=1e-10)
    assert np.allclose(beta_ss, g["it0_beta_ss"], rtol=1e-7, atol=1e-12)
    bad = RefShaped()
    bad.__dict__.update(m.__dict__)
    bad.beta = m.beta.copy(); bad.beta[0, 1, int(g["indices"][0])] = -1.0
    bad.betaindex = np.ones(m.N, dtype=np.int64) * 0
    with pytest.raises(AssertionError):       # stm.py:534
        ns["E_step"](bad)


# ------------------------------------------------------------------ where parity ends: the long-run regime
def test_long_run_regime_of_config5_is_within_the_oracles_own_sensitivity(oracle, monkeypatch):
    """Config 5's shape (content covariate, A = 2 levels of beta) scaled to 4000 documents and driven on the device until the
    fit is in the long-run regime -- mean scipy `nit` >= 8: every document takes about ten BFGS iterations, and ten
    iterations amplify a last-bit difference into one accepted step more or less (DESIGN.md sections 2 and 7; profiles/HISTORY.md section 9).  One teacher-forced
    E-step from that state, four ways:
      (i)  the HIP path with and without its outcome-preserving line-search cuts (STM_DEBUG_FLAGS = 0 / 6) -- identical
           status / nit / eta, bit for bit: the cuts are not what moves a document there;
      (ii) the HIP path against the oracle, and the oracle against itself from a start moved by a relative 1e-13: the
           number of documents whose nit / status differ between GPU and oracle must not exceed what the oracle's own
           sensitivity to a perturbation of the last bits produces (floor: three documents), and eta must
           agree as well as the oracle agrees with itself."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    N, V, K, A = 4000, 10_000, 50, 2
    syn = synthetic_corpus(N, V, K, n_words=150, seed=12345)
    c = syn.corpus
    aspect = np.random.default_rng(777).integers(0, A, size=N).astype(np.int32)
    from strutopy_amd.engine import HipEstepEngine
    m = STM(documents=c, dictionary=None, content=True, K=K, X=syn.X, kappa_interactions=True, A=A, beta_index=aspect,
            max_em_iter=200, sigma_prior=0, convergence_threshold=1e-12, init_type="random", engine=HipEstepEngine(0, testing=True))
    its, mean_nit = 0, 0.0
    while its < 120 and mean_nit < 8.0:
        m._em_iteration_resident()
        its += 1
        if its >= 10:
            mean_nit = float(m.solver_diagnostics()["nit"].mean())
    assert mean_nit >= 8.0, f"the fit never reached the long-run regime: mean nit {mean_nit:.2f} after {its} EM iterations"
    beta, mu, eta = m.beta.copy(), m.mu.copy(), m.eta.copy()
    m._preamble()
    siginv, sigent = m.siginv.copy(), float(m.sigmaentropy)
    res = {}
    for flags in ("0", "6"):
        m._engine.debug_set("STM_DEBUG_FLAGS", int(flags))
        m.eta = eta.copy()
        m._estep_device()
        d = m.solver_diagnostics()
        res[flags] = (d["status"].copy(), d["nit"].copy(), m.eta.copy(), d["nfev"].copy())
    m._engine.debug_set("STM_DEBUG_FLAGS", 0)
    assert np.array_equal(res["0"][0], res["6"][0]) and np.array_equal(res["0"][1], res["6"][1])
    assert np.array_equal(res["0"][2], res["6"][2])                      # the same bits
    assert res["6"][3].mean() > res["0"][3].mean()                       # (and the cuts did skip evaluations)
    o = oracle.estep(c.indptr, c.indices, c.counts, beta, mu, eta, siginv, sigent, aspect=aspect, nthreads=0)
    op = oracle.estep(c.indptr, c.indices, c.counts, beta, mu, eta * (1.0 + 1e-13), siginv, sigent, aspect=aspect, nthreads=0)
    assert o["nit"].mean() >= 8.0
    self_nit, self_status = int(np.sum(op["nit"] != o["nit"])), int(np.sum(op["status"] != o["status"]))
    self_eta = float(np.max(np.abs(op["eta"] - o["eta"])))
    gpu_nit, gpu_status = int(np.sum(res["0"][1] != o["nit"])), int(np.sum(res["0"][0] != o["status"]))
    gpu_eta = float(np.max(np.abs(res["0"][2] - o["eta"])))
    print(f"long-run regime after {its} EM iterations: mean nit {o['nit'].mean():.2f}; GPU vs oracle: {gpu_nit} nit / {gpu_status} status differ, "
          f"eta {gpu_eta:.2e}; oracle vs oracle(1 + 1e-13): {self_nit} / {self_status}, eta {self_eta:.2e}")
    assert gpu_nit <= max(2 * self_nit, 3) and gpu_status <= max(2 * self_status, 3)
    assert gpu_eta <= max(10 * self_eta, 1e-6)
    m.close()


def test_long_run_regime_against_the_reference_on_the_gpu():
    """tests/golden/c5_long.npz: the imported reference, teacher-forced for one E-step on the state of a config-5-shaped device fit at
    EM iteration 26 (mean scipy nit 12).  The bar is the reference's own sensitivity there (it differs from itself in `ref_self_nit`
    of 300 documents when its start moves by a relative 1e-13): the HIP path may differ from the reference in at most
    max(2 x that, 3) documents, and on the documents that take the reference's path eta agrees within 10 x its eta sensitivity."""
    from strutopy_amd.engine import estep_host
    g = load_golden("c5_long")
    d = estep_host(g["indptr"], g["indices"], g["counts"], g["beta"], g["mu"], g["eta"], g["siginv"], float(g["sigmaentropy"]), aspect=g["aspect"])
    nit_bar, status_bar = max(2 * int(g["ref_self_nit"]), 3), max(2 * int(g["ref_self_status"]), 3)
    d_nit, d_status = int(np.sum(d["nit"] != g["out_nit"])), int(np.sum(d["status"] != g["out_status"]))
    same = (d["nit"] == g["out_nit"]) & (d["status"] == g["out_status"])
    d_eta = float(np.max(np.abs(d["eta"] - g["out_eta"])[same]))
    print(f"long-run golden: HIP vs reference {d_nit} nit / {d_status} status of {len(same)} documents differ, eta {d_eta:.2e} on the same path; "
          f"reference vs itself {int(g['ref_self_nit'])} / {int(g['ref_self_status'])}, {float(g['ref_self_eta']):.2e}; oracle vs reference "
          f"{int(g['ref_vs_oracle_nit'])} / {int(g['ref_vs_oracle_status'])}")
    assert g["out_nit"].mean() >= 8.0
    assert d_nit <= nit_bar and d_status <= status_bar and d_eta <= 10.0 * float(g["ref_self_eta"])
    assert np.array_equal(d["pd_path"], g["out_pd_path"])
    assert abs(d["bound"] - float(g["out_bound"])) <= 1e-8 * abs(float(g["out_bound"]))
    rel = float(np.max(np.abs(d["sigma_ss"] - g["out_sigma_ss"])) / np.max(np.abs(g["out_sigma_ss"])))
    assert rel <= 1e-6
    assert np.allclose(d["beta_ss"].sum(axis=-1), g["out_beta_ss_rowsum"], rtol=1e-7) and np.allclose(d["beta_ss"].sum(axis=-2), g["out_beta_ss_colsum"], rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------ config 4 as a corpus: the shards' statistics add up
def test_config4_shape_cut_into_eight_shards_adds_up():
    """BASELINE configs[3] (V = 50k, K = 100) is a document-sharded 8-GPU run: shard_bounds(indptr, 8), one E-step per shard,
    one all-reduce.  On one GPU: the whole corpus' E-step against the eight shards run one after another through their own
    handles -- every document's eta / status / nit / bound the same bits in its shard as in the whole corpus, the shards'
    bound / sigma_ss / beta_ss sums equal to the whole corpus' to 1e-12, nnz balance within 1 %.  Here at 80k documents (two
    E-steps: the cold start and the warm second one); tools/c4_corpus_shards.py runs the same check at the configuration's full
    1M documents (profiles/r05_c4_corpus_shards.json)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c4_corpus_shards.py"), "80000", "50000", "100", "8"],
                       capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1])
    assert r.returncode == 0 and res["ok"], res
    assert res["nnz_imbalance"] <= 0.01 and len(res["shards"]) == 8
    for e in res["estep"]:
        assert all(e["per_document_bits_equal"].values()), e
        assert max(e["bound_rel"], e["sigma_ss_rel"], e["beta_ss_rel"]) <= 1e-12, e
