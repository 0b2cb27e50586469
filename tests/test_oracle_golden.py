"""The CPU oracle (oracle/stm_oracle.c) pinned against the golden vectors.

The goldens under tests/golden/ were produced by importing the reference
(tools/make_golden.py, numpy/scipy versions recorded in every file), so these
tests are what makes the oracle trustworthy as the checker of the HIP path.
Tolerances follow SURVEY.md appendix A.5: status / nit / PD path exact,
eta <= 1e-7 abs, per-document bound <= 1e-9 rel, total ELBO <= 1e-10 rel;
nfev / njev are informational (the terminal line search runs in rounding noise).
"""
import numpy as np
import pytest

from conftest import load_golden, reference_beta0

CASES = [("toy_ctm", 2), ("edge", 2), ("content_a2", 2), ("c1_k10", 3), ("k50_v10k", 2), ("wiki_k50", 2)]


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def _beta_for(g, it, prev):
    """beta entering iteration `it`: stored beta0, the reference's seeded random init, or the
    row-normalised beta_ss of the previous iteration (stm.py:741-745)."""
    if it == 0:
        if "beta0" in g.files:
            return g["beta0"]
        return reference_beta0(int(g["K"]), int(g["V"]))
    key = f"it{it - 1}_beta_out"
    if key in g.files:
        return g[key]
    bss = prev["beta_ss"]
    rs = bss.sum(axis=1)[:, None]
    return np.divide(bss, rs, out=np.zeros_like(bss), where=rs != 0)


@pytest.mark.parametrize("name,its", CASES)
def test_estep_matches_reference(oracle, name, its):
    g = load_golden(name)
    aspect = g["aspect"] if "aspect" in g.files else None
    prev = None
    for it in range(its):
        p = f"it{it}_"
        beta = _beta_for(g, it, prev)
        o = oracle.estep(g["indptr"], g["indices"], g["counts"], beta, g[p + "mu_in"], g[p + "eta_in"],
                         g[p + "siginv"], float(g[p + "sigmaentropy"]), aspect=aspect, nthreads=0)
        prev = o
        assert np.array_equal(o["status"], g[p + "status"]), f"{name} it{it}: scipy status differs"
        assert np.array_equal(o["nit"], g[p + "nit"]), f"{name} it{it}: BFGS iteration counts differ"
        assert np.array_equal(o["pd_path"], g[p + "pd_path"]), f"{name} it{it}: PD-fix path differs"
        assert np.max(np.abs(o["eta"] - g[p + "eta"])) <= 1e-7
        assert np.max(np.abs(o["theta"] - g[p + "theta"])) <= 1e-7
        assert np.max(np.abs(o["bound_doc"] - g[p + "bound_doc"]) / np.abs(g[p + "bound_doc"])) <= 1e-9
        assert abs(o["bound"] - float(g[p + "bound"])) <= 1e-10 * abs(float(g[p + "bound"]))
        assert _rel(o["sigma_ss"], g[p + "sigma_ss"]) <= 1e-8
        if p + "beta_ss" in g.files:
            assert _rel(o["beta_ss"], g[p + "beta_ss"]) <= 1e-8
        else:  # large cases store marginals + a 64-column sample
            assert _rel(o["beta_ss"].sum(axis=1), g[p + "beta_ss_rowsum"]) <= 1e-9
            assert _rel(o["beta_ss"].sum(axis=0), g[p + "beta_ss_colsum"]) <= 1e-8
            assert _rel(o["beta_ss"][:, g["sample_cols"]], g[p + "beta_ss_cols"]) <= 1e-8
        assert _rel(o["phi_last"], g[p + "phi_last"]) <= 1e-8


def test_k50_later_iterations_teacher_forced(oracle):
    """tests/golden/k50_late.npz (tools/make_golden.py k50_late): EM iterations 3, 4, 5, 8 of the reference at
    K=50 / V=10k with each E-step's full input state -- the regime where about half of the documents take BFGS
    steps that move (iterations 0-1 of the other K=50 goldens never do)."""
    g = load_golden("k50_late")
    moved = 0
    for it in g["kept"]:
        p = f"it{int(it)}_"
        o = oracle.estep(g["indptr"], g["indices"], g["counts"], g[p + "beta_in"], g[p + "mu_in"], g[p + "eta_in"],
                         g[p + "siginv"], float(g[p + "sigmaentropy"]), nthreads=0)
        for k in ("status", "nit", "pd_path"):
            assert np.array_equal(o[k], g[p + k]), f"it{it}: {k}"
        assert np.max(np.abs(o["eta"] - g[p + "eta"])) <= 1e-7
        assert np.max(np.abs(o["bound_doc"] - g[p + "bound_doc"]) / np.abs(g[p + "bound_doc"])) <= 1e-9
        assert abs(o["bound"] - float(g[p + "bound"])) <= 1e-10 * abs(float(g[p + "bound"]))
        assert _rel(o["sigma_ss"], g[p + "sigma_ss"]) <= 1e-8
        assert _rel(o["beta_ss"].sum(axis=1), g[p + "beta_ss_rowsum"]) <= 1e-9
        moved += int(np.sum(g[p + "nit"] > 0))
    assert moved > 2000          # it 4: 41 %, it 5: 100 %, it 8: 116 % BFGS iterations per document


def _teacher_forced(g, its, beta0, aspect=None):
    for it in range(its):
        p = f"it{it}_"
        beta = beta0 if it == 0 else g[p + "beta_in"]
        yield it, p, (g["indptr"], g["indices"], g["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                      float(g[p + "sigmaentropy"]))


def _check_against_reference(o, g, p, tag):
    for k in ("status", "nit", "pd_path"):
        assert np.array_equal(o[k], g[p + k]), f"{tag}: {k}"
    assert np.max(np.abs(o["eta"] - g[p + "eta"])) <= 1e-7, tag
    assert np.max(np.abs(o["bound_doc"] - g[p + "bound_doc"]) / np.abs(g[p + "bound_doc"])) <= 1e-9, tag
    assert abs(o["bound"] - float(g[p + "bound"])) <= 1e-10 * abs(float(g[p + "bound"])), tag
    assert _rel(o["sigma_ss"], g[p + "sigma_ss"]) <= 1e-8, tag
    assert _rel(o["beta_ss"].sum(axis=-1), g[p + "beta_ss_rowsum"]) <= 1e-9, tag
    assert _rel(o["beta_ss"].sum(axis=-2), g[p + "beta_ss_colsum"]) <= 1e-8, tag


def test_k100_against_the_reference(oracle):
    """tests/golden/k100_v5k.npz: K = 100 (BASELINE config 3's topic count) run by the reference itself, EM iterations 0-2."""
    g = load_golden("k100_v5k")
    for it, p, args in _teacher_forced(g, 3, reference_beta0(100, int(g["V"]))):
        _check_against_reference(oracle.estep(*args, nthreads=0), g, p, f"k100 it{it}")


def test_content_covariate_at_k50_against_the_reference(oracle):
    """tests/golden/content_k50.npz: BASELINE config 4's shape (K = 50, A = 2 levels of beta) run by the reference."""
    g = load_golden("content_k50")
    b0 = reference_beta0(50, int(g["V"]))
    for it, p, args in _teacher_forced(g, 2, np.repeat(b0[None], 2, axis=0)):
        _check_against_reference(oracle.estep(*args, aspect=g["aspect"], nthreads=0), g, p, f"content k50 it{it}")


def test_wiki_known_answer_shipped_by_reference(oracle):
    """ELBO[0] of src/artifacts/reference_model/50/lower_bound.pickle: the one number the
    reference itself ships for this path."""
    g = load_golden("wiki_k50")
    shipped = float(g["shipped_lower_bound"][0])
    assert abs(shipped - (-855111.02)) < 0.01
    beta = reference_beta0(50, int(g["V"]))
    o = oracle.estep(g["indptr"], g["indices"], g["counts"], beta, g["it0_mu_in"], g["it0_eta_in"],
                     g["it0_siginv"], float(g["it0_sigmaentropy"]), nthreads=0)
    assert abs(o["bound"] - shipped) <= 1e-10 * abs(shipped)


def test_wiki_k70_known_answer_and_teacher_forced_iteration(oracle):
    """The second of the two numbers the reference ships for this path: wiki corpus at K = 70,
    src/artifacts/reference_model/70/lower_bound.pickle[0] = -868098.47 (producer src/03_fit_reference_model.py:40-74).
    K = 70 is a K > 64 shape on real data (N_d ~ 60); EM iteration 1 is teacher-forced with the reference's own beta."""
    g = load_golden("wiki_k70")
    c = load_golden(str(g["corpus"]))
    shipped = float(g["shipped_lower_bound"][0])
    assert abs(shipped - (-868098.47)) < 0.01
    for it, beta in ((0, reference_beta0(70, int(g["V"]))), (1, g["it1_beta_in"])):
        p = f"it{it}_"
        o = oracle.estep(c["indptr"], c["indices"], c["counts"], beta, g[p + "mu_in"], g[p + "eta_in"],
                         g[p + "siginv"], float(g[p + "sigmaentropy"]), nthreads=0)
        _check_against_reference(o, g, p, f"wiki k70 it{it}")
        assert _rel(o["beta_ss"][:, g["sample_cols"]], g[p + "beta_ss_cols"]) <= 1e-8
        if it == 0:
            assert abs(o["bound"] - shipped) <= 1e-10 * abs(shipped)


def test_pivot_tolerance_band_is_inert_on_baseline_shapes(oracle):
    """PIVOT_TOL (a Cholesky pivot below 32 ulp of its diagonal entry counts as failed: oracle chol_lower, kernels
    stm_post_common.h) is the one deliberate deviation from stm.py:1017-1021 / np.linalg.cholesky's "pivot <= 0" shared by
    the oracle and the kernels.  On every BASELINE shape no pivot of any rung of any document's PD ladder -- accepted or
    rejected -- comes within 1e3 x that band: the deviation cannot have changed an outcome there."""
    from strutopy_amd.corpus import synthetic_corpus
    band = 1e3 * 32.0 * 2.220446049250313e-16
    worst = {}
    g = load_golden("c2_full")                      # configs[1] at full size, EM iteration 0 (eta = mu = 0, the seeded beta)
    K = int(g["K"])
    c = synthetic_corpus(int(g["n_docs"]), int(g["V_requested"]), K, n_words=int(g["n_words"]), seed=int(g["seed"])).corpus
    z = np.zeros((c.N, K - 1))
    o = oracle.estep(c.indptr, c.indices, c.counts, reference_beta0(K, c.V), z, z, g["it0_siginv"], float(g["it0_sigmaentropy"]), nthreads=0)
    assert np.array_equal(o["pd_path"], g["it0_pd_path"])
    worst["c2_full it0"] = float(o["pivot_margin"].min())
    for name, K in (("wiki_k50", 50), ("wiki_k70", 70)):
        g = load_golden(name)
        c = load_golden(str(g["corpus"])) if "corpus" in g.files else g
        beta = reference_beta0(K, int(g["V"]))
        for it in range(2):
            p = f"it{it}_"
            o = oracle.estep(c["indptr"], c["indices"], c["counts"], beta, g[p + "mu_in"], g[p + "eta_in"], g[p + "siginv"],
                             float(g[p + "sigmaentropy"]), nthreads=0)
            worst[f"{name} it{it}"] = float(o["pivot_margin"].min())
            rs = o["beta_ss"].sum(axis=1)[:, None]
            beta = g["it1_beta_in"] if "it1_beta_in" in g.files else np.divide(o["beta_ss"], rs, out=np.zeros_like(o["beta_ss"]), where=rs != 0)
    g = load_golden("k100_v5k")
    for it, p, args in _teacher_forced(g, 3, reference_beta0(100, int(g["V"]))):
        worst[f"k100_v5k it{it}"] = float(oracle.estep(*args, nthreads=0)["pivot_margin"].min())
    g = load_golden("k50_late")
    for it in g["kept"]:
        p = f"it{int(it)}_"
        o = oracle.estep(g["indptr"], g["indices"], g["counts"], g[p + "beta_in"], g[p + "mu_in"], g[p + "eta_in"],
                         g[p + "siginv"], float(g[p + "sigmaentropy"]), nthreads=0)
        worst[f"k50_late it{int(it)}"] = float(o["pivot_margin"].min())
    assert all(v > band for v in worst.values()), worst


def test_toy_pipeline_final_bound(oracle):
    """tests/test_integration.py::_run_toy_pipeline of the reference: final_bound after 2 EM its."""
    g = load_golden("toy_ctm")
    assert float(g["final_bound"]) == pytest.approx(-8185.904356877947, rel=1e-14)
    o = oracle.estep(g["indptr"], g["indices"], g["counts"], g["it0_beta_out"], g["it1_mu_in"], g["it1_eta_in"],
                     g["it1_siginv"], float(g["it1_sigmaentropy"]))
    assert abs(o["bound"] - float(g["final_bound"])) <= 1e-10 * abs(float(g["final_bound"]))


def test_hessian_cholesky_nu_per_document(oracle):
    for name in ("toy_ctm", "edge"):
        g = load_golden(name)
        o = oracle.estep(g["indptr"], g["indices"], g["counts"], g["beta0"], g["it0_mu_in"], g["it0_eta_in"],
                         g["it0_siginv"], float(g["it0_sigmaentropy"]), dump_mats=True)
        assert _rel(o["hess"], g["it0_hess"]) <= 1e-8
        assert _rel(o["chol"], g["it0_chol"]) <= 1e-8
        assert _rel(o["nu"], g["it0_nu"]) <= 1e-7


@pytest.mark.parametrize("dense", [False, True])
def test_objective_gradient_and_bfgs(oracle, dense):
    """f / df (stm.py:920-958) at random points and scipy BFGS end points, diagonal siginv (what
    stm.py:501 produces) and a dense one."""
    g = load_golden("functions")
    K = int(g["K"])
    sfx = "_dense" if dense else ""
    siginv = g["siginv_dense"] if dense else g["siginv"]
    indptr = g["indptr"]
    for d in range(len(indptr) - 1):
        sl = slice(indptr[d], indptr[d + 1])
        betad = np.ascontiguousarray(g["beta0"][:, g["indices"][sl]])
        cnt, mu = g["counts"][sl], g["mus"][d]
        for j in range(4):
            fv = oracle.f(K, g["etas"][d, j], mu, cnt, betad, siginv)
            assert abs(fv - g["fvals" + sfx][d, j]) <= 1e-12 * max(1.0, abs(g["fvals" + sfx][d, j]))
            gv = oracle.df(K, g["etas"][d, j], mu, cnt, betad, siginv)
            assert np.max(np.abs(gv - g["gvals" + sfx][d, j])) <= 1e-10 * max(1.0, np.max(np.abs(g["gvals" + sfx][d, j])))
        r = oracle.bfgs(K, g["etas"][d, 1], mu, cnt, betad, siginv)
        assert r["status"] == int(g["bfgs_status" + sfx][d])
        assert r["nit"] == int(g["bfgs_nit" + sfx][d])
        assert np.max(np.abs(r["x"] - g["bfgs_x" + sfx][d])) <= 1e-7


def test_make_pd_and_decompose(oracle):
    g = load_golden("functions")
    for i, nm in enumerate(g["mat_names"]):
        M = g["mats"][i]
        assert np.array_equal(oracle.make_pd(M), g["make_pd"][i]), nm
        path, L, nu = oracle.decompose(M)
        assert path >= 0, nm
        assert _rel(L, g["chol"][i]) <= 1e-12, nm
        assert _rel(nu, g["nu"][i]) <= 1e-9, nm


def test_oracle_rejects_bad_beta(oracle):
    g = load_golden("toy_ctm")
    beta = g["beta0"].copy()
    beta[1, int(g["indices"][0])] = -1e-3
    with pytest.raises(AssertionError):  # stm.py:534
        oracle.estep(g["indptr"], g["indices"], g["counts"], beta, g["it0_mu_in"], g["it0_eta_in"],
                     g["it0_siginv"], float(g["it0_sigmaentropy"]))


def test_oracle_thread_count_does_not_change_results(oracle):
    g = load_golden("edge")
    args = (g["indptr"], g["indices"], g["counts"], g["beta0"], g["it0_mu_in"], g["it0_eta_in"],
            g["it0_siginv"], float(g["it0_sigmaentropy"]))
    a = oracle.estep(*args, nthreads=1)
    b = oracle.estep(*args, nthreads=0)
    assert np.array_equal(a["eta"], b["eta"]) and np.array_equal(a["bound_doc"], b["bound_doc"])
    assert _rel(a["beta_ss"], b["beta_ss"]) <= 1e-13


def test_heldout_restatement_matches_reference(oracle):
    """eval_heldout of the reference (src/modules/heldout.py:88-97) on the halved C1 corpus."""
    g = load_golden("heldout")
    per_doc = oracle.eval_heldout_docs(g["second_indptr"], g["second_indices"], g["second_counts"], g["theta"], g["beta"])
    assert np.allclose(per_doc, g["per_doc"], rtol=1e-13, atol=0)
    assert np.mean(per_doc) == pytest.approx(float(g["mean"]), rel=1e-14)


def test_oracle_on_the_full_size_reference_golden(oracle):
    """tests/golden/c2_full.npz holds the reference's own E-step over BASELINE configs[1] (100k documents,
    V=10k, K=50; tools/make_golden_c2.py).  The oracle is run on its 500 sampled documents of EM iteration 0."""
    from strutopy_amd.corpus import synthetic_corpus
    g = load_golden("c2_full")
    K, n = int(g["K"]), int(g["K"]) - 1
    c = synthetic_corpus(int(g["n_docs"]), int(g["V_requested"]), K, n_words=int(g["n_words"]), seed=int(g["seed"])).corpus
    assert int(np.sum(c.indices.astype(np.int64) * (np.arange(len(c.indices)) % 9973))) == int(g["checksum_indices"])
    rows = g["sample_docs"]
    lens = c.indptr[rows + 1] - c.indptr[rows]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    take = np.concatenate([np.arange(c.indptr[r], c.indptr[r + 1]) for r in rows])
    beta = reference_beta0(K, c.V)
    z = np.zeros((len(rows), n))
    o = oracle.estep(indptr, c.indices[take], c.counts[take], beta, z, z, g["it0_siginv"], float(g["it0_sigmaentropy"]), nthreads=0)
    assert np.array_equal(o["status"], g["it0_status"][rows]) and np.array_equal(o["nit"], g["it0_nit"][rows])
    assert np.array_equal(o["pd_path"], g["it0_pd_path"][rows])
    assert np.max(np.abs(o["eta"] - g["it0_eta_sample"])) <= 1e-7
    assert np.max(np.abs(o["theta"] - g["it0_theta_sample"])) <= 1e-7
    assert np.max(np.abs(o["bound_doc"] - g["it0_bound_doc_sample"]) / np.abs(g["it0_bound_doc_sample"])) <= 1e-9


def _long_run_bars(g):
    """The long-run regime's bar (tests/golden/c5_long.npz, tools/make_golden.py long_c5): the reference there takes ~12 BFGS
    iterations per document, and it differs from ITSELF in `ref_self_nit` of 300 documents when its start is moved by a relative
    1e-13.  A faithful restatement may differ from it in up to max(2 x that, 3) documents; on the documents that take the same
    path, eta within 10 x the reference's own eta sensitivity."""
    return (max(2 * int(g["ref_self_nit"]), 3), max(2 * int(g["ref_self_status"]), 3), 10.0 * float(g["ref_self_eta"]))


def test_long_run_regime_against_the_reference(oracle):
    """Config 5's shape at EM iteration 26 of a device fit (mean scipy nit 12): ONE teacher-forced E-step of the imported
    reference is the golden; the oracle must land within the reference's own sensitivity, and no further from it than when
    the golden was generated (`ref_vs_oracle_*`, recorded in the file)."""
    g = load_golden("c5_long")
    o = oracle.estep(g["indptr"], g["indices"], g["counts"], g["beta"], g["mu"], g["eta"], g["siginv"], float(g["sigmaentropy"]),
                     aspect=g["aspect"], nthreads=1)
    assert g["out_nit"].mean() >= 8.0
    nit_bar, status_bar, eta_bar = _long_run_bars(g)
    d_nit, d_status = int(np.sum(o["nit"] != g["out_nit"])), int(np.sum(o["status"] != g["out_status"]))
    same = (o["nit"] == g["out_nit"]) & (o["status"] == g["out_status"])
    d_eta = float(np.max(np.abs(o["eta"] - g["out_eta"])[same]))
    assert d_nit <= nit_bar and d_status <= status_bar and d_eta <= eta_bar, (d_nit, d_status, d_eta)
    assert d_nit <= int(g["ref_vs_oracle_nit"]) + 1 and d_status <= int(g["ref_vs_oracle_status"]) + 1   # (as recorded; + 1: another libm)
    assert np.array_equal(o["pd_path"], g["out_pd_path"])
    assert abs(o["bound"] - float(g["out_bound"])) <= 1e-8 * abs(float(g["out_bound"]))
    assert _rel(o["sigma_ss"], g["out_sigma_ss"]) <= 1e-6
