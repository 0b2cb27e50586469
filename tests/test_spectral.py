"""Spectral initialisation (reference src/modules/stm.py:30-296).

CPU: the NumPy restatement (oracle/spectral_oracle.py) and the host side of strutopy_amd.spectral (with the oracle
standing in for the device steps) against goldens made by running the reference's own gram / fastAnchor /
recover_l2 / spectral_init (tools/make_golden.py spectral_c1 spectral_wiki).  gram and fastAnchor are the
reference as imported; recover_l2's QP went through the stand-in solve_qp of tools/refshim (qpsolvers / quadprog are
absent from the image; the QP is strictly convex, its minimiser solver-independent) -- the fixtures say so (`qp_solver`).
GPU: the same checks through the C-ABI (stm_spectral_*)."""
import numpy as np
import pytest

from conftest import load_golden

CASES = ["spectral_c1", "spectral_wiki"]


def _corpus(g):
    from strutopy_amd.corpus import PackedCorpus
    c = load_golden(str(g["corpus"]))
    return PackedCorpus(c["indptr"], c["indices"], c["counts"], int(c["V"]))


def _check_parts(g, keep, wprob, anchor, q_rows=None, beta_kept=None, beta=None):
    assert np.array_equal(keep, g["keep"]) and np.allclose(wprob, g["wprob"], rtol=1e-15, atol=0)
    assert np.array_equal(np.asarray(anchor, dtype=np.int64), g["anchor"].astype(np.int64)), "anchor terms differ"
    if q_rows is not None:
        # the diagonal is a difference of two sums (Htilde^T Htilde - Hhat): absolute error relative to the matrix scale
        assert np.allclose(q_rows, g["Q_gram_rows"], rtol=1e-12, atol=1e-13 * np.abs(g["Q_gram_rows"]).max())
    if beta_kept is not None and "beta_kept" in g.files:
        assert np.allclose(beta_kept, g["beta_kept"], rtol=1e-7, atol=1e-11)
    if beta is not None:
        K = int(g["K"])
        assert np.allclose(beta.sum(axis=1), 1.0 / K, rtol=1e-12)          # the reference divides by the TOTAL sum (stm.py:83)
        if "beta" in g.files:
            assert np.allclose(beta, g["beta"], rtol=1e-7, atol=1e-12)
        else:
            assert np.allclose(beta.sum(axis=0), g["beta_colsum"], rtol=1e-7, atol=1e-12)
            assert np.allclose(beta[:, g["sample_cols"]], g["beta_cols"], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("name", CASES)
def test_oracle_restatement_matches_the_reference(name):
    from oracle import spectral_oracle as so
    g = load_golden(name)
    c = _corpus(g)
    beta, p = so.spectral_init(c.indptr, c.indices, c.counts, int(g["K"]), int(g["V"]))
    _check_parts(g, p["keep"], p["wprob"], p["anchor"], p["Q"][g["sample_rows"]], p["beta_kept"], beta)
    assert np.allclose(p["Q"].sum(axis=1), g["Q_gram_rowsum"], rtol=1e-11)
    assert np.allclose((p["Q"] ** 2).sum(axis=0), g["Q_gram_colsq"], rtol=1e-11)
    # gram's result is NOT row-normalised (sklearn normalises a discarded copy, stm.py:156)
    assert not np.allclose(np.sqrt((p["Q"] ** 2).sum(axis=1)), 1.0, atol=0.05)
    if "Q_caller_anchor_rows" in g.files:   # fastAnchor rescales the first anchor's row of its caller's matrix
        assert np.allclose(p["Q_caller"][np.intp(g["anchor"])], g["Q_caller_anchor_rows"], rtol=1e-12, atol=1e-15)


def test_host_side_with_the_oracle_engine_and_the_stm_surface():
    from _oracle_engine import OracleEngine
    from strutopy_amd.spectral import spectral_init
    from strutopy_amd.stm import STM
    g = load_golden("spectral_c1")
    c = _corpus(g)
    det = {}
    beta = spectral_init(c, int(g["K"]), int(g["V"]), verbose=False, engine=OracleEngine(), details=det)
    _check_parts(g, det["keep"], det["wprob"], det["anchor"], None, det["beta_kept"], beta)
    beta2 = spectral_init(c.to_bow(), int(g["K"]), int(g["V"]), verbose=False, engine=OracleEngine())   # the reference's BoW lists
    assert np.array_equal(beta, beta2)
    X = load_golden("c1_k10")["X"][:, 0]
    m = STM(documents=c, dictionary=None, content=False, K=int(g["K"]), X=X, kappa_interactions=False, max_em_iter=1,
            sigma_prior=0, convergence_threshold=1e-5, init_type="spectral", engine=OracleEngine())   # stm.py:419-422
    assert np.allclose(m.beta, g["beta"], rtol=1e-7, atol=1e-12)
    m.expectation_maximization(saving=False)
    assert np.isfinite(m.bound)


def test_qp_step_is_the_nonnegative_least_squares_fit():
    """recover_l2's QP (stm.py:271-285: min 1/2 x'Px + q'x, x <= 0, weights = -x) against its KKT conditions."""
    from strutopy_amd.spectral import solve_weights
    rng = np.random.default_rng(0)
    Vk, K = 120, 7
    Q = np.abs(rng.normal(size=(Vk, Vk))) * (rng.random((Vk, Vk)) < 0.3)
    anchor = rng.choice(Vk, K, replace=False)
    M = Q[anchor]
    q = Q @ M.T
    w = solve_weights(q, anchor)
    P = M @ M.T
    for i in range(Vk):
        if i in anchor:
            assert w[i].sum() == 1 and w[i, list(anchor).index(i)] == 1
            continue
        grad = P @ w[i] - q[i]                    # gradient of 1/2 w'Pw - q'w
        assert np.all(w[i] >= 0)
        assert np.all(grad >= -1e-9 * max(1.0, np.abs(q[i]).max()))                 # dual feasibility
        assert np.all(np.abs(grad[w[i] > 0]) <= 1e-9 * max(1.0, np.abs(q[i]).max()))  # complementarity


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_gram_anchors_and_beta_match_the_reference(name):
    from strutopy_amd.engine import HipEstepEngine
    from strutopy_amd.spectral import gram_inputs, kept_terms, spectral_init
    g = load_golden(name)
    c = _corpus(g)
    K, V = int(g["K"]), int(g["V"])
    e = HipEstepEngine(0)
    wprob, keep = kept_terms(c, 5000)
    e.spectral_gram(c.N, len(keep), gram_inputs(c, keep))
    rows = g["sample_rows"]
    q_rows = e.spectral_q_rows(rows)
    allq = np.concatenate([e.spectral_q_rows(np.arange(lo, min(lo + 500, len(keep)))) for lo in range(0, len(keep), 500)])
    assert np.allclose(allq.sum(axis=1), g["Q_gram_rowsum"], rtol=1e-11)
    assert np.allclose((allq ** 2).sum(axis=0), g["Q_gram_colsq"], rtol=1e-11)
    assert np.array_equal(allq, allq.T)                     # every (a, b) sum runs over the same documents in the same order
    anchor = e.spectral_anchors(K)
    _check_parts(g, keep, wprob, anchor, q_rows)
    if "Q_caller_anchor_rows" in g.files:
        assert np.allclose(e.spectral_q_rows(anchor), g["Q_caller_anchor_rows"], rtol=1e-12, atol=1e-15)
    # the per-term QPs on the device against SciPy's NNLS on the same inputs (and the KKT conditions of the QP)
    from strutopy_amd.spectral import solve_weights
    q = e.spectral_project(anchor)
    w_dev, w_ref = e.spectral_weights(anchor), solve_weights(q, anchor)
    assert w_dev.min() >= 0 and np.allclose(w_dev, w_ref, rtol=1e-8, atol=1e-10 * np.abs(w_ref).max())
    P = q[np.intp(anchor)]
    grad = w_dev @ P - q
    free = np.ones(len(q), dtype=bool); free[np.intp(anchor)] = False
    scale = np.abs(q).max()
    assert grad[free].min() >= -1e-9 * scale and np.abs(grad[free][w_dev[free] > 0]).max() <= 1e-9 * scale
    e.spectral_release()
    det = {}
    beta = spectral_init(c, K, V, verbose=False, engine=e, details=det)
    _check_parts(g, det["keep"], det["wprob"], det["anchor"], None, det["beta_kept"], beta)
    e.close()


@pytest.mark.gpu
def test_stm_with_spectral_init_on_the_gpu():
    """src/05_train.py's configuration in miniature: init_type="spectral", then EM on the device."""
    from strutopy_amd import STM
    g = load_golden("spectral_c1")
    c = _corpus(g)
    X = load_golden("c1_k10")["X"][:, 0]
    m = STM(documents=c, dictionary=None, content=False, K=int(g["K"]), X=X, kappa_interactions=False, max_em_iter=3,
            sigma_prior=0, convergence_threshold=1e-5, init_type="spectral")
    assert np.allclose(m.beta, g["beta"], rtol=1e-7, atol=1e-12)
    m.expectation_maximization(saving=False)
    assert len(m.last_bounds) >= 2 and np.all(np.isfinite(m.last_bounds)) and m.last_bounds[-1] > m.last_bounds[0]
    m.close()
