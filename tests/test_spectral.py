"""Spectral initialisation (reference src/modules/stm.py:30-296).

CPU: the NumPy restatement (oracle/spectral_oracle.py) and the host side of strutopy_amd.spectral (with the oracle
standing in for the device steps) against goldens made by running the reference's own gram / fastAnchor /
recover_l2 / spectral_init (tools/make_golden.py spectral_c1 spectral_wiki).  gram and fastAnchor are the
reference as imported; recover_l2's solve_qp call reaches quadprog's algorithm -- the dual active-set method of Goldfarb &
Idnani (1983) -- as restated in oracle/spectral_oracle.py (qpsolvers / quadprog themselves are absent from the image; the
fixtures say so in `qp_solver`).
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


def _kkt(P, q, w, tol=1e-9):
    """KKT conditions of min 1/2 x'Px + q'x, x <= 0 at x = -w (stm.py:271-285)."""
    grad = P @ w - q                                   # gradient of 1/2 w'Pw - q'w
    scale = max(1.0, float(np.abs(q).max()))
    free = w > 1e-12 * max(1.0, float(np.abs(w).max()))   # a weight the active-set method left at rounding level is a bound one
    return bool(np.all(w >= -1e-15) and np.all(grad >= -tol * scale) and np.all(np.abs(grad[free]) <= tol * scale))


def test_goldfarb_idnani_restatement_on_known_answers():
    """The restated quadprog algorithm on QPs with answers known in closed form, incl. the worked example of the paper
    (Goldfarb & Idnani 1983, section 5; also the example of R's solve.QP documentation)."""
    from oracle.spectral_oracle import solve_qp_goldfarb_idnani as gi
    # solve.QP's example: min -d'b + 1/2 b'Db, D = I, d = (0, 5, 0), A'b >= b0
    Amat = np.array([[-4.0, -3.0, 0.0], [2.0, 1.0, 0.0], [0.0, -2.0, 1.0]]).T        # columns of R's Amat
    b0 = np.array([-8.0, 2.0, 0.0])
    x = gi(np.eye(3), -np.array([0.0, 5.0, 0.0]), -Amat.T, -b0)          # Amat' b >= b0  <=>  (-Amat') b <= -b0
    assert np.allclose(x, [0.4761905, 1.0476190, 2.0952381], atol=1e-7)              # the documented solution
    # unconstrained minimum inside the feasible set / on one face / at a vertex
    P = np.array([[2.0, 0.5], [0.5, 1.0]])
    assert np.allclose(gi(P, np.array([1.0, 1.0]), np.eye(2), np.zeros(2)), -np.linalg.solve(P, [1.0, 1.0]))
    assert np.allclose(gi(P, np.array([-1.0, 1.0]), np.eye(2), np.zeros(2)), [0.0, -1.0])
    assert np.allclose(gi(P, np.array([-1.0, -2.0]), np.eye(2), np.zeros(2)), [0.0, 0.0])


@pytest.mark.parametrize("name", CASES)
def test_qp_step_goldfarb_idnani_equals_the_least_squares_form(name):
    """recover_l2's QP on the fixtures' own inputs: quadprog's algorithm (restated) and the non-negative least-squares form
    the device solves give the same weights to 1e-10, and both satisfy the KKT conditions."""
    from oracle import spectral_oracle as so
    g = load_golden(name)
    c = _corpus(g)
    wprob, keep = so.word_probabilities(c.indptr, c.indices, c.counts)
    Q = so.fast_anchor(so.gram(c.indptr, c.indices, c.counts, keep), int(g["K"]))[1]
    anchor = np.intp(g["anchor"])
    M = Q[anchor]
    P = M @ M.T
    rows = [i for i in np.linspace(0, len(keep) - 1, 160 if name == "spectral_c1" else 60).astype(np.int64) if i not in set(anchor)]
    w_gi = so.recover_l2_weights(Q, anchor, rows=rows)
    for i in rows:
        q = M @ Q[i]
        w_ls = so.nnls_weights(P, q)
        assert np.max(np.abs(w_gi[i] - w_ls)) <= 1e-10 * max(1.0, float(np.abs(w_ls).max())), (name, int(i))
        assert _kkt(P, q, w_gi[i]) and _kkt(P, q, w_ls)


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
    # the per-term QPs on the device against quadprog's algorithm (restated, a sample of terms) on the same inputs, and
    # against the KKT conditions of the QP (every term)
    from oracle import spectral_oracle as so
    q = e.spectral_project(anchor)
    w_dev = e.spectral_weights(anchor)
    P = q[np.intp(anchor)]
    sample = [i for i in np.linspace(0, len(q) - 1, 80).astype(np.int64) if i not in set(np.intp(anchor))]
    for i in sample:
        w_gi = -so.solve_qp_goldfarb_idnani(P, q[i], np.eye(K), np.zeros(K))
        assert np.max(np.abs(w_dev[i] - w_gi)) <= 1e-8 * max(1.0, float(np.abs(w_gi).max())), int(i)
    assert w_dev.min() >= 0
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
@pytest.mark.parametrize("name", CASES)
def test_resident_gram_whole_and_sharded_against_the_prepared_orientations(name):
    """stm_spectral_gram_resident (gram on the handle's resident CSR: no NumPy preparation) gives the matrix of stm_spectral_gram
    fed with the prepared orientations -- the same products added in the same order, bit for bit -- and, run on two document
    shards (flags = 1) whose matrices are then summed and pushed back, the reference's anchors again (gram is a sum over
    documents, stm.py:122-157: how a sharded fit initialises)."""
    from strutopy_amd.engine import HipEstepEngine
    from strutopy_amd.spectral import gram_inputs, kept_terms
    g = load_golden(name)
    c = _corpus(g)
    K, V = int(g["K"]), int(g["V"])
    wprob, keep = kept_terms(c, 5000)
    allrows = np.arange(len(keep), dtype=np.int32)
    e = HipEstepEngine(0)
    e.spectral_gram(c.N, len(keep), gram_inputs(c, keep))
    q_prepared = e.spectral_q_rows(allrows)
    e.set_corpus(c.indptr, c.indices, c.counts, V)
    e.spectral_gram_resident(keep)
    q_resident = e.spectral_q_rows(allrows)
    assert np.array_equal(q_resident, q_prepared)
    anchor_whole = e.spectral_anchors(K)
    e.spectral_release()
    # two shards, each on a handle of its own
    mid = c.N // 2
    parts = []
    for lo, hi in ((0, mid), (mid, c.N)):
        sh = c.slice(lo, hi)
        es = HipEstepEngine(0)
        es.set_corpus(sh.indptr, sh.indices, sh.counts, V)
        es.spectral_gram_resident(keep, check=False)
        parts.append(es.spectral_q_rows(allrows))
        es.spectral_release(); es.close()
    e.spectral_gram_resident(keep)                      # (allocates the state; its matrix is replaced by the shards' sum)
    e.spectral_put_q(parts[0] + parts[1])
    e.spectral_check()
    assert np.allclose(parts[0] + parts[1], q_prepared, rtol=1e-12, atol=1e-13 * np.abs(q_prepared).max())
    anchor = e.spectral_anchors(K)
    assert np.array_equal(anchor, anchor_whole) and np.array_equal(anchor.astype(np.int64), g["anchor"].astype(np.int64))
    e.spectral_release()
    e.close()


@pytest.mark.gpu
def test_device_qp_with_a_dependent_anchor_row_is_still_a_minimiser():
    """Two identical anchor rows make P = M M^T singular: the passive-set factorisation of the second one fails, the column
    is banned (not re-factorised until the iteration cap) and the result must still satisfy the QP's KKT conditions -- a
    minimiser, if not a unique one; terms whose result does not are reported as STM_ERR_LINALG (what quadprog does with a P
    that is not positive definite), never returned silently."""
    from strutopy_amd.engine import HipEstepEngine
    from strutopy_amd.spectral import gram_inputs, kept_terms
    g = load_golden("spectral_c1")
    c = _corpus(g)
    e = HipEstepEngine(0)
    wprob, keep = kept_terms(c, 5000)
    e.spectral_gram(c.N, len(keep), gram_inputs(c, keep))
    anchor = np.array(e.spectral_anchors(int(g["K"])), dtype=np.float64)
    anchor[-1] = anchor[0]                                   # a dependent row
    q = e.spectral_project(anchor)
    w = e.spectral_weights(anchor)
    P = q[np.intp(anchor)]
    free = np.ones(len(q), dtype=bool); free[np.intp(anchor)] = False
    grad = w @ P - q
    scale = np.abs(q).max()
    assert w.min() >= 0 and np.isfinite(w).all()
    assert grad[free].min() >= -1e-7 * scale and np.abs(grad[free][w[free] > 0]).max() <= 1e-7 * scale
    e.spectral_release()
    e.close()


@pytest.mark.gpu
def test_device_qp_beyond_128_topics_is_a_minimiser():
    """K = 140 anchors (nnls_kernel<512>: per-thread vectors of up to K_LIMIT entries): every term's weights satisfy the KKT
    conditions of recover_l2's QP (stm.py:257-285), and a fit with init_type="spectral" runs at that K."""
    from strutopy_amd import STM
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    from strutopy_amd.spectral import gram_inputs, kept_terms
    K = 140
    syn = synthetic_corpus(3000, 1200, K, n_words=80, seed=140)
    c = syn.corpus
    e = HipEstepEngine(0)
    wprob, keep = kept_terms(c, 5000)
    e.spectral_gram(c.N, len(keep), gram_inputs(c, keep))
    anchor = np.array(e.spectral_anchors(K), dtype=np.float64)
    assert len(np.unique(anchor)) == K
    q = e.spectral_project(anchor)
    w = e.spectral_weights(anchor)
    P = q[np.intp(anchor)]
    free = np.ones(len(q), dtype=bool); free[np.intp(anchor)] = False
    grad = w @ P - q
    scale = np.abs(q).max()
    assert w.min() >= 0 and np.isfinite(w).all()
    assert grad[free].min() >= -1e-7 * scale and np.abs(grad[free][w[free] > 0]).max() <= 1e-7 * scale
    e.spectral_release()
    e.close()
    m = STM(documents=c, dictionary=None, content=False, K=K, X=syn.X, kappa_interactions=False, max_em_iter=2,
            sigma_prior=0, convergence_threshold=1e-5, init_type="spectral")
    assert np.isfinite(m.beta).all() and m.beta.min() >= 0 and m.beta.shape == (K, c.V)
    m.expectation_maximization(saving=False)
    assert np.all(np.isfinite(m.last_bounds))
    m.close()


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
