"""Shared body of the M-step-branch parity tests (tests/golden/mstep_modes.npz, tools/make_golden.py mstep_modes): the
reference's two EM iterations with mode="ridge", mode="lasso" (stm.py:678-689) and sigma_prior = 0.5 (stm.py:721-728) on a
three-level covariate."""
import numpy as np
import pytest

CONFIGS = [("ridge", "ridge", 0.0), ("lasso", "lasso", 0.0), ("sp05", "ols", 0.5)]


def run(g, tag, mode, sp, resident, engine=None):
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.stm import STM
    kw = {} if engine is None else dict(engine=engine)
    m = STM(documents=PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"])), dictionary=None, content=False,
            K=int(g["K"]), X=g["X"], kappa_interactions=False, max_em_iter=2, sigma_prior=sp, convergence_threshold=1e-12,
            init_type="random", model_type="STM", mode=mode, **kw)
    assert np.array_equal(m.beta, g["beta0"])
    assert m._Xenc.shape[1] == 3                       # the three levels, one-hot (stm.py:665-667)
    m.expectation_maximization(saving=False, resident=resident)
    p = tag + "_it"
    assert m.last_bounds[0] == pytest.approx(float(g[p + "0_bound"]), rel=1e-10)
    assert m.last_bounds[1] == pytest.approx(float(g[p + "1_bound"]), rel=1e-8)
    assert np.allclose(m.gamma, g[p + "1_gamma"], rtol=1e-5, atol=1e-7)
    assert np.allclose(m.mu, g[p + "1_mu_out"], rtol=0, atol=1e-6)
    assert np.allclose(m.sigma, g[p + "1_sigma_out"], rtol=1e-6, atol=1e-8)
    assert np.allclose(m.beta, g[p + "1_beta_out"], rtol=1e-6, atol=1e-12)
    if sp:                                             # the prior pulls the off-diagonals towards 0 (stm.py:728)
        full = (g[p + "1_sigma_out"] - np.diag(np.diag(g[p + "1_sigma_out"]))) / (1 - sp)
        assert np.abs(full).max() > 0 and np.allclose(np.diag(m.sigma), np.diag(g[p + "1_sigma_out"]), rtol=1e-6)
    if mode == "lasso":
        assert not m.gamma.any() or np.count_nonzero(m.gamma) < m.gamma.size   # alpha = 1 shrinks coefficients to 0
    if hasattr(m, "close"):
        m.close()
    return m
