"""`STM` -- drop-in for the hot path of strutopy's STM class on MI355X.

Mirrors the method surface of the reference class (reference
src/modules/stm.py:310-1149): same constructor keywords (stm.py:311-329), same
attributes (`beta, mu, eta, sigma, theta, gamma, bound, last_bounds, siginv,
sigmaentropy, phi, wcounts, N, K, V, A`), `E_step()` -> `(beta_ss, sigma_ss)`
(stm.py:489-597), `M_step(beta_ss, sigma_ss)` (stm.py:622-747),
`expectation_maximization(saving, output_dir)` (stm.py:855-880; also reachable
as `fit()`), `EM_is_converged`, `max_its_reached`, `save_model` (stm.py:1120).

The E-step runs in hand-written HIP kernels behind the C-ABI of
include/stm_estep.h; host code is Python + NumPy + ctypes.  There is no CPU
fallback: constructing an STM without a usable GPU / built library raises.

`init_type="spectral"` (stm.py:30-296) runs through strutopy_amd.spectral (gram / fastAnchor on the GPU).
Out of scope (raise NotImplementedError): `mnreg` (`lda_beta=False`,
stm.py:749-853, broken in the reference on current scipy), labelling/plot helpers.
"""
import logging
import os
import pickle
import time

import numpy as np

from .corpus import PackedCorpus, pack_bow
from .dist import SingleComm

logger = logging.getLogger(__name__)
_GATHER_BLOCK = 256 << 20     # bytes per rank and message when save_model gathers the shards' rows


def _default_engine(device):
    from .engine import HipEstepEngine  # raises if libstm_hip.so / a GPU is missing
    return HipEstepEngine(device)


def lasso_from_moments(Sxx, Sxe, syy, n_samples, alpha=1.0, tol=1e-4, max_iter=1000):
    """sklearn.linear_model.Lasso(alpha, fit_intercept=True).fit(X, eta).coef_ (stm.py:677-681) from the centred moments
    Sxx = Xc^T Xc (p x p), Sxe = Xc^T eta_c (p x n) and syy = diag(eta_c^T eta_c) (n): cyclic coordinate descent on the Gram
    matrix -- sklearn's own `enet_coordinate_descent_gram` (its `precompute` form of the solver the reference runs): the
    objective 1/(2 N) |y - X w|^2 + alpha |w|_1 scaled by N, coefficients from zero, one target at a time in sklearn, all
    targets side by side here (each with its own stopping test: largest coordinate update below `tol`, then the duality gap
    below tol * |y|^2).  X and eta enter through these moments only, so document shards need nothing beyond what the
    iteration's all-reduce already carries.  Returns coef (n x p)."""
    Q = np.ascontiguousarray(Sxx, dtype=np.float64)
    q = np.ascontiguousarray(Sxe, dtype=np.float64)
    p, n = q.shape
    a = float(alpha) * float(n_samples)
    W = np.zeros((p, n))
    H = np.zeros((p, n))                     # Q @ W, kept up to date
    y2 = np.asarray(syy, dtype=np.float64)
    gap_tol = tol * y2
    active = np.ones(n, dtype=bool)
    diag = np.diag(Q)
    for it in range(max_iter):
        idx = np.flatnonzero(active)
        if idx.size == 0:
            break
        w_max = np.zeros(idx.size)
        d_w_max = np.zeros(idx.size)
        Wa, Ha, qa = W[:, idx], H[:, idx], q[:, idx]
        for ii in range(p):
            if diag[ii] == 0.0:
                continue
            w_ii = Wa[ii].copy()
            Ha -= Q[:, ii][:, None] * w_ii[None, :]
            tmp = qa[ii] - Ha[ii]
            w_new = np.sign(tmp) * np.maximum(np.abs(tmp) - a, 0.0) / diag[ii]
            Wa[ii] = w_new
            Ha += Q[:, ii][:, None] * w_new[None, :]
            d_w_max = np.maximum(d_w_max, np.abs(w_new - w_ii))
            w_max = np.maximum(w_max, np.abs(w_new))
        W[:, idx], H[:, idx] = Wa, Ha
        with np.errstate(divide="ignore", invalid="ignore"):
            check = (w_max == 0.0) | (d_w_max / w_max < tol) | (it == max_iter - 1)
        if not check.any():
            continue
        c = idx[check]
        Wc, Hc, qc = W[:, c], H[:, c], q[:, c]
        q_dot_w = np.sum(Wc * qc, axis=0)
        dual = np.max(np.abs(qc - Hc), axis=0)
        r2 = y2[c] + np.sum(Wc * Hc, axis=0) - 2.0 * q_dot_w
        big = dual > a
        const = np.where(big, a / np.where(big, dual, 1.0), 1.0)
        gap = np.where(big, 0.5 * (r2 + r2 * const ** 2), r2)
        gap = gap + a * np.sum(np.abs(Wc), axis=0) - const * y2[c] + const * q_dot_w
        active[c[gap < gap_tol[c]]] = False
    if active.any():
        import warnings
        warnings.warn(f"lasso_from_moments: {int(active.sum())} target(s) did not reach the duality-gap tolerance in {max_iter} sweeps")
    return W.T.copy()


def encode_covariates(X, comm=None):
    """The covariate preparation of update_mu (stm.py:656-671): 2-D, one-hot unless already 0/1.

    With a multi-rank `comm`, X is this rank's document shard: whether the columns count as "already 0/1"
    and the category list of every column are agreed on over ALL shards first, so every rank builds the
    same columns in the same order (what the reference's OneHotEncoder sees is the whole corpus)."""
    if X is None:
        return None
    try:
        import pandas as pd  # stm.py:657 tries `.astype("category")` (a pandas-only no-op here)
        if isinstance(X, (pd.DataFrame, pd.Series)):
            X = X.to_numpy()
    except Exception:
        pass
    prev_cov = np.array(X)[:, None]
    if prev_cov.ndim > 2:
        prev_cov = np.squeeze(prev_cov, axis=1)
    is_bool = bool(np.array_equal(prev_cov, prev_cov.astype(bool)))
    cats = [np.unique(prev_cov[:, j]) for j in range(prev_cov.shape[1])]
    if comm is not None and comm.size > 1:
        infos = comm.allgather((prev_cov.shape[1], is_bool, cats))
        if len({i[0] for i in infos}) != 1:
            raise ValueError("X has a different number of columns on different ranks")
        is_bool = all(i[1] for i in infos)
        cats = [np.unique(np.concatenate([i[2][j] for i in infos])) for j in range(prev_cov.shape[1])]
    if not is_bool:
        # sklearn OneHotEncoder semantics: per column, sorted categories -> indicator columns
        prev_cov = np.concatenate([(prev_cov[:, j][:, None] == cats[j][None, :]).astype(np.float64)
                                   for j in range(prev_cov.shape[1])], axis=1)
    return np.ascontiguousarray(prev_cov, dtype=np.float64)


class STM:
    def __init__(self, documents, dictionary, content, K, X, kappa_interactions, max_em_iter,
                 sigma_prior, convergence_threshold, lda_beta=True, beta_index=None, A=None,
                 dtype=np.float32, init_type="spectral", model_type="STM", mode="ols",
                 device=0, comm=None, engine=None, n_total=None, exchange="split"):
        """Same arguments as the reference constructor (stm.py:311-329) plus

        device  : GPU ordinal of this process
        comm    : strutopy_amd.dist communicator when `documents` is one shard of a
                  document-sharded corpus (None: single GPU)
        engine  : test hook -- an object with the HipEstepEngine interface
        exchange: "split" (default) | "single" -- how a sharded fit's EM iteration sends its sufficient statistics over RCCL: two
                  all-reduces ([bound | sigma_ss | moments] in front of the host's read-back, beta_ss behind it) or ONE of the whole
                  packed buffer (what BASELINE.json's north_star names); same sums (`STM.exchange` can be changed between iterations,
                  on every rank alike)
        `documents` may be the reference's BoW list or a strutopy_amd.corpus.PackedCorpus.
        """
        np.random.seed(123456)  # stm.py:361 reseeds numpy's legacy global RNG; kept for drop-in parity
        self.dtype = np.finfo(dtype).dtype

        if not hasattr(documents, "__len__") or isinstance(documents, (str, tuple)) or hasattr(documents, "tocsr"):
            # a MatrixMarket path, a scipy.sparse matrix or the CSR triple itself (corpus.pack_bow): packed here, once
            documents = pack_bow(documents, V=len(dictionary) if dictionary is not None else None)
        self.documents = documents
        self.dictionary = dictionary
        self.init = init_type
        self.model = model_type
        self.mode = mode
        self.content = content
        self.K = K
        self.A = A
        self.V = len(self.dictionary) if dictionary is not None else int(documents.V)
        self.X = X
        self.interactions = kappa_interactions
        self.beta_index = beta_index
        self.max_em_iter = max_em_iter
        self.sigma_prior = sigma_prior
        self.convergence_threshold = convergence_threshold
        self.N = len(self.documents)
        self.LDAbeta = lda_beta
        self.betaindex = beta_index
        self.last_bounds = []
        self.max_em_its = max_em_iter
        if self.K == 0:
            raise ValueError("Number of topics must be specified")
        if self.A == 1:
            logging.warning("no dimension for the topical content provided")

        self.comm = comm if comm is not None else SingleComm()
        self.N_total = int(n_total) if n_total is not None else int(
            self.comm.allreduce_host(np.array([float(self.N)]))[0])

        # ---- device state
        self._corpus = pack_bow(documents, V=self.V)
        self._levels = int(self.A) if (self.interactions and self.A) else 1
        self._aspect = None
        if self._levels > 1:
            if beta_index is None:
                raise ValueError("kappa_interactions=True needs beta_index")
            self._aspect = np.ascontiguousarray(beta_index, dtype=np.int32)
        self._engine = engine if engine is not None else _default_engine(device)
        self._engine.set_corpus(self._corpus.indptr, self._corpus.indices, self._corpus.counts, self.V,
                                aspect=self._aspect, A=self._levels)
        self._engine.set_topics(self.K)
        self.comm.attach(self._engine)
        self._Xenc = encode_covariates(X, self.comm) if model_type == "STM" else None
        if self._Xenc is not None and len(self._Xenc) != self.N:
            raise ValueError("X must have one row per document")
        self._cov_on_device = False
        self._phi = None
        self._phi_stale = False
        if exchange not in ("split", "single"):
            raise ValueError("exchange must be 'split' or 'single'")
        self.exchange = exchange
        self._exchange_set = None
        self.cov_exchange = "moments"    # "exact": always take the second (K-1)^2 all-reduce of the local covariance
        self.cov_exchanges = []          # per resident iteration: which form was used
        # which side holds the fresh copy of each array ("host" | "device" | "both")
        self._fresh = dict(beta="host", eta="host", mu="host", theta="host")
        self.init_params()
        self.timings = []

    # ------------------------------------------------------------------ state plumbing
    def _host_value(self, name):
        if self._fresh[name] == "device":
            getter = getattr(self._engine, "get_" + name)
            setattr(self, "_" + name, getter())
            self._fresh[name] = "both"
        return getattr(self, "_" + name)

    def _set_host(self, name, value):
        setattr(self, "_" + name, value)
        self._fresh[name] = "host"

    def _push(self, name):
        if self._fresh[name] == "host":
            getattr(self._engine, "put_" + name)(getattr(self, "_" + name))
            self._fresh[name] = "both"

    beta = property(lambda s: s._host_value("beta"), lambda s, v: s._set_host("beta", v))
    eta = property(lambda s: s._host_value("eta"), lambda s, v: s._set_host("eta", v))
    mu = property(lambda s: s._host_value("mu"), lambda s, v: s._set_host("mu", v))

    @property
    def phi(self):
        """stm.py:1116 leaves the last document's phi in self.phi; fetched from the device on first use."""
        if self._phi_stale:
            self._phi = self._engine.get_phi_last()     # (a failing read-back raises: the reference leaves a matrix here, never None)
            self._phi_stale = False
        return self._phi

    @phi.setter
    def phi(self, v):
        self._phi, self._phi_stale = v, False

    @property
    def theta(self):
        return self._host_value("theta")

    @theta.setter
    def theta(self, v):
        self._theta = v
        self._fresh["theta"] = "both"  # theta is an output only

    # ------------------------------------------------------------------ initialisation (stm.py:402-486)
    def init_params(self):
        self.init_beta()
        self.init_mu()
        self.init_eta()
        self.init_sigma()
        self.wcounts = self._corpus.word_counts()
        self.init_theta()

    def init_beta(self):
        if self.init == "spectral":
            from .spectral import spectral_init
            # stm.py:419-422; a 2-D beta also when kappa_interactions is set, like the reference.  On a sharded fit every rank
            # runs gram on its resident shard, the matrices are summed once, and all ranks end with the same beta.
            self.beta = spectral_init(self._corpus, self.K, self.V, maxV=5000, verbose=False, engine=self._engine,
                                      comm=self.comm, resident=True)
        elif self.init == "random":
            # stm.py:425-439, numpy legacy RNG stream seeded at stm.py:361
            beta_init = np.random.gamma(0.1, 1, self.V * self.K).reshape(self.K, self.V)
            row_sums = np.sum(beta_init, axis=1)[:, None]
            beta_init_normalized = np.divide(beta_init, row_sums, out=np.zeros_like(beta_init),
                                             where=row_sums != 0)
            if self.interactions:
                self.beta = np.repeat(beta_init_normalized[None, :], self.A, axis=0)
            else:
                self.beta = beta_init_normalized
        else:
            raise ValueError("init_type must be 'random' or 'spectral'")

    def init_mu(self):
        self.mu = np.zeros((self.N, self.K - 1))

    def init_sigma(self):
        self.sigma = np.zeros(((self.K - 1), (self.K - 1)))
        np.fill_diagonal(self.sigma, 20)

    def init_eta(self):
        self.eta = np.zeros((self.N, self.K - 1))

    def init_theta(self):
        self.theta = np.zeros((self.N, self.K))

    # ------------------------------------------------------------------ E-step (stm.py:489-597)
    def _preamble(self):
        """stm.py:497-510 with the reference's own numpy expression (element-wise `*`)."""
        sigobj = np.linalg.cholesky(self.sigma)  # LinAlgError if Sigma is not PD
        self.sigmaentropy = np.sum(np.log(np.diag(sigobj)))
        inv = np.linalg.inv(sigobj)                 # the reference inverts twice; the two results are the same array
        self.siginv = inv.T * inv

    def _estep_device(self):
        """Run the kernels on the resident state; results stay in HBM."""
        self._preamble()
        for name in ("beta", "eta", "mu"):
            self._push(name)
        t0 = time.time()
        bound_local = self._engine.estep(self.siginv, float(self.sigmaentropy))
        self._fresh["eta"] = "device"
        self._fresh["theta"] = "device"
        self._phi_stale = True
        self._estep_seconds = time.time() - t0
        return bound_local

    def E_step(self):
        """Drop-in for STM.E_step: returns (beta_ss, sigma_ss) as numpy arrays."""
        start_time = time.time()
        bound_local = self._estep_device()
        bound, _ = self.comm.allreduce_suffstats(self._engine, np.zeros(0))
        if self.comm.size == 1:
            bound = bound_local
        beta_ss = self._engine.get_beta_ss()
        sigma_ss = self._engine.get_sigma_ss()
        self.bound = bound
        self.last_bounds.append(self.bound)
        logger.info(f"Lower Bound: {self.bound}")
        logger.info(f"Completed E-Step in {np.round(time.time() - start_time, 3)} seconds. \n")
        return beta_ss, sigma_ss

    def get_beta(self, words, aspect):
        """stm.py:599-620 (host view; the kernels gather on the device)."""
        if self.interactions:
            return self.beta[aspect][:, np.array(np.intp(words))]
        return self.beta[:, np.array(np.intp(words))]

    def solver_diagnostics(self):
        """Per-document scipy-style status / nit / nfev / njev and the PD-fix path of the last E-step."""
        return self._engine.get_diagnostics()

    # ------------------------------------------------------------------ M-step (stm.py:622-747)
    def M_step(self, beta_ss, sigma_ss):
        """Host M-step on numpy arrays, mirroring the reference statement by statement."""
        start_time = time.time()
        self.update_mu()
        self.update_sigma(nu=sigma_ss, sigprior=self.sigma_prior)
        self.update_beta(beta_ss)
        logger.info(f"Completed M-Step in {np.round(time.time() - start_time, 3)} seconds. \n")

    def _regress(self, prev_cov, eta, intercept=True):
        # sklearn flattens coef_ for a single target (K = 2), which sends the reference's mu = X @ coef_.T (stm.py:704) to
        # shape (N,) and its Sigma to N x N; here the coefficients keep their (K-1) x p shape
        shape = (eta.shape[1], prev_cov.shape[1])
        if self.mode == "lasso":
            import sklearn.linear_model
            return np.reshape(sklearn.linear_model.Lasso(alpha=1, fit_intercept=intercept).fit(prev_cov, eta).coef_, shape)
        if self.mode == "ridge":
            import sklearn.linear_model
            return np.reshape(sklearn.linear_model.Ridge(alpha=0.1, fit_intercept=intercept).fit(prev_cov, eta).coef_, shape)
        if self.mode != "ols":
            print("Need to specify the estimation mode of prevalence covariate coefficients. Uses default 'ols'.")
        # sklearn LinearRegression(fit_intercept=True): centre, then minimum-norm least squares
        Xc = prev_cov - prev_cov.mean(axis=0)
        yc = eta - eta.mean(axis=0)
        coef, *_ = np.linalg.lstsq(Xc, yc, rcond=max(Xc.shape) * np.finfo(np.float64).eps)
        return coef.T

    def update_mu(self, intercept=True):
        if self.comm.size > 1:
            raise RuntimeError("the host M-step is single-process; sharded fits use expectation_maximization()")
        if self.model == "CTM":
            self.mu = np.repeat(np.mean(self.eta, axis=0)[None, :], self.N, axis=0)  # stm.py:651
        elif self.model == "STM":
            prev_cov = self._Xenc
            self.gamma = self._regress(prev_cov, self.eta, intercept)  # stm.py:703: coef_ only
            self.mu = prev_cov @ self.gamma.T                          # stm.py:704-706: no intercept
        else:
            raise ValueError('Updating the topical prevalence parameter requires a mode. Choose from "CTM", '
                             '"Pooled" or "L1" (default).')

    def update_sigma(self, nu, sigprior=0):
        assert 0 <= sigprior <= 1, "weight needs to be defined between 0 and 1"
        diff = self.eta - self.mu
        covariance = np.array(diff.T @ diff, dtype="float64")
        self._finish_sigma(covariance, nu, sigprior)

    def _finish_sigma(self, covariance, nu, sigprior):
        sigma = np.array((covariance + nu) / self.N_total, dtype="float64")  # stm.py:725
        self.sigma = np.diag(np.diag(sigma)) * sigprior + (1 - sigprior) * sigma  # stm.py:728

    def update_beta(self, beta_ss):
        if self.LDAbeta:
            assert np.any(np.sum(beta_ss, axis=1) >= 0), "break here"
            row_sums = np.sum(beta_ss, axis=1)[:, None]  # 3-D beta_ss: this sums over topics (reference quirk)
            self.beta = np.divide(beta_ss, row_sums, out=np.zeros_like(beta_ss), where=row_sums != 0)
        else:
            raise NotImplementedError("lda_beta=False (mnreg, reference stm.py:749-853) is out of scope")

    # ------------------------------------------------------------------ device-resident EM iteration
    def _covariance_from_moments(self, ete, n_tot, se, XtX=None, Xte=None):
        """(eta - mu)^T (eta - mu) (stm.py:723) from the all-reduced moments, with mu_d = x_d gamma^T (stm.py:706)
        or the column mean (CTM, stm.py:651).  Returns (covariance, smallest diag(cov) / diag(eta^T eta))."""
        if XtX is None:
            m = se / n_tot
            cov = ete - n_tot * np.outer(m, m)
        else:
            C = self.gamma
            M = C @ Xte
            Q = C @ XtX @ C.T
            cov = ete - M - M.T + 0.5 * (Q + Q.T)
        d = np.diag(ete)
        ratio = float(np.min(np.diag(cov)[d > 0] / d[d > 0])) if np.any(d > 0) else 1.0
        return cov, ratio

    def _em_iteration_resident(self):
        """E-step + ONE all-reduce + M-step with eta / mu / beta / theta kept in HBM.

        Same arithmetic as E_step() + M_step(); the regression and the covariance are formed from
        moments so that document shards only exchange O(K^2 + K V) numbers, once.
        """
        eng = self._engine
        n = self.K - 1
        t0 = time.time()
        if self.model not in ("STM", "CTM"):
            raise ValueError("model_type must be 'STM' or 'CTM'")
        use_reg = self.model == "STM"
        if use_reg and self.mode not in ("ols", "ridge", "lasso") and not getattr(self, "_mode_notice", False):
            # the reference's own fallback (stm.py:696-700)
            print("Need to specify the estimation mode of prevalence covariate coefficients. Uses default 'ols'.")
            self._mode_notice = True
        if use_reg and not self._cov_on_device:
            eng.put_covariates(self._Xenc)      # before the E-step: a wide X re-allocates the packed buffer
            self._cov_on_device = True
        p = self._Xenc.shape[1] if use_reg else 0
        # engines with device-side collectives run the iteration with ONE host wait (stm_em_begin / stm_em_finish);
        # the host reduction (HostComm) and the test double take the call-by-call path
        fused = getattr(self.comm, "device_collective", False) and hasattr(eng, "em_begin")
        if fused:
            self._preamble()
            for name in ("beta", "eta", "mu"):
                self._push(name)
            if self._exchange_set != self.exchange and hasattr(eng, "set_exchange") and (self.exchange != "split" or self._exchange_set is not None):
                eng.set_exchange(self.exchange)       # ("split" is a fresh handle's state)
            self._exchange_set = self.exchange
            bound, sigma_ss, mom = eng.em_begin(self.siginv, float(self.sigmaentropy), p)
            self._fresh["eta"] = self._fresh["theta"] = "device"
            self._phi_stale = True
            t1 = time.time()
        else:
            bound_local = self._estep_device()
            t1 = time.time()
            mom = eng.moments(p)
            bound, mom = self.comm.allreduce_suffstats(eng, mom)
            if self.comm.size == 1:
                bound = bound_local
            sigma_ss = None
        self.bound = bound
        self.last_bounds.append(self.bound)
        Ntot = mom[0]
        sx = mom[1:1 + p]
        se = mom[1 + p:1 + p + n]
        o = 1 + p + n
        ete = mom[o + p * p + p * n:].reshape(n, n)
        if use_reg:
            XtX = mom[o:o + p * p].reshape(p, p)
            Xte = mom[o + p * p:o + p * p + p * n].reshape(p, n)
            xbar, ebar = sx / Ntot, se / Ntot
            Sxx = XtX - Ntot * np.outer(xbar, xbar)
            Sxe = Xte - Ntot * np.outer(xbar, ebar)
            if self.mode == "ridge":
                coef = np.linalg.solve(Sxx + 0.1 * np.eye(p), Sxe)          # Ridge(alpha=0.1)
            elif self.mode == "lasso":                                       # Lasso(alpha=1), stm.py:677-681
                coef = lasso_from_moments(Sxx, Sxe, np.diag(ete) - Ntot * ebar * ebar, Ntot).T
            else:
                coef = np.linalg.pinv(Sxx, rcond=1e-12, hermitian=True) @ Sxe  # minimum-norm OLS
            self.gamma = coef.T                                              # (K-1) x p, stm.py:703
            cov, ratio = self._covariance_from_moments(ete, Ntot, se, XtX, Xte)
            mu_arg = dict(gamma=self.gamma)
        else:
            cov, ratio = self._covariance_from_moments(ete, Ntot, se)
            mu_arg = dict(mean_eta=se / Ntot)                                # stm.py:651
        if not self.LDAbeta:
            raise NotImplementedError("lda_beta=False (mnreg, reference stm.py:749-853) is out of scope")
        # the expansion loses -log10(ratio) digits to cancellation; every rank sees the same reduced moments,
        # so every rank takes the same branch
        exact = self.cov_exchange == "exact" or not np.isfinite(ratio) or ratio < 1e-4
        if fused and not exact:
            eng.em_finish(**mu_arg)                                          # mu, beta: enqueued, no wait
        else:
            if use_reg:
                eng.set_mu_regression(self.gamma)
            else:
                eng.set_mu_constant(mu_arg["mean_eta"])
            if exact:
                cov = self.comm.allreduce_small(eng, eng.covariance())       # stm.py:723, second all-reduce
            eng.update_beta()                                                # stm.py:741-745
        self.cov_exchanges.append("exact" if exact else "moments")
        self._fresh["mu"] = "device"
        self._fresh["beta"] = "device"
        if sigma_ss is None:
            sigma_ss = eng.get_sigma_ss()
        self._finish_sigma(cov, sigma_ss, self.sigma_prior)
        t2 = time.time()
        self.timings.append(dict(estep=t1 - t0, mstep=t2 - t1, kernels=eng.kernel_ms()))

    # ------------------------------------------------------------------ EM driver (stm.py:855-903)
    def expectation_maximization(self, saving, output_dir=None, resident=True):
        first_start_time = time.time()
        logger.info(f"Fit STM for {self.K} topics")
        for _iteration in range(100):
            if resident:
                self._em_iteration_resident()
            else:
                beta_ss, sigma_ss = self.E_step()
                self.M_step(beta_ss, sigma_ss)
            if self.EM_is_converged(_iteration):
                self.time_processed = time.time() - first_start_time
                logger.info(f"model converged in iteration {_iteration} after {self.time_processed}s")
                break
            if self.max_its_reached(_iteration):
                self.time_processed = time.time() - first_start_time
                logger.info(f"maximum number of iterations ({self.max_em_its}) reached after "
                            f"{self.time_processed} seconds")
                break
        if saving:
            assert output_dir is not None
            self.save_model(output_dir)

    fit = expectation_maximization  # the name BASELINE.json's north_star uses

    def estep(self):
        return self.E_step()

    def EM_is_converged(self, _iteration, convergence=None):
        if _iteration < 1:
            return False
        new = self.bound
        old = self.last_bounds[-2]
        convergence_check = np.abs((new - old) / np.abs(old))
        logger.info(f"relative change: {convergence_check}")
        return bool(convergence_check < self.convergence_threshold)

    def max_its_reached(self, _iteration):
        return _iteration == self.max_em_its - 1

    def stable_softmax(self, x):
        xshift = x - np.max(x)
        exps = np.exp(xshift)
        return exps / np.sum(exps)

    # ------------------------------------------------------------------ persistence (stm.py:1120-1149)
    def save_model(self, output_dir):
        """stm.py:1120-1149: the same eight files with the same names, shapes and dtypes.

        On a document-sharded fit (comm.size > 1) theta / eta / mu / X are this rank's rows: the shards travel ONCE, one array
        at a time, over the host group to rank 0 (`comm.gather`; rank order = corpus order, dist.shard_bounds cuts contiguous
        ranges), and rank 0 alone writes the N x K arrays the reference's callers read back (src/05_train.py:116,
        src/04_create_synthetic_corpora.py:61-62).  beta, sigma, gamma and the bound trace are replicated.  Every rank
        returns after rank 0 has written; a failure to write raises on every rank."""
        comm = self.comm
        if comm.size <= 1:
            self._write_model(output_dir, self.theta, self.eta, self.mu, self.X)
            return
        from .dist import sendable
        rows = {}
        for name in ("theta", "eta", "mu", "X"):
            part = getattr(self, name)
            part = None if part is None else np.asarray(part)
            # what every rank is about to send, agreed on BEFORE anything travels: a rank whose shard cannot be sent (an X dtype the
            # host group does not carry) or that holds nothing where others hold rows raises on EVERY rank here, instead of failing
            # alone inside the gather while rank 0 waits for it
            why = "" if part is None else sendable(part[:1])
            info = comm.allgather((None if part is None else (tuple(part.shape), int(part[:1].nbytes)), why))
            bad = [f"rank {r}: {w}" for r, (_, w) in enumerate(info) if w]
            if bad:
                raise RuntimeError(f"save_model: {name} cannot be gathered ({'; '.join(bad)})")
            if any(i[0] is None for i in info):
                if not all(i[0] is None for i in info):
                    raise RuntimeError(f"save_model: {name} is None on some ranks only")
                rows[name] = None
                continue
            # row blocks of at most _GATHER_BLOCK bytes per rank and message (a shard of any size stays below the host group's frame
            # limit, and rank 0 never holds more than one block per rank beside the array it assembles)
            row_bytes = max(max(i[0][1] for i in info), 1)
            per = max(1, _GATHER_BLOCK // row_bytes)
            nblk = max((i[0][0][0] + per - 1) // per for i in info)
            got = [[] for _ in range(comm.size)]
            for b in range(max(nblk, 1)):
                parts = comm.gather(part[b * per:(b + 1) * per], 0)
                if comm.rank == 0:
                    for r, q in enumerate(parts):
                        if len(q):
                            got[r].append(q)
            if comm.rank == 0:
                rows[name] = np.concatenate([q for r in got for q in r], axis=0) if any(got) else part[:0]
        err = ""
        if comm.rank == 0:
            try:
                if len(rows["theta"]) != self.N_total:
                    raise RuntimeError(f"save_model: gathered {len(rows['theta'])} rows of theta, the corpus has {self.N_total}")
                self._write_model(output_dir, rows["theta"], rows["eta"], rows["mu"], rows["X"])
            except Exception as e:      # the peers are waiting in the collective below: tell them instead of leaving them there
                err = f"{type(e).__name__}: {e}"
        err = comm.allgather(err)[0]
        if err:
            raise RuntimeError("save_model failed on rank 0: " + err)

    def _write_model(self, output_dir, theta, eta, mu, X):
        os.makedirs(output_dir, exist_ok=True)
        np.save(os.path.join(output_dir, "beta_hat"), self.beta)
        np.save(os.path.join(output_dir, "theta_hat"), theta)
        np.save(os.path.join(output_dir, "sigma_hat"), self.sigma)
        np.save(os.path.join(output_dir, "eta_hat"), eta)
        np.save(os.path.join(output_dir, "mu_hat"), mu)
        np.save(os.path.join(output_dir, "X"), X)
        if self.model == "STM":
            np.save(os.path.join(output_dir, "gamma_hat"), self.gamma)
        with open(os.path.join(output_dir, "lower_bound.pickle"), "wb") as f:
            pickle.dump(self.last_bounds, f)

    def close(self):
        if hasattr(self._engine, "close"):
            self._engine.close()
