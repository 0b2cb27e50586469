"""CPU stand-in for strutopy_amd.engine.HipEstepEngine, backed by the oracle.

TEST INFRASTRUCTURE: lets the `-m "not gpu"` suite exercise the host logic of
strutopy_amd.stm.STM (state plumbing, M-step from moments, sharding + all-reduce)
without a GPU.  It is injected through the `engine=` test hook; the product never
imports it.
"""
import numpy as np

from oracle import stm_oracle


class OracleEngine:
    def __init__(self, nthreads=0):
        self.nthreads = nthreads
        self.last_bound = 0.0

    def set_corpus(self, indptr, indices, counts, V, aspect=None, A=1):
        self.indptr = np.asarray(indptr, dtype=np.int64)
        self.indices = np.asarray(indices, dtype=np.int32)
        self.counts = np.asarray(counts, dtype=np.float64)
        self.V, self.A = int(V), int(max(A, 1))
        self.aspect = None if aspect is None else np.asarray(aspect, dtype=np.int32)
        self.N = len(self.indptr) - 1

    def set_topics(self, K):
        self.K = int(K)
        n = K - 1
        self.eta = np.zeros((self.N, n))
        self.mu = np.zeros((self.N, n))
        self.theta = np.zeros((self.N, K))

    def put_beta(self, b): self.beta = np.array(b, dtype=np.float64)
    def put_eta(self, e): self.eta = np.array(e, dtype=np.float64).reshape(self.N, self.K - 1)
    def put_mu(self, m): self.mu = np.array(m, dtype=np.float64).reshape(self.N, self.K - 1)
    def get_beta(self): return self.beta.copy()
    def get_eta(self): return self.eta.copy()
    def get_mu(self): return self.mu.copy()
    def get_theta(self): return self.theta.copy()
    def get_sigma_ss(self): return self.sigma_ss.copy()
    def get_beta_ss(self): return self.beta_ss.copy()
    def put_sigma_ss(self, s): self.sigma_ss = np.array(s, dtype=np.float64).reshape(self.K - 1, self.K - 1)
    def put_beta_ss(self, b): self.beta_ss = np.array(b, dtype=np.float64).reshape(self.beta.shape)
    def get_bound_docs(self): return self.bound_doc.copy()
    def get_bound_total(self): return self.last_bound
    def get_phi_last(self): return self.phi_last.copy()
    def get_diagnostics(self): return dict(self.diag)
    def kernel_ms(self): return dict(solver=0.0, post=0.0, estep=0.0)
    def synchronize(self): pass
    def close(self): pass

    def estep(self, siginv, sigmaentropy):
        o = stm_oracle.estep(self.indptr, self.indices, self.counts, self.beta, self.mu, self.eta, siginv,
                             sigmaentropy, aspect=self.aspect if self.A > 1 else None, nthreads=self.nthreads)
        self.eta, self.theta = o["eta"], o["theta"]
        self.sigma_ss, self.beta_ss, self.bound_doc = o["sigma_ss"], o["beta_ss"], o["bound_doc"]
        self.phi_last = o["phi_last"]
        self.diag = {k: o[k] for k in ("status", "nit", "nfev", "njev", "pd_path")}
        self.last_bound = o["bound"]
        return o["bound"]

    # ---- M-step pieces (numpy statements of the kernels in stm_mstep.h)
    def put_covariates(self, X):
        self.X = np.array(X, dtype=np.float64).reshape(self.N, -1)

    def moments(self, p):
        n = self.K - 1
        ete = (self.eta.T @ self.eta).ravel()
        if p == 0:
            return np.concatenate([[float(self.N)], self.eta.sum(0), ete])
        X = self.X
        return np.concatenate([[float(self.N)], X.sum(0), self.eta.sum(0), (X.T @ X).ravel(), (X.T @ self.eta).ravel(), ete])

    def set_mu_regression(self, gamma): self.mu = self.X @ np.asarray(gamma).T
    def set_mu_constant(self, mean_eta): self.mu = np.repeat(np.asarray(mean_eta)[None, :], self.N, axis=0)
    def covariance(self):
        d = self.eta - self.mu
        return d.T @ d

    def update_beta(self):
        bss = self.beta_ss
        rs = np.sum(bss, axis=1)[:, None]
        self.beta = np.divide(bss, rs, out=np.zeros_like(bss), where=rs != 0)

    def allreduce_suffstats(self, extra):
        return self.last_bound, np.array(extra, dtype=np.float64, copy=True)

    def allreduce_small(self, buf):
        return np.array(buf, dtype=np.float64, copy=True)

    # ---- spectral initialisation: the NumPy restatement (oracle/spectral_oracle.py) behind the engine's interface
    def spectral_gram(self, N, Vk, g):
        from oracle import spectral_oracle as so
        indptr = np.asarray(g["doc_ptr"])
        D = np.zeros((N, Vk))
        doc = np.repeat(np.arange(N), np.diff(indptr))
        D[doc, g["doc_word"]] = g["doc_h"]
        self._sq0 = D.T @ D - np.diag(g["hhat"])
        assert np.all(self._sq0.sum(axis=1) > 0), "Encountered zeroes in Q row sums, can not normalize."
        self._so = so

    def spectral_gram_resident(self, keep, check=True):
        """gram over this engine's own (shard of the) corpus, as stm_spectral_gram_resident: the oracle's NumPy restatement"""
        from oracle import spectral_oracle as so
        from strutopy_amd.corpus import PackedCorpus
        from strutopy_amd.spectral import gram_inputs
        c = PackedCorpus(self.indptr, self.indices, self.counts, self.V)
        g = gram_inputs(c, np.asarray(keep))
        Vk = len(keep)
        D = np.zeros((c.N, Vk))
        doc = np.repeat(np.arange(c.N), np.diff(g["doc_ptr"]))
        D[doc, g["doc_word"]] = g["doc_h"]
        self._sq0 = D.T @ D - np.diag(g["hhat"])
        self._so = so
        if check:
            self.spectral_check()

    def spectral_terms(self): return self._sq0.shape[0]
    def spectral_put_q(self, Q): self._sq0 = np.array(Q, dtype=np.float64)

    def spectral_check(self):
        assert np.all(self._sq0.sum(axis=1) > 0), "Encountered zeroes in Q row sums, can not normalize."

    def spectral_anchors(self, K):
        anchor, self._sq0 = self._so.fast_anchor(self._sq0, K)
        return np.asarray(anchor, dtype=np.int32)

    def spectral_q_rows(self, rows): return self._sq0[np.intp(rows)].copy()
    def spectral_project(self, anchor): return self._sq0 @ self._sq0[np.intp(anchor)].T
    def spectral_weights(self, anchor): return self._so.recover_l2_weights(self._sq0, anchor)   # Goldfarb-Idnani, per term
    def spectral_release(self): self._sq0 = None
