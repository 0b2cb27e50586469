"""Device engine: a thin object over one `stm_handle` (one GPU, one HIP stream).

Holds the packed corpus (CSR), beta, eta, mu, theta and the sufficient
statistics resident in HBM and runs the E-step kernels.  Used by
strutopy_amd.stm.STM; nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import dptr, f64, iptr, lptr


check = _lib.check   # (the product library's error string; an engine on the testing build passes its own)


def device_count():
    """Usable GPUs (raises StmError when there is none -- there is no CPU fallback)."""
    n = C.c_int(0)
    check(_lib.lib().stm_device_count(C.byref(n)))
    return n.value


class HipEstepEngine:
    def __init__(self, device=0, testing=False, debug=None):
        """testing=True: the -DSTM_TESTING build of the library (strutopy_amd/libstm_hip_testing.so) -- the only one with debug
        switches; `debug`: {"STM_DEBUG_DUMP": 1, ...} set on the handle before the corpus (tests and tools)."""
        self._L = _lib.lib(testing=testing or bool(debug))
        self._h = C.c_void_p()
        self.check(self._L.stm_create(C.byref(self._h), int(device)))
        for k, v in (debug or {}).items():
            self.debug_set(k, v)
        self.device = int(device)
        self.N = self.V = self.K = self.A = 0
        self.indptr = None
        self.last_bound = 0.0

    def check(self, rc):
        _lib.check(rc, self._L)

    def debug_set(self, name, value):
        """A debug switch (named like its environment variable) on the live handle; the product build refuses."""
        self.check(self._L.stm_debug_set(self._h, name.encode(), int(value)))

    # -- lifetime -----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.stm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        hbm = C.c_int64(0)
        self.check(self._L.stm_device_info(self._h, name, 256, C.byref(cu), C.byref(hbm)))
        return dict(name=name.value.decode(), cu=cu.value, hbm_bytes=hbm.value)

    # -- corpus / model state ---------------------------------------------------------
    def set_corpus(self, indptr, indices, counts, V, aspect=None, A=1):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        counts = f64(counts)
        asp = None
        if aspect is not None and A > 1:
            asp = np.ascontiguousarray(aspect, dtype=np.int32)
        self.check(self._L.stm_set_corpus(self._h, len(indptr) - 1, int(V), lptr(indptr), iptr(indices),
                                     dptr(counts), iptr(asp) if asp is not None else None, int(A)))
        self.N, self.V, self.A = len(indptr) - 1, int(V), int(max(A, 1))
        self.indptr = indptr

    def set_topics(self, K):
        self.check(self._L.stm_set_topics(self._h, int(K)))
        self.K = int(K)

    def _beta_shape(self):
        return (self.K, self.V) if self.A == 1 else (self.A, self.K, self.V)

    def put_beta(self, beta):
        beta = f64(beta)
        if beta.shape != self._beta_shape():
            raise ValueError(f"beta has shape {beta.shape}, expected {self._beta_shape()}")
        self.check(self._L.stm_put_beta(self._h, dptr(beta)))

    def put_eta(self, eta):
        eta = f64(eta).reshape(self.N, self.K - 1)
        self.check(self._L.stm_put_eta(self._h, dptr(eta)))

    def put_mu(self, mu):
        mu = f64(mu).reshape(self.N, self.K - 1)
        self.check(self._L.stm_put_mu(self._h, dptr(mu)))

    def _get(self, fn, shape):
        out = np.empty(shape, dtype=np.float64)
        self.check(fn(self._h, dptr(out)))
        return out

    def get_beta(self):
        return self._get(self._L.stm_get_beta, self._beta_shape())

    def get_eta(self):
        return self._get(self._L.stm_get_eta, (self.N, self.K - 1))

    def get_mu(self):
        return self._get(self._L.stm_get_mu, (self.N, self.K - 1))

    def get_theta(self):
        return self._get(self._L.stm_get_theta, (self.N, self.K))

    def get_sigma_ss(self):
        return self._get(self._L.stm_get_sigma_ss, (self.K - 1, self.K - 1))

    def get_beta_ss(self):
        return self._get(self._L.stm_get_beta_ss, self._beta_shape())

    def put_sigma_ss(self, s):
        s = f64(s).reshape(self.K - 1, self.K - 1)
        self.check(self._L.stm_put_sigma_ss(self._h, dptr(s)))

    def put_beta_ss(self, b):
        b = f64(b).reshape(self._beta_shape())
        self.check(self._L.stm_put_beta_ss(self._h, dptr(b)))

    def get_bound_total(self):
        return self.last_bound

    def get_bound_docs(self):
        return self._get(self._L.stm_get_bound_docs, (self.N,))

    def get_phi_last(self):
        nd = int(self.indptr[-1] - self.indptr[-2])
        out = np.empty((self.K, nd))
        self.check(self._L.stm_get_phi(self._h, self.N - 1, dptr(out)))
        return out

    def get_diagnostics(self):
        out = {k: np.empty(self.N, dtype=np.int32) for k in ("status", "nit", "nfev", "njev", "pd_path")}
        self.check(self._L.stm_get_diagnostics(self._h, *(iptr(out[k]) for k in ("status", "nit", "nfev", "njev", "pd_path"))))
        return out

    def debug_mats(self):
        n = self.K - 1
        hess, chol, nu = (np.empty((self.N, n, n)) for _ in range(3))
        self.check(self._L.stm_debug_get_mats(self._h, dptr(hess), dptr(chol), dptr(nu)))
        return hess, chol, nu

    # -- the hot path -----------------------------------------------------------------
    def estep(self, siginv, sigmaentropy):
        """Run the E-step kernels on the resident state; returns the summed bound."""
        siginv = f64(siginv).reshape(self.K - 1, self.K - 1)
        tot = C.c_double(0.0)
        self.check(self._L.stm_estep(self._h, dptr(siginv), float(sigmaentropy), C.byref(tot)))
        self.last_bound = tot.value
        return tot.value

    def kernel_ms(self):
        """HIP-event times of the last E-step: solver kernel, post kernel + beta_ss pass ("post"), the pass alone, first to last kernel."""
        ms = (C.c_float * 3)()
        self.check(self._L.stm_last_kernel_ms(self._h, ms))
        ps = C.c_float(0.0)
        if hasattr(self._L, "stm_last_pass_ms"):      # (absent from an older build loaded for an A/B run)
            self.check(self._L.stm_last_pass_ms(self._h, C.byref(ps)))
        return {"solver": ms[0], "post": ms[1], "estep": ms[2], "pass": ps.value}

    def synchronize(self):
        self.check(self._L.stm_synchronize(self._h))

    # -- M-step pieces ---------------------------------------------------------------
    def put_covariates(self, X):
        X = f64(X).reshape(self.N, -1)
        self.check(self._L.stm_put_covariates(self._h, dptr(X), X.shape[1]))
        self.p = X.shape[1]

    def moments(self, p):
        """[N | sum_x | sum_eta | XtX | Xt_eta | eta^T eta] of this shard; also left in the packed device buffer."""
        n = self.K - 1
        L = 1 + p + n + p * p + p * n + n * n
        out = np.zeros(L)
        self.check(self._L.stm_mstep_moments(self._h, dptr(out), L))
        return out

    def set_mu_regression(self, gamma):
        gamma = f64(gamma)
        self.check(self._L.stm_mstep_set_mu(self._h, dptr(gamma), None))

    def set_mu_constant(self, mean_eta):
        mean_eta = f64(mean_eta)
        self.check(self._L.stm_mstep_set_mu(self._h, None, dptr(mean_eta)))

    def covariance(self):
        return self._get(self._L.stm_mstep_covariance, (self.K - 1, self.K - 1))

    def update_beta(self):
        self.check(self._L.stm_mstep_update_beta(self._h))

    # -- the whole iteration with one host wait ------------------------------------------------
    def em_begin(self, siginv, sigmaentropy, p):
        """E-step + moments (+ all-reduce with a communicator attached), one wait: (bound, sigma_ss, moments)."""
        n = self.K - 1
        siginv = f64(siginv).reshape(n, n)
        tot = C.c_double(0.0)
        sig = np.empty((n, n))
        mom = np.empty(1 + p + n + p * p + p * n + n * n)
        self.check(self._L.stm_em_begin(self._h, dptr(siginv), float(sigmaentropy), C.byref(tot), dptr(sig), dptr(mom), len(mom)))
        self.last_bound = tot.value
        return tot.value, sig, mom

    def em_finish(self, gamma=None, mean_eta=None):
        """mu and beta of the M-step, enqueued behind the E-step (no wait)."""
        g = None if gamma is None else f64(gamma)
        m = None if mean_eta is None else f64(mean_eta)
        self.check(self._L.stm_em_finish(self._h, dptr(g) if g is not None else None, dptr(m) if m is not None else None))

    # -- held-out likelihood ---------------------------------------------------------------
    def eval_heldout(self, indptr, indices, counts, theta=None):
        """Per-document held-out per-word log-likelihood against the resident beta (heldout.py:88-97)."""
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        counts = f64(counts)
        n = len(indptr) - 1
        th = None if theta is None else f64(theta).reshape(n, self.K)
        out = np.empty(n, dtype=np.float64)
        self.check(self._L.stm_eval_heldout(self._h, n, lptr(indptr), iptr(indices), dptr(counts),
                                       dptr(th) if th is not None else None, dptr(out)))
        return out

    # -- spectral initialisation (stm.py:30-296) ----------------------------------------------
    def spectral_gram(self, N, Vk, g):
        a = {k: np.ascontiguousarray(v) for k, v in g.items()}
        self.check(self._L.stm_spectral_gram(self._h, int(N), int(Vk), lptr(a["doc_ptr"]), iptr(a["doc_word"]), dptr(a["doc_h"]),
                                        lptr(a["word_ptr"]), iptr(a["word_doc"]), dptr(a["word_h"]), dptr(f64(a["hhat"]))))
        self._Vk = int(Vk)

    def spectral_gram_resident(self, keep, check=True):  # noqa: A002
        """gram over the resident corpus (this rank's shard), restricted to the kept terms; check=False: the caller
        sums the shards' matrices (spectral_allreduce / spectral_put_q) and then calls spectral_check."""
        keep = np.ascontiguousarray(keep, dtype=np.int32)
        self.check(self._L.stm_spectral_gram_resident(self._h, len(keep), iptr(keep), 0 if check else 1))
        self._Vk = len(keep)

    def spectral_terms(self):
        return self._Vk

    def spectral_allreduce(self):
        self.check(self._L.stm_spectral_allreduce(self._h))

    def spectral_put_q(self, Q):
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        assert Q.shape == (self._Vk, self._Vk)
        self.check(self._L.stm_spectral_put_q(self._h, dptr(Q)))

    def spectral_check(self):
        self.check(self._L.stm_spectral_check(self._h))

    def spectral_q_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        out = np.empty((len(rows), self._Vk))
        self.check(self._L.stm_spectral_get_q(self._h, iptr(rows), len(rows), dptr(out)))
        return out

    def spectral_anchors(self, K):
        out = np.zeros(int(K), dtype=np.int32)
        self.check(self._L.stm_spectral_anchors(self._h, int(K), iptr(out)))
        return out

    def spectral_project(self, anchor):
        anchor = np.ascontiguousarray(anchor, dtype=np.int32)
        out = np.empty((self._Vk, len(anchor)))
        self.check(self._L.stm_spectral_project(self._h, len(anchor), iptr(anchor), dptr(out)))
        return out

    def spectral_weights(self, anchor):
        anchor = np.ascontiguousarray(anchor, dtype=np.int32)
        out = np.empty((self._Vk, len(anchor)))
        self.check(self._L.stm_spectral_weights(self._h, len(anchor), iptr(anchor), dptr(out)))
        return out

    def spectral_release(self):
        self.check(self._L.stm_spectral_release(self._h))

    # -- multi-GPU ---------------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self.check(self._L.stm_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, rank, nranks):
        buf = C.create_string_buffer(uid, 128)
        self.check(self._L.stm_comm_init(self._h, buf, int(rank), int(nranks)))

    def set_exchange(self, mode):
        """"split" (two all-reduces per EM iteration, beta_ss behind the read-back) | "single" (one, of the whole packed buffer)."""
        if mode not in ("split", "single"):
            raise ValueError("exchange must be 'split' or 'single'")
        self.check(self._L.stm_comm_set_exchange(self._h, 1 if mode == "single" else 0))

    def comm_info(self):
        """RCCL's own view of this handle's communicator: dict(nranks, rank, device); nranks = 0 without one."""
        n, r, d = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.check(self._L.stm_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)))
        return dict(nranks=n.value, rank=r.value, device=d.value)

    def allreduce_suffstats(self, moments):
        """All-reduce the packed [bound | sigma_ss | moments | beta_ss] in place on the device; returns
        (bound, reduced moments).  `moments` is what moments() returned (its values already sit in the buffer)."""
        out = np.zeros(len(np.asarray(moments).ravel()))
        tot = C.c_double(0.0)
        self.check(self._L.stm_allreduce_suffstats(self._h, C.byref(tot), dptr(out) if len(out) else None, len(out)))
        return tot.value, out

    def allreduce_small(self, buf):
        buf = f64(buf).copy()
        flat = buf.reshape(-1)
        self.check(self._L.stm_allreduce_small(self._h, dptr(flat), flat.size))
        return buf


def estep_host(indptr, indices, counts, beta, mu, eta, siginv, sigmaentropy, aspect=None, device=0, testing=False):
    """One-shot E-step over host arrays through stm_estep_host (upload, run, download).  testing=True: through the -DSTM_TESTING
    build, whose handles read the STM_DEBUG_* environment switches when they are created (tests)."""
    L = _lib.lib(testing=testing)
    beta = f64(beta)
    if beta.ndim == 2:
        A, (K, V) = 1, beta.shape
    else:
        A, K, V = beta.shape
    n = K - 1
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    counts = f64(counts)
    N = len(indptr) - 1
    mu = f64(mu).reshape(N, n)
    eta = f64(eta).reshape(N, n).copy()
    siginv = f64(siginv).reshape(n, n)
    out = dict(eta=eta, theta=np.zeros((N, K)), bound_doc=np.zeros(N), sigma_ss=np.zeros((n, n)),
               beta_ss=np.zeros_like(beta))
    for k in ("status", "nit", "nfev", "njev", "pd_path"):
        out[k] = np.zeros(N, dtype=np.int32)
    tot = np.zeros(1)
    a = _lib.EstepArgs()
    a.N, a.K, a.V, a.A = N, K, V, A
    a.indptr, a.indices, a.counts = lptr(indptr), iptr(indices), dptr(counts)
    if aspect is not None and A > 1:
        aspect = np.ascontiguousarray(aspect, dtype=np.int32)
        a.aspect = iptr(aspect)
    a.beta, a.mu, a.eta, a.siginv = dptr(beta), dptr(mu), dptr(eta), dptr(siginv)
    a.sigmaentropy = float(sigmaentropy)
    a.theta, a.bound, a.sigma_ss, a.beta_ss, a.bound_total = (
        dptr(out["theta"]), dptr(out["bound_doc"]), dptr(out["sigma_ss"]), dptr(out["beta_ss"]), dptr(tot))
    for k in ("status", "nit", "nfev", "njev", "pd_path"):
        setattr(a, k, iptr(out[k]))
    _lib.check(L.stm_estep_host(C.byref(a), int(device)), L)
    out["bound"] = float(tot[0])
    return out
