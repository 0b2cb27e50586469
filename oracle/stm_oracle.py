"""ctypes loader for the CPU oracle (oracle/libstm_oracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/stm_oracle.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
strutopy_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class _Args(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("K", C.c_int32), ("V", C.c_int32), ("A", C.c_int32),
        ("indptr", _lp), ("indices", _ip), ("counts", _dp), ("aspect", _ip),
        ("beta", _dp), ("mu", _dp), ("eta", _dp), ("siginv", _dp), ("sigmaentropy", C.c_double),
        ("theta", _dp), ("bound", _dp), ("sigma_ss", _dp), ("beta_ss", _dp), ("bound_total", _dp),
        ("status", _ip), ("nit", _ip), ("nfev", _ip), ("njev", _ip), ("pd_path", _ip),
        ("hess_out", _dp), ("chol_out", _dp), ("nu_out", _dp), ("phi_last", _dp),
        ("pivot_margin", _dp),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libstm_oracle.so")
    src = os.path.join(_HERE, "stm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libstm_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libstm_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.stm_oracle_estep.argtypes = [C.POINTER(_Args), C.c_int]
        L.stm_oracle_estep.restype = C.c_int
        L.stm_oracle_last_error.restype = C.c_char_p
        L.stm_oracle_max_threads.restype = C.c_int
        L.stm_oracle_f.restype = C.c_double
        L.stm_oracle_f.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.stm_oracle_df.restype = None
        L.stm_oracle_df.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.stm_oracle_bfgs.restype = C.c_int
        L.stm_oracle_bfgs.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip, _dp]
        L.stm_oracle_make_pd.restype = None
        L.stm_oracle_make_pd.argtypes = [C.c_int, _dp]
        L.stm_oracle_decompose.restype = C.c_int
        L.stm_oracle_decompose.argtypes = [C.c_int, _dp, _dp, _dp]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def max_threads():
    return lib().stm_oracle_max_threads()


def preamble(sigma):
    """stm.py:499-501, the same numpy expression the reference evaluates."""
    sigobj = np.linalg.cholesky(sigma)
    sigmaentropy = np.sum(np.log(np.diag(sigobj)))
    siginv = np.linalg.inv(sigobj).T * np.linalg.inv(sigobj)
    return siginv, float(sigmaentropy)


def estep(indptr, indices, counts, beta, mu, eta, siginv, sigmaentropy, aspect=None,
          nthreads=1, dump_mats=False):
    """Run the oracle E-step.  Returns a dict of outputs (eta is copied, not modified)."""
    beta = _f64(beta)
    if beta.ndim == 2:
        A, (K, V) = 1, beta.shape
    else:
        A, K, V = beta.shape
    n = K - 1
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    counts = _f64(counts)
    N = len(indptr) - 1
    mu = _f64(mu).reshape(N, n)
    eta = _f64(eta).reshape(N, n).copy()
    siginv = _f64(siginv).reshape(n, n)
    out = dict(
        eta=eta, theta=np.zeros((N, K)), bound_doc=np.zeros(N), sigma_ss=np.zeros((n, n)),
        beta_ss=np.zeros_like(beta), status=np.zeros(N, np.int32), nit=np.zeros(N, np.int32),
        nfev=np.zeros(N, np.int32), njev=np.zeros(N, np.int32), pd_path=np.zeros(N, np.int32),
    )
    tot = np.zeros(1)
    last_nd = int(indptr[-1] - indptr[-2]) if N > 0 else 0
    phi_last = np.zeros((K, max(last_nd, 1)))
    a = _Args()
    a.N, a.K, a.V, a.A = N, K, V, A
    a.indptr = indptr.ctypes.data_as(_lp)
    a.indices = indices.ctypes.data_as(_ip)
    a.counts = _d(counts)
    if aspect is not None:
        aspect = np.ascontiguousarray(aspect, dtype=np.int32)
        a.aspect = aspect.ctypes.data_as(_ip)
    a.beta, a.mu, a.eta, a.siginv = _d(beta), _d(mu), _d(eta), _d(siginv)
    a.sigmaentropy = float(sigmaentropy)
    a.theta, a.bound, a.sigma_ss, a.beta_ss, a.bound_total = (
        _d(out["theta"]), _d(out["bound_doc"]), _d(out["sigma_ss"]), _d(out["beta_ss"]), _d(tot))
    for k in ("status", "nit", "nfev", "njev", "pd_path"):
        setattr(a, k, out[k].ctypes.data_as(_ip))
    if dump_mats:
        for k, f in (("hess", "hess_out"), ("chol", "chol_out"), ("nu", "nu_out")):
            out[k] = np.zeros((N, n, n))
            setattr(a, f, _d(out[k]))
    a.phi_last = _d(phi_last)
    out["pivot_margin"] = np.zeros(N)
    a.pivot_margin = _d(out["pivot_margin"])
    rc = lib().stm_oracle_estep(C.byref(a), int(nthreads))
    if rc != 0:
        msg = lib().stm_oracle_last_error().decode()
        if rc == 2:
            raise AssertionError(msg)
        raise np.linalg.LinAlgError(msg)
    out["bound"] = float(tot[0])
    out["phi_last"] = phi_last[:, :last_nd]
    return out


def f(K, eta, mu, counts, betad, siginv):
    eta, mu, counts, betad, siginv = map(_f64, (eta, mu, counts, betad, siginv))
    return lib().stm_oracle_f(K, len(counts), _d(eta), _d(mu), _d(counts), _d(betad), _d(siginv))


def df(K, eta, mu, counts, betad, siginv):
    eta, mu, counts, betad, siginv = map(_f64, (eta, mu, counts, betad, siginv))
    g = np.zeros(K - 1)
    lib().stm_oracle_df(K, len(counts), _d(eta), _d(mu), _d(counts), _d(betad), _d(siginv), _d(g))
    return g


def bfgs(K, eta0, mu, counts, betad, siginv):
    mu, counts, betad, siginv = map(_f64, (mu, counts, betad, siginv))
    x = _f64(eta0).copy()
    nit, nfev, njev = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    fun = C.c_double(0)
    st = lib().stm_oracle_bfgs(K, len(counts), _d(x), _d(mu), _d(counts), _d(betad), _d(siginv),
                               C.byref(nit), C.byref(nfev), C.byref(njev), C.byref(fun))
    return dict(x=x, status=st, nit=nit.value, nfev=nfev.value, njev=njev.value, fun=fun.value)


def make_pd(M):
    M = _f64(M).copy()
    lib().stm_oracle_make_pd(M.shape[0], _d(M))
    return M


def decompose(H):
    H = _f64(H).copy()
    n = H.shape[0]
    L = np.zeros((n, n))
    nu = np.zeros((n, n))
    path = lib().stm_oracle_decompose(n, _d(H), _d(L), _d(nu))
    return path, L, nu


def eval_heldout_docs(indptr, indices, counts, theta, beta):
    """Per-document values of the reference's eval_heldout (src/modules/heldout.py:88-97), numpy loop:
    sum_w c_w log(theta_d @ beta[:, w]) / sum_w c_w; the reference returns np.mean of these."""
    indptr = np.asarray(indptr, dtype=np.int64)
    out = np.empty(len(indptr) - 1)
    for i in range(len(out)):
        sl = slice(indptr[i], indptr[i + 1])
        word_ll = counts[sl] * np.log(theta[i] @ beta[:, indices[sl]])
        out[i] = np.sum(word_ll) / np.sum(counts[sl])
    return out
