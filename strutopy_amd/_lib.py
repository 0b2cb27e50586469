"""ctypes binding of the C-ABI in include/stm_estep.h (libstm_hip.so).

Host code stays Python + NumPy; this module is the only place that touches the
shared library.  There is NO CPU fallback: if the HIP library is missing or no
GPU is usable, loading / creating a handle raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STM_LIB_PATH") or os.path.join(_HERE, "libstm_hip.so")   # STM_LIB_PATH: A/B builds of the same source
# the same sources built with -DSTM_TESTING: the debug switches (dumps, cycle counters, poisoned LDS, partial E-steps, the fault
# injector) exist only there -- what tests/ and tools/ load when they need one (HipEstepEngine(testing=True))
TESTING_LIB_PATH = os.environ.get("STM_TESTING_LIB_PATH") or os.path.join(_HERE, "libstm_hip_testing.so")

STM_OK = 0
STM_ERR_INVALID, STM_ERR_BETA, STM_ERR_LINALG, STM_ERR_HIP = 1, 2, 3, 4
STM_ERR_NO_DEVICE, STM_ERR_COMM, STM_ERR_PHI = 5, 6, 7

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_h = C.c_void_p


class EstepArgs(C.Structure):
    """struct stm_estep_args (include/stm_estep.h)."""
    _fields_ = [
        ("N", C.c_int64), ("K", C.c_int32), ("V", C.c_int32), ("A", C.c_int32),
        ("indptr", _lp), ("indices", _ip), ("counts", _dp), ("aspect", _ip),
        ("beta", _dp), ("mu", _dp), ("eta", _dp), ("siginv", _dp), ("sigmaentropy", C.c_double),
        ("theta", _dp), ("bound", _dp), ("sigma_ss", _dp), ("beta_ss", _dp), ("bound_total", _dp),
        ("status", _ip), ("nit", _ip), ("nfev", _ip), ("njev", _ip), ("pd_path", _ip),
    ]


# name -> (restype, argtypes); every symbol include/stm_estep.h declares
SIGNATURES = {
    "stm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "stm_create": (C.c_int, [C.POINTER(_h), C.c_int]),
    "stm_destroy": (None, [_h]),
    "stm_last_error": (C.c_char_p, []),
    "stm_device_info": (C.c_int, [_h, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "stm_set_corpus": (C.c_int, [_h, C.c_int64, C.c_int32, _lp, _ip, _dp, _ip, C.c_int32]),
    "stm_set_topics": (C.c_int, [_h, C.c_int32]),
    "stm_put_beta": (C.c_int, [_h, _dp]),
    "stm_put_eta": (C.c_int, [_h, _dp]),
    "stm_put_mu": (C.c_int, [_h, _dp]),
    "stm_get_beta": (C.c_int, [_h, _dp]),
    "stm_get_eta": (C.c_int, [_h, _dp]),
    "stm_get_mu": (C.c_int, [_h, _dp]),
    "stm_get_theta": (C.c_int, [_h, _dp]),
    "stm_estep": (C.c_int, [_h, _dp, C.c_double, _dp]),
    "stm_get_sigma_ss": (C.c_int, [_h, _dp]),
    "stm_get_beta_ss": (C.c_int, [_h, _dp]),
    "stm_get_bound_docs": (C.c_int, [_h, _dp]),
    "stm_put_sigma_ss": (C.c_int, [_h, _dp]),
    "stm_put_beta_ss": (C.c_int, [_h, _dp]),
    "stm_get_diagnostics": (C.c_int, [_h, _ip, _ip, _ip, _ip, _ip]),
    "stm_get_phi": (C.c_int, [_h, C.c_int64, _dp]),
    "stm_estep_host": (C.c_int, [C.POINTER(EstepArgs), C.c_int]),
    "stm_put_covariates": (C.c_int, [_h, _dp, C.c_int32]),
    "stm_mstep_moments": (C.c_int, [_h, _dp, C.c_int64]),
    "stm_mstep_set_mu": (C.c_int, [_h, _dp, _dp]),
    "stm_mstep_covariance": (C.c_int, [_h, _dp]),
    "stm_mstep_update_beta": (C.c_int, [_h]),
    "stm_em_begin": (C.c_int, [_h, _dp, C.c_double, _dp, _dp, _dp, C.c_int64]),
    "stm_em_finish": (C.c_int, [_h, _dp, _dp]),
    "stm_eval_heldout": (C.c_int, [_h, C.c_int64, _lp, _ip, _dp, _dp, _dp]),
    "stm_spectral_gram": (C.c_int, [_h, C.c_int64, C.c_int32, _lp, _ip, _dp, _lp, _ip, _dp, _dp]),
    "stm_spectral_gram_resident": (C.c_int, [_h, C.c_int32, _ip, C.c_int32]),
    "stm_spectral_allreduce": (C.c_int, [_h]),
    "stm_spectral_put_q": (C.c_int, [_h, _dp]),
    "stm_spectral_check": (C.c_int, [_h]),
    "stm_spectral_get_q": (C.c_int, [_h, _ip, C.c_int32, _dp]),
    "stm_spectral_anchors": (C.c_int, [_h, C.c_int32, _ip]),
    "stm_spectral_project": (C.c_int, [_h, C.c_int32, _ip, _dp]),
    "stm_spectral_weights": (C.c_int, [_h, C.c_int32, _ip, _dp]),
    "stm_spectral_release": (C.c_int, [_h]),
    "stm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "stm_comm_init": (C.c_int, [_h, C.c_void_p, C.c_int, C.c_int]),
    "stm_comm_info": (C.c_int, [_h, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stm_comm_set_exchange": (C.c_int, [_h, C.c_int32]),
    "stm_allreduce_suffstats": (C.c_int, [_h, _dp, _dp, C.c_int64]),
    "stm_allreduce_small": (C.c_int, [_h, _dp, C.c_int64]),
    "stm_last_kernel_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "stm_last_pass_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "stm_synchronize": (C.c_int, [_h]),
}
# not part of the public header: debug dumps used by the parity tests
_DEBUG_SIGNATURES = {
    "stm_debug_get_mats": (C.c_int, [_h, _dp, _dp, _dp]),
    "stm_debug_get_prof": (C.c_int, [_h, C.POINTER(C.c_longlong)]),
    "stm_debug_set": (C.c_int, [_h, C.c_char_p, C.c_int]),
    "stm_is_testing_build": (C.c_int, []),
}

_SINCE_ROUND6 = ("stm_comm_set_exchange", "stm_last_pass_ms", "stm_debug_set", "stm_is_testing_build")
_LIB = None
_TESTING_LIB = None


class StmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libstm_hip error {code}: {msg}")
        self.code = code


def _load(path):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  strutopy_amd has no CPU fallback.")
    L = C.CDLL(path)
    for name, (res, args) in {**SIGNATURES, **_DEBUG_SIGNATURES}.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            # an older build loaded through STM_LIB_PATH for an A/B run (tools/bitcmp.py against last round's library) may lack the
            # newest entry points; the library next to this file must have every one of them (tests/test_cabi.py)
            if name in _SINCE_ROUND6 and os.environ.get("STM_LIB_PATH"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    return L


def lib(testing=False):
    """Load libstm_hip.so (built by __graft_entry__.build()); raise if it is missing.  testing=True: the -DSTM_TESTING build."""
    global _LIB, _TESTING_LIB
    if testing:
        if _TESTING_LIB is None:
            _TESTING_LIB = _load(TESTING_LIB_PATH)
            if not _TESTING_LIB.stm_is_testing_build():
                raise ImportError(f"{TESTING_LIB_PATH} was not built with -DSTM_TESTING")
        return _TESTING_LIB
    if _LIB is None:
        _LIB = _load(LIB_PATH)
    return _LIB


def check(rc, L=None):
    """Map C-ABI error codes onto the exceptions the reference raises for the same condition."""
    if rc == STM_OK:
        return
    msg = (L or lib()).stm_last_error().decode(errors="replace")
    if rc in (STM_ERR_BETA, STM_ERR_PHI):
        raise AssertionError(msg)                 # stm.py:534 / stm.py:1117 are `assert`s
    if rc == STM_ERR_LINALG:
        raise np.linalg.LinAlgError(msg)          # np.linalg.cholesky failure, stm.py:1040
    if rc == STM_ERR_INVALID:
        raise ValueError(msg)
    raise StmError(rc, msg)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def lptr(a):
    return a.ctypes.data_as(_lp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
