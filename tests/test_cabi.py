"""The C-ABI boundary (include/stm_estep.h <-> strutopy_amd/libstm_hip.so), checked without a GPU:
the library loads, exports every symbol the header declares, the ctypes table covers them all,
and -- there being no CPU fallback -- creating a handle without a GPU fails loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu

from strutopy_amd import _lib

HEADER = os.path.join(ROOT, "include", "stm_estep.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(stm_[a-z0-9_]+)\s*\(", txt)))


def test_header_cites_the_reference_interface():
    txt = open(HEADER).read()
    for cite in ("stm.py:489-597", "stm.py:534", "stm.py:1040", "stm.py:1117", "stm.py:622-747"):
        assert cite in txt
    assert 'extern "C"' in txt and "torch" not in txt.replace("no torch", "")


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 30
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/stm_estep.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert _lib.lib() is _lib.lib()


def test_estep_args_struct_layout():
    # struct stm_estep_args: int64 N; int32 K, V, A; (pad) 8 pointers ... ; must match the header order
    fields = [f[0] for f in _lib.EstepArgs._fields_]
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct stm_estep_args \{(.*?)\} stm_estep_args;", txt, flags=re.S).group(1)
    order = re.findall(r"\*?\s*\b([A-Za-z_]+)\s*(?:,|;)", body)
    assert [o for o in order if o in fields] == fields
    assert ctypes.sizeof(_lib.EstepArgs) == 8 + 4 * 3 + 4 + 8 * 8 + 8 + 8 * 10  # 176 bytes on LP64


def test_error_codes_map_to_reference_exceptions(monkeypatch):
    class Fake:
        def stm_last_error(self):
            return b"msg"
    monkeypatch.setattr(_lib, "_LIB", Fake())
    with pytest.raises(AssertionError):
        _lib.check(_lib.STM_ERR_BETA)          # stm.py:534 is an assert
    with pytest.raises(AssertionError):
        _lib.check(_lib.STM_ERR_PHI)           # stm.py:1117
    with pytest.raises(np.linalg.LinAlgError):
        _lib.check(_lib.STM_ERR_LINALG)        # np.linalg.cholesky, stm.py:1040
    with pytest.raises(ValueError):
        _lib.check(_lib.STM_ERR_INVALID)
    with pytest.raises(_lib.StmError):
        _lib.check(_lib.STM_ERR_NO_DEVICE)
    _lib.check(_lib.STM_OK)


@pytest.mark.skipif(has_gpu(), reason="this box has a GPU")
def test_no_cpu_fallback_without_a_gpu():
    from strutopy_amd.corpus import synthetic_corpus
    from strutopy_amd.engine import HipEstepEngine
    from strutopy_amd.stm import STM
    with pytest.raises(_lib.StmError) as ei:
        HipEstepEngine(0)
    assert ei.value.code == _lib.STM_ERR_NO_DEVICE
    syn = synthetic_corpus(8, 50, 3, n_words=10, seed=1)
    with pytest.raises(_lib.StmError):
        STM(documents=syn.corpus, dictionary=None, content=False, K=3, X=syn.X, kappa_interactions=False,
            max_em_iter=1, sigma_prior=0, convergence_threshold=1e-5, init_type="random")


def test_null_and_order_checks_do_not_need_a_device():
    L = _lib.lib()
    assert L.stm_set_topics(None, 5) == _lib.STM_ERR_INVALID
    assert L.stm_estep(None, None, 0.0, None) == _lib.STM_ERR_INVALID
    assert L.stm_create(None, 0) == _lib.STM_ERR_INVALID
    assert b"" != L.stm_last_error()
    L.stm_destroy(None)


def test_integration_stub_matches_the_binding():
    """The ctypes stub INTEGRATION.md hands to a reference maintainer declares struct stm_estep_args exactly like the package's own
    binding (field names, order, C types), and only calls entry points the header declares."""
    import re
    import ctypes as C
    from strutopy_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("class _Args(C.Structure)"):text.index("_lib.stm_estep_host.argtypes")]
    doc_fields = re.findall(r'\("(\w+)",\s*([\w.]+)\)', block)
    types = {"C.c_int64": C.c_int64, "C.c_int32": C.c_int32, "C.c_double": C.c_double, "_dp": C.POINTER(C.c_double),
             "_ip": C.POINTER(C.c_int32), "_lp": C.POINTER(C.c_int64)}
    ours = [(n, t) for n, t in _lib.EstepArgs._fields_]
    assert [n for n, _ in doc_fields] == [n for n, _ in ours]
    for (n, tname), (_, t) in zip(doc_fields, ours):
        assert C.sizeof(types[tname]) == C.sizeof(t) and types[tname]._type_ == t._type_, n
    header = open(os.path.join(ROOT, "include", "stm_estep.h")).read()
    for sym in set(re.findall(r"_lib\.(stm_\w+)", text)):
        assert re.search(r"\b" + sym + r"\s*\(", header), sym
