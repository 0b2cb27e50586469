import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def reference_beta0(K, V):
    """The reference's random init (stm.py:361,425-429): numpy legacy RNG seeded with 123456."""
    rs = np.random.RandomState(123456)
    b = rs.gamma(0.1, 1, V * K).reshape(K, V)
    return b / b.sum(axis=1)[:, None]


@pytest.fixture(scope="session")
def oracle():
    from oracle import stm_oracle
    stm_oracle.build()
    return stm_oracle


def has_gpu():
    try:
        from strutopy_amd.engine import HipEstepEngine
        e = HipEstepEngine(0)
        e.close()
        return True
    except Exception:
        return False
