"""Stand-in for `qpsolvers` (absent from this image; tools/make_golden.py only).

`solve_qp` is called by the reference in exactly one place, recover_l2 of the spectral initialisation
(src/modules/stm.py:271), with solver="quadprog".  quadprog implements the dual active-set method of Goldfarb & Idnani
(1983); this stand-in forwards to the restatement of that published algorithm in oracle/spectral_oracle.py.  Contains no
reference code.  Goldens that depend on it say so (tests/golden/spectral_*.npz, key `qp_solver`).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle.spectral_oracle import solve_qp  # noqa: E402,F401
