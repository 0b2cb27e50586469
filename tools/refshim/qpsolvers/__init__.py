"""Stand-in for `qpsolvers` (absent here). Only spectral init uses it; goldens use random init."""


def solve_qp(*args, **kwargs):
    raise NotImplementedError("qpsolvers is not available in this image (spectral init out of scope)")
