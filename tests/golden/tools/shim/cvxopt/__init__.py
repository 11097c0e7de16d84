"""Stand-in so that /root/reference/car_racing/utils/base.py:11-12 and control/lmpc_helper.py:2,6
import and run (golden tooling only).  cvxopt==1.3.0 is not installable offline.  The reference uses
it for ONE thing: `qp(Q, b)` WITHOUT constraints (lmpc_helper.py:358-366), i.e. the minimiser of
1/2 x'Qx + b'x, which cvxopt obtains from the linear system Q x = -b; `matrix` is its dense
column-major container.  Both are restated with numpy here."""
import numpy as np


def spmatrix(*a, **k):
    raise NotImplementedError("cvxopt.spmatrix is not used on any path the goldens exercise")


def matrix(a, *k, **kw):
    return np.array(a, dtype=float)
