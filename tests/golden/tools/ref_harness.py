"""Import the (Python) reference in THIS container with inert/recording stand-ins for the three
third-party wheels that cannot be installed here (casadi, cvxopt, pathos).

TEST TOOLING ONLY: used by make_golden.py to generate tests/golden/*.npz.  Nothing here is
imported by the product, by `-m gpu` tests, by bench.py or by smoke(); /root/reference does not
exist on the GPU box.
"""
import os
import sys

import numpy as np

REF_ROOT = os.environ.get("CRX_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "car_racing"))


def install():
    """Put shims + the reference's top-level packages (setup.cfg:16-17 `package_dir = =car_racing`)
    on sys.path, chdir to the reference root (its *Param defaults read CSVs relative to CWD at
    import time, utils/base.py:124-125) and return the imported modules."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.setrecursionlimit(100000)
    if not hasattr(np, "asscalar"):  # removed in NumPy 1.23; racing_env.py:25,86 still calls it
        np.asscalar = lambda a: a.item()
    shim = os.path.join(HERE, "shim")
    for p in (os.path.join(REF_ROOT, "car_racing"), shim, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("MPLBACKEND", "Agg")
    os.chdir(REF_ROOT)
    import casadi  # the shim
    import nlp_solve

    casadi.Opti.solver_fn = staticmethod(nlp_solve.solve_recorded)
    from control import control
    from planning import overtake_traj_planner, planner_helper
    from racing import offboard
    from utils import base, racing_env

    return dict(
        casadi=casadi,
        control=control,
        planner=overtake_traj_planner,
        planner_helper=planner_helper,
        offboard=offboard,
        base=base,
        racing_env=racing_env,
    )
