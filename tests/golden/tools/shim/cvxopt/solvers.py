"""Stand-in (golden tooling only); see cvxopt/__init__.py."""
import numpy as np

options = {}


def qp(P, q, *a, **k):
    if a or k:
        raise NotImplementedError("only the unconstrained form qp(Q, b) is restated")
    P = np.asarray(P, float)
    q = np.asarray(q, float).reshape(-1)
    return {"x": np.linalg.solve(P, -q).reshape(-1, 1), "status": "optimal"}
