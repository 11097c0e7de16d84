"""Inert stand-in so that /root/reference/car_racing/utils/base.py:11-12 imports (golden tooling only)."""


def spmatrix(*a, **k):
    raise NotImplementedError("cvxopt is not available; shim is import-only")


def matrix(*a, **k):
    raise NotImplementedError("cvxopt is not available; shim is import-only")
