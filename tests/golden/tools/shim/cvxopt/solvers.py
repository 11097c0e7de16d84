"""Inert stand-in (golden tooling only); see cvxopt/__init__.py."""
options = {}


def qp(*a, **k):
    raise NotImplementedError("cvxopt is not available; shim is import-only")
