"""Recording stand-in for the subset of CasADi's ``Opti`` stack that the reference touches.

TEST TOOLING ONLY (never imported by the product, never shipped to the GPU box as part of a
solve path).  The reference's solver front-ends build their problems through ``casadi.Opti``
(/root/reference/car_racing/control/control.py:492-597,
/root/reference/car_racing/planning/overtake_traj_planner.py:263-364).  CasADi 3.5.5 / IPOPT are
not installable in this container, so ``make_golden.py`` puts THIS module on
``sys.modules['casadi']`` and runs the reference's own, unmodified code against it.  The shim

  * records every decision variable, constraint, cost term and initial guess the reference emits,
    as small expression graphs with exact forward-mode derivatives,
  * hands the recorded NLP to an independent solver (``nlp_solve.py``: SciPy SLSQP + an
    active-set Newton polish + an explicit KKT certificate), and
  * raises ``RuntimeError`` from ``Opti.solve`` when the problem is infeasible, which is what
    CasADi does for every non-success IPOPT return status, so that the reference's own
    ``except RuntimeError`` fallbacks run.

The recorded problem + certified solution become the committed fixtures in tests/golden/*.npz.
"""
import numpy as np

__all__ = ["Opti", "mtimes", "MX", "vertcat", "horzcat", "inf"]


class Node:
    """Scalar expression node; value + dense gradient are computed by ``ev``."""

    __slots__ = ("op", "a", "b", "_cache_key", "_cache_val")
    __array_ufunc__ = None  # make numpy scalars defer to our reflected operators

    def __init__(self, op, a=None, b=None):
        self.op, self.a, self.b = op, a, b
        self._cache_key = None
        self._cache_val = None

    # -- construction helpers -------------------------------------------------------------
    @staticmethod
    def lift(v):
        if isinstance(v, Node):
            return v
        if isinstance(v, MX):
            assert v.a.shape == (1, 1)
            return v.a[0, 0]
        v = np.asarray(v, dtype=float)
        assert v.size == 1, "scalar expected"
        return Node("const", float(v.reshape(-1)[0]))

    def __add__(self, o):
        return Node("add", self, Node.lift(o))

    __radd__ = __add__

    def __sub__(self, o):
        return Node("sub", self, Node.lift(o))

    def __rsub__(self, o):
        return Node("sub", Node.lift(o), self)

    def __mul__(self, o):
        return Node("mul", self, Node.lift(o))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return Node("mul", self, Node("const", 1.0 / float(o)))

    def __neg__(self):
        return Node("mul", self, Node("const", -1.0))

    def __pow__(self, p):
        p = int(p)
        assert p >= 1
        return Node("pow", self, p)

    # -- evaluation -----------------------------------------------------------------------
    def ev(self, z, key):
        """Return (value, gradient[n]) at point z; ``key`` identifies z for memoisation."""
        if self._cache_key == key:
            return self._cache_val
        op = self.op
        if op == "const":
            r = (self.a, 0.0)
        elif op == "var":
            g = np.zeros(z.size)
            g[self.a] = 1.0
            r = (z[self.a], g)
        elif op == "add":
            va, ga = self.a.ev(z, key)
            vb, gb = self.b.ev(z, key)
            r = (va + vb, ga + gb)
        elif op == "sub":
            va, ga = self.a.ev(z, key)
            vb, gb = self.b.ev(z, key)
            r = (va - vb, ga - gb)
        elif op == "mul":
            va, ga = self.a.ev(z, key)
            vb, gb = self.b.ev(z, key)
            r = (va * vb, ga * vb + gb * va)
        elif op == "pow":
            va, ga = self.a.ev(z, key)
            r = (va ** self.b, (self.b * va ** (self.b - 1)) * ga)
        else:  # pragma: no cover
            raise ValueError(op)
        self._cache_key, self._cache_val = key, r
        return r


class Constraint:
    """lhs (==|>=) 0 on a scalar node."""

    def __init__(self, node, kind):
        self.node, self.kind = node, kind  # kind in {"eq", "ge"}


class MX:
    """2-D matrix of scalar nodes with CasADi-like shape semantics (always 2-D, column vectors)."""

    __array_ufunc__ = None

    def __init__(self, a):
        a = np.asarray(a, dtype=object)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        self.a = a

    @property
    def shape(self):
        return self.a.shape

    @property
    def T(self):
        return MX(self.a.T)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx, slice(None)) if self.a.shape[1] > 1 else (idx, 0)
        r, c = idx
        rr = np.arange(self.a.shape[0])[r]
        cc = np.arange(self.a.shape[1])[c]
        rr = np.atleast_1d(rr)
        cc = np.atleast_1d(cc)
        return MX(self.a[np.ix_(rr, cc)])

    @staticmethod
    def _coerce(o):
        if isinstance(o, MX):
            return o.a
        if isinstance(o, Node):
            return np.array([[o]], dtype=object)
        o = np.asarray(o, dtype=float)
        if o.ndim == 0:
            return o.reshape(1, 1)
        if o.ndim == 1:
            return o.reshape(-1, 1)
        return o

    def _bin(self, o, fn):
        a, b = self.a, MX._coerce(o)
        if a.shape != b.shape:
            if a.shape == (1, 1) or b.shape == (1, 1):
                a, b = np.broadcast_arrays(a, b)
            else:
                raise ValueError("shape mismatch %s vs %s" % (a.shape, b.shape))
        out = np.empty(a.shape, dtype=object)
        for i in range(a.shape[0]):
            for j in range(a.shape[1]):
                out[i, j] = fn(a[i, j], b[i, j])
        return MX(out)

    @staticmethod
    def _n(v):
        return v if isinstance(v, Node) else Node("const", float(v))

    def __add__(self, o):
        return self._bin(o, lambda p, q: MX._n(p) + MX._n(q))

    __radd__ = __add__

    def __sub__(self, o):
        return self._bin(o, lambda p, q: MX._n(p) - MX._n(q))

    def __rsub__(self, o):
        return self._bin(o, lambda p, q: MX._n(q) - MX._n(p))

    def __mul__(self, o):
        return self._bin(o, lambda p, q: MX._n(p) * MX._n(q))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._bin(o, lambda p, q: MX._n(p) * (1.0 / float(q)))

    def __neg__(self):
        return self * -1.0

    def __pow__(self, p):
        out = np.empty(self.a.shape, dtype=object)
        for i in range(self.a.shape[0]):
            for j in range(self.a.shape[1]):
                out[i, j] = MX._n(self.a[i, j]) ** p
        return MX(out)

    def _cmp(self, o, kind, sign):
        d = (self - o) if sign > 0 else (MX(MX._coerce(o)) - self)
        return [Constraint(MX._n(v), kind) for v in d.a.reshape(-1)]

    def __eq__(self, o):  # noqa: D105
        return self._cmp(o, "eq", +1)

    def __ge__(self, o):
        return self._cmp(o, "ge", +1)

    def __le__(self, o):
        return self._cmp(o, "ge", -1)

    # CasADi's Opti treats strict inequalities as non-strict (overtake_path_planner.py:280,297 use < and >)
    def __gt__(self, o):
        return self._cmp(o, "ge", +1)

    def __lt__(self, o):
        return self._cmp(o, "ge", -1)

    __hash__ = None


def mtimes(a, b):
    A, B = MX._coerce(a), MX._coerce(b)
    assert A.shape[1] == B.shape[0], (A.shape, B.shape)
    out = np.empty((A.shape[0], B.shape[1]), dtype=object)
    for i in range(A.shape[0]):
        for j in range(B.shape[1]):
            acc = None
            for k in range(A.shape[1]):
                p, q = A[i, k], B[k, j]
                if not isinstance(p, Node) and not isinstance(q, Node):
                    t = Node("const", float(p) * float(q))
                elif not isinstance(p, Node):
                    if float(p) == 0.0:
                        continue
                    t = q * float(p)
                elif not isinstance(q, Node):
                    if float(q) == 0.0:
                        continue
                    t = p * float(q)
                else:
                    t = p * q
                acc = t if acc is None else acc + t
            out[i, j] = acc if acc is not None else Node("const", 0.0)
    return MX(out)


def vertcat(*xs):
    return MX(np.vstack([MX._coerce(x) for x in xs]))


def horzcat(*xs):
    return MX(np.hstack([MX._coerce(x) for x in xs]))


class _Sol:
    def __init__(self, z, key):
        self.z, self.key = z, key

    def value(self, e):
        if isinstance(e, MX):
            out = np.empty(e.a.shape)
            for i in range(e.a.shape[0]):
                for j in range(e.a.shape[1]):
                    out[i, j] = MX._n(e.a[i, j]).ev(self.z, self.key)[0]
            if out.shape == (1, 1):
                return float(out[0, 0])
            if out.shape[1] == 1:
                return out[:, 0].copy()
            return out
        if isinstance(e, Node):
            return float(e.ev(self.z, self.key)[0])
        return e


_EVAL_KEY = [0]


def _next_key():
    _EVAL_KEY[0] += 1
    return _EVAL_KEY[0]


class Opti:
    """Recording Opti.  ``Opti.hook`` (if set) is called with the finished record dict."""

    hook = None
    solver_fn = None  # set by make_golden: (opti) -> (z, info); raises RuntimeError when infeasible

    def __init__(self):
        self.nvar = 0
        self.vars = []  # (MX, offset, shape)
        self.cons = []
        self.cost = None
        self.init = {}
        self.debug = self
        self._last = None

    def variable(self, r=1, c=1):
        a = np.empty((r, c), dtype=object)
        # CasADi numbers a matrix variable column-major
        for j in range(c):
            for i in range(r):
                a[i, j] = Node("var", self.nvar)
                self.nvar += 1
        m = MX(a)
        self.vars.append((m, self.nvar - r * c, (r, c)))
        return m

    def subject_to(self, cons):
        if isinstance(cons, Constraint):
            cons = [cons]
        self.cons.extend(cons)

    def minimize(self, cost):
        self.cost = Node.lift(cost)

    def set_initial(self, var, val):
        a = MX._coerce(var)
        v = np.broadcast_to(np.asarray(val, dtype=float).reshape(-1, 1) if np.ndim(val) == 1 else np.asarray(val, dtype=float), a.shape)
        for i in range(a.shape[0]):
            for j in range(a.shape[1]):
                n = a[i, j]
                assert isinstance(n, Node) and n.op == "var"
                self.init[n.a] = float(v[i, j])

    def solver(self, name, opts=None):
        self.solver_name, self.solver_opts = name, opts

    # evaluation helpers used by nlp_solve ---------------------------------------------------
    def z0(self):
        z = np.zeros(self.nvar)
        for k, v in self.init.items():
            z[k] = v
        return z

    def eval_all(self, z):
        key = _next_key()
        f, gf = self.cost.ev(z, key)
        ce, Je, ci, Ji = [], [], [], []
        for c in self.cons:
            v, g = c.node.ev(z, key)
            g = np.zeros(self.nvar) if np.isscalar(g) else g
            if c.kind == "eq":
                ce.append(v)
                Je.append(g)
            else:
                ci.append(v)
                Ji.append(g)
        n = self.nvar
        return (
            f,
            gf if not np.isscalar(gf) else np.zeros(n),
            np.array(ce),
            np.array(Je).reshape(len(ce), n),
            np.array(ci),
            np.array(Ji).reshape(len(ci), n),
        )

    def solve(self):
        z, info = Opti.solver_fn(self)
        self._last = _Sol(z, _next_key())
        self.info = info
        if Opti.hook is not None:
            Opti.hook(self, z, info)
        if not info["success"]:
            raise RuntimeError("shim: solver did not succeed: %s" % info.get("reason"))
        return self._last

    def value(self, e):  # opti.debug.value(...)
        return self._last.value(e)
inf = float("inf")  # casadi exports it; overtake_path_planner.py:316 relies on `from casadi import *`
