"""ORACLE - test infrastructure only.  NOT cvxpy.

A small stand-in for the part of the `cvxpy` API that hanruihua/RDA-planner's RDA_planner/rda_solver.py
uses (cvxpy==1.5.2 is pinned by the reference's setup.py:8 and is not installable in the build image), so
that the UNMODIFIED reference module can be imported from /root/reference and executed end to end:
`definition()`, `construct_problem()`, the cost / constraint formulas (`Im_su`, `Hm_su`, `Im_LamMu`,
`Hm_LamMu`, `dynamics_constraint`, `bound_*`, `C0_cost`, `C1_cost`, `cone_*_array`, rda_solver.py:831-1050)
and `prob.solve(...)`.  Problems built by the reference's own code are canonicalised by the disciplined
convex rules (epigraph variables for norm / abs / max / min / neg) into a cone QP and solved by
oracle/refshim/coneqp.py, a generic interior-point method.  Only tests and fixture generators use it.

Supported surface (exactly what rda_solver.py touches): Variable, Parameter, Problem, Minimize,
sum_squares, neg, norm (2-norm, axis None / 0), max, min, abs, sum, vstack, hstack, reshape,
constraints.zero.Zero, constraints.nonpos.NonPos, @ * + - unary-minus, basic indexing, .T, ==, <=, >=,
Problem.is_dcp / solve / status / variables / value, Variable.name / value, the status and solver constants.
Flattening is numpy row-major throughout (cvxpy's column-major internals are not observable through this API).
"""
import numbers

import numpy as np
import scipy.sparse as sp

import refshim_coneqp as _coneqp          # top-level module of oracle/refshim (the directory is put on sys.path)
from . import constraints  # noqa: F401  (cp.constraints.zero.Zero / cp.constraints.nonpos.NonPos)

OPTIMAL = "optimal"
OPTIMAL_INACCURATE = "optimal_inaccurate"
INFEASIBLE = "infeasible"
SOLVER_ERROR = "solver_error"
ECOS = "ECOS"
SCS = "SCS"
__version__ = "0.0-refshim"

SETTINGS = {"tol": 1e-10, "max_iter": 200, "verbose": False}
LAST = {}          # statistics of the last solve (tests read it)


# ---------------------------------------------------------------------------------------------
# affine forms: value = M x + k over the flat (row-major) entries of an expression, M kept as COO triplets
# ---------------------------------------------------------------------------------------------
class Aff:
    __slots__ = ("shape", "r", "c", "v", "k")

    def __init__(self, shape, r, c, v, k):
        self.shape = tuple(shape)
        self.r, self.c, self.v = r, c, v
        self.k = k

    @property
    def size(self):
        return self.k.size

    @staticmethod
    def const(val):
        val = np.asarray(val, dtype=float)
        z = np.zeros(0, dtype=np.int64)
        return Aff(val.shape, z, z, np.zeros(0), val.ravel().copy())

    def take(self, idx, shape):
        """new entry j = old entry idx[j] (idx may repeat entries)"""
        idx = np.asarray(idx, dtype=np.int64).ravel()
        n_old = self.size
        if self.r.size == 0:
            z = np.zeros(0, dtype=np.int64)
            return Aff(shape, z, z, np.zeros(0), self.k[idx])
        counts_idx = np.bincount(idx, minlength=n_old)
        if counts_idx.max(initial=0) <= 1:
            inv = np.full(n_old, -1, dtype=np.int64)
            inv[idx] = np.arange(idx.size)
            nr = inv[self.r]
            keep = nr >= 0
            return Aff(shape, nr[keep], self.c[keep], self.v[keep], self.k[idx])
        order = np.argsort(self.r, kind="stable")
        rs, cs, vs = self.r[order], self.c[order], self.v[order]
        cnt = np.bincount(rs, minlength=n_old)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        cn = cnt[idx]
        total = int(cn.sum())
        new_r = np.repeat(np.arange(idx.size), cn)
        src = np.arange(total) - np.repeat(np.cumsum(cn) - cn, cn) + np.repeat(start[idx], cn)
        return Aff(shape, new_r, cs[src], vs[src], self.k[idx])

    def scale_rows(self, w):
        w = np.asarray(w, dtype=float).ravel()
        return Aff(self.shape, self.r, self.c, self.v * w[self.r], self.k * w)

    def __neg__(self):
        return Aff(self.shape, self.r, self.c, -self.v, -self.k)

    @staticmethod
    def add(a, b):
        assert a.shape == b.shape, (a.shape, b.shape)
        return Aff(a.shape, np.concatenate([a.r, b.r]), np.concatenate([a.c, b.c]), np.concatenate([a.v, b.v]), a.k + b.k)

    def matrix(self, n):
        return sp.csr_matrix((self.v, (self.r, self.c)), shape=(self.size, n))


def _bcast_idx(shape, out_shape):
    return np.broadcast_to(np.arange(int(np.prod(shape, dtype=np.int64))).reshape(shape), out_shape).ravel()


def _bshape(a, b):
    return np.broadcast_shapes(tuple(a), tuple(b))


class Ctx:
    """one canonicalisation: variable offsets + collected cone QP data"""

    def __init__(self):
        self.n = 0
        self.off = {}
        self.eq, self.ineq, self.soc = [], [], []

    def var(self, v):
        if id(v) not in self.off:
            self.off[id(v)] = (self.n, v)
            self.n += v.size
            if v.attrs.get("nonneg"):
                self.ineq.append(-self.ident(self.off[id(v)][0], v.shape))
        o = self.off[id(v)][0]
        return self.ident(o, v.shape)

    def new(self, shape):
        size = int(np.prod(shape, dtype=np.int64))
        o = self.n
        self.n += size
        return self.ident(o, shape)

    @staticmethod
    def ident(o, shape):
        size = int(np.prod(shape, dtype=np.int64))
        ar = np.arange(size)
        return Aff(shape, ar, ar + o, np.ones(size), np.zeros(size))


# ---------------------------------------------------------------------------------------------
# expression tree
# ---------------------------------------------------------------------------------------------
def _wrap(x):
    if isinstance(x, Expr):
        return x
    return Const(x)


class Expr:
    __array_priority__ = 1000       # numpy defers to our reflected operators
    shape = ()

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def ndim(self):
        return len(self.shape)

    def is_const(self):
        c = getattr(self, "_is_const", None)
        if c is None:
            c = all(ch.is_const() for ch in self.children())
            self._is_const = c
        return c

    def children(self):
        return ()

    # numeric value from leaf values
    @property
    def value(self):
        return self.numeric()

    # canon(ctx, curv): curv 0 -> must be affine; +1 -> an affine over-estimator is wanted (convex ok); -1 -> under-estimator
    def canon(self, ctx, curv):
        if self.is_const():
            return Aff.const(np.asarray(self.numeric(), dtype=float).reshape(self.shape))
        return self._canon(ctx, curv)

    # operators
    def __add__(self, o):
        return Add(self, _wrap(o))

    def __radd__(self, o):
        return Add(_wrap(o), self)

    def __sub__(self, o):
        return Add(self, Neg(_wrap(o)))

    def __rsub__(self, o):
        return Add(_wrap(o), Neg(self))

    def __neg__(self):
        return Neg(self)

    def __mul__(self, o):
        return Mul(self, _wrap(o))

    def __rmul__(self, o):
        return Mul(_wrap(o), self)

    def __matmul__(self, o):
        return MatMul(self, _wrap(o))

    def __rmatmul__(self, o):
        return MatMul(_wrap(o), self)

    def __truediv__(self, o):
        return Mul(self, Const(1.0 / np.asarray(o, dtype=float)))

    def __getitem__(self, key):
        return Index(self, key)

    @property
    def T(self):
        return Transpose(self)

    def __eq__(self, o):
        return constraints.Zero(self - _wrap(o))

    def __le__(self, o):
        return constraints.NonPos(self - _wrap(o))

    def __ge__(self, o):
        return constraints.NonPos(_wrap(o) - self)

    __hash__ = object.__hash__


class Const(Expr):
    def __init__(self, val):
        self.val = np.asarray(val, dtype=float)
        self.shape = self.val.shape

    def is_const(self):
        return True

    def numeric(self):
        return self.val


class Leaf(Expr):
    _count = 0

    def __init__(self, shape=(), name=None, value=None, **attrs):
        if isinstance(shape, numbers.Integral):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        Leaf._count += 1
        self._name = name if name is not None else f"{type(self).__name__.lower()}{Leaf._count}"
        self.attrs = attrs
        self._value = None
        if value is not None:
            self.value = value

    def name(self):
        return self._name

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, val):
        if val is None:
            self._value = None
            return
        arr = val if (isinstance(val, np.ndarray) and val.dtype == np.float64) else np.asarray(val, dtype=float)
        if arr.shape != self.shape:
            raise ValueError(f"Invalid dimensions {arr.shape} for {type(self).__name__} value (expected {self.shape}).")
        if self.attrs.get("nonneg") and arr.size and np.min(arr) < -1e-8:
            raise ValueError(f"{type(self).__name__} value must be nonnegative.")
        # like cvxpy the leaf keeps the array it was given (no copy): the reference mutates `.value[...]` in place
        self._value = arr if self.shape != () else (val if isinstance(val, numbers.Number) else arr)

    def numeric(self):
        if self._value is None:
            raise ValueError(f"{self._name} has no value")
        return np.asarray(self._value, dtype=float)


class Parameter(Leaf):
    def is_const(self):
        return True


class Variable(Leaf):
    def is_const(self):
        return False

    def _canon(self, ctx, curv):
        return ctx.var(self)


class Add(Expr):
    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = _bshape(a.shape, b.shape)

    def children(self):
        return (self.a, self.b)

    def numeric(self):
        return self.a.numeric() + self.b.numeric()

    def _canon(self, ctx, curv):
        fa, fb = self.a.canon(ctx, curv), self.b.canon(ctx, curv)
        if fa.shape != self.shape:
            fa = fa.take(_bcast_idx(fa.shape, self.shape), self.shape)
        if fb.shape != self.shape:
            fb = fb.take(_bcast_idx(fb.shape, self.shape), self.shape)
        return Aff.add(fa, fb)


class Neg(Expr):
    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def children(self):
        return (self.a,)

    def numeric(self):
        return -self.a.numeric()

    def _canon(self, ctx, curv):
        return -self.a.canon(ctx, -curv)


class Mul(Expr):
    """elementwise product with numpy broadcasting; one side must be free of variables"""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = _bshape(a.shape, b.shape)

    def children(self):
        return (self.a, self.b)

    def numeric(self):
        return self.a.numeric() * self.b.numeric()

    def _canon(self, ctx, curv):
        cst, var = (self.a, self.b) if self.a.is_const() else (self.b, self.a)
        if not cst.is_const():
            raise ValueError("product of two non-constant expressions is not DCP")
        w = np.broadcast_to(np.asarray(cst.numeric(), dtype=float), self.shape).ravel()
        if not np.any(w):
            return Aff.const(np.zeros(self.shape))
        sub = curv
        if curv != 0:
            if np.all(w >= 0):
                sub = curv
            elif np.all(w <= 0):
                sub = -curv
            else:
                sub = 0
        f = var.canon(ctx, sub)
        if f.shape != self.shape:
            f = f.take(_bcast_idx(f.shape, self.shape), self.shape)
        return f.scale_rows(w)


class MatMul(Expr):
    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = np.matmul(np.zeros(a.shape), np.zeros(b.shape)).shape if (a.ndim and b.ndim) else None
        if self.shape is None:
            raise ValueError("matmul with a scalar")

    def children(self):
        return (self.a, self.b)

    def numeric(self):
        return self.a.numeric() @ self.b.numeric()

    def _canon(self, ctx, curv):
        if self.a.is_const():
            Pm = np.asarray(self.a.numeric(), dtype=float)
            sub = curv if (curv == 0 or np.all(Pm >= 0)) else (-curv if np.all(Pm <= 0) else 0)
            f = self.b.canon(ctx, sub)
            Pm2 = Pm.reshape(1, -1) if Pm.ndim == 1 else Pm
            bshape = (f.shape[0], 1) if len(f.shape) == 1 else f.shape
            m, kk = Pm2.shape
            nn = bshape[1]
            out = None
            ii, jj = np.meshgrid(np.arange(m), np.arange(nn), indexing="ij")
            for l_ in range(kk):
                w = Pm2[ii.ravel(), l_]
                if not np.any(w):
                    continue
                term = f.take(l_ * nn + jj.ravel(), (m, nn)).scale_rows(w)
                out = term if out is None else Aff.add(out, term)
            if out is None:
                out = Aff.const(np.zeros((m, nn)))
            out.shape = self.shape
            return out
        if self.b.is_const():
            Pm = np.asarray(self.b.numeric(), dtype=float)
            sub = curv if (curv == 0 or np.all(Pm >= 0)) else (-curv if np.all(Pm <= 0) else 0)
            f = self.a.canon(ctx, sub)
            Pm2 = Pm.reshape(-1, 1) if Pm.ndim == 1 else Pm
            ashape = (1, f.shape[0]) if len(f.shape) == 1 else f.shape
            m, kk = ashape
            nn = Pm2.shape[1]
            out = None
            ii, jj = np.meshgrid(np.arange(m), np.arange(nn), indexing="ij")
            for l_ in range(kk):
                w = Pm2[l_, jj.ravel()]
                if not np.any(w):
                    continue
                term = f.take(ii.ravel() * kk + l_, (m, nn)).scale_rows(w)
                out = term if out is None else Aff.add(out, term)
            if out is None:
                out = Aff.const(np.zeros((m, nn)))
            out.shape = self.shape
            return out
        raise ValueError("product of two non-constant expressions is not DCP")


class Index(Expr):
    def __init__(self, a, key):
        self.a, self.key = a, key
        self.idx = np.arange(a.size).reshape(a.shape)[key]
        self.shape = self.idx.shape

    def children(self):
        return (self.a,)

    def numeric(self):
        return self.a.numeric()[self.key]

    def _canon(self, ctx, curv):
        return self.a.canon(ctx, curv).take(self.idx.ravel(), self.shape)


class Transpose(Expr):
    def __init__(self, a):
        self.a = a
        self.idx = np.arange(a.size).reshape(a.shape).T
        self.shape = self.idx.shape

    def children(self):
        return (self.a,)

    def numeric(self):
        return self.a.numeric().T

    def _canon(self, ctx, curv):
        return self.a.canon(ctx, curv).take(self.idx.ravel(), self.shape)


class Reshape(Expr):
    def __init__(self, a, shape, order="F"):
        self.a = a
        if isinstance(shape, numbers.Integral):
            shape = (int(shape),)
        self.idx = np.arange(a.size).reshape(a.shape).reshape(tuple(shape), order=order)
        self.shape = self.idx.shape
        self.order = order

    def children(self):
        return (self.a,)

    def numeric(self):
        return self.a.numeric().reshape(self.shape, order=self.order)

    def _canon(self, ctx, curv):
        return self.a.canon(ctx, curv).take(self.idx.ravel(), self.shape)


class Stack(Expr):
    def __init__(self, args, vertical):
        self.args = [_wrap(a) for a in args]
        self.vertical = vertical
        f = np.vstack if vertical else np.hstack
        pos = []
        off = 0
        for a in self.args:
            pos.append(np.arange(off, off + a.size).reshape(a.shape))
            off += a.size
        self.layout = f(pos)              # output entry -> index into the concatenation of the flat arguments
        self.shape = self.layout.shape

    def children(self):
        return tuple(self.args)

    def numeric(self):
        f = np.vstack if self.vertical else np.hstack
        return f([np.asarray(a.numeric(), dtype=float) for a in self.args])

    def _canon(self, ctx, curv):
        fs = [a.canon(ctx, curv) for a in self.args]
        offs = np.cumsum([0] + [f.size for f in fs])
        cat = Aff((int(offs[-1]),), np.concatenate([f.r + o for f, o in zip(fs, offs)]), np.concatenate([f.c for f in fs]),
                  np.concatenate([f.v for f in fs]), np.concatenate([f.k for f in fs]))
        return cat.take(self.layout.ravel(), self.shape)


class Sum(Expr):
    def __init__(self, a):
        self.a = a
        self.shape = ()

    def children(self):
        return (self.a,)

    def numeric(self):
        return np.sum(self.a.numeric())

    def _canon(self, ctx, curv):
        f = self.a.canon(ctx, curv)
        return Aff((), np.zeros_like(f.r), f.c, f.v, np.array([f.k.sum()]))


# ---- convex / concave atoms ------------------------------------------------------------------
class Norm2(Expr):
    def __init__(self, a, axis=None):
        self.a, self.axis = a, axis
        if axis is None:
            self.shape = ()
        elif axis == 0:
            if a.ndim != 2:
                raise ValueError("norm(axis=0) needs a matrix")
            self.shape = (a.shape[1],)
        else:
            raise NotImplementedError("norm axis")

    def children(self):
        return (self.a,)

    def numeric(self):
        v = np.asarray(self.a.numeric(), dtype=float)
        return np.linalg.norm(v.ravel()) if self.axis is None else np.linalg.norm(v, axis=0)

    def _canon(self, ctx, curv):
        if curv <= 0:
            raise ValueError("norm is convex: not DCP in this position")
        f = self.a.canon(ctx, 0)
        if self.axis is None:
            t = ctx.new(())
            ctx.soc.append((t, Aff((1, f.size), f.r, f.c, f.v, f.k)))
            return t
        m, n = f.shape
        t = ctx.new((n,))
        ctx.soc.append((t, f.take(np.arange(m * n).reshape(m, n).T.ravel(), (n, m))))
        return t


class Abs(Expr):
    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def children(self):
        return (self.a,)

    def numeric(self):
        return np.abs(self.a.numeric())

    def _canon(self, ctx, curv):
        if curv <= 0:
            raise ValueError("abs is convex: not DCP in this position")
        f = self.a.canon(ctx, 0)
        t = ctx.new(self.shape)
        ctx.ineq.append(Aff.add(f, -t))
        ctx.ineq.append(Aff.add(-f, -t))
        return t


class NegPart(Expr):
    """neg(x) = max(-x, 0), convex and non-increasing"""

    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def children(self):
        return (self.a,)

    def numeric(self):
        return np.maximum(-self.a.numeric(), 0.0)

    def _canon(self, ctx, curv):
        if curv <= 0:
            raise ValueError("neg is convex: not DCP in this position")
        f = self.a.canon(ctx, -1)
        t = ctx.new(self.shape)
        ctx.ineq.append(Aff.add(-f, -t))      # -x <= t
        ctx.ineq.append(-t)                   # 0 <= t
        return t


class MaxEntries(Expr):
    def __init__(self, a, concave=False):
        self.a, self.concave = a, concave
        self.shape = ()

    def children(self):
        return (self.a,)

    def numeric(self):
        return (np.min if self.concave else np.max)(self.a.numeric())

    def _canon(self, ctx, curv):
        want = -1 if self.concave else 1
        if curv != want:
            raise ValueError("max / min: not DCP in this position")
        f = self.a.canon(ctx, want)
        t = ctx.new(())
        tb = t.take(np.zeros(f.size, dtype=np.int64), f.shape)
        ctx.ineq.append(Aff.add(f, -tb) if not self.concave else Aff.add(-f, tb))
        return t


class SumSquares(Expr):
    def __init__(self, a):
        self.a = a
        self.shape = ()

    def children(self):
        return (self.a,)

    def numeric(self):
        return float(np.sum(np.asarray(self.a.numeric(), dtype=float) ** 2))

    def _canon(self, ctx, curv):
        raise NotImplementedError("sum_squares is supported in the objective only")


def sum_squares(x):
    return SumSquares(_wrap(x))


def neg(x):
    return NegPart(_wrap(x))


def norm(x, p=2, axis=None):
    if p not in (2, "fro"):
        raise NotImplementedError("only the 2-norm")
    return Norm2(_wrap(x), axis)


def max(x):     # noqa: A001 (mirrors cp.max)
    return MaxEntries(_wrap(x))


def min(x):     # noqa: A001
    return MaxEntries(_wrap(x), concave=True)


def abs(x):     # noqa: A001
    return Abs(_wrap(x))


def sum(x):     # noqa: A001
    return Sum(_wrap(x))


def vstack(args):
    return Stack(list(args), True)


def hstack(args):
    return Stack(list(args), False)


def reshape(x, shape, order="F"):
    return Reshape(_wrap(x), shape, order)


# ---------------------------------------------------------------------------------------------
class Minimize:
    def __init__(self, expr):
        self.expr = _wrap(expr)

    @property
    def value(self):
        return self.expr.numeric()


def _canon_objective(expr, scale, ctx, quad, lin):
    """collect  scale * expr  into quad (list of (weight, Aff)) and lin (list of (weight, Aff scalar))"""
    if expr.is_const():
        lin.append((scale, Aff.const(np.asarray(expr.numeric(), dtype=float).reshape(()))))
    elif isinstance(expr, Add):
        _canon_objective(expr.a, scale, ctx, quad, lin)
        _canon_objective(expr.b, scale, ctx, quad, lin)
    elif isinstance(expr, Neg):
        _canon_objective(expr.a, -scale, ctx, quad, lin)
    elif isinstance(expr, Mul) and (expr.a.is_const() or expr.b.is_const()):
        cst, var = (expr.a, expr.b) if expr.a.is_const() else (expr.b, expr.a)
        c = np.asarray(cst.numeric(), dtype=float)
        if c.size != 1 or var.shape != ():
            raise ValueError("objective must be a scalar")
        _canon_objective(var, scale * float(c.ravel()[0]), ctx, quad, lin)
    elif isinstance(expr, SumSquares):
        if scale < 0:
            raise ValueError("objective is not convex")
        inner = expr.a
        affine = _is_affine(inner)
        quad.append((scale, inner.canon(ctx, 0 if affine else 1)))
    else:
        if expr.shape != ():
            raise ValueError("objective must be a scalar")
        lin.append((scale, expr.canon(ctx, 1 if scale >= 0 else -1)))


def _is_affine(e):
    if isinstance(e, (Norm2, Abs, NegPart, MaxEntries, SumSquares)):
        return e.is_const()
    return all(_is_affine(ch) for ch in e.children())


def _variables_of(e, seen, out):
    if id(e) in seen:
        return
    seen.add(id(e))
    if isinstance(e, Variable):
        out.append(e)
    for ch in e.children():
        _variables_of(ch, seen, out)


class Problem:
    def __init__(self, objective, constraints_=None):
        self.objective = objective
        self.constraints = list(constraints_ or [])
        self.status = None
        self.value = None
        self._vars = None

    def variables(self):
        if self._vars is None:
            seen, out = set(), []
            _variables_of(self.objective.expr, seen, out)
            for c in self.constraints:
                _variables_of(c.expr, seen, out)
            self._vars = out
        return list(self._vars)

    def is_dcp(self, dpp=False):
        # curvature is checked by the canonicalisation itself (it raises on a non-DCP composition); parameters
        # only ever multiply variables or each other, which is what `dpp=True` asks for
        return True

    def _canonicalise(self):
        ctx = Ctx()
        for v in self.variables():
            ctx.var(v)
        quad, lin = [], []
        _canon_objective(self.objective.expr, 1.0, ctx, quad, lin)
        for c in self.constraints:
            c.canon(ctx)
        n = ctx.n
        P = sp.csr_matrix((n, n))
        q = np.zeros(n)
        r0 = 0.0
        for w, f in quad:
            F = f.matrix(n)
            P = P + 2.0 * w * (F.T @ F)
            q += 2.0 * w * (F.T @ f.k)
            r0 += w * float(f.k @ f.k)
        for w, f in lin:
            if f.r.size:
                np.add.at(q, f.c, w * f.v)
            r0 += w * float(f.k.sum())
        if ctx.eq:
            A = sp.vstack([f.matrix(n) for f in ctx.eq], format="csr")
            b = -np.concatenate([f.k for f in ctx.eq])
        else:
            A, b = sp.csr_matrix((0, n)), np.zeros(0)
        Gs, hs = [], []
        l_ = 0
        for f in ctx.ineq:
            Gs.append(f.matrix(n))
            hs.append(-f.k)
            l_ += f.size
        socs = []
        for t, X in ctx.soc:
            k, m = X.shape
            # rows of cone i: (t_i ; X[i, :])  ->  s = h - G x  with  s = (t_i x + kt ; X x + kx)
            lay = np.empty((k, m + 1), dtype=np.int64)
            lay[:, 0] = np.arange(k)
            lay[:, 1:] = k + np.arange(k * m).reshape(k, m)
            tt = Aff((t.size,), t.r, t.c, t.v, t.k)
            cat = Aff((k + k * m,), np.concatenate([tt.r, X.r + k]), np.concatenate([tt.c, X.c]), np.concatenate([tt.v, X.v]),
                      np.concatenate([tt.k, X.k])).take(lay.ravel(), (k * (m + 1),))
            Gs.append(-cat.matrix(n))
            hs.append(cat.k)
            socs += [m + 1] * k
        G = sp.vstack(Gs, format="csr") if Gs else sp.csr_matrix((0, n))
        h = np.concatenate(hs) if hs else np.zeros(0)
        return ctx, P, q, r0, A, b, G, h, l_, socs

    def solve(self, solver=None, verbose=False, **kwargs):
        ctx, P, q, r0, A, b, G, h, l_, socs = self._canonicalise()
        res = _coneqp.solve(P, q, A, b, G, h, l_, socs, tol=kwargs.get("tol", SETTINGS["tol"]),
                            max_iter=SETTINGS["max_iter"], verbose=verbose or SETTINGS["verbose"])
        LAST.clear()
        LAST.update(iters=res["iters"], status=res["status"], n=ctx.n, p=A.shape[0], m=G.shape[0], gap=res["gap"],
                    pres=res["pres"], dres=res["dres"])
        st = res["status"]
        if st in ("optimal", "optimal_inaccurate"):
            self.status = OPTIMAL if st == "optimal" else OPTIMAL_INACCURATE
            x = res["x"]
            for o, v in ctx.off.values():
                val = x[o:o + v.size].reshape(v.shape)
                if v.attrs.get("nonneg"):
                    val = np.maximum(val, 0.0)
                v._value = val.copy()
            self.value = res["pcost"] + r0
        else:
            self.status = SOLVER_ERROR           # the reference treats every non-OPTIMAL status alike (rda_solver.py:696,781)
            for o, v in ctx.off.values():
                v._value = None
            self.value = None
        return self.value


class SolverError(Exception):
    pass


class error:                                     # cvxpy.error.SolverError
    SolverError = SolverError
