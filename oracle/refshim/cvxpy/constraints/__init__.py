"""ORACLE - test infrastructure only (part of the cvxpy stand-in, see ../__init__.py)."""


class Constraint:
    def __init__(self, expr):
        self.expr = expr

    def violation(self):
        raise NotImplementedError


class Zero(Constraint):
    """expr == 0"""

    def canon(self, ctx):
        ctx.eq.append(self.expr.canon(ctx, 0))

    def violation(self):
        import numpy as np
        return float(np.max(np.abs(self.expr.numeric()), initial=0.0))


class NonPos(Constraint):
    """expr <= 0 (expr convex)"""

    def canon(self, ctx):
        ctx.ineq.append(self.expr.canon(ctx, 1))

    def violation(self):
        import numpy as np
        return float(np.max(np.maximum(self.expr.numeric(), 0.0), initial=0.0))


from . import zero, nonpos  # noqa: E402,F401
