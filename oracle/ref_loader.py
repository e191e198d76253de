"""ORACLE - test infrastructure only.

Imports the UNMODIFIED reference (`/root/reference/RDA_planner/{rda_solver,mpc}.py`) in this process.  The
reference needs `cvxpy` (1.5.2, setup.py:8) and `pathos`, neither installable here; `oracle/refshim/` provides
stand-ins for exactly the names it uses (see oracle/refshim/cvxpy/__init__.py), put on sys.path *only* when
the real packages are absent.  /root/reference exists in the build container only - everything that needs it
(tests marked `needs_reference`, tests/golden/make_ref_golden.py) is skipped / not run on the GPU box, which
sees the committed fixtures instead.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("RDA_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "RDA_planner", "rda_solver.py"))


def load():
    """-> (rda_solver module, mpc module, backend) with backend 'cvxpy' (the real one) or 'refshim'"""
    if not available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    backend = "cvxpy"
    try:
        import cvxpy as cp
        if getattr(cp, "__version__", "").endswith("refshim"):
            backend = "refshim"
    except ImportError:
        backend = "refshim"
        if _SHIM not in sys.path:
            sys.path.insert(0, _SHIM)
    try:
        import pathos.multiprocessing  # noqa: F401
    except ImportError:
        if _SHIM not in sys.path:
            sys.path.insert(0, _SHIM)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    rs = importlib.import_module("RDA_planner.rda_solver")
    mp = importlib.import_module("RDA_planner.mpc")
    return rs, mp, backend
