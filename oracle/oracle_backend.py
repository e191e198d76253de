"""ORACLE backend (test infrastructure only): lets tests run the host-side `RDA_solver` / `MPC`
classes on top of oracle/librda_oracle.so instead of the HIP library.

    solver = RDA_solver(..., _backend=oracle_backend)
"""
import ctypes as C

from rda_planner_amd._capi import CApi, Cfg, c_double_p, c_int_p
from rda_planner_amd.rda_solver import _Backend
from . import oracle_lib

_api = None


def api():
    global _api
    if _api is None:
        lib = oracle_lib.load()
        _api = CApi(lib, "orc")
        lib.orc_lammuz_one.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, c_double_p, C.c_double,
                                       c_double_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        lib.orc_lammuz_one.restype = C.c_int
        lib.orc_su_solve.argtypes = [C.POINTER(Cfg)] + [c_double_p] * 3 + [C.c_double] + [c_double_p] * 7 + [c_int_p]
        lib.orc_su_solve.restype = C.c_int
        lib.orc_set_threads.argtypes = [C.c_int]
    return _api


def oracle_backend(cfg, G, h):
    return _Backend(api(), cfg, G, h)
