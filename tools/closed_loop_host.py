"""ctypes view of tools/libclosed_loop_host.so (tools/closed_loop_host.c: the closed MPC loop around the C-ABI written in C).
Used by bench.py (the timed closed loops) and tests/test_gpu_track.py."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libclosed_loop_host.so")


def build():
    subprocess.check_call(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-o", SO, os.path.join(HERE, "closed_loop_host.c"), "-lm"])
    return SO


class Api(C.Structure):        # struct closed_loop_api
    _fields_ = [(n_, C.c_void_p) for n_ in ("step_tracked", "tracked_begin", "upload_scene_async", "tracked_finish", "scene_resort")]


class Scene(C.Structure):      # struct closed_loop_scene
    _fields_ = [("n", C.c_int32), ("maxv", C.c_int32), ("order", C.c_int32), ("moving", C.c_int32), ("kind", C.POINTER(C.c_int32)),
                ("nvert", C.POINTER(C.c_int32)), ("geom", C.POINTER(C.c_double)), ("geom0", C.POINTER(C.c_double)), ("vel", C.POINTER(C.c_double))]


class FleetApi(C.Structure):   # struct closed_loop_fleet_api
    _fields_ = [(n_, C.c_void_p) for n_ in ("fleet_step_tracked", "scene_resort", "fleet_scene_resort")]


class Host:
    """`run` = closed_loop_run; `api` = the four entry points of librda_hip.so it calls, as function pointers"""

    def __init__(self, rda_lib):
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "closed_loop_host.c")):
            build()
        lib = C.CDLL(SO)
        self.api = Api(*[C.cast(getattr(rda_lib, "rda_" + n_), C.c_void_p).value for n_, _ in Api._fields_])
        self.Scene = Scene
        lib.closed_loop_run.restype = C.c_int
        lib.closed_loop_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self.run = lib.closed_loop_run
        self.fleet_api = FleetApi(*[C.cast(getattr(rda_lib, "rda_" + n_), C.c_void_p).value if hasattr(rda_lib, "rda_" + n_) else None
                                    for n_, _ in FleetApi._fields_])
        lib.closed_loop_fleet_run.restype = C.c_int
        lib.closed_loop_fleet_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                              C.c_double, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self.fleet_run = lib.closed_loop_fleet_run
