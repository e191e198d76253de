"""Loader for librda_hip.so.  Fails loudly: there is no CPU fallback in the product path."""
import ctypes as C
import os
import subprocess

from ._capi import CApi, Cfg, c_double_p, c_int_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILT_SO = os.path.join(_HERE, "librda_hip.so")
SO_PATH = os.environ.get("RDA_HIP_SO") or _BUILT_SO     # RDA_HIP_SO: another build of the SAME library (A/B runs of tools/)
_api = None


def build(force=False):
    """compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)"""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + \
           [os.path.join(os.path.dirname(_HERE), "include", "rda_hip.h")]
    stale = force or not os.path.exists(_BUILT_SO) or any(os.path.getmtime(s) > os.path.getmtime(_BUILT_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", src_dir, "-s"] + (["-B"] if force else []))   # -B: `force` must not depend on mtimes
    build_flatten_ext(force)
    return _BUILT_SO


def build_flatten_ext(force=False):
    """csrc/flatten_ext.c -> rda_planner_amd/_flatten.so (CPython extension, plain gcc): the optional accelerator of
    RDA_solver.flatten_scene.  A failure here is not fatal - the numpy implementation is used."""
    import sysconfig
    src, so = os.path.join(_HERE, "csrc", "flatten_ext.c"), os.path.join(_HERE, "_flatten.so")
    if not force and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    try:
        subprocess.check_call(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-I" + sysconfig.get_paths()["include"], "-o", so, src])
    except (OSError, subprocess.CalledProcessError) as e:
        print(f"rda_planner_amd: _flatten extension not built ({e}); flatten_scene uses numpy")
        return None
    return so


def load_library():
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(needs hipcc); rda_planner_amd has no CPU fallback")
    return C.CDLL(SO_PATH)


def hip_api():
    global _api
    if _api is None:
        lib = load_library()
        api = CApi(lib, "rda")
        lib.rda_device_count.restype = C.c_int
        lib.rda_set_device.argtypes = [C.c_int]
        lib.rda_set_device.restype = C.c_int
        lib.rda_strerror.restype = C.c_char_p
        lib.rda_strerror.argtypes = [C.c_int]
        lib.rda_last_nonconvex.argtypes = [C.c_void_p]
        lib.rda_last_nonconvex.restype = C.c_int
        lib.rda_lammuz_kernel.argtypes = [C.c_void_p]
        lib.rda_lammuz_kernel.restype = C.c_char_p
        lib.rda_shard_comm_count.argtypes = [C.c_void_p]
        lib.rda_shard_comm_count.restype = C.c_int
        lib.rda_debug_su_prof.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        lib.rda_debug_su_prof.restype = C.c_int
        lib.rda_shard_unique_id.argtypes = [C.c_void_p, C.c_void_p]
        lib.rda_shard_comm_init.argtypes = [C.c_void_p, C.c_void_p]
        lib.rda_shard_unique_id.restype = C.c_int
        lib.rda_shard_comm_init.restype = C.c_int
        lib.rda_upload_trace.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        lib.rda_enqueue_step.argtypes = [C.c_void_p, C.c_int]
        lib.rda_sync.argtypes = [C.c_void_p]
        lib.rda_fetch_result.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, C.c_void_p]
        lib.rda_timing_reset.argtypes = [C.c_void_p, C.c_int]
        lib.rda_timing_read.argtypes = [C.c_void_p, C.c_int, c_double_p, c_int_p]
        lib.rda_timing_launches.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int, c_int_p]
        lib.rda_lammuz_batch.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p, c_double_p, c_double_p,
                                         c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_double, C.c_double,
                                         C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        lib.rda_su_solve.argtypes = [C.POINTER(Cfg)] + [c_double_p] * 3 + [C.c_double] + [c_double_p] * 7 + [c_int_p]
        lib.rda_su_solve_opts.argtypes = [C.POINTER(Cfg), C.c_void_p] + [c_double_p] * 3 + [C.c_double] + [c_double_p] * 7 + [c_int_p]
        lib.rda_su_solve_opts.restype = C.c_int
        for name in ("upload_trace", "enqueue_step", "sync", "fetch_result", "timing_reset",
                     "timing_read", "timing_launches", "lammuz_batch", "su_solve"):
            getattr(lib, "rda_" + name).restype = C.c_int
        if lib.rda_device_count() < 1:
            raise RuntimeError("librda_hip.so loaded but no HIP device is visible; rda_planner_amd has no CPU fallback")
        _api = api
    return _api
