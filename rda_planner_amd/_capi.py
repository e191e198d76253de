"""ctypes view of the C-ABI declared in include/rda_hip.h.

The binding class is parametrised by the symbol prefix: the product instantiates it on
`librda_hip.so` (symbols `rda_*`); the test suite re-uses it on its CPU checker, which
deliberately shares struct layouts and argument order (symbols `orc_*`).
"""
import ctypes as C
import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Cfg(C.Structure):
    _fields_ = [("T", C.c_int), ("N", C.c_int), ("E", C.c_int), ("R", C.c_int),
                ("dynamics", C.c_int), ("accelerated", C.c_int), ("iter_num", C.c_int),
                ("robot_norm2", C.c_int),
                ("dt", C.c_double), ("L", C.c_double),
                ("max_speed", C.c_double * 2), ("acce_bound", C.c_double * 2),
                ("iter_threshold", C.c_double), ("ws", C.c_double), ("wu", C.c_double),
                ("slack_gain", C.c_double), ("max_sd", C.c_double), ("min_sd", C.c_double),
                ("ro1", C.c_double), ("ro2", C.c_double),
                ("delta", C.c_double), ("eps_u", C.c_double)]


class Opts(C.Structure):
    """rda_opts of include/rda_hip.h: per-handle solver options that are not reference arguments (filled by rda_opts_init)"""
    _fields_ = [("lmz_mode", C.c_int), ("tie_centre", C.c_int), ("lmz_mu", C.c_double), ("su_tol", C.c_double * 3), ("su_tol_early", C.c_double * 3), ("su_hard_warm", C.c_double * 2),
                ("lmz_warm", C.c_int), ("lmz_rows", C.c_int), ("lmz_dense_from", C.c_int), ("lmz_split", C.c_int),
                ("lmz_ip_rows", C.c_int), ("lmz_ip_warm", C.c_int), ("su_pre", C.c_int), ("su_light", C.c_int),
                ("su_warm_first", C.c_int), ("su_warm_cap", C.c_int), ("su_easy_max", C.c_int), ("su_easy_nopred", C.c_int),
                ("su_cold_from", C.c_int), ("su_cold_probe", C.c_int), ("zero_copy", C.c_int), ("early_finish", C.c_int),
                ("fuse_track", C.c_int), ("su_prof", C.c_int), ("su_split", C.c_int), ("duals_follow", C.c_int), ("su_accept", C.c_int), ("su_first_attempt", C.c_int),
                ("su_warm", C.c_double * 2), ("su_warm_endgame", C.c_double * 2), ("su_warm_clip", C.c_double),
                ("su_easy", C.c_double * 5), ("su_land", C.c_int), ("su_land_tol", C.c_double * 3), ("su_land_rho", C.c_double), ("su_land_first", C.c_int), ("su_land_blind_from", C.c_int)]


class Info(C.Structure):
    _fields_ = [("resi_dual", C.c_double), ("resi_pri", C.c_double), ("iters", C.c_int),
                ("su_status", C.c_int), ("su_ipm_iters", C.c_int), ("lmz_fail", C.c_int)]


DYNAMICS = {"acker": 0, "diff": 1, "omni": 2}


def dptr(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(c_int_p) if a is not None else None


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class CApi:
    """Thin typed wrapper over `<prefix>_*` entry points of a loaded shared library."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        f = self._f
        f("create").argtypes = [C.POINTER(Cfg), c_double_p, c_double_p, C.POINTER(C.c_void_p)]
        f("create").restype = C.c_int
        self.has_opts = hasattr(lib, f"{prefix}_create_opts")
        if self.has_opts:
            f("opts_init").argtypes = [C.POINTER(Opts)]
            f("opts_init").restype = None
            f("create_opts").argtypes = [C.POINTER(Cfg), C.POINTER(Opts), c_double_p, c_double_p, C.POINTER(C.c_void_p)]
            f("create_opts").restype = C.c_int
            f("get_su_history").argtypes = [C.c_void_p, c_int_p, c_double_p]
            f("set_su_history").argtypes = [C.c_void_p, c_int_p, c_double_p]
            f("get_su_history").restype = f("set_su_history").restype = C.c_int
            if hasattr(lib, f"{prefix}_get_su_history_n"):
                f("get_su_history_n").argtypes = [C.c_void_p, c_int_p, C.c_int, c_double_p]
                f("set_su_history_n").argtypes = [C.c_void_p, c_int_p, C.c_int, c_double_p]
                f("get_su_history_n").restype = f("set_su_history_n").restype = C.c_int
        f("destroy").argtypes = [C.c_void_p]
        f("destroy").restype = None
        f("set_adjust").argtypes = [C.c_void_p] + [C.c_double] * 5
        f("set_adjust").restype = C.c_int
        f("reset").argtypes = [C.c_void_p]
        f("reset").restype = C.c_int
        f("step").argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_double, C.c_int,
                              c_double_p, c_double_p, c_int_p, C.c_int, c_double_p, c_double_p,
                              C.POINTER(Info)]
        f("step").restype = C.c_int
        f("get_state").argtypes = [C.c_void_p] + [c_double_p] * 8
        f("get_state").restype = C.c_int
        f("set_state").argtypes = [C.c_void_p] + [c_double_p] * 8
        f("set_state").restype = C.c_int
        # obstacle sharding + host-driven ADMM pieces
        f("upload_obstacles").argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int_p, C.c_int]
        f("shard_config").argtypes = [C.c_void_p, C.c_int, C.c_int]
        f("shard_chunk_doubles").argtypes = [C.c_void_p]
        f("shard_get_chunk").argtypes = [C.c_void_p, c_double_p]
        f("shard_set_chunks").argtypes = [C.c_void_p, c_double_p]
        f("admm_begin").argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_double]
        f("admm_su").argtypes = [C.c_void_p, C.c_int, c_int_p]
        f("admm_lammuz").argtypes = [C.c_void_p]
        f("admm_finish").argtypes = [C.c_void_p, c_double_p, c_double_p, C.POINTER(Info)]
        for name in ("upload_obstacles", "shard_config", "shard_chunk_doubles", "shard_get_chunk", "shard_set_chunks",
                     "admm_begin", "admm_su", "admm_lammuz", "admm_finish"):
            f(name).restype = C.c_int

        # caller-side obstacle pipeline on the device (HIP library only)
        self.has_scene = hasattr(lib, f"{prefix}_step_scene")
        if self.has_scene:
            f("upload_scene").argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p, C.c_int, c_int_p]
            f("upload_scene").restype = C.c_int
            f("step_scene").argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_double, C.c_int, c_int_p, c_int_p,
                                        c_double_p, c_double_p, c_double_p, C.c_int, c_double_p, c_double_p, C.POINTER(Info)]
            f("step_scene").restype = C.c_int
            f("get_obstacles").argtypes = [C.c_void_p, c_double_p, c_double_p, c_int_p, c_int_p]
            f("get_obstacles").restype = C.c_int

        # caller-side pre_process on the device (HIP library only)
        self.has_track = hasattr(lib, f"{prefix}_step_tracked")
        self.has_pipeline = False
        if self.has_track:
            f("upload_path").argtypes = [C.c_void_p, C.c_int, c_double_p]
            f("upload_path").restype = C.c_int
            f("step_tracked").argtypes = [C.c_void_p, c_double_p, C.c_double, C.c_int, C.c_double, C.c_int, c_double_p,
                                          c_double_p, c_double_p, C.POINTER(Info), c_double_p, c_double_p, c_int_p, c_double_p]
            f("step_tracked").restype = C.c_int
            self.has_pipeline = hasattr(lib, f"{prefix}_tracked_begin")
            if self.has_pipeline:
                f("tracked_begin").argtypes = [C.c_void_p, c_double_p, C.c_double, C.c_int, C.c_double, C.c_int, c_double_p]
                f("upload_scene_async").argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p, C.c_int]
                f("tracked_finish").argtypes = [C.c_void_p, c_double_p, c_double_p, C.POINTER(Info), c_double_p, c_double_p,
                                                c_int_p, c_double_p]
                for name in ("tracked_begin", "upload_scene_async", "tracked_finish"):
                    f(name).restype = C.c_int
                if hasattr(lib, f"{prefix}_scene_resort"):
                    f("scene_resort").argtypes = [C.c_void_p, c_double_p]
                    f("scene_resort").restype = C.c_int
            if hasattr(lib, f"{prefix}_fleet_step_tracked"):
                f("fleet_step_tracked").argtypes = [C.c_void_p, c_double_p, c_double_p, c_int_p, C.c_double, C.c_int, c_double_p,
                                                    c_double_p, c_double_p, C.POINTER(Info), c_double_p, c_int_p, c_double_p]
                f("fleet_step_tracked").restype = C.c_int
        # batched multi-ego stepping (HIP library only)
        self.has_fleet = hasattr(lib, f"{prefix}_fleet_step")
        if self.has_fleet:
            f("fleet_create").argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
            f("fleet_destroy").argtypes = [C.c_void_p]
            f("fleet_destroy").restype = None
            f("fleet_size").argtypes = [C.c_void_p]
            f("fleet_step").argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                        C.POINTER(Info)]
            f("fleet_enqueue_range").argtypes = [C.c_void_p, C.c_int, C.c_int]
            f("fleet_sync").argtypes = [C.c_void_p]
            for name in ("fleet_create", "fleet_size", "fleet_step", "fleet_enqueue_range", "fleet_sync"):
                f(name).restype = C.c_int
            if hasattr(lib, f"{prefix}_fleet_scene_resort"):
                f("fleet_scene_resort").argtypes = [C.c_void_p, c_double_p, C.c_int]
                f("fleet_scene_resort").restype = C.c_int
            self.has_fleet_scenes = hasattr(lib, f"{prefix}_fleet_upload_scenes")
            if self.has_fleet_scenes:
                f("fleet_upload_scenes").argtypes = [C.c_void_p, c_int_p, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p, c_int_p]
                f("fleet_upload_scenes").restype = C.c_int

    def _f(self, name):
        return getattr(self.lib, f"{self.prefix}_{name}")

    def __getattr__(self, name):
        return self._f(name)
