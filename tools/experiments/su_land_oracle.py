"""CPU study behind the landing of the su solve (oracle/rda_oracle.c su_polish): the SAME closed loop solved along two different
interior-point paths of the oracle - warm starts (mirror of the kernel's start rules) and cold starts - step by step from the same solver state.
Without the landing the two stop at different points of the central path; with it both should end on the same vertex.
    python tools/experiments/su_polish_oracle.py [n_obs] [T] [steps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402
from oracle.oracle_backend import oracle_backend, api as orc_api  # noqa: E402
from rda_planner_amd.mpc import MPC                            # noqa: E402
from rda_planner_amd import scenarios as sc                    # noqa: E402


def run(n_obs, T, steps, polish, moving=False, tight=False):
    lib = orc_api().lib
    lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    lib.orc_set_threads(8)
    lib.orc_set_su_land(polish)
    car_t, path, obstacles, kw = bench.build_workload(n_obs=n_obs, T=T, n_steps=steps + 20, moving=moving)
    if tight:
        clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
        obstacles = sc.scene_polygons(n_obs, lo=(8, 17), hi=(50, 33), seed=sc.SEED + 5, keep_clear=clear, clear_radius=2.0, moving=moving)
    a = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    b = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    state = path[0].copy().reshape(3, 1)
    worst, over, ipm_a, ipm_b, flips = 0.0, 0, 0, 0, 0
    for k in range(steps):
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
        lib.orc_set_su_warm(1e-3, 1e-3, 30)
        ua, ia = a.control(state.copy(), 4.0, list(cur))
        lib.orc_set_su_warm(0.0, 0.0, 0)
        ub, ib = b.control(state.copy(), 4.0, list(cur))
        if ia["iters"] == ib["iters"]:
            d = float(np.abs(a.cur_vel_array - b.cur_vel_array).max())
            worst = max(worst, d); over += d > 1e-7
        else:
            flips += 1
        ipm_a += ia["su_ipm_iters"]; ipm_b += ib["su_ipm_iters"]
        b.rda.set_state(a.rda.get_state()); b.cur_vel_array = a.cur_vel_array.copy()
        state = sc.kinematic_step(state, ua, car_t, 0.1)
    lib.orc_set_su_warm(1e-3, 1e-3, 30); lib.orc_set_su_land(0)
    st = (C.c_long * 8)(); lib.orc_get_su_land_stats(st)
    if polish:
        print(f"   landing: {st[0]} solves, {st[1]} accepted, {st[2] / max(st[0], 1):.2f} rounds per solve; refused: factor failed {st[3]}, active set still moving {st[4]}, "
              f"not stationary {st[5]}; rows moved in / out {st[6]}")
    return worst, over, flips, ipm_a, ipm_b


if __name__ == "__main__":
    n_obs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for tight in (False, True):
        for moving in (False, True):
            for polish in (0, 1):
                w, o, f, ia, ib = run(n_obs, T, steps, polish, moving, tight)
                print(f"N={n_obs} T={T} tight={tight} moving={moving} polish={polish}: max |du| warm path vs cold path over the horizon {w:.2e}, steps > 1e-7: {o}/{steps - f}, "
                      f"ADMM-count flips {f}, interior-point iterations {ia} / {ib}", flush=True)
