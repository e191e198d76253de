"""Round 6: what the landing of the su solve does per configuration - GPU closed loops alone (Python MPC caller, the reference's re-sorting protocol and the
fixed binding), with and without `rda_opts::su_land`: interior-point iterations per su-solve, landings accepted / refused, landing rounds and passes per solve.

    python tools/experiments/land_stats.py [--steps 30]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rda_planner_amd import scenarios as sc          # noqa: E402
from rda_planner_amd.mpc import MPC                  # noqa: E402
from rda_planner_amd.rda_solver import hip_options   # noqa: E402


def workload(n_obs, T, n_steps, moving=False):
    from test_gpu_baseline_sizes import _workload
    return _workload(n_obs, T, n_steps, moving=moving)


def run(name, n_obs, T, moving, steps, ordered, **opts):
    car_t, path, obstacles, kw = workload(n_obs, T, steps + 10, moving)
    kw["obstacle_order"] = ordered
    gpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(**opts), **kw)
    state = path[0].copy().reshape(3, 1)
    ipm = solves = 0
    for i in range(steps):
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * i)) for o in obstacles]
        u, info = gpu.control(state.copy(), 4.0, list(cur))
        ipm += info["su_ipm_iters"]; solves += info["iters"]
        state = sc.kinematic_step(state, u, car_t, 0.1)
    st = (C.c_int32 * 20)()
    gpu.rda._be.api.lib.rda_debug_su_land_n(gpu.rda._be.handle, st, 20)
    st = list(st)
    print(f"{name:10s} ordered={int(ordered)} {str(opts):40s}: {solves / steps:4.2f} su-solves/step, interior-point its/solve {ipm / solves:5.2f}, "
          f"landings accepted {st[0]} refused {st[1]} rounds/solve {st[2] / solves:4.2f} landing passes/solve {st[3] / solves:4.2f}; speculative tried {st[4]} accepted {st[5]}, "
          f"by decade of the start's rd0 (<1e-4 .. >=1): tried {st[6:12]} accepted {st[12:18]}; blind tried {st[18]} accepted {st[19]}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    for name, n_obs, T, moving in (("NS", 200, 20, False), ("N=20", 20, 20, False), ("N=2000", 2000, 20, False), ("C4", 200, 30, True), ("C5shape", 100, 25, False)):
        for ordered in (True, False):
            run(name, n_obs, T, moving, a.steps, ordered, su_land=0)
            run(name, n_obs, T, moving, a.steps, ordered, su_land_first=1)
            run(name, n_obs, T, moving, a.steps, ordered, su_land_first=2)
