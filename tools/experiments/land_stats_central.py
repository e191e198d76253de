"""Round 6: the su-solves of the interior-point LamMuZ mode (lmz_central = 1e-3) on the headline loop: interior-point iterations, landing rounds, speculative landings.
    python tools/experiments/land_stats_central.py [--steps 30]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rda_planner_amd import scenarios as sc          # noqa: E402
from rda_planner_amd.mpc import MPC                  # noqa: E402
from rda_planner_amd.rda_solver import hip_options   # noqa: E402


def run(ordered, steps, central, **opts):
    from test_gpu_baseline_sizes import _workload
    car_t, path, obstacles, kw = _workload(200, 20, steps + 10)
    kw["obstacle_order"] = ordered
    if central:
        kw["lmz_central"] = 1e-3
    gpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(**opts), **kw)
    state = path[0].copy().reshape(3, 1)
    ipm = solves = 0
    per_it = {}
    for i in range(steps):
        u, info = gpu.control(state.copy(), 4.0, list(obstacles))
        ipm += info["su_ipm_iters"]; solves += info["iters"]
        state = sc.kinematic_step(state, u, car_t, 0.1)
    st = (C.c_int32 * 20)()
    gpu.rda._be.api.lib.rda_debug_su_land_n(gpu.rda._be.handle, st, 20)
    st = list(st)
    print(f"central={central} ordered={int(ordered)} {str(opts):30s}: {solves / steps:4.2f} su-solves/step, interior-point its/solve {ipm / solves:5.2f}, landings accepted {st[0]} refused {st[1]} "
          f"rounds/solve {st[2] / solves:4.2f} passes/solve {st[3] / solves:4.2f}; speculative tried {st[4]} accepted {st[5]} by decade tried {st[6:12]} accepted {st[12:18]}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    for central in (False, True):
        for ordered in (True, False):
            run(ordered, a.steps, central)
            run(ordered, a.steps, central, su_land=0)
