"""debug: how many rows the common-path LamMuZ kernel defers to the work-list kernel (split launch form), per step, in a closed loop
python tools/worklist_stats.py [--n-obs N] [--horizon T] [--moving] [--steps K]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n-obs", type=int, default=2000); ap.add_argument("--horizon", type=int, default=20); ap.add_argument("--moving", action="store_true")
ap.add_argument("--steps", type=int, default=40); ap.add_argument("--order", action="store_true")
ap.add_argument("--iter-num", type=int, default=0, help="ADMM iterations per step (1: the counter then belongs to the FIRST LamMuZ launch of every tick)")
a = ap.parse_args()
import bench
from rda_planner_amd.mpc import MPC
from rda_planner_amd import scenarios as sc
from rda_planner_amd._lib import hip_api
car_t, path, obstacles, kw = bench.build_workload(n_obs=a.n_obs, T=a.horizon, n_steps=a.steps + 20, moving=a.moving)
kw["obstacle_order"] = bool(a.order)
if a.iter_num: kw["iter_num"] = a.iter_num
mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, **kw)
lib = hip_api().lib
lib.rda_debug_worklist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
state = path[0].copy().reshape(3, 1); rows = C.c_int(0); hist = []
for k in range(a.steps):
    cur = obstacles if not a.moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
    u, info = mpc.control(state, 4.0, list(cur))
    state = sc.kinematic_step(state, u, car_t, 0.1)
    lib.rda_debug_worklist(mpc.rda._be.handle, C.byref(rows)); hist.append(rows.value)
tot = a.n_obs * a.horizon
print(f"N={a.n_obs} T={a.horizon} moving={a.moving}: {tot} rows per launch; work list of the last iteration of each step: first steps {hist[:4]}, "
      f"then median {int(np.median(hist[5:]))} ({np.median(hist[5:]) / tot:.1%}), max {max(hist[5:])}, kernel form {lib.rda_lammuz_kernel(mpc.rda._be.handle)}")
