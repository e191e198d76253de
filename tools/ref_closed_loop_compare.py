#!/usr/bin/env python
"""Closed-loop comparison of the UNMODIFIED reference (on oracle/refshim: cvxpy stand-in + generic interior-point
solver) with this repo's oracle (tie-break T1) on the reference's example scenes.  Test infrastructure / evidence for
DESIGN.md 2 ("tie-breaks"); needs /root/reference, runs on the CPU, minutes per scene (the stand-in solves every
LamMuZ problem with a general sparse IPM).

    python tools/ref_closed_loop_compare.py path_track|corridor [iter_num] [steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh                      # noqa: E402
from oracle.oracle_backend import oracle_backend          # noqa: E402
from rda_planner_amd import scenarios as sc               # noqa: E402
from rda_planner_amd.mpc import MPC                       # noqa: E402

rs, mp, backend = rh.load()
which = sys.argv[1] if len(sys.argv) > 1 else "corridor"
if which == "path_track":                                 # example/path_track/path_track_diff.py
    car_t = sc.rectangle_robot(dynamics="diff", wheelbase=0)
    path, obstacles = sc.path_track_ref(), sc.scene_path_track()
    kw = dict(receding=10, sample_time=0.1, iter_num=2, max_edge_num=4, max_obs_num=11, ro1=300, obstacle_order=True)
    steps = 400
else:                                                     # example/corridor/corridor.py (dubins (0,20,0)->(60,20,0) is a line)
    car_t = sc.rectangle_robot(dynamics="acker")
    path, obstacles = sc.line_path([0, 20, 0], [60, 20, 0], 0.1), sc.scene_corridor(0)
    kw = dict(receding=10, sample_time=0.1, iter_num=4, max_edge_num=4, max_obs_num=6)
    steps = 300
if len(sys.argv) > 2:
    kw["iter_num"] = int(sys.argv[2])
if len(sys.argv) > 3:
    steps = int(sys.argv[3])
start, speed = path[0].copy().reshape(3, 1), 4.0


def loop(m, name, count=None):
    state, traj, clr, arrived, t0 = start.copy(), [], np.inf, None, time.time()
    for k in range(steps):
        u, info = m.control(state, speed, list(obstacles))
        traj.append(state.ravel().copy())
        state = sc.kinematic_step(state, u, car_t, 0.1)
        clr = min(clr, sc.clearance(car_t, state, obstacles))
        if info["arrive"]:
            arrived = k
            break
    its = np.mean(count) if count else None
    print(f"{name:16s} iter_num {kw['iter_num']} steps {k + 1:4d} arrived {arrived} min clearance {clr:7.3f} m  mean ADMM iterations {its}  "
          f"({time.time() - t0:.0f} s)", flush=True)
    return np.array(traj)


ref = mp.MPC(car_t, [p.copy() for p in path], time_print=False, process_num=1, **kw)
cnt = []
orig, oc = ref.rda.rda_solver, ref.control


def counted():
    cnt[-1] += 1
    return orig()


def control(*a, **k):
    cnt.append(0)
    return oc(*a, **k)


ref.rda.rda_solver, ref.control = counted, control
ours = MPC(car_t, [p.copy() for p in path], time_print=False, _backend=oracle_backend, **kw)
ours_it = []
oo = ours.control


def ocontrol(*a, **k):
    u, info = oo(*a, **k)
    ours_it.append(info["iters"])
    return u, info


ours.control = ocontrol
tr_o = loop(ours, "oracle (T1)", ours_it)
tr_r = loop(ref, f"reference+{backend}", cnt)
n = min(len(tr_o), len(tr_r))
print("max |xy| deviation over the common steps: %.3f m" % np.max(np.linalg.norm(tr_o[:n, :2] - tr_r[:n, :2], axis=1)))
