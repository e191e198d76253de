"""Round 6: circle rows in the DENSE launch form (common-path kernel + work-list kernel): closed-loop rate and LamMuZ kernel times of a big scene whose
obstacles are circles / half circles, Python API loop.   python tools/experiments/circle_dense.py [N]   (RDA_HIP_SO: another build, A/B)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rda_planner_amd import scenarios as sc  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = 20
car_t = sc.rectangle_robot(dynamics="acker")
path = sc.line_path([4, 25, 0], [60, 25, 0], 0.1)
clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
polys = sc.scene_polygons(N, lo=(6, -30), hi=(62, 80), seed=3, keep_clear=clear, clear_radius=3.2, moving=False)
circles = [sc.circle(float(o.vertex[0].mean()), float(o.vertex[1].mean()), 0.5, (0.0, 0.0)) for o in polys]
for name, scene in (("polygons", polys), ("circles", circles), ("half / half", polys[::2] + circles[1::2])):
    m = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, receding=T, iter_num=4, max_edge_num=4, max_obs_num=N, ro1=200)
    st = path[0].copy().reshape(3, 1)
    ts, us = [], []
    for k in range(40):
        t0 = time.perf_counter()
        u, info = m.control(st, 4.0, list(scene))
        ts.append(time.perf_counter() - t0)
        assert info["status"] == 0
        us.append(np.asarray(u).ravel().copy())
        st = sc.kinematic_step(st, u, car_t, 0.1)
    ts = np.array(ts[10:])
    print(f"N={N} {name:12s}: {1.0 / ts.mean():8.1f} steps/s (median {1.0 / np.median(ts):8.1f}), launch form {m.rda._be.api.lib.rda_lammuz_kernel(m.rda._be.handle).decode()}, checksum {float(np.sum(np.abs(us))):.15e}", flush=True)
