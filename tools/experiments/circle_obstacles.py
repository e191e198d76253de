"""One-off (round 5): closed-loop rate with CIRCLE obstacles (norm2 cone: the reference's dynamic_obs example) against polygons of the same
field - the Python API loop of both, same path, same N, kernel times from the library's own events."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rda_planner_amd import scenarios as sc  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
T = 20
car_t = sc.rectangle_robot(dynamics="acker")
path = sc.line_path([4, 25, 0], [60, 25, 0], 0.1)
clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
polys = sc.scene_polygons(N, lo=(6, 5), hi=(62, 45), seed=3, keep_clear=clear, clear_radius=3.2, moving=True)
circles = [sc.circle(float(o.vertex[0].mean()), float(o.vertex[1].mean()), 0.7, tuple(o.velocity.ravel())) for o in polys]
for name, scene in (("polygons", polys), ("circles", circles), ("half / half", polys[::2] + circles[1::2])):
    m = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, receding=T, iter_num=4, max_edge_num=4, max_obs_num=N, ro1=200)
    st = path[0].copy().reshape(3, 1)
    ts, its = [], []
    for k in range(70):
        cur = [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive" else o._replace(center=o.center + o.velocity * (0.1 * k)) for o in scene]
        t0 = time.perf_counter()
        u, info = m.control(st, 4.0, cur)
        ts.append(time.perf_counter() - t0); its.append(info["iters"])
        assert info["status"] == 0
        st = sc.kinematic_step(st, u, car_t, 0.1)
    ts = np.array(ts[10:])
    print(f"{name:12s} N={N}: {1.0 / ts.mean():7.1f} steps/s (Python API, obstacle objects rebuilt per tick), median {np.median(ts) * 1e3:.3f} ms, ADMM iterations {np.mean(its[10:]):.2f}")
