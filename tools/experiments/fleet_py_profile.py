"""One-off (round 5): where Fleet.control spends its host time (64 egos, north-star scene): cProfile top functions."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rda_planner_amd import scenarios as sc  # noqa: E402
from rda_planner_amd.fleet import Fleet  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
car_t, path, obstacles, kw = bench.build_workload(n_obs=200, T=20, n_steps=60)
members = [MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, **kw) for _ in range(M)]
fl = Fleet(members)
states = [path[0].copy().reshape(3, 1) for _ in range(M)]
obs = [list(obstacles) for _ in range(M)]


def loop(n):
    for _ in range(n):
        res = fl.control([s.copy() for s in states], 4.0, obs)
        for i in range(M):
            states[i] = sc.kinematic_step(states[i], res[i][0], car_t, 0.1)


loop(4)
t0 = time.perf_counter(); loop(10); print(f"{M} egos: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per fleet step")
pr = cProfile.Profile(); pr.enable(); loop(10); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
