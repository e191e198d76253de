"""One-off (round 5): which wave bounds phase (4) of the su interior-point iteration - the Riccati recursion (waves 0 / 1), the adjoint sweep + early
verdict (wave 2) or the measures (wave 3)?  Needs a library built with -DSU_P4 (su_device.h: per-wave cycles from the start of the phase to the
wave's arrival at the barrier, slots 0..3 of rda_debug_su_prof; slot 8 = passes):  RDA_HIP_SO=tools/_bin/librda_hip_p4_<v>.so python tools/experiments/p4_waves.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rda_planner_amd import scenarios as sc  # noqa: E402
from rda_planner_amd._lib import hip_api  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402
from rda_planner_amd.rda_solver import hip_options  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
car_t, path, obstacles, kw = bench.build_workload(n_obs=200, T=T, n_steps=120)
mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(su_prof=1), **kw)
lib = hip_api().lib
state = path[0].copy().reshape(3, 1)
out = (C.c_longlong * 16)()
for k in range(80):
    u, info = mpc.control(state, 4.0, list(obstacles))
    state = sc.kinematic_step(state, u, car_t, 0.1)
    if k == 9:
        lib.rda_debug_su_prof(mpc.rda._be.handle, out)
lib.rda_debug_su_prof(mpc.rda._be.handle, out)
n = max(out[8], 1)
print(f"T={T}: {n} passes; cycles from the start of phase (4) to the barrier, per pass: " + ", ".join(f"wave {w}: {out[w] / n:.0f}" for w in range(4)))
