"""CPU study behind rda_opts::su_hard_warm (test infrastructure: runs the ORACLE's closed loops, no GPU).
    python tools/experiments/su_hard_warm_cpu.py floors     # slack floor / mu0 of the oracle's warm attempts on the re-sorted north star
    python tools/experiments/su_hard_warm_cpu.py key        # the shipped rule (after an unconverged step, last solve > 3 iterations) on eight loops
Prints interior-point iterations per su-solve (steps 10.. of each loop).  Round-4 results: floors (1e-3, 1e-3) 7.54, (1e-2, 1e-2) 7.30,
(0.1, 1e-3) 5.88, (1, 1e-3) 5.29, (1, 1e-2) 5.71, (3, 1e-2) 6.13;  key: north star re-sorted 7.29 -> 5.75, C4 re-sorted 25.5 -> 14.1,
converged / easy loops unchanged (iter_num = 1 with the first condition alone: 1.0 -> 3.0)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402
from rda_planner_amd import scenarios as sc  # noqa: E402
from oracle.oracle_backend import oracle_backend, api as orc_api  # noqa: E402

lib = orc_api().lib
lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
lib.orc_set_su_hard_warm.argtypes = [C.c_double, C.c_double]
lib.orc_set_threads.argtypes = [C.c_int]
lib.orc_set_threads(min(16, os.cpu_count() or 1))


def run(order, iter_num=4, n_obs=200, T=20, moving=False, steps=60, thr=None):
    car_t, path, obstacles, kw = bench.build_workload(n_obs=n_obs, T=T, n_steps=steps + 20, moving=moving)
    kw["obstacle_order"] = order; kw["iter_num"] = iter_num
    if thr is not None:
        kw["iter_threshold"] = thr
    m = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    st = path[0].copy().reshape(3, 1); ipm = sol = 0
    for k in range(steps):
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
        u, info = m.control(st, 4.0, list(cur))
        assert info["status"] == 0
        st = sc.kinematic_step(st, u, car_t, 0.1)
        if k >= 10:
            ipm += info["su_ipm_iters"]; sol += info["iters"]
    return ipm / sol


try:
    if sys.argv[1:] == ["floors"]:
        for wfl, mu0 in [(0, 0), (1e-3, 1e-3), (1e-2, 1e-2), (0.1, 1e-2), (0.1, 1e-3), (0.3, 1e-3), (1.0, 1e-2), (1.0, 1e-3), (1.0, 1e-4), (3.0, 1e-2)]:
            lib.orc_set_su_warm(wfl, mu0, 30)
            print(f"warm attempts from (slack floor {wfl:g}, mu0 {mu0:g}): {run(True):.2f} interior-point iterations per su-solve", flush=True)
    else:
        cases = [("north star, fixed binding", dict(order=False)), ("fixed, iter_num 1", dict(order=False, iter_num=1)), ("fixed, iter_num 2, iter_threshold 0.02", dict(order=False, iter_num=2, thr=0.02)),
                 ("north star, re-sorted", dict(order=True)), ("re-sorted, iter_num 2", dict(order=True, iter_num=2)), ("C4 fixed", dict(order=False, T=30, moving=True, steps=40)),
                 ("C4 re-sorted", dict(order=True, T=30, moving=True, steps=40)), ("N=20 re-sorted", dict(order=True, n_obs=20))]
        for name, kw in cases:
            lib.orc_set_su_hard_warm(0.0, 0.0); a = run(**kw)
            lib.orc_set_su_hard_warm(1.0, 1e-3); b = run(**kw)
            print(f"{name}: {a:.2f} -> {b:.2f} interior-point iterations per su-solve with su_hard_warm = (1, 1e-3)", flush=True)
finally:
    lib.orc_set_su_warm(1e-3, 1e-3, 30); lib.orc_set_su_hard_warm(1.0, 1e-3)
