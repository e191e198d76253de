"""Random closed-loop soak: the HIP path (default options: device-side obstacle pipeline, tracking, pipelined tick) against the CPU
oracle, step by step from the SAME solver state (the oracle's duals / nominal controls are re-synchronised to the GPU's after
every step, so differences cannot accumulate).  Test infrastructure, like everything that touches oracle/.
    gpurun -- 'python tools/soak.py --scenes 24 --steps 120'
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rda_planner_amd import scenarios as sc                      # noqa: E402
from rda_planner_amd.mpc import MPC                              # noqa: E402
from oracle.oracle_backend import oracle_backend, api as orc_api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=12)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--only", type=int, default=-1, help="run this scene only (the random draws of the others are still made)")
    ap.add_argument("--lmz-central", type=float, default=0.0, help="interior-point LamMuZ mode on both sides (central-path point at this barrier parameter, e.g. 1e-3)")
    ap.add_argument("--dump", default="", help="record the oracle's su-problems (same state as the GPU's: re-synchronised every step) for tools/su_replay.py")
    ap.add_argument("--so", default="", help="another build of librda_hip.so (A/B against an older commit)")
    a = ap.parse_args()
    if a.so:
        from rda_planner_amd import _lib
        _lib.SO_PATH = os.path.abspath(a.so)
    if a.dump:
        import ctypes as C
        orc_api().lib.orc_set_su_dump.argtypes = [C.c_char_p]
        os.makedirs(os.path.dirname(os.path.abspath(a.dump)), exist_ok=True)
        orc_api().lib.orc_set_su_dump(os.path.abspath(a.dump).encode())
    if a.lmz_central > 0:
        import ctypes as C
        orc_api().lib.orc_set_lmz_ipm_mu.argtypes = [C.c_double]
        orc_api().lib.orc_set_lmz_mode(1); orc_api().lib.orc_set_lmz_ipm_mu(a.lmz_central)
    orc_api().lib.orc_set_threads(min(16, os.cpu_count() or 1))      # more threads than that slow the oracle down (bench.py thread sweep)
    rng = np.random.default_rng(a.seed)
    tot = bad_u = bad_it = failed = 0
    worst = 0.0
    t0 = time.time()
    for s in range(a.scenes):
        dyn = ["acker", "diff", "omni"][int(rng.integers(3))]
        T = int(rng.choice([10, 15, 20, 25]))
        N = int(rng.integers(8, 60))
        car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
        y = 25.0
        path = sc.line_path([4, y, 0], [4 + 0.4 * a.steps + 12, y, 0], 0.1)
        clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
        moving = bool(rng.integers(2))
        scene = sc.scene_polygons(N, lo=(6, y - 12), hi=(4 + 0.4 * a.steps + 14, y + 12), seed=1000 * a.seed + s, keep_clear=clear,
                                  clear_radius=float(rng.uniform(2.4, 3.4)), moving=moving)
        for _ in range(int(rng.integers(0, 4))):
            scene.append(sc.circle(float(rng.uniform(10, 40)), y + float(rng.choice([-1, 1])) * float(rng.uniform(3.5, 8)),
                                   float(rng.uniform(0.4, 1.2)), (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)))))
        kw = dict(receding=T, iter_num=int(rng.integers(2, 5)), max_edge_num=4, max_obs_num=int(rng.integers(max(4, N // 2), N + 6)),
                  ro1=float(rng.choice([200, 300])), time_print=False)
        if a.lmz_central > 0:
            kw["lmz_central"] = a.lmz_central
        gpu = MPC(car_t, [p.copy() for p in path], **kw)
        cpu = MPC(car_t, [p.copy() for p in path], _backend=oracle_backend, **kw)
        st = path[0].copy().reshape(3, 1)
        if dyn == "omni":
            st[2, 0] = 0.0
        speed = float(rng.uniform(2.5, 4.5))
        if a.only >= 0 and s != a.only:
            continue
        for k in range(a.steps):
            cur = [o if not np.any(o.velocity) else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                                     else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in scene]
            ug, ig = gpu.control(st.copy(), speed, list(cur))
            uc, ic = cpu.control(st.copy(), speed, list(cur))
            tot += 1
            du = float(np.abs(ug - uc).max())
            worst = max(worst, du)
            if ig["status"] or ic["status"]:
                failed += 1
                print(f"scene {s} ({dyn} T={T} N={N}) step {k}: su status gpu {ig['status']} (ipm {ig['su_ipm_iters']}), oracle {ic['status']} (ipm {ic['su_ipm_iters']})")
            if ig["iters"] != ic["iters"]:
                bad_it += 1
                print(f"scene {s} ({dyn} T={T} N={N}) step {k}: iterations {ig['iters']} vs {ic['iters']}, du {du:.2e}")
            elif du > a.tol:
                bad_u += 1
                print(f"scene {s} ({dyn} T={T} N={N}) step {k}: du {du:.2e} (ipm {ig['su_ipm_iters']} vs {ic['su_ipm_iters']})")
            # the oracle continues from the GPU's state
            cpu.rda.set_state(gpu.rda.get_state())
            cpu.cur_vel_array = gpu.cur_vel_array.copy()
            cpu.cur_index = gpu.cur_index
            st = sc.kinematic_step(st, ug, car_t, 0.1)
            if ig["arrive"]:
                break
    print(f"soak: {tot} steps over {a.scenes} scenes in {time.time() - t0:.0f} s; max |du| {worst:.2e}; control mismatches > {a.tol:g}: {bad_u}; "
          f"iteration-count mismatches: {bad_it}; steps with a failed su-solve: {failed}")


if __name__ == "__main__":
    main()
