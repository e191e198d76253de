"""Randomised closed-loop soak: the HIP path (defaults: device-side obstacle pipeline, tracking, pipelined tick, all start rules of the
su interior point) against the CPU oracle, step by step from the SAME solver state - the oracle's duals / nominal controls are
re-synchronised to the GPU's after every step, so differences cannot accumulate and every step is an independent sample.
Test infrastructure (it drives oracle/): used by tests/test_gpu_soak.py (the -m gpu mini-soak the driver runs) and tools/soak.py (CLI).

What is compared per step
    du_raw   max |u_gpu - u_oracle| of the APPLIED control (u[:, 0]) in the solver's own coordinates
    du_body  the same control expressed in what the robot does with it (`body_rates`): linear velocity and yaw rate for acker / diff,
             the Cartesian velocity for omni - reported, not asserted (for omni it is |v| times the heading difference)
    iters    ADMM iteration counts (early stop, rda_solver.py:594)
    status   su-solves that did not converge (either side)
"""
import ctypes as C
import os

import numpy as np

from rda_planner_amd import scenarios as sc
from rda_planner_amd.mpc import MPC
from oracle.oracle_backend import oracle_backend, api as orc_api


def body_rates(u, dyn, L):
    """applied control -> what the kinematic model integrates (mpc.py:293-336): (v, yaw rate) or, omni, (vx, vy)"""
    v, w = float(u[0]), float(u[1])
    if dyn == "acker":
        return np.array([v, v * np.tan(w) / L])
    if dyn == "diff":
        return np.array([v, w])
    return np.array([v * np.cos(w), v * np.sin(w)])


def draw_scene(rng, seed, s, steps, large=False, exotic=False, circle_robot=False, tight=False, robots=False, circles=False):
    """the random draws of scene s (kinematics, horizon, obstacle field, solver arguments) - one rng stream per soak, consumed in scene order.
    `large`: the BASELINE regime instead of the examples' (T in {20, 25, 30}, 100 - 420 obstacles in a field 2.5 times as wide).
    `exotic`: what the examples do not use but the reference interface allows - the reference's default max_edge_num = 5 and more (polygons
    with 3 .. E vertices), a circle robot (norm2 cone, R = 3; `circle_robot`: the interior-point LamMuZ mode only, like the library), accelerated=False, horizons outside the compiled instantiations (5, 12, 40),
    obstacle_order=False, other penalty weights.
    `robots` (with exotic): convex bodies with 3 / 5 / 6 / 8 edges instead of the rectangle.
    `circles`: two of three polygons become circle obstacles (norm2 cone) of about the same size, moving ones stay moving.
    `tight`: half the clearance between path and obstacles (1.2 - 1.7 m for a 1.6 m wide body: the lane is blocked here and there)"""
    dyn = ["acker", "diff", "omni"][int(rng.integers(3))]
    T = int(rng.choice([20, 25, 30] if large else [10, 15, 20, 25]))
    N = int(rng.integers(100, 420)) if large else int(rng.integers(8, 60))
    E, extra, kmax = 4, {}, 4
    if exotic:
        T = int(rng.choice([5, 12, 20, 40]))
        E = int(rng.choice([5, 6, 8])); kmax = E
        if rng.random() < 0.4 and dyn != "acker" and circle_robot:        # (the draw is made either way: same scenes in both LamMuZ modes)
            car_t = sc.circle_robot(radius=float(rng.uniform(0.5, 1.0)), dynamics=dyn)
        else:
            car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
        if robots and car_t.cone_type == "Rpositive" and rng.random() < 0.6:
            # a convex k-gon body, k in {3, 5, 6, 8} (R = k half-spaces; E + R + 1 > 16 leaves the packed LamMuZ kernel for the one-row-per-wave form)
            k = int(rng.choice([3, 5, 6, 8]))
            ang = 2 * np.pi * (np.arange(k) + 0.5) / k
            V = np.vstack((1.5 + 2.3 * np.cos(ang) if dyn == "acker" else 2.3 * np.cos(ang), 0.9 * np.sin(ang)))
            Gk, hk = sc.polygon_halfspaces(V)
            car_t = car_t._replace(G=Gk, h=hk)
        if rng.random() < 0.25:
            extra["accelerated"] = False
        if rng.random() < 0.3:
            extra["obstacle_order"] = False
        if rng.random() < 0.5:
            extra.update(ro2=float(rng.choice([0.5, 2.0])), slack_gain=float(rng.choice([4, 8, 12])), max_sd=float(rng.choice([0.8, 1.0, 1.5])),
                         min_sd=float(rng.choice([0.05, 0.1, 0.3])))
    else:
        car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
    y = 25.0
    path = sc.line_path([4, y, 0], [4 + 0.4 * steps + 12, y, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    moving = bool(rng.integers(2))
    half = 30 if large else 12
    scene = sc.scene_polygons(N, lo=(6, y - half), hi=(4 + 0.4 * steps + 14, y + half), seed=1000 * seed + s, keep_clear=clear,
                              clear_radius=float(rng.uniform(2.4, 3.4)) * (0.5 if tight else 1.0), moving=moving)
    if kmax > 4:             # every other polygon redrawn with 3 .. E vertices (same centre region, own stream)
        r2 = np.random.default_rng(7000 * seed + s)
        for i in range(0, len(scene), 2):
            o = scene[i]; c = o.vertex.mean(axis=1)
            scene[i] = sc.regular_polygon(c[0], c[1], int(r2.integers(3, kmax + 1)), float(r2.uniform(0.5, 1.0)), float(r2.uniform(-np.pi, np.pi)),
                                          tuple(o.velocity.ravel()))
    if circles:
        scene = [o if i % 3 == 0 else sc.circle(float(o.vertex[0].mean()), float(o.vertex[1].mean()), 0.45 * float(np.ptp(o.vertex[0]) + np.ptp(o.vertex[1])) / 2 + 0.3,
                                                tuple(o.velocity.ravel())) for i, o in enumerate(scene)]
    for _ in range(int(rng.integers(0, 4))):
        scene.append(sc.circle(float(rng.uniform(10, 40)), y + float(rng.choice([-1, 1])) * float(rng.uniform(3.5, 8)),
                               float(rng.uniform(0.4, 1.2)), (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)))))
    kw = dict(receding=T, iter_num=int(rng.integers(2, 5)), max_edge_num=E, max_obs_num=int(rng.integers(max(4, N // 2), N + 6)),
              ro1=float(rng.choice([200, 300])), time_print=False, **extra)
    speed = float(rng.uniform(2.5, 4.5))
    return dict(dyn=dyn, T=T, N=N, car=car_t, path=path, scene=scene, kw=kw, speed=speed, moving=moving)


def run_soak(scenes=12, steps=100, seed=0, lmz_central=0.0, cold_oracle=False, only=-1, threads=None, dump_dir="", dump_tol=1e-5,
             su_dump="", so="", log=print, hip_kw=None, large=False, exotic=False, tight=False, robots=False, circles=False):
    """returns a dict of totals + the per-step outliers; `log` receives one line per remarkable step"""
    lib = orc_api().lib
    if so:
        from rda_planner_amd import _lib
        _lib.SO_PATH = os.path.abspath(so)
    lib.orc_set_su_dump.argtypes = [C.c_char_p]
    lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_set_lmz_ipm_mu.argtypes = [C.c_double]
    if su_dump:
        os.makedirs(os.path.dirname(os.path.abspath(su_dump)), exist_ok=True)
        lib.orc_set_su_dump(os.path.abspath(su_dump).encode())
    if lmz_central > 0:
        lib.orc_set_lmz_mode(1); lib.orc_set_lmz_ipm_mu(lmz_central)
    if cold_oracle:
        lib.orc_set_su_warm(0.0, 0.0, 0)
    lib.orc_set_threads(threads or min(16, os.cpu_count() or 1))      # more threads than that slow the oracle down (bench.py thread sweep)
    if dump_dir:
        os.makedirs(dump_dir, exist_ok=True)
    step_dump = os.path.join(dump_dir, "_step.bin") if dump_dir and not su_dump else ""
    rng = np.random.default_rng(seed)
    out = dict(steps=0, worst_raw=0.0, worst_body=0.0, worst_hor_body=0.0, iter_mismatch=0, failed=0, over_raw=0, outliers=[], flips=[],
               ipm_gpu=0, ipm_cpu=0)
    try:
        for s in range(scenes):
            d = draw_scene(rng, seed, s, steps, large, exotic, lmz_central > 0, tight, robots, circles)
            if only >= 0 and s != only:
                continue
            kw = dict(d["kw"])
            if lmz_central > 0:
                kw["lmz_central"] = lmz_central
            gpu = MPC(d["car"], [p.copy() for p in d["path"]], **kw, **(hip_kw or {}))
            cpu = MPC(d["car"], [p.copy() for p in d["path"]], _backend=oracle_backend, **kw)
            st = d["path"][0].copy().reshape(3, 1)
            if d["dyn"] == "omni":
                st[2, 0] = 0.0
            L = d["car"].wheelbase or 1.0
            tag = f"scene {s} ({d['dyn']} T={d['T']} N={d['N']}{' moving' if d['moving'] else ''})"
            if exotic:
                tag = tag[:-1] + f" E={kw['max_edge_num']} robot={d['car'].cone_type} R={np.shape(d['car'].G)[0]}" + "".join(f" {k}={v}" for k, v in kw.items() if k in ("accelerated", "obstacle_order", "ro2")) + ")"
            for k in range(steps):
                cur = [o if not np.any(o.velocity) else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                                         else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in d["scene"]]
                ug, ig = gpu.control(st.copy(), d["speed"], list(cur))
                if step_dump:
                    lib.orc_set_su_dump(step_dump.encode())
                uc, ic = cpu.control(st.copy(), d["speed"], list(cur))
                if step_dump:
                    lib.orc_set_su_dump(b"")
                out["steps"] += 1
                out["ipm_gpu"] += int(ig["su_ipm_iters"]); out["ipm_cpu"] += int(ic["su_ipm_iters"])
                du = float(np.abs(ug - uc).max())
                db = float(np.abs(body_rates(ug.ravel(), d["dyn"], L) - body_rates(uc.ravel(), d["dyn"], L)).max())
                hg, hc = gpu.cur_vel_array, cpu.cur_vel_array
                dh = max(float(np.abs(body_rates(hg[:, j], d["dyn"], L) - body_rates(hc[:, j], d["dyn"], L)).max()) for j in range(hg.shape[1]))
                rec = dict(scene=s, step=k, dyn=d["dyn"], T=d["T"], N=d["N"], du_raw=du, du_body=db, du_hor_body=dh, iters=(int(ig["iters"]), int(ic["iters"])),
                           ipm=(int(ig["su_ipm_iters"]), int(ic["su_ipm_iters"])), status=(int(ig["status"]), int(ic["status"])),
                           u_gpu=ug.ravel().copy(), u_cpu=uc.ravel().copy())
                if ig["status"] or ic["status"]:
                    out["failed"] += 1
                    log(f"{tag} step {k}: su status gpu {ig['status']} (ipm {ig['su_ipm_iters']}), oracle {ic['status']} (ipm {ic['su_ipm_iters']})")
                if ig["iters"] != ic["iters"]:
                    out["iter_mismatch"] += 1
                    out["flips"].append(rec)
                    log(f"{tag} step {k}: iterations {ig['iters']} vs {ic['iters']}, du raw {du:.2e} body {db:.2e}")
                else:
                    out["worst_raw"] = max(out["worst_raw"], du); out["worst_body"] = max(out["worst_body"], db)
                    out["worst_hor_body"] = max(out["worst_hor_body"], dh)
                    if du > dump_tol:
                        out["over_raw"] += 1
                        out["outliers"].append(rec)
                        log(f"{tag} step {k}: du raw {du:.2e} body {db:.2e} horizon body {dh:.2e} (ipm {ig['su_ipm_iters']} vs {ic['su_ipm_iters']}) "
                            f"u_gpu {ug.ravel()} u_cpu {uc.ravel()}")
                        if step_dump and os.path.exists(step_dump):
                            base = os.path.join(dump_dir, f"seed{seed}_scene{s}_step{k}")
                            os.replace(step_dump, base + ".bin")
                            np.savez(base + ".npz", hor_gpu=hg, hor_cpu=hc, u_gpu=ug, u_cpu=uc, dyn=d["dyn"], T=d["T"], N=d["N"], state=st,
                                     speed=d["speed"], iters=ig["iters"])
                # the oracle continues from the GPU's state
                cpu.rda.set_state(gpu.rda.get_state())
                cpu.cur_vel_array = gpu.cur_vel_array.copy()
                cpu.cur_index = gpu.cur_index
                st = sc.kinematic_step(st, ug, d["car"], 0.1)
                if ig["arrive"]:
                    break
    finally:
        lib.orc_set_su_dump(b"")
        if cold_oracle:
            lib.orc_set_su_warm(1e-3, 1e-3, 30)
        if lmz_central > 0:
            lib.orc_set_lmz_mode(0)
        lib.orc_set_threads(1)
        if step_dump and os.path.exists(step_dump):
            os.remove(step_dump)
    return out
