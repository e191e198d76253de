"""The timed closed loop of bench.py: state in / control out per step through the C-ABI, ONE host synchronisation per step.
Split out of bench.py in round 5; the headline `value` comes from `run(ctx, per_tick_scene=<moving>, ordered=True)`."""
import ctypes as C
import time
from dataclasses import dataclass
from typing import Any

import numpy as np


@dataclass
class Loop:
    elapsed: float               # wall time of the K timed steps (max over ranks)
    times: Any                   # per-step wall times [K]
    du: float                    # max |u - recorded Python closed loop| (0 when not compared)
    iters: list                  # executed ADMM iterations per timed step
    info: Any = None             # [K][3] final resi_dual, resi_pri, interior-point iterations of every timed step (C driver)
    second_window: Any = None    # more steps of the same loop right behind the timed ones (C driver, one rank)
    kernel_ms: Any = None        # timing pass: hipEvent times per launch, {"k_lammuz": [...], "k_su": [...]}
    lmz_kernel: str = ""
    elapsed_per_rank: Any = None # this loop's wall time on every rank (N > 1: the per-rank values of the driver line)


def residual_summary(info, iter_threshold):
    """where the ADMM of the timed steps ENDS (in the reference's default protocol it never meets iter_threshold)"""
    if info is None or not len(info):
        return None
    rd, rp, ipm = info[:, 0], info[:, 1], info[:, 2]
    fin = np.isfinite(rd) & np.isfinite(rp)
    conv = fin & (rd < iter_threshold) & (rp < iter_threshold)
    return {"iter_threshold": iter_threshold, "median_resi_dual": round(float(np.median(rd[fin])), 5) if fin.any() else None,
            "median_resi_pri": round(float(np.median(rp[fin])), 7) if fin.any() else None,
            "max_resi_dual": round(float(rd[fin].max()), 5) if fin.any() else None,
            "steps_below_threshold": int(conv.sum()), "steps": int(len(rd)),
            "su_interior_point_iters_per_step": round(float(ipm.mean()), 3)}


def run(ctx, per_tick_scene, driver="c", car=None, compare=True, ordered=False, timing=False, **solver_kw):
    """state in / control out per step; scene resident in HBM (per_tick_scene False) or handed over from host memory on every tick (True,
    BASELINE.md 2.4 'including H2D of obstacles').  driver "c": the loop is tools/closed_loop_host.c (C-ABI calls and the kinematic model in C,
    nothing of the interpreter between two steps); "python": the same loop written with ctypes / numpy.
    ordered: the reference's default obstacle_order=True - the scene is re-sorted by distance to the robot on EVERY tick and the nearest
    max_obs_num are staged (mpc.py:205-206): per_tick_scene -> rda_upload_scene_async(order = 1), resident scene -> rda_scene_resort (the same
    conversion kernels on the resident raw scene, no copy); compared with the ordered Python closed loop.
    timing: hipEvents around every solver launch of the timed steps (switches the zero-copy hand-over off: a pass of its own, never the one
    `value` comes from)."""
    from rda_planner_amd._capi import Info, dptr, iptr
    api, args, kw, T, K, W = ctx.api, ctx.args, ctx.kw, ctx.T, ctx.K, ctx.W
    path, obstacles = ctx.path, ctx.obstacles
    sv = ctx.new_solver(car, **solver_kw)
    hh = sv._be.handle
    car_l = car or ctx.car_t
    n_sc, kind, nvert, geom, vel = sv.flatten_scene(list(obstacles))
    kind, nvert = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(nvert, np.int32)
    geom, vel = np.ascontiguousarray(geom, float), np.ascontiguousarray(vel, float)
    geom0 = geom.copy()
    P = np.ascontiguousarray(np.hstack(path)[0:3, :].T, dtype=float)
    assert api.upload_path(hh, int(P.shape[0]), dptr(P)) == 0
    state = np.ascontiguousarray(path[0], float).ravel()[0:3].copy()
    out_u, out_s, inf = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
    mi, eh = np.zeros(1, np.int32), np.zeros(1)
    nom_u0 = np.zeros((2, T))
    order = 1 if ordered else int(bool(ctx.kw_rec["obstacle_order"]))
    want_u = None
    if compare and not args.moving:
        want_u = ctx.u_ord if ordered else (np.array([ctx.trace["u"][k].ravel() for k in range(W + K)]) if ctx.trace is not None else None)
    if not per_tick_scene:
        assert api.upload_scene(hh, int(n_sc), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(state), order, None) == 0
    cur, du, its, times = 0, 0.0, [], []
    L, wb = car_l.wheelbase, car_l.dynamics
    t_start = 0.0
    if driver == "c":
        host = ctx.closed_loop_host()
        scn = host.Scene(int(n_sc) if per_tick_scene else 0, int(geom.shape[1]), order, int(bool(args.moving)), iptr(kind), iptr(nvert),
                         dptr(geom), dptr(geom0), dptr(vel))
        cur_c = C.c_int32(0)
        # More steps of the same loop right behind the timed ones, when the path is long enough: the driver's 20-step window after 5 warm-up
        # steps is all start-up (no solver history yet), and the metric's own protocol is the MEDIAN of >= 50 timed steps -
        # so this window has max(K, 50) steps where they fit
        fits = lambda n: 0.4 * (W + K + n) + 8.0 <= ctx.path_length
        K2 = max(K, 50) if fits(max(K, 50)) else (K if fits(K) else 0)
        if ctx.world != 1 or timing:
            K2 = 0
        n_all = W + K + K2
        u_log, t_log, it_log, info_log = np.zeros((n_all, 2)), np.zeros(n_all), np.zeros(n_all, np.int32), np.zeros((n_all, 3))
        dyn = {"acker": 0, "diff": 1, "omni": 2}[wb]

        def go(k0, n):
            rc = host.run(C.byref(host.api), hh, C.byref(scn), T, dyn, float(L or 0.0), 0.1, 4.0, 0.1, 10, len(path), k0, n, dptr(nom_u0),
                          dptr(state), C.byref(cur_c), dptr(u_log[k0:]), dptr(t_log[k0:]), iptr(it_log[k0:]), None, dptr(info_log[k0:]))
            assert rc == 0, ("workload invalid: the robot reached the goal inside the timed region" if rc == 1 else rc)
        go(0, W)
        api.lib.rda_sync(hh)
        ctx.barrier_all()
        if timing:
            api.lib.rda_timing_reset(hh, 1)
        t_start = time.perf_counter()
        go(W, K)
        api.lib.rda_sync(hh)
        ctx.barrier_all()
        el_local = time.perf_counter() - t_start
        el = ctx.max_over_ranks(el_local)
        res = Loop(el, t_log[W:W + K].copy(), 0.0, [int(v) for v in it_log[W:W + K]], info=info_log[W:W + K].copy())
        if ctx.world > 1:
            res.elapsed_per_rank = ctx.gather_over_ranks(el_local)
        if timing:
            kt_ = {}
            for which, name in ((0, "k_lammuz"), (1, "k_su")):
                cap = K * (kw["iter_num"] + 1) + 8
                buf, n_ = np.zeros(cap), C.c_int(0)
                api.lib.rda_timing_launches(hh, which, dptr(buf), cap, C.cast(C.byref(n_), C.POINTER(C.c_int)))
                kt_[name] = buf[:min(n_.value, cap)].copy()
            api.lib.rda_timing_reset(hh, 0)
            res.kernel_ms, res.lmz_kernel = kt_, api.lib.rda_lammuz_kernel(hh).decode()
            return res
        if K2:
            t2 = time.perf_counter()
            go(W + K, K2)
            api.lib.rda_sync(hh)
            el2 = time.perf_counter() - t2
            res.second_window = {"steps": K2, "after_steps": W + K, "steps_per_s": round(K2 / el2, 2),
                                 "median_ms_per_step": round(float(np.median(t_log[W + K:]) * 1e3), 5),
                                 "median_steps_per_s": round(1.0 / float(np.median(t_log[W + K:])), 2),
                                 "mean_admm_iters": round(float(np.mean(it_log[W + K:])), 3),
                                 "residuals": residual_summary(info_log[W + K:], float(kw.get("iter_threshold", 0.2)))}
        if want_u is not None:
            res.du = float(np.abs(u_log[:W + K] - want_u[:W + K]).max())
        return res
    for k in range(W + K):
        if k == W:
            api.lib.rda_sync(hh)
            ctx.barrier_all()
            t_start = time.perf_counter()
        t0 = time.perf_counter()
        nu = dptr(nom_u0) if k == 0 else None          # afterwards the previous controls are resident (MPC.cur_vel_array)
        if per_tick_scene:
            if args.moving:                             # obstacles advance every tick like in the dynamic_obs example
                geom[:, :, :] = geom0 + (vel * (0.1 * k))[:, None, :] * (np.arange(geom.shape[1])[None, :, None] < nvert[:, None, None])
            rc = api.tracked_begin(hh, dptr(state), 4.0, int(cur), 0.1, 10, nu)
            rc |= api.upload_scene_async(hh, int(n_sc), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(state), order)
            rc |= api.tracked_finish(hh, dptr(out_u), dptr(out_s), C.byref(inf), None, None, iptr(mi), dptr(eh))
        elif ordered:
            rc = api.tracked_begin(hh, dptr(state), 4.0, int(cur), 0.1, 10, nu)
            rc |= api.scene_resort(hh, dptr(state))
            rc |= api.tracked_finish(hh, dptr(out_u), dptr(out_s), C.byref(inf), None, None, iptr(mi), dptr(eh))
        else:
            rc = api.step_tracked(hh, dptr(state), 4.0, int(cur), 0.1, 10, nu, dptr(out_u), dptr(out_s), C.byref(inf), None, None,
                                  iptr(mi), dptr(eh))
        assert rc >= 0, rc
        cur = int(mi[0])
        assert cur < len(path) - 1, "workload invalid: the robot reached the goal inside the timed region"
        # the host side of the loop: apply the first control to the kinematic model (what ir-sim's env.step does)
        v, w, phi = float(out_u[0, 0]), float(out_u[1, 0]), float(state[2])
        if wb == "acker":
            state += 0.1 * np.array([v * np.cos(phi), v * np.sin(phi), v * np.tan(w) / L])
        elif wb == "diff":
            state += 0.1 * np.array([v * np.cos(phi), v * np.sin(phi), w])
        else:
            state += 0.1 * np.array([v * np.cos(w), v * np.sin(w), 0.0])
        if k >= W:
            times.append(time.perf_counter() - t0)
            its.append(inf.iters)
        if want_u is not None:
            du = max(du, float(np.abs(out_u[:, 0] - want_u[k]).max()))
    api.lib.rda_sync(hh)
    ctx.barrier_all()
    el = ctx.max_over_ranks(time.perf_counter() - t_start)
    return Loop(el, np.array(times), du, its)
