"""The legs of bench.py beside the headline closed loop: Python-API closed loops, device-resident replay, multi-ego, fleet, the `sizes`
sub-runs and the obstacle-shard leg.  None of them is ever `value` (except the replay in --mode shard).  Split out of bench.py in round 5."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

from .context import ROOT

# CPU affinity of this process when the module is imported - before libgomp is loaded.  With OMP_PROC_BIND=close (bench.py sets it for the cpu_baseline
# leg) libgomp binds the MAIN thread of this process to its first place once the oracle's first parallel region has run; a sub-process started afterwards
# inherits that one-core mask, and its own 16 OpenMP threads then share a core: the cpu_baseline of the `sizes` legs read 5 - 7 x too low in rounds 4 - 5
# (C4: 8.1 steps/s in the size leg, 56 in a stand-alone run of the same command).  The size legs are started with the original mask.
_AFFINITY0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None


def _restore_affinity():
    if _AFFINITY0:
        try:
            os.sched_setaffinity(0, _AFFINITY0)
        except OSError:
            pass

from .workload import build_workload, record_trace


def python_api_closed_loop(ctx, track):
    """the closed loop through the Python `MPC.control` API with the caller-side obstacle pipeline on the device (rda_step_scene)
    and, track=True, MPC.pre_process on the device as well (rda_step_tracked)"""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    args, K, W = ctx.args, ctx.K, ctx.W
    mpc_d = MPC(ctx.car_t, [p.copy() for p in ctx.path], sample_time=0.1, time_print=False, device_track=track, **ctx.kw_rec)
    if not mpc_d.rda.has_scene or (track and not mpc_d.rda.has_track):
        return None
    st = ctx.path[0].copy().reshape(3, 1)
    nd = min(W + K, 100)
    du = 0.0
    t0 = time.perf_counter()
    for k in range(nd):
        cur = ctx.obstacles if not args.moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in ctx.obstacles]
        u, _ = mpc_d.control(st, 4.0, list(cur))
        if not args.moving:
            du = max(du, float(np.abs(u - ctx.trace["u"][k]).max()))
        st = sc.kinematic_step(st, u, ctx.car_t, 0.1)
    return {"steps_per_s": round(nd / (time.perf_counter() - t0), 2), "max_du_vs_host_staging": None if args.moving else du,
            "obstacles_advance_every_tick": bool(args.moving)}


def oversubscribed_shard_run(ctx, mpc_rec):
    """Plumbing run on a box with fewer GPUs than ranks (the 1-GPU test box): the same obstacle shards, but the per-iteration exchange is done by
    the host (rda_shard_get_chunk -> gloo all_gather -> rda_shard_set_chunks) instead of RCCL.  Functional check of the sharded code path, NOT
    a performance number.  Prints its own line."""
    import torch
    from rda_planner_amd.rda_solver import RDA_solver
    from rda_planner_amd.sharded import ShardedRDA
    dist, kw, T, N, K, W, trace = ctx.dist, ctx.kw, ctx.T, ctx.N, ctx.K, ctx.W, ctx.trace

    def all_gather(chunk):
        mine = torch.from_numpy(np.ascontiguousarray(chunk))
        everyone = torch.zeros(ctx.world * mine.numel(), dtype=torch.float64)
        dist.all_gather_into_tensor(everyone, mine)
        return everyone.numpy()
    sv = RDA_solver(T, ctx.car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
    sh = ShardedRDA(sv, ctx.rank, ctx.world, all_gather)
    rl = mpc_rec.convert_rda_obstacle(ctx.obstacles, ctx.path[0].copy().reshape(3, 1), False)
    du, its = 0.0, []
    for k in range(W + K):
        if k == W:
            dist.barrier()
            t0 = time.perf_counter()
        u, info = sh.iterative_solve(trace["nom_s"][k], trace["nom_u"][k], [trace["ref"][k][:, j:j + 1] for j in range(T + 1)],
                                     float(trace["speed"][k]), list(rl))
        du = max(du, float(np.abs(u - trace["u_solver"][k]).max()))
        if k >= W:
            its.append(info["iters"])
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if ctx.rank == 0:
        print(json.dumps({"metric": f"MPC steps/sec (ADMM early stop or iter_num cap), T={T}, N_obs={N}", "value": round(K / float(tt.item()), 3), "unit": "steps/s",
                          "n_gpus": ctx.world, "steps": K, "warmup": W, "ms_per_step": round(float(tt.item()) / K * 1e3, 5), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": f"acker rectangle robot, T={T}, N_obs={N}, obstacles sharded {ctx.world}-way",
                                     "parallelism": f"{ctx.world} ranks OVERSUBSCRIBED on {ctx.ndev} GPU(s): host-driven exchange over gloo, plumbing check only"},
                          "mean_admm_iters": round(float(np.mean(its)), 3), "max_du_vs_unsharded_closed_loop": du}))


def replay_legs(ctx):
    """device-resident replay: the recorded step inputs (fixed slot binding) back-to-back, no per-step synchronisation; an instrumented pass
    (hipEvents around every kernel), an un-instrumented one, and one with a host synchronisation per step.  Returns a dict of everything
    the line needs from it, incl. the handle of the instrumented solver (alive: the caller asks it for the LamMuZ launch form)."""
    from rda_planner_amd._capi import Info, dptr, iptr
    api, args, kw, T, K, W, trace, staged = ctx.api, ctx.args, ctx.kw, ctx.T, ctx.K, ctx.W, ctx.trace, ctx.staged

    def load(sv):
        h_ = sv._be.handle
        assert api.lib.rda_upload_obstacles(h_, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"]) == 0
        assert api.lib.rda_upload_trace(h_, W + K, dptr(trace["nom_s"]), dptr(trace["nom_u"]), dptr(trace["ref"]), dptr(trace["speed"])) == 0
        return h_
    solver = ctx.new_solver()
    h = load(solver)
    for k in range(W):
        api.lib.rda_enqueue_step(h, k)
    api.lib.rda_sync(h); ctx.barrier_all()
    api.lib.rda_timing_reset(h, 1)                       # hipEvents around every kernel of the timed region
    t0 = time.perf_counter()
    for k in range(W, W + K):
        api.lib.rda_enqueue_step(h, k)
    api.lib.rda_sync(h); ctx.barrier_all()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    # per-launch GPU times from the events recorded inside that region, in launch order
    kt = {}
    for which, name in ((0, "k_lammuz"), (1, "k_su")):
        cap = K * kw["iter_num"] + 8
        buf, n = np.zeros(cap), C.c_int(0)
        api.lib.rda_timing_launches(h, which, dptr(buf), cap, C.cast(C.byref(n), C.POINTER(C.c_int)))
        kt[name] = buf[:min(n.value, cap)].copy()
    api.lib.rda_timing_reset(h, 0)
    # un-instrumented pass (events perturb the stream slightly)
    solver2 = ctx.new_solver()
    h2 = load(solver2)
    for k in range(W):
        api.lib.rda_enqueue_step(h2, k)
    api.lib.rda_sync(h2); ctx.barrier_all()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        api.lib.rda_enqueue_step(h2, k)
    api.lib.rda_sync(h2); ctx.barrier_all()
    elapsed2 = ctx.max_over_ranks(time.perf_counter() - t0)
    # the same replay with one host synchronisation per step: how long the host needs to queue a step (all launches of one MPC step)
    # and what a step costs when the device starts from an empty stream - the latency floor of the closed loop
    sync_replay = None
    if not args.size_leg:
        solver3 = ctx.new_solver()
        h3 = load(solver3)
        t_enq, t_tot = [], []
        for k in range(W + K):
            ta = time.perf_counter()
            api.lib.rda_enqueue_step(h3, k)
            tb = time.perf_counter()
            api.lib.rda_sync(h3)
            tc = time.perf_counter()
            if k >= W:
                t_enq.append(tb - ta)
                t_tot.append(tc - ta)
        sync_replay = {"median_ms_per_step": round(float(np.median(t_tot)) * 1e3, 5), "median_host_enqueue_ms": round(float(np.median(t_enq)) * 1e3, 5),
                       "what": "replay with rda_sync after every step: host time to queue one step's launches, and the step latency from an idle stream"}
        del solver3
    # replay must reproduce the recorded closed loop (same inputs, same initial state)
    u_last, s_last, info = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
    api.lib.rda_fetch_result(h2, W + K - 1, dptr(u_last), dptr(s_last), C.byref(info))
    replay_err = float(np.abs(u_last - trace["u_solver"][W + K - 1]).max())
    assert trace["arrived_steps"] == 0, "workload invalid: the robot reached the goal inside the timed region"
    iters = []
    for k in range(W, W + K):
        api.lib.rda_fetch_result(h2, k, None, None, C.byref(info))
        iters.append(info.iters)
    return {"kt": kt, "iters": iters, "elapsed_instrumented": elapsed, "elapsed": elapsed2, "sync_replay": sync_replay, "replay_err": replay_err,
            "lmz_kernel": api.lib.rda_lammuz_kernel(h).decode(), "keep_alive": (solver, solver2)}


def multi_ego(ctx):
    """batched multi-ego on ONE GPU (BASELINE "batched multi-ego", replicas only): M independent handles, one HIP stream each, the same recorded
    step inputs; k_su occupies one CU per ego, so the egos overlap on the device"""
    from rda_planner_amd._capi import Info, dptr, iptr
    from rda_planner_amd.rda_solver import RDA_solver
    api, kw, T, N, K, W, trace, staged = ctx.api, ctx.kw, ctx.T, ctx.N, ctx.K, ctx.W, ctx.trace, ctx.staged
    M, Km = ctx.args.egos, min(K, 100)
    hs = []
    for _ in range(M):
        sm = RDA_solver(T, ctx.car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
        hm = sm._be.handle
        api.lib.rda_upload_obstacles(hm, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"])
        api.lib.rda_upload_trace(hm, W + Km, dptr(trace["nom_s"][:W + Km]), dptr(trace["nom_u"][:W + Km]), dptr(trace["ref"][:W + Km]), dptr(trace["speed"][:W + Km]))
        hs.append((sm, hm))
    for k in range(W):
        for _, hm in hs:
            api.lib.rda_enqueue_step(hm, k)
    for _, hm in hs:
        api.lib.rda_sync(hm)
    t0 = time.perf_counter()
    for k in range(W, W + Km, 10):                    # ten steps per ego per host call, egos interleaved
        for _, hm in hs:
            api.lib.rda_enqueue_range(hm, k, min(k + 10, W + Km))
    for _, hm in hs:
        api.lib.rda_sync(hm)
    el = time.perf_counter() - t0
    um, sm_, info = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
    api.lib.rda_fetch_result(hs[-1][1], W + Km - 1, dptr(um), dptr(sm_), C.byref(info))
    return {"egos": M, "steps_per_ego": Km, "aggregate_steps_per_s": round(M * Km / el, 1),
            "max_du_vs_single": float(np.abs(um - trace["u_solver"][W + Km - 1]).max())}


def fleet(ctx):
    """the same, as a FLEET: one set of launches per ADMM iteration with an ego dimension in the grid (rda_fleet_*): k_su runs one workgroup
    per ego side by side, the k_lammuz grid is egos x N*T/4 workgroups"""
    from rda_planner_amd._capi import Info, dptr, iptr
    from rda_planner_amd.rda_solver import RDA_solver
    api, kw, T, N, K, W, trace, staged = ctx.api, ctx.kw, ctx.T, ctx.N, ctx.K, ctx.W, ctx.trace, ctx.staged
    M, Km = ctx.args.fleet_egos, min(K, 100)
    members = []
    for _ in range(M):
        sm = RDA_solver(T, ctx.car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
        hm = sm._be.handle
        api.lib.rda_upload_obstacles(hm, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"])
        api.lib.rda_upload_trace(hm, W + Km, dptr(trace["nom_s"][:W + Km]), dptr(trace["nom_u"][:W + Km]), dptr(trace["ref"][:W + Km]), dptr(trace["speed"][:W + Km]))
        members.append(sm)
    arr = (C.c_void_p * M)(*[m._be.handle for m in members])
    F = C.c_void_p()
    assert api.fleet_create(arr, M, C.byref(F)) == 0
    api.fleet_enqueue_range(F, 0, W)
    api.fleet_sync(F)
    t0 = time.perf_counter()
    api.fleet_enqueue_range(F, W, W + Km)
    api.fleet_sync(F)
    el = time.perf_counter() - t0
    worst = 0.0
    um, sm_, info = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
    for m in (members[0], members[M // 2], members[-1]):
        api.lib.rda_fetch_result(m._be.handle, W + Km - 1, dptr(um), dptr(sm_), C.byref(info))
        worst = max(worst, float(np.abs(um - trace["u_solver"][W + Km - 1]).max()))
    out = {"egos": M, "steps_per_ego": Km, "aggregate_steps_per_s": round(M * Km / el, 1),
           "ms_per_fleet_step": round(el / Km * 1e3, 4), "max_du_vs_single": worst}
    api.fleet_destroy(F)
    del members
    try:
        out["python_api_closed_loop"] = fleet_python_api(ctx, M)
    except Exception as e:                                  # the line must not depend on this leg
        out["python_api_closed_loop"] = {"error": repr(e)[:200]}
    try:
        out["c_abi_closed_loop"] = fleet_closed_loop(ctx, M)
        out["c_abi_closed_loop"]["member_by_member_resort_ego_steps_per_s"] = fleet_closed_loop(ctx, M, steps=12, warm=4, check=(), resort=1)["ego_steps_per_s"]
        if M % 4 == 0 and M >= 16:
            # the same M egos as FOUR fleets of M / 4, each ticked by its own host thread: one fleet's su launch (one workgroup per member, the launch waits for
            # its slowest member) runs beside the other fleets' LamMuZ grids.  2 and 8 fleets are slower (tools/experiments/fleet_groups.py)
            g4 = fleet_closed_loop(ctx, M, check=(0,), groups=4)
            out["c_abi_closed_loop"]["fleets_4_host_threads"] = {k: g4[k] for k in ("ego_steps_per_s", "ms_per_fleet_tick", "mean_admm_iters", "host_threads",
                                                                                     "max_du_vs_solo_closed_loop")}
    except Exception as e:
        out["c_abi_closed_loop"] = {"error": repr(e)[:200]}
    return out


def fleet_closed_loop(ctx, M, steps=30, warm=6, check=(0, 1), resort=2, groups=1):
    """BASELINE config C5 as a CLOSED LOOP through the C-ABI: M egos, each with its OWN seeded scene (seed + ego), its own path index and state;
    the caller is C (tools/closed_loop_host.c closed_loop_fleet_run): per fleet tick rda_fleet_scene_resort (resort = 2; 1: rda_scene_resort member by member -
    the reference re-sorts each robot's list on every tick, mpc.py:205-206), ONE rda_fleet_step_tracked (every member's pre_process + ADMM loop, one host synchronisation), kinematics of every member in
    C.  Members `check` are also run SOLO through closed_loop_run in the same protocol: the fleet is documented to be bit-identical to solo handles."""
    import ctypes as C
    from rda_planner_amd._capi import dptr, iptr
    api, T, N, kw = ctx.api, ctx.T, ctx.N, ctx.kw
    host = ctx.closed_loop_host()
    n_all = warm + steps
    solvers, states, scenes = [], np.zeros((M, 3)), []

    def stage(sv, e):
        car_t, path, obstacles, _ = build_workload(seed_offset=e, n_obs=N, T=T, n_steps=n_all + 10)
        hh = sv._be.handle
        n_sc, kind, nvert, geom, vel = sv.flatten_scene(list(obstacles))
        kind, nvert = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(nvert, np.int32)
        geom, vel = np.ascontiguousarray(geom, float), np.ascontiguousarray(vel, float)
        P = np.ascontiguousarray(np.hstack(path)[0:3, :].T, dtype=float)
        st = np.ascontiguousarray(path[0], float).ravel()[0:3].copy()
        assert api.upload_path(hh, int(P.shape[0]), dptr(P)) == 0
        assert api.upload_scene(hh, int(n_sc), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(st), 1, None) == 0
        return st, int(P.shape[0])
    plen = np.zeros(M, np.int32)
    for e in range(M):
        sv = ctx.new_solver()
        states[e], plen[e] = stage(sv, e)
        solvers.append(sv)
    # `groups` fleets of M / groups members each, every fleet ticked by its own host thread (the members are independent robots: nothing couples two fleets; the
    # C loop runs without the interpreter lock): one fleet's su launch - the wait for its slowest member - overlaps the other fleets' LamMuZ grids
    assert M % groups == 0
    Mg = M // groups
    L, dyn = float(ctx.car_t.wheelbase or 0.0), {"acker": 0, "diff": 1, "omni": 2}[ctx.car_t.dynamics]
    G = []
    for g in range(groups):
        arr = (C.c_void_p * Mg)(*[sv._be.handle for sv in solvers[g * Mg:(g + 1) * Mg]])
        F = C.c_void_p()
        assert api.fleet_create(arr, Mg, C.byref(F)) == 0
        G.append(dict(arr=arr, F=F, cur=np.zeros(Mg, np.int32), nom_u0=np.zeros((Mg, 2, T)), states=np.ascontiguousarray(states[g * Mg:(g + 1) * Mg]),
                      plen=np.ascontiguousarray(plen[g * Mg:(g + 1) * Mg]), u_log=np.zeros((n_all, Mg, 2)), t_log=np.zeros(n_all),
                      it_log=np.zeros((n_all, Mg), np.int32), ipm_log=np.zeros((n_all, Mg), np.int32), rc=0))

    def go_one(q, k0, n):
        q["rc"] = host.fleet_run(C.byref(host.fleet_api), q["F"], q["arr"], Mg, T, dyn, L, 0.1, 4.0, 0.1, 10, iptr(q["plen"]), resort, k0, n, dptr(q["nom_u0"]),
                                 dptr(q["states"]), iptr(q["cur"]), dptr(q["u_log"][k0:]), dptr(q["t_log"][k0:]), iptr(q["it_log"][k0:]), iptr(q["ipm_log"][k0:]))

    def go(k0, n):
        if groups == 1:
            go_one(G[0], k0, n)
        else:
            import threading
            th = [threading.Thread(target=go_one, args=(q, k0, n)) for q in G]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
        for q in G:
            assert q["rc"] == 0, q["rc"]
            api.fleet_sync(q["F"])
    go(0, warm)
    t0 = time.perf_counter()
    go(warm, steps)
    el = time.perf_counter() - t0
    u_log = np.concatenate([q["u_log"] for q in G], axis=1)
    t_log = np.max([q["t_log"] for q in G], axis=0)
    it_log, ipm_log = np.concatenate([q["it_log"] for q in G], axis=1), np.concatenate([q["ipm_log"] for q in G], axis=1)
    out = {"egos": M, "steps_per_ego": steps, "warmup": warm, "ego_steps_per_s": round(M * steps / el, 1), "ms_per_fleet_tick": round(el / steps * 1e3, 4),
           "median_ms_per_fleet_tick": round(float(np.median(t_log[warm:])) * 1e3, 4), "mean_admm_iters": round(float(it_log[warm:].mean()), 3),
           "su_interior_point_iters_per_ego_step": round(float(ipm_log[warm:].mean()), 2),
           "resort": "rda_fleet_scene_resort (one launch set)" if resort == 2 else "rda_scene_resort member by member",
           "what": "closed loop through the C-ABI, caller in C: per fleet tick the members' scenes re-sorted + ONE rda_fleet_step_tracked (one host "
                   "synchronisation) + every member's kinematics; every member its own seeded scene, re-sorted about its robot on every tick (the headline protocol)"}
    out["host_threads"] = groups
    for q in G:
        api.fleet_destroy(q["F"])
    del solvers
    # solo runs of the same members in the same protocol (closed_loop_run: rda_tracked_begin + rda_scene_resort + rda_tracked_finish)
    worst = 0.0
    for e in check:
        sv = ctx.new_solver()
        st, pl = stage(sv, e)
        scn = host.Scene(0, 0, 1, 0, None, None, None, None, None)
        cur_c = C.c_int32(0)
        ul, tl, il = np.zeros((n_all, 2)), np.zeros(n_all), np.zeros(n_all, np.int32)
        rc = host.run(C.byref(host.api), sv._be.handle, C.byref(scn), T, dyn, L, 0.1, 4.0, 0.1, 10, pl, 0, n_all, dptr(np.zeros((2, T))), dptr(st),
                      C.byref(cur_c), dptr(ul), dptr(tl), iptr(il), None, None)
        assert rc == 0, rc
        worst = max(worst, float(np.abs(ul - u_log[:, e, :]).max()))
    out["max_du_vs_solo_closed_loop"] = worst
    out["members_checked_against_solo"] = list(check)
    return out


def fleet_python_api(ctx, M, steps=24):
    """the same fleet driven END TO END from the Python API (SURVEY 8 f3 'batched for multi-ego'): `Fleet.control` over M `MPC` members - every
    member's pre_process on the device (rda_fleet_step_tracked), all raw scenes flattened in one pass and staged by one call
    (rda_fleet_upload_scenes), one set of launches per ADMM iteration; closed loops (each member applies its control to its own kinematic model)"""
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.fleet import Fleet
    from rda_planner_amd.mpc import MPC
    members = [MPC(ctx.car_t, [p.copy() for p in ctx.path], sample_time=0.1, time_print=False, **ctx.kw) for _ in range(M)]
    fl = Fleet(members)
    states = [ctx.path[0].copy().reshape(3, 1) for _ in range(M)]
    obs = [list(ctx.obstacles) for _ in range(M)]
    its, t0 = [], 0.0
    for k in range(4 + steps):
        if k == 4:
            t0 = time.perf_counter()
        res = fl.control([s.copy() for s in states], 4.0, obs)
        for i in range(M):
            states[i] = sc.kinematic_step(states[i], res[i][0], ctx.car_t, 0.1)
        if k >= 4:
            its.append(res[0][1]["iters"])
    el = time.perf_counter() - t0
    out = {"egos": M, "steps_per_ego": steps, "aggregate_steps_per_s": round(M * steps / el, 1), "ms_per_fleet_step": round(el / steps * 1e3, 3),
           "mean_admm_iters": round(float(np.mean(its)), 3), "one_pass_scene_staging_ticks": int(fl.batched_ticks),
           "what": "Fleet.control: Python objects in, controls out; obstacle_order=True (every member's scene re-sorted on the device every tick)"}
    fl.close()
    return out


def shard_leg(ctx):
    """N > 1, default (replica) mode: the OTHER way to use the node - ONE ego whose obstacles are sharded over the ranks, the north-star scaling
    point (T=20, N_obs=2000): every rank solves the LamMuZ problems of its slots, one in-library ncclAllGather per ADMM iteration replicates
    what the su-problem reads (3 arrays + the reduced sums / masks: DESIGN.md 6), every rank solves the identical su-problem.
    Device-resident replay of a recorded closed loop, barrier + max over ranks like the headline.  All ranks call this at the same point; a
    watchdog bounds it (a collective that never completes must not cost the line)."""
    import torch
    from rda_planner_amd._capi import Info, dptr, iptr
    from rda_planner_amd.rda_solver import RDA_solver
    from rda_planner_amd.sharded import enable_rccl
    api, args, dist, rank, world, K, W = ctx.api, ctx.args, ctx.dist, ctx.rank, ctx.world, ctx.K, ctx.W
    Ns, Ts = args.shard_n_obs, 20
    Ks, Ws = min(K, 40), min(W, 4)
    car_s, path_s, obs_s, kw_s = build_workload(seed_offset=0, n_obs=Ns, T=Ts, n_steps=Ks + Ws)
    tr, stg, _ = record_trace(car_s, path_s, obs_s, dict(kw_s, obstacle_order=False), Ws + Ks)

    def replay(sv):
        hh = sv._be.handle
        assert api.lib.rda_upload_obstacles(hh, stg["n"], dptr(stg["A"]), dptr(stg["b"]), iptr(stg["cone"]), stg["per_t"]) == 0
        assert api.lib.rda_upload_trace(hh, Ws + Ks, dptr(tr["nom_s"]), dptr(tr["nom_u"]), dptr(tr["ref"]), dptr(tr["speed"])) == 0
        for k in range(Ws):
            api.lib.rda_enqueue_step(hh, k)
        api.lib.rda_sync(hh); ctx.barrier_all()
        api.lib.rda_timing_reset(hh, 1)
        t0 = time.perf_counter()
        for k in range(Ws, Ws + Ks):
            assert api.lib.rda_enqueue_step(hh, k) == 0
        api.lib.rda_sync(hh); ctx.barrier_all()
        el = ctx.max_over_ranks(time.perf_counter() - t0)
        per = {}
        for which, name in ((0, "lammuz"), (1, "su"), (2, "gather")):
            buf, n = np.zeros(Ks * kw_s["iter_num"] + 8), C.c_int(0)
            api.lib.rda_timing_launches(hh, which, dptr(buf), buf.size, C.cast(C.byref(n), C.POINTER(C.c_int)))
            per[name] = buf[:min(n.value, buf.size)]
        api.lib.rda_timing_reset(hh, 0)
        u_last, s_last, inf = np.zeros((2, Ts)), np.zeros((3, Ts + 1)), Info()
        api.lib.rda_fetch_result(hh, Ws + Ks - 1, dptr(u_last), dptr(s_last), C.byref(inf))
        its = []
        for k in range(Ws, Ws + Ks):
            api.lib.rda_fetch_result(hh, k, None, None, C.byref(inf)); its.append(inf.iters)
        return el, per, float(np.abs(u_last - tr["u_solver"][Ws + Ks - 1]).max()), float(np.mean(its))
    mk = lambda: RDA_solver(Ts, car_s, kw_s["max_edge_num"], Ns, iter_num=kw_s["iter_num"], step_time=0.1, time_print=False, ro1=kw_s["ro1"])
    el1, per1, err1, _ = replay(mk())                    # every rank alone (unsharded): the one-GPU number of the same workload
    sv = mk()

    def bcast(buf):
        if dist is None:                                 # (--force-shard-leg on one GPU: a one-rank communicator, plumbing only)
            return bytes(buf)
        t = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone().cuda()
        dist.broadcast(t, 0)
        return bytes(t.cpu().numpy().tobytes())
    enable_rccl(sv, rank, world, bcast)
    elP, perP, errP, itsP = replay(sv)
    n_exec = int(round(itsP * Ks))
    ex = lambda v: np.sort(v)[max(v.size - n_exec, 0):] if v.size else v        # the executed launches are the longest ones
    # Amdahl, from THIS run's one-GPU kernel times: every rank still solves the whole su-problem, only the LamMuZ launch shards
    su1, lm1 = float(ex(per1["su"]).mean()) * 1e3, float(ex(per1["lammuz"]).mean()) * 1e3
    gat = float(perP["gather"].mean()) * 1e3 if perP["gather"].size else 0.0
    amdahl = {"one_gpu_us_per_iteration": {"su": round(su1, 2), "lammuz": round(lm1, 2)},
              "bound_speedup_without_exchange": round((su1 + lm1) / (su1 + lm1 / world), 3),
              "bound_speedup_with_measured_gather": round((su1 + lm1) / (su1 + lm1 / world + gat), 3),
              "what": f"(t_su + t_lmz) / (t_su + t_lmz / {world} [+ t_gather]): the su-problem is replicated, not sharded - read the measured speed-up against this"}
    return {"amdahl": amdahl, "workload": f"T={Ts}, N_obs={Ns} static seeded polygons, obstacles sharded {world}-way ({-(-Ns // world)} slots per rank)",
            "steps_per_s": round(Ks / elP, 2), "ms_per_step": round(elP / Ks * 1e3, 4), "mean_admm_iters": round(itsP, 3),
            "unsharded_one_gpu_steps_per_s": round(Ks / el1, 2), "speedup_vs_one_gpu": round(el1 / elP, 3),
            "gather_us_per_iteration": round(float(perP["gather"].mean()) * 1e3, 2) if perP["gather"].size else None,
            "gathers": int(perP["gather"].size), "nccl_comm_count": int(api.lib.rda_shard_comm_count(sv._be.handle)),
            "chunk_bytes_per_rank": int(api.shard_chunk_doubles(sv._be.handle)) * 8,
            "lammuz_us_per_executed_launch": {"one_gpu": round(float(ex(per1["lammuz"]).mean()) * 1e3, 2), "sharded": round(float(ex(perP["lammuz"]).mean()) * 1e3, 2)},
            "su_us_per_executed_launch": {"one_gpu": round(float(ex(per1["su"]).mean()) * 1e3, 2), "sharded": round(float(ex(perP["su"]).mean()) * 1e3, 2)},
            "max_du_vs_recorded_closed_loop": {"one_gpu": err1, "sharded": errP}, "steps": Ks, "warmup": Ws,
            "what": "device-resident replay, barrier + max over ranks; every rank enqueues the same steps, one ncclAllGather per executed ADMM iteration"}


SIZE_LEGS = [("n20_T20", ["--n-obs", "20", "--steps", "40", "--warmup", "10", "--fleet-egos", "0"]),
             ("n2000_T20", ["--n-obs", "2000", "--steps", "30", "--warmup", "8", "--fleet-egos", "0"]),
             ("c4_dynamic_obs_n200_T30_moving", ["--n-obs", "200", "--horizon", "30", "--moving", "--steps", "30", "--warmup", "8", "--fleet-egos", "0"]),
             ("c5_shape_n100_T25_fleet64", ["--n-obs", "100", "--horizon", "25", "--steps", "30", "--warmup", "8", "--fleet-egos", "64"])]


def sizes(budget_s, script):
    """every size the metric names + the moving-obstacle and multi-ego configurations, in the SAME driver-run line: one short sub-run of bench.py
    each (own process: a fresh HIP context per shape; --size-leg keeps the closed loops, the timed replay and one 16-thread cpu_baseline
    sample).  BASELINE.json: N in {20, 200, 2000} at T=20; C4 = 200 moving polygons, T=30; C5 = 64 egos x 100 obstacles, T=25."""
    t_sz, out = time.perf_counter(), {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    for name, extra in SIZE_LEGS:
        left = budget_s - (time.perf_counter() - t_sz)
        if left < 8.0:
            out[name] = {"skipped": f"sizes budget of {budget_s:.0f} s used up"}
            continue
        try:
            pr = subprocess.run([sys.executable, script, "--gpus", "1", "--size-leg", "--cpu-threads", "16"] + extra,
                                capture_output=True, text=True, timeout=left + 20.0, env=env, preexec_fn=_restore_affinity)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith("DETAIL {")]      # the sub-run's full dictionary (its last line is the compact one)
            j = json.loads(line[-1][len("DETAIL "):])
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "median_ms_per_step", "mean_admm_iters", "max_du_vs_python_closed_loop",
                    "second_window", "residuals", "roofline", "roofline_secondary", "cpu_baseline", "multi_ego_fleet")
            e = {k: j.get(k) for k in keep}
            e["workload"] = j["config"]["workload"]
            e["fixed_slot_binding_steps_per_s"] = (j.get("fixed_slot_binding") or {}).get("steps_per_s")
            e["su_hard_warm_off_steps_per_s"] = (j.get("su_hard_warm_off") or {}).get("steps_per_s")
            e["pcie_inclusive_steps_per_s"] = (j.get("pcie_inclusive") or {}).get("steps_per_s")
            e["replay_steps_per_s"] = j["device_resident_replay"]["steps_per_s"]
            if e.get("cpu_baseline"):
                e["cpu_baseline"] = {k: e["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "steps", "sample", "max_du_vs_gpu")}
                e["gpu_over_cpu_port"] = round(j["value"] / e["cpu_baseline"]["value"], 1) if e["cpu_baseline"]["value"] else None
            out[name] = e
        except Exception as ex:                         # the headline must not depend on these legs
            out[name] = {"error": repr(ex)[:300]}
    return out, round(time.perf_counter() - t_sz, 1)
