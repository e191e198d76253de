#!/usr/bin/env python
"""
bench.py - MPC steps/s of the MI355X-native RDA ADMM inner solver (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one RDA_solver.iterative_solve (reference rda_solver.py:573-610): up to iter_num
ADMM iterations with early stop.  Workload (north-star point of BASELINE.json / SURVEY.md 8d):
Ackermann rectangle robot, T=20, N_obs=200 static polygons, E=4, synthetic seeded scene.

Protocol: a closed-loop run (solver in the loop, kinematic robot model) records the inputs of
W+K consecutive MPC steps; obstacles and that trace are then uploaded once, the solver state is
reset, W steps are replayed untimed and EXACTLY K steps are enqueued back-to-back on the device
(no host synchronisation inside or between steps) between two barriers + device synchronisation.
`value` is therefore the device-resident rate; the host-synchronous closed-loop rate (H2D of the
nominal, D2H of the control every step) is reported next to it as `closed_loop_steps_per_s`.

N > 1: one process per GPU, independent ego replicas (BASELINE config "batched multi-ego":
scenario batch sharded, no data-path collective) - weak scaling, value = sum over ranks.
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # the multi-ego leg runs one HIP stream per ego; the default 4 hardware queues serialise them
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_workload(seed_offset=0, n_obs=200, T=20, n_steps=110, moving=False):
    """straight reference path through a seeded field of polygons; long enough that the robot never arrives
    (an arrived robot would make every later step trivial)"""
    from rda_planner_amd import scenarios as sc
    car_t = sc.rectangle_robot(dynamics="acker")
    length = max(40.0, 0.4 * n_steps + 12.0)
    path = sc.line_path([4, 25, 0], [4 + length, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(4 + length - 4, 40), seed=sc.SEED + seed_offset, keep_clear=clear, clear_radius=3.2,
                                  moving=moving)     # moving: velocities U[-1,1]^2 m/s, (A, b) per horizon stage (BASELINE dynamic_obs)
    kw = dict(receding=T, iter_num=4, max_edge_num=4, max_obs_num=n_obs, ro1=200, obstacle_order=True)
    return car_t, path, obstacles, kw


def record_trace(car_t, path, obstacles, kw, n_steps, backend=None, post_init=None):
    """closed loop with the solver in the loop; returns per-step inputs and the staged obstacle arrays"""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    extra = {"_backend": backend} if backend is not None else {}
    # host-side obstacle staging here: the spy below needs the staged arrays for the device-resident replay
    mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, device_obstacles=False, device_track=False, **kw, **extra)
    if post_init is not None:
        post_init(mpc.rda)
    T = kw["receding"]
    state = path[0].copy().reshape(3, 1)
    tr = {"nom_s": [], "nom_u": [], "ref": [], "speed": [], "u": [], "u_solver": []}
    arrived = 0
    orig = mpc.rda.iterative_solve
    staged = {}

    def spy(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k):
        tr["nom_s"].append(np.array(nom_s, float).reshape(3, T + 1))
        tr["nom_u"].append(np.array(nom_u, float).reshape(2, T))
        tr["ref"].append(np.array(np.hstack(ref_states)[0:3, :], float))
        tr["speed"].append(float(ref_speed))
        if not staged:
            n, A, b, cone, per_t = mpc.rda._stage(list(obstacle_list))
            staged.update(n=n, A=A, b=b, cone=cone, per_t=per_t)
        u_sol, info_sol = orig(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k)
        tr["u_solver"].append(np.array(u_sol, float))
        return u_sol, info_sol

    mpc.rda.iterative_solve = spy
    t0 = time.perf_counter()
    min_clear = np.inf
    for _ in range(n_steps):
        # static obstacles + obstacle_order=False semantics for the replay: keep slot binding fixed
        u, info = mpc.control(state, 4.0, list(obstacles))
        tr["u"].append(u.copy())
        arrived += int(info["arrive"])
        state = sc.kinematic_step(state, u, car_t, 0.1)
    dt = time.perf_counter() - t0
    min_clear = sc.clearance(car_t, state, obstacles)
    out = {k: np.ascontiguousarray(np.array(v)) for k, v in tr.items()}
    out["closed_loop_s_per_step"] = dt / n_steps
    out["final_clearance"] = float(min_clear)
    out["arrived_steps"] = arrived
    return out, staged, mpc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 200 timed MPC steps = 80 m of driving
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-obs", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--moving", action="store_true", help="moving obstacles: per-stage (A, b) over the horizon (dynamic_obs config)")
    ap.add_argument("--egos", type=int, default=16, help="extra leg: this many independent egos concurrently on one GPU (0/1 = skip)")
    ap.add_argument("--fleet-egos", type=int, default=64, help="extra leg: this many egos stepped as one fleet (batched launches; 0/1 = skip)")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas",
                    help="N>1: independent ego replicas (default, no collective) or ONE ego whose obstacles are sharded over the ranks "
                         "with an RCCL all-gather per ADMM iteration (strong scaling, --n-obs = total obstacles)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        oversub = ndev < world                     # fewer GPUs than ranks (plumbing test on a 1-GPU box): gloo + shared device
        dev_index = local_rank % max(ndev, 1)
        torch.cuda.set_device(dev_index)
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        tdev = "cpu" if oversub else "cuda"

    from rda_planner_amd._lib import hip_api
    from rda_planner_amd._capi import Info, dptr, iptr
    api = hip_api()
    api.lib.rda_set_device(dev_index if world > 1 else local_rank)

    K, W = args.steps, args.warmup
    shard = args.mode == "shard" and world > 1
    car_t, path, obstacles, kw = build_workload(seed_offset=0 if shard else rank, n_obs=args.n_obs, T=args.horizon, n_steps=K + W, moving=args.moving)

    def make_sharded(solver):
        """obstacle shards + in-library ncclAllGather; the 128-byte unique id travels over torch.distributed"""
        if not shard:
            return
        import torch
        from rda_planner_amd.sharded import enable_rccl

        def bcast(buf):
            t = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone().cuda()
            dist.broadcast(t, 0)
            return bytes(t.cpu().numpy().tobytes())
        enable_rccl(solver, rank, world, bcast)
    T, N = kw["receding"], kw["max_obs_num"]
    # obstacle slots must not be re-sorted between recording and replay: record with the distance
    # order of the first step frozen (static scene), i.e. obstacle_order only affects slot binding
    kw_rec = dict(kw, obstacle_order=False)
    trace, staged, mpc_rec = record_trace(car_t, path, obstacles, kw_rec, W + K, post_init=make_sharded)
    # the same closed loop with the caller-side obstacle pipeline on the device (rda_step_scene)
    cl_dev = cl_trk = None
    if rank == 0 and not shard:
        from rda_planner_amd.mpc import MPC
        from rda_planner_amd import scenarios as sc

        def closed_loop(track):
            mpc_d = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, device_track=track, **kw_rec)
            if not mpc_d.rda.has_scene or (track and not mpc_d.rda.has_track):
                return None
            st = path[0].copy().reshape(3, 1)
            nd = min(W + K, 100)
            du = 0.0
            t0 = time.perf_counter()
            for k in range(nd):
                cur = obstacles if not args.moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
                u, _ = mpc_d.control(st, 4.0, list(cur))
                if not args.moving:
                    du = max(du, float(np.abs(u - trace["u"][k]).max()))
                st = sc.kinematic_step(st, u, car_t, 0.1)
            return {"steps_per_s": round(nd / (time.perf_counter() - t0), 2), "max_du_vs_host_staging": None if args.moving else du,
                    "obstacles_advance_every_tick": bool(args.moving)}
        cl_dev = closed_loop(False)
        # ... and with MPC.pre_process on the device as well (rda_step_tracked): state in, control out
        cl_trk = closed_loop(True)

    # ---- device-resident replay -------------------------------------------------------------------
    from rda_planner_amd.rda_solver import RDA_solver
    solver = RDA_solver(T, car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1,
                        time_print=False, ro1=kw["ro1"])
    make_sharded(solver)
    h = solver._be.handle
    assert api.lib.rda_upload_obstacles(h, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"]) == 0
    assert api.lib.rda_upload_trace(h, W + K, dptr(trace["nom_s"]), dptr(trace["nom_u"]), dptr(trace["ref"]), dptr(trace["speed"])) == 0

    def barrier():
        api.lib.rda_sync(h)
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(W):
        api.lib.rda_enqueue_step(h, k)
    barrier()
    api.lib.rda_timing_reset(h, 1)                       # hipEvents around every kernel of the timed region
    t0 = time.perf_counter()
    for k in range(W, W + K):
        api.lib.rda_enqueue_step(h, k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-kernel GPU time from the events recorded inside the timed region
    kt = {}
    for which, name in ((0, "k_lammuz"), (1, "k_su")):
        ms = C.c_double(0)
        n = C.c_int(0)
        api.lib.rda_timing_read(h, which, C.cast(C.byref(ms), C.POINTER(C.c_double)), C.cast(C.byref(n), C.POINTER(C.c_int)))
        kt[name] = (ms.value, n.value)
    api.lib.rda_timing_reset(h, 0)
    # un-instrumented pass for the headline number (events perturb the stream slightly)
    solver2 = RDA_solver(T, car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
    make_sharded(solver2)
    h2 = solver2._be.handle
    api.lib.rda_upload_obstacles(h2, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"])
    api.lib.rda_upload_trace(h2, W + K, dptr(trace["nom_s"]), dptr(trace["nom_u"]), dptr(trace["ref"]), dptr(trace["speed"]))
    for k in range(W):
        api.lib.rda_enqueue_step(h2, k)
    api.lib.rda_sync(h2)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        api.lib.rda_enqueue_step(h2, k)
    api.lib.rda_sync(h2)
    if dist is not None:
        import torch
        dist.barrier()
        torch.cuda.synchronize()
    elapsed2 = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed2], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed2 = float(tt.item())

    # replay must reproduce the recorded closed loop (same inputs, same initial state)
    u_last = np.zeros((2, T))
    s_last = np.zeros((3, T + 1))
    info = Info()
    api.lib.rda_fetch_result(h2, W + K - 1, dptr(u_last), dptr(s_last), C.byref(info))
    replay_err = float(np.abs(u_last - trace["u_solver"][W + K - 1]).max())
    assert trace["arrived_steps"] == 0, "workload invalid: the robot reached the goal inside the timed region"
    iters = []
    for k in range(W, W + K):
        api.lib.rda_fetch_result(h2, k, None, None, C.byref(info))
        iters.append(info.iters)
    mean_iters = float(np.mean(iters))

    # ---- batched multi-ego on ONE GPU (BASELINE "batched multi-ego", replicas only): M independent handles, one HIP
    #      stream each, the same recorded step inputs; k_su occupies one CU per ego, so the egos overlap on the device
    multi = None
    if rank == 0 and world == 1 and args.egos > 1:
        M, Km = args.egos, min(K, 100)
        hs = []
        for _ in range(M):
            sm = RDA_solver(T, car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
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
        um = np.zeros((2, T)); sm_ = np.zeros((3, T + 1))
        api.lib.rda_fetch_result(hs[-1][1], W + Km - 1, dptr(um), dptr(sm_), C.byref(info))
        multi = {"egos": M, "steps_per_ego": Km, "aggregate_steps_per_s": round(M * Km / el, 1),
                 "max_du_vs_single": float(np.abs(um - trace["u_solver"][W + Km - 1]).max())}
        del hs

    # ---- the same, as a FLEET: one set of launches per ADMM iteration with an ego dimension in the grid (rda_fleet_*):
    #      k_su runs one workgroup per ego side by side, the k_lammuz grid is egos x N*T/4 workgroups
    fleet = None
    if rank == 0 and world == 1 and args.fleet_egos > 1 and getattr(api, "has_fleet", False):
        M, Km = args.fleet_egos, min(K, 100)
        members = []
        for _ in range(M):
            sm = RDA_solver(T, car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
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
        um = np.zeros((2, T)); sm_ = np.zeros((3, T + 1))
        for m in (members[0], members[M // 2], members[-1]):
            api.lib.rda_fetch_result(m._be.handle, W + Km - 1, dptr(um), dptr(sm_), C.byref(info))
            worst = max(worst, float(np.abs(um - trace["u_solver"][W + Km - 1]).max()))
        fleet = {"egos": M, "steps_per_ego": Km, "aggregate_steps_per_s": round(M * Km / el, 1),
                 "ms_per_fleet_step": round(el / Km * 1e3, 4), "max_du_vs_single": worst}
        api.fleet_destroy(F)
        del members

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    E, R = kw["max_edge_num"], 4
    unit_bytes = 8 * (5 * E + 2 * R + 8)                 # 288 B per (obstacle, stage) at E=R=4
    lm_ms, lm_n = kt["k_lammuz"]
    su_ms, su_n = kt["k_su"]
    lm_avg = lm_ms / max(lm_n, 1) * 1e-3
    su_avg = su_ms / max(su_n, 1) * 1e-3
    peak = 8000.0

    def roof(name, avg_s, bytes_per_launch, launches, total_ms):
        ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 3), "peak": peak, "unit": "GB/s",
                "frac": round(ach / peak, 6), "traffic": None, "avg_launch_us": round(avg_s * 1e6, 2),
                "launches": launches, "total_ms": round(total_ms, 3), "algorithmic_bytes_per_launch": bytes_per_launch}
    # the LamMuZ kernel that was actually launched (rda_hip.hip launch_lammuz): packed rows when E+R+1 <= 16, the two-workgroup
    # build above 256 workgroups; RDA_LMZ_ROWS=0 selects the one-sub-problem-per-wave kernel
    lm_kernel = "k_lammuz"
    if E + R + 1 <= 16 and os.environ.get("RDA_LMZ_ROWS", "1") != "0":
        n_loc = N // world if shard else N
        lm_kernel = "k_lammuz_rows_dense" if (n_loc * T + 15) // 16 > int(os.environ.get("RDA_LMZ_DENSE_FROM", "256")) else "k_lammuz_rows"
    r_lm = roof(lm_kernel, lm_avg, unit_bytes * N * T, lm_n, lm_ms)
    r_su = roof(f"k_su<{T}>" if T in (10, 20, 25, 30) else "k_su<0>", su_avg, 48 * N * T + 8 * (8 * (T + 1) + 5 * T), su_n, su_ms)
    tr_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_file):
        try:
            tj = json.load(open(tr_file))
            if tj.get("workload") == {"n_obs": N, "horizon": T}:       # PMC bytes are per launch of THIS workload only
                r_lm["traffic"] = tj.get("k_lammuz")
                r_su["traffic"] = tj.get("k_su")
        except Exception:
            pass
    dominant, secondary = (r_su, r_lm) if su_ms >= lm_ms else (r_lm, r_su)

    kind_word = "moving" if args.moving else "static"
    out = {
        "metric": f"MPC steps/sec (ADMM-converged), T={T}, N_obs={N}", "value": round(K * (1 if shard else world) / elapsed2, 3), "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed2 / K * 1e3, 5),
        "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"north-star: acker rectangle robot, T={T}, N_obs={N} {kind_word} seeded polygons, E={E}, iter_num={kw['iter_num']}, iter_threshold=0.2, ro1={kw['ro1']}",
                   "parallelism": "single GPU" if world == 1 else (f"one ego, obstacles sharded {world}-way, RCCL all-gather per ADMM iteration" if shard else f"{world} independent ego replicas (no collective)")},
        "mean_admm_iters": round(mean_iters, 3), "replay_vs_closed_loop_max_du": replay_err,
        "closed_loop_steps_per_s": round(1.0 / trace["closed_loop_s_per_step"], 2),
        "closed_loop_device_obstacles": cl_dev,
        "closed_loop_device_resident": cl_trk,
        "multi_ego_one_gpu": multi,
        "multi_ego_fleet": fleet,
        "instrumented_ms_per_step": round(elapsed / K * 1e3, 5),
        "roofline": dominant, "roofline_secondary": secondary,
    }

    if not args.no_cpu_baseline and world == 1:
        from oracle.oracle_backend import oracle_backend, api as orc_api
        ncore = os.cpu_count() or 1
        orc_api().lib.orc_set_threads(ncore)
        cpu = RDA_solver(T, car_t, E, N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"], _backend=oracle_backend)
        info_c = Info()
        ou = np.zeros((2, T))
        os_ = np.zeros((3, T + 1))
        n_cpu, budget, t_cpu = 0, 15.0, 0.0
        err = 0.0
        while n_cpu < W + K and (t_cpu < budget or n_cpu < 3):
            k = n_cpu
            t1 = time.perf_counter()
            cpu._be.api.step(cpu._be.handle, dptr(trace["nom_s"][k]), dptr(trace["nom_u"][k]), dptr(trace["ref"][k]), float(trace["speed"][k]),
                             staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"], dptr(ou), dptr(os_), C.byref(info_c))
            t_cpu += time.perf_counter() - t1
            err = max(err, float(np.abs(ou - trace["u_solver"][k]).max()))
            n_cpu += 1
        out["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 3), "unit": "steps/s", "cores": ncore, "kind": "port",
                               "sample": f"first {n_cpu} steps of the same recorded trace (oracle/rda_oracle.c, OpenMP over obstacles; su-problem serial)",
                               "max_du_vs_gpu": err}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
