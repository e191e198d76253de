#!/usr/bin/env python
"""
bench.py - MPC steps/s of the MI355X-native RDA ADMM inner solver (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one RDA_solver.iterative_solve (reference rda_solver.py:573-610): up to iter_num
ADMM iterations with early stop.  Workload (north-star point of BASELINE.json / SURVEY.md 8d):
Ackermann rectangle robot, T=20, N_obs=200 static polygons, E=4, synthetic seeded scene.

Protocol: closed loop, W warm-up steps, then EXACTLY K steps, each one a C-ABI call that takes the
robot state and returns the control with ONE host synchronisation at its end (`rda_step_tracked`: MPC.pre_process, the
ADMM loop and the D2H of the control on the device; the host applies the control to the kinematic model and calls
again).  The caller of those K steps is C (tools/closed_loop_host.c, loaded here): C-ABI calls and the kinematic model only, no
interpreter objects between two steps; `python_caller_closed_loop` is the same loop written with ctypes / numpy.
The obstacle scene is resident in HBM when the timed region starts (static scene: staged once with
`rda_upload_scene`).  `value` = K / wall time of those K steps (max over ranks); `median_ms_per_step` is the median of
the K per-step wall times.  Reported beside it, never as `value`:
  * `pcie_inclusive`        - the same loop with the raw scene (vertices, velocities) handed over from host memory on
                              every tick (`rda_tracked_begin` + `rda_upload_scene_async` + `rda_tracked_finish`);
  * `device_resident_replay`- the recorded step inputs replayed back-to-back with no per-step synchronisation
                              (throughput of the device pipeline, what round 1 reported as the headline);
  * the same closed loop through the Python `MPC.control` API with host / device obstacle staging.
A Python closed loop first records the W+K step inputs; every other leg must reproduce its controls.

N > 1: one process per GPU, independent ego replicas (BASELINE config "batched multi-ego":
scenario batch sharded, no data-path collective) - weak scaling, value = sum over ranks.
"""
import os
os.environ.setdefault("OMP_PROC_BIND", "close")   # cpu_baseline leg: keep the oracle's OpenMP threads on neighbouring cores (read when libgomp loads)
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


def record_trace(car_t, path, obstacles, kw, n_steps, backend=None, post_init=None, stage_every_step=False, moving=False):
    """closed loop with the solver in the loop; returns per-step inputs and the staged obstacle arrays (of the first step, or - for
    a scene that is re-sorted every tick - of every step: trace["staged"])"""
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
    per_step = []

    def spy(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k):
        tr["nom_s"].append(np.array(nom_s, float).reshape(3, T + 1))
        tr["nom_u"].append(np.array(nom_u, float).reshape(2, T))
        tr["ref"].append(np.array(np.hstack(ref_states)[0:3, :], float))
        tr["speed"].append(float(ref_speed))
        if not staged or stage_every_step:
            n, A, b, cone, per_t = mpc.rda._stage(list(obstacle_list))
            if not staged:
                staged.update(n=n, A=A, b=b, cone=cone, per_t=per_t)
            if stage_every_step:
                per_step.append((n, A.copy(), b.copy(), cone.copy(), per_t))
        u_sol, info_sol = orig(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k)
        tr["u_solver"].append(np.array(u_sol, float))
        return u_sol, info_sol

    mpc.rda.iterative_solve = spy
    t0 = time.perf_counter()
    min_clear = np.inf
    for k_ in range(n_steps):
        # static obstacles + obstacle_order=False semantics for the replay: keep slot binding fixed
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k_)) for o in obstacles]
        u, info = mpc.control(state, 4.0, list(cur))
        tr["u"].append(u.copy())
        arrived += int(info["arrive"])
        state = sc.kinematic_step(state, u, car_t, 0.1)
    dt = time.perf_counter() - t0
    min_clear = sc.clearance(car_t, state, obstacles)
    out = {k: np.ascontiguousarray(np.array(v)) for k, v in tr.items()}
    out["closed_loop_s_per_step"] = dt / n_steps
    out["final_clearance"] = float(min_clear)
    out["arrived_steps"] = arrived
    out["staged"] = per_step
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
    ap.add_argument("--no-ip-legs", action="store_true", help="skip the interior-point LamMuZ closed loops (extra keys)")
    ap.add_argument("--no-shard-leg", action="store_true", help="N > 1, replica mode: skip the extra obstacle-shard leg (one ego, N_obs = --shard-n-obs)")
    ap.add_argument("--force-shard-leg", action="store_true", help="run the obstacle-shard leg on ONE GPU with a one-rank communicator (plumbing check)")
    ap.add_argument("--shard-n-obs", type=int, default=2000)
    ap.add_argument("--shard-leg-timeout", type=float, default=120.0)
    ap.add_argument("--no-sizes", action="store_true", help="skip the `sizes` legs (N=20, N=2000, C4, C5 shape: one short sub-run of this script each)")
    ap.add_argument("--sizes-budget-s", type=float, default=75.0, help="wall-clock budget of the `sizes` legs; a leg that would start after it is skipped (and says so)")
    ap.add_argument("--size-leg", action="store_true", help="(internal) reduced set of legs: what one entry of `sizes` needs")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cpu_baseline with this one thread count instead of the sweep")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas",
                    help="N>1: independent ego replicas (default, no collective) or ONE ego whose obstacles are sharded over the ranks "
                         "with an RCCL all-gather per ADMM iteration (strong scaling, --n-obs = total obstacles)")
    args = ap.parse_args()
    if args.size_leg:                                   # one entry of `sizes`: the closed loops, the timed replay, one cpu_baseline sample
        args.egos, args.no_ip_legs, args.no_sizes = 0, True, True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    oversub = False
    if args.force_shard_leg:
        import torch                               # torch (its HIP runtime, its RCCL) must be in the process BEFORE librda_hip.so is loaded
        torch.cuda.init()
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
    path_length = max(40.0, 0.4 * (K + W) + 12.0)

    def make_sharded(solver):
        """obstacle shards + in-library ncclAllGather; the 128-byte unique id travels over torch.distributed"""
        if not shard or oversub:            # ranks sharing one GPU cannot form an RCCL communicator: host-driven exchange below
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
    u_ord = None
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
        if not args.size_leg:
            cl_dev = closed_loop(False)
            # ... and with MPC.pre_process on the device as well (rda_step_tracked): state in, control out
            cl_trk = closed_loop(True)
        # the reference's default: obstacle_order=True, the list re-sorted by distance on EVERY tick (mpc.py:205-206).  That closed loop
        # through the Python API with host-side staging: the controls the ordered C-ABI legs must reproduce, and the staged slots of
        # every step for the cpu_baseline leg (the oracle is timed on the SAME ordered workload)
        trace_o, _, _ = record_trace(car_t, path, obstacles, kw, W + K, stage_every_step=True, moving=args.moving)
        if not args.moving:
            u_ord = np.array([u.ravel() for u in trace_o["u"]])
    else:
        trace_o = None

    from rda_planner_amd.rda_solver import RDA_solver
    from rda_planner_amd import scenarios as sc

    if shard and oversub:
        # Plumbing run on a box with fewer GPUs than ranks (the 1-GPU test box): the same obstacle shards, but the per-iteration
        # exchange is done by the host (rda_shard_get_chunk -> gloo all_gather -> rda_shard_set_chunks) instead of RCCL.
        # Functional check of the sharded code path, NOT a performance number.
        import torch
        from rda_planner_amd.sharded import ShardedRDA

        def all_gather(chunk):
            mine = torch.from_numpy(np.ascontiguousarray(chunk))
            everyone = torch.zeros(world * mine.numel(), dtype=torch.float64)
            dist.all_gather_into_tensor(everyone, mine)
            return everyone.numpy()
        sv = RDA_solver(T, car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"])
        sh = ShardedRDA(sv, rank, world, all_gather)
        rl = mpc_rec.convert_rda_obstacle(obstacles, path[0].copy().reshape(3, 1), False)
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
        if rank == 0:
            print(json.dumps({"metric": f"MPC steps/sec (ADMM-converged), T={T}, N_obs={N}", "value": round(K / float(tt.item()), 3), "unit": "steps/s",
                              "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(float(tt.item()) / K * 1e3, 5), "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": f"acker rectangle robot, T={T}, N_obs={N}, obstacles sharded {world}-way",
                                         "parallelism": f"{world} ranks OVERSUBSCRIBED on {ndev} GPU(s): host-driven exchange over gloo, plumbing check only"},
                              "mean_admm_iters": round(float(np.mean(its)), 3), "max_du_vs_unsharded_closed_loop": du}))
        dist.destroy_process_group()
        return

    def barrier_all():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        tt = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def closed_loop_host():
        """tools/libclosed_loop_host.so: the caller's loop in C, entry points of librda_hip.so handed over (tools/closed_loop_host.py)"""
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import closed_loop_host as clh
        return clh.Host(api.lib)

    def new_solver(car=None, **extra):
        sv = RDA_solver(T, car or car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"], **extra)
        make_sharded(sv)
        return sv

    # ---- HEADLINE: closed loop through the C-ABI, one host synchronisation per MPC step -------------------------------
    def cabi_closed_loop(per_tick_scene, driver="c", car=None, compare=True, ordered=False, timing=False, **solver_kw):
        """state in / control out per step; scene resident in HBM (per_tick_scene False) or handed over from host memory on every
        tick (True, BASELINE.md 2.4 'including H2D of obstacles').  driver "c": the loop is tools/closed_loop_host.c (C-ABI calls and the
        kinematic model in C, nothing of the interpreter between two steps); "python": the same loop written with ctypes / numpy.
        ordered: the reference's default obstacle_order=True - the scene is re-sorted by distance to the robot on EVERY tick and the nearest
        max_obs_num are staged (mpc.py:205-206): per_tick_scene -> rda_upload_scene_async(order = 1), resident scene -> rda_scene_resort (the
        same conversion kernels on the resident raw scene, no copy); compared with the ordered Python closed loop.
        Returns (elapsed of the K timed steps, per-step times, max |u - recorded Python closed loop|, iterations per step)."""
        sv = new_solver(car, **solver_kw)
        hh = sv._be.handle
        car_l = car or car_t
        scene = sv.flatten_scene(list(obstacles))
        n_sc, kind, nvert, geom, vel = scene
        kind, nvert = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(nvert, np.int32)
        geom, vel = np.ascontiguousarray(geom, float), np.ascontiguousarray(vel, float)
        geom0 = geom.copy()
        P = np.ascontiguousarray(np.hstack(path)[0:3, :].T, dtype=float)
        assert api.upload_path(hh, int(P.shape[0]), dptr(P)) == 0
        state = np.ascontiguousarray(path[0], float).ravel()[0:3].copy()
        out_u, out_s, inf = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
        mi, eh = np.zeros(1, np.int32), np.zeros(1)
        nom_u0 = np.zeros((2, T))
        order = 1 if ordered else int(bool(kw_rec["obstacle_order"]))
        want_u = (u_ord if ordered else np.array([trace["u"][k].ravel() for k in range(W + K)])) if (compare and not args.moving) else None
        if ordered and want_u is None and not args.moving:
            compare = False
        if not per_tick_scene:
            assert api.upload_scene(hh, int(n_sc), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(state), order, None) == 0
        cur, du, its, times = 0, 0.0, [], []
        L, wb = car_l.wheelbase, car_l.dynamics
        t_start = 0.0
        if driver == "c":
            host = closed_loop_host()
            scn = host.Scene(int(n_sc) if per_tick_scene else 0, int(geom.shape[1]), order, int(bool(args.moving)), iptr(kind), iptr(nvert),
                             dptr(geom), dptr(geom0), dptr(vel))
            cur_c = C.c_int32(0)
            # a second window of K steps right behind the timed one, when the path is long enough (the driver's 20-step window after 5
            # warm-up steps is all start-up: no solver history yet, 2.0 instead of ~2.6 ADMM iterations per step)
            K2 = K if 0.4 * (W + 2 * K) + 8.0 <= path_length else 0
            u_log, t_log, it_log = np.zeros((W + K + K2, 2)), np.zeros(W + K + K2), np.zeros(W + K + K2, np.int32)
            dyn = {"acker": 0, "diff": 1, "omni": 2}[wb]

            def run(k0, n):
                rc = host.run(C.byref(host.api), hh, C.byref(scn), T, dyn, float(L or 0.0), 0.1, 4.0, 0.1, 10, len(path), k0, n, dptr(nom_u0),
                              dptr(state), C.byref(cur_c), dptr(u_log[k0:]), dptr(t_log[k0:]), iptr(it_log[k0:]), None)
                assert rc == 0, ("workload invalid: the robot reached the goal inside the timed region" if rc == 1 else rc)
            run(0, W)
            api.lib.rda_sync(hh)
            barrier_all()
            if timing:                                   # hipEvents around every solver launch of the timed steps (switches the zero-copy hand-over off:
                api.lib.rda_timing_reset(hh, 1)          # a pass of its own, never the one `value` comes from)
            t_start = time.perf_counter()
            run(W, K)
            api.lib.rda_sync(hh)
            barrier_all()
            el = max_over_ranks(time.perf_counter() - t_start)
            cabi_closed_loop.second_window = None
            if timing:
                kt_ = {}
                for which, name in ((0, "k_lammuz"), (1, "k_su")):
                    cap = K * (kw["iter_num"] + 1) + 8
                    buf, n_ = np.zeros(cap), C.c_int(0)
                    api.lib.rda_timing_launches(hh, which, dptr(buf), cap, C.cast(C.byref(n_), C.POINTER(C.c_int)))
                    kt_[name] = buf[:min(n_.value, cap)].copy()
                api.lib.rda_timing_reset(hh, 0)
                cabi_closed_loop.kernel_ms = kt_
                cabi_closed_loop.lmz_kernel = api.lib.rda_lammuz_kernel(hh).decode()
                return el, t_log[W:W + K].copy(), 0.0, [int(v) for v in it_log[W:W + K]]
            if K2 and world == 1:
                t2 = time.perf_counter()
                run(W + K, K2)
                api.lib.rda_sync(hh)
                el2 = time.perf_counter() - t2
                cabi_closed_loop.second_window = {"steps": K2, "after_steps": W + K, "steps_per_s": round(K2 / el2, 2),
                                                  "median_ms_per_step": round(float(np.median(t_log[W + K:]) * 1e3), 5),
                                                  "mean_admm_iters": round(float(np.mean(it_log[W + K:])), 3)}
            if want_u is not None:
                du = float(np.abs(u_log[:W + K] - want_u[:W + K]).max())
            return el, t_log[W:W + K].copy(), du, [int(v) for v in it_log[W:W + K]]
        for k in range(W + K):
            if k == W:
                api.lib.rda_sync(hh)
                barrier_all()
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
        barrier_all()
        el = max_over_ranks(time.perf_counter() - t_start)
        return el, np.array(times), du, its

    head = None
    if not shard:
        # HEADLINE = the reference's default semantics: obstacle_order=True, the scene re-sorted about the robot on every tick
        el_h, times_h, du_h, its_h = cabi_closed_loop(per_tick_scene=bool(args.moving), ordered=True)
        head = {"elapsed": el_h, "median_ms": float(np.median(times_h) * 1e3), "du": du_h, "iters": its_h, "second_window": cabi_closed_loop.second_window}
        pcie = None
        pydrv = None
        fixed = None
        if rank == 0 and world == 1:
            # the protocol of rounds 2-3: slots bound once (obstacle_order=False), no conversion kernel inside the timed region
            el_f, times_f, du_f, its_f = cabi_closed_loop(per_tick_scene=bool(args.moving))
            fixed = {"steps_per_s": round(K / el_f, 2), "median_ms_per_step": round(float(np.median(times_f) * 1e3), 5),
                     "mean_admm_iters": round(float(np.mean(its_f)), 3), "max_du_vs_python_closed_loop": du_f if not args.moving else None,
                     "second_window": cabi_closed_loop.second_window,
                     "what": "obstacle_order=False: slots bound once at staging, rda_step_tracked per step (the headline protocol of rounds 2-3; the reference's default re-sorts every tick)"}
        follow = None
        if rank == 0 and world == 1:
            # NOT the reference's semantics (opt-in, rda_opts::duals_follow): the headline loop - scene re-sorted every tick - with the duals
            # moving WITH their obstacles through the re-binding instead of staying with the slot (quirk Q5)
            try:
                el_w, times_w, _, its_w = cabi_closed_loop(per_tick_scene=bool(args.moving), ordered=True, compare=False, duals_follow_obstacles=True)
                follow = {"steps_per_s": round(K / el_w, 2), "median_ms_per_step": round(float(np.median(times_w) * 1e3), 5),
                          "mean_admm_iters": round(float(np.mean(its_w)), 3), "second_window": cabi_closed_loop.second_window,
                          "what": "the headline protocol (obstacle_order=True, re-sorted on the device every tick) with duals_follow_obstacles=True: "
                                  "an extension, NOT reference semantics - never `value`"}
            except (AssertionError, RuntimeError) as e:          # the headline must not depend on an opt-in leg
                follow = {"error": str(e)[:200]}
        early = None
        hardw = None
        if rank == 0 and world == 1 and not args.size_leg:
            # opt-in rda_opts::su_tol_early: the su-problems before the last ADMM iteration of a step at the reference solver's own class of
            # tolerance (ECOS defaults, 1e-8) instead of the 1000 x tighter su_tol the parity tolerance is stated against
            from rda_planner_amd.rda_solver import hip_options as _ho
            try:
                el_e, times_e, _, its_e = cabi_closed_loop(per_tick_scene=bool(args.moving), ordered=True, compare=False, hip_opts=_ho(su_tol_early=(1e-6, 1e-7, 1e-8)))
                early = {"steps_per_s": round(K / el_e, 2), "median_ms_per_step": round(float(np.median(times_e) * 1e3), 5),
                         "mean_admm_iters": round(float(np.mean(its_e)), 3),
                         "what": "the headline protocol with su_tol_early = (1e-6, 1e-7, 1e-8): opt-in, the stated parity tolerance does not hold with it - never `value`"}
            except (AssertionError, RuntimeError) as e:
                early = {"error": str(e)[:200]}
            # opt-in rda_opts::su_hard_warm (found at the end of round 4, default off until it has been soaked): the warm attempts of a step that
            # follows an UNCONVERGED step start well inside the boxes (slack floor 1) with the previous multipliers and mu0 = 1e-3
            try:
                el_g, times_g, _, its_g = cabi_closed_loop(per_tick_scene=bool(args.moving), ordered=True, compare=False, hip_opts=_ho(su_hard_warm=(1.0, 1e-3)))
                hardw = {"steps_per_s": round(K / el_g, 2), "median_ms_per_step": round(float(np.median(times_g) * 1e3), 5),
                         "mean_admm_iters": round(float(np.mean(its_g)), 3),
                         "what": "the headline protocol with su_hard_warm = (1, 1e-3): same su-problems solved to the same tolerance from another start; "
                                 "opt-in until validated by the soak - never `value`"}
            except (AssertionError, RuntimeError) as e:
                hardw = {"error": str(e)[:200]}
        if rank == 0 and world == 1 and not args.size_leg:
            el_y, times_y, du_y, _ = cabi_closed_loop(per_tick_scene=bool(args.moving), driver="python", ordered=True)
            pydrv = {"steps_per_s": round(K / el_y, 2), "median_ms_per_step": round(float(np.median(times_y) * 1e3), 5),
                     "max_du_vs_python_closed_loop": du_y if not args.moving else None,
                     "what": "the headline loop with the caller written in Python (ctypes calls + numpy kinematics between two steps)"}
        if rank == 0 and not args.moving:
            el_p, times_p, du_p, _ = cabi_closed_loop(per_tick_scene=True, ordered=True) if world == 1 else (None, None, None, None)
            if el_p is not None:
                pcie = {"steps_per_s": round(K / el_p, 2), "median_ms_per_step": round(float(np.median(times_p) * 1e3), 5),
                        "max_du_vs_python_closed_loop": du_p,
                        "what": "raw scene (vertices, velocities) handed over from host memory AND re-sorted every tick: rda_tracked_begin + rda_upload_scene_async(order=1) + rda_tracked_finish"}

    # per-launch GPU times of the HEADLINE loop (ordered): one more pass of the same closed loop with hipEvents around every launch
    head_kt = None
    if head is not None and rank == 0 and world == 1:
        _, _, _, its_t = cabi_closed_loop(per_tick_scene=bool(args.moving), ordered=True, timing=True, compare=False)
        head_kt = {"kt": cabi_closed_loop.kernel_ms, "lmz_kernel": cabi_closed_loop.lmz_kernel, "n_exec": int(np.sum(its_t))}

    # ---- interior-point LamMuZ mode (row-parallel kernel k_lammuz_ip): the robust setting lmz_central = 1e-3 on the headline scene, and a
    #      CIRCLE robot (norm2 robot cone, rda_solver.py:1034-1039: always this mode).  Same closed-loop protocol as the headline.
    ip_legs = None
    if rank == 0 and world == 1 and not args.moving and not args.no_ip_legs:
        ip_legs = {}
        try:
            el_i, times_i, _, its_i = cabi_closed_loop(per_tick_scene=False, compare=False, lmz_central=1e-3)
            ip_legs["rectangle_robot_lmz_central_1e-3"] = {"steps_per_s": round(K / el_i, 2), "median_ms_per_step": round(float(np.median(times_i) * 1e3), 5),
                                                           "mean_admm_iters": round(float(np.mean(its_i)), 3)}
            from rda_planner_amd import scenarios as sc_
            circ = sc_.circle_robot(radius=0.8, dynamics="diff")
            el_c, times_c, _, its_c = cabi_closed_loop(per_tick_scene=False, compare=False, car=circ)
            ip_legs["circle_robot_norm2_cone"] = {"steps_per_s": round(K / el_c, 2), "median_ms_per_step": round(float(np.median(times_c) * 1e3), 5),
                                                  "mean_admm_iters": round(float(np.mean(its_c)), 3)}
        except (AssertionError, RuntimeError) as e:          # (a robot that reaches the goal inside the timed region, ...)
            ip_legs["error"] = str(e)

    # ---- device-resident replay: the recorded step inputs back-to-back, no per-step synchronisation ---------------------
    solver = new_solver()
    h = solver._be.handle
    assert api.lib.rda_upload_obstacles(h, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"]) == 0
    assert api.lib.rda_upload_trace(h, W + K, dptr(trace["nom_s"]), dptr(trace["nom_u"]), dptr(trace["ref"]), dptr(trace["speed"])) == 0

    def barrier():
        api.lib.rda_sync(h)
        barrier_all()

    for k in range(W):
        api.lib.rda_enqueue_step(h, k)
    barrier()
    api.lib.rda_timing_reset(h, 1)                       # hipEvents around every kernel of the timed region
    t0 = time.perf_counter()
    for k in range(W, W + K):
        api.lib.rda_enqueue_step(h, k)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # per-launch GPU times from the events recorded inside that region, in launch order
    kt = {}
    for which, name in ((0, "k_lammuz"), (1, "k_su")):
        cap = K * kw["iter_num"] + 8
        buf = np.zeros(cap)
        n = C.c_int(0)
        api.lib.rda_timing_launches(h, which, dptr(buf), cap, C.cast(C.byref(n), C.POINTER(C.c_int)))
        kt[name] = buf[:min(n.value, cap)].copy()
    api.lib.rda_timing_reset(h, 0)
    # un-instrumented pass (events perturb the stream slightly)
    solver2 = new_solver()
    h2 = solver2._be.handle
    api.lib.rda_upload_obstacles(h2, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"])
    api.lib.rda_upload_trace(h2, W + K, dptr(trace["nom_s"]), dptr(trace["nom_u"]), dptr(trace["ref"]), dptr(trace["speed"]))
    for k in range(W):
        api.lib.rda_enqueue_step(h2, k)
    api.lib.rda_sync(h2)
    barrier_all()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        api.lib.rda_enqueue_step(h2, k)
    api.lib.rda_sync(h2)
    barrier_all()
    elapsed2 = max_over_ranks(time.perf_counter() - t0)

    # the same replay with one host synchronisation per step: how long the host needs to queue a step (all launches of one MPC step)
    # and what a step costs when the device starts from an empty stream - the latency floor of the closed loop
    sync_replay = None
    if not args.size_leg:
        solver3 = new_solver()
        h3 = solver3._be.handle
        api.lib.rda_upload_obstacles(h3, staged["n"], dptr(staged["A"]), dptr(staged["b"]), iptr(staged["cone"]), staged["per_t"])
        api.lib.rda_upload_trace(h3, W + K, dptr(trace["nom_s"]), dptr(trace["nom_u"]), dptr(trace["ref"]), dptr(trace["speed"]))
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

    def shard_leg():
        """N > 1, default (replica) mode: the OTHER way to use the node - ONE ego whose obstacles are sharded over the ranks, the
        north-star scaling point (T=20, N_obs=2000): every rank solves the LamMuZ problems of its slots, one in-library ncclAllGather
        per ADMM iteration replicates what the su-problem reads (3 arrays + the reduced sums / masks: DESIGN.md 6), every rank solves
        the identical su-problem.  Device-resident replay of a recorded closed loop, barrier + max over ranks like the headline.
        All ranks call this at the same point; a watchdog bounds it (a collective that never completes must not cost the line)."""
        import torch
        from rda_planner_amd.sharded import enable_rccl
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
            api.lib.rda_sync(hh); barrier_all()
            api.lib.rda_timing_reset(hh, 1)
            t0 = time.perf_counter()
            for k in range(Ws, Ws + Ks):
                assert api.lib.rda_enqueue_step(hh, k) == 0
            api.lib.rda_sync(hh); barrier_all()
            el = max_over_ranks(time.perf_counter() - t0)
            per = {}
            for which, name in ((0, "lammuz"), (1, "su"), (2, "gather")):
                buf, n = np.zeros(Ks * kw_s["iter_num"] + 8), C.c_int(0)
                api.lib.rda_timing_launches(hh, which, dptr(buf), buf.size, C.cast(C.byref(n), C.POINTER(C.c_int)))
                v = buf[:min(n.value, buf.size)]
                per[name] = v
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

    want_shard_leg = (world > 1 and not shard and not oversub and not args.no_shard_leg) or (world == 1 and args.force_shard_leg)
    if rank != 0:
        if want_shard_leg:
            import threading
            wd = threading.Timer(args.shard_leg_timeout, lambda: os._exit(0))
            wd.daemon = True; wd.start()
            try:
                shard_leg()
            except Exception:
                pass
            wd.cancel()
        if dist is not None:
            dist.destroy_process_group()
        return

    E, R = kw["max_edge_num"], 4
    unit_bytes = 8 * (5 * E + 2 * R + 8)                 # 288 B per (obstacle, stage) at E=R=4
    peak = 8000.0
    n_exec = int(np.sum(iters))                          # executed ADMM iterations of the timed replay = executed launches per kernel

    def roof(name, ms, bytes_per_launch, n_exec=n_exec):
        """per EXECUTED launch: launches queued behind the device early-stop flag return at once (no bytes, ~3 us) and are
        separated from the executed ones by their count (sum of rda_info.iters) - the n_exec longest launches are the executed ones"""
        ms = np.sort(np.asarray(ms, float))
        n_noop = max(ms.size - n_exec, 0)
        ex, noop = ms[n_noop:], ms[:n_noop]
        avg_s = float(ex.mean()) * 1e-3 if ex.size else 0.0
        ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 3), "peak": peak, "unit": "GB/s",
                "frac": round(ach / peak, 6), "traffic": None, "avg_launch_us": round(avg_s * 1e6, 2),
                "launches": int(ex.size), "total_ms": round(float(ex.sum()), 3), "algorithmic_bytes_per_launch": bytes_per_launch,
                "skipped_launches": int(noop.size), "skipped_avg_us": round(float(noop.mean()) * 1e3, 2) if noop.size else None,
                "avg_us_over_all_launches": round(float(ms.mean()) * 1e3, 2) if ms.size else None}
    # the LamMuZ launch form that was actually used: asked of the library (rda_lammuz_kernel; a dense grid is three launches, timed together)
    n_loc = -(-N // world) if shard else N
    lm_kernel = api.lib.rda_lammuz_kernel(h).decode()
    su_name = f"k_su<{T}>" if T in (10, 20, 25, 30) else "k_su<0>"
    J = -(-n_loc // 8)
    su_bytes = 32 * T * J * (world if shard else 1) + 8 * (8 * (T + 1) + 5 * T + 10 * T + 4 * T)
    # the same two kernels in the fixed-slot-binding replay (the loop the rooflines of rounds 1-3 were taken from)
    replay_roofs = {"k_su": roof(su_name, kt["k_su"], su_bytes), "k_lammuz": roof(lm_kernel, kt["k_lammuz"], unit_bytes * n_loc * T),
                    "what": "device-resident replay of the recorded closed loop with obstacle_order=False (slots bound once)"}
    if head_kt is not None:                              # the rooflines of the line: the launches of the HEADLINE loop (re-sorted every tick)
        kt, lm_kernel, n_exec = head_kt["kt"], head_kt["lmz_kernel"], head_kt["n_exec"]
    r_lm = roof(lm_kernel, kt["k_lammuz"], unit_bytes * n_loc * T, n_exec)
    # k_su has NO pass over the N terms any more: its set-up reads the reduced form the LamMuZ launch leaves behind - per (stage, 8-slot
    # block) three sums and a near mask (32 of the 48 bytes of a block record) - plus the nominal / reference / kept multipliers; per
    # interior-point pass it visits the NEAR terms only (24 B each, data dependent: not counted, so the fraction is a lower bound)
    r_su = roof(su_name, kt["k_su"], su_bytes, n_exec)
    r_su["includes"] = f"k_su_tracked<{T}> (first solve of every tick) and {su_name} launches of the timed closed loop"
    # latency roof of the su kernel: ONE workgroup (4 waves) on one CU walks a dependent chain; what bounds it is the length of
    # that chain, not bytes - stated next to the HBM fraction so the fraction is not read as a bandwidth problem
    r_su["cus_occupied"] = 1
    r_lm["cus_occupied"] = min(256, (T * J * 128 + 255) // 256) if "rows" in lm_kernel or "k_lammuz_ip" in lm_kernel else min(256, (n_loc * T + 63) // 64)
    tr_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_file):
        try:
            tj = json.load(open(tr_file))
            mode_now = 1 if "k_lammuz_ip" in lm_kernel or "k_lammuz_cp" in lm_kernel else 0
            for wl in tj.get("workloads", {}).values():            # PMC bytes per executed launch of THIS workload only
                if (wl["n_obs"], wl["horizon"], bool(wl["moving"]), wl.get("lmz_mode", 0)) == (N, T, bool(args.moving), mode_now) and world == 1:
                    r_lm["traffic"], r_su["traffic"] = wl.get("k_lammuz"), wl.get("k_su")
                    r_su["traffic_source"] = r_lm["traffic_source"] = tj.get("source")
        except Exception:
            pass
    # What actually bounds these kernels: instruction issue, not bytes.  From the SQ counters of the committed profile of
    # this workload (profiles/issue.json, written by tools/profile_collect.py from the --pmc passes of tools/profile_round.sh; per dispatch,
    # averaged over executed and skipped launches alike, so every figure is a RATIO of two counters of the same pass):
    #   ipc_per_wave  = (VALU + SALU + LDS + VMEM wave-instructions) / (4 SQ_WAVE_CYCLES)   (SQ_WAVE_CYCLES counts quad-cycles summed over the
    #                   waves: 4 x 51.8 k / 4 waves = 51.8 k cycles = 21.6 us at 2.4 GHz for k_su<20>, rocprofv3's average launch is 20.6 us;
    #                   1 = a wave issuing every cycle it is resident)
    #   fp64_frac     = fp64 FLOP (2 FMA + MUL + ADD, x 64 lanes = upper bound) per launch / measured launch time / 78.6 TFLOP/s (vector fp64)
    #   serial_cycles = k_su only: instructions of ONE wave x 6.4 cycles (measured issue interval of a lone wave, tools/latency_micro.cpp)
    #                   = the length of the dependent chain the launch walks; serial_frac = that / the measured launch time at 2.4 GHz
    is_file = os.path.join(ROOT, "profiles", "issue.json")
    if os.path.exists(is_file):
        try:
            ij = json.load(open(is_file))
            for wl in ij.get("workloads", {}).values():
                if (wl["n_obs"], wl["horizon"], bool(wl["moving"])) != (N, T, bool(args.moving)) or world != 1:
                    continue
                for r in (r_su, r_lm):
                    c = wl["kernels"].get(r["kernel"].split("+")[0])
                    if not c:
                        continue
                    insts = c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_VMEM_RD", 0)
                    if c.get("SQ_WAVE_CYCLES"):
                        r["ipc_per_wave"] = round(insts / (4.0 * c["SQ_WAVE_CYCLES"]), 4)
                    flop = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0))
                    t_all = (r["avg_us_over_all_launches"] or 0) * 1e-6
                    if flop and t_all:
                        r["fp64_gflops"] = round(flop / t_all / 1e9, 2)
                        r["fp64_frac"] = round(flop / t_all / 78.6e12, 6)
                    if r is r_su and c.get("SQ_WAVES"):
                        per_wave = insts / c["SQ_WAVES"]
                        r["serial_cycles"] = round(per_wave * 6.4)
                        if t_all:
                            r["serial_frac"] = round(per_wave * 6.4 / (t_all * 2.4e9), 4)
                    r["issue_source"] = ij.get("source")
        except Exception:
            pass
    dominant, secondary = (r_su, r_lm) if r_su["total_ms"] >= r_lm["total_ms"] else (r_lm, r_su)

    kind_word = "moving" if args.moving else "static"
    env_switches = {k: v for k, v in os.environ.items() if k.startswith("RDA_")}
    replay = {"steps_per_s": round(K * (1 if shard else world) / elapsed2, 3), "ms_per_step": round(elapsed2 / K * 1e3, 5),
              "instrumented_ms_per_step": round(elapsed / K * 1e3, 5), "max_du_vs_python_closed_loop": replay_err,
              "what": "recorded step inputs replayed back-to-back on the device, no per-step host synchronisation",
              "synchronised_per_step": sync_replay}
    if head is not None:
        value, ms_step = K * world / head["elapsed"], head["elapsed"] / K * 1e3
        protocol = ("closed loop through the C-ABI, caller in C (tools/closed_loop_host.c), the reference's default obstacle_order=True: per step "
                    "rda_tracked_begin(state) + rda_scene_resort(state) + rda_tracked_finish -> control (the resident scene is re-sorted about the robot "
                    "and the nearest max_obs_num re-staged by k_keys / k_rank / k_build / k_prepare INSIDE the timed region, mpc.py:205-206), one host "
                    "sync per step, host applies the control to the kinematic model; raw scene resident in HBM" if not args.moving else
                    "closed loop through the C-ABI, caller in C (tools/closed_loop_host.c), obstacles advance every tick and are re-sorted "
                    "(obstacle_order=True): rda_tracked_begin + rda_upload_scene_async(order=1) + rda_tracked_finish per step")
    else:                   # obstacle shards: the RCCL path is driven by the replay (every rank enqueues the same steps)
        value, ms_step, protocol = replay["steps_per_s"], replay["ms_per_step"], replay["what"]
    out = {
        "metric": f"MPC steps/sec (ADMM-converged), T={T}, N_obs={N}", "value": round(value, 3), "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_step, 5),
        "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"north-star: acker rectangle robot, T={T}, N_obs={N} {kind_word} seeded polygons, E={E}, iter_num={kw['iter_num']}, iter_threshold=0.2, ro1={kw['ro1']}",
                   "parallelism": "single GPU" if world == 1 else (f"one ego, obstacles sharded {world}-way, RCCL all-gather per ADMM iteration" if shard else f"{world} independent ego replicas (no collective)"),
                   "protocol": protocol, "env_switches": env_switches},
        "median_ms_per_step": round(head["median_ms"], 5) if head else None,
        "max_du_vs_python_closed_loop": head["du"] if head and not args.moving else None,
        "mean_admm_iters": round(float(np.mean(head["iters"])) if head else mean_iters, 3),
        "second_window": head["second_window"] if head else None,
        "fixed_slot_binding": fixed if head else None,
        "duals_follow_obstacles": follow if head else None,
        "su_tol_early": early if head else None,
        "su_hard_warm": hardw if head else None,
        "pcie_inclusive": pcie if head else None,
        "python_caller_closed_loop": pydrv if head else None,
        "device_resident_replay": replay,
        "python_api_closed_loop": {"host_obstacle_staging_steps_per_s": round(1.0 / trace["closed_loop_s_per_step"], 2),
                                   "device_obstacles": cl_dev, "device_obstacles_and_tracking": cl_trk},
        "multi_ego_one_gpu": multi,
        "multi_ego_fleet": fleet,
        "lammuz_interior_point_closed_loops": ip_legs,
        "roofline": dominant, "roofline_secondary": secondary,
        "roofline_fixed_slot_binding_replay": {k: ({kk: v[kk] for kk in ("kernel", "avg_launch_us", "launches", "skipped_launches", "frac", "achieved")} if isinstance(v, dict) else v)
                                               for k, v in replay_roofs.items()},
        "parity": {"stated_tolerance_applied_control": 5e-4, "where": "tests/helpers.py TOL_U; asserted by tests/test_gpu_soak.py (random scenes) and tests/test_gpu_baseline_sizes.py; DESIGN.md 7"},
    }

    if not args.no_cpu_baseline and world == 1:
        from oracle.oracle_backend import oracle_backend, api as orc_api
        ncore = os.cpu_count() or 1
        cpu = RDA_solver(T, car_t, E, N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"], _backend=oracle_backend)
        info_c = Info()
        ou = np.zeros((2, T))
        os_ = np.zeros((3, T + 1))
        sweep, err, n_total = {}, 0.0, 0
        k = 0
        counts = sorted({t for t in (1, 8, 16, 32, 64, ncore) if t <= ncore}) if not args.cpu_threads else [min(args.cpu_threads, ncore)]
        for nthr in counts:
            orc_api().lib.orc_set_threads(nthr)
            n_cpu, t_cpu = 0, 0.0
            while t_cpu < 2.5 or n_cpu < 3:                  # consecutive steps of ONE closed loop (the duals stay warm) ...
                kk = k % (W + K)                             # ... wrapping around the recorded trace when it is used up
                tr_c = trace_o if trace_o is not None else trace     # the headline workload: the scene re-sorted on every tick
                n_c, A_c, b_c, cone_c, pt_c = tr_c["staged"][kk] if tr_c.get("staged") else (staged["n"], staged["A"], staged["b"], staged["cone"], staged["per_t"])
                t1 = time.perf_counter()
                cpu._be.api.step(cpu._be.handle, dptr(tr_c["nom_s"][kk]), dptr(tr_c["nom_u"][kk]), dptr(tr_c["ref"][kk]), float(tr_c["speed"][kk]),
                                 n_c, dptr(A_c), dptr(b_c), iptr(cone_c), pt_c, dptr(ou), dptr(os_), C.byref(info_c))
                t_cpu += time.perf_counter() - t1
                if k < W + K:                                # first pass only: the same state history as the GPU run
                    err = max(err, float(np.abs(ou - tr_c["u_solver"][kk]).max()))
                n_cpu += 1
                k += 1
            sweep[nthr] = round(n_cpu / t_cpu, 3)
            n_total += n_cpu
        best = max(sweep, key=sweep.get)
        out["cpu_baseline"] = {"value": sweep[best], "unit": "steps/s", "cores": best, "kind": "port",
                               "single_thread": sweep.get(1), "thread_sweep": sweep, "host_cores": ncore,
                               "sample": f"{n_total} steps of the headline closed loop (obstacle_order=True: the staged slots of every tick as the GPU run had them; consecutive, "
                                         "wrapping around), ~2.5 s per thread count (oracle/rda_oracle.c: OpenMP over obstacles, OMP_PROC_BIND=close, su-problem serial); best thread count reported",
                               "note": "a restatement of the ADMM in C, NOT the reference's CVXPY+ECOS+pathos path (not installable here): the "
                                       "north-star '>=100x the reference CPU path' cannot be measured against this number",
                               "max_du_vs_gpu": err}
    # ---- every size the metric names + the moving-obstacle and multi-ego configurations, in the SAME driver-run line: one short sub-run of
    #      this script each (own process: a fresh HIP context per shape; --size-leg keeps the closed loops, the timed replay and one
    #      16-thread cpu_baseline sample).  BASELINE.json: N in {20, 200, 2000} at T=20; C4 = 200 moving polygons, T=30; C5 = 64 egos x 100
    #      obstacles, T=25.
    if world == 1 and not args.no_sizes and (N, T, bool(args.moving)) == (200, 20, False):
        import subprocess
        legs = [("n20_T20", ["--n-obs", "20", "--steps", "40", "--warmup", "10", "--fleet-egos", "0"]),
                ("n2000_T20", ["--n-obs", "2000", "--steps", "30", "--warmup", "8", "--fleet-egos", "0"]),
                ("c4_dynamic_obs_n200_T30_moving", ["--n-obs", "200", "--horizon", "30", "--moving", "--steps", "30", "--warmup", "8", "--fleet-egos", "0"]),
                ("c5_shape_n100_T25_fleet64", ["--n-obs", "100", "--horizon", "25", "--steps", "30", "--warmup", "8", "--fleet-egos", "64"])]
        t_sz, sizes = time.perf_counter(), {}
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        for name, extra in legs:
            left = args.sizes_budget_s - (time.perf_counter() - t_sz)
            if left < 8.0:
                sizes[name] = {"skipped": f"sizes budget of {args.sizes_budget_s:.0f} s used up"}
                continue
            try:
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--size-leg", "--cpu-threads", "16"] + extra,
                                    capture_output=True, text=True, timeout=left + 20.0, env=env)
                line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                j = json.loads(line[-1])
                keep = ("value", "unit", "steps", "warmup", "ms_per_step", "median_ms_per_step", "mean_admm_iters", "max_du_vs_python_closed_loop",
                        "second_window", "roofline", "roofline_secondary", "cpu_baseline", "multi_ego_fleet")
                e = {k: j.get(k) for k in keep}
                e["workload"] = j["config"]["workload"]
                e["fixed_slot_binding_steps_per_s"] = (j.get("fixed_slot_binding") or {}).get("steps_per_s")
                e["duals_follow_obstacles_steps_per_s"] = (j.get("duals_follow_obstacles") or {}).get("steps_per_s")
                e["pcie_inclusive_steps_per_s"] = (j.get("pcie_inclusive") or {}).get("steps_per_s")
                e["replay_steps_per_s"] = j["device_resident_replay"]["steps_per_s"]
                if e.get("cpu_baseline"):
                    e["cpu_baseline"] = {k: e["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample", "max_du_vs_gpu")}
                    e["gpu_over_cpu_port"] = round(j["value"] / e["cpu_baseline"]["value"], 1) if e["cpu_baseline"]["value"] else None
                e["leg_wall_s"] = None
                sizes[name] = e
            except Exception as ex:                         # the headline must not depend on these legs
                sizes[name] = {"error": repr(ex)[:300]}
        out["sizes"] = sizes
        out["sizes_wall_s"] = round(time.perf_counter() - t_sz, 1)
    if want_shard_leg:
        import threading

        def give_up():
            out["obstacle_shard_leg"] = {"error": f"no result within {args.shard_leg_timeout:.0f} s (collective did not complete)"}
            print(json.dumps(out), flush=True)
            os._exit(0)
        wd = threading.Timer(args.shard_leg_timeout, give_up)
        wd.daemon = True; wd.start()
        try:
            out["obstacle_shard_leg"] = shard_leg()
        except Exception as e:                               # the headline line must not depend on this leg
            out["obstacle_shard_leg"] = {"error": repr(e)}
        wd.cancel()
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
