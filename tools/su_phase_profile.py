"""Phase cycle counters of the su-solves of a closed loop (rda_opts::su_prof + rda_debug_su_prof): where k_su's time goes in steady
state - not in the cold stand-alone hook.  Needs a PROFILING build of the library (the product build has no counters since round 5):
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSU_PROF -shared -o tools/_bin/librda_hip_prof.so rda_planner_amd/csrc/rda_hip.hip   (-DSU_FINE for --fine)
  RDA_HIP_SO=tools/_bin/librda_hip_prof.so python tools/su_phase_profile.py [--n-obs N] [--horizon T] [--moving] [--steps K]
Prints, per su-solve, the clock64 ticks of every phase marker of su::solve (su_device.h `mark(k)`)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "load nominal", 11: "linearise", 13: "clip + roll-out", 3: "term sums / masks", 14: "density + reach check", 12: "stage sums",
         9: "duals start + reference", 1: "hinge sums", 2: "gradients + Hessian bases", 4: "Riccati / adjoint / measures", 5: "closed-loop matrices + verdict",
         6: "rhs", 7: "vector sweeps", 8: "slack / multiplier step", 15: "final roll-out", 10: "write-back + pose table"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-obs", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--moving", action="store_true")
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--order", action="store_true", help="re-sort the obstacles by distance every tick (MPC default; the headline loop of bench.py keeps the slot binding fixed)")
    ap.add_argument("--fine", action="store_true", help="the library was built with -DSU_FINE (make -C rda_planner_amd/csrc CXXFLAGS+=-DSU_FINE): the set-up slots carry sub-phases of the iteration, the whole set-up is booked under slot 10")
    ap.add_argument("--iter-num", type=int, default=0, help="ADMM iterations per step (1: only the FIRST su-solve of every tick - the one inside k_su_tracked - is profiled)")
    ap.add_argument("--lmz-central", type=float, default=0.0, help="interior-point LamMuZ mode (central duals at this mu)")
    ap.add_argument("--circle", action="store_true", help="circle robot (norm2 cone: the interior-point LamMuZ kernel)")
    ap.add_argument("--per-step", action="store_true", help="print (ADMM iterations, su interior-point iterations) of every step")
    args = ap.parse_args()
    import bench
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.rda_solver import hip_options
    from rda_planner_amd._lib import hip_api
    car_t, path, obstacles, kw = bench.build_workload(n_obs=args.n_obs, T=args.horizon, n_steps=args.steps + 20, moving=args.moving)
    kw["obstacle_order"] = bool(args.order)
    if args.iter_num:
        kw["iter_num"] = args.iter_num
    if args.lmz_central > 0:
        kw["lmz_central"] = args.lmz_central
    if args.circle:
        car_t = sc.circle_robot(radius=0.8, dynamics="diff")
    mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(su_prof=1), **kw)
    lib = hip_api().lib
    state = path[0].copy().reshape(3, 1)
    out = (C.c_longlong * 16)()
    solves = ipm = 0
    for k in range(args.steps):
        cur = obstacles if not args.moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
        u, info = mpc.control(state, 4.0, list(cur))
        state = sc.kinematic_step(state, u, car_t, 0.1)
        if k == 9:                                   # warm-up over: drop what has accumulated
            lib.rda_debug_su_prof(mpc.rda._be.handle, out)
        elif k > 9:
            solves += info["iters"]; ipm += info["su_ipm_iters"]
        if args.per_step:
            print(k, info["iters"], info["su_ipm_iters"], info["status"])
    assert lib.rda_debug_su_prof(mpc.rda._be.handle, out) == 0
    if args.fine:
        NAMES.update({14: "(2) stage derivatives + inequality rows", 12: "(3a) stage gradients", 2: "(3b) Hessian bases", 9: "(6a) corrector rc + gh", 6: "(6b) sweep constants",
                      15: "(8a) slack / multiplier rows", 0: "(8b) step-length reduction", 11: "(8c) update", 13: "(8d) reach check", 8: "(8e) sigma / light check",
                      10: "set-up + final roll-out + write-back", 3: "(7a) backward sweeps + interface (time split)", 7: "(7b) forward sweeps (split: + interface solve)"})
    tot = sum(out)
    print(f"T={args.horizon} N={args.n_obs} moving={args.moving}: {solves} su-solves, {ipm / solves:.2f} interior-point iterations per solve, "
          f"{tot / solves:.0f} ticks per solve")
    for k in ((1, 14, 12, 2, 4, 5, 9, 6, 3, 7, 15, 0, 11, 13, 8, 10) if args.fine else (0, 11, 13, 9, 3, 14, 12, 1, 2, 4, 5, 6, 7, 8, 15, 10)):
        print(f"  [{k:2d}] {NAMES[k]:34s} {out[k] / solves:9.0f} ticks/solve  {100.0 * out[k] / tot:5.1f} %")


if __name__ == "__main__":
    main()
