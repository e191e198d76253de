"""Per-wave timeline of ONE su launch (the last of a closed loop): which wave works and which waits in every phase of the interior-point iterations.
Needs a -DSU_TRACE build of the library:
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSU_TRACE -shared -o tools/_bin/librda_hip_trace.so rda_planner_amd/csrc/rda_hip.hip
  RDA_HIP_SO=tools/_bin/librda_hip_trace.so python tools/su_trace.py [--n-obs N] [--horizon T] [--moving] [--steps K] [--raw]
Event ids (su_device.h TR): 100 entry, 101 set-up done, 1 iteration top, 3 hinge terms summed per lane, 4 DPP group sums, 5 stage derivatives, 6 gradient entry,
7 Hessian row (-> barrier) 8, 9 Riccati / adjoint / measures done (-> barrier) 10, 11 closed-loop rows + verdict (-> barrier), per pass p (0 predictor, 1 corrector) 20+10p start,
31 corrector rhs (behind its barrier), 22+10p sweep constants | unit sweeps (-> barrier) 23+10p, 24+10p backward sweeps | interface (-> barrier) 25+10p, 26+10p interface solve +
forward sweeps (-> barrier) 27+10p, 28+10p row steps (-> reduction) 29+10p, 40 update (-> barrier) 41, 42 reach / complementarity + rows of the next iteration (-> barrier) 43,
102 loop left, 103 final roll-out, 104 written back.  Prints, per interval between two consecutive events of wave 0, the cycles every wave spent between ITS same two events
(summed over the iterations of the launch) - a wave that reaches a barrier early shows a long 'x -> barrier released' interval."""
import argparse
import ctypes as C
import os
import sys
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-obs", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--moving", action="store_true")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--launches", type=int, default=8, help="average over the last launch of this many consecutive steps")
    ap.add_argument("--raw", action="store_true", help="print the event list of every wave of the last launch")
    args = ap.parse_args()
    import bench
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.rda_solver import hip_options
    from rda_planner_amd._lib import hip_api
    car_t, path, obstacles, kw = bench.build_workload(n_obs=args.n_obs, T=args.horizon, n_steps=args.steps + 20, moving=args.moving)
    mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(su_prof=1), **kw)
    lib = hip_api().lib
    lib.rda_debug_su_trace.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_int)]
    lib.rda_debug_su_trace.restype = C.c_int
    cap = 1024
    buf, n = (C.c_longlong * (4 * cap * 2))(), C.c_int(0)
    state = path[0].copy().reshape(3, 1)
    acc, total, iters, nl = OrderedDict(), 0, 0, 0
    for k in range(args.steps):
        cur = obstacles if not args.moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
        u, info = mpc.control(state, 4.0, list(cur))
        state = sc.kinematic_step(state, u, car_t, 0.1)
        if k < args.steps - args.launches:
            continue
        assert lib.rda_debug_su_trace(mpc.rda._be.handle, buf, cap, C.byref(n)) == 0, "not a -DSU_TRACE build"
        ev = np.array(buf[:], dtype=np.int64).reshape(4, cap, 2)
        waves = []
        for w in range(4):
            e = ev[w]
            m = int(np.argmax(e[:, 0] == 104)) + 1 if (e[:, 0] == 104).any() else cap
            waves.append(e[:m])
        if args.raw and k == args.steps - 1:
            for w in range(4):
                print(f"wave {w}:", " ".join(f"{int(i)}@{int(c - waves[0][0, 1])}" for i, c in waves[w]))
        nl += 1
        total += int(waves[0][-1, 1] - waves[0][0, 1])
        iters += int((waves[0][:, 0] == 1).sum())
        for w in range(4):
            e = waves[w]
            for a_, b_ in zip(e[:-1], e[1:]):
                key = (int(a_[0]), int(b_[0]))
                acc.setdefault(key, np.zeros(5))
                acc[key][w] += float(b_[1] - a_[1])
                if w == 0:
                    acc[key][4] += 1
    print(f"T={args.horizon} N={args.n_obs} moving={args.moving}: last su launch of {nl} steps, {total / nl:.0f} cycles per launch, {iters / nl:.2f} iteration tops per launch")
    print(f"{'interval':>12s} {'count':>6s} {'wave0':>9s} {'wave1':>9s} {'wave2':>9s} {'wave3':>9s}   cycles per launch (sum over the launch's iterations)")
    for (a_, b_), v in acc.items():
        print(f"{a_:5d} ->{b_:4d} {v[4] / nl:6.1f} {v[0] / nl:9.0f} {v[1] / nl:9.0f} {v[2] / nl:9.0f} {v[3] / nl:9.0f}")


if __name__ == "__main__":
    main()
