#!/usr/bin/env python
"""CPU study aid for the su interior point (test infrastructure, uses the ORACLE): records the su-problems of an oracle closed loop and
replays them through `orc_su_solve` (the cold interior point the kernel's cold attempt mirrors), so that changes to the iteration can be
counted in interior-point iterations before any GPU time is spent.

    python tools/su_replay.py record c4 --n-obs 200 --horizon 30 --steps 30 --moving     # -> /tmp/rda_su_replay/su_c4.bin (cfg + arguments per solve)
    python tools/su_replay.py count c4                                                    # iterations per solve, by ADMM iteration index
    python tools/su_replay.py trace c4 40 41                                              # per-iteration trace of problems 40, 41 (stderr)
    python tools/su_replay.py distance c4          # how far a solution is from its nominal / from the previous step's solution (shifted)

The experimental variants this was used for in round 3 (end-game floors, no second-order term, separate primal / dual steps, multiple
centrality correctors, cross-over to an active-set step) are kept as a patch of the oracle: tools/experiments/su_ipm_variants.patch
(DESIGN section 9 has the counts)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rda_planner_amd._capi import Cfg, c_double_p, c_int_p  # noqa: E402
from oracle import oracle_lib  # noqa: E402


def _lib():
    oracle_lib.build(force=False)
    l = oracle_lib.load()
    l.orc_su_solve.argtypes = [C.POINTER(Cfg)] + [c_double_p] * 3 + [C.c_double] + [c_double_p] * 7 + [c_int_p]
    l.orc_su_solve.restype = C.c_int
    l.orc_set_su_dump.argtypes = [C.c_char_p]
    l.orc_set_su_trace.argtypes = [C.c_int]
    l.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    l.orc_set_threads.argtypes = [C.c_int]
    return l


SCRATCH = os.environ.get("RDA_SU_REPLAY_DIR", "/tmp/rda_su_replay")     # recordings are tens of MB: kept out of the tree (it travels to the GPU box)


def _path(name):
    os.makedirs(SCRATCH, exist_ok=True)
    return os.path.join(SCRATCH, f"su_{name}.bin")


def record(name, n_obs, T, steps, moving):
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.mpc import MPC
    from oracle.oracle_backend import oracle_backend
    l = _lib()
    l.orc_set_su_warm(0.0, 0.0, 0); l.orc_set_threads(16)
    l.orc_set_su_dump(_path(name).encode())
    car_t = sc.rectangle_robot(dynamics="acker")
    length = max(40.0, 0.4 * steps + 12.0)
    path = sc.line_path([4, 25, 0], [4 + length, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(4 + length - 4, 40), seed=sc.SEED, keep_clear=clear, clear_radius=3.2, moving=moving)
    kw = dict(receding=T, iter_num=4, max_edge_num=4, max_obs_num=n_obs, ro1=200, obstacle_order=True, iter_threshold=0.2)
    cpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    state = path[0].copy().reshape(3, 1)
    for i in range(steps):
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * i)) for o in obstacles]
        uc, ic = cpu.control(state.copy(), 4.0, list(cur))
        state = sc.kinematic_step(state, uc, car_t, 0.1)
    l.orc_set_su_dump(b"")
    print("recorded", _path(name), os.path.getsize(_path(name)), "bytes")


def load(name):
    raw = open(_path(name), "rb").read()
    off, probs, csz = 0, [], C.sizeof(Cfg)
    while off < len(raw):
        cfg = Cfg.from_buffer_copy(raw[off:off + csz]); off += csz
        hd = np.frombuffer(raw, dtype=np.float64, count=4, offset=off); off += 32
        T, N = int(hd[0]), int(hd[1])

        def take(n):
            nonlocal off
            a = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
            return a
        probs.append(dict(cfg=cfg, T=T, N=N, ref_speed=float(hd[2]), it=int(hd[3]), s=take(3 * (T + 1)), u=take(2 * T), ref=take(3 * (T + 1)),
                          a=take(2 * N * T), cc=take(N * T), g=take(2 * N * T), d=take(T)))
    return probs


def solve(l, pr):
    T = pr["T"]
    so, uo, do, it = np.zeros(3 * (T + 1)), np.zeros(2 * T), np.zeros(T), C.c_int(0)
    P = lambda x: x.ctypes.data_as(c_double_p)  # noqa: E731
    st = l.orc_su_solve(C.byref(pr["cfg"]), P(pr["s"]), P(pr["u"]), P(pr["ref"]), pr["ref_speed"], P(pr["a"]), P(pr["cc"]), P(pr["g"]), P(pr["d"]),
                        P(so), P(uo), P(do), C.byref(it))
    return st, it.value, so, uo, do


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["record", "count", "trace", "distance", "gpu"]); ap.add_argument("name"); ap.add_argument("problems", nargs="*", type=int)
    ap.add_argument("--n-obs", type=int, default=200); ap.add_argument("--horizon", type=int, default=30); ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--moving", action="store_true")
    a = ap.parse_args()
    if a.cmd == "record":
        return record(a.name, a.n_obs, a.horizon, a.steps, a.moving)
    l, probs = _lib(), load(a.name)
    if a.cmd == "gpu":            # the same recorded problems through the kernel's cold solve (rda_su_solve) beside the oracle's: status, iterations, |du|
        from rda_planner_amd._lib import hip_api
        hl = hip_api().lib
        for i, pr in enumerate(probs):
            T = pr["T"]
            so, uo, do, it = np.zeros(3 * (T + 1)), np.zeros(2 * T), np.zeros(T), C.c_int(0)
            P = lambda x: x.ctypes.data_as(c_double_p)  # noqa: E731
            sg = hl.rda_su_solve(C.byref(pr["cfg"]), P(pr["s"]), P(pr["u"]), P(pr["ref"]), pr["ref_speed"], P(pr["a"]), P(pr["cc"]), P(pr["g"]), P(pr["d"]),
                                 P(so), P(uo), P(do), C.byref(it))
            sc_, ic, _, uc, _ = solve(l, pr)
            print(f"problem {i} (ADMM iteration {pr['it']}): gpu status {sg} / {it.value} iterations, oracle status {sc_} / {ic} iterations, max |u_gpu - u_oracle| {np.abs(uo - uc).max():.2e}")
        return
    if a.cmd == "count":
        its = np.array([solve(l, pr)[1] for pr in probs])
        print(f"{len(its)} su-problems: {its.mean():.2f} interior-point iterations per cold solve (median {np.median(its):.0f}, max {its.max()}); by ADMM iteration " +
              " ".join(f"{np.mean([i for i, p in zip(its, probs) if p['it'] == k]):.1f}" for k in range(4)))
    elif a.cmd == "trace":
        l.orc_set_su_trace(1)
        for i in a.problems:
            print(f"problem {i} (ADMM iteration {probs[i]['it']})", file=sys.stderr)
            st, it = solve(l, probs[i])[:2]
            print(f"  status {st}, {it} iterations", file=sys.stderr)
    else:
        sols, step = {}, -1
        for pr in probs:
            step += pr["it"] == 0
            _, _, _, uo, _ = solve(l, pr)
            sols[(step, pr["it"])] = (uo.reshape(2, -1), pr["u"].reshape(2, -1))
        dn, ds = [], []
        for (j, k), (u, nom) in sorted(sols.items()):
            if (j - 1, k) in sols:
                up = sols[(j - 1, k)][0]
                dn.append(np.abs(u - nom).max()); ds.append(np.abs(u - np.concatenate([up[:, 1:], up[:, -1:]], axis=1)).max())
        print(f"max |u* - nominal| per solve: median {np.median(dn):.3f}, 90 % {np.quantile(dn, 0.9):.3f};  max |u* - previous step's u* of the same ADMM "
              f"iteration, shifted|: median {np.median(ds):.3f}, 90 % {np.quantile(ds, 0.9):.3f}")


if __name__ == "__main__":
    main()
