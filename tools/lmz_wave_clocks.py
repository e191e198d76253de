"""Where a wave of k_lammuz_rows spends its time (debug build only).
  hipcc ... -DRDA_LMZ_CLK -o tools/librda_hip_clk.so   (python tools/lmz_wave_clocks.py --build, on the build machine)
  python tools/lmz_wave_clocks.py [--n-obs N] [--steps K]   (on the GPU box)
clock64 ticks per section of a wave, summed over the waves of the executed launches of a closed loop; the sections are the LMZ_CLK(k)
marks of lammuz_body_rows<0> in csrc/rda_hip.hip."""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.environ.get("RDA_LMZ_CLK_SO") or os.path.join(ROOT, "tools", "librda_hip_clk.so")      # (RDA_LMZ_CLK_SO: another -DRDA_LMZ_CLK build, A/B)
NAMES = ["stop flag", "robot data, half-spaces -> LDS, barrier", "pose, duals, pose products, support cache", "warm candidate + certificate",
         "shared enumeration of failing rows (2 barriers)", "central normal (T1)", "dual / residual updates, row record", "block partial"]


def _decode(c, n):
    """support candidate index -> tuple of rows (lammuz_device.h eval_candidate / decode_pair)"""
    if c == 0:
        return ()
    if c <= n:
        return (c - 1,)
    k, i1, rowlen = c - 1 - n, 0, n - 1
    while k >= rowlen:
        k -= rowlen; i1 += 1; rowlen -= 1
    return (i1, i1 + 1 + k)


def fail_report(flog, E, R):
    """how the support of a failing row differs from the remembered one"""
    import collections
    n = int(flog[0]); rows = flog[1:1 + 3 * min(n, 200000)].reshape(-1, 3)
    nm = 1 + R + R * (R - 1) // 2
    kinds = collections.Counter()
    for old, new, circ in rows:
        if circ:
            kinds["circle obstacle (no warm path)"] += 1; continue
        if old < 0:
            kinds["no hint"] += 1; continue
        lo, mo, ln, mn = _decode(old // nm, E), _decode(old % nm, R), _decode(new // nm, E), _decode(new % nm, R)
        def rel(a, b):
            if a == b: return "same"
            if set(a) < set(b) or set(b) < set(a): return "sub/superset"
            if set(a) & set(b): return "shares a row"
            return "disjoint"
        kinds[f"lam {rel(lo, ln)} | mu {rel(mo, mn)}"] += 1
    print(f"  {n} failing rows:")
    for k, v in kinds.most_common():
        print(f"    {v:7d}  {100.0 * v / max(n, 1):5.1f} %  {k}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--n-obs", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--steps", type=int, default=120)
    args = ap.parse_args()
    if args.build:
        src = os.path.join(ROOT, "rda_planner_amd", "csrc")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DRDA_LMZ_CLK",
                               "-shared", "-o", SO, os.path.join(src, "rda_hip.hip")])
        print("built", SO)
        return
    from rda_planner_amd import _lib
    _lib.SO_PATH = SO
    import bench
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    car_t, path, obstacles, kw = bench.build_workload(n_obs=args.n_obs, T=args.horizon, n_steps=args.steps + 20, moving=False)
    kw["obstacle_order"] = not os.environ.get("LMZ_CLK_FIXED")      # the reference default: re-sorted every tick (LMZ_CLK_FIXED=1: fixed binding)
    mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, **kw)
    lib = _lib.hip_api().lib
    import numpy as np
    W = 4096
    lib.rda_debug_lmz_clk.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    buf = np.zeros((W, 16), dtype=np.uint64)
    flog = np.zeros(1 + 3 * 200000, dtype=np.int32)
    state = path[0].copy().reshape(3, 1)
    for k in range(args.steps):
        u, info = mpc.control(state, 4.0, list(obstacles))
        state = sc.kinematic_step(state, u, car_t, 0.1)
        if k == 9:
            lib.rda_debug_lmz_clk(mpc.rda._be.handle, buf.ctypes.data, W)       # warm-up over: arm (first call) the per-wave slots
            lib.rda_debug_lmz_clk(mpc.rda._be.handle, flog.ctypes.data, -1)     # ... and the fail log
    assert lib.rda_debug_lmz_clk(mpc.rda._be.handle, buf.ctypes.data, W) == 0
    assert lib.rda_debug_lmz_clk(mpc.rda._be.handle, flog.ctypes.data, -1) == 0
    fail_report(flog, kw["max_edge_num"], 4)
    used = buf[:, 8] > 0
    out = buf[used].astype(float)
    launches = out[:, 8].max()
    per = out[:, :8].sum(axis=0) / out[:, 8].sum()
    print(f"T={args.horizon} N={args.n_obs}: {int(used.sum())} waves per launch, {int(launches)} executed launches, {out[:, 9].sum() / out[:, 8].sum():.0f} ticks per wave, "
          f"slowest wave {int(out[:, 10].max())} ticks; {int(out[:, 11].sum())} wave-runs in a workgroup that enumerated ({out[:, 12].sum() / max(out[:, 11].sum(), 1):.0f} ticks each)")
    for k, name in enumerate(NAMES):
        print(f"  [{k}] {name:58s} {per[k]:8.0f} ticks/wave")
    # the LAST executed launch on the device-wide 100 MHz clock: when its waves started and ended
    st, en = out[:, 13] * 0.01, out[:, 14] * 0.01      # us
    t0 = st.min()
    print(f"  last launch: waves start {np.percentile(st - t0, 50):.2f} us (median) / {np.percentile(st - t0, 99):.2f} (99 %) / {(st - t0).max():.2f} (last) after the first one; "
          f"wave duration median {np.median(en - st):.2f} us, 99 % {np.percentile(en - st, 99):.2f}, max {(en - st).max():.2f}; last wave ends {(en - t0).max():.2f} us after the first start")
    # the slowest wave slot on average: the launch ends with it
    avg = out[:, 9] / out[:, 8]
    print(f"  per-slot average: min {avg.min():.0f}  median {np.median(avg):.0f}  max {avg.max():.0f}")


if __name__ == "__main__":
    main()
