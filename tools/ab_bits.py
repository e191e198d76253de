"""Do two builds of librda_hip.so compute the same thing?  Closed loops through the Python MPC.control API on a set of shapes (every su instantiation:
T = 10 / 20 / 25 / 30 compile-time, T = 15 / 40 generic; static / moving; N up to 2000; the three motion models), one process per build
(RDA_HIP_SO), the applied controls and iteration counts of every step compared with the first tag's:
    python tools/ab_bits.py base new            (tags = tools/_bin/librda_hip_<tag>.so; `cur` = the in-tree build)
    python tools/ab_bits.py --worker out.npz     (internal)
Prints per shape the largest |u - u_first_tag| and the steps whose ADMM / interior-point iteration counts differ."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("ns_T20_N200", dict(n_obs=200, T=20), "acker", 40), ("T10_N24", dict(n_obs=24, T=10), "acker", 40), ("T25_N100", dict(n_obs=100, T=25), "acker", 30),
          ("c4_T30_N200_moving", dict(n_obs=200, T=30, moving=True), "acker", 30), ("T15_N60_generic", dict(n_obs=60, T=15), "diff", 30),
          ("T40_N40_generic", dict(n_obs=40, T=40), "omni", 20), ("n2000_T20", dict(n_obs=2000, T=20), "acker", 12), ("ns_fixed_binding", dict(n_obs=200, T=20), "acker", 40)]


def worker(out):
    import bench
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    res = {}
    for name, wl, dyn, steps in SHAPES:
        car_t, path, obstacles, kw = bench.build_workload(n_steps=steps + 20, **wl)
        if dyn != "acker":
            car_t = sc.rectangle_robot(dynamics=dyn)
        if name.endswith("fixed_binding"):
            kw["obstacle_order"] = False
        mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, **kw)
        state = path[0].copy().reshape(3, 1)
        us, its = [], []
        for k in range(steps):
            cur = obstacles if not wl.get("moving") else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
            u, info = mpc.control(state, 4.0, list(cur))
            us.append(np.asarray(u, float).ravel()); its.append([info["iters"], info["su_ipm_iters"], info["status"]])
            state = sc.kinematic_step(state, u, car_t, 0.1)
        res[name + ":u"] = np.array(us); res[name + ":it"] = np.array(its)
    np.savez(out, **res)


def main():
    if sys.argv[1] == "--worker":
        return worker(sys.argv[2])
    tags, outs = sys.argv[1:], {}
    for tag in tags:
        env = dict(os.environ)
        if tag != "cur":
            env["RDA_HIP_SO"] = os.path.join(ROOT, "tools", "_bin", f"librda_hip_{tag}.so")
        out = os.path.join(ROOT, "gpurun_out", f"ab_bits_{tag}.npz")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", out], env=env, capture_output=True, text=True)
        if pr.returncode:
            print(f"{tag}: worker FAILED rc={pr.returncode}\n{pr.stderr[-1500:]}")
            continue
        outs[tag] = np.load(out)
    first = tags[0]
    for tag in tags[1:]:
        if tag not in outs or first not in outs:
            continue
        for name, _, _, steps in SHAPES:
            ua, ub, ia, ib = outs[first][name + ":u"], outs[tag][name + ":u"], outs[first][name + ":it"], outs[tag][name + ":it"]
            d = np.abs(ua - ub).max(axis=1)
            first_diff = int(np.argmax(d > 0)) if (d > 0).any() else -1
            print(f"{tag} vs {first}  {name:22s} max|du| {d.max():.3e}  first differing step {first_diff:3d}  admm iters differ on {int((ia[:, 0] != ib[:, 0]).sum())} steps, "
                  f"ipm iters {int(ia[:, 1].sum())} -> {int(ib[:, 1].sum())}, status != 0: {int((ia[:, 2] != 0).sum())} -> {int((ib[:, 2] != 0).sum())}", flush=True)


if __name__ == "__main__":
    main()
