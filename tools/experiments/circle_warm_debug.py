"""One-off (round 5): the packed-rows LamMuZ kernel (remembered circle supports: lmz::warm_circle) against the one-row-per-wave kernel (circle rows always
enumerated) on a soak scene, step by step from the same state: first row whose duals differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import soak_lib  # noqa: E402
from rda_planner_amd import scenarios as sc  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402
from rda_planner_amd.rda_solver import hip_options  # noqa: E402

seed, scene, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for s in range(scene + 1):
    d = soak_lib.draw_scene(rng, seed, s, 80, circles=True)
if len(sys.argv) > 4:
    d["kw"]["iter_num"] = int(sys.argv[4])
a = MPC(d["car"], [p.copy() for p in d["path"]], **d["kw"], hip_opts=hip_options(lmz_rows=1))
b = MPC(d["car"], [p.copy() for p in d["path"]], **d["kw"], hip_opts=hip_options(lmz_rows=0))
st = d["path"][0].copy().reshape(3, 1)
for k in range(steps):
    cur = [o if not np.any(o.velocity) else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                             else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in d["scene"]]
    ua, ia = a.control(st.copy(), d["speed"], list(cur))
    ub, ib = b.control(st.copy(), d["speed"], list(cur))
    sa, sb = a.rda.get_state(), b.rda.get_state()
    dl = np.abs(sa["lam"] - sb["lam"]).max(axis=(1, 2)) if sa["lam"].ndim == 3 else np.abs(sa["lam"] - sb["lam"])
    worst = float(np.abs(sa["lam"] - sb["lam"]).max())
    if worst > 1e-9 or np.abs(ua - ub).max() > 1e-9:
        idx = np.unravel_index(np.argmax(np.abs(sa["lam"] - sb["lam"])), sa["lam"].shape)
        print(f"step {k}: |du| {np.abs(ua - ub).max():.2e}, max |dlam| {worst:.2e} at {idx} (shape {sa['lam'].shape})")
        n = idx[0]
        np.set_printoptions(linewidth=200, precision=10)
        dm = np.abs(sa["lam"][n] - sb["lam"][n]).reshape(sa["lam"].shape[1], -1).max(axis=1) if sa["lam"].ndim == 3 else None
        print(" per-column |dlam| of that obstacle:", dm)
        t = int(np.argmax(dm))
        print(f" column {t}: lam rows {sa['lam'][n][t]} enum {sb['lam'][n][t]}\n            mu rows {sa['mu'][n][t]} enum {sb['mu'][n][t]}")
        print(f" |a| rows {np.hypot(*sa['lam'][n][t][:2]):.12f} enum {np.hypot(*sb['lam'][n][t][:2]):.12f}; rows differing at all: {int((np.abs(sa['lam'] - sb['lam']).reshape(-1, sa['lam'].shape[-1]).max(axis=1) > 0).sum())}")
        break
    b.rda.set_state(sa); b.cur_vel_array = a.cur_vel_array.copy(); b.cur_index = a.cur_index
    st = sc.kinematic_step(st, ua, d["car"], 0.1)
else:
    print("no difference in", steps, "steps")
