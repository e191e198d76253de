"""Round 6: what the GPU's su-solves did on the one soak step outside the stated tolerance (seed 23000 --exotic --robots, scene 52, step 46): the landing
statistics of the handle (rda_debug_su_land_n) before and after that step, the GPU closed loop alone (same states as in the soak: the soak's oracle
continues from the GPU's state, so the GPU trajectory does not depend on it).

    python tools/experiments/soak_outlier_land.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import soak_lib                                     # noqa: E402
from rda_planner_amd import scenarios as sc         # noqa: E402
from rda_planner_amd.mpc import MPC                 # noqa: E402

SEED, SCENE, STEP, STEPS = 23000, 52, 46, 100
rng = np.random.default_rng(SEED)
for s in range(SCENE + 1):
    d = soak_lib.draw_scene(rng, SEED, s, STEPS, False, True, False, False, True, False)
gpu = MPC(d["car"], [p.copy() for p in d["path"]], **dict(d["kw"]))
st = d["path"][0].copy().reshape(3, 1)
L = d["car"].wheelbase or 1.0
lib, h = gpu.rda._be.api.lib, gpu.rda._be.handle
prev = [0] * 20
for k in range(STEP + 2):
    cur = [o if not np.any(o.velocity) else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                             else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in d["scene"]]
    u, info = gpu.control(st.copy(), d["speed"], list(cur))
    stt = (C.c_int32 * 20)(); lib.rda_debug_su_land_n(h, stt, 20); stt = list(stt)
    if k >= STEP - 2:
        dl = stt          # (the counters are the step's own: rda_tracked_begin clears them)
        print(f"step {k}: u {u.ravel()} iters {info['iters']} ipm {info['su_ipm_iters']} status {info['status']}; this step: landings accepted {dl[0]} refused {dl[1]} "
              f"rounds {dl[2]} passes {dl[3]} speculative tried {dl[4]} accepted {dl[5]} blind tried {dl[18]} accepted {dl[19]}", flush=True)
    prev = stt
    st = sc.kinematic_step(st, u, d["car"], 0.1) if hasattr(sc, "kinematic_step") else st
