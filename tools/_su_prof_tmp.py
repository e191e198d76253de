import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, helpers as hp
from rda_planner_amd import _lib
lib = _lib.hip_api().lib
rng = np.random.default_rng(3)
cfg = hp.make_cfg(T=20, N=200, dynamics=0, L=3.0)
si = hp.su_inputs(rng, cfg)
for k in range(3):
    st, s, u, d, it = hp.su_solve(lib.rda_su_solve, cfg, si)
    print("status", st, "iters", it)
