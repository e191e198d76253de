"""how far do two identical loops drift when one forgets its remembered supports before every third step (diagnostic of
tests/test_gpu_supports.py::test_flushing_the_supports_cache_mid_loop_changes_no_bit)"""
import ctypes as C
import sys
import numpy as np
from rda_planner_amd import scenarios as sc
from rda_planner_amd.mpc import MPC
from rda_planner_amd._lib import hip_api

n_obs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
car_t = sc.rectangle_robot(dynamics="acker")
path = sc.line_path([4, 25, 0], [44, 25, 0], 0.1)
clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(40, 40), seed=sc.SEED + 3, keep_clear=clear, clear_radius=3.2, moving=False)
kw = dict(sample_time=0.1, time_print=False, receding=20, iter_num=3, max_edge_num=4, max_obs_num=n_obs, ro1=200, obstacle_order=True)
a = MPC(car_t, [p.copy() for p in path], **kw)
b = MPC(car_t, [p.copy() for p in path], **kw)
lib = hip_api().lib
lib.rda_debug_flush_supports.argtypes = [C.c_void_p]
lib.rda_lammuz_kernel.restype = C.c_char_p
state = path[0].copy().reshape(3, 1)
for k in range(18):
    if k % 3 == 2:
        lib.rda_debug_flush_supports(b.rda._be.handle)
    ua, ia = a.control(state.copy(), 4.0, list(obstacles))
    ub, ib = b.control(state.copy(), 4.0, list(obstacles))
    sa, sb = a.rda.get_state(), b.rda.get_state()
    d = {key: float(np.abs(sa[key] - sb[key]).max()) for key in ("lam", "mu", "z", "xi", "zeta")}
    nd = {key: int((sa[key] != sb[key]).sum()) for key in ("lam", "mu", "z")}
    print(k, lib.rda_lammuz_kernel(a.rda._be.handle).decode(), "du", float(np.abs(ua - ub).max()), "iters", ia["iters"], ib["iters"], d, nd, flush=True)
    if nd["lam"]:
        w = np.argwhere(sa["lam"] != sb["lam"])
        print("   first differing lam entries", w[:6].tolist(), sa["lam"][tuple(w[0])], sb["lam"][tuple(w[0])])
        if k == 2:
            np.set_printoptions(precision=17, linewidth=200)
            print("   shapes", {key: sa[key].shape for key in sa})
            for idx in w[:3]:
                i0, i1 = int(idx[0]), int(idx[1])
                for key in ("lam", "mu"):
                    print("   ", key, (i0, i1), sa[key][i0, i1], sb[key][i0, i1])
                print("    z", [(sa["z"][i], sb["z"][i]) for i in [(i0, i1), (i1, i0)] if i[0] < sa["z"].shape[0] and i[1] < sa["z"].shape[1]])
    state = sc.kinematic_step(state, ua, car_t, 0.1)
