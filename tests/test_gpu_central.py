"""-m gpu : the interior-point LamMuZ kernel (csrc/lammuz_cp_device.h, `k_lammuz_cp_*`) against the oracle's interior-point
restatement (oracle/lmz_ipm.c, pinned on the reference's own one-stage problems by tests/test_reference_pinned.py).

Both return the point of the central path of the reference's cone program at the same barrier parameter mu - a well-conditioned
function of the data - so the two implementations agree to ~1e-5 in the duals (un-normalised half-spaces put factors of 100 between
a multiplier and Im) and within the stated 1e-4 closed-loop tolerance of tests/test_gpu_baseline_sizes.py in the controls.
Covers the norm2 (circle) ROBOT cone of rda_solver.py:1034-1039, which the enumeration kernels do not have."""
import ctypes as C

import numpy as np
import pytest

from rda_planner_amd import scenarios as sc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def central_orc(orc):
    orc.lib.orc_set_lmz_ipm_mu.argtypes = [C.c_double]
    orc.lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    orc.lib.orc_set_lmz_mode(1)
    yield orc
    orc.lib.orc_set_lmz_mode(0)
    orc.lib.orc_set_lmz_ipm_mu(1e-6)


def _closed_loop(car_t, path, obstacles, kw, steps, mu, orc, start=None):
    from oracle.oracle_backend import oracle_backend
    from rda_planner_amd.mpc import MPC
    orc.lib.orc_set_lmz_ipm_mu(mu)
    cpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    gpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, lmz_central=mu, **kw)
    state = (path[0] if start is None else start).copy().reshape(3, 1)
    worst = dict(u=0.0, state=0.0, res=0.0)
    min_clear = np.inf
    for i in range(steps):
        uc, ic = cpu.control(state.copy(), 4.0, list(obstacles))
        ug, ig = gpu.control(state.copy(), 4.0, list(obstacles))
        assert ic["iters"] == ig["iters"] and ic["lmz_fail"] == ig["lmz_fail"], (i, ic["iters"], ig["iters"], ic["lmz_fail"], ig["lmz_fail"])
        worst["u"] = max(worst["u"], float(np.abs(cpu.cur_vel_array - gpu.cur_vel_array).max()))
        if np.isfinite(ic["resi_dual"]):
            worst["res"] = max(worst["res"], abs(ic["resi_dual"] - ig["resi_dual"]) / (1 + ic["resi_dual"]), abs(ic["resi_pri"] - ig["resi_pri"]))
        sc_, sg_ = cpu.rda.get_state(), gpu.rda.get_state()
        for k in sc_:
            worst["state"] = max(worst["state"], float(np.abs(sc_[k] - sg_[k]).max()))
        gpu.rda.set_state(sc_)
        gpu.cur_vel_array = cpu.cur_vel_array.copy()
        state = sc.kinematic_step(state, uc, car_t, 0.1)
        min_clear = min(min_clear, sc.clearance(car_t, state, obstacles))
    return worst, min_clear


@pytest.mark.parametrize("mu", [1e-6, 1e-3])
def test_central_path_duals_rectangle_robot(central_orc, mu):
    """polygon robot, polygons + circles, padded slots: the interior-point kernel as an alternative to the enumeration"""
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_boxes(7, (8, 16), (36, 34), keep_clear=clear, clear_radius=3.5) + [sc.circle(20, 29.5, 1.0), sc.circle(28, 20.5, 0.7)]
    kw = dict(receding=10, iter_num=3, max_edge_num=4, max_obs_num=10, obstacle_order=True)
    worst, _ = _closed_loop(car_t, path, obstacles, kw, 14, mu, central_orc)
    print(f"central path mu={mu}: max |du| {worst['u']:.2e}, dual state {worst['state']:.2e}, residuals {worst['res']:.2e}")
    assert worst["u"] < 1e-4 and worst["state"] < 1e-4 and worst["res"] < 1e-6, worst


def test_norm2_robot_cone(central_orc):
    """circle robot (car_tuple.cone_type == 'norm2', rda_solver.py:1034-1039) among polygons and circles: GPU == oracle step by
    step, and the closed loop keeps its distance"""
    car_t = sc.circle_robot(radius=0.8, dynamics="diff")
    path = sc.line_path([4, 25, 0], [34, 25, 0], 0.1)
    obstacles = [sc.box(14, 27.1, 2.0, 1.5, 0.4), sc.circle(22, 22.7, 1.0), sc.regular_polygon(28, 27.6, 3, 1.2, 0.3), sc.box(20, 30, 3, 2, 0.0)]
    kw = dict(receding=10, iter_num=4, max_edge_num=4, max_obs_num=4, obstacle_order=True)
    worst, min_clear = _closed_loop(car_t, path, obstacles, kw, 60, 1e-6, central_orc)
    print(f"norm2 robot: max |du| {worst['u']:.2e}, dual state {worst['state']:.2e}, residuals {worst['res']:.2e}, min clearance {min_clear:.3f}")
    assert worst["u"] < 1e-4 and worst["state"] < 1e-4 and worst["res"] < 1e-6, worst
    assert min_clear > 0.0


def test_norm2_robot_is_not_rejected_any_more(hip):
    from rda_planner_amd.rda_solver import RDA_solver
    s = RDA_solver(8, sc.circle_robot(), 4, 3, iter_num=2, time_print=False)
    T = 8
    nom_s = np.zeros((3, T + 1)); nom_s[0] = 0.4 * np.arange(T + 1)
    u, info = s.iterative_solve(nom_s, np.vstack([np.full(T, 4.0), np.zeros(T)]), [nom_s[:, j:j + 1] for j in range(T + 1)], 4.0, [])
    assert np.isfinite(u).all() and info["iters"] >= 1
