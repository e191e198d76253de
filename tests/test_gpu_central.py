"""-m gpu : the interior-point LamMuZ kernels - the row-parallel one (csrc/lammuz_ip_device.h, `k_lammuz_ip`: 16 lanes per
sub-problem, what runs for every shape that fits 16 variables) and the per-thread one it replaced (csrc/lammuz_cp_device.h,
`k_lammuz_cp_*`, kept for bigger shapes; RDA_LMZ_IP_ROWS=0) - against the oracle's interior-point restatement (oracle/lmz_ipm.c,
pinned on the reference's own one-stage problems by tests/test_reference_pinned.py).  The arithmetic of the row-parallel kernel is
additionally pinned on the CPU (tests/test_ip_rows_emu.py: the same template on a host lane vector).

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


def test_central_path_at_the_north_star_size(central_orc):
    """T=20, N=200 (4000 sub-problems per launch: the size the kernel's launch time is quoted on), robust mode mu = 1e-3"""
    from test_gpu_baseline_sizes import _workload
    central_orc.lib.orc_set_threads.argtypes = [C.c_int]
    central_orc.lib.orc_set_threads(16)
    try:
        car_t, path, obstacles, kw = _workload(200, 20, 40)
        worst, _ = _closed_loop(car_t, path, obstacles, kw, 8, 1e-3, central_orc)
    finally:
        central_orc.lib.orc_set_threads(1)
    print(f"central path mu=1e-3, T=20 N=200: max |du| {worst['u']:.2e}, dual state {worst['state']:.2e}, residuals {worst['res']:.2e}")
    assert worst["u"] < 1e-4 and worst["state"] < 1e-4 and worst["res"] < 1e-6, worst


def test_per_thread_kernel_equals_the_row_parallel_one(monkeypatch):
    """RDA_LMZ_IP_ROWS=0 (the per-thread solver for the same shape): both kernels end on the same point of the central path"""
    from rda_planner_amd.mpc import MPC
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_boxes(7, (8, 16), (36, 34), keep_clear=clear, clear_radius=3.5) + [sc.circle(20, 29.5, 1.0)]
    kw = dict(receding=10, iter_num=3, max_edge_num=4, max_obs_num=8, obstacle_order=True)
    rows = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, lmz_central=1e-3, **kw)
    monkeypatch.setenv("RDA_LMZ_IP_ROWS", "0")
    thr = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, lmz_central=1e-3, **kw)
    state = path[0].copy().reshape(3, 1)
    worst = 0.0
    for i in range(10):
        ur, ir = rows.control(state.copy(), 4.0, list(obstacles))
        ut, it_ = thr.control(state.copy(), 4.0, list(obstacles))
        assert ir["iters"] == it_["iters"]
        worst = max(worst, float(np.abs(ur - ut).max()))
        thr.rda.set_state(rows.rda.get_state()); thr.cur_vel_array = rows.cur_vel_array.copy(); thr._dev_u = None
        state = sc.kinematic_step(state, ur, car_t, 0.1)
    assert worst < 1e-5, worst


def test_robust_mode_reaches_the_goal_from_perturbed_starts():
    """`MPC(..., lmz_central=1e-3)` - interior duals at a moderate barrier parameter, the robust choice in tight scenes (DESIGN.md 2:
    16 / 16 in both scenes on the oracle, against 2 / 16 and 8 / 16 for the default tie-break T1) - on the HIP library with the
    row-parallel kernel: the reference's corridor (C2) and lidar (C3) examples from 16 starts perturbed by +-0.3 m / +-0.08 rad
    each.  Success = goal reached without contact.  The RATE is asserted (closed loops are chaotic in these scenes)."""
    from rda_planner_amd.mpc import MPC
    from test_host_api import lidar_closed_loop
    rng = np.random.default_rng(11)
    starts = [(0.0, 0.0, 0.0)] + [(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.08, 0.08)) for _ in range(15)]
    car_a = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([0, 20, 0], [60, 20, 0], 0.1)
    obs = sc.scene_corridor(n_extra=0)
    ok_c = 0
    for dx, dy, dth in starts:
        mpc = MPC(car_a, [p.copy() for p in path], sample_time=0.1, max_edge_num=4, max_obs_num=6, lmz_central=1e-3)
        state = np.array([[0.0 + dx], [20.0 + dy], [0.0 + dth]])
        minc, arrived = np.inf, False
        for i in range(300):
            u, info = mpc.control(state, 4, list(obs))
            state = sc.kinematic_step(state, u, car_a, 0.1)
            minc = min(minc, sc.clearance(car_a, state, obs))
            if info["arrive"]:
                arrived = True
                break
        ok_c += int(arrived and minc > 0.0)
        mpc.rda._be.close()
    runs = [lidar_closed_loop(s_, lmz_central=1e-3) for s_ in starts]
    ok_l = sum(a and not c and mc > 0.0 for a, c, mc, _ in runs)
    print(f"robust mode (central path at mu = 1e-3) on the GPU: corridor {ok_c} / 16, lidar {ok_l} / 16")
    assert ok_c >= 13 and ok_l >= 13, (ok_c, ok_l)
