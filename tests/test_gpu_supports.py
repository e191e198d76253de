"""-m gpu : the remembered supports of the LamMuZ rows (a cache: csrc/rda_hip.hip Dev::hint) follow their obstacles through the
re-binding of a re-sorted scene and the horizon through the tick - pinned by what they are for: how many rows of the FIRST LamMuZ
launch of a tick lose their support and go to the work-list kernel (split launch form, `rda_debug_worklist`)."""
import ctypes as C

import numpy as np
import pytest

from rda_planner_amd import scenarios as sc

pytestmark = pytest.mark.gpu


def _first_launch_worklist(order, moving, n_obs=600, T=20, steps=14):
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd._lib import hip_api
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [44, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(40, 40), seed=sc.SEED, keep_clear=clear, clear_radius=3.2, moving=moving)
    # iter_num = 1: the only LamMuZ launch of a step is the first one of its tick, and the counter read after the step is its work list
    mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, receding=T, iter_num=1, max_edge_num=4, max_obs_num=n_obs,
              ro1=200, obstacle_order=order)
    lib = hip_api().lib
    lib.rda_debug_worklist.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert lib.rda_lammuz_kernel(mpc.rda._be.handle) is not None
    state = path[0].copy().reshape(3, 1)
    rows, hist = C.c_int(0), []
    for k in range(steps):
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
        u, info = mpc.control(state, 4.0, list(cur))
        assert info["status"] == 0
        state = sc.kinematic_step(state, u, car_t, 0.1)
        assert lib.rda_debug_worklist(mpc.rda._be.handle, C.byref(rows)) == 0
        hist.append(rows.value)
    return hist, n_obs * T


def test_supports_follow_the_obstacles_of_a_resorted_scene():
    """obstacle_order=True (the reference's default): the scene is sorted by distance on every tick, most slots are re-bound when two
    obstacles swap ranks; keyed by slot, 80 % of the rows lost their support in the first launch of every tick"""
    hist, rows = _first_launch_worklist(order=True, moving=False)
    assert hist[0] == rows                      # nothing remembered on the very first step: every row is enumerated
    assert np.median(hist[4:]) <= 0.04 * rows, hist


def test_supports_follow_the_horizon_with_a_fixed_binding():
    """the first launch of a tick reads the support of stage t+1 of the previous tick (5 % -> 0.3 % of the rows at N = 2000)"""
    hist, rows = _first_launch_worklist(order=False, moving=False)
    assert np.median(hist[4:]) <= 0.02 * rows, hist


def test_supports_with_moving_obstacles_stay_mostly_valid():
    hist, rows = _first_launch_worklist(order=True, moving=True)
    assert np.median(hist[4:]) <= 0.10 * rows, hist


@pytest.mark.parametrize("n_obs,order,moving,circles", [(60, True, False, False), (40, False, True, False), (600, True, False, False), (600, False, True, False),
                                                        (200, True, True, False), (60, True, True, True), (200, True, False, True), (600, False, True, True)])
def test_flushing_the_supports_cache_mid_loop_changes_no_bit(n_obs, order, moving, circles):
    """the remembered supports are NOT part of rda_get_state - results may not depend on them.  Two identical loops;
    one forgets every remembered support (rda_debug_flush_supports) before every third step: controls, states, residuals, iteration
    counts and the whole dual state must agree bit for bit.  A remembered support is accepted on the optimality certificate of the FULL
    problem with STRICT complementarity (lammuz_device.h, certify): where a row outside the support has a zero gradient the optimum is
    described by two supports whose closed forms agree to 1-2 ulp only (found by this test at 600 obstacles: 3 of 12 600 rows at the first
    flush, 1e-16 in the duals, before the rule was strict) - such rows go to the enumeration, which ranks by (cost, candidate id).
    600 obstacles: the split launch form (common path + work list)."""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd._lib import hip_api
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [44, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(40, 40), seed=sc.SEED + 3, keep_clear=clear, clear_radius=3.2, moving=moving)
    if circles:        # two of three obstacles as circles (norm2 cone; remembered supports since round 5: lammuz_device.h certify_circle)
        obstacles = [o if i % 3 == 0 else sc.circle(float(o.vertex[0].mean()), float(o.vertex[1].mean()), 0.7, tuple(o.velocity.ravel()))
                     for i, o in enumerate(obstacles)]
    kw = dict(sample_time=0.1, time_print=False, receding=20, iter_num=3, max_edge_num=4, max_obs_num=n_obs, ro1=200, obstacle_order=order)
    a = MPC(car_t, [p.copy() for p in path], **kw)
    b = MPC(car_t, [p.copy() for p in path], **kw)
    lib = hip_api().lib
    lib.rda_debug_flush_supports.argtypes = [C.c_void_p]
    state = path[0].copy().reshape(3, 1)
    for k in range(18):
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive" else o._replace(center=o.center + o.velocity * (0.1 * k))
                                            for o in obstacles]
        if k % 3 == 2:
            assert lib.rda_debug_flush_supports(b.rda._be.handle) == 0
        ua, ia = a.control(state.copy(), 4.0, list(cur))
        ub, ib = b.control(state.copy(), 4.0, list(cur))
        assert ia["iters"] == ib["iters"] and ia["resi_dual"] == ib["resi_dual"] and ia["resi_pri"] == ib["resi_pri"], k
        assert np.array_equal(ua, ub) and np.array_equal(a.cur_vel_array, b.cur_vel_array), (k, float(np.abs(ua - ub).max()))
        state = sc.kinematic_step(state, ua, car_t, 0.1)
    sa, sb = a.rda.get_state(), b.rda.get_state()
    for key in ("lam", "mu", "z", "xi", "zeta"):
        assert np.array_equal(sa[key], sb[key]), key
