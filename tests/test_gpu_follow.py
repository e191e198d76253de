"""rda_opts::duals_follow / RDA_solver(duals_follow_obstacles=True): the dual state moves with its obstacle when the device pipeline
re-binds the slots.  NOT reference semantics (the reference keeps the duals with the slot: SURVEY quirk Q5, the default of this
library) - an opt-in for callers that re-sort their obstacle list every tick, the reference's default caller.

What is pinned here
  * with every obstacle of the scene in a slot the result does not depend on the slot order: re-sorted + follow == fixed binding
    (which the parity suite pins on the oracle) up to the summation order of the su hinge terms;
  * with more obstacles than slots (obstacles enter and leave) the step equals the oracle driven piecewise (orc_admm_*) with the same
    re-arrangement of its duals between the first su-problem and the first LamMuZ pass of the tick;
  * slots staged on the host carry no obstacle identity, sharded handles move no rows between ranks: both are refused."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL_U
from follow_lib import PiecewiseOracle, follow
from rda_planner_amd import scenarios as sc

pytestmark = pytest.mark.gpu


def _scene(n_obs, moving=False, seed=5):
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [54, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 12), hi=(52, 38), seed=sc.SEED + seed, keep_clear=clear, clear_radius=3.2, moving=moving)
    return car_t, path, obstacles


def _at(obstacles, k, moving):
    return obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]


@pytest.mark.parametrize("n_obs,moving", [(60, False), (40, True), (320, False)])
def test_result_does_not_depend_on_the_slot_order(n_obs, moving):
    from rda_planner_amd.mpc import MPC
    car_t, path, obstacles = _scene(n_obs, moving)
    kw = dict(sample_time=0.1, time_print=False, receding=20, iter_num=4, max_edge_num=4, max_obs_num=n_obs, ro1=200)
    fixed = MPC(car_t, [p.copy() for p in path], obstacle_order=False, **kw)
    follow = MPC(car_t, [p.copy() for p in path], obstacle_order=True, duals_follow_obstacles=True, **kw)
    slot = MPC(car_t, [p.copy() for p in path], obstacle_order=True, **kw)              # the reference's semantics, for the iteration counts
    state = path[0].copy().reshape(3, 1)
    worst, it_fixed, it_follow, it_slot = 0.0, [], [], []
    for k in range(40):
        cur = _at(obstacles, k, moving)
        ua, ia = fixed.control(state.copy(), 4.0, list(cur))
        ub, ib = follow.control(state.copy(), 4.0, list(cur))
        _, ic = slot.control(state.copy(), 4.0, list(cur))
        assert ia["status"] == 0 and ib["status"] == 0
        it_fixed.append(ia["iters"]); it_follow.append(ib["iters"]); it_slot.append(ic["iters"])
        if ia["iters"] == ib["iters"]:
            worst = max(worst, float(np.abs(ua - ub).max()), float(np.abs(fixed.cur_vel_array - follow.cur_vel_array).max()))
        state = sc.kinematic_step(state, ua, car_t, 0.1)
    assert worst <= 1e-6, worst
    assert np.sum(np.array(it_fixed) != np.array(it_follow)) <= 1, (it_fixed, it_follow)        # (a residual within rounding of the threshold)
    assert np.mean(it_follow[5:]) <= np.mean(it_fixed[5:]) + 0.05
    assert np.mean(it_slot[5:]) >= np.mean(it_follow[5:]) + 0.5, (np.mean(it_slot[5:]), np.mean(it_follow[5:]))   # what the option is for


def _slot_src(mpc):
    from rda_planner_amd._lib import hip_api
    lib = hip_api().lib
    lib.rda_debug_slot_src.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    N = mpc.rda.max_obs_num
    src, used = (C.c_int32 * N)(), C.c_int32(0)
    assert lib.rda_debug_slot_src(mpc.rda._be.handle, src, C.byref(used)) == 0
    return np.array(src[:used.value], dtype=int)


@pytest.mark.parametrize("n_obs,slots,moving", [(90, 40, False), (70, 30, True)])
def test_obstacles_entering_and_leaving_the_slots_match_the_oracle(n_obs, slots, moving):
    """more obstacles than slots: the nearest `slots` are staged, the set changes as the robot drives.  Per step from the same solver
    state (the oracle continues from the GPU's state, as in the soak)"""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd.rda_solver import _Backend
    from oracle.oracle_backend import api as orc_api
    car_t, path, obstacles = _scene(n_obs, moving, seed=9)
    kw = dict(sample_time=0.1, time_print=False, receding=15, iter_num=3, max_edge_num=4, max_obs_num=slots, ro1=200, obstacle_order=True)
    gpu = MPC(car_t, [p.copy() for p in path], duals_follow_obstacles=True, **kw)
    papi = PiecewiseOracle(orc_api(), kw["iter_num"])
    cpu = MPC(car_t, [p.copy() for p in path], _backend=lambda cfg, G, h: _Backend(papi, cfg, G, h), **kw)
    orc_api().lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    orc_api().lib.orc_set_su_warm(0.0, 0.0, 0)                   # the cold oracle: the independent checker
    try:
        state = path[0].copy().reshape(3, 1)
        prev, worst, changed, flips = None, 0.0, 0, 0
        for k in range(36):
            cur = _at(obstacles, k, moving)
            ug, ig = gpu.control(state.copy(), 4.0, list(cur))
            now = _slot_src(gpu)
            assert len(now) == slots

            def hook(prev=prev, now=now):
                if prev is not None:
                    cpu.rda.set_state(follow(cpu.rda.get_state(), prev, now))
            papi.hook = hook
            uc, ic = cpu.control(state.copy(), 4.0, list(cur))
            assert ig["status"] == 0 and ic["status"] == 0
            if prev is not None:
                changed += int(set(prev.tolist()) != set(now.tolist()))
            if ig["iters"] == ic["iters"]:
                worst = max(worst, float(np.abs(ug - uc).max()))
            else:
                flips += 1
            prev = now
            cpu.rda.set_state(gpu.rda.get_state())
            cpu.cur_vel_array = gpu.cur_vel_array.copy(); cpu.cur_index = gpu.cur_index
            state = sc.kinematic_step(state, ug, car_t, 0.1)
        assert changed >= 5, changed                              # the slot SET did change (obstacles entered / left)
        assert flips <= 1 and worst <= TOL_U, (flips, worst)
    finally:
        orc_api().lib.orc_set_su_warm(1e-3, 1e-3, 30)


def test_host_staged_slots_and_shards_are_refused():
    from rda_planner_amd.rda_solver import RDA_solver
    from rda_planner_amd._lib import hip_api
    car_t, path, obstacles = _scene(6)
    rda = RDA_solver(10, car_t, max_edge_num=4, max_obs_num=6, iter_num=2, time_print=False, duals_follow_obstacles=True)
    lib = hip_api().lib
    A, b, cone = np.zeros((6, 4, 2)), np.zeros((6, 4)), np.zeros(6, dtype=np.int32)
    A[:, :, 0] = 1.0
    rc = hip_api().upload_obstacles(rda._be.handle, 6, A.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                                    cone.ctypes.data_as(C.POINTER(C.c_int)), 0)
    assert rc != 0 and b"unsupported" in lib.rda_strerror(rc).lower()
    rc = hip_api().shard_config(rda._be.handle, 0, 2)
    assert rc != 0 and b"unsupported" in lib.rda_strerror(rc).lower()
    assert hip_api().shard_config(rda._be.handle, 0, 1) == 0


def test_follow_in_the_interior_point_lammuz_mode():
    """lmz_central: the kept central-path points of a re-bound slot are dropped (cold start of that row), the duals follow.  The
    interior-point LamMuZ answers are tolerance-level objects (tests/test_gpu_central.py: 1e-4), so is this comparison."""
    from rda_planner_amd.mpc import MPC
    car_t, path, obstacles = _scene(40)
    kw = dict(sample_time=0.1, time_print=False, receding=15, iter_num=3, max_edge_num=4, max_obs_num=40, ro1=200, lmz_central=1e-3)
    fixed = MPC(car_t, [p.copy() for p in path], obstacle_order=False, **kw)
    follow = MPC(car_t, [p.copy() for p in path], obstacle_order=True, duals_follow_obstacles=True, **kw)
    state = path[0].copy().reshape(3, 1)
    worst = 0.0
    for k in range(25):
        ua, ia = fixed.control(state.copy(), 4.0, list(obstacles))
        ub, ib = follow.control(state.copy(), 4.0, list(obstacles))
        assert ia["status"] == 0 and ib["status"] == 0 and ib["lmz_fail"] == 0
        if ia["iters"] == ib["iters"]:
            worst = max(worst, float(np.abs(ua - ub).max()))
        state = sc.kinematic_step(state, ua, car_t, 0.1)
    assert worst <= 5e-3, worst
