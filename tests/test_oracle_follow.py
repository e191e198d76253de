"""CPU mirror of rda_opts::duals_follow (tests/test_gpu_follow.py uses it as the checker): the oracle driven piecewise (orc_admm_*), its
duals re-arranged between the first su-problem and the first LamMuZ pass of a tick the way the device pipeline re-arranges them when it
re-binds the slots.  With every obstacle in a slot, the re-sorted loop with that re-arrangement must equal the loop with a fixed slot
binding - the ADMM state is then a function of the obstacles, not of the slot order.  (Test infrastructure on test infrastructure: no
product code runs here.)"""
import numpy as np

from rda_planner_amd import scenarios as sc
from rda_planner_amd.mpc import MPC
from rda_planner_amd.rda_solver import _Backend
from oracle.oracle_backend import oracle_backend, api as orc_api


from follow_lib import PiecewiseOracle, follow


def test_resorted_loop_with_following_duals_equals_the_fixed_binding():
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [34, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    n = 24
    obstacles = sc.scene_polygons(n, lo=(8, 14), hi=(34, 36), seed=sc.SEED + 21, keep_clear=clear, clear_radius=3.2)
    kw = dict(sample_time=0.1, time_print=False, receding=10, iter_num=4, max_edge_num=4, max_obs_num=n, ro1=200)
    fixed = MPC(car_t, [p.copy() for p in path], _backend=oracle_backend, obstacle_order=False, **kw)
    slot = MPC(car_t, [p.copy() for p in path], _backend=oracle_backend, obstacle_order=True, **kw)
    papi = PiecewiseOracle(orc_api(), kw["iter_num"])
    foll = MPC(car_t, [p.copy() for p in path], _backend=lambda cfg, G, h: _Backend(papi, cfg, G, h), obstacle_order=True, **kw)
    state = path[0].copy().reshape(3, 1)
    binding = {"prev": None}

    def hook():
        objs = foll.convert_rda_obstacle(obstacles, foll.state, False)
        now = np.argsort([foll.rda_obs_distance(o) for o in objs], kind="stable")[:n]
        if binding["prev"] is not None:
            foll.rda.set_state(follow(foll.rda.get_state(), binding["prev"], now))
        binding["prev"] = now
    papi.hook = hook
    worst, it_fixed, it_foll, it_slot, moved = 0.0, [], [], [], 0
    for k in range(30):
        before = None if binding["prev"] is None else binding["prev"].copy()
        ua, ia = fixed.control(state.copy(), 4.0, list(obstacles))
        ub, ib = foll.control(state.copy(), 4.0, list(obstacles))
        _, ic = slot.control(state.copy(), 4.0, list(obstacles))
        assert ia["status"] == 0 and ib["status"] == 0
        moved += int(before is not None and not np.array_equal(before, binding["prev"]))
        it_fixed.append(ia["iters"]); it_foll.append(ib["iters"]); it_slot.append(ic["iters"])
        if ia["iters"] == ib["iters"]:
            worst = max(worst, float(np.abs(ua - ub).max()))
        state = sc.kinematic_step(state, ua, car_t, 0.1)
    assert moved >= 5, moved                                       # the binding did change
    assert worst <= 1e-5, worst
    assert np.sum(np.array(it_fixed) != np.array(it_foll)) <= 1, (it_fixed, it_foll)
    assert np.mean(it_slot[5:]) > np.mean(it_foll[5:]), (np.mean(it_slot[5:]), np.mean(it_foll[5:]))     # slot-bound duals: the quirk the option removes
