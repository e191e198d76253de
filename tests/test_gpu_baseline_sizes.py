"""-m gpu : parity at the sizes BASELINE.json names, against the COLD oracle.

The HIP path runs with its defaults (device-side obstacle pipeline and tracking, pipelined tick, interior-point warm
starts of the su-problem within a step and across steps); the oracle runs with `orc_set_su_warm(0, 0, 0)`: every
su-problem starts cold, like ECOS in the reference (no warm start, rda_solver.py:693).  The two therefore reach the
(unique) su solution along DIFFERENT iteration paths, which is what makes this an accuracy statement rather than a
same-code-path identity (that identity is `tests/test_gpu_parity.py`, warm oracle, rounding level).

Stated fp64 tolerance of the closed-loop parity (state re-synchronised to the oracle every step), ONE number for every closed-loop
HIP-vs-oracle test (tests/helpers.py, where the reason for its size is written down; tests/test_gpu_soak.py asserts it on random scenes):
    applied control  |u_gpu - u_oracle|  <=  TOL_U = 1e-6   (speed, m/s; steering / yaw rate / heading, rad; rounds 3-5: 5e-4 - both sides land the su solve now)
    whole horizon    2 x T controls       <=  TOL_U
    residuals        relative             <=  1e-4
  (measured on THESE fixed scenes: <= 3e-11 since round 6 (3e-5 before the landing), and what is ASSERTED on them is TOL_U_FIXED = 1e-7, so that a regression of the su kernel on
   the BASELINE sizes cannot hide inside the randomised soak's bound; the tests print their values with -s)
    ADMM iteration counts equal on >= 95 % of the steps (the early-stop test `resi < 0.2` may flip when a residual
    sits within 1e-6 of the threshold).  A step on which the counts differ is NOT skipped: its applied control must agree
    to TOL_U_FLIP = 5e-2 (the two sides ended one ADMM iteration apart; below the stop threshold one more iteration moves
    the control by that order), and the `..._every_iteration` variants run the same loops with the early stop switched
    off (iter_threshold = 0: both sides run all iter_num iterations on every step), so that EVERY step of the loop is
    compared at TOL_U with no exclusion.
Measured values are printed by the tests (run with -s).
"""
import ctypes as C

import numpy as np
import pytest

from rda_planner_amd import scenarios as sc
from rda_planner_amd._capi import Info, dptr

from helpers import TOL_U, TOL_U_FIXED, TOL_U_FLIP

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cold_orc(orc):
    orc.lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    orc.lib.orc_set_threads.argtypes = [C.c_int]
    orc.lib.orc_set_su_warm(0.0, 0.0, 0)
    orc.lib.orc_set_threads(16)
    yield orc
    orc.lib.orc_set_su_warm(1e-3, 1e-3, 30)
    orc.lib.orc_set_threads(1)


def _workload(n_obs, T, n_steps, moving=False, seed_offset=0, iter_num=4, iter_threshold=0.2):
    """bench.py's workload: acker rectangle robot, straight path through a seeded polygon field"""
    car_t = sc.rectangle_robot(dynamics="acker")
    length = max(40.0, 0.4 * n_steps + 12.0)
    path = sc.line_path([4, 25, 0], [4 + length, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(4 + length - 4, 40), seed=sc.SEED + seed_offset, keep_clear=clear,
                                  clear_radius=3.2, moving=moving)
    kw = dict(receding=T, iter_num=iter_num, max_edge_num=4, max_obs_num=n_obs, ro1=200, obstacle_order=True, iter_threshold=iter_threshold)
    return car_t, path, obstacles, kw


def _closed_loop_vs_cold_oracle(car_t, path, obstacles, kw, steps, advance=False):
    """returns (worst |du0|, worst |du| over the horizon, worst residual error) over the steps with EQUAL iteration counts, the fraction
    of such steps, and the worst |du0| over the steps whose counts differ (0 if none)"""
    from oracle.oracle_backend import oracle_backend
    from rda_planner_amd.mpc import MPC
    cpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    gpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, **kw)
    state = path[0].copy().reshape(3, 1)
    worst_u0 = worst_u = worst_res = worst_flip = 0.0
    same_iters = 0
    for i in range(steps):
        cur = obstacles if not advance else [o._replace(vertex=o.vertex + o.velocity * (0.1 * i)) for o in obstacles]
        uc, ic = cpu.control(state.copy(), 4.0, list(cur))
        ug, ig = gpu.control(state.copy(), 4.0, list(cur))
        same_iters += int(ic["iters"] == ig["iters"])
        if ic["iters"] == ig["iters"]:
            worst_u0 = max(worst_u0, float(np.abs(uc - ug).max()))
            worst_u = max(worst_u, float(np.abs(cpu.cur_vel_array - gpu.cur_vel_array).max()))
            worst_res = max(worst_res, abs(ic["resi_dual"] - ig["resi_dual"]) / (1 + ic["resi_dual"]), abs(ic["resi_pri"] - ig["resi_pri"]))
        else:
            worst_flip = max(worst_flip, float(np.abs(uc - ug).max()))
        assert ic["status"] == 0 and ig["status"] == 0, (i, ic["status"], ig["status"])
        gpu.rda.set_state(cpu.rda.get_state())               # re-synchronise: isolate the per-step error
        gpu.cur_vel_array = cpu.cur_vel_array.copy()
        gpu._dev_u = None                                    # the device copy of the nominal controls is stale now
        state = sc.kinematic_step(state, uc, car_t, 0.1)
    return worst_u0, worst_u, worst_res, same_iters / steps, worst_flip


def test_north_star_T20_N200_closed_loop_vs_cold_oracle(cold_orc):
    car_t, path, obstacles, kw = _workload(200, 20, 60)
    u0, u, res, same, flip = _closed_loop_vs_cold_oracle(car_t, path, obstacles, kw, 60)
    print(f"NS T=20 N=200, 60 steps: max |du0| {u0:.2e}, max |du| horizon {u:.2e}, residual {res:.2e}, same iteration count {same:.0%}, "
          f"max |du0| on the other steps {flip:.2e}")
    assert u0 <= TOL_U_FIXED and u <= TOL_U_FIXED and res <= 1e-4 and same >= 0.95 and flip <= TOL_U_FLIP


def test_north_star_every_iteration_vs_cold_oracle(cold_orc):
    """early stop off (iter_threshold = 0): all four ADMM iterations on every step, every step compared"""
    car_t, path, obstacles, kw = _workload(200, 20, 40, iter_threshold=0.0)
    u0, u, res, same, flip = _closed_loop_vs_cold_oracle(car_t, path, obstacles, kw, 30)
    print(f"NS T=20 N=200, 30 steps x 4 iterations: max |du0| {u0:.2e}, max |du| horizon {u:.2e}, residual {res:.2e}")
    assert same == 1.0 and u0 <= TOL_U_FIXED and u <= TOL_U_FIXED and res <= 1e-4


def test_c4_dynamic_obs_T30_N200_closed_loop_vs_cold_oracle(cold_orc):
    """BASELINE config C4: 200 moving polygons, T=30 - per-stage (A, b) over the horizon, obstacles advance every tick"""
    car_t, path, obstacles, kw = _workload(200, 30, 40, moving=True)
    u0, u, res, same, flip = _closed_loop_vs_cold_oracle(car_t, path, obstacles, kw, 24, advance=True)
    print(f"C4 T=30 N=200 moving, 24 steps: max |du0| {u0:.2e}, max |du| horizon {u:.2e}, residual {res:.2e}, same iteration count {same:.0%}, "
          f"max |du0| on the other steps {flip:.2e}")
    assert u0 <= TOL_U_FIXED and u <= TOL_U_FIXED and res <= 1e-4 and same >= 0.9 and flip <= TOL_U_FLIP


def test_c4_dynamic_obs_every_iteration_vs_cold_oracle(cold_orc):
    """C4 with the early stop off: every step, all iterations, no exclusion"""
    car_t, path, obstacles, kw = _workload(200, 30, 40, moving=True, iter_threshold=0.0)
    u0, u, res, same, flip = _closed_loop_vs_cold_oracle(car_t, path, obstacles, kw, 12, advance=True)
    print(f"C4 T=30 N=200 moving, 12 steps x 4 iterations: max |du0| {u0:.2e}, max |du| horizon {u:.2e}, residual {res:.2e}")
    assert same == 1.0 and u0 <= TOL_U_FIXED and u <= TOL_U_FIXED and res <= 1e-4


def test_scaling_point_T20_N2000_vs_cold_oracle(cold_orc):
    car_t, path, obstacles, kw = _workload(2000, 20, 40)
    u0, u, res, same, flip = _closed_loop_vs_cold_oracle(car_t, path, obstacles, kw, 20)
    print(f"S8 T=20 N=2000, 20 steps: max |du0| {u0:.2e}, max |du| horizon {u:.2e}, residual {res:.2e}, same iteration count {same:.0%}, "
          f"max |du0| on the other steps {flip:.2e}")
    assert u0 <= TOL_U_FIXED and u <= TOL_U_FIXED and res <= 1e-4 and same >= 0.95 and flip <= TOL_U_FLIP


def test_c5_fleet_64x100_T25_members_vs_cold_oracle(cold_orc, hip):
    """BASELINE config C5: 64 egos x 100 obstacles, T=25, stepped as ONE fleet; EVERY member is checked step by step against its
    own oracle instance (not against a solo HIP run) over 3 fleet steps"""
    from oracle.oracle_backend import oracle_backend
    from rda_planner_amd.fleet import Fleet
    from rda_planner_amd.mpc import MPC
    B, T, N, steps = 64, 25, 100, 3
    members, scenes, twins = [], [], {}
    for e in range(B):
        car_t, path, obstacles, kw = _workload(N, T, 40, moving=(e % 2 == 1), seed_offset=e)
        members.append(MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, **kw))
        scenes.append((car_t, path, obstacles))
        twins[e] = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    fleet = Fleet(members)
    states = [scenes[e][1][0].copy().reshape(3, 1) for e in range(B)]
    worst, worst_flip, same, total = 0.0, 0.0, 0, 0
    for k in range(steps):
        res = fleet.control([s.copy() for s in states], 4.0, [list(scenes[e][2]) for e in range(B)])
        for e in range(B):
            uc, ic = twins[e].control(states[e].copy(), 4.0, list(scenes[e][2]))
            total += 1
            if ic["iters"] == res[e][1]["iters"]:
                same += 1
                worst = max(worst, float(np.abs(uc - res[e][0]).max()), float(np.abs(twins[e].cur_vel_array - members[e].cur_vel_array).max()))
            else:
                worst_flip = max(worst_flip, float(np.abs(uc - res[e][0]).max()))
            members[e].rda.set_state(twins[e].rda.get_state())
            members[e].cur_vel_array = twins[e].cur_vel_array.copy()
            members[e]._dev_u = None
            states[e] = sc.kinematic_step(states[e], uc, scenes[e][0], 0.1)
    print(f"C5 64x100 T=25, {steps} fleet steps, all {B} members: max |du| {worst:.2e}, same iteration count {same}/{total}, max |du0| on the others {worst_flip:.2e}")
    assert worst <= TOL_U_FIXED and same >= 0.97 * total and worst_flip <= TOL_U_FLIP
    fleet.close()


@pytest.mark.parametrize("T,N,dyn,acc,ro1", [(20, 200, 0, 1, 200), (30, 200, 0, 1, 200), (25, 100, 1, 1, 300), (20, 2000, 0, 1, 200)])
def test_su_solve_baseline_shapes_stated_tolerance(orc, hip, T, N, dyn, acc, ro1):
    """the su-problem hooks (both start cold) at the BASELINE shapes: 1e-6 on s, u, d"""
    import helpers as hp
    rng = np.random.default_rng(T * 7 + N)
    cfg = hp.make_cfg(T=T, N=N, dynamics=dyn, accelerated=acc, ro1=ro1)
    worst = 0.0
    for _ in range(3):
        si = hp.su_inputs(rng, cfg)
        so = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
        sh = hp.su_solve(hip.lib.rda_su_solve, cfg, si)
        assert so[0] == 0 and sh[0] == 0
        worst = max(worst, max(float(np.abs(so[k] - sh[k]).max()) for k in (1, 2, 3)))
    assert worst < 1e-6, worst
