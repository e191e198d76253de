"""-m gpu : the LANDING of the su interior point (rda_opts::su_land, round 6; oracle mirror orc_set_su_land).

The interior point stops ON the central path: a row that is only just active (multiplier lam* ~ 1e-7) keeps the slack mu / lam*, so two iterations that stop
at different mu - the kernel's warm Riccati iteration and the oracle's cold dense one - hand back controls up to 1e-4 apart; that, not rounding, is why the
stated tolerance of rounds 3-5 was TOL_U = 5e-4 (tests/helpers.py: TOL_U_IP today).  With the landing (the default since round 6) both sides run the interior point only until the active set
can be read off and then compute the vertex itself (active rows as equalities, verified on the true objective), and the answer no longer depends on the
path: TOL_U_LANDED below, five orders of magnitude tighter, on the su-problems themselves and on closed loops at the BASELINE sizes against the COLD oracle.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import helpers as hp
from rda_planner_amd import scenarios as sc
from rda_planner_amd._capi import dptr

pytestmark = pytest.mark.gpu

TOL_U_LANDED = 1e-9            # applied control and whole horizon, GPU (landed) vs cold oracle (landed); measured 5e-13 ... 3e-11 (printed by the tests)


@pytest.fixture()
def landed_cold_orc(orc):
    orc.lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    orc.lib.orc_set_su_warm(0.0, 0.0, 0)
    orc.lib.orc_set_su_land(1)
    orc.lib.orc_set_threads(16)
    yield orc
    orc.lib.orc_set_su_land(1)                      # (the default since round 6)
    orc.lib.orc_set_su_warm(1e-3, 1e-3, 30)
    orc.lib.orc_set_threads(1)


def _hip_landed(hip):
    from rda_planner_amd.rda_solver import hip_options
    o = hip_options(su_land=1)
    return lambda *a: hip.lib.rda_su_solve_opts(a[0], C.byref(o), *a[1:])


@pytest.mark.parametrize("T,N,dyn,ro1", [(20, 200, 0, 200), (30, 200, 0, 200), (25, 100, 1, 300), (10, 24, 2, 200), (15, 40, 0, 100), (40, 30, 1, 200)])
def test_landed_su_solve_equals_the_landed_oracle(landed_cold_orc, hip, T, N, dyn, ro1):
    """the su-problem hooks, both cold, both landed: s, u, d to 1e-9 (measured 7e-13 ... 6e-11; default mode: 1e-6, tests/test_gpu_baseline_sizes.py)"""
    rng = np.random.default_rng(T * 11 + N)
    cfg = hp.make_cfg(T=T, N=N, dynamics=dyn, ro1=ro1)
    fn = _hip_landed(hip)
    worst = 0.0
    for _ in range(4):
        si = hp.su_inputs(rng, cfg)
        so = hp.su_solve(landed_cold_orc.lib.orc_su_solve, cfg, si)
        sh = hp.su_solve(fn, cfg, si)
        assert so[0] == 0 and sh[0] == 0
        assert landed_cold_orc.lib.orc_get_su_landed() == 1
        worst = max(worst, max(float(np.abs(so[k] - sh[k]).max()) for k in (1, 2, 3)))
    print(f"T={T} N={N} dyn={dyn}: |(s, u, d)_gpu - oracle|, both landed: {worst:.2e}")
    assert worst < 1e-9, worst


def test_landed_hard_fixtures(landed_cold_orc, hip):
    """the recorded hard su-problems (cycling rows, end-game noise, ...): landed on both sides where the landing is accepted, and equal then"""
    fn = _hip_landed(hip)
    root = os.path.join(os.path.dirname(__file__), "golden", "su_hard")
    n_landed = 0
    for path in sorted(glob.glob(os.path.join(root, "*.npz"))):
        cfg, si = hp.load_su_case(path)
        so = hp.su_solve(landed_cold_orc.lib.orc_su_solve, cfg, si)
        landed = landed_cold_orc.lib.orc_get_su_landed()
        sh = hp.su_solve(fn, cfg, si)
        assert so[0] == 0 and sh[0] == 0, path
        d = max(float(np.abs(so[k] - sh[k]).max()) for k in (1, 2, 3))
        print(f"{os.path.basename(path)}: oracle landed {landed}, |(s, u, d)_gpu - oracle| {d:.2e}")
        n_landed += landed
        assert d < (1e-9 if landed else 5e-5), (path, d)
    assert n_landed >= 10


def _closed_loop(car_t, path, obstacles, kw, steps, advance=False):
    from oracle.oracle_backend import oracle_backend
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd.rda_solver import hip_options
    cpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, _backend=oracle_backend, **kw)
    gpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(su_land=1), **kw)
    state = path[0].copy().reshape(3, 1)
    worst, flips = 0.0, 0
    for i in range(steps):
        cur = obstacles if not advance else [o._replace(vertex=o.vertex + o.velocity * (0.1 * i)) for o in obstacles]
        uc, ic = cpu.control(state.copy(), 4.0, list(cur))
        ug, ig = gpu.control(state.copy(), 4.0, list(cur))
        assert ic["status"] == 0 and ig["status"] == 0, (i, ic["status"], ig["status"])
        if ic["iters"] == ig["iters"]:
            worst = max(worst, float(np.abs(uc - ug).max()), float(np.abs(cpu.cur_vel_array - gpu.cur_vel_array).max()))
        else:
            flips += 1
        gpu.rda.set_state(cpu.rda.get_state())
        gpu.cur_vel_array = cpu.cur_vel_array.copy()
        gpu._dev_u = None
        state = sc.kinematic_step(state, uc, car_t, 0.1)
    st = (C.c_int32 * 6)()
    assert gpu.rda._be.api.lib.rda_debug_su_land_n(gpu.rda._be.handle, st, 6) == 0
    return worst, flips, list(st)


@pytest.mark.parametrize("name,n_obs,T,moving,steps", [("north star", 200, 20, False, 40), ("C4", 200, 30, True, 20), ("C5 shape", 100, 25, False, 24), ("N=2000", 2000, 20, False, 10)])
def test_landed_closed_loop_vs_landed_cold_oracle(landed_cold_orc, name, n_obs, T, moving, steps):
    from test_gpu_baseline_sizes import _workload
    car_t, path, obstacles, kw = _workload(n_obs, T, steps + 10, moving=moving)
    worst, flips, st = _closed_loop(car_t, path, obstacles, kw, steps, advance=moving)
    print(f"{name} T={T} N={n_obs} moving={moving}, {steps} steps, both landed: max |du| over the horizon {worst:.2e}, ADMM-count flips {flips}; "
          f"GPU landings accepted {st[0]}, refused {st[1]}, rounds {st[2]}, passes {st[3]}; speculative (su_land_first = 2) tried {st[4]}, accepted {st[5]}")
    assert worst <= TOL_U_LANDED and flips == 0, (worst, flips)
    # every solve landed (some at a later stop of the interior point: a refusal is a retry; a refused SPECULATIVE landing - from the start of a warm attempt,
    # before any interior-point iteration - hands over to the interior point, whose iterate is landed later)
    assert st[0] > 0 and st[1] - (st[4] - st[5]) <= 0.1 * st[0], st


def _gpu_loop(car_t, path, obstacles, kw, steps, advance, **opts):
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd.rda_solver import hip_options
    gpu = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, hip_opts=hip_options(**opts), **kw)
    state = path[0].copy().reshape(3, 1)
    us, its, ipm = [], [], 0
    for i in range(steps):
        cur = obstacles if not advance else [o._replace(vertex=o.vertex + o.velocity * (0.1 * i)) for o in obstacles]
        u, info = gpu.control(state.copy(), 4.0, list(cur))
        assert info["status"] == 0
        us.append(gpu.cur_vel_array.copy()); its.append(info["iters"]); ipm += info["su_ipm_iters"]
        state = sc.kinematic_step(state, u, car_t, 0.1)
    st = (C.c_int32 * 6)()
    assert gpu.rda._be.api.lib.rda_debug_su_land_n(gpu.rda._be.handle, st, 6) == 0
    return np.array(us), its, ipm, list(st)


@pytest.mark.parametrize("name,n_obs,T,moving,steps", [("north star", 200, 20, False, 40), ("C4", 200, 30, True, 16), ("C5 shape", 100, 25, False, 30)])
def test_landing_first_reaches_the_same_vertex_with_fewer_interior_point_iterations(name, n_obs, T, moving, steps):
    """rda_opts::su_land_first (round 6).  Mode 1 (the first pass of a warm attempt is a light one) is bit-identical to mode 0.  Mode 2 (the default: where the start
    of a warm attempt does not meet the landing's stop, the landing is tried from the start with the active set of the kept multipliers - the warm-started
    active-set method) takes another path to the SAME vertex: accepted only on the verified optimality conditions of the true problem, so whole closed loops -
    free-running, never re-synchronised - agree to 1e-9 with the same ADMM iteration counts, and the static sizes spend fewer interior-point iterations."""
    from test_gpu_baseline_sizes import _workload
    car_t, path, obstacles, kw = _workload(n_obs, T, steps + 10, moving=moving)
    u0, it0, ipm0, st0 = _gpu_loop(car_t, path, obstacles, kw, steps, moving, su_land_first=0)
    u1, it1, ipm1, st1 = _gpu_loop(car_t, path, obstacles, kw, steps, moving, su_land_first=1)
    u2, it2, ipm2, st2 = _gpu_loop(car_t, path, obstacles, kw, steps, moving, su_land_first=2)
    d2 = float(np.abs(u2 - u0).max())
    print(f"{name}: interior-point iterations per step {ipm0 / steps:.2f} (mode 0) / {ipm1 / steps:.2f} (1) / {ipm2 / steps:.2f} (2); speculative landings tried {st2[4]}, "
          f"accepted {st2[5]}; max |du| over the horizon, mode 2 vs mode 0: {d2:.2e}")
    assert np.array_equal(u1, u0) and it1 == it0 and ipm1 == ipm0                  # the light first pass changes nothing
    assert st0[4] == 0 and st1[4] == 0
    assert it2 == it0 and d2 <= 1e-9, (d2, it2, it0)
    if not moving:
        assert st2[5] > 0 and ipm2 < ipm0, (st2, ipm2, ipm0)                        # static sizes: some speculative landings are accepted and save iterations
