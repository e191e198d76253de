"""-m gpu : the solver history of a handle (VERDICT r02 4b / ADVICE): `Ctrl::su_last`, `su_probe`, `prev_unconv`, `su_hardlike` and the kept multipliers of the last
su-solve pick the START of the next interior-point solve (easy / moderate / cold).  They are not reference-visible state, but two
handles that agree in everything the reference can see return the same controls only if they agree in this too - so it is part of
the state accessors (rda_get_su_history / rda_set_su_history) and cleared by rda_reset."""
import ctypes as C

import numpy as np
import pytest

from rda_planner_amd import scenarios as sc
from rda_planner_amd._capi import dptr, iptr

pytestmark = pytest.mark.gpu


def _solver(N=24, T=12):
    from rda_planner_amd.rda_solver import RDA_solver
    car_t = sc.rectangle_robot(dynamics="acker")
    return car_t, RDA_solver(T, car_t, 4, N, iter_num=3, time_print=False, ro1=200)


def _inputs(T, k):
    rng = np.random.default_rng(100 + k)
    nom_u = np.vstack([np.full(T, 3.0) + rng.uniform(-0.2, 0.2, T), rng.uniform(-0.1, 0.1, T)])
    nom_s = np.zeros((3, T + 1)); nom_s[:, 0] = [0.3 * k, 0.02 * k, 0.0]
    for t in range(T):
        nom_s[:, t + 1] = nom_s[:, t] + 0.1 * np.array([nom_u[0, t] * np.cos(nom_s[2, t]), nom_u[0, t] * np.sin(nom_s[2, t]), nom_u[0, t] * np.tan(nom_u[1, t]) / 3.0])
    ref = [np.array([[0.3 * k + 0.4 * t], [0.0], [0.0]]) for t in range(T + 1)]
    return nom_s, nom_u, ref


def _obstacles(N):
    from rda_planner_amd.mpc import MPC
    obstacles = sc.scene_polygons(N, lo=(3, -7), hi=(20, 7), seed=3)
    conv = MPC.__new__(MPC)
    conv.receding, conv.dt, conv.state = 12, 0.1, np.zeros((3, 1))
    return MPC.convert_rda_obstacle(conv, obstacles, np.zeros((3, 1)), False)


def _history(api, s):
    hist, keep = np.zeros(8, np.int32), np.zeros(10 * s.T)          # RDA_SU_HISTORY_INTS = 8 since round 6 (+ the credit of the speculative landings, the landing level key, the easy-landing count, the gate of the blind landings)
    assert api.get_su_history(s._be.handle, iptr(hist), dptr(keep)) == 0
    return hist, keep


def test_counted_history_accessors_serve_callers_of_other_header_versions(hip):
    """`hist` grew from 2 to 4 ints in round 5 with no count in the signature.  The `_n` forms take the caller's own count: a round-4 caller
    (2 ints) reads / writes its two entries and nothing past its buffer; a caller with MORE entries than this library has reads zeros for them."""
    T, N = 12, 24
    _, a = _solver(N, T)
    rl = _obstacles(N)
    for k in range(4):
        a.iterative_solve(*_inputs(T, k), 4.0, list(rl))
    full, _ = _history(hip, a)
    two = np.full(4, -7, np.int32)
    assert hip.get_su_history_n(a._be.handle, iptr(two), 2, None) == 0
    assert np.array_equal(two[:2], full[:2]) and (two[2:] == -7).all()          # nothing written behind the caller's two ints
    ten = np.full(10, -7, np.int32)
    assert hip.get_su_history_n(a._be.handle, iptr(ten), 10, None) == 0
    assert np.array_equal(ten[:8], full) and (ten[8:] == 0).all()
    mod = np.array([3, 1], np.int32)
    assert hip.set_su_history_n(a._be.handle, iptr(mod), 2, None) == 0            # the other two keys keep their values
    after, _ = _history(hip, a)
    assert np.array_equal(after[:2], mod) and np.array_equal(after[2:], full[2:])
    assert hip.get_su_history_n(a._be.handle, iptr(two), -1, None) != 0


def test_two_handles_with_the_same_state_and_history_return_the_same_controls(hip):
    T, N = 12, 24
    car_t, a = _solver(N, T)
    _, b = _solver(N, T)
    _, c = _solver(N, T)
    rl = _obstacles(N)
    for k in range(6):                                   # give A a past
        a.iterative_solve(*_inputs(T, k), 4.0, list(rl))
    hist, keep = _history(hip, a)
    assert hist[0] != 99 and np.abs(keep).max() > 0       # A has solver history ...
    h0, k0 = _history(hip, b)
    assert h0[0] == 99 and not h0[1:].any() and not k0.any()    # ... a fresh handle has none
    st = a.get_state()
    b.set_state(st); c.set_state(st)
    assert hip.set_su_history(b._be.handle, iptr(hist), dptr(keep)) == 0          # B: state AND history; C: state only
    worst_b = worst_c = 0.0
    same_ipm_b = True
    for k in range(6, 12):
        ua, ia = a.iterative_solve(*_inputs(T, k), 4.0, list(rl))
        ub, ib = b.iterative_solve(*_inputs(T, k), 4.0, list(rl))
        uc, ic = c.iterative_solve(*_inputs(T, k), 4.0, list(rl))
        assert ia["iters"] == ib["iters"]
        same_ipm_b = same_ipm_b and ia["su_ipm_iters"] == ib["su_ipm_iters"]
        worst_b = max(worst_b, float(np.abs(ua - ub).max()))
        worst_c = max(worst_c, float(np.abs(ua - uc).max()))
        if k == 6:
            first_ipm = (ia["su_ipm_iters"], ic["su_ipm_iters"])
    print(f"same state + history: max |du| {worst_b:.2e} (same interior-point iteration counts: {same_ipm_b}); state only: max |du| {worst_c:.2e}, "
          f"interior-point iterations of the first step {first_ipm}")
    # set_state re-condenses the terms from the duals (rounding level), the interior-point paths are the same ones
    assert same_ipm_b and worst_b <= 1e-9
    assert first_ipm[1] > first_ipm[0]                     # without the history the first solve starts cold: more iterations
    for s_ in (a, b, c):
        s_._be.close()


def test_reset_clears_the_solver_history(hip):
    T, N = 12, 24
    _, a = _solver(N, T)
    rl = _obstacles(N)
    for k in range(4):
        a.iterative_solve(*_inputs(T, k), 4.0, list(rl))
    assert _history(hip, a)[0][0] != 99
    a.reset()
    hist, keep = _history(hip, a)
    assert hist[0] == 99 and not hist[1:].any() and not keep.any()
    a._be.close()
