"""CPU tests pinning the ORACLE's su-problem solver (reference rda_solver.py:216-231,313-387,831-872,911-947)
against an independent un-condensed formulation solved by scipy (states as variables, dynamics as equalities)."""
import numpy as np
import pytest

import helpers as hp


def _lin(dyn, s, u, dt, L):
    phi, v, psi = s[2], u[0], u[1]
    if dyn == 2:
        phi = u[1]
        A = np.eye(3)
        B = np.array([[np.cos(phi) * dt, -v * np.sin(phi) * dt], [np.sin(phi) * dt, v * np.cos(phi) * dt], [0, 0]])
        C = np.array([phi * v * np.sin(phi) * dt, -phi * v * np.cos(phi) * dt, 0])
        return A, B, C
    A = np.array([[1, 0, -v * dt * np.sin(phi)], [0, 1, v * dt * np.cos(phi)], [0, 0, 1]])
    if dyn == 0:
        B = np.array([[np.cos(phi) * dt, 0], [np.sin(phi) * dt, 0], [np.tan(psi) * dt / L, v * dt / (L * np.cos(psi) ** 2)]])
        C = np.array([phi * v * np.sin(phi) * dt, -phi * v * np.cos(phi) * dt, -psi * v * dt / (L * np.cos(psi) ** 2)])
    else:
        B = np.array([[np.cos(phi) * dt, 0], [np.sin(phi) * dt, 0], [0, dt]])
        C = np.array([phi * v * np.sin(phi) * dt, -phi * v * np.cos(phi) * dt, 0])
    return A, B, C


def _objective(cfg, si, S, U, D):
    """the reference's su cost, literally (rda_solver.py:1011-1032, 846-851, 868, 376-383)"""
    T, N, dyn = cfg.T, cfg.N, cfg.dynamics
    w = np.array([1, 1, 0.0 if dyn == 2 else 1.0])[:, None]
    J = cfg.ws * np.sum(w * (S - si["ref"]) ** 2) + cfg.wu * np.sum((U[0] - si["vref"]) ** 2) - cfg.slack_gain * np.sum(D)
    J += 0.5 * cfg.eps_u * np.sum(U ** 2)
    for t in range(T):
        ph = si["nom_s"][2, t]
        Rm = np.array([[np.cos(ph), -np.sin(ph)], [np.sin(ph), np.cos(ph)]])
        dR = np.array([[-np.sin(ph), -np.cos(ph)], [np.cos(ph), -np.sin(ph)]])
        rot = Rm + dR * (S[2, t + 1] - ph)
        for n in range(N):
            Im = si["a"][n, t] @ S[0:2, t + 1] - si["cc"][n, t] - D[t]
            J += 0.5 * cfg.ro1 * (min(Im, 0) ** 2 if cfg.accelerated else Im ** 2)
            Hm = si["g"][n, t] + si["a"][n, t] @ rot
            J += 0.5 * cfg.ro2 * Hm @ Hm
    return J


@pytest.fixture(autouse=True)
def _interior_point_only(request, orc):
    """this module is about the oracle's interior-point iteration (stop tolerances, cycling instances, start rules): the landing that is default since
    round 6 (orc_set_su_land) is switched off - except in the tests of the landing itself"""
    if "landing" in request.node.name:
        yield
        return
    orc.lib.orc_set_su_land(0)
    yield
    orc.lib.orc_set_su_land(1)


@pytest.mark.parametrize("trial", range(6))
def test_su_against_scipy(orc, trial):
    from scipy.optimize import minimize, LinearConstraint, Bounds
    rng = np.random.default_rng(100 + trial)
    cfg = hp.make_cfg(T=int(rng.integers(3, 7)), N=int(rng.integers(1, 5)), dynamics=trial % 3, accelerated=int(trial != 4),
                      ro1=[200, 300, 200, 1, 200, 300][trial])
    si = hp.su_inputs(rng, cfg)
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st == 0
    T = cfg.T
    ns = 3 * (T + 1)

    def unpack(y):
        return y[:ns].reshape(3, T + 1), y[ns:ns + 2 * T].reshape(2, T), y[ns + 2 * T:]
    rows, rhs = [], []
    nvar = ns + 3 * T
    for r in range(3):
        row = np.zeros(nvar); row[r * (T + 1)] = 1; rows.append(row); rhs.append(si["nom_s"][r, 0])
    for t in range(T):
        A, B, Cc = _lin(cfg.dynamics, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
        for r in range(3):
            row = np.zeros(nvar); row[r * (T + 1) + t + 1] = 1
            for k in range(3):
                row[k * (T + 1) + t] -= A[r, k]
            row[ns + t] -= B[r, 0]; row[ns + T + t] -= B[r, 1]
            rows.append(row); rhs.append(Cc[r])
    cons = [LinearConstraint(np.array(rows), rhs, rhs)]
    rr = []
    for t in range(T - 1):
        for i in range(2):
            row = np.zeros(nvar); row[ns + i * T + t + 1] = 1; row[ns + i * T + t] = -1; rr.append(row)
    ab = np.array([cfg.acce_bound[0], cfg.acce_bound[1]] * (T - 1))
    cons.append(LinearConstraint(np.array(rr), -ab, ab))
    lb = np.r_[np.full(ns, -np.inf), np.full(T, -cfg.max_speed[0]), np.full(T, -cfg.max_speed[1]), np.full(T, cfg.min_sd)]
    ub = np.r_[np.full(ns, np.inf), np.full(T, cfg.max_speed[0]), np.full(T, cfg.max_speed[1]), np.full(T, cfg.max_sd)]
    f = lambda y: _objective(cfg, si, *unpack(y))
    ours = f(np.r_[s.ravel(), u.ravel(), d])
    y0 = np.r_[si["nom_s"].ravel(), si["nom_u"].ravel(), np.full(T, 0.5)]
    r = minimize(f, y0, method="trust-constr", bounds=Bounds(lb, ub), constraints=cons,
                 options={"gtol": 1e-9, "xtol": 1e-11, "maxiter": 3000})
    S, U, D = unpack(r.x)
    # the oracle must be feasible and at least as good as scipy; solutions agree to scipy's accuracy
    assert ours <= r.fun + 1e-6 * (1 + abs(r.fun))
    assert np.abs(U - u).max() < 5e-4 and np.abs(S - s).max() < 5e-4 and np.abs(D - d).max() < 5e-4
    assert (np.abs(u[0]) <= cfg.max_speed[0] + 1e-9).all() and (np.abs(u[1]) <= cfg.max_speed[1] + 1e-9).all()
    assert (np.abs(np.diff(u, axis=1)) <= np.array([[cfg.acce_bound[0]], [cfg.acce_bound[1]]]) + 1e-9).all()
    assert (d >= cfg.min_sd - 1e-9).all() and (d <= cfg.max_sd + 1e-9).all()


def test_su_dynamics_and_first_order_optimality(orc):
    """independent certificate at full size (T=20, N=200): dynamics hold exactly and a projected-gradient
    step from the returned point cannot decrease the cost"""
    rng = np.random.default_rng(5)
    cfg = hp.make_cfg(T=20, N=200)
    si = hp.su_inputs(rng, cfg)
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st == 0 and it < 60
    for t in range(cfg.T):
        A, B, Cc = _lin(0, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
        assert np.abs(s[:, t + 1] - (A @ s[:, t] + B @ u[:, t] + Cc)).max() < 1e-10
    assert np.allclose(s[:, 0], si["nom_s"][:, 0])

    def roll(U):
        S = np.zeros((3, cfg.T + 1)); S[:, 0] = si["nom_s"][:, 0]
        for t in range(cfg.T):
            A, B, Cc = _lin(0, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
            S[:, t + 1] = A @ S[:, t] + B @ U[:, t] + Cc
        return S
    f0 = _objective(cfg, si, s, u, d)
    for k in range(20):
        du = rng.normal(0, 1e-4, u.shape); dd = rng.normal(0, 1e-4, d.shape)
        U2 = np.clip(u + du, -np.array([[10.0], [1.0]]), np.array([[10.0], [1.0]]))
        for t in range(1, cfg.T):                       # keep the rate constraint
            U2[:, t] = np.clip(U2[:, t], U2[:, t - 1] - [1.0, 0.05], U2[:, t - 1] + [1.0, 0.05])
        D2 = np.clip(d + dd, cfg.min_sd, cfg.max_sd)
        assert _objective(cfg, si, roll(U2), U2, D2) >= f0 - 1e-7 * (1 + abs(f0))


def test_su_formerly_cycling_instance(orc):
    """a recorded closed-loop instance (omni, T=10, 33 obstacles) on which the predictor-corrector iteration used to cycle
    with a fixed 0.995 fraction to the boundary (100 iterations, then the restart from the more central point); with the
    adaptive fraction it converges directly, and the result is a minimiser (dynamics exact, no feasible perturbation
    lowers the cost)"""
    import os
    cfg, si = hp.load_su_case(os.path.join(os.path.dirname(__file__), "golden", "su_hard", "omni_T10_N33_restart.npz"))
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st == 0 and it <= 40
    si = dict(si, nom_u=si["nom_u"].reshape(2, -1))

    def roll(U):
        S = np.zeros((3, cfg.T + 1)); S[:, 0] = si["nom_s"][:, 0]
        for t in range(cfg.T):
            A, B, Cc = _lin(2, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
            S[:, t + 1] = A @ S[:, t] + B @ U[:, t] + Cc
        return S
    assert np.abs(roll(u) - s).max() < 1e-9
    f0 = _objective(cfg, si, s, u, d)
    rng = np.random.default_rng(0)
    for k in range(20):
        U2 = np.clip(u + rng.normal(0, 1e-4, u.shape), -np.array([[10.0], [1.0]]), np.array([[10.0], [1.0]]))
        for t in range(1, cfg.T):
            U2[:, t] = np.clip(U2[:, t], U2[:, t - 1] - [1.0, 0.05], U2[:, t - 1] + [1.0, 0.05])
        D2 = np.clip(d + rng.normal(0, 1e-4, d.shape), cfg.min_sd, cfg.max_sd)
        assert _objective(cfg, si, roll(U2), U2, D2) >= f0 - 1e-7 * (1 + abs(f0))


def test_su_rate_rows_trading_places_instance(orc):
    """recorded by the soak run of round 3 (acker, T=15, 45 obstacles, scene 30 step 39): two neighbouring steering-rate rows, both
    active at the solution, used to trade places for ever - w_63 = 1e-8 with lam w / mu = 4e-7 in one iterate, the same for row 66 in
    the next, mu stuck at 3e-6 - and the cold attempt AND the central restart ran into their caps (200 iterations, status 1).  With
    every pair kept at lam w >= 1e-5 mu once a cold attempt has passed 25 iterations the solve converges (32), and the point is a minimiser."""
    import os
    cfg, si = hp.load_su_case(os.path.join(os.path.dirname(__file__), "golden", "su_hard", "acker_T15_N45_rate_rows_cycle.npz"))
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st == 0 and it <= 40, (st, it)
    si = dict(si, nom_u=si["nom_u"].reshape(2, -1))
    f0 = _objective(cfg, si, s, u, d)
    rng = np.random.default_rng(0)

    def roll(U):
        S = np.zeros((3, cfg.T + 1)); S[:, 0] = si["nom_s"][:, 0]
        for t in range(cfg.T):
            A, B, Cc = _lin(0, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
            S[:, t + 1] = A @ S[:, t] + B @ U[:, t] + Cc
        return S
    assert np.abs(roll(u) - s).max() < 1e-9
    for k in range(20):
        U2 = np.clip(u + rng.normal(0, 1e-4, u.shape), -np.array([[10.0], [1.0]]), np.array([[10.0], [1.0]]))
        for t in range(1, cfg.T):
            U2[:, t] = np.clip(U2[:, t], U2[:, t - 1] - [1.0, 0.05], U2[:, t - 1] + [1.0, 0.05])
        D2 = np.clip(d + rng.normal(0, 1e-4, d.shape), cfg.min_sd, cfg.max_sd)
        assert _objective(cfg, si, roll(U2), U2, D2) >= f0 - 1e-7 * (1 + abs(f0))


def test_su_hinge_term_switching_on_and_off_instance(orc):
    """recorded by the soak of the interior-point LamMuZ mode (omni, T=15, 40 obstacles, scene 56 step 61): a hinge term switches on and
    off for ever (5 <-> 6 active terms, period 3, steps of 0.005 along a direction of length 1) - 200 iterations and status 1 in kernel
    AND oracle.  With the hinge terms smoothed over 0.1 sqrt(mu) once a cold attempt has passed 25 iterations the solve converges (37)."""
    import os
    cfg, si = hp.load_su_case(os.path.join(os.path.dirname(__file__), "golden", "su_hard", "omni_T15_N40_hinge_flips.npz"))
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st == 0 and it <= 45, (st, it)
    si = dict(si, nom_u=si["nom_u"].reshape(2, -1))
    f0 = _objective(cfg, si, s, u, d)
    rng = np.random.default_rng(0)

    def roll(U):
        S = np.zeros((3, cfg.T + 1)); S[:, 0] = si["nom_s"][:, 0]
        for t in range(cfg.T):
            A, B, Cc = _lin(2, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
            S[:, t + 1] = A @ S[:, t] + B @ U[:, t] + Cc
        return S
    assert np.abs(roll(u) - s).max() < 1e-9
    for k in range(20):
        U2 = np.clip(u + rng.normal(0, 1e-4, u.shape), -np.array([[10.0], [1.0]]), np.array([[10.0], [1.0]]))
        for t in range(1, cfg.T):
            U2[:, t] = np.clip(U2[:, t], U2[:, t - 1] - [1.0, 0.05], U2[:, t - 1] + [1.0, 0.05])
        D2 = np.clip(d + rng.normal(0, 1e-4, d.shape), cfg.min_sd, cfg.max_sd)
        assert _objective(cfg, si, roll(U2), U2, D2) >= f0 - 1e-7 * (1 + abs(f0))


def test_su_stagnating_dual_residual_is_accepted(orc):
    """recorded instance (omni, T=25, 26 obstacles): with barrier weights lam/w of 1e10 the dual residual cannot fall below
    ~5e-8 relative while the complementarity keeps shrinking until the factorisation breaks down; the second termination
    clause (primal feasible, complementarity 1e-12, dual residual 1e-7) stops there and the point is a minimiser"""
    import os
    cfg, si = hp.load_su_case(os.path.join(os.path.dirname(__file__), "golden", "su_hard", "omni_T25_N26_stagnating_dual.npz"))
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st == 0 and it <= 40
    si = dict(si, nom_u=si["nom_u"].reshape(2, -1))
    f0 = _objective(cfg, si, s, u, d)
    rng = np.random.default_rng(0)

    def roll(U):
        S = np.zeros((3, cfg.T + 1)); S[:, 0] = si["nom_s"][:, 0]
        for t in range(cfg.T):
            A, B, Cc = _lin(2, si["nom_s"][:, t], si["nom_u"][:, t], cfg.dt, cfg.L)
            S[:, t + 1] = A @ S[:, t] + B @ U[:, t] + Cc
        return S
    for k in range(20):
        U2 = np.clip(u + rng.normal(0, 1e-4, u.shape), -np.array([[10.0], [1.0]]), np.array([[10.0], [1.0]]))
        for t in range(1, cfg.T):
            U2[:, t] = np.clip(U2[:, t], U2[:, t - 1] - [1.0, 0.05], U2[:, t - 1] + [1.0, 0.05])
        D2 = np.clip(d + rng.normal(0, 1e-4, d.shape), cfg.min_sd, cfg.max_sd)
        assert _objective(cfg, si, roll(U2), U2, D2) >= f0 - 1e-7 * (1 + abs(f0))


WEAK = ["omni_T15_N30_weakly_active_a", "omni_T15_N13_weakly_active_b", "acker_T15_N27_weakly_active_c"]


@pytest.mark.parametrize("name", WEAK)
def test_stop_tolerance_vs_weakly_active_rows(orc, name):
    """Where the stated closed-loop tolerance TOL_U (tests/helpers.py) comes from.  Three su-problems recorded on the steps of the round-4
    soak (GPU vs cold oracle, 9600 steps) with the largest control differences: each has inequality rows that are only just active at
    the solution (multipliers ~1e-4), so the central-path point at complementarity mu lies ~mu / lam* from the solution and the answer
    moves with the STOP tolerance: against a solve at 1e-12 / 1e-12 / 1e-15 the default stop (1e-9 / 1e-10 / 1e-11) lands within 5e-5
    (this is what two different interior-point paths - kernel and oracle - can differ by per su-problem), while a solve at the ECOS-class
    tolerances of the reference's solver (1e-8 throughout) lands 2e-4 ... 3e-3 away.  The TOL_U = 5e-4 of rounds 3-5 (TOL_U_IP today: the interior-point-only
    mode) sits between the two; the landed solve of round 6 is checked against the same three problems below."""
    import os
    import ctypes as C
    orc.lib.orc_set_su_tol.argtypes = [C.c_double] * 3
    cfg, si = hp.load_su_case(os.path.join(os.path.dirname(__file__), "golden", "su_hard", name + ".npz"))
    try:
        orc.lib.orc_set_su_tol(1e-12, 1e-12, 1e-15)
        st_t, _, u_t, _, it_t = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
        orc.lib.orc_set_su_tol(1e-9, 1e-10, 1e-11)
        st_d, _, u_d, _, it_d = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
        orc.lib.orc_set_su_tol(1e-8, 1e-8, 1e-8)
        st_e, _, u_e, _, it_e = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
        # round 6: the LANDED solve (the default; this module switches it off) - the interior point stops at the 1e-3 class, the vertex is computed exactly
        orc.lib.orc_set_su_tol(1e-9, 1e-10, 1e-11)
        orc.lib.orc_set_su_land(1)
        st_l, _, u_l, _, it_l = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
        landed = orc.lib.orc_get_su_landed()
    finally:
        orc.lib.orc_set_su_tol(1e-9, 1e-10, 1e-11)
        orc.lib.orc_set_su_land(0)
    assert st_t == 0 and st_d == 0 and st_e == 0 and it_e <= it_d <= it_t
    d_def, d_ecos = float(np.abs(u_d - u_t).max()), float(np.abs(u_e - u_t).max())
    d_land = float(np.abs(u_l - u_t).max())
    print(f"{name}: default stop {it_d} iterations, |u - u_tight| {d_def:.2e}; ECOS-class stop {it_e} iterations, {d_ecos:.2e}; tight {it_t}; "
          f"landed ({landed}) after {it_l} interior-point iterations: {d_land:.2e}")
    assert d_def <= 5e-5 < hp.TOL_U_IP
    assert d_ecos >= 2e-4
    # the landed answer lies where the tight interior point converges to (which is itself 1e-15 / lam* ~ 1e-11 short of the vertex), with FEWER iterations than the default stop
    assert st_l == 0 and landed == 1 and d_land <= 5e-8 < hp.TOL_U and it_l < it_d


def test_warm_started_su_reaches_the_cold_solution():
    """Inside an MPC step the su-problems of ADMM iterations >= 1 start from the previous multipliers (oracle and kernel share the
    rule).  The su solution is unique, so a step solved with the warm start (default) and without it (orc_set_su_warm(0, 0, 0))
    from the same solver state must coincide to the solver tolerance, with the same ADMM iteration count.  What the warm start
    saves depends on the workload (one interior-point iteration per solve on the N=200 benchmark, nothing on this small scene);
    it must not cost more than a few per cent anywhere."""
    import ctypes as C
    from oracle.oracle_backend import api, oracle_backend
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.mpc import MPC
    lib = api().lib
    lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    lib.orc_set_su_warm.restype = None
    try:
        for dyn in ("acker", "diff", "omni"):
            car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
            path = sc.line_path([4, 20, 0], [20, 20, 0], 0.1)
            clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
            scene = sc.scene_polygons(8, lo=(6, 13), hi=(22, 27), seed=7, keep_clear=clear, clear_radius=2.6)
            kw = dict(receding=10, iter_num=4, max_edge_num=4, max_obs_num=8, _backend=oracle_backend, time_print=False)
            a, b = MPC(car_t, [p.copy() for p in path], **kw), MPC(car_t, [p.copy() for p in path], **kw)
            st = path[0].copy().reshape(3, 1)
            if dyn == "omni":
                st[2, 0] = 0.0
            ipm_w = ipm_c = 0
            for k in range(25):
                lib.orc_set_su_warm(1e-3, 1e-3, 30)
                uw, iw = a.control(st.copy(), 4.0, list(scene))
                lib.orc_set_su_warm(0.0, 0.0, 0)
                uc, ic = b.control(st.copy(), 4.0, list(scene))
                assert iw["iters"] == ic["iters"], (dyn, k)
                # the speed row is pinned by the wu term; the second control (steering / turn rate) enters the cost only through the
                # predicted states, a direction in which the cost is nearly flat: two runs that stop at the same residual tolerance
                # (1e-9 relative) along different iteration paths agree there to ~1e-4 only (3e-5 observed)
                du = np.abs(uw - uc).max(axis=1)
                assert du[0] < 1e-5 and du[1] < 2e-4, (dyn, k, du)
                ipm_w += iw["su_ipm_iters"]; ipm_c += ic["su_ipm_iters"]
                b.rda.set_state(a.rda.get_state())              # the cold run continues from the warm run's state
                b.cur_vel_array, b.cur_index = a.cur_vel_array.copy(), a.cur_index
                st = sc.kinematic_step(st, uw, car_t, 0.1)
            assert ipm_w <= 1.05 * ipm_c, (dyn, ipm_w, ipm_c)
    finally:
        lib.orc_set_su_warm(1e-3, 1e-3, 30)


def test_su_replay_tool_records_and_replays(tmp_path, monkeypatch):
    """tools/su_replay.py (the CPU study aid of DESIGN section 9): the su-problems an oracle closed loop records are the ones orc_su_solve
    gets - replayed cold they converge, and the iteration trace hook prints one line per interior-point iteration"""
    import importlib, os, sys
    monkeypatch.setenv("RDA_SU_REPLAY_DIR", str(tmp_path))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "tools"))
    sys.modules.pop("su_replay", None)
    su_replay = importlib.import_module("su_replay")
    try:
        su_replay.record("t", n_obs=12, T=8, steps=3, moving=True)
        probs = su_replay.load("t")
        assert 3 <= len(probs) <= 12 and probs[0]["T"] == 8 and probs[0]["N"] == 12 and probs[0]["it"] == 0
        lib = su_replay._lib()
        for pr in probs:
            st, its, so, uo, do = su_replay.solve(lib, pr)
            assert st == 0 and 1 <= its <= 40
            assert np.all(np.isfinite(uo)) and np.all(do <= pr["cfg"].max_sd + 1e-9) and np.all(do >= pr["cfg"].min_sd - 1e-9)
    finally:                                   # the recording run switched the oracle's su start rules and thread count: back to the defaults
        lib = su_replay._lib()
        lib.orc_set_su_warm(1e-3, 1e-3, 30); lib.orc_set_threads(1); lib.orc_set_su_dump(b""); lib.orc_set_su_trace(0)


def test_end_game_lost_in_rounding_returns_the_near_converged_iterate(orc):
    """recorded instance (soak seed 9, scene 13, step 15: omni, T=25, 20 slots).  At complementarity 2e-9 the dual residual is 8e-10 -
    one decade of mu short of the stop test - and from there it GROWS (4e-7, 5e-6, ... 3e-5: barrier weights lam/w beyond 1e10) while mu
    falls to 1e-15, where the Cholesky factor breaks down.  Round 4: the central restart ended the same way, and the checker's safety net (the
    best iterate that is primal feasible and within 10 x / 1000 x of the dual / complementarity tolerances - the class ECOS stops at;
    orc_set_su_accept, since round 5 also rda_opts::su_accept of the kernel) returned its remembered iterate instead of `no update`.  Round 5: the
    last-resort attempt is a plain long-step path-following iteration and CONVERGES here (16 + 20 iterations), with or without the net; the net's
    own hand-over is exercised by its test switch (accept = 2: always hand the remembered iterate back).  Either way the point is a minimiser to 1e-6."""
    import ctypes as C
    import os
    cfg, si = hp.load_su_case(os.path.join(os.path.dirname(__file__), "golden", "su_hard", "omni_T25_N20_end_game_noise.npz"))
    orc.lib.orc_set_su_accept.argtypes = [C.c_int]
    try:
        orc.lib.orc_set_su_accept(0)
        st0, _, u0, _, it0 = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
        orc.lib.orc_set_su_accept(2)
        st2, _, u2, _, it2 = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    finally:
        orc.lib.orc_set_su_accept(1)
    st, s, u, d, it = hp.su_solve(orc.lib.orc_su_solve, cfg, si)
    assert st0 == 0 and st == 0 and st2 == 0 and it == it0 == it2 and 20 < it <= 40, (st0, st, st2, it0, it, it2)        # (cold attempt 16, last resort 20)
    assert np.array_equal(u, u0) and 0 < np.abs(u2 - u).max() <= 2e-4           # the remembered iterate: of the looser class, next to the converged one
    si2 = dict(si, nom_u=si["nom_u"].reshape(2, -1))
    f0 = _objective(cfg, si2, s, u, d)
    rng = np.random.default_rng(1)

    def roll(U):
        S = np.zeros((3, cfg.T + 1)); S[:, 0] = si2["nom_s"][:, 0]
        for t in range(cfg.T):
            A, B, Cc = _lin(2, si2["nom_s"][:, t], si2["nom_u"][:, t], cfg.dt, cfg.L)
            S[:, t + 1] = A @ S[:, t] + B @ U[:, t] + Cc
        return S
    for k in range(20):
        U2 = np.clip(u + rng.normal(0, 1e-4, u.shape), -np.array([[10.0], [1.0]]), np.array([[10.0], [1.0]]))
        for t in range(1, cfg.T):
            U2[:, t] = np.clip(U2[:, t], U2[:, t - 1] - [1.0, 0.05], U2[:, t - 1] + [1.0, 0.05])
        D2 = np.clip(d + rng.normal(0, 1e-4, d.shape), cfg.min_sd, cfg.max_sd)
        assert _objective(cfg, si2, roll(U2), U2, D2) >= f0 - 1e-6 * (1 + abs(f0))


def test_warm_start_after_an_unconverged_step_saves_iterations_and_moves_nothing(orc):
    """mirror of rda_opts::su_hard_warm (default since round 5): when the ADMM of the previous step did not converge (here: a caller that re-sorts its
    obstacle list every tick while the duals stay with their slots, quirk Q5), the warm attempts start from a point well inside the boxes
    (slack floor 1) with the previous multipliers and mu0 = 1e-3 - while consecutive su-problems really are far apart (the last solve's first
    iterate had a relative dual residual above 1e-2; without that second key a loop of EASY problems that merely runs out of ADMM iterations
    is locked out of its easy start: a hard-started solve costs three iterations whatever the problem).
    Same su-problems, same stop tolerance: fewer interior-point iterations,
    the controls of the closed loop within 1e-4; a loop whose steps converge never sees the rule (bit-identical)."""
    import ctypes as C
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.mpc import MPC
    from oracle.oracle_backend import oracle_backend
    orc.lib.orc_set_su_hard_warm.argtypes = [C.c_double, C.c_double]
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(200, lo=(8, 10), hi=(40, 40), seed=sc.SEED, keep_clear=clear, clear_radius=3.2)      # (the north-star scene)

    def loop(order, hard, iter_num=4, steps=40):
        orc.lib.orc_set_su_hard_warm(*hard)
        m = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, receding=20, iter_num=iter_num, max_edge_num=4, max_obs_num=200,
                ro1=200, obstacle_order=order, _backend=oracle_backend)
        st, us, ipm, its = path[0].copy().reshape(3, 1), [], 0, 0
        for k in range(steps):
            u, info = m.control(st, 4.0, list(obstacles))
            assert info["status"] == 0
            us.append(u.ravel().copy()); ipm += info["su_ipm_iters"]; its += info["iters"]
            st = sc.kinematic_step(st, u, car_t, 0.1)
        return np.array(us), ipm, its
    try:
        u0, ipm0, its0 = loop(True, (0.0, 0.0))
        u1, ipm1, its1 = loop(True, (1.0, 1e-3))
        f0, fi0, _ = loop(False, (0.0, 0.0))
        f1, fi1, _ = loop(False, (1.0, 1e-3))
        g0, gi0, _ = loop(False, (0.0, 0.0), iter_num=1, steps=25)          # every step 'unconverged' (one ADMM iteration), every su-problem easy
        g1, gi1, _ = loop(False, (1.0, 1e-3), iter_num=1, steps=25)
    finally:
        orc.lib.orc_set_su_hard_warm(1.0, 1e-3)
    print(f"re-sorted loop: {ipm0} -> {ipm1} interior-point iterations over 40 steps ({its0} / {its1} ADMM iterations), max |du| {np.abs(u0 - u1).max():.1e}")
    assert its0 == its1 and ipm1 <= 0.85 * ipm0, (ipm0, ipm1)
    assert np.abs(u0 - u1).max() <= 1e-4
    assert fi0 == fi1 and np.array_equal(f0, f1)
    assert gi1 <= 1.1 * gi0 and np.abs(g0 - g1).max() <= 1e-4, (gi0, gi1)        # not locked out (1.0 -> 3.0 iterations per solve with a key on the last solve's iteration count)


@pytest.mark.parametrize("moving", [False, True])
def test_landing_makes_the_su_answer_independent_of_the_interior_point_path(orc, moving):
    """Round 6 (oracle su_land, mirror of rda_opts::su_land; study: tools/experiments/su_land_oracle.py).  The same closed loop solved along two different
    interior-point paths - warm starts that mirror the kernel's start rules, and cold starts - step by step from the same solver state.  Without the
    landing the two stop at different points of the central path (1e-6 .. 4e-5 apart in the controls: a row that is only just active keeps the slack
    mu / lam*); landed, both end on the same vertex: 1e-10."""
    import ctypes as C
    from oracle.oracle_backend import oracle_backend
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    lib = orc.lib
    lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(24, lo=(8, 17), hi=(40, 33), seed=sc.SEED + 5, keep_clear=clear, clear_radius=2.2, moving=moving)
    kw = dict(receding=12, iter_num=3, max_edge_num=4, max_obs_num=24, ro1=200, obstacle_order=True, time_print=False)
    worst = {}
    try:
        for land in (0, 1):
            lib.orc_set_su_land(land)
            a = MPC(car_t, [p.copy() for p in path], sample_time=0.1, _backend=oracle_backend, **kw)
            b = MPC(car_t, [p.copy() for p in path], sample_time=0.1, _backend=oracle_backend, **kw)
            state, w = path[0].copy().reshape(3, 1), 0.0
            for k in range(16):
                cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obstacles]
                lib.orc_set_su_warm(1e-3, 1e-3, 30)
                ua, ia = a.control(state.copy(), 4.0, list(cur))
                lib.orc_set_su_warm(0.0, 0.0, 0)
                ub, ib = b.control(state.copy(), 4.0, list(cur))
                assert ia["iters"] == ib["iters"]
                w = max(w, float(np.abs(a.cur_vel_array - b.cur_vel_array).max()))
                b.rda.set_state(a.rda.get_state()); b.cur_vel_array = a.cur_vel_array.copy()
                state = sc.kinematic_step(state, ua, car_t, 0.1)
            worst[land] = w
    finally:
        lib.orc_set_su_warm(1e-3, 1e-3, 30); lib.orc_set_su_land(1)
    print(f"moving={moving}: warm path vs cold path, max |du| over the horizon: {worst[0]:.2e} without the landing, {worst[1]:.2e} with it")
    assert worst[1] <= 1e-10 and worst[1] < worst[0]
