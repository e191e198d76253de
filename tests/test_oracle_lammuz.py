"""CPU tests that pin the ORACLE's LamMuZ sub-problem solver (reference rda_solver.py:389-421).

The reference ships no tests and CVXPY/ECOS cannot be installed here (parity unpinned),
so the oracle is pinned by (i) an independent numpy restatement, (ii) KKT certificates on the
reference's own formulation, (iii) scipy cross-checks of the optimal value, (iv) geometry
known-answer tests, (v) committed golden vectors.
"""
import json
import os

import numpy as np
import pytest

import helpers as hp
from oracle.lammuz_np import solve_lammuz
from rda_planner_amd import scenarios as sc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lammuz_golden.json")


def test_c_oracle_matches_numpy_restatement(orc):
    rng = np.random.default_rng(11)
    inp = hp.lammuz_batch_inputs(rng, 300)
    lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp)
    for i in range(300):
        l2, m2, z2, info = solve_lammuz(inp["A"][i], inp["b"][i], bool(inp["cone"][i]), inp["p"][i], inp["phi"][i],
                                        hp.G, hp.H, inp["xi"][i], inp["zeta"][i], inp["dbar"][i], 1.0)
        assert abs(info["cost"] - cmh[i, 0]) < 1e-11
        assert abs(info["m"] - cmh[i, 1]) < 1e-9
        # (lam, mu) individually only where the optimum is not an exact tie
        if np.abs(l2 - lam[i]).max() > 1e-8:
            assert abs(info["cost"] - cmh[i, 0]) < 1e-13      # tie: same cost, different vertex
        else:
            assert np.abs(m2 - mu[i]).max() < 1e-8 and abs(z2 - z[i]) < 1e-9


def _kkt_residual(A, b, p, phi, xi, zeta, dbar, lam, mu, ro2=1.0, delta=1e-6):
    """stationarity / complementarity of the FULL polygon problem (all E + R variables)"""
    c, s = np.cos(phi), np.sin(phi)
    Rm = np.array([[c, -s], [s, c]])
    q = A @ p - b
    M = A @ Rm
    m = lam @ q - hp.H @ mu + zeta - dbar
    Hv = M.T @ lam + hp.G.T @ mu + xi
    dpsi = min(m, 0.0) - delta
    g_lam = dpsi * q + ro2 * (M @ Hv)
    g_mu = -dpsi * hp.H + ro2 * (hp.G @ Hv)
    a = A.T @ lam
    na = np.linalg.norm(a)
    # multiplier of 1/2(|a|^2 - 1) <= 0 from the support rows
    tau = 0.0
    if na > 1 - 1e-9:
        sup = lam > 1e-9
        if sup.any():
            Aa = A[sup] @ a
            tau = float(np.mean(-g_lam[sup] / Aa)) if np.all(np.abs(Aa) > 1e-12) else 0.0
            tau = max(tau, 0.0)
    g_lam = g_lam + tau * (A @ a)
    res = 0.0
    for g, v in ((g_lam, lam), (g_mu, mu)):
        res = max(res, float(np.max(np.maximum(-g, 0))))              # dual feasibility g >= 0
        res = max(res, float(np.max(np.abs(g * v))))                  # complementarity
    assert na <= 1 + 1e-9 and (lam >= 0).all() and (mu >= 0).all()
    return res


def test_kkt_certificate_polygons(orc):
    """the enumeration (tie-break T1 = max clearance, i.e. the delta-perturbed problem) returns a KKT point"""
    rng = np.random.default_rng(3)
    inp = hp.lammuz_batch_inputs(rng, 400, circles=0.0)
    orc.lib.orc_set_centre(0)
    try:
        lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp)
    finally:
        orc.lib.orc_set_centre(1)
    worst = 0.0
    for i in range(400):
        worst = max(worst, _kkt_residual(inp["A"][i], inp["b"][i], inp["p"][i], inp["phi"][i], inp["xi"][i],
                                         inp["zeta"][i], inp["dbar"][i], lam[i], mu[i]))
    assert worst < 1e-7, worst


def test_value_not_worse_than_scipy(orc):
    from scipy.optimize import minimize, NonlinearConstraint, Bounds
    rng = np.random.default_rng(8)
    inp = hp.lammuz_batch_inputs(rng, 25, circles=0.0)
    lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp, delta=1e-3)
    for i in range(25):
        A, b, p, phi = inp["A"][i], inp["b"][i], inp["p"][i], inp["phi"][i]
        c, s = np.cos(phi), np.sin(phi)
        q = A @ p - b
        M = A @ np.array([[c, -s], [s, c]])
        E = A.shape[0]

        def f(y):
            m = y[:E] @ q - y[E:] @ hp.H + inp["zeta"][i] - inp["dbar"][i]
            Hv = M.T @ y[:E] + hp.G.T @ y[E:] + inp["xi"][i]
            return 0.5 * min(m, 0) ** 2 - 1e-3 * m + 0.5 * Hv @ Hv
        cons = [NonlinearConstraint(lambda y: np.sum((A.T @ y[:E]) ** 2), -np.inf, 1.0)]
        best = np.inf
        for k in range(2):
            y0 = np.r_[lam[i], mu[i]] + (rng.uniform(0, 0.2, E + 4) if k else 0)
            r = minimize(f, y0, method="trust-constr", bounds=Bounds(0, np.inf), constraints=cons,
                         options={"gtol": 1e-10, "xtol": 1e-12, "maxiter": 500})
            y = np.maximum(r.x, 0)
            na = np.linalg.norm(A.T @ y[:E])
            if na > 1:
                y[:E] /= na
            best = min(best, f(y))
        assert cmh[i, 0] <= best + 1e-8 * (1 + abs(best)), (i, cmh[i, 0], best)


def test_circle_value_not_worse_than_scipy(orc):
    """norm2 obstacles (cone rda_solver.py:1041-1050): lam = (a, lam_3 <= -||a||), ||a|| <= 1.  Near a predicted
    overlap or with a large xi the optimum has 0 < ||a|| < 1 - the circle-interior candidate must find it."""
    from scipy.optimize import minimize, NonlinearConstraint, Bounds
    rng = np.random.default_rng(21)
    n_interior = 0
    for trial in range(16):
        delta = 10 ** rng.uniform(-6, -2)
        ro2 = float(rng.choice([1.0, 0.3, 5.0]))
        p = rng.uniform(-5, 5, 2)
        phi = rng.uniform(-np.pi, np.pi)
        th = rng.uniform(0, 2 * np.pi)
        cen = p + rng.choice([0.3, 0.8, 2.0, 4.0]) * np.array([np.cos(th), np.sin(th)])
        A = np.array([[1, 0], [0, 1], [0, 0], [0, 0.0]])
        b = np.array([cen[0], cen[1], -rng.uniform(0.3, 1.5), 0])
        xi = rng.normal(0, rng.choice([0, 0.05, 0.5]), 2)
        zeta = rng.normal(0, rng.choice([0, 0.3, 2.0]))
        dbar = rng.uniform(0.1, 1.0)
        inp = dict(A=A[None], b=b[None], cone=np.ones(1, np.int32), p=p[None], phi=np.array([phi]), xi=xi[None],
                   zeta=np.array([zeta]), dbar=np.array([dbar]))
        lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp, ro2=ro2, delta=delta)
        na = np.hypot(lam[0, 0], lam[0, 1])
        assert na <= 1 + 1e-12 and lam[0, 2] <= -na + 1e-12 and (mu >= 0).all()
        n_interior += 1e-9 < na < 1 - 1e-9
        c, s = np.cos(phi), np.sin(phi)
        q = A @ p - b
        M = A @ np.array([[c, -s], [s, c]])

        def f(y):
            l = np.r_[y[:3], 0]
            m = l @ q - y[3:] @ hp.H + zeta - dbar
            Hv = M.T @ l + hp.G.T @ y[3:] + xi
            return 0.5 * min(m, 0) ** 2 - delta * m + 0.5 * ro2 * Hv @ Hv
        cons = [NonlinearConstraint(lambda y: y[0] ** 2 + y[1] ** 2, -np.inf, 1.0),
                NonlinearConstraint(lambda y: -y[2] - np.hypot(y[0], y[1]), 0, np.inf)]
        bnd = Bounds(np.r_[-np.inf, -np.inf, -np.inf, 0, 0, 0, 0], np.full(7, np.inf))
        best = np.inf
        for k in range(2):
            y0 = np.r_[lam[0, :3], mu[0]] + (rng.normal(0, 0.1, 7) if k else 0)
            y0[3:] = np.abs(y0[3:])
            r = minimize(f, y0, method="trust-constr", bounds=bnd, constraints=cons,
                         options={"gtol": 1e-11, "xtol": 1e-13, "maxiter": 1500})
            y = r.x.copy()
            y[3:] = np.maximum(y[3:], 0)
            n2 = np.hypot(y[0], y[1])
            if n2 > 1:
                y[:2] /= n2
                n2 = 1
            y[2] = min(y[2], -n2)
            best = min(best, f(y))
        assert cmh[0, 0] <= best + 1e-8 * (1 + abs(best)), (trial, cmh[0, 0], best)
    assert n_interior >= 3, n_interior


def test_central_normal_rule(orc):
    """tie-break T1 in the slack regime (the default): whenever the max-clearance optimum has m > 0 and H ~ 0, the duals
    returned are an optimal point of the REFERENCE problem (cost 0: H = 0, m >= 0, cone and norm constraints hold) whose
    unit normal sits in the middle of the arc of separating directions - checked against a brute-force scan of that arc"""
    rng = np.random.default_rng(12)
    inp = hp.lammuz_batch_inputs(rng, 300, circles=0.3)
    orc.lib.orc_set_centre(0)
    lam0, mu0, z0, cmh0 = hp.oracle_lammuz_batch(orc, inp)
    orc.lib.orc_set_centre(1)
    lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp)
    n_central = 0
    for i in range(300):
        A, b, p, phi, xi = inp["A"][i], inp["b"][i], inp["p"][i], inp["phi"][i], inp["xi"][i]
        a0 = A.T @ lam0[i]
        slack = cmh0[i, 1] > 0 and cmh0[i, 2] ** 2 + cmh0[i, 3] ** 2 < 1e-8 and a0 @ a0 >= 1 - 1e-9
        changed = np.abs(lam[i] - lam0[i]).max() > 1e-12
        if not slack:
            assert not changed
            continue
        if not changed:
            continue                                         # full-circle arc or a symmetric configuration
        n_central += 1
        c, s = np.cos(phi), np.sin(phi)
        Rm = np.array([[c, -s], [s, c]])
        kappa0 = inp["zeta"][i] - inp["dbar"][i]
        a = A.T @ lam[i]
        assert abs(np.hypot(*a) - 1) < 1e-9 and (mu[i] >= 0).all()
        if inp["cone"][i]:
            assert lam[i, 2] <= -np.hypot(lam[i, 0], lam[i, 1]) + 1e-12
        else:
            assert (lam[i] >= 0).all()
        Hc = (A @ Rm).T @ lam[i] + hp.G.T @ mu[i] + xi
        m = lam[i] @ (A @ p - b) - mu[i] @ hp.H + kappa0
        assert np.abs(Hc).max() < 1e-9 and -1e-12 <= m <= cmh0[i, 1] + 1e-5 and abs(m - cmh[i, 1]) < 1e-9
        assert abs(z[i] - 0.5 * m) < 1e-12                   # T2 on the new m

        def clear(th):                                       # clearance of the unit normal a(th) with H = 0 duals
            aa = np.array([np.cos(th), np.sin(th)])
            if inp["cone"][i]:
                so = aa @ b[0:2] - b[2]                      # support of the disc: c'a + r
            else:
                so = max(aa @ np.array(v[:2]) for v in _verts(A, b))
            g = -(Rm.T @ aa) - xi
            sr = max(g @ np.array(v[:2]) for v in _verts(hp.G, hp.H))
            return aa @ p - so - sr + kappa0
        th0 = np.arctan2(*(A.T @ lam0[i])[::-1])
        thc = np.arctan2(a[1], a[0])
        step = 2 * np.pi / 20000
        up = next(k for k in range(1, 20001) if clear(th0 + k * step) < 0) * step
        dn = next(k for k in range(1, 20001) if clear(th0 - k * step) < 0) * step
        mid = th0 + 0.5 * (up - dn)
        assert abs((thc - mid + np.pi) % (2 * np.pi) - np.pi) < 2 * step, (i, thc, mid)
        if n_central >= 40:
            break
    assert n_central >= 20


def _verts(A, b):
    from oracle.lammuz_np import polygon_vertices
    return polygon_vertices(np.asarray(A, float), np.asarray(b, float).ravel())


KAT = [  # obstacle, robot pose (x, y, phi), distance  - SURVEY.md appendix B
    ("poly", (25, 26, 0.0), 2.2),
    ("poly", (25, 30, 0.7), 4.5949905),
    ("poly", (36, 31, 2.5), 3.3988473),
    ("circ", (12, 34, 0.0), 2.7),
    ("circ", (18, 40, -1.2), 1.0169500),
]


@pytest.mark.parametrize("kind,pose,dist", KAT)
def test_known_answer_distances(orc, kind, pose, dist):
    """with xi = 0 and an inactive hinge the max-clearance duals are the polytope-distance duals:
    m + d - zeta = distance(robot, obstacle)"""
    E = 4
    if kind == "poly":
        A, b = sc.polygon_halfspaces(np.array([[31, 33, 33, 31], [24, 24, 28, 28.0]]))
        A = np.ascontiguousarray(A)
        b = b.ravel()
        cone = 0
    else:
        A = np.array([[1, 0], [0, 1], [0, 0], [0, 0.0]])
        b = np.array([20, 34, -1.5, 0.0])
        cone = 1
    inp = dict(A=A[None], b=b[None], cone=np.array([cone], np.int32), p=np.array([pose[:2]], float), phi=np.array([pose[2]]),
               xi=np.zeros((1, 2)), zeta=np.zeros(1), dbar=np.array([0.1]))
    orc.lib.orc_set_centre(0)
    try:
        lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp)
    finally:
        orc.lib.orc_set_centre(1)
    # max clearance is rewarded with delta = 1e-6, which buys delta*|x_R|^2/ro2 <= 1.6e-5 of extra m for H = -delta/ro2 * x_R
    assert abs(cmh[0, 1] + 0.1 - dist) < 3e-5
    assert abs(np.linalg.norm(A.T @ lam[0]) - 1) < 1e-9


def test_edge_cases(orc):
    rng = np.random.default_rng(2)
    inp = hp.lammuz_batch_inputs(rng, 4, circles=0.0)
    # all-zero obstacle rows (padding only) -> zero duals
    inp["A"][0] = 0
    inp["b"][0] = 0
    # robot deep inside the obstacle -> hinge active, duals shrink towards 0, still finite
    inp["A"][1], inp["b"][1] = hp.random_polygon(rng, inp["p"][1], 4, 8.0, 4)
    lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp)
    assert np.all(lam[0] == 0) and np.isfinite(cmh).all()
    assert np.isfinite(lam).all() and np.isfinite(mu).all() and (z >= 0).all()
    # non-accelerated: z absorbs the whole margin
    l2, m2, z2, c2 = hp.oracle_lammuz_batch(orc, inp, accelerated=0)
    assert np.allclose(z2, np.maximum(c2[:, 1], 0)) and np.allclose(z, 0.5 * np.maximum(cmh[:, 1], 0))


def test_golden_vectors(orc):
    """committed fixtures (tests/golden/make_golden.py) - guards the oracle against silent drift"""
    gold = json.load(open(GOLD))
    inp = {k: np.array(v) for k, v in gold["inputs"].items()}
    inp["cone"] = inp["cone"].astype(np.int32)
    lam, mu, z, cmh = hp.oracle_lammuz_batch(orc, inp)
    assert np.abs(cmh - np.array(gold["cmh"])).max() < 1e-10
    assert np.abs(z - np.array(gold["z"])).max() < 1e-10
    ok = np.array(gold["unique"], bool)
    assert np.abs(lam - np.array(gold["lam"]))[ok].max() < 1e-8
    assert np.abs(mu - np.array(gold["mu"]))[ok].max() < 1e-8


def test_ipm_norm2_robot_certifies_exactly_the_geometric_distance(orc):
    """Known-answer test of the cone conventions of the interior-point restatement (oracle/lmz_ipm.c) for a CIRCLE robot
    (car_tuple.cone_type == 'norm2', rda_solver.py:1034-1039): the sub-problem has optimal value 0 iff duals exist that certify
    `distance(robot, obstacle) >= dbar` (Im >= 0 with H = 0), so the value must be ~0 for dbar just below the true distance and
    positive just above it - for polygon and circle obstacles, any heading."""
    import ctypes as C
    from rda_planner_amd._capi import c_double_p, c_int_p, dptr
    L = orc.lib
    L.orc_lammuz_ipm_one.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, C.c_double, c_double_p,
                                     c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p,
                                     c_double_p, c_double_p, c_int_p]
    L.orc_lammuz_ipm_one.restype = C.c_int
    L.orc_set_lmz_ipm_mu.argtypes = [C.c_double]
    L.orc_set_lmz_ipm_mu(0.0)
    r = 0.8
    G = np.ascontiguousarray([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]])
    h = np.ascontiguousarray([0.0, 0.0, -r])
    rng = np.random.default_rng(2)
    for trial in range(12):
        p = rng.uniform(-3, 3, 2)
        phi = rng.uniform(-3, 3)
        if trial % 2 == 0:          # polygon obstacle: distance from the centre to the polygon minus r
            V = np.array([[4.0, 6.0, 6.5, 4.5], [-1.0, -1.5, 1.0, 1.5]]) + rng.uniform(-1, 1, (2, 1))
            A, b = sc.polygon_halfspaces(V)
            A, b, cone = np.ascontiguousarray(A), np.ascontiguousarray(b.ravel()), 0
            k = V.shape[1]
            dist = min(np.linalg.norm(V[:, i] + np.clip((p - V[:, i]) @ (V[:, (i + 1) % k] - V[:, i]) / np.sum((V[:, (i + 1) % k] - V[:, i]) ** 2), 0, 1)
                                      * (V[:, (i + 1) % k] - V[:, i]) - p) for i in range(k)) - r
            E = 4
        else:                       # circle obstacle
            c0, ro = rng.uniform(4, 7, 2), rng.uniform(0.3, 1.2)
            A = np.ascontiguousarray([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]]); b = np.ascontiguousarray([c0[0], c0[1], -ro]); cone = 1
            dist = np.linalg.norm(c0 - p) - ro - r
            E = 3
        vals = []
        for dbar in (dist - 0.01, dist + 0.01):
            lo, mo, zo, cmh, it = np.zeros(E), np.zeros(3), C.c_double(0), np.zeros(4), C.c_int(0)
            st = L.orc_lammuz_ipm_one(E, 3, dptr(A), dptr(b), cone, 1, dptr(np.ascontiguousarray(p)), float(phi), dptr(G), dptr(h), dptr(np.zeros(2)),
                                      0.0, float(dbar), 50.0, 1, dptr(lo), dptr(mo), C.cast(C.byref(zo), c_double_p), dptr(cmh), C.cast(C.byref(it), c_int_p))
            assert st in (0, 1), (trial, st)
            assert np.hypot(mo[0], mo[1]) <= -mo[2] + 1e-7                      # mu in the robot's cone
            vals.append(cmh[0])
        assert vals[0] < 1e-9 and vals[1] > 2e-5, (trial, dist, vals)
    L.orc_set_lmz_ipm_mu(1e-6)
