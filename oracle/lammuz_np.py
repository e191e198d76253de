"""
ORACLE (test infrastructure only - never imported by the product path).

Plain-numpy restatement of ONE LamMuZ sub-problem of the reference
(/root/reference/RDA_planner/rda_solver.py:389-421 `LamMuZ_cost_cons`,
:874-909 `Hm_LamMu` / `Im_LamMu`, :1034-1050 cone helpers), i.e. for one
obstacle n and one stage t (dual column t+1):

    minimise   1/2 * neg(Im)^2 + 1/2*ro2*||Hm||^2                 (accelerated)
    Im = lam'(A p - b) - mu'h - d - z + zeta        (rda_solver.py:902-907)
    Hm = G'mu + (A R)'lam + xi                      (rda_solver.py:886)
    s.t. ||A'lam|| <= 1 (:409-416),  lam in K_obs (:1042-1050),
         mu in K_robot (:1034-1039),  z >= 0 (:100)

The reference hands this to CVXPY/ECOS.  The minimiser is NOT unique (the cost
only sees 3 linear functionals of the 9+ unknowns), so ECOS
returns an interior point of the optimal face.  PARITY UNPINNED: cvxpy/ecos are
not installable here, so this restatement fixes the tie-break explicitly:

  (T1) slack regime (some (lam, mu) reaches H = 0 with m >= 0: every such point is optimal): the duals that support
       the UNIT normal in the middle of the arc of all separating directions (`central_normal`); the arc is taken
       around the max-clearance normal a*, which the enumeration below delivers by solving the problem with the extra
       cost -delta*m, delta = 1e-6 (perturbs the reference optimum by O(delta)).  Outside the slack regime the
       optimum is unique and the enumeration result is final;
  (T2) z = 1/2 * max(Im|z=0, 0) in accelerated mode (mid-point of its optimal
       interval [0, Im+]);  z = max(Im|z=0, 0) when accelerated=False, where
       that value is the unique minimiser;
  (T3) pass 1 ranks the hinge-inactive candidates by the hinge-inactive model cost
       (a lower bound of the true cost, exact for m >= 0) and returns its minimiser
       when that has m >= 0 (it is then globally optimal); otherwise pass 2 lets the
       hinge-active candidates and that minimiser compete on the true cost.  Exact
       ties go to the lowest candidate index 2*(lam_index*n_mu + mu_index) + hinge.

Method: because -psi'(m) > 0 everywhere, for the optimal a = A'lam and g = G'mu
the pair (lam, mu) solves two LPs  ->  basic optimal solutions have at most two
non-zero lam_i and two non-zero mu_j.  We enumerate those supports ("closest
features"), solve each small piecewise-quadratic problem in closed form (a 2-D
trust-region sub-problem when the separating direction is free), evaluate the
TRUE cost at every sign-feasible candidate and keep the best.  Only polygon
robots (Rpositive) are covered; circle obstacles (norm2) must be in the
canonical form produced by mpc.py:440-458 (A=[[1,0],[0,1],[0,0]], b=[cx,cy,-r]).
"""
import itertools
import numpy as np

DELTA = 1e-6
SIGN_TOL = 1e-12


def trs2(Q, c, disc):
    """min 1/2 x'Qx + c'x  over ||x||<=1 (disc=True) or ||x||==1 (disc=False).
    Q symmetric PSD 2x2.  Returns list of candidate minimisers (1 or 2)."""
    q11, q12, q22 = Q[0, 0], Q[0, 1], Q[1, 1]
    mean = 0.5 * (q11 + q22)
    dif = 0.5 * (q11 - q22)
    rad = np.hypot(dif, q12)
    l1, l2 = mean - rad, mean + rad
    # eigenvector of l2
    if dif >= 0:
        v2 = np.array([dif + rad, q12])
    else:
        v2 = np.array([q12, rad - dif])
    nv = np.hypot(v2[0], v2[1])
    v2 = v2 / nv if nv > 0 else np.array([1.0, 0.0])
    v1 = np.array([-v2[1], v2[0]])
    c1, c2 = v1 @ c, v2 @ c
    cn = np.hypot(c1, c2)
    out = []
    scale = max(abs(l2), 1e-300)
    if disc and l1 > 1e-13 * scale:
        x = -(c1 / l1) * v1 - (c2 / l2) * v2
        if x @ x <= 1.0:
            return [x]
    if cn == 0.0:
        # flat model: only the hard case remains (direction of smallest curvature)
        return [v1] if not disc else []
    if abs(c1) <= 1e-9 * cn:
        # hard case: c is (numerically) orthogonal to the low-curvature eigenvector -> two minimisers
        gap = l2 - l1
        if gap > 0:
            y2 = -c2 / gap
            if abs(y2) < 1.0:
                sq = np.sqrt(1.0 - y2 * y2)
                return [y2 * v2 + sq * v1, y2 * v2 - sq * v1]
        c1 = 0.0
    lo = max(cn - l2, abs(c1) - l1)
    if disc:
        lo = max(lo, 0.0)
    tau = lo
    for _ in range(20):
        s1, s2 = l1 + tau, l2 + tau
        if s1 <= 0 or s2 <= 0:
            tau = max(-l1, -l2) + 1e-300
            s1, s2 = l1 + tau, l2 + tau
        a1 = c1 / s1 if c1 != 0 else 0.0
        a2 = c2 / s2 if c2 != 0 else 0.0
        phi = a1 * a1 + a2 * a2
        if phi <= 0:
            break
        dphi = -2.0 * (a1 * a1 / s1 + a2 * a2 / s2)
        sq = np.sqrt(phi)
        g = 1.0 / sq - 1.0
        dg = -0.5 * dphi / (phi * sq)
        step = g / dg
        tau_new = tau - step
        if abs(step) <= 4e-16 * max(1.0, abs(tau)):
            tau = tau_new
            break
        tau = tau_new
    s1, s2 = l1 + tau, l2 + tau
    x = -(c1 / s1 if c1 != 0 else 0.0) * v1 - (c2 / s2 if c2 != 0 else 0.0) * v2
    nx = np.hypot(x[0], x[1])
    if nx > 0:
        x = x / nx
    out.append(x)
    return out


def circle_interior(gstar, chi, ut, l0, kappa0, xi, ro2, delta):
    """Circle obstacle, candidate with 0 < ||a|| < 1 (lam_3 = -||a|| tight): minimise the smooth convex
    Phi(at) = model(t = at'ut + l0*||at|| + kappa0, e = at + xi) by damped Newton; l0 = -radius.
    Returns at or None (no interior stationary point: the ||a|| in {0, 1} candidates cover it)."""
    def G(t, e):
        gam, m, H = gstar(t, e)
        return np.array([chi * m - delta, ro2 * H[0], ro2 * H[1]]), 0.5 * chi * m * m - delta * m + 0.5 * ro2 * (H @ H)
    G0, _ = G(0.0, np.zeros(2))
    Hm = np.column_stack([G(1.0, np.zeros(2))[0] - G0, G(0.0, np.array([1.0, 0.0]))[0] - G0, G(0.0, np.array([0.0, 1.0]))[0] - G0])
    Hm = 0.5 * (Hm + Hm.T)
    nu = np.hypot(ut[0], ut[1])
    # start: steepest-descent ray out of the kink at at = 0, exact minimiser along it (Phi is quadratic on a ray)
    g0 = G0_ = G(kappa0, xi)[0]
    gk = g0[0] * ut + g0[1:]
    ck = g0[0] * l0
    ng = np.hypot(gk[0], gk[1])
    if not ng > ck * (1.0 + 1e-12):
        return None                    # at = 0 is the minimiser of this (convex) model: candidate L0 covers it
    v = -gk / ng
    w = np.array([v @ ut + l0, v[0], v[1]])
    curv = w @ Hm @ w
    s0 = (ng - ck) / curv if curv > (ng - ck) / 0.9 else 0.9
    x = s0 * v
    def fval(x):
        s_ = np.hypot(x[0], x[1])
        return G(x @ ut + l0 * s_ + kappa0, x + xi)[1]
    f = fval(x)
    nclip = 0
    gscale = lambda g3: 1.0 + abs(g3[0]) * nu + np.hypot(g3[1], g3[2])
    for _ in range(30):
        s_ = np.hypot(x[0], x[1])
        ah = x / s_
        g3, _ = G(x @ ut + l0 * s_ + kappa0, x + xi)
        Jt = ut + l0 * ah
        grad = g3[0] * Jt + g3[1:]
        Jm = np.vstack([Jt, np.eye(2)])
        Hs = Jm.T @ Hm @ Jm + g3[0] * l0 / s_ * (np.eye(2) - np.outer(ah, ah))
        # the hinge-active model is not convex in at where m > 0: shift an indefinite Hessian (modified Newton)
        tr_, df_ = 0.5 * (Hs[0, 0] + Hs[1, 1]), 0.5 * (Hs[0, 0] - Hs[1, 1])
        rad_ = np.hypot(df_, Hs[0, 1])
        lmin, lmax = tr_ - rad_, tr_ + rad_
        shift = 0.0
        if lmin < 1e-8 * max(abs(lmax), 1e-300):
            shift = 1e-8 * max(abs(lmax), 1e-300) - lmin
        Hs = Hs + shift * np.eye(2)
        det = Hs[0, 0] * Hs[1, 1] - Hs[0, 1] * Hs[1, 0]
        if not det > 0:
            return None
        d = -np.array([Hs[1, 1] * grad[0] - Hs[0, 1] * grad[1], -Hs[1, 0] * grad[0] + Hs[0, 0] * grad[1]]) / det
        if np.hypot(grad[0], grad[1]) <= 1e-13 * gscale(g3) or np.hypot(d[0], d[1]) <= 1e-15 * max(1.0, s_):
            break
        al = 1.0
        ok = False
        clipped = False
        for _bt in range(30):
            xn = x + al * d
            sn = np.hypot(xn[0], xn[1])
            if 1e-12 < sn < 1.0:
                fn = fval(xn)
                if fn <= f + 1e-4 * al * (grad @ d) + 1e-13 * abs(f):
                    ok = True
                    break
            elif sn >= 1.0:
                clipped = True
            al *= 0.5
        if clipped:
            nclip += 1
            if nclip >= 3:
                return None            # heading for ||a|| = 1: the boundary candidate LC covers it
        if not ok:
            break                      # no further decrease possible: judged by the final gradient test
        x, f = xn, fn
    s_ = np.hypot(x[0], x[1])
    ah = x / s_
    g3, _ = G(x @ ut + l0 * s_ + kappa0, x + xi)
    grad = g3[0] * (ut + l0 * ah) + g3[1:]
    if np.hypot(grad[0], grad[1]) > 1e-9 * gscale(g3):
        return None
    return x



CENTRE_H2 = 1e-8      # slack regime: the max-clearance optimum has m > 0 and |H|^2 below this (H = O(delta))


def polygon_vertices(A, b):
    """vertices of {x : A x <= b} with the two rows meeting there: pairs (i1 < i2) whose lines intersect in a point
    of the polygon (same test as the candidate pruning of the kernel)"""
    out = []
    E = A.shape[0]
    for i1 in range(E):
        for i2 in range(i1 + 1, E):
            det = A[i1, 0] * A[i2, 1] - A[i1, 1] * A[i2, 0]
            if det == 0 or not det * det > 1e-24 * (A[i1] @ A[i1]) * (A[i2] @ A[i2]):
                continue
            wx = b[i1] * A[i2, 1] - A[i1, 1] * b[i2]
            wy = A[i1, 0] * b[i2] - b[i1] * A[i2, 0]
            sg, ad = (1.0 if det > 0 else -1.0), abs(det)
            ok = True
            for k in range(E):
                viol = sg * (A[k, 0] * wx + A[k, 1] * wy - b[k] * det)
                if viol > 1e-9 * (ad + abs(b[k]) * ad + abs(A[k, 0] * wx) + abs(A[k, 1] * wy)):
                    ok = False
            if ok:
                out.append((wx / det, wy / det, i1, i2))
    return out


def central_normal(A, b, cone_norm2, p, phi, G, h, xi, kappa0, a_star):
    """Tie-break T1 in the slack regime (every (lam, mu) with H = 0, m >= 0 is optimal): instead of the max-clearance
    normal a*, the UNIT normal in the middle of the arc {theta : m(a(theta)) >= 0} of all separating directions around
    a*, with the duals that support it (H = 0 exactly).  For unit a the clearance is
        m(a) = min_{k,j} [ a'(p - v_k + R r_j) + xi'r_j ] + kappa0   (v_k obstacle vertices, r_j robot vertices; a circle
    obstacle contributes its centre and -radius), so every vertex pair admits an arc centred at the direction of its
    w_kj = p - v_k + R r_j with half-width acos(-c_j / |w_kj|); the feasible arc is their intersection.
    Returns (lam, mu, m) or None (arc undefined / degenerate -> keep the max-clearance solution)."""
    c, s = np.cos(phi), np.sin(phi)
    Rm = np.array([[c, -s], [s, c]])
    E, Rn = A.shape[0], G.shape[0]
    rv = polygon_vertices(G, h)
    if len(rv) < 3:
        return None
    if cone_norm2:
        ov = [(b[0], b[1], -1, -1)]
        off = b[2]                       # = -radius
    else:
        ov = polygon_vertices(A, b)
        off = 0.0
        if len(ov) < 3:
            return None
    th0 = np.arctan2(a_star[1], a_star[0])
    lo = hi = np.pi
    for (vx, vy, _, _) in ov:
        for (rx, ry, _, _) in rv:
            w = p - np.array([vx, vy]) + Rm @ np.array([rx, ry])
            cj = xi[0] * rx + xi[1] * ry + kappa0 + off
            nw = np.hypot(w[0], w[1])
            if not nw > 0:
                if cj < 0:
                    return None
                continue
            q = -cj / nw
            if q >= 1.0:
                return None              # this pair admits no direction: not the slack regime
            if q <= -1.0:
                continue                 # every direction is fine for this pair
            beta = np.arccos(q)
            d = th0 - np.arctan2(w[1], w[0])
            d = (d + np.pi) % (2 * np.pi) - np.pi
            if abs(d) > beta:
                return None              # a* itself does not separate for unit length: keep it
            hi = min(hi, beta - d)
            lo = min(lo, beta + d)
    if hi >= np.pi and lo >= np.pi:
        return None
    thc = th0 + 0.5 * (hi - lo)
    a = np.array([np.cos(thc), np.sin(thc)])
    lam = np.zeros(E)
    if cone_norm2:
        lam[0], lam[1], lam[2] = a[0], a[1], -1.0
    else:
        k = int(np.argmax([a[0] * v[0] + a[1] * v[1] for v in ov]))
        i1, i2 = ov[k][2], ov[k][3]
        lam[[i1, i2]] = np.linalg.solve(A[[i1, i2]].T, a)
    g = -(Rm.T @ a) - xi
    j = int(np.argmax([g[0] * r[0] + g[1] * r[1] for r in rv]))
    j1, j2 = rv[j][2], rv[j][3]
    mu = np.zeros(Rn)
    mu[[j1, j2]] = np.linalg.solve(G[[j1, j2]].T, g)
    if (not cone_norm2 and lam.min() < -1e-9) or mu.min() < -1e-9:
        return None
    if not cone_norm2:
        lam = np.maximum(lam, 0.0)
    mu = np.maximum(mu, 0.0)
    m = lam @ (A @ p - b) - mu @ h + kappa0
    if m < 0:
        return None
    return lam, mu, m


def solve_lammuz(A, b, cone_norm2, p, phi, G, h, xi, zeta, dbar, ro2,
                 accelerated=True, delta=DELTA, return_all=False, centre=True):
    """One (obstacle, stage) sub-problem.  A:(E,2) b:(E,) p:(2,) nominal position
    (column t+1), phi nominal heading (column t, quirk Q1), G:(R,2) h:(R,),
    xi:(2,), zeta, dbar scalars.  Returns lam(E), mu(R), z, info dict."""
    A = np.asarray(A, float); b = np.asarray(b, float).ravel()
    G = np.asarray(G, float); h = np.asarray(h, float).ravel()
    E, Rn = A.shape[0], G.shape[0]
    c, s = np.cos(phi), np.sin(phi)
    Rm = np.array([[c, -s], [s, c]])
    q = A @ p - b                      # obsA_trans - b  (rda_solver.py:562,902)
    M = A @ Rm                         # obsA_rot        (rda_solver.py:561)
    kappa0 = zeta - dbar
    xi = np.asarray(xi, float).ravel()

    def true_cost(lam, mu):
        m = lam @ q - mu @ h + kappa0
        H = M.T @ lam + G.T @ mu + xi
        return 0.5 * min(m, 0.0) ** 2 - delta * m + 0.5 * ro2 * (H @ H), m, H

    # ---- mu candidates -------------------------------------------------------
    mu_cands = [()]
    mu_cands += [(j,) for j in range(Rn) if G[j] @ G[j] > 0]
    for j1, j2 in itertools.combinations(range(Rn), 2):
        det = G[j1, 0] * G[j2, 1] - G[j1, 1] * G[j2, 0]
        if abs(det) > 1e-12 * np.linalg.norm(G[j1]) * np.linalg.norm(G[j2]) and det != 0:
            mu_cands.append((j1, j2))
    # ---- lam candidates ------------------------------------------------------
    if cone_norm2:
        lam_cands = [("L0",), ("LC",), ("LI",)]
    else:
        lam_cands = [("L0",)]
        lam_cands += [("L1", i) for i in range(E) if A[i] @ A[i] > 0]
        for i1, i2 in itertools.combinations(range(E), 2):
            det = A[i1, 0] * A[i2, 1] - A[i1, 1] * A[i2, 0]
            if abs(det) > 1e-12 * np.linalg.norm(A[i1]) * np.linalg.norm(A[i2]) and det != 0:
                lam_cands.append(("L2", i1, i2))

    def gamma_star(t, e, mc, chi):
        """minimise over the mu-support coefficients; returns gamma, m, H"""
        if len(mc) == 0:
            return np.zeros(0), t, e
        if len(mc) == 1:
            g = G[mc[0]]; eta = h[mc[0]]
            den = chi * eta * eta + ro2 * (g @ g)
            gam = (chi * eta * t - delta * eta - ro2 * (g @ e)) / den
            return np.array([gam]), t - eta * gam, e + gam * g
        j1, j2 = mc
        P = np.array([[G[j1, 0], G[j2, 0]], [G[j1, 1], G[j2, 1]]])   # columns = G rows
        r = np.linalg.solve(P.T, np.array([h[j1], h[j2]]))           # robot vertex
        rr = r @ r
        beta = (delta - chi * (t + r @ e)) / (chi * rr + ro2)
        H = -beta * r
        gam = np.linalg.solve(P, H - e)
        return gam, t + r @ e + beta * rr, H

    best = None
    allc = []
    nm = len(mu_cands)
    # rule T3.  Pass 1: the hinge-inactive candidates are ranked by the cost of the hinge-inactive MODEL
    # (-delta*m + ro2/2|H|^2, a lower bound of the true cost that is exact for m >= 0); if its minimiser c0 has
    # m >= 0 it is the global optimum.  Otherwise pass 2: the hinge-active candidates (and c0) compete on the
    # TRUE cost.  Exact ties go to the lowest candidate id.
    for chi in (0.0, 1.0):
      if chi == 1.0:
        if best[4] >= 0:
            break
        c0 = best
        tc = true_cost(c0[2], c0[3])[0]
        best = (tc,) + c0[1:]
      for il, lc in enumerate(lam_cands):
        for im, mc in enumerate(mu_cands):
            if True:
                idx = 2 * (il * nm + im) + int(chi)
                sols = []       # list of (lam_support_values as full vector)
                if lc[0] == "L0":
                    gam, m, H = gamma_star(kappa0, xi, mc, chi)
                    sols.append((np.zeros(E), gam))
                elif lc[0] == "L1":
                    i = lc[1]
                    amax = 1.0 / np.sqrt(A[i] @ A[i])
                    def dphi(al):
                        gam, m, H = gamma_star(al * q[i] + kappa0, al * M[i] + xi, mc, chi)
                        return (chi * m - delta) * q[i] + ro2 * (M[i] @ H), gam
                    d0, _ = dphi(0.0)
                    d1, _ = dphi(amax)
                    if d1 <= 0:
                        al = amax
                    elif d0 >= 0:
                        al = 0.0
                    else:
                        al = amax * d0 / (d0 - d1)
                    _, gam = dphi(al)
                    lam = np.zeros(E); lam[i] = al
                    sols.append((lam, gam))
                else:
                    if lc[0] == "L2":
                        i1, i2 = lc[1], lc[2]
                        AS = A[[i1, i2]]
                        v = np.linalg.solve(AS, b[[i1, i2]])
                        ut = Rm.T @ (p - v)
                        l0 = 0.0
                    else:   # LC circle
                        ut = Rm.T @ (p - b[0:2])
                        l0 = b[2]          # = -radius
                    def grad(at):
                        gam, m, H = gamma_star(at @ ut + l0 + kappa0, at + xi, mc, chi)
                        return (chi * m - delta) * ut + ro2 * H
                    g0 = grad(np.zeros(2))
                    g1 = grad(np.array([1.0, 0.0])) - g0
                    g2 = grad(np.array([0.0, 1.0])) - g0
                    Q = np.array([[g1[0], 0.5 * (g1[1] + g2[0])], [0.5 * (g1[1] + g2[0]), g2[1]]])
                    for at in (trs2(Q, g0, disc=(lc[0] == "L2")) if lc[0] != "LI" else []):
                        gam, m, H = gamma_star(at @ ut + l0 + kappa0, at + xi, mc, chi)
                        a = Rm @ at
                        lam = np.zeros(E)
                        if lc[0] == "L2":
                            ls = np.linalg.solve(AS.T, a)
                            lam[i1], lam[i2] = ls
                        else:
                            lam[0:2] = a
                            lam[2] = -np.hypot(a[0], a[1])
                        sols.append((lam, gam))
                    if lc[0] == "LI" and len(mc) < 2:
                        at = circle_interior(lambda t_, e_: gamma_star(t_, e_, mc, chi), chi, ut, l0, kappa0, xi, ro2, delta)
                        if at is not None:
                            s_ = np.hypot(at[0], at[1])
                            gam, m, H = gamma_star(at @ ut + l0 * s_ + kappa0, at + xi, mc, chi)
                            a = Rm @ at
                            lam = np.zeros(E)
                            lam[0:2] = a
                            lam[2] = -s_
                            sols.append((lam, gam))
                for lam, gam in sols:
                    mu = np.zeros(Rn)
                    for k, j in enumerate(mc):
                        mu[j] = gam[k]
                    # sign feasibility
                    if cone_norm2:
                        ok_l = True
                    else:
                        ok_l = np.all(lam >= -SIGN_TOL)
                    if not ok_l or not np.all(mu >= -SIGN_TOL):
                        continue
                    if not cone_norm2:
                        lam = np.maximum(lam, 0.0)
                    mu = np.maximum(mu, 0.0)
                    cost, m, H = true_cost(lam, mu)
                    if chi == 0.0:
                        cost = cost - 0.5 * min(m, 0.0) ** 2          # model cost in pass 1
                    allc.append((cost, idx, lc, mc, chi))
                    if best is None or cost < best[0] or (cost == best[0] and idx < best[1]):
                        best = (cost, idx, lam, mu, m, H, lc, mc, chi)
    cost, idx, lam, mu, m, H, lc, mc, chi = best
    central = False
    a_st = A.T @ lam
    if centre and m > 0 and H @ H < CENTRE_H2 and a_st @ a_st >= 1.0 - 1e-9:      # a* on the unit circle: it has a direction
        cn = central_normal(A, b, cone_norm2, np.asarray(p, float), phi, G, h, xi, kappa0, A.T @ lam)
        if cn is not None:
            lam, mu, m = cn
            H = M.T @ lam + G.T @ mu + xi
            central = True
    z = (0.5 if accelerated else 1.0) * max(m, 0.0)
    info = dict(cost=cost, m=m, H=H, cand=(lc, mc, chi), index=idx, central=central)
    if return_all:
        info["all"] = allc
    return lam, mu, z, info
