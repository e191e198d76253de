"""ORACLE - test infrastructure only.

Generic primal-dual interior-point method for

    minimise   1/2 x'Px + q'x
    subject to A x = b,   G x + s = h,   s in K = R+^l  x  Q^{m_1} x ... x Q^{m_k}

(second-order cones Q^m = {(u0, u1): u0 >= ||u1||}), Nesterov-Todd scaling, Mehrotra predictor-corrector -
the textbook algorithm behind ECOS / CVXOPT `coneqp` (Vandenberghe, "The CVXOPT linear and quadratic cone
program solvers", 2010), restated from that description with scipy.sparse.  It is the stand-in for the
`prob.solve(solver=cp.ECOS)` calls of the reference (rda_solver.py:693,768,800) when the *unmodified*
reference runs on top of oracle/refshim/cvxpy: ECOS itself (a third-party C library, version unpinned by
the reference's setup.py) is not installable here.  Like ECOS it follows the central path, so on a
non-unique optimum it returns (approximately) the analytic centre of the optimal face.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class ConeDims:
    def __init__(self, l, q):
        self.l = int(l)
        self.q = [int(m) for m in q]
        self.m = self.l + sum(self.q)
        self.deg = self.l + len(self.q)
        # cones grouped by dimension for batched arithmetic: dim -> (index array [k, dim])
        self.groups = {}
        off = self.l
        tmp = {}
        for m in self.q:
            tmp.setdefault(m, []).append(np.arange(off, off + m))
            off += m
        for m, lst in tmp.items():
            self.groups[m] = np.array(lst, dtype=np.int64)


def _jdet(u):            # u [k, m] -> u0^2 - |u1|^2
    return u[:, 0] ** 2 - np.einsum("ij,ij->i", u[:, 1:], u[:, 1:])


def min_eig(dims, u):
    """smallest 'eigenvalue' of u w.r.t. the cone (u is interior iff > 0)"""
    vals = [np.inf]
    if dims.l:
        vals.append(u[:dims.l].min())
    for idx in dims.groups.values():
        uu = u[idx]
        vals.append((uu[:, 0] - np.linalg.norm(uu[:, 1:], axis=1)).min())
    return min(vals)


def unit(dims):
    e = np.zeros(dims.m)
    e[:dims.l] = 1.0
    for idx in dims.groups.values():
        e[idx[:, 0]] = 1.0
    return e


def jprod(dims, u, v):
    """Jordan product u o v"""
    out = np.empty(dims.m)
    out[:dims.l] = u[:dims.l] * v[:dims.l]
    for idx in dims.groups.values():
        uu, vv = u[idx], v[idx]
        r = np.empty_like(uu)
        r[:, 0] = np.einsum("ij,ij->i", uu, vv)
        r[:, 1:] = uu[:, :1] * vv[:, 1:] + vv[:, :1] * uu[:, 1:]
        out[idx] = r
    return out


def jdiv(dims, lam, b):
    """solve lam o u = b"""
    out = np.empty(dims.m)
    out[:dims.l] = b[:dims.l] / lam[:dims.l]
    for idx in dims.groups.values():
        ll, bb = lam[idx], b[idx]
        det = _jdet(ll)
        l0, l1 = ll[:, 0], ll[:, 1:]
        b0, b1 = bb[:, 0], bb[:, 1:]
        l1b1 = np.einsum("ij,ij->i", l1, b1)
        r = np.empty_like(ll)
        r[:, 0] = (l0 * b0 - l1b1) / det
        r[:, 1:] = (-l1 * b0[:, None] + (det[:, None] * b1 + l1 * l1b1[:, None]) / l0[:, None]) / det[:, None]
        out[idx] = r
    return out


class Scaling:
    """Nesterov-Todd scaling W (symmetric): W z = W^{-1} s = lam"""

    def __init__(self, dims, s, z):
        self.dims = dims
        self.d = np.sqrt(s[:dims.l] / z[:dims.l])
        self.soc = {}
        for m, idx in dims.groups.items():
            ss, zz = s[idx], z[idx]
            ns, nz = np.sqrt(_jdet(ss)), np.sqrt(_jdet(zz))
            sb, zb = ss / ns[:, None], zz / nz[:, None]
            gamma = np.sqrt(0.5 * (1.0 + np.einsum("ij,ij->i", sb, zb)))
            w = sb.copy()
            w[:, 0] += zb[:, 0]
            w[:, 1:] -= zb[:, 1:]
            w /= (2.0 * gamma)[:, None]
            self.soc[m] = (np.sqrt(ns / nz), w)

    def _apply(self, u, inverse):
        dims = self.dims
        out = np.empty(dims.m)
        out[:dims.l] = u[:dims.l] / self.d if inverse else u[:dims.l] * self.d
        for m, idx in dims.groups.items():
            beta, w = self.soc[m]
            uu = u[idx]
            w0, w1 = w[:, 0], w[:, 1:]
            sgn = -1.0 if inverse else 1.0
            w1u1 = np.einsum("ij,ij->i", w1, uu[:, 1:])
            r = np.empty_like(uu)
            r[:, 0] = w0 * uu[:, 0] + sgn * w1u1
            r[:, 1:] = sgn * w1 * uu[:, :1] + uu[:, 1:] + w1 * (w1u1 / (1.0 + w0))[:, None]
            out[idx] = r / beta[:, None] if inverse else r * beta[:, None]
        return out

    def W(self, u):
        return self._apply(u, False)

    def Winv(self, u):
        return self._apply(u, True)

    def W2_matrix(self):
        """sparse block-diagonal W^2"""
        dims = self.dims
        rows, cols, vals = [np.arange(dims.l)], [np.arange(dims.l)], [self.d ** 2]
        for m, idx in dims.groups.items():
            beta, w = self.soc[m]
            k = idx.shape[0]
            # W = beta * [[w0, w1'], [w1, I + w1 w1'/(1+w0)]]
            Wm = np.zeros((k, m, m))
            Wm[:, 0, 0] = w[:, 0]
            Wm[:, 0, 1:] = w[:, 1:]
            Wm[:, 1:, 0] = w[:, 1:]
            Wm[:, 1:, 1:] = np.eye(m - 1)[None] + np.einsum("ki,kj->kij", w[:, 1:], w[:, 1:]) / (1.0 + w[:, 0])[:, None, None]
            Wm *= beta[:, None, None]
            W2 = np.einsum("kij,kjl->kil", Wm, Wm)
            rows.append(np.repeat(idx, m, axis=1).ravel())
            cols.append(np.tile(idx, (1, m)).ravel())
            vals.append(W2.ravel())
        return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(dims.m, dims.m))


def max_step(dims, u, du):
    """largest alpha with u + alpha du in the cone"""
    a = np.inf
    if dims.l:
        neg = du[:dims.l] < 0
        if neg.any():
            a = min(a, (-u[:dims.l][neg] / du[:dims.l][neg]).min())
    for idx in dims.groups.values():
        uu, dd = u[idx], du[idx]
        # (u0 + a d0)^2 - |u1 + a d1|^2 = c + 2 b a + qa a^2
        qa = _jdet(dd)
        b = uu[:, 0] * dd[:, 0] - np.einsum("ij,ij->i", uu[:, 1:], dd[:, 1:])
        c = _jdet(uu)
        for k in range(uu.shape[0]):
            cand = []
            if abs(qa[k]) < 1e-300:
                if b[k] < 0:
                    cand.append(-c[k] / (2 * b[k]))
            else:
                disc = b[k] * b[k] - qa[k] * c[k]
                if disc >= 0:
                    sq = np.sqrt(disc)
                    # stable roots of qa a^2 + 2 b a + c
                    t = -(b[k] + np.copysign(sq, b[k]))
                    r1 = t / qa[k]
                    r2 = c[k] / t if t != 0 else np.inf
                    cand += [r for r in (r1, r2) if r > 0]
            if dd[k, 0] < 0:
                cand.append(-uu[k, 0] / dd[k, 0])
            if cand:
                a = min(a, min(cand))
    return a


def solve(P, q, A, b, G, h, l, soc, tol=1e-10, max_iter=200, verbose=False):
    """returns dict(x, y, z, s, status, iters, pcost, gap, pres, dres)"""
    n = q.size
    dims = ConeDims(l, soc)
    P = sp.csc_matrix(P) if P is not None else sp.csc_matrix((n, n))
    A = sp.csc_matrix(A) if A is not None else sp.csc_matrix((0, n))
    G = sp.csc_matrix(G)
    p = A.shape[0]
    assert G.shape == (dims.m, n) and h.size == dims.m and b.size == p
    e = unit(dims)
    reg_p, reg_d = 1e-11, 1e-11

    def kkt_factor(W2):
        """LU of the quasi-definite system [[P, A', G'], [A, 0, 0], [G, 0, -W^2]] (static regularisation + refinement)"""
        K = sp.bmat([[P + reg_p * sp.identity(n, format="csc"), A.T, G.T],
                     [A, -reg_d * sp.identity(p, format="csc") if p else None, None],
                     [G, None, -W2 - reg_d * sp.identity(dims.m, format="csc")]], format="csc")
        Ktrue = sp.bmat([[P, A.T, G.T], [A, None, None], [G, None, -W2]], format="csc")
        lu = spla.splu(K)

        def solve_(bx, by, bz):
            rhs = np.concatenate([bx, by, bz])
            sol = lu.solve(rhs)
            for _ in range(4):      # iterative refinement against the un-regularised system
                r = rhs - Ktrue @ sol
                if np.abs(r).max(initial=0) < 1e-15 * (1 + np.abs(rhs).max(initial=0)):
                    break
                sol = sol + lu.solve(r)
            return sol[:n], sol[n:n + p], sol[n + p:]
        return solve_

    # initial point
    f0 = kkt_factor(sp.identity(dims.m, format="csc"))
    x, y, z = f0(-q, b, h)
    s = -z
    ts = -min_eig(dims, s)
    if ts >= -1e-8 * max(1.0, np.linalg.norm(s)):
        s = s + (1.0 + ts) * e
    tz = -min_eig(dims, z)
    if tz >= -1e-8 * max(1.0, np.linalg.norm(z)):
        z = z + (1.0 + tz) * e

    nq, nb, nh = 1 + np.abs(q).max(initial=0), 1 + np.abs(b).max(initial=0), 1 + np.abs(h).max(initial=0)
    status = "max_iter"
    best = None
    for it in range(max_iter):
        rx = P @ x + q + (A.T @ y if p else 0) + G.T @ z
        ry = A @ x - b if p else np.zeros(0)
        rz = G @ x + s - h
        gap = float(s @ z)
        pcost = 0.5 * float(x @ (P @ x)) + float(q @ x)
        dres = np.abs(rx).max(initial=0) / nq
        pres = max(np.abs(ry).max(initial=0) / nb, np.abs(rz).max(initial=0) / nh)
        relgap = gap / max(1.0, abs(pcost))
        if verbose:
            print(f"{it:3d} pcost {pcost: .10e} gap {gap:.2e} pres {pres:.2e} dres {dres:.2e}")
        meas = max(dres, pres, relgap)
        if best is None or meas < best[0]:
            best = (meas, x.copy(), y.copy(), z.copy(), s.copy(), it)
        if dres <= tol and pres <= tol and relgap <= tol:
            status = "optimal"
            break
        try:
            with np.errstate(all="ignore"):
                W = Scaling(dims, s, z)
                lam = W.W(z)
                W2 = W.W2_matrix()
            if not (np.all(np.isfinite(lam)) and np.all(np.isfinite(W2.data))):
                raise FloatingPointError
            fk = kkt_factor(W2)
        except Exception:                                # numerical breakdown at the boundary: keep the best iterate
            status = "breakdown"
            break

        def newton(bx, by, bz, bs):
            u = jdiv(dims, lam, bs)
            dx, dy, dz = fk(bx, by, bz - W.W(u))
            ds = W.W(u - W.W(dz))
            return dx, dy, dz, ds

        mu = gap / dims.deg
        with np.errstate(all="ignore"):
            dxa, dya, dza, dsa = newton(-rx, -ry, -rz, -jprod(dims, lam, lam))
            ok = bool(np.all(np.isfinite(dsa)) and np.all(np.isfinite(dza)))
            if ok:
                aa = min(1.0, max_step(dims, s, dsa), max_step(dims, z, dza))
                sigma = (1.0 - aa) ** 3
                bs = -jprod(dims, lam, lam) - jprod(dims, W.Winv(dsa), W.W(dza)) + sigma * mu * e
                dx, dy, dz, ds = newton(-(1 - sigma) * rx, -(1 - sigma) * ry, -(1 - sigma) * rz, bs)
                a = min(1.0, 0.99 * min(max_step(dims, s, ds), max_step(dims, z, dz)))
                ok = bool(np.isfinite(a) and a > 0 and np.all(np.isfinite(dx)) and np.all(np.isfinite(dz)) and np.all(np.isfinite(ds)))
        if not ok:
            status = "breakdown"
            break
        x, y, z, s = x + a * dx, y + a * dy, z + a * dz, s + a * ds
    if status != "optimal" and best is not None:
        meas, x, y, z, s, _ = best
        status = "optimal_inaccurate" if meas <= 1e-6 else status
        rx = P @ x + q + (A.T @ y if p else 0) + G.T @ z
        dres = np.abs(rx).max(initial=0) / nq
        rz = G @ x + s - h
        ry = A @ x - b if p else np.zeros(0)
        pres = max(np.abs(ry).max(initial=0) / nb, np.abs(rz).max(initial=0) / nh)
        gap = float(s @ z)
        pcost = 0.5 * float(x @ (P @ x)) + float(q @ x)
    return dict(x=x, y=y, z=z, s=s, status=status, iters=it + 1, pcost=pcost, gap=gap, pres=pres, dres=dres)
