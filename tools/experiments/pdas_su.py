"""Round-3 prototype (CPU, dense numpy, test infrastructure): a primal-dual active-set / semismooth-Newton iteration for the su-problem in
the condensed form of oracle/rda_oracle.c, run on su-problems recorded by tools/su_replay.py.  Result on the C4 recording (120 problems): from
the cold start it needs 15-40 iterations where the interior point needs 12.8, and 37 of 120 runs do not settle within 40 iterations (the rate
constraints couple neighbouring stages: the active-set map cycles); systems with a singular reduced Hessian (a distance variable whose bound
leaves the guess) are solved in the least-squares sense, so its `converged' points are not reliable either.  Not pursued; kept because
tools/experiments/scan_riccati.py uses its lin_model()."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from su_replay import load, _lib as lib, solve

def lin_model(c, st, ut):
    dt = c.dt; A = np.eye(3); B = np.zeros((3, 2)); Cv = np.zeros(3)
    if c.dynamics == 2:
        phi, v = ut[1], ut[0]
        B[0, 0] = np.cos(phi) * dt; B[0, 1] = -v * np.sin(phi) * dt; B[1, 0] = np.sin(phi) * dt; B[1, 1] = v * np.cos(phi) * dt
        Cv[0] = phi * v * np.sin(phi) * dt; Cv[1] = -phi * v * np.cos(phi) * dt
        return A, B, Cv
    phi, v = st[2], ut[0]
    A[0, 2] = -v * dt * np.sin(phi); A[1, 2] = v * dt * np.cos(phi)
    B[0, 0] = np.cos(phi) * dt; B[1, 0] = np.sin(phi) * dt
    Cv[0] = phi * v * np.sin(phi) * dt; Cv[1] = -phi * v * np.cos(phi) * dt
    if c.dynamics == 0:
        psi = ut[1]; cp = np.cos(psi)
        B[2, 0] = np.tan(psi) * dt / c.L; B[2, 1] = v * dt / (c.L * cp * cp); Cv[2] = -psi * v * dt / (c.L * cp * cp)
    else:
        B[2, 1] = dt
    return A, B, Cv

class Prob:
    def __init__(self, pr):
        c = self.c = pr["cfg"]; T = self.T = pr["T"]; N = self.N = pr["N"]; n = self.n = 3 * T
        s = pr["s"].reshape(3, T + 1); u = pr["u"].reshape(2, T)
        self.ref = pr["ref"].reshape(3, T + 1); self.vref = pr["ref_speed"]; self.nom_s = s; self.nom_u = u; self.d0 = pr["d"]
        self.a = pr["a"].reshape(N, T, 2); self.cc = pr["cc"].reshape(N, T); g = pr["g"].reshape(N, T, 2)
        # affine map  s_{t+1} = S0[t] + sum_k Gam[t][k] u_k   (x = [u_0(2) .. u_{T-1}(2) | d(T)])
        self.G = np.zeros((T, 3, n)); self.S0 = np.zeros((T, 3))
        sc = s[:, 0].copy(); Gc = np.zeros((3, n))
        self.Q1 = np.zeros(T); self.Q2 = np.zeros(T)
        for t in range(T):
            A, B, Cv = lin_model(c, s[:, t], u[:, t])
            Gc = A @ Gc; Gc[:, 2 * t:2 * t + 2] += B; sc = A @ sc + Cv
            self.G[t] = Gc; self.S0[t] = sc
            cs, sn = np.cos(s[2, t]), np.sin(s[2, t]); an = self.a[:, t]; gn = g[:, t]
            k0x = gn[:, 0] + cs * an[:, 0] + sn * an[:, 1]; k0y = gn[:, 1] - sn * an[:, 0] + cs * an[:, 1]
            k1x = -sn * an[:, 0] + cs * an[:, 1]; k1y = -cs * an[:, 0] - sn * an[:, 1]
            self.Q1[t] = 2 * np.sum(k0x * k1x + k0y * k1y); self.Q2[t] = np.sum(k1x ** 2 + k1y ** 2)
        rows = []; e = []
        def row(i1, c1, i2=None, c2=0.0):
            r = np.zeros(n); r[i1] = c1
            if i2 is not None: r[i2] = c2
            return r
        for t in range(T):
            for i in range(2):
                rows.append(row(2 * t + i, 1.0)); e.append(c.max_speed[i]); rows.append(row(2 * t + i, -1.0)); e.append(c.max_speed[i])
        for t in range(T - 1):
            for i in range(2):
                rows.append(row(2 * (t + 1) + i, 1.0, 2 * t + i, -1.0)); e.append(c.acce_bound[i])
                rows.append(row(2 * (t + 1) + i, -1.0, 2 * t + i, 1.0)); e.append(c.acce_bound[i])
        for t in range(T):
            rows.append(row(2 * T + t, 1.0)); e.append(c.max_sd); rows.append(row(2 * T + t, -1.0)); e.append(-c.min_sd)
        self.C = np.array(rows); self.e = np.array(e)
    def states(self, x):
        return self.S0 + np.einsum('tij,j->ti', self.G, x)          # [T][3] = s_{t+1}
    def eval(self, x, want_H=True):
        c, T, n = self.c, self.T, self.n
        st = self.states(x); d = x[2 * T:]
        wz = 0.0 if c.dynamics == 2 else 1.0; w3 = np.array([1, 1, wz])
        grad = np.zeros(n); H = np.zeros((n, n)) if want_H else None
        f = 0.0; nact = 0; act = []
        for t in range(T):
            df = st[t] - self.ref[:, t + 1]
            f += c.ws * np.sum(w3 * df * df); gs = 2 * c.ws * w3 * df; Hs = np.diag(2 * c.ws * w3)
            dl = st[t, 2] - self.nom_s[2, t]
            f += 0.5 * c.ro2 * (self.Q1[t] * dl + self.Q2[t] * dl * dl); gs[2] += 0.5 * c.ro2 * (self.Q1[t] + 2 * self.Q2[t] * dl); Hs[2, 2] += c.ro2 * self.Q2[t]
            Im = self.a[:, t, 0] * st[t, 0] + self.a[:, t, 1] * st[t, 1] - self.cc[:, t] - d[t]
            m = Im < 0 if c.accelerated else np.ones_like(Im, bool)
            act.append(m); nact += int(m.sum())
            am = self.a[m, t]; Imm = Im[m]
            f += 0.5 * c.ro1 * np.sum(Imm ** 2) - c.slack_gain * d[t]
            gs[:2] += c.ro1 * (Imm @ am); gd = -c.ro1 * Imm.sum() - c.slack_gain
            Gt = self.G[t]
            grad += Gt.T @ gs; grad[2 * T + t] += gd
            if want_H:
                Hs[:2, :2] += c.ro1 * am.T @ am; hsd = -c.ro1 * am.sum(0); hdd = c.ro1 * m.sum()
                H += Gt.T @ Hs @ Gt
                v = Gt[:2].T @ hsd; H[:, 2 * T + t] += v; H[2 * T + t, :] += v; H[2 * T + t, 2 * T + t] += hdd
        u0 = x[0:2 * T:2]; u1 = x[1:2 * T:2]
        f += c.wu * np.sum((u0 - self.vref) ** 2) + 0.5 * c.eps_u * np.sum(x[:2 * T] ** 2)
        grad[0:2 * T:2] += 2 * c.wu * (u0 - self.vref) + c.eps_u * u0; grad[1:2 * T:2] += c.eps_u * u1
        if want_H:
            idx = np.arange(0, 2 * T, 2); H[idx, idx] += 2 * c.wu + c.eps_u; H[idx + 1, idx + 1] += c.eps_u
        return f, grad, H, np.concatenate(act)

def start(P):
    c, T = P.c, P.T
    x = np.zeros(P.n)
    for i in range(2):
        lim = 0.99 * c.max_speed[i]; x[i:2 * T:2] = np.clip(P.nom_u[i], -lim, lim)
    lo = c.min_sd + 0.01 * (c.max_sd - c.min_sd); hi = c.max_sd - 0.01 * (c.max_sd - c.min_sd)
    x[2 * T:] = np.clip(P.d0, lo, hi)
    return x

def pdas(P, gamma=1e3, maxit=40, verbose=False, x=None, lam=None):
    x = start(P) if x is None else x.copy()
    if lam is None:            # the distance rows d_t <= max_sd start active (the reward -slack_gain d has no curvature of its own)
        lam = np.zeros(len(P.e)); lam[8 * P.T - 4::2] = P.c.slack_gain
    else: lam = lam.copy()
    A_prev = None; h_prev = None
    for it in range(maxit):
        f, g, H, hact = P.eval(x)
        A = (lam + gamma * (P.C @ x - P.e)) > 0
        if A_prev is not None and np.array_equal(A, A_prev) and np.array_equal(hact, h_prev):
            return x, lam, it, True
        CA = P.C[A]
        # equality-constrained Newton step on the current quadratic model:  [H CA'; CA 0] [dx; lamA] = [-g ; eA - CA x]
        nA = CA.shape[0]
        K = np.block([[H, CA.T], [CA, np.zeros((nA, nA))]])
        rhs = np.concatenate([-g, P.e[A] - CA @ x])
        try: sol = np.linalg.solve(K, rhs)
        except np.linalg.LinAlgError: sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        x = x + sol[:P.n]; lam = np.zeros(len(P.e)); lam[A] = sol[P.n:]
        if verbose: print(f"   it {it} nA {nA} nH {hact.sum()} |dx| {np.abs(sol[:P.n]).max():.2e} min lamA {lam[A].min() if nA else 0:.2e} maxviol {np.max(P.C @ x - P.e):.2e}")
        A_prev, h_prev = A, hact
    return x, lam, maxit, False

if __name__ == "__main__":
    name = sys.argv[1]; l = lib(); probs = load(name)
    its = []; bad = 0; errs = []; ipm = []
    sel = range(len(probs)) if len(sys.argv) < 3 else eval(sys.argv[2])
    for i in sel:
        pr = probs[i]; P = Prob(pr)
        st, nit, so, uo, do = solve(l, pr); ipm.append(nit)
        xo = np.concatenate([uo.reshape(2, -1).T.ravel(), do])
        x, lam, it, ok = pdas(P, verbose=len(sys.argv) > 3)
        err = np.abs(x - xo).max()
        its.append(it); bad += (not ok); errs.append(err)
        if not ok or err > 1e-5: print("problem", i, "admm", pr["it"], "pdas its", it, "ok", ok, "err vs ipm", err, "ipm its", nit)
    print(f"PDAS: mean {np.mean(its):.2f} max {np.max(its)} not converged {bad}; IPM mean {np.mean(ipm):.2f}; max err {np.max(errs):.2e} median err {np.median(errs):.2e}")
