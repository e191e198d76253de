"""numpy check of the two-segment (time-split) Riccati solve of k_su (round 4) in the kernel's conventions (csrc/su_device.h):
stage y = [x(5); v(3)], x+ = F y, M = H + F'PF, W = Mxv Minv, P = Mxx - W Mvx, backward p = g_x - W g_v + Acl' p+, kk = -Minv (g_v + Fv' p+),
forward dx+ = Acl dx + Fv kk, v = kk - W' dx.  Segment B = stages m..T-1 from P_T = 0; segment A = stages 0..m-1 from P = 0 with the LINEAR terminal
term pi = P_m x_m + p_m found from a 5x5 interface system."""
import numpy as np
rng = np.random.default_rng(0)

def stage(T):
    Hs, Fs, gs = [], [], []
    for t in range(T):
        A = np.eye(3); A[0, 2] = rng.normal(0, .3); A[1, 2] = rng.normal(0, .3)
        B = rng.normal(0, .3, (3, 2))
        F = np.zeros((5, 8)); F[:3, :3] = A; F[:3, 5:7] = B; F[3, 5] = 1; F[4, 6] = 1
        R = rng.normal(0, 1, (8, 8)); H = R @ R.T * 0.1
        H[5:, 5:] += np.diag(10 ** rng.uniform(-1, 6, 3))          # barrier weights
        Hs.append(H); Fs.append(F); gs.append(rng.normal(0, 1, 8))
    return Hs, Fs, gs

def recursion(Hs, Fs, lo, hi):
    """matrix recursion over stages hi-1 .. lo from P = 0; returns per-stage (W, Minv, Acl, Fv) and P_lo"""
    P = np.zeros((5, 5)); out = {}
    for t in range(hi - 1, lo - 1, -1):
        F = Fs[t]; M = Hs[t] + F.T @ P @ F
        Minv = np.linalg.inv(M[5:, 5:]); W = M[:5, 5:] @ Minv
        P = M[:5, :5] - W @ M[5:, :5]; P = 0.5 * (P + P.T)
        Fx, Fv = F[:, :5], F[:, 5:]
        out[t] = (W, Minv, Fx - Fv @ W.T, Fv)
    return out, P

def bwd(G, gs, lo, hi, p_end):
    p = p_end.copy(); kk = {}
    for t in range(hi - 1, lo - 1, -1):
        W, Minv, Acl, Fv = G[t]; g = gs[t]
        kk[t] = -Minv @ (g[5:] + Fv.T @ p)
        p = g[:5] - W @ g[5:] + Acl.T @ p
    return p, kk

def fwd(G, kk, lo, hi, x0):
    x = x0.copy(); dx, v = {}, {}
    for t in range(lo, hi):
        W, Minv, Acl, Fv = G[t]
        dx[t] = x; v[t] = kk[t] - W.T @ x
        x = Acl @ x + Fv @ kk[t]
    return x, dx, v

for T, m in ((20, 10), (30, 15), (30, 10), (10, 5)):
    Hs, Fs, gs = stage(T)
    # reference: one recursion over the whole horizon
    G, _ = recursion(Hs, Fs, 0, T)
    _, kk = bwd(G, gs, 0, T, np.zeros(5)); _, dx_ref, v_ref = fwd(G, kk, 0, T, np.zeros(5))
    # two segments
    GB, Pm = recursion(Hs, Fs, m, T)
    GA, _ = recursion(Hs, Fs, 0, m)
    pm, kkB = bwd(GB, gs, m, T, np.zeros(5))
    _, kkA0 = bwd(GA, gs, 0, m, np.zeros(5)); xm0, dxA0, vA0 = fwd(GA, kkA0, 0, m, np.zeros(5))
    zero = [np.zeros(8)] * T
    X = np.zeros((5, 5)); unit = []
    for i in range(5):
        e = np.zeros(5); e[i] = 1
        _, kki = bwd(GA, zero, 0, m, e); xi, dxi, vi = fwd(GA, kki, 0, m, np.zeros(5))
        X[:, i] = xi; unit.append((dxi, vi))
    S = np.eye(5) - X @ Pm
    xm = np.linalg.solve(S, xm0 + X @ pm); pi = Pm @ xm + pm
    _, dxB, vB = fwd(GB, kkB, m, T, xm)
    err = 0.0
    for t in range(T):
        if t < m:
            d = dxA0[t] + sum(pi[i] * unit[i][0][t] for i in range(5)); vv = vA0[t] + sum(pi[i] * unit[i][1][t] for i in range(5))
        else:
            d, vv = dxB[t], vB[t]
        err = max(err, np.abs(d - dx_ref[t]).max() / (1 + np.abs(dx_ref[t]).max()), np.abs(vv - v_ref[t]).max() / (1 + np.abs(v_ref[t]).max()))
    print(f"T={T} m={m}: max rel err {err:.2e}; cond(I - X P_m) {np.linalg.cond(S):.1e}; eig(X) max {np.linalg.eigvalsh(0.5*(X+X.T)).max():.2e} (X = -Phi <= 0)")

# ---- design 2: no unit FORWARD sweeps.  X and x0 as reductions over the unit / main BACKWARD sweeps; combined kk; one forward sweep of A
def bwd_full(G, gs, lo, hi, p_end):
    """like bwd but also returns p_t for every stage (p[t] = value AFTER processing stage t) """
    p = p_end.copy(); kk = {}; ps = {hi: p_end.copy()}
    for t in range(hi - 1, lo - 1, -1):
        W, Minv, Acl, Fv = G[t]; g = gs[t]
        kk[t] = -Minv @ (g[5:] + Fv.T @ p)
        p = g[:5] - W @ g[5:] + Acl.T @ p; ps[t] = p.copy()
    return p, kk, ps

print("design 2:")
for T, m in ((20, 10), (30, 15), (25, 12), (10, 5)):
    Hs, Fs, gs = stage(T)
    G, _ = recursion(Hs, Fs, 0, T)
    _, kk = bwd(G, gs, 0, T, np.zeros(5)); _, dx_ref, v_ref = fwd(G, kk, 0, T, np.zeros(5))
    GB, Pm = recursion(Hs, Fs, m, T); GA, _ = recursion(Hs, Fs, 0, m)
    pm, kkB = bwd(GB, gs, m, T, np.zeros(5))
    _, kkA0, _ = bwd_full(GA, gs, 0, m, np.zeros(5))
    zero = [np.zeros(8)] * T
    UK, UB = [], []
    for j in range(5):
        e = np.zeros(5); e[j] = 1
        _, kkj, psj = bwd_full(GA, zero, 0, m, e)
        UK.append(kkj); UB.append({t: GA[t][3].T @ psj[t + 1] for t in range(m)})       # b = Fv' p_{t+1}
    X = np.array([[sum(UB[j][t] @ UK[i][t] for t in range(m)) for i in range(5)] for j in range(5)])
    x0 = np.array([sum(UB[j][t] @ kkA0[t] for t in range(m)) for j in range(5)])
    S = np.eye(5) - X @ Pm
    xm = np.linalg.solve(S, x0 + X @ pm); pi = Pm @ xm + pm
    kkA = {t: kkA0[t] + sum(pi[i] * UK[i][t] for i in range(5)) for t in range(m)}
    xmA, dxA, vA = fwd(GA, kkA, 0, m, np.zeros(5))
    _, dxB, vB = fwd(GB, kkB, m, T, xm)
    err = max(max(np.abs((dxA if t < m else dxB)[t] - dx_ref[t]).max(), np.abs((vA if t < m else vB)[t] - v_ref[t]).max()) for t in range(T))
    print(f"T={T} m={m}: max abs err {err:.2e}; |x_m(A) - x_m| {np.abs(xmA - xm).max():.1e}; X symmetric {np.abs(X - X.T).max():.1e}")
