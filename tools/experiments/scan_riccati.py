"""numerical experiment: backward Riccati recursion (what k_su does) vs the associative-scan form of the same recursion (Sarkka & Garcia-Fernandez 2023)
on the stage matrices of recorded C4 su-problems along the central path: error of P_t against an 80-bit sequential reference"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from su_replay import load, _lib, solve
from pdas_su import lin_model

def iterates(l, pr):
    """(x, w, lam) of every interior-point iteration of the cold solve of one recorded problem (oracle trace mode 2)"""
    import os
    from su_replay import SCRATCH
    path = os.path.join(SCRATCH, "_iter.bin")
    l.orc_set_su_dump(path.encode()); l.orc_set_su_trace(2); solve(l, pr); l.orc_set_su_trace(0); l.orc_set_su_dump(b"")
    raw = np.fromfile(path); off = 0; out = []
    while off < len(raw):
        it, mc, n = int(raw[off]), int(raw[off + 1]), int(raw[off + 2]); off += 3
        x = raw[off:off + n]; off += n; w = raw[off:off + mc]; off += mc; lm = raw[off:off + mc]; off += mc
        out.append((x.copy(), w.copy(), lm.copy()))
    return out

def stage_data_it(pr, x, w, lm):
    """stage matrices of the Newton system at an interior-point iterate; oracle row order: 4T speed rows (t, i, +-), 4(T-1) rate rows, 2T distance rows"""
    T = pr["T"]; uo = np.concatenate([x[0:2 * T:2], x[1:2 * T:2]]); do = x[2 * T:]
    bwt = lm / w
    return stage_data(pr, uo, do, None, bwt=bwt)

def stage_data(pr, uo, do, mu, dtype=np.float64, bwt=None):
    c = pr["cfg"]; T = pr["T"]; N = pr["N"]
    s = pr["s"].reshape(3, T + 1); unom = pr["u"].reshape(2, T)
    a = pr["a"].reshape(N, T, 2); cc = pr["cc"].reshape(N, T); g = pr["g"].reshape(N, T, 2)
    u = uo.reshape(2, T)
    # roll the solution out
    st = np.zeros((3, T + 1)); st[:, 0] = s[:, 0]
    AB = []
    for t in range(T):
        A, B, Cv = lin_model(c, s[:, t], unom[:, t]); AB.append((A, B)); st[:, t + 1] = A @ st[:, t] + B @ u[:, t] + Cv
    Hs, Fs = [], []
    wz = 0.0 if c.dynamics == 2 else 1.0
    for t in range(T):
        A, B = AB[t]
        F = np.zeros((5, 8)); F[:3, :3] = A; F[:3, 5:7] = B; F[3, 5] = 1; F[4, 6] = 1
        Im = a[:, t, 0] * st[0, t + 1] + a[:, t, 1] * st[1, t + 1] - cc[:, t] - do[t]
        m = Im < 0; am = a[m, t]
        cs, sn = np.cos(s[2, t]), np.sin(s[2, t]); an = a[:, t]
        Q2 = np.sum((-sn * an[:, 0] + cs * an[:, 1]) ** 2 + (-cs * an[:, 0] - sn * an[:, 1]) ** 2)
        Hw = np.zeros((4, 4)); Hw[0, 0] = Hw[1, 1] = 2 * c.ws; Hw[2, 2] = 2 * c.ws * wz + c.ro2 * Q2
        Hw[:2, :2] += c.ro1 * am.T @ am; Hw[:2, 3] = Hw[3, :2] = -c.ro1 * am.sum(0); Hw[3, 3] = c.ro1 * m.sum()
        J = np.zeros((4, 8)); J[:3] = F[:3]; J[3, 7] = 1
        H = J.T @ Hw @ J
        H[5, 5] += 2 * c.wu + c.eps_u; H[6, 6] += c.eps_u
        # barrier weights lam / w with lam = mu / w on the central path
        for i in range(2):
            H[5 + i, 5 + i] += bwt[4 * t + 2 * i] + bwt[4 * t + 2 * i + 1]
            if t >= 1:
                br = bwt[4 * T + 4 * (t - 1) + 2 * i] + bwt[4 * T + 4 * (t - 1) + 2 * i + 1]
                H[5 + i, 5 + i] += br; H[3 + i, 3 + i] += br; H[5 + i, 3 + i] -= br; H[3 + i, 5 + i] -= br
        H[7, 7] += bwt[8 * T - 4 + 2 * t] + bwt[8 * T - 4 + 2 * t + 1]
        Hs.append(H.astype(dtype)); Fs.append(F.astype(dtype))
    return Hs, Fs

def riccati(Hs, Fs, dtype):
    T = len(Hs); P = np.zeros((5, 5), dtype); out = [None] * T
    for t in range(T - 1, -1, -1):
        M = Hs[t].astype(dtype) + Fs[t].astype(dtype).T @ P @ Fs[t].astype(dtype)
        Mxx, Mxv, Mvv = M[:5, :5], M[:5, 5:], M[5:, 5:]
        W = np.linalg.solve(Mvv.astype(np.float64), Mxv.T.astype(np.float64)).T if dtype == np.float64 else solve_ld(Mvv, Mxv.T).T
        P = Mxx - W @ Mxv.T; P = 0.5 * (P + P.T); out[t] = P
    return out

def solve_ld(A, B):          # Gaussian elimination with partial pivoting in the array's own precision (numpy.linalg is float64 only)
    A = A.copy(); B = B.copy(); n = A.shape[0]
    for k in range(n):
        p = k + np.argmax(np.abs(A[k:, k])); A[[k, p]] = A[[p, k]]; B[[k, p]] = B[[p, k]]
        for i in range(k + 1, n):
            f = A[i, k] / A[k, k]; A[i, k:] -= f * A[k, k:]; B[i] -= f * B[k]
    X = np.zeros_like(B)
    for k in range(n - 1, -1, -1): X[k] = (B[k] - A[k, k + 1:] @ X[k + 1:]) / A[k, k]
    return X

def scan(Hs, Fs):
    """suffix combination of the conditional value functions, float64; J of element (t -> T) is P_t"""
    T = len(Hs); el = []
    for t in range(T):
        H, F = Hs[t], Fs[t]; A, B = F[:, :5], F[:, 5:]
        Hxx, Hxv, Hvv = H[:5, :5], H[:5, 5:], H[5:, 5:]
        Ri = np.linalg.inv(Hvv)
        el.append((A - B @ Ri @ Hxv.T, B @ Ri @ B.T, Hxx - Hxv @ Ri @ Hxv.T))       # (A~, C, J)
    def comb(ei, ej):      # i earlier, j later
        Ai, Ci, Ji = ei; Aj, Cj, Jj = ej; I = np.eye(5)
        X = np.linalg.solve(I + Ci @ Jj, np.concatenate([Ai, Ci], axis=1)); XA, XC = X[:, :5], X[:, 5:]
        Y = np.linalg.solve(I + Jj @ Ci, Jj @ Ai)
        return (Aj @ XA, Aj @ XC @ Aj.T + Cj, Ai.T @ Y + Ji), np.linalg.cond(I + Ci @ Jj)
    # Hillis-Steele suffix scan: log2(T) levels, every element combined with the one `off` stages later
    cur = list(el); off = 1; worst = 0.0
    while off < T:
        nxt = list(cur)
        for t in range(T - off):
            nxt[t], cd = comb(cur[t], cur[t + off]); worst = max(worst, cd)
        cur = nxt; off *= 2
    return [e[2] for e in cur], worst

if __name__ == "__main__":
    l = _lib(); probs = load(sys.argv[1])
    l.orc_set_su_dump.argtypes = [__import__("ctypes").c_char_p]
    LD = np.longdouble
    rel = lambda X, R: float(np.abs(X - R.astype(np.float64)).max() / np.abs(R.astype(np.float64)).max())
    print("problem it   mean lam*w   max lam/w   rel err of P_t (worst stage): sequential fp64 | scan fp64 | worst cond(I + C_i J_j)")
    for i in (40, 42, 61, 100):
        its = iterates(l, probs[i])
        for k, (x, w, lm) in enumerate(its):
            if k % 2 and k != len(its) - 1: continue
            Hs, Fs = stage_data_it(probs[i], x, w, lm)
            ref = riccati([h.astype(LD) for h in Hs], [f.astype(LD) for f in Fs], LD)
            seq = riccati(Hs, Fs, np.float64); sc_, cd = scan(Hs, Fs)
            T = probs[i]["T"]
            print(f"{i:4d} {k:3d}   {np.mean(lm * w):.1e}    {np.max(lm / w):.1e}    {max(rel(seq[t], ref[t]) for t in range(T)):.2e}   {max(rel(sc_[t], ref[t]) for t in range(T)):.2e}   {cd:.1e}")
