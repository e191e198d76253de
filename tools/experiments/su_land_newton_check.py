"""Does k_su solve its Newton system?  A -DSU_LAND_DEBUG build of the library (rda_su_solve_opts with su_prof = 2 and SU_LAND_DUMP=<prefix> in the environment)
dumps the Newton system of the first landing round of a su-solve - stage Hessians Hb, transitions Ft, stage gradients gst, row terms xd, the eliminated
column m7 - and the step the kernel computed from it; this script condenses the same system densely in numpy and compares the steps.
    python tools/experiments/su_land_newton_check.py <prefix>_1.bin
Round 6: the landing's first version handed back steps 180 x too long on some problems although weights, right-hand sides and Hessians were right - the dense
solve of the DUMPED system gave the expected step, i.e. the factorisation was stale: the early "converged" verdict of wave 2 (rows of a landing round carry
no residual, so the measures of that pass always pass) had made waves 0 / 1 abandon the Riccati recursion.  With the verdict switched off in landing rounds
every landing of the test loops is accepted (tests/test_gpu_land.py)."""
import numpy as np, sys
def load(path):
    raw = np.fromfile(path)
    T = int(raw[0]); big = raw[1:]
    q = big[2800:]
    Hb = q[:64*T].reshape(T,8,8); Ft = q[64*T:112*T].reshape(T,8,6); gst = q[112*T:120*T].reshape(T,8); m7 = q[120*T:128*T].reshape(T,8); xd = q[128*T:133*T].reshape(T,5)
    dbg = big[400:2800]
    dy = [dbg[1800+300*p:1800+300*p+150][:8*T].reshape(T,8) for p in (0,1)]
    vv = [dbg[1800+300*p+150:1800+300*p+300][:8*T].reshape(T,8) for p in (0,1)]
    return T, Hb, Ft, gst, m7, xd, dy, vv
T, Hb, Ft, gst, m7, xd, dy, vv = load(sys.argv[1])
F = np.zeros((T,5,8))
for t in range(T):
    for q in range(8):
        for i in range(5): F[t,i,q] = Ft[t,q,i]
g = np.zeros((T,7))
for t in range(T):
    g7 = gst[t,7] + xd[t,2]; c7 = g7*m7[t,7]
    add = np.zeros(7); add[3] = -xd[t,3]; add[4] = -xd[t,4]; add[5] = xd[t,0]+xd[t,3]; add[6] = xd[t,1]+xd[t,4]
    g[t] = gst[t,:7] + add - m7[t,:7]*c7
H = Hb[:, :7, :7]
n = 2*T
# x_t = S_t u  (5 x n)
S = [np.zeros((5,n))]
for t in range(T):
    Sx = F[t][:, :5] @ S[t]
    Sx[:, 2*t:2*t+2] += F[t][:, 5:7]
    S.append(Sx)
K = np.zeros((n,n)); rhs = np.zeros(n)
for t in range(T):
    Y = np.zeros((7,n)); Y[:5] = S[t]; Y[5, 2*t] = 1; Y[6, 2*t+1] = 1
    K += Y.T @ H[t] @ Y; rhs += Y.T @ g[t]
print("cond K %.2e, min eig %.3e" % (np.linalg.cond(K), np.linalg.eigvalsh(K).min()))
du = np.linalg.solve(K, -rhs).reshape(T,2)
dk = np.array([[vv[0][t,3], vv[0][t,4]] for t in range(T)])
for t in range(T): print(f"t={t:2d} dense du {du[t,0]: .4e} {du[t,1]: .4e}   kernel pass0 du {dk[t,0]: .4e} {dk[t,1]: .4e}")
print("max |dense - kernel| %.3e   max |dense| %.3e" % (np.abs(du-dk).max(), np.abs(du).max()))
