// K3 device code: the state/control ("su") problem of one ADMM iteration, solved by ONE
// workgroup as a primal-dual interior point method whose Newton systems are block-tridiagonal
// and solved by a Riccati recursion over the T stages (reference: construct_su_prob
// rda_solver.py:216-231, nav_cost_cons :313-328, update_su_cost_cons :330-387, Im_su/Hm_su
// :831-872, dynamics/bounds :911-947, C0/C1 cost :1011-1032).
//
//   stage vector   y_t = [ s_t(3) | up_t(2) = u_{t-1} | u_t(2) | d_t ]            (8)
//   dynamics       s_{t+1} = A_t s_t + B_t u_t + C_t ,  up_{t+1} = u_t
//   stage cost     q_t(s_{t+1}, d_t)  [tracking + rotation penalty + sum_n hinge^2]
//                  + wu (u_t[0]-v_ref)^2 + eps_u/2 |u_t|^2 - slack_gain d_t
//   inequalities   |u_t| <= u_max, d_min <= d_t <= d_max, |u_t - up_t| <= a_max dt (t >= 1)
//
// The only N-dependent work per interior-point iteration is a per-stage reduction of nine sums
// over the obstacles whose hinge is active - done by all waves of the workgroup from the
// [T][N] structure-of-arrays coefficients (coalesced over n); the serial Riccati sweeps run on
// wave 0 with the 8x8 stage matrix spread one entry per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace su {

constexpr int NT = 256;          // workgroup size
constexpr int NC = 10;           // inequality rows per stage

struct Cfg {
    int T, N, dynamics, accelerated;
    double dt, L, umax0, umax1, ab0, ab1, ws, wu, slack_gain, max_sd, min_sd, ro1, ro2, eps_u;
};

struct Args {
    Cfg c;
    const double *in_s, *in_u;       // linearisation point (3x(T+1), 2xT)
    const double *ref;               // 3x(T+1)
    const double *ref_speed;         // scalar on device
    const double *ax, *ay, *blam, *ee, *gx, *gy;   // [T][N] condensed obstacle terms
    const double *d_in;              // [T] initial guess for d
    double *out_s, *out_u, *out_d;   // results (may alias in_*)
    int *status;                     // 0 ok / 1 not converged / 2 factorisation failed
    int *ipm_iters;
};

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// linearised motion models, rda_solver.py:949-994
__device__ inline void lin_model(const Cfg &c, const double *st, const double *ut, double *A, double *B, double *C)
{
    double dt = c.dt;
    for (int i = 0; i < 9; ++i) A[i] = 0;
    for (int i = 0; i < 6; ++i) B[i] = 0;
    C[0] = C[1] = C[2] = 0;
    A[0] = A[4] = A[8] = 1;
    if (c.dynamics == 2) {
        double phi = ut[1], v = ut[0];
        B[0] = cos(phi) * dt; B[1] = -v * sin(phi) * dt; B[2] = sin(phi) * dt; B[3] = v * cos(phi) * dt;
        C[0] = phi * v * sin(phi) * dt; C[1] = -phi * v * cos(phi) * dt;
        return;
    }
    double phi = st[2], v = ut[0];
    A[2] = -v * dt * sin(phi); A[5] = v * dt * cos(phi);
    B[0] = cos(phi) * dt; B[2] = sin(phi) * dt;
    C[0] = phi * v * sin(phi) * dt; C[1] = -phi * v * cos(phi) * dt;
    if (c.dynamics == 0) {
        double psi = ut[1], cp = cos(psi);
        B[4] = tan(psi) * dt / c.L; B[5] = v * dt / (c.L * cp * cp);
        C[2] = -psi * v * dt / (c.L * cp * cp);
    } else {
        B[5] = dt;
    }
}

// LDS carve-up (doubles)
struct Lds {
    double *s, *u, *d, *phin, *ref, *Ak, *Bk, *Ck, *Q0, *Q1, *Q2;
    double *hs;        // [T][9] hinge sums
    double *Hw, *gw;   // [T][10], [T][4]
    double *gst;       // [T][8]  stage gradient (objective + C'lam)
    double *gh;        // [T][8]  Newton right-hand side gradient
    double *cw, *cl, *rp, *rc, *dw, *dl;   // [T][NC]
    double *Minv, *Mxv, *gv, *dy;          // [T][9], [T][15], [T][3], [T][8]
    double *Mm, *W6, *P, *pv, *red;        // 64, 36, 25, 8, NT
    __device__ static size_t doubles(int T) {
        return (size_t)3 * (T + 1) + 2 * T + T + T + 3 * (T + 1) + 9 * T + 6 * T + 3 * T + 3 * T
             + 9 * T + 10 * T + 4 * T + 8 * T + 8 * T + 6 * NC * T + 9 * T + 15 * T + 3 * T + 8 * T
             + 64 + 36 + 25 + 8 + NT;
    }
    __device__ void carve(double *base, int T) {
        double *p = base;
        s = p; p += 3 * (T + 1); u = p; p += 2 * T; d = p; p += T; phin = p; p += T; ref = p; p += 3 * (T + 1);
        Ak = p; p += 9 * T; Bk = p; p += 6 * T; Ck = p; p += 3 * T; Q0 = p; p += T; Q1 = p; p += T; Q2 = p; p += T;
        hs = p; p += 9 * T; Hw = p; p += 10 * T; gw = p; p += 4 * T; gst = p; p += 8 * T; gh = p; p += 8 * T;
        cw = p; p += NC * T; cl = p; p += NC * T; rp = p; p += NC * T; rc = p; p += NC * T; dw = p; p += NC * T; dl = p; p += NC * T;
        Minv = p; p += 9 * T; Mxv = p; p += 15 * T; gv = p; p += 3 * T; dy = p; p += 8 * T;
        Mm = p; p += 64; W6 = p; p += 36; P = p; p += 25; pv = p; p += 8; red = p; p += NT;
    }
};
inline size_t lds_bytes(int T)
{
    size_t n = (size_t)3 * (T + 1) + 2 * T + T + T + 3 * (T + 1) + 9 * T + 6 * T + 3 * T + 3 * T
             + 9 * T + 10 * T + 4 * T + 8 * T + 8 * T + 6 * NC * T + 9 * T + 15 * T + 3 * T + 8 * T
             + 64 + 36 + 25 + 8 + NT;
    return n * sizeof(double);
}

// constraint row k of stage t: value c'y, rhs e.  y = [s(3) up(2) u(2) d]
__device__ __forceinline__ double con_val(int k, double u0, double u1, double up0, double up1, double dd)
{
    switch (k) {
        case 0: return u0; case 1: return -u0; case 2: return u1; case 3: return -u1;
        case 4: return dd; case 5: return -dd;
        case 6: return u0 - up0; case 7: return -(u0 - up0); case 8: return u1 - up1; default: return -(u1 - up1);
    }
}
__device__ __forceinline__ double con_rhs(const Cfg &c, int k)
{
    switch (k) {
        case 0: case 1: return c.umax0; case 2: case 3: return c.umax1;
        case 4: return c.max_sd; case 5: return -c.min_sd;
        case 6: case 7: return c.ab0; default: return c.ab1;
    }
}
// index in y (5..7 direct, 3..4 up) and sign for the (at most two) non-zeros of row k
__device__ __forceinline__ void con_pat(int k, int &ia, double &ca, int &ib, double &cb)
{
    ib = -1; cb = 0;
    switch (k) {
        case 0: ia = 5; ca = 1; break; case 1: ia = 5; ca = -1; break;
        case 2: ia = 6; ca = 1; break; case 3: ia = 6; ca = -1; break;
        case 4: ia = 7; ca = 1; break; case 5: ia = 7; ca = -1; break;
        case 6: ia = 5; ca = 1; ib = 3; cb = -1; break; case 7: ia = 5; ca = -1; ib = 3; cb = 1; break;
        case 8: ia = 6; ca = 1; ib = 4; cb = -1; break; default: ia = 6; ca = -1; ib = 4; cb = 1; break;
    }
}
__device__ __forceinline__ bool con_on(int t, int k) { return k < 6 || t >= 1; }

__device__ __forceinline__ double block_reduce(double v, double *red, int tid, bool is_max)
{
    // wave reduce then across 4 waves via LDS
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = is_max ? (o > v ? o : v) : v + o;
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < NT / 64; ++w) r = is_max ? (red[w] > r ? red[w] : r) : r + red[w];
    return r;
}

// F' p for stage t:  out(8) = F_t' p(5),  F = [A 0 B 0; 0 0 I 0]
__device__ __forceinline__ double Ft_p(const double *A, const double *B, const double *p, int i)
{
    if (i < 3) return A[0 * 3 + i] * p[0] + A[1 * 3 + i] * p[1] + A[2 * 3 + i] * p[2];
    if (i < 5) return 0.0;
    if (i < 7) { int j = i - 5; return B[0 * 2 + j] * p[0] + B[1 * 2 + j] * p[1] + B[2 * 2 + j] * p[2] + p[3 + j]; }
    return 0.0;
}
// Kt[a][i] : 6x8 map y -> (s_next(3), up_next(2), d)
__device__ __forceinline__ double Kt(const double *A, const double *B, int a, int i)
{
    if (a < 3) { if (i < 3) return A[a * 3 + i]; if (i >= 5 && i < 7) return B[a * 2 + (i - 5)]; return 0.0; }
    if (a < 5) return (i == 5 + (a - 3)) ? 1.0 : 0.0;
    return i == 7 ? 1.0 : 0.0;
}

// The whole solve.  Must be called by all NT threads of the block with `smem` >= lds_bytes(T).
__device__ inline void solve(const Args &a, double *smem)
{
    const Cfg &c = a.c;
    const int T = c.T, N = c.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Lds L; L.carve(smem, T);
    const double vref = *a.ref_speed;
    // ---- load nominal, reference; linearise -------------------------------------------------
    for (int i = tid; i < 3 * (T + 1); i += NT) { L.s[i] = a.in_s[i]; L.ref[i] = a.ref[i]; }
    for (int i = tid; i < 2 * T; i += NT) L.u[i] = a.in_u[i];
    __syncthreads();
    if (tid < T) {
        int t = tid;
        double st[3] = { L.s[t], L.s[(T + 1) + t], L.s[2 * (T + 1) + t] }, ut[2] = { L.u[t], L.u[T + t] };
        lin_model(c, st, ut, &L.Ak[9 * t], &L.Bk[6 * t], &L.Ck[3 * t]);
        L.phin[t] = st[2];
    }
    __syncthreads();
    // ---- rotation-consistency penalty -> scalar quadratic per stage --------------
    for (int t = wave; t < T; t += NT / 64) {
        double cs = cos(L.phin[t]), sn = sin(L.phin[t]);
        double q0 = 0, q1 = 0, q2 = 0;
        for (int n = lane; n < N; n += 64) {
            double ax = a.ax[t * N + n], ay = a.ay[t * N + n], gx = a.gx[t * N + n], gy = a.gy[t * N + n];
            double k0x = gx + cs * ax + sn * ay, k0y = gy - sn * ax + cs * ay;
            double k1x = -sn * ax + cs * ay, k1y = -cs * ax - sn * ay;
            q0 += k0x * k0x + k0y * k0y; q1 += 2 * (k0x * k1x + k0y * k1y); q2 += k1x * k1x + k1y * k1y;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { q0 += __shfl_xor(q0, off, 64); q1 += __shfl_xor(q1, off, 64); q2 += __shfl_xor(q2, off, 64); }
        if (lane == 0) { L.Q0[t] = q0; L.Q1[t] = q1; L.Q2[t] = q2; }
    }
    // ---- initial point (same rule as the oracle) ------------------------------------------------
    if (tid < T) {
        int t = tid;
        double lim0 = 0.99 * c.umax0, lim1 = 0.99 * c.umax1;
        double v0 = L.u[t], v1 = L.u[T + t];
        L.u[t] = v0 > lim0 ? lim0 : (v0 < -lim0 ? -lim0 : v0);
        L.u[T + t] = v1 > lim1 ? lim1 : (v1 < -lim1 ? -lim1 : v1);
        double lo = c.min_sd + 0.01 * (c.max_sd - c.min_sd), hi = c.max_sd - 0.01 * (c.max_sd - c.min_sd);
        double dv = a.d_in ? a.d_in[t] : c.max_sd;
        L.d[t] = dv > hi ? hi : (dv < lo ? lo : dv);
    }
    __syncthreads();
    if (tid == 0) {          // roll the state out with the clipped controls
        for (int t = 0; t < T; ++t) {
            const double *A = &L.Ak[9 * t], *B = &L.Bk[6 * t], *C = &L.Ck[3 * t];
            for (int r = 0; r < 3; ++r) {
                double v = C[r];
                for (int k = 0; k < 3; ++k) v += A[3 * r + k] * L.s[k * (T + 1) + t];
                v += B[2 * r] * L.u[t] + B[2 * r + 1] * L.u[T + t];
                L.s[r * (T + 1) + t + 1] = v;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < NC * T; i += NT) {
        int t = i / NC, k = i % NC;
        double up0 = t ? L.u[t - 1] : 0, up1 = t ? L.u[T + t - 1] : 0;
        double sl = con_rhs(c, k) - con_val(k, L.u[t], L.u[T + t], up0, up1, L.d[t]);
        bool on = con_on(t, k);
        L.cw[i] = on ? (sl > 1e-2 ? sl : 1e-2) : 1.0;
        L.cl[i] = on ? 1.0 : 0.0;
    }
    __syncthreads();
    const double mcnt = (double)(6 * T + 4 * (T - 1));
    const double wz = c.dynamics == 2 ? 0.0 : 1.0;
    int status = 1, it;
    for (it = 0; it < 100; ++it) {
        // ---- (1) hinge sums per stage -----------------------------------------------------------
        for (int t = wave; t < T; t += NT / 64) {
            double px = L.s[t + 1], py = L.s[(T + 1) + t + 1], dd = L.d[t];
            double sxx = 0, sxy = 0, syy = 0, sx = 0, sy = 0, s1 = 0, ix = 0, iy = 0, i1 = 0;
            for (int n = lane; n < N; n += 64) {
                double ax = a.ax[t * N + n], ay = a.ay[t * N + n];
                double Im = ax * px + ay * py - (a.blam[t * N + n] + a.ee[t * N + n]) - dd;
                if (!c.accelerated || Im < 0) {
                    sxx += ax * ax; sxy += ax * ay; syy += ay * ay; sx += ax; sy += ay; s1 += 1.0;
                    ix += Im * ax; iy += Im * ay; i1 += Im;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                sxx += __shfl_xor(sxx, off, 64); sxy += __shfl_xor(sxy, off, 64); syy += __shfl_xor(syy, off, 64);
                sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); s1 += __shfl_xor(s1, off, 64);
                ix += __shfl_xor(ix, off, 64); iy += __shfl_xor(iy, off, 64); i1 += __shfl_xor(i1, off, 64);
            }
            if (lane == 0) {
                double *h = &L.hs[9 * t];
                h[0] = sxx; h[1] = sxy; h[2] = syy; h[3] = sx; h[4] = sy; h[5] = s1; h[6] = ix; h[7] = iy; h[8] = i1;
            }
        }
        __syncthreads();
        // ---- (2) stage cost derivatives wrt w = (s_next, d) and stage gradient -------------------
        if (tid < T) {
            int t = tid;
            const double *h = &L.hs[9 * t];
            double st[3] = { L.s[t + 1], L.s[(T + 1) + t + 1], L.s[2 * (T + 1) + t + 1] };
            double w3[3] = { 1, 1, wz };
            double gs[3], Hs00, Hs01, Hs11, Hs22;
            for (int r = 0; r < 3; ++r) gs[r] = 2 * c.ws * w3[r] * (st[r] - L.ref[r * (T + 1) + t + 1]);
            Hs00 = 2 * c.ws; Hs11 = 2 * c.ws; Hs22 = 2 * c.ws * wz; Hs01 = 0;
            double dl = st[2] - L.phin[t];
            gs[2] += 0.5 * c.ro2 * (L.Q1[t] + 2 * L.Q2[t] * dl); Hs22 += c.ro2 * L.Q2[t];
            gs[0] += c.ro1 * h[6]; gs[1] += c.ro1 * h[7];
            Hs00 += c.ro1 * h[0]; Hs01 += c.ro1 * h[1]; Hs11 += c.ro1 * h[2];
            double *Hw = &L.Hw[10 * t], *gw = &L.gw[4 * t];
            // packed symmetric 4x4: (00,01,02,03,11,12,13,22,23,33), index 3 = d
            Hw[0] = Hs00; Hw[1] = Hs01; Hw[2] = 0; Hw[3] = -c.ro1 * h[3];
            Hw[4] = Hs11; Hw[5] = 0; Hw[6] = -c.ro1 * h[4];
            Hw[7] = Hs22; Hw[8] = 0; Hw[9] = c.ro1 * h[5];
            gw[0] = gs[0]; gw[1] = gs[1]; gw[2] = gs[2]; gw[3] = -c.ro1 * h[8] - c.slack_gain;
        }
        __syncthreads();
        // stage gradient gst[t][i] = K_t' gw + direct terms + C' lam ; residuals rp
        for (int i = tid; i < 8 * T; i += NT) {
            int t = i >> 3, j = i & 7;
            const double *A = &L.Ak[9 * t], *B = &L.Bk[6 * t], *gw = &L.gw[4 * t];
            double v = Kt(A, B, 0, j) * gw[0] + Kt(A, B, 1, j) * gw[1] + Kt(A, B, 2, j) * gw[2] + Kt(A, B, 5, j) * gw[3];
            if (j == 5) v += 2 * c.wu * (L.u[t] - vref) + c.eps_u * L.u[t];
            if (j == 6) v += c.eps_u * L.u[T + t];
            for (int k = 0; k < NC; ++k) {
                int ia, ib; double ca, cb; con_pat(k, ia, ca, ib, cb);
                if (ia == j) v += ca * L.cl[t * NC + k];
                if (ib == j) v += cb * L.cl[t * NC + k];
            }
            L.gst[i] = v;
        }
        for (int i = tid; i < NC * T; i += NT) {
            int t = i / NC, k = i % NC;
            double up0 = t ? L.u[t - 1] : 0, up1 = t ? L.u[T + t - 1] : 0;
            L.rp[i] = con_on(t, k) ? con_val(k, L.u[t], L.u[T + t], up0, up1, L.d[t]) + L.cw[i] - con_rhs(c, k) : 0.0;
        }
        __syncthreads();
        // ---- (3) reduced gradient by an adjoint sweep (wave 0), norms ----------------------------
        if (wave == 0) {
            if (lane < 8) L.pv[lane] = 0;
            wsync();
            for (int t = T - 1; t >= 0; --t) {
                double v = 0;
                if (lane < 8) v = L.gst[8 * t + lane] + Ft_p(&L.Ak[9 * t], &L.Bk[6 * t], L.pv, lane);
                wsync();
                if (lane < 8) { L.gh[8 * t + lane] = v; if (lane < 5) L.pv[lane] = v; }
                wsync();
            }
        }
        __syncthreads();
        double rdn = 0, gn = 0, rpn = 0, mu = 0;
        for (int i = tid; i < 8 * T; i += NT) {
            int j = i & 7;
            if (j >= 5) { double v = fabs(L.gh[i]); if (v > rdn) rdn = v; }
        }
        for (int i = tid; i < 4 * T; i += NT) { double v = fabs(L.gw[i]); if (v > gn) gn = v; }
        for (int i = tid; i < NC * T; i += NT) { double v = fabs(L.rp[i]); if (v > rpn) rpn = v; mu += L.cl[i] * L.cw[i]; }
        rdn = block_reduce(rdn, L.red, tid, true);
        gn = block_reduce(gn, L.red, tid, true);
        rpn = block_reduce(rpn, L.red, tid, true);
        mu = block_reduce(mu, L.red, tid, false) / mcnt;
        double sc = 1 + gn;
        if (rdn <= 1e-9 * sc && rpn <= 1e-10 && mu <= 1e-11 * sc) { status = 0; break; }

        double sigma = 0;
        bool fail = false;
        for (int pass = 0; pass < 2; ++pass) {
            // ---- (4) complementarity target and Newton gradient -------------------------------------
            for (int i = tid; i < NC * T; i += NT)
                L.rc[i] = L.cl[i] * L.cw[i] + (pass ? L.dl[i] * L.dw[i] - sigma * mu : 0.0);
            __syncthreads();
            for (int i = tid; i < 8 * T; i += NT) {
                int t = i >> 3, j = i & 7;
                double v = L.gst[i];
                for (int k = 0; k < NC; ++k) {
                    if (!con_on(t, k)) continue;
                    int ia, ib; double ca, cb; con_pat(k, ia, ca, ib, cb);
                    double q = (L.cl[t * NC + k] * L.rp[t * NC + k] - L.rc[t * NC + k]) / L.cw[t * NC + k];
                    if (ia == j) v += ca * q;
                    if (ib == j) v += cb * q;
                }
                L.gh[i] = v;
            }
            __syncthreads();
            // ---- (5) Riccati backward sweep (wave 0) ---------------------------------------------------
            if (wave == 0) {
                if (lane < 25) L.P[lane] = 0;
                if (lane < 8) L.pv[lane] = 0;
                wsync();
                for (int t = T - 1; t >= 0; --t) {
                    const double *A = &L.Ak[9 * t], *B = &L.Bk[6 * t];
                    if (pass == 0) {
                        // W6 = [Hss+Pss Psu Hsd; Pus Puu 0; Hds 0 Hdd]
                        if (lane < 36) {
                            int r = lane / 6, q = lane % 6;
                            const double *Hw = &L.Hw[10 * t];
                            double v = 0;
                            if (r < 5 && q < 5) v = L.P[r * 5 + q];
                            int rr = r < 3 ? r : (r == 5 ? 3 : -1), qq = q < 3 ? q : (q == 5 ? 3 : -1);
                            if (rr >= 0 && qq >= 0) {
                                int lo = rr < qq ? rr : qq, hi = rr < qq ? qq : rr;
                                const int base[4] = { 0, 4, 7, 9 };
                                v += Hw[base[lo] + (hi - lo)];
                            }
                            L.W6[lane] = v;
                        }
                        wsync();
                        int i = lane >> 3, j = lane & 7;
                        double m = 0;
                        for (int r = 0; r < 6; ++r) {
                            double kr = Kt(A, B, r, i);
                            if (kr == 0.0) continue;
                            double acc = 0;
                            for (int q = 0; q < 6; ++q) acc += L.W6[r * 6 + q] * Kt(A, B, q, j);
                            m += kr * acc;
                        }
                        // direct objective terms and barrier terms
                        if (i == 5 && j == 5) m += 2 * c.wu + c.eps_u;
                        if (i == 6 && j == 6) m += c.eps_u;
                        for (int k = 0; k < NC; ++k) {
                            if (!con_on(t, k)) continue;
                            int ia, ib; double ca, cb; con_pat(k, ia, ca, ib, cb);
                            double dg = L.cl[t * NC + k] / L.cw[t * NC + k];
                            double ci = (ia == i ? ca : 0.0) + (ib == i ? cb : 0.0);
                            double cj = (ia == j ? ca : 0.0) + (ib == j ? cb : 0.0);
                            m += dg * ci * cj;
                        }
                        L.Mm[lane] = m;
                        wsync();
                        // Cholesky of Mvv (rows/cols 5..7), every lane redundantly
                        double m00 = L.Mm[5 * 8 + 5], m10 = L.Mm[6 * 8 + 5], m20 = L.Mm[7 * 8 + 5];
                        double m11 = L.Mm[6 * 8 + 6], m21 = L.Mm[7 * 8 + 6], m22 = L.Mm[7 * 8 + 7];
                        double l00 = sqrt(m00), l10 = m10 / l00, l20 = m20 / l00;
                        double d11 = m11 - l10 * l10; double l11 = sqrt(d11), l21 = (m21 - l20 * l10) / l11;
                        double d22 = m22 - l20 * l20 - l21 * l21; double l22 = sqrt(d22);
                        if (!(m00 > 0) || !(d11 > 0) || !(d22 > 0)) fail = true;
                        // inverse of L (lower): Li
                        double i00 = 1 / l00, i11 = 1 / l11, i22 = 1 / l22;
                        double i10 = -l10 * i00 * i11, i21 = -l21 * i11 * i22, i20 = -(l20 * i00 + l21 * i10) * i22;
                        // Minv = Li' Li
                        double n00 = i00 * i00 + i10 * i10 + i20 * i20, n01 = i10 * i11 + i20 * i21, n02 = i20 * i22;
                        double n11 = i11 * i11 + i21 * i21, n12 = i21 * i22, n22 = i22 * i22;
                        if (lane == 0) {
                            double *Mi = &L.Minv[9 * t];
                            Mi[0] = n00; Mi[1] = n01; Mi[2] = n02; Mi[3] = n01; Mi[4] = n11; Mi[5] = n12; Mi[6] = n02; Mi[7] = n12; Mi[8] = n22;
                        }
                        if (lane < 15) L.Mxv[15 * t + lane] = L.Mm[(lane / 3) * 8 + 5 + (lane % 3)];
                        wsync();
                        // P_t = Mxx - Mxv Minv Mvx
                        if (lane < 25) {
                            int r = lane / 5, q = lane % 5;
                            const double *Mi = &L.Minv[9 * t], *Xr = &L.Mxv[15 * t + 3 * r], *Xq = &L.Mxv[15 * t + 3 * q];
                            double acc = 0;
                            for (int x = 0; x < 3; ++x) acc += Xr[x] * (Mi[3 * x] * Xq[0] + Mi[3 * x + 1] * Xq[1] + Mi[3 * x + 2] * Xq[2]);
                            L.P[lane] = L.Mm[r * 8 + q] - acc;
                        }
                    }
                    // vector part: ghat_t = gh_t + F' p_{t+1}; gv; p_t = ghat_x - Mxv Minv gv
                    double v = 0;
                    if (lane < 8) v = L.gh[8 * t + lane] + Ft_p(A, B, L.pv, lane);
                    wsync();
                    if (lane < 8) { L.gh[8 * t + lane] = v; if (lane >= 5) L.gv[3 * t + lane - 5] = v; }
                    wsync();
                    if (lane < 5) {
                        const double *Mi = &L.Minv[9 * t], *X = &L.Mxv[15 * t + 3 * lane], *g = &L.gv[3 * t];
                        double acc = 0;
                        for (int x = 0; x < 3; ++x) acc += X[x] * (Mi[3 * x] * g[0] + Mi[3 * x + 1] * g[1] + Mi[3 * x + 2] * g[2]);
                        L.pv[lane] = v - acc;
                    }
                    wsync();
                }
                // ---- (6) forward sweep --------------------------------------------------------------------
                // dx (5) in pv[0..4]
                if (lane < 8) L.pv[lane] = 0;
                wsync();
                for (int t = 0; t < T; ++t) {
                    const double *A = &L.Ak[9 * t], *B = &L.Bk[6 * t];
                    double dv = 0;
                    if (lane < 3) {
                        const double *Mi = &L.Minv[9 * t], *g = &L.gv[3 * t];
                        double r3[3];
                        for (int x = 0; x < 3; ++x) {
                            double acc = g[x];
                            for (int r = 0; r < 5; ++r) acc += L.Mxv[15 * t + 3 * r + x] * L.pv[r];
                            r3[x] = acc;
                        }
                        dv = -(Mi[3 * lane] * r3[0] + Mi[3 * lane + 1] * r3[1] + Mi[3 * lane + 2] * r3[2]);
                    }
                    wsync();
                    if (lane < 3) L.dy[8 * t + 5 + lane] = dv;
                    if (lane < 5) L.dy[8 * t + lane] = L.pv[lane];
                    wsync();
                    double nx = 0;
                    if (lane < 3) {
                        const double *y = &L.dy[8 * t];
                        nx = A[3 * lane] * y[0] + A[3 * lane + 1] * y[1] + A[3 * lane + 2] * y[2] + B[2 * lane] * y[5] + B[2 * lane + 1] * y[6];
                    } else if (lane < 5) nx = L.dy[8 * t + 5 + (lane - 3)];
                    wsync();
                    if (lane < 5) L.pv[lane] = nx;
                    wsync();
                }
                // dx_T (state step at the horizon end) kept in pv[0..2]
            }
            __syncthreads();
            // ---- (7) slack / multiplier steps, step length -------------------------------------------
            double al = 1.0, muaff = 0;
            for (int i = tid; i < NC * T; i += NT) {
                int t = i / NC, k = i % NC;
                if (!con_on(t, k)) { L.dw[i] = 0; L.dl[i] = 0; continue; }
                const double *y = &L.dy[8 * t];
                double cdx = con_val(k, y[5], y[6], y[3], y[4], y[7]);
                double dwv = -L.rp[i] - cdx, dlv = -(L.rc[i] + L.cl[i] * dwv) / L.cw[i];
                L.dw[i] = dwv; L.dl[i] = dlv;
                double fr = pass ? 0.995 : 1.0;
                if (dwv < 0) { double x = -fr * L.cw[i] / dwv; if (x < al) al = x; }
                if (dlv < 0) { double x = -fr * L.cl[i] / dlv; if (x < al) al = x; }
            }
            al = -block_reduce(-al, L.red, tid, true);
            if (pass == 0) {
                for (int i = tid; i < NC * T; i += NT) muaff += (L.cl[i] + al * L.dl[i]) * (L.cw[i] + al * L.dw[i]);
                muaff = block_reduce(muaff, L.red, tid, false) / mcnt;
                double r = muaff / mu; sigma = r * r * r;
            } else {
                // ---- (8) update the iterate ---------------------------------------------------------------
                for (int i = tid; i < NC * T; i += NT) { L.cw[i] += al * L.dw[i]; L.cl[i] += al * L.dl[i]; }
                if (tid < T) {
                    int t = tid; const double *y = &L.dy[8 * t];
                    L.u[t] += al * y[5]; L.u[T + t] += al * y[6]; L.d[t] += al * y[7];
                    if (t >= 1) for (int r = 0; r < 3; ++r) L.s[r * (T + 1) + t] += al * y[r];
                }
                if (tid < 3) L.s[tid * (T + 1) + T] += al * L.pv[tid];
                __syncthreads();
            }
        }
        int anyfail = __syncthreads_or(fail ? 1 : 0);
        if (anyfail) { status = 2; break; }
    }
    __syncthreads();
    // consistent final rollout (removes accumulated rounding in s)
    if (tid == 0) {
        for (int t = 0; t < T; ++t) {
            const double *A = &L.Ak[9 * t], *B = &L.Bk[6 * t], *C = &L.Ck[3 * t];
            for (int r = 0; r < 3; ++r) {
                double v = C[r];
                for (int k = 0; k < 3; ++k) v += A[3 * r + k] * L.s[k * (T + 1) + t];
                v += B[2 * r] * L.u[t] + B[2 * r + 1] * L.u[T + t];
                L.s[r * (T + 1) + t + 1] = v;
            }
        }
    }
    __syncthreads();
    if (status == 0) {       // otherwise keep the nominal (reference :696-700)
        for (int i = tid; i < 3 * (T + 1); i += NT) a.out_s[i] = L.s[i];
        for (int i = tid; i < 2 * T; i += NT) a.out_u[i] = L.u[i];
        for (int i = tid; i < T; i += NT) a.out_d[i] = L.d[i];
    } else if (a.out_s != a.in_s) {
        for (int i = tid; i < 3 * (T + 1); i += NT) a.out_s[i] = a.in_s[i];
        for (int i = tid; i < 2 * T; i += NT) a.out_u[i] = a.in_u[i];
        if (a.d_in) for (int i = tid; i < T; i += NT) a.out_d[i] = a.d_in[i];
    }
    if (tid == 0) { *a.status = status; *a.ipm_iters = it; }
}

}  // namespace su
