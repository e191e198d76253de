// K3 device code: the state/control ("su") problem of one ADMM iteration, solved by ONE
// workgroup as a primal-dual interior point method (Mehrotra predictor-corrector) whose Newton
// systems are block-tridiagonal and solved by a Riccati recursion over the T stages
// (reference: construct_su_prob rda_solver.py:216-231, nav_cost_cons :313-328,
// update_su_cost_cons :330-387, Im_su/Hm_su :831-872, dynamics/bounds :911-947, C0/C1 cost
// :1011-1032).
//
//   stage vector   y_t = [ s_t(3) | up_t(2) = u_{t-1} | u_t(2) | d_t ]            (8)
//   dynamics       s_{t+1} = A_t s_t + B_t u_t + C_t ,  up_{t+1} = u_t
//   stage cost     q_t(s_{t+1}, d_t)  [tracking + rotation penalty + sum_n hinge^2]
//                  + wu (u_t[0]-v_ref)^2 + eps_u/2 |u_t|^2 - slack_gain d_t
//   inequalities   |u_t| <= u_max, d_min <= d_t <= d_max, |u_t - up_t| <= a_max dt (t >= 1)
//
// Work split inside an interior-point iteration
//   all 4 waves : per-stage sums over the obstacles with an active hinge (the only N-dependent
//                 work; [T][N] structure-of-arrays coefficients, (stage, chunk) thread mapping),
//                 stage gradients / Hessian bases, slack and multiplier updates, reductions
//   wave 0      : the serial Riccati sweeps.  Every lane carries the cost-to-go (P 5x5, p 5) in
//                 registers and performs the stage update redundantly, with the next stage's
//                 constants prefetched from LDS: no barrier and no LDS round trip on the serial
//                 critical path
//   wave 1      : adjoint sweep for the reduced gradient, concurrently with wave 0
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
    const double *ax, *ay, *blam, *ee, *gx, *gy;   // condensed obstacle terms of obstacle shard 0, each [T][Nloc]
    int P, Nloc; size_t chunk;       // P obstacle shards (N = P*Nloc); shard r's arrays start `chunk` doubles after shard r-1's
    const double *d_in;              // [T] initial guess for d
    double *out_s, *out_u, *out_d;   // results (may alias in_*)
    int *status;                     // 0 ok / 1 not converged / 2 factorisation failed
    int *ipm_iters;
    long long *prof;                 // optional per-phase cycle counters (debug), may be null
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

// LDS carve-up (doubles).  Per-stage constants of the serial sweeps are packed into 16-byte aligned records so
// that they are fetched with ds_read_b128 and prefetched one stage ahead.
constexpr int RM = 52;   // matrix-sweep record: upper triangle of the 8x8 stage Hessian base (36) | B (6) | a13 a23 | pad
constexpr int RV = 40;   // vector-sweep record: A (9) | B (6) | W (15) | Minv sym (6) | kk (3) | pad
__device__ __host__ inline int ev(int n) { return (n + 1) & ~1; }
struct Lds {
    double *s, *u, *d, *phin, *ref, *Ak, *Bk, *Ck, *Q0, *Q1, *Q2;
    double *Jm;        // [T][32] d(s_next, d)/dy : 4x8 per stage (constant during the solve)
    double *part;      // [NT][9] partial sums of the chunked reductions
    double *hs;        // [T][9]  hinge sums
    double *Hw, *gw;   // [T][16], [T][4]
    double *bw;        // [T][5]  barrier weights (u0 box, u1 box, d box, rate u0, rate u1)
    double *cy;        // [T][5]  C'lam per stage (entries 3..7 of y)
    double *gst;       // [T][8]  stage gradient (objective + C'lam)
    double *gh;        // [T][8]  Newton right-hand side gradient
    double *gad;       // [T][3]  reduced gradient (adjoint sweep)
    double *recM;      // [T][RM]
    double *recV;      // [T][RV]
    double *cw, *cl, *rp, *rc, *dw, *dl;   // [T][NC]
    double *dy;                            // [T][8]
    double *pv, *red;                      // 8, NT
    __device__ void carve(double *b, int T) {
        double *p = b;
        s = p; p += ev(3 * (T + 1)); u = p; p += 2 * T; d = p; p += ev(T); phin = p; p += ev(T); ref = p; p += ev(3 * (T + 1));
        Ak = p; p += ev(9 * T); Bk = p; p += 6 * T; Ck = p; p += ev(3 * T); Q0 = p; p += ev(T); Q1 = p; p += ev(T); Q2 = p; p += ev(T);
        Jm = p; p += 32 * T; hs = p; p += ev(9 * T); Hw = p; p += 16 * T; gw = p; p += 4 * T; bw = p; p += ev(5 * T); cy = p; p += ev(5 * T);
        gst = p; p += 8 * T; gh = p; p += 8 * T; gad = p; p += ev(3 * T); recM = p; part = p; p += (RM * T > 9 * NT ? RM * T : 9 * NT); recV = p;   // part (phase 1) and recM (phases 3-4) never live together
         p += RV * T;
        cw = p; p += NC * T; cl = p; p += NC * T; rp = p; p += NC * T; rc = p; p += NC * T; dw = p; p += NC * T; dl = p; p += NC * T;
        dy = p; p += 8 * T; pv = p; p += 8; red = p; p += NT;
    }
};
inline size_t lds_bytes(int T)
{
    size_t n = (size_t)2 * ev(3 * (T + 1)) + 2 * T + 2 * ev(T) + ev(9 * T) + 6 * T + ev(3 * T) + 3 * ev(T)
             + 32 * T + ev(9 * T) + 16 * T + 4 * T + 2 * ev(5 * T) + 8 * T + 8 * T + ev(3 * T) + (RM * T > 9 * NT ? RM * T : 9 * NT) + RV * T
             + 6 * NC * T + 8 * T + 8 + NT;
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
__device__ __forceinline__ bool con_on(int t, int k) { return k < 6 || t >= 1; }

__device__ __forceinline__ double block_reduce(double v, double *red, int tid, bool is_max)
{
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

// C' x for one stage: x = per-row values of the 10 inequality rows -> entries 3..7 of y
__device__ __forceinline__ void con_T(const double *x, int t, double &y3, double &y4, double &y5, double &y6, double &y7)
{
    double r0 = t >= 1 ? x[6] - x[7] : 0.0, r1 = t >= 1 ? x[8] - x[9] : 0.0;
    y5 = x[0] - x[1] + r0; y6 = x[2] - x[3] + r1; y7 = x[4] - x[5]; y3 = -r0; y4 = -r1;
}

// The whole solve.  Must be called by all NT threads of the block with `smem` >= lds_bytes(T).
__device__ inline void solve(const Args &a, double *smem)
{
    const Cfg &c = a.c;
    const int T = c.T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Lds L; L.carve(smem, T);
    const double vref = *a.ref_speed;
    // (stage, chunk) mapping of the obstacle reductions
    const int nch = NT / T;                       // chunks per stage (T <= 64 -> nch >= 4)
    const int rt = tid / nch, rc_ = tid % nch;    // this thread's stage and chunk
    const bool ract = rt < T;

    // ---- load nominal, reference; linearise -------------------------------------------------
    for (int i = tid; i < 3 * (T + 1); i += NT) { L.s[i] = a.in_s[i]; L.ref[i] = a.ref[i]; }
    for (int i = tid; i < 2 * T; i += NT) L.u[i] = a.in_u[i];
    __syncthreads();
    if (tid < T) {
        int t = tid;
        double st[3] = { L.s[t], L.s[(T + 1) + t], L.s[2 * (T + 1) + t] }, ut[2] = { L.u[t], L.u[T + t] };
        lin_model(c, st, ut, &L.Ak[9 * t], &L.Bk[6 * t], &L.Ck[3 * t]);
        L.phin[t] = st[2];
        // J_t = d(s_next, d)/dy : rows 0..2 = [A | 0 | B | 0], row 3 = e_7
        double *J = &L.Jm[32 * t];
        for (int i = 0; i < 32; ++i) J[i] = 0;
        for (int r = 0; r < 3; ++r) {
            for (int q = 0; q < 3; ++q) J[r * 8 + q] = L.Ak[9 * t + 3 * r + q];
            for (int q = 0; q < 2; ++q) J[r * 8 + 5 + q] = L.Bk[6 * t + 2 * r + q];
        }
        J[3 * 8 + 7] = 1.0;
        // constant parts of the sweep records
        double *rv = &L.recV[RV * t];
        for (int i = 0; i < 6; ++i) rv[9 + i] = L.Bk[6 * t + i];
        for (int i = 0; i < 9; ++i) rv[i] = L.Ak[9 * t + i];
    }
    __syncthreads();
    // ---- rotation-consistency penalty -> scalar quadratic per stage --------------
    {
        double q0 = 0, q1 = 0, q2 = 0;
        if (ract) {
            double cs = cos(L.phin[rt]), sn = sin(L.phin[rt]);
            for (int r = 0; r < a.P; ++r) {
                const size_t o = r * a.chunk + (size_t)rt * a.Nloc;
                for (int n = rc_; n < a.Nloc; n += nch) {
                    double ax = a.ax[o + n], ay = a.ay[o + n], gx = a.gx[o + n], gy = a.gy[o + n];
                    double k0x = gx + cs * ax + sn * ay, k0y = gy - sn * ax + cs * ay;
                    double k1x = -sn * ax + cs * ay, k1y = -cs * ax - sn * ay;
                    q0 += k0x * k0x + k0y * k0y; q1 += 2 * (k0x * k1x + k0y * k1y); q2 += k1x * k1x + k1y * k1y;
                }
            }
        }
        L.part[tid * 9] = q0; L.part[tid * 9 + 1] = q1; L.part[tid * 9 + 2] = q2;
        __syncthreads();
        if (tid < T) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int k = 0; k < nch; ++k) { const double *pp = &L.part[(tid * nch + k) * 9]; s0 += pp[0]; s1 += pp[1]; s2 += pp[2]; }
            L.Q0[tid] = s0; L.Q1[tid] = s1; L.Q2[tid] = s2;
        }
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
        L.cl[i] = on ? 1.0 / L.cw[i] : 0.0;            // lam = mu0 / w with mu0 = 1
    }
    __syncthreads();
    const double mcnt = (double)(6 * T + 4 * (T - 1));
    const double wz = c.dynamics == 2 ? 0.0 : 1.0;

    // Newton right-hand side gradient gh = gst + C'((lam*rp - rc)/w)   (threads < T, one stage each)
    auto build_gh = [&]() {
        if (tid < T) {
            int t = tid;
            double x[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) { int i = t * NC + k; x[k] = (L.cl[i] * L.rp[i] - L.rc[i]) / L.cw[i]; }
            double y3, y4, y5, y6, y7; con_T(x, t, y3, y4, y5, y6, y7);
            const double *g = &L.gst[8 * t]; double *o = &L.gh[8 * t];
            o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; o[3] = g[3] + y3; o[4] = g[4] + y4; o[5] = g[5] + y5; o[6] = g[6] + y6; o[7] = g[7] + y7;
        }
    };
    // ---- serial sweeps (wave 0; every lane redundantly, all state in registers) -----------------------------
    typedef double d2 __attribute__((ext_vector_type(2)));
    struct RecV { d2 v[RV / 2]; };
    struct RecM { d2 v[RM / 2]; };
    struct RecG { d2 v[4]; };
    // The records are wave-uniform; an opaque VGPR offset keeps the fetches as plain vector LDS loads (otherwise
    // the compiler scalarises every value through v_readlane_b32 with a wait per group).
    auto opaque = [](int off) { asm volatile("" : "+v"(off)); return off; };
    auto ldV = [&](int t, RecV &k) {
        const d2 *q = reinterpret_cast<const d2 *>(__builtin_assume_aligned(L.recV + opaque(RV * t), 16));
#pragma unroll
        for (int i = 0; i < RV / 2; ++i) k.v[i] = q[i]; };
    auto ldM = [&](int t, RecM &k) {
        const d2 *q = reinterpret_cast<const d2 *>(__builtin_assume_aligned(L.recM + opaque(RM * t), 16));
#pragma unroll
        for (int i = 0; i < RM / 2; ++i) k.v[i] = q[i]; };
    auto ldG = [&](int t, RecG &k) {
        const d2 *q = reinterpret_cast<const d2 *>(__builtin_assume_aligned(L.gh + opaque(8 * t), 16));
#pragma unroll
        for (int i = 0; i < 4; ++i) k.v[i] = q[i]; };
#define RVAL(k, i) ((k).v[(i) >> 1][(i) & 1])
    // one backward vector step: p <- ghat_x - W ghat_v ; stores kk = -Minv ghat_v
    auto bwd_step = [&](int t, const RecV &k, const RecG &gg, double (&p)[5], int lane_) {
        const double g0 = RVAL(gg, 0), g1 = RVAL(gg, 1), g2 = RVAL(gg, 2), g3 = RVAL(gg, 3), g4 = RVAL(gg, 4), g5 = RVAL(gg, 5), g6 = RVAL(gg, 6), g7 = RVAL(gg, 7);
        double gv0 = g5 + RVAL(k, 9) * p[0] + RVAL(k, 11) * p[1] + RVAL(k, 13) * p[2] + p[3];
        double gv1 = g6 + RVAL(k, 10) * p[0] + RVAL(k, 12) * p[1] + RVAL(k, 14) * p[2] + p[4];
        double gv2 = g7;
        double gx0 = g0 + RVAL(k, 0) * p[0] + RVAL(k, 3) * p[1] + RVAL(k, 6) * p[2];
        double gx1 = g1 + RVAL(k, 1) * p[0] + RVAL(k, 4) * p[1] + RVAL(k, 7) * p[2];
        double gx2 = g2 + RVAL(k, 2) * p[0] + RVAL(k, 5) * p[1] + RVAL(k, 8) * p[2];
        p[0] = gx0 - (RVAL(k, 15) * gv0 + RVAL(k, 16) * gv1 + RVAL(k, 17) * gv2);
        p[1] = gx1 - (RVAL(k, 18) * gv0 + RVAL(k, 19) * gv1 + RVAL(k, 20) * gv2);
        p[2] = gx2 - (RVAL(k, 21) * gv0 + RVAL(k, 22) * gv1 + RVAL(k, 23) * gv2);
        p[3] = g3 - (RVAL(k, 24) * gv0 + RVAL(k, 25) * gv1 + RVAL(k, 26) * gv2);
        p[4] = g4 - (RVAL(k, 27) * gv0 + RVAL(k, 28) * gv1 + RVAL(k, 29) * gv2);
        if (lane_ == 0) {
            double *o = &L.recV[RV * t + 36];
            o[0] = -(RVAL(k, 30) * gv0 + RVAL(k, 31) * gv1 + RVAL(k, 32) * gv2);
            o[1] = -(RVAL(k, 31) * gv0 + RVAL(k, 33) * gv1 + RVAL(k, 34) * gv2);
            o[2] = -(RVAL(k, 32) * gv0 + RVAL(k, 34) * gv1 + RVAL(k, 35) * gv2);
        }
    };
    auto bwd_all = [&](int lane_) {
        double p[5] = {0, 0, 0, 0, 0};
        RecV ka, kb; RecG ga, gb;
        ldV(T - 1, ka); ldG(T - 1, ga);
        for (int t = T - 1; t >= 0; t -= 2) {
            if (t >= 1) { ldV(t - 1, kb); ldG(t - 1, gb); }
            bwd_step(t, ka, ga, p, lane_);
            if (t >= 1) {
                if (t >= 2) { ldV(t - 2, ka); ldG(t - 2, ga); }
                bwd_step(t - 1, kb, gb, p, lane_);
            }
        }
    };
    // one forward step: dv = kk - W' dx ; dx+ = F [dx; dv]
    auto fwd_step = [&](int t, const RecV &k, double (&dx)[5], int lane_) {
        double v0 = RVAL(k, 36) - (RVAL(k, 15) * dx[0] + RVAL(k, 18) * dx[1] + RVAL(k, 21) * dx[2] + RVAL(k, 24) * dx[3] + RVAL(k, 27) * dx[4]);
        double v1 = RVAL(k, 37) - (RVAL(k, 16) * dx[0] + RVAL(k, 19) * dx[1] + RVAL(k, 22) * dx[2] + RVAL(k, 25) * dx[3] + RVAL(k, 28) * dx[4]);
        double v2 = RVAL(k, 38) - (RVAL(k, 17) * dx[0] + RVAL(k, 20) * dx[1] + RVAL(k, 23) * dx[2] + RVAL(k, 26) * dx[3] + RVAL(k, 29) * dx[4]);
        if (lane_ == 0) {
            double *y = &L.dy[8 * t];
            y[0] = dx[0]; y[1] = dx[1]; y[2] = dx[2]; y[3] = dx[3]; y[4] = dx[4]; y[5] = v0; y[6] = v1; y[7] = v2;
        }
        double n0 = RVAL(k, 0) * dx[0] + RVAL(k, 1) * dx[1] + RVAL(k, 2) * dx[2] + RVAL(k, 9) * v0 + RVAL(k, 10) * v1;
        double n1 = RVAL(k, 3) * dx[0] + RVAL(k, 4) * dx[1] + RVAL(k, 5) * dx[2] + RVAL(k, 11) * v0 + RVAL(k, 12) * v1;
        double n2 = RVAL(k, 6) * dx[0] + RVAL(k, 7) * dx[1] + RVAL(k, 8) * dx[2] + RVAL(k, 13) * v0 + RVAL(k, 14) * v1;
        dx[0] = n0; dx[1] = n1; dx[2] = n2; dx[3] = v0; dx[4] = v1;
    };
    auto fwd_all = [&](int lane_) {
        double dx[5] = {0, 0, 0, 0, 0};
        RecV ka, kb;
        ldV(0, ka);
        for (int t = 0; t < T; t += 2) {
            if (t + 1 < T) ldV(t + 1, kb);
            fwd_step(t, ka, dx, lane_);
            if (t + 1 < T) {
                if (t + 2 < T) ldV(t + 2, ka);
                fwd_step(t + 1, kb, dx, lane_);
            }
        }
        if (lane_ == 0) { L.pv[0] = dx[0]; L.pv[1] = dx[1]; L.pv[2] = dx[2]; }
    };
    // one Riccati matrix step fused with the predictor's backward vector step
    auto mat_step = [&](int t, const RecM &k, const RecG &gg, double (&Pm)[5][5], double (&p)[5], bool &fail_, int lane_) {
        double M[8][8];
        {
            int o = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = r; q < 8; ++q) { M[r][q] = RVAL(k, o); ++o; }
        }
        const double B0 = RVAL(k, 36), B1 = RVAL(k, 37), B2 = RVAL(k, 38), B3 = RVAL(k, 39), B4 = RVAL(k, 40), B5 = RVAL(k, 41);
        const double a13 = RVAL(k, 42), a23 = RVAL(k, 43);
        const double Bm[3][2] = { { B0, B1 }, { B2, B3 }, { B4, B5 } };
        // A = [[1,0,a13],[0,1,a23],[0,0,1]] in all three motion models (rda_solver.py:955,971,987)
        double T1c[3], T2[3][2], T3[2][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T1c[r] = Pm[r][0] * a13 + Pm[r][1] * a23 + Pm[r][2];                 // (Pss A)[:,2]
#pragma unroll
            for (int q = 0; q < 2; ++q) T2[r][q] = Pm[r][0] * Bm[0][q] + Pm[r][1] * Bm[1][q] + Pm[r][2] * Bm[2][q] + Pm[r][3 + q];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) T3[r][q] = Pm[3 + r][0] * Bm[0][q] + Pm[3 + r][1] * Bm[1][q] + Pm[3 + r][2] * Bm[2][q] + Pm[3 + r][3 + q];
        // M += F' P F (upper triangle)
        M[0][0] += Pm[0][0]; M[0][1] += Pm[0][1]; M[0][2] += T1c[0];
        M[1][1] += Pm[1][1]; M[1][2] += T1c[1];
        M[2][2] += a13 * T1c[0] + a23 * T1c[1] + T1c[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            M[0][5 + q] += T2[0][q]; M[1][5 + q] += T2[1][q];
            M[2][5 + q] += a13 * T2[0][q] + a23 * T2[1][q] + T2[2][q];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = r; q < 2; ++q) M[5 + r][5 + q] += Bm[0][r] * T2[0][q] + Bm[1][r] * T2[1][q] + Bm[2][r] * T2[2][q] + T3[r][q];
        // inverse of Mvv (rows/cols 5..7) by the adjugate: one division on the critical path
        double m00 = M[5][5], m01 = M[5][6], m02 = M[5][7], m11 = M[6][6], m12 = M[6][7], m22 = M[7][7];
        double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
        double c11 = m00 * m22 - m02 * m02, c12 = m01 * m02 - m00 * m12, c22 = m00 * m11 - m01 * m01;
        double det = m00 * c00 + m01 * c01 + m02 * c02;
        if (!(m00 > 0) || !(c22 > 0) || !(det > 0)) fail_ = true;
        double id = 1.0 / det;
        double n00 = c00 * id, n01 = c01 * id, n02 = c02 * id, n11 = c11 * id, n12 = c12 * id, n22 = c22 * id;
        double W[5][3];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            double x0 = M[r][5], x1 = M[r][6], x2 = M[r][7];
            W[r][0] = x0 * n00 + x1 * n01 + x2 * n02;
            W[r][1] = x0 * n01 + x1 * n11 + x2 * n12;
            W[r][2] = x0 * n02 + x1 * n12 + x2 * n22;
        }
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int q = r; q < 5; ++q) {
                double v = M[r][q] - (W[r][0] * M[q][5] + W[r][1] * M[q][6] + W[r][2] * M[q][7]);
                Pm[r][q] = v; Pm[q][r] = v;
            }
        const double g0 = RVAL(gg, 0), g1 = RVAL(gg, 1), g2 = RVAL(gg, 2), g3 = RVAL(gg, 3), g4 = RVAL(gg, 4), g5 = RVAL(gg, 5), g6 = RVAL(gg, 6), g7 = RVAL(gg, 7);
        double gv0 = g5 + B0 * p[0] + B2 * p[1] + B4 * p[2] + p[3];
        double gv1 = g6 + B1 * p[0] + B3 * p[1] + B5 * p[2] + p[4];
        double gv2 = g7;
        double gx0 = g0 + p[0], gx1 = g1 + p[1], gx2 = g2 + a13 * p[0] + a23 * p[1] + p[2];
        p[0] = gx0 - (W[0][0] * gv0 + W[0][1] * gv1 + W[0][2] * gv2);
        p[1] = gx1 - (W[1][0] * gv0 + W[1][1] * gv1 + W[1][2] * gv2);
        p[2] = gx2 - (W[2][0] * gv0 + W[2][1] * gv1 + W[2][2] * gv2);
        p[3] = g3 - (W[3][0] * gv0 + W[3][1] * gv1 + W[3][2] * gv2);
        p[4] = g4 - (W[4][0] * gv0 + W[4][1] * gv1 + W[4][2] * gv2);
        if (lane_ == 0) {
            double *o = &L.recV[RV * t + 15];
#pragma unroll
            for (int r = 0; r < 5; ++r) { o[3 * r] = W[r][0]; o[3 * r + 1] = W[r][1]; o[3 * r + 2] = W[r][2]; }
            o[15] = n00; o[16] = n01; o[17] = n02; o[18] = n11; o[19] = n12; o[20] = n22;
            o[21] = -(n00 * gv0 + n01 * gv1 + n02 * gv2);
            o[22] = -(n01 * gv0 + n11 * gv1 + n12 * gv2);
            o[23] = -(n02 * gv0 + n12 * gv1 + n22 * gv2);
        }
    };

    int status = 1, it;
    long long tprev = clock64();
    auto mark = [&](int k) { if (a.prof && tid == 0) { long long now = clock64(); a.prof[k] += now - tprev; tprev = now; } };
    mark(0);
    for (it = 0; it < 100; ++it) {
        // ---- (1) hinge sums per stage: (stage, chunk) partials, then one thread per (stage, quantity) --
        {
            double sxx = 0, sxy = 0, syy = 0, sx = 0, sy = 0, s1 = 0, ix = 0, iy = 0, i1 = 0;
            if (ract) {
                const double px = L.s[rt + 1], py = L.s[(T + 1) + rt + 1], dd = L.d[rt];
                auto term = [&](double ax, double ay, double cb) {
                    double Im = ax * px + ay * py - cb - dd;
                    if (!c.accelerated || Im < 0) {
                        sxx += ax * ax; sxy += ax * ay; syy += ay * ay; sx += ax; sy += ay; s1 += 1.0;
                        ix += Im * ax; iy += Im * ay; i1 += Im;
                    }
                };
                const int Nl = a.Nloc;
                for (int r = 0; r < a.P; ++r) {
                    const size_t o = r * a.chunk + (size_t)rt * Nl;
                    const double *pax = a.ax + o, *pay = a.ay + o, *pb = a.blam + o, *pe = a.ee + o;
                    int n = rc_;
                    for (; n + 3 * nch < Nl; n += 4 * nch) {      // four independent loads in flight per array
                        double a0 = pax[n], a1 = pax[n + nch], a2 = pax[n + 2 * nch], a3 = pax[n + 3 * nch];
                        double b0 = pay[n], b1 = pay[n + nch], b2 = pay[n + 2 * nch], b3 = pay[n + 3 * nch];
                        double c0 = pb[n] + pe[n], c1 = pb[n + nch] + pe[n + nch], c2 = pb[n + 2 * nch] + pe[n + 2 * nch], c3 = pb[n + 3 * nch] + pe[n + 3 * nch];
                        term(a0, b0, c0); term(a1, b1, c1); term(a2, b2, c2); term(a3, b3, c3);
                    }
                    for (; n < Nl; n += nch) term(pax[n], pay[n], pb[n] + pe[n]);
                }
            }
            double *pp = &L.part[tid * 9];
            pp[0] = sxx; pp[1] = sxy; pp[2] = syy; pp[3] = sx; pp[4] = sy; pp[5] = s1; pp[6] = ix; pp[7] = iy; pp[8] = i1;
            __syncthreads();
            for (int i = tid; i < 9 * T; i += NT) {
                int t = i / 9, k = i % 9;
                double acc = 0;
                for (int ch = 0; ch < nch; ++ch) acc += L.part[(t * nch + ch) * 9 + k];
                L.hs[i] = acc;
            }
        }
        __syncthreads();
        mark(1);
        // ---- (2) per-stage derivatives wrt w = (s_next, d); barrier weights; C'lam; residuals --------
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
            double *Hw = &L.Hw[16 * t], *gw = &L.gw[4 * t];
            double hsd0 = -c.ro1 * h[3], hsd1 = -c.ro1 * h[4];
            Hw[0] = Hs00; Hw[1] = Hs01; Hw[2] = 0; Hw[3] = hsd0;
            Hw[4] = Hs01; Hw[5] = Hs11; Hw[6] = 0; Hw[7] = hsd1;
            Hw[8] = 0; Hw[9] = 0; Hw[10] = Hs22; Hw[11] = 0;
            Hw[12] = hsd0; Hw[13] = hsd1; Hw[14] = 0; Hw[15] = c.ro1 * h[5];
            gw[0] = gs[0]; gw[1] = gs[1]; gw[2] = gs[2]; gw[3] = -c.ro1 * h[8] - c.slack_gain;
            // inequality rows of this stage
            double up0 = t ? L.u[t - 1] : 0, up1 = t ? L.u[T + t - 1] : 0;
            double lam[NC], dg[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                int i = t * NC + k;
                bool on = con_on(t, k);
                lam[k] = L.cl[i]; dg[k] = on ? L.cl[i] / L.cw[i] : 0.0;
                L.rp[i] = on ? con_val(k, L.u[t], L.u[T + t], up0, up1, L.d[t]) + L.cw[i] - con_rhs(c, k) : 0.0;
                L.rc[i] = L.cl[i] * L.cw[i];                              // affine (predictor) target
            }
            double *bw = &L.bw[5 * t];
            bw[0] = dg[0] + dg[1]; bw[1] = dg[2] + dg[3]; bw[2] = dg[4] + dg[5]; bw[3] = dg[6] + dg[7]; bw[4] = dg[8] + dg[9];
            double y3, y4, y5, y6, y7; con_T(lam, t, y3, y4, y5, y6, y7);
            double *rm = &L.recM[RM * t];
            for (int i = 0; i < 6; ++i) rm[36 + i] = L.Bk[6 * t + i];
            rm[42] = L.Ak[9 * t + 2]; rm[43] = L.Ak[9 * t + 5];
            double *cy = &L.cy[5 * t];
            cy[0] = y3; cy[1] = y4; cy[2] = y5 + 2 * c.wu * (L.u[t] - vref) + c.eps_u * L.u[t]; cy[3] = y6 + c.eps_u * L.u[T + t]; cy[4] = y7;
        }
        __syncthreads();
        // ---- (3) stage Hessian bases  J' Hw J + direct + barrier  (all threads), predictor rhs ----------
        // stage gradient gst = J' gw + direct terms + C' lam   (one thread per entry)
        for (int i = tid; i < 8 * T; i += NT) {
            int t = i >> 3, j = i & 7;
            const double *J = &L.Jm[32 * t], *gw = &L.gw[4 * t];
            double v = J[j] * gw[0] + J[8 + j] * gw[1] + J[16 + j] * gw[2] + J[24 + j] * gw[3];
            if (j >= 3) v += L.cy[5 * t + j - 3];
            L.gst[i] = v;
        }
        // upper triangle of the stage Hessian base, written straight into the matrix-sweep records
        for (int i = tid; i < 36 * T; i += NT) {
            int t = i / 36, o = i % 36;
            int r = 0, rem = o;
            while (rem >= 8 - r) { rem -= 8 - r; ++r; }
            int q = r + rem;
            const double *J = &L.Jm[32 * t], *Hw = &L.Hw[16 * t], *bw = &L.bw[5 * t];
            double m = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                double acc = Hw[4 * x] * J[q] + Hw[4 * x + 1] * J[8 + q] + Hw[4 * x + 2] * J[16 + q] + Hw[4 * x + 3] * J[24 + q];
                m += J[8 * x + r] * acc;
            }
            if (r == q) {
                if (r == 5) m += 2 * c.wu + c.eps_u + bw[0] + bw[3];
                else if (r == 6) m += c.eps_u + bw[1] + bw[4];
                else if (r == 7) m += bw[2];
                else if (r == 3) m += bw[3];
                else if (r == 4) m += bw[4];
            } else if (r == 3 && q == 5) m -= bw[3];
            else if (r == 4 && q == 6) m -= bw[4];
            L.recM[RM * t + o] = m;
        }
        __syncthreads();
        build_gh();
        __syncthreads();
        mark(2);
        // ---- (4) wave 0: Riccati matrix recursion fused with the predictor's backward vector sweep;
        //          wave 1: adjoint sweep for the reduced gradient -----------------------------------
        bool fail = false;
        if (wave == 0) {
            double p[5] = {0, 0, 0, 0, 0};
            double Pm[5][5];
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int q = 0; q < 5; ++q) Pm[r][q] = 0;
            RecM ka, kb; RecG ga, gb;
            ldM(T - 1, ka); ldG(T - 1, ga);
            for (int t = T - 1; t >= 0; t -= 2) {
                if (t >= 1) { ldM(t - 1, kb); ldG(t - 1, gb); }
                mat_step(t, ka, ga, Pm, p, fail, lane);
                if (t >= 1) {
                    if (t >= 2) { ldM(t - 2, ka); ldG(t - 2, ga); }
                    mat_step(t - 1, kb, gb, Pm, p, fail, lane);
                }
            }
            wsync();
        } else if (wave == 1) {
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
            for (int t = T - 1; t >= 0; --t) {
                const double *A = &L.Ak[9 * t], *B = &L.Bk[6 * t], *g = &L.gst[8 * t];
                double g0 = g[0] + A[0] * p0 + A[3] * p1 + A[6] * p2;
                double g1 = g[1] + A[1] * p0 + A[4] * p1 + A[7] * p2;
                double g2 = g[2] + A[2] * p0 + A[5] * p1 + A[8] * p2;
                double v0 = g[5] + B[0] * p0 + B[2] * p1 + B[4] * p2 + p3;
                double v1 = g[6] + B[1] * p0 + B[3] * p1 + B[5] * p2 + p4;
                if (lane == 0) { L.gad[3 * t] = v0; L.gad[3 * t + 1] = v1; L.gad[3 * t + 2] = g[7]; }
                p0 = g0; p1 = g1; p2 = g2; p3 = g[3]; p4 = g[4];
            }
        }
        __syncthreads();
        mark(4);
        double rdn = 0, gn = 0, rpn = 0, mu = 0;
        for (int i = tid; i < 3 * T; i += NT) { double v = fabs(L.gad[i]); if (v > rdn) rdn = v; }
        for (int i = tid; i < 4 * T; i += NT) { double v = fabs(L.gw[i]); if (v > gn) gn = v; }
        for (int i = tid; i < NC * T; i += NT) { double v = fabs(L.rp[i]); if (v > rpn) rpn = v; mu += L.cl[i] * L.cw[i]; }
        rdn = block_reduce(rdn, L.red, tid, true);
        gn = block_reduce(gn, L.red, tid, true);
        rpn = block_reduce(rpn, L.red, tid, true);
        mu = block_reduce(mu, L.red, tid, false) / mcnt;
        double sc = 1 + gn;
        if (rdn <= 1e-9 * sc && rpn <= 1e-10 && mu <= 1e-11 * sc) { status = 0; break; }
        if (__syncthreads_or(fail ? 1 : 0)) { status = 2; break; }
        mark(5);

        double sigma = 0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                // corrector right-hand side, vector-only backward sweep with the stored factors
                for (int i = tid; i < NC * T; i += NT) L.rc[i] = L.cl[i] * L.cw[i] + L.dl[i] * L.dw[i] - sigma * mu;
                __syncthreads();
                build_gh();
                __syncthreads();
                if (wave == 0) { bwd_all(lane); wsync(); }
            }
            mark(6);
            if (wave == 0) fwd_all(lane);
            __syncthreads();
            mark(7);
            // ---- slack / multiplier steps, step length ------------------------------------------------
            double al = 1.0;
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
                // centering parameter from the predictor step length, floored (see the oracle for why)
                double q = 1 - al, fl = al >= 0.95 ? 0.003 : 0.03; sigma = q * q * q; if (sigma < fl) sigma = fl;
            } else {
                for (int i = tid; i < NC * T; i += NT) { L.cw[i] += al * L.dw[i]; L.cl[i] += al * L.dl[i]; }
                if (tid < T) {
                    int t = tid; const double *y = &L.dy[8 * t];
                    L.u[t] += al * y[5]; L.u[T + t] += al * y[6]; L.d[t] += al * y[7];
                    if (t >= 1) for (int r = 0; r < 3; ++r) L.s[r * (T + 1) + t] += al * y[r];
                }
                if (tid < 3) L.s[tid * (T + 1) + T] += al * L.pv[tid];
                __syncthreads();
            }
            mark(8);
        }
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
