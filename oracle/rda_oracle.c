/*
 * ORACLE - test infrastructure only (see rda_oracle.h).  Plain C99, fp64.
 *
 * Restates, function by function, /root/reference/RDA_planner/rda_solver.py:
 *   orc_step            <- iterative_solve :573-610 + rda_solver :612-637
 *   stage_obstacles     <- assign_obstacle_parameter :483-526   (quirk Q3: pad = duplicate last)
 *   linearise           <- assign_state_parameter :436-460, models :949-994 (quirk Q1)
 *   orc_lammuz_one      <- LamMuZ_cost_cons :389-421, Hm_LamMu/Im_LamMu :874-909, cones :1034-1050
 *   dual/residual part  <- solve_parallel :781-793, assign_combine_parameter_lamobs :529-542,
 *                          update_xi :668-690, update_zeta :639-666
 *   orc_su_solve        <- construct_su_prob :216-231, nav_cost_cons :313-328,
 *                          update_su_cost_cons :330-387, Im_su/Hm_su :831-872,
 *                          dynamics/bounds :911-947, C0/C1 cost :1011-1032
 * The two convex programs the reference gives to CVXPY/ECOS are solved here by
 *   (i)  exhaustive enumeration of basic supports (LamMuZ; see oracle/lammuz_np.py for the
 *        derivation and the explicit tie-break T1-T3), and
 *  (ii)  a dense primal-dual interior point method on the control-condensed problem (su); inside an MPC step the
 *        su-problems of ADMM iterations >= 1 start from the multipliers of the previous one (su_solve_impl, the same
 *        start rule as the kernel's, so both walk the same iteration path; the solution itself is unique).
 */
#include "rda_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EMAX 16
#define RMAX 16
#define SIGN_TOL 1e-12

static int g_threads = 1;
static double g_su_warm_wfl = 1e-3, g_su_warm_mu0 = 1e-3; static int g_su_warm_cap = 30, g_su_warm_first = 1;   /* su warm start (orc_set_su_warm(0,0,0): cold) */
void orc_set_su_warm(double wfl, double mu0, int cap) { g_su_warm_wfl = wfl; g_su_warm_mu0 = mu0; g_su_warm_cap = cap; }
/* end game of a WARM-started su solve: floor of the fraction to the boundary and of the centering parameter (cold: 0.995, 1e-3) */
static double g_su_warm_tau = 0.9999, g_su_warm_sig = 1e-5;
void orc_set_su_warm_endgame(double tau, double sig) { g_su_warm_tau = tau; g_su_warm_sig = sig; }
/* relative margin by which the start of a WARM attempt is pulled inside the control / distance boxes (cold: 0.01) */
static double g_su_warm_clip = 0.01;
void orc_set_su_warm_clip(double m) { g_su_warm_clip = m; }
/* start used while the su-solves are easy (the last one took <= max iterations): wfl, mu0, clip, tau, sigma; max = 0 disables */
static double g_su_easy[5] = {1e-12, 1e-12, 1e-12, 0.999999, 1e-7}; static int g_su_easy_max = 2;   /* = rda_opts_init (csrc/rda_hip.hip) */
static double cur_warm_tau = 0.9999, cur_warm_sig = 1e-5, cur_warm_clip = 0.01;     /* what the warm attempt of the solve in progress uses */
void orc_set_su_easy(double wfl, double mu0, double clip, double tau, double sig, int max) { g_su_easy[0] = wfl; g_su_easy[1] = mu0; g_su_easy[2] = clip; g_su_easy[3] = tau; g_su_easy[4] = sig; g_su_easy_max = max; }
static double g_su_tol[3] = {1e-9, 1e-10, 1e-11};   /* interior-point stop of the su-problem: rd, rp, mu */
void orc_set_su_tol(double rd, double rp, double mu) { if (rd > 0 && rp > 0 && mu > 0) { g_su_tol[0] = rd; g_su_tol[1] = rp; g_su_tol[2] = mu; } }
/* mirror of rda_opts::su_tol_early: the su-problems of the ADMM iterations BEFORE the last one of a step stop at these (0: same as su_tol) */
static double g_su_tol_early[3] = {0, 0, 0};
void orc_set_su_tol_early(double rd, double rp, double mu) { g_su_tol_early[0] = rd; g_su_tol_early[1] = rp; g_su_tol_early[2] = mu; }
static int g_lmz_mode = 0;   /* 0: support enumeration + tie-breaks T1-T3, 1: interior point (oracle/lmz_ipm.c) */
void orc_set_lmz_mode(int mode) { g_lmz_mode = mode ? 1 : 0; }
static int g_centre = 1;     /* tie-break T1: central separating normal in the slack regime (orc_set_centre(0): max clearance) */
void orc_set_centre(int on) { g_centre = on; }
/* debug aids of the su interior point (not used by any test's comparison): orc_set_su_dump(path) records every su-problem orc_admm_su
 * solves (NULL / "": off), orc_set_su_trace(1) prints one line per interior-point iteration to stderr */
static FILE *g_su_dump = NULL; static int g_su_trace = 0;
void orc_set_su_dump(const char *path) { if (g_su_dump) fclose(g_su_dump); g_su_dump = (path && path[0]) ? fopen(path, "wb") : NULL; }
void orc_set_su_trace(int on) { g_su_trace = on; }
static double g_su_hard_wfl = 1.0, g_su_hard_mu0 = 1e-3;   /* mirror of rda_opts::su_hard_warm: start of the warm attempts of a step that follows an UNCONVERGED step (0 = off) */
void orc_set_su_hard_warm(double wfl, double mu0) { g_su_hard_wfl = wfl; g_su_hard_mu0 = mu0; }
/* mirror of rda_opts::su_cold_from / su_cold_probe: a warm attempt that follows a solve with more than `from` interior-point iterations is
 * skipped (the solve starts cold), except every `probe`-th such solve (0 = never) */
static int g_su_cold_from = 7, g_su_cold_probe = 8;
void orc_set_su_cold_from(int from, int probe) { g_su_cold_from = from; g_su_cold_probe = probe > 0 ? probe : 1; }
#define SU_HARD_RD0 1e-2      /* = csrc/su_device.h */
#define SU_HARD_DMU 1.0       /* = csrc/su_device.h */
static int g_su_first_attempt = 0;               /* mirror of rda_opts::su_first_attempt (test switch): 1 = start with the last-resort attempt */
void orc_set_su_first_attempt(int a) { g_su_first_attempt = a ? 1 : 0; }
static int g_su_accept = 1;                      /* su_solve_impl: the near-converged iterate kept as a safety net (see there) */
void orc_set_su_accept(int on) { g_su_accept = on; }      /* 2: test switch - ALWAYS return the remembered iterate (= csrc/su_device.h Args::accept) */
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }

struct orc_handle {
    orc_cfg c;
    double *G, *h;                 /* R*2, R */
    double *lam, *mu, *z, *xi, *zeta, *dis;   /* dual state, persists across steps (Q5,Q6) */
    double *a_lam, *b_lam;         /* obsA_lam / obsb_lam (rda_solver.py:172-173), stale at step start (Q4) */
    double *A, *b; int *cone;      /* staged obstacles [N][T+1][E][2], [N][T+1][E] */
    double *s, *u;                 /* current nominal (para_s, para_u) */
    int obstacle_num;
    double *resp;                  /* [N*T][2] residual partials of the last LamMuZ pass */
    double *ref, ref_speed;        /* step inputs */
    int stop, iters, su_status, ipm_total, lmz_fail; double resi_dual, resi_pri;
    double *su_lam_keep;          /* inequality multipliers of the last converged su-solve (10T - 4 rows) */
    int su_last;                  /* interior-point iterations of the last su-solve (99: none / not converged): picks the next warm start */
    int prev_unconv;              /* the previous step ended with a residual above iter_threshold (all iter_num iterations, no early stop) */
    int su_probe;                 /* consecutive su-solves in the hard regime (su_cold_from) */
    int su_hardlike;              /* the last su-solve started far from its solution: relative dual residual of its first iterate > SU_HARD_RD0 */
    int ipm_hist[32];             /* debug: interior-point iterations of the su-solve of every ADMM iteration of the last step (orc_get_su_ipm_hist) */
    int P, rank, Nloc, have_gath; size_t chunk; double *gath;      /* obstacle sharding */
};

/* ------------------------------------------------------------------------------------------ */
/* 2-D trust-region sub-problem: min 1/2 x'Qx + c'x,  ||x||<=1 (disc) or ||x||==1.            */
/* returns number of solutions written to x (2 doubles each): 0, 1, or 2 in the hard case          */
static int trs2(double q11, double q12, double q22, double c0, double c1, int disc, double *x)
{
    double mean = 0.5 * (q11 + q22), dif = 0.5 * (q11 - q22);
    double rad = hypot(dif, q12);
    double l1 = mean - rad, l2 = mean + rad;
    double v2x, v2y;
    if (dif >= 0) { v2x = dif + rad; v2y = q12; } else { v2x = q12; v2y = rad - dif; }
    double nv = hypot(v2x, v2y);
    if (nv > 0) { v2x /= nv; v2y /= nv; } else { v2x = 1; v2y = 0; }
    double v1x = -v2y, v1y = v2x;
    double h1 = v1x * c0 + v1y * c1, h2 = v2x * c0 + v2y * c1;
    double cn = hypot(h1, h2);
    double scale = fabs(l2) > 1e-300 ? fabs(l2) : 1e-300;
    if (disc && l1 > 1e-13 * scale) {
        double y1 = -h1 / l1, y2 = -h2 / l2;
        if (y1 * y1 + y2 * y2 <= 1.0) { x[0] = y1 * v1x + y2 * v2x; x[1] = y1 * v1y + y2 * v2y; return 1; }
    }
    if (cn == 0.0) {
        if (disc) return 0;
        x[0] = v1x; x[1] = v1y; return 1;
    }
    if (fabs(h1) <= 1e-9 * cn) {
        /* hard case: c orthogonal to the low-curvature eigenvector -> two minimisers */
        double gap = l2 - l1;
        if (gap > 0) {
            double y2 = -h2 / gap;
            if (fabs(y2) < 1.0) {
                double sq = sqrt(1.0 - y2 * y2);
                x[0] = y2 * v2x + sq * v1x; x[1] = y2 * v2y + sq * v1y;
                x[2] = y2 * v2x - sq * v1x; x[3] = y2 * v2y - sq * v1y;
                return 2;
            }
        }
        h1 = 0.0;
    }
    double lo = cn - l2, lo2 = fabs(h1) - l1;
    if (lo2 > lo) lo = lo2;
    if (disc && lo < 0) lo = 0;
    double tau = lo;
    for (int it = 0; it < 20; ++it) {
        double s1 = l1 + tau, s2 = l2 + tau;
        if (s1 <= 0 || s2 <= 0) { tau = (-l1 > -l2 ? -l1 : -l2) + 1e-300; s1 = l1 + tau; s2 = l2 + tau; }
        double a1 = h1 != 0 ? h1 / s1 : 0.0, a2 = h2 != 0 ? h2 / s2 : 0.0;
        double phi = a1 * a1 + a2 * a2;
        if (!(phi > 0)) break;
        double dphi = -2.0 * (a1 * a1 / s1 + a2 * a2 / s2);
        double sq = sqrt(phi);
        double g = 1.0 / sq - 1.0, dg = -0.5 * dphi / (phi * sq);
        double step = g / dg;
        tau -= step;
        if (fabs(step) <= 4e-16 * (fabs(tau) > 1 ? fabs(tau) : 1)) break;
    }
    double s1 = l1 + tau, s2 = l2 + tau;
    double y1 = h1 != 0 ? -h1 / s1 : 0.0, y2 = h2 != 0 ? -h2 / s2 : 0.0;
    double xx = y1 * v1x + y2 * v2x, xy = y1 * v1y + y2 * v2y;
    double nx = hypot(xx, xy);
    if (nx > 0) { xx /= nx; xy /= nx; }
    x[0] = xx; x[1] = xy;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int E, R;
    const double *A, *b, *G, *h;
    double q[EMAX], M[EMAX][2];       /* q = A p - b ; M = A R (rows) */
    double xi[2], kappa0, ro2, delta;
} lmz_ctx;

typedef struct { int k; int j[2]; double P[2][2]; double r[2]; double rr; } mu_cand;

/* minimise over the mu-support coefficients for given (t, e):  returns m, H, gamma */
static void gamma_star(const lmz_ctx *c, const mu_cand *mc, double chi, double t, const double *e,
                       double *gam, double *m, double *H)
{
    if (mc->k == 0) { *m = t; H[0] = e[0]; H[1] = e[1]; return; }
    if (mc->k == 1) {
        const double *g = &c->G[2 * mc->j[0]]; double eta = c->h[mc->j[0]];
        double den = chi * eta * eta + c->ro2 * (g[0] * g[0] + g[1] * g[1]);
        double ga = (chi * eta * t - c->delta * eta - c->ro2 * (g[0] * e[0] + g[1] * e[1])) / den;
        gam[0] = ga; *m = t - eta * ga; H[0] = e[0] + ga * g[0]; H[1] = e[1] + ga * g[1];
        return;
    }
    double re = mc->r[0] * e[0] + mc->r[1] * e[1];
    double beta = (c->delta - chi * (t + re)) / (chi * mc->rr + c->ro2);
    H[0] = -beta * mc->r[0]; H[1] = -beta * mc->r[1];
    /* gamma = P^{-1}(H - e), P columns = G rows */
    double d0 = H[0] - e[0], d1 = H[1] - e[1];
    double det = mc->P[0][0] * mc->P[1][1] - mc->P[0][1] * mc->P[1][0];
    gam[0] = (mc->P[1][1] * d0 - mc->P[0][1] * d1) / det;
    gam[1] = (-mc->P[1][0] * d0 + mc->P[0][0] * d1) / det;
    *m = t + re + beta * mc->rr;
}

/* value / gradient of the candidate model in (t, e) after eliminating the mu-support coefficients */
static void model_g3(const lmz_ctx *c, const mu_cand *mc, double chi, double t, const double *e, double *g3, double *f)
{
    double gam[2] = {0, 0}, m, H[2];
    gamma_star(c, mc, chi, t, e, gam, &m, H);
    g3[0] = chi * m - c->delta; g3[1] = c->ro2 * H[0]; g3[2] = c->ro2 * H[1];
    if (f) *f = 0.5 * chi * m * m - c->delta * m + 0.5 * c->ro2 * (H[0] * H[0] + H[1] * H[1]);
}

/* Circle obstacle, candidate with 0 < ||a|| < 1 (lam_3 = -||a|| tight): damped Newton on
 * Phi(at) = model(t = at'ut + l0*||at|| + kappa0, e = at + xi), l0 = -radius, started on the steepest-descent
 * ray out of the kink at at = 0 (Phi is quadratic along a ray).  Returns 1 and at[2], or 0 when the
 * ||a|| in {0, 1} candidates cover the optimum.  Same steps as oracle/lammuz_np.py:circle_interior. */
static int circle_interior(const lmz_ctx *c, const mu_cand *mc, double chi, const double *ut, double l0, double *at)
{
    double z2[2] = {0, 0}, e[2], G0[3], g3[3], Hm[3][3], col[3];
    model_g3(c, mc, chi, 0.0, z2, G0, NULL);
    model_g3(c, mc, chi, 1.0, z2, col, NULL); for (int i = 0; i < 3; ++i) Hm[i][0] = col[i] - G0[i];
    e[0] = 1; e[1] = 0; model_g3(c, mc, chi, 0.0, e, col, NULL); for (int i = 0; i < 3; ++i) Hm[i][1] = col[i] - G0[i];
    e[0] = 0; e[1] = 1; model_g3(c, mc, chi, 0.0, e, col, NULL); for (int i = 0; i < 3; ++i) Hm[i][2] = col[i] - G0[i];
    for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) { double a = 0.5 * (Hm[i][j] + Hm[j][i]); Hm[i][j] = Hm[j][i] = a; }
    double nu = hypot(ut[0], ut[1]);
    model_g3(c, mc, chi, c->kappa0, c->xi, g3, NULL);
    double gk0 = g3[0] * ut[0] + g3[1], gk1 = g3[0] * ut[1] + g3[2], ck = g3[0] * l0, ng = hypot(gk0, gk1);
    if (!(ng > ck * (1.0 + 1e-12))) return 0;
    double v0 = -gk0 / ng, v1 = -gk1 / ng;
    double w[3] = { v0 * ut[0] + v1 * ut[1] + l0, v0, v1 }, curv = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) curv += w[i] * Hm[i][j] * w[j];
    double s0 = curv > (ng - ck) / 0.9 ? (ng - ck) / curv : 0.9;
    double x0 = s0 * v0, x1 = s0 * v1, f, s_;
    s_ = hypot(x0, x1); e[0] = x0 + c->xi[0]; e[1] = x1 + c->xi[1];
    model_g3(c, mc, chi, x0 * ut[0] + x1 * ut[1] + l0 * s_ + c->kappa0, e, g3, &f);
    int nclip = 0;
    for (int it = 0; it < 30; ++it) {
        s_ = hypot(x0, x1);
        double a0 = x0 / s_, a1 = x1 / s_;
        e[0] = x0 + c->xi[0]; e[1] = x1 + c->xi[1];
        model_g3(c, mc, chi, x0 * ut[0] + x1 * ut[1] + l0 * s_ + c->kappa0, e, g3, NULL);
        double J[3][2] = { { ut[0] + l0 * a0, ut[1] + l0 * a1 }, { 1, 0 }, { 0, 1 } };
        double gr0 = g3[0] * J[0][0] + g3[1], gr1 = g3[0] * J[0][1] + g3[2];
        double Hs[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
            double acc = 0;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) acc += J[i][a] * Hm[i][j] * J[j][b];
            Hs[a][b] = acc;
        }
        double kq = g3[0] * l0 / s_;
        Hs[0][0] += kq * (1 - a0 * a0); Hs[0][1] += kq * (-a0 * a1); Hs[1][0] += kq * (-a0 * a1); Hs[1][1] += kq * (1 - a1 * a1);
        double tr = 0.5 * (Hs[0][0] + Hs[1][1]), df = 0.5 * (Hs[0][0] - Hs[1][1]), rad = hypot(df, Hs[0][1]);
        double lmin = tr - rad, lmax = fabs(tr + rad) > 1e-300 ? fabs(tr + rad) : 1e-300;
        if (lmin < 1e-8 * lmax) { double sh = 1e-8 * lmax - lmin; Hs[0][0] += sh; Hs[1][1] += sh; }
        double det = Hs[0][0] * Hs[1][1] - Hs[0][1] * Hs[1][0];
        if (!(det > 0)) return 0;
        double d0 = -(Hs[1][1] * gr0 - Hs[0][1] * gr1) / det, d1 = -(-Hs[1][0] * gr0 + Hs[0][0] * gr1) / det;
        double gsc = 1.0 + fabs(g3[0]) * nu + hypot(g3[1], g3[2]);
        if (hypot(gr0, gr1) <= 1e-13 * gsc || hypot(d0, d1) <= 1e-15 * (s_ > 1 ? s_ : 1.0)) break;
        double al = 1.0, xn0 = x0, xn1 = x1, fn = f; int ok = 0, clipped = 0;
        for (int bt = 0; bt < 30; ++bt) {
            xn0 = x0 + al * d0; xn1 = x1 + al * d1;
            double sn = hypot(xn0, xn1);
            if (sn > 1e-12 && sn < 1.0) {
                double g3n[3];
                e[0] = xn0 + c->xi[0]; e[1] = xn1 + c->xi[1];
                model_g3(c, mc, chi, xn0 * ut[0] + xn1 * ut[1] + l0 * sn + c->kappa0, e, g3n, &fn);
                if (fn <= f + 1e-4 * al * (gr0 * d0 + gr1 * d1) + 1e-13 * fabs(f)) { ok = 1; break; }
            } else if (sn >= 1.0) clipped = 1;
            al *= 0.5;
        }
        if (clipped && ++nclip >= 3) return 0;
        if (!ok) break;
        x0 = xn0; x1 = xn1; f = fn;
    }
    s_ = hypot(x0, x1);
    e[0] = x0 + c->xi[0]; e[1] = x1 + c->xi[1];
    model_g3(c, mc, chi, x0 * ut[0] + x1 * ut[1] + l0 * s_ + c->kappa0, e, g3, NULL);
    double gr0 = g3[0] * (ut[0] + l0 * x0 / s_) + g3[1], gr1 = g3[0] * (ut[1] + l0 * x1 / s_) + g3[2];
    if (hypot(gr0, gr1) > 1e-9 * (1.0 + fabs(g3[0]) * nu + hypot(g3[1], g3[2]))) return 0;
    at[0] = x0; at[1] = x1;
    return 1;
}


/* ------------------------------------------------------------------------------------------ */
/* Tie-break T1 in the slack regime: the unit normal in the middle of the arc of all separating   */
/* directions around the max-clearance normal a*, with the duals that support it (H = 0).        */
/* Same steps as oracle/lammuz_np.py:central_normal.                                              */
typedef struct { double x, y; int i1, i2; } pvert;
static int polygon_vertices(int E, const double *A, const double *b, pvert *out)
{
    int n = 0;
    for (int i1 = 0; i1 < E; ++i1) for (int i2 = i1 + 1; i2 < E; ++i2) {
        const double a00 = A[2 * i1], a01 = A[2 * i1 + 1], a10 = A[2 * i2], a11 = A[2 * i2 + 1];
        const double det = a00 * a11 - a01 * a10;
        if (det == 0 || !(det * det > 1e-24 * (a00 * a00 + a01 * a01) * (a10 * a10 + a11 * a11))) continue;
        const double wx = b[i1] * a11 - a01 * b[i2], wy = a00 * b[i2] - b[i1] * a10, sg = det > 0 ? 1.0 : -1.0, ad = fabs(det);
        int ok = 1;
        for (int k = 0; k < E; ++k) {
            const double viol = sg * (A[2 * k] * wx + A[2 * k + 1] * wy - b[k] * det);
            if (viol > 1e-9 * (ad + fabs(b[k]) * ad + fabs(A[2 * k] * wx) + fabs(A[2 * k + 1] * wy))) ok = 0;
        }
        if (ok) { out[n].x = wx / det; out[n].y = wy / det; out[n].i1 = i1; out[n].i2 = i2; n++; }
    }
    return n;
}

static int central_normal(int E, int R, const double *A, const double *b, int cone_norm2, const double *p, double cs, double sn,
                          const double *G, const double *h, const double *xi, double kappa0, const double *a_star,
                          double *lam, double *mu, double *m_out)
{
    const double PI = 3.14159265358979323846;
    pvert rv[EMAX * (EMAX - 1) / 2 > RMAX * (RMAX - 1) / 2 ? EMAX * (EMAX - 1) / 2 : RMAX * (RMAX - 1) / 2], ov[EMAX * (EMAX - 1) / 2 + 1];
    int nr = polygon_vertices(R, G, h, rv), nv; double off = 0;
    if (nr < 3) return 0;
    if (cone_norm2) { ov[0].x = b[0]; ov[0].y = b[1]; ov[0].i1 = ov[0].i2 = -1; nv = 1; off = b[2]; }
    else { nv = polygon_vertices(E, A, b, ov); if (nv < 3) return 0; }
    const double th0 = atan2(a_star[1], a_star[0]);
    double lo = PI, hi = PI;
    for (int k = 0; k < nv; ++k) for (int j = 0; j < nr; ++j) {
        const double wx = p[0] - ov[k].x + (cs * rv[j].x - sn * rv[j].y), wy = p[1] - ov[k].y + (sn * rv[j].x + cs * rv[j].y);
        const double cj = xi[0] * rv[j].x + xi[1] * rv[j].y + kappa0 + off;
        const double nw = hypot(wx, wy);
        if (!(nw > 0)) { if (cj < 0) return 0; continue; }
        const double q = -cj / nw;
        if (q >= 1.0) return 0;
        if (q <= -1.0) continue;
        const double beta = acos(q);
        double d = th0 - atan2(wy, wx);
        d = fmod(d + PI, 2 * PI); if (d < 0) d += 2 * PI; d -= PI;
        if (fabs(d) > beta) return 0;
        if (beta - d < hi) hi = beta - d;
        if (beta + d < lo) lo = beta + d;
    }
    if (hi >= PI && lo >= PI) return 0;
    const double thc = th0 + 0.5 * (hi - lo), a0 = cos(thc), a1 = sin(thc);
    for (int i = 0; i < E; ++i) lam[i] = 0;
    for (int j = 0; j < R; ++j) mu[j] = 0;
    if (cone_norm2) { lam[0] = a0; lam[1] = a1; lam[2] = -1.0; }
    else {
        int kb = 0; double best = -INFINITY;
        for (int k = 0; k < nv; ++k) { double v = a0 * ov[k].x + a1 * ov[k].y; if (v > best) { best = v; kb = k; } }
        const int i1 = ov[kb].i1, i2 = ov[kb].i2;
        const double det = A[2 * i1] * A[2 * i2 + 1] - A[2 * i1 + 1] * A[2 * i2];
        lam[i1] = (a0 * A[2 * i2 + 1] - A[2 * i2] * a1) / det;          /* A_S' lam_S = a */
        lam[i2] = (A[2 * i1] * a1 - a0 * A[2 * i1 + 1]) / det;
    }
    const double gx = -(cs * a0 + sn * a1) - xi[0], gy = -(-sn * a0 + cs * a1) - xi[1];    /* g = -R'a - xi */
    {
        int jb = 0; double best = -INFINITY;
        for (int j = 0; j < nr; ++j) { double v = gx * rv[j].x + gy * rv[j].y; if (v > best) { best = v; jb = j; } }
        const int j1 = rv[jb].i1, j2 = rv[jb].i2;
        const double det = G[2 * j1] * G[2 * j2 + 1] - G[2 * j1 + 1] * G[2 * j2];
        mu[j1] = (gx * G[2 * j2 + 1] - G[2 * j2] * gy) / det;
        mu[j2] = (G[2 * j1] * gy - gx * G[2 * j1 + 1]) / det;
    }
    if (!cone_norm2) for (int i = 0; i < E; ++i) { if (lam[i] < -1e-9) return 0; if (lam[i] < 0) lam[i] = 0; }
    for (int j = 0; j < R; ++j) { if (mu[j] < -1e-9) return 0; if (mu[j] < 0) mu[j] = 0; }
    double m = kappa0;
    for (int i = 0; i < E; ++i) m += lam[i] * (A[2 * i] * p[0] + A[2 * i + 1] * p[1] - b[i]);
    for (int j = 0; j < R; ++j) m -= mu[j] * h[j];
    if (m < 0) return 0;
    *m_out = m;
    return 1;
}

int orc_lammuz_one(int E, int R, const double *A, const double *b, int cone_norm2,
                   const double *p, double phi, const double *G, const double *h,
                   const double *xi, double zeta, double dbar, double ro2, double delta,
                   int accelerated, double *lam_out, double *mu_out, double *z_out, double *cmh)
{
    if (E > EMAX || R > RMAX) return -1;
    lmz_ctx c; c.E = E; c.R = R; c.A = A; c.b = b; c.G = G; c.h = h;
    double cs = cos(phi), sn = sin(phi);
    for (int i = 0; i < E; ++i) {
        c.q[i] = A[2 * i] * p[0] + A[2 * i + 1] * p[1] - b[i];
        c.M[i][0] = A[2 * i] * cs + A[2 * i + 1] * sn;       /* (A R)[i][0] */
        c.M[i][1] = -A[2 * i] * sn + A[2 * i + 1] * cs;
    }
    c.xi[0] = xi[0]; c.xi[1] = xi[1]; c.kappa0 = zeta - dbar; c.ro2 = ro2; c.delta = delta;

    /* mu candidates */
    mu_cand mcs[1 + RMAX + RMAX * (RMAX - 1) / 2]; int nm = 0;
    mcs[nm].k = 0; nm++;
    for (int j = 0; j < R; ++j)
        if (G[2 * j] * G[2 * j] + G[2 * j + 1] * G[2 * j + 1] > 0) { mcs[nm].k = 1; mcs[nm].j[0] = j; nm++; }
    for (int j1 = 0; j1 < R; ++j1) for (int j2 = j1 + 1; j2 < R; ++j2) {
        double det = G[2 * j1] * G[2 * j2 + 1] - G[2 * j1 + 1] * G[2 * j2];
        double n1 = hypot(G[2 * j1], G[2 * j1 + 1]), n2 = hypot(G[2 * j2], G[2 * j2 + 1]);
        if (det != 0 && fabs(det) > 1e-12 * n1 * n2) {
            mu_cand *m = &mcs[nm++]; m->k = 2; m->j[0] = j1; m->j[1] = j2;
            m->P[0][0] = G[2 * j1]; m->P[0][1] = G[2 * j2]; m->P[1][0] = G[2 * j1 + 1]; m->P[1][1] = G[2 * j2 + 1];
            /* robot vertex: G_S r = h_S */
            m->r[0] = (h[j1] * G[2 * j2 + 1] - G[2 * j1 + 1] * h[j2]) / det;
            m->r[1] = (G[2 * j1] * h[j2] - h[j1] * G[2 * j2]) / det;
            m->rr = m->r[0] * m->r[0] + m->r[1] * m->r[1];
        }
    }
    /* lam candidates: type 0 L0, 1 L1(i), 2 L2(i1,i2), 3 LC (||a|| = 1), 4 LI (0 < ||a|| < 1) */
    int lt[1 + EMAX + EMAX * (EMAX - 1) / 2], li1[1 + EMAX + EMAX * (EMAX - 1) / 2], li2[1 + EMAX + EMAX * (EMAX - 1) / 2];
    int nl = 0;
    lt[nl] = 0; li1[nl] = li2[nl] = -1; nl++;
    if (cone_norm2) { lt[nl] = 3; li1[nl] = li2[nl] = -1; nl++; lt[nl] = 4; li1[nl] = li2[nl] = -1; nl++; }
    else {
        for (int i = 0; i < E; ++i)
            if (A[2 * i] * A[2 * i] + A[2 * i + 1] * A[2 * i + 1] > 0) { lt[nl] = 1; li1[nl] = i; li2[nl] = -1; nl++; }
        for (int i1 = 0; i1 < E; ++i1) for (int i2 = i1 + 1; i2 < E; ++i2) {
            double det = A[2 * i1] * A[2 * i2 + 1] - A[2 * i1 + 1] * A[2 * i2];
            double n1 = hypot(A[2 * i1], A[2 * i1 + 1]), n2 = hypot(A[2 * i2], A[2 * i2 + 1]);
            if (det != 0 && fabs(det) > 1e-12 * n1 * n2) { lt[nl] = 2; li1[nl] = i1; li2[nl] = i2; nl++; }
        }
    }

    double best_cost = INFINITY; int best_idx = 0;
    double best_m = 0, best_H[2] = {0, 0};
    for (int i = 0; i < E; ++i) lam_out[i] = 0;
    for (int j = 0; j < R; ++j) mu_out[j] = 0;
    /* Rule T3.  Pass 1 (ic = 0): the hinge-inactive candidates are ranked by the cost of the hinge-inactive MODEL
     * (-delta*m + ro2/2|H|^2: a lower bound of the true cost, exact for m >= 0); if its minimiser c0 has m >= 0 it
     * is the global optimum.  Otherwise pass 2 (ic = 1): the hinge-active candidates and c0 compete on the TRUE
     * cost.  Exact ties go to the lowest id = 2*(il*nm+im)+ic. */
    int best_id = 0x7fffffff;
    for (int ic = 0; ic < 2; ++ic) {
    if (ic == 1) {
        if (best_m >= 0) break;
        best_cost += 0.5 * best_m * best_m;            /* c0: model cost -> true cost (m < 0 here) */
    }
    for (int il = 0; il < nl; ++il) for (int im = 0; im < nm; ++im) {
        int idx = 2 * (il * nm + im) + ic;
        const mu_cand *mc = &mcs[im]; double chi = (double)ic;
        double lam[EMAX]; for (int i = 0; i < E; ++i) lam[i] = 0;
        double gam[2] = {0, 0}, m, H[2];
        int nsol = 1, interior = -1; double ats[4] = {0, 0, 0, 0}, ut_[2] = {0, 0}, l0_ = 0, detS_ = 1, AS_[4] = {0, 0, 0, 0};
        if (lt[il] == 0) {
            gamma_star(&c, mc, chi, c.kappa0, c.xi, gam, &m, H);
        } else if (lt[il] == 1) {
            int i = li1[il];
            double amax = 1.0 / hypot(A[2 * i], A[2 * i + 1]);
            double e[2], d0, d1, al;
            gamma_star(&c, mc, chi, c.kappa0, c.xi, gam, &m, H);
            d0 = (chi * m - delta) * c.q[i] + ro2 * (c.M[i][0] * H[0] + c.M[i][1] * H[1]);
            e[0] = amax * c.M[i][0] + c.xi[0]; e[1] = amax * c.M[i][1] + c.xi[1];
            gamma_star(&c, mc, chi, amax * c.q[i] + c.kappa0, e, gam, &m, H);
            d1 = (chi * m - delta) * c.q[i] + ro2 * (c.M[i][0] * H[0] + c.M[i][1] * H[1]);
            if (d1 <= 0) al = amax; else if (d0 >= 0) al = 0; else al = amax * d0 / (d0 - d1);
            e[0] = al * c.M[i][0] + c.xi[0]; e[1] = al * c.M[i][1] + c.xi[1];
            gamma_star(&c, mc, chi, al * c.q[i] + c.kappa0, e, gam, &m, H);
            lam[i] = al;
        } else {
            double ut[2], l0 = 0, AS[2][2] = {{0, 0}, {0, 0}}, detS = 1;
            int i1 = li1[il], i2 = li2[il];
            double dv[2];
            if (lt[il] == 2) {
                AS[0][0] = A[2 * i1]; AS[0][1] = A[2 * i1 + 1]; AS[1][0] = A[2 * i2]; AS[1][1] = A[2 * i2 + 1];
                detS = AS[0][0] * AS[1][1] - AS[0][1] * AS[1][0];
                double vx = (b[i1] * AS[1][1] - AS[0][1] * b[i2]) / detS;
                double vy = (AS[0][0] * b[i2] - b[i1] * AS[1][0]) / detS;
                dv[0] = p[0] - vx; dv[1] = p[1] - vy;
            } else { dv[0] = p[0] - b[0]; dv[1] = p[1] - b[1]; l0 = b[2]; }
            ut[0] = cs * dv[0] + sn * dv[1]; ut[1] = -sn * dv[0] + cs * dv[1];      /* R' (p - v) */
            double g0[2], g1[2], g2[2], e[2];
            /* gradient of the reduced model at 0, e1, e2 */
            gamma_star(&c, mc, chi, l0 + c.kappa0, c.xi, gam, &m, H);
            g0[0] = (chi * m - delta) * ut[0] + ro2 * H[0]; g0[1] = (chi * m - delta) * ut[1] + ro2 * H[1];
            e[0] = 1 + c.xi[0]; e[1] = c.xi[1];
            gamma_star(&c, mc, chi, ut[0] + l0 + c.kappa0, e, gam, &m, H);
            g1[0] = (chi * m - delta) * ut[0] + ro2 * H[0] - g0[0]; g1[1] = (chi * m - delta) * ut[1] + ro2 * H[1] - g0[1];
            e[0] = c.xi[0]; e[1] = 1 + c.xi[1];
            gamma_star(&c, mc, chi, ut[1] + l0 + c.kappa0, e, gam, &m, H);
            g2[0] = (chi * m - delta) * ut[0] + ro2 * H[0] - g0[0]; g2[1] = (chi * m - delta) * ut[1] + ro2 * H[1] - g0[1];
            double q12 = 0.5 * (g1[1] + g2[0]);
            if (lt[il] == 4) { nsol = mc->k < 2 ? circle_interior(&c, mc, chi, ut, l0, ats) : 0; interior = 0; }
            else nsol = trs2(g1[0], q12, g2[1], g0[0], g0[1], lt[il] == 2, ats);
            ut_[0] = ut[0]; ut_[1] = ut[1]; l0_ = l0; detS_ = detS;
            AS_[0] = AS[0][0]; AS_[1] = AS[0][1]; AS_[2] = AS[1][0]; AS_[3] = AS[1][1];
        }
        for (int sol = 0; sol < nsol; ++sol) {
        if (lt[il] >= 2) {
            int i1 = li1[il], i2 = li2[il];
            double at0 = ats[2 * sol], at1 = ats[2 * sol + 1], e[2];
            for (int i = 0; i < E; ++i) lam[i] = 0;
            e[0] = at0 + c.xi[0]; e[1] = at1 + c.xi[1];
            double sc_ = sol == interior ? hypot(at0, at1) : 1.0;   /* circle: lam_3 = -||a|| (1 on the boundary) */
            gamma_star(&c, mc, chi, at0 * ut_[0] + at1 * ut_[1] + l0_ * sc_ + c.kappa0, e, gam, &m, H);
            double ax = cs * at0 - sn * at1, ay = sn * at0 + cs * at1;   /* a = R at */
            if (lt[il] == 2) {
                /* A_S' lamS = a */
                lam[i1] = (ax * AS_[3] - AS_[2] * ay) / detS_;
                lam[i2] = (AS_[0] * ay - ax * AS_[1]) / detS_;
            } else { lam[0] = ax; lam[1] = ay; lam[2] = -hypot(ax, ay); }
        }
        double mu[RMAX]; for (int j = 0; j < R; ++j) mu[j] = 0;
        for (int k = 0; k < mc->k; ++k) mu[mc->j[k]] = gam[k];
        int ok = 1;
        if (!cone_norm2) for (int i = 0; i < E; ++i) { if (lam[i] < -SIGN_TOL) ok = 0; else if (lam[i] < 0) lam[i] = 0; }
        for (int j = 0; j < R; ++j) { if (mu[j] < -SIGN_TOL) ok = 0; else if (mu[j] < 0) mu[j] = 0; }
        if (!ok) continue;
        /* true cost */
        double mm = c.kappa0, HH[2] = {c.xi[0], c.xi[1]};
        for (int i = 0; i < E; ++i) { mm += lam[i] * c.q[i]; HH[0] += lam[i] * c.M[i][0]; HH[1] += lam[i] * c.M[i][1]; }
        for (int j = 0; j < R; ++j) { mm -= mu[j] * h[j]; HH[0] += mu[j] * G[2 * j]; HH[1] += mu[j] * G[2 * j + 1]; }
        double ng = mm < 0 ? mm : 0;
        double cost = (ic ? 0.5 * ng * ng : 0.0) - delta * mm + 0.5 * ro2 * (HH[0] * HH[0] + HH[1] * HH[1]);
        if (cost < best_cost || (cost == best_cost && idx < best_id)) {
            best_cost = cost; best_idx = idx; best_id = idx; best_m = mm; best_H[0] = HH[0]; best_H[1] = HH[1];
            for (int i = 0; i < E; ++i) lam_out[i] = lam[i];
            for (int j = 0; j < R; ++j) mu_out[j] = mu[j];
        }
        }
    }
    }
    /* tie-break T1 (slack regime): central separating normal instead of the max-clearance one */
    if (g_centre && best_m > 0 && best_H[0] * best_H[0] + best_H[1] * best_H[1] < 1e-8) {
        double as[2] = {0, 0}, lc[EMAX], mc_[RMAX], mm;
        for (int i = 0; i < E; ++i) { as[0] += lam_out[i] * A[2 * i]; as[1] += lam_out[i] * A[2 * i + 1]; }
        if (as[0] * as[0] + as[1] * as[1] >= 1.0 - 1e-9 &&      /* a* on the unit circle: it has a direction */
            central_normal(E, R, A, b, cone_norm2, p, cs, sn, G, h, xi, c.kappa0, as, lc, mc_, &mm)) {
            for (int i = 0; i < E; ++i) lam_out[i] = lc[i];
            for (int j = 0; j < R; ++j) mu_out[j] = mc_[j];
            best_m = mm; best_H[0] = c.xi[0]; best_H[1] = c.xi[1];
            for (int i = 0; i < E; ++i) { best_H[0] += lc[i] * c.M[i][0]; best_H[1] += lc[i] * c.M[i][1]; }
            for (int j = 0; j < R; ++j) { best_H[0] += mc_[j] * G[2 * j]; best_H[1] += mc_[j] * G[2 * j + 1]; }
        }
    }
    *z_out = (accelerated ? 0.5 : 1.0) * (best_m > 0 ? best_m : 0);          /* tie-break T2 */
    if (cmh) { cmh[0] = best_cost; cmh[1] = best_m; cmh[2] = best_H[0]; cmh[3] = best_H[1]; }
    return best_idx;
}

/* ------------------------------------------------------------------------------------------ */
/* linearised motion models, rda_solver.py:949-994 (A 3x3, B 3x2, C 3) about (s_t, u_t)       */
static void lin_model(const orc_cfg *c, const double *st, const double *ut, double *A, double *B, double *C)
{
    double dt = c->dt;
    memset(A, 0, 9 * sizeof(double)); memset(B, 0, 6 * sizeof(double)); memset(C, 0, 3 * sizeof(double));
    A[0] = A[4] = A[8] = 1;
    if (c->dynamics == 2) {         /* omni: phi := velocity heading u[1] */
        double phi = ut[1], v = ut[0];
        B[0] = cos(phi) * dt; B[1] = -v * sin(phi) * dt; B[2] = sin(phi) * dt; B[3] = v * cos(phi) * dt;
        C[0] = phi * v * sin(phi) * dt; C[1] = -phi * v * cos(phi) * dt;
        return;
    }
    double phi = st[2], v = ut[0];
    A[2] = -v * dt * sin(phi); A[5] = v * dt * cos(phi);
    B[0] = cos(phi) * dt; B[2] = sin(phi) * dt;
    C[0] = phi * v * sin(phi) * dt; C[1] = -phi * v * cos(phi) * dt;
    if (c->dynamics == 0) {
        double psi = ut[1], cp = cos(psi);
        B[4] = tan(psi) * dt / c->L; B[5] = v * dt / (c->L * cp * cp);
        C[2] = -psi * v * dt / (c->L * cp * cp);
    } else {
        B[5] = dt;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* su-problem: dense primal-dual IPM on x = (u_0..u_{T-1}, d_0..d_{T-1})                       */
typedef struct { int i1, i2; double c1, c2, e; } lincon;

typedef struct {
    const orc_cfg *c; int T, N;
    const double *a, *cc, *g, *ref, *nom_s;
    double ref_speed;
    double *Ak, *Bk, *Ck;       /* T x (9,6,3) */
    double *Gam;                /* [t][k] 3x2, k<=t : d s_{t+1} / d u_k */
    double *Q0, *Q1, *Q2;       /* per t */
    double s0[3];
} su_ctx;

static void su_rollout(const su_ctx *S, const double *x, double *s /*3x(T+1) row-major*/)
{
    int T = S->T;
    for (int r = 0; r < 3; ++r) s[r * (T + 1)] = S->s0[r];
    for (int t = 0; t < T; ++t) {
        const double *A = &S->Ak[9 * t], *B = &S->Bk[6 * t], *C = &S->Ck[3 * t];
        for (int r = 0; r < 3; ++r) {
            double v = C[r];
            for (int k = 0; k < 3; ++k) v += A[3 * r + k] * s[k * (T + 1) + t];
            v += B[2 * r] * x[2 * t] + B[2 * r + 1] * x[2 * t + 1];
            s[r * (T + 1) + t + 1] = v;
        }
    }
}

static __thread double t_su_rd0 = 0.0;       /* relative dual residual rd / (1 + |g|) of the FIRST iterate of the last su_solve_impl call (start-rule key) */
static __thread double t_su_eps = 0.0;       /* smoothing width of the hinge terms in su_eval: 0 except in the rescue phase below */
/* objective, gradient (n) and generalised Hessian (n x n) at x */
static double su_eval(const su_ctx *S, const double *x, double *s, double *grad, double *Hm)
{
    const orc_cfg *c = S->c; int T = S->T, N = S->N, n = 3 * T;
    su_rollout(S, x, s);
    double f = 0;
    if (grad) memset(grad, 0, n * sizeof(double));
    if (Hm) memset(Hm, 0, (size_t)n * n * sizeof(double));
    double wz = c->dynamics == 2 ? 0.0 : 1.0;
    for (int t = 0; t < T; ++t) {
        double st[3] = { s[t + 1], s[(T + 1) + t + 1], s[2 * (T + 1) + t + 1] };
        double gs[3], Hs[3][3] = {{0}}, gd = 0, Hsd[2] = {0, 0}, Hdd = 0;
        double w[3] = { 1, 1, wz };
        for (int r = 0; r < 3; ++r) {
            double df = st[r] - S->ref[r * (T + 1) + t + 1];
            f += c->ws * w[r] * df * df; gs[r] = 2 * c->ws * w[r] * df; Hs[r][r] = 2 * c->ws * w[r];
        }
        double dl = st[2] - S->nom_s[2 * (T + 1) + t];
        f += 0.5 * c->ro2 * (S->Q0[t] + S->Q1[t] * dl + S->Q2[t] * dl * dl);
        gs[2] += 0.5 * c->ro2 * (S->Q1[t] + 2 * S->Q2[t] * dl); Hs[2][2] += c->ro2 * S->Q2[t];
        double dt_ = x[2 * T + t];
        for (int nn = 0; nn < N; ++nn) {
            const double *a = &S->a[(nn * T + t) * 2];
            double Im = a[0] * st[0] + a[1] * st[1] - S->cc[nn * T + t] - dt_;
            if (c->accelerated && t_su_eps > 0) {       /* rescue phase of a cycling cold attempt (su_solve_impl): neg(Im) -> (sqrt(Im^2 + 4 eps^2) - Im) / 2 */
                const double e2 = 4 * t_su_eps * t_su_eps, rt = sqrt(Im * Im + e2), sv = 0.5 * (rt - Im), ds = 0.5 * (Im / rt - 1.0), d2 = 0.5 * e2 / (rt * rt * rt);
                const double c1 = c->ro1 * sv * ds, c2 = c->ro1 * (ds * ds + sv * d2);
                f += 0.5 * c->ro1 * sv * sv;
                gs[0] += c1 * a[0]; gs[1] += c1 * a[1]; gd -= c1;
                Hs[0][0] += c2 * a[0] * a[0]; Hs[0][1] += c2 * a[0] * a[1]; Hs[1][1] += c2 * a[1] * a[1];
                Hsd[0] -= c2 * a[0]; Hsd[1] -= c2 * a[1]; Hdd += c2;
            } else
            if (!c->accelerated || Im < 0) {
                f += 0.5 * c->ro1 * Im * Im;
                gs[0] += c->ro1 * Im * a[0]; gs[1] += c->ro1 * Im * a[1]; gd -= c->ro1 * Im;
                Hs[0][0] += c->ro1 * a[0] * a[0]; Hs[0][1] += c->ro1 * a[0] * a[1]; Hs[1][1] += c->ro1 * a[1] * a[1];
                Hsd[0] -= c->ro1 * a[0]; Hsd[1] -= c->ro1 * a[1]; Hdd += c->ro1;
            }
        }
        Hs[1][0] = Hs[0][1];
        f -= c->slack_gain * dt_;
        if (grad) {
            grad[2 * T + t] += gd - c->slack_gain;
            for (int k = 0; k <= t; ++k) {
                const double *Gm = &S->Gam[(t * T + k) * 6];
                for (int j = 0; j < 2; ++j)
                    grad[2 * k + j] += Gm[j] * gs[0] + Gm[2 + j] * gs[1] + Gm[4 + j] * gs[2];
            }
        }
        if (Hm) {
            Hm[(2 * T + t) * n + 2 * T + t] += Hdd;
            for (int k = 0; k <= t; ++k) {
                const double *Gk = &S->Gam[(t * T + k) * 6];
                double HG[3][2];
                for (int r = 0; r < 3; ++r) for (int j = 0; j < 2; ++j)
                    HG[r][j] = Hs[r][0] * Gk[j] + Hs[r][1] * Gk[2 + j] + Hs[r][2] * Gk[4 + j];
                for (int j = 0; j < 2; ++j) {
                    double v = Gk[j] * Hsd[0] + Gk[2 + j] * Hsd[1];
                    Hm[(2 * k + j) * n + 2 * T + t] += v; Hm[(2 * T + t) * n + 2 * k + j] += v;
                }
                for (int l = 0; l <= t; ++l) {
                    const double *Gl = &S->Gam[(t * T + l) * 6];
                    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
                        Hm[(2 * l + i) * n + 2 * k + j] += Gl[i] * HG[0][j] + Gl[2 + i] * HG[1][j] + Gl[4 + i] * HG[2][j];
                }
            }
        }
    }
    for (int t = 0; t < T; ++t) {
        double dv = x[2 * t] - S->ref_speed;
        f += c->wu * dv * dv + 0.5 * c->eps_u * (x[2 * t] * x[2 * t] + x[2 * t + 1] * x[2 * t + 1]);
        if (grad) { grad[2 * t] += 2 * c->wu * dv + c->eps_u * x[2 * t]; grad[2 * t + 1] += c->eps_u * x[2 * t + 1]; }
        if (Hm) { Hm[(2 * t) * n + 2 * t] += 2 * c->wu + c->eps_u; Hm[(2 * t + 1) * n + 2 * t + 1] += c->eps_u; }
    }
    return f;
}

static int chol_factor(double *K, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = K[j * n + j];
        for (int k = 0; k < j; ++k) d -= K[j * n + k] * K[j * n + k];
        if (!(d > 0)) return 1;
        d = sqrt(d); K[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = K[i * n + j];
            for (int k = 0; k < j; ++k) v -= K[i * n + k] * K[j * n + k];
            K[i * n + j] = v / d;
        }
    }
    return 0;
}
static void chol_solve(const double *K, int n, double *rhs)
{
    for (int i = 0; i < n; ++i) { double v = rhs[i]; for (int k = 0; k < i; ++k) v -= K[i * n + k] * rhs[k]; rhs[i] = v / K[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double v = rhs[i]; for (int k = i + 1; k < n; ++k) v -= K[k * n + i] * rhs[k]; rhs[i] = v / K[i * n + i]; }
}

#ifndef SIGMA_FLOOR
#define SIGMA_FLOOR 1e-3
#endif
#define SU_CENTRE_GAMMA 1e-5      /* = su_device.h */
#define SU_SMOOTH_K 0.1           /* = su_device.h */
#define SU_COLD_CAP 50            /* iterations granted to the cold attempt (= su_device.h) */
#define SU_SAFE_SIGMA 0.3         /* last-resort attempt: centring parameter ... */
#define SU_SAFE_SIGMA_END 0.05    /* ... once the last step was >= 0.9 */
#define SU_SAFE_TAU 0.9           /* ... fraction to the boundary */
#define SU_SAFE_GAMMA 1e-2        /* ... lam w >= this x mu after every step          (all four = su_device.h) */
#ifndef SU_CENTRE_FROM
#define SU_CENTRE_FROM 25         /* = su_device.h */
#endif
/* Warm start of the su-problems of ADMM iterations >= 1 (same rule as csrc/su_device.h): lam_keep[mc] = the inequality multipliers
 * the last converged solve ended with (this function's own row order); warm != 0: the slacks of the start are floored at warm_wfl
 * and the multipliers are the larger of warm_mu0 / w and lam_keep; the warm attempt gets warm_cap iterations, then the cold rule. */
/* ---- landing of a converged interior-point iterate on the vertex it approaches (round 6) ------------------------------------------
 * The interior point stops ON the central path: a row that is only just active (multiplier lam* ~ 1e-7) still has the slack mu / lam*, so two
 * iterations that stop at different mu hand back controls up to ~1e-4 apart (tests/test_oracle_su.py::test_stop_tolerance_vs_weakly_active_rows) - the
 * reason for TOL_U.  The landing: freeze the active set A = {lam_i > w_i}, solve the equality-constrained quadratic model at x (the hinge terms'
 * generalised Hessian at x) for x+, the multipliers nu of A come out of it; accept when x+ is feasible for the other rows, nu >= 0 and the TRUE
 * gradient at x+ (hinge terms re-evaluated) is stationary; otherwise move rows in / out of A by their signs (primal-dual active set) and try again, a few
 * rounds; no acceptance: the interior-point iterate is returned as before.  The model is solved by the method of multipliers on
 * K = H + rho C_A' C_A (one Cholesky factor, a few solves: converges like (|H| / rho)^k), so no row of C_A has to be independent of the others. */
static int g_su_land = 1; static double g_su_land_tol[3] = {1e-3, 1e-4, 1e-5};   /* = rda_opts::su_land, su_land_tol (first stop; a refused landing is tried again at 1e-2 x, then never) */
static __thread int t_su_landed = 0, t_su_land_rounds = 0;
static long g_pol_stat[8];       /* calls, accepted, rounds summed, last verdict: chol failed, set still moving, not stationary; rows moved */
void orc_get_su_land_stats(long *out8) { for (int i = 0; i < 8; ++i) { out8[i] = g_pol_stat[i]; g_pol_stat[i] = 0; } }
void orc_set_su_land(int on) { g_su_land = on; }
void orc_set_su_land_tol(double rd, double rp, double mu) { if (rd > 0 && rp > 0 && mu > 0) { g_su_land_tol[0] = rd; g_su_land_tol[1] = rp; g_su_land_tol[2] = mu; } }
int orc_get_su_landed(void) { return t_su_landed; }
static int su_land(const su_ctx *S, const lincon *con, int mc, int n, double *x, double *lm, double *w, double *s, double *grad, double *Hm, double *K, double tol_rd)
{
    double *dx = malloc(sizeof(double) * n), *rhs = malloc(sizeof(double) * n), *nu = malloc(sizeof(double) * mc), *xn = malloc(sizeof(double) * n), *g2 = malloc(sizeof(double) * n);
    char *act = malloc(mc);
    int ok = 0, last = 0;
    __sync_fetch_and_add(&g_pol_stat[0], 1);
    for (int i = 0; i < mc; ++i) { act[i] = lm[i] > w[i]; nu[i] = act[i] ? lm[i] : 0.0; }
    for (int round = 0; round < 4 && !ok; ++round) {
        t_su_land_rounds = round + 1;
        su_eval(S, x, s, grad, Hm);
        double hmax = 0, gn = 0;
        for (int i = 0; i < n; ++i) { if (Hm[i * n + i] > hmax) hmax = Hm[i * n + i]; if (fabs(grad[i]) > gn) gn = fabs(grad[i]); }
        const double rho = 1e4 * (hmax > 1 ? hmax : 1.0), sc = 1 + gn;
        memcpy(K, Hm, sizeof(double) * n * n);
        for (int i = 0; i < mc; ++i) if (act[i]) {
            int i1 = con[i].i1, i2 = con[i].i2;
            K[i1 * n + i1] += rho * con[i].c1 * con[i].c1;
            if (i2 >= 0) { K[i2 * n + i2] += rho * con[i].c2 * con[i].c2; K[i1 * n + i2] += rho * con[i].c1 * con[i].c2; K[i2 * n + i1] += rho * con[i].c1 * con[i].c2; }
        }
        if (chol_factor(K, n)) { last = 3; break; }
        for (int k = 0; k < 6; ++k) {                       /* method of multipliers on the quadratic model: dx always measured from x */
            for (int i = 0; i < n; ++i) rhs[i] = -grad[i];
            for (int i = 0; i < mc; ++i) if (act[i]) {
                double r = con[i].c1 * x[con[i].i1] + (con[i].i2 >= 0 ? con[i].c2 * x[con[i].i2] : 0) - con[i].e, v = nu[i] + rho * r;
                rhs[con[i].i1] -= con[i].c1 * v; if (con[i].i2 >= 0) rhs[con[i].i2] -= con[i].c2 * v;
            }
            memcpy(dx, rhs, sizeof(double) * n);
            chol_solve(K, n, dx);
            for (int i = 0; i < mc; ++i) if (act[i]) {
                double r = con[i].c1 * (x[con[i].i1] + dx[con[i].i1]) + (con[i].i2 >= 0 ? con[i].c2 * (x[con[i].i2] + dx[con[i].i2]) : 0) - con[i].e;
                nu[i] += rho * r;
            }
        }
        for (int i = 0; i < n; ++i) xn[i] = x[i] + dx[i];
        /* verdict on x+: feasibility of the rows outside A, signs of nu, stationarity of the TRUE objective */
        int bad = 0;
        su_eval(S, xn, s, g2, NULL);
        for (int i = 0; i < mc; ++i) if (act[i]) { g2[con[i].i1] += con[i].c1 * nu[i]; if (con[i].i2 >= 0) g2[con[i].i2] += con[i].c2 * nu[i]; }
        double rdn = 0; for (int i = 0; i < n; ++i) if (fabs(g2[i]) > rdn) rdn = fabs(g2[i]);
        for (int i = 0; i < mc; ++i) {
            double cx = con[i].c1 * xn[con[i].i1] + (con[i].i2 >= 0 ? con[i].c2 * xn[con[i].i2] : 0);
            if (act[i]) { if (nu[i] < -1e-9 * sc) { act[i] = 0; nu[i] = 0; bad = 1; __sync_fetch_and_add(&g_pol_stat[6], 1); } }
            else if (cx > con[i].e + 1e-11 * (1 + fabs(con[i].e))) { act[i] = 1; nu[i] = 0; bad = 1; __sync_fetch_and_add(&g_pol_stat[6], 1); }
        }
        last = bad ? 4 : 5;
        if (getenv("ORC_LAND_DEBUG")) fprintf(stderr, "landing round %d: bad %d rdn %.2e (tol %.2e) |dx| %.2e\n", round, bad, rdn, tol_rd * sc, fabs(dx[0]));
        if (!bad && rdn <= 100 * tol_rd * sc) {
            memcpy(x, xn, sizeof(double) * n);
            for (int i = 0; i < mc; ++i) {
                double cx = con[i].c1 * x[con[i].i1] + (con[i].i2 >= 0 ? con[i].c2 * x[con[i].i2] : 0);
                if (act[i]) { lm[i] = nu[i] > 0 ? nu[i] : 0.0; w[i] = 0.0; } else { lm[i] = 0.0; w[i] = con[i].e - cx; }
            }
            ok = 1;
        } else if (!bad) memcpy(x, xn, sizeof(double) * n), ok = 0, bad = 2;      /* same set, a hinge term switched: linearise again at x+ */
        if (bad == 2) { /* x moved: the interior-point multipliers no longer belong to it, keep nu */ }
    }
    free(dx); free(rhs); free(nu); free(xn); free(g2); free(act);
    __sync_fetch_and_add(&g_pol_stat[ok ? 1 : last], 1); __sync_fetch_and_add(&g_pol_stat[2], t_su_land_rounds);
    return ok;
}

static int su_solve_impl(const orc_cfg *c, const double *nom_s, const double *nom_u, const double *ref_s,
                         double ref_speed, const double *a, const double *cc, const double *g,
                         const double *d0, double *s_out, double *u_out, double *d_out, int *ipm_iters,
                         double *lam_keep, int warm, double warm_wfl, double warm_mu0, int warm_cap, int warm_shift, double hard_dmu)
{
    int T = c->T, N = c->N, n = 3 * T;
    su_ctx S; S.c = c; S.T = T; S.N = N; S.a = a; S.cc = cc; S.g = g; S.ref = ref_s; S.nom_s = nom_s; S.ref_speed = ref_speed;
    S.Ak = malloc(sizeof(double) * 9 * T); S.Bk = malloc(sizeof(double) * 6 * T); S.Ck = malloc(sizeof(double) * 3 * T);
    S.Gam = calloc((size_t)T * T * 6, sizeof(double));
    S.Q0 = calloc(T, sizeof(double)); S.Q1 = calloc(T, sizeof(double)); S.Q2 = calloc(T, sizeof(double));
    for (int r = 0; r < 3; ++r) S.s0[r] = nom_s[r * (T + 1)];
    for (int t = 0; t < T; ++t) {
        double st[3] = { nom_s[t], nom_s[(T + 1) + t], nom_s[2 * (T + 1) + t] }, ut[2] = { nom_u[t], nom_u[T + t] };
        lin_model(c, st, ut, &S.Ak[9 * t], &S.Bk[6 * t], &S.Ck[3 * t]);
        /* Gam[t][t] = B_t ; Gam[t][k] = A_t Gam[t-1][k] */
        memcpy(&S.Gam[(t * T + t) * 6], &S.Bk[6 * t], 6 * sizeof(double));
        for (int k = 0; k < t; ++k) {
            const double *P = &S.Gam[((t - 1) * T + k) * 6]; double *Q = &S.Gam[(t * T + k) * 6];
            for (int r = 0; r < 3; ++r) for (int j = 0; j < 2; ++j)
                Q[2 * r + j] = S.Ak[9 * t + 3 * r] * P[j] + S.Ak[9 * t + 3 * r + 1] * P[2 + j] + S.Ak[9 * t + 3 * r + 2] * P[4 + j];
        }
        /* rotation-consistency penalty reduced to a scalar quadratic in delta_t */
        double phi = st[2], cs = cos(phi), sn = sin(phi);
        for (int nn = 0; nn < N; ++nn) {
            const double *an = &a[(nn * T + t) * 2], *gn = &g[(nn * T + t) * 2];
            double k0x = gn[0] + cs * an[0] + sn * an[1], k0y = gn[1] - sn * an[0] + cs * an[1];
            double k1x = -sn * an[0] + cs * an[1], k1y = -cs * an[0] - sn * an[1];
            S.Q0[t] += k0x * k0x + k0y * k0y; S.Q1[t] += 2 * (k0x * k1x + k0y * k1y); S.Q2[t] += k1x * k1x + k1y * k1y;
        }
    }
    /* constraints C x <= e */
    int mc = 4 * T + 4 * (T - 1) + 2 * T;
    lincon *con = malloc(sizeof(lincon) * mc); int m = 0;
    for (int t = 0; t < T; ++t) for (int i = 0; i < 2; ++i) {
        con[m++] = (lincon){ 2 * t + i, -1, 1.0, 0, c->max_speed[i] };
        con[m++] = (lincon){ 2 * t + i, -1, -1.0, 0, c->max_speed[i] };
    }
    for (int t = 0; t + 1 < T; ++t) for (int i = 0; i < 2; ++i) {
        con[m++] = (lincon){ 2 * (t + 1) + i, 2 * t + i, 1.0, -1.0, c->acce_bound[i] };
        con[m++] = (lincon){ 2 * (t + 1) + i, 2 * t + i, -1.0, 1.0, c->acce_bound[i] };
    }
    for (int t = 0; t < T; ++t) {
        con[m++] = (lincon){ 2 * T + t, -1, 1.0, 0, c->max_sd };
        con[m++] = (lincon){ 2 * T + t, -1, -1.0, 0, -c->min_sd };
    }
    mc = m;
    double *x = malloc(sizeof(double) * n), *grad = malloc(sizeof(double) * n), *Hm = malloc(sizeof(double) * n * n);
    double *K = malloc(sizeof(double) * n * n), *rhs = malloc(sizeof(double) * n), *dx = malloc(sizeof(double) * n);
    double *w = malloc(sizeof(double) * mc), *lm = malloc(sizeof(double) * mc), *rp = malloc(sizeof(double) * mc);
    double *dw = malloc(sizeof(double) * mc), *dl = malloc(sizeof(double) * mc), *rc = malloc(sizeof(double) * mc);
    double *s = malloc(sizeof(double) * 3 * (T + 1));
    /* Safety net of the CHECKER (not mirrored in the kernel; orc_set_su_accept(0) switches it off): the best iterate that is primal
     * feasible to tolerance, dual feasible to 10 x and complementary to 1000 x the stop tolerances (the class ECOS stops at) is kept.
     * A solve whose every attempt then loses its end game in rounding - the dual residual GROWS from 1e-9 to 1e-5 while mu falls from 1e-9 to
     * 1e-15 and the Cholesky factor breaks down (soak seed 9, scene 13, step 15; tests/golden/su_hard/omni_T25_N20_end_game_noise.npz:
     * the kernel's arithmetic converges in 16 iterations) - returns that iterate instead of "no update". */
    double *x_acc = malloc(sizeof(double) * n), *lm_acc = malloc(sizeof(double) * mc), acc_merit = INFINITY; int have_acc = 0;
    double *x_sav = malloc(sizeof(double) * n), *lm_sav = malloc(sizeof(double) * mc), *w_sav = malloc(sizeof(double) * mc); int land_failed = 0;
    t_su_landed = 0;
    /* Attempts: [-1 the warm start,] 0 the cold start, 1 the last resort.  The last one only runs when the others end without convergence
     * (the iteration cap, ~0.1% of closed-loop solves, where the iterates cycle): it restarts from the same nominal with a more central
     * point (slack floor 0.1, mu0 = 10) and - since round 5 - as a plain long-step path-following iteration (`safe` below). */
    int status = 1, it = 0, used = 0;
    warm = warm && lam_keep != NULL;
    for (int attempt = g_su_first_attempt ? 1 : (warm ? -1 : 0); attempt < 2 && status != 0; ++attempt) {
    const double wfl = attempt < 0 ? warm_wfl : (attempt ? 1e-1 : 1e-2), mu0 = attempt < 0 ? warm_mu0 : (attempt ? 10.0 : 1.0);
    /* (the cold attempt: 50 since round 5 - it was 100 while the last resort was a second Mehrotra attempt; every recorded solve that the cold
     * attempt finishes at all takes <= 37 iterations, a cycling one is better off in the last resort after 50 than after 100) */
    const int it_cap = attempt < 0 ? warm_cap : (attempt == 0 ? SU_COLD_CAP : 100);
    for (int t = 0; t < T; ++t) for (int i = 0; i < 2; ++i) {
        const double clipm = attempt < 0 ? cur_warm_clip : 0.01;
        double v = nom_u[i * T + t], lim = (1.0 - clipm) * c->max_speed[i];
        x[2 * t + i] = v > lim ? lim : (v < -lim ? -lim : v);
    }
    for (int t = 0; t < T; ++t) {
        const double clipm = attempt < 0 ? cur_warm_clip : 0.01;
        double v = d0 ? d0[t] : c->max_sd, lo = c->min_sd + clipm * (c->max_sd - c->min_sd), hi = c->max_sd - clipm * (c->max_sd - c->min_sd);
        x[2 * T + t] = v > hi ? hi : (v < lo ? lo : v);
    }
    for (int i = 0; i < mc; ++i) {
        double cx = con[i].c1 * x[con[i].i1] + (con[i].i2 >= 0 ? con[i].c2 * x[con[i].i2] : 0);
        double sl = con[i].e - cx;
        w[i] = sl > wfl ? sl : wfl; lm[i] = mu0 / w[i];
        if (attempt < 0) {
            /* first su-problem of a step: the multipliers are those of the previous STEP, whose stage t+1 is this step's stage t */
            int src = i;
            if (warm_shift) {
                if (i < 4 * T) { int t = i / 4 + 1; if (t > T - 1) t = T - 1; src = 4 * t + i % 4; }
                else if (i < 8 * T - 4) { int j = i - 4 * T, t = j / 4 + 1; if (t > T - 2) t = T - 2; src = 4 * T + 4 * t + j % 4; }
                else { int j = i - (8 * T - 4), t = j / 2 + 1; if (t > T - 1) t = T - 1; src = 8 * T - 4 + 2 * t + j % 2; }
            }
            if (lam_keep[src] > lm[i]) lm[i] = lam_keep[src];
            /* hard start (su_hard_warm): the rows of d get a barrier of their own where NEITHER bound was active - lam+ and lam- are raised by
             * the same delta = hard_dmu - max(kept+, kept-) >= 0.  (i) With lam = 1e-3 on slacks of 1 the safety distance of a stage that has
             * lost its active hinge terms (re-sorted slots) has next to no curvature (H77 = 2e-3) against the gradient -slack_gain: the first
             * Newton step asks for |dd| ~ 4e3 and is cut to 3e-4 of its length - a lost iteration (C4: in every solve).  (ii) Both slacks of the
             * pair are floored alike (the box is narrower than the floor), so lam+ - lam- and with it the dual residual of the first iterate
             * stay those of the kept multipliers: the key su_hardlike does not see the start it follows (a one-sided max(kept, dmu / w) did -
             * iter_num = 1: 1.0 -> 3.0 iterations per solve, the easy start locked out). */
            if (hard_dmu > 0 && i >= 8 * T - 4) {
                const int j = i - (8 * T - 4), t = j / 2, s0 = 8 * T - 4 + 2 * (warm_shift ? (t + 1 > T - 1 ? T - 1 : t + 1) : t);
                const double km = fmax(lam_keep[s0], lam_keep[s0 + 1]), dl_ = hard_dmu - km;
                if (dl_ > 0) lm[i] += dl_ / w[i];
            }
        }
    }
    status = 1; land_failed = 0;
    double mu_prev = 1.0, al_prev = 0.0;
    for (it = 0; it < it_cap; ++it) {
        /* ... and in that rescue phase the hinge terms are smoothed over a width eps = SU_SMOOTH_K sqrt(mu) (mu of the previous iterate;
         * -> 0 with the complementarity: 3e-6 at the stop, 2e-8 in the controls): the other cycle of the semismooth iteration is a hinge
         * term that switches on and off - 5 <-> 6 active terms, period 3, steps of 0.005 along a direction of length 1 (soak of the
         * interior-point LamMuZ mode, scene 56 step 61: 200 iterations / status 1 on both sides -> 37 iterations; another problem of
         * that loop 79 -> 42) */
        t_su_eps = (attempt >= 0 && it >= SU_CENTRE_FROM) ? SU_SMOOTH_K * sqrt(mu_prev) : 0.0;
        su_eval(&S, x, s, grad, Hm);
        t_su_eps = 0.0;
        double gn = 0, rdn = 0, rpn = 0, mu = 0;
        for (int i = 0; i < n; ++i) { if (fabs(grad[i]) > gn) gn = fabs(grad[i]); rhs[i] = grad[i]; }
        for (int i = 0; i < mc; ++i) {
            rhs[con[i].i1] += con[i].c1 * lm[i]; if (con[i].i2 >= 0) rhs[con[i].i2] += con[i].c2 * lm[i];
            double cx = con[i].c1 * x[con[i].i1] + (con[i].i2 >= 0 ? con[i].c2 * x[con[i].i2] : 0);
            rp[i] = cx + w[i] - con[i].e; if (fabs(rp[i]) > rpn) rpn = fabs(rp[i]);
            mu += lm[i] * w[i];
        }
        mu /= mc; mu_prev = mu;
        for (int i = 0; i < n; ++i) if (fabs(rhs[i]) > rdn) rdn = fabs(rhs[i]);
        double sc = 1 + gn;
        if (used == 0 && it == 0) t_su_rd0 = rdn / sc;
#ifdef ORC_DEBUG
        fprintf(stderr, "it %d rdn %.3e rpn %.3e mu %.3e sc %.3e\n", it, rdn, rpn, mu, sc);
#endif
        /* (second clause) past that complementarity the barrier weights lam/w (1e10 and more) put rounding noise into the dual residual: a
         * point that is primal feasible and complementary to 1e-12 is accepted with the residual the arithmetic can deliver */
#ifndef SU_LAND_FALLBACK
#define SU_LAND_FALLBACK 1e-3
#endif
#define SU_CONV(tol) ((rdn <= (tol)[0] * sc && rpn <= (tol)[1] && mu <= (tol)[2] * sc) || (rdn <= 100 * (tol)[0] * sc && rpn <= (tol)[1] && mu <= 0.1 * (tol)[2] * sc))
        if (g_su_land && land_failed < 99) {
            /* Landing (round 6, see su_land): the interior point only has to get close enough for the active set to be read off - su_land_tol - and the
             * vertex is then computed exactly.  Refused (the active-set rounds can cycle while borderline rows are undecided): tried again at 1e-2 x
             * su_land_tol, 1e-4 x ... down to su_tol itself, then the iterate at su_tol is returned as without the landing. */
            double lt[3], lsc = 1.0; int last = 1;
            for (int k = 0; k < land_failed && k < 8; ++k) lsc *= 1e-2;
            for (int k = 0; k < 3; ++k) { lt[k] = lsc * g_su_land_tol[k]; if (lt[k] <= g_su_tol[k]) lt[k] = g_su_tol[k]; else last = 0; }
            if (SU_CONV(lt)) {
                memcpy(x_sav, x, sizeof(double) * n); memcpy(lm_sav, lm, sizeof(double) * mc); memcpy(w_sav, w, sizeof(double) * mc);
                if (su_land(&S, con, mc, n, x, lm, w, s, grad, Hm, K, g_su_tol[0])) { t_su_landed = 1; status = 0; break; }
                memcpy(x, x_sav, sizeof(double) * n); memcpy(lm, lm_sav, sizeof(double) * mc); memcpy(w, w_sav, sizeof(double) * mc);
                land_failed = last ? 99 : land_failed + 1;
                su_eval(&S, x, s, grad, Hm);
            }
        }
        /* every landing refused: the fallback is the interior point itself, and it is run SU_LAND_FALLBACK x tighter than su_tol - where the su-problem is nearly
         * singular (the steering of an Ackermann robot at v ~ 0) the point at su_tol is 1e-5 .. 1e-4 from the vertex, one or two iterations later 1e-7 .. 1e-8
         * (mirrors csrc/su_device.h).  Not reached within the cap: the safety net below has the iterate that met su_tol. */
        {
            const double ft[3] = { SU_LAND_FALLBACK * g_su_tol[0], SU_LAND_FALLBACK * g_su_tol[1], SU_LAND_FALLBACK * g_su_tol[2] };
            if ((!g_su_land && SU_CONV(g_su_tol)) || (g_su_land && land_failed >= 99 && SU_CONV(ft))) { status = 0; break; }
        }
        if (g_su_accept && rpn <= g_su_tol[1] && rdn <= 10 * g_su_tol[0] * sc && mu <= 1e3 * g_su_tol[2] * sc) {
            const double merit = fmax(rdn / (g_su_tol[0] * sc), mu / (g_su_tol[2] * sc));
            if (!have_acc || merit < acc_merit) { memcpy(x_acc, x, sizeof(double) * n); memcpy(lm_acc, lm, sizeof(double) * mc); have_acc = 1; acc_merit = merit; }
        }
        /* K = H + C' diag(lm/w) C */
        memcpy(K, Hm, sizeof(double) * n * n);
        for (int i = 0; i < mc; ++i) {
            double dgn = lm[i] / w[i]; int i1 = con[i].i1, i2 = con[i].i2;
            K[i1 * n + i1] += dgn * con[i].c1 * con[i].c1;
            if (i2 >= 0) { K[i2 * n + i2] += dgn * con[i].c2 * con[i].c2; K[i1 * n + i2] += dgn * con[i].c1 * con[i].c2; K[i2 * n + i1] += dgn * con[i].c1 * con[i].c2; }
        }
        if (chol_factor(K, n)) { status = 2; break; }
        double sigma = 0, mu_aff = 0;
        /* Last resort (attempt 1, round 5): plain long-step path following - no predictor, a fixed centring parameter (smaller once the steps
         * are nearly full), a shorter fraction to the boundary, and every pair kept in the wide neighbourhood lam w >= SU_SAFE_GAMMA mu after
         * each step.  Mehrotra's heuristics can cycle on this problem class (two rows trading places with steps of 0.02 / 0.6 for ever:
         * tests/golden/su_hard/omni_T15_N51_rate_and_distance_rows_cycle.npz - both former attempts ran into their caps); this iteration has the
         * textbook guarantee and is only reached when the two others have failed. */
        const int safe = attempt == 1;
        if (safe) sigma = al_prev >= 0.9 ? SU_SAFE_SIGMA_END : SU_SAFE_SIGMA;
        for (int pass = safe ? 1 : 0; pass < 2; ++pass) {
            /* rc = lm*w (+ corrector) - sigma*mu */
            for (int i = 0; i < mc; ++i) rc[i] = lm[i] * w[i] + (pass ? (safe ? 0.0 : dl[i] * dw[i]) - sigma * mu : 0.0);
            for (int i = 0; i < n; ++i) dx[i] = -rhs[i];
            for (int i = 0; i < mc; ++i) {
                double v = (lm[i] * rp[i] - rc[i]) / w[i];
                dx[con[i].i1] -= con[i].c1 * v; if (con[i].i2 >= 0) dx[con[i].i2] -= con[i].c2 * v;
            }
            chol_solve(K, n, dx);
            for (int i = 0; i < mc; ++i) {
                double cdx = con[i].c1 * dx[con[i].i1] + (con[i].i2 >= 0 ? con[i].c2 * dx[con[i].i2] : 0);
                dw[i] = -rp[i] - cdx; dl[i] = -(rc[i] + lm[i] * dw[i]) / w[i];
            }
            if (pass == 0) {
                double al = 1.0;
                for (int i = 0; i < mc; ++i) {
                    if (dw[i] < 0 && -w[i] / dw[i] < al) al = -w[i] / dw[i];
                    if (dl[i] < 0 && -lm[i] / dl[i] < al) al = -lm[i] / dl[i];
                }
                mu_aff = 0; for (int i = 0; i < mc; ++i) mu_aff += (lm[i] + al * dl[i]) * (w[i] + al * dw[i]);
                mu_aff /= mc;
                /* centering parameter from the predictor step length, floored: the classical (mu_aff/mu)^3
                 * rule can cycle on the piecewise-quadratic hinge terms (observed with ro1 = 1) */
                { double q = 1 - al, fl = al >= 0.95 ? (attempt < 0 ? cur_warm_sig : SIGMA_FLOOR) : 0.03;
                if (it >= 25) fl = it >= 50 ? 0.3 : 0.1;      /* a solve that is still running is cycling: centre harder */
                sigma = q * q * q; if (sigma < fl) sigma = fl; }
            }
        }
        /* fraction to the boundary: 0.995 far from the solution, -> 1 with the complementarity (superlinear end game) */
        const double tau_min = attempt < 0 ? cur_warm_tau : 0.995;
        double al = 1.0, tau = 1.0 - mu; if (tau < tau_min) tau = tau_min;
        if (safe) tau = SU_SAFE_TAU;
        for (int i = 0; i < mc; ++i) {
            if (dw[i] < 0 && -tau * w[i] / dw[i] < al) al = -tau * w[i] / dw[i];
            if (dl[i] < 0 && -tau * lm[i] / dl[i] < al) al = -tau * lm[i] / dl[i];
        }
        if (g_su_trace == 2 && g_su_dump) {      /* (tools/experiments/scan_riccati.py) the iterate the factorisation of this iteration belongs to */
            double hd[3] = { (double)it, (double)mc, (double)n };
            fwrite(hd, sizeof(double), 3, g_su_dump); fwrite(x, sizeof(double), n, g_su_dump); fwrite(w, sizeof(double), mc, g_su_dump); fwrite(lm, sizeof(double), mc, g_su_dump);
        }
        if (g_su_trace == 1) {       /* residuals of this iterate, the step taken from it, active hinge terms, the row that blocks the step */
            int nact = 0, blk = -1; double dxn = 0, bal = 2;
            for (int t = 0; t < T; ++t) for (int nn = 0; nn < N; ++nn) {
                const double *aa = &a[(nn * T + t) * 2];
                nact += aa[0] * s[t + 1] + aa[1] * s[(T + 1) + t + 1] - cc[nn * T + t] - x[2 * T + t] < 0;
            }
            for (int i = 0; i < n; ++i) if (fabs(dx[i]) > dxn) dxn = fabs(dx[i]);
            for (int i = 0; i < mc; ++i) {
                if (dw[i] < 0 && -w[i] / dw[i] < bal) { bal = -w[i] / dw[i]; blk = i; }
                if (dl[i] < 0 && -lm[i] / dl[i] < bal) { bal = -lm[i] / dl[i]; blk = -i - 1; }
            }
            fprintf(stderr, "it %2d rd %.2e rp %.2e mu %.2e sigma %.1e step %.4f active hinges %4d |dx| %.2e blocked by %s %d\n",
                    it, rdn, rpn, mu, sigma, al, nact, dxn, blk < 0 ? "multiplier" : "slack", blk < 0 ? -blk - 1 : blk);
        }
        for (int i = 0; i < n; ++i) x[i] += al * dx[i];
        for (int i = 0; i < mc; ++i) { w[i] += al * dw[i]; lm[i] += al * dl[i]; }
        /* A cold attempt that is still running after SU_CENTRE_FROM iterations is cycling (same cue as the centring floor above): from
         * there on every pair is kept inside a (very) wide neighbourhood of the central path, lam_i w_i >= SU_CENTRE_GAMMA mu after the
         * step, by raising the multiplier.  The cycle it breaks: two neighbouring rate rows, both active at the solution, trade places
         * for ever - one iterate has w_63 = 1e-8 with lam_63 w_63 / mu = 4e-7, the next one the same for row 66, mu stays at 3e-6 and
         * the controls jump by 1e-2 (soak scene 30, step 39: the cold attempt AND the central restart ran into their caps;
         * tests/golden/su_hard/acker_T15_N45_rate_rows_cycle.npz: 200 iterations and status 1 -> 32 iterations).  Solves that end
         * earlier - all of a recorded C4 / north-star closed loop (tools/su_replay.py) - are untouched. */
        if ((attempt >= 0 && it >= SU_CENTRE_FROM) || safe) {
            const double gam = safe ? SU_SAFE_GAMMA : SU_CENTRE_GAMMA;
            double mun = 0; for (int i = 0; i < mc; ++i) mun += lm[i] * w[i];
            mun /= mc;
            for (int i = 0; i < mc; ++i) if (lm[i] * w[i] < gam * mun) lm[i] = gam * mun / w[i];
        }
        al_prev = al;
    }
    used += it;
    }
    /* every attempt failed: the safety net (see above) */
    if ((status != 0 || g_su_accept == 2) && have_acc) { memcpy(x, x_acc, sizeof(double) * n); memcpy(lm, lm_acc, sizeof(double) * mc); status = 0; }
    if (ipm_iters) *ipm_iters = used;
    if (status == 0 && lam_keep) memcpy(lam_keep, lm, sizeof(double) * mc);
    su_rollout(&S, x, s);
    memcpy(s_out, s, sizeof(double) * 3 * (T + 1));
    for (int t = 0; t < T; ++t) { u_out[t] = x[2 * t]; u_out[T + t] = x[2 * t + 1]; d_out[t] = x[2 * T + t]; }
    free(S.Ak); free(S.Bk); free(S.Ck); free(S.Gam); free(S.Q0); free(S.Q1); free(S.Q2);
    free(con); free(x); free(grad); free(Hm); free(K); free(rhs); free(dx); free(w); free(lm); free(rp); free(dw); free(dl); free(rc); free(s); free(x_acc); free(lm_acc); free(x_sav); free(lm_sav); free(w_sav);
    return status;
}

int orc_su_solve(const orc_cfg *c, const double *nom_s, const double *nom_u, const double *ref_s,
                 double ref_speed, const double *a, const double *cc, const double *g,
                 const double *d0, double *s_out, double *u_out, double *d_out, int *ipm_iters)
{
    return su_solve_impl(c, nom_s, nom_u, ref_s, ref_speed, a, cc, g, d0, s_out, u_out, d_out, ipm_iters, NULL, 0, 0, 0, 0, 0, 0);
}

/* ------------------------------------------------------------------------------------------ */
int orc_create(const orc_cfg *cfg, const double *G, const double *h, orc_handle **out)
{
    if (cfg->E > EMAX || cfg->R > RMAX || cfg->N < 1 || cfg->T < 1) return -1;
    if (cfg->robot_norm2 && !g_lmz_mode) return -2;      /* the enumeration has no norm2 robot candidates: interior-point mode only */
    orc_handle *H = calloc(1, sizeof(*H));
    H->c = *cfg;
    int T = cfg->T, N = cfg->N, E = cfg->E, R = cfg->R;
    H->G = malloc(sizeof(double) * 2 * R); memcpy(H->G, G, sizeof(double) * 2 * R);
    H->h = malloc(sizeof(double) * R); memcpy(H->h, h, sizeof(double) * R);
    H->lam = calloc((size_t)N * (T + 1) * E, sizeof(double));
    H->mu = calloc((size_t)N * (T + 1) * R, sizeof(double));
    H->z = calloc((size_t)N * T, sizeof(double));
    H->xi = calloc((size_t)N * (T + 1) * 2, sizeof(double));
    H->zeta = calloc((size_t)N * T, sizeof(double));
    H->dis = malloc(sizeof(double) * T); for (int t = 0; t < T; ++t) H->dis[t] = 1.0;   /* rda_solver.py:119 */
    H->a_lam = calloc((size_t)N * (T + 1) * 2, sizeof(double));
    H->b_lam = calloc((size_t)N * (T + 1), sizeof(double));
    H->A = calloc((size_t)N * (T + 1) * E * 2, sizeof(double));
    H->b = calloc((size_t)N * (T + 1) * E, sizeof(double));
    H->cone = malloc(sizeof(int) * N); for (int n = 0; n < N; ++n) H->cone[n] = 1;        /* rda_solver.py:158 */
    H->s = calloc(3 * (T + 1), sizeof(double)); H->u = calloc(2 * T, sizeof(double));
    H->resp = calloc((size_t)2 * N * T, sizeof(double)); H->ref = calloc(3 * (T + 1), sizeof(double));
    H->P = 1; H->rank = 0; H->Nloc = N; H->chunk = (size_t)8 * T * N; H->gath = NULL; H->have_gath = 0;
    H->su_lam_keep = calloc((size_t)10 * T, sizeof(double)); H->su_last = 99;
    *out = H; return 0;
}
void orc_destroy(orc_handle *H)
{
    if (!H) return;
    free(H->G); free(H->h); free(H->lam); free(H->mu); free(H->z); free(H->xi); free(H->zeta); free(H->dis);
    free(H->a_lam); free(H->b_lam); free(H->A); free(H->b); free(H->cone); free(H->s); free(H->u); free(H->resp); free(H->ref); free(H->gath); free(H->su_lam_keep); free(H);
}
int orc_set_adjust(orc_handle *H, double slack_gain, double max_sd, double min_sd, double ro1, double ro2)
{ H->c.slack_gain = slack_gain; H->c.max_sd = max_sd; H->c.min_sd = min_sd; H->c.ro1 = ro1; H->c.ro2 = ro2; return 0; }
int orc_reset(orc_handle *H)
{
    int T = H->c.T, N = H->c.N;
    for (int n = 0; n < N; ++n) for (int t = 0; t < T; ++t) {
        H->a_lam[(n * (T + 1) + t + 1) * 2] = H->a_lam[(n * (T + 1) + t + 1) * 2 + 1] = 0; H->b_lam[n * (T + 1) + t + 1] = 0;
    }
    /* the solver history that picks the start of the next su-solves goes with it, like k_reset of csrc/rda_hip.hip */
    H->su_last = 99; H->su_probe = 0; H->su_hardlike = 0; H->prev_unconv = 0; memset(H->su_lam_keep, 0, sizeof(double) * 10 * T);
    return 0;
}
/* debug: interior-point iterations of the su-solve of each ADMM iteration of the last step (n <= 32 entries) */
void orc_get_su_ipm_hist(const orc_handle *H, int *out, int n) { for (int i = 0; i < n && i < 32; ++i) out[i] = i < H->iters ? H->ipm_hist[i] : 0; }
int orc_get_state(orc_handle *H, double *lam, double *mu, double *z, double *xi, double *zeta, double *dis, double *a_lam, double *b_lam)
{
    int T = H->c.T, N = H->c.N, E = H->c.E, R = H->c.R;
    if (lam) memcpy(lam, H->lam, sizeof(double) * N * (T + 1) * E);
    if (mu) memcpy(mu, H->mu, sizeof(double) * N * (T + 1) * R);
    if (z) memcpy(z, H->z, sizeof(double) * N * T);
    if (xi) memcpy(xi, H->xi, sizeof(double) * N * (T + 1) * 2);
    if (zeta) memcpy(zeta, H->zeta, sizeof(double) * N * T);
    if (dis) memcpy(dis, H->dis, sizeof(double) * T);
    if (a_lam) memcpy(a_lam, H->a_lam, sizeof(double) * N * (T + 1) * 2);
    if (b_lam) memcpy(b_lam, H->b_lam, sizeof(double) * N * (T + 1));
    return 0;
}
int orc_set_state(orc_handle *H, const double *lam, const double *mu, const double *z, const double *xi, const double *zeta, const double *dis, const double *a_lam, const double *b_lam)
{
    int T = H->c.T, N = H->c.N, E = H->c.E, R = H->c.R;
    if (lam) memcpy(H->lam, lam, sizeof(double) * N * (T + 1) * E);
    if (mu) memcpy(H->mu, mu, sizeof(double) * N * (T + 1) * R);
    if (z) memcpy(H->z, z, sizeof(double) * N * T);
    if (xi) memcpy(H->xi, xi, sizeof(double) * N * (T + 1) * 2);
    if (zeta) memcpy(H->zeta, zeta, sizeof(double) * N * T);
    if (dis) memcpy(H->dis, dis, sizeof(double) * T);
    if (a_lam) memcpy(H->a_lam, a_lam, sizeof(double) * N * (T + 1) * 2);
    if (b_lam) memcpy(H->b_lam, b_lam, sizeof(double) * N * (T + 1));
    return 0;
}

/* assign_obstacle_parameter, rda_solver.py:483-526 */
static void stage_obstacles(orc_handle *H, int n_obs, const double *A, const double *b, const int *cone, int per_t)
{
    int T = H->c.T, N = H->c.N, E = H->c.E;
    H->obstacle_num = n_obs;
    if (n_obs <= 0) return;                      /* nothing written: stale A,b stay */
    int nt = per_t ? T + 1 : 1;
    for (int n = 0; n < N; ++n) {
        int src = n < n_obs ? n : n_obs - 1;     /* pad by duplicating the last obstacle (Q3) */
        for (int t = 0; t <= T; ++t) {
            int ts = per_t ? t : 0;
            memcpy(&H->A[((size_t)(n * (T + 1) + t) * E) * 2], &A[((size_t)(src * nt + ts) * E) * 2], sizeof(double) * E * 2);
            memcpy(&H->b[(size_t)(n * (T + 1) + t) * E], &b[(size_t)(src * nt + ts) * E], sizeof(double) * E);
        }
        H->cone[n] = cone[src];
    }
    H->obstacle_num = N;                          /* rda_solver.py:492 after padding; >N truncates */
}

/* ---- the ADMM loop in its pieces (rda_solver.py:588-637) ------------------------------------------
 * admm_begin; for it: admm_su (early-stop test of the previous iteration, then the su-problem);
 * admm_lammuz (this rank's obstacle shard); [exchange of the shard chunks]; admm_finish.
 * With one shard (P = 1) orc_step runs them back to back. */
static void shard_chunk_of(orc_handle *H, int n, int t, double *o8)
{   /* the 8 numbers per (obstacle, stage) the su-problem and the residual test need */
    const orc_cfg *c = &H->c; int T = c->T, R = c->R;
    if (n >= c->N) { for (int k = 0; k < 8; ++k) o8[k] = 0; o8[3] = -1e30; return; }    /* padding slot of an uneven shard: never an active hinge */
    const double *mu = &H->mu[(size_t)(n * (T + 1) + t + 1) * R];
    double muh = 0, gx = H->xi[(n * (T + 1) + t + 1) * 2], gy = H->xi[(n * (T + 1) + t + 1) * 2 + 1];
    for (int j = 0; j < R; ++j) { muh += mu[j] * H->h[j]; gx += mu[j] * H->G[2 * j]; gy += mu[j] * H->G[2 * j + 1]; }
    o8[0] = H->a_lam[(n * (T + 1) + t + 1) * 2]; o8[1] = H->a_lam[(n * (T + 1) + t + 1) * 2 + 1];
    o8[2] = H->b_lam[n * (T + 1) + t + 1]; o8[3] = muh + H->z[n * T + t] - H->zeta[n * T + t];
    o8[4] = gx; o8[5] = gy; o8[6] = H->resp[2 * (n * T + t)]; o8[7] = H->resp[2 * (n * T + t) + 1];
}

int orc_shard_config(orc_handle *H, int rank, int world)
{
    if (!H || world < 1 || rank < 0 || rank >= world) return -1;
    if (H->c.N % world && !H->c.accelerated) return -2;
    H->P = world; H->rank = rank; H->Nloc = (H->c.N + world - 1) / world; H->chunk = (size_t)8 * H->c.T * H->Nloc;
    free(H->gath); H->gath = calloc(H->chunk * world, sizeof(double)); H->have_gath = 0;
    return 0;
}
int orc_shard_chunk_doubles(orc_handle *H) { return (int)H->chunk; }
int orc_shard_get_chunk(orc_handle *H, double *out)
{   /* layout [k][t][nl], identical to the HIP library */
    int T = H->c.T, Nl = H->Nloc;
    for (int nl = 0; nl < Nl; ++nl) for (int t = 0; t < T; ++t) {
        double o8[8]; shard_chunk_of(H, H->rank * Nl + nl, t, o8);
        for (int k = 0; k < 8; ++k) out[(size_t)k * T * Nl + t * Nl + nl] = o8[k];
    }
    return 0;
}
int orc_shard_set_chunks(orc_handle *H, const double *all)
{ memcpy(H->gath, all, sizeof(double) * H->chunk * H->P); H->have_gath = 1; return 0; }

int orc_admm_begin(orc_handle *H, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed)
{
    int T = H->c.T;
    memcpy(H->s, nom_s, sizeof(double) * 3 * (T + 1)); memcpy(H->u, nom_u, sizeof(double) * 2 * T);
    memcpy(H->ref, ref_s, sizeof(double) * 3 * (T + 1)); H->ref_speed = ref_speed;
    H->stop = 0; H->iters = 0; H->su_status = 0; H->ipm_total = 0; H->lmz_fail = 0; H->resi_dual = 0; H->resi_pri = 0;
    return 0;
}
static void admm_residuals(orc_handle *H)
{
    const orc_cfg *c = &H->c; int T = c->T, N = c->N;
    double rd = 0, rp = 0;
    if (H->obstacle_num != 0) {
        if (H->P > 1 && H->have_gath) {
            for (int r = 0; r < H->P; ++r) for (int nl = 0; nl < H->Nloc; ++nl) for (int t = 0; t < T; ++t) {
                rd += H->gath[r * H->chunk + (size_t)6 * T * H->Nloc + t * H->Nloc + nl];
                rp += H->gath[r * H->chunk + (size_t)7 * T * H->Nloc + t * H->Nloc + nl];
            }
        } else for (int i = 0; i < N * T; ++i) { rd += H->resp[2 * i]; rp += H->resp[2 * i + 1]; }
    }
    H->resi_dual = rd / N; H->resi_pri = sqrt(rp);                                            /* :737,:688 */
}
int orc_admm_su(orc_handle *H, int it, int *stopped)
{
    const orc_cfg *c = &H->c; int T = c->T, N = c->N;
    if (H->stop) { if (stopped) *stopped = 1; return 0; }
    if (it > 0) {
        admm_residuals(H);
        if (H->resi_dual < c->iter_threshold && H->resi_pri < c->iter_threshold) {            /* :594 */
            H->stop = 1; if (stopped) *stopped = 1; return 0;
        }
    }
    double *ca = malloc(sizeof(double) * N * T * 2), *cc = malloc(sizeof(double) * N * T), *cg = malloc(sizeof(double) * N * T * 2);
    double *s_new = malloc(sizeof(double) * 3 * (T + 1)), *u_new = malloc(sizeof(double) * 2 * T), *d_new = malloc(sizeof(double) * T);
    /* ---- su-problem (rda_solver.py:617,692-700) ------------------------------------- */
    for (int n = 0; n < N; ++n) for (int t = 0; t < T; ++t) {
        double o8[8];
        if (H->P > 1 && H->have_gath) {
            int r = n / H->Nloc, nl = n % H->Nloc;
            for (int k = 0; k < 6; ++k) o8[k] = H->gath[r * H->chunk + (size_t)k * T * H->Nloc + t * H->Nloc + nl];
        } else shard_chunk_of(H, n, t, o8);
        ca[(n * T + t) * 2] = o8[0]; ca[(n * T + t) * 2 + 1] = o8[1];
        cc[n * T + t] = o8[2] + o8[3];
        cg[(n * T + t) * 2] = o8[4]; cg[(n * T + t) * 2 + 1] = o8[5];
    }
    int ipm = 0;
    if (g_su_dump) {            /* debug: every su-problem of a run, cfg + orc_su_solve's arguments (tools/su_replay.py replays them) */
        double hd[4] = { (double)T, (double)N, H->ref_speed, (double)it };
        fwrite(c, sizeof(orc_cfg), 1, g_su_dump); fwrite(hd, sizeof(double), 4, g_su_dump);
        fwrite(H->s, sizeof(double), 3 * (T + 1), g_su_dump); fwrite(H->u, sizeof(double), 2 * T, g_su_dump);
        fwrite(H->ref, sizeof(double), 3 * (T + 1), g_su_dump); fwrite(ca, sizeof(double), N * T * 2, g_su_dump); fwrite(cc, sizeof(double), N * T, g_su_dump);
        fwrite(cg, sizeof(double), N * T * 2, g_su_dump); fwrite(H->dis, sizeof(double), T, g_su_dump); fflush(g_su_dump);
    }
    /* ADMM iterations >= 1 start from the multipliers of the previous su-solve of this step, if that one converged */
    /* ... and the first one from those of the previous step, shifted by one stage */
    const int warm = g_su_warm_mu0 > 0 && (it > 0 ? !((H->su_status >> (it - 1)) & 1) : g_su_warm_first);
    /* while the su-solves are easy (the last one took <= max iterations) the warm attempt starts 1e-6 from the previous solution's
     * active bounds and takes near-full steps - the same rule as csrc/rda_hip.hip su_body */
    const int easy = warm && g_su_easy_max > 0 && H->su_last <= g_su_easy_max;
    /* after an unconverged step, and only while consecutive su-problems really are far apart: the last solve's FIRST iterate (the previous
     * solution with its multipliers) had a relative dual residual above SU_HARD_RD0.  (Round 4 keyed on the last solve's iteration count
     * > 3; a hard-started solve costs >= 3 iterations whatever the problem, so that key either locked the easy start out - iter_num = 1:
     * 1.0 -> 3.0 iterations per solve - or, set to > 3, left most of the gain: re-sorted north star 7.6 -> 6.2 against 5.3 with this key.) */
    const int hard = warm && !easy && g_su_hard_mu0 > 0 && H->prev_unconv && H->su_hardlike && H->su_last < 99;
    /* hard regime without that rule (many moving obstacles): a warm attempt needs MORE iterations than a cold start - start cold, probe the warm start now and then */
    const int go_cold = warm && !easy && !hard && g_su_cold_from > 0 && H->su_last > g_su_cold_from && H->su_last < 99 && H->su_probe % g_su_cold_probe != g_su_cold_probe - 1;
    cur_warm_clip = easy ? g_su_easy[2] : g_su_warm_clip; cur_warm_tau = easy ? g_su_easy[3] : g_su_warm_tau; cur_warm_sig = easy ? g_su_easy[4] : g_su_warm_sig;
    double tol_keep[3] = { g_su_tol[0], g_su_tol[1], g_su_tol[2] };
    if (it < c->iter_num - 1 && g_su_tol_early[0] > 0 && g_su_tol_early[1] > 0 && g_su_tol_early[2] > 0) memcpy(g_su_tol, g_su_tol_early, sizeof g_su_tol);
    int st = su_solve_impl(c, H->s, H->u, H->ref, H->ref_speed, ca, cc, cg, H->dis, s_new, u_new, d_new, &ipm,
                           H->su_lam_keep, warm && !go_cold, easy ? g_su_easy[0] : (hard ? g_su_hard_wfl : g_su_warm_wfl), easy ? g_su_easy[1] : (hard ? g_su_hard_mu0 : g_su_warm_mu0), g_su_warm_cap, it == 0, hard ? SU_HARD_DMU : 0.0);
    memcpy(g_su_tol, tol_keep, sizeof g_su_tol);
    H->su_last = st == 0 ? ipm : 99;
    H->su_hardlike = t_su_rd0 > SU_HARD_RD0;
    H->su_probe = (g_su_cold_from > 0 && H->su_last > g_su_cold_from && H->su_last < 99) ? H->su_probe + 1 : 0;
    H->ipm_total += ipm; if (it < 32) H->ipm_hist[it] = ipm;
    if (st == 0) { memcpy(H->s, s_new, sizeof(double) * 3 * (T + 1)); memcpy(H->u, u_new, sizeof(double) * 2 * T); memcpy(H->dis, d_new, sizeof(double) * T); }
    else H->su_status |= 1 << it;                 /* 'No update of state and control vector' :699 */
    H->iters = it + 1;
    free(ca); free(cc); free(cg); free(s_new); free(u_new); free(d_new);
    if (stopped) *stopped = 0;
    return 0;
}
int orc_admm_lammuz(orc_handle *H)
{
    const orc_cfg *c = &H->c; int T = c->T, N = c->N, E = c->E, R = c->R;
    if (H->stop) return 0;
    if (H->obstacle_num == 0) {                   /* Q9: only the last slot is cleared, :564-568 */
        if (H->rank == (N - 1) / H->Nloc) {
            int n = N - 1;
            for (int t = 0; t < T; ++t) { H->a_lam[(n * (T + 1) + t + 1) * 2] = H->a_lam[(n * (T + 1) + t + 1) * 2 + 1] = 0; H->b_lam[n * (T + 1) + t + 1] = 0; }
        }
        return 0;
    }
    /* ---- LamMuZ problems + dual updates of this rank's shard (rda_solver.py:628-635) ----------------- */
    const int n0 = H->rank * H->Nloc, n1 = n0 + H->Nloc < N ? n0 + H->Nloc : N;      /* the last shard may be short (N % P != 0) */
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_threads) schedule(static)
#endif
    for (int n = n0; n < n1; ++n) {
        for (int t = 0; t < T; ++t) {
            size_t o = (size_t)(n * (T + 1) + t + 1);
            const double *At = &H->A[o * E * 2], *bt = &H->b[o * E];
            double p[2] = { H->s[t + 1], H->s[(T + 1) + t + 1] }, phi = H->s[2 * (T + 1) + t];
            double lam[EMAX], mu[RMAX], z, res = 0;
            int fail = 0;
            int finite_in = isfinite(p[0]) && isfinite(p[1]) && isfinite(phi) && isfinite(H->xi[o * 2]) && isfinite(H->xi[o * 2 + 1])
                            && isfinite(H->zeta[n * T + t]) && isfinite(H->dis[t]);
            for (int i = 0; i < E; ++i) finite_in = finite_in && isfinite(At[2 * i]) && isfinite(At[2 * i + 1]) && isfinite(bt[i]);
            if (!finite_in) fail = 2;
            else if (g_lmz_mode) {
                int st = orc_lammuz_ipm_one(E, R, At, bt, H->cone[n], c->robot_norm2, p, phi, H->G, H->h, &H->xi[o * 2], H->zeta[n * T + t],
                                            H->dis[t], c->ro2, c->accelerated, lam, mu, &z, NULL, NULL);
                /* the reference accepts OPTIMAL only (rda_solver.py:781,816; Q8): anything else keeps the previous duals and the
                 * residual of the obstacle becomes inf, which blocks the early stop (:791-793) */
                if (st != 0) { fail = 1; memcpy(lam, &H->lam[o * E], sizeof(double) * E); memcpy(mu, &H->mu[o * R], sizeof(double) * R); z = H->z[n * T + t]; }
            } else {
                double cmh[4] = {0, 0, 0, 0};
                orc_lammuz_one(E, R, At, bt, H->cone[n], p, phi, H->G, H->h, &H->xi[o * 2], H->zeta[n * T + t],
                                              H->dis[t], c->ro2, c->delta, c->accelerated, lam, mu, &z, cmh);
                if (!isfinite(cmh[0]) || !isfinite(cmh[1]) || !isfinite(cmh[2]) || !isfinite(cmh[3])) fail = 2;
            }
            if (fail == 2) {    /* non-finite data or result: previous lam, mu, z, xi, zeta stay, the stage drops out of the su hinge */
                H->a_lam[o * 2] = H->a_lam[o * 2 + 1] = 0; H->b_lam[o] = 0;
                H->resp[2 * (n * T + t)] = INFINITY; H->resp[2 * (n * T + t) + 1] = 0;
#ifdef _OPENMP
#pragma omp atomic
#endif
                H->lmz_fail++;
                continue;
            }
            double cs = cos(phi), sn = sin(phi);
            double ax = 0, ay = 0, bl = 0, im = 0, gx = 0, gy = 0;
            for (int i = 0; i < E; ++i) {
                double dl_ = lam[i] - H->lam[o * E + i]; res += dl_ * dl_; H->lam[o * E + i] = lam[i];
                ax += lam[i] * At[2 * i]; ay += lam[i] * At[2 * i + 1]; bl += lam[i] * bt[i];
            }
            for (int j = 0; j < R; ++j) {
                double dm = mu[j] - H->mu[o * R + j]; res += dm * dm; H->mu[o * R + j] = mu[j];
                im -= mu[j] * H->h[j]; gx += mu[j] * H->G[2 * j]; gy += mu[j] * H->G[2 * j + 1];
            }
            double dz = z - H->z[n * T + t]; res += dz * dz; H->z[n * T + t] = z;
            H->a_lam[o * 2] = ax; H->a_lam[o * 2 + 1] = ay; H->b_lam[o] = bl;          /* :541-542 */
            double hx = gx + cs * ax + sn * ay, hy = gy - sn * ax + cs * ay;          /* :682 */
            H->xi[o * 2] += hx; H->xi[o * 2 + 1] += hy;                               /* :683 */
            im += ax * p[0] + ay * p[1] - bl;                                         /* :659 */
            H->zeta[n * T + t] += im - H->dis[t] - z;                                 /* :666 */
            H->resp[2 * (n * T + t)] = fail ? INFINITY : res; H->resp[2 * (n * T + t) + 1] = hx * hx + hy * hy;
            if (fail) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                H->lmz_fail++;
            }
        }
    }
    H->have_gath = 0;       /* the gathered copy is stale until the next exchange */
    return 0;
}
int orc_lmz_failures(orc_handle *H) { return H ? H->lmz_fail : -1; }
int orc_admm_finish(orc_handle *H, double *out_u, double *out_s, orc_info *info)
{
    int T = H->c.T;
    if (!H->stop) admm_residuals(H);
    H->prev_unconv = !(H->resi_dual < H->c.iter_threshold && H->resi_pri < H->c.iter_threshold);
    memcpy(out_u, H->u, sizeof(double) * 2 * T); memcpy(out_s, H->s, sizeof(double) * 3 * (T + 1));
    if (info) { info->resi_dual = H->resi_dual; info->resi_pri = H->resi_pri; info->iters = H->iters; info->su_status = H->su_status; info->su_ipm_iters = H->ipm_total; info->lmz_fail = H->lmz_fail; }
    return 0;
}

int orc_step(orc_handle *H, const double *nom_s, const double *nom_u, const double *ref_s,
             double ref_speed, int n_obs, const double *A, const double *b, const int *cone,
             int per_t, double *out_u, double *out_s, orc_info *info)
{
    if (H->P != 1) return -1;                     /* sharded handles are driven through orc_admm_* */
    stage_obstacles(H, n_obs, A, b, cone, per_t);
    orc_admm_begin(H, nom_s, nom_u, ref_s, ref_speed);
    for (int it = 0; it < H->c.iter_num; ++it) {
        int stopped = 0;
        orc_admm_su(H, it, &stopped);
        if (stopped) break;
        orc_admm_lammuz(H);
    }
    return orc_admm_finish(H, out_u, out_s, info);
}
int orc_upload_obstacles(orc_handle *H, int n_obs, const double *A, const double *b, const int *cone, int per_t)
{ stage_obstacles(H, n_obs, A, b, cone, per_t); return 0; }
