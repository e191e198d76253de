/*
 * ORACLE - test infrastructure only (see rda_oracle.h).  Plain C99, fp64.
 *
 * The LamMuZ sub-problem of ONE (obstacle, stage) solved the way the reference solves it: as the cone program its
 * own construction code builds (rda_solver.py:389-421 `LamMuZ_cost_cons`, :874-909 `Hm_LamMu` / `Im_LamMu`,
 * :1034-1050 the cones) for a one-stage horizon, by a primal-dual interior-point method (Mehrotra predictor-
 * corrector, Nesterov-Todd scaling - the algorithm class of ECOS, rda_solver.py:768,800).  Unlike the support
 * enumeration of `orc_lammuz_one` it returns, where the minimiser is not unique (the "slack regime": some (lam, mu)
 * reaches H = 0 with Im >= 0), an INTERIOR point of the optimal face near its analytic centre - all multipliers
 * positive, ||A'lam|| < 1, z = Im - which is what the su-problem of the reference is fed with.
 *
 * Canonical form (disciplined-convex reductions of the reference's expression tree, the same ones cvxpy applies):
 *   x = [ lam (E) | mu (R) | z | th | tn | mm | (tl) | (tr) ]
 *   minimise  1/2 th^2 + 1/2 ro2 |M'lam + G'mu + xi|^2                    (accelerated: th is the epigraph of neg(Im))
 *             1/2 Im^2 + 1/2 ro2 |...|^2                                  (not accelerated: no th)
 *   s.t.      z >= 0 ; th >= -Im ; th >= 0                                 Im = q'lam - h'mu - z + zeta - dbar
 *             (tn ; A'lam) in Q^3 ; tn <= mm ; mm <= 1                     cp.max(cp.vstack([cp.norm(A'lam_t)])) <= 1, one stage
 *             obstacle cone  Rpositive: lam >= 0     norm2: (tl ; lam_0, lam_1) in Q^3, tl + lam_2 <= 0 (E identical rows)
 *             robot cone     Rpositive: mu >= 0      norm2: (tr ; mu_0..mu_{R-2}) in Q^R, tr + mu_{R-1} <= 0
 * The equality-constrained auxiliary variables Im, Hm of the reference are eliminated (the Newton iterates of the
 * eliminated and the full system coincide).  Deviation from the reference, stated: it couples the T stages of an
 * obstacle through ONE `mm` (max over t of the norms); here every stage has its own.  Column 0 of lam, mu is only
 * cone-constrained in the reference (unbounded optimal face, the value an interior-point solver returns there is
 * arbitrary) and is not touched.
 *
 * The solver is the dense, normal-equations form of the algorithm of oracle/refshim/refshim_coneqp.py (same start,
 * same step rule), stopped at the 1e-8 class tolerances ECOS runs with by default.
 */
#include "rda_oracle.h"
#include <math.h>
#include <string.h>

#define NX 28            /* max variables  */
#define MX 64            /* max cone rows  */
#define KQ 3             /* max second-order cones */

typedef struct {
    int n, l, nq, qd[KQ], m;           /* variables, LP rows, cones, cone dims, total rows */
    double P[NX][NX], q[NX], G[MX][NX], h[MX];
} cqp;

static double jdet(const double *u, int d) { double s = u[0] * u[0]; for (int i = 1; i < d; ++i) s -= u[i] * u[i]; return s; }
static double nrm1(const double *u, int d) { double s = 0; for (int i = 1; i < d; ++i) s += u[i] * u[i]; return sqrt(s); }

static double min_eig(const cqp *c, const double *u)
{
    double v = INFINITY;
    for (int i = 0; i < c->l; ++i) if (u[i] < v) v = u[i];
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) { double e = u[o] - nrm1(u + o, c->qd[k]); if (e < v) v = e; }
    return v;
}
static void jprod(const cqp *c, const double *u, const double *v, double *o_)
{
    for (int i = 0; i < c->l; ++i) o_[i] = u[i] * v[i];
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) {
        int d = c->qd[k]; double s = 0;
        for (int i = 0; i < d; ++i) s += u[o + i] * v[o + i];
        for (int i = 1; i < d; ++i) o_[o + i] = u[o] * v[o + i] + v[o] * u[o + i];
        o_[o] = s;
    }
}
static void jdiv(const cqp *c, const double *lam, const double *b, double *o_)      /* lam o u = b */
{
    for (int i = 0; i < c->l; ++i) o_[i] = b[i] / lam[i];
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) {
        int d = c->qd[k]; const double *l = lam + o, *bb = b + o;
        double det = jdet(l, d), l1b1 = 0;
        for (int i = 1; i < d; ++i) l1b1 += l[i] * bb[i];
        o_[o] = (l[0] * bb[0] - l1b1) / det;
        for (int i = 1; i < d; ++i) o_[o + i] = (-l[i] * bb[0] + (det * bb[i] + l[i] * l1b1) / l[0]) / det;
    }
}
typedef struct { double d[MX]; double beta[KQ]; double w[KQ][NX]; } nt_scaling;
static int nt_compute(const cqp *c, const double *s, const double *z, nt_scaling *W)
{
    for (int i = 0; i < c->l; ++i) { if (!(s[i] > 0) || !(z[i] > 0)) return 1; W->d[i] = sqrt(s[i] / z[i]); }
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) {
        int d = c->qd[k];
        double ds = jdet(s + o, d), dz = jdet(z + o, d);
        if (!(ds > 0) || !(dz > 0) || !(s[o] > 0) || !(z[o] > 0)) return 1;
        double ns = sqrt(ds), nz = sqrt(dz), g = 0;
        for (int i = 0; i < d; ++i) g += (s[o + i] / ns) * (z[o + i] / nz);
        g = sqrt(0.5 * (1.0 + g));
        W->w[k][0] = (s[o] / ns + z[o] / nz) / (2 * g);
        for (int i = 1; i < d; ++i) W->w[k][i] = (s[o + i] / ns - z[o + i] / nz) / (2 * g);
        W->beta[k] = sqrt(ns / nz);
    }
    return 0;
}
static void nt_apply(const cqp *c, const nt_scaling *W, const double *u, double *o_, int inverse)
{
    for (int i = 0; i < c->l; ++i) o_[i] = inverse ? u[i] / W->d[i] : u[i] * W->d[i];
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) {
        int d = c->qd[k]; const double *w = W->w[k]; double sg = inverse ? -1.0 : 1.0, w1u1 = 0;
        for (int i = 1; i < d; ++i) w1u1 += w[i] * u[o + i];
        double r0 = w[0] * u[o] + sg * w1u1, f = w1u1 / (1.0 + w[0]);
        double sc = inverse ? 1.0 / W->beta[k] : W->beta[k];
        for (int i = 1; i < d; ++i) o_[o + i] = (sg * w[i] * u[o] + u[o + i] + w[i] * f) * sc;
        o_[o] = r0 * sc;
    }
}
static double max_step(const cqp *c, const double *u, const double *du)
{
    double a = INFINITY;
    for (int i = 0; i < c->l; ++i) if (du[i] < 0) { double t = -u[i] / du[i]; if (t < a) a = t; }
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) {
        int d = c->qd[k];
        double qa = jdet(du + o, d), cc = jdet(u + o, d), b = u[o] * du[o];
        for (int i = 1; i < d; ++i) b -= u[o + i] * du[o + i];
        if (fabs(qa) < 1e-300) { if (b < 0) { double t = -cc / (2 * b); if (t < a) a = t; } }
        else {
            double disc = b * b - qa * cc;
            if (disc >= 0) {
                double t = -(b + copysign(sqrt(disc), b)), r1 = t / qa, r2 = t != 0 ? cc / t : INFINITY;
                if (r1 > 0 && r1 < a) a = r1;
                if (r2 > 0 && r2 < a) a = r2;
            }
        }
        if (du[o] < 0) { double t = -u[o] / du[o]; if (t < a) a = t; }
    }
    return a;
}
static int chol(double H[NX][NX], int n)
{
    for (int j = 0; j < n; ++j) {
        double d = H[j][j];
        for (int k = 0; k < j; ++k) d -= H[j][k] * H[j][k];
        if (!(d > 0)) return 1;
        d = sqrt(d); H[j][j] = d;
        for (int i = j + 1; i < n; ++i) { double v = H[i][j]; for (int k = 0; k < j; ++k) v -= H[i][k] * H[j][k]; H[i][j] = v / d; }
    }
    return 0;
}
static void chol_solve2(double H[NX][NX], int n, double *b)
{
    for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= H[i][k] * b[k]; b[i] = v / H[i][i]; }
    for (int i = n - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < n; ++k) v -= H[k][i] * b[k]; b[i] = v / H[i][i]; }
}

/* returns 0 optimal, 1 inaccurate (<= 1e-6), 2 failed; x, z, s hold the best iterate */
static int cqp_solve(const cqp *c, double tol, double mu_target, double *x, double *z, double *s, int *iters_out)
{
    const int n = c->n, m = c->m;
    double e[MX]; memset(e, 0, sizeof(e));
    for (int i = 0; i < c->l; ++i) e[i] = 1;
    for (int k = 0, o = c->l; k < c->nq; o += c->qd[k], ++k) e[o] = 1;
    const int deg = c->l + c->nq;
    static const double REG = 1e-11;
    double H[NX][NX], rhs[NX];
    /* start: W = I */
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = c->P[i][j]; for (int r = 0; r < m; ++r) v += c->G[r][i] * c->G[r][j]; H[i][j] = v + (i == j ? REG : 0); }
    if (chol(H, n)) return 2;
    for (int i = 0; i < n; ++i) { double v = -c->q[i]; for (int r = 0; r < m; ++r) v += c->G[r][i] * c->h[r]; x[i] = v; }
    chol_solve2(H, n, x);
    for (int r = 0; r < m; ++r) { double v = -c->h[r]; for (int i = 0; i < n; ++i) v += c->G[r][i] * x[i]; z[r] = v; s[r] = -v; }
    { double ns = 0, nz = 0; for (int r = 0; r < m; ++r) { ns += s[r] * s[r]; nz += z[r] * z[r]; } ns = sqrt(ns); nz = sqrt(nz);
      double ts = -min_eig(c, s); if (ts >= -1e-8 * (ns > 1 ? ns : 1)) for (int r = 0; r < m; ++r) s[r] += (1 + ts) * e[r];
      double tz = -min_eig(c, z); if (tz >= -1e-8 * (nz > 1 ? nz : 1)) for (int r = 0; r < m; ++r) z[r] += (1 + tz) * e[r]; }
    double nq = 1, nh = 1;
    for (int i = 0; i < n; ++i) if (1 + fabs(c->q[i]) > nq) nq = 1 + fabs(c->q[i]);
    for (int r = 0; r < m; ++r) if (1 + fabs(c->h[r]) > nh) nh = 1 + fabs(c->h[r]);
    double best = INFINITY, bx[NX], bz[MX], bs[MX];
    int status = 2, it;
    for (it = 0; it < 60; ++it) {
        double rx[NX], rz[MX], gap = 0, pc = 0, dres = 0, pres = 0;
        for (int i = 0; i < n; ++i) { double v = c->q[i], pv = 0; for (int j = 0; j < n; ++j) pv += c->P[i][j] * x[j]; for (int r = 0; r < m; ++r) v += c->G[r][i] * z[r];
                                      rx[i] = v + pv; pc += x[i] * (0.5 * pv + c->q[i]); if (fabs(rx[i]) > dres) dres = fabs(rx[i]); }
        for (int r = 0; r < m; ++r) { double v = s[r] - c->h[r]; for (int i = 0; i < n; ++i) v += c->G[r][i] * x[i]; rz[r] = v; if (fabs(v) > pres) pres = fabs(v); gap += s[r] * z[r]; }
        dres /= nq; pres /= nh;
        double relgap = gap / (fabs(pc) > 1 ? fabs(pc) : 1), meas = dres > pres ? dres : pres; if (relgap > meas) meas = relgap;
        if (meas < best) { best = meas; memcpy(bx, x, sizeof(double) * n); memcpy(bz, z, sizeof(double) * m); memcpy(bs, s, sizeof(double) * m); }
        if (mu_target <= 0 && dres <= tol && pres <= tol && relgap <= tol) { status = 0; break; }
        nt_scaling W;
        if (nt_compute(c, s, z, &W)) break;
        double lam[MX]; nt_apply(c, &W, z, lam, 0);
        /* With mu_target > 0 the answer is the point OF THE CENTRAL PATH at that barrier parameter (s o z = mu_target e, zero
         * residuals) instead of "the iterate at which the gap test fires": once the complementarity has come down to it the
         * iteration switches from predictor-corrector steps to pure centring steps.  Along an optimal face that is not a
         * single point the curvature of the Newton system is of the order mu / slack^2, so the iterates of an ever smaller mu
         * drift along the face with the rounding errors; at a fixed mu the centre is a well-conditioned function of the data. */
        const int centring = mu_target > 0 && gap / deg <= 10 * mu_target;
        if (centring) {
            double lc[MX], cent = 0; jprod(c, lam, lam, lc);
            for (int r = 0; r < m; ++r) { double v = fabs(lc[r] - mu_target * e[r]); if (v > cent) cent = v; }
            if (dres <= 1e-10 && pres <= 1e-10 && cent <= 1e-7 * mu_target) { status = 0; break; }
        }
        /* H = P + G' W^-2 G : rows scaled by W^-1 first (GW = W^-1 G, column by column) */
        double GW[MX][NX];
        { double col[MX], o_[MX];
          for (int i = 0; i < n; ++i) { for (int r = 0; r < m; ++r) col[r] = c->G[r][i]; nt_apply(c, &W, col, o_, 1); for (int r = 0; r < m; ++r) GW[r][i] = o_[r]; } }
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = c->P[i][j]; for (int r = 0; r < m; ++r) v += GW[r][i] * GW[r][j]; H[i][j] = v + (i == j ? REG : 0); }
        if (chol(H, n)) break;
        double ll[MX]; jprod(c, lam, lam, ll);
        double dxa[NX], dza[MX], dsa[MX], dx[NX], dz[MX], ds[MX];
        double sigma = 0, mu = gap / deg;
        int bad = 0;
        if (centring) { sigma = 1.0; mu = mu_target; }
        for (int pass = centring ? 1 : 0; pass < 2 && !bad; ++pass) {
            double bsv[MX], u[MX], wu[MX], t[MX], wt[MX], sc = (pass && !centring) ? 1 - sigma : 1.0;
            if (!pass) for (int r = 0; r < m; ++r) bsv[r] = -ll[r];
            else if (centring) for (int r = 0; r < m; ++r) bsv[r] = -ll[r] + mu * e[r];
            else {
                double a1[MX], a2[MX], pr[MX];
                nt_apply(c, &W, dsa, a1, 1); nt_apply(c, &W, dza, a2, 0); jprod(c, a1, a2, pr);
                for (int r = 0; r < m; ++r) bsv[r] = -ll[r] - pr[r] + sigma * mu * e[r];
            }
            jdiv(c, lam, bsv, u);
            nt_apply(c, &W, u, wu, 0);
            for (int r = 0; r < m; ++r) t[r] = -sc * rz[r] - wu[r];          /* t = bz - W u */
            nt_apply(c, &W, t, wt, 1);                                         /* W^-1 t */
            for (int i = 0; i < n; ++i) { double v = -sc * rx[i]; for (int r = 0; r < m; ++r) v += GW[r][i] * wt[r]; rhs[i] = v; }
            chol_solve2(H, n, rhs);
            double *pdx = pass ? dx : dxa, *pdz = pass ? dz : dza, *pds = pass ? ds : dsa;
            for (int i = 0; i < n; ++i) pdx[i] = rhs[i];
            double gd[MX], v1[MX];
            for (int r = 0; r < m; ++r) { double v = -wt[r]; for (int i = 0; i < n; ++i) v += GW[r][i] * pdx[i]; gd[r] = v; }     /* W^-1 (G dx - t) */
            nt_apply(c, &W, gd, pdz, 1);                                                                                           /* dz = W^-2 (G dx - t) */
            for (int r = 0; r < m; ++r) v1[r] = u[r] - gd[r];                                                                      /* u - W dz */
            nt_apply(c, &W, v1, pds, 0);
            for (int r = 0; r < m; ++r) if (!isfinite(pdz[r]) || !isfinite(pds[r])) bad = 1;
            if (!pass && !bad) {
                double aa = max_step(c, s, dsa), ab = max_step(c, z, dza); if (ab < aa) aa = ab; if (aa > 1) aa = 1;
                sigma = (1 - aa) * (1 - aa) * (1 - aa);
            }
        }
        if (bad) break;
        double a = max_step(c, s, ds), a2 = max_step(c, z, dz); if (a2 < a) a = a2; a *= 0.99; if (a > 1) a = 1;
        if (!(a > 0) || !isfinite(a)) break;
        for (int i = 0; i < n; ++i) x[i] += a * dx[i];
        for (int r = 0; r < m; ++r) { z[r] += a * dz[r]; s[r] += a * ds[r]; }
    }
    if (iters_out) *iters_out = it;
    if (status != 0 && best < INFINITY) {
        memcpy(x, bx, sizeof(double) * n); memcpy(z, bz, sizeof(double) * m); memcpy(s, bs, sizeof(double) * m);
        status = best <= 1e-6 ? 1 : 2;
    }
    return status;
}

static double g_ipm_tol = 1e-8, g_ipm_mu = 1e-6;
void orc_set_lmz_ipm_tol(double tol) { if (tol > 0) g_ipm_tol = tol; }
void orc_set_lmz_ipm_mu(double mu) { g_ipm_mu = mu; }      /* > 0: return the central-path point at this barrier parameter; 0: stop by the gap test */

/* One (obstacle, stage) sub-problem by the interior-point method.  Arguments as orc_lammuz_one plus robot_norm2.
 * cmh = (optimal value, Im, H0, H1).  Returns 0 optimal, 1 inaccurate, 2 failed (outputs untouched), <0 bad argument. */
int orc_lammuz_ipm_one(int E, int R, const double *A, const double *b, int cone_norm2, int robot_norm2,
                       const double *p, double phi, const double *G, const double *h,
                       const double *xi, double zeta, double dbar, double ro2, int accelerated,
                       double *lam_out, double *mu_out, double *z_out, double *cmh, int *iters)
{
    if (E < 1 || R < 1 || E > 8 || R > 8 || (cone_norm2 && E < 3) || (robot_norm2 && R < 2)) return -1;
    for (int i = 0; i < E; ++i) if (!isfinite(A[2 * i]) || !isfinite(A[2 * i + 1]) || !isfinite(b[i])) return 2;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(phi) || !isfinite(xi[0]) || !isfinite(xi[1]) || !isfinite(zeta) || !isfinite(dbar)) return 2;
    cqp c; memset(&c, 0, sizeof(c));
    const double cs = cos(phi), sn = sin(phi), kap = zeta - dbar;
    double q[8], M[8][2];
    for (int i = 0; i < E; ++i) {
        q[i] = A[2 * i] * p[0] + A[2 * i + 1] * p[1] - b[i];
        M[i][0] = A[2 * i] * cs + A[2 * i + 1] * sn; M[i][1] = -A[2 * i] * sn + A[2 * i + 1] * cs;
    }
    /* variable layout */
    const int iz = E + R, ith = accelerated ? iz + 1 : -1, itn = iz + 1 + (accelerated ? 1 : 0), imm = itn + 1;
    int n = imm + 1;
    const int itl = cone_norm2 ? n++ : -1, itr = robot_norm2 ? n++ : -1;
    c.n = n;
    /* cost: 1/2 ro2 |B x + xi|^2,  B = [M' | G' | 0] */
    double B[2][NX]; memset(B, 0, sizeof(B));
    for (int i = 0; i < E; ++i) { B[0][i] = M[i][0]; B[1][i] = M[i][1]; }
    for (int j = 0; j < R; ++j) { B[0][E + j] = G[2 * j]; B[1][E + j] = G[2 * j + 1]; }
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) c.P[i][j] = ro2 * (B[0][i] * B[0][j] + B[1][i] * B[1][j]); c.q[i] = ro2 * (B[0][i] * xi[0] + B[1][i] * xi[1]); }
    /* Im = cv'x + kap */
    double cv[NX]; memset(cv, 0, sizeof(cv));
    for (int i = 0; i < E; ++i) cv[i] = q[i];
    for (int j = 0; j < R; ++j) cv[E + j] = -h[j];
    cv[iz] = -1;
    if (accelerated) c.P[ith][ith] += 1.0;
    else for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) c.P[i][j] += cv[i] * cv[j]; c.q[i] += kap * cv[i]; }
    /* LP rows  G x <= h */
    int r = 0;
    c.G[r][iz] = -1; c.h[r] = 0; ++r;                                                      /* z >= 0 */
    if (accelerated) {
        for (int i = 0; i < n; ++i) c.G[r][i] = -cv[i];
        c.G[r][ith] = -1; c.h[r] = kap; ++r;                                               /* -Im - th <= 0 */
        c.G[r][ith] = -1; c.h[r] = 0; ++r;                                                 /* th >= 0 */
    }
    c.G[r][itn] = 1; c.G[r][imm] = -1; c.h[r] = 0; ++r;                                    /* tn <= mm */
    c.G[r][imm] = 1; c.h[r] = 1; ++r;                                                      /* mm <= 1 */
    /* zero-padded edge rows (A_i = 0, b_i = 0: rda_solver.py:507-508,520-521) carry a multiplier that enters nothing; the reference
     * still constrains it (lam_i >= 0), which leaves it without a central value (its dual is 0) - here it is simply left at 0 */
    if (!cone_norm2) for (int i = 0; i < E; ++i) { if (A[2 * i] == 0 && A[2 * i + 1] == 0 && b[i] == 0) continue; c.G[r][i] = -1; c.h[r] = 0; ++r; }
    else for (int i = 0; i < E; ++i) { c.G[r][itl] = 1; c.G[r][2] = 1; c.h[r] = 0; ++r; }  /* tl + lam_2 <= 0, E identical rows (:1044-1048) */
    if (!robot_norm2) for (int j = 0; j < R; ++j) { c.G[r][E + j] = -1; c.h[r] = 0; ++r; }
    else { c.G[r][itr] = 1; c.G[r][E + R - 1] = 1; c.h[r] = 0; ++r; }                      /* tr + mu_{R-1} <= 0 (:1039) */
    c.l = r;
    /* cones: s = h - G x = (t ; vector) */
    c.nq = 0;
    c.G[r][itn] = -1; ++r;                                                                 /* (tn ; A'lam) */
    for (int k = 0; k < 2; ++k) { for (int i = 0; i < E; ++i) c.G[r][i] = -A[2 * i + k]; ++r; }
    c.qd[c.nq++] = 3;
    if (cone_norm2) { c.G[r][itl] = -1; ++r; c.G[r][0] = 1; ++r; c.G[r][1] = 1; ++r; c.qd[c.nq++] = 3; }      /* (tl ; -lam_0, -lam_1) */
    if (robot_norm2) { c.G[r][itr] = -1; ++r; for (int j = 0; j < R - 1; ++j) { c.G[r][E + j] = 1; ++r; } c.qd[c.nq++] = R; }
    c.m = r;
    double x[NX], z[MX], s[MX];
    int st = cqp_solve(&c, g_ipm_tol, g_ipm_mu, x, z, s, iters);
    if (st == 2) return 2;
    for (int i = 0; i < E; ++i) lam_out[i] = x[i];
    for (int j = 0; j < R; ++j) mu_out[j] = x[E + j];
    if (!cone_norm2) for (int i = 0; i < E; ++i) if (lam_out[i] < 0) lam_out[i] = 0;
    if (!robot_norm2) for (int j = 0; j < R; ++j) if (mu_out[j] < 0) mu_out[j] = 0;
    *z_out = x[iz] > 0 ? x[iz] : 0;
    if (cmh) {
        double Im = kap - *z_out, H0 = xi[0], H1 = xi[1];
        for (int i = 0; i < E; ++i) { Im += q[i] * lam_out[i]; H0 += M[i][0] * lam_out[i]; H1 += M[i][1] * lam_out[i]; }
        for (int j = 0; j < R; ++j) { Im -= h[j] * mu_out[j]; H0 += G[2 * j] * mu_out[j]; H1 += G[2 * j + 1] * mu_out[j]; }
        double hin = accelerated ? (Im < 0 ? Im : 0) : Im;
        cmh[0] = 0.5 * hin * hin + 0.5 * ro2 * (H0 * H0 + H1 * H1); cmh[1] = Im; cmh[2] = H0; cmh[3] = H1;
    }
    return st;
}
