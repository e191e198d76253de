// K1, interior-point variant: one LamMuZ sub-problem (obstacle n, stage t) per THREAD, solved as the cone program the
// reference builds for it (rda_solver.py:389-421 LamMuZ_cost_cons, :874-909 Hm_LamMu / Im_LamMu, :1034-1050 the cones):
//
//   x = [ lam (E) | mu (R) | z | th | tn | mm | (tl) | (tr) ]
//   minimise  1/2 th^2 + 1/2 ro2 |M'lam + G'mu + xi|^2                 (accelerated; otherwise 1/2 Im^2 instead of the th term)
//   s.t.      z >= 0 ; th >= -Im ; th >= 0 ;  (tn ; A'lam) in Q^3 ; tn <= mm ; mm <= 1
//             obstacle cone  Rpositive: lam >= 0   |  norm2: (tl ; lam_0, lam_1) in Q^3, tl + lam_2 <= 0 (E identical rows)
//             robot cone     Rpositive: mu >= 0    |  norm2: (tr ; mu_0 .. mu_{R-2}) in Q^R, tr + mu_{R-1} <= 0
//
// by a primal-dual interior-point method (Mehrotra predictor-corrector, Nesterov-Todd scaling, normal equations) that ends
// ON THE CENTRAL PATH at a prescribed barrier parameter mu* (s o z = mu* e, residuals at rounding level) rather than at
// whatever iterate a gap test accepts: where the optimal face is not a single point (the "slack regime") the
// centre at a fixed mu* is a well-conditioned function of the data, the end point of an ever smaller mu is not.
//
// Why it exists next to the support enumeration of lammuz_device.h: (i) the enumeration has no candidates for a norm2
// (circle) ROBOT cone - this solver takes any combination of cones; (ii) it returns interior duals (all multipliers
// positive, ||A'lam|| < 1, z = Im) like the interior-point solver behind the reference does, which the enumeration's
// basic solutions cannot.  It is the SLOW path (dense per-thread linear algebra in scratch memory): the enumeration is
// two orders of magnitude faster and stays the default for polygon robots.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace cpq {

struct Problem {             // inputs of one sub-problem (what lmz::Params + the LDS slab carry for the enumeration)
    int E, R, cone_norm2, robot_norm2, accelerated;
    const double *A, *b;     // [E][2], [E]
    const double *G, *h;     // [R][2], [R]
    double px, py, cs, sn, xi0, xi1, kappa0, ro2, mu_target;
};
struct Result { int status; double lam[8], mu[8], z, Im, H0, H1; };    // status 0 ok, 2 failed

template <int NX, int MX> struct Solver {
    static constexpr int KQ = 3;
    int n, l, nq, qd[KQ], m;
    double P[NX][NX], q[NX], G[MX][NX], h[MX];
    double Hm[NX][NX], GW[MX][NX];
    double wd[MX], wbeta[KQ], ww[KQ][8];

    __device__ static double jdet(const double *u, int d) { double s = u[0] * u[0]; for (int i = 1; i < d; ++i) s -= u[i] * u[i]; return s; }
    __device__ double min_eig(const double *u) const
    {
        double v = INFINITY;
        for (int i = 0; i < l; ++i) if (u[i] < v) v = u[i];
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) {
            double s = 0; for (int i = 1; i < qd[k]; ++i) s += u[o + i] * u[o + i];
            double e = u[o] - sqrt(s); if (e < v) v = e;
        }
        return v;
    }
    __device__ void jprod(const double *u, const double *v, double *o_) const
    {
        for (int i = 0; i < l; ++i) o_[i] = u[i] * v[i];
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) {
            const int d = qd[k]; double s = 0;
            for (int i = 0; i < d; ++i) s += u[o + i] * v[o + i];
            for (int i = 1; i < d; ++i) o_[o + i] = u[o] * v[o + i] + v[o] * u[o + i];
            o_[o] = s;
        }
    }
    __device__ void jdiv(const double *lam, const double *b, double *o_) const
    {
        for (int i = 0; i < l; ++i) o_[i] = b[i] / lam[i];
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) {
            const int d = qd[k]; const double *lm = lam + o, *bb = b + o;
            double det = jdet(lm, d), l1b1 = 0;
            for (int i = 1; i < d; ++i) l1b1 += lm[i] * bb[i];
            o_[o] = (lm[0] * bb[0] - l1b1) / det;
            for (int i = 1; i < d; ++i) o_[o + i] = (-lm[i] * bb[0] + (det * bb[i] + lm[i] * l1b1) / lm[0]) / det;
        }
    }
    __device__ bool nt_compute(const double *s, const double *z)
    {
        for (int i = 0; i < l; ++i) { if (!(s[i] > 0) || !(z[i] > 0)) return false; wd[i] = sqrt(s[i] / z[i]); }
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) {
            const int d = qd[k];
            double ds = jdet(s + o, d), dz = jdet(z + o, d);
            if (!(ds > 0) || !(dz > 0) || !(s[o] > 0) || !(z[o] > 0)) return false;
            double ns = sqrt(ds), nz = sqrt(dz), g = 0;
            for (int i = 0; i < d; ++i) g += (s[o + i] / ns) * (z[o + i] / nz);
            g = sqrt(0.5 * (1.0 + g));
            ww[k][0] = (s[o] / ns + z[o] / nz) / (2 * g);
            for (int i = 1; i < d; ++i) ww[k][i] = (s[o + i] / ns - z[o + i] / nz) / (2 * g);
            wbeta[k] = sqrt(ns / nz);
        }
        return true;
    }
    __device__ void nt_apply(const double *u, double *o_, bool inverse) const
    {
        for (int i = 0; i < l; ++i) o_[i] = inverse ? u[i] / wd[i] : u[i] * wd[i];
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) {
            const int d = qd[k]; const double *w = ww[k]; const double sg = inverse ? -1.0 : 1.0;
            double w1u1 = 0;
            for (int i = 1; i < d; ++i) w1u1 += w[i] * u[o + i];
            const double r0 = w[0] * u[o] + sg * w1u1, f = w1u1 / (1.0 + w[0]), sc = inverse ? 1.0 / wbeta[k] : wbeta[k];
            for (int i = 1; i < d; ++i) o_[o + i] = (sg * w[i] * u[o] + u[o + i] + w[i] * f) * sc;
            o_[o] = r0 * sc;
        }
    }
    __device__ double max_step(const double *u, const double *du) const
    {
        double a = INFINITY;
        for (int i = 0; i < l; ++i) if (du[i] < 0) { double t = -u[i] / du[i]; if (t < a) a = t; }
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) {
            const int d = qd[k];
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
    __device__ bool chol()
    {
        for (int j = 0; j < n; ++j) {
            double d = Hm[j][j];
            for (int k = 0; k < j; ++k) d -= Hm[j][k] * Hm[j][k];
            if (!(d > 0)) return false;
            d = sqrt(d); Hm[j][j] = d;
            for (int i = j + 1; i < n; ++i) { double v = Hm[i][j]; for (int k = 0; k < j; ++k) v -= Hm[i][k] * Hm[j][k]; Hm[i][j] = v / d; }
        }
        return true;
    }
    __device__ void chol_solve(double *b) const
    {
        for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= Hm[i][k] * b[k]; b[i] = v / Hm[i][i]; }
        for (int i = n - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < n; ++k) v -= Hm[k][i] * b[k]; b[i] = v / Hm[i][i]; }
    }

    // ---- the cone program of one (obstacle, stage) -------------------------------------------------------------
    int iz, ith, itn, imm, itl, itr;
    double qv[8], Mv[8][2], cv[NX];
    __device__ void build(const Problem &p)
    {
        const int E = p.E, R = p.R;
        for (int i = 0; i < NX; ++i) { q[i] = 0; cv[i] = 0; for (int j = 0; j < NX; ++j) P[i][j] = 0; }
        for (int r = 0; r < MX; ++r) { h[r] = 0; for (int i = 0; i < NX; ++i) G[r][i] = 0; }
        for (int i = 0; i < E; ++i) {
            qv[i] = p.A[2 * i] * p.px + p.A[2 * i + 1] * p.py - p.b[i];
            Mv[i][0] = p.A[2 * i] * p.cs + p.A[2 * i + 1] * p.sn; Mv[i][1] = -p.A[2 * i] * p.sn + p.A[2 * i + 1] * p.cs;
        }
        iz = E + R; ith = p.accelerated ? iz + 1 : -1; itn = iz + 1 + (p.accelerated ? 1 : 0); imm = itn + 1;
        n = imm + 1;
        itl = p.cone_norm2 ? n++ : -1; itr = p.robot_norm2 ? n++ : -1;
        double B0[NX], B1[NX];
        for (int i = 0; i < n; ++i) { B0[i] = 0; B1[i] = 0; }
        for (int i = 0; i < E; ++i) { B0[i] = Mv[i][0]; B1[i] = Mv[i][1]; }
        for (int j = 0; j < R; ++j) { B0[E + j] = p.G[2 * j]; B1[E + j] = p.G[2 * j + 1]; }
        for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) P[i][j] = p.ro2 * (B0[i] * B0[j] + B1[i] * B1[j]); q[i] = p.ro2 * (B0[i] * p.xi0 + B1[i] * p.xi1); }
        for (int i = 0; i < E; ++i) cv[i] = qv[i];
        for (int j = 0; j < R; ++j) cv[E + j] = -p.h[j];
        cv[iz] = -1;
        if (p.accelerated) P[ith][ith] += 1.0;
        else for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) P[i][j] += cv[i] * cv[j]; q[i] += p.kappa0 * cv[i]; }
        int r = 0;
        G[r][iz] = -1; ++r;
        if (p.accelerated) {
            for (int i = 0; i < n; ++i) G[r][i] = -cv[i];
            G[r][ith] = -1; h[r] = p.kappa0; ++r;
            G[r][ith] = -1; ++r;
        }
        G[r][itn] = 1; G[r][imm] = -1; ++r;
        G[r][imm] = 1; h[r] = 1; ++r;
        // zero-padded edge rows: their multiplier enters nothing and has no central value (its dual is 0): left unconstrained at 0
        if (!p.cone_norm2) for (int i = 0; i < E; ++i) { if (p.A[2 * i] == 0 && p.A[2 * i + 1] == 0 && p.b[i] == 0) continue; G[r][i] = -1; ++r; }
        else for (int i = 0; i < E; ++i) { G[r][itl] = 1; G[r][2] = 1; ++r; }
        if (!p.robot_norm2) for (int j = 0; j < R; ++j) { G[r][E + j] = -1; ++r; }
        else { G[r][itr] = 1; G[r][E + R - 1] = 1; ++r; }
        l = r;
        nq = 0;
        G[r][itn] = -1; ++r;
        for (int k = 0; k < 2; ++k) { for (int i = 0; i < E; ++i) G[r][i] = -p.A[2 * i + k]; ++r; }
        qd[nq++] = 3;
        if (p.cone_norm2) { G[r][itl] = -1; ++r; G[r][0] = 1; ++r; G[r][1] = 1; ++r; qd[nq++] = 3; }
        if (p.robot_norm2) { G[r][itr] = -1; ++r; for (int j = 0; j < R - 1; ++j) { G[r][E + j] = 1; ++r; } qd[nq++] = R; }
        m = r;
    }

    // returns 0 (on the central path at mu_target), 2 failed
    __device__ int solve(double mu_target, double *x)
    {
        double z[MX], s[MX], e[MX];
        for (int r = 0; r < m; ++r) e[r] = 0;
        for (int i = 0; i < l; ++i) e[i] = 1;
        for (int k = 0, o = l; k < nq; o += qd[k], ++k) e[o] = 1;
        const int deg = l + nq;
        const double REG = 1e-11;
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = P[i][j]; for (int r = 0; r < m; ++r) v += G[r][i] * G[r][j]; Hm[i][j] = v + (i == j ? REG : 0); }
        if (!chol()) return 2;
        for (int i = 0; i < n; ++i) { double v = -q[i]; for (int r = 0; r < m; ++r) v += G[r][i] * h[r]; x[i] = v; }
        chol_solve(x);
        double ns = 0, nz = 0;
        for (int r = 0; r < m; ++r) { double v = -h[r]; for (int i = 0; i < n; ++i) v += G[r][i] * x[i]; z[r] = v; s[r] = -v; ns += v * v; }
        nz = ns = sqrt(ns);
        { double ts = -min_eig(s); if (ts >= -1e-8 * (ns > 1 ? ns : 1)) for (int r = 0; r < m; ++r) s[r] += (1 + ts) * e[r];
          double tz = -min_eig(z); if (tz >= -1e-8 * (nz > 1 ? nz : 1)) for (int r = 0; r < m; ++r) z[r] += (1 + tz) * e[r]; }
        double nqn = 1, nhn = 1;
        for (int i = 0; i < n; ++i) if (1 + fabs(q[i]) > nqn) nqn = 1 + fabs(q[i]);
        for (int r = 0; r < m; ++r) if (1 + fabs(h[r]) > nhn) nhn = 1 + fabs(h[r]);
        for (int it = 0; it < 60; ++it) {
            double rx[NX], rz[MX], gap = 0, dres = 0, pres = 0;
            for (int i = 0; i < n; ++i) { double v = q[i]; for (int j = 0; j < n; ++j) v += P[i][j] * x[j]; for (int r = 0; r < m; ++r) v += G[r][i] * z[r];
                                          rx[i] = v; if (fabs(v) > dres) dres = fabs(v); }
            for (int r = 0; r < m; ++r) { double v = s[r] - h[r]; for (int i = 0; i < n; ++i) v += G[r][i] * x[i]; rz[r] = v; if (fabs(v) > pres) pres = fabs(v); gap += s[r] * z[r]; }
            dres /= nqn; pres /= nhn;
            if (!nt_compute(s, z)) return 2;
            double lam[MX]; nt_apply(z, lam, false);
            double ll[MX]; jprod(lam, lam, ll);
            const bool centring = gap / deg <= 10 * mu_target;
            if (centring) {
                double cent = 0;
                for (int r = 0; r < m; ++r) { double v = fabs(ll[r] - mu_target * e[r]); if (v > cent) cent = v; }
                if (dres <= 1e-10 && pres <= 1e-10 && cent <= 1e-7 * mu_target) return 0;
            }
            { double col[MX], o_[MX];
              for (int i = 0; i < n; ++i) { for (int r = 0; r < m; ++r) col[r] = G[r][i]; nt_apply(col, o_, true); for (int r = 0; r < m; ++r) GW[r][i] = o_[r]; } }
            for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = P[i][j]; for (int r = 0; r < m; ++r) v += GW[r][i] * GW[r][j]; Hm[i][j] = v + (i == j ? REG : 0); }
            if (!chol()) return 2;
            double dxa[NX], dza[MX], dsa[MX], dx[NX], dz[MX], ds[MX], rhs[NX];
            double sigma = 0, mu = gap / deg;
            bool bad = false;
            if (centring) { sigma = 1.0; mu = mu_target; }
            for (int pass = centring ? 1 : 0; pass < 2 && !bad; ++pass) {
                double bsv[MX], u[MX], wu[MX], t[MX], wt[MX];
                const double sc = (pass && !centring) ? 1 - sigma : 1.0;
                if (!pass) for (int r = 0; r < m; ++r) bsv[r] = -ll[r];
                else if (centring) for (int r = 0; r < m; ++r) bsv[r] = -ll[r] + mu * e[r];
                else {
                    double a1[MX], a2[MX], pr[MX];
                    nt_apply(dsa, a1, true); nt_apply(dza, a2, false); jprod(a1, a2, pr);
                    for (int r = 0; r < m; ++r) bsv[r] = -ll[r] - pr[r] + sigma * mu * e[r];
                }
                jdiv(lam, bsv, u);
                nt_apply(u, wu, false);
                for (int r = 0; r < m; ++r) t[r] = -sc * rz[r] - wu[r];
                nt_apply(t, wt, true);
                for (int i = 0; i < n; ++i) { double v = -sc * rx[i]; for (int r = 0; r < m; ++r) v += GW[r][i] * wt[r]; rhs[i] = v; }
                chol_solve(rhs);
                double *pdx = pass ? dx : dxa, *pdz = pass ? dz : dza, *pds = pass ? ds : dsa;
                for (int i = 0; i < n; ++i) pdx[i] = rhs[i];
                double gd[MX], v1[MX];
                for (int r = 0; r < m; ++r) { double v = -wt[r]; for (int i = 0; i < n; ++i) v += GW[r][i] * pdx[i]; gd[r] = v; }
                nt_apply(gd, pdz, true);
                for (int r = 0; r < m; ++r) v1[r] = u[r] - gd[r];
                nt_apply(v1, pds, false);
                for (int r = 0; r < m; ++r) if (!isfinite(pdz[r]) || !isfinite(pds[r])) bad = true;
                if (!pass && !bad) {
                    double aa = max_step(s, dsa), ab = max_step(z, dza); if (ab < aa) aa = ab; if (aa > 1) aa = 1;
                    sigma = (1 - aa) * (1 - aa) * (1 - aa);
                }
            }
            if (bad) return 2;
            double a = max_step(s, ds), a2 = max_step(z, dz); if (a2 < a) a = a2; a *= 0.99; if (a > 1) a = 1;
            if (!(a > 0) || !isfinite(a)) return 2;
            for (int i = 0; i < n; ++i) x[i] += a * dx[i];
            for (int r = 0; r < m; ++r) { z[r] += a * dz[r]; s[r] += a * ds[r]; }
        }
        return 2;
    }

    __device__ void run(const Problem &p, Result &out)
    {
        build(p);
        double x[NX];
        out.status = solve(p.mu_target, x);
        if (out.status != 0) return;
        const int E = p.E, R = p.R;
        for (int i = 0; i < E; ++i) { double v = x[i]; if (!p.cone_norm2 && v < 0) v = 0; out.lam[i] = v; }
        for (int j = 0; j < R; ++j) { double v = x[E + j]; if (!p.robot_norm2 && v < 0) v = 0; out.mu[j] = v; }
        out.z = x[iz] > 0 ? x[iz] : 0;
        double Im = p.kappa0 - out.z, H0 = p.xi0, H1 = p.xi1;
        for (int i = 0; i < E; ++i) { Im += qv[i] * out.lam[i]; H0 += Mv[i][0] * out.lam[i]; H1 += Mv[i][1] * out.lam[i]; }
        for (int j = 0; j < R; ++j) { Im -= p.h[j] * out.mu[j]; H0 += p.G[2 * j] * out.mu[j]; H1 += p.G[2 * j + 1] * out.mu[j]; }
        out.Im = Im; out.H0 = H0; out.H1 = H1;
        for (int k = 0; k < 4; ++k) if (!isfinite(out.lam[k < E ? k : 0]) || !isfinite(out.mu[k < R ? k : 0])) out.status = 2;
        if (!isfinite(out.z) || !isfinite(Im) || !isfinite(H0) || !isfinite(H1)) out.status = 2;
    }
};

}  // namespace cpq
