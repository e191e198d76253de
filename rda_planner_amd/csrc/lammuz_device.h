// K1 device code: one LamMuZ sub-problem (obstacle n, stage t) per 64-lane wavefront.
//
// Problem solved (reference rda_solver.py:389-421, 874-909, 1034-1050):
//     min  1/2 neg(m)^2 - delta*m + 1/2 ro2 ||H||^2,   m = lam'q - mu'h + (zeta - d),  H = M'lam + G'mu + xi
//     s.t. lam in K_obs, mu >= 0, ||A'lam|| <= 1,       q = A p - b,  M = A R(phi)
// Because the cost is strictly decreasing in m, optimal (lam, mu) are LP-optimal for their own
// a = A'lam and g = G'mu, so a basic optimal solution has <= 2 non-zero lam_i and <= 2 non-zero
// mu_j ("closest features").  The wavefront ENUMERATES all (lam-support, mu-support, hinge-state)
// candidates, one per lane per pass, solves each tiny piecewise-quadratic problem in closed form
// (a 2-D trust-region secular equation when the separating direction is free), evaluates the TRUE
// cost of every sign-feasible candidate and takes the wave-wide arg-min (ties: lowest candidate
// id).  No iteration over the problem, no divergence across sub-problems, no global scratch:
// the obstacle half-spaces and the derived q, M live in the wave's LDS slab.
// Around the enumeration: prepare_wave (candidate pruning), solve_wave_warm (previous support + optimality certificate)
// and central_normal_wave (tie-break T1 of the degenerate slack regime).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// Floating-point contraction: `on` (a product is fused with the sum it feeds only inside ONE source expression, decided by the
// front end) instead of the HIP default `fast` (the backend fuses opportunistically, across statements, depending on what else was
// inlined around the code).  The same device functions are compiled into several kernels (one sub-problem per wave / four per
// wave / common path only / work list; single ego / fleet) and their results must not depend on which kernel solved a row.
#pragma clang fp contract(on)

namespace lmz {

constexpr int EMAX = 8;
constexpr int RMAX = 8;
constexpr double SIGN_TOL = 1e-12;

struct WaveLDS {            // one slab per wave (stage-local obstacle data)
    double A[EMAX][2];
    double b[EMAX];
    double q[EMAX];
    double M[EMAX][2];
    unsigned char lamc[40];            // surviving lam support candidates (original indices), heavy types first
    double vtx[28][2];                 // polygon vertices of the surviving pairs, in list order (= lamc[0..npv))
    int npv, nlv;                      // number of vertex pairs / of all lam candidates
};
struct RobotLDS {           // one per block
    double G[RMAX][2];
    double h[RMAX];
    unsigned char muc[40]; int nmv;    // surviving mu support candidates (robot geometry only: built once on the host)
    double rv[28][2]; int nrv;         // robot vertices of the surviving pairs, in list order (= muc[0..nrv))
};

struct Sol {
    double cost; int id;
    double m, H0, H1;
    int i1, i2, j1, j2;     // supports (-1 = unused)
    double l1, l2, g1, g2;  // lam / mu values on the supports
};

struct MuCand { int k; int j0, j1; double gx, gy, eta; double P00, P01, P10, P11, det, r0, r1, rr;
                double iden;   // reciprocal of the mu-elimination pivot for the current hinge state (k = 1, 2)
                double idet; };

__device__ __forceinline__ void gamma_star(const MuCand &mc, double chi, double ro2, double delta,
                                           double t, double e0, double e1,
                                           double &ga0, double &ga1, double &m, double &H0, double &H1)
{
    if (mc.k == 0) { m = t; H0 = e0; H1 = e1; ga0 = 0; ga1 = 0; return; }
    if (mc.k == 1) {
        double ga = (chi * mc.eta * t - delta * mc.eta - ro2 * (mc.gx * e0 + mc.gy * e1)) * mc.iden;
        ga0 = ga; ga1 = 0; m = t - mc.eta * ga; H0 = e0 + ga * mc.gx; H1 = e1 + ga * mc.gy;
        return;
    }
    double re = mc.r0 * e0 + mc.r1 * e1;
    double beta = (delta - chi * (t + re)) * mc.iden;
    H0 = -beta * mc.r0; H1 = -beta * mc.r1;
    double d0 = H0 - e0, d1 = H1 - e1;
    ga0 = (mc.P11 * d0 - mc.P01 * d1) * mc.idet;
    ga1 = (-mc.P10 * d0 + mc.P00 * d1) * mc.idet;
    m = t + re + beta * mc.rr;
}

// 2-D trust-region sub-problem  min 1/2 x'Qx + c'x,  ||x||<=1 (disc) or ||x||==1.
// Returns the number of minimisers written (0, 1, or 2 in the hard case).
__device__ __forceinline__ int trs2(double q11, double q12, double q22, double c0, double c1, bool disc,
                                    double &x0, double &x1, double &x2, double &x3)
{
    double mean = 0.5 * (q11 + q22), dif = 0.5 * (q11 - q22);
    double rad = sqrt(dif * dif + q12 * q12);
    double l1 = mean - rad, l2 = mean + rad;
    double v2x, v2y;
    if (dif >= 0) { v2x = dif + rad; v2y = q12; } else { v2x = q12; v2y = rad - dif; }
    double nv2 = v2x * v2x + v2y * v2y;
    if (nv2 > 0) { double inv = rsqrt(nv2); v2x *= inv; v2y *= inv; } else { v2x = 1; v2y = 0; }
    double v1x = -v2y, v1y = v2x;
    double h1 = v1x * c0 + v1y * c1, h2 = v2x * c0 + v2y * c1;
    double cn = sqrt(h1 * h1 + h2 * h2);
    double scale = fabs(l2) > 1e-300 ? fabs(l2) : 1e-300;
    if (disc && l1 > 1e-13 * scale) {
        double y1 = -h1 / l1, y2 = -h2 / l2;
        if (y1 * y1 + y2 * y2 <= 1.0) { x0 = y1 * v1x + y2 * v2x; x1 = y1 * v1y + y2 * v2y; return 1; }
    }
    if (cn == 0.0) {
        if (disc) return 0;
        x0 = v1x; x1 = v1y; return 1;
    }
    if (fabs(h1) <= 1e-9 * cn) {
        // hard case: c orthogonal to the low-curvature eigenvector -> two minimisers
        double gap = l2 - l1;
        if (gap > 0) {
            double y2 = -h2 / gap;
            if (fabs(y2) < 1.0) {
                double sq = sqrt(1.0 - y2 * y2);
                x0 = y2 * v2x + sq * v1x; x1 = y2 * v2y + sq * v1y;
                x2 = y2 * v2x - sq * v1x; x3 = y2 * v2y - sq * v1y;
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
        const double s1i = 1.0 / s1, s2i = 1.0 / s2;
        double a1 = h1 != 0 ? h1 * s1i : 0.0, a2 = h2 != 0 ? h2 * s2i : 0.0;
        double phi = a1 * a1 + a2 * a2;
        if (!(phi > 0)) break;
        double dphi = -2.0 * (a1 * a1 * s1i + a2 * a2 * s2i);
        double rs = rsqrt(phi);                               // g = 1/||y|| - 1,  dg = -1/2 dphi / ||y||^3
        double g = rs - 1.0, dg = -0.5 * dphi * rs * rs * rs;
        double step = g / dg;
        tau -= step;
        if (fabs(step) <= 4e-16 * (fabs(tau) > 1 ? fabs(tau) : 1)) break;
    }
    double s1 = l1 + tau, s2 = l2 + tau;
    double y1 = h1 != 0 ? -h1 / s1 : 0.0, y2 = h2 != 0 ? -h2 / s2 : 0.0;
    double xx = y1 * v1x + y2 * v2x, xy = y1 * v1y + y2 * v2y;
    double nx2 = xx * xx + xy * xy;
    if (nx2 > 0) { double inv = rsqrt(nx2); xx *= inv; xy *= inv; }
    x0 = xx; x1 = xy;
    return 1;
}

// gradient (d/dt, d/de) and value of the candidate model after eliminating the mu-support coefficients
__device__ __forceinline__ void model_g3(const MuCand &mc, double chi, double ro2, double delta, double t, double e0, double e1,
                                         double &gt, double &ge0, double &ge1, double &f)
{
    double ga0, ga1, m, H0, H1;
    gamma_star(mc, chi, ro2, delta, t, e0, e1, ga0, ga1, m, H0, H1);
    gt = chi * m - delta; ge0 = ro2 * H0; ge1 = ro2 * H1;
    f = 0.5 * chi * m * m - delta * m + 0.5 * ro2 * (H0 * H0 + H1 * H1);
}

// Circle obstacle, candidate with 0 < ||a|| < 1 (lam_3 = -||a|| tight): damped Newton on
// Phi(at) = model(t = at'ut + l0*||at|| + kappa0, e = at + xi), l0 = -radius, started on the steepest-descent ray out
// of the kink at at = 0 (Phi is quadratic along a ray).  Returns false when the ||a|| in {0, 1} candidates cover
// the optimum.  mc.k < 2 only.  Lane-local: runs on the lanes that own an LI candidate.
__device__ __forceinline__ bool circle_interior(const MuCand &mc, double chi, double ro2, double delta, double ut0, double ut1,
                                                double l0, double kappa0, double xi0, double xi1, double &at0, double &at1)
{
    double G0t, G00, G01, c0, c1, c2, f;
    model_g3(mc, chi, ro2, delta, 0.0, 0.0, 0.0, G0t, G00, G01, f);
    double h00, h01, h02, h11, h12, h22;                      // symmetric 3x3 model Hessian in (t, e0, e1)
    model_g3(mc, chi, ro2, delta, 1.0, 0.0, 0.0, c0, c1, c2, f); h00 = c0 - G0t; double h10 = c1 - G00, h20 = c2 - G01;
    model_g3(mc, chi, ro2, delta, 0.0, 1.0, 0.0, c0, c1, c2, f); h01 = c0 - G0t; h11 = c1 - G00; double h21 = c2 - G01;
    model_g3(mc, chi, ro2, delta, 0.0, 0.0, 1.0, c0, c1, c2, f); h02 = c0 - G0t; h12 = c1 - G00; h22 = c2 - G01;
    h01 = 0.5 * (h01 + h10); h02 = 0.5 * (h02 + h20); h12 = 0.5 * (h12 + h21);
    const double nu = hypot(ut0, ut1);
    double gt, ge0, ge1;
    model_g3(mc, chi, ro2, delta, kappa0, xi0, xi1, gt, ge0, ge1, f);
    double gk0 = gt * ut0 + ge0, gk1 = gt * ut1 + ge1, ck = gt * l0, ng = hypot(gk0, gk1);
    if (!(ng > ck * (1.0 + 1e-12))) return false;
    double v0 = -gk0 / ng, v1 = -gk1 / ng;
    double w0 = v0 * ut0 + v1 * ut1 + l0;
    double curv = w0 * (h00 * w0 + h01 * v0 + h02 * v1) + v0 * (h01 * w0 + h11 * v0 + h12 * v1) + v1 * (h02 * w0 + h12 * v0 + h22 * v1);
    double s0 = curv > (ng - ck) / 0.9 ? (ng - ck) / curv : 0.9;
    double x0 = s0 * v0, x1 = s0 * v1, s_ = hypot(x0, x1);
    model_g3(mc, chi, ro2, delta, x0 * ut0 + x1 * ut1 + l0 * s_ + kappa0, x0 + xi0, x1 + xi1, gt, ge0, ge1, f);
    int nclip = 0;
    for (int it = 0; it < 30; ++it) {
        s_ = hypot(x0, x1);
        double a0 = x0 / s_, a1 = x1 / s_, fdummy;
        model_g3(mc, chi, ro2, delta, x0 * ut0 + x1 * ut1 + l0 * s_ + kappa0, x0 + xi0, x1 + xi1, gt, ge0, ge1, fdummy);
        double j0 = ut0 + l0 * a0, j1 = ut1 + l0 * a1;          // dt/dat
        double gr0 = gt * j0 + ge0, gr1 = gt * j1 + ge1;
        // Hs = J' Hm J with J = [j; I]
        double H00 = j0 * (h00 * j0 + h01) + (h01 * j0 + h11);
        double H01 = j0 * (h00 * j1 + h02) + (h01 * j1 + h12);
        double H10 = j1 * (h00 * j0 + h01) + (h02 * j0 + h12);
        double H11 = j1 * (h00 * j1 + h02) + (h02 * j1 + h22);
        double kq = gt * l0 / s_;
        H00 += kq * (1 - a0 * a0); H01 += kq * (-a0 * a1); H10 += kq * (-a0 * a1); H11 += kq * (1 - a1 * a1);
        double tr = 0.5 * (H00 + H11), df = 0.5 * (H00 - H11), rad = hypot(df, H01);
        double lmin = tr - rad, lmax = fabs(tr + rad) > 1e-300 ? fabs(tr + rad) : 1e-300;
        if (lmin < 1e-8 * lmax) { double sh = 1e-8 * lmax - lmin; H00 += sh; H11 += sh; }
        double det = H00 * H11 - H01 * H10;
        if (!(det > 0)) return false;
        double d0 = -(H11 * gr0 - H01 * gr1) / det, d1 = -(-H10 * gr0 + H00 * gr1) / det;
        double gsc = 1.0 + fabs(gt) * nu + hypot(ge0, ge1);
        if (hypot(gr0, gr1) <= 1e-13 * gsc || hypot(d0, d1) <= 1e-15 * (s_ > 1 ? s_ : 1.0)) break;
        double al = 1.0, xn0 = x0, xn1 = x1, fn = f; bool ok = false, clipped = false;
        for (int bt = 0; bt < 30; ++bt) {
            xn0 = x0 + al * d0; xn1 = x1 + al * d1;
            double sn = hypot(xn0, xn1);
            if (sn > 1e-12 && sn < 1.0) {
                double t0, t1, t2;
                model_g3(mc, chi, ro2, delta, xn0 * ut0 + xn1 * ut1 + l0 * sn + kappa0, xn0 + xi0, xn1 + xi1, t0, t1, t2, fn);
                if (fn <= f + 1e-4 * al * (gr0 * d0 + gr1 * d1) + 1e-13 * fabs(f)) { ok = true; break; }
            } else if (sn >= 1.0) clipped = true;
            al *= 0.5;
        }
        if (clipped && ++nclip >= 3) return false;
        if (!ok) break;
        x0 = xn0; x1 = xn1; f = fn;
    }
    s_ = hypot(x0, x1);
    model_g3(mc, chi, ro2, delta, x0 * ut0 + x1 * ut1 + l0 * s_ + kappa0, x0 + xi0, x1 + xi1, gt, ge0, ge1, f);
    double gr0 = gt * (ut0 + l0 * x0 / s_) + ge0, gr1 = gt * (ut1 + l0 * x1 / s_) + ge1;
    if (hypot(gr0, gr1) > 1e-9 * (1.0 + fabs(gt) * nu + hypot(ge0, ge1))) return false;
    at0 = x0; at1 = x1;
    return true;
}

// decode the k-th pair (lexicographic i1<i2) of {0..n-1}
__device__ __forceinline__ void decode_pair(int k, int n, int &i1, int &i2)
{
    i1 = 0;
    int rowlen = n - 1;
    while (k >= rowlen) { k -= rowlen; ++i1; --rowlen; }
    i2 = i1 + 1 + k;
}

struct Params { int E, R; int norm2; double px, py, cs, sn, xi0, xi1, kappa0, ro2, delta; long long *prof = nullptr; };

// sign feasibility, clamping and cost of one candidate point: the hinge-inactive MODEL cost in pass ic = 0
// (a lower bound of the true cost, exact for m >= 0), the TRUE cost in pass ic = 1
__device__ __forceinline__ bool finish(const WaveLDS &W, const RobotLDS &Rb, const Params &P, const MuCand &mc, int type, int ic,
                                       int i1, int i2, double la1, double la2, double ga0, double ga1, Sol &s)
{
    if (!P.norm2) {
        if (la1 < -SIGN_TOL || la2 < -SIGN_TOL) return false;
        if (la1 < 0) la1 = 0;
        if (la2 < 0) la2 = 0;
    }
    if (ga0 < -SIGN_TOL || ga1 < -SIGN_TOL) return false;
    if (ga0 < 0) ga0 = 0;
    if (ga1 < 0) ga1 = 0;
    double mm = P.kappa0, HH0 = P.xi0, HH1 = P.xi1;
    if (type >= 3) {
        double l3 = -hypot(la1, la2);
        mm += la1 * W.q[0] + la2 * W.q[1] + l3 * W.q[2];
        HH0 += la1 * W.M[0][0] + la2 * W.M[1][0] + l3 * W.M[2][0];
        HH1 += la1 * W.M[0][1] + la2 * W.M[1][1] + l3 * W.M[2][1];
    } else {
        if (i1 >= 0) { mm += la1 * W.q[i1]; HH0 += la1 * W.M[i1][0]; HH1 += la1 * W.M[i1][1]; }
        if (i2 >= 0) { mm += la2 * W.q[i2]; HH0 += la2 * W.M[i2][0]; HH1 += la2 * W.M[i2][1]; }
    }
    if (mc.k >= 1) { mm -= ga0 * Rb.h[mc.j0]; HH0 += ga0 * Rb.G[mc.j0][0]; HH1 += ga0 * Rb.G[mc.j0][1]; }
    if (mc.k == 2) { mm -= ga1 * Rb.h[mc.j1]; HH0 += ga1 * Rb.G[mc.j1][0]; HH1 += ga1 * Rb.G[mc.j1][1]; }
    double ng = mm < 0 ? mm : 0;
    s.cost = (ic ? 0.5 * ng * ng : 0.0) - P.delta * mm + 0.5 * P.ro2 * (HH0 * HH0 + HH1 * HH1);
    s.m = mm; s.H0 = HH0; s.H1 = HH1;
    s.i1 = i1; s.i2 = i2; s.l1 = la1; s.l2 = la2;
    s.j1 = mc.j0; s.j2 = mc.j1; s.g1 = ga0; s.g2 = ga1;
    s.id = 0;
    return true;
}

// Evaluate candidate (il, im, ic); returns false when the candidate does not exist / is infeasible.
__device__ __forceinline__ bool eval_candidate(const WaveLDS &W, const RobotLDS &Rb, const Params &P,
                                               int il, int im, int ic, Sol &s)
{
    const double chi = (double)ic, ro2 = P.ro2, delta = P.delta;
    // ---- mu candidate ----------------------------------------------------------------------
    MuCand mc; mc.k = 0; mc.j0 = mc.j1 = -1;
    if (im >= 1 && im <= P.R) {
        mc.k = 1; mc.j0 = im - 1; mc.gx = Rb.G[mc.j0][0]; mc.gy = Rb.G[mc.j0][1]; mc.eta = Rb.h[mc.j0];
        const double g2 = mc.gx * mc.gx + mc.gy * mc.gy;
        if (!(g2 > 0)) return false;
        mc.iden = 1.0 / (chi * mc.eta * mc.eta + ro2 * g2);
    } else if (im > P.R) {
        mc.k = 2; decode_pair(im - 1 - P.R, P.R, mc.j0, mc.j1);
        double a0 = Rb.G[mc.j0][0], a1 = Rb.G[mc.j0][1], b0 = Rb.G[mc.j1][0], b1 = Rb.G[mc.j1][1];
        double det = a0 * b1 - a1 * b0;
        if (det == 0 || !(det * det > 1e-24 * (a0 * a0 + a1 * a1) * (b0 * b0 + b1 * b1))) return false;
        mc.P00 = a0; mc.P01 = b0; mc.P10 = a1; mc.P11 = b1; mc.det = det; mc.idet = 1.0 / det;
        double h0 = Rb.h[mc.j0], h1 = Rb.h[mc.j1];
        mc.r0 = (h0 * b1 - a1 * h1) * mc.idet; mc.r1 = (a0 * h1 - h0 * b0) * mc.idet;
        mc.rr = mc.r0 * mc.r0 + mc.r1 * mc.r1;
        mc.iden = 1.0 / (chi * mc.rr + ro2);
    }
    // ---- lam candidate ---------------------------------------------------------------------
    int i1 = -1, i2 = -1; double la1 = 0, la2 = 0;
    double ga0 = 0, ga1 = 0, m, H0, H1;
    int type;                                   // 0 L0, 1 L1, 2 L2, 3 LC (||a|| = 1), 4 LI (0 < ||a|| < 1)
    if (il == 0) type = 0;
    else if (P.norm2) type = il == 1 ? 3 : 4;
    else if (il <= P.E) { type = 1; i1 = il - 1; }
    else { type = 2; decode_pair(il - 1 - P.E, P.E, i1, i2); }

    if (type == 0) {
        gamma_star(mc, chi, ro2, delta, P.kappa0, P.xi0, P.xi1, ga0, ga1, m, H0, H1);
    } else if (type == 1) {
        double ax = W.A[i1][0], ay = W.A[i1][1];
        double n2 = ax * ax + ay * ay;
        if (!(n2 > 0)) return false;
        double amax = rsqrt(n2);
        double qi = W.q[i1], m0 = W.M[i1][0], m1 = W.M[i1][1];
        gamma_star(mc, chi, ro2, delta, P.kappa0, P.xi0, P.xi1, ga0, ga1, m, H0, H1);
        double d0 = (chi * m - delta) * qi + ro2 * (m0 * H0 + m1 * H1);
        gamma_star(mc, chi, ro2, delta, amax * qi + P.kappa0, amax * m0 + P.xi0, amax * m1 + P.xi1, ga0, ga1, m, H0, H1);
        double d1 = (chi * m - delta) * qi + ro2 * (m0 * H0 + m1 * H1);
        double al;
        if (d1 <= 0) al = amax; else if (d0 >= 0) al = 0; else al = amax * d0 / (d0 - d1);
        gamma_star(mc, chi, ro2, delta, al * qi + P.kappa0, al * m0 + P.xi0, al * m1 + P.xi1, ga0, ga1, m, H0, H1);
        la1 = al;
    } else {
        double dvx, dvy, l0 = 0, a00 = 0, a01 = 0, a10 = 0, a11 = 0, detS = 1, idetS = 1;
        if (type == 2) {
            a00 = W.A[i1][0]; a01 = W.A[i1][1]; a10 = W.A[i2][0]; a11 = W.A[i2][1];
            detS = a00 * a11 - a01 * a10;
            if (detS == 0 || !(detS * detS > 1e-24 * (a00 * a00 + a01 * a01) * (a10 * a10 + a11 * a11))) return false;
            idetS = 1.0 / detS;
            double vx = (W.b[i1] * a11 - a01 * W.b[i2]) * idetS;
            double vy = (a00 * W.b[i2] - W.b[i1] * a10) * idetS;
            dvx = P.px - vx; dvy = P.py - vy;
        } else { dvx = P.px - W.b[0]; dvy = P.py - W.b[1]; l0 = W.b[2]; i1 = 0; i2 = 1; }
        double ut0 = P.cs * dvx + P.sn * dvy, ut1 = -P.sn * dvx + P.cs * dvy;       // R'(p - v)
        if (type == 4) {
            double at0, at1;
            if (mc.k == 2 || !circle_interior(mc, chi, ro2, delta, ut0, ut1, l0, P.kappa0, P.xi0, P.xi1, at0, at1)) return false;
            gamma_star(mc, chi, ro2, delta, at0 * ut0 + at1 * ut1 + l0 * hypot(at0, at1) + P.kappa0, at0 + P.xi0, at1 + P.xi1,
                       ga0, ga1, m, H0, H1);
            la1 = P.cs * at0 - P.sn * at1; la2 = P.sn * at0 + P.cs * at1;              // a = R at
            return finish(W, Rb, P, mc, type, ic, i1, i2, la1, la2, ga0, ga1, s);
        }
        double g00, g01, g10, g11, g20, g21;
        gamma_star(mc, chi, ro2, delta, l0 + P.kappa0, P.xi0, P.xi1, ga0, ga1, m, H0, H1);
        g00 = (chi * m - delta) * ut0 + ro2 * H0; g01 = (chi * m - delta) * ut1 + ro2 * H1;
        gamma_star(mc, chi, ro2, delta, ut0 + l0 + P.kappa0, 1 + P.xi0, P.xi1, ga0, ga1, m, H0, H1);
        g10 = (chi * m - delta) * ut0 + ro2 * H0 - g00; g11 = (chi * m - delta) * ut1 + ro2 * H1 - g01;
        gamma_star(mc, chi, ro2, delta, ut1 + l0 + P.kappa0, P.xi0, 1 + P.xi1, ga0, ga1, m, H0, H1);
        g20 = (chi * m - delta) * ut0 + ro2 * H0 - g00; g21 = (chi * m - delta) * ut1 + ro2 * H1 - g01;
        double ats[4] = {0, 0, 0, 0};
        int nsol = trs2(g10, 0.5 * (g11 + g20), g21, g00, g01, type == 2, ats[0], ats[1], ats[2], ats[3]);
        bool any = false;
        for (int sol = 0; sol < nsol; ++sol) {
            double at0 = ats[2 * sol], at1 = ats[2 * sol + 1];
            gamma_star(mc, chi, ro2, delta, at0 * ut0 + at1 * ut1 + l0 + P.kappa0, at0 + P.xi0, at1 + P.xi1, ga0, ga1, m, H0, H1);
            double ax = P.cs * at0 - P.sn * at1, ay = P.sn * at0 + P.cs * at1;          // a = R at
            if (type == 2) {
                la1 = (ax * a11 - a10 * ay) * idetS;
                la2 = (a00 * ay - ax * a01) * idetS;
            } else { la1 = ax; la2 = ay; }
            Sol c2;
            if (finish(W, Rb, P, mc, type, ic, i1, i2, la1, la2, ga0, ga1, c2) && (!any || c2.cost < s.cost)) { s = c2; any = true; }
        }
        return any;
    }
    return finish(W, Rb, P, mc, type, ic, i1, i2, la1, la2, ga0, ga1, s);
}

// Per-wave preparation, all 64 lanes: q = A p - b, M = A R, and the lam candidate list.  A two-element support {i1,i2}
// can only be optimal if the intersection of its two lines is a vertex of the polygon (the optimal lam is LP-optimal
// for its a = A'lam, and an optimal basis is primal feasible); same for the robot (list built once on the host).
// Infeasible pairs and null rows are dropped up front, so the lane loop of solve_wave runs over the survivors only: a
// quadrilateral x rectangle problem shrinks from 121 to 81 candidates per hinge state, a triangle from 121 to 63 (one
// lane pass).  Pairs first: the last pass then carries only the cheap types.  The vertices are kept for the
// central-normal rule.
// q = A p - b ; M = A R  (lanes < E; the caller synchronises the wave afterwards)
__device__ __forceinline__ void pose_products(WaveLDS &W, const Params &P, int lane)
{
    if (lane < P.E) {
        double ax = W.A[lane][0], ay = W.A[lane][1];
        W.q[lane] = ax * P.px + ay * P.py - W.b[lane];
        W.M[lane][0] = ax * P.cs + ay * P.sn;
        W.M[lane][1] = -ax * P.sn + ay * P.cs;
    }
}

// candidate list + vertices of the obstacle in W.A / W.b (pose independent: k_prepare caches it per obstacle slot)
__device__ __forceinline__ void build_lists(WaveLDS &W, int E, int norm2, int lane)
{
    if (norm2) {
        if (lane == 0) { W.lamc[0] = 1; W.lamc[1] = 2; W.lamc[2] = 0; W.npv = 0; W.nlv = 3; }
    } else {
        bool vp = false; double vx = 0, vy = 0;
        if (lane < E * (E - 1) / 2) {
            int i1, i2; decode_pair(lane, E, i1, i2);
            const double a00 = W.A[i1][0], a01 = W.A[i1][1], a10 = W.A[i2][0], a11 = W.A[i2][1];
            const double det = a00 * a11 - a01 * a10;
            if (det != 0 && det * det > 1e-24 * (a00 * a00 + a01 * a01) * (a10 * a10 + a11 * a11)) {
                // v det = (b1 a11 - a01 b2, a00 b2 - b1 a10);  A_k v - b_k <= tol  <=>  sign(det) (A_k (v det) - b_k det) <= tol |det|
                const double wx = W.b[i1] * a11 - a01 * W.b[i2], wy = a00 * W.b[i2] - W.b[i1] * a10, sg = det > 0 ? 1.0 : -1.0, ad = fabs(det);
                vp = true;
                for (int k = 0; k < E; ++k) {
                    const double ak0 = W.A[k][0], ak1 = W.A[k][1], bk = W.b[k];
                    const double viol = sg * (ak0 * wx + ak1 * wy - bk * det);
                    if (viol > 1e-9 * (ad + fabs(bk) * ad + fabs(ak0 * wx) + fabs(ak1 * wy))) vp = false;
                }
                vx = wx / det; vy = wy / det;
            }
        }
        const bool vs = lane < E && (W.A[lane < E ? lane : 0][0] != 0 || W.A[lane < E ? lane : 0][1] != 0);
        const unsigned long long bp = __ballot(vp), bs = __ballot(vs), below = (1ull << lane) - 1;
        const int npv = __popcll(bp), nsv = __popcll(bs);
        if (vp) { const int k = __popcll(bp & below); W.lamc[k] = (unsigned char)(1 + E + lane); W.vtx[k][0] = vx; W.vtx[k][1] = vy; }
        if (vs) W.lamc[npv + __popcll(bs & below)] = (unsigned char)(1 + lane);
        if (lane == 0) { W.lamc[npv + nsv] = 0; W.npv = npv; W.nlv = npv + nsv + 1; }
    }
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void prepare_wave(WaveLDS &W, const Params &P, int lane)
{
    pose_products(W, P, lane);
    build_lists(W, P.E, P.norm2, lane);
    wave_sync();
}

// Whole-wave enumeration.  All 64 lanes must call after prepare_wave.  On return every lane holds the winning
// solution in `best`.
__device__ __forceinline__ void solve_wave(WaveLDS &W, const RobotLDS &Rb, const Params &P, int lane, Sol &best)
{
    const int nm = 1 + P.R + P.R * (P.R - 1) / 2;            // original candidate numbering (ids, tie-break T3)
    const int nlv = W.nlv, nmv = Rb.nmv;
    const int half = nlv * nmv;
    best.cost = INFINITY; best.id = 0x7fffffff;
    best.m = 0; best.H0 = 0; best.H1 = 0; best.i1 = best.i2 = best.j1 = best.j2 = -1;
    best.l1 = best.l2 = best.g1 = best.g2 = 0;
    long long tprev = P.prof ? clock64() : 0;
    auto mark = [&](int k) { if (P.prof && lane == 0) { long long now = clock64(); P.prof[k] += now - tprev; tprev = now; } };
    // Rule T3.  Pass 1 (ic = 0): the hinge-inactive candidates are ranked by the cost of the hinge-inactive MODEL
    // (-delta*m + ro2/2|H|^2: a lower bound of the true cost, exact for m >= 0); if its minimiser c0 has m >= 0 it is
    // the global optimum.  Otherwise pass 2 (ic = 1): the hinge-active candidates and c0 compete on the TRUE cost.
    // Exact ties go to the lowest id = 2*(il*nm+im)+ic.
    for (int ic = 0; ic < 2; ++ic) {
        int itn = 0;
        for (int c = lane; c < half; c += 64) {
            const int ilx = c / nmv, il = W.lamc[ilx], im = Rb.muc[c - ilx * nmv];
            Sol s;
            if (eval_candidate(W, Rb, P, il, im, ic, s)) {
                s.id = 2 * (il * nm + im) + ic;
                if (s.cost < best.cost || (s.cost == best.cost && s.id < best.id)) best = s;
            }
            mark(1 + 2 * ic + (itn < 1 ? itn : 1)); ++itn;
        }
        // wave arg-min on (cost, id)
        double bc = best.cost; int bid = best.id;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            double oc = __shfl_xor(bc, off, 64); int oid = __shfl_xor(bid, off, 64);
            if (oc < bc || (oc == bc && oid < bid)) { bc = oc; bid = oid; }
        }
        unsigned long long win = __ballot(best.id == bid);
        int src = __ffsll((long long)win) - 1;
        best.cost = __shfl(best.cost, src, 64); best.id = bid;
        best.m = __shfl(best.m, src, 64); best.H0 = __shfl(best.H0, src, 64); best.H1 = __shfl(best.H1, src, 64);
        best.i1 = __shfl(best.i1, src, 64); best.i2 = __shfl(best.i2, src, 64);
        best.j1 = __shfl(best.j1, src, 64); best.j2 = __shfl(best.j2, src, 64);
        best.l1 = __shfl(best.l1, src, 64); best.l2 = __shfl(best.l2, src, 64);
        best.g1 = __shfl(best.g1, src, 64); best.g2 = __shfl(best.g2, src, 64);
        mark(5 + ic);
        if (best.m >= 0) break;                 // wave-uniform after the broadcast
        if (ic == 0) best.cost += 0.5 * best.m * best.m;      // c0: model cost -> true cost (m < 0 here)
    }
}

// wave-wide max of a double on every lane: DPP inside the 16-lane rows (xor 1, xor 2, half mirror, mirror), two
// ds_bpermute rounds across the rows
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v)
{
    double o;
    o = dpp_mov_f64<0xB1>(v); v = o > v ? o : v;
    o = dpp_mov_f64<0x4E>(v); v = o > v ? o : v;
    o = dpp_mov_f64<0x141>(v); v = o > v ? o : v;
    o = dpp_mov_f64<0x140>(v); v = o > v ? o : v;
    o = __shfl_xor(v, 16, 64); v = o > v ? o : v;
    o = __shfl_xor(v, 32, 64); v = o > v ? o : v;
    return v;
}

// Lane groups.  GW = 64: the wave works on one sub-problem.  GW = 16: four sub-problems per wave, one per DPP row - every
// cross-lane step below stays inside a row, so the rows may diverge from each other (different supports, early exits).
__device__ __forceinline__ double row_max(double v)
{
    double o;
    o = dpp_mov_f64<0xB1>(v); v = o > v ? o : v;
    o = dpp_mov_f64<0x4E>(v); v = o > v ? o : v;
    o = dpp_mov_f64<0x141>(v); v = o > v ? o : v;
    o = dpp_mov_f64<0x140>(v); v = o > v ? o : v;
    return v;
}
template <int GW> struct Grp {
    static __device__ __forceinline__ unsigned long long ballot(bool p, int lane)
    {
        const unsigned long long b = __ballot(p);
        if (GW == 64) return b;
        return (b >> (lane & 48)) & 0xffffull;
    }
    static __device__ __forceinline__ double max(double v) { return GW == 64 ? wave_max(v) : row_max(v); }
    template <typename Tv> static __device__ __forceinline__ Tv bcast(Tv v, int src) { return __shfl(v, src, GW); }
};

// Warm start (polygon obstacles; since round 5 circles too - warm_circle): the support (non-zero pattern) of the previous (lam, mu) of this (obstacle, stage)
// usually survives from one ADMM iteration / MPC step to the next.  Lanes 0 and 1 solve that one support for the
// two hinge states, every lane then checks the optimality conditions of the FULL problem for the result
// (lam_i >= 0 : g_i + nu A_i'a^ >= 0 off the support, mu_j >= 0 : g_j >= 0 off the support, nu >= 0 for |A'lam| <= 1).
// The problem is convex, so a point that passes is a global minimiser and the enumeration is skipped; any doubt
// (sign, tolerance, no hint yet) falls back to solve_wave.  Call after prepare_wave.  `hint` is the
// candidate index il*n_mu + im remembered from the last solve of this (n, t) (-1: none); it is only a hint - whatever it
// is, a result is accepted on the certificate alone.
// k-th neighbour of a support candidate c in the list cand[0..ncand) of n rows (0 = empty, 1..n = one row, beyond = a pair that meets in
// a vertex): the supports that differ from it by ONE row - a pair's two rows, a row's pairs and the empty support, the empty support's
// rows.  -1 when there is no k-th one.
__device__ __forceinline__ int support_neighbour(int c, int n, const unsigned char *cand, int ncand, int k)
{
    if (c == 0) { int seen = 0; for (int q = 0; q < ncand; ++q) { const int v = cand[q]; if (v >= 1 && v <= n) { if (seen == k) return v; ++seen; } } return -1; }
    if (c <= n) {
        if (k == 0) return 0;
        int seen = 1;
        for (int q = 0; q < ncand; ++q) {
            const int v = cand[q];
            if (v > n) { int a, b; decode_pair(v - 1 - n, n, a, b); if (a == c - 1 || b == c - 1) { if (seen == k) return v; ++seen; } }
        }
        return -1;
    }
    int a, b; decode_pair(c - 1 - n, n, a, b);
    return k == 0 ? 1 + a : (k == 1 ? 1 + b : -1);
}

// ---- remembered support of a CIRCLE obstacle (round 5).  With lam_3 = -|a| (always tight: the cost falls with m) the problem is: min over the disc |a| <= 1 and
// mu >= 0 of phi(m) + ro2/2 |H|^2,  m = kappa0 + at'ut + l0 |at| - h'mu (l0 = -radius),  H = at + xi + G'mu,  at = R'a - convex, with a kink at a = 0.  The three
// lam candidates of the enumeration are the three places the minimiser can be: a = 0 (gradient of the smooth part inside the kink's subdifferential),
// 0 < |a| < 1 (gradient zero), |a| = 1 (gradient = -nu a, nu >= 0).  STRICT margins between them, like the strict complementarity of the polygon rows: a point on the
// border of two cases is described by two candidates and goes to the enumeration, which ranks them by (cost, id).  Round 1: the remembered (lam case, mu support) in
// both hinge states; round 2: the other two lam cases and the mu supports one row away.  Before this, every circle row went through the 64-lane enumeration on
// every ADMM iteration, one row after the other: 200 circles 658 -> 871 steps/s, 40 circles 748 -> 1134 (Python API loop, tools/experiments/circle_obstacles.py).
// A routine of its own, called by the single-ego launch form only (lammuz_body_rows<0, true>) BEHIND the polygon rows' attempt: sharing solve_wave_warm's candidate
// evaluations put the circle cases (the interior one is a Newton iteration) into the polygon rows' code - 121 instead of 53 spilled VGPRs in the dense launch form,
// N = 2000: LamMuZ 75.6 -> 82.6 us, the 64-ego fleet 99 k -> 89 k ego-steps/s (same-box A/B, tools/experiments/ab_so.sh); out of line (noinline) its 248 VGPRs
// became the register count of every kernel that can reach it (occupancy 3 -> 1).  The dense forms and the fleet keep sending their circle rows to the enumeration.
template <int GW> __device__ __forceinline__ bool warm_circle(const WaveLDS &W, const RobotLDS &Rb, const Params &P, int wlane, int hint, Sol &best)
{
    const int lane = wlane & (GW - 1), R = P.R, E = P.E, nm0 = 1 + R + R * (R - 1) / 2;
    if (hint < 0) return false;
    const int il = hint / nm0, im = hint - il * nm0;
    if (il > 2) return false;
    // Every tolerance RELATIVE to the size of the gradient's own terms (no "1 +"): in the slack regime (m > 0, H -> 0) they are all of size delta = 1e-6, and
    // the interior candidate comes from a Newton iteration that accepts its point at an ABSOLUTE 1e-9 - there it can hand back a point that is not stationary
    // at all (found by tools/soak.py --circles: |a| = 0.51 with one mu row, the minimiser had |a| = 1 and two; the enumeration drops such a candidate by its
    // cost, a certificate with absolute tolerances let it pass).  Anything that does not pass goes to the enumeration, as ever.
    auto certify = [&]() -> bool {
        const double na = best.i1 >= 0 ? hypot(best.l1, best.l2) : 0.0;
        if (na > 1.0 + 1e-12) return false;
        // a mu support row whose multiplier is (numerically) zero describes the same point as the support without it: two candidates, one optimum - the
        // enumeration ranks them by (cost, id) (found by tools/experiments/circle_warm_debug2.py: mu = (0, 3e-17, 0, 0) remembered, the enumeration's answer 0)
        if ((best.j1 >= 0 && !(best.g1 > 1e-12)) || (best.j2 >= 0 && !(best.g2 > 1e-12))) return false;
        const double phim = (best.m < 0 ? best.m : 0.0) - P.delta;
        const double dvx = P.px - W.b[0], dvy = P.py - W.b[1], l0 = W.b[2];
        const double ut0 = P.cs * dvx + P.sn * dvy, ut1 = -P.sn * dvx + P.cs * dvy;                    // R'(p - centre)
        const double at0 = P.cs * best.l1 + P.sn * best.l2, at1 = -P.sn * best.l1 + P.cs * best.l2;    // R'a
        const double g0 = phim * ut0 + P.ro2 * best.H0, g1 = phim * ut1 + P.ro2 * best.H1, ck = phim * l0;
        const double gs = fabs(phim) * (hypot(ut0, ut1) + fabs(l0)) + P.ro2 * hypot(best.H0, best.H1);
        bool pass;
        if (!(na > 0)) pass = hypot(g0, g1) < ck * (1.0 - 1e-6);
        else {
            const double gr0 = g0 + ck * at0 / na, gr1 = g1 + ck * at1 / na;
            if (na < 1.0 - 1e-6) pass = na > 1e-6 && hypot(gr0, gr1) <= 1e-7 * gs;
            else { const double nu = -(gr0 * at0 + gr1 * at1), tz = gr1 * at0 - gr0 * at1; pass = na >= 1.0 - 1e-12 && nu > 1e-6 * gs && fabs(tz) <= 1e-7 * gs; }
        }
        if (lane >= E && lane < E + R) {          // the robot side, one row per lane: mu_j >= 0 with gradient g_j - zero on the support, positive off it
            const int j = lane - E;
            const double gj = -phim * Rb.h[j] + P.ro2 * (Rb.G[j][0] * best.H0 + Rb.G[j][1] * best.H1);
            const double sj = fabs(phim * Rb.h[j]) + P.ro2 * (fabs(Rb.G[j][0] * best.H0) + fabs(Rb.G[j][1] * best.H1));
            const bool pos = (j == best.j1 && best.g1 > 0) || (j == best.j2 && best.g2 > 0);
            pass = pass && (pos ? fabs(gj) <= 1e-7 * sj : gj > 1e-6 * sj);
        }
        return Grp<GW>::ballot(!pass, wlane) == 0;
    };
    auto take = [&](const Sol &s, int src, int cid) {
        best.cost = Grp<GW>::bcast(s.cost, src); best.id = cid;
        best.m = Grp<GW>::bcast(s.m, src); best.H0 = Grp<GW>::bcast(s.H0, src); best.H1 = Grp<GW>::bcast(s.H1, src);
        best.i1 = Grp<GW>::bcast(s.i1, src); best.i2 = Grp<GW>::bcast(s.i2, src);
        best.j1 = Grp<GW>::bcast(s.j1, src); best.j2 = Grp<GW>::bcast(s.j2, src);
        best.l1 = Grp<GW>::bcast(s.l1, src); best.l2 = Grp<GW>::bcast(s.l2, src);
        best.g1 = Grp<GW>::bcast(s.g1, src); best.g2 = Grp<GW>::bcast(s.g2, src);
    };
    Sol s; s.m = 0; s.H0 = s.H1 = 0; s.i1 = s.i2 = s.j1 = s.j2 = -1; s.l1 = s.l2 = s.g1 = s.g2 = 0; s.cost = 0; s.id = 0;
    {
        bool ok = false;
        if (lane < 2) ok = eval_candidate(W, Rb, P, il, im, lane, s);
        const double m0 = Grp<GW>::bcast(s.m, 0), m1 = Grp<GW>::bcast(s.m, 1);
        const unsigned long long okb = Grp<GW>::ballot(ok, wlane);
        int src = -1;
        if ((okb & 1) && m0 >= 0) src = 0; else if ((okb & 2) && m1 < 0) src = 1;
        if (src >= 0) {
            take(s, src, 2 * (il * nm0 + im) + src);
            if (certify()) return true;
        }
    }
    if (GW >= 16) {
        const int q = (lane >> 1) & 7, c = lane & 1;
        int nil = -1, nim = -1;
        if (lane < 16) {
            if (q < 4) { nil = q < 2 ? (il + 1 + q) % 3 : -1; nim = im; }          // the other two of a = 0, |a| = 1, 0 < |a| < 1
            else { nil = il; nim = support_neighbour(im, R, Rb.muc, Rb.nmv, q - 4); }
        }
        bool ok = false;
        if (lane < 16 && nil >= 0 && nim >= 0) ok = eval_candidate(W, Rb, P, nil, nim, c, s);
        const bool valid = ok && (c == 0 ? s.m >= 0 : s.m < 0);
        const unsigned long long vb = Grp<GW>::ballot(valid, wlane);
        const bool use = valid && !(c == 1 && ((vb >> (lane - 1)) & 1ull));
        const double cost = use ? s.cost : INFINITY;
        const double cmin = -Grp<GW>::max(-cost);
        if (cmin < INFINITY) {
            const unsigned long long wb = Grp<GW>::ballot(use && cost == cmin, wlane);
            const int src = __ffsll((long long)wb) - 1;
            const int cid = Grp<GW>::bcast(2 * (nil * nm0 + nim) + c, src);
            take(s, src, cid);
            if (certify()) return true;
        }
    }
    return false;
}

template <int GW> __device__ __forceinline__ bool solve_wave_warm(WaveLDS &W, const RobotLDS &Rb, const Params &P, int wlane, int hint, Sol &best)
{
    const int lane = wlane & (GW - 1);                          // lane inside the group
    if (P.norm2 || hint < 0) return false;                      // (circle obstacles: warm_circle, where the launch form calls it)
    const int R = P.R, E = P.E, nm0 = 1 + R + R * (R - 1) / 2;
    const int il = hint / nm0, im = hint - il * nm0;            // support of the last max-clearance optimum of this (n, t)
    if (il >= 1 + E + E * (E - 1) / 2) return false;
    // ---- optimality conditions of the full problem for the (group-uniform) point in `best` -------------------------------------------
    auto certify = [&]() -> bool {
        double ax = 0, ay = 0;
        if (best.i1 >= 0) { ax += best.l1 * W.A[best.i1][0]; ay += best.l1 * W.A[best.i1][1]; }
        if (best.i2 >= 0) { ax += best.l2 * W.A[best.i2][0]; ay += best.l2 * W.A[best.i2][1]; }
        const double na = sqrt(ax * ax + ay * ay);
        if (na > 1.0 + 1e-12) return false;
        const double phim = (best.m < 0 ? best.m : 0.0) - P.delta;          // d cost / d m
        const bool tight = na >= 1.0 - 1e-9;
        const double ux = tight ? ax / na : 0.0, uy = tight ? ay / na : 0.0;
        auto glam = [&](int i) { return phim * W.q[i] + P.ro2 * (W.M[i][0] * best.H0 + W.M[i][1] * best.H1); };
        double nu = 0.0;
        if (tight) {           // multiplier of |A'lam| <= 1 from the support row with the larger A_i'a^
            int ib = best.i1;
            double d1 = best.i1 >= 0 ? W.A[best.i1][0] * ux + W.A[best.i1][1] * uy : 0.0;
            double d2 = best.i2 >= 0 ? W.A[best.i2][0] * ux + W.A[best.i2][1] * uy : 0.0;
            double db = d1;
            if (fabs(d2) > fabs(d1)) { ib = best.i2; db = d2; }
            if (ib < 0 || !(fabs(db) > 1e-12)) return false;
            nu = -glam(ib) / db;
            if (!(nu >= -1e-10)) return false;
        }
        bool pass = true;
        if (lane < E) {
            const double gi = glam(lane) + nu * (W.A[lane][0] * ux + W.A[lane][1] * uy);
            const double tol = 1e-10 * (1.0 + fabs(phim * W.q[lane]) + P.ro2 * (fabs(W.M[lane][0] * best.H0) + fabs(W.M[lane][1] * best.H1)));
            // complementarity: a support entry that came out at its bound (0) is just an inactive-side constraint
            const bool pos = (lane == best.i1 && best.l1 > 0) || (lane == best.i2 && best.l2 > 0);
            // STRICT complementarity for the rows outside the support (null rows of a padded obstacle aside): a row with a zero
            // multiplier AND a zero gradient means the optimum is described by two supports, whose closed forms agree to rounding
            // only - such a row goes to the enumeration, which ranks them by (cost, id), so that the answer does not depend on which
            // support happened to be remembered (tests/test_gpu_supports.py: flushing the cache changes no bit)
            const bool null_row = W.A[lane][0] == 0 && W.A[lane][1] == 0;
            pass = pos ? fabs(gi) <= 1e3 * tol : (null_row ? gi >= -tol : gi > tol);
        } else if (lane < E + R) {
            const int j = lane - E;
            const double gj = -phim * Rb.h[j] + P.ro2 * (Rb.G[j][0] * best.H0 + Rb.G[j][1] * best.H1);
            const double tol = 1e-10 * (1.0 + fabs(phim * Rb.h[j]) + P.ro2 * (fabs(Rb.G[j][0] * best.H0) + fabs(Rb.G[j][1] * best.H1)));
            const bool pos = (j == best.j1 && best.g1 > 0) || (j == best.j2 && best.g2 > 0);
            pass = pos ? fabs(gj) <= 1e3 * tol : gj > tol;
        }
        return Grp<GW>::ballot(!pass, wlane) == 0;
    };
    auto take = [&](const Sol &s, int src, int cid) {          // the solution lane `src` holds becomes the group's `best`
        best.cost = Grp<GW>::bcast(s.cost, src); best.id = cid;
        best.m = Grp<GW>::bcast(s.m, src); best.H0 = Grp<GW>::bcast(s.H0, src); best.H1 = Grp<GW>::bcast(s.H1, src);
        best.i1 = Grp<GW>::bcast(s.i1, src); best.i2 = Grp<GW>::bcast(s.i2, src);
        best.j1 = Grp<GW>::bcast(s.j1, src); best.j2 = Grp<GW>::bcast(s.j2, src);
        best.l1 = Grp<GW>::bcast(s.l1, src); best.l2 = Grp<GW>::bcast(s.l2, src);
        best.g1 = Grp<GW>::bcast(s.g1, src); best.g2 = Grp<GW>::bcast(s.g2, src);
    };
    Sol s; s.m = 0; s.H0 = s.H1 = 0; s.i1 = s.i2 = s.j1 = s.j2 = -1; s.l1 = s.l2 = s.g1 = s.g2 = 0; s.cost = 0; s.id = 0;
    // ---- round 1: the remembered support, lanes 0 / 1 = hinge inactive / active ------------------------------------------------------
    {
        bool ok = false;
        if (lane < 2) ok = eval_candidate(W, Rb, P, il, im, lane, s);
        // hinge-inactive solution with m >= 0, else hinge-active solution with m < 0
        const double m0 = Grp<GW>::bcast(s.m, 0), m1 = Grp<GW>::bcast(s.m, 1);
        const unsigned long long okb = Grp<GW>::ballot(ok, wlane);
        int src = -1;
        if ((okb & 1) && m0 >= 0) src = 0; else if ((okb & 2) && m1 < 0) src = 1;
        if (src >= 0) {
            take(s, src, 2 * (il * nm0 + im) + src);
            if (certify()) return true;
        }
    }
    // ---- round 2: the supports ONE row away from the remembered one (what a row that loses its support usually moves to: a vertex
    // contact becomes an edge contact or the other way round, on the obstacle's side or on the robot's - 99 % of the failures of a
    // closed loop, tools/lmz_wave_clocks.py).  Lane 2 q + c: neighbour q (0..3: a lam row changes, 4..7: a mu row), hinge state c;
    // every neighbour's own valid point (inactive with m >= 0, else active with m < 0), the cheapest of them, ONE more certificate.
    // Only a point that passes it is taken (the problem is convex: it is then a global minimiser); anything else goes to the enumeration.
    if (GW >= 16) {
        const int q = (lane >> 1) & 7, c = lane & 1;
        int nil = -1, nim = -1;
        if (lane < 16) {
            if (q < 4) { nil = support_neighbour(il, E, W.lamc, W.nlv, q); nim = im; }
            else { nil = il; nim = support_neighbour(im, R, Rb.muc, Rb.nmv, q - 4); }
        }
        bool ok = false;
        if (lane < 16 && nil >= 0 && nim >= 0) ok = eval_candidate(W, Rb, P, nil, nim, c, s);
        const bool valid = ok && (c == 0 ? s.m >= 0 : s.m < 0);
        // a support whose hinge-inactive point is valid does not also offer its hinge-active one (same rule as round 1)
        const unsigned long long vb = Grp<GW>::ballot(valid, wlane);
        const bool use = valid && !(c == 1 && ((vb >> (lane - 1)) & 1ull));
        const double cost = use ? s.cost : INFINITY;
        const double cmin = -Grp<GW>::max(-cost);
        if (cmin < INFINITY) {
            const unsigned long long wb = Grp<GW>::ballot(use && cost == cmin, wlane);
            const int src = __ffsll((long long)wb) - 1;
            const int cid = Grp<GW>::bcast(2 * (nil * nm0 + nim) + c, src);
            take(s, src, cid);
            if (certify()) return true;
        }
    }
    return false;
}

// Tie-break T1 in the slack regime (every (lam, mu) with H = 0, m >= 0 is optimal for the reference's problem): replace
// the max-clearance solution in `best` by the duals of the UNIT normal in the middle of the arc {theta : m(a(theta)) >= 0}
// of all separating directions around it.  For unit a:  m(a) = min_{k,j} [a'(p - v_k + R r_j) + xi'r_j] + kappa0 (a circle
// obstacle contributes its centre and -radius): one (obstacle vertex, robot vertex) pair per lane, its admissible arc is
// centred at the direction of w_kj with half-width acos(-c_j/|w_kj|); the feasible arc is the intersection (two wave
// minima).  Same steps as oracle/lammuz_np.py:central_normal.  All lanes call, after prepare_wave; `best` is wave-uniform.
template <int GW> __device__ __forceinline__ bool central_normal_wave(const WaveLDS &W, const RobotLDS &Rb, const Params &P, int wlane, Sol &best)
{
    const int lane = wlane & (GW - 1);
    if (!(best.m > 0) || !(best.H0 * best.H0 + best.H1 * best.H1 < 1e-8)) return false;
    double as0 = 0, as1 = 0;
    if (P.norm2) { if (best.i1 >= 0) { as0 = best.l1; as1 = best.l2; } }
    else {
        if (best.i1 >= 0) { as0 += best.l1 * W.A[best.i1][0]; as1 += best.l1 * W.A[best.i1][1]; }
        if (best.i2 >= 0) { as0 += best.l2 * W.A[best.i2][0]; as1 += best.l2 * W.A[best.i2][1]; }
    }
    if (!(as0 * as0 + as1 * as1 >= 1.0 - 1e-9)) return false;          // a* on the unit circle: it has a direction
    const int nv = P.norm2 ? 1 : W.npv, nr = Rb.nrv;
    if (nr < 3 || (!P.norm2 && nv < 3) || nv * nr > 64 || nv > GW || nr > GW) return false;
    // The oracle does this with atan2 / acos; here the same arc arithmetic is carried out on unit vectors (no
    // trigonometric calls): the end points of a pair's arc are w^ rotated by -+beta (cos beta = q), the room from a* to an
    // end point is an angle in [0, pi] iff the cross product is >= 0, and tan(angle/2) = sin/(1 + cos) orders such angles
    // accurately even when they are tiny.
    const double off = P.norm2 ? W.b[2] : 0.0;
    const double ins = rsqrt(as0 * as0 + as1 * as1), u0 = as0 * ins, u1 = as1 * ins;
    bool fail = false; double thi = INFINITY, tlo = INFINITY;           // tan(hi/2), tan(lo/2); INFINITY = the cap hi = lo = pi
    for (int pi = lane; pi < nv * nr; pi += GW) {           // one (obstacle vertex, robot vertex) pair per lane and round
        const int k = pi / nr, j = pi - k * nr;
        const double vx = P.norm2 ? W.b[0] : W.vtx[k][0], vy = P.norm2 ? W.b[1] : W.vtx[k][1];
        const double rx = Rb.rv[j][0], ry = Rb.rv[j][1];
        const double wx = P.px - vx + (P.cs * rx - P.sn * ry), wy = P.py - vy + (P.sn * rx + P.cs * ry);
        const double cj = P.xi0 * rx + P.xi1 * ry + P.kappa0 + off;
        const double n2 = wx * wx + wy * wy;
        if (!(n2 > 0)) fail = fail || cj < 0;
        else {
            const double inw = rsqrt(n2), q = -cj * inw, hx = wx * inw, hy = wy * inw;
            if (q >= 1.0) fail = true;
            else if (q > -1.0) {
                if (u0 * hx + u1 * hy < q - 1e-15) fail = true;                  // a* itself must separate at unit length
                else {
                    const double sb = sqrt((1.0 - q) * (1.0 + q));
                    const double epx = q * hx - sb * hy, epy = sb * hx + q * hy;  // w^ rotated by +beta: upper end of the arc
                    const double emx = q * hx + sb * hy, emy = -sb * hx + q * hy; // w^ rotated by -beta: lower end
                    const double cp = u0 * epx + u1 * epy, sp = u0 * epy - u1 * epx;     // angle a* -> upper end
                    const double cm = u0 * emx + u1 * emy, sm = emx * u1 - emy * u0;     // angle lower end -> a*
                    double th = INFINITY, tl = INFINITY;
                    if (sp >= 0 && cp > -1.0) th = sp / (1.0 + cp);
                    else if (sp < 0 && sp > -1e-15) th = 0.0;                     // a* on the upper end up to rounding
                    if (sm >= 0 && cm > -1.0) tl = sm / (1.0 + cm);
                    else if (sm < 0 && sm > -1e-15) tl = 0.0;
                    if (th < thi) thi = th;
                    if (tl < tlo) tlo = tl;
                }
            }
        }
    }
    if (Grp<GW>::ballot(fail, wlane)) return false;
    thi = -Grp<GW>::max(-thi); tlo = -Grp<GW>::max(-tlo);
    if (isinf(thi) && isinf(tlo)) return false;
    // a_c = a* rotated by (hi - lo)/2, from the half-angle tangents
    double chh, shh, chl, shl;
    if (isinf(thi)) { chh = 0; shh = 1; } else { const double r = rsqrt(1.0 + thi * thi); chh = r; shh = thi * r; }
    if (isinf(tlo)) { chl = 0; shl = 1; } else { const double r = rsqrt(1.0 + tlo * tlo); chl = r; shl = tlo * r; }
    const double Cr = chh * chl + shh * shl, Sr = shh * chl - chh * shl;
    const double a0 = u0 * Cr - u1 * Sr, a1 = u1 * Cr + u0 * Sr;
    // supporting duals: lam for a, mu for g = -R'a - xi  (argmax over the vertices, lowest index on ties)
    int i1 = -1, i2 = -1; double l1 = 0, l2 = 0;
    if (P.norm2) { i1 = 0; i2 = 1; l1 = a0; l2 = a1; }
    else {
        const double val = lane < nv ? a0 * W.vtx[lane < nv ? lane : 0][0] + a1 * W.vtx[lane < nv ? lane : 0][1] : -INFINITY, mx = Grp<GW>::max(val);
        const int kb = __ffsll((long long)Grp<GW>::ballot(val == mx, wlane)) - 1;
        decode_pair((int)W.lamc[kb] - 1 - P.E, P.E, i1, i2);
        const double a00 = W.A[i1][0], a01 = W.A[i1][1], a10 = W.A[i2][0], a11 = W.A[i2][1], det = a00 * a11 - a01 * a10;
        l1 = (a0 * a11 - a10 * a1) / det;            // A_S' lam_S = a
        l2 = (a00 * a1 - a0 * a01) / det;
        if (l1 < -1e-9 || l2 < -1e-9) return false;
        if (l1 < 0) l1 = 0;
        if (l2 < 0) l2 = 0;
    }
    const double gx = -(P.cs * a0 + P.sn * a1) - P.xi0, gy = -(-P.sn * a0 + P.cs * a1) - P.xi1;
    int j1, j2; double g1, g2;
    {
        const double val = lane < nr ? gx * Rb.rv[lane < nr ? lane : 0][0] + gy * Rb.rv[lane < nr ? lane : 0][1] : -INFINITY, mx = Grp<GW>::max(val);
        const int jb = __ffsll((long long)Grp<GW>::ballot(val == mx, wlane)) - 1;
        decode_pair((int)Rb.muc[jb] - 1 - P.R, P.R, j1, j2);
        const double a00 = Rb.G[j1][0], a01 = Rb.G[j1][1], a10 = Rb.G[j2][0], a11 = Rb.G[j2][1], det = a00 * a11 - a01 * a10;
        g1 = (gx * a11 - a10 * gy) / det;
        g2 = (a00 * gy - gx * a01) / det;
        if (g1 < -1e-9 || g2 < -1e-9) return false;
        if (g1 < 0) g1 = 0;
        if (g2 < 0) g2 = 0;
    }
    double m = P.kappa0 - g1 * Rb.h[j1] - g2 * Rb.h[j2];
    double H0 = P.xi0 + g1 * Rb.G[j1][0] + g2 * Rb.G[j2][0], H1 = P.xi1 + g1 * Rb.G[j1][1] + g2 * Rb.G[j2][1];
    if (P.norm2) {
        m += l1 * W.q[0] + l2 * W.q[1] - W.q[2];
        H0 += l1 * W.M[0][0] + l2 * W.M[1][0] - W.M[2][0]; H1 += l1 * W.M[0][1] + l2 * W.M[1][1] - W.M[2][1];
    } else {
        m += l1 * W.q[i1] + l2 * W.q[i2];
        H0 += l1 * W.M[i1][0] + l2 * W.M[i2][0]; H1 += l1 * W.M[i1][1] + l2 * W.M[i2][1];
    }
    if (m < 0) return false;
    best.i1 = i1; best.i2 = i2; best.l1 = l1; best.l2 = l2; best.j1 = j1; best.j2 = j2; best.g1 = g1; best.g2 = g2;
    best.m = m; best.H0 = H0; best.H1 = H1;
    return true;
}

// value of lam[e] / mu[j] encoded by a solution
__device__ __forceinline__ double lam_of(const Sol &s, int norm2, int e)
{
    if (norm2) {
        if (s.i1 < 0) return 0.0;
        return e == 0 ? s.l1 : (e == 1 ? s.l2 : (e == 2 ? -hypot(s.l1, s.l2) : 0.0));
    }
    return e == s.i1 ? s.l1 : (e == s.i2 ? s.l2 : 0.0);
}
__device__ __forceinline__ double mu_of(const Sol &s, int j) { return j == s.j1 ? s.g1 : (j == s.j2 ? s.g2 : 0.0); }

}  // namespace lmz

#pragma clang fp contract(fast)      // back to the HIP default for whatever is compiled after this header
