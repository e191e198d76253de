// K1, interior-point variant, ROW-PARALLEL: one LamMuZ sub-problem (obstacle slot n, stage t) per 16-lane DPP row of a wavefront,
// four per wave - the cone program the reference builds for it (rda_solver.py:389-421 LamMuZ_cost_cons, :874-909 Hm_LamMu / Im_LamMu,
// :1034-1050 the cones), solved by the same primal-dual interior-point method as oracle/lmz_ipm.c / lammuz_cp_device.h (Mehrotra
// predictor-corrector, Nesterov-Todd scaling, normal equations) that ENDS ON THE CENTRAL PATH at a prescribed barrier parameter mu*
// (s o z = mu* e, residuals at rounding level): at fixed mu* the answer is a well-conditioned function of the data, so two
// implementations that reach it agree whatever their iteration paths.
//
//   x = [ lam (E) | mu (R) | z | th | tn | mm | (tl) | (tr) ]                                    n <= 16: LANE i OWNS VARIABLE i
//   minimise  1/2 th^2 + 1/2 ro2 |M'lam + G'mu + xi|^2                 (accelerated; otherwise 1/2 Im^2 instead of the th term)
//   s.t.      z >= 0 ; th >= -Im ; th >= 0 ;  (tn ; A'lam) in Q^3 ; tn <= mm ; mm <= 1
//             obstacle cone  Rpositive: lam >= 0   |  norm2: (tl ; -lam_0, -lam_1) in Q^3, tl + lam_2 <= 0 (E identical rows)
//             robot cone     Rpositive: mu >= 0    |  norm2: (tr ; -mu_0 .. -mu_{R-2}) in Q^R, tr + mu_{R-1} <= 0
//
// How the 16 lanes of a row share the work.  The constraint rows G x + s = h, s in K come in two kinds:
//   * DIAGONAL rows - one non-zero, +-1 (lam_i >= 0, mu_j >= 0, z >= 0, th >= 0, mm <= 1): at most one per variable, kept by the
//     lane of that variable (slack / multiplier `sd`, `zd`);
//   * GENERAL rows - at FIXED lanes, so that every cross-lane access is a DPP row broadcast with an immediate lane:
//       0-2 (tn ; A'lam) in Q^3 | 3-5 (tl ; -lam_0, -lam_1) in Q^3 | 6-9 (tr ; -mu_0 ..) in Q^R, R <= 4 | 10 th >= -Im | 11 tn <= mm |
//       12 tl + lam_2 <= 0 with MULTIPLICITY E (the reference writes that row E times; the copies stay equal on the whole path, so
//       one row with weight E in G'z, the normal matrix, the gap and the degree is the same iteration) | 13 tr + mu_{R-1} <= 0
//     lane k keeps row k of G (`gr`), every lane i keeps column i of the general rows (`gc`): G x and G'y are 16 broadcast-FMAs each.
// The normal matrix H = P + G'W^-2 G (16 x 16, lane i holds row i) is built from the column form of W^-1 G, factorised by a
// right-looking Cholesky whose column updates are broadcast-FMAs, and the two triangular solves per Newton system run on the
// lane-distributed right-hand side.  Everything lives in registers; no LDS, no scratch.
//
// The code is written ONCE against a small lane-vector interface `L` (V = one value per lane of the row, bc<J> = broadcast of lane J,
// rsum / rmax / rmin = row all-reduce, sel = per-lane select, uni = a row-uniform predicate as a bool): on the device V is a double
// and bc a DPP row_newbcast; tests/emu/rip_emu.cpp instantiates the very same template with a 16-wide host vector class and pins it
// against oracle/lmz_ipm.c on the CPU (tests/test_ip_rows_emu.py), so the kernel's arithmetic is checked in the build container.
#pragma once
#include <math.h>
#include <type_traits>

#ifndef RIP_HD
#ifdef __HIPCC__
#define RIP_HD __device__ __forceinline__
#else
#define RIP_HD inline
#endif
#endif

namespace rip {

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E_, class F> RIP_HD void sfor(F &&f)
{
    if constexpr (B < E_) { f(IC<B>{}); sfor<B + 1, E_>(f); }
}
template <int B, int E_, class F> RIP_HD void sfor_down(F &&f)        // E_-1, E_-2, ..., B
{
    if constexpr (B < E_) { f(IC<E_ - 1>{}); sfor_down<B, E_ - 1>(f); }
}

struct Problem {             // inputs of one sub-problem (what lmz::Params + the LDS slab carry for the enumeration)
    int E, R, cone_norm2, robot_norm2, accelerated;
    const double *A, *b;     // [E][2], [E]   (the row's LDS slab on the device)
    const double *G, *h;     // [R][2], [R]
    double px, py, cs, sn, xi0, xi1, kappa0, ro2, mu_target;
};

// fixed lanes of the general rows
constexpr int Q0 = 0, Q1 = 3, Q2 = 6, RIM = 10, RTN = 11, RTL = 12, RTR = 13;

// shape test: does the sub-problem fit the 16 lanes?  (else the per-thread solver of lammuz_cp_device.h is launched)
inline bool fits(int E, int R, int cone_any_norm2, int robot_norm2, int accelerated)
{
    const int n = E + R + 3 + (accelerated ? 1 : 0) + (cone_any_norm2 ? 1 : 0) + (robot_norm2 ? 1 : 0);
    return n <= 16 && (!robot_norm2 || R <= 4);
}

template <class L> struct Solver {
    typedef typename L::V V;
    typedef typename L::M M;
    struct Q4 { V c[4]; };            // the entries of one second-order cone (dimension <= 4, unused entries zero), same on every lane
    struct RV { V d, g; };            // a vector over the constraint rows: diagonal row of variable `lane` | general row `lane`

    // ---- problem ------------------------------------------------------------------------------------------------------------
    int n, iz, ith, itn, imm, itl, itr, E, R;
    bool hq1, hq2;                    // second-order cones 1 (circle obstacle) and 2 (norm2 robot) exist; cone 0 always does
    int deg;                          // degree of the cone (rows counted with multiplicity, one per second-order cone)
    V gc[16], gr[16];                 // column `lane` / row `lane` of the general rows
    V hg, mult;                       // rhs and multiplicity (0 = the row does not exist) of general row `lane`
    M glp, gq;                        // general row `lane` is an LP row / belongs to a second-order cone
    M gq_first;                       // ... and is the first entry of its cone
    V pv[3]; double pw[3];            // P = sum_c pw[c] pv[c] pv[c]'   (pv[c] = entry of variable `lane`)
    V qx;                             // linear cost
    V dsg, dh; M dact;                // diagonal row of variable `lane`: sign, rhs, exists
    M xdead;                          // variable `lane` enters nothing (beyond n, or the multiplier of a zero-padded edge row): stays 0
    V cvx; V qv_, mv0, mv1;           // Im = cv'x + kappa0 ; (q, M) of the lam entries (outputs)
    double kappa0, xi0, xi1;
    // ---- iterate ------------------------------------------------------------------------------------------------------------
    V x; RV s, z;
    // ---- scaling / factorisation ----------------------------------------------------------------------------------------------
    RV wd;                            // LP rows: sqrt(s / z)
    Q4 ww[3]; V wbeta[3];             // Nesterov-Todd scaling of the three cones
    V hm[16];                         // row `lane` of the normal matrix, then of its Cholesky factor
    V rdiag;                          // 1 / L_jj on lane j

    RIP_HD static V cst(double c) { return L::cst(c); }
    RIP_HD static M lane_is(int k) { return L::lane_eq(k); }

    // ---- cone gathers -------------------------------------------------------------------------------------------------------
    template <int B> RIP_HD static Q4 gather(const V &g)
    {
        Q4 q; q.c[0] = L::template bc<B>(g); q.c[1] = L::template bc<B + 1>(g); q.c[2] = L::template bc<B + 2>(g);
        q.c[3] = B == Q2 ? L::template bc<(B == Q2 ? B + 3 : B)>(g) : cst(0.0);
        return q;
    }
    template <int B> RIP_HD static V scatter(const Q4 &q, const V &g)
    {
        V o = L::sel(lane_is(B), q.c[0], g);
        o = L::sel(lane_is(B + 1), q.c[1], o); o = L::sel(lane_is(B + 2), q.c[2], o);
        if (B == Q2) o = L::sel(lane_is(B + 3), q.c[3], o);
        return o;
    }
    RIP_HD static V jdet(const Q4 &u) { return u.c[0] * u.c[0] - (u.c[1] * u.c[1] + u.c[2] * u.c[2] + u.c[3] * u.c[3]); }
    RIP_HD static V dot1(const Q4 &u, const Q4 &v) { return u.c[1] * v.c[1] + u.c[2] * v.c[2] + u.c[3] * v.c[3]; }
    // general row k exists for this sub-problem's cone configuration (row-uniform)
    template <int k> RIP_HD bool row_on() const
    {
        if (k >= Q1 && k < Q1 + 3) return hq1;
        if (k >= Q2 && k < Q2 + 4) return hq2 && k - Q2 < R;
        if (k == RIM) return ith >= 0;
        if (k == RTL) return hq1;
        if (k == RTR) return hq2;
        return k < 14;
    }
    template <class F> RIP_HD void each_cone(F &&f) const
    {
        f(IC<Q0>{}, IC<0>{});
        if (hq1) f(IC<Q1>{}, IC<1>{});
        if (hq2) f(IC<Q2>{}, IC<2>{});
    }

    // ---- Jordan algebra on row vectors -----------------------------------------------------------------------------------------
    RIP_HD RV jprod(const RV &u, const RV &v) const
    {
        RV o; o.d = u.d * v.d; o.g = u.g * v.g;
        each_cone([&](auto B, auto) {
            constexpr int b = decltype(B)::value;
            const Q4 a = gather<b>(u.g), c = gather<b>(v.g);
            Q4 r; r.c[0] = a.c[0] * c.c[0] + dot1(a, c);
            for (int i = 1; i < 4; ++i) r.c[i] = a.c[0] * c.c[i] + c.c[0] * a.c[i];
            o.g = scatter<b>(r, o.g);
        });
        return o;
    }
    RIP_HD RV jdiv(const RV &lam, const RV &bv) const        // lam o u = bv
    {
        RV o; o.d = L::sel(dact, bv.d / lam.d, cst(0.0)); o.g = L::sel(glp, bv.g / lam.g, cst(0.0));
        each_cone([&](auto B, auto) {
            constexpr int b = decltype(B)::value;
            const Q4 l = gather<b>(lam.g), c = gather<b>(bv.g);
            const V det = jdet(l), l1b1 = dot1(l, c);
            Q4 r; r.c[0] = (l.c[0] * c.c[0] - l1b1) / det;
            for (int i = 1; i < 4; ++i) r.c[i] = (-l.c[i] * c.c[0] + (det * c.c[i] + l.c[i] * l1b1) / l.c[0]) / det;
            o.g = scatter<b>(r, o.g);
        });
        return o;
    }
    // Nesterov-Todd scaling of (s, z); false: not interior
    RIP_HD bool nt_compute()
    {
        const M okd = !dact || (s.d > cst(0.0) && z.d > cst(0.0)), okg = !glp || (s.g > cst(0.0) && z.g > cst(0.0));
        bool ok = L::uni(L::all(okd && okg));
        wd.d = L::sel(dact, L::sqrt_(s.d / z.d), cst(1.0)); wd.g = L::sel(glp, L::sqrt_(s.g / z.g), cst(1.0));
        each_cone([&](auto B, auto C) {
            constexpr int b = decltype(B)::value, c = decltype(C)::value;
            const Q4 sq = gather<b>(s.g), zq = gather<b>(z.g);
            const V ds = jdet(sq), dz = jdet(zq);
            if (!L::uni(ds > cst(0.0) && dz > cst(0.0) && sq.c[0] > cst(0.0) && zq.c[0] > cst(0.0))) { ok = false; return; }
            const V ns = L::sqrt_(ds), nz = L::sqrt_(dz);
            V g = cst(0.0);
            for (int i = 0; i < 4; ++i) g = g + (sq.c[i] / ns) * (zq.c[i] / nz);
            g = L::sqrt_(cst(0.5) * (cst(1.0) + g));
            ww[c].c[0] = (sq.c[0] / ns + zq.c[0] / nz) / (cst(2.0) * g);
            for (int i = 1; i < 4; ++i) ww[c].c[i] = (sq.c[i] / ns - zq.c[i] / nz) / (cst(2.0) * g);
            wbeta[c] = L::sqrt_(ns / nz);
        });
        return ok;
    }
    // one cone: W u (inverse = false) or W^-1 u
    template <int c> RIP_HD Q4 nt_cone(const Q4 &u, bool inverse) const
    {
        const Q4 &w = ww[c];
        const V sg = cst(inverse ? -1.0 : 1.0), w1u1 = dot1(w, u);
        const V r0 = w.c[0] * u.c[0] + sg * w1u1, f = w1u1 / (cst(1.0) + w.c[0]);
        const V sc = inverse ? cst(1.0) / wbeta[c] : wbeta[c];
        Q4 o; o.c[0] = r0 * sc;
        for (int i = 1; i < 4; ++i) o.c[i] = (sg * w.c[i] * u.c[0] + u.c[i] + w.c[i] * f) * sc;
        return o;
    }
    RIP_HD RV nt_apply(const RV &u, bool inverse) const
    {
        RV o;
        o.d = inverse ? u.d / wd.d : u.d * wd.d;
        o.g = inverse ? u.g / wd.g : u.g * wd.g;
        each_cone([&](auto B, auto C) {
            constexpr int b = decltype(B)::value, c = decltype(C)::value;
            o.g = scatter<b>(nt_cone<c>(gather<b>(u.g), inverse), o.g);
        });
        return o;
    }
    RIP_HD V min_eig(const RV &u) const
    {
        V v = L::rmin(L::sel(dact, u.d, cst(INFINITY)));
        v = L::fmin_(v, L::rmin(L::sel(glp, u.g, cst(INFINITY))));
        each_cone([&](auto B, auto) {
            constexpr int b = decltype(B)::value;
            const Q4 q = gather<b>(u.g);
            v = L::fmin_(v, q.c[0] - L::sqrt_(dot1(q, q)));
        });
        return v;
    }
    RIP_HD V max_step(const RV &u, const RV &du) const
    {
        const V inf = cst(INFINITY);
        V a = L::rmin(L::sel(dact && du.d < cst(0.0), -u.d / du.d, inf));
        a = L::fmin_(a, L::rmin(L::sel(glp && du.g < cst(0.0), -u.g / du.g, inf)));
        each_cone([&](auto B, auto) {
            constexpr int b = decltype(B)::value;
            const Q4 q = gather<b>(u.g), dq = gather<b>(du.g);
            const V qa = jdet(dq), cc = jdet(q), bb = q.c[0] * dq.c[0] - dot1(q, dq);
            // roots of qa t^2 + 2 bb t + cc = 0 (the boundary of the cone along the step), the smallest positive one
            const M lin = L::fabs_(qa) < cst(1e-300);
            const V tl = L::sel(bb < cst(0.0), -cc / (cst(2.0) * bb), inf);
            const V disc = bb * bb - qa * cc;
            const V sq = L::sqrt_(L::fmax_(disc, cst(0.0)));
            const V t = -(bb + L::sel(bb < cst(0.0), -sq, sq));
            const V r1 = t / qa, r2 = L::sel(t != cst(0.0), cc / t, inf);
            V tq = inf;
            tq = L::sel(disc >= cst(0.0) && r1 > cst(0.0), L::fmin_(tq, r1), tq);
            tq = L::sel(disc >= cst(0.0) && r2 > cst(0.0), L::fmin_(tq, r2), tq);
            a = L::fmin_(a, L::sel(lin, tl, tq));
            a = L::fmin_(a, L::sel(dq.c[0] < cst(0.0), -q.c[0] / dq.c[0], inf));
        });
        return a;
    }

    // ---- products with G and P ---------------------------------------------------------------------------------------------
    RIP_HD V Gt(const RV &y) const           // G' y (rows weighted with their multiplicity)
    {
        const V yg = y.g * mult;
        V v = L::sel(dact, dsg * y.d, cst(0.0));
        sfor<0, 14>([&](auto K) { constexpr int k = decltype(K)::value; if (row_on<k>()) v = L::template fma_bc<k>(v, gc[k], yg); });
        return v;
    }
    RIP_HD RV Gx(const V &xv) const
    {
        RV o; o.d = L::sel(dact, dsg * xv, cst(0.0));
        V acc = cst(0.0);
        sfor<0, 16>([&](auto J) { constexpr int j = decltype(J)::value; acc = L::template fma_bc<j>(acc, gr[j], xv); });
        o.g = acc;
        return o;
    }
    RIP_HD V Pmul(const V &xv) const
    {
        V o = cst(0.0);
        sfor<0, 3>([&](auto C) { constexpr int c = decltype(C)::value; if (pw[c] != 0.0) o = o + cst(pw[c]) * pv[c] * L::rsum(pv[c] * xv); });
        return o;
    }

    // ---- normal matrix H = P + G' W^-2 G + REG I, Cholesky, solves ------------------------------------------------------------------
    // scaled = false: W = I (the starting point)
    RIP_HD bool factor(bool scaled)
    {
        // column `lane` of W^-1 G over the general rows
        V gw[16];
        const V wrow = scaled ? L::sqrt_(mult) / wd.g : L::sqrt_(mult);          // LP rows: sqrt(mult) / d   (0 for rows that do not exist)
        sfor<0, 16>([&](auto K) { constexpr int k = decltype(K)::value; gw[k] = (k < 14 && row_on<(k < 14 ? k : 0)>()) ? gc[k] * L::template bc<k>(wrow) : cst(0.0); });
        if (scaled) each_cone([&](auto B, auto C) {
            constexpr int b = decltype(B)::value, c = decltype(C)::value;
            Q4 u; u.c[0] = gc[b]; u.c[1] = gc[b + 1]; u.c[2] = gc[b + 2]; u.c[3] = b == Q2 ? gc[b == Q2 ? b + 3 : b] : cst(0.0);
            const Q4 o = nt_cone<c>(u, true);
            gw[b] = o.c[0]; gw[b + 1] = o.c[1]; gw[b + 2] = o.c[2];
            if (b == Q2) gw[b == Q2 ? b + 3 : b] = o.c[3];
        });
        // (cone rows unscaled: gw = gc there already, mult = 1)
        const V dd = L::sel(dact, scaled ? cst(1.0) / (wd.d * wd.d) : cst(1.0), cst(0.0));
        sfor<0, 16>([&](auto J) {
            constexpr int j = decltype(J)::value;
            V acc = L::sel(lane_is(j), dd + cst(1e-11) + L::sel(L::lane_lt(n), cst(0.0), cst(1.0)), cst(0.0));     // unused variables: identity
            sfor<0, 3>([&](auto C) { constexpr int c = decltype(C)::value; if (pw[c] != 0.0) acc = acc + cst(pw[c]) * pv[c] * L::template bc<j>(pv[c]); });
            sfor<0, 14>([&](auto K) { constexpr int k = decltype(K)::value; if (row_on<k>()) acc = L::template fma_bc<j>(acc, gw[k], gw[k]); });
            hm[j] = acc;
        });
        // right-looking Cholesky: lane i holds row i; the upper triangle fills with values nobody reads
        bool ok = true;
        rdiag = cst(1.0);
        sfor<0, 16>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const V pj = L::template bc<j>(hm[j]);
            if (!L::uni(pj > cst(0.0))) ok = false;
            const V rj = L::rsqrt_(L::fmax_(pj, cst(1e-300)));
            rdiag = L::sel(lane_is(j), rj, rdiag);
            const V lj = hm[j] * rj;
            hm[j] = lj;
            sfor<j + 1, 16>([&](auto K) { constexpr int k = decltype(K)::value; hm[k] = L::template fma_bc<k>(hm[k], -lj, lj); });
        });
        return ok;
    }
    RIP_HD V solve(V bvec) const             // H y = b, b and y distributed over the lanes
    {
        sfor<0, 16>([&](auto J) {             // L y = b
            constexpr int j = decltype(J)::value;
            const V yj = L::template bc<j>(bvec * rdiag);
            bvec = L::sel(L::lane_gt(j), bvec - hm[j] * yj, L::sel(lane_is(j), yj, bvec));
        });
        sfor_down<0, 16>([&](auto J) {        // L' x = y
            constexpr int j = decltype(J)::value;
            const V sm = L::rsum(L::sel(L::lane_gt(j), hm[j] * bvec, cst(0.0)));
            const V xj = (L::template bc<j>(bvec) - sm) * L::template bc<j>(rdiag);
            bvec = L::sel(lane_is(j), xj, bvec);
        });
        return bvec;
    }

    // ---- the cone program of one (obstacle, stage) ----------------------------------------------------------------------------
    RIP_HD void build(const Problem &p)
    {
        E = p.E; R = p.R;
        iz = E + R; ith = p.accelerated ? iz + 1 : -1; itn = iz + 1 + (p.accelerated ? 1 : 0); imm = itn + 1;
        n = imm + 1;
        itl = p.cone_norm2 ? n++ : -1; itr = p.robot_norm2 ? n++ : -1;
        hq1 = p.cone_norm2 != 0; hq2 = p.robot_norm2 != 0;
        kappa0 = p.kappa0; xi0 = p.xi0; xi1 = p.xi1;
        const V zero = cst(0.0);
        // per-variable data: lane i picks its own entry (per-lane loads of the E + R rows; the rest are constants)
        V ai0 = zero, ai1 = zero, bi = zero, gj0 = zero, gj1 = zero, hj = zero;
        for (int i = 0; i < E; ++i) { const M m = lane_is(i); ai0 = L::sel(m, cst(p.A[2 * i]), ai0); ai1 = L::sel(m, cst(p.A[2 * i + 1]), ai1); bi = L::sel(m, cst(p.b[i]), bi); }
        for (int j = 0; j < R; ++j) { const M m = lane_is(E + j); gj0 = L::sel(m, cst(p.G[2 * j]), gj0); gj1 = L::sel(m, cst(p.G[2 * j + 1]), gj1); hj = L::sel(m, cst(p.h[j]), hj); }
        const M islam = L::lane_lt(E), ismu = L::lane_lt(E + R) && !islam;
        qv_ = L::sel(islam, ai0 * cst(p.px) + ai1 * cst(p.py) - bi, zero);
        mv0 = L::sel(islam, ai0 * cst(p.cs) + ai1 * cst(p.sn), zero);
        mv1 = L::sel(islam, -ai0 * cst(p.sn) + ai1 * cst(p.cs), zero);
        const V b0 = L::sel(islam, mv0, L::sel(ismu, gj0, zero)), b1 = L::sel(islam, mv1, L::sel(ismu, gj1, zero));
        cvx = L::sel(islam, qv_, L::sel(ismu, -hj, L::sel(lane_is(iz), cst(-1.0), zero)));
        pv[0] = b0; pw[0] = p.ro2; pv[1] = b1; pw[1] = p.ro2;
        qx = cst(p.ro2) * (b0 * cst(p.xi0) + b1 * cst(p.xi1));
        if (p.accelerated) { pv[2] = L::sel(lane_is(ith), cst(1.0), zero); pw[2] = 1.0; }
        else { pv[2] = cvx; pw[2] = 1.0; qx = qx + cst(p.kappa0) * cvx; }
        // diagonal rows.  Zero-padded edge rows carry a multiplier that enters nothing and has no central value: left unconstrained at 0
        const M nullrow = islam && ai0 == zero && ai1 == zero && bi == zero;
        xdead = nullrow || !L::lane_lt(n);
        dact = (islam && !nullrow && !cst_m(p.cone_norm2 != 0)) || (ismu && !cst_m(p.robot_norm2 != 0)) || lane_is(iz) || lane_is(ith) || lane_is(imm);
        dsg = L::sel(lane_is(imm), cst(1.0), cst(-1.0));
        dh = L::sel(lane_is(imm), cst(1.0), zero);
        // general rows: column form gc[k] (entry of variable `lane`) and row form gr[j] (row `lane`, variable j)
        sfor<0, 16>([&](auto K) { constexpr int k = decltype(K)::value; gc[k] = zero; gr[k] = zero; });
        // cone 0: (tn ; A'lam):  rows -e_tn, -A[:,0]' lam, -A[:,1]' lam
        gc[Q0] = L::sel(lane_is(itn), cst(-1.0), zero);
        gc[Q0 + 1] = L::sel(islam, -ai0, zero);
        gc[Q0 + 2] = L::sel(islam, -ai1, zero);
        if (hq1) {      // (tl ; -lam_0, -lam_1): rows -e_tl, +e_0, +e_1
            gc[Q1] = L::sel(lane_is(itl), cst(-1.0), zero);
            gc[Q1 + 1] = L::sel(lane_is(0), cst(1.0), zero);
            gc[Q1 + 2] = L::sel(lane_is(1), cst(1.0), zero);
        }
        if (hq2) {      // (tr ; -mu_0 .. -mu_{R-2})
            gc[Q2] = L::sel(lane_is(itr), cst(-1.0), zero);
            sfor<0, 3>([&](auto Jx) { constexpr int j = decltype(Jx)::value; if (j < R - 1) gc[Q2 + 1 + j] = L::sel(lane_is(E + j), cst(1.0), zero); });
        }
        if (p.accelerated) gc[RIM] = L::sel(lane_is(ith), cst(-1.0), -cvx);        // th >= -Im:  -cv'x - th <= kappa0
        gc[RTN] = L::sel(lane_is(itn), cst(1.0), L::sel(lane_is(imm), cst(-1.0), zero));
        if (hq1) gc[RTL] = L::sel(lane_is(itl), cst(1.0), L::sel(lane_is(2), cst(1.0), zero));
        if (hq2) gc[RTR] = L::sel(lane_is(itr), cst(1.0), L::sel(lane_is(E + R - 1), cst(1.0), zero));
        // row form by transposition: gr[j] on lane k = gc[k] on lane j
        sfor<0, 16>([&](auto J) {
            constexpr int j = decltype(J)::value;
            V acc = zero;
            sfor<0, 14>([&](auto K) { constexpr int k = decltype(K)::value; acc = L::sel(lane_is(k), L::template bc<j>(gc[k]), acc); });
            gr[j] = acc;
        });
        hg = L::sel(lane_is(RIM), cst(p.accelerated ? p.kappa0 : 0.0), zero);
        const M r_q0 = L::lane_lt(Q0 + 3), r_q1 = cst_m(hq1) && L::lane_lt(Q1 + 3) && !L::lane_lt(Q1),
                r_q2 = cst_m(hq2) && L::lane_lt(Q2 + (R < 4 ? R : 4)) && !L::lane_lt(Q2);
        gq = r_q0 || r_q1 || r_q2;
        gq_first = lane_is(Q0) || (cst_m(hq1) && lane_is(Q1)) || (cst_m(hq2) && lane_is(Q2));
        glp = (cst_m(p.accelerated != 0) && lane_is(RIM)) || lane_is(RTN) || (cst_m(hq1) && lane_is(RTL)) || (cst_m(hq2) && lane_is(RTR));
        mult = L::sel(gq || glp, cst(1.0), zero);
        if (hq1) mult = L::sel(lane_is(RTL), cst((double)E), mult);
        // degree: rows with multiplicity + one per second-order cone
        const V cntd = L::rsum(L::sel(dact, cst(1.0), zero)) + L::rsum(L::sel(glp, mult, zero));
        deg = (int)(L::first(cntd) + 0.5) + 1 + (hq1 ? 1 : 0) + (hq2 ? 1 : 0);
    }
    RIP_HD static M cst_m(bool b) { return L::cst_m(b); }

    RIP_HD RV evec() const { RV e; e.d = L::sel(dact, cst(1.0), cst(0.0)); e.g = L::sel(glp || gq_first, cst(1.0), cst(0.0)); return e; }
    RIP_HD RV hvec() const { RV h; h.d = L::sel(dact, dh, cst(0.0)); h.g = hg; return h; }
    RIP_HD static RV axpy(const V &a, const RV &x_, const RV &y) { RV o; o.d = a * x_.d + y.d; o.g = a * x_.g + y.g; return o; }
    RIP_HD V rdot(const RV &a, const RV &b) const { return L::rsum(a.d * b.d) + L::rsum(a.g * b.g * mult); }     // with multiplicity
    RIP_HD V rmaxabs(const RV &a) const { return L::fmax_(L::rmax(L::sel(dact, L::fabs_(a.d), cst(0.0))), L::rmax(L::sel(glp || gq, L::fabs_(a.g), cst(0.0)))); }

    // take over a kept central-path point as the start of a warm run (after build): entries that do not exist in THIS problem are
    // cleared - the slot may have held another obstacle last time (re-ordered list: a triangle where a quadrilateral was, a circle
    // where a polygon was); rows that exist now but did not then arrive as zeros, which nt_compute rejects -> cold start
    RIP_HD void load(const V &x_, const V &sd, const V &zd, const V &sg, const V &zg)
    {
        const V zero = cst(0.0);
        x = L::sel(xdead, zero, x_);
        s.d = L::sel(dact, sd, zero); z.d = L::sel(dact, zd, zero);
        s.g = L::sel(glp || gq, sg, zero); z.g = L::sel(glp || gq, zg, zero);
    }
    // returns 0 (on the central path at mu_target), 2 failed.  warm: (x, s, z) hold an interior point - the central-path point of the
    // previous solve of this (slot, stage): the cones are the same ones, so it is interior for the new data too, its gap is deg mu*,
    // and the iteration is in its centring phase from the first step (one factorisation + ONE solve per iteration, Newton on the
    // mu*-perturbed optimality conditions with full residual reduction): 2 - 4 iterations where consecutive problems are close.  A warm
    // run that has not arrived after `warm_cap` iterations (or leaves the cone numerically) gives way to the cold start.
    RIP_HD int run(double mu_target, bool warm = false, int *iters = nullptr)
    {
        if (iters) *iters = 0;
        if (warm) {
            const int st = iterate(mu_target, 12, iters);
            if (st == 0) return 0;
        }
        const RV e = evec(), h = hvec();
        // ---- starting point: x = argmin 1/2 x'Px + q'x + 1/2 |Gx - h|^2, z = Gx - h, s = -z, both shifted into the cone ------------
        if (!factor(false)) return 2;
        x = solve(Gt(h) - qx);
        { const RV gx = Gx(x); z.d = gx.d - h.d; z.g = L::sel(glp || gq, gx.g - h.g, cst(0.0)); s.d = -z.d; s.g = -z.g; }
        {
            const V ns = L::sqrt_(rdot(s, s));
            const V ts = -min_eig(s), tz = -min_eig(z), thr = cst(-1e-8) * L::fmax_(ns, cst(1.0));
            if (L::uni(ts >= thr)) s = axpy(cst(1.0) + ts, e, s);
            if (L::uni(tz >= thr)) z = axpy(cst(1.0) + tz, e, z);
        }
        return iterate(mu_target, 60, iters);
    }
    RIP_HD int iterate(double mu_target, int it_cap, int *iters)
    {
        const RV e = evec(), h = hvec();
        const V nqn = L::rmax(L::sel(L::lane_lt(n), cst(1.0) + L::fabs_(qx), cst(1.0)));
        const V nhn = L::fmax_(cst(1.0), cst(1.0) + rmaxabs(h));
        const V mut = cst(mu_target);
        for (int it = 0; it < it_cap; ++it) {
            if (iters) *iters += 1;
            const V rx = L::sel(L::lane_lt(n), qx + Pmul(x) + Gt(z), cst(0.0));
            RV rz; { const RV gx = Gx(x); rz.d = L::sel(dact, s.d - h.d + gx.d, cst(0.0)); rz.g = L::sel(glp || gq, s.g - h.g + gx.g, cst(0.0)); }
            const V gap = rdot(s, z);
            const V dres = L::rmax(L::fabs_(rx)) / nqn, pres = rmaxabs(rz) / nhn;
            if (!nt_compute()) return 2;
            const RV lam = nt_apply(z, false);
            const RV ll = jprod(lam, lam);
            const bool centring = L::uni(gap / cst((double)deg) <= cst(10.0) * mut);
            if (centring) {
                const V cent = rmaxabs(axpy(-mut, e, ll));
                if (L::uni(dres <= cst(1e-10) && pres <= cst(1e-10) && cent <= cst(1e-7) * mut)) return 0;
            }
            if (!factor(true)) return 2;
            V sigma = cst(0.0), mu = gap / cst((double)deg);
            if (centring) { sigma = cst(1.0); mu = mut; }
            RV dsa, dza; dsa.d = dsa.g = dza.d = dza.g = cst(0.0);
            V dx = cst(0.0); RV ds, dz; ds = dsa; dz = dza;
            bool bad = false;
            for (int pass = centring ? 1 : 0; pass < 2 && !bad; ++pass) {
                const V sc = (pass && !centring) ? cst(1.0) - sigma : cst(1.0);
                RV bsv;
                if (!pass) bsv = axpy(cst(-1.0), ll, RV{cst(0.0), cst(0.0)});
                else if (centring) bsv = axpy(mu, e, axpy(cst(-1.0), ll, RV{cst(0.0), cst(0.0)}));
                else {
                    const RV pr = jprod(nt_apply(dsa, true), nt_apply(dza, false));
                    bsv = axpy(sigma * mu, e, axpy(cst(-1.0), pr, axpy(cst(-1.0), ll, RV{cst(0.0), cst(0.0)})));
                }
                const RV u = jdiv(lam, bsv);
                const RV wu = nt_apply(u, false);
                RV t; t.d = -sc * rz.d - wu.d; t.g = -sc * rz.g - wu.g;
                const RV wt = nt_apply(t, true);                   // W^-1 t
                // rhs = -sc rx + (W^-1 G)' wt = -sc rx + G' (W^-1 wt)
                const V rhs = L::sel(L::lane_lt(n), -sc * rx + Gt(nt_apply(wt, true)), cst(0.0));
                const V dxp = solve(rhs);
                // gd = (W^-1 G) dx - wt ;  dz = W^-1 gd ;  ds = W (u - gd)
                RV gd; { const RV gdx = nt_apply(Gx(dxp), true); gd.d = L::sel(dact, gdx.d - wt.d, cst(0.0)); gd.g = L::sel(glp || gq, gdx.g - wt.g, cst(0.0)); }
                const RV pdz = nt_apply(gd, true);
                RV v1; v1.d = u.d - gd.d; v1.g = u.g - gd.g;
                const RV pds = nt_apply(v1, false);
                if (!L::uni(L::all(L::finite_(pdz.d) && L::finite_(pdz.g) && L::finite_(pds.d) && L::finite_(pds.g)))) { bad = true; break; }
                if (!pass) {
                    dsa = pds; dza = pdz;
                    V aa = L::fmin_(cst(1.0), L::fmin_(max_step(s, dsa), max_step(z, dza)));
                    sigma = (cst(1.0) - aa) * (cst(1.0) - aa) * (cst(1.0) - aa);
                } else { dx = dxp; ds = pds; dz = pdz; }
            }
            if (bad) return 2;
            V a = L::fmin_(max_step(s, ds), max_step(z, dz)) * cst(0.99);
            a = L::fmin_(a, cst(1.0));
            if (!L::uni(a > cst(0.0) && L::finite_(a))) return 2;
            x = x + a * dx;
            s = axpy(a, ds, s); z = axpy(a, dz, z);
        }
        return 2;
    }
};


#ifdef __HIPCC__
// Device lanes: one value per lane of a 16-lane DPP row; every cross-lane step is a DPP operation inside the row, so the four rows of
// a wavefront run four different sub-problems and may diverge from each other (row-uniform branches).
struct DevLanes {
    typedef double V;
    typedef bool M;
    static __device__ __forceinline__ int lid() { return (int)(threadIdx.x & 15u); }
    static __device__ __forceinline__ V cst(double c) { return c; }
    static __device__ __forceinline__ M cst_m(bool b) { return b; }
    static __device__ __forceinline__ M lane_eq(int k) { return lid() == k; }
    static __device__ __forceinline__ M lane_lt(int k) { return lid() < k; }
    static __device__ __forceinline__ M lane_gt(int k) { return lid() > k; }
    template <int CTRL> static __device__ __forceinline__ V dpp(V v)
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    template <int J> static __device__ __forceinline__ V bc(V v) { return dpp<0x150 + J>(v); }          // row_newbcast:J
    // acc + a * (lane J's b): v_fmac_f64 with a DPP row_newbcast source operand (gfx90a+: DPP on 64-bit VALU ops for row_newbcast);
    // s_nop 1: a VGPR written by the previous VALU op needs two wait states before a DPP read
    template <int J> static __device__ __forceinline__ V fma_bc(V acc, V a, V b)
    {
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(J));
        return acc;
    }
    static __device__ __forceinline__ V rsum(V v) { v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v); return v; }
    static __device__ __forceinline__ V rmax(V v) { v = fmax(v, dpp<0xB1>(v)); v = fmax(v, dpp<0x4E>(v)); v = fmax(v, dpp<0x141>(v)); v = fmax(v, dpp<0x140>(v)); return v; }
    static __device__ __forceinline__ V rmin(V v) { v = fmin(v, dpp<0xB1>(v)); v = fmin(v, dpp<0x4E>(v)); v = fmin(v, dpp<0x141>(v)); v = fmin(v, dpp<0x140>(v)); return v; }
    static __device__ __forceinline__ V sel(M m, V a, V b) { return m ? a : b; }
    static __device__ __forceinline__ bool uni(M m) { return m; }
    static __device__ __forceinline__ M all(M m) { return ((__ballot(m) >> (threadIdx.x & 48u)) & 0xffffull) == 0xffffull; }
    static __device__ __forceinline__ double first(V a) { return a; }
    static __device__ __forceinline__ V sqrt_(V a) { return sqrt(a); }
    // 1 / sqrt(a): v_rsq_f64 + two Newton steps (full double accuracy for the well-scaled pivots it is used on)
    static __device__ __forceinline__ V rsqrt_(V a)
    {
        V y = __builtin_amdgcn_rsq(a);
        y = y * (1.5 - 0.5 * a * y * y); y = y * (1.5 - 0.5 * a * y * y);
        return y;
    }
    static __device__ __forceinline__ V fabs_(V a) { return fabs(a); }
    static __device__ __forceinline__ V fmin_(V a, V b) { return fmin(a, b); }
    static __device__ __forceinline__ V fmax_(V a, V b) { return fmax(a, b); }
    static __device__ __forceinline__ M finite_(V a) { return isfinite(a); }
};
#endif

}  // namespace rip
