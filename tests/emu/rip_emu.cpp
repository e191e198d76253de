// TEST INFRASTRUCTURE: host emulation of the row-parallel interior-point LamMuZ kernel.  Instantiates the very template the HIP
// kernel runs (rda_planner_amd/csrc/lammuz_ip_device.h, rip::Solver<L>) with a 16-wide host lane vector, so that its arithmetic is
// pinned against oracle/lmz_ipm.c in the build container (tests/test_ip_rows_emu.py).  Built by the test with g++ into tests/emu/_build/.
#include <cmath>
#include <cstring>
#include "../../rda_planner_amd/csrc/lammuz_ip_device.h"

struct M16 {
    bool v[16];
    M16 operator&&(const M16 &o) const { M16 r; for (int i = 0; i < 16; ++i) r.v[i] = v[i] && o.v[i]; return r; }
    M16 operator||(const M16 &o) const { M16 r; for (int i = 0; i < 16; ++i) r.v[i] = v[i] || o.v[i]; return r; }
    M16 operator!() const { M16 r; for (int i = 0; i < 16; ++i) r.v[i] = !v[i]; return r; }
};
struct V16 {
    double v[16];
#define BINOP(op) V16 operator op(const V16 &o) const { V16 r; for (int i = 0; i < 16; ++i) r.v[i] = v[i] op o.v[i]; return r; }
    BINOP(+) BINOP(-) BINOP(*) BINOP(/)
#undef BINOP
    V16 operator-() const { V16 r; for (int i = 0; i < 16; ++i) r.v[i] = -v[i]; return r; }
#define CMP(op) M16 operator op(const V16 &o) const { M16 r; for (int i = 0; i < 16; ++i) r.v[i] = v[i] op o.v[i]; return r; }
    CMP(<) CMP(>) CMP(<=) CMP(>=) CMP(==) CMP(!=)
#undef CMP
};
struct HostLanes {
    typedef V16 V; typedef M16 M;
    static V cst(double c) { V r; for (int i = 0; i < 16; ++i) r.v[i] = c; return r; }
    static M cst_m(bool b) { M r; for (int i = 0; i < 16; ++i) r.v[i] = b; return r; }
    static M lane_eq(int k) { M r; for (int i = 0; i < 16; ++i) r.v[i] = i == k; return r; }
    static M lane_lt(int k) { M r; for (int i = 0; i < 16; ++i) r.v[i] = i < k; return r; }
    static M lane_gt(int k) { M r; for (int i = 0; i < 16; ++i) r.v[i] = i > k; return r; }
    template <int J> static V bc(const V &a) { return cst(a.v[J]); }
    template <int J> static V fma_bc(const V &acc, const V &a, const V &b) { V r; for (int i = 0; i < 16; ++i) r.v[i] = acc.v[i] + a.v[i] * b.v[J]; return r; }
    static V rsum(const V &a) { double s = 0; for (int i = 0; i < 16; ++i) s += a.v[i]; return cst(s); }
    static V rmax(const V &a) { double s = -INFINITY; for (int i = 0; i < 16; ++i) if (a.v[i] > s) s = a.v[i]; return cst(s); }
    static V rmin(const V &a) { double s = INFINITY; for (int i = 0; i < 16; ++i) if (a.v[i] < s) s = a.v[i]; return cst(s); }
    static V sel(const M &m, const V &a, const V &b) { V r; for (int i = 0; i < 16; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    static bool uni(const M &m) { return m.v[0]; }
    static M all(const M &m) { bool a = true; for (int i = 0; i < 16; ++i) a = a && m.v[i]; return cst_m(a); }
    static double first(const V &a) { return a.v[0]; }
    static V sqrt_(const V &a) { V r; for (int i = 0; i < 16; ++i) r.v[i] = std::sqrt(a.v[i]); return r; }
    static V rsqrt_(const V &a) { V r; for (int i = 0; i < 16; ++i) r.v[i] = 1.0 / std::sqrt(a.v[i]); return r; }
    static V fabs_(const V &a) { V r; for (int i = 0; i < 16; ++i) r.v[i] = std::fabs(a.v[i]); return r; }
    static V fmin_(const V &a, const V &b) { V r; for (int i = 0; i < 16; ++i) r.v[i] = std::fmin(a.v[i], b.v[i]); return r; }
    static V fmax_(const V &a, const V &b) { V r; for (int i = 0; i < 16; ++i) r.v[i] = std::fmax(a.v[i], b.v[i]); return r; }
    static M finite_(const V &a) { M r; for (int i = 0; i < 16; ++i) r.v[i] = std::isfinite(a.v[i]); return r; }
};

extern "C" int rip_emu_fits(int E, int R, int cone_norm2, int robot_norm2, int accelerated) { return rip::fits(E, R, cone_norm2, robot_norm2, accelerated) ? 1 : 0; }

// arguments as oracle/lmz_ipm.c:orc_lammuz_ipm_one; returns 0 on the central path at mu_target, 2 failed
extern "C" int rip_emu_solve(int E, int R, const double *A, const double *b, int cone_norm2, int robot_norm2,
                             const double *p, double phi, const double *G, const double *h,
                             const double *xi, double zeta, double dbar, double ro2, int accelerated, double mu_target,
                             double *lam_out, double *mu_out, double *z_out, double *x_out, double *state /* 80: x | s.d | z.d | s.g | z.g, in (if warm) / out */,
                             int warm, int *iters)
{
    rip::Problem pr;
    pr.E = E; pr.R = R; pr.cone_norm2 = cone_norm2; pr.robot_norm2 = robot_norm2; pr.accelerated = accelerated;
    pr.A = A; pr.b = b; pr.G = G; pr.h = h;
    pr.px = p[0]; pr.py = p[1]; pr.cs = std::cos(phi); pr.sn = std::sin(phi); pr.xi0 = xi[0]; pr.xi1 = xi[1];
    pr.kappa0 = zeta - dbar; pr.ro2 = ro2; pr.mu_target = mu_target;
    static rip::Solver<HostLanes> sv;
    sv.build(pr);
    if (warm) {
        V16 a, b_, c_, d_, e_;
        for (int i = 0; i < 16; ++i) { a.v[i] = state[i]; b_.v[i] = state[16 + i]; c_.v[i] = state[32 + i]; d_.v[i] = state[48 + i]; e_.v[i] = state[64 + i]; }
        sv.load(a, b_, c_, d_, e_);
    }
    const int st = sv.run(mu_target, warm != 0, iters);
    if (state && st == 0) for (int i = 0; i < 16; ++i) { state[i] = sv.x.v[i]; state[16 + i] = sv.s.d.v[i]; state[32 + i] = sv.z.d.v[i]; state[48 + i] = sv.s.g.v[i]; state[64 + i] = sv.z.g.v[i]; }
    if (x_out) for (int i = 0; i < 16; ++i) x_out[i] = sv.x.v[i];
    if (st != 0) return st;
    for (int i = 0; i < E; ++i) { double v = sv.x.v[i]; if (!cone_norm2 && v < 0) v = 0; lam_out[i] = v; }
    for (int j = 0; j < R; ++j) { double v = sv.x.v[E + j]; if (!robot_norm2 && v < 0) v = 0; mu_out[j] = v; }
    *z_out = sv.x.v[E + R] > 0 ? sv.x.v[E + R] : 0;
    return 0;
}
