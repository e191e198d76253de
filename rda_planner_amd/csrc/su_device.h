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
//                 stage gradients / Hessian bases, closed-loop sweep matrices, slack and multiplier
//                 updates, reductions
//   wave 0      : the serial Riccati sweeps, LANE-PARALLEL.  Matrix recursion: lane 8r+q owns entry
//                 (r,q) of the 8x8 stage matrix M = Hb + F'PF; the two small products exchange
//                 operands with ds_bpermute, the 3x3 pivot block is broadcast with v_readlane and
//                 inverted by the adjugate on every lane.  Vector sweeps: one affine map per stage,
//                 x+ = Mrow . x + c, lane r owns row r, x is broadcast with v_readlane (16 VALU
//                 instructions per stage).  All per-stage constants are prefetched one stage ahead.
//   wave 1      : adjoint sweep for the reduced gradient, concurrently with wave 0
//   wave 2      : Newton right-hand side of the predictor, concurrently with wave 0
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#ifndef SIGMA_FLOOR
#define SIGMA_FLOOR 1e-3
#endif
#define SU_CENTRE_GAMMA 1e-5     // cold attempts still running after SU_CENTRE_FROM iterations: lam w >= SU_CENTRE_GAMMA mu after every step (= oracle/rda_oracle.c)
#define SU_CENTRE_FROM 25
#define SU_SMOOTH_K 0.1          // ... and smooths its hinge terms over SU_SMOOTH_K sqrt(mu)
// Last-resort attempt (attempt 1, round 5; = oracle/rda_oracle.c): plain long-step path following - no predictor, a fixed centring parameter (smaller once
// the steps are nearly full), a shorter fraction to the boundary, every pair kept in the wide neighbourhood lam w >= SU_SAFE_GAMMA mu after each step.
// Mehrotra's heuristics can cycle on this problem class (two rows trading places with steps of 0.02 / 0.6 for ever:
// tests/golden/su_hard/omni_T15_N51_rate_and_distance_rows_cycle.npz - both former attempts of the oracle ran into their caps); this iteration has the
// textbook guarantee (16 - 22 iterations on every recorded hard instance) and is only reached when the other attempts have failed.
#define SU_COLD_CAP 50
#define SU_SAFE_SIGMA 0.3
#define SU_SAFE_SIGMA_END 0.05
#define SU_SAFE_TAU 0.9
#define SU_SAFE_GAMMA 1e-2
// fp contraction per source expression, not per optimiser context: see lammuz_device.h (k_su, k_su_tracked, k_su_fleet and the
// rda_su_solve hook inline the same solve and must round alike)
#pragma clang fp contract(on)

namespace su {

constexpr int NT = 256;          // workgroup size
// start rule su_hard_warm (rda_hip.hip su_body; = oracle/rda_oracle.c): a solve whose FIRST iterate - the previous solution with its multipliers - has a
// relative dual residual above HARD_RD0 started far from its solution; in the hard start the rows of d begin with the barrier HARD_DMU / w
constexpr double HARD_RD0 = 1e-2, HARD_DMU = 1.0;
constexpr int NC = 10;           // inequality rows per stage
constexpr int LAND_STATS = 20;   // counters of Args::land_stat
// -DSU_TRACE (tools/su_trace.py): lane 0 of every wave logs (event id, clock64) at the phase boundaries of the solve - before and behind every barrier - into the
// profiling buffer behind the 16 phase counters, [4 waves][TRACE_CAP][2]; every launch starts over, so the buffer holds the LAST launch of the handle
constexpr int TRACE_CAP = 1024;
#ifdef SU_TRACE
constexpr int PROF_WORDS = 16 + 2 * 4 * TRACE_CAP;
#else
constexpr int PROF_WORDS = 16;
#endif

struct Cfg {
    int T, N, dynamics, accelerated;
    double dt, L, umax0, umax1, ab0, ab1, ws, wu, slack_gain, max_sd, min_sd, ro1, ro2, eps_u;
    // interior-point stop: |r_dual|_inf <= tol_rd (1 + |g|_inf), |r_prim|_inf <= tol_rp, mean complementarity <= tol_mu (1 + |g|_inf)
    double tol_rd = 1e-9, tol_rp = 1e-10, tol_mu = 1e-11;
    int light_check = 1;             // convergence pass without the factorisation when the step predicts convergence (RDA_SU_LIGHT=0: off)
};

// One condensed obstacle term (a, g, cb) of an (obstacle slot, stage): what it adds to the three POSE-INDEPENDENT sums the stage's
// rotation-consistency penalty reduces to (SURVEY A.3 gives Q1 = 2 sum k0.k1, Q2 = sum |k1|^2 with k0 = g + R'a, k1 = dR'a; expanding
// in (cos, sin) of the nominal heading:  Q2 = sum |a|^2,  Q1 = 2 (cos sum g x a - sin sum g.a) - so the N-dependent part never sees
// the pose), and the screening verdict of its hinge (`near`: it may become active while the stage position stays within SCREEN_DELTA of
// the reference position (px, py)).  ONE definition: k_lammuz evaluates it per row right after the dual update (block partial sums
// of GS slots + a GS-bit near mask), k_lmz_finalize re-evaluates it from the stored terms, and the su set-up evaluates it itself when
// it is handed raw terms only (rda_su_solve hook) - all three must round alike.
constexpr double SCREEN_DELTA = 2.0;
constexpr int GS = 8;            // slots per block partial (one LamMuZ workgroup: 2 waves x 4 rows)
constexpr int NBS = 5;           // doubles per block partial: sum |a|^2, sum g.a, sum g x a, dual residual, |Hm|^2
struct RowTerm { double aa, ga, gxa; bool near; };
__device__ __forceinline__ RowTerm row_term(double ax, double ay, double gx, double gy, double cb, double px, double py, double max_sd, bool pose_ok)
{
    RowTerm r;
    r.aa = ax * ax + ay * ay; r.ga = gx * ax + gy * ay; r.gxa = gx * ay - gy * ax;
    const double margin = ax * px + ay * py - cb - max_sd;
    r.near = !(pose_ok && margin > 0 && margin * margin > SCREEN_DELTA * SCREEN_DELTA * r.aa);
    return r;
}

struct Args {
    Cfg c;
    const double *in_s, *in_u;       // linearisation point (3x(T+1), 2xT)
    const double *ref;               // 3x(T+1)
    const double *ref_speed;         // scalar on device
    const double *ax, *ay, *cb, *gx, *gy;   // condensed obstacle terms of obstacle shard 0, each [T][Nloc]: a = A'lam, cb = b'lam + mu'h + z - zeta, g = G'mu + xi
    // reduced form of the terms, written by k_lammuz / k_lmz_finalize: per (stage, GS-slot block) NBS sums and a near mask (bit r = slot
    // GS j + r may become active within SCREEN_DELTA of the pose table's position).  null: the set-up evaluates row_term itself.
    const double *bsum = nullptr; const unsigned long long *bmask = nullptr; int J = 0;
    const double *pose = nullptr;    // [T][4] px, py (column t+1), cos, sin (heading of column t) of the trajectory the masks refer to
    int pose_ok = 0;                 // the pose table is valid (a LamMuZ launch has run since the terms last changed)
    int pose_lin = 0;                // the pose table IS the linearisation point (ADMM iterations >= 1): its cos / sin are reused
    double *pose_out = nullptr;      // (may equal pose) pose table of the trajectory this solve hands back
    int P, Nloc; size_t chunk;       // P obstacle shards (N = P*Nloc); shard r's arrays start `chunk` doubles after shard r-1's
    const double *d_in;              // [T] initial guess for d
    double *out_s, *out_u, *out_d;   // results (may alias in_*)
    int *status;                     // 0 ok / 1 not converged / 2 factorisation failed
    int *ipm_iters;
    double *rd0 = nullptr;           // optional: relative dual residual rd / (1 + |g|) of the first iterate of the first attempt (start-rule key)
    long long *prof;                 // optional per-phase cycle counters (debug), may be null
    double *dbg = nullptr;           // optional per-iteration trace (rdn, rpn, mu, sc) x 100 (debug), may be null
    // Warm start of the su-problems of ADMM iterations >= 1 (the problem differs from the previous iteration's only through the
    // duals, the linearisation point IS the previous solution): slack floor / barrier parameter of the start (0 = the cold
    // rule) and the multipliers of the previous converged solve [NC*T] (read when warm, written by every converged solve)
    double warm_wfl = 0, warm_mu0 = 0; int warm_cap = 30; int warm_shift = 0;
    // end game of the warm attempt: floors of the fraction to the boundary and of the centering parameter.  A warm start begins next
    // to the solution (the dual residual is at rounding level after one step), what is left is driving the complementarity down;
    // with the cold floors (0.995, 1e-3) that takes three iterations from mu ~ 1e-2, with these two.
    double warm_tau = 0.9999, warm_sig = 1e-5;
    int warm_nopred = 0;             // the first iteration of the warm attempt is a plain Newton step towards sigma*mu (no predictor)
    double warm_clip = 0.01;         // relative margin by which the start of a warm attempt is pulled inside the control / distance boxes
    // hard start (> 0): the rows of d get a barrier of their own where NEITHER bound was active - lam+ and lam- are raised by the same
    // delta = hard_dmu - max(kept+, kept-) >= 0 (over their slacks).  (i) With lam = 1e-3 on slacks of 1 the safety distance of a stage that
    // has lost its active hinge terms (re-sorted slots) has next to no curvature (H77 = 2e-3) against the gradient -slack_gain: the first
    // Newton step asks for |dd| ~ 4e3 and is cut to 3e-4 of its length - a lost iteration (C4: in every solve).  (ii) Both slacks of the
    // pair are floored alike (the box is narrower than the floor), so lam+ - lam- and with it the dual residual of the first iterate stay
    // those of the kept multipliers: the key Args::rd0 does not see the start it follows (a one-sided max(kept, dmu / w) did - oracle,
    // iter_num = 1: 1.0 -> 3.0 iterations per solve, the easy start locked out).
    double hard_dmu = 0;
    double *lam_keep = nullptr;
    // time split of the Newton system (TT = 10, 20, 25, 30): the stages [T/2, T) are factorised by wave 0 and the stages [0, T/2) by wave 1 at
    // the same time (see solve); 0 = one recursion over the whole horizon on wave 0 (rounds 1-3)
    int split = 1;
    // safety net (= oracle/rda_oracle.c su_solve_impl, rda_opts::su_accept): the best iterate that is primal feasible to tol_rp, dual feasible to
    // 10 x tol_rd and complementary to 1000 x tol_mu is remembered (controls, distances, multipliers: Lds::acc) and returned when every attempt fails
    int accept = 1;                  // (2: test switch, see the end of solve)
    int first_attempt = 0;           // test switch (rda_opts::su_first_attempt): 1 = start with the last-resort attempt
    // LANDING (round 6, rda_opts::su_land; = oracle/rda_oracle.c su_land): the interior point runs to land_tol only - close enough for the active set to be
    // read off its multipliers and slacks - and the vertex it approaches is then computed exactly: active rows (lam > w) become equalities, the others are
    // dropped, the equality-constrained quadratic model at the iterate is solved with the SAME factorisation and sweeps (weights rho on the active rows, the
    // two passes of an iteration = two steps of the method of multipliers), the result is verified on the true objective (a light pass: stationarity with
    // the hinge terms re-evaluated, feasibility of the dropped rows, signs of the multipliers); rows move in / out by those signs and the round is repeated
    // (at most 4); refused: the interior-point iterate is restored and the iteration goes on (all landings refused: to SU_LAND_FALLBACK x tol_rd / tol_rp / tol_mu).
    // Why: the interior point stops ON the central path, a row that is only just active keeps the slack mu / lam*, and two iterations that stop at different
    // mu differ by up to 1e-4 in the controls (TOL_U); the vertex does not depend on the path (tools/experiments/su_land_oracle.py: 1e-5 -> 1e-13).
    // Stops: the landing is first tried where the interior point reaches land_tol (1e-3 class: the active set is usually readable there - re-sorted north
    // star: 10 instead of 21 interior-point iterations per step, every landing accepted); a refused landing (the primal-dual active-set rounds can cycle
    // while borderline rows are still undecided) is tried again at 1e-2 x land_tol, 1e-4 x ... down to the tight tolerances themselves, then never.
    int land = 0; double land_tol[3] = {1e-3, 1e-4, 1e-5}; double land_rho = 1e4;      // (land_rho: penalty of the active rows relative to the largest stage-Hessian entry)
    // LANDING FIRST (rda_opts::su_land_first; warm attempts only: the su-problems of ADMM iterations >= 1 start from the previous solution of the step and its
    // multipliers, the first one of a tick from the previous tick's solution shifted by one stage).  1: the first pass of the attempt is a LIGHT one (true measures only - no Hessian bases, no Riccati recursion, no sweep
    // matrices): the start usually meets the landing's stop as it stands, and the factorisation of that pass was thrown away by the landing round anyway.
    // 2: ... and when the start does NOT meet the stop the landing is tried all the same - from the start itself, active set = the rows whose KEPT
    // multiplier exceeds the slack - with at most two rounds: the warm-started active-set method.  What it returns has passed the verification on the
    // true objective (stationarity with the hinge terms re-evaluated, feasibility, signs), i.e. it IS the vertex - the answer does not depend on how the
    // active set was guessed; refused: back to the start, the interior point takes over as without the switch (the landing level is not raised).
    int land_first = 0;
    // BLIND landing (the launch sets it after easy solves - zero interior-point iterations, one landing round - of the same step): the warm attempt starts WITH a
    // landing round, no measuring pass in front of it (penalty scale: land_rho_prev, what the last landing used); one round only, then as a refused speculation
    int land_blind = 0; double land_rho_prev = 0;
    int land_level0 = 0;             // first landing level of the attempts (0: the interior point stops at land_tol; 1: at 1e-2 x land_tol): the launch raises it after a solve whose landing took three or more rounds
    double land_first_rd0 = 0.5;     // ... only from a start whose relative dual residual lies below this (60-step loops, accepted / tried below 0.5 | 0.5 .. 1 | above 1: north star 47 / 53 | 7 / 23 | 3 / 14;
                                     // N = 2000 77 / 79 | 2 / 62 | 1 / 49; C5 shape 47 / 54 | 7 / 17 | 0 / 16; C4 0 / 3 | 2 / 43 | 0 / 86)
    int *land_stat = nullptr;        // optional [LAND_STATS]: landings accepted, refused, rounds, interior-point iterations that were landing / verification passes;
                                     // [4] speculative landings (land_first = 2) tried, [5] accepted, [6 + k] / [12 + k] tried / accepted by the decade k of the start's relative dual residual (< 1e-4, .. < 1, >= 1)
    // the reference may still be in the making when the solve starts (another workgroup samples it, k_su_tracked): it is then
    // fetched at its first use (the stage gradients of the first interior-point pass), once *ref_flag == ref_seq (agent scope)
    const unsigned long long *ref_flag = nullptr; unsigned long long ref_seq = 0;
#ifdef SU_TRACE
    long long t_entry = 0;           // clock64() at kernel entry (event 99 of the trace: what the launch spends before the solve starts)
    long long t_mark[4] = {0, 0, 0, 0};   // ... and at four points of the launch's prologue (events 95 .. 98)
#endif
};

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// linearised motion models, rda_solver.py:949-994.  (cp, sp) = cos / sin of the state heading st[2] (the caller has them: the pose
// table for ADMM iterations >= 1, one sincos otherwise); the omni model linearises about the VELOCITY heading ut[1] instead.
__device__ inline void lin_model(const Cfg &c, const double *st, const double *ut, double cp, double sp, double *A, double *B, double *C)
{
    double dt = c.dt;
    for (int i = 0; i < 9; ++i) A[i] = 0;
    for (int i = 0; i < 6; ++i) B[i] = 0;
    C[0] = C[1] = C[2] = 0;
    A[0] = A[4] = A[8] = 1;
    if (c.dynamics == 2) {
        double phi = ut[1], v = ut[0];
        sincos(phi, &sp, &cp);
        B[0] = cp * dt; B[1] = -v * sp * dt; B[2] = sp * dt; B[3] = v * cp * dt;
        C[0] = phi * v * sp * dt; C[1] = -phi * v * cp * dt;
        return;
    }
    double phi = st[2], v = ut[0];
    A[2] = -v * dt * sp; A[5] = v * dt * cp;
    B[0] = cp * dt; B[2] = sp * dt;
    C[0] = phi * v * sp * dt; C[1] = -phi * v * cp * dt;
    if (c.dynamics == 0) {
        double psi = ut[1], cs = cos(psi);
        B[4] = tan(psi) * dt / c.L; B[5] = v * dt / (c.L * cs * cs);
        C[2] = -psi * v * dt / (c.L * cs * cs);
    } else {
        B[5] = dt;
    }
}

// LDS carve-up (doubles).  Per-lane rows of the sweep matrices are 16-byte aligned (row stride 6).
// near terms of a solve kept in LDS, one or two per thread in the hinge sums (a solve with more visits its masks in global memory instead);
// the long horizons have no room for the second half (160 KB of LDS per workgroup)
__device__ __host__ constexpr int near_max(int T) { return T <= 40 ? 512 : 256; }
// ... and what the LIST itself may hold (round 5).  A solve with more near terms than near_max keeps them in LDS all the same and sums them as (stage,
// chunk) partials - what it did over the masks in global memory before (N = 2000: 13.6 k cycles per hinge pass, 25 % of the solve).  Sized to the LDS
// the other arrays leave (160 KB per workgroup; 32 B per term)
__device__ __host__ constexpr int near_cap(int T) { return T <= 20 ? 1792 : (T <= 25 ? 1536 : (T <= 30 ? 1280 : (T <= 35 ? 1024 : (T <= 40 ? 896 : (T <= 50 ? 1024 : 256))))); }
// Stage strides of the big per-stage arrays (doubles).  The natural sizes 48 / 64 are multiples of 32 dwords: the rows of stage t and t + 2 (or t + 1)
// then start on the same LDS banks and every (stage, row)-thread phase runs 2-way conflicted.  Round 5, k_su<20> in the headline loop,
// SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS per dispatch (tools/experiments/lds_conflicts.sh): 48/64/24/36 -> 1.04, HB = 66 -> 0.94, + FT = 50 -> 0.84,
// + WN = 26, MF = 38 -> 0.86; launch time 108.6 -> 107.9 us.  The LDS array is busy ~12 % of a launch (active + conflict cycles over 4 x wave cycles):
// bank conflicts are not what bounds this kernel, the padding that is free is kept.
#ifndef SU_FT
#define SU_FT 50
#define SU_HB 66
#define SU_WN 24
#define SU_MF 36
#endif
constexpr int FT = SU_FT;   // F' of the stage, [8 columns q][6]: F[0..4][q] | pad   (x+ = F y, x = [s(3) up(2)])   (48 used)
constexpr int HB = SU_HB;   // full 8x8 stage Hessian base; re-used after the matrix sweep for Mb [8][6]               (64 used)
constexpr int WN = SU_WN;   // W (5x3) | Minv sym (6) | pad                                                          (21 used)
constexpr int MF = SU_MF;   // forward sweep rows [6][6]                                                             (36 used)
static_assert(FT >= 48 && HB >= 64 && WN >= 21 && MF >= 36 && FT % 2 == 0 && HB % 2 == 0 && MF % 2 == 0, "stage strides: rows are read as 16-byte pairs");
__device__ __host__ constexpr int ev(int n) { return (n + 1) & ~1; }
// time split of the Newton system: the horizons with a compile-time instantiation are cut in two halves (0 = no split)
__device__ __host__ constexpr int split_point(int T) { return (T == 10 || T == 20 || T == 25 || T == 30) ? T / 2 : 0; }
struct Lds {
    double *s, *u, *d, *phin, *ref, *Ak, *Bk, *Ck, *csn, *Q1, *Q2;   // csn: cos [T] | sin [T] of the nominal headings
    double *Ft;        // [T][FT]  (constant during the solve)
    double *part;      // [NT][9] partial sums of the chunked reductions (aliases Hb)
    double *hs;        // [T][9]  hinge sums
    double *Hw, *gw;   // [T][9] (7 used; the region keeps its 16 T: Mf overlays it), [T][4]
    double *bw;        // [T][5]  barrier weights lam/w of the inequality pairs, + and - row summed (u0 box, u1 box, d box, rate u0, rate u1)
    double *cy;        // [T][5]  lam+ - lam- of the pairs (C'lam of the stage gradient)
    double *gst;       // [T][8]  stage gradient (objective + C'lam)
    double *gad;       // [T][3]  reduced gradient (adjoint sweep)
    double *Hb;        // [T][HB]  stage Hessian base -> (after the matrix sweep) backward rows Mb [8][6]
    double *Wn;        // [T][WN]
    double *Mf;        // [T][MF]  (overlays hs..cy, which are dead once the stage Hessians are assembled)
    double *kk;        // [T][8]   backward sweep outputs per stage: p (5) | feed-forward kk (3)
    double *vv;        // [T][8]   forward sweep outputs per stage: dx+ (5) | v_2 ; v = entries 3..5
    double *xd, *lw, *ra;                  // [T][5] per inequality PAIR (see the pair threads in solve): x+ - x-, lam w (+ and -), max |r_p|
    double *m7;                            // [T][8] column of d_t in the stage Hessian (entries 0..6) | 1 / its diagonal: d_t is eliminated before the recursion
    double *dy;                            // [T][8]
    double *pv, *red;                      // 8, NT + 32 (reductions, flags, the two waves' 2 x 64-double scratch of the matrix recursion)
    double *p0;                            // [2][T] reference positions of the hinge screening
    double *uk, *ub, *xs;                  // time split (split_point(T) > 0): unit backward sweeps [5][8 m], their F_v' p [5][m][2], interface block [96]
    double *acc;                           // [13 T] safety net: u (2T) | d (T) | multipliers of the pairs [T][10]
    double *hmx, *sav, *base;              // [T] largest diagonal entry of the stage's derivative block (scale of the landing's penalty); [3 T] u | d of the iterate a landing started from; [3 T] ... of the point its current round linearises at
    double *near; int *ncnt, *sto;   // near list of the hinge screening: [near_cap][4] = (ax, ay, cb, stage) of the terms that may be active, stage-major and compact; [NT] per-thread counts; [T+1] first entry of a stage
    __device__ void carve(double *b, int T) {
        double *p = b;
        s = p; p += ev(3 * (T + 1)); u = p; p += 2 * T; d = p; p += ev(T); phin = p; p += ev(T); ref = p; p += ev(3 * (T + 1));
        Ak = p; p += ev(9 * T); Bk = p; p += 6 * T; Ck = p; p += ev(3 * T); csn = p; p += 2 * T; Q1 = p; p += ev(T); Q2 = p; p += ev(T);
        Ft = p; p += FT * T; hs = p; Mf = p; p += ev(9 * T);      // Mf (after the matrix sweep) overlays hs|Hw|gw|bw|cy: 39T >= MF*T
        Hw = p; p += 16 * T; gw = p; p += 4 * T; bw = p; p += ev(5 * T); cy = p; p += ev(5 * T);
        gst = p; p += 8 * T; gad = p; p += ev(3 * T);
        Hb = p; part = p; p += (HB * T > 9 * NT ? HB * T : 9 * NT);    // part (phase 1) is dead before Hb is written (phase 3)
        Wn = p; p += WN * T; kk = p; p += 8 * T; vv = p; p += 8 * T;
        xd = p; p += ev(5 * T); lw = p; p += ev(5 * T); ra = p; p += ev(5 * T); m7 = p; p += 8 * T;
        dy = p; p += 8 * T; pv = p; p += 8; red = p; p += NT + 32; p0 = p; p += 2 * T;
        const int m = split_point(T);
        uk = p; p += 40 * m; ub = p; p += 10 * m; xs = p; p += m ? 96 : 0;
        acc = p; p += ev(13 * T);
        hmx = p; p += ev(T); sav = p; p += ev(3 * T); base = p; p += ev(3 * T);
        near = p; p += 4 * near_cap(T); ncnt = (int *)p; p += NT / 2; sto = (int *)p; p += ev(T + 2) / 2 + 1;
    }
};
constexpr size_t lds_bytes(int T)
{
    size_t n = (size_t)2 * ev(3 * (T + 1)) + 2 * T + 2 * ev(T) + ev(9 * T) + 6 * T + ev(3 * T) + 2 * T + 2 * ev(T)
             + FT * T + ev(9 * T) + 16 * T + 4 * T + 2 * ev(5 * T) + 8 * T + ev(3 * T) + (HB * T > 9 * NT ? HB * T : 9 * NT)
             + WN * T + 16 * T + 3 * ev(5 * T) + 8 * T + 8 * T + 8 + NT + 32 + 2 * T + 50 * split_point(T) + (split_point(T) ? 96 : 0)
             + ev(13 * T) + ev(T) + 2 * ev(3 * T) + 4 * near_cap(T) + NT / 2 + ev(T + 2) / 2 + 1;
    return n * sizeof(double);
}
// every horizon the interface accepts (RDA_TMAX = 64) must fit the 160 KB a workgroup can have (round 5: T = 36 .. 40 did not for a while - the near
// list was sized per range of T, the safety-net block and the padded strides came on top, and no test created such a handle)
constexpr bool lds_fits_all() { for (int T = 1; T <= 64; ++T) if (lds_bytes(T) > 160 * 1024) return false; return true; }
static_assert(lds_fits_all(), "su::Lds: some horizon T <= 64 needs more than 160 KB of LDS - shrink near_cap(T) there");

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

// wave-uniform broadcast of lane `src`'s value (src is a compile-time constant after unrolling): two v_readlane_b32,
// the result lives in an SGPR pair and feeds the FMAs as a scalar operand
__device__ __forceinline__ double bcast(double v, int src)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// (lanes without a source - shifted in from outside the row, or rows outside ROWS - read 0)
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 64 lanes: DPP inside the 16-lane rows (xor 1, xor 2, half mirror, mirror), v_readlane across rows
__device__ __forceinline__ double wave_allreduce(double v, bool is_max)
{
    auto op = [&](double x, double y) { return is_max ? (y > x ? y : x) : x + y; };
    v = op(v, dpp_f64<0xB1>(v));
    v = op(v, dpp_f64<0x4E>(v));
    v = op(v, dpp_f64<0x141>(v));
    v = op(v, dpp_f64<0x140>(v));
    return op(op(bcast(v, 0), bcast(v, 16)), op(bcast(v, 32), bcast(v, 48)));
}
__device__ __forceinline__ double block_reduce(double v, double *red, int tid, bool is_max)
{
    v = wave_allreduce(v, is_max);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < NT / 64; ++w) r = is_max ? (red[w] > r ? red[w] : r) : r + red[w];
    return r;
}

// ... with ONE barrier: every call site owns its NT / 64 slots, so the barrier that keeps a later writer from a slow reader of the previous reduction
// is any barrier between two executions of the same site (round 6: the step-length and the reach / complementarity reductions run twice per iteration)
__device__ __forceinline__ double block_reduce1(double v, double *slots, int tid, bool is_max)
{
    v = wave_allreduce(v, is_max);
    if ((tid & 63) == 0) slots[tid >> 6] = v;
    __syncthreads();
    double r = slots[0];
    for (int w = 1; w < NT / 64; ++w) r = is_max ? (slots[w] > r ? slots[w] : r) : r + slots[w];
    return r;
}

// C' x for one stage: x = per-row values of the 10 inequality rows -> entries 3..7 of y
__device__ __forceinline__ void con_T(const double *x, int t, double &y3, double &y4, double &y5, double &y6, double &y7)
{
    double r0 = t >= 1 ? x[6] - x[7] : 0.0, r1 = t >= 1 ? x[8] - x[9] : 0.0;
    y5 = x[0] - x[1] + r0; y6 = x[2] - x[3] + r1; y7 = x[4] - x[5]; y3 = -r0; y4 = -r1;
}

// entry F[i][q] of the stage transition x+ = F y from the packed transpose
__device__ __forceinline__ double Fel(const double *Ft_t, int i, int q) { return Ft_t[6 * q + i]; }

// Inequality pair (stage t, linear form k) -> thread.  Horizons with a compile-time instantiation up to 32 stages: the form k is WAVE-UNIFORM per
// half wave - wave 0 holds k = 0 (lanes 0..31) and k = 1 (lanes 32..63), wave 1 the rate pairs k = 3 / 4, wave 2 (lanes 0..31) the distance pair
// k = 2, lane & 31 = the stage - so the three shapes of a pair's linear form (a control, the distance with its eliminated-variable dot product,
// a control difference) are branches whole waves skip instead of three divergent passes in every wave (round 6; rounds 3-5 dealt pair
// p = 5 t + k to thread p: 5 T <= 150 threads in waves 0 / 1 / 2, every wave holding every k).  Other horizons: pair p = 5 t + k on thread p (two per thread beyond NT).
template <int TT> __device__ __forceinline__ bool pair_map(const int T, const int j, int &t, int &k)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (TT > 0 && TT <= 32) {
        const int half = lane >> 5, tl = lane & 31;
        k = wave == 0 ? half : (wave == 1 ? 3 + half : 2);
        const bool ok = j == 0 && tl < T && (wave < 2 || (wave == 2 && half == 0));
        t = ok ? tl : 0; if (!ok) k = 0;
        return ok;
    }
    const int pi = tid + j * NT;
    const bool ok = pi < 5 * T;
    t = ok ? pi / 5 : 0; k = ok ? pi % 5 : 0;
    return ok;
}

// EVERY global input of the set-up, requested in one go into registers (round 6).  The set-up's phases (nominal, linearisation, start of the duals, term
// sums) each began with a dependent trip to memory - what the LamMuZ launch wrote sits in HBM / other XCDs' L2s: ~1.5 us a trip - and so did the launch's
// own prologue before the solve (control block, residual partials: rda_hip.hip su_body).  The caller issues prefetch() FIRST, next to its own loads, so that
// all of them overlap into one trip; a caller that passes no Pre (the rda_su_solve hook) has the solve issue it at its entry.
// the verdict of a solve once more, in registers of every thread (uniform): the launch's bookkeeping behind the solve then needs no trip to memory
#ifndef SU_LAND_FALLBACK
#define SU_LAND_FALLBACK 1e-3
#endif
struct Result { int status = 1, iters = 0; double rd0 = 0; int spec = 0; int land_rounds = 0; int rounds_all = 0; int blind = 0; double land_rho = 0; };      // (rd0: Args::rd0; left alone by a solve that does not measure it.  spec: a speculative landing (Args::land_first = 2) was 1 accepted, 2 refused; land_rounds: rounds of the non-speculative landings of the solve, rounds_all: of all of them; blind: a blind landing was 1 accepted, 2 refused; land_rho: penalty of the last landing)
struct Pre {
    double vref = 0;
    double u0 = 0, u1 = 0, d = 0, cp = 0, sp = 0;      // thread t < T: nominal controls / distance, cos / sin of the pose table (pose_lin)
    double lkp[2] = {0, 0}, lkm[2] = {0, 0};           // kept multipliers of this thread's inequality pairs (+ row, - row)
    double q0[8], q1[8], q2[8]; unsigned long long mk[8];   // block partials / near masks of the thread's (stage, chunk) slice, first eight blocks
    double sv = 0, rv = 0;                             // element tid of the nominal states / the reference (3 (T+1) <= 195 < NT)
    double p0a = 0, p0b = 0;                           // element tid < 2 T of the screening reference: from the pose table / from the nominal states
};
template <int TT> __device__ __forceinline__ void prefetch(const Args &a, Pre &p)
{
    const Cfg &c = a.c;
    const int T = TT > 0 ? TT : c.T, tid = threadIdx.x;
    p.vref = *a.ref_speed;
    p.d = c.max_sd;
    if (tid < 128 && (tid & 63) < T) {          // (lane t of wave 0 - the linearisation - and of wave 1 - the clip of the start, beside it)
        const int t = tid & 63;
        p.u0 = a.in_u[t]; p.u1 = a.in_u[T + t]; if (a.d_in) p.d = a.d_in[t];
        if (a.pose_lin) { p.cp = a.pose[4 * t + 2]; p.sp = a.pose[4 * t + 3]; }
    }
    if (a.lam_keep)
        for (int j = 0; j < 2; ++j) {
            int t, kk;
            if (pair_map<TT>(T, j, t, kk)) {
                const int ts = a.warm_shift ? (t + 1 < T ? t + 1 : T - 1) : t;
                p.lkp[j] = a.lam_keep[ts * NC + 2 * kk]; p.lkm[j] = a.lam_keep[ts * NC + 2 * kk + 1];
            }
        }
    const int nch = NT / T, rt = tid / nch, rc_ = tid % nch;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p.q0[k] = 0; p.q1[k] = 0; p.q2[k] = 0; p.mk[k] = 0; }
    if (a.bsum != nullptr && rt < T) {
        const int J = a.J, KB = (J + nch - 1) / nch;
        const double *bs = a.bsum + (size_t)rt * J * NBS;
        const unsigned long long *bm = a.bmask + (size_t)rt * J;
        // (blocks beyond the slice are read from block 0 and never used: the consumer tests the same bounds.  Zeroing them here made every load of
        // the batch CONDITIONAL code with a wait of its own - eight dependent trips instead of one, found in the ISA, round 6)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = rc_ + k * nch; const bool in = k < KB && j < J; const int jj = in ? j : 0;
            p.q0[k] = bs[(size_t)jj * NBS]; p.q1[k] = bs[(size_t)jj * NBS + 1]; p.q2[k] = bs[(size_t)jj * NBS + 2]; p.mk[k] = bm[jj];
        }
    }
    if (tid < 3 * (T + 1)) { p.sv = a.in_s[tid]; if (!a.ref_flag) p.rv = a.ref[tid]; }
    if (tid < 2 * T) {
        if (a.bsum != nullptr && a.pose != nullptr) p.p0a = a.pose[4 * (tid % T) + tid / T];
        p.p0b = a.in_s[(tid / T) * (T + 1) + (tid % T) + 1];
    }
}

// The whole solve.  Must be called by all NT threads of the block with `smem` >= lds_bytes(T).  TT > 0 fixes the horizon
// at compile time (every LDS offset becomes an immediate, the stage loops get constant bounds); TT == 0 reads it from c.T.
// RefWait: called by all threads before the first use of the reference, when a.ref_flag is set - returns when the reference (a.ref) is complete
// (k_su_tracked: another workgroup samples it meanwhile; the functor bounds the wait and samples it itself on expiry).
struct NoRefWait { __device__ __forceinline__ void operator()(double *) const {} };
// (pre / have_pre: see Pre.  The struct is handed over by reference and never through a selected pointer, so that its members stay registers)
template <int TT, typename RefWait = NoRefWait> __device__ __forceinline__ bool solve(const Args &a, double *smem, Pre &pre, const bool have_pre, Result &res, RefWait ref_wait = RefWait())
{
    static_assert(3 * (64 + 1) <= NT && 2 * 64 <= NT, "Pre: one element of the nominal states / screening reference per thread");
    if (!have_pre) prefetch<TT>(a, pre);
    const Cfg &c = a.c;
    const int T = TT > 0 ? TT : c.T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Lds L; L.carve(smem, T);
#ifdef SU_POISON_LDS      // debug build (tools/experiments/README.md): every double of the workgroup's LDS starts as a NaN - a read of a word this solve has not written shows
    for (int i = tid; i < (int)(lds_bytes(T) / sizeof(double)); i += NT) smem[i] = __builtin_nan("");
    __syncthreads();
#endif
    long long tprev = clock64();
    long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // phase cycle counters stay in registers (k is a literal)
    // The phase counters exist in the PROFILING builds only (-DSU_PROF, -DSU_FINE: tools/su_phase_profile.py).  As a run-time switch of the product build
    // they cost 32 VGPRs (the 16 counters) and a branch per mark: k_su<20> 108.6 -> 107.1 us in the headline loop without them (round 5, same box).
#if defined(SU_PROF) || defined(SU_FINE)
    const bool prof_on = a.prof != nullptr;
#else
    constexpr bool prof_on = false;
#endif

    auto mark = [&](int k) { if (prof_on) { long long now = clock64(); pacc[k] += now - tprev; tprev = now; } };
#ifdef SU_TRACE
    int trn = 0;
    auto tr_ = [&](int id) {
        if (a.prof && (threadIdx.x & 63) == 0 && trn < TRACE_CAP) {
            long long *q = a.prof + 16 + ((threadIdx.x >> 6) * TRACE_CAP + trn) * 2;
            q[0] = id; q[1] = clock64(); ++trn;
        }
    };
#define TR(id) tr_(id)
#else
#define TR(id)
#endif
#ifdef SU_FINE      // one-off build for tools/su_phase_profile.py --fine: the set-up slots are re-used for sub-phases of the iteration
#define MS(k) mark(10)
#define MF(k) mark(k)
#else
#define MS(k) mark(k)
#define MF(k)
#endif
#ifdef SU_TRACE
    if (a.prof && (threadIdx.x & 63) == 0) { long long *q = a.prof + 16 + ((threadIdx.x >> 6) * TRACE_CAP + trn) * 2; q[0] = 99; q[1] = a.t_entry; ++trn; for (int k = 0; k < 4; ++k) { q += 2; q[0] = 95 + k; q[1] = a.t_mark[k]; ++trn; } }
#endif
    // the landing counters: fire-and-forget atomic adds of thread 0.  As `a.land_stat[k] += 1` each was a global load - wait - store on the solve's critical path
    // (the whole workgroup meets the waiting wave at the next barrier): ~1.5 k cycles per counter, three to five per solve
#define LSTAT(k, v) do { if (a.land_stat && tid == 0) __hip_atomic_fetch_add(&a.land_stat[k], (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
    TR(100);
    const double vref = pre.vref;
    // (stage, chunk) mapping of the obstacle reductions
    const int nch = NT / T;                       // chunks per stage (T <= 64 -> nch >= 4)
    const int rt = tid / nch, rc_ = tid % nch;    // this thread's stage and chunk
    const bool ract = rt < T;
    const bool masks_in = a.bsum != nullptr;      // the reduced form of the terms is handed in (k_lammuz / k_lmz_finalize)
    const bool warm = a.warm_mu0 > 0 && a.lam_keep != nullptr;
    const int J = masks_in ? a.J : (a.Nloc + GS - 1) / GS;     // GS-slot blocks per shard
    const int KB = (J + nch - 1) / nch;                         // blocks per shard in one thread's slice: j = rc_ + kb nch

    // ---- every global input of the set-up was requested by prefetch() (see Pre): the registers it filled
    const double pf_u0 = pre.u0, pf_u1 = pre.u1, pf_d = pre.d, pf_cp = pre.cp, pf_sp = pre.sp;
    auto pair_map = [&](const int j, int &t, int &k) -> bool { return su::pair_map<TT>(T, j, t, k); };
    const double pf_lkp[2] = {warm ? pre.lkp[0] : 0.0, warm ? pre.lkp[1] : 0.0}, pf_lkm[2] = {warm ? pre.lkm[0] : 0.0, warm ? pre.lkm[1] : 0.0};   // kept multipliers of this thread's inequality pairs
    // block partials / near masks of the thread's slice: eight blocks at a time, all their loads in flight together
    auto load_batch = [&](int r, int kb0, double *q0, double *q1, double *q2, unsigned long long *mk) {
        const double *bs = a.bsum + r * a.chunk + (size_t)rt * J * NBS;
        const unsigned long long *bm = a.bmask + r * a.chunk + (size_t)rt * J;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = rc_ + (kb0 + k) * nch; const bool in = kb0 + k < KB && j < J; const int jj = in ? j : 0;
            q0[k] = bs[(size_t)jj * NBS]; q1[k] = bs[(size_t)jj * NBS + 1]; q2[k] = bs[(size_t)jj * NBS + 2]; mk[k] = bm[jj];      // (no zeroing: see prefetch)
        }
    };

    // ---- load nominal, reference; linearise -------------------------------------------------
    if (tid < 3 * (T + 1)) { L.s[tid] = pre.sv; if (!a.ref_flag) L.ref[tid] = pre.rv; }
    for (int i = tid; i < WN * T; i += NT) L.Wn[i] = 0.0;       // (W[.][2], Minv[.][2] stay zero: d_t is not part of the recursion)
    for (int i = tid; i < FT * T; i += NT) L.Ft[i] = 0.0;       // (the stage threads fill in the non-zeros after the barrier)
    if (tid < T) { L.u[tid] = pf_u0; L.u[T + tid] = pf_u1; }
    // reference positions of the hinge screening: where the masks were made (pose table), else the nominal positions of stages 1..T
    if (tid < 2 * T) L.p0[tid] = (masks_in && a.pose_ok) ? pre.p0a : pre.p0b;
    __syncthreads();
    MS(0); TR(120);
    if (tid < T) {
        int t = tid;
        double st[3] = { L.s[t], L.s[(T + 1) + t], L.s[2 * (T + 1) + t] }, ut[2] = { pf_u0, pf_u1 };      // (the nominal controls from the registers: wave 1 is clipping L.u meanwhile)
        double cp, sp;
        if (a.pose_lin) { cp = pf_cp; sp = pf_sp; } else sincos(st[2], &sp, &cp);
        L.csn[t] = cp; L.csn[T + t] = sp;
        // (the model in registers, then to LDS: Ak / Bk / Ck for the roll-outs, and the non-zeros of F = [[A 0 B 0],[0 0 I2 0]] (5 x 8, stored
        // transposed and padded: Ft[q][i]) straight from the registers - the rest of Ft was zeroed by all threads above.  Round 5: the T stage
        // threads used to write all 48 entries and read A, B back from LDS: 4.8 k cycles per solve)
        double Am[9], Bm[6], Cm[3];
        lin_model(c, st, ut, cp, sp, Am, Bm, Cm);
#pragma unroll
        for (int i = 0; i < 9; ++i) L.Ak[9 * t + i] = Am[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) L.Bk[6 * t + i] = Bm[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) L.Ck[3 * t + i] = Cm[i];
        L.phin[t] = st[2];
        double *F = &L.Ft[FT * t];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int q = 0; q < 3; ++q) F[6 * q + r] = Am[3 * r + q];
#pragma unroll
            for (int q = 0; q < 2; ++q) F[6 * (5 + q) + r] = Bm[2 * r + q];
        }
        F[6 * 5 + 3] = 1.0; F[6 * 6 + 4] = 1.0;
    }
    // ... and BESIDE the linearisation (wave 0) wave 1 pulls the start inside the boxes (round 6: it was a phase of its own, the T threads of wave 0 and a barrier).
    // A warm attempt starts next to the previous solution: pulled inside by warm_clip only (a cold start by 1 %), so that the kept multipliers of the active rows
    // meet slacks of that size and the start is already nearly complementary
    if (wave == 1 && lane < T) {
        const int t = lane;
        const double clipm = warm ? a.warm_clip : 0.01;
        const double lim0 = (1.0 - clipm) * c.umax0, lim1 = (1.0 - clipm) * c.umax1;
        const double v0 = pf_u0, v1 = pf_u1;
        L.u[t] = v0 > lim0 ? lim0 : (v0 < -lim0 ? -lim0 : v0);
        L.u[T + t] = v1 > lim1 ? lim1 : (v1 < -lim1 ? -lim1 : v1);
        const double lo = c.min_sd + clipm * (c.max_sd - c.min_sd), hi = c.max_sd - clipm * (c.max_sd - c.min_sd);
        const double dv = pf_d;
        L.d[t] = dv > hi ? hi : (dv < lo ? lo : dv);
    }
    __syncthreads();
    MS(11); TR(131);
    // ---- initial point (same rule as the oracle) ------------------------------------------------
    auto clip_controls = [&](const double clipm) {          // (the nominal controls / distances: the registers prefetched above)
        if (tid < T) {
            int t = tid;
            double lim0 = (1.0 - clipm) * c.umax0, lim1 = (1.0 - clipm) * c.umax1;
            double v0 = pf_u0, v1 = pf_u1;
            L.u[t] = v0 > lim0 ? lim0 : (v0 < -lim0 ? -lim0 : v0);
            L.u[T + t] = v1 > lim1 ? lim1 : (v1 < -lim1 ? -lim1 : v1);
            double lo = c.min_sd + clipm * (c.max_sd - c.min_sd), hi = c.max_sd - clipm * (c.max_sd - c.min_sd);
            double dv = pf_d;
            L.d[t] = dv > hi ? hi : (dv < lo ? lo : dv);
        }
        __syncthreads();
    };
    // (the set-up's own clip: done by wave 1 beside the linearisation, above)
    // state rollout with the current controls.  In all three motion models A = [[1,0,a13],[0,1,a23],[0,0,1]]
    // (rda_solver.py:955,971,987), so the heading is a running sum of per-stage increments and, once it is known, so
    // are x and y: the increments are formed by one lane per stage, the two running sums are 3T dependent additions
    // on values that were fetched in one go (wave 0 only).
    auto rollout = [&](const int wv = 0) {       // (one whole wave; the set-up's runs on wave 3, beside the start of the duals on the pair threads of waves 0 - 2)
        if (wave == wv) {
            double *inc = L.dy;                         // scratch: dy is dead outside the sweeps
            for (int t = lane; t < T; t += 64) {
                const double *B = &L.Bk[6 * t], *C = &L.Ck[3 * t];
                const double u0 = L.u[t], u1 = L.u[T + t];
                inc[3 * t + 2] = (B[4] * u0 + B[5] * u1) + C[2];
                inc[3 * t] = (B[0] * u0 + B[1] * u1) + C[0];
                inc[3 * t + 1] = (B[2] * u0 + B[3] * u1) + C[1];
            }
            wsync();
            // the running sums as wave prefix scans, one stage per lane (T <= 64): six shuffle steps instead of T dependent additions on one lane
            // holding T registers (round 4: the unrolled register arrays of the serial form were the reason the T = 25 / 30 instantiations
            // spilled to scratch).  Another summation order than the serial loop: rounding level, like every reduction of this kernel.
            // (round 5: as DPP moves - row_shr 1, 2, 4, 8 inside the 16-lane rows, then the row totals across with row_bcast:15 / :31 - instead of six
            // __shfl_up = ds_bpermute round trips per scan: two dependent scan rounds per roll-out, two roll-outs per solve)
            auto prefix = [&](double v) {
                v += dpp_f64<0x111>(v); v += dpp_f64<0x112>(v); v += dpp_f64<0x114>(v); v += dpp_f64<0x118>(v);
                v += dpp_f64<0x142, 0xA>(v); v += dpp_f64<0x143, 0xC>(v);
                return v;
            };
            {
                const double ph0 = L.s[2 * (T + 1)];
                const double v = prefix(lane < T ? inc[3 * lane + 2] : 0.0);
                if (lane < T) L.s[2 * (T + 1) + lane + 1] = ph0 + v;
            }
            wsync();
            {
                const double x0 = L.s[0], y0 = L.s[T + 1];
                double ix = 0, iy = 0;
                if (lane < T) {                                  // x, y increments need the heading of their own stage
                    const double ph = L.s[2 * (T + 1) + lane];
                    ix = inc[3 * lane] + L.Ak[9 * lane + 2] * ph; iy = inc[3 * lane + 1] + L.Ak[9 * lane + 5] * ph;
                }
                const double px = prefix(ix), py = prefix(iy);
                if (lane < T) { L.s[lane + 1] = x0 + px; L.s[(T + 1) + lane + 1] = y0 + py; }
            }
        }
    };
    rollout(3);
    MS(13); TR(133);
    // ---- the inequality rows live in REGISTERS.  The 10 rows of a stage are 5 pairs (+val <= e+, -val <= e-) of one linear form each
    // (k = 0 u0, 1 u1, 2 d, 3 u0 - up0, 4 u1 - up1; the rate pairs exist for t >= 1): pair p = 5 t + k is owned by thread p (5 T <= 256
    // for T <= 51; the generic instantiation takes two per thread), which keeps the pair's slacks and multipliers (w+, w-, lam+, lam-),
    // residuals, targets and steps for the whole solve.  What the other phases need of the rows travels through three small [T][5]
    // arrays per pass (barrier weights and lam+ - lam- for the stage Hessians / gradients, x+ - x- for the Newton right-hand side) and
    // two for the termination measures - instead of six [T][10] arrays read and written by every phase (round 3: the row phases were
    // ~40 % of an interior-point iteration, all LDS trips and divisions; reciprocals of w and lam are now formed once per iteration).
    constexpr int NPR = TT > 0 ? (5 * TT + NT - 1) / NT : 2;
    int  p_t[NPR], p_k[NPR]; bool p_ok[NPR], p_on[NPR];
    double p_ep[NPR], p_em[NPR];                              // right-hand sides of the + and - row
    double Pwp[NPR], Pwm[NPR], Plp[NPR], Plm[NPR];            // slacks, multipliers
    double Prpp[NPR], Prpm[NPR], Prcp[NPR], Prcm[NPR];        // primal residuals, complementarity targets of the current right-hand side
    double Piwp[NPR], Piwm[NPR], Pilp[NPR], Pilm[NPR];        // reciprocals (once per iteration)
    double Pdwp[NPR], Pdwm[NPR], Pdlp[NPR], Pdlm[NPR];        // steps
#pragma unroll
    for (int j = 0; j < NPR; ++j) {
        p_ok[j] = pair_map(j, p_t[j], p_k[j]);
        p_on[j] = p_ok[j] && (p_k[j] < 3 || p_t[j] >= 1);
        p_ep[j] = con_rhs(c, 2 * p_k[j]); p_em[j] = con_rhs(c, 2 * p_k[j] + 1);
        Pwp[j] = Pwm[j] = 1.0; Plp[j] = Plm[j] = 0.0; Prpp[j] = Prpm[j] = Prcp[j] = Prcm[j] = 0.0;
        Piwp[j] = Piwm[j] = 1.0; Pilp[j] = Pilm[j] = 0.0; Pdwp[j] = Pdwm[j] = Pdlp[j] = Pdlm[j] = 0.0;
    }
    auto frcp = [](double x) {                                 // 1 / x: v_rcp_f64 + two Newton steps (full precision for normal x > 0)
        double r = __builtin_amdgcn_rcp(x);
        r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
        return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    };
    // value of pair (t, k)'s linear form at the current controls / distances (con_val of its + row)
    auto pair_val = [&](int t, int k) {
        if (k == 2) return L.d[t];
        const double *uu = (k == 1 || k == 4) ? &L.u[T] : &L.u[0];       // k = 0, 3: u0 ; k = 1, 4: u1
        const double v = uu[t];
        return k < 3 ? v : v - (t ? uu[t - 1] : 0.0);
    };
    // ... and of a step: [dx (5) | .] in dy, v = (du0, du1, dd) in vv[3..5]
    auto pair_step = [&](int t, int k) {
        const double *y = &L.dy[8 * t], *v = &L.vv[8 * t + 3];
        if (k == 2) {                                         // dd = -(g7 + m7' [dx; du]) / H77 ; also left where the update reads it
            const double *m7 = &L.m7[8 * t];
            const double g7 = L.gst[8 * t + 7] + L.xd[5 * t + 2];
            const double dd = -(g7 + (m7[0] * y[0] + m7[1] * y[1] + m7[2] * y[2] + m7[3] * y[3] + m7[4] * y[4]) + (m7[5] * v[0] + m7[6] * v[1])) * m7[7];
            L.vv[8 * t + 5] = dd;
            return dd;
        }
        const bool second = k == 1 || k == 4;
        const double dv = second ? v[1] : v[0];
        return k < 3 ? dv : dv - (second ? y[4] : y[3]);
    };
    // slacks floored at wfl, multipliers lam = mu0 / w  (first attempt: 1e-2 and 1)
    auto centre_duals = [&](double wfl, double mu0) {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            if (p_on[j]) {
                const double cv = pair_val(p_t[j], p_k[j]);
                const double sp = p_ep[j] - cv, sm = p_em[j] + cv;
                Pwp[j] = sp > wfl ? sp : wfl; Pwm[j] = sm > wfl ? sm : wfl;
                Plp[j] = mu0 / Pwp[j]; Plm[j] = mu0 / Pwm[j];
            } else { Pwp[j] = Pwm[j] = 1.0; Plp[j] = Plm[j] = 0.0; }
        }
    };
    // What the stage phases of an interior-point iteration need of the rows, from the current slacks / multipliers and controls / distances: primal
    // residuals, reciprocals (once per iteration), affine (predictor) targets in registers; barrier weights lam/w, lam+ - lam-, x+ - x-, lam w and
    // max |r_p| per pair to the [T][5] arrays.  Called where the pairs change: at the start of an attempt and right after the update of every
    // iteration (between the two barriers of its reach / complementarity reduction - round 6: it used to be a phase of its own at the top of the
    // next iteration), so the iteration itself starts with the fused stage phase.  A barrier must follow before the stage phase reads the arrays.
    auto pair_rows = [&]() {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            if (p_ok[j]) {
                const int o = 5 * p_t[j] + p_k[j];
                if (p_on[j]) {
                    const double cv = pair_val(p_t[j], p_k[j]);
                    const double wp = Pwp[j], wm = Pwm[j], lp = Plp[j], lm = Plm[j];
                    Prpp[j] = cv + wp - p_ep[j]; Prpm[j] = wm - cv - p_em[j];
                    const double iwp = frcp(wp), iwm = frcp(wm);
                    Piwp[j] = iwp; Piwm[j] = iwm; Pilp[j] = frcp(lp); Pilm[j] = frcp(lm);
                    Prcp[j] = lp * wp; Prcm[j] = lm * wm;
                    L.bw[o] = lp * iwp + lm * iwm;
                    L.cy[o] = lp - lm;
                    L.xd[o] = (lp * Prpp[j] - Prcp[j]) * iwp - (lm * Prpm[j] - Prcm[j]) * iwm;
                    L.lw[o] = Prcp[j] + Prcm[j];
                    L.ra[o] = fmax(fabs(Prpp[j]), fabs(Prpm[j]));
                } else { L.bw[o] = 0; L.cy[o] = 0; L.xd[o] = 0; L.lw[o] = 0; L.ra[o] = 0; }
            }
        }
    };
    // ---- landing (Args::land): state of the pairs - which of the two rows is active, the iterate the landing started from (slacks / multipliers in
    //      registers, controls / distances in L.sav) - and what the stage phases read of them while a landing round runs
    bool Lap[NPR], Lam[NPR];
    double Swp[NPR], Swm[NPR], Slp[NPR], Slm[NPR];
#pragma unroll
    for (int j = 0; j < NPR; ++j) { Lap[j] = Lam[j] = false; Swp[j] = Swm[j] = 1.0; Slp[j] = Slm[j] = 0.0; }
    // rows of a landing round at the current point x: r+ = c'x - e+, r- = -c'x - e- (kept in Prpp / Prpm), multiplier estimates nu in Plp / Plm;
    // barrier weight -> penalty rho on the active rows, lam+ - lam- -> nu+ - nu-, x+ - x- -> rho (r+ - r-) over the active rows
    auto land_rows = [&](const double rho) {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            if (p_ok[j]) {
                const int o = 5 * p_t[j] + p_k[j];
                if (p_on[j]) {
                    const double cv = pair_val(p_t[j], p_k[j]);
                    Prpp[j] = cv - p_ep[j]; Prpm[j] = -cv - p_em[j];
                    if (!Lap[j]) Plp[j] = 0.0;
                    if (!Lam[j]) Plm[j] = 0.0;
                    L.bw[o] = rho * ((Lap[j] ? 1.0 : 0.0) + (Lam[j] ? 1.0 : 0.0));
                    L.cy[o] = Plp[j] - Plm[j];
                    L.xd[o] = rho * ((Lap[j] ? Prpp[j] : 0.0) - (Lam[j] ? Prpm[j] : 0.0));
                } else { L.bw[o] = 0; L.cy[o] = 0; L.xd[o] = 0; }
                L.lw[o] = 0; L.ra[o] = 0;
            }
        }
    };
    // ... and of the verification pass behind it, at x+ (slacks w = e - c'x+ in Pwp / Pwm, multipliers nu in Plp / Plm): lam+ - lam- for the stage
    // gradients; "primal residual" -> the largest violation (a dropped row that is violated, an active row off its bound), relative to 1 + |e|;
    // "complementarity" -> the negative parts of the multipliers
    auto verify_rows = [&]() {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            if (p_ok[j]) {
                const int o = 5 * p_t[j] + p_k[j];
                if (p_on[j]) {
                    // (an active row sits on its bound to (nu* - nu_2) / rho after the two steps of the method of multipliers: held to 1e-9, a hundred times
                    // the threshold of a dropped row's violation, by the factor)
                    const double vp = (Lap[j] ? 1e-2 * fabs(Pwp[j]) : fmax(-Pwp[j], 0.0)) / (1.0 + fabs(p_ep[j]));
                    const double vm = (Lam[j] ? 1e-2 * fabs(Pwm[j]) : fmax(-Pwm[j], 0.0)) / (1.0 + fabs(p_em[j]));
                    L.cy[o] = Plp[j] - Plm[j];
                    L.ra[o] = fmax(vp, vm);
                    L.lw[o] = fmax(-Plp[j], 0.0) + fmax(-Plm[j], 0.0);
                } else { L.cy[o] = 0; L.ra[o] = 0; L.lw[o] = 0; }
                L.bw[o] = 0; L.xd[o] = 0;
            }
        }
    };
    // Start.  Cold: slacks floored at 1e-2, lam = 1/w (mu0 = 1).  Warm (ADMM iterations >= 1): the primal point is the previous
    // solution, so the slacks are the previous ones; they are floored at warm_wfl, the multipliers are the larger of the centred
    // ones (warm_mu0 / w) and those the previous solve ended with.  Starting with a small mu0 WITHOUT the old multipliers costs
    // iterations (measured: 75 -> 92 us per launch), with them it saves about one per solve (75 -> 67 us).
    if (warm) {
        centre_duals(a.warm_wfl, a.warm_mu0);
#pragma unroll
        for (int j = 0; j < NPR; ++j)
            if (p_on[j]) {
                if (pf_lkp[j] > Plp[j]) Plp[j] = pf_lkp[j];
                if (pf_lkm[j] > Plm[j]) Plm[j] = pf_lkm[j];
                if (a.hard_dmu > 0 && p_k[j] == 2) {
                    const double dl = a.hard_dmu - fmax(pf_lkp[j], pf_lkm[j]);
                    if (dl > 0) { Plp[j] += dl / Pwp[j]; Plm[j] += dl / Pwm[j]; }
                }
            }
    } else centre_duals(1e-2, 1.0);
    __syncthreads();                                          // (the roll-out of wave 0 is visible to everybody from here on)
    TR(140);
    // ---- rotation-consistency penalty -> scalar quadratic per stage from three POSE-INDEPENDENT sums over the terms (see
    // row_term), and the HINGE SCREENING: Im_su = a'p - cb - d >= a'p0 - cb - max_sd - |a| |p - p0|, so a term whose margin at the
    // reference position p0 exceeds DELTA |a| cannot be active while the stage position stays within DELTA of p0.  Both arrive in
    // reduced form - per (stage, GS-slot block) partial sums and a GS-bit near mask, made by the LamMuZ launch that produced the terms
    // (p0 = the trajectory it worked with: the pose table) - so this set-up has NO pass over the N terms; handed raw terms only
    // (rda_su_solve) it evaluates them here in the same grouping, so both forms round alike.  Each thread keeps the near masks of the
    // blocks of ITS (stage, chunk) slice in registers and only those terms are visited by the per-iteration hinge sums (same visiting
    // order as the streaming loop -> the sums are unchanged).  The assumption |p - p0| <= DELTA is verified on the nominal and after
    // every step; if violated the iteration continues with all terms.  Excluded terms are inactive at the verified solution, so it
    // satisfies the optimality conditions of the full problem.
    constexpr int MW = 4;
    constexpr double DELTA = SCREEN_DELTA;
    unsigned long long amask[MW] = {0, 0, 0, 0};
    bool screened = c.accelerated && a.P * KB * GS <= 64 * MW && (!masks_in || a.pose_ok);
    bool listed = false;                                       // (uniform) the near terms are in L.near
    {
        double saa = 0, sga = 0, sgx = 0;
        if (ract) {
            const double p0x = L.p0[rt], p0y = L.p0[T + rt];
            auto place = [&](int r, int kb, unsigned long long m8) {
                // the bit position only depends on the loop counters, i.e. it is the same in every thread (scalar word index and shift)
                const int bp = (r * KB + kb) * GS;
                const unsigned long long mm = screened ? m8 << (bp & 63) : 0ull;
                switch (bp >> 6) { case 0: amask[0] |= mm; break; case 1: amask[1] |= mm; break; case 2: amask[2] |= mm; break; default: amask[3] |= mm; break; }
            };
            for (int r = 0; r < a.P; ++r) {
                if (masks_in) {
                    for (int kb0 = 0; kb0 < KB; kb0 += 8) {
                        double q0[8], q1[8], q2[8]; unsigned long long mk[8];
                        if (r == 0 && kb0 == 0) {                 // (prefetched at kernel entry)
#pragma unroll
                            for (int k = 0; k < 8; ++k) { q0[k] = pre.q0[k]; q1[k] = pre.q1[k]; q2[k] = pre.q2[k]; mk[k] = pre.mk[k]; }
                        } else load_batch(r, kb0, q0, q1, q2, mk);
#pragma unroll
                        for (int k = 0; k < 8; ++k) if (kb0 + k < KB && rc_ + (kb0 + k) * nch < J) {
                            saa += q0[k]; sga += q1[k]; sgx += q2[k];
                            place(r, kb0 + k, mk[k] & ((1ull << GS) - 1));
                        }
                    }
                } else
                for (int kb = 0; kb < KB; ++kb) {
                    const int j = rc_ + kb * nch;
                    if (j >= J) break;
                    unsigned long long m8 = 0;
                    const size_t o = r * a.chunk + (size_t)rt * a.Nloc + GS * j;
                    double b0 = 0, b1 = 0, b2 = 0;
                    for (int row = 0; row < GS && GS * j + row < a.Nloc; ++row) {
                        const RowTerm q = row_term(a.ax[o + row], a.ay[o + row], a.gx[o + row], a.gy[o + row], a.cb[o + row], p0x, p0y, c.max_sd, true);
                        b0 += q.aa; b1 += q.ga; b2 += q.gxa;
                        m8 |= q.near ? 1ull << row : 0ull;
                    }
                    saa += b0; sga += b1; sgx += b2;
                    place(r, kb, m8);
                }
            }
        }
        MS(3); TR(123);
        // ---- near list (round 4).  The terms the masks name are fetched ONCE, here, into a compact stage-major list in LDS: (ax, ay, cb, stage).
        // The per-iteration hinge sums (phase 1 below) then take one term per thread - no trip to memory, no loop, no register cache - and
        // one thread per (stage, quantity) adds the stage's contributions up in list order (chunk, then bit: the order of the mask walk).
        // A dense active set is served better by the streaming loop (the sparse path is kept for < 30 %), and the nominal itself must lie within DELTA / 2 of
        // the screening reference.  (Round 6: the count of near terms IS the total of the list's scan and the reach of the nominal travels through the same
        // barrier pair - rounds 4-5 ran two block reductions, four barriers, in front of the scan for the same two numbers.)
        int mine = 0;
        if (screened) {
            if (ract) for (int w = 0; w < MW; ++w) mine += __popcll(amask[w]);
            L.ncnt[tid] = mine;
            double dv = 0;
            if (masks_in && tid < T) { double ex = L.s[tid + 1] - L.p0[tid], ey = L.s[(T + 1) + tid + 1] - L.p0[T + tid]; dv = sqrt(ex * ex + ey * ey); }
            if (masks_in) { dv = wave_allreduce(dv, true); if (lane == 0) L.red[280 + wave] = dv; }
            __syncthreads();
            if (wave == 0) {                                      // first entry of every stage (sto[T]: the total): stage totals, one wave scan (T <= 64)
                int tot = 0;
                if (lane < T) for (int k = 0; k < nch; ++k) tot += L.ncnt[lane * nch + k];
                int inc = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
                if (lane < T) L.sto[lane] = inc - tot;
                if (lane == T - 1) L.sto[T] = inc;
            }
            __syncthreads();
            const double cnt = (double)L.sto[T];
            const double dvm = masks_in ? fmax(fmax(L.red[280], L.red[281]), fmax(L.red[282], L.red[283])) : 0.0;
            if (cnt > 0.3 * (double)a.P * a.Nloc * T || dvm > 0.5 * DELTA) screened = false;
        }
        MS(14); TR(134);
        if (screened) {
            listed = L.sto[T] <= near_cap(T);
            if (listed && ract && mine > 0) {
                int off = L.sto[rt];
                for (int k = 0; k < rc_; ++k) off += L.ncnt[rt * nch + k];
                double *q = &L.near[4 * (size_t)off];
                const int Nl = a.Nloc;
                for (int w = 0; w < MW; ++w) {
                    unsigned long long m = amask[w];
                    while (m) {                                   // four loads in flight: UNCONDITIONAL ones (a slot without a term repeats the last one's
                        size_t o4[4]; int cnt = 0, bit = 0;       // address) - a load under `k < cnt` is conditional code with a wait of its own
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (m) { bit = 64 * w + __ffsll((long long)m) - 1; m &= m - 1; ++cnt; }
                            const int blk = bit / GS, row = bit % GS;
                            const int r = a.P == 1 ? 0 : blk / KB, kb = blk - r * KB;
                            o4[k] = r * a.chunk + (size_t)rt * Nl + GS * (rc_ + kb * nch) + row;
                        }
                        double x[4], y[4], cb[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { x[k] = a.ax[o4[k]]; y[k] = a.ay[o4[k]]; cb[k] = a.cb[o4[k]]; }
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k < cnt) { q[0] = x[k]; q[1] = y[k]; q[2] = cb[k]; q[3] = (double)rt; q += 4; }
                    }
                }
            }
        }
        TR(141);
        L.part[tid * 9] = saa; L.part[tid * 9 + 1] = sga; L.part[tid * 9 + 2] = sgx;
        __syncthreads();
        if (tid < T) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int k = 0; k < nch; ++k) { const double *pp = &L.part[(tid * nch + k) * 9]; s0 += pp[0]; s1 += pp[1]; s2 += pp[2]; }
            // Q2 = sum |dR'a|^2 = sum |a|^2 ;  Q1 = 2 sum (g + R'a).(dR'a) = 2 (cos sum g x a - sin sum g.a)
            L.Q2[tid] = s0; L.Q1[tid] = 2 * (L.csn[tid] * s2 - L.csn[T + tid] * s1);
        }
    }
    MS(12); TR(132);
    const double mcnt = (double)(6 * T + 4 * (T - 1));
    const double wz = c.dynamics == 2 ? 0.0 : 1.0;

    // Newton right-hand side gradient gh = gst + C'((lam*rp - rc)/w): with xd = x+ - x- per pair (L.xd, written by the pair threads for the
    // current right-hand side) the entries 3..7 of a stage are -xd3, -xd4, xd0 + xd3, xd1 + xd4, xd2 - formed where they are used
    // constants of the backward affine map for the current right-hand side: cb = [g_x - W g_v ; -Minv g_v]
    auto cb_entry = [&](const int t, const int r) {
            const double *gs = &L.gst[8 * t], *xd = &L.xd[5 * t], *wn = &L.Wn[WN * t], *m7 = &L.m7[8 * t];
            // (d_t eliminated: g' = g - m7 g7 / H77 on the entries 0..6; the rows of d in W / Minv are zero)
            const double g7 = gs[7] + xd[2], c7 = g7 * m7[7];
            const double g5 = gs[5] + (xd[0] + xd[3]) - m7[5] * c7, g6 = gs[6] + (xd[1] + xd[4]) - m7[6] * c7;
            double v;
            if (r < 5) {
                const double gr = (r < 3 ? gs[r] : (r == 3 ? gs[3] - xd[3] : gs[4] - xd[4])) - m7[r] * c7;
                v = gr - (wn[3 * r] * g5 + wn[3 * r + 1] * g6);
            } else {
                int k = r - 5;
                double n0 = k == 0 ? wn[15] : (k == 1 ? wn[16] : wn[17]);
                double n1 = k == 0 ? wn[16] : (k == 1 ? wn[18] : wn[19]);
                double n2 = k == 0 ? wn[17] : (k == 1 ? wn[19] : wn[20]);
                v = -(n0 * g5 + n1 * g6 + n2 * g7);
            }
            return v;
    };
    auto build_cb = [&](const int first = threadIdx.x, const int stride = NT) {
        for (int i = first; i < 8 * T; i += stride) {
            const int t = i >> 3, r = i & 7;
            L.Hb[HB * t + 6 * r + 5] = cb_entry(t, r);
        }
    };
    // ---- vector sweeps (wave 0, lane-parallel): one affine map per stage -----------------------------------
    typedef double d2 __attribute__((ext_vector_type(2)));
    struct Row { d2 v[3]; };
#define RW(k, i) ((k).v[(i) >> 1][(i) & 1])
// s_waitcnt lgkmcnt(0) at the END of a sweep stage: the next stage's rows (issued one stage ahead) have landed, so the
// compiler does not have to drain the freshly issued prefetch before the first use at the top of the next stage
#define LDS_DRAIN() __builtin_amdgcn_s_waitcnt(0xc07f)
    // (lin: the linear part only, entries 0..4 - the unit sweeps of the time split run while waves 0 / 1 write the constants, entry 5 of the same rows)
    auto ldrow = [&](const double *base, Row &k, const bool lin = false) {
        const d2 *q = reinterpret_cast<const d2 *>(__builtin_assume_aligned(base, 16));
        k.v[0] = q[0]; k.v[1] = q[1];
        if (lin) { k.v[2][0] = base[4]; k.v[2][1] = 0.0; } else k.v[2] = q[2];
    };
    // x+ = row . x[0..4] + c with x held by lanes 0..4 of the row: v_fmac_f64 with a DPP row_newbcast source operand
    // (gfx90a+ allows DPP on 64-bit VALU ops for row_newbcast) - one instruction per term instead of two v_readlane
    // and an FMA.  s_nop 1: a VGPR written by the previous VALU op needs two wait states before a DPP read.
    // One accumulator: a single wave issues an instruction every ~6.4 cycles whether or not it depends on the previous one
    // (tools/latency_micro.cpp), so a second accumulator only adds its initialisation and the final addition to the stage.
    auto affine = [&](const Row &k, double x, const bool with_const = true) {
        double e0 = with_const ? RW(k, 5) : 0.0;
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(e0) : "v"(x), "v"(RW(k, 0)), "v"(RW(k, 1)), "v"(RW(k, 2)), "v"(RW(k, 3)), "v"(RW(k, 4)));
        return e0;
    };
    // backward over the stages hi-1 .. lo: [p ; kk] <- Mb [p] + cb   (lanes 0..4 of a 16-lane DPP row carry p, lanes 5..7 deliver kk; r8 = the
    // calling lane's position in its row, < 8).  pl = p_hi in lanes 0..4; returns p_lo.  out: [8 t + r8].  with_const = false: the linear part
    // only (unit sweeps of the time split).  Only lanes with r8 < 8 may call; row_newbcast reads inside the 16-lane row, so the four rows of a
    // wave can run four different sweeps with the same instructions.
    // The rows of a stage are requested PD stages ahead (a ring of PD register rows): one stage of the map is ~56 cycles of dependent
    // FMAs (tools/dpp_micro.cpp) while an LDS read needs ~100 - with the rows of the NEXT stage only (round 1-2) every stage waited
    // for its rows: 151 cycles per stage measured in the kernel.
    constexpr int PD = 4;
    auto bwd_seg = [&](const int lo, const int hi, double pl, double *out, const bool with_const, const int r8) {
        Row k[PD];
#pragma unroll
        for (int j = 0; j < PD; ++j) if (hi - 1 - j >= lo) ldrow(L.Hb + HB * (hi - 1 - j) + 6 * r8, k[j]);
        for (int t0 = hi - 1; t0 >= lo; t0 -= PD) {
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int t = t0 - j;
                if (t >= lo) {
                    pl = affine(k[j], pl, with_const);
                    out[8 * t + r8] = pl;
                    if (t - PD >= lo) ldrow(L.Hb + HB * (t - PD) + 6 * r8, k[j]);
                }
            }
        }
        return pl;
    };
    // constants of the forward map of the stages lo .. hi-1: cf = [Fv kk ; kk_2]   (one wave; wsync before and after by the caller)
    auto cf_seg = [&](const int lo, const int hi) {
        for (int i = 6 * lo + lane; i < 6 * hi; i += 64) {
            int t = i / 6, r = i % 6;
            const double *kq = &L.kk[8 * t + 5], *F = &L.Ft[FT * t];
            L.Mf[MF * t + 6 * r + 5] = r < 5 ? Fel(F, r, 5) * kq[0] + Fel(F, r, 6) * kq[1] : kq[2];
        }
    };
    // forward over the stages lo .. hi-1: [dx+ ; v_2] <- Mf [dx] + cf   (lanes 0..4 carry dx; v = outputs 3..5); xl = dx_lo in lanes 0..4; lanes < 8 only
    auto fwd_seg = [&](const int lo, const int hi, double xl) {
        const int row = lane < 6 ? lane : 5;
        Row k[PD];
#pragma unroll
        for (int j = 0; j < PD; ++j) if (lo + j < hi) ldrow(L.Mf + MF * (lo + j) + 6 * row, k[j]);
        for (int t0 = lo; t0 < hi; t0 += PD) {
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int t = t0 + j;
                if (t < hi) {
                    L.dy[8 * t + lane] = xl;                    // entries 0..4 = dx_t
                    xl = affine(k[j], xl);
                    L.vv[8 * t + lane] = xl;                    // entries 3..5 = v_t
                    if (t + PD < hi) ldrow(L.Mf + MF * (t + PD) + 6 * row, k[j]);
                }
            }
        }
        return xl;
    };
    // ---- Riccati matrix recursion (wave 0): lane 8r+q owns entry (r,q) ------------------------------------------
    struct Row5 { d2 a, b; double c; };
    auto ldrow5 = [&](const double *base, Row5 &k) {
        const d2 *q = reinterpret_cast<const d2 *>(__builtin_assume_aligned(base, 16));
        k.a = q[0]; k.b = q[1]; k.c = base[4];
    };
#define R5(k, i) ((i) < 2 ? (k).a[(i) & 1] : ((i) < 4 ? (k).b[(i) & 1] : (k).c))
    struct MatK { double hb; Row5 fc, fr; };
    const int mq_ = lane >> 3, mr_ = lane & 7;          // lane 8q+r owns entry (r,q): a column of M per 8-lane group
    auto ldmat = [&](int t, MatK &k) {
        k.hb = L.Hb[HB * t + lane];
        ldrow5(L.Ft + FT * t + 6 * mq_, k.fc);
        ldrow5(L.Ft + FT * t + 6 * mr_, k.fr);
    };
    // Cross-lane operands: rows of P and the blocks of M travel through a 2 x 64-double LDS scratch of this wave (LDS
    // executes one wave's instructions in order, so a write followed by reads needs no barrier); the column of X a
    // lane needs sits in its own 8-lane group and is consumed straight from the neighbours' registers by
    // v_fmac_f64 with a DPP row_newbcast operand (bank_mask selects the lower / upper group of the 16-lane row).
    double *const Px = L.red + 16, *const Ms = L.red + 80;             // wave 0 (stages [msp, T)): P_msp stays in Px for the interface of the time split
    double *const PxA = L.red + 144, *const MsA = L.red + 208;          // wave 1 (stages [0, msp))
    // time split (see the header of the pass loop): compile-time split point of this instantiation, runtime switch
    constexpr int MSP = split_point(TT > 0 ? TT : 1);
    const int msp = (MSP > 0 && a.split) ? MSP : 0;
    constexpr int MQ = MSP > 0 ? MSP : 1;                               // = msp wherever the split code runs (compile-time: constant addresses, unrolled loops)
    unsigned long long stopf = 0, seq = 0;               // seq = tag of the current interior-point iteration (early-verdict flags)
    unsigned long long *const flag_meas = reinterpret_cast<unsigned long long *>(L.red + 12), *const flag_stop = flag_meas + 1;
    // (round 6) flag_fail = seq: a recursion of this iteration broke down (instead of __syncthreads_or: 1.4 k cycles per iteration on the traced timeline);
    // flag_unit = seq: wave 3's unit sweep of this factorisation is in LDS (wave 2 goes on to the interface matrix without a block barrier)
    unsigned long long *const flag_fail = flag_meas + 2, *const flag_unit = flag_meas + 3, *const flag_unit2 = reinterpret_cast<unsigned long long *>(L.pv + 4);
    double okmin = 1.0, lastp = 0.0;
    auto mat_step = [&](int t, const MatK &k, double *const Px, double *const Ms) {
        // X = P F : lane (q,i) needs row i of P
        Row5 pr; ldrow5(Px + 8 * mr_, pr);
        double x = R5(pr, 0) * R5(k.fc, 0) + R5(pr, 1) * R5(k.fc, 1);
        x += R5(pr, 2) * R5(k.fc, 2); x += R5(pr, 3) * R5(k.fc, 3); x += R5(pr, 4) * R5(k.fc, 4);
        // M = Hb + F' X : lane (q,r) needs X[0..4][q] = positions 0..4 of its own group
        // (64-bit DPP takes full row/bank masks only: both halves of the row are formed and the own group's is kept)
        double me = 0.0, mo = 0.0;
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %3 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %6 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %7 row_newbcast:12 row_mask:0xf bank_mask:0xf"
                     : "+v"(me), "+v"(mo) : "v"(x), "v"(R5(k.fr, 0)), "v"(R5(k.fr, 1)), "v"(R5(k.fr, 2)), "v"(R5(k.fr, 3)), "v"(R5(k.fr, 4)));
        const double m = k.hb + ((mq_ & 1) ? mo : me);
        Ms[8 * mr_ + mq_] = m;
        // pivot block Mvv (rows/cols 5, 6: d_t was eliminated when the stage Hessians were assembled), M[r][5..6], M[5..6][q]
        const double m00 = Ms[45], m01 = Ms[46], m11 = Ms[54];
        const double a0 = Ms[8 * mr_ + 5], a1 = Ms[8 * mr_ + 6];
        double b0 = Ms[40 + mq_], b1 = Ms[48 + mq_];
        stopf = __atomic_load_n(flag_stop, __ATOMIC_RELAXED);   // "this iterate has converged" (wave 2), same LDS round
        asm volatile("" : "+v"(b0), "+v"(b1));               // keep these loads in the same LDS round as the pivot block
        // inverse of the 2 x 2 block on every lane; reciprocal of the determinant by v_rcp_f64 + two Newton steps
        const double det = m00 * m11 - m01 * m01;
        double id = __builtin_amdgcn_rcp(det);
        id = __builtin_fma(id, __builtin_fma(-det, id, 1.0), id);
        id = __builtin_fma(id, __builtin_fma(-det, id, 1.0), id);
        const double n00 = m11 * id, n01 = -m01 * id, n11 = m00 * id;
        // W row r = M[r][5..6] Minv ;  P = Mxx - W Mvx
        const double w0 = a0 * n00 + a1 * n01;
        const double w1 = a0 * n01 + a1 * n11;
        double pn = m - (w0 * b0 + w1 * b1);
        Px[8 * mr_ + mq_] = (mr_ < 5 && mq_ < 5) ? pn : 0.0;
        double *o = &L.Wn[WN * t];                            // (the entries that belong to d - W[.][2], Minv[.][2] - are zero for the whole solve: set-up)
        if (mq_ == 0 && mr_ < 5) { o[3 * mr_] = w0; o[3 * mr_ + 1] = w1; }      // (the q = 0 lane of a row holds its whole W row: no selects)
        if (lane == 63) { o[15] = n00; o[16] = n01; o[18] = n11; }
        // breakdown check: the running minimum of the leading minors' signs (one v_min each; a NaN shows in the last P instead)
        okmin = fmin(okmin, fmin(m00, det));
        lastp = pn;
        return true;
    };
    // the recursion over the stages hi-1 .. lo from P = 0 (one wave, its own scratch); false = the factorisation broke down
    auto mat_range = [&](const int lo, const int hi, double *const Px_, double *const Ms_) {
        Px_[lane] = 0.0;
        MatK ka, kb;
        okmin = 1.0; lastp = 0.0;
        if (hi <= lo) return true;
        ldmat(hi - 1, ka);
        // The termination measures do not need the factorisation: waves 2 and 3 evaluate the test while this wave
        // factorises and raise *flag_stop = seq when the iterate has converged; the recursion of that (last, unused)
        // factorisation is then abandoned.  The test itself is repeated by all threads after the barrier.
        for (int t = hi - 1; t >= lo; t -= 2) {
            if (t - 1 >= lo) ldmat(t - 1, kb);
            mat_step(t, ka, Px_, Ms_);
            if (stopf == seq) break;
            if (t - 1 >= lo) {
                if (t - 2 >= lo) ldmat(t - 2, ka);
                mat_step(t - 1, kb, Px_, Ms_);
            }
        }
        return (okmin > 0 && lastp == lastp) || stopf == seq;
    };
    // ---- time split, interface block xs: X (25) | S^-1 (25) | x0 (5) at 76 | p_m (5) at 82
    double *const XS_X = L.xs, *const XS_SI = L.xs + 25, *const XS_x0 = L.xs + 76, *const XS_pm = L.xs + 82;
    // unit backward sweeps of segment A (no constants): p_msp = e_i; waves 2 (i = DPP row 0..3) and 3 (i = 4).  Then b = F_v' p_{t+1} per stage.
    auto unit_sweeps = [&]() {
        const int r8 = lane & 15, ui = wave == 2 ? (lane >> 4) : 4;
        const bool act = r8 < 10 && (wave == 2 || lane < 16);
        if (act) {
            // lanes 0..7 of the row: the backward map of the stage (rows of Mb) without its constant - p(i)_t, kk(i)_t; lanes 8, 9: the columns 5, 6 of F as
            // rows - b(i)_t = F_v' p(i)_{t+1} comes out of the same five v_fmac_f64_dpp (the input of stage t IS p_{t+1})
            const double *base = r8 < 8 ? L.Hb + 6 * r8 : L.Ft + 6 * (r8 - 3);
            const int stride = r8 < 8 ? HB : FT;
            double *const outp = r8 < 8 ? L.uk + 8 * MQ * ui + r8 : L.ub + 2 * MQ * ui + (r8 - 8);
            const int ostride = r8 < 8 ? 8 : 2;
            double pl = r8 == ui ? 1.0 : 0.0;
            Row k[PD];
#pragma unroll
            for (int j = 0; j < PD; ++j) if (MQ - 1 - j >= 0) ldrow(base + stride * (MQ - 1 - j), k[j], true);
            for (int t0 = MQ - 1; t0 >= 0; t0 -= PD) {
#pragma unroll
                for (int j = 0; j < PD; ++j) {
                    const int t = t0 - j;
                    if (t >= 0) {
                        pl = affine(k[j], pl, false);
                        outp[ostride * t] = pl;
                        if (t - PD >= 0) ldrow(base + stride * (t - PD), k[j], true);
                    }
                }
            }
        }
    };
    // sum over the 8 lanes of an aligned group (xor 1, xor 2, half mirror)
    auto sum8 = [&](double v) { v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); return v; };
    // X[j][i] = sum_t b(j)_t . kk(i)_t : symmetric, 15 entries x 4 lanes (each lane the stages t = q mod 4), wave 2.  Then S = I - X P_msp (rows in
    // lanes 0..4), S^-1 by Gauss-Jordan and Y = S^-1 X.  The elimination runs WITHOUT row exchanges, the pivot row reaching the other lanes as a DPP
    // operand of the update itself (one v_fmac_f64_dpp per entry): on the recorded interface matrices (tools/experiments/two_segment.py) the
    // pivots stay above 0.05 max|S| and the inverse is exact to 1e-15 - S = I + Phi P with Phi, P positive semidefinite has real eigenvalues >= 1,
    // but its leading minors are not guaranteed, so a pivot below 1e-6 max|S| sends the iteration to the variant with partial pivoting.
    auto interface_matrix = [&]() {
        if (lane < 60) {
            const int e = lane >> 2, q = lane & 3;
            // (j, i), j <= i, of entry e = 0..14:  rows start at 0, 5, 9, 12, 14
            const int j = e >= 14 ? 4 : (e >= 12 ? 3 : (e >= 9 ? 2 : (e >= 5 ? 1 : 0)));
            const int i = e - (j == 0 ? 0 : (j == 1 ? 5 : (j == 2 ? 9 : (j == 3 ? 12 : 14)))) + j;
            double x = 0;
            const double *ubj = &L.ub[j * MQ * 2], *uki = &L.uk[8 * MQ * i + 5];
#pragma unroll
            for (int tt = 0; tt < (MSP + 3) / 4; ++tt) {
                const int t = q + 4 * tt;
                if (t < MQ) x += ubj[2 * t] * uki[8 * t] + ubj[2 * t + 1] * uki[8 * t + 1];
            }
            x += dpp_f64<0xB1>(x); x += dpp_f64<0x4E>(x);
            if (q == 0) { XS_X[5 * j + i] = x; XS_X[5 * i + j] = x; }
        }
        wsync();
        double am[5], ai[5];
        const int r = lane < 5 ? lane : 0;
        {
            double xr[5];
#pragma unroll
            for (int k2 = 0; k2 < 5; ++k2) xr[k2] = XS_X[5 * r + k2];
#pragma unroll
            for (int c2 = 0; c2 < 5; ++c2) {
                double v = 0;
#pragma unroll
                for (int k2 = 0; k2 < 5; ++k2) v += xr[k2] * Px[8 * k2 + c2];
                am[c2] = lane < 5 ? (r == c2 ? 1.0 : 0.0) - v : 0.0;     // (the other lanes run along; nothing of theirs is read)
                ai[c2] = (lane < 5 && r == c2) ? 1.0 : 0.0;
            }
        }
        double smax = 0;
#pragma unroll
        for (int c2 = 0; c2 < 5; ++c2) smax = fmax(smax, fabs(am[c2]));
        smax = fmax(fmax(bcast(smax, 0), bcast(smax, 1)), fmax(fmax(bcast(smax, 2), bcast(smax, 3)), bcast(smax, 4)));
        double s0[5], pmin = smax;
#pragma unroll
        for (int c2 = 0; c2 < 5; ++c2) s0[c2] = am[c2];
        // Gauss-Jordan, pivot (k, k): every lane adds nf x (row k) to its row, nf = -a[.][k] / a[k][k]; the pivot lane itself nf = 1 / a[k][k] - 1
#define SU_GJ_STEP(K)                                                                                                                   \
        {                                                                                                                                \
            const double pk = dpp_f64<0x150 + K>(am[K]);                                                                                  \
            pmin = fmin(pmin, fabs(pk));                                                                                                  \
            const double ip = frcp(pk);                                                                                                    \
            const double nf = lane == K ? ip - 1.0 : -am[K] * ip;                                                                           \
            asm volatile("s_nop 1\n\t"                                                                                                   \
                         "v_fmac_f64_dpp %0, %0, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %1, %1, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %2, %2, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %3, %3, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %4, %4, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %5, %5, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %6, %6, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %7, %7, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %8, %8, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"                                  \
                         "v_fmac_f64_dpp %9, %9, %10 row_newbcast:" #K " row_mask:0xf bank_mask:0xf"                                       \
                         : "+v"(am[0]), "+v"(am[1]), "+v"(am[2]), "+v"(am[3]), "+v"(am[4]), "+v"(ai[0]), "+v"(ai[1]), "+v"(ai[2]), "+v"(ai[3]), "+v"(ai[4]) \
                         : "v"(nf));                                                                                                      \
        }
        SU_GJ_STEP(0) SU_GJ_STEP(1) SU_GJ_STEP(2) SU_GJ_STEP(3) SU_GJ_STEP(4)
#undef SU_GJ_STEP
        pmin = fmin(fmin(bcast(pmin, 0), bcast(pmin, 1)), fmin(fmin(bcast(pmin, 2), bcast(pmin, 3)), bcast(pmin, 4)));
        int mycol = lane < 5 ? lane : 0;
        if (!(pmin > 1e-6 * smax)) {               // (uniform; not seen on any recorded system) the same elimination with partial pivoting
#pragma unroll
            for (int c2 = 0; c2 < 5; ++c2) { am[c2] = lane < 5 ? s0[c2] : 0.0; ai[c2] = (lane < 5 && r == c2) ? 1.0 : 0.0; }
            bool usedrow = lane >= 5;
#pragma unroll
            for (int k2 = 0; k2 < 5; ++k2) {
                const double cand = usedrow ? -1.0 : fabs(am[k2]);
                int pl_ = 0; double best = bcast(cand, 0);
                { const double c1 = bcast(cand, 1); if (c1 > best) { best = c1; pl_ = 1; } }
                { const double c1 = bcast(cand, 2); if (c1 > best) { best = c1; pl_ = 2; } }
                { const double c1 = bcast(cand, 3); if (c1 > best) { best = c1; pl_ = 3; } }
                { const double c1 = bcast(cand, 4); if (c1 > best) { best = c1; pl_ = 4; } }
                const int pv_ = __builtin_amdgcn_readfirstlane(pl_);
                double pm_[5], pi_[5];
#pragma unroll
                for (int c2 = 0; c2 < 5; ++c2) {
                    pm_[c2] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(am[c2]), pv_), __builtin_amdgcn_readlane(__double2loint(am[c2]), pv_));
                    pi_[c2] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ai[c2]), pv_), __builtin_amdgcn_readlane(__double2loint(ai[c2]), pv_));
                }
                const double ip = frcp(pm_[k2]);
                if (lane == pv_) {
#pragma unroll
                    for (int c2 = 0; c2 < 5; ++c2) { am[c2] = pm_[c2] * ip; ai[c2] = pi_[c2] * ip; }
                    usedrow = true; mycol = k2;
                } else {
                    const double f = am[k2] * ip;
#pragma unroll
                    for (int c2 = 0; c2 < 5; ++c2) { am[c2] -= f * pm_[c2]; ai[c2] -= f * pi_[c2]; }
                }
            }
        }
        if (lane < 5) {
#pragma unroll
            for (int c2 = 0; c2 < 5; ++c2) XS_SI[5 * mycol + c2] = ai[c2];
        }
    };
    // x0[j] = sum_t b(j)_t . kk_t of A's own backward sweep (wave 1: 5 x 8 lanes, each the stages t = q mod 8)
    auto interface_x0 = [&]() {
        if (lane < 40) {
            const int j = lane >> 3, q = lane & 7;
            double x = 0;
            const double *ubj = &L.ub[j * MQ * 2];
#pragma unroll
            for (int tt = 0; tt < (MSP + 7) / 8; ++tt) {
                const int t = q + 8 * tt;
                if (t < MQ) x += ubj[2 * t] * L.kk[8 * t + 5] + ubj[2 * t + 1] * L.kk[8 * t + 6];
            }
            x = sum8(x);
            if (q == 0) XS_x0[j] = x;
        }
    };
    // x_m = S^-1 (x0 + X p_m) (lanes 0..4), pi = P_msp x_m + p_m; returns (x_m, pi) of lane j < 5.  Called by waves 0 and 1 alike.
    auto interface_solve = [&](double &xm, double &pi) {
        const int j = lane < 5 ? lane : 0;
        double si[5], xx[5], pmv[5], pj[5];
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) { si[k2] = XS_SI[5 * j + k2]; xx[k2] = XS_X[5 * j + k2]; pmv[k2] = XS_pm[k2]; pj[k2] = Px[8 * j + k2]; }
        double rr = XS_x0[j], w = XS_pm[j];
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) rr += xx[k2] * pmv[k2];
        double v = 0.0;
        // x_m,j = sum_k S^-1[j][k] rr_k and pi_j = p_m,j + sum_k P[j][k] x_m,k : the vectors sit in lanes 0..4
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(v) : "v"(rr), "v"(si[0]), "v"(si[1]), "v"(si[2]), "v"(si[3]), "v"(si[4]));
        xm = v;
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(w) : "v"(xm), "v"(pj[0]), "v"(pj[1]), "v"(pj[2]), "v"(pj[3]), "v"(pj[4]));
        pi = w;
    };

    // Attempts (same rule as the oracle): when the cold one ends without convergence -- the iteration cap, ~0.1 % of closed-loop
    // solves, where the iterates cycle -- restart from the same nominal with a more central point (slack floor 0.1, mu0 = 10),
    // every hinge term in play and, since round 5, as the plain path-following iteration SU_SAFE_* (`safe`).
    int status = 1, it = 0, used = 0;
    bool have_acc = false; double acc_merit = 0.0;            // (uniform) safety net
    if (tid == 0) { *flag_meas = 0; *flag_stop = 0; *flag_fail = 0; *flag_unit = 0; *flag_unit2 = 0; }
    bool ref_pending = a.ref_flag != nullptr;      // the reference of a tracked tick is sampled by a second workgroup: picked up at its first use
    MS(9); TR(101);
    // Attempts: [-1: the warm start, at most warm_cap = 30 iterations.  Where consecutive su-problems are close (static scenes) it
    // converges within 3-4; with many moving obstacles it needs as many iterations as the cold start (8-20) but does arrive:
    // cutting it at 5 / 7 / 12 iterations and starting over cost +29 / +35 / +5 % on the dynamic_obs benchmark, 30 costs
    // nothing], 0: the cold start, 1: the central restart described above.
    for (int attempt = a.first_attempt ? 1 : (warm ? -1 : 0); attempt < 2 && status != 0; ++attempt) {
    const bool safe = attempt == 1;                // (uniform) the last-resort iteration, see SU_SAFE_*
    double al_prev = 0.0;
    int land = 0, land_rounds = 0, land_its = 0;   // (uniform) 0: interior point; 1: this pass is a landing round; 2: this pass verifies one; passes spent on landings
    int land_level = a.land_level0;                // (uniform) landings refused so far in this attempt (+ the level the launch starts at): k -> stop at 1e-2^k x land_tol (not below the tight tolerances); 99 -> no landing
    double land_rho = 0.0;
    bool expect_conv = attempt < 0 && a.land != 0 && a.land_first != 0;      // (landing first: the first pass of a warm attempt is a light one)
    int spec_dec = 0;                              // decade of the start's relative dual residual (statistics of the speculative landings)
    bool blind = false;                            // (uniform) the running landing started blind
    bool spec = false, spec_tried = false;         // (uniform) the running landing is a speculative one (land_first = 2); one has been refused in this attempt
    // a landing is refused: back to the interior-point iterate it started from, and on to the tight tolerances (all threads; ends with a barrier)
    auto land_refuse = [&](const bool last) {
        __syncthreads();
        for (int i = tid; i < 2 * T; i += NT) L.u[i] = L.sav[i];
        for (int i = tid; i < T; i += NT) L.d[i] = L.sav[2 * T + i];
#pragma unroll
        for (int j = 0; j < NPR; ++j) { Pwp[j] = Swp[j]; Pwm[j] = Swm[j]; Plp[j] = Slp[j]; Plm[j] = Slm[j]; }
        __syncthreads();
        rollout();
        __syncthreads();
        LSTAT(1, 1);
        land = 0; expect_conv = false;
        if (blind) { blind = false; res.blind = 2; }
        if (spec) { spec = false; spec_tried = true; res.spec = 2; } else land_level = last ? 99 : land_level + 1;      // (a refused speculation does not use up a landing level)
        pair_rows();
        __syncthreads();
    };
    if (attempt == 1 || (attempt == 0 && warm)) {
        __syncthreads();
        clip_controls(0.01);
        rollout();
        __syncthreads();
        if (attempt == 1) { screened = false; centre_duals(1e-1, 10.0); }
        else centre_duals(1e-2, 1.0);
        __syncthreads();
        if (screened) {            // (the restart re-clips the nominal with another margin: its positions get the reach check of their own - rare path)
            double dv = 0;
            if (tid < T) { double ex = L.s[tid + 1] - L.p0[tid], ey = L.s[(T + 1) + tid + 1] - L.p0[T + tid]; dv = sqrt(ex * ex + ey * ey); }
            dv = block_reduce(dv, L.red, tid, true);
            if (dv > DELTA) screened = false;
        }
        status = 1;
    }
    TR(150);
    pair_rows();
    TR(151);
    __syncthreads();
    if (attempt < 0 && a.land != 0 && a.land_blind != 0 && a.land_rho_prev > 0) {
        // ---- blind landing (Args::land_blind): the state a speculative landing starts in, without the measuring pass in front of it
        for (int i = tid; i < 2 * T; i += NT) { L.sav[i] = L.u[i]; L.base[i] = L.u[i]; }
        for (int i = tid; i < T; i += NT) { L.sav[2 * T + i] = L.d[i]; L.base[2 * T + i] = L.d[i]; }
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            Swp[j] = Pwp[j]; Swm[j] = Pwm[j]; Slp[j] = Plp[j]; Slm[j] = Plm[j];
            Lap[j] = p_on[j] && pf_lkp[j] > Pwp[j]; Lam[j] = p_on[j] && pf_lkm[j] > Pwm[j];
        }
        land_rho = a.land_rho_prev;
        land = 1; land_rounds = 1; expect_conv = false; spec = true; blind = true;
        LSTAT(4, 1); LSTAT(18, 1);
        __syncthreads();
        land_rows(land_rho);
        __syncthreads();
    }
    const int it_cap = attempt < 0 ? a.warm_cap : (attempt == 0 ? SU_COLD_CAP : 100);       // (the cold attempt: 50 since round 5, see the oracle)
    const double tau_min = attempt < 0 ? a.warm_tau : 0.995;
    double mu_prev = 1.0;
    for (it = 0; it < it_cap || land != 0; ++it) {
        TR(1);
        seq += 1;
        if (land != 0) land_its += 1;
        const double heps = (land == 0 && attempt >= 0 && it >= SU_CENTRE_FROM) ? SU_SMOOTH_K * sqrt(mu_prev) : 0.0;      // (a landing works on the true hinge terms)
        // stop of this pass (also the early verdict of wave 2, phase 4): the tight tolerances, or - while the solve is to be landed - 1e-2^level x land_tol
        const bool landing = a.land != 0 && land_level < 99;       // (uniform) the interior point stops at 1e-2^level x land_tol and is landed
        double lsc = 1.0;
        for (int k = 0; k < land_level && k < 8; ++k) lsc *= 1e-2;
        // every landing refused (level 99 with the switch on): the fallback is the interior point itself, run SU_LAND_FALLBACK x tighter than su_tol - in a direction the
        // su-problem is nearly singular in, the point at su_tol is 1e-5 .. 1e-4 from the vertex (mirrors oracle/rda_oracle.c).  Not reached within the
        // cap: the safety net (Args::accept) has the iterate that met su_tol.
        const double fb = (a.land != 0 && !landing) ? SU_LAND_FALLBACK : 1.0;
        const double t_rd = landing ? fmax(lsc * a.land_tol[0], c.tol_rd) : fb * c.tol_rd, t_rp = landing ? fmax(lsc * a.land_tol[1], c.tol_rp) : fb * c.tol_rp,
                     t_mu = landing ? fmax(lsc * a.land_tol[2], c.tol_mu) : fb * c.tol_mu;
        const bool land_last = t_rd <= c.tol_rd && t_rp <= c.tol_rp && t_mu <= c.tol_mu;      // a landing refused at the tight tolerances is the last one
        // (rescue phase: the smoothed hinge is non-zero for EVERY term - the oracle sums them all, so does this pass; such iterations are rare)
        const bool screened_now = screened && !(heps > 0);
        // ---- (1)-(3) ONE phase, no barrier inside (round 6; rounds 1-5: hinge contributions | barrier | stage sums | barrier | stage derivatives and
        //      pair rows | barrier | gradients and Hessian bases - four phases of 1.5-4 k cycles each, 9.3 k of a 44.7 k-cycle iteration).  Lane i = 8 t + r
        //      belongs to the aligned 8-lane group of stage t and owns row r of its 8 x 8 Hessian base: the group (a) sums the stage's stretch of the
        //      near list - lane r the terms j0 + r, j0 + r + 8, ... - and adds the eight partials up with three DPP steps (every lane of the group then
        //      holds the nine hinge sums), (b) forms the stage derivatives wrt (s_next, d) in registers (each lane for itself: same loads, no trip
        //      through LDS), (c) writes entry r of the stage gradient and (d) row r of the Hessian base.  The rows' barrier weights / multiplier
        //      differences (L.bw, L.cy) were left by the pair threads at the END of the previous iteration (pair_rows).
        //      Solves whose near terms are not in the LDS list (more than near_cap, unscreened, rescue phase): (stage, chunk) partials over the masks / over
        //      every term -> L.hs first, as before.
        const bool fused_hinge = screened_now && listed;
        if (!fused_hinge) {
            double sxx = 0, sxy = 0, syy = 0, sx = 0, sy = 0, s1 = 0, ix = 0, iy = 0, i1 = 0;
            if (ract) {
                const double px = L.s[rt + 1], py = L.s[(T + 1) + rt + 1], dd = L.d[rt];
                // (rescue phase of a cycling cold attempt: the hinge terms are smoothed over a width heps = SU_SMOOTH_K sqrt(mu), same rule
                // and reason as the oracle's su_solve_impl - a term that switches on and off for ever; heps = 0 otherwise)
                auto term = [&](double ax, double ay, double cb) {
                    double Im = ax * px + ay * py - cb - dd;
                    if (c.accelerated && heps > 0) {
                        const double e2 = 4 * heps * heps, rr = sqrt(Im * Im + e2), sv = 0.5 * (rr - Im), ds = 0.5 * (Im / rr - 1.0), d2 = 0.5 * e2 / (rr * rr * rr);
                        const double c1 = sv * ds, c2 = ds * ds + sv * d2;
                        sxx += c2 * ax * ax; sxy += c2 * ax * ay; syy += c2 * ay * ay; sx += c2 * ax; sy += c2 * ay; s1 += c2;
                        ix += c1 * ax; iy += c1 * ay; i1 += c1;
                    } else
                    if (!c.accelerated || Im < 0) {
                        sxx += ax * ax; sxy += ax * ay; syy += ay * ay; sx += ax; sy += ay; s1 += 1.0;
                        ix += Im * ax; iy += Im * ay; i1 += Im;
                    }
                };
                const int Nl = a.Nloc;
                if (screened_now) {
                    // (more near terms than the list holds) visit only the terms that may be active, four loads in flight
                    for (int w = 0; w < MW; ++w) {
                        unsigned long long m = amask[w];
                        while (m) {
                            size_t off[4]; int cnt = 0, bit = 0;           // (unconditional loads, as in the near-list fill of the set-up)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (m) { bit = 64 * w + __ffsll((long long)m) - 1; m &= m - 1; ++cnt; }
                                const int blk = bit / GS, row = bit % GS;
                                const int r = a.P == 1 ? 0 : blk / KB, kb = blk - r * KB;
                                off[q] = r * a.chunk + (size_t)rt * Nl + GS * (rc_ + kb * nch) + row;
                            }
                            double x[4], y[4], cb[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { x[q] = a.ax[off[q]]; y[q] = a.ay[off[q]]; cb[q] = a.cb[off[q]]; }
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (q < cnt) term(x[q], y[q], cb[q]);
                        }
                    }
                } else
                for (int r = 0; r < a.P; ++r) {                    // every term, in the same order: shard, block of the slice, slot
                    for (int kb = 0; kb < KB; ++kb) {
                        const int j = rc_ + kb * nch;
                        if (j >= J) break;
                        const size_t o = r * a.chunk + (size_t)rt * Nl + GS * j;
                        const double *pax = a.ax + o, *pay = a.ay + o, *pcb = a.cb + o;
                        const int nr = Nl - GS * j < GS ? Nl - GS * j : GS;
                        int n = 0;
                        for (; n + 8 <= nr; n += 8) {                  // eight independent loads in flight per array
                            double x[8], y[8], cb[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) { x[k] = pax[n + k]; y[k] = pay[n + k]; cb[k] = pcb[n + k]; }
#pragma unroll
                            for (int k = 0; k < 8; ++k) term(x[k], y[k], cb[k]);
                        }
                        for (; n < nr; ++n) term(pax[n], pay[n], pcb[n]);
                    }
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
            __syncthreads();
        }
        for (int i0 = 0; i0 < 8 * T; i0 += NT) {                   // (one round for T <= 32)
            const int i = i0 + tid; const bool act = i < 8 * T;
            const int t = act ? i >> 3 : 0, r = i & 7;
            double h[9];
            if (fused_hinge) {
                double sxx = 0, sxy = 0, syy = 0, sx = 0, sy = 0, s1 = 0, ix = 0, iy = 0, i1 = 0;
                const double px = L.s[t + 1], py = L.s[(T + 1) + t + 1], dd = L.d[t];
                const int j1 = act ? L.sto[t + 1] : 0;
                int j = L.sto[t] + r;
                auto term = [&](const d2 q01, const double cb) {
                    const double ax = q01[0], ay = q01[1];
                    const double Im = ax * px + ay * py - cb - dd;
                    if (Im < 0) {
                        sxx += ax * ax; sxy += ax * ay; syy += ay * ay; sx += ax; sy += ay; s1 += 1.0;
                        ix += Im * ax; iy += Im * ay; i1 += Im;
                    }
                };
                for (; j + 8 < j1; j += 16) {                      // two terms in flight, added in list order
                    const d2 qa = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(&L.near[4 * j], 16)); const double ca = L.near[4 * j + 2];
                    const d2 qb = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(&L.near[4 * (j + 8)], 16)); const double cb = L.near[4 * (j + 8) + 2];
                    term(qa, ca); term(qb, cb);
                }
                if (j < j1) {
                    const d2 qa = *reinterpret_cast<const d2 *>(__builtin_assume_aligned(&L.near[4 * j], 16)); const double ca = L.near[4 * j + 2];
                    term(qa, ca);
                }
                TR(3);
                h[0] = sum8(sxx); h[1] = sum8(sxy); h[2] = sum8(syy); h[3] = sum8(sx); h[4] = sum8(sy); h[5] = sum8(s1);
                h[6] = sum8(ix); h[7] = sum8(iy); h[8] = sum8(i1);
                TR(4);
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) h[k] = L.hs[9 * t + k];
            }
            mark(1);
            if (ref_pending) {                     // (uniform) first pass of a tracked tick: the hinge sums above did not need the reference,
                ref_pending = false;               // the stage gradients below do.  L.part (the functor's scratch) is free from here to the next phase 1
                ref_wait(L.part);
                for (int q = tid; q < 3 * (T + 1); q += NT) L.ref[q] = a.ref[q];
                __syncthreads();
                MS(9);
            }
            // ---- (2) derivatives of the stage cost wrt w = (s_next, d): the seven distinct entries of the symmetric 4 x 4 block, the gradient ---------
            double gw0, gw1, gw2, gw3, h00, h01, hd0, h11, hd1, h22, hdd;
            {
                const double st0 = L.s[t + 1], st1 = L.s[(T + 1) + t + 1], st2 = L.s[2 * (T + 1) + t + 1];
                double gs0 = 2 * c.ws * 1.0 * (st0 - L.ref[t + 1]), gs1 = 2 * c.ws * 1.0 * (st1 - L.ref[(T + 1) + t + 1]);
                double gs2 = 2 * c.ws * wz * (st2 - L.ref[2 * (T + 1) + t + 1]);
                double Hs00 = 2 * c.ws, Hs11 = 2 * c.ws, Hs22 = 2 * c.ws * wz, Hs01 = 0;
                const double dl = st2 - L.phin[t];
                gs2 += 0.5 * c.ro2 * (L.Q1[t] + 2 * L.Q2[t] * dl); Hs22 += c.ro2 * L.Q2[t];
                gs0 += c.ro1 * h[6]; gs1 += c.ro1 * h[7];
                Hs00 += c.ro1 * h[0]; Hs01 += c.ro1 * h[1]; Hs11 += c.ro1 * h[2];
                h00 = Hs00; h01 = Hs01; hd0 = -c.ro1 * h[3]; h11 = Hs11; hd1 = -c.ro1 * h[4]; h22 = Hs22; hdd = c.ro1 * h[5];
                gw0 = gs0; gw1 = gs1; gw2 = gs2; gw3 = -c.ro1 * h[8] - c.slack_gain;
                if (act && r == 0) {       // (termination measure |g|_inf, scale of the landing's penalty: wave 3, phase 4)
                    double *gw = &L.gw[4 * t]; gw[0] = gw0; gw[1] = gw1; gw[2] = gw2; gw[3] = gw3;
                    L.hmx[t] = fmax(fmax(h00, h11), fmax(h22, hdd));
                }
            }
            MF(14); TR(5);
            // ---- (3) stage gradient entry r and row r of the stage Hessian base  J' Hw J + direct + barrier ;  J = d(s_next, d)/dy: rows 0..2 = rows 0..2 of F, row 3 = e_7
            const double *F = &L.Ft[FT * t];
            const double a0 = Fel(F, 0, r), a1 = Fel(F, 1, r), a2 = Fel(F, 2, r), a3 = r == 7 ? 1.0 : 0.0;
            if (act) {
                const double *ld = &L.cy[5 * t];                  // ld = lam+ - lam- per pair
                double v = a0 * gw0 + a1 * gw1 + a2 * gw2 + (r == 7 ? gw3 : 0.0);
                // + direct control cost + C' lam   (the rate pairs exist for t >= 1; their multipliers are 0 at t = 0).  Every operand is fetched by every lane
                // and the row's term is SELECTED (round 6): as branches on r each of the five cases was a load -> wait -> add round of its own
                const double l0 = ld[0], l1 = ld[1], l2 = ld[2], r0 = ld[3], r1 = ld[4], u0 = L.u[t], u1 = L.u[T + t];
                const double e5 = l0 + r0 + 2 * c.wu * (u0 - vref) + c.eps_u * u0, e6 = l1 + r1 + c.eps_u * u1;
                const double ad = r == 3 ? -r0 : (r == 4 ? -r1 : (r == 5 ? e5 : (r == 6 ? e6 : l2)));
                if (r >= 3) v += ad;
                L.gst[i] = v;
            }
            MF(12); TR(6);
            // (skipped in a pass that is expected to be the convergence check only, see `expect_conv`)
            if (!expect_conv && act) {
                const double *dg = &L.bw[5 * t];
                // v = Hw J[:, r]
                const double v0 = h00 * a0 + h01 * a1 + hd0 * a3, v1 = h01 * a0 + h11 * a1 + hd1 * a3, v2 = h22 * a2, v3 = hd0 * a0 + hd1 * a1 + hdd * a3;
                const double bu0 = dg[0], bu1 = dg[1], bd = dg[2], br0 = dg[3], br1 = dg[4];
                double *row = &L.Hb[HB * t + 8 * r];
                // d_t enters only its own stage (F has no column for it), so it is eliminated HERE, before the recursion: column m7 = H[0..6][7],
                // H' = H - m7 m7' / H77 on the entries 0..6, d decoupled (round 4: the pivot block of the recursion becomes 2 x 2; the step of d is
                // recovered per stage by its inequality pair: dd = -(g7 + m7'y) / H77)
                const double i77 = frcp(hdd + bd);
                const double m7r = r < 7 ? v3 : 0.0;              // (v3 of a row r < 7 = hd0 a0 + hd1 a1)
                L.m7[8 * t + r] = r < 7 ? m7r : i77;
                const double f7 = m7r * i77;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    double m = Fel(F, 0, q) * v0 + Fel(F, 1, q) * v1 + Fel(F, 2, q) * v2;
                    // barrier weights: u0 box, u1 box, d box, rate u0 (rows u0 - up0), rate u1
                    if (r == q) {
                        if (r == 5) m += 2 * c.wu + c.eps_u + bu0 + br0;
                        else if (r == 6) m += c.eps_u + bu1 + br1;
                        else if (r == 3) m += br0;
                        else if (r == 4) m += br1;
                    } else if ((r == 3 && q == 5) || (r == 5 && q == 3)) m -= br0;
                    else if ((r == 4 && q == 6) || (r == 6 && q == 4)) m -= br1;
                    if (q < 7) m -= f7 * (hd0 * Fel(F, 0, q) + hd1 * Fel(F, 1, q));
                    row[q] = (r == 7 || q == 7) ? (r == q ? 1.0 : 0.0) : m;
                }
            }
        }
        TR(7);
        __syncthreads();
        mark(2); TR(8);
        // ---- (4) waves 0 / 1: Riccati matrix recursion of the stages [msp, T) / [0, msp) (time split; msp = 0: wave 0 takes them all);
        //          wave 2: adjoint sweep for the reduced gradient + early verdict; wave 3: the other termination measures ------------
        bool fail = false;
        if (wave == 0) {
            if (!expect_conv) fail = msp > 0 ? !mat_range(MSP, T, Px, Ms) : !mat_range(0, T, Px, Ms);
        } else if (wave == 1) {
            if (!expect_conv && msp > 0) fail = !mat_range(0, MSP, PxA, MsA);
        } else if (wave == 2) {
            // Adjoint sweep p_t = g_x,t + A_t' p_{t+1}, one stage per lane.  A_t = [[1,0,a13],[0,1,a23],[0,0,1]] in all three
            // motion models, so p0 and p1 are suffix sums of g0, g1 and p2 is the suffix sum of g2 + a13 p0' + a23 p1'
            // (' = the value of stage t+1): three wave scans instead of T dependent stages.  Only the termination measure
            // |reduced gradient|_inf is taken from it.
            const bool on = lane < T;
            const double *A = &L.Ak[9 * (on ? lane : 0)], *B = &L.Bk[6 * (on ? lane : 0)], *g = &L.gst[8 * (on ? lane : 0)];
            // (round 6) the suffix sums as total - inclusive prefix + own value, the prefix by DPP moves (row_shr 1, 2, 4, 8, then the row totals across with
            // row_bcast:15 / :31 - the roll-out's scan) and the neighbour's value by wave_shl:1 - rounds 1-5: six __shfl_down = ds_bpermute round trips per scan,
            // three dependent scans per pass, on the critical path of every light pass.  Another summation order: rounding level of a termination measure.
            auto suffix = [&](double v) {
                double q = v;
                q += dpp_f64<0x111>(q); q += dpp_f64<0x112>(q); q += dpp_f64<0x114>(q); q += dpp_f64<0x118>(q);
                q += dpp_f64<0x142, 0xA>(q); q += dpp_f64<0x143, 0xC>(q);
                return v + (bcast(q, 63) - q);
            };
            auto next = [&](double v) { return dpp_f64<0x130>(v); };      // wave_shl:1: lane i takes lane i + 1's value, lane 63 reads 0
            const double s0 = suffix(on ? g[0] : 0.0), s1 = suffix(on ? g[1] : 0.0);
            const double p0 = next(s0), p1 = next(s1);
            const double s2 = suffix(on ? g[2] + A[2] * p0 + A[5] * p1 : 0.0);
            const double p2 = next(s2), p3 = next(on ? g[3] : 0.0), p4 = next(on ? g[4] : 0.0);
            double rd = 0;
            if (on) {
                const double v0 = g[5] + B[0] * p0 + B[2] * p1 + B[4] * p2 + p3;
                const double v1 = g[6] + B[1] * p0 + B[3] * p1 + B[5] * p2 + p4;
                L.gad[3 * lane] = v0; L.gad[3 * lane + 1] = v1; L.gad[3 * lane + 2] = g[7];
                rd = fmax(fabs(v0), fmax(fabs(v1), fabs(g[7])));
            }
            rd = wave_allreduce(rd, true);
            if (lane == 0) L.red[8] = rd;
            // early verdict for wave 0 (see there): wave 3's measures are published under *flag_meas = seq
            while (__atomic_load_n(flag_meas, __ATOMIC_ACQUIRE) != seq) __builtin_amdgcn_s_sleep(1);
            {
                // (plain LDS reads behind the acquire above.  As `*(volatile double *)&L.red[k]` they were FLAT loads with sc0 sc1 - the cast loses the LDS address
                // space - each followed by s_waitcnt vmcnt(0): three trips through the flat path on the critical path of every light pass, found in the ISA, round 6)
                const double gn_ = L.red[9], rpn_ = L.red[10];
                const double mu_ = L.red[11] / mcnt, sc_ = 1 + gn_;
                // (not in a landing round: its rows carry no residual / complementarity, the "measures" of that pass would always pass - and the round NEEDS its
                // factorisation: an abandoned recursion left the stale factors of the last interior-point iteration in place, found with a dense solve of the
                // dumped Newton system, scratch of round 6)
                if (land == 0 && ((rd <= t_rd * sc_ && rpn_ <= t_rp && mu_ <= t_mu * sc_) || (rd <= 100 * t_rd * sc_ && rpn_ <= t_rp && mu_ <= 0.1 * t_mu * sc_)))
                    if (lane == 0) __atomic_store_n(flag_stop, seq, __ATOMIC_RELAXED);
            }
        } else {
            // termination measures that do not depend on the sweeps
            double g = 0, rp_ = 0, m_ = 0, hm_ = 0;
            for (int i = lane; i < T; i += 64) hm_ = fmax(hm_, L.hmx[i]);
            hm_ = wave_allreduce(hm_, true);
            for (int i = lane; i < 4 * T; i += 64) { double v = fabs(L.gw[i]); if (v > g) g = v; }
            for (int i = lane; i < 5 * T; i += 64) { double v = L.ra[i]; if (v > rp_) rp_ = v; m_ += L.lw[i]; }
            g = wave_allreduce(g, true); rp_ = wave_allreduce(rp_, true); m_ = wave_allreduce(m_, false);
            if (lane == 0) {
                L.red[9] = g; L.red[10] = rp_; L.red[11] = m_; L.pv[5] = hm_;
                __atomic_store_n(flag_meas, seq, __ATOMIC_RELEASE);
            }
        }
        TR(9);
        if (fail) __atomic_store_n(flag_fail, seq, __ATOMIC_RELAXED);
        __syncthreads();
        const bool anyfail = __atomic_load_n(flag_fail, __ATOMIC_RELAXED) == seq;
        mark(4); TR(10);
        // The first iteration of an easy-mode warm attempt skips the predictor (a.warm_nopred): next to the solution the affine step
        // is a full step, so sigma ends at its floor anyway and the second-order term dl*dw is O(error^2) - one sweep pair instead of
        // two.  Should that iteration not finish the solve, the following ones are ordinary predictor-corrector iterations.
        const bool nopred = land != 1 && ((attempt < 0 && a.warm_nopred != 0 && it == 0) || safe);      // (a landing round always runs both passes)
        // ---- (4b) closed-loop sweep matrices from W, Minv (all threads; Mb overwrites the consumed Hb,
        //           Mf overwrites the consumed hs..cy) --------------------------------------------------------------
        if (!expect_conv)
        for (int i = tid; i < 8 * T; i += NT) {
            int t = i >> 3, r = i & 7;
            const double *F = &L.Ft[FT * t], *wn = &L.Wn[WN * t];
            double *mb = &L.Hb[HB * t + 6 * r];
            if (r < 5) {
                double *mf = &L.Mf[MF * t + 6 * r];
                const double w0 = wn[3 * r], w1 = wn[3 * r + 1], fr5 = Fel(F, r, 5), fr6 = Fel(F, r, 6);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    mb[j] = Fel(F, j, r) - (Fel(F, j, 5) * w0 + Fel(F, j, 6) * w1);            // Acl[j][r]
                    mf[j] = Fel(F, r, j) - (fr5 * wn[3 * j] + fr6 * wn[3 * j + 1]);              // Acl[r][j]
                }
            } else {
                int k = r - 5;
                double n0 = k == 0 ? wn[15] : (k == 1 ? wn[16] : wn[17]);
                double n1 = k == 0 ? wn[16] : (k == 1 ? wn[18] : wn[19]);
#pragma unroll
                for (int j = 0; j < 5; ++j) mb[j] = -(n0 * Fel(F, j, 5) + n1 * Fel(F, j, 6));
                if (k == 2) {
                    double *mf = &L.Mf[MF * t + 6 * 5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) mf[j] = -wn[3 * j + 2];
                }
            }
            // (round 6) ... and the constant of the PREDICTOR's backward map with its row - build_cb for the first right-hand side used to be a phase of
            // its own on waves 0 / 1 (2.3 k cycles on the traced timeline, beside the unit sweeps of waves 2 / 3)
            if (!nopred) mb[5] = cb_entry(t, r);
        }
        TR(12);
        const double rdn = L.red[8], gn = L.red[9], rpn = L.red[10], mu = L.red[11] / mcnt;
        double sc = 1 + gn;
        if (a.dbg && tid == 0) { a.dbg[4 * it] = rdn; a.dbg[4 * it + 1] = rpn; a.dbg[4 * it + 2] = mu; a.dbg[4 * it + 3] = sc; }
        if (a.rd0 && used == 0 && it == 0 && tid == 0) *a.rd0 = rdn / sc;
        if (used == 0 && it == 0) res.rd0 = rdn / sc;
        // second clause: see the oracle (rounding noise of the dual residual once lam/w reaches 1e10)
        const bool conv_now = land == 0 && ((rdn <= t_rd * sc && rpn <= t_rp && mu <= t_mu * sc) || (rdn <= 100 * t_rd * sc && rpn <= t_rp && mu <= 0.1 * t_mu * sc));
        if (land == 2) {
            // ---- verdict on a landing round (this pass measured x+ with the hinge terms re-evaluated: rdn = stationarity, rpn = largest violation relative to
            //      1 + |e|, mu mcnt = sum of the negative parts of the multipliers; same thresholds as the oracle's su_land)
            LSTAT(2, 1);
            if (!spec) res.land_rounds += 1;
            res.rounds_all += 1;
            const bool moved = rpn > 1e-11 || mu * mcnt > 1e-9 * sc;
            if (!moved && rdn <= 100 * c.tol_rd * sc) {                      // landed
#pragma unroll
                for (int j = 0; j < NPR; ++j) if (p_on[j]) { Plp[j] = fmax(Plp[j], 0.0); Plm[j] = fmax(Plm[j], 0.0); }
                LSTAT(0, 1); if (spec) { LSTAT(5, 1); if (!blind) LSTAT(12 + spec_dec, 1); }
                if (spec) res.spec = 1;
                if (blind) { res.blind = 1; LSTAT(19, 1); }
                res.land_rho = land_rho;
                status = 0; break;
            }
            if (land_rounds >= (blind ? 1 : (spec ? 2 : 4)) || !(rdn == rdn)) { land_refuse(land_last); continue; }      // refused
            // next round: rows move in / out of the active set by their signs (primal-dual active set) and the model is solved again from the SAME point -
            // a full step along a weakly curved direction may have left the boxes by far, x+ is then no place to linearise the hinge terms at; a round whose
            // set did not move was not stationary because a hinge term switched: that one is linearised again at x+ (= the oracle's su_land)
#pragma unroll
            for (int j = 0; j < NPR; ++j)
                if (p_on[j]) {
                    if (Lap[j] && Plp[j] < -1e-9 * sc) { Lap[j] = false; Plp[j] = 0.0; } else if (!Lap[j] && Pwp[j] < -1e-11 * (1.0 + fabs(p_ep[j]))) { Lap[j] = true; Plp[j] = 0.0; }
                    if (Lam[j] && Plm[j] < -1e-9 * sc) { Lam[j] = false; Plm[j] = 0.0; } else if (!Lam[j] && Pwm[j] < -1e-11 * (1.0 + fabs(p_em[j]))) { Lam[j] = true; Plm[j] = 0.0; }
                }
            land = 1; land_rounds += 1; expect_conv = false;
            __syncthreads();
            if (moved) {
                for (int i = tid; i < 2 * T; i += NT) L.u[i] = L.base[i];
                for (int i = tid; i < T; i += NT) L.d[i] = L.base[2 * T + i];
                __syncthreads();
                rollout();
                __syncthreads();
            } else {
                for (int i = tid; i < 2 * T; i += NT) L.base[i] = L.u[i];
                for (int i = tid; i < T; i += NT) L.base[2 * T + i] = L.d[i];
            }
            land_rows(land_rho);
            __syncthreads();
            continue;
        }
        // (may a remembered iterate have ignored hinge terms outside the near list?  No: the measures tested here were formed by THIS pass's stage
        // phase, which sums the near list only while `screened` holds - and `screened` is dropped by the reach check right behind every update (and by the
        // set-up for the nominal) as soon as a stage has left the DELTA ball; from then on every pass sums every term.  So an iterate that passes this test
        // was measured on a term set that contains every term that can be active at it - the same argument as for the converged iterate.)
        if (a.accept && land == 0 && !conv_now && rpn <= c.tol_rp && rdn <= 10 * c.tol_rd * sc && mu <= 1e3 * c.tol_mu * sc) {     // (uniform) safety net: see Args::accept
            const double merit = fmax(rdn / (c.tol_rd * sc), mu / (c.tol_mu * sc));
            if (!have_acc || merit < acc_merit) {
                have_acc = true; acc_merit = merit;
                for (int i = tid; i < 2 * T; i += NT) L.acc[i] = L.u[i];
                for (int i = tid; i < T; i += NT) L.acc[2 * T + i] = L.d[i];
#pragma unroll
                for (int j = 0; j < NPR; ++j)
                    if (p_ok[j]) { const int o = 3 * T + 10 * p_t[j] + 2 * p_k[j]; L.acc[o] = p_on[j] ? Plp[j] : 0.0; L.acc[o + 1] = p_on[j] ? Plm[j] : 0.0; }
            }
        }
        // landing first, mode 2: the light first pass of a warm attempt did not meet the stop - land from the start all the same (see Args::land_first)
        const bool spec_now = !conv_now && expect_conv && land == 0 && landing && a.land_first == 2 && attempt < 0 && it == 0 && !spec_tried && rdn < a.land_first_rd0 * sc && rpn == rpn;
        if (conv_now || spec_now) {
            // (rounds 1-5 re-checked here, with a block reduction of its own, that the positions lie within DELTA of the screening reference: the check behind every
            // update - and the set-up's, at half the radius - has already taken that verdict for exactly these positions: they do not move between an update
            // and the stop test of the next pass, and a refused landing goes back to a checked iterate.  2 k cycles per solve, never taken.  Round 6.)
            if (landing) {
                // ---- start of the landing: remember the iterate, read the active set off it (lam > w), first round
                for (int i = tid; i < 2 * T; i += NT) { L.sav[i] = L.u[i]; L.base[i] = L.u[i]; }
                for (int i = tid; i < T; i += NT) { L.sav[2 * T + i] = L.d[i]; L.base[2 * T + i] = L.d[i]; }
#pragma unroll
                for (int j = 0; j < NPR; ++j) {
                    Swp[j] = Pwp[j]; Swm[j] = Pwm[j]; Slp[j] = Plp[j]; Slm[j] = Plm[j];
                    // (a speculative landing reads the set off the KEPT multipliers: the floors of the warm start make every row within sqrt(mu0) of its bound look active)
                    Lap[j] = p_on[j] && (spec_now ? pf_lkp[j] : Plp[j]) > Pwp[j]; Lam[j] = p_on[j] && (spec_now ? pf_lkm[j] : Plm[j]) > Pwm[j];
                }
                land_rho = a.land_rho * fmax(1.0, fmax(L.pv[5], 2 * c.wu + c.eps_u));
                land = 1; land_rounds = 1; expect_conv = false; spec = spec_now;
                if (spec_now) {
                    const double q = rdn / sc;
                    spec_dec = q < 1e-4 ? 0 : (q < 1e-3 ? 1 : (q < 1e-2 ? 2 : (q < 1e-1 ? 3 : (q < 1.0 ? 4 : 5))));
                    LSTAT(4, 1); LSTAT(6 + spec_dec, 1);
                }
                TR(13);
                __syncthreads();
                land_rows(land_rho);
                TR(14);
                __syncthreads();
                continue;
            }
            status = 0; break;
        }
        if (expect_conv) {         // the prediction was wrong: repeat this pass with the factorisation (same iteration number)
            expect_conv = false; __syncthreads(); --it; continue;
        }
        mu_prev = mu;              // (after the repeat decision: the repeated pass smooths with the same width as the light one)
        if (anyfail && land == 1) { land_refuse(land_last); continue; }       // (a landing whose frozen set made the factorisation break down)
        if (anyfail) { status = 2; break; }
        mark(5);

        double sigma = 0;
        if (nopred) sigma = safe ? (al_prev >= 0.9 ? SU_SAFE_SIGMA_END : SU_SAFE_SIGMA) : a.warm_sig;
        bool unit_done = false;                    // time split: the unit sweeps / interface matrix of this factorisation exist
        TR(11);
        if (msp > 0) __syncthreads();              // (the unit sweeps of waves 2 / 3 read the closed-loop rows all threads have just written)
        for (int pass = nopred ? 1 : 0; pass < 2; ++pass) {
            TR(20 + 10 * pass);
            if (pass == 1) {
                // corrector right-hand side: targets lam w + dlam dw - sigma mu (no second-order term without a predictor), new x+ - x-
                const double smu = sigma * mu;
                if (land == 1) {
                    // second step of the method of multipliers: nu_1 = nu_0 + rho r(x + dx_0) (active rows; r(x + dx_0) is in Pdwp / Pdwm), the gradient was
                    // formed with nu_0, so the right-hand side gets rho (r(x) + r(x + dx_0)) through x+ - x-
#pragma unroll
                    for (int j = 0; j < NPR; ++j)
                        if (p_on[j])
                            L.xd[5 * p_t[j] + p_k[j]] = land_rho * ((Lap[j] ? Prpp[j] + Pdwp[j] : 0.0) - (Lam[j] ? Prpm[j] + Pdwm[j] : 0.0));
                } else
#pragma unroll
                for (int j = 0; j < NPR; ++j)
                    if (p_on[j]) {
                        Prcp[j] = Plp[j] * Pwp[j] + (nopred ? 0.0 : Pdlp[j] * Pdwp[j]) - smu;
                        Prcm[j] = Plm[j] * Pwm[j] + (nopred ? 0.0 : Pdlm[j] * Pdwm[j]) - smu;
                        L.xd[5 * p_t[j] + p_k[j]] = (Plp[j] * Prpp[j] - Prcp[j]) * Piwp[j] - (Plm[j] * Prpm[j] - Prcm[j]) * Piwm[j];
                    }
                __syncthreads();
                MF(9); TR(31);
            }
            if (MSP == 0 || msp == 0) {
                if (pass == 1) build_cb();                  // (the predictor's constants were formed with the closed-loop rows, 4b)
                __syncthreads();
                mark(6);
                if (wave == 0) {
                    if (lane < 8) bwd_seg(0, T, 0.0, L.kk, true, lane);
                    wsync(); cf_seg(0, T); wsync();
                    if (lane < 8) { const double xe = fwd_seg(0, T, 0.0); if (lane < 3) L.pv[lane] = xe; }
                }
            } else {
                // TIME SPLIT.  Segment B = stages [msp, T) was factorised from P_T = 0 (wave 0), segment A = [0, msp) from P = 0 (wave 1): A's gains are
                // those of a horizon that ends at msp with a LINEAR terminal term pi' x_msp.  With pi = P_msp x_msp + p_msp (the gradient of B's
                // cost-to-go at the true x_msp) both halves solve the same Newton system as one recursion over [0, T) - the first-order conditions
                // are identical.  x_msp responds linearly to pi: x_msp = x0 + X pi, where x0 is A's own answer (pi = 0) and X[j][i] = sum_t b(j)_t . kk(i)_t
                // comes from five UNIT backward sweeps of A (p_msp = e_i, no constants: kk(i)_t; b(j)_t = F_v' p(j)_{t+1}) - so
                //     (I - X P_msp) x_msp = x0 + X p_msp ,   pi = P_msp x_msp + p_msp ,   kk_t += sum_i pi_i kk(i)_t  (t < msp)
                // and the two forward sweeps start from 0 and x_msp.  Unit sweeps, X, S^-1 and S^-1 X depend on the factorisation only: once per
                // interior-point iteration, by waves 2 / 3 beside the sweep constants and the first backward sweeps.
                // tools/experiments/two_segment.py: the algebra in numpy, on random and on recorded systems.
                // Predictor (pass 0): its constants were formed with the closed-loop rows (4b), so the backward sweeps of waves 0 / 1 start at once; waves 2 / 3
                // run the unit sweeps beside them and wave 2 goes straight on to the interface matrix once wave 3's sweep is in LDS (flag_unit, no block
                // barrier).  A corrector that is the FIRST pass of its factorisation (no predictor): constants on waves 0 / 1 beside the unit sweeps, barrier,
                // backward sweeps beside the interface matrix - as in rounds 4-5.
                if (pass == 1) {
                    if (!unit_done) { if (wave < 2) build_cb(tid, NT / 2); else unit_sweeps(); }
                    else build_cb(tid, NT);
                    TR(22 + 10 * pass);
                    __syncthreads();
                    mark(6); TR(23 + 10 * pass);
                }
                if (wave == 0) {
                    double pe = 0;
                    if (lane < 8) pe = bwd_seg(MSP, T, 0.0, L.kk, true, lane);
                    if (lane < 5) XS_pm[lane] = pe;
                } else if (wave == 1) {
                    if (lane < 8) bwd_seg(0, MSP, 0.0, L.kk, true, lane);
                    if (pass == 0) {                       // (x0 reads the b(j)_t of the unit sweeps: both waves' must be in LDS)
                        while (__atomic_load_n(flag_unit, __ATOMIC_ACQUIRE) != seq) __builtin_amdgcn_s_sleep(1);
                        while (__atomic_load_n(flag_unit2, __ATOMIC_ACQUIRE) != seq) __builtin_amdgcn_s_sleep(1);
                    }
                    wsync(); interface_x0();
                } else if (pass == 0) {
                    unit_sweeps();
                    if (lane == 0) __atomic_store_n(wave == 3 ? flag_unit : flag_unit2, seq, __ATOMIC_RELEASE);
                    if (wave == 2) {
                        while (__atomic_load_n(flag_unit, __ATOMIC_ACQUIRE) != seq) __builtin_amdgcn_s_sleep(1);
                        interface_matrix();
                    }
                } else if (wave == 2 && !unit_done) interface_matrix();
                unit_done = true;
                TR(24 + 10 * pass);
                __syncthreads();
                MF(3); TR(25 + 10 * pass);
                if (wave < 2) {
                    double xm, pi;
                    interface_solve(xm, pi);
                    if (wave == 0) {
                        cf_seg(MSP, T); wsync();          // (the forward constants of B here rather than behind the backward sweep: this wave has the shorter prelude)
                        if (lane < 8) { const double xe = fwd_seg(MSP, T, lane < 5 ? xm : 0.0); if (lane < 3) L.pv[lane] = xe; }
                    } else {
                        double pv5[5];
#pragma unroll
                        for (int i = 0; i < 5; ++i) pv5[i] = bcast(pi, i);
                        for (int e = lane; e < 3 * MSP; e += 64) {
                            const int t = e / 3, k2 = 5 + e % 3;
                            double v = L.kk[8 * t + k2];
#pragma unroll
                            for (int i = 0; i < 5; ++i) v += pv5[i] * L.uk[8 * MSP * i + 8 * t + k2];
                            L.kk[8 * t + k2] = v;
                        }
                        wsync(); cf_seg(0, MSP); wsync();
                        if (lane < 8) fwd_seg(0, MSP, 0.0);
                    }
                }
            }
            TR(26 + 10 * pass);
            __syncthreads();
            mark(7); TR(27 + 10 * pass);
            // ---- slack / multiplier steps (pair threads, registers), step length ---------------------------------------
            // step to the boundary: the largest ratio -dx / x over all slacks and multipliers; al = min(1, fr / ratio)
            double ratio = 0.0;
            if (land == 1) {
                // landing: the rows at x + dx of this pass, r+ + c'dx and r- - c'dx; pass 0 keeps them for the second right-hand side (Pdwp / Pdwm),
                // pass 1 ends the round: nu_2 = nu_0 + rho (r(x + dx_0) + r(x + dx_1)) on the active rows, slacks e - c'x+ of all rows
#pragma unroll
                for (int j = 0; j < NPR; ++j)
                    if (p_on[j]) {
                        const double cdx = pair_step(p_t[j], p_k[j]);
                        const double rp_ = Prpp[j] + cdx, rm_ = Prpm[j] - cdx;
                        if (pass == 0) { Pdwp[j] = rp_; Pdwm[j] = rm_; }
                        else {
                            if (Lap[j]) Plp[j] += land_rho * (Pdwp[j] + rp_);
                            if (Lam[j]) Plm[j] += land_rho * (Pdwm[j] + rm_);
                            Pwp[j] = -rp_; Pwm[j] = -rm_;
                        }
                    }
            } else
#pragma unroll
            for (int j = 0; j < NPR; ++j)
                if (p_on[j]) {
                    const double cdx = pair_step(p_t[j], p_k[j]);
                    const double dwp = -Prpp[j] - cdx, dwm = -Prpm[j] + cdx;
                    const double dlp = -(Prcp[j] + Plp[j] * dwp) * Piwp[j], dlm = -(Prcm[j] + Plm[j] * dwm) * Piwm[j];
                    Pdwp[j] = dwp; Pdwm[j] = dwm; Pdlp[j] = dlp; Pdlm[j] = dlm;
                    ratio = fmax(ratio, fmax(fmax(-dwp * Piwp[j], -dwm * Piwm[j]), fmax(-dlp * Pilp[j], -dlm * Pilm[j])));
                }
            MF(15); TR(28 + 10 * pass);
            ratio = block_reduce1(ratio, L.red + 272 + 4 * pass, tid, true);
            MF(0); TR(29 + 10 * pass);
            // fraction to the boundary: 1 for the predictor; corrector: 0.995 far from the solution, -> 1 with the complementarity (superlinear end game)
            double fr = 1.0;
            if (pass) { fr = 1.0 - mu; if (fr < tau_min) fr = tau_min; if (safe) fr = SU_SAFE_TAU; }
            const double al = land == 1 ? 1.0 : (ratio > fr ? fr / ratio : 1.0);      // (a landing takes the full step of its model)
            if (pass == 0) {
                // centering parameter from the predictor step length, floored (see the oracle for why)
                double q = 1 - al, fl = al >= 0.95 ? (attempt < 0 ? a.warm_sig : SIGMA_FLOOR) : 0.03;
                if (it >= 25) fl = it >= 50 ? 0.3 : 0.1;      /* a solve that is still running is cycling: centre harder */
                sigma = q * q * q; if (sigma < fl) sigma = fl;
            } else {
#pragma unroll
                for (int j = 0; j < NPR; ++j)
                    if (p_on[j] && land != 1) { Pwp[j] += al * Pdwp[j]; Pwm[j] += al * Pdwm[j]; Plp[j] += al * Pdlp[j]; Plm[j] += al * Pdlm[j]; }
                if (tid < T) {
                    int t = tid; const double *y = &L.dy[8 * t], *v = &L.vv[8 * t + 3];
                    L.u[t] += al * v[0]; L.u[T + t] += al * v[1]; L.d[t] += al * v[2];
                    if (t >= 1) for (int r = 0; r < 3; ++r) L.s[r * (T + 1) + t] += al * y[r];
                }
                if (tid < 3) L.s[tid * (T + 1) + T] += al * L.pv[tid];
                TR(40);
                __syncthreads();
                MF(11); TR(41);
                // ONE block reduction for (i) the reach of the hinge screening - it holds only while every stage stays within DELTA of its
                // reference position - and (ii) the mean complementarity after the step (light convergence pass, recentring); slots of their own (see
                // block_reduce1), and between its two barriers the pair threads prepare the rows of the NEXT iteration (pair_rows)
                double dv = 0, m_ = 0;
                if (screened && tid < T) { double ex = L.s[tid + 1] - L.p0[tid], ey = L.s[(T + 1) + tid + 1] - L.p0[T + tid]; dv = sqrt(ex * ex + ey * ey); }
#pragma unroll
                for (int j = 0; j < NPR; ++j) if (p_on[j]) m_ += Plp[j] * Pwp[j] + Plm[j] * Pwm[j];
                dv = wave_allreduce(dv, true); m_ = wave_allreduce(m_, false);
                if (lane == 0) { L.red[280 + wave] = dv; L.red[284 + wave] = m_; }
                if (land == 1) verify_rows(); else pair_rows();
                TR(42);
                __syncthreads();
                TR(43);
                dv = fmax(fmax(L.red[280], L.red[281]), fmax(L.red[282], L.red[283]));
                m_ = ((L.red[284] + L.red[285]) + (L.red[286] + L.red[287])) / mcnt;
                if (screened && dv > DELTA) screened = false;      // from the next iteration on: every term (the streaming loop)
                MF(13);
                // Light convergence pass: the residuals of a Newton step of length al shrink by (1 - al) (the dynamics are
                // eliminated exactly, the constraints are affine) and the new complementarity is known now.  When these predict
                // that the stop test will hold, the next pass evaluates the TRUE measures only - no Hessian bases, no Riccati
                // recursion, no sweep matrices - and is repeated in full should the test fail after all.
                // A cold attempt that is still running after SU_CENTRE_FROM iterations is cycling: from there on every pair is kept inside a
                // (very) wide neighbourhood of the central path, lam w >= SU_CENTRE_GAMMA mu after the step, by raising the multiplier
                // (same rule and reason as the oracle's su_solve_impl: two neighbouring rate rows traded places for ever,
                // tests/golden/su_hard/acker_T15_N45_rate_rows_cycle.npz).
                const bool recentre = land == 0 && ((attempt >= 0 && it >= SU_CENTRE_FROM) || safe);
                al_prev = al;
                if (recentre) {
                    const double floor_ = (safe ? SU_SAFE_GAMMA : SU_CENTRE_GAMMA) * m_;
#pragma unroll
                    for (int j = 0; j < NPR; ++j)
                        if (p_on[j]) {
                            if (Plp[j] * Pwp[j] < floor_) Plp[j] = floor_ / Pwp[j];
                            if (Plm[j] * Pwm[j] < floor_) Plm[j] = floor_ / Pwm[j];
                        }
                    pair_rows();                                   // (rare path: the rows of the next iteration once more, with the raised multipliers)
                    __syncthreads();
                }
                const double prd = (1 - al) * rdn, prp = (1 - al) * rpn;
                expect_conv = c.light_check && ((prd <= t_rd * sc && prp <= t_rp && m_ <= t_mu * sc) ||
                                                (prd <= 100 * t_rd * sc && prp <= t_rp && m_ <= 0.1 * t_mu * sc));
                if (land == 1) { land = 2; expect_conv = true; }      // (the next pass verifies the landing: true measures only)
            }
            mark(8);
        }
    }
    used += it - land_its > 0 ? it - land_its : 0;      // (an accepted BLIND landing has no measuring pass in front of it: two landing passes, the second one breaks at it = 1)
    LSTAT(3, land_its);
    }
    __syncthreads();
    if ((status != 0 || a.accept == 2) && have_acc) {          // every attempt failed: the safety net (the remembered iterate is primal feasible: inside the boxes).  accept == 2: test switch -
                                                                // ALWAYS hand the remembered iterate back (tests/test_gpu_parity.py: this path is otherwise never taken)
        for (int i = tid; i < 2 * T; i += NT) L.u[i] = L.acc[i];
        for (int i = tid; i < T; i += NT) L.d[i] = L.acc[2 * T + i];
#pragma unroll
        for (int j = 0; j < NPR; ++j)
            if (p_ok[j]) { const int o = 3 * T + 10 * p_t[j] + 2 * p_k[j]; Plp[j] = L.acc[o]; Plm[j] = L.acc[o + 1]; }
        status = 0;
        __syncthreads();
    }
    // consistent final rollout (removes accumulated rounding in s)
    TR(102);
    rollout();
    __syncthreads();
    mark(15); TR(103);
    if (status == 0 && a.lam_keep) {
#pragma unroll
        for (int j = 0; j < NPR; ++j)
            if (p_ok[j]) { const int o = p_t[j] * NC + 2 * p_k[j]; a.lam_keep[o] = p_on[j] ? Plp[j] : 0.0; a.lam_keep[o + 1] = p_on[j] ? Plm[j] : 0.0; }
    }
    if (status == 0) {       // otherwise keep the nominal (reference :696-700)
        for (int i = tid; i < 3 * (T + 1); i += NT) a.out_s[i] = L.s[i];
        for (int i = tid; i < 2 * T; i += NT) a.out_u[i] = L.u[i];
        for (int i = tid; i < T; i += NT) a.out_d[i] = L.d[i];
    } else if (a.out_s != a.in_s) {
        for (int i = tid; i < 3 * (T + 1); i += NT) a.out_s[i] = a.in_s[i];
        for (int i = tid; i < 2 * T; i += NT) a.out_u[i] = a.in_u[i];
        if (a.d_in) for (int i = tid; i < T; i += NT) a.out_d[i] = a.d_in[i];
    }
    // pose table of the trajectory handed back (the solution, or the nominal when the solve failed): position of column t+1,
    // cos / sin of the heading of column t (quirk Q1) - what every LamMuZ row and the next su set-up would otherwise recompute
    if (a.pose_out && tid < T) {
        const bool okk = status == 0;
        const double px = okk ? L.s[tid + 1] : a.in_s[tid + 1], py = okk ? L.s[(T + 1) + tid + 1] : a.in_s[(T + 1) + tid + 1];
        const double ph = okk ? L.s[2 * (T + 1) + tid] : a.in_s[2 * (T + 1) + tid];
        double sp, cp; sincos(ph, &sp, &cp);
        a.pose_out[4 * tid] = px; a.pose_out[4 * tid + 1] = py; a.pose_out[4 * tid + 2] = cp; a.pose_out[4 * tid + 3] = sp;
    }
    if (tid == 0) { *a.status = status; *a.ipm_iters = used; }
    res.status = status; res.iters = used;
    mark(10); TR(104);
    if (prof_on && tid == 0) for (int k = 0; k < 16; ++k) a.prof[k] += pacc[k];
    return true;
}
#undef RW
#undef LSTAT
#undef R5
#undef MS
#undef MF
#undef LDS_DRAIN
#undef TR

}  // namespace su

#pragma clang fp contract(fast)
