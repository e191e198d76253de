// librda_hip.so - MI355X (gfx950) implementation of the C-ABI in include/rda_hip.h.
//
// Per ADMM iteration two launches on the handle's stream, no host synchronisation in between:
//   k_su      <<<1, 256>>>      su-problem (interior point + Riccati), relinearisation point update,
//                               residual reduction / early-stop flag of the previous iteration
//   k_lammuz_rows <<<N*T/16, 256>>>  one (obstacle, stage) LamMuZ sub-problem per 16-lane row of a wavefront (k_lammuz
//                               <<<N*T/4, 256>>>: one per wavefront, for E+R+1 > 16), fused with
//                               the lam'A / lam'b products, the xi / zeta updates and the residual
//                               partials (reference rda_solver.py:529-542, 639-690, 781-793)
// Dense grids (more than 256 workgroups per ego, fleets) run the LamMuZ step as two launches: k_lammuz_rows_fast (the common path
// only, three waves per SIMD) and k_lammuz_enum (the rows whose warm candidate failed its certificate).
// In front of them, optional: scene::k_* (the caller's obstacle conversion / ordering) and the caller's pre_process, which
// shares a launch with the first su-problem of the tick (k_su_tracked: two workgroups); rda_fleet_* launches the same bodies
// once for B egos.  The launch that ends a step (the su launch that detects the early stop, else k_finish) writes the result
// slot into pinned host memory and publishes a sequence word the host polls.
// All solver state (duals, products, nominal trajectory, staged obstacles) stays resident in HBM
// between iterations and between MPC steps, exactly like the reference keeps it in CVXPY Parameter
// values (quirks Q4-Q6 come for free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <dlfcn.h>
#include "../../include/rda_hip.h"
#include "lammuz_device.h"
#include "lammuz_cp_device.h"
#include "lammuz_ip_device.h"
#include "su_device.h"
#include "scene_device.h"
#include "track_device.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "librda_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return RDA_ERR_HIP; } } while (0)

struct Ctrl {
    int stop, iters, su_status, ipm_iters, st_tmp, it_tmp;
    int su_last;             // interior-point iterations of the last su-solve of this handle (99 = none / it did not converge): picks the next start
                             // (a history per ADMM iteration index was measured: no gain on dynamic_obs, and it leaves the oracle's mirrored path)
    int lmz_fail;            // sub-problems of this step that kept their previous duals (non-finite input or result), rda_solver.py:791-793
    double resi_dual, resi_pri;
    int finished;                 // the result slot of this step has been written (by the launch that ended the step)
    int su_probe;                 // consecutive su-solves in the hard regime (see su_body)
    int hint_par;                 // which of the two Dev::hint buffers the LamMuZ launch of this iteration WRITES (flipped by every executed su launch)
    int wl_count;                 // entries of Dev::wl written by the common-path LamMuZ kernel of this iteration (reset by k_su)
    int wlc_count;                // ... and its CIRCLE rows, which fill Dev::wl from the back (round 6: the work-list kernel serves them four per wave, remembered case first)
    int resi_iter;                // ADMM iterations of this step whose residuals are in resi_dual / resi_pri (k_su / k_finish)
    int pose_ok;                  // Dev::pose and the near masks of Dev::coef describe the same terms (a LamMuZ launch / k_lmz_finalize made them)
    int prev_unconv;              // the previous step ended with a residual above iter_threshold (all iter_num iterations, no early stop): picks the warm start (su_hard_warm)
    int su_hardlike;              // the last su-solve started far from its solution (relative dual residual of its first iterate > su::HARD_RD0): the other key of su_hard_warm
    double rd0_tmp;               // ... that residual, written by the solve
    int land_easy;                // consecutive su-solves of this handle that needed no interior-point iteration and ONE landing round: the next warm su-problem of the same step starts with a blind landing (su_body): solver history
    double land_rho_prev;         // penalty scale of the last landing (what a blind landing uses)
    int blind_ok;                 // blind landings are tried while this is >= 0: +1 per accepted one (capped at 4), -6 after a refused one, +1 per opportunity skipped (N = 2000: 7 of 37 accepted, north star 54 of 62): solver history
    int land_hard;                // > 0: one of the last four su-solves' landings took three or more rounds (many rows / hinge terms still undecided at the 1e-3-class stop: moving obstacles): the next solve's interior point runs to 1e-2 x su_land_tol before it is landed (C4: 2.7 -> rounds per solve, +4 % steps/s; north star: 1.1 - 1.3 rounds, unaffected): solver history
    int spec_credit;              // su_land_first = 2: speculative landings are tried while this is >= 0 (+3 per accepted one, capped at 6; -2 per refused one; +1 per eligible solve that had to skip): solver history
    int land_stat[su::LAND_STATS];  // su_land: landings accepted, refused, rounds, passes spent on landings, speculative landings (rda_debug_su_land, rda_debug_su_land_n)
    unsigned long long ref_seq;   // tick number whose reference is complete (k_su_tracked: written by the sampling workgroup)
#ifdef RDA_LMZ_STATS
    unsigned lmz_stat[8];    // debug build only: [0] executed launches, [1] rows that needed the enumeration, [2+k] waves with k such rows
#endif
};

constexpr int NCOEF = 9;      // per-(slot, stage) term arrays: NGATH of them travel with the shard chunk (Dev::coef), the rest stay local (Dev::coefL)
constexpr int NGATH = 3;      // ax, ay, cb: what the su hinge reads per term
constexpr int NBS = su::NBS;  // doubles per (stage, GS-slot block) partial; one more 8-byte word per block carries the near mask
constexpr int GS = su::GS;    // slots per block partial = rows of a packed LamMuZ workgroup (2 waves x 4 rows of 16 lanes)
struct Dev;
__host__ __device__ inline double *coef_arr(const Dev &d, int r, int k);

struct Dev {
    rda_cfg c;
    int nt;                  // time slots of the staged obstacles (T+1 or 1)
    int warm;                // k_lammuz tries the previous support first (RDA_LMZ_WARM=0 disables)
    int rows;                // four sub-problems per wave (k_lammuz_rows) when E+R+1 <= 16 (RDA_LMZ_ROWS=0 disables)
    unsigned char muc[40]; int nmv;   // robot support candidates that survive the vertex test (host, rda_create)
    double rv[28][2]; int nrv;        // robot vertices of the surviving pairs (list order)
    double *su_lam_keep;               // inequality multipliers of the last converged su-solve [10*T] (interior-point warm start)
    int su_warm_first;                 // the first su-problem of a step starts from the previous step's multipliers, shifted by one stage
    int su_warm_cap;                   // iterations granted to the warm start before the cold one takes over
    double su_tol[3];                  // interior-point stop of the su-problem (rda_opts::su_tol)
    double su_tol_early[3];            // ... of the ADMM iterations before the last one of a step (rda_opts::su_tol_early; 0 = su_tol)
    int su_light;                        // su_device Cfg::light_check
    int su_split;                        // su_device Args::split (time split of the Newton system)
    int su_accept;                       // su_device Args::accept (safety net: the best near-converged iterate)
    int su_first_attempt;                // su_device Args::first_attempt (test switch)
    int su_land; double su_land_tol[3], su_land_rho;  // su_device Args::land (rda_opts::su_land)
    int su_land_first;                   // su_device Args::land_first (rda_opts::su_land_first)
    int su_land_blind_from;              // blind landings after this many easy solves in a row (0: never)
    int *wl; int wl_cap;                 // [wl_cap >= N*T] work list: sub-problems whose warm candidate failed its certificate (split LamMuZ launch): polygon rows from the front, circle rows from the back
    int *sc_bad;                         // non-convex counter of the staged raw scene (null: obstacles were staged as (A, b) slots)
    int su_easy_nopred;
    int su_pre;                        // the su set-up reads the block sums / near masks of the LamMuZ launch (0: it evaluates every term itself)
    int su_cold_probe;
    int su_cold_from;                  // a solve that follows one with more interior-point iterations than this starts cold (0 = never)
    double su_easy[5]; int su_easy_max;  // wfl, mu0, clip, tau, sigma of the start used while the su-solves are EASY (the last one took <= su_easy_max
                                       // interior-point iterations; max = 0 disables)
    double su_warm_clip;               // start of a warm attempt: relative margin inside the boxes (cold 0.01)
    double su_warm_tau, su_warm_sig;   // end game of the warm attempt (floors; cold solves: 0.995, 1e-3)
    double su_warm_wfl, su_warm_mu0;   // interior-point start of the su-problems of ADMM iterations >= 1 ("0,0" = cold)
    double su_hard_wfl, su_hard_mu0;   // ... of the steps that follow an UNCONVERGED step (rda_opts::su_hard_warm; mu0 = 0: the rule is off)
    int lmz_mode;            // 0: support enumeration + tie-breaks T1-T3 (default), 1: interior point, central path at lmz_mu (norm2 robots: always)
    double lmz_mu;           // barrier parameter of the returned central-path point (mode 1)
    int centre;              // tie-break T1: central separating normal in the slack regime
    int obstacle_num;        // 0 or N
    double *G, *h;
    double *A, *b; int *cone;                 // [N][nt][E][2], [N][nt][E], [N]
    // per (slot, time slot) candidate list and vertices of the staged obstacle (pose independent): k_prepare, at upload
    unsigned char *oc_lamc; double *oc_vtx; int *oc_cnt;      // [N*nt][oc_ls], [N*nt][oc_vs], [N*nt][2] = (npv, nlv)
    int oc_ls, oc_vs;        // strides of the two tables: 1 + E + E (E-1)/2 candidates (rounded up to 4), E (E-1) vertex coordinates - NOT the 40 / 56 of E = 8 (round 6: with moving
                             // obstacles the tables are per (slot, STAGE): 496 B a row against 96 B of half-spaces; E = 4: 128 B)
    // Remembered supports (candidate index of the last max-clearance optimum, -1 = none) - a pure cache: whatever it holds is only the
    // first candidate of a row, which is accepted on its certificate alone.  Two buffers of hint_len ints, [T][hint_stride] each: a
    // LamMuZ launch reads the one the previous executed launch wrote and writes the other (Ctrl::hint_par, flipped by the su launch of
    // the iteration), so that the FIRST launch of a tick can read stage t+1 of the previous tick - the horizon has moved on by one
    // stage, that support belongs to this pose - without racing against the row that rewrites it.  Key of a row: the SOURCE obstacle
    // of its slot when the scene was staged by the device pipeline (slot_src = its rank table: a re-sorted scene re-binds most slots
    // every tick, the supports stay with their obstacles), else the slot itself (host-staged slots, padding copies).
    int *hint; int hint_len, hint_stride, src_cap;
    const int *slot_src; int src_used;        // slot -> index in the caller's raw scene, valid for slots < src_used (null: host-staged)
    // Dual state, STAGE-MAJOR: lam [T+1][N][E], mu [T+1][N][R], xi [T+1][N][2], z / zeta [T][N] - a LamMuZ workgroup owns GS consecutive
    // slots of one stage, so what it reads and writes are whole lines (the accessors rda_get_state / rda_set_state speak the
    // reference's [N][T+1][.] shapes)
    double *lam, *mu, *z, *xi, *zeta, *dis;
    // Condensed su terms + residual partials per obstacle shard (P = 1 on a single GPU), arrays [T][Nloc] indexed t*Nloc + nl:
    //   k = 0 ax, 1 ay (a = A'lam), 8 cb = lam'b + mu'h + z - zeta (the offset of the su hinge)      -> the shard CHUNK, coef[r*chunk + ..]
    //   k = 2 lam'b, 3 mu'h + z - zeta, 4 / 5 G'mu + xi, 6 dual residual, 7 |Hm|^2                    -> local only, coefL[r*lchunk + ..]
    //   bsum : coef[r*chunk + NGATH*T*Nloc + ((t*J + j)*NBS + q)]   sums over the slots GS j .. GS j + GS-1 of stage t (su::row_term; [6], [7])
    //   bmask: the 8-byte words behind bsum, [t*J + j]: bit r = slot GS j + r is NEAR (su hinge screening) at the pose table's position
    // chunk r (what the su-problem reads: three arrays + the reduced sums and masks) is produced by rank r's LamMuZ launch and replicated
    // by one all-gather per ADMM iteration; the local arrays serve the state accessors, failed rows and k_lmz_finalize of the own rank.
    double *coef, *coefL; int P, rank, Nloc, J; size_t chunk, lchunk;
    int Nlive;                                // obstacle slots of THIS rank's shard that exist (< Nloc on the last ranks when N % P != 0)
    double *s, *u;                            // nominal (para_s, para_u)
    double *pose;                             // [T][4] px, py (column t+1), cos, sin (heading of column t) of Dev::s: written by every su launch,
                                              // read by the LamMuZ rows (no trigonometry per row) and by the next su set-up
    double *ref, *ref_speed;                  // current step reference (device)
    Ctrl *ctrl;
    long long *su_prof;                       // optional phase cycle counters of the su-solves of this handle (RDA_SU_PROF), else null
    // interior-point LamMuZ mode, row-parallel kernel: the central-path point (x | s diag | z diag | s general | z general, 5 x 16 doubles)
    // every (stage, slot) ended on and whether it is valid - the start of that sub-problem's next solve (solver history, like su_lam_keep)
    double *ipw; int *ipf;                    // [T][N][5][16], [T][N]; null: always the cold start
};

__host__ __device__ inline double *coef_arr(const Dev &d, int r, int k)
{
    const size_t tn = (size_t)d.c.T * d.Nloc;
    if (k < 2) return d.coef + (size_t)r * d.chunk + (size_t)k * tn;
    if (k == 8) return d.coef + (size_t)r * d.chunk + 2 * tn;
    return d.coefL + (size_t)r * d.lchunk + (size_t)(k - 2) * tn;
}
__host__ __device__ inline double *bsum_arr(const Dev &d, int r) { return d.coef + (size_t)r * d.chunk + (size_t)NGATH * d.c.T * d.Nloc; }
__host__ __device__ inline unsigned long long *bmask_arr(const Dev &d, int r) { return (unsigned long long *)(bsum_arr(d, r) + (size_t)NBS * d.c.T * d.J); }
__host__ __device__ inline size_t chunk_doubles(int T, int Nloc) { const size_t J = (Nloc + GS - 1) / GS; return (size_t)NGATH * T * Nloc + (size_t)(NBS + 1) * T * J; }
__host__ __device__ inline size_t lchunk_doubles(int T, int Nloc) { return (size_t)(NCOEF - NGATH) * T * Nloc; }
// row index of (slot n, dual column tt) / (slot n, stage t) in the stage-major dual arrays
__host__ __device__ inline size_t drow(const Dev &d, int n, int tt) { return (size_t)tt * d.c.N + n; }

// ------------------------------------------------------------------------------------------------
// resi_dual (mean over the slots of the squared dual change, rda_solver.py:735-737) and resi_pri (|stack(Hm)|, :688) from the block
// partials of every shard.  ONE wave (the first 64 threads of the calling workgroup), lane l sums the partials l, l+64, ... in that
// order, then a wave reduction: the values do not depend on the kernel (or workgroup size) that takes the verdict, and they are
// identical on every rank.  All threads of the workgroup call.
struct ResPre { double a[16], b[16]; };       // first batch of shard 0's residual partials, requested ahead (su_body: with everything else the launch reads)
__device__ __forceinline__ void residual_prefetch(const Dev &d, int tid, ResPre &p)
{
    const double *bs = bsum_arr(d, 0);
    const int nb = d.J * d.c.T;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int i = tid + 64 * k; const int ii = (tid < 64 && i < nb) ? i : 0; p.a[k] = bs[(size_t)ii * NBS + 3]; p.b[k] = bs[(size_t)ii * NBS + 4]; }      // (unconditional loads: see reduce_residuals)
}
// `iters`: the value of Ctrl::iters (the caller has it).  `bc` (LDS, 2 doubles, optional): the verdict reaches the other threads through it - (rd, rp) are
// returned to every thread - instead of a store to the control block and a trip back.
// (pre by reference + a flag, never a selected pointer: its members stay registers)
__device__ __forceinline__ void reduce_residuals(const Dev &d, int tid, int iters, const ResPre &pre, const bool have_pre, double *bc, double &out_rd, double &out_rp)
{
    const int T = d.c.T, N = d.c.N;
    if (tid < 64) {
        double rd = 0, rp = 0;
        if (d.obstacle_num != 0)
            for (int r = 0; r < d.P; ++r) {
                const double *bs = bsum_arr(d, r);
                const int nb = d.J * T;
                for (int base = 0; base < nb; base += 64 * 16) {        // sixteen independent loads in flight per lane and quantity,
                    double a[16], b[16];                                 // accumulated in the order of the plain loop
                    if (have_pre && r == 0 && base == 0) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) { a[k] = pre.a[k]; b[k] = pre.b[k]; }
                    } else {
                        // (entries beyond nb are read from entry 0 and not added below.  `in ? load : 0` became sixteen conditional loads with a wait each - found in the ISA, round 6)
#pragma unroll
                        for (int k = 0; k < 16; ++k) { const int i = base + tid + 64 * k; const int ii = i < nb ? i : 0; a[k] = bs[(size_t)ii * NBS + 3]; b[k] = bs[(size_t)ii * NBS + 4]; }
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) if (base + tid + 64 * k < nb) { rd += a[k]; rp += b[k]; }
                }
            }
        rd = su::wave_allreduce(rd, false);
        rp = su::wave_allreduce(rp, false);
        if (tid == 0) {
            d.ctrl->resi_dual = rd / N; d.ctrl->resi_pri = sqrt(rp); d.ctrl->resi_iter = iters;
            if (bc) { bc[0] = rd / N; bc[1] = sqrt(rp); }
        }
    }
    __syncthreads();
    if (bc) { out_rd = bc[0]; out_rp = bc[1]; }
}

extern __shared__ __attribute__((aligned(16))) double smem_su[];

// The kernel arguments of the one-workgroup launches (Dev by value: 1.3 KB = 21 lines of the scalar cache) are read by s_load where the code needs them - a
// dependent chain of scalar-cache misses through the launch's prologue (~25 s_load / s_waitcnt pairs in k_su<20>, round 6).  One load per line up front, all
// in flight together: the later ones hit.
template <int BYTES> __device__ __forceinline__ void warm_kernargs()
{
    const int __attribute__((address_space(4))) *kp = (const int __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    int acc = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) acc |= kp[o / 4];
    asm volatile("" :: "s"(acc));
}

// Where a step's result goes: the device slot [u | s | info | track out] (one contiguous block starting at out_u), and optionally
// `mirror` (pinned host memory): the same block written straight into the caller-visible host buffer and published with a
// sequence number (system-scope release) - the host polls that word instead of queueing a device-to-host copy and waiting for
// the stream (fetch_result).
struct Fin { double *out_u, *out_s; rda_info *info; double *mirror; unsigned long long seq; int slot;   // slot: out_u is the handle's contiguous result slot
             // obstacle shards: the su launch that takes the early-stop verdict itself publishes it (2 vseq + stop) in pinned host memory
             // BEFORE it starts its solve - the host polls that word and queues the next LamMuZ launch + all-gather (or stops queueing)
             // while the su-problem is still being solved: no stream synchronisation inside a step
             unsigned long long *verdict = nullptr; unsigned long long vseq = 0; };

// all threads of the calling workgroup (any size); the residuals in d.ctrl are final
__device__ __forceinline__ void publish_result(const Dev &d, const Fin &f)
{
    const int tid = threadIdx.x, nth = blockDim.x, T = d.c.T;
    for (int i = tid; i < 2 * T; i += nth) f.out_u[i] = d.u[i];
    for (int i = tid; i < 3 * (T + 1); i += nth) f.out_s[i] = d.s[i];
    if (tid == 0) {
        f.info->resi_dual = d.ctrl->resi_dual; f.info->resi_pri = d.ctrl->resi_pri;
        f.info->iters = d.ctrl->iters; f.info->su_status = d.ctrl->su_status; f.info->su_ipm_iters = d.ctrl->ipm_iters;
        f.info->lmz_fail = d.ctrl->lmz_fail;
        d.ctrl->prev_unconv = !(d.ctrl->resi_dual < d.c.iter_threshold && d.ctrl->resi_pri < d.c.iter_threshold);
    }
    // polygons of the staged scene that failed the reference's convexity test (mpc.py:476-549 prints a warning per polygon)
    if (f.slot && tid == 1) *(long long *)(f.out_u + 2 * T + 3 * (T + 1) + 6) = d.sc_bad ? (long long)*d.sc_bad : 0ll;
    if (!f.mirror) return;
    __syncthreads();
    const int n = 2 * T + 3 * (T + 1) + 7;                   // out_u, out_s, info (4 doubles), the track::Out (2), the non-convex count
    for (int i = tid; i < n; i += nth) f.mirror[i] = f.out_u[i];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store((unsigned long long *)(f.mirror + n), f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int TT, typename RefWait = su::NoRefWait>
__device__ __forceinline__ void su_body(const Dev &d, int it, const double *in_s, const double *in_u,
                                        const double *ref, const double *ref_speed,
                                        const unsigned long long *ref_flag = nullptr, unsigned long long ref_seq = 0,
                                        const Fin *fin = nullptr, RefWait ref_wait = RefWait())
{
#ifdef SU_TRACE
    const long long t_entry_ = clock64();
#endif
    const int tid = threadIdx.x;
    // ---- (round 6) ONE trip to memory for everything this launch reads before its first Newton step.  The prologue used to be a chain of dependent
    //      trips (~1.5 us each: what the previous launch wrote sits in HBM / another XCD's L2): control block (stop?) -> residual partials -> their verdict
    //      back through the control block -> solver history for the start rule -> the solve's own inputs.  Now: the part of su::Args that follows from the
    //      kernel arguments alone is filled first, the solve's inputs (su::prefetch), a copy of the control block and the first batch of the residual
    //      partials are requested together, and the bookkeeping below works on the copy (thread 0 stores what changes; nothing is read back).
    su::Args a;
    a.c.T = d.c.T; a.c.N = d.c.N; a.c.dynamics = d.c.dynamics; a.c.accelerated = d.c.accelerated;
    a.c.dt = d.c.dt; a.c.L = d.c.L; a.c.umax0 = d.c.max_speed[0]; a.c.umax1 = d.c.max_speed[1];
    a.c.ab0 = d.c.acce_bound[0]; a.c.ab1 = d.c.acce_bound[1]; a.c.ws = d.c.ws; a.c.wu = d.c.wu;
    a.c.slack_gain = d.c.slack_gain; a.c.max_sd = d.c.max_sd; a.c.min_sd = d.c.min_sd; a.c.ro1 = d.c.ro1; a.c.ro2 = d.c.ro2;
    a.c.eps_u = d.c.eps_u; a.c.tol_rd = d.su_tol[0]; a.c.tol_rp = d.su_tol[1]; a.c.tol_mu = d.su_tol[2]; a.c.light_check = d.su_light;
    if (it < d.c.iter_num - 1 && d.su_tol_early[0] > 0) { a.c.tol_rd = d.su_tol_early[0]; a.c.tol_rp = d.su_tol_early[1]; a.c.tol_mu = d.su_tol_early[2]; }
    a.in_s = it == 0 ? in_s : d.s; a.in_u = it == 0 ? in_u : d.u;
    a.ref = ref; a.ref_speed = ref_speed; a.ref_flag = ref_flag; a.ref_seq = ref_seq;
    a.ax = coef_arr(d, 0, 0); a.ay = coef_arr(d, 0, 1); a.cb = coef_arr(d, 0, 8); a.gx = coef_arr(d, 0, 4); a.gy = coef_arr(d, 0, 5);
    a.P = d.P; a.Nloc = d.Nloc; a.chunk = d.chunk;
    // the reduced form of the terms (block sums, near masks at the pose table's positions): no pass over the N terms in the set-up
    if (d.su_pre) { a.bsum = bsum_arr(d, 0); a.bmask = bmask_arr(d, 0); a.J = d.J; a.pose = d.pose; }
    // iterations >= 1 are linearised about the previous solution, whose pose table the previous su launch wrote
    if (it > 0) { a.pose = d.pose; a.pose_lin = 1; }
    a.pose_out = d.pose;
    a.d_in = d.dis; a.out_s = d.s; a.out_u = d.u; a.out_d = d.dis;
    a.lam_keep = d.su_lam_keep;
    if (it == 0 && d.su_warm_mu0 > 0 && d.su_warm_first) a.warm_shift = 1;
    su::Pre pre;
    su::prefetch<TT>(a, pre);
    Ctrl cl = *d.ctrl;
    ResPre rpre;
    const bool res_pre = it > 0 && d.obstacle_num != 0;
    if (res_pre) residual_prefetch(d, tid, rpre);
#ifdef SU_TRACE
    long long t_mark_[4] = {0, 0, 0, 0};
    t_mark_[0] = clock64();                  // every load of the prologue has been requested
#endif
    if (it == 0) {                      // first su-problem of a step: the step's bookkeeping starts here (no separate launch)
        if (tid == 0) {
            d.ctrl->stop = 0; d.ctrl->iters = 0; d.ctrl->su_status = 0; d.ctrl->ipm_iters = 0; d.ctrl->lmz_fail = 0;
            d.ctrl->resi_dual = 0; d.ctrl->resi_pri = 0; d.ctrl->finished = 0; d.ctrl->resi_iter = 0;
        }
        cl.stop = 0; cl.iters = 0; cl.su_status = 0; cl.ipm_iters = 0; cl.lmz_fail = 0; cl.resi_dual = 0; cl.resi_pri = 0; cl.finished = 0; cl.resi_iter = 0;
    } else if (cl.stop) {
        if (fin && fin->verdict && tid == 0) __hip_atomic_store(fin->verdict, 2 * fin->vseq + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#ifdef SU_TRACE
    t_mark_[1] = clock64();                  // the copy of the control block has arrived (stop flag tested)
#endif
    // The early stop of rda_solver.py:594 after iteration it-1: the residual partials of the LamMuZ launch are reduced and the verdict taken here
    // (a tail of the LamMuZ launch that did it instead - rda_opts::lmz_tail, rounds 2-5 - measured 2-4 % slower: tools/experiments/lmz_tail.patch).
    if (it > 0) {
        double rd = cl.resi_dual, rp = cl.resi_pri;
        if (cl.resi_iter != it) reduce_residuals(d, tid, cl.iters, rpre, res_pre, smem_su, rd, rp);          // (k_finish of a host-driven caller may have reduced them already)
        const bool stop_now = rd < d.c.iter_threshold && rp < d.c.iter_threshold;   // rda_solver.py:594
        if (fin && fin->verdict && tid == 0) __hip_atomic_store(fin->verdict, 2 * fin->vseq + (stop_now ? 1ull : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (stop_now) {
            __syncthreads();
            if (tid == 0) d.ctrl->stop = 1;
            // The step ends here (rda_solver.py:594-596): hand the result over now - the launches still queued behind this one return
            // at once and k_finish finds the slot written.
            if (fin && fin->out_u) {
                publish_result(d, *fin);
                if (tid == 0) d.ctrl->finished = 1;
            }
            return;
        }
        __syncthreads();                // (the broadcast slot of the verdict is LDS of the solve)
    }
#ifdef SU_TRACE
    t_mark_[2] = clock64();                  // residuals reduced, verdict taken
#endif
    if (tid == 0) { d.ctrl->wl_count = 0; d.ctrl->wlc_count = 0; d.ctrl->hint_par = cl.hint_par ^ 1; }     // the LamMuZ launches of this iteration start with an empty work list and write the other support buffer
    if (d.su_pre) a.pose_ok = cl.pose_ok;
    a.status = &d.ctrl->st_tmp; a.ipm_iters = &d.ctrl->it_tmp; a.rd0 = &d.ctrl->rd0_tmp; a.prof = d.su_prof; a.split = d.su_split; a.accept = d.su_accept; a.first_attempt = d.su_first_attempt;
    su::Result res; res.rd0 = cl.rd0_tmp;
    a.land = d.su_land; a.land_tol[0] = d.su_land_tol[0]; a.land_tol[1] = d.su_land_tol[1]; a.land_tol[2] = d.su_land_tol[2]; a.land_stat = d.ctrl->land_stat; a.land_rho = d.su_land_rho;
    // warm start of iterations >= 1 from the multipliers of the previous su-solve of THIS step (only if that one converged)
    if (it > 0 && d.su_warm_mu0 > 0 && !((cl.su_status >> (it - 1)) & 1)) { a.warm_wfl = d.su_warm_wfl; a.warm_mu0 = d.su_warm_mu0; a.warm_cap = d.su_warm_cap; }
    if (it == 0 && d.su_warm_mu0 > 0 && d.su_warm_first) { a.warm_wfl = d.su_warm_wfl; a.warm_mu0 = d.su_warm_mu0; a.warm_cap = d.su_warm_cap; }
    a.warm_tau = d.su_warm_tau; a.warm_sig = d.su_warm_sig; a.warm_clip = d.su_warm_clip;
    // While consecutive su-problems are close (static scenes: the previous solve needed one or two iterations) the warm attempt starts
    // 1e-6 from the previous solution's active bounds with its multipliers and takes near-full steps: ONE iteration + the
    // convergence pass.  After a solve that needed more (moving obstacles, a changed active set) the moderate start above is used.
    // (su_last / su_probe / su_lam_keep are solver history of the handle: rda_reset clears them, rda_get/set_su_history carry them.)
    bool hard = false;
    if (a.warm_mu0 > 0 && d.su_easy_max > 0 && cl.su_last <= d.su_easy_max) {
        a.warm_wfl = d.su_easy[0]; a.warm_mu0 = d.su_easy[1]; a.warm_clip = d.su_easy[2]; a.warm_tau = d.su_easy[3]; a.warm_sig = d.su_easy[4];
        a.warm_nopred = d.su_easy_nopred;
    } else if (a.warm_mu0 > 0 && d.su_hard_mu0 > 0 && cl.prev_unconv && cl.su_hardlike && cl.su_last < 99) {
        // The ADMM of the previous step did not converge (a caller that re-sorts its obstacles every tick, quirk Q5; many moving obstacles)
        // AND consecutive su-problems really are far apart: the last solve's first iterate - the previous solution with its multipliers - had a
        // relative dual residual above su::HARD_RD0.  A warm attempt then does best from a point WELL inside the boxes (slack floor 1) with
        // the previous multipliers and next to no barrier (mu0 1e-3; the rows of d: su::HARD_DMU) - and it beats the cold start, so that rule is
        // skipped.  Oracle (same rule), interior-point iterations per su-solve: re-sorted north star 7.6 -> 5.7, N = 2000 8.1 -> 6.3, C4
        // re-sorted 14.2 -> 12.7, converged / easy loops unchanged.  (Round 4 keyed on the last solve's iteration count: a hard-started solve
        // costs >= 3 iterations whatever the problem, so `> 2` locked the easy start out - iter_num = 1: 1.0 -> 3.0 iterations per solve -
        // and `> 3` left most of the gain: 7.6 -> 6.2.)
        a.warm_wfl = d.su_hard_wfl; a.warm_mu0 = d.su_hard_mu0; a.hard_dmu = su::HARD_DMU; hard = true;
    }
    // Hard regime (many moving obstacles: consecutive su-problems are far apart): a warm attempt then needs MORE iterations than a cold
    // start.  While the last solve needed more than su_cold_from iterations the solve starts cold; every su_cold_probe-th such solve tries the
    // warm start again, so that the handle finds its way back when the scene calms down.
    // landing first: the warm-started su-problems
    a.land_level0 = (a.land && cl.land_hard > 0) ? 1 : 0;
    a.land_rho_prev = cl.land_rho_prev;
    const bool lf_eligible = a.warm_mu0 > 0 && a.land;       // (every warm attempt: ADMM iterations >= 1, and the first su-problem of a tick - the previous tick's solution shifted by one stage)
    if (lf_eligible) a.land_first = d.su_land_first == 2 ? (cl.spec_credit >= 0 ? 2 : 1) : d.su_land_first;
    const bool blind_opportunity = lf_eligible && d.su_land_first == 2 && it > 0 && cl.land_easy >= d.su_land_blind_from && d.su_land_blind_from > 0 && cl.land_rho_prev > 0;
    if (blind_opportunity && cl.blind_ok >= 0) a.land_blind = 1;
    if (!hard && a.warm_mu0 > 0 && d.su_cold_from > 0 && cl.su_last > d.su_cold_from && cl.su_last < 99 && cl.su_probe % d.su_cold_probe != d.su_cold_probe - 1) a.warm_mu0 = 0;
#ifdef SU_TRACE
    a.t_entry = t_entry_; t_mark_[3] = clock64();
    for (int k = 0; k < 4; ++k) a.t_mark[k] = t_mark_[k];
#endif
    su::solve<TT>(a, smem_su, pre, true, res, ref_wait);
    if (tid == 0) {                     // (from the verdict the solve left in registers and the copy of the control block: stores only)
        const int su_last = res.status == 0 ? res.iters : 99;
        d.ctrl->iters = it + 1;
        if (res.status != 0) d.ctrl->su_status = cl.su_status | (1 << it);
        d.ctrl->ipm_iters = cl.ipm_iters + res.iters;
        d.ctrl->su_last = su_last;
        d.ctrl->su_hardlike = res.rd0 > su::HARD_RD0;
        d.ctrl->su_probe = (d.su_cold_from > 0 && su_last > d.su_cold_from && su_last < 99) ? cl.su_probe + 1 : 0;
        d.ctrl->pose_ok = 0;          // the pose table has moved on; the LamMuZ launch that follows makes the masks that go with it
        if (a.land) {
            const bool easy1 = res.status == 0 && res.iters == 0 && res.rounds_all == 1;
            d.ctrl->land_easy = easy1 ? (cl.land_easy < 8 ? cl.land_easy + 1 : 8) : 0;
            if (res.land_rho > 0) d.ctrl->land_rho_prev = res.land_rho;
            if (blind_opportunity) d.ctrl->blind_ok = res.blind == 1 ? (cl.blind_ok < 4 ? cl.blind_ok + 1 : 4) : (res.blind == 2 ? -6 : (cl.blind_ok < 0 ? cl.blind_ok + 1 : cl.blind_ok));
        }
        if (a.land && res.status == 0) {
            // sticky: a landing of three or more rounds sends the next FOUR solves to the later stop (a good landing at the later stop says nothing about the
            // earlier one: with a one-solve memory C4 alternated between the two levels)
            int lh = cl.land_hard;
            if (res.land_rounds >= 3) lh = 4; else if (lh > 0) lh -= 1;
            d.ctrl->land_hard = lh;
        }
        if (lf_eligible && d.su_land_first == 2) {
            // speculative landings pay where consecutive su-problems keep their active set (static scenes: 50 - 65 % accepted) and cost two landing rounds where
            // they do not (C4, moving obstacles: none accepted): a handle that keeps failing tries every third eligible solve only; at one success in two the credit grows
            int cr = cl.spec_credit;
            if (res.spec == 1) cr = cr + 3 > 6 ? 6 : cr + 3; else if (res.spec == 2) cr -= 2; else if (cr < 0) cr += 1;
            d.ctrl->spec_credit = cr;
        }
    }
}

template <int TT> __global__ __launch_bounds__(su::NT) void k_su(Dev d, int it, const double *in_s, const double *in_u, Fin fin)
{
    warm_kernargs<sizeof(Dev) + 2 * sizeof(void *) + sizeof(int) + sizeof(Fin)>();
    su_body<TT>(d, it, in_s, in_u, d.ref, d.ref_speed, nullptr, 0, &fin);
}

// final bookkeeping of a step that ran all its iterations: residuals of the last one, result slot
__device__ __forceinline__ void finish_body(const Dev &d, const Fin &f)
{
    if (d.ctrl->finished) return;                   // (uniform) the launch that ended the step already handed the result over
    if (!d.ctrl->stop && d.ctrl->resi_iter != d.ctrl->iters) { ResPre none; double rd, rp; reduce_residuals(d, threadIdx.x, d.ctrl->iters, none, false, nullptr, rd, rp); }
    __syncthreads();
    publish_result(d, f);
}

__global__ __launch_bounds__(su::NT) void k_finish(Dev d, Fin f) { finish_body(d, f); }

__device__ __forceinline__ void begin_body(const Dev &d)
{
    if (threadIdx.x == 0) {
        d.ctrl->stop = 0; d.ctrl->iters = 0; d.ctrl->su_status = 0; d.ctrl->ipm_iters = 0; d.ctrl->lmz_fail = 0;
        d.ctrl->resi_dual = 0; d.ctrl->resi_pri = 0; d.ctrl->finished = 0; d.ctrl->wl_count = 0; d.ctrl->wlc_count = 0; d.ctrl->resi_iter = 0;
    }
}

__global__ void k_begin(Dev d) { begin_body(d); }

// ------------------------------------------------------------------------------------------------
// K1: one wavefront per (obstacle n, stage t); 4 wavefronts per workgroup.
// candidate list + vertices of every staged obstacle slot (one wave per (slot, time slot)); runs once per upload
__device__ __forceinline__ void prepare_body(const Dev &d, lmz::WaveLDS *wl)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, E = d.c.E;
    const int w = blockIdx.x * 4 + wv;
    if (w >= d.c.N * d.nt) return;
    const int n = w / d.nt;
    lmz::WaveLDS &W = wl[wv];
    if (lane < 2 * E) W.A[lane >> 1][lane & 1] = d.A[(size_t)w * E * 2 + lane];
    if (lane < E) W.b[lane] = d.b[(size_t)w * E + lane];
    lmz::wave_sync();
    lmz::build_lists(W, E, d.cone[n], lane);
    lmz::wave_sync();
    if (lane < d.oc_ls) d.oc_lamc[(size_t)w * d.oc_ls + lane] = W.lamc[lane];
    if (lane < d.oc_vs) d.oc_vtx[(size_t)w * d.oc_vs + lane] = (&W.vtx[0][0])[lane];
    if (lane == 0) { d.oc_cnt[2 * w] = W.npv; d.oc_cnt[2 * w + 1] = W.nlv; }
}
__global__ __launch_bounds__(256) void k_prepare(Dev d) { __shared__ lmz::WaveLDS wl[4]; prepare_body(d, wl); }
__global__ __launch_bounds__(256) void k_prepare_fleet(const Dev *devs) { __shared__ lmz::WaveLDS wl[4]; prepare_body(devs[blockIdx.y], wl); }

// Unit w of a rank's LamMuZ grid -> (stage t, local slot nl): the SLOT runs fastest and a stage is padded to GS J units, so a
// workgroup of the packed kernels (GS = 8 rows: 2 waves x 4 rows) owns the slots GS j .. GS j + GS-1 of ONE stage: with the stage-major
// dual arrays and the [T][Nloc] condensed-term arrays everything it reads and writes are whole 64-byte half lines, and its rows reduce
// to one block partial.  (Two-wave workgroups: the north-star grid T J = 500 of them is ONE round on 256 CUs at one wave per SIMD.)
__device__ __forceinline__ void unit_of(const Dev &d, int w, int &t, int &nl) { const int S = GS * d.J; t = w / S; nl = w - t * S; }
// Launch index of a packed workgroup -> its block (stage t, GS-slot group j), or t = -1 for a filler.  Workgroups are handed to the
// 8 XCDs round robin (block b runs on XCD b % 8 - observed, used for speed only), and each XCD has its own L2.  The T workgroups that
// share the staged data of the same obstacles (half-spaces, candidate lists, vertices: ~0.5 KB per slot, read once per STAGE) should
// meet in ONE L2, and every XCD must get the same number of workgroups (the north-star grid is exactly one round of the chip): the
// blocks are numbered group-major (u = j T + t) and cut into 8 equal contiguous ranges, one per XCD - a group's stages land on one
// XCD (two for the few groups a cut goes through), neighbouring groups (which share the 128-byte lines of the [T][N] arrays) too.
//   b = 8 q + x ;  u = x per + q ,  per = ceil(T J / 8) ;  j = u / T ,  t = u % T
__host__ __device__ inline int packed_grid(int T, int J) { return 8 * ((T * J + 7) / 8); }
__device__ __forceinline__ void block_of(const Dev &d, int b, int &t, int &j)
{
    const int T = d.c.T, W = T * d.J, per = (W + 7) / 8, x = b & 7, q = b >> 3, u = x * per + q;
    j = u / T; t = u - j * T;
    if (u >= W) t = -1;
}

// What one solved (or failed) row hands to the su-problem and to the residuals, as stored in the coef arrays
struct RowOut { double ax, ay, bl, c3, c4, c5, res, hh; };
__device__ __forceinline__ void store_row(const Dev &d, int k, const RowOut &o)
{
    coef_arr(d, d.rank, 0)[k] = o.ax; coef_arr(d, d.rank, 1)[k] = o.ay; coef_arr(d, d.rank, 2)[k] = o.bl;   // rda_solver.py:541-542
    coef_arr(d, d.rank, 3)[k] = o.c3; coef_arr(d, d.rank, 4)[k] = o.c4; coef_arr(d, d.rank, 5)[k] = o.c5;
    coef_arr(d, d.rank, 6)[k] = o.res; coef_arr(d, d.rank, 7)[k] = o.hh; coef_arr(d, d.rank, 8)[k] = o.bl + o.c3;
}
// a row that cannot be solved (non-finite data or result): previous lam, mu, z, xi, zeta stay; the stage drops out of the su hinge
// (a = 0, offset and g as they were); its residual is inf (rda_solver.py:781-793)
__device__ __forceinline__ RowOut failed_row(const Dev &d, int k)
{
    RowOut o; o.ax = 0; o.ay = 0; o.bl = 0; o.c3 = coef_arr(d, d.rank, 3)[k]; o.c4 = coef_arr(d, d.rank, 4)[k]; o.c5 = coef_arr(d, d.rank, 5)[k];
    o.res = INFINITY; o.hh = 0;
    return o;
}

// GS row records [GS][6] = (|a|^2, g.a, g x a, dual residual, |Hm|^2, near) in LDS -> the block partial of (stage t, block j): threads
// q = 0..4 sum one quantity over the rows IN ROW ORDER (dead rows hold zeros), thread 5 packs the near mask.  Stored write-through
// (sc1: 8-byte agent-scope stores), as round 2's tail of the launch read them without a release fence (tools/experiments/lmz_tail.patch); kept: the next su launch reads them from L2 either way.
__device__ __forceinline__ void block_partial(const Dev &d, int t, int j, const double (*rowv)[6], int q)
{
    const size_t bi = (size_t)t * d.J + j;
    if (q < NBS) {
        double acc = 0;
        for (int r = 0; r < GS; ++r) acc += rowv[r][q];
        __hip_atomic_store(bsum_arr(d, d.rank) + bi * NBS + q, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (q == NBS) {
        unsigned long long m = 0;
        for (int r = 0; r < GS; ++r) m |= rowv[r][5] != 0.0 ? 1ull << r : 0ull;
        __hip_atomic_store(bmask_arr(d, d.rank) + bi, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void row_record(double *rv, const Dev &d, const RowOut &o, double px, double py)
{
    const su::RowTerm q = su::row_term(o.ax, o.ay, o.c4, o.c5, o.bl + o.c3, px, py, d.c.max_sd, true);
    rv[0] = q.aa; rv[1] = q.ga; rv[2] = q.gxa; rv[3] = o.res; rv[4] = o.hh; rv[5] = (!d.c.accelerated || q.near) ? 1.0 : 0.0;
}

// K1, one (slot, stage) sub-problem per wavefront; 4 wavefronts per workgroup (shapes with E+R+1 > 16, RDA_LMZ_ROWS=0, and the
// no-obstacle case).  Writes the per-row terms only: k_lmz_finalize forms the block partials and runs the tail behind it.
__device__ __forceinline__ size_t hint_key(const Dev &d, int n, int t)
{
    int key = n;
    if (d.slot_src && n < d.src_used) { const int src = d.slot_src[n]; if (src >= 0 && src < d.src_cap) key = d.c.N + src; }
    return (size_t)t * d.hint_stride + key;
}
// `first` = first LamMuZ launch of a tick: the support remembered for stage t+1 of the previous tick
__device__ __forceinline__ int hint_read(const Dev &d, int par, int n, int t, bool first)
{
    return d.hint[(size_t)(par ^ 1) * d.hint_len + hint_key(d, n, first && t + 1 < d.c.T ? t + 1 : t)];
}
__device__ __forceinline__ void hint_write(const Dev &d, int par, int n, int t, int v) { d.hint[(size_t)par * d.hint_len + hint_key(d, n, t)] = v; }

__device__ __forceinline__ void lammuz_body(const Dev &d, const int block, const int it)
{
#pragma clang fp contract(on)          // see lammuz_device.h: results independent of the kernel the body is compiled into
    __shared__ lmz::WaveLDS wl[4];
    __shared__ lmz::RobotLDS rb;
    const int T = d.c.T, N = d.c.N, E = d.c.E, R = d.c.R;
    if (d.ctrl->stop) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (d.obstacle_num == 0) {
        // quirk Q9 (rda_solver.py:564-568): only slot N-1 loses its lam'A / lam'b products
        if (block == 0 && d.rank == (N - 1) / d.Nloc)
            for (int t = threadIdx.x; t < T; t += 256) {
                int i = t * d.Nloc + (N - 1) % d.Nloc;
                coef_arr(d, d.rank, 0)[i] = 0; coef_arr(d, d.rank, 1)[i] = 0; coef_arr(d, d.rank, 2)[i] = 0;
                coef_arr(d, d.rank, 8)[i] = coef_arr(d, d.rank, 3)[i];
            }
        return;
    }
    if (threadIdx.x < 2 * R) rb.G[threadIdx.x >> 1][threadIdx.x & 1] = d.G[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + R) rb.h[threadIdx.x - 64] = d.h[threadIdx.x - 64];
    if (threadIdx.x >= 128 && threadIdx.x < 128 + 40) rb.muc[threadIdx.x - 128] = d.muc[threadIdx.x - 128];
    if (threadIdx.x >= 192 && threadIdx.x < 192 + 56) (&rb.rv[0][0])[threadIdx.x - 192] = (&d.rv[0][0])[threadIdx.x - 192];
    if (threadIdx.x == 255) { rb.nmv = d.nmv; rb.nrv = d.nrv; }
    const int w = block * 4 + wv;
    int t, nl; unit_of(d, w, t, nl);
    const bool live = t < T && nl < d.Nlive;
    if (!live) { t = 0; nl = 0; }
    const int n = d.rank * d.Nloc + nl;                        // this rank's obstacle shard [rank*Nloc, (rank+1)*Nloc)
    lmz::WaveLDS &W = wl[wv];
    const size_t ao = ((size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0)) * E;
    if (lane < 2 * E) W.A[lane >> 1][lane & 1] = d.A[ao * 2 + lane];
    if (lane < E) W.b[lane] = d.b[ao + lane];
    __syncthreads();
    if (!live) return;
    lmz::Params P;
    P.E = E; P.R = R; P.norm2 = d.cone[n];
    const double *ps = d.pose + 4 * t;                        // position of column t+1, heading of column t (quirk Q1)
    P.px = ps[0]; P.py = ps[1]; P.cs = ps[2]; P.sn = ps[3];
    const size_t o = drow(d, n, t + 1), zi = drow(d, n, t);
    P.xi0 = d.xi[2 * o]; P.xi1 = d.xi[2 * o + 1];
    const double zeta = d.zeta[zi], dbar = d.dis[t];
    P.kappa0 = zeta - dbar; P.ro2 = d.c.ro2; P.delta = d.c.delta;
    lmz::Sol best;
    // previous value of this lane's dual entry: warm start of the support + the dual residual below
    double prev = 0.0;
    if (lane < E) prev = d.lam[o * E + lane];
    else if (lane < E + R) prev = d.mu[o * R + lane - E];
    // A sub-problem whose data are not finite cannot be solved: like a LamMuZ solve of the reference that does not end OPTIMAL
    // (rda_solver.py:781-793) it keeps its previous duals and its residual is inf (no early stop).  The solver below is run on
    // harmless stand-in data instead (its loops then see no NaN) and its answer is dropped.
    bool bad = !isfinite(P.px + P.py + P.cs + P.sn + P.xi0 + P.xi1 + P.kappa0);
    if (lane < 2 * E) bad = bad || !isfinite(W.A[lane >> 1][lane & 1]);
    if (lane < E) bad = bad || !isfinite(W.b[lane]);
    bad = __ballot(bad) != 0;
    if (bad) {
        if (lane < 2 * E) W.A[lane >> 1][lane & 1] = 0;
        if (lane < E) W.b[lane] = 0;
        P.px = P.py = P.sn = P.xi0 = P.xi1 = P.kappa0 = 0; P.cs = 1;
        lmz::wave_sync();
    }
    lmz::pose_products(W, P, lane);
    {   // candidate list and vertices of this slot from the upload-time cache
        const size_t oc = (size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0);
        if (lane < d.oc_ls) W.lamc[lane] = d.oc_lamc[oc * d.oc_ls + lane];
        if (lane < d.oc_vs) (&W.vtx[0][0])[lane] = d.oc_vtx[oc * d.oc_vs + lane];
        if (lane == 0) { W.npv = d.oc_cnt[2 * oc]; W.nlv = d.oc_cnt[2 * oc + 1]; }
    }
    lmz::wave_sync();
    const int hpar = d.ctrl->hint_par & 1;
    if (!d.warm || !lmz::solve_wave_warm<64>(W, rb, P, lane, hint_read(d, hpar, n, t, it == 0), best)) lmz::solve_wave(W, rb, P, lane, best);
    if (lane == 0) hint_write(d, hpar, n, t, best.id >> 1);
    if (d.centre) lmz::central_normal_wave<64>(W, rb, P, lane, best);
    bad = bad || !(isfinite(best.cost) && isfinite(best.m) && isfinite(best.H0) && isfinite(best.H1));
    const int k = t * d.Nloc + nl;
    if (bad) {
        if (lane == 0) {
            store_row(d, k, failed_row(d, k));
            hint_write(d, hpar, n, t, -1);
            atomicAdd(&d.ctrl->lmz_fail, 1);
        }
        return;
    }
    // ---- fused dual / residual updates (every lane holds the winner) ----------------------------
    const double znew = (d.c.accelerated ? 0.5 : 1.0) * (best.m > 0 ? best.m : 0.0);     // tie-break T2
    double res = 0;
    if (lane < E) {
        double v = lmz::lam_of(best, P.norm2, lane);
        res = (v - prev) * (v - prev); d.lam[o * E + lane] = v;
    } else if (lane < E + R) {
        int j = lane - E;
        double v = lmz::mu_of(best, j);
        res = (v - prev) * (v - prev); d.mu[o * R + j] = v;
    } else if (lane == E + R) {
        double old = d.z[zi];
        res = (znew - old) * (znew - old); d.z[zi] = znew;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) res += __shfl_xor(res, off, 64);
    if (lane == 0) {
        double ax = 0, ay = 0, bl = 0, mh = 0, gx = 0, gy = 0;
        for (int i = 0; i < E; ++i) {
            double v = lmz::lam_of(best, P.norm2, i);
            ax += v * W.A[i][0]; ay += v * W.A[i][1]; bl += v * W.b[i];
        }
        for (int j = 0; j < R; ++j) {
            double v = lmz::mu_of(best, j);
            mh += v * rb.h[j]; gx += v * rb.G[j][0]; gy += v * rb.G[j][1];
        }
        const double hx = gx + P.cs * ax + P.sn * ay, hy = gy - P.sn * ax + P.cs * ay;      // Hm, :682
        const double xin0 = P.xi0 + hx, xin1 = P.xi1 + hy;                                  // :683
        d.xi[2 * o] = xin0; d.xi[2 * o + 1] = xin1;
        const double im = ax * P.px + ay * P.py - bl - mh;                                  // :659
        const double zetan = zeta + im - dbar - znew;                                       // :666
        d.zeta[zi] = zetan;
        RowOut ro; ro.ax = ax; ro.ay = ay; ro.bl = bl; ro.c3 = mh + znew - zetan; ro.c4 = gx + xin0; ro.c5 = gy + xin1;
        ro.res = res; ro.hh = hx * hx + hy * hy;
        store_row(d, k, ro);
    }
}

__global__ __launch_bounds__(256) void k_lammuz(Dev d, int it) { lammuz_body(d, blockIdx.x, it); }

// K1, packed: FOUR sub-problems per wavefront, one per 16-lane DPP row (16 per workgroup).  The warm-started path - one
// candidate on two lanes, the optimality certificate on E+R lanes, the central-normal step on (vertex, vertex) pairs, the
// dual updates on E+R+1 lanes - never used more than a quarter of a wave, and the kernel is bound by instruction issue
// (DESIGN.md section 5): the rows run it side by side.  Only a row whose certificate fails needs the 64-lane enumeration;
// those rows are served one after the other by the whole wave.  Same device functions, same arithmetic, same results as
// lammuz_body.  Requires E + R + 1 <= 16 (else the one-per-wave body is launched).
// MODE 0: everything in one kernel (single ego, latency), INCLUDING the block partial of the workgroup's GS slots.  Dense grids are served by
//         three launches instead:
// MODE 1: the common path only - three waves per SIMD, no spills; a row whose warm candidate fails its certificate goes on the
//         handle's work list (Dev::wl) and writes nothing;
// MODE 2: the rows on the work list, ONE per wave (row 0 of the wave; rows 1-3 idle) and pass, straight to the enumeration, then
//         the very same 16-lane code as modes 0 / 1 - so the results do not depend on which launch solved a row;
//         then k_lmz_finalize: block partials from the stored terms (same function, same order as mode 0) and the tail.
#ifdef RDA_LMZ_CLK
__device__ unsigned long long *g_lmz_clk = nullptr;
__device__ int *g_lmz_faillog = nullptr;          // [0] count, then (old hint, new support id, circle?) triples of the rows whose certificate failed
#endif
template <int MODE = 0, bool CW = false> __device__ __forceinline__ void lammuz_body_rows(const Dev &d, const int block, const int nblocks, const int it, const Fin &fin)
{
#pragma clang fp contract(on)          // see lammuz_device.h
    constexpr int WPB = GS / 4;        // waves per workgroup (4 rows each)
    __shared__ lmz::WaveLDS wl[GS];
    __shared__ lmz::RobotLDS rb;
    __shared__ double rowv[MODE == 0 ? GS : 1][6];
    const int T = d.c.T, E = d.c.E, R = d.c.R;
#ifdef RDA_LMZ_CLK
    // debug build only (tools/lmz_wave_clocks.py): clock64 ticks per section of a wave, in the wave's own 16-word slot of g_lmz_clk
    long long clk_prev = clock64(); const long long clk_in = clk_prev; bool clk_enum = false; const unsigned long long clk_wall_in = (unsigned long long)wall_clock64();
    unsigned long long *const clk_slot = g_lmz_clk ? g_lmz_clk + 16 * (size_t)(block * (GS / 4) + (threadIdx.x >> 6)) : nullptr;
#define LMZ_CLK(k) do { if (MODE == 0) { const long long now_ = clock64(); if (clk_slot && (threadIdx.x & 63) == 0) clk_slot[k] += (unsigned long long)(now_ - clk_prev); clk_prev = now_; } } while (0)
#else
#define LMZ_CLK(k) do { } while (0)
#endif
    int tb = 0, jb = 0;                                        // (modes 0, 1) the workgroup's block: stage and GS-slot group
    if (MODE != 2) { block_of(d, block, tb, jb); if (tb < 0) return; }      // a filler of the XCD-aware launch order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = lane >> 4, gl = lane & 15;
    // MODE 0 (one ego, one wave per SIMD: the launch IS the dependent chain of one wave, and six of its links were trips to the L2 -
    // stop flag, half-spaces, pose / duals, support cache, slot -> obstacle, remembered support): everything a row reads is requested
    // here, in two batches, before the first of them is waited for.  The throughput forms (modes 1, 2) keep their loads where the
    // values are used: there the registers are worth more than the latency (three waves per SIMD hide it).
    struct Early { double px, py, cs, sn, xi0, xi1, zeta, dbar, prev, vtx[4], A, b, G, h, zold; int cone, src, hpar, hint, npv, nlv; unsigned char lamc[3]; } ey;
    if (MODE == 0) {
        const int nl0 = jb * GS + wv * 4 + row, nl = nl0 < d.Nlive ? nl0 : 0, n = d.rank * d.Nloc + nl;
        ey.src = (d.slot_src && n < d.src_used) ? d.slot_src[n] : -1;
        ey.hpar = d.ctrl->hint_par & 1;
    }
    if (d.ctrl->stop) return;                                 // (not merged into the batch below: launches behind the stop flag are on the next tick's way and must stay short - measured)
    if (MODE == 2 && block * WPB >= d.ctrl->wl_count + (d.ctrl->wlc_count + 3) / 4) return;  // work-list form: nothing for this workgroup (the list is short or empty since round 3: no trip for the robot data)
    LMZ_CLK(0);
    if (MODE == 0) {
        const int t = tb, nl0 = jb * GS + wv * 4 + row, nl = nl0 < d.Nlive ? nl0 : 0, n = d.rank * d.Nloc + nl;
        const double *ps = d.pose + 4 * t;
        const size_t o = drow(d, n, t + 1), zi = drow(d, n, t), oc = (size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0);
        ey.px = ps[0]; ey.py = ps[1]; ey.cs = ps[2]; ey.sn = ps[3];
        ey.xi0 = d.xi[2 * o]; ey.xi1 = d.xi[2 * o + 1]; ey.zeta = d.zeta[zi]; ey.dbar = d.dis[t]; ey.cone = d.cone[n];
        ey.zold = d.z[zi];                // (the previous z of the row's residual: it was fetched where it is used, a trip to memory inside the update phase)
        // (ONE unconditional load through a selected address: as `gl < E ? lam : (gl < E + R ? mu : 0)` the two loads shared their destination register and the
        // second one waited - vmcnt(0), in the middle of the batch - for the first; lanes beyond E + R read lam[0] and are zeroed where `prev` is used.  Round 6)
        ey.prev = *(gl < E ? &d.lam[o * E + gl] : (gl < E + R ? &d.mu[o * R + gl - E] : &d.lam[o * E]));
        // half-spaces of the row and the robot's (G, h): they used to be fetched one after the other behind this batch - load, vmcnt(0), LDS store, four times
        // (G, h, A, b): four dependent trips to the L2 in a launch that IS one wave's dependent chain
        {
            const size_t ao = ((size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0)) * E;
            ey.A = d.A[ao * 2 + (gl < 2 * E ? gl : 0)]; ey.b = d.b[ao + (gl < E ? gl : 0)];
            ey.G = d.G[(int)threadIdx.x < 2 * R ? threadIdx.x : 0]; ey.h = d.h[(int)threadIdx.x < R ? threadIdx.x : 0];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) ey.lamc[k] = gl + 16 * k < d.oc_ls ? d.oc_lamc[oc * d.oc_ls + gl + 16 * k] : (unsigned char)0;      // (wave-uniform bounds for k >= 1 at E <= 5: whole loads drop out)
#pragma unroll
        for (int k = 0; k < 4; ++k) ey.vtx[k] = gl + 16 * k < d.oc_vs ? d.oc_vtx[oc * d.oc_vs + gl + 16 * k] : 0.0;
        ey.npv = d.oc_cnt[2 * oc]; ey.nlv = d.oc_cnt[2 * oc + 1];
        {   // the remembered support (hint_read with the slot's source already at hand)
            const int key = (ey.src >= 0 && ey.src < d.src_cap) ? d.c.N + ey.src : n, tr = (it == 0 && t + 1 < T) ? t + 1 : t;
            ey.hint = d.hint[(size_t)(ey.hpar ^ 1) * d.hint_len + (size_t)tr * d.hint_stride + key];
        }
    }
    if (MODE == 0) {                                          // (2 R <= 16 < the workgroup: one element per thread, from the batch above)
        if ((int)threadIdx.x < 2 * R) rb.G[threadIdx.x >> 1][threadIdx.x & 1] = ey.G;
        if ((int)threadIdx.x < R) rb.h[threadIdx.x] = ey.h;
    } else {
        for (int i = threadIdx.x; i < 2 * R; i += 64 * WPB) rb.G[i >> 1][i & 1] = d.G[i];
        for (int i = threadIdx.x; i < R; i += 64 * WPB) rb.h[i] = d.h[i];
    }
    for (int i = threadIdx.x; i < 40; i += 64 * WPB) rb.muc[i] = d.muc[i];
    for (int i = threadIdx.x; i < 56; i += 64 * WPB) (&rb.rv[0][0])[i] = (&d.rv[0][0])[i];
    if (threadIdx.x == 0) { rb.nmv = d.nmv; rb.nrv = d.nrv; }
    // Work-list form: the list has two ends.  Polygon rows (from the front, Ctrl::wl_count) take a wave each: they go straight to the 64-lane enumeration.
    // Circle rows (from the back, Ctrl::wlc_count) take a 16-lane row each, FOUR per wave: their remembered case runs side by side like in the single-ego
    // form, and only a row whose certificate fails is enumerated by its wave (round 6).  A wave slot = one polygon row or one group of four circle rows.
    const int cntP = MODE == 2 ? d.ctrl->wl_count : 1, cntC = MODE == 2 ? d.ctrl->wlc_count : 0;
    const int cnt = cntP + (cntC + 3) / 4;
    for (int base = MODE == 2 ? block * WPB : 0; base < cnt; base += MODE == 2 ? nblocks * WPB : 1) {       // modes 0, 1: one pass
    if (MODE == 2) __syncthreads();                            // the slabs of the previous pass are free again
    const bool entry = MODE == 2 && base + wv < cnt;            // (wave-uniform) this wave has a work-list slot in this pass
    // a row past the end (a dead slot of the padded stage, an idle row of mode 2) shadows an existing unit and writes nothing
    int t, nl;
    bool live2 = false;
    if (MODE == 2) {
        const int sl = entry ? base + wv : base;               // (a wave without a slot shadows the first slot of the pass)
        int widx;
        if (sl < cntP) { widx = sl; live2 = entry && row == 0; }
        else { const int c0 = 4 * (sl - cntP), ci = c0 + row; live2 = entry && ci < cntC; widx = d.wl_cap - 1 - (live2 ? ci : c0); }
        unit_of(d, d.wl[widx], t, nl);
    } else { t = tb; nl = jb * GS + wv * 4 + row; }
    const bool live = MODE == 2 ? live2 : nl < d.Nlive;
    if (!live) nl = 0;
    const int n = d.rank * d.Nloc + nl;
    lmz::WaveLDS &W = wl[wv * 4 + row];
    const size_t ao = ((size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0)) * E;
    if (MODE == 0) {
        if (gl < 2 * E) W.A[gl >> 1][gl & 1] = ey.A;
        if (gl < E) W.b[gl] = ey.b;
    } else {
        if (gl < 2 * E) W.A[gl >> 1][gl & 1] = d.A[ao * 2 + gl];
        if (gl < E) W.b[gl] = d.b[ao + gl];
    }
    __syncthreads();
    LMZ_CLK(1);
    lmz::Params P;
    P.E = E; P.R = R; P.norm2 = MODE == 0 ? ey.cone : d.cone[n];
    const double *ps = d.pose + 4 * t;                        // position of column t+1, cos / sin of the heading of column t (quirk Q1)
    if (MODE == 0) { P.px = ey.px; P.py = ey.py; P.cs = ey.cs; P.sn = ey.sn; }
    else { P.px = ps[0]; P.py = ps[1]; P.cs = ps[2]; P.sn = ps[3]; }
    const size_t o = drow(d, n, t + 1), zi = drow(d, n, t);
    if (MODE == 0) { P.xi0 = ey.xi0; P.xi1 = ey.xi1; } else { P.xi0 = d.xi[2 * o]; P.xi1 = d.xi[2 * o + 1]; }
    const double zeta = MODE == 0 ? ey.zeta : d.zeta[zi], dbar = MODE == 0 ? ey.dbar : d.dis[t];
    P.kappa0 = zeta - dbar; P.ro2 = d.c.ro2; P.delta = d.c.delta;
    lmz::Sol best;
    double prev = 0.0;
    if (MODE == 0) prev = gl < E + R ? ey.prev : 0.0;
    else if (gl < E) prev = d.lam[o * E + gl];
    else if (gl < E + R) prev = d.mu[o * R + gl - E];
    // non-finite data: see lammuz_body (the row solves harmless stand-in data, keeps its previous duals, residual inf)
    const double px0 = P.px, py0 = P.py;                       // (the pose as read: a failed row overwrites it with stand-ins)
    bool bad = !isfinite(P.px + P.py + P.cs + P.sn + P.xi0 + P.xi1 + P.kappa0);
    if (gl < 2 * E) bad = bad || !isfinite(W.A[gl >> 1][gl & 1]);
    if (gl < E) bad = bad || !isfinite(W.b[gl]);
    bad = ((__ballot(bad) >> (16 * row)) & 0xffffull) != 0;
    if (bad) {
        if (gl < 2 * E) W.A[gl >> 1][gl & 1] = 0;
        if (gl < E) W.b[gl] = 0;
        P.px = P.py = P.sn = P.xi0 = P.xi1 = P.kappa0 = 0; P.cs = 1;
    }
    lmz::wave_sync();
    lmz::pose_products(W, P, gl);
    {
        const size_t oc = (size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0);
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) if (gl + 16 * k < d.oc_ls) W.lamc[gl + 16 * k] = ey.lamc[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (gl + 16 * k < d.oc_vs) (&W.vtx[0][0])[gl + 16 * k] = ey.vtx[k];
            if (gl == 0) { W.npv = ey.npv; W.nlv = ey.nlv; }
        } else {
            for (int i = gl; i < d.oc_ls; i += 16) W.lamc[i] = d.oc_lamc[oc * d.oc_ls + i];
            for (int i = gl; i < d.oc_vs; i += 16) (&W.vtx[0][0])[i] = d.oc_vtx[oc * d.oc_vs + i];
            if (gl == 0) { W.npv = d.oc_cnt[2 * oc]; W.nlv = d.oc_cnt[2 * oc + 1]; }
        }
    }
    lmz::wave_sync();
    LMZ_CLK(2);
    const int hpar = MODE == 0 ? ey.hpar : (d.ctrl->hint_par & 1);
    // (work-list form: only a CIRCLE row reads its remembered support - round 6, see below)
    const int hint_in = MODE == 0 ? ey.hint : (MODE != 2 ? hint_read(d, hpar, n, t, it == 0) : ((P.norm2 && live) ? hint_read(d, hpar, n, t, it == 0) : -1));
    bool ok = MODE != 2 && d.warm && lmz::solve_wave_warm<16>(W, rb, P, lane, hint_in, best);
    // CW (the single-ego kernel only, see lmz::warm_circle): the remembered case of a circle obstacle's row, four rows of the wave side by side
    if (CW && MODE == 0 && d.warm && P.norm2 && live && !bad) ok = lmz::warm_circle<16>(W, rb, P, lane, hint_in, best);
    // ... and in the WORK-LIST form (round 6): the common-path kernel of the dense launch forms and the fleet cannot afford the circle cases inline (registers:
    // lmz::warm_circle's header), so it defers every circle row - and the work-list kernel used to enumerate each of them, on every ADMM iteration (64 lanes, the
    // interior candidate a Newton iteration: ~31 k cycles a row).  This kernel has the registers (one wave per SIMD): the remembered case first, the enumeration
    // only when its certificate fails - the same routine as the single-ego form, so a circle row's answer no longer depends on the launch form.
    if (MODE == 2 && d.warm && P.norm2 && live && !bad) ok = lmz::warm_circle<16>(W, rb, P, lane, hint_in, best);
    LMZ_CLK(3);
    if (MODE != 2 && !live && !ok) {                           // a dead row never asks for the enumeration; nothing of `best` is used
        best.cost = 0; best.id = 0; best.m = -1; best.H0 = best.H1 = 0; best.i1 = best.i2 = best.j1 = best.j2 = -1;
        best.l1 = best.l2 = best.g1 = best.g2 = 0;
        ok = true;
    }
    unsigned long long need = __ballot(!ok);                   // rows that need the enumeration (wave-uniform from here)
    if (MODE == 2) {
        need = __ballot(live && !ok);                            // the polygon row of the slot, or the circle rows whose remembered case was not accepted
        if (!live) {                                            // idle rows: nothing of `best` is used below
            best.cost = 0; best.id = 0; best.m = -1; best.H0 = best.H1 = 0; best.i1 = best.i2 = best.j1 = best.j2 = -1;
            best.l1 = best.l2 = best.g1 = best.g2 = 0;
        }
    }
#ifdef RDA_LMZ_STATS
    if (lane == 0) {
        const int nf = __popcll(need) >> 4;
        if (block == 0 && wv == 0) atomicAdd(&d.ctrl->lmz_stat[0], 1u);
        atomicAdd(&d.ctrl->lmz_stat[1], (unsigned)nf); atomicAdd(&d.ctrl->lmz_stat[2 + nf], 1u);
    }
#endif
    bool defer = false;
    if (MODE == 1) {
        defer = ((need >> (16 * row)) & 1) != 0 && !bad;
        if ((need >> (16 * row)) & 1) {     // nothing of `best` is used below (deferred, or a non-finite row): a harmless value
            best.cost = 0; best.id = 0; best.m = -1; best.H0 = best.H1 = 0; best.i1 = best.i2 = best.j1 = best.j2 = -1;
            best.l1 = best.l2 = best.g1 = best.g2 = 0;
        }
        if (defer && gl == 0 && live) {
            if (P.norm2) d.wl[d.wl_cap - 1 - atomicAdd(&d.ctrl->wlc_count, 1)] = t * GS * d.J + nl;      // circle rows: from the back (see the work-list form)
            else d.wl[atomicAdd(&d.ctrl->wl_count, 1)] = t * GS * d.J + nl;
        }
        need = 0;
    }
    if (MODE == 0) {
        // The rows of the WORKGROUP whose certificate failed are shared out over its waves: the launch time of a small grid is its
        // slowest wave, and that was a wave with two failing rows (2 - 3 % of the rows fail; ~13 us common path + ~6 us per
        // enumeration served one after the other) - with the list in LDS a workgroup's two waves take one each.  A row's result
        // does not depend on which wave enumerated it (same function on the same slab).
        struct Prm { int norm2; double px, py, cs, sn, xi0, xi1, kappa0; };
        __shared__ int nfail; __shared__ unsigned char flist[GS]; __shared__ Prm prm[GS]; __shared__ lmz::Sol sol[GS];
        if (threadIdx.x == 0) nfail = 0;
        __syncthreads();
        const bool mine = ((need >> (16 * row)) & 1) != 0;
        if (mine && gl == 0) {
            const int rid = wv * 4 + row; flist[atomicAdd(&nfail, 1)] = (unsigned char)rid;
            Prm &q = prm[rid]; q.norm2 = P.norm2; q.px = P.px; q.py = P.py; q.cs = P.cs; q.sn = P.sn; q.xi0 = P.xi0; q.xi1 = P.xi1; q.kappa0 = P.kappa0;
        }
        __syncthreads();
        const int nf = nfail;
        for (int i = wv; i < nf; i += WPB) {
            const int rid = flist[i];
            lmz::Params Pg;
            { const Prm &q = prm[rid]; Pg.E = E; Pg.R = R; Pg.norm2 = q.norm2; Pg.px = q.px; Pg.py = q.py; Pg.cs = q.cs; Pg.sn = q.sn; Pg.xi0 = q.xi0; Pg.xi1 = q.xi1;
              Pg.kappa0 = q.kappa0; Pg.ro2 = P.ro2; Pg.delta = P.delta; }
            lmz::Sol bg;
            lmz::solve_wave(wl[rid], rb, Pg, lane, bg);
            if (lane == 0) sol[rid] = bg;
        }
        __syncthreads();
        if (mine) best = sol[wv * 4 + row];
        need = 0;
#ifdef RDA_LMZ_CLK
        clk_enum = nf > 0;
        if (mine && gl == 0 && live && g_lmz_faillog) {
            const int k = atomicAdd(&g_lmz_faillog[0], 1);
            if (k < 200000) { g_lmz_faillog[1 + 3 * k] = hint_in; g_lmz_faillog[2 + 3 * k] = best.id >> 1; g_lmz_faillog[3 + 3 * k] = P.norm2; }
        }
#endif
    }
    LMZ_CLK(4);
    while (need) {
        const int g = (__ffsll((long long)need) - 1) >> 4;
        need &= ~(0xffffull << (16 * g));
        lmz::Params Pg;
        const int src = 16 * g;
        Pg.E = E; Pg.R = R; Pg.norm2 = __shfl(P.norm2, src, 64);
        Pg.px = __shfl(P.px, src, 64); Pg.py = __shfl(P.py, src, 64); Pg.cs = __shfl(P.cs, src, 64); Pg.sn = __shfl(P.sn, src, 64);
        Pg.xi0 = __shfl(P.xi0, src, 64); Pg.xi1 = __shfl(P.xi1, src, 64); Pg.kappa0 = __shfl(P.kappa0, src, 64);
        Pg.ro2 = P.ro2; Pg.delta = P.delta;
        lmz::Sol bg;
        lmz::solve_wave(wl[wv * 4 + g], rb, Pg, lane, bg);
        if (row == g) best = bg;
    }
    if (gl == 0 && live && !defer) hint_write(d, hpar, n, t, best.id >> 1);
    if (d.centre) lmz::central_normal_wave<16>(W, rb, P, lane, best);
    LMZ_CLK(5);
    bad = bad || !(isfinite(best.cost) && isfinite(best.m) && isfinite(best.H0) && isfinite(best.H1));      // uniform over the row
    // ---- fused dual / residual updates (every lane of the row holds the row's winner) ----------------
    const double znew = (d.c.accelerated ? 0.5 : 1.0) * (best.m > 0 ? best.m : 0.0);     // tie-break T2
    double res = 0;
    const bool wr = live && !bad && !defer;
    if (gl < E) {
        double v = lmz::lam_of(best, P.norm2, gl);
        res = (v - prev) * (v - prev); if (wr) d.lam[o * E + gl] = v;
    } else if (gl < E + R) {
        int j = gl - E;
        double v = lmz::mu_of(best, j);
        res = (v - prev) * (v - prev); if (wr) d.mu[o * R + j] = v;
    } else if (gl == E + R) {
        const double old = MODE == 0 ? ey.zold : d.z[zi];
        res = (znew - old) * (znew - old); if (wr) d.z[zi] = znew;
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) res += __shfl_xor(res, off, 16);
    if (gl == 0) {
        const int k = t * d.Nloc + nl;
        RowOut ro;
        bool have = false;
        if (live && bad) {
            ro = failed_row(d, k); have = true;
            store_row(d, k, ro);
            hint_write(d, hpar, n, t, -1);
            atomicAdd(&d.ctrl->lmz_fail, 1);
        } else if (wr) {
            double ax = 0, ay = 0, bl = 0, mh = 0, gx = 0, gy = 0;
            for (int i = 0; i < E; ++i) {
                double v = lmz::lam_of(best, P.norm2, i);
                ax += v * W.A[i][0]; ay += v * W.A[i][1]; bl += v * W.b[i];
            }
            for (int j = 0; j < R; ++j) {
                double v = lmz::mu_of(best, j);
                mh += v * rb.h[j]; gx += v * rb.G[j][0]; gy += v * rb.G[j][1];
            }
            const double hx = gx + P.cs * ax + P.sn * ay, hy = gy - P.sn * ax + P.cs * ay;      // Hm, :682
            const double xin0 = P.xi0 + hx, xin1 = P.xi1 + hy;                                  // :683
            d.xi[2 * o] = xin0; d.xi[2 * o + 1] = xin1;
            const double im = ax * P.px + ay * P.py - bl - mh;                                  // :659
            const double zetan = zeta + im - dbar - znew;                                       // :666
            d.zeta[zi] = zetan;
            ro.ax = ax; ro.ay = ay; ro.bl = bl; ro.c3 = mh + znew - zetan; ro.c4 = gx + xin0; ro.c5 = gy + xin1;
            ro.res = res; ro.hh = hx * hx + hy * hy; have = true;
            store_row(d, k, ro);
        }
        if (MODE == 0) {                  // this row's record for the block partial (a dead slot: zeros)
            double *rv = rowv[wv * 4 + row];
            if (have) row_record(rv, d, ro, px0, py0);
            else { rv[0] = rv[1] = rv[2] = rv[3] = rv[4] = rv[5] = 0.0; }
        }
    }
    LMZ_CLK(6);
    if (MODE == 0) {
        __syncthreads();
        block_partial(d, tb, jb, rowv, threadIdx.x);
        if (block == 0 && threadIdx.x == 0) d.ctrl->pose_ok = 1;
    }
    LMZ_CLK(7);
#ifdef RDA_LMZ_CLK
    if (MODE == 0 && clk_slot && (threadIdx.x & 63) == 0) {
        const unsigned long long tot = (unsigned long long)(clock64() - clk_in);
        clk_slot[8] += 1; clk_slot[9] += tot; if (tot > clk_slot[10]) clk_slot[10] = tot;
        clk_slot[13] = clk_wall_in; clk_slot[14] = (unsigned long long)wall_clock64();      // (last launch) start / end on the device-wide 100 MHz clock
        if (clk_enum) { clk_slot[11] += 1; clk_slot[12] += tot; }
    }
#endif
    }
#undef LMZ_CLK
}

// two builds: all registers and one wave per SIMD (no spills: the shorter critical path a single ego wants), or two
// waves per SIMD with a few spilled registers (more sub-problems in flight: what a full chip wants)
__global__ __launch_bounds__(64 * GS / 4) void k_lammuz_rows(Dev d, int it, Fin fin) { warm_kernargs<sizeof(Dev) + sizeof(int) + sizeof(Fin)>(); lammuz_body_rows<0, true>(d, blockIdx.x, gridDim.x, it, fin); }
__global__ __launch_bounds__(64 * GS / 4, 2) void k_lammuz_rows_dense(Dev d, int it, Fin fin) { lammuz_body_rows<0>(d, blockIdx.x, gridDim.x, it, fin); }
__global__ __launch_bounds__(64 * GS / 4, 3) void k_lammuz_rows_fast(Dev d, int it) { lammuz_body_rows<1>(d, blockIdx.x, gridDim.x, it, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0}); }
// (measured: the common path with four waves per SIMD and 31 spilled registers is 9 % slower.  Round 4, same-box A/B (tools/experiments/ab_so.sh): the common
// path at TWO waves per SIMD - no scratch spills - is 9 % slower at N = 2000 and 20 % slower for the 64-ego C5 fleet than at three waves with its 53 spilled
// registers, so it stays at three; the work list at ONE wave per SIMD (no spills; round 2-3: two waves, 91 spilled registers) is +1 % / +-0 %: kept)
__global__ __launch_bounds__(64 * GS / 4, 1) void k_lammuz_enum(Dev d, int it) { lammuz_body_rows<2>(d, blockIdx.x, gridDim.x, it, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0}); }

// Block partials from the STORED terms - the same row_term, the same row order as mode 0 of the packed kernel forms them in flight - and
// the tail of the step.  Runs behind every LamMuZ form that leaves its rows to more than one workgroup or launch (split launch, one
// sub-problem per wave, the per-thread interior-point kernel), after the no-obstacle launch (quirk Q9 edits terms) and whenever the
// host rewrites terms (rda_set_state, rda_reset, rda_shard_config).  One GS-slot block per GS threads, 256 / GS blocks per workgroup;
// it < 0: no tail.
constexpr int FPB = 256 / GS;          // block partials per workgroup of k_lmz_finalize
__device__ __forceinline__ void finalize_body(const Dev &d, const int block, const int nblocks, const int it, const Fin &fin)
{
#pragma clang fp contract(on)
    __shared__ double rowv[FPB][GS][6];
    if (it >= 0 && d.ctrl->stop) return;
    const int T = d.c.T, g = threadIdx.x / GS, row = threadIdx.x % GS;
    const int B = block * FPB + g, t = B / d.J, j = B - t * d.J, nl = GS * j + row;
    double *rv = rowv[g][row];
    if (B < T * d.J && nl < d.Nlive) {
        const int k = t * d.Nloc + nl;
        RowOut ro; ro.ax = coef_arr(d, d.rank, 0)[k]; ro.ay = coef_arr(d, d.rank, 1)[k]; ro.bl = coef_arr(d, d.rank, 2)[k];
        ro.c3 = coef_arr(d, d.rank, 3)[k]; ro.c4 = coef_arr(d, d.rank, 4)[k]; ro.c5 = coef_arr(d, d.rank, 5)[k];
        ro.res = coef_arr(d, d.rank, 6)[k]; ro.hh = coef_arr(d, d.rank, 7)[k];
        row_record(rv, d, ro, d.pose[4 * t], d.pose[4 * t + 1]);
    } else { rv[0] = rv[1] = rv[2] = rv[3] = rv[4] = rv[5] = 0.0; }
    __syncthreads();
    if (B < T * d.J) block_partial(d, t, j, rowv[g], row);
    if (block == 0 && threadIdx.x == 0) d.ctrl->pose_ok = 1;
}
__global__ __launch_bounds__(256) void k_lmz_finalize(Dev d, int it, Fin fin) { finalize_body(d, blockIdx.x, gridDim.x, it, fin); }

// K1, interior-point variant (lammuz_cp_device.h): one (obstacle, stage) sub-problem per thread, same fused dual / residual
// updates as lammuz_body.  A solve that does not end on the central path keeps the previous duals of its stage and makes the
// residual inf (rda_solver.py:781-793); unlike non-finite data (where everything of the stage is left alone) the xi / zeta
// updates then run with the kept duals, as the reference's do.
template <int NX, int MX> __device__ __forceinline__ void lammuz_cp_body(const Dev &d)
{
    const int T = d.c.T, E = d.c.E, R = d.c.R;
    if (d.ctrl->stop) return;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.obstacle_num == 0) {          // quirk Q9, as in lammuz_body
        if (w < T && d.rank == (d.c.N - 1) / d.Nloc) {
            const int i = w * d.Nloc + (d.c.N - 1) % d.Nloc;
            coef_arr(d, d.rank, 0)[i] = 0; coef_arr(d, d.rank, 1)[i] = 0; coef_arr(d, d.rank, 2)[i] = 0;
            coef_arr(d, d.rank, 8)[i] = coef_arr(d, d.rank, 3)[i];
        }
        return;
    }
    if (w >= d.Nlive * T) return;
    const int nl = w % d.Nlive, t = w / d.Nlive, n = d.rank * d.Nloc + nl;
    const size_t ao = ((size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0)) * E, o = drow(d, n, t + 1), zi = drow(d, n, t);
    const int k = t * d.Nloc + nl;
    double A[16], b[8];
    for (int i = 0; i < 2 * E; ++i) A[i] = d.A[ao * 2 + i];
    for (int i = 0; i < E; ++i) b[i] = d.b[ao + i];
    cpq::Problem p;
    p.E = E; p.R = R; p.cone_norm2 = d.cone[n]; p.robot_norm2 = d.c.robot_norm2; p.accelerated = d.c.accelerated;
    p.A = A; p.b = b; p.G = d.G; p.h = d.h;
    const double *ps = d.pose + 4 * t;                        // position of column t+1, heading of column t (quirk Q1)
    p.px = ps[0]; p.py = ps[1]; p.cs = ps[2]; p.sn = ps[3];
    p.xi0 = d.xi[2 * o]; p.xi1 = d.xi[2 * o + 1];
    const double zeta = d.zeta[zi], dbar = d.dis[t];
    p.kappa0 = zeta - dbar; p.ro2 = d.c.ro2; p.mu_target = d.lmz_mu;
    bool finite_in = isfinite(p.px + p.py + p.cs + p.sn + p.xi0 + p.xi1 + p.kappa0);
    for (int i = 0; i < 2 * E; ++i) finite_in = finite_in && isfinite(A[i]);
    for (int i = 0; i < E; ++i) finite_in = finite_in && isfinite(b[i]);
    if (!finite_in) {
        store_row(d, k, failed_row(d, k));
        atomicAdd(&d.ctrl->lmz_fail, 1);
        return;
    }
    cpq::Result rs;
    {
        cpq::Solver<NX, MX> sv;
        sv.run(p, rs);
    }
    const bool fail = rs.status != 0;
    double res = 0, lam[8], mu[8], znew;
    for (int i = 0; i < E; ++i) { const double old = d.lam[o * E + i]; lam[i] = fail ? old : rs.lam[i]; res += (lam[i] - old) * (lam[i] - old); }
    for (int j = 0; j < R; ++j) { const double old = d.mu[o * R + j]; mu[j] = fail ? old : rs.mu[j]; res += (mu[j] - old) * (mu[j] - old); }
    { const double old = d.z[zi]; znew = fail ? old : rs.z; res += (znew - old) * (znew - old); }
    if (!fail) {
        for (int i = 0; i < E; ++i) d.lam[o * E + i] = lam[i];
        for (int j = 0; j < R; ++j) d.mu[o * R + j] = mu[j];
        d.z[zi] = znew;
    } else atomicAdd(&d.ctrl->lmz_fail, 1);
    double ax = 0, ay = 0, bl = 0, mh = 0, gx = 0, gy = 0;
    for (int i = 0; i < E; ++i) { ax += lam[i] * A[2 * i]; ay += lam[i] * A[2 * i + 1]; bl += lam[i] * b[i]; }
    for (int j = 0; j < R; ++j) { mh += mu[j] * d.h[j]; gx += mu[j] * d.G[2 * j]; gy += mu[j] * d.G[2 * j + 1]; }
    const double hx = gx + p.cs * ax + p.sn * ay, hy = gy - p.sn * ax + p.cs * ay;      // Hm, :682
    const double xin0 = p.xi0 + hx, xin1 = p.xi1 + hy;                                  // :683
    d.xi[2 * o] = xin0; d.xi[2 * o + 1] = xin1;
    const double im = ax * p.px + ay * p.py - bl - mh;                                  // :659
    const double zetan = zeta + im - dbar - znew;                                       // :666
    d.zeta[zi] = zetan;
    RowOut ro; ro.ax = ax; ro.ay = ay; ro.bl = bl; ro.c3 = mh + znew - zetan; ro.c4 = gx + xin0; ro.c5 = gy + xin1;
    ro.res = fail ? INFINITY : res; ro.hh = hx * hx + hy * hy;
    store_row(d, k, ro);
}
__global__ __launch_bounds__(64) void k_lammuz_cp_small(Dev d) { lammuz_cp_body<16, 24>(d); }      // E, R <= 4
__global__ __launch_bounds__(64) void k_lammuz_cp_large(Dev d) { lammuz_cp_body<24, 36>(d); }      // E, R <= 8

// K1, interior-point variant, ROW-PARALLEL (lammuz_ip_device.h): one sub-problem per 16-lane row, lane i = variable i of the cone
// program, GS rows per workgroup in the launch order and with the epilogue of the packed enumeration kernel - duals, condensed terms,
// block partial, tail of the step all in this ONE launch.  A solve that does not reach the central path keeps the previous duals of
// its stage and makes the residual inf (rda_solver.py:781-793); unlike non-finite data (where everything of the stage is left alone)
// the xi / zeta updates then run with the kept duals, as the reference's do.
__device__ __forceinline__ void lammuz_ip_body(const Dev &d, const int block, const int it, const Fin &fin)
{
    constexpr int WPB = GS / 4;
    __shared__ lmz::WaveLDS wl[GS];
    __shared__ lmz::RobotLDS rb;
    __shared__ double rowv[GS][6];
    typedef rip::DevLanes L;
    const int E = d.c.E, R = d.c.R;
    if (d.ctrl->stop) return;
    int t, jb; block_of(d, block, t, jb);
    if (t < 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = lane >> 4, gl = lane & 15;
    for (int i = threadIdx.x; i < 2 * R; i += 64 * WPB) rb.G[i >> 1][i & 1] = d.G[i];
    for (int i = threadIdx.x; i < R; i += 64 * WPB) rb.h[i] = d.h[i];
    int nl = jb * GS + wv * 4 + row;
    const bool live = nl < d.Nlive;
    if (!live) nl = 0;
    const int n = d.rank * d.Nloc + nl;
    lmz::WaveLDS &W = wl[wv * 4 + row];
    const size_t ao = ((size_t)n * d.nt + (d.nt > 1 ? t + 1 : 0)) * E;
    if (gl < 2 * E) W.A[gl >> 1][gl & 1] = d.A[ao * 2 + gl];
    if (gl < E) W.b[gl] = d.b[ao + gl];
    __syncthreads();
    rip::Problem p;
    p.E = E; p.R = R; p.cone_norm2 = d.cone[n]; p.robot_norm2 = d.c.robot_norm2; p.accelerated = d.c.accelerated;
    p.A = &W.A[0][0]; p.b = W.b; p.G = &rb.G[0][0]; p.h = rb.h;
    const double *ps = d.pose + 4 * t;                        // position of column t+1, cos / sin of the heading of column t (quirk Q1)
    p.px = ps[0]; p.py = ps[1]; p.cs = ps[2]; p.sn = ps[3];
    const size_t o = drow(d, n, t + 1), zi = drow(d, n, t);
    p.xi0 = d.xi[2 * o]; p.xi1 = d.xi[2 * o + 1];
    const double zeta = d.zeta[zi], dbar = d.dis[t];
    p.kappa0 = zeta - dbar; p.ro2 = d.c.ro2; p.mu_target = d.lmz_mu;
    double prev = 0.0;                                        // this lane's dual entry before the solve: lam | mu | z
    if (gl < E) prev = d.lam[o * E + gl];
    else if (gl < E + R) prev = d.mu[o * R + gl - E];
    else if (gl == E + R) prev = d.z[zi];
    bool bad = !isfinite(p.px + p.py + p.cs + p.sn + p.xi0 + p.xi1 + p.kappa0);
    if (gl < 2 * E) bad = bad || !isfinite(W.A[gl >> 1][gl & 1]);
    if (gl < E) bad = bad || !isfinite(W.b[gl]);
    bad = ((__ballot(bad) >> (16 * row)) & 0xffffull) != 0;                      // uniform over the row
    int status = 2;
    double v = prev;
    double *const wst = d.ipw ? d.ipw + (zi * 5) * 16 : nullptr;      // this sub-problem's kept central-path point
    if (!bad) {
        rip::Solver<L> sv;
        sv.build(p);
        const bool warm = wst && d.ipf[zi] != 0;                      // (uniform over the row)
        if (warm) sv.load(wst[gl], wst[16 + gl], wst[32 + gl], wst[48 + gl], wst[64 + gl]);
        status = sv.run(p.mu_target, warm);
        if (wst && live) {
            if (status == 0) { wst[gl] = sv.x; wst[16 + gl] = sv.s.d; wst[32 + gl] = sv.z.d; wst[48 + gl] = sv.s.g; wst[64 + gl] = sv.z.g; }
            if (gl == 0) d.ipf[zi] = status == 0 ? 1 : 0;
        }
        if (status == 0) {
            v = sv.x;
            if (gl < E) { if (!p.cone_norm2 && v < 0) v = 0; }
            else if (gl < E + R) { if (!p.robot_norm2 && v < 0) v = 0; }
            else if (gl == E + R) { if (!(v > 0)) v = 0; }
            const bool fin_ok = gl > E + R || isfinite(v);
            if (((__ballot(!fin_ok) >> (16 * row)) & 0xffffull) != 0) status = 2;
        }
    }
    const bool fail = status != 0;                             // (row-uniform) previous duals kept
    if (fail) v = prev;
    const double dv = gl <= E + R ? v - prev : 0.0;
    const double res = L::rsum(dv * dv);
    const bool wr = live && !bad;
    if (wr && !fail) {
        if (gl < E) d.lam[o * E + gl] = v;
        else if (gl < E + R) d.mu[o * R + gl - E] = v;
        else if (gl == E + R) d.z[zi] = v;
    }
    // lam'A, lam'b, mu'h, G'mu with the duals in force (new, or kept): row reductions of the lanes' products
    const double la = gl < E ? v : 0.0, ma = (gl >= E && gl < E + R) ? v : 0.0;
    const int ie = gl < E ? gl : 0, jr = (gl >= E && gl < E + R) ? gl - E : 0;
    const double ax = L::rsum(la * W.A[ie][0]), ay = L::rsum(la * W.A[ie][1]), bl = L::rsum(la * W.b[ie]);
    const double mh = L::rsum(ma * rb.h[jr]), gx = L::rsum(ma * rb.G[jr][0]), gy = L::rsum(ma * rb.G[jr][1]);
    const double znew = L::bc<0>(__shfl(v, (lane & 48) + E + R, 64));           // lane E+R of the row
    if (gl == 0) {
        const int k = t * d.Nloc + nl;
        RowOut ro;
        bool have = false;
        if (live && bad) {
            ro = failed_row(d, k); have = true;
            store_row(d, k, ro);
            atomicAdd(&d.ctrl->lmz_fail, 1);
        } else if (wr) {
            const double hx = gx + p.cs * ax + p.sn * ay, hy = gy - p.sn * ax + p.cs * ay;      // Hm, :682
            const double xin0 = p.xi0 + hx, xin1 = p.xi1 + hy;                                  // :683
            d.xi[2 * o] = xin0; d.xi[2 * o + 1] = xin1;
            const double im = ax * p.px + ay * p.py - bl - mh;                                  // :659
            const double zetan = zeta + im - dbar - znew;                                       // :666
            d.zeta[zi] = zetan;
            ro.ax = ax; ro.ay = ay; ro.bl = bl; ro.c3 = mh + znew - zetan; ro.c4 = gx + xin0; ro.c5 = gy + xin1;
            ro.res = fail ? INFINITY : res; ro.hh = hx * hx + hy * hy; have = true;
            store_row(d, k, ro);
            if (fail) atomicAdd(&d.ctrl->lmz_fail, 1);
        }
        double *rv = rowv[wv * 4 + row];
        if (have) row_record(rv, d, ro, p.px, p.py);
        else { rv[0] = rv[1] = rv[2] = rv[3] = rv[4] = rv[5] = 0.0; }
    }
    __syncthreads();
    block_partial(d, t, jb, rowv, threadIdx.x);
    if (block == 0 && threadIdx.x == 0) d.ctrl->pose_ok = 1;
}
__global__ __launch_bounds__(64 * GS / 4) void k_lammuz_ip(Dev d, int it, Fin fin) { lammuz_ip_body(d, blockIdx.x, it, fin); }

struct RobotCands { unsigned char muc[40]; int nmv; double rv[28][2]; int nrv; int centre; };

// pure-function batch hook (rda_lammuz_batch)
__global__ __launch_bounds__(256) void k_lammuz_batch(int B, int E, int R, const double *A, const double *b, const int *cone,
                                                      const double *p, const double *phi, const double *G, const double *h,
                                                      const double *xi, const double *zeta, const double *dbar, double ro2,
                                                      double delta, int accelerated, double *lam, double *mu, double *z, double *cmh, long long *prof, RobotCands rc)
{
    __shared__ lmz::WaveLDS wl[4];
    __shared__ lmz::RobotLDS rb;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long t_begin = prof ? clock64() : 0;
    if (threadIdx.x < 2 * R) rb.G[threadIdx.x >> 1][threadIdx.x & 1] = G[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + R) rb.h[threadIdx.x - 64] = h[threadIdx.x - 64];
    if (threadIdx.x >= 128 && threadIdx.x < 128 + 40) rb.muc[threadIdx.x - 128] = rc.muc[threadIdx.x - 128];
    if (threadIdx.x >= 192 && threadIdx.x < 192 + 56) (&rb.rv[0][0])[threadIdx.x - 192] = (&rc.rv[0][0])[threadIdx.x - 192];
    if (threadIdx.x == 255) { rb.nmv = rc.nmv; rb.nrv = rc.nrv; }
    const int w = blockIdx.x * 4 + wv;
    const bool live = w < B;
    const int k = live ? w : 0;
    lmz::WaveLDS &W = wl[wv];
    if (lane < 2 * E) W.A[lane >> 1][lane & 1] = A[(size_t)k * E * 2 + lane];
    if (lane < E) W.b[lane] = b[(size_t)k * E + lane];
    __syncthreads();
    if (!live) return;
    lmz::Params P;
    P.E = E; P.R = R; P.norm2 = cone[k]; P.px = p[2 * k]; P.py = p[2 * k + 1];
    P.cs = cos(phi[k]); P.sn = sin(phi[k]); P.xi0 = xi[2 * k]; P.xi1 = xi[2 * k + 1];
    P.kappa0 = zeta[k] - dbar[k]; P.ro2 = ro2; P.delta = delta;
    P.prof = (prof && w == 0) ? prof : nullptr;               // wave 0 of the batch reports its phase cycles
    if (P.prof && lane == 0) prof[0] += clock64() - t_begin;
    lmz::Sol best;
    lmz::prepare_wave(W, P, lane);
    lmz::solve_wave(W, rb, P, lane, best);
    if (rc.centre) lmz::central_normal_wave<64>(W, rb, P, lane, best);
    if (P.prof && lane == 0) prof[7] += clock64() - t_begin;
    if (lane < E) lam[(size_t)k * E + lane] = lmz::lam_of(best, P.norm2, lane);
    else if (lane < E + R) mu[(size_t)k * R + lane - E] = lmz::mu_of(best, lane - E);
    else if (lane == E + R) {
        z[k] = (accelerated ? 0.5 : 1.0) * (best.m > 0 ? best.m : 0.0);
        cmh[4 * k] = best.cost; cmh[4 * k + 1] = best.m; cmh[4 * k + 2] = best.H0; cmh[4 * k + 3] = best.H1;
    }
}

// standalone su-solve hook
template <int TT> __global__ __launch_bounds__(su::NT) void k_su_hook(su::Args a) { su::Pre pre; su::Result res; su::solve<TT>(a, smem_su, pre, false, res); }

// horizons with a compile-time specialisation of the su kernel (BASELINE configs C1, north star, C5, C4); any other T runs the generic one
#define RDA_SU_DISPATCH(T, CALL) do { switch (T) { case 10: { constexpr int TT = 10; CALL; } break; case 20: { constexpr int TT = 20; CALL; } break; \
                                                    case 25: { constexpr int TT = 25; CALL; } break; case 30: { constexpr int TT = 30; CALL; } break; \
                                                    default: { constexpr int TT = 0; CALL; } break; } } while (0)

template <int TT> __global__ __launch_bounds__(su::NT) void k_su_tracked(Dev d, track::In in, double *path, int L, const double *nom_u, double *step,
                                                                          track::Out *out, unsigned long long seq);

// [N][T+1][2] / [N][T+1] <-> [T][N] transposes for the state accessors
__global__ void k_products_get(Dev d, double *a_lam, double *b_lam)
{
    const int T = d.c.T, N = d.c.N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * (T + 1); i += gridDim.x * blockDim.x) {
        int n = i / (T + 1), tt = i % (T + 1);
        double ax = 0, ay = 0, bl = 0;
        if (tt >= 1) { int r = n / d.Nloc, k = (tt - 1) * d.Nloc + (n - r * d.Nloc); ax = coef_arr(d, r, 0)[k]; ay = coef_arr(d, r, 1)[k]; bl = coef_arr(d, r, 2)[k]; }
        a_lam[2 * i] = ax; a_lam[2 * i + 1] = ay; b_lam[i] = bl;
    }
}
// rebuild every condensed term from the dual state (after rda_set_state); k_lmz_finalize then rebuilds the block partials
__global__ void k_products_set(Dev d, const double *a_lam, const double *b_lam)
{
    const int T = d.c.T, N = d.c.N, R = d.c.R;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * T; i += gridDim.x * blockDim.x) {
        int n = i / T, t = i % T, r = n / d.Nloc, k = t * d.Nloc + (n - r * d.Nloc);
        const size_t o = drow(d, n, t + 1), zi = drow(d, n, t), oh = (size_t)n * (T + 1) + t + 1;     // oh: the caller's [N][T+1] layout
        if (a_lam) { coef_arr(d, r, 0)[k] = a_lam[2 * oh]; coef_arr(d, r, 1)[k] = a_lam[2 * oh + 1]; }
        if (b_lam) coef_arr(d, r, 2)[k] = b_lam[oh];
        double mh = 0, gx = 0, gy = 0;
        for (int j = 0; j < R; ++j) { double v = d.mu[o * R + j]; mh += v * d.h[j]; gx += v * d.G[2 * j]; gy += v * d.G[2 * j + 1]; }
        coef_arr(d, r, 3)[k] = mh + d.z[zi] - d.zeta[zi];
        coef_arr(d, r, 4)[k] = gx + d.xi[2 * o]; coef_arr(d, r, 5)[k] = gy + d.xi[2 * o + 1];
        coef_arr(d, r, 8)[k] = coef_arr(d, r, 2)[k] + coef_arr(d, r, 3)[k];
        coef_arr(d, r, 6)[k] = 0; coef_arr(d, r, 7)[k] = 0;
    }
}
// reset() of the reference (rda_solver.py:1060-1068) clears the lam'A / lam'b products only - the duals stay (quirk Q6)
__global__ void k_reset(Dev d)
{
    const int T = d.c.T;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.P * d.Nloc * T; i += gridDim.x * blockDim.x) {     // incl. padding slots (zero anyway)
        int r = i / (d.Nloc * T), k = i % (d.Nloc * T);
        coef_arr(d, r, 0)[k] = 0; coef_arr(d, r, 1)[k] = 0; coef_arr(d, r, 2)[k] = 0;
        coef_arr(d, r, 8)[k] = coef_arr(d, r, 3)[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.ctrl->su_last = 99; d.ctrl->su_probe = 0; d.ctrl->prev_unconv = 0; d.ctrl->su_hardlike = 0; d.ctrl->spec_credit = 0; d.ctrl->land_hard = 0; d.ctrl->land_easy = 0; d.ctrl->land_rho_prev = 0; d.ctrl->blind_ok = 0; }      // solver history of the handle
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < su::NC * T; i += blockDim.x) d.su_lam_keep[i] = 0;
    if (d.ipf) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.c.N * T; i += gridDim.x * blockDim.x) d.ipf[i] = 0;
}
// the same rebuild for every shard of a handle (the host rewrote terms: all P chunks are local copies)
__global__ __launch_bounds__(256) void k_lmz_finalize_all(Dev d)
{
    Dev q = d; q.rank = blockIdx.y; q.Nlive = d.c.N - q.rank * d.Nloc; if (q.Nlive > d.Nloc) q.Nlive = d.Nloc; if (q.Nlive < 0) q.Nlive = 0;
    finalize_body(q, blockIdx.x, gridDim.x, -1, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0});
}

// ---- rda_opts::duals_follow: the dual state moves with its obstacle when the device pipeline re-binds the slots ----------------------
// One workgroup per slot s.  map[s] = the slot that held, at the previous staging, the raw-scene entry slot s holds now (-1: it was in
// no slot -> initial duals, zeros).  Padding slots (s >= used: copies of the last obstacle, quirk Q3) keep their own state for as long
// as they stay padding slots.  The rows (stage, slot) of lam | mu | xi (T+1 stages) and z | zeta (T stages) of a slot that changes hands
// are gathered through the map into `tmp` (same layouts, one after the other) and written back by a second launch: the state never
// aliases itself.
__global__ __launch_bounds__(64) void k_follow_gather(Dev d, const int *now, int used, const int *prev, int prev_used, int *map, double *tmp)
{
    const int T = d.c.T, N = d.c.N, E = d.c.E, R = d.c.R, sl = blockIdx.x, lane = threadIdx.x;
    int m = -1;
    if (sl < used) {
        const int want = now[sl];
        for (int j = lane; j < prev_used; j += 64) if (prev[j] == want) m = j;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    } else m = sl >= prev_used ? sl : -1;
    if (lane == 0) map[sl] = m;
    if (m == sl) return;                                      // the slot kept its obstacle
    const size_t rows = (size_t)(T + 1) * N;
    double *tl = tmp, *tm = tl + rows * E, *tx = tm + rows * R, *tz = tx + rows * 2, *tzt = tz + (size_t)T * N;
    for (int t = lane; t <= T; t += 64) {
        const size_t i = (size_t)t * N + sl, o = (size_t)t * N + (m >= 0 ? m : 0);
        for (int e = 0; e < E; ++e) tl[i * E + e] = m >= 0 ? d.lam[o * E + e] : 0.0;
        for (int j = 0; j < R; ++j) tm[i * R + j] = m >= 0 ? d.mu[o * R + j] : 0.0;
        tx[2 * i] = m >= 0 ? d.xi[2 * o] : 0.0; tx[2 * i + 1] = m >= 0 ? d.xi[2 * o + 1] : 0.0;
        if (t < T) { tz[i] = m >= 0 ? d.z[o] : 0.0; tzt[i] = m >= 0 ? d.zeta[o] : 0.0; }
    }
}
// map == nullptr: the first staging of a handle, only the binding is recorded
__global__ __launch_bounds__(64) void k_follow_back(Dev d, const int *map, const double *tmp, const int *now, int used, int *prev)
{
    const int T = d.c.T, N = d.c.N, E = d.c.E, R = d.c.R, sl = blockIdx.x, lane = threadIdx.x;
    if (lane == 0 && sl < used) prev[sl] = now[sl];
    if (!map || map[sl] == sl) return;
    const size_t rows = (size_t)(T + 1) * N;
    const double *tl = tmp, *tm = tl + rows * E, *tx = tm + rows * R, *tz = tx + rows * 2, *tzt = tz + (size_t)T * N;
    for (int t = lane; t <= T; t += 64) {
        const size_t i = (size_t)t * N + sl;
        for (int e = 0; e < E; ++e) d.lam[i * E + e] = tl[i * E + e];
        for (int j = 0; j < R; ++j) d.mu[i * R + j] = tm[i * R + j];
        d.xi[2 * i] = tx[2 * i]; d.xi[2 * i + 1] = tx[2 * i + 1];
        if (t < T) {
            d.z[i] = tz[i]; d.zeta[i] = tzt[i];
            if (d.ipf) d.ipf[i] = 0;                              // interior-point mode: the kept central point belonged to another obstacle (cold start)
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct rda_handle {
    Dev d;
    rda_opts opts;                                        // as handed to rda_create_opts (copied)
    hipStream_t stream;
    size_t su_lds;
    // staging
    double *h_stage_A, *h_stage_b; int *h_stage_cone;    // pinned, N slots
    double *h_step;                                       // pinned: nom_s | nom_u | ref | speed
    double *d_step;                                       // device copy of the above (slot 0 of the step path)
    double *d_out_u, *d_out_s; rda_info *d_info;          // result slot of the step path
    double *h_out; rda_info *h_info;                      // pinned
    unsigned long long res_seq; int zero_copy;           // result mirror written by k_finish (see there); RDA_ZERO_COPY=0: D2H copy + stream sync
    unsigned long long *h_verdict = nullptr; unsigned long long vseq = 0;   // pinned early-stop verdict word of the su launches (sharded handles with a communicator)
    // trace path
    int K; double *d_tr_s, *d_tr_u, *d_tr_ref, *d_tr_speed, *d_tr_out_u, *d_tr_out_s; rda_info *d_tr_info;
    // obstacle-shard exchange (RCCL, resolved lazily with dlopen so that single-GPU use has no rccl dependency)
    void *nccl_lib, *comm;
    int (*p_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
    int (*p_comm_destroy)(void *);
    // a tick opened by rda_tracked_begin and not yet closed by rda_tracked_finish
    int pending, pending_scene; const double *pending_in_u;
    int tick_stages, tick_has_event;     // see rda_tracked_begin
    int fuse_track; size_t su_trk_lds; unsigned long long trk_seq;
    int lmz_split;                        // dense LamMuZ grids as two launches: common path, then the deferred rows (RDA_LMZ_SPLIT=0: one fused kernel)
    int early_finish;                     // the su launch that detects the early stop writes the result slot (RDA_EARLY_FINISH=0: k_finish does)    // k_su_tracked (RDA_FUSE_TRACK=0: k_track and k_su as two launches)
    hipStream_t stream2; hipEvent_t ev_tick, ev_scene; int scene_on_s2;   // in-tick scene staging runs beside the first su-problem
    // timing
    int timing; std::vector<hipEvent_t> ev[3]; size_t ev_used[3];      // 0 LamMuZ launches, 1 su launches, 2 shard all-gathers
    // device-side obstacle pipeline (rda_upload_scene): scene description and scratch, grown on demand
    int sc_cap; int *d_sc_sel; double *d_sc_blk, *d_sc_key;     // d_sc_blk mirrors the pinned block h_sc (ONE H2D copy per upload)
    scene::Args sc_args; int sc_n;                              // the resident raw scene as the conversion kernels were last given it (sc_n = 0: none)
    void *h_sc; size_t h_sc_bytes;
    // device-side pre_process (rda_upload_path / rda_step_tracked)
    double *d_path; int path_len; track::Out *d_trk, *h_trk;
    int dense_from;          // grids above this many workgroups use the dense form of the LamMuZ launch (rda_opts::lmz_dense_from)
    int ip_rows;             // interior-point mode runs the row-parallel kernel (shape allows it and rda_opts::lmz_ip_rows)
    int admm_it;             // host-driven ADMM pieces (rda_admm_*): the iteration rda_admm_su was last called with
    int stepped;             // a step has been queued on this handle (sharded handles: rda_reset / rda_set_state are refused from then on)
    // rda_opts::duals_follow: slot -> raw-scene entry of the staging the dual state is arranged by (prev_used = -1: none yet)
    int follow; int *d_prev_sel, *d_follow_map; double *d_follow_tmp; int prev_used;
};

static void dev_free(void *p) { if (p) (void)hipFree(p); }

extern "C" const char *rda_strerror(int code)
{
    switch (code) {
        case RDA_OK: return "ok";
        case RDA_ERR_ARG: return "invalid argument";
        case RDA_ERR_UNSUPPORTED: return "unsupported configuration (E/R/T above the compiled limits, non-canonical circle obstacle, interior-point mode in a fleet)";
        case RDA_ERR_HIP: return "HIP runtime error";
        case RDA_ERR_NODEVICE: return "no HIP device";
        default: return code > 0 ? "soft status" : "unknown error";
    }
}
extern "C" int rda_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" int rda_set_device(int dev) { HIPCHK(hipSetDevice(dev)); return RDA_OK; }

static size_t res_doubles(size_t T) { return 2 * T + 3 * (T + 1) + 8; }     // u | s | info (4) | track::Out (2) | non-convex count | sequence word
static_assert(9 * su::NT >= track::LDS_DOUBLES, "TrackedRefWait runs track::run in the su solve's `part` scratch");

template <typename Tp> static int dalloc(Tp **p, size_t n)
{
    HIPCHK(hipMalloc((void **)p, n * sizeof(Tp)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(Tp)));
    return 0;
}

// Solver options: library defaults, then the RDA_* environment overrides (experiments and A/B runs; read HERE only)
extern "C" void rda_opts_init(rda_opts *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->lmz_mode = 0; o->tie_centre = 1; o->lmz_mu = 1e-6;
    o->su_tol[0] = 1e-9; o->su_tol[1] = 1e-10; o->su_tol[2] = 1e-11; o->su_tol_early[0] = o->su_tol_early[1] = o->su_tol_early[2] = 0; o->su_hard_warm[0] = 1.0; o->su_hard_warm[1] = 1e-3;
    o->lmz_warm = 1; o->lmz_rows = 1; o->lmz_dense_from = 256; o->lmz_split = 1; o->lmz_ip_rows = 1; o->lmz_ip_warm = 1;
    o->su_pre = 1; o->su_light = 1; o->su_warm_first = 1; o->su_warm_cap = 30; o->su_easy_max = 2; o->su_easy_nopred = 1;
    o->su_cold_from = 7; o->su_cold_probe = 8; o->zero_copy = 1; o->early_finish = 1; o->fuse_track = 1; o->su_prof = 0; o->su_split = 1; o->duals_follow = 0; o->su_accept = 1; o->su_first_attempt = 0;
    o->su_land_blind_from = 1; o->su_land_first = 2; o->su_land = 1; o->su_land_tol[0] = 1e-3; o->su_land_tol[1] = 1e-4; o->su_land_tol[2] = 1e-5; o->su_land_rho = 1e4;
    o->su_warm[0] = 1e-3; o->su_warm[1] = 1e-3; o->su_warm_endgame[0] = 0.9999; o->su_warm_endgame[1] = 1e-5; o->su_warm_clip = 0.01;
    // easy start = the previous solution ITSELF: slack floor, barrier parameter and clip margin below the stop tolerances (1e-12 against
    // mu <= 1e-11 (1 + |grad|), |r_p| <= 1e-10), so that the stop test can accept the start when the new su-problem's optimality
    // conditions already hold there (a converging ADMM in a static scene: 57 % of the north-star solves) - zero Newton steps
    { const double ez[5] = {1e-12, 1e-12, 1e-12, 0.999999, 1e-7}; for (int i = 0; i < 5; ++i) o->su_easy[i] = ez[i]; }
    // (No environment overrides here since round 5: a library call has no process-global configuration.  The RDA_* switches of the A/B tools and
    // tests are applied by the Python host package - rda_planner_amd.rda_solver.hip_options - to the struct it hands to rda_create_opts.)
    if (o->su_cold_probe < 1) o->su_cold_probe = 1;
}

// mu support candidates of a polygon robot: pairs whose intersection is a vertex of the robot, then the non-null rows, then
// the empty support (same rule and order as the lam lists built per wave in lmz::solve_wave)
static int robot_candidates(int R, const double *G, const double *h, unsigned char *out, double (*rv)[2], int *nrv)
{
    int n = 0, p = 0;
    for (int j1 = 0; j1 < R; ++j1) for (int j2 = j1 + 1; j2 < R; ++j2, ++p) {
        const double a00 = G[2 * j1], a01 = G[2 * j1 + 1], a10 = G[2 * j2], a11 = G[2 * j2 + 1];
        const double det = a00 * a11 - a01 * a10;
        if (det == 0 || !(det * det > 1e-24 * (a00 * a00 + a01 * a01) * (a10 * a10 + a11 * a11))) continue;
        const double wx = h[j1] * a11 - a01 * h[j2], wy = a00 * h[j2] - h[j1] * a10, sg = det > 0 ? 1.0 : -1.0, ad = fabs(det);
        bool ok = true;
        for (int k = 0; k < R; ++k) {
            const double viol = sg * (G[2 * k] * wx + G[2 * k + 1] * wy - h[k] * det);
            if (viol > 1e-9 * (ad + fabs(h[k]) * ad + fabs(G[2 * k] * wx) + fabs(G[2 * k + 1] * wy))) ok = false;
        }
        if (ok) { rv[n][0] = wx / det; rv[n][1] = wy / det; out[n++] = (unsigned char)(1 + R + p); }
    }
    *nrv = n;
    for (int j = 0; j < R; ++j) if (G[2 * j] != 0 || G[2 * j + 1] != 0) out[n++] = (unsigned char)(1 + j);
    out[n++] = 0;
    return n;
}

static int create_impl(const rda_cfg *cfg, const rda_opts *opts, const double *G, const double *h, rda_handle **out, rda_handle **partial);
static int terms_rebuild(rda_handle *H);
extern "C" void rda_destroy(rda_handle *H);

extern "C" int rda_create_opts(const rda_cfg *cfg, const rda_opts *opts, const double *G, const double *h, rda_handle **out)
{
    rda_handle *partial = nullptr;                      // a failure half-way releases what was built (rda_destroy takes partial handles)
    rda_opts def;
    if (!opts) { rda_opts_init(&def); opts = &def; }
    const int rc = create_impl(cfg, opts, G, h, out, &partial);
    if (rc != RDA_OK && partial) { rda_destroy(partial); if (out) *out = nullptr; }
    return rc;
}
extern "C" int rda_create(const rda_cfg *cfg, const double *G, const double *h, rda_handle **out) { return rda_create_opts(cfg, nullptr, G, h, out); }

static int create_impl(const rda_cfg *cfg, const rda_opts *opts, const double *G, const double *h, rda_handle **out, rda_handle **partial)
{
    if (!cfg || !G || !h || !out) return RDA_ERR_ARG;
    if (cfg->robot_norm2 && cfg->R < 2) return RDA_ERR_UNSUPPORTED;
    if (cfg->E < 1 || cfg->E > RDA_EMAX || cfg->R < 1 || cfg->R > RDA_RMAX || cfg->T < 1 || cfg->T > RDA_TMAX || cfg->N < 1) return RDA_ERR_UNSUPPORTED;
    if (cfg->E + cfg->R + 1 > 64) return RDA_ERR_UNSUPPORTED;
    if (rda_device_count() < 1) return RDA_ERR_NODEVICE;
    rda_handle *H = new rda_handle();
    *partial = H;
    memset(&H->d, 0, sizeof(Dev));
    H->d.c = *cfg; H->d.nt = 1; H->d.obstacle_num = 0; H->K = 0; H->timing = 0;
    H->opts = *opts;
    const rda_opts &o = H->opts;
    H->d.warm = o.lmz_warm; H->dense_from = o.lmz_dense_from;
    H->d.rows = (cfg->E + cfg->R + 1 <= 16) && o.lmz_rows != 0;
    H->d.nmv = robot_candidates(cfg->R, G, h, H->d.muc, H->d.rv, &H->d.nrv);
    H->d.centre = o.tie_centre ? 1 : 0;
    H->d.lmz_mode = (o.lmz_mode || cfg->robot_norm2) ? 1 : 0; H->d.lmz_mu = o.lmz_mu > 0 ? o.lmz_mu : 1e-6;      // the enumeration has no norm2-robot candidates
    H->d.su_warm_wfl = o.su_warm[0]; H->d.su_warm_mu0 = o.su_warm[1]; H->d.su_warm_cap = o.su_warm_cap; H->d.su_warm_first = o.su_warm_first;
    H->d.su_hard_wfl = o.su_hard_warm[0]; H->d.su_hard_mu0 = o.su_hard_warm[0] > 0 ? o.su_hard_warm[1] : 0.0;
    H->d.su_warm_tau = o.su_warm_endgame[0]; H->d.su_warm_sig = o.su_warm_endgame[1]; H->d.su_warm_clip = o.su_warm_clip;
    for (int i = 0; i < 5; ++i) H->d.su_easy[i] = o.su_easy[i];
    H->d.su_easy_max = o.su_easy_max; H->d.su_easy_nopred = o.su_easy_nopred;
    H->d.su_pre = o.su_pre;
    H->d.su_cold_from = o.su_cold_from; H->d.su_cold_probe = o.su_cold_probe < 1 ? 1 : o.su_cold_probe;
    H->d.su_light = o.su_light; H->d.su_split = o.su_split; H->d.su_accept = o.su_accept; H->d.su_first_attempt = o.su_first_attempt;
    H->d.su_land_first = o.su_land_first < 0 ? 0 : (o.su_land_first > 2 ? 2 : o.su_land_first);
    H->d.su_land_blind_from = o.su_land_blind_from;
    H->d.su_land = o.su_land ? 1 : 0; H->d.su_land_rho = o.su_land_rho > 0 ? o.su_land_rho : 1e4;
    { const bool ok = o.su_land_tol[0] > 0 && o.su_land_tol[1] > 0 && o.su_land_tol[2] > 0; const double dflt[3] = {1e-3, 1e-4, 1e-5}; for (int i = 0; i < 3; ++i) H->d.su_land_tol[i] = ok ? o.su_land_tol[i] : dflt[i]; }
    H->follow = o.duals_follow != 0; H->prev_used = -1; H->d_prev_sel = nullptr; H->d_follow_map = nullptr; H->d_follow_tmp = nullptr;
    for (int i = 0; i < 3; ++i) H->d.su_tol[i] = o.su_tol[i] > 0 ? o.su_tol[i] : (i == 0 ? 1e-9 : (i == 1 ? 1e-10 : 1e-11));
    { const bool on = o.su_tol_early[0] > 0 && o.su_tol_early[1] > 0 && o.su_tol_early[2] > 0; for (int i = 0; i < 3; ++i) H->d.su_tol_early[i] = on ? o.su_tol_early[i] : 0.0; }
    H->nccl_lib = nullptr; H->comm = nullptr; H->p_allgather = nullptr; H->p_comm_destroy = nullptr; H->ev_used[0] = H->ev_used[1] = H->ev_used[2] = 0;
    H->d_tr_s = H->d_tr_u = H->d_tr_ref = H->d_tr_speed = H->d_tr_out_u = H->d_tr_out_s = nullptr; H->d_tr_info = nullptr;
    const size_t T = cfg->T, N = cfg->N, E = cfg->E, R = cfg->R;
    HIPCHK(hipStreamCreate(&H->stream));
    HIPCHK(hipStreamCreate(&H->stream2));
    HIPCHK(hipEventCreateWithFlags(&H->ev_tick, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&H->ev_scene, hipEventDisableTiming));
    Dev &d = H->d;
    int rc = 0;
    rc |= dalloc(&d.G, 2 * R); rc |= dalloc(&d.h, R);
    rc |= dalloc(&d.A, N * (T + 1) * E * 2); rc |= dalloc(&d.b, N * (T + 1) * E); rc |= dalloc(&d.cone, N);
    rc |= dalloc(&d.wl, (N + GS) * T); d.wl_cap = (int)((N + GS) * T); d.src_cap = (int)(4 * N + 256); d.hint_stride = (int)N + d.src_cap; d.hint_len = d.hint_stride * (int)T; d.slot_src = nullptr; d.src_used = 0;
    rc |= dalloc(&d.hint, 2 * (size_t)d.hint_len); d.oc_ls = (1 + (int)E + (int)E * ((int)E - 1) / 2 + 3) & ~3; d.oc_vs = (int)E * ((int)E - 1) > 2 ? (int)E * ((int)E - 1) : 2;
    rc |= dalloc(&d.oc_lamc, N * (T + 1) * d.oc_ls); rc |= dalloc(&d.oc_vtx, N * (T + 1) * d.oc_vs); rc |= dalloc(&d.oc_cnt, N * (T + 1) * 2);
    rc |= dalloc(&d.lam, N * (T + 1) * E); rc |= dalloc(&d.mu, N * (T + 1) * R); rc |= dalloc(&d.z, N * T);
    rc |= dalloc(&d.xi, N * (T + 1) * 2); rc |= dalloc(&d.zeta, N * T); rc |= dalloc(&d.dis, T);
    d.P = 1; d.rank = 0; d.Nloc = (int)N; d.Nlive = (int)N; d.J = (int)((N + GS - 1) / GS); d.chunk = chunk_doubles((int)T, (int)N); d.lchunk = lchunk_doubles((int)T, (int)N);
    rc |= dalloc(&d.coef, d.chunk); rc |= dalloc(&d.coefL, d.lchunk);
    rc |= dalloc(&d.s, 3 * (T + 1)); rc |= dalloc(&d.u, 2 * T); rc |= dalloc(&d.pose, 4 * T);
    rc |= dalloc(&d.ctrl, 1);
    rc |= dalloc(&d.su_lam_keep, 10 * T);
    if (H->follow) {
        rc |= dalloc(&H->d_prev_sel, N); rc |= dalloc(&H->d_follow_map, N);
        rc |= dalloc(&H->d_follow_tmp, N * (T + 1) * (E + R + 2) + 2 * N * T);
    }
#if defined(SU_TRACE)
    if (o.su_prof) rc |= dalloc(&d.su_prof, su::PROF_WORDS);                   // 16 + the per-wave event trace of the LAST su launch (tools/su_trace.py)
#elif defined(SU_PROF) || defined(SU_FINE)
    if (o.su_prof) rc |= dalloc(&d.su_prof, 16);
#else
    if (o.su_prof) { rda_destroy(H); *partial = nullptr; return RDA_ERR_UNSUPPORTED; }     // phase counters: profiling builds only (-DSU_PROF / -DSU_FINE, tools/su_phase_profile.py)
#endif
    const size_t step_n = 3 * (T + 1) + 2 * T + 3 * (T + 1) + 1;
    rc |= dalloc(&H->d_step, step_n);
    // result block, identical on the device and in pinned memory: u [2T] | s [3(T+1)] | rda_info (4 doubles) | track::Out (4 doubles):
    // ONE copy back per step
    rc |= dalloc(&H->d_out_u, res_doubles(T));
    if (!rc) { H->d_out_s = H->d_out_u + 2 * T; H->d_info = (rda_info *)(H->d_out_s + 3 * (T + 1)); H->d_trk = (track::Out *)(H->d_out_s + 3 * (T + 1) + 4); }
    if (rc) { rda_destroy(H); *partial = nullptr; return RDA_ERR_HIP; }
    { const int hard = 99; HIPCHK(hipMemcpy(&d.ctrl->su_last, &hard, sizeof(int), hipMemcpyHostToDevice)); }     // no su history yet
    HIPCHK(hipMemcpy(d.G, G, 2 * R * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.h, h, R * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> ones(T, 1.0);                                       // para_dis init, rda_solver.py:119
    HIPCHK(hipMemcpy(d.dis, ones.data(), T * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> cn(N, 1);                                              // para_cone init, rda_solver.py:158
    HIPCHK(hipMemcpy(d.cone, cn.data(), N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipHostMalloc((void **)&H->h_stage_A, N * (T + 1) * E * 2 * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&H->h_stage_b, N * (T + 1) * E * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&H->h_stage_cone, N * sizeof(int)));
    HIPCHK(hipHostMalloc((void **)&H->h_step, step_n * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&H->h_out, res_doubles(T) * sizeof(double)));
    memset(H->h_out, 0, res_doubles(T) * sizeof(double)); H->res_seq = 0; H->zero_copy = o.zero_copy;
    H->h_info = (rda_info *)(H->h_out + 2 * T + 3 * (T + 1)); H->h_trk = (track::Out *)(H->h_out + 2 * T + 3 * (T + 1) + 4);
    H->su_lds = su::lds_bytes((int)T);
    RDA_SU_DISPATCH((int)T, HIPCHK(hipFuncSetAttribute((const void *)k_su<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)H->su_lds)));
    HIPCHK(hipFuncSetAttribute((const void *)k_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)H->su_lds));
    H->su_trk_lds = H->su_lds > track::LDS_DOUBLES * sizeof(double) ? H->su_lds : track::LDS_DOUBLES * sizeof(double);
    RDA_SU_DISPATCH((int)T, HIPCHK(hipFuncSetAttribute((const void *)k_su_tracked<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)H->su_trk_lds)));
    H->fuse_track = o.fuse_track; H->tick_stages = 0; H->tick_has_event = 0; H->trk_seq = 0;
    H->early_finish = o.early_finish;
    H->lmz_split = o.lmz_split;
    H->ip_rows = o.lmz_ip_rows && rip::fits(cfg->E, cfg->R, cfg->E >= 3, cfg->robot_norm2, cfg->accelerated); H->admm_it = 0;
    if (H->d.lmz_mode && H->ip_rows && o.lmz_ip_warm) { if (dalloc(&H->d.ipw, N * T * 80) || dalloc(&H->d.ipf, N * T)) return RDA_ERR_HIP; }
    // the terms of a fresh handle are all zero: their block partials (zero sums, every slot NEAR: a = 0 puts the hinge at -d < 0)
    int rcf = terms_rebuild(H);
    if (rcf != RDA_OK) return rcf;
    HIPCHK(hipStreamSynchronize(H->stream));
    *out = H;
    return RDA_OK;
}

extern "C" void rda_destroy(rda_handle *H)
{
    if (!H) return;
    if (H->stream) (void)hipStreamSynchronize(H->stream);
    if (H->comm && H->p_comm_destroy) H->p_comm_destroy(H->comm);
    Dev &d = H->d;
    void *ptrs[] = { d.wl, d.hint, d.oc_lamc, d.oc_vtx, d.oc_cnt, d.G, d.h, d.A, d.b, d.cone, d.lam, d.mu, d.z, d.xi, d.zeta, d.dis, d.coef, d.coefL,
                     d.s, d.u, d.pose, d.su_prof, d.ipw, d.ipf, d.ctrl, d.su_lam_keep, H->d_step, H->d_out_u,
                     H->d_tr_s, H->d_tr_u, H->d_tr_ref, H->d_tr_speed, H->d_tr_out_u, H->d_tr_out_s, H->d_tr_info,
                     H->d_sc_sel, H->d_sc_blk, H->d_sc_key, H->d_path, H->d_prev_sel, H->d_follow_map, H->d_follow_tmp };
    for (void *p : ptrs) dev_free(p);
    if (H->h_stage_A) (void)hipHostFree(H->h_stage_A);
    if (H->h_stage_b) (void)hipHostFree(H->h_stage_b);
    if (H->h_stage_cone) (void)hipHostFree(H->h_stage_cone);
    if (H->h_step) (void)hipHostFree(H->h_step);
    if (H->h_out) (void)hipHostFree(H->h_out);
    if (H->h_sc) (void)hipHostFree(H->h_sc);
    if (H->h_verdict) (void)hipHostFree(H->h_verdict);
    for (int w = 0; w < 3; ++w) for (hipEvent_t e : H->ev[w]) (void)hipEventDestroy(e);
    if (H->stream2) { (void)hipStreamSynchronize(H->stream2); (void)hipStreamDestroy(H->stream2); }
    if (H->ev_tick) (void)hipEventDestroy(H->ev_tick);
    if (H->ev_scene) (void)hipEventDestroy(H->ev_scene);
    if (H->stream) (void)hipStreamDestroy(H->stream);
    delete H;
}

extern "C" int rda_set_adjust(rda_handle *H, double slack_gain, double max_sd, double min_sd, double ro1, double ro2)
{
    if (!H) return RDA_ERR_ARG;
    H->d.c.slack_gain = slack_gain; H->d.c.max_sd = max_sd; H->d.c.min_sd = min_sd; H->d.c.ro1 = ro1; H->d.c.ro2 = ro2;
    return RDA_OK;
}

// block partials (sums, near masks) of every local chunk from the stored terms: after the host rewrote terms
static int terms_rebuild(rda_handle *H)
{
    const Dev &d = H->d;
    const int nb = (d.c.T * d.J + FPB - 1) / FPB;
    hipLaunchKernelGGL(k_lmz_finalize_all, dim3(nb, d.P), dim3(256), 0, H->stream, d);
    HIPCHK(hipGetLastError());
    return RDA_OK;
}

// Obstacle shards: the local term arrays (hinge offsets, G'mu + xi) of the OTHER ranks' slots are never populated on this rank, so the
// terms of remote slots cannot be rebuilt here once the ranks have stepped - a reset / state rewrite would leave every rank with a
// different su-problem (and different early-stop verdicts: the next all-gather would hang).  Refused from the first step on.
static inline bool shard_frozen(const rda_handle *H) { return H->d.P > 1 && H->stepped; }
extern "C" int rda_reset(rda_handle *H)
{
    if (!H) return RDA_ERR_ARG;
    if (shard_frozen(H)) return RDA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_reset, dim3(64), dim3(256), 0, H->stream, H->d);
    HIPCHK(hipGetLastError());
    int rc = terms_rebuild(H);
    if (rc != RDA_OK) return rc;
    HIPCHK(hipStreamSynchronize(H->stream));
    return RDA_OK;
}

extern "C" int rda_get_su_history_n(rda_handle *H, int32_t *hist, int n_hist, double *lam_keep)
{
    if (!H || n_hist < 0) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    if (hist && n_hist > 0) {
        Ctrl c;
        HIPCHK(hipMemcpy(&c, H->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
        const int32_t all[RDA_SU_HISTORY_INTS] = { c.su_last, c.su_probe, c.prev_unconv, c.su_hardlike, c.spec_credit, c.land_hard, c.land_easy, c.blind_ok };
        for (int i = 0; i < n_hist; ++i) hist[i] = i < RDA_SU_HISTORY_INTS ? all[i] : 0;      // (entries a later version may add read as 0 here)
    }
    if (lam_keep) HIPCHK(hipMemcpy(lam_keep, H->d.su_lam_keep, (size_t)su::NC * H->d.c.T * sizeof(double), hipMemcpyDeviceToHost));
    return RDA_OK;
}
extern "C" int rda_set_su_history_n(rda_handle *H, const int32_t *hist, int n_hist, const double *lam_keep)
{
    if (!H || n_hist < 0) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    if (hist) {
        int *dst[RDA_SU_HISTORY_INTS] = { &H->d.ctrl->su_last, &H->d.ctrl->su_probe, &H->d.ctrl->prev_unconv, &H->d.ctrl->su_hardlike, &H->d.ctrl->spec_credit, &H->d.ctrl->land_hard, &H->d.ctrl->land_easy, &H->d.ctrl->blind_ok };
        for (int i = 0; i < n_hist && i < RDA_SU_HISTORY_INTS; ++i) HIPCHK(hipMemcpy(dst[i], &hist[i], sizeof(int), hipMemcpyHostToDevice));   // entries the caller does not have keep their value
    }
    if (lam_keep) HIPCHK(hipMemcpy(H->d.su_lam_keep, lam_keep, (size_t)su::NC * H->d.c.T * sizeof(double), hipMemcpyHostToDevice));
    return RDA_OK;
}
// the forms without a count: RDA_SU_HISTORY_INTS (= 8 since round 6; round 5: 4, round 4: 2) entries - a caller built against an older header must use the _n forms
extern "C" int rda_get_su_history(rda_handle *H, int32_t *hist, double *lam_keep) { return rda_get_su_history_n(H, hist, RDA_SU_HISTORY_INTS, lam_keep); }
extern "C" int rda_set_su_history(rda_handle *H, const int32_t *hist, const double *lam_keep) { return rda_set_su_history_n(H, hist, RDA_SU_HISTORY_INTS, lam_keep); }
extern "C" int rda_lmz_history_doubles(rda_handle *H) { return !H ? RDA_ERR_ARG : (H->d.ipw ? 80 * H->d.c.N * H->d.c.T : 0); }
extern "C" int rda_get_lmz_history(rda_handle *H, double *points, int32_t *valid)
{
    if (!H) return RDA_ERR_ARG;
    if (!H->d.ipw) return RDA_ERR_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(H->stream));
    const size_t nt = (size_t)H->d.c.N * H->d.c.T;
    if (points) HIPCHK(hipMemcpy(points, H->d.ipw, nt * 80 * sizeof(double), hipMemcpyDeviceToHost));
    if (valid) HIPCHK(hipMemcpy(valid, H->d.ipf, nt * sizeof(int), hipMemcpyDeviceToHost));
    return RDA_OK;
}
extern "C" int rda_set_lmz_history(rda_handle *H, const double *points, const int32_t *valid)
{
    if (!H) return RDA_ERR_ARG;
    if (!H->d.ipw) return RDA_ERR_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(H->stream));
    const size_t nt = (size_t)H->d.c.N * H->d.c.T;
    if (points) HIPCHK(hipMemcpy(H->d.ipw, points, nt * 80 * sizeof(double), hipMemcpyHostToDevice));
    if (valid) HIPCHK(hipMemcpy(H->d.ipf, valid, nt * sizeof(int), hipMemcpyHostToDevice));
    return RDA_OK;
}
extern "C" int rda_debug_su_prof(rda_handle *H, long long *out16)
{
    if (!H || !out16) return RDA_ERR_ARG;
    if (!H->d.su_prof) return RDA_ERR_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(out16, H->d.su_prof, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(H->d.su_prof, 0, 16 * sizeof(long long)));
    return RDA_OK;
}

extern "C" int rda_debug_su_land(rda_handle *H, int32_t *out4)
{
    if (!H || !out4) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(out4, H->d.ctrl->land_stat, 4 * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(H->d.ctrl->land_stat, 0, su::LAND_STATS * sizeof(int)));
    return RDA_OK;
}

extern "C" int rda_debug_su_land_n(rda_handle *H, int32_t *out, int n)
{
    if (!H || !out || n < 1 || n > su::LAND_STATS) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(out, H->d.ctrl->land_stat, n * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(H->d.ctrl->land_stat, 0, su::LAND_STATS * sizeof(int)));
    return RDA_OK;
}

// debug, -DSU_TRACE builds only: (event id, clock64) pairs of the four waves of the LAST su launch of this handle, [4][cap][2]; *n_out = events per wave
extern "C" int rda_debug_su_trace(rda_handle *H, long long *out, int cap, int *n_out)
{
    if (!H || !out || !n_out || cap < 1) return RDA_ERR_ARG;
#if defined(SU_TRACE)
    if (!H->d.su_prof) return RDA_ERR_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(H->stream));
    std::vector<long long> buf(2 * 4 * su::TRACE_CAP);
    HIPCHK(hipMemcpy(buf.data(), H->d.su_prof + 16, buf.size() * sizeof(long long), hipMemcpyDeviceToHost));
    const int n = cap < su::TRACE_CAP ? cap : su::TRACE_CAP;
    for (int w = 0; w < 4; ++w)
        for (int e = 0; e < n; ++e) { out[(w * cap + e) * 2] = buf[(w * su::TRACE_CAP + e) * 2]; out[(w * cap + e) * 2 + 1] = buf[(w * su::TRACE_CAP + e) * 2 + 1]; }
    *n_out = n;
    return RDA_OK;
#else
    (void)H;
    return RDA_ERR_UNSUPPORTED;
#endif
}

// debug: rows the common-path LamMuZ kernel of the LAST executed iteration put on the work list (split launch form only)
// debug / test hook: forget every remembered support (both buffers -> "none").  The supports are a cache - every answer is accepted on its
// optimality certificate alone - so a loop must return the same bits with or without them (tests/test_gpu_supports.py).
extern "C" int rda_debug_flush_supports(rda_handle *H)
{
    if (!H) return RDA_ERR_ARG;
    HIPCHK(hipMemsetAsync(H->d.hint, 0xff, 2 * (size_t)H->d.hint_len * sizeof(int), H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    return RDA_OK;
}
extern "C" int rda_debug_slot_src(rda_handle *H, int32_t *src, int32_t *used)
{
    if (!H || !src || !used) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    if (H->scene_on_s2) HIPCHK(hipStreamSynchronize(H->stream2));
    *used = H->d.slot_src ? H->d.src_used : 0;
    if (*used > 0) HIPCHK(hipMemcpy(src, H->d.slot_src, (size_t)*used * sizeof(int), hipMemcpyDeviceToHost));
    return RDA_OK;
}
extern "C" int rda_debug_worklist(rda_handle *H, int *rows)
{
    if (!H || !rows) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(rows, &H->d.ctrl->wl_count, sizeof(int), hipMemcpyDeviceToHost));
    return RDA_OK;
}

// assign_obstacle_parameter (rda_solver.py:483-526): pad / truncate into N slots, then upload
static int obstacles_stage(rda_handle *H, int n_obs, const double *A, const double *b, const int32_t *cone, int per_t, bool sync)
{
    if (!H) return RDA_ERR_ARG;
    Dev &d = H->d;
    const size_t T = d.c.T, N = d.c.N, E = d.c.E;
    if (n_obs <= 0) { d.obstacle_num = 0; return RDA_OK; }       // nothing written: stale A, b stay
    if (!A || !b || !cone) return RDA_ERR_ARG;
    if (H->follow) return RDA_ERR_UNSUPPORTED;                   // duals_follow: host-staged slots carry no obstacle identity
    d.sc_bad = nullptr;
    H->sc_n = 0;                                                 // the slots no longer come from the resident raw scene: rda_scene_resort must not rebuild them from it
    const size_t nt = per_t ? T + 1 : 1;
    for (size_t n = 0; n < N; ++n) {
        size_t src = n < (size_t)n_obs ? n : (size_t)n_obs - 1;   // quirk Q3: duplicate the last obstacle
        memcpy(&H->h_stage_A[n * nt * E * 2], &A[src * nt * E * 2], nt * E * 2 * sizeof(double));
        memcpy(&H->h_stage_b[n * nt * E], &b[src * nt * E], nt * E * sizeof(double));
        H->h_stage_cone[n] = cone[src];
        if (cone[src] == 1) {         // canonical circle only (mpc.py:440-458)
            for (size_t t = 0; t < nt; ++t) {
                const double *At = &H->h_stage_A[(n * nt + t) * E * 2];
                if (E < 3 || At[0] != 1 || At[1] != 0 || At[2] != 0 || At[3] != 1 || At[4] != 0 || At[5] != 0) return RDA_ERR_UNSUPPORTED;
            }
        }
    }
    HIPCHK(hipMemcpyAsync(d.A, H->h_stage_A, N * nt * E * 2 * sizeof(double), hipMemcpyHostToDevice, H->stream));
    HIPCHK(hipMemcpyAsync(d.b, H->h_stage_b, N * nt * E * sizeof(double), hipMemcpyHostToDevice, H->stream));
    HIPCHK(hipMemcpyAsync(d.cone, H->h_stage_cone, N * sizeof(int), hipMemcpyHostToDevice, H->stream));
    d.nt = (int)nt; d.obstacle_num = (int)N;
    d.slot_src = nullptr; d.src_used = 0;                    // host-staged slots: the remembered supports are keyed by slot
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)((N * nt + 3) / 4)), dim3(256), 0, H->stream, d);
    if (sync) HIPCHK(hipStreamSynchronize(H->stream));      // staging buffers are reused by the next call
    return RDA_OK;
}

extern "C" int rda_upload_obstacles(rda_handle *H, int n_obs, const double *A, const double *b, const int32_t *cone, int per_t)
{
    return obstacles_stage(H, n_obs, A, b, cone, per_t, true);
}

// ---- device-side obstacle pipeline -----------------------------------------------------------------------------
// raw scene block, identical on the host (pinned) and on the device: geom [n][E][2] | vel [n][2] | robot [2] | nonconvex
// counter (int, padded to 8 bytes) | kind [n] | nvert [n]
static size_t scene_block_bytes(size_t n, size_t E) { return n * (E * 2 + 2) * sizeof(double) + 3 * sizeof(double) + 2 * n * sizeof(int); }
static int scene_reserve(rda_handle *H, int n)
{
    const size_t E = H->d.c.E;
    if (n > H->sc_cap) {
        dev_free(H->d_sc_blk); dev_free(H->d_sc_sel); dev_free(H->d_sc_key);
        H->d_sc_blk = nullptr; H->d_sc_sel = nullptr; H->d_sc_key = nullptr; H->sc_cap = 0; H->sc_n = 0;
        int cap = n + n / 2 + 16, rc = 0;
        rc |= dalloc(&H->d_sc_blk, scene_block_bytes((size_t)cap, E) / sizeof(double) + 1);
        rc |= dalloc(&H->d_sc_sel, (size_t)cap); rc |= dalloc(&H->d_sc_key, (size_t)cap);
        if (rc) return RDA_ERR_HIP;
        H->sc_cap = cap;
    }
    const size_t need = scene_block_bytes((size_t)n, E);
    if (need > H->h_sc_bytes) {
        if (H->h_sc) (void)hipHostFree(H->h_sc);
        H->h_sc = nullptr; H->h_sc_bytes = 0;
        HIPCHK(hipHostMalloc(&H->h_sc, need * 2));
        H->h_sc_bytes = need * 2;
    }
    return RDA_OK;
}

// the conversion kernels on a raw scene that is in device memory: distance keys, stable rank, half-space slots, candidate lists
static void scene_kernels(rda_handle *H, const scene::Args &a, hipStream_t st)
{
    Dev &d = H->d;
    const int n = a.n, N = a.N;
    hipLaunchKernelGGL(scene::k_keys, dim3((n + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(scene::k_rank, dim3((n + 15) / 16), dim3(256), 0, st, a);
    hipLaunchKernelGGL(scene::k_build, dim3((N * a.nt + 255) / 256), dim3(256), 0, st, a);
    d.nt = a.nt; d.obstacle_num = N; d.sc_bad = a.nonconvex;
    d.slot_src = H->d_sc_sel; d.src_used = n < N ? n : N;          // the remembered supports follow the obstacles through the re-binding (Dev::hint)
    if (H->follow) {
        // rda_opts::duals_follow: the dual state is re-arranged from the previous binding to this one.  Nothing of a tick's head reads
        // lam, mu, xi, z, zeta (the su-problem reads the condensed terms, which the first LamMuZ launch of the tick rewrites for every
        // slot): inside a tick this runs on the second stream beside the first su-problem like the rest of the staging.
        const int used = d.src_used;
        const bool have = H->prev_used >= 0;
        if (have) hipLaunchKernelGGL(k_follow_gather, dim3(N), dim3(64), 0, st, d, (const int *)H->d_sc_sel, used, (const int *)H->d_prev_sel, H->prev_used, H->d_follow_map, H->d_follow_tmp);
        hipLaunchKernelGGL(k_follow_back, dim3(N), dim3(64), 0, st, d, have ? (const int *)H->d_follow_map : (const int *)nullptr, (const double *)H->d_follow_tmp,
                           (const int *)H->d_sc_sel, used, H->d_prev_sel);
        H->prev_used = used;
    }
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)((N * a.nt + 3) / 4)), dim3(256), 0, st, d);
}

static int scene_stage(rda_handle *H, int n, const int32_t *kind, const int32_t *nvert, const double *geom,
                       const double *vel, const double *robot_xy, int order, int32_t *n_nonconvex, bool sync, hipStream_t st = nullptr)
{
    if (!H) return RDA_ERR_ARG;
    if (!st) st = H->stream;
    Dev &d = H->d;
    if (n_nonconvex) *n_nonconvex = 0;
    if (n <= 0) { d.obstacle_num = 0; return RDA_OK; }           // nothing written: stale A, b stay (rda_solver.py:485)
    if (!kind || !nvert || !geom || !vel || (order && !robot_xy)) return RDA_ERR_ARG;
    const int E = d.c.E, T = d.c.T, N = d.c.N;
    bool any_moving = false;
    for (int i = 0; i < n; ++i) {
        if (kind[i] == 1) { if (E < 3) return RDA_ERR_UNSUPPORTED; }
        else if (kind[i] != 0 || nvert[i] < 0 || nvert[i] > E) return RDA_ERR_ARG;
        any_moving = any_moving || sqrt(vel[2 * i] * vel[2 * i] + vel[2 * i + 1] * vel[2 * i + 1]) > 0.01;
    }
    int rc = scene_reserve(H, n);
    if (rc != RDA_OK) return rc;
    // one pinned staging block -> its device mirror, one copy
    const size_t o_vel = (size_t)n * E * 2, o_rob = o_vel + (size_t)n * 2, o_bad = o_rob + 2, o_int = o_bad + 1;   // in doubles
    double *hb = (double *)H->h_sc, *db = H->d_sc_blk;
    memcpy(hb, geom, o_vel * sizeof(double));
    memcpy(hb + o_vel, vel, (size_t)n * 2 * sizeof(double));
    hb[o_rob] = robot_xy ? robot_xy[0] : 0; hb[o_rob + 1] = robot_xy ? robot_xy[1] : 0;
    hb[o_bad] = 0.0;                                                       // all-zero bits: the int counter starts at 0
    memcpy((int *)(hb + o_int), kind, (size_t)n * sizeof(int));
    memcpy((int *)(hb + o_int) + n, nvert, (size_t)n * sizeof(int));
    HIPCHK(hipMemcpyAsync(db, hb, scene_block_bytes((size_t)n, (size_t)E), hipMemcpyHostToDevice, st));
    int *const d_bad = (int *)(db + o_bad);
    d.sc_bad = d_bad;
    scene::Args a;
    a.n = n; a.N = N; a.E = E; a.T = T; a.nt = any_moving ? T + 1 : 1; a.order = order; a.dt = d.c.dt;
    a.kind = (int *)(db + o_int); a.nvert = (int *)(db + o_int) + n; a.geom = db; a.vel = db + o_vel; a.robot = db + o_rob;
    a.key = H->d_sc_key; a.sel = H->d_sc_sel; a.A = d.A; a.b = d.b; a.cone = d.cone; a.nonconvex = d_bad;
    a.rx = 0; a.ry = 0; a.robot_val = 0;
    H->sc_args = a; H->sc_n = n;
    scene_kernels(H, a, st);
    HIPCHK(hipGetLastError());
    if (n_nonconvex) {
        HIPCHK(hipMemcpyAsync(H->h_sc, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        *n_nonconvex = *(int *)H->h_sc;
    } else if (sync) {
        HIPCHK(hipStreamSynchronize(st));                  // the staging block is reused by the next call
    }
    return RDA_OK;
}

extern "C" int rda_last_nonconvex(rda_handle *H)
{
    if (!H) return RDA_ERR_ARG;
    const size_t T = H->d.c.T;
    return (int)*(const long long *)(H->h_out + 2 * T + 3 * (T + 1) + 6);
}

extern "C" int rda_upload_scene(rda_handle *H, int n, const int32_t *kind, const int32_t *nvert, const double *geom,
                                const double *vel, const double *robot_xy, int order, int32_t *n_nonconvex)
{
    return scene_stage(H, n, kind, nvert, geom, vel, robot_xy, order, n_nonconvex, true);
}

// test hook: the staged obstacle slots as the solver sees them
extern "C" int rda_get_obstacles(rda_handle *H, double *A, double *b, int32_t *cone, int32_t *nt)
{
    if (!H || !A || !b || !cone || !nt) return RDA_ERR_ARG;
    const Dev &d = H->d;
    const size_t N = d.c.N, E = d.c.E;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(A, d.A, N * d.nt * E * 2 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b, d.b, N * d.nt * E * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cone, d.cone, N * sizeof(int), hipMemcpyDeviceToHost));
    *nt = d.nt;
    return RDA_OK;
}

static hipEvent_t next_event(rda_handle *H, int which)
{
    if (H->ev_used[which] == H->ev[which].size()) { hipEvent_t e; (void)hipEventCreate(&e); H->ev[which].push_back(e); }
    return H->ev[which][H->ev_used[which]++];
}

// K1 launch of ADMM iteration `it`.  Packed rows when the shape allows: a small grid is ONE launch that also forms the block
// partials; a dense grid is the common-path kernel + the work-list kernel, with k_lmz_finalize (partials) behind them.  One sub-problem per wave (big
// shapes, the no-obstacle case of quirk Q9) and the per-thread interior-point kernel likewise end with k_lmz_finalize.
static void launch_finalize(rda_handle *H, const Dev &d, int it, const Fin &fin)
{
    hipLaunchKernelGGL(k_lmz_finalize, dim3((d.c.T * d.J + FPB - 1) / FPB), dim3(256), 0, H->stream, d, it, fin);
}
// (the row-parallel interior-point kernel: see k_lammuz_ip below)
static void launch_lammuz_ip(rda_handle *H, const Dev &d, int it, const Fin &fin)
{
    hipLaunchKernelGGL(k_lammuz_ip, dim3(packed_grid(d.c.T, d.J)), dim3(64 * GS / 4), 0, H->stream, d, it, fin);
}
// the LamMuZ launch form launch_lammuz picks for the handle's shape and staged obstacles (bench.py labels its per-launch times with it)
extern "C" const char *rda_lammuz_kernel(rda_handle *H)
{
    if (!H) return "";
    const Dev &d = H->d;
    if (d.lmz_mode && !(H->ip_rows && d.obstacle_num)) return (d.c.E <= 4 && d.c.R <= 4) ? "k_lammuz_cp_small+k_lmz_finalize" : "k_lammuz_cp_large+k_lmz_finalize";
    if (d.lmz_mode) return "k_lammuz_ip";
    if (d.rows && d.obstacle_num) {
        const int cus = (d.c.T * d.J * (64 * GS / 4) + 255) / 256, dense_from = d.nt > 1 ? H->dense_from * 7 / 4 : H->dense_from;
        if (cus > dense_from && H->lmz_split) return "k_lammuz_rows_fast+k_lammuz_enum+k_lmz_finalize";
        return cus > dense_from ? "k_lammuz_rows_dense" : "k_lammuz_rows";
    }
    return "k_lammuz+k_lmz_finalize";
}
// Grid size (in compute units at one wave per SIMD) from which the split form of the LamMuZ launch is used.  Scenes with per-stage
// obstacle data (moving obstacles) lose the remembered support of more rows per launch, so the work list of the split form is longer and
// the single launch stays ahead up to a larger grid: measured cross-over ~280 CUs for static scenes (N = 300, T = 20: 35.7 vs 38.7 us),
// ~560 for moving ones (T = 30: N = 200 40.9 vs 47.1 us for the single launch, N = 300 52.2 vs 50.1, N = 400 60.7 vs 51.1).
static inline int dense_threshold(const rda_handle *H, const Dev &d) { return d.nt > 1 ? H->dense_from * 7 / 4 : H->dense_from; }
static void launch_lammuz(rda_handle *H, const Dev &d, int it, const Fin &fin)
{
    if (d.Nlive == 0) return;                    // a shard without obstacles (N < P)
    const int units = d.c.T * GS * d.J;          // rows of the grid (a stage is padded to whole GS-slot blocks)
    if (d.lmz_mode && !(H->ip_rows && d.obstacle_num)) {
        const int nth = d.Nlive * d.c.T, nb = d.obstacle_num ? (nth + 63) / 64 : (d.c.T + 63) / 64;
        if (d.c.E <= 4 && d.c.R <= 4) hipLaunchKernelGGL(k_lammuz_cp_small, dim3(nb), dim3(64), 0, H->stream, d);
        else hipLaunchKernelGGL(k_lammuz_cp_large, dim3(nb), dim3(64), 0, H->stream, d);
        launch_finalize(H, d, it, fin);
        return;
    }
    if (d.lmz_mode) { launch_lammuz_ip(H, d, it, fin); return; }
    if (d.rows && d.obstacle_num) {
        constexpr int NTH = 64 * GS / 4;             // threads of a packed workgroup (2 waves)
        const int nb = packed_grid(d.c.T, d.J), cus = (units / GS * NTH + 255) / 256;      // launch indices (XCD-aware order, a few fillers); CUs the grid asks for at one wave per SIMD
        const int dense_from = dense_threshold(H, d);
        if (cus > dense_from && H->lmz_split) {
            // dense grid: common path with three waves per SIMD, then the deferred rows one per wave (see lammuz_body_rows)
            hipLaunchKernelGGL(k_lammuz_rows_fast, dim3(nb), dim3(NTH), 0, H->stream, d, it);
            int ne = units / 32; if (ne < 64) ne = 64; if (ne > 2048) ne = 2048;
            hipLaunchKernelGGL(k_lammuz_enum, dim3(ne), dim3(NTH), 0, H->stream, d, it);
            launch_finalize(H, d, it, fin);
        } else if (cus > dense_from) hipLaunchKernelGGL(k_lammuz_rows_dense, dim3(nb), dim3(NTH), 0, H->stream, d, it, fin);
        else hipLaunchKernelGGL(k_lammuz_rows, dim3(nb), dim3(NTH), 0, H->stream, d, it, fin);
    } else {
        hipLaunchKernelGGL(k_lammuz, dim3(d.obstacle_num ? (units + 3) / 4 : 1), dim3(256), 0, H->stream, d, it);
        launch_finalize(H, d, it, fin);
    }
}

// queue the whole ADMM loop of one MPC step (rda_solver.py:588-596) - no host synchronisation
// The ADMM loop of one MPC step in two parts.  The HEAD (the first su-problem, which also resets the step's control block) reads the nominal trajectory and the
// condensed terms of the PREVIOUS step (quirk Q4) but nothing of the staged obstacles, so a caller may stage this tick's
// obstacles on the stream between head and tail while the first su-problem is being solved (rda_tracked_begin/_finish).
static void launch_su(rda_handle *H, const Dev &d, int it, const double *in_s, const double *in_u, const Fin &fin = Fin{nullptr, nullptr, nullptr, nullptr, 0, 0})
{
    const int T = d.c.T;
    if (H->timing) (void)hipEventRecord(next_event(H, 1), H->stream);
    RDA_SU_DISPATCH(T, hipLaunchKernelGGL(k_su<TT>, dim3(1), dim3(su::NT), H->su_lds, H->stream, d, it, in_s, in_u, fin));
    if (H->timing) (void)hipEventRecord(next_event(H, 1), H->stream);
}
// Hand-over of a step's result slot to the host block h_out.  Zero-copy form: the launch that ends the step (the su launch that
// detects the early stop, else k_finish) writes the block itself and the host polls the sequence word behind it.  Everything the
// step queued before that launch has completed by then; what is still queued behind it are launches that return at once.
// After 20 ms without the word - or with the timing events on, or RDA_ZERO_COPY=0 - the stream is synchronised the ordinary
// way, which also surfaces a device fault.
static Fin make_fin(rda_handle *H, double *out_u, double *out_s, rda_info *info)
{
    const bool zc = H->zero_copy && !H->timing && out_u == H->d_out_u;
    if (zc) H->res_seq += 1;
    return Fin{ out_u, out_s, info, zc ? H->h_out : nullptr, H->res_seq, out_u == H->d_out_u ? 1 : 0 };
}
static int launch_finish(rda_handle *H, const Dev &d, const Fin &fin)
{
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(su::NT), H->su_lds, H->stream, d, fin);
    HIPCHK(hipGetLastError());
    return RDA_OK;
}
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}
static int fetch_result(rda_handle *H)
{
    const size_t T = H->d.c.T;
    if (H->zero_copy && !H->timing) {
        volatile unsigned long long *flag = (volatile unsigned long long *)(H->h_out + 2 * T + 3 * (T + 1) + 7);
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            if (*flag == H->res_seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return RDA_OK; }
            cpu_relax();
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        // 20 ms without the word (a healthy step takes a fraction of a millisecond): synchronise the ordinary way.  A device fault -
        // in this step, or in the launches that were still queued behind the hand-over of the PREVIOUS step (the zero-copy hand-over
        // returns before they have drained) - never publishes the word and surfaces here as the stream's error.
        HIPCHK(hipStreamSynchronize(H->stream));
        if (*flag == H->res_seq) return RDA_OK;
        return RDA_ERR_HIP;
    }
    // D2H form: the result block without the sequence word behind it (that word belongs to the zero-copy protocol)
    HIPCHK(hipMemcpyAsync(H->h_out, H->d_out_u, (res_doubles(T) - 1) * sizeof(double), hipMemcpyDeviceToHost, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    return RDA_OK;
}
static int enqueue_admm_head(rda_handle *H, const double *in_s, const double *in_u, const double *ref, const double *speed)
{
    Dev d = H->d;
    H->stepped = 1;
    d.ref = const_cast<double *>(ref); d.ref_speed = const_cast<double *>(speed);
    launch_su(H, d, 0, in_s, in_u);                    // resets the step's control block itself (su_body, it == 0)
    HIPCHK(hipGetLastError());
    return RDA_OK;
}
static int enqueue_admm_tail(rda_handle *H, const double *in_s, const double *in_u, const double *ref, const double *speed,
                             double *out_u, double *out_s, rda_info *info)
{
    Dev d = H->d;                                     // taken AFTER the obstacles of this tick were staged (nt, obstacle_num)
    d.ref = const_cast<double *>(ref); d.ref_speed = const_cast<double *>(speed);
    const Fin fin = make_fin(H, out_u, out_s, info);  // the su launch that detects the early stop hands the result over itself
    for (int it = 0; it < d.c.iter_num; ++it) {
        if (it > 0) {
            Fin f = H->early_finish ? fin : Fin{nullptr, nullptr, nullptr, nullptr, 0, 0};
            if (H->comm) { f.verdict = H->h_verdict; f.vseq = ++H->vseq; }
            launch_su(H, d, it, in_s, in_u, f);
        }
        if (it > 0 && H->comm) {
            // The early stop (rda_solver.py:594) is a device flag and kernels queued behind it return at once - a collective cannot, so
            // with a communicator no all-gather may be queued for an iteration that does not run.  The su launch publishes its verdict
            // in pinned host memory as soon as it has reduced the residuals (before its solve; every rank takes the same verdict: the
            // su-problems are bitwise identical); the host polls that word - no copy, no stream synchronisation - and goes on queueing
            // (or stops) while the su-problem is being solved.
            volatile unsigned long long *vw = H->h_verdict;
            const auto t0 = std::chrono::steady_clock::now();
            bool seen = false;
            for (unsigned spin = 0; !seen; ++spin) {
                if ((*vw >> 1) == H->vseq) { seen = true; break; }
                cpu_relax();
                if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
            }
            if (!seen) { HIPCHK(hipStreamSynchronize(H->stream)); if ((*vw >> 1) != H->vseq) return RDA_ERR_HIP; }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (*vw & 1ull) break;
        }
        if (H->timing) (void)hipEventRecord(next_event(H, 0), H->stream);
        launch_lammuz(H, d, it, H->early_finish ? fin : Fin{nullptr, nullptr, nullptr, nullptr, 0, 0});
        if (H->timing) (void)hipEventRecord(next_event(H, 0), H->stream);
        if (H->comm) {      // one exchange per ADMM iteration: every rank's chunk to every rank (in place)
            if (H->timing) (void)hipEventRecord(next_event(H, 2), H->stream);
            int nrc = H->p_allgather(d.coef + (size_t)d.rank * d.chunk, d.coef, d.chunk, /*ncclDouble*/ 8, H->comm, H->stream);
            if (H->timing) (void)hipEventRecord(next_event(H, 2), H->stream);
            if (nrc != 0) { fprintf(stderr, "librda_hip: ncclAllGather failed (%d)\n", nrc); return RDA_ERR_HIP; }
        }
    }
    return launch_finish(H, d, fin);
}
static int enqueue_admm(rda_handle *H, const double *in_s, const double *in_u, const double *ref, const double *speed,
                        double *out_u, double *out_s, rda_info *info)
{
    int rc = enqueue_admm_head(H, in_s, in_u, ref, speed);
    if (rc != RDA_OK) return rc;
    return enqueue_admm_tail(H, in_s, in_u, ref, speed, out_u, out_s, info);
}

// nominal / reference in, ADMM loop, control / state / info out (host buffers, synchronous).  `stage` queues this tick's
// obstacles (pinned staging block -> H2D -> conversion kernels) WITHOUT waiting: the one synchronisation of the step is
// the one at its end, which also makes the staging blocks reusable by the next call.
template <typename Stage>
static int step_common(rda_handle *H, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed,
                       double *out_u, double *out_s, rda_info *info, Stage stage)
{
    if (H->pending) return RDA_ERR_ARG;
    int rc = stage();
    if (rc != RDA_OK) { (void)hipStreamSynchronize(H->stream); return rc; }
    const size_t T = H->d.c.T;
    const size_t ns = 3 * (T + 1), nu = 2 * T;
    memcpy(H->h_step, nom_s, ns * sizeof(double));
    memcpy(H->h_step + ns, nom_u, nu * sizeof(double));
    memcpy(H->h_step + ns + nu, ref_s, ns * sizeof(double));
    H->h_step[ns + nu + ns] = ref_speed;
    HIPCHK(hipMemcpyAsync(H->d_step, H->h_step, (2 * ns + nu + 1) * sizeof(double), hipMemcpyHostToDevice, H->stream));
    rc = enqueue_admm(H, H->d_step, H->d_step + ns, H->d_step + ns + nu, H->d_step + ns + nu + ns, H->d_out_u, H->d_out_s, H->d_info);
    if (rc != RDA_OK) return rc;
    rc = fetch_result(H);
    if (rc != RDA_OK) return rc;
    memcpy(out_u, H->h_out, nu * sizeof(double));
    memcpy(out_s, H->h_out + nu, ns * sizeof(double));
    if (info) *info = *H->h_info;
    return RDA_OK;
}

extern "C" int rda_step(rda_handle *H, const double *nom_s, const double *nom_u, const double *ref_s,
                        double ref_speed, int n_obs, const double *A, const double *b, const int32_t *cone,
                        int per_t, double *out_u, double *out_s, rda_info *info)
{
    if (!H || !nom_s || !nom_u || !ref_s || !out_u || !out_s) return RDA_ERR_ARG;
    return step_common(H, nom_s, nom_u, ref_s, ref_speed, out_u, out_s, info,
                       [&]() { return obstacles_stage(H, n_obs, A, b, cone, per_t, false); });
}

extern "C" int rda_step_scene(rda_handle *H, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed,
                              int n, const int32_t *kind, const int32_t *nvert, const double *geom, const double *vel,
                              const double *robot_xy, int order, double *out_u, double *out_s, rda_info *info)
{
    if (!H || !nom_s || !nom_u || !ref_s || !out_u || !out_s) return RDA_ERR_ARG;
    return step_common(H, nom_s, nom_u, ref_s, ref_speed, out_u, out_s, info,
                       [&]() { return scene_stage(H, n, kind, nvert, geom, vel, robot_xy, order, nullptr, false); });
}

// ---- device-side pre_process --------------------------------------------------------------------
__global__ void k_track(Dev d, track::In in, double *path, int L, const double *nom_u, double *step, track::Out *out)
{
    __shared__ double win[track::LDS_DOUBLES];
    const int T = d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    track::Ego e;
    e.path = path; e.L = L; e.nom_u = nom_u; e.nom_s = step; e.ref = step + ns + nu; e.speed = step + 2 * ns + nu;
    e.T = T; e.dynamics = d.c.dynamics; e.dt = d.c.dt; e.wheelbase = d.c.L;
    track::run(e, in, *out, win, threadIdx.x);
}

// k_track and the first su-problem of the step as ONE launch of two workgroups.  Workgroup 0: wave 0 rolls the nominal out (the
// only part of pre_process the su set-up needs), then the su-problem starts; workgroup 1 samples the reference from the path
// meanwhile (closest_point, inter_point: 20 dependent searches, ~12 us) and publishes it under the tick number; the solve picks it
// up after its set-up (su::Args::ref_flag).  Same arithmetic as k_track followed by k_su<TT>(it = 0).
// HIP does not promise that the two workgroups run concurrently (CU masking, a saturated GPU, a debugger): the wait is BOUNDED, and on
// expiry workgroup 0 samples the reference itself (the same track::run part, the same values - should workgroup 1 still run later it
// rewrites them identically).
struct TrackedRefWait {
    track::Ego e; track::In in; track::Out *out; const unsigned long long *flag; unsigned long long seq;
    __device__ __forceinline__ void operator()(double *scratch) const
    {
        int *okw = reinterpret_cast<int *>(scratch);
        if (threadIdx.x == 0) {
            int ok = 0;
            for (int spin = 0; spin < 40000; ++spin) {            // ~ 20 ms at the slowest; a normal wait is 10 - 30 us
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            *okw = ok;
        }
        __syncthreads();
        const int ok = *(volatile int *)okw;
        __syncthreads();
        if (!ok) {
            if (threadIdx.x < 64) track::run(e, in, *out, scratch, threadIdx.x, 2);
            __syncthreads();
        }
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
};
template <int TT> __global__ __launch_bounds__(su::NT) void k_su_tracked(Dev d, track::In in, double *path, int L, const double *nom_u, double *step,
                                                                          track::Out *out, unsigned long long seq)
{
    warm_kernargs<sizeof(Dev) + sizeof(track::In) + 5 * sizeof(void *)>();
    const int T = d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    track::Ego e;
    e.path = path; e.L = L; e.nom_u = nom_u; e.nom_s = step; e.ref = step + ns + nu; e.speed = step + 2 * ns + nu;
    e.T = T; e.dynamics = d.c.dynamics; e.dt = d.c.dt; e.wheelbase = d.c.L;
    if (blockIdx.x == 1) {
        if (threadIdx.x < 64) {
            track::run(e, in, *out, smem_su, threadIdx.x, 2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_store(&d.ctrl->ref_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (threadIdx.x < 64) track::run(e, in, *out, smem_su, threadIdx.x, 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    su_body<TT>(d, 0, step, nom_u, d.ref, d.ref_speed, &d.ctrl->ref_seq, seq, nullptr, TrackedRefWait{ e, in, out, &d.ctrl->ref_seq, seq });
}

extern "C" int rda_upload_path(rda_handle *H, int L, const double *path)
{
    if (!H || L < 1 || !path) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    if (L > H->path_len || !H->d_path) { dev_free(H->d_path); H->d_path = nullptr; if (dalloc(&H->d_path, (size_t)3 * L)) return RDA_ERR_HIP; }
    H->path_len = L;
    HIPCHK(hipMemcpy(H->d_path, path, (size_t)3 * L * sizeof(double), hipMemcpyHostToDevice));
    return RDA_OK;
}

extern "C" int rda_tracked_begin(rda_handle *H, const double *state, double ref_speed, int cur_index, double threshold, int ind_range,
                                 const double *nom_u)
{
    if (!H || !state || H->pending) return RDA_ERR_ARG;
    if (!H->d_path || cur_index < 0 || cur_index >= H->path_len || ind_range < 1) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    const double *in_u = H->d.u;                       // the controls of the last solve are still resident
    if (nom_u) {
        memcpy(H->h_step + ns, nom_u, nu * sizeof(double));
        HIPCHK(hipMemcpyAsync(H->d_step + ns, H->h_step + ns, nu * sizeof(double), hipMemcpyHostToDevice, H->stream));
        in_u = H->d_step + ns;
    }
    track::In in; in.sx = state[0]; in.sy = state[1]; in.sth = state[2]; in.speed = ref_speed; in.threshold = threshold;
    in.cur_index = cur_index; in.ind_range = ind_range;
    // everything queued before this tick: an in-tick scene staging (rda_upload_scene_async) starts behind this event, beside the first
    // su-problem.  Recorded only for callers that stage inside their ticks (learned from the previous tick); a first in-tick staging
    // without it orders itself behind the head of the tick instead.
    H->tick_has_event = 0;
    if (H->tick_stages) { HIPCHK(hipEventRecord(H->ev_tick, H->stream)); H->tick_has_event = 1; }
    H->tick_stages = 0;
    if (H->fuse_track) {
        Dev d = H->d;
        d.ref = H->d_step + ns + nu; d.ref_speed = H->d_step + 2 * ns + nu;
        if (H->timing) (void)hipEventRecord(next_event(H, 1), H->stream);
        H->trk_seq += 1;
        RDA_SU_DISPATCH((int)T, hipLaunchKernelGGL(k_su_tracked<TT>, dim3(2), dim3(su::NT), H->su_trk_lds, H->stream, d, in, H->d_path, H->path_len,
                                                   in_u, H->d_step, H->d_trk, H->trk_seq));
        if (H->timing) (void)hipEventRecord(next_event(H, 1), H->stream);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_track, dim3(1), dim3(64), 0, H->stream, H->d, in, H->d_path, H->path_len, in_u, H->d_step, H->d_trk);
        int rc = enqueue_admm_head(H, H->d_step, in_u, H->d_step + ns + nu, H->d_step + 2 * ns + nu);
        if (rc != RDA_OK) return rc;
    }
    H->pending = 1; H->pending_in_u = in_u;
    return RDA_OK;
}

// asynchronous scene work (stage(stream) -> rc; `live`: it queued something), outside or inside a tick
template <typename Stage> static int scene_async(rda_handle *H, bool live, Stage stage)
{
    // one pinned staging block per handle: a second upload before the caller's synchronising call (rda_tracked_finish,
    // rda_sync, a fleet step) has to wait for the first one to have been copied
    if (H->pending_scene) { HIPCHK(hipStreamSynchronize(H->stream)); HIPCHK(hipStreamSynchronize(H->stream2)); H->pending_scene = 0; H->scene_on_s2 = 0; }
    if (!H->pending) {
        int rc = stage(H->stream);
        if (rc == RDA_OK && live) H->pending_scene = 1;
        return rc;
    }
    // inside a tick: the copy and the conversion kernels touch only the obstacle slots, which nothing of the head (k_track,
    // the first su-problem) reads - they run on the second stream BESIDE the first su-problem; rda_tracked_finish joins
    if (!H->tick_has_event) { HIPCHK(hipEventRecord(H->ev_tick, H->stream)); H->tick_has_event = 1; }
    H->tick_stages = 1;
    HIPCHK(hipStreamWaitEvent(H->stream2, H->ev_tick, 0));
    int rc = stage(H->stream2);
    if (rc == RDA_OK && live) { HIPCHK(hipEventRecord(H->ev_scene, H->stream2)); H->pending_scene = 1; H->scene_on_s2 = 1; }
    return rc;
}

extern "C" int rda_upload_scene_async(rda_handle *H, int n, const int32_t *kind, const int32_t *nvert, const double *geom,
                                      const double *vel, const double *robot_xy, int order)
{
    if (!H) return RDA_ERR_ARG;
    return scene_async(H, n > 0, [&](hipStream_t st) -> int { return scene_stage(H, n, kind, nvert, geom, vel, robot_xy, order, nullptr, false, st); });
}

// The reference re-sorts the obstacle list by distance on EVERY tick (mpc.py:205-206, obstacle_order=True) before the first max_obs_num
// are staged.  For a scene that is already resident (rda_upload_scene*) this re-ranks it about a new robot position and rebuilds the
// slots - the same three kernels, the position travels in the kernel arguments: no host-to-device copy.  Asynchronous like
// rda_upload_scene_async (inside a tick it runs beside the first su-problem).  Obstacles with a velocity are predicted from the
// geometry that was uploaded, i.e. a scene that MOVES between ticks has to be uploaded again instead.
extern "C" int rda_scene_resort(rda_handle *H, const double *robot_xy)
{
    if (!H || !robot_xy) return RDA_ERR_ARG;
    if (H->sc_n <= 0 || H->d.obstacle_num == 0) return RDA_ERR_ARG;              // no resident raw scene
    return scene_async(H, true, [&](hipStream_t st) -> int {
        scene::Args a = H->sc_args;
        a.order = 1; a.robot_val = 1; a.rx = robot_xy[0]; a.ry = robot_xy[1];
        HIPCHK(hipMemsetAsync(a.nonconvex, 0, sizeof(int), st));
        scene_kernels(H, a, st);
        HIPCHK(hipGetLastError());
        return (int)RDA_OK;
    });
}

extern "C" int rda_tracked_finish(rda_handle *H, double *out_u, double *out_s, rda_info *info,
                                  double *nom_s_out, double *ref_out, int32_t *min_index, double *end_heading)
{
    if (!H || !H->pending) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    const double *in_u = H->pending_in_u;
    if (H->scene_on_s2) HIPCHK(hipStreamWaitEvent(H->stream, H->ev_scene, 0));
    H->pending = 0; H->pending_scene = 0; H->scene_on_s2 = 0;
    int rc = enqueue_admm_tail(H, H->d_step, in_u, H->d_step + ns + nu, H->d_step + 2 * ns + nu, H->d_out_u, H->d_out_s, H->d_info);
    if (rc != RDA_OK) { (void)hipStreamSynchronize(H->stream); return rc; }
    if (nom_s_out || ref_out) {
        HIPCHK(hipMemcpyAsync(H->h_step, H->d_step, (2 * ns + nu) * sizeof(double), hipMemcpyDeviceToHost, H->stream));
        HIPCHK(hipStreamSynchronize(H->stream));
    }
    rc = fetch_result(H);
    if (rc != RDA_OK) return rc;
    if (out_u) memcpy(out_u, H->h_out, nu * sizeof(double));
    if (out_s) memcpy(out_s, H->h_out + nu, ns * sizeof(double));
    if (info) *info = *H->h_info;
    if (nom_s_out) memcpy(nom_s_out, H->h_step, ns * sizeof(double));
    if (ref_out) memcpy(ref_out, H->h_step + ns + nu, ns * sizeof(double));
    if (min_index) *min_index = H->h_trk->min_index;
    if (end_heading) *end_heading = H->h_trk->end_heading;
    return RDA_OK;
}

extern "C" int rda_step_tracked(rda_handle *H, const double *state, double ref_speed, int cur_index, double threshold, int ind_range,
                                const double *nom_u, double *out_u, double *out_s, rda_info *info,
                                double *nom_s_out, double *ref_out, int32_t *min_index, double *end_heading)
{
    if (!H || !state || !out_u || !out_s) return RDA_ERR_ARG;
    int rc = rda_tracked_begin(H, state, ref_speed, cur_index, threshold, ind_range, nom_u);
    if (rc != RDA_OK) return rc;
    return rda_tracked_finish(H, out_u, out_s, info, nom_s_out, ref_out, min_index, end_heading);
}

extern "C" int rda_upload_trace(rda_handle *H, int K, const double *nom_s, const double *nom_u, const double *ref_s, const double *ref_speed)
{
    if (!H || K < 1 || !nom_s || !nom_u || !ref_s || !ref_speed) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    dev_free(H->d_tr_s); dev_free(H->d_tr_u); dev_free(H->d_tr_ref); dev_free(H->d_tr_speed);
    dev_free(H->d_tr_out_u); dev_free(H->d_tr_out_s); dev_free(H->d_tr_info);
    int rc = 0;
    rc |= dalloc(&H->d_tr_s, K * ns); rc |= dalloc(&H->d_tr_u, K * nu); rc |= dalloc(&H->d_tr_ref, K * ns); rc |= dalloc(&H->d_tr_speed, (size_t)K);
    rc |= dalloc(&H->d_tr_out_u, K * nu); rc |= dalloc(&H->d_tr_out_s, K * ns); rc |= dalloc(&H->d_tr_info, (size_t)K);
    if (rc) return RDA_ERR_HIP;
    HIPCHK(hipMemcpy(H->d_tr_s, nom_s, K * ns * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(H->d_tr_u, nom_u, K * nu * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(H->d_tr_ref, ref_s, K * ns * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(H->d_tr_speed, ref_speed, K * sizeof(double), hipMemcpyHostToDevice));
    H->K = K;
    return RDA_OK;
}

extern "C" int rda_enqueue_step(rda_handle *H, int k)
{
    if (!H || k < 0 || k >= H->K) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    return enqueue_admm(H, H->d_tr_s + k * ns, H->d_tr_u + k * nu, H->d_tr_ref + k * ns, H->d_tr_speed + k,
                        H->d_tr_out_u + k * nu, H->d_tr_out_s + k * ns, H->d_tr_info + k);
}

// steps k0 .. k1-1 of the uploaded trace, queued back to back (one host call instead of k1-k0)
extern "C" int rda_enqueue_range(rda_handle *H, int k0, int k1)
{
    if (!H || k0 < 0 || k1 > H->K || k0 > k1) return RDA_ERR_ARG;
    for (int k = k0; k < k1; ++k) { int rc = rda_enqueue_step(H, k); if (rc != RDA_OK) return rc; }
    return RDA_OK;
}

#ifdef RDA_LMZ_CLK
extern "C" int rda_debug_lmz_clk(rda_handle *H, unsigned long long *out, int max_waves)     // out[max_waves][16]; reads and clears
{
    static unsigned long long *buf = nullptr; static int cap = 0;
    HIPCHK(hipStreamSynchronize(H->stream));
    if (!buf) {
        cap = max_waves;
        HIPCHK(hipMalloc((void **)&buf, (size_t)cap * 16 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(buf, 0, (size_t)cap * 16 * sizeof(unsigned long long)));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_lmz_clk), &buf, sizeof(buf)));
        return RDA_OK;
    }
    if (max_waves < 0) {          // the fail log instead: out = int[1 + 3 * 200000]; first call arms it
        static int *flog = nullptr;
        if (!flog) {
            HIPCHK(hipMalloc((void **)&flog, (1 + 3 * 200000) * sizeof(int))); HIPCHK(hipMemset(flog, 0, (1 + 3 * 200000) * sizeof(int)));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_lmz_faillog), &flog, sizeof(flog)));
            return RDA_OK;
        }
        HIPCHK(hipMemcpy(out, flog, (1 + 3 * 200000) * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(flog, 0, sizeof(int)));
        return RDA_OK;
    }
    if (max_waves > cap) return RDA_ERR_ARG;
    HIPCHK(hipMemcpy(out, buf, (size_t)max_waves * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(buf, 0, (size_t)cap * 16 * sizeof(unsigned long long)));
    return RDA_OK;
}
#endif
#ifdef RDA_LMZ_STATS
extern "C" int rda_debug_lmz_stats(rda_handle *H, unsigned *out)
{
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(out, (char *)H->d.ctrl + offsetof(Ctrl, lmz_stat), 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
    return RDA_OK;
}
#endif
extern "C" int rda_sync(rda_handle *H) { if (!H) return RDA_ERR_ARG; HIPCHK(hipStreamSynchronize(H->stream)); H->pending_scene = 0; return RDA_OK; }

extern "C" int rda_fetch_result(rda_handle *H, int k, double *out_u, double *out_s, rda_info *info)
{
    if (!H || k < 0 || k >= H->K) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    HIPCHK(hipStreamSynchronize(H->stream));
    if (out_u) HIPCHK(hipMemcpy(out_u, H->d_tr_out_u + k * nu, nu * sizeof(double), hipMemcpyDeviceToHost));
    if (out_s) HIPCHK(hipMemcpy(out_s, H->d_tr_out_s + k * ns, ns * sizeof(double), hipMemcpyDeviceToHost));
    if (info) HIPCHK(hipMemcpy(info, H->d_tr_info + k, sizeof(rda_info), hipMemcpyDeviceToHost));
    return RDA_OK;
}

extern "C" int rda_timing_reset(rda_handle *H, int enable)
{
    if (!H) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    H->timing = enable; H->ev_used[0] = H->ev_used[1] = H->ev_used[2] = 0;
    return RDA_OK;
}
extern "C" int rda_timing_read(rda_handle *H, int which, double *total_ms, int *launches)
{
    if (!H || which < 0 || which > 2) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    double tot = 0; int n = 0;
    for (size_t i = 0; i + 1 < H->ev_used[which]; i += 2) {
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev[which][i], H->ev[which][i + 1]));
        tot += ms; ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return RDA_OK;
}

extern "C" int rda_timing_launches(rda_handle *H, int which, double *ms_out, int cap, int *launches)
{
    if (!H || which < 0 || which > 2 || (cap > 0 && !ms_out)) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    int n = 0;
    for (size_t i = 0; i + 1 < H->ev_used[which]; i += 2, ++n) {
        if (n >= cap) continue;
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev[which][i], H->ev[which][i + 1]));
        ms_out[n] = ms;
    }
    if (launches) *launches = n;
    return RDA_OK;
}

// The dual state lives stage-major on the device ([T+1][N][.], [T][N]); the accessors speak the reference's shapes ([N][T+1][.],
// [N][T]): transposed on the host (test / checkpoint path, not the hot path)
static void tr_to_host(const std::vector<double> &dev, double *host, size_t N, size_t TT, size_t W)
{
    for (size_t n = 0; n < N; ++n) for (size_t t = 0; t < TT; ++t) memcpy(host + (n * TT + t) * W, dev.data() + (t * N + n) * W, W * sizeof(double));
}
static void tr_to_dev(const double *host, std::vector<double> &dev, size_t N, size_t TT, size_t W)
{
    for (size_t n = 0; n < N; ++n) for (size_t t = 0; t < TT; ++t) memcpy(dev.data() + (t * N + n) * W, host + (n * TT + t) * W, W * sizeof(double));
}
extern "C" int rda_get_state(rda_handle *H, double *lam, double *mu, double *z, double *xi, double *zeta,
                             double *dis, double *a_lam, double *b_lam)
{
    if (!H) return RDA_ERR_ARG;
    Dev &d = H->d; const size_t T = d.c.T, N = d.c.N, E = d.c.E, R = d.c.R;
    HIPCHK(hipStreamSynchronize(H->stream));
    std::vector<double> tmp;
    auto get = [&](const double *dev, double *host, size_t TT, size_t W) -> int {
        tmp.resize(N * TT * W);
        HIPCHK(hipMemcpy(tmp.data(), dev, N * TT * W * sizeof(double), hipMemcpyDeviceToHost));
        tr_to_host(tmp, host, N, TT, W);
        return RDA_OK;
    };
    if (lam && get(d.lam, lam, T + 1, E)) return RDA_ERR_HIP;
    if (mu && get(d.mu, mu, T + 1, R)) return RDA_ERR_HIP;
    if (z && get(d.z, z, T, 1)) return RDA_ERR_HIP;
    if (xi && get(d.xi, xi, T + 1, 2)) return RDA_ERR_HIP;
    if (zeta && get(d.zeta, zeta, T, 1)) return RDA_ERR_HIP;
    if (dis) HIPCHK(hipMemcpy(dis, d.dis, T * sizeof(double), hipMemcpyDeviceToHost));
    if (a_lam || b_lam) {
        double *ta = nullptr, *tb = nullptr;
        if (dalloc(&ta, N * (T + 1) * 2) || dalloc(&tb, N * (T + 1))) return RDA_ERR_HIP;
        hipLaunchKernelGGL(k_products_get, dim3(64), dim3(256), 0, H->stream, d, ta, tb);
        HIPCHK(hipStreamSynchronize(H->stream));
        if (a_lam) HIPCHK(hipMemcpy(a_lam, ta, N * (T + 1) * 2 * sizeof(double), hipMemcpyDeviceToHost));
        if (b_lam) HIPCHK(hipMemcpy(b_lam, tb, N * (T + 1) * sizeof(double), hipMemcpyDeviceToHost));
        dev_free(ta); dev_free(tb);
    }
    return RDA_OK;
}

extern "C" int rda_set_state(rda_handle *H, const double *lam, const double *mu, const double *z,
                             const double *xi, const double *zeta, const double *dis,
                             const double *a_lam, const double *b_lam)
{
    if (!H) return RDA_ERR_ARG;
    if (shard_frozen(H)) return RDA_ERR_UNSUPPORTED;
    Dev &d = H->d; const size_t T = d.c.T, N = d.c.N, E = d.c.E, R = d.c.R;
    HIPCHK(hipStreamSynchronize(H->stream));
    std::vector<double> tmp;
    auto put = [&](const double *host, double *dev, size_t TT, size_t W) -> int {
        tmp.resize(N * TT * W);
        tr_to_dev(host, tmp, N, TT, W);
        HIPCHK(hipMemcpy(dev, tmp.data(), N * TT * W * sizeof(double), hipMemcpyHostToDevice));
        return RDA_OK;
    };
    if (lam && put(lam, d.lam, T + 1, E)) return RDA_ERR_HIP;
    if (mu && put(mu, d.mu, T + 1, R)) return RDA_ERR_HIP;
    if (z && put(z, d.z, T, 1)) return RDA_ERR_HIP;
    if (xi && put(xi, d.xi, T + 1, 2)) return RDA_ERR_HIP;
    if (zeta && put(zeta, d.zeta, T, 1)) return RDA_ERR_HIP;
    if (dis) HIPCHK(hipMemcpy(d.dis, dis, T * sizeof(double), hipMemcpyHostToDevice));
    double *ta = nullptr, *tb = nullptr;
    if (a_lam) { if (dalloc(&ta, N * (T + 1) * 2)) return RDA_ERR_HIP; HIPCHK(hipMemcpy(ta, a_lam, N * (T + 1) * 2 * sizeof(double), hipMemcpyHostToDevice)); }
    if (b_lam) { if (dalloc(&tb, N * (T + 1))) return RDA_ERR_HIP; HIPCHK(hipMemcpy(tb, b_lam, N * (T + 1) * sizeof(double), hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_products_set, dim3(64), dim3(256), 0, H->stream, d, ta, tb);
    int rc = terms_rebuild(H);                          // block sums / near masks of the rewritten terms
    HIPCHK(hipStreamSynchronize(H->stream));
    dev_free(ta); dev_free(tb);
    return rc;
}


// ---- obstacle sharding ------------------------------------------------------------------------------
// shard slots past the last obstacle (N % P != 0): terms the su-problem must not see.  a = g = 0 and the offset `ee` so low that
// the hinge margin a'p - (blam + ee) - d is astronomically positive: never active, screened out, no rotation term, no residual.
__global__ void k_dead_slots(Dev d)
{
    const int T = d.c.T;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.P * d.Nloc * T; i += gridDim.x * blockDim.x) {
        const int r = i / (d.Nloc * T), k = i % (d.Nloc * T), nl = k % d.Nloc;
        if (r * d.Nloc + nl >= d.c.N) { coef_arr(d, r, 3)[k] = -1e30; coef_arr(d, r, 8)[k] = -1e30; }      // (their block partials: zero sums, bit clear)
    }
}
extern "C" int rda_shard_config(rda_handle *H, int rank, int world)
{
    if (!H || world < 1 || rank < 0 || rank >= world) return RDA_ERR_ARG;
    if (H->d.c.N % world != 0 && !H->d.c.accelerated) return RDA_ERR_UNSUPPORTED;    // padded shards rely on the hinge of the accelerated cost
    if (world > 1 && H->follow) return RDA_ERR_UNSUPPORTED;                          // duals_follow moves rows between slots = between ranks
    HIPCHK(hipStreamSynchronize(H->stream));
    Dev &d = H->d;
    dev_free(d.coef); d.coef = nullptr; dev_free(d.coefL); d.coefL = nullptr;
    // su_pre = 0 (the su set-up evaluates every term itself) reads g = G'mu + xi of every slot, which lives in the LOCAL chunk and is
    // never gathered: with more than one rank the reduced form (block sums / near masks, part of the gathered chunk) is the only one
    if (world > 1) d.su_pre = 1;
    d.P = world; d.rank = rank; d.Nloc = (d.c.N + world - 1) / world; d.J = (d.Nloc + GS - 1) / GS; d.chunk = chunk_doubles(d.c.T, d.Nloc);
    d.lchunk = lchunk_doubles(d.c.T, d.Nloc);
    const int first = rank * d.Nloc;
    d.Nlive = first >= d.c.N ? 0 : (d.c.N - first < d.Nloc ? d.c.N - first : d.Nloc);
    if (dalloc(&d.coef, d.chunk * world) || dalloc(&d.coefL, d.lchunk * world)) return RDA_ERR_HIP;      // (dalloc zeroes)
    if (d.Nloc * world != d.c.N) hipLaunchKernelGGL(k_dead_slots, dim3(64), dim3(256), 0, H->stream, d);
    int rc = terms_rebuild(H);                          // (the duals of a handle that is re-sharded are NOT re-condensed: shard before the first step)
    if (rc != RDA_OK) return rc;
    HIPCHK(hipStreamSynchronize(H->stream));
    return RDA_OK;
}
static void *rccl_sym(rda_handle *H, const char *name);
// ranks of the handle's communicator as RCCL counts them (ncclCommCount); 0 without a communicator
extern "C" int rda_shard_comm_count(rda_handle *H)
{
    if (!H) return RDA_ERR_ARG;
    if (!H->comm) return 0;
    typedef int (*fn)(void *, int *);
    fn f = (fn)rccl_sym(H, "ncclCommCount");
    int n = 0;
    if (!f || f(H->comm, &n) != 0) return RDA_ERR_HIP;
    return n;
}
extern "C" int rda_shard_chunk_doubles(rda_handle *H) { return H ? (int)H->d.chunk : RDA_ERR_ARG; }
extern "C" int rda_shard_get_chunk(rda_handle *H, double *host)
{
    if (!H || !host) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(host, H->d.coef + (size_t)H->d.rank * H->d.chunk, H->d.chunk * sizeof(double), hipMemcpyDeviceToHost));
    return RDA_OK;
}
extern "C" int rda_shard_set_chunks(rda_handle *H, const double *host_all)
{
    if (!H || !host_all) return RDA_ERR_ARG;
    HIPCHK(hipStreamSynchronize(H->stream));
    HIPCHK(hipMemcpy(H->d.coef, host_all, H->d.chunk * H->d.P * sizeof(double), hipMemcpyHostToDevice));
    return RDA_OK;
}
static void *rccl_sym(rda_handle *H, const char *name)
{
    if (!H->nccl_lib) {
        // The RCCL the process ALREADY has, if any: a host that imported torch brought its own librccl.so.1 (and HIP runtime) with it,
        // and a second copy next to it breaks - default-visibility symbols of the second resolve into the first (measured:
        // ncclCommInitRank of /opt/rocm's copy fails with ncclUnhandledCudaError once torch's is mapped).  RTLD_NOLOAD matches by SONAME.
        H->nccl_lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!H->nccl_lib) H->nccl_lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!H->nccl_lib) H->nccl_lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!H->nccl_lib) H->nccl_lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    return H->nccl_lib ? dlsym(H->nccl_lib, name) : nullptr;
}
extern "C" int rda_shard_unique_id(rda_handle *H, void *out128)
{
    if (!H || !out128) return RDA_ERR_ARG;
    typedef int (*fn)(void *);
    fn f = (fn)rccl_sym(H, "ncclGetUniqueId");
    if (!f) return RDA_ERR_UNSUPPORTED;
    return f(out128) == 0 ? RDA_OK : RDA_ERR_HIP;
}
extern "C" int rda_shard_comm_init(rda_handle *H, const void *uid128)
{
    if (!H || !uid128 || H->d.P < 1) return RDA_ERR_ARG;     // a one-rank communicator is legal (plumbing check on a 1-GPU box)
    struct uid_t { char b[128]; } uid;
    memcpy(&uid, uid128, 128);
    typedef int (*init_fn)(void **, int, uid_t, int);
    init_fn f = (init_fn)rccl_sym(H, "ncclCommInitRank");
    H->p_allgather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))rccl_sym(H, "ncclAllGather");
    H->p_comm_destroy = (int (*)(void *))rccl_sym(H, "ncclCommDestroy");
    if (!f || !H->p_allgather) return RDA_ERR_UNSUPPORTED;
    int rc = f(&H->comm, H->d.P, uid, H->d.rank);
    if (rc != 0) { fprintf(stderr, "librda_hip: ncclCommInitRank failed (%d)\n", rc); H->comm = nullptr; return RDA_ERR_HIP; }
    if (!H->h_verdict) { HIPCHK(hipHostMalloc((void **)&H->h_verdict, sizeof(unsigned long long))); *H->h_verdict = 0; }
    return RDA_OK;
}

// ---- host-driven ADMM iteration (exchange done by the caller: tests, gloo, MPI ...) -----------------------
extern "C" int rda_admm_begin(rda_handle *H, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed)
{
    if (!H || !nom_s || !nom_u || !ref_s) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    H->stepped = 1;
    HIPCHK(hipStreamSynchronize(H->stream));
    memcpy(H->h_step, nom_s, ns * sizeof(double));
    memcpy(H->h_step + ns, nom_u, nu * sizeof(double));
    memcpy(H->h_step + ns + nu, ref_s, ns * sizeof(double));
    H->h_step[ns + nu + ns] = ref_speed;
    HIPCHK(hipMemcpyAsync(H->d_step, H->h_step, (2 * ns + nu + 1) * sizeof(double), hipMemcpyHostToDevice, H->stream));
    hipLaunchKernelGGL(k_begin, dim3(1), dim3(64), 0, H->stream, H->d);
    HIPCHK(hipGetLastError());
    return RDA_OK;
}
extern "C" int rda_admm_su(rda_handle *H, int it, int *stopped)
{
    if (!H || it < 0) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    H->admm_it = it;
    Dev d = H->d;
    d.ref = H->d_step + ns + nu; d.ref_speed = H->d_step + ns + nu + ns;
    RDA_SU_DISPATCH((int)T, hipLaunchKernelGGL(k_su<TT>, dim3(1), dim3(su::NT), H->su_lds, H->stream, d, it, H->d_step, H->d_step + ns, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0}));
    HIPCHK(hipGetLastError());
    if (stopped) {
        Ctrl c;
        HIPCHK(hipMemcpyAsync(&c, d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, H->stream));
        HIPCHK(hipStreamSynchronize(H->stream));
        *stopped = c.stop;
    }
    return RDA_OK;
}
extern "C" int rda_admm_lammuz(rda_handle *H)
{
    if (!H) return RDA_ERR_ARG;
    Dev d = H->d;
    launch_lammuz(H, d, H->admm_it, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0});
    HIPCHK(hipGetLastError());
    return RDA_OK;
}
extern "C" int rda_admm_finish(rda_handle *H, double *out_u, double *out_s, rda_info *info)
{
    if (!H || !out_u || !out_s) return RDA_ERR_ARG;
    const size_t T = H->d.c.T, ns = 3 * (T + 1), nu = 2 * T;
    Dev d = H->d;
    { int rc = launch_finish(H, d, make_fin(H, H->d_out_u, H->d_out_s, H->d_info)); if (rc != RDA_OK) return rc; }
    { int rc = fetch_result(H); if (rc != RDA_OK) return rc; }
    memcpy(out_u, H->h_out, nu * sizeof(double));
    memcpy(out_s, H->h_out + nu, ns * sizeof(double));
    if (info) *info = *H->h_info;
    return RDA_OK;
}

// ------------------------------------------------------------------------------------------------
// Fleet: B independent egos (BASELINE config C5) advanced by ONE set of launches per ADMM iteration.  Every member keeps
// its own handle (state, obstacles, trace) - the fleet only holds a device array of their `Dev` records and a table of
// per-ego input / output locations.  The grid gets an ego dimension: k_su runs B workgroups (one CU each) side by side,
// k_lammuz B * N*T/4 workgroups, so the launch fills the 256 CUs that a single ego cannot.
struct EgoIO {            // base pointers, indexed by the step number k inside the kernels
    const double *s, *u, *ref, *speed; double *out_u, *out_s; rda_info *info;
};


template <int TT> __global__ __launch_bounds__(su::NT) void k_su_fleet(const Dev *devs, const EgoIO *io, int it, int k)
{
    const Dev &d = devs[blockIdx.x];
    const EgoIO e = io[blockIdx.x];
    const size_t ns = 3 * (d.c.T + 1), nu = 2 * d.c.T;
    su_body<TT>(d, it, e.s + k * ns, e.u + k * nu, e.ref + k * ns, e.speed + k);
}

// a member without obstacles (Q9) or of a shape the packed body does not take runs the one-per-wave body on the first quarter
// of the (packed-size) grid... kept simple: the fleet uses the packed kernel only when every member can
__device__ __forceinline__ Fin fleet_fin(const Dev &d, const EgoIO &e, int k)
{
    const size_t ns = 3 * (d.c.T + 1), nu = 2 * d.c.T;
    return Fin{ e.out_u + k * nu, e.out_s + k * ns, e.info + k, nullptr, 0, 0 };
}
__global__ __launch_bounds__(256) void k_lammuz_fleet(const Dev *devs, int it) { lammuz_body(devs[blockIdx.y], blockIdx.x, it); }
__global__ __launch_bounds__(64 * GS / 4, 2) void k_lammuz_fleet_rows(const Dev *devs, const EgoIO *io, int it, int k)
{
    lammuz_body_rows<0>(devs[blockIdx.y], blockIdx.x, gridDim.x, it, fleet_fin(devs[blockIdx.y], io[blockIdx.y], k));
}
#ifndef LMZ_FLEET_FAST_OCC
#define LMZ_FLEET_FAST_OCC 3
#endif
__global__ __launch_bounds__(64 * GS / 4, LMZ_FLEET_FAST_OCC) void k_lammuz_fleet_rows_fast(const Dev *devs, int it) { lammuz_body_rows<1>(devs[blockIdx.y], blockIdx.x, gridDim.x, it, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0}); }
__global__ __launch_bounds__(64 * GS / 4, 1) void k_lammuz_fleet_enum(const Dev *devs, int it) { lammuz_body_rows<2>(devs[blockIdx.y], blockIdx.x, gridDim.x, it, Fin{nullptr, nullptr, nullptr, nullptr, 0, 0}); }
__global__ __launch_bounds__(256) void k_lmz_finalize_fleet(const Dev *devs, const EgoIO *io, int it, int k)
{
    finalize_body(devs[blockIdx.y], blockIdx.x, gridDim.x, it, fleet_fin(devs[blockIdx.y], io[blockIdx.y], k));
}

__global__ __launch_bounds__(su::NT) void k_finish_fleet(const Dev *devs, const EgoIO *io, int k)
{
    const Dev &d = devs[blockIdx.x];
    const EgoIO e = io[blockIdx.x];
    const size_t ns = 3 * (d.c.T + 1), nu = 2 * d.c.T;
    const Fin f = { e.out_u + k * nu, e.out_s + k * ns, e.info + k, nullptr, 0, 0 };
    finish_body(d, f);
}

struct rda_fleet {
    int B;
    std::vector<rda_handle *> egos;
    hipStream_t stream;
    Dev *h_devs, *d_devs;                 // pinned mirror / device array
    EgoIO *h_io, *d_io_step, *d_io_trace;
    double *h_in, *d_in;                  // step path, per ego: nom_s | nom_u | ref | speed
    double *h_out, *d_out;                // per ego: u | s
    rda_info *h_info, *d_info;
    hipEvent_t ev;
    int T, iter_num, J, rows, lmz_split;
    size_t su_lds;
    // tracked stepping (device-side pre_process), allocated on first use
    track::In *h_trk_in, *d_trk_in; track::Out *h_trk_out, *d_trk_out;
    double **h_paths, **d_paths; int *h_lens, *d_lens; EgoIO *h_io_track, *d_io_track;
    // rda_fleet_scene_resort (allocated on first use): the members' scene arguments, their robots' positions; rob_pending: a copy out of h_rob may be queued
    scene::Args *h_sc, *d_sc; double *h_rob, *d_rob; int rob_pending;
};

extern "C" void rda_fleet_destroy(rda_fleet *F)
{
    if (!F) return;
    (void)hipStreamSynchronize(F->stream);
    void *dp[] = { F->d_devs, F->d_io_step, F->d_io_trace, F->d_in, F->d_out, F->d_info, F->d_trk_in, F->d_trk_out, F->d_paths, F->d_lens, F->d_io_track, F->d_sc, F->d_rob };
    for (void *q : dp) dev_free(q);
    void *hp[] = { F->h_devs, F->h_io, F->h_in, F->h_out, F->h_info, F->h_trk_in, F->h_trk_out, F->h_paths, F->h_lens, F->h_io_track, F->h_sc, F->h_rob };
    for (void *q : hp) if (q) (void)hipHostFree(q);
    (void)hipEventDestroy(F->ev);
    (void)hipStreamDestroy(F->stream);
    delete F;
}

extern "C" int rda_fleet_create(rda_handle *const *egos, int B, rda_fleet **out)
{
    if (!egos || B < 1 || !out) return RDA_ERR_ARG;
    for (int i = 0; i < B; ++i) {
        if (!egos[i]) return RDA_ERR_ARG;
        const rda_cfg &a = egos[0]->d.c, &b = egos[i]->d.c;
        // one grid for all members: the problem SHAPE must agree (weights, bounds, kinematics and robots may differ)
        if (a.T != b.T || a.N != b.N || a.E != b.E || a.R != b.R || a.iter_num != b.iter_num) return RDA_ERR_UNSUPPORTED;
        if (egos[i]->comm || egos[i]->d.P != 1) return RDA_ERR_UNSUPPORTED;        // egos are replicas, obstacle shards are not
        if (egos[i]->d.lmz_mode) return RDA_ERR_UNSUPPORTED;                       // the fused fleet launches run the enumeration kernels
    }
    rda_fleet *F = new rda_fleet();
    F->B = B; F->egos.assign(egos, egos + B);
    F->d_devs = nullptr; F->d_io_step = F->d_io_trace = nullptr; F->d_in = F->d_out = nullptr; F->d_info = nullptr;
    F->h_devs = nullptr; F->h_io = nullptr; F->h_in = F->h_out = nullptr; F->h_info = nullptr;
    const rda_cfg &c = egos[0]->d.c;
    F->T = c.T; F->iter_num = c.iter_num; F->J = (c.N + GS - 1) / GS; F->su_lds = egos[0]->su_lds;
    HIPCHK(hipStreamCreate(&F->stream));
    HIPCHK(hipEventCreateWithFlags(&F->ev, hipEventDisableTiming));
    const size_t T = c.T, ns = 3 * (T + 1), nu = 2 * T, nin = 2 * ns + nu + 1, nout = nu + ns;
    int rc = 0;
    rc |= dalloc(&F->d_devs, (size_t)B); rc |= dalloc(&F->d_io_step, (size_t)B); rc |= dalloc(&F->d_io_trace, (size_t)B);
    rc |= dalloc(&F->d_in, B * nin); rc |= dalloc(&F->d_out, B * nout); rc |= dalloc(&F->d_info, (size_t)B);
    if (rc) { rda_fleet_destroy(F); return RDA_ERR_HIP; }
    HIPCHK(hipHostMalloc((void **)&F->h_devs, B * sizeof(Dev)));
    HIPCHK(hipHostMalloc((void **)&F->h_io, B * sizeof(EgoIO)));
    HIPCHK(hipHostMalloc((void **)&F->h_in, B * nin * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&F->h_out, B * nout * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&F->h_info, B * sizeof(rda_info)));
    memset(F->h_devs, 0, B * sizeof(Dev));
    for (int i = 0; i < B; ++i) {
        EgoIO &e = F->h_io[i];
        e.s = F->d_in + i * nin; e.u = e.s + ns; e.ref = e.u + nu; e.speed = e.ref + ns;
        e.out_u = F->d_out + i * nout; e.out_s = e.out_u + nu; e.info = F->d_info + i;
    }
    HIPCHK(hipMemcpy(F->d_io_step, F->h_io, B * sizeof(EgoIO), hipMemcpyHostToDevice));
    RDA_SU_DISPATCH((int)T, HIPCHK(hipFuncSetAttribute((const void *)k_su_fleet<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F->su_lds)));
    HIPCHK(hipFuncSetAttribute((const void *)k_finish_fleet, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F->su_lds));
    *out = F;
    return RDA_OK;
}

// members' records (obstacle count, staged pointers, weights may have changed since the last call) -> device; the fleet
// stream then waits for whatever the members still have in flight on their own streams (obstacle uploads)
static int fleet_refresh(rda_fleet *F)
{
    bool changed = false;
    for (int i = 0; i < F->B; ++i) {
        if (memcmp(&F->h_devs[i], &F->egos[i]->d, sizeof(Dev)) != 0) changed = true;
        // the fleet's stream must run behind whatever a member still has queued on its OWN stream (an upload, a solo step).  An idle member stream needs no
        // edge: the event record + wait pair per member cost ~8 us of host time each, twice per fleet tick (re-sort + step) = 1 ms of a 4.2 ms tick of the
        // 64-ego closed loop with the GPU idle meanwhile (rocprofv3 kernel trace cut into ticks, tools/experiments/fleet_tick_trace.sh, round 6)
        const hipError_t q = hipStreamQuery(F->egos[i]->stream);
        if (q == hipSuccess) continue;
        if (q != hipErrorNotReady) { fprintf(stderr, "librda_hip: hipStreamQuery failed: %s (%s:%d)\n", hipGetErrorString(q), __FILE__, __LINE__); return RDA_ERR_HIP; }
        (void)hipGetLastError();                            // ("not ready" is an answer, not an error to be found by a later hipGetLastError)
        HIPCHK(hipEventRecord(F->ev, F->egos[i]->stream));
        HIPCHK(hipStreamWaitEvent(F->stream, F->ev, 0));
    }
    F->rows = 1;
    F->lmz_split = 1;
    for (int i = 0; i < F->B; ++i) if (!F->egos[i]->lmz_split) F->lmz_split = 0;
    for (int i = 0; i < F->B; ++i) if (!F->egos[i]->d.rows || !F->egos[i]->d.obstacle_num) F->rows = 0;
    if (changed) {
        HIPCHK(hipStreamSynchronize(F->stream));            // an earlier copy out of the pinned mirror may still be queued
        for (int i = 0; i < F->B; ++i) memcpy(&F->h_devs[i], &F->egos[i]->d, sizeof(Dev));
        HIPCHK(hipMemcpyAsync(F->d_devs, F->h_devs, F->B * sizeof(Dev), hipMemcpyHostToDevice, F->stream));
    }
    return RDA_OK;
}

static int fleet_enqueue(rda_fleet *F, const EgoIO *io, int k)
{
    const int B = F->B;
    for (int it = 0; it < F->iter_num; ++it) {          // iteration 0 resets every member's control block (su_body)
        RDA_SU_DISPATCH(F->T, hipLaunchKernelGGL(k_su_fleet<TT>, dim3(B), dim3(su::NT), F->su_lds, F->stream, F->d_devs, io, it, k));
        constexpr int NTH = 64 * GS / 4;
        const int nbr = F->T * F->J, nfin = (nbr + FPB - 1) / FPB, nbp = packed_grid(F->T, F->J);       // one workgroup per (stage, GS-slot block), XCD-aware order
        if (F->rows && F->lmz_split) {
            hipLaunchKernelGGL(k_lammuz_fleet_rows_fast, dim3(nbp, B), dim3(NTH), 0, F->stream, F->d_devs, it);
            int ne = nbr / 8; if (ne < 8) ne = 8; if (ne > 128) ne = 128;
            hipLaunchKernelGGL(k_lammuz_fleet_enum, dim3(ne, B), dim3(NTH), 0, F->stream, F->d_devs, it);
            hipLaunchKernelGGL(k_lmz_finalize_fleet, dim3(nfin, B), dim3(256), 0, F->stream, F->d_devs, io, it, k);
        } else if (F->rows) hipLaunchKernelGGL(k_lammuz_fleet_rows, dim3(nbp, B), dim3(NTH), 0, F->stream, F->d_devs, io, it, k);
        else {
            hipLaunchKernelGGL(k_lammuz_fleet, dim3(nbr * GS / 4, B), dim3(256), 0, F->stream, F->d_devs, it);
            hipLaunchKernelGGL(k_lmz_finalize_fleet, dim3(nfin, B), dim3(256), 0, F->stream, F->d_devs, io, it, k);
        }
    }
    hipLaunchKernelGGL(k_finish_fleet, dim3(B), dim3(su::NT), F->su_lds, F->stream, F->d_devs, io, k);
    HIPCHK(hipGetLastError());
    return RDA_OK;
}

// one synchronous MPC step of every member: inputs / outputs are the per-ego arrays of rda_step, concatenated
extern "C" int rda_fleet_step(rda_fleet *F, const double *nom_s, const double *nom_u, const double *ref_s, const double *ref_speed,
                              double *out_u, double *out_s, rda_info *info)
{
    if (!F || !nom_s || !nom_u || !ref_s || !ref_speed || !out_u || !out_s) return RDA_ERR_ARG;
    const size_t T = F->T, ns = 3 * (T + 1), nu = 2 * T, nin = 2 * ns + nu + 1, nout = nu + ns, B = F->B;
    for (size_t i = 0; i < B; ++i) {
        double *q = F->h_in + i * nin;
        memcpy(q, nom_s + i * ns, ns * sizeof(double)); memcpy(q + ns, nom_u + i * nu, nu * sizeof(double));
        memcpy(q + ns + nu, ref_s + i * ns, ns * sizeof(double)); q[2 * ns + nu] = ref_speed[i];
    }
    int rc = fleet_refresh(F);
    if (rc != RDA_OK) return rc;
    HIPCHK(hipMemcpyAsync(F->d_in, F->h_in, B * nin * sizeof(double), hipMemcpyHostToDevice, F->stream));
    rc = fleet_enqueue(F, F->d_io_step, 0);
    if (rc != RDA_OK) return rc;
    HIPCHK(hipMemcpyAsync(F->h_out, F->d_out, B * nout * sizeof(double), hipMemcpyDeviceToHost, F->stream));
    HIPCHK(hipMemcpyAsync(F->h_info, F->d_info, B * sizeof(rda_info), hipMemcpyDeviceToHost, F->stream));
    HIPCHK(hipStreamSynchronize(F->stream));
    F->rob_pending = 0;
    for (rda_handle *Hm : F->egos) Hm->pending_scene = 0;      // their staged scenes have been consumed
    for (size_t i = 0; i < B; ++i) {
        memcpy(out_u + i * nu, F->h_out + i * nout, nu * sizeof(double));
        memcpy(out_s + i * ns, F->h_out + i * nout + nu, ns * sizeof(double));
        if (info) info[i] = F->h_info[i];
    }
    return RDA_OK;
}

// device-side pre_process of every member (one wave per ego), then the fleet step on what it wrote
__global__ void k_track_fleet(const Dev *devs, const EgoIO *io, const track::In *ins, double *const *paths, const int *lens,
                              track::Out *outs, int B)
{
    __shared__ double win[track::LDS_DOUBLES];
    const int b = blockIdx.x;
    if (b >= B) return;
    const Dev &d = devs[b];
    track::Ego e;
    e.path = paths[b]; e.L = lens[b]; e.nom_u = io[b].u; e.nom_s = const_cast<double *>(io[b].s); e.ref = const_cast<double *>(io[b].ref);
    e.speed = const_cast<double *>(io[b].speed);
    e.T = d.c.T; e.dynamics = d.c.dynamics; e.dt = d.c.dt; e.wheelbase = d.c.L;
    track::run(e, ins[b], outs[b], win, threadIdx.x);
}

// every member's raw scene in one call (member i owns counts[i] consecutive entries of the arrays): staged on the members'
// streams without waiting - the next fleet step orders itself behind them (fleet_refresh) and ends with a synchronisation
extern "C" int rda_fleet_upload_scenes(rda_fleet *F, const int32_t *counts, const int32_t *kind, const int32_t *nvert, const double *geom,
                                       const double *vel, const double *robot_xy, const int32_t *order)
{
    if (!F || !counts || !robot_xy || !order) return RDA_ERR_ARG;
    size_t off = 0;
    for (int i = 0; i < F->B; ++i) {
        rda_handle *H = F->egos[i];
        const size_t E = H->d.c.E;
        const int n = counts[i];
        if (n < 0) return RDA_ERR_ARG;
        int rc = rda_upload_scene_async(H, n, kind ? kind + off : nullptr, nvert ? nvert + off : nullptr, geom ? geom + off * E * 2 : nullptr,
                                        vel ? vel + off * 2 : nullptr, robot_xy + 2 * i, order[i]);
        if (rc != RDA_OK) return rc;
        off += (size_t)n;
    }
    return RDA_OK;
}

// rda_scene_resort for every member in ONE launch set (round 6): the members' resident raw scenes (rda_upload_scene*, static obstacles) re-ranked about
// states[i * stride + 0..1] and their slots rebuilt on the fleet's stream; the next fleet step runs behind it.  Same device code per member as
// rda_scene_resort: bit-identical slots.  Members with rda_opts::duals_follow, without a resident scene, or inside a tick: RDA_ERR_UNSUPPORTED / _ARG.
extern "C" int rda_fleet_scene_resort(rda_fleet *F, const double *states, int stride)
{
    if (!F || !states || stride < 2) return RDA_ERR_ARG;
    const size_t B = F->B;
    for (rda_handle *H : F->egos) {
        if (H->sc_n <= 0 || H->d.obstacle_num == 0 || H->pending) return RDA_ERR_ARG;
        if (H->follow) return RDA_ERR_UNSUPPORTED;
    }
    if (!F->d_sc) {
        int rc = 0;
        rc |= dalloc(&F->d_sc, B); rc |= dalloc(&F->d_rob, 2 * B);
        if (rc) return RDA_ERR_HIP;
        HIPCHK(hipHostMalloc((void **)&F->h_sc, B * sizeof(scene::Args)));
        HIPCHK(hipHostMalloc((void **)&F->h_rob, 2 * B * sizeof(double)));
        memset((void *)F->h_sc, 0, B * sizeof(scene::Args));
    }
    int rc = fleet_refresh(F);
    if (rc != RDA_OK) return rc;
    bool changed = false;
    int nmax = 0, wmax = 0;
    for (size_t i = 0; i < B; ++i) {
        scene::Args a = F->egos[i]->sc_args;
        a.order = 1; a.robot_val = 0; a.rx = 0; a.ry = 0;
        if (memcmp(&a, &F->h_sc[i], sizeof(scene::Args)) != 0) changed = true;
        nmax = a.n > nmax ? a.n : nmax; wmax = a.N * a.nt > wmax ? a.N * a.nt : wmax;
    }
    if (changed || F->rob_pending) { HIPCHK(hipStreamSynchronize(F->stream)); F->rob_pending = 0; }
    if (changed) {
        for (size_t i = 0; i < B; ++i) { scene::Args a = F->egos[i]->sc_args; a.order = 1; a.robot_val = 0; a.rx = 0; a.ry = 0; memcpy((void *)&F->h_sc[i], &a, sizeof(a)); }
        HIPCHK(hipMemcpyAsync(F->d_sc, F->h_sc, B * sizeof(scene::Args), hipMemcpyHostToDevice, F->stream));
    }
    for (size_t i = 0; i < B; ++i) { F->h_rob[2 * i] = states[i * stride]; F->h_rob[2 * i + 1] = states[i * stride + 1]; }
    HIPCHK(hipMemcpyAsync(F->d_rob, F->h_rob, 2 * B * sizeof(double), hipMemcpyHostToDevice, F->stream));
    F->rob_pending = 1;
    hipLaunchKernelGGL(scene::k_keys_fleet, dim3((nmax + 255) / 256, (unsigned)B), dim3(256), 0, F->stream, (const scene::Args *)F->d_sc, (const double *)F->d_rob);
    hipLaunchKernelGGL(scene::k_rank_fleet, dim3((nmax + 15) / 16, (unsigned)B), dim3(256), 0, F->stream, (const scene::Args *)F->d_sc, (const double *)F->d_rob);
    hipLaunchKernelGGL(scene::k_build_fleet, dim3((wmax + 255) / 256, (unsigned)B), dim3(256), 0, F->stream, (const scene::Args *)F->d_sc, (const double *)F->d_rob);
    hipLaunchKernelGGL(k_prepare_fleet, dim3((unsigned)((wmax + 3) / 4), (unsigned)B), dim3(256), 0, F->stream, (const Dev *)F->d_devs);
    HIPCHK(hipGetLastError());
    return RDA_OK;
}

extern "C" int rda_fleet_step_tracked(rda_fleet *F, const double *states, const double *ref_speed, const int32_t *cur_index,
                                      double threshold, int ind_range, const double *nom_u,
                                      double *out_u, double *out_s, rda_info *info, double *ref_out, int32_t *min_index, double *end_heading)
{
    if (!F || !states || !ref_speed || !cur_index || !out_u || !out_s || ind_range < 1) return RDA_ERR_ARG;
    const size_t T = F->T, ns = 3 * (T + 1), nu = 2 * T, nin = 2 * ns + nu + 1, nout = nu + ns, B = F->B;
    if (!F->d_trk_in) {
        int rc = 0;
        rc |= dalloc(&F->d_trk_in, B); rc |= dalloc(&F->d_trk_out, B); rc |= dalloc(&F->d_paths, B); rc |= dalloc(&F->d_lens, B);
        rc |= dalloc(&F->d_io_track, B);
        if (rc) return RDA_ERR_HIP;
        HIPCHK(hipHostMalloc((void **)&F->h_trk_in, B * sizeof(track::In)));
        HIPCHK(hipHostMalloc((void **)&F->h_trk_out, B * sizeof(track::Out)));
        HIPCHK(hipHostMalloc((void **)&F->h_paths, B * sizeof(double *)));
        HIPCHK(hipHostMalloc((void **)&F->h_lens, B * sizeof(int)));
        HIPCHK(hipHostMalloc((void **)&F->h_io_track, B * sizeof(EgoIO)));
        memset(F->h_paths, 0, B * sizeof(double *)); memset(F->h_lens, 0, B * sizeof(int)); memset(F->h_io_track, 0, B * sizeof(EgoIO));
    }
    // the result of the tick is written where the host reads it (round 6): k_finish_fleet and k_track_fleet store straight into the pinned blocks (write-only,
    // fire-and-forget stores over the link, complete at the end of their kernels) instead of three device-to-host copies queued behind the last launch
    // (~10 us each on the tick's critical path).  rda_opts::zero_copy = 0 on any member keeps the copies.
    bool zc = true;
    for (size_t i = 0; i < B; ++i) if (!F->egos[i]->zero_copy) zc = false;
    double *const out_base = zc ? F->h_out : F->d_out;
    rda_info *const info_base = zc ? F->h_info : F->d_info;
    bool tables = false;
    for (size_t i = 0; i < B; ++i) {
        rda_handle *H = F->egos[i];
        if (!H->d_path || cur_index[i] < 0 || cur_index[i] >= H->path_len) return RDA_ERR_ARG;
        track::In &in = F->h_trk_in[i];
        in.sx = states[3 * i]; in.sy = states[3 * i + 1]; in.sth = states[3 * i + 2]; in.speed = ref_speed[i]; in.threshold = threshold;
        in.cur_index = cur_index[i]; in.ind_range = ind_range;
        EgoIO e;
        e.s = F->d_in + i * nin; e.u = nom_u ? e.s + ns : H->d.u; e.ref = e.s + ns + nu; e.speed = e.ref + ns;
        e.out_u = out_base + i * nout; e.out_s = e.out_u + nu; e.info = info_base + i;
        if (memcmp(&e, &F->h_io_track[i], sizeof(EgoIO)) != 0 || F->h_paths[i] != H->d_path || F->h_lens[i] != H->path_len) tables = true;
    }
    int rc = fleet_refresh(F);
    if (rc != RDA_OK) return rc;
    if (tables) {
        HIPCHK(hipStreamSynchronize(F->stream));
        for (size_t i = 0; i < B; ++i) {
            rda_handle *H = F->egos[i];
            EgoIO &e = F->h_io_track[i];
            e.s = F->d_in + i * nin; e.u = nom_u ? e.s + ns : H->d.u; e.ref = e.s + ns + nu; e.speed = e.ref + ns;
            e.out_u = out_base + i * nout; e.out_s = e.out_u + nu; e.info = info_base + i;
            F->h_paths[i] = H->d_path; F->h_lens[i] = H->path_len;
        }
        HIPCHK(hipMemcpyAsync(F->d_io_track, F->h_io_track, B * sizeof(EgoIO), hipMemcpyHostToDevice, F->stream));
        HIPCHK(hipMemcpyAsync(F->d_paths, F->h_paths, B * sizeof(double *), hipMemcpyHostToDevice, F->stream));
        HIPCHK(hipMemcpyAsync(F->d_lens, F->h_lens, B * sizeof(int), hipMemcpyHostToDevice, F->stream));
    }
    if (nom_u) {
        for (size_t i = 0; i < B; ++i) memcpy(F->h_in + i * nin + ns, nom_u + i * nu, nu * sizeof(double));
        HIPCHK(hipMemcpy2DAsync(F->d_in + ns, nin * sizeof(double), F->h_in + ns, nin * sizeof(double), nu * sizeof(double), B,
                                hipMemcpyHostToDevice, F->stream));
    }
    HIPCHK(hipMemcpyAsync(F->d_trk_in, F->h_trk_in, B * sizeof(track::In), hipMemcpyHostToDevice, F->stream));
    hipLaunchKernelGGL(k_track_fleet, dim3((unsigned)B), dim3(64), 0, F->stream, F->d_devs, F->d_io_track, F->d_trk_in,
                       F->d_paths, F->d_lens, zc ? F->h_trk_out : F->d_trk_out, (int)B);
    rc = fleet_enqueue(F, F->d_io_track, 0);
    if (rc != RDA_OK) return rc;
    if (!zc) {
        HIPCHK(hipMemcpyAsync(F->h_out, F->d_out, B * nout * sizeof(double), hipMemcpyDeviceToHost, F->stream));
        HIPCHK(hipMemcpyAsync(F->h_info, F->d_info, B * sizeof(rda_info), hipMemcpyDeviceToHost, F->stream));
        HIPCHK(hipMemcpyAsync(F->h_trk_out, F->d_trk_out, B * sizeof(track::Out), hipMemcpyDeviceToHost, F->stream));
    }
    if (ref_out) HIPCHK(hipMemcpyAsync(F->h_in, F->d_in, B * nin * sizeof(double), hipMemcpyDeviceToHost, F->stream));
    HIPCHK(hipStreamSynchronize(F->stream));
    for (rda_handle *Hm : F->egos) Hm->pending_scene = 0;      // their staged scenes have been consumed
    for (size_t i = 0; i < B; ++i) {
        memcpy(out_u + i * nu, F->h_out + i * nout, nu * sizeof(double));
        memcpy(out_s + i * ns, F->h_out + i * nout + nu, ns * sizeof(double));
        if (info) info[i] = F->h_info[i];
        if (ref_out) memcpy(ref_out + i * ns, F->h_in + i * nin + ns + nu, ns * sizeof(double));
        if (min_index) min_index[i] = F->h_trk_out[i].min_index;
        if (end_heading) end_heading[i] = F->h_trk_out[i].end_heading;
    }
    return RDA_OK;
}

// steps k0 .. k1-1 of every member's uploaded trace (rda_upload_trace), no host synchronisation; results are read with
// rda_fetch_result on the members after rda_fleet_sync
extern "C" int rda_fleet_enqueue_range(rda_fleet *F, int k0, int k1)
{
    if (!F || k0 < 0 || k0 > k1) return RDA_ERR_ARG;
    for (int i = 0; i < F->B; ++i) {
        rda_handle *H = F->egos[i];
        if (k1 > H->K) return RDA_ERR_ARG;
        EgoIO &e = F->h_io[i];
        e.s = H->d_tr_s; e.u = H->d_tr_u; e.ref = H->d_tr_ref; e.speed = H->d_tr_speed;
        e.out_u = H->d_tr_out_u; e.out_s = H->d_tr_out_s; e.info = H->d_tr_info;
    }
    int rc = fleet_refresh(F);
    if (rc != RDA_OK) return rc;
    HIPCHK(hipMemcpyAsync(F->d_io_trace, F->h_io, F->B * sizeof(EgoIO), hipMemcpyHostToDevice, F->stream));
    for (int k = k0; k < k1; ++k) { rc = fleet_enqueue(F, F->d_io_trace, k); if (rc != RDA_OK) return rc; }
    return RDA_OK;
}

extern "C" int rda_fleet_sync(rda_fleet *F) { if (!F) return RDA_ERR_ARG; HIPCHK(hipStreamSynchronize(F->stream)); F->rob_pending = 0; return RDA_OK; }
extern "C" int rda_fleet_size(rda_fleet *F) { return F ? F->B : RDA_ERR_ARG; }

// ---- pure-function hooks ------------------------------------------------------------------------
extern "C" int rda_lammuz_batch(int B, int E, int R, const double *A, const double *b, const int32_t *cone,
                                const double *p, const double *phi, const double *G, const double *h,
                                const double *xi, const double *zeta, const double *dbar, double ro2, double delta,
                                int accelerated, double *lam, double *mu, double *z, double *cmh)
{
    if (B < 1 || E < 1 || E > RDA_EMAX || R < 1 || R > RDA_RMAX || E + R + 1 > 64) return RDA_ERR_UNSUPPORTED;
    if (rda_device_count() < 1) return RDA_ERR_NODEVICE;
    double *dA, *db, *dp, *dphi, *dG, *dh, *dxi, *dzeta, *ddbar, *dlam, *dmu, *dz, *dcmh; int *dcone;
    struct Cp { void **dst; const void *src; size_t bytes; };
    const size_t sB = (size_t)B;
    Cp ins[] = { {(void **)&dA, A, sB * E * 2 * 8}, {(void **)&db, b, sB * E * 8}, {(void **)&dcone, cone, sB * 4}, {(void **)&dp, p, sB * 2 * 8},
                 {(void **)&dphi, phi, sB * 8}, {(void **)&dG, G, (size_t)R * 2 * 8}, {(void **)&dh, h, (size_t)R * 8}, {(void **)&dxi, xi, sB * 2 * 8},
                 {(void **)&dzeta, zeta, sB * 8}, {(void **)&ddbar, dbar, sB * 8} };
    for (auto &c : ins) { HIPCHK(hipMalloc(c.dst, c.bytes)); HIPCHK(hipMemcpy(*c.dst, c.src, c.bytes, hipMemcpyHostToDevice)); }
    HIPCHK(hipMalloc((void **)&dlam, sB * E * 8)); HIPCHK(hipMalloc((void **)&dmu, sB * R * 8)); HIPCHK(hipMalloc((void **)&dz, sB * 8)); HIPCHK(hipMalloc((void **)&dcmh, sB * 4 * 8));
    rda_opts od; rda_opts_init(&od);                   // tie-break T1 of the defaults
    RobotCands rcands; rcands.nmv = robot_candidates(R, G, h, rcands.muc, rcands.rv, &rcands.nrv); rcands.centre = od.tie_centre ? 1 : 0;
    long long *dprof = nullptr;
#ifdef RDA_LMZ_PROF     // debug build only: phase cycles of sub-problem 0 on stderr
    HIPCHK(hipMalloc((void **)&dprof, 8 * sizeof(long long))); HIPCHK(hipMemset(dprof, 0, 8 * sizeof(long long)));
#endif
    hipLaunchKernelGGL(k_lammuz_batch, dim3((B + 3) / 4), dim3(256), 0, 0, B, E, R, dA, db, dcone, dp, dphi, dG, dh, dxi, dzeta, ddbar,
                       ro2, delta, accelerated, dlam, dmu, dz, dcmh, dprof, rcands);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if (dprof) {
        long long hp[8]; HIPCHK(hipMemcpy(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost));
        fprintf(stderr, "lammuz prof (ticks, sub-problem 0): setup=%lld p1.it0=%lld p1.it1=%lld p2.it0=%lld p2.it1=%lld argmin1=%lld argmin2=%lld total=%lld\n",
                hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
        dev_free(dprof);
    }
    HIPCHK(hipMemcpy(lam, dlam, sB * E * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(mu, dmu, sB * R * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(z, dz, sB * 8, hipMemcpyDeviceToHost));
    if (cmh) HIPCHK(hipMemcpy(cmh, dcmh, sB * 4 * 8, hipMemcpyDeviceToHost));
    for (auto &c : ins) dev_free(*c.dst);
    dev_free(dlam); dev_free(dmu); dev_free(dz); dev_free(dcmh);
    return RDA_OK;
}

extern "C" int rda_su_solve_opts(const rda_cfg *cfg, const rda_opts *opts, const double *nom_s, const double *nom_u, const double *ref_s,
                            double ref_speed, const double *a, const double *cc, const double *g,
                            const double *d0, double *s, double *u, double *dd, int32_t *ipm_iters)
{
    if (!cfg || cfg->T < 1 || cfg->T > RDA_TMAX || cfg->N < 1) return RDA_ERR_UNSUPPORTED;
    if (rda_device_count() < 1) return RDA_ERR_NODEVICE;
    const size_t T = cfg->T, N = cfg->N, ns = 3 * (T + 1), nu = 2 * T;
    std::vector<double> soa(6 * T * N, 0.0);
    for (size_t n = 0; n < N; ++n) for (size_t t = 0; t < T; ++t) {
        size_t k = t * N + n;
        soa[0 * T * N + k] = a[(n * T + t) * 2]; soa[1 * T * N + k] = a[(n * T + t) * 2 + 1];
        soa[2 * T * N + k] = cc[n * T + t];
        soa[4 * T * N + k] = g[(n * T + t) * 2]; soa[5 * T * N + k] = g[(n * T + t) * 2 + 1];
    }
    double *dsoa, *dns, *dnu, *dref, *dspeed, *dd0, *dos, *dou, *dod; int *dst;
    HIPCHK(hipMalloc((void **)&dsoa, soa.size() * 8)); HIPCHK(hipMemcpy(dsoa, soa.data(), soa.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&dns, ns * 8)); HIPCHK(hipMemcpy(dns, nom_s, ns * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&dnu, nu * 8)); HIPCHK(hipMemcpy(dnu, nom_u, nu * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&dref, ns * 8)); HIPCHK(hipMemcpy(dref, ref_s, ns * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&dspeed, 8)); HIPCHK(hipMemcpy(dspeed, &ref_speed, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&dd0, T * 8));
    if (d0) HIPCHK(hipMemcpy(dd0, d0, T * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&dos, ns * 8)); HIPCHK(hipMalloc((void **)&dou, nu * 8)); HIPCHK(hipMalloc((void **)&dod, T * 8));
    HIPCHK(hipMalloc((void **)&dst, 2 * sizeof(int)));
    su::Args ar;
    ar.c.T = cfg->T; ar.c.N = cfg->N; ar.c.dynamics = cfg->dynamics; ar.c.accelerated = cfg->accelerated;
    ar.c.dt = cfg->dt; ar.c.L = cfg->L; ar.c.umax0 = cfg->max_speed[0]; ar.c.umax1 = cfg->max_speed[1];
    ar.c.ab0 = cfg->acce_bound[0]; ar.c.ab1 = cfg->acce_bound[1]; ar.c.ws = cfg->ws; ar.c.wu = cfg->wu;
    ar.c.slack_gain = cfg->slack_gain; ar.c.max_sd = cfg->max_sd; ar.c.min_sd = cfg->min_sd; ar.c.ro1 = cfg->ro1; ar.c.ro2 = cfg->ro2;
    rda_opts od; rda_opts_init(&od);                   // the stop tolerances / switches: the caller's, else the defaults
    if (opts) od = *opts;
    ar.c.eps_u = cfg->eps_u; ar.c.tol_rd = od.su_tol[0] > 0 ? od.su_tol[0] : 1e-9; ar.c.tol_rp = od.su_tol[1] > 0 ? od.su_tol[1] : 1e-10; ar.c.tol_mu = od.su_tol[2] > 0 ? od.su_tol[2] : 1e-11;    // (same fallback as rda_create_opts)
    ar.split = od.su_split; ar.accept = od.su_accept; ar.first_attempt = od.su_first_attempt;
    ar.land = od.su_land ? 1 : 0; if (od.su_land_rho > 0) ar.land_rho = od.su_land_rho;
    if (od.su_land_tol[0] > 0 && od.su_land_tol[1] > 0 && od.su_land_tol[2] > 0) { ar.land_tol[0] = od.su_land_tol[0]; ar.land_tol[1] = od.su_land_tol[1]; ar.land_tol[2] = od.su_land_tol[2]; }
    ar.in_s = dns; ar.in_u = dnu; ar.ref = dref; ar.ref_speed = dspeed;
    ar.ax = dsoa; ar.ay = dsoa + T * N; ar.cb = dsoa + 2 * T * N; ar.gx = dsoa + 4 * T * N; ar.gy = dsoa + 5 * T * N;
    ar.P = 1; ar.Nloc = (int)N; ar.chunk = 0;
    ar.d_in = d0 ? dd0 : nullptr; ar.out_s = dos; ar.out_u = dou; ar.out_d = dod; ar.status = dst; ar.ipm_iters = dst + 1;
    long long *dprof = nullptr;
    if (od.su_prof) { HIPCHK(hipMalloc((void **)&dprof, su::PROF_WORDS * sizeof(long long))); HIPCHK(hipMemset(dprof, 0, su::PROF_WORDS * sizeof(long long))); }
    ar.prof = dprof;
    double *ddbg = nullptr;
    if (od.su_prof > 1) { HIPCHK(hipMalloc((void **)&ddbg, 400 * sizeof(double))); HIPCHK(hipMemset(ddbg, 0, 400 * sizeof(double))); }
    ar.dbg = ddbg;
    const size_t lds = su::lds_bytes((int)T);
    RDA_SU_DISPATCH((int)T, HIPCHK(hipFuncSetAttribute((const void *)k_su_hook<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)));
    RDA_SU_DISPATCH((int)T, hipLaunchKernelGGL(k_su_hook<TT>, dim3(1), dim3(su::NT), lds, 0, ar));
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    int st[2];
    HIPCHK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost));
    if (ddbg) {
        double hd[400]; HIPCHK(hipMemcpy(hd, ddbg, sizeof(hd), hipMemcpyDeviceToHost));
        for (int i = 0; i < 100 && (i == 0 || hd[4 * i + 3] != 0); ++i)
            fprintf(stderr, "it %d rdn %.3e rpn %.3e mu %.3e sc %.3e\n", i, hd[4 * i], hd[4 * i + 1], hd[4 * i + 2], hd[4 * i + 3]);
        dev_free(ddbg);
    }
    if (dprof) {
        long long hp[16]; HIPCHK(hipMemcpy(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost));
        fprintf(stderr, "su prof (cycles) iters=%d:", st[1]); for (int i = 0; i < 16; ++i) fprintf(stderr, " [%d]=%lld", i, hp[i]); fprintf(stderr, "\n");
        dev_free(dprof);
    }
    HIPCHK(hipMemcpy(s, dos, ns * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(u, dou, nu * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(dd, dod, T * 8, hipMemcpyDeviceToHost));
    if (ipm_iters) *ipm_iters = st[1];
    void *fr[] = { dsoa, dns, dnu, dref, dspeed, dd0, dos, dou, dod, dst };
    for (void *p : fr) dev_free(p);
    return st[0] == 0 ? 0 : 1;
}
extern "C" int rda_su_solve(const rda_cfg *cfg, const double *nom_s, const double *nom_u, const double *ref_s,
                            double ref_speed, const double *a, const double *cc, const double *g,
                            const double *d0, double *s, double *u, double *dd, int32_t *ipm_iters)
{
    return rda_su_solve_opts(cfg, nullptr, nom_s, nom_u, ref_s, ref_speed, a, cc, g, d0, s, u, dd, ipm_iters);
}
