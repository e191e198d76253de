/*
 * rda_hip.h - C-ABI of librda_hip.so, the MI355X-native RDA ADMM inner solver.
 *
 * Drop-in boundary: these entry points are what a ctypes / cffi binding inside
 * the reference's RDA_planner/rda_solver.py would call instead of CVXPY + pathos.  Plain
 * pointers and sizes only; every array is C-contiguous float64 (int32 for cone codes); host
 * pointers unless the name says `dev`.  The library owns all device memory and keeps no
 * caller pointer past a call.  Return value: 0 ok, >0 soft status, <0 hard error
 * (rda_strerror).  One HIP stream per handle; a handle is not thread-safe, distinct handles
 * are independent; no global state: everything that configures a solver travels in rda_cfg (the
 * reference's constructor arguments) and rda_opts (solver options that are not reference arguments),
 * both copied into the handle at creation.  The library reads NO environment variable (since round 5): the RDA_* switches of the
 * A/B tools are applied by the Python host package (rda_planner_amd.rda_solver.hip_options) to the rda_opts it hands over.
 *
 * reference interface replaced                       | entry point
 * ---------------------------------------------------+-------------------------------------
 * RDA_solver.__init__            rda_solver.py:18-61 | rda_create / rda_create_opts
 * assign_adjust_parameter        rda_solver.py:426   | rda_set_adjust
 * reset                          rda_solver.py:1060  | rda_reset
 * iterative_solve                rda_solver.py:573   | rda_step (host buffers in/out)
 * assign_obstacle_parameter      rda_solver.py:483   | rda_upload_obstacles
 * MPC.convert_rda_obstacle + sort mpc.py:189-218,440 | rda_upload_scene / rda_step_scene (caller-side obstacle pipeline on device)
 * rda_solver loop body           rda_solver.py:612   | rda_enqueue_step (device-resident inputs)
 * solve_parallel (pure function) rda_solver.py:743   | rda_lammuz_batch
 * su_prob_solve                  rda_solver.py:692   | rda_su_solve
 * para_*.value getters/setters   rda_solver.py:129   | rda_get_state / rda_set_state
 */
#ifndef RDA_HIP_H
#define RDA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RDA_EMAX 8     /* max_edge_num supported by the kernels */
#define RDA_RMAX 8     /* robot half-spaces supported           */
#define RDA_TMAX 64    /* receding horizon supported            */

typedef struct rda_cfg {
    int32_t T;            /* receding                      rda_solver.py:34  */
    int32_t N;            /* max_obs_num                   rda_solver.py:38  */
    int32_t E;            /* max_edge_num                  rda_solver.py:39  */
    int32_t R;            /* car_tuple.G.shape[0]          rda_solver.py:99  */
    int32_t dynamics;     /* 0 acker, 1 diff, 2 omni       rda_solver.py:446 */
    int32_t accelerated;  /*                               rda_solver.py:47  */
    int32_t iter_num;     /*                               rda_solver.py:42  */
    int32_t robot_norm2;  /* car_tuple.cone_type=='norm2' (rda_solver.py:1034-1039): solved by the interior-point LamMuZ kernel */
    double dt, L;
    double max_speed[2];  /* rda_solver.py:37 */
    double acce_bound[2]; /* max_acce*dt, rda_solver.py:44 */
    double iter_threshold;
    double ws, wu;                                /* rda_solver.py:218-219 */
    double slack_gain, max_sd, min_sd, ro1, ro2;  /* rda_solver.py:196-201 */
    double delta;         /* tie-break T1 (clearance reward), 1e-6 */
    double eps_u;         /* tie-break for an undetermined steering column, 1e-8 */
} rda_cfg;

typedef struct rda_info {
    double resi_dual, resi_pri;   /* rda_solver.py:607-608 */
    int32_t iters;                /* ADMM iterations executed (early stop, :594) */
    int32_t su_status;            /* bit i: su-solve of iteration i not converged -> nominal kept (:699) */
    int32_t su_ipm_iters;         /* interior-point iterations, summed */
    int32_t lmz_fail;             /* (obstacle, stage) sub-problems that kept their previous duals - non-finite data or result; their
                                     residual is inf, which blocks the early stop like a non-OPTIMAL solve does (:781-793, :816-826) */
} rda_info;

typedef struct rda_handle rda_handle;

enum { RDA_OK = 0, RDA_ERR_ARG = -1, RDA_ERR_UNSUPPORTED = -2, RDA_ERR_HIP = -3, RDA_ERR_NODEVICE = -4 };

/* Solver options that are NOT arguments of the reference's constructor.  rda_opts_init fills the defaults below - and nothing else: the
 * library reads no environment variable (round 5; the RDA_* names in the right-hand column are switches of the Python host package,
 * rda_planner_amd.rda_solver.hip_options, which applies them to the struct it hands over).  A caller changes what it wants and hands the
 * struct to rda_create_opts, which copies it.  The first block is what a user may want to choose; the rest are the A/B switches of the kernels' restructurings -
 * every one of them changes where / when work is done, never the result (tests/test_gpu_switches.py). */
typedef struct rda_opts {
    int32_t lmz_mode;        /* 0 (default) support enumeration with the tie-breaks T1-T3 of DESIGN.md 2; 1 interior point ending on the
                                central path of the reference's own cone program at barrier parameter lmz_mu (interior duals like the
                                reference's solver returns, any combination of cones).  A norm2 robot always uses 1.     RDA_LMZ_MODE */
    int32_t tie_centre;      /* tie-break T1 in the slack regime: 1 (default) duals of the central separating normal, 0 max clearance.  RDA_TIE_CENTRE */
    double  lmz_mu;          /* 1e-6                                                                                      RDA_LMZ_MU */
    double  su_tol[3];       /* interior-point stop of the su-problem: |r_dual|_inf <= [0] (1+|grad|_inf), |r_prim|_inf <= [1], mean
                                complementarity <= [2] (1+|grad|_inf); 1e-9, 1e-10, 1e-11                                  RDA_SU_TOL */
    double  su_tol_early[3]; /* [0, 0, 0 = off] opt-in: the su-problems of the ADMM iterations BEFORE the last one of a step (it < iter_num - 1)
                                stop at these instead of su_tol.  The reference's own solver stops at ECOS defaults (1e-8 class) in EVERY
                                iteration; su_tol is 1000 x tighter because the parity tolerance is stated against it.  "1e-6,1e-7,1e-8":
                                two interior-point iterations fewer per su-problem.  A step that stops early returns the control of an
                                su-problem solved to THIS tolerance (some 1e-3 from the exact one where an inequality is weakly active,
                                tests/test_oracle_su.py), so the stated tolerance TOL_U does not hold with it.            RDA_SU_TOL_EARLY */
    double  su_hard_warm[2]; /* [1, 1e-3; 0, 0 = off] slack floor / barrier parameter of the warm attempts of a step that follows an UNCONVERGED step
                                (its ADMM used all iter_num iterations with a residual above iter_threshold) while consecutive su-problems are far
                                apart (the last solve's first iterate - the previous solution with its multipliers - had a relative dual residual
                                above 1e-2): a point well inside the boxes with the previous multipliers, a barrier of their own for the rows of the
                                safety distance; the cold-start rule is skipped there.  Same su-problems, same stop tolerance, another start: the
                                reference's default protocol (obstacle list re-sorted every tick, quirk Q5) needs 25 % fewer interior-point
                                iterations, converged loops are untouched.  Default since round 5 (it has been through the parity suite and the
                                soak; both keys are part of rda_get_su_history).                                          RDA_SU_HARD_WARM */
    /* ---- A/B switches (defaults in brackets) ---- */
    int32_t lmz_warm;        /* [1] try the remembered support first, then the supports one row away from it, before a row is
                                enumerated; every answer is accepted on its optimality certificate alone.  The supports are a cache of
                                the handle (not part of rda_get_state): they follow the source obstacle of a slot through the
                                re-binding of a re-sorted scene (rda_upload_scene*) and the horizon through the tick.  Circle obstacles
                                (norm2 cone) have one since round 5 in the single-ego launch form; in the dense forms and in a
                                fleet their rows are enumerated as before                                                RDA_LMZ_WARM */
    int32_t lmz_rows;        /* [1] four sub-problems per wave when E+R+1 <= 16                                        RDA_LMZ_ROWS */
    int32_t lmz_dense_from;  /* [256] grid size (CUs at one wave per SIMD) above which the split form of the LamMuZ launch is used; x 7/4 for moving scenes  RDA_LMZ_DENSE_FROM */
    int32_t lmz_split;       /* [1] dense grids: common-path kernel + work-list kernel + finalize                      RDA_LMZ_SPLIT */
    int32_t lmz_ip_rows;     /* [1] interior-point mode: the row-parallel kernel (16 lanes per sub-problem) when the shape allows
                                (0: one sub-problem per thread)                                                         RDA_LMZ_IP_ROWS */
    int32_t lmz_ip_warm;     /* [1] interior-point mode, row-parallel kernel: every sub-problem starts from the central-path point its last
                                solve ended on (same end point, 2 - 4 instead of ~9 iterations; solver history: rda_get_lmz_history) RDA_LMZ_IP_WARM */
    int32_t su_pre;          /* [1] the su set-up reads the block sums / near masks the LamMuZ launch wrote (0: evaluates every term) RDA_SU_PRE */
    int32_t su_light;        /* [1] convergence pass without the factorisation when the last step predicts convergence  RDA_SU_LIGHT */
    int32_t su_warm_first;   /* [1] the first su-problem of a step starts from the previous step's multipliers          RDA_SU_WARM_FIRST */
    int32_t su_warm_cap;     /* [30] iterations granted to a warm attempt                                               RDA_SU_WARM (3rd) */
    int32_t su_easy_max;     /* [2] easy mode while the last su-solve needed <= this many iterations (0 = never)        RDA_SU_EASY (6th) */
    int32_t su_easy_nopred;  /* [1] first iteration of an easy attempt without the predictor                            RDA_SU_EASY_NOPRED */
    int32_t su_cold_from;    /* [7] cold start after a solve with more iterations than this (0 = never) ...              RDA_SU_COLD_FROM */
    int32_t su_cold_probe;   /* [8] ... every this-many-th such solve tries the warm start                              RDA_SU_COLD_FROM (2nd) */
    int32_t zero_copy;       /* [1] result slot written into pinned host memory by the kernel, host polls a sequence word  RDA_ZERO_COPY */
    int32_t early_finish;    /* [1] the launch that detects the early stop hands the result over (0: k_finish does)      RDA_EARLY_FINISH */
    int32_t fuse_track;      /* [1] k_su_tracked (tracking beside su-problem 0; 0: k_track then k_su)                   RDA_FUSE_TRACK */
    int32_t su_prof;         /* [0] phase cycle counters of the su-solves (rda_debug_su_prof): PROFILING builds of the library only (-DSU_PROF /
                                -DSU_FINE, tools/su_phase_profile.py) - the product build refuses 1 with RDA_ERR_UNSUPPORTED   RDA_SU_PROF */
    int32_t su_split;        /* [1] su Newton system of the horizons 10, 20, 25, 30 cut in two halves that two waves factorise and sweep at
                                the same time, joined by a 5 x 5 interface system (same linear system, same answers up to rounding; 0: one
                                recursion over the whole horizon)                                                          RDA_SU_SPLIT */
    int32_t duals_follow;    /* [0] NOT reference semantics (opt-in).  The reference keeps lam, mu, z, xi, zeta per obstacle SLOT while its
                                default caller re-sorts the obstacle list on every tick (mpc.py:205-206), so after a re-sort most slots
                                continue from another obstacle's duals and the ADMM does not reach iter_threshold within iter_num (SURVEY
                                quirk Q5).  1: when the device pipeline re-binds the slots (rda_upload_scene*, rda_scene_resort) the dual
                                state moves WITH its obstacle - slot s takes the duals of the slot that held the same entry of the caller's
                                raw scene at the previous staging, an obstacle that was in no slot starts from the initial duals (zeros).
                                Obstacle i of the raw scene must denote the same obstacle from call to call.  The first su-problem of a
                                tick reads the terms of the previous tick's slots (as always).  Needs slots staged by the device pipeline
                                (rda_upload_obstacles / rda_step: RDA_ERR_UNSUPPORTED) and an unsharded handle.  rda_get_state /
                                rda_set_state speak the CURRENT slot order (rda_debug_slot_src tells which obstacle a slot holds); the
                                kept central points of the interior-point LamMuZ mode are dropped for a re-bound slot.  RDA_DUALS_FOLLOW */
    int32_t su_accept;       /* [1] safety net of the su interior point: the best iterate that is primal feasible to su_tol[1], dual feasible to
                                10 x su_tol[0] and complementary to 1000 x su_tol[2] (the class the reference's solver stops at) is remembered; a
                                solve whose every attempt then loses its end game in rounding (dual residual growing while the complementarity
                                falls below 1e-15, factorisation breaking down: one oracle solve in 32 000 soak steps) returns it instead of
                                'no update' (rda_solver.py:696-700).  Same rule as the oracle's orc_set_su_accept.  0: off; 2: test switch - every solve
                                hands back its remembered iterate (the path is otherwise never taken)                     RDA_SU_ACCEPT */
    int32_t su_first_attempt; /* [0] test switch: 1 = every su-solve starts with its LAST-RESORT attempt (plain long-step path following: no
                                predictor, fixed centring, lam w >= 1e-2 mu; normally reached only when the warm and the cold attempt have both
                                failed - csrc/su_device.h SU_SAFE_*).  Same solution, more iterations.                    (no env switch) */
    double  su_warm[2];      /* [1e-3, 1e-3] slack floor / barrier parameter of a warm start ("0,0" = always cold)      RDA_SU_WARM */
    double  su_warm_endgame[2]; /* [0.9999, 1e-5] floors of the fraction to the boundary / centering parameter, warm attempts  RDA_SU_WARM_ENDGAME */
    double  su_warm_clip;    /* [0.01]                                                                                  RDA_SU_WARM_CLIP */
    double  su_easy[5];      /* [1e-12, 1e-12, 1e-12, 0.999999, 1e-7] wfl, mu0, clip, tau, sigma of the easy start: the first three lie
                                BELOW the stop tolerances, i.e. the easy start is the previous solution itself and the stop test may
                                accept it without a Newton step when the new problem's optimality conditions hold there  RDA_SU_EASY */
    int32_t su_land;         /* [1] LANDING of the su interior point (round 6; oracle mirror orc_set_su_land): 1 = the interior point runs to
                                su_land_tol only - close enough for the active set to be read off (lam > w) - and the vertex it approaches is
                                computed exactly: active rows as equalities, the others dropped, the equality-constrained quadratic model solved
                                with the same factorisation and sweeps (two steps of the method of multipliers = the two passes of an iteration),
                                verified on the true objective, rows moved in / out by their signs for at most 4 rounds; refused -> the iterate
                                is restored and the iteration goes on; once every landing of an attempt has been refused it stops at 1e-3 x su_tol (SU_LAND_FALLBACK).  The answer no longer depends on WHERE on
                                the central path the iteration stopped (the reason for the stated tolerance of rounds 3-5, 5e-4: a row that is only just
                                active keeps the slack mu / lam*): kernel and cold oracle agree to 1e-9 with it (tests/test_gpu_land.py).
                                Costs one factorisation + a verification pass per solve, saves the last interior-point iteration.  RDA_SU_LAND */
    double  su_land_tol[3];  /* [1e-3, 1e-4, 1e-5] first stop of the interior point when it is landed; a refused landing is tried once more at 1e-2 x
                                these values, then the iteration runs to su_tol                                                  RDA_SU_LAND_TOL */
    double  su_land_rho;     /* [1e4] penalty of the landing's active rows, relative to the largest entry of the stage Hessians  RDA_SU_LAND_RHO */
    int32_t su_land_first;   /* [2] landing FIRST, for the warm-started su-problems (ADMM iterations >= 1: from the previous solution of the step and its multipliers; the first
                                one of a tick: from the previous tick's solution, shifted by one stage).
                                1: the first pass of the warm attempt is a light one (measures only - the start usually meets the landing's stop as it stands and the
                                factorisation of that pass was thrown away by the landing round anyway); results are bit-identical to 0.  2: ... and when the start
                                does not meet the stop the landing is tried all the same, from the start, active set = the rows whose kept multiplier exceeds the
                                slack, at most two rounds (a warm-started active-set method; accepted only on the verified optimality conditions of the true
                                problem, so the answer is the same vertex: differences at rounding level); refused: the interior point takes over.  RDA_SU_LAND_FIRST */
    int32_t su_land_blind_from; /* [1] with su_land_first = 2: after this many su-solves in a row that needed no interior-point iteration and one landing round, the next warm
                                su-problem of the same step starts WITH its landing round - no measuring pass in front of it (one round; refused: the interior
                                point takes over).  0 = never.                                                                  RDA_SU_LAND_BLIND_FROM */
} rda_opts;
void rda_opts_init(rda_opts *o);

int  rda_create(const rda_cfg *cfg, const double *G /*R*2*/, const double *h /*R*/, rda_handle **out);   /* = rda_create_opts(cfg, NULL, ...) */
int  rda_create_opts(const rda_cfg *cfg, const rda_opts *opts /* NULL: rda_opts_init */, const double *G, const double *h, rda_handle **out);
void rda_destroy(rda_handle *h);
int  rda_set_adjust(rda_handle *h, double slack_gain, double max_sd, double min_sd, double ro1, double ro2);
/* reset() of the reference (rda_solver.py:1060-1068): clears the lam'A / lam'b products, NOT the duals (quirk Q6); also clears the
 * handle's solver history (rda_get_su_history) */
int  rda_reset(rda_handle *h);
const char *rda_strerror(int code);
int  rda_device_count(void);
int  rda_set_device(int dev);     /* device used by handles created afterwards (one process per GPU).  HIP's current device belongs to the calling HOST THREAD
                                   * (default 0): a thread that drives handles or fleets created on device `dev` calls this once first.  The library keeps
                                   * no state outside its handles and fleets: distinct handles / fleets may be driven by distinct threads concurrently
                                   * (one thread at a time per handle or fleet) */

/* One MPC step, host buffers: nom_s 3x(T+1), nom_u 2xT, ref_s 3x(T+1) row-major;
 * obstacles A [n_obs][per_t? T+1 : 1][E][2], b [n_obs][per_t? T+1 : 1][E], cone [n_obs] (0 Rpositive, 1 norm2);
 * n_obs < N pads by duplicating the last obstacle, n_obs > N uses the first N, n_obs == 0 skips the
 * dual side (rda_solver.py:483-526,625).  out_u 2xT, out_s 3x(T+1). */
int  rda_step(rda_handle *h, const double *nom_s, const double *nom_u, const double *ref_s,
              double ref_speed, int n_obs, const double *A, const double *b, const int32_t *cone,
              int per_t, double *out_u, double *out_s, rda_info *info);

/* Caller-side obstacle pipeline on the device.  Replaces, per MPC tick: MPC.convert_rda_obstacle,
 * rda_obs_distance and the distance sort (mpc.py:189-218), convert_inequal_circle / convert_inequal_polygon with
 * the constant-velocity prediction `+ velocity * (t * dt)` (mpc.py:440-472), gen_inequal_global / is_convex_and_ordered
 * (mpc.py:476-549), and RDA_solver.assign_obstacle_parameter (rda_solver.py:483-526: first max_obs_num by distance,
 * padding with the last one, zeroed spare rows, per-t replication).  The solver's obstacle slots come out bit-identical
 * to what the Python caller stages.
 *   kind [n]      0 = polygon (cone 'Rpositive'), 1 = circle (cone 'norm2')
 *   nvert [n]     polygon vertex count, <= E (ignored for circles)
 *   geom [n][E][2] polygon: vertices in the caller's order (CW input is reversed like the reference does);
 *                  circle: geom[i][0] = centre, geom[i][1][0] = radius
 *   vel [n][2]    obstacle velocity; |vel| > 0.01 makes the obstacle time-varying over the horizon
 *   robot_xy [2]  robot position for the ordering (order != 0: nearest first, stable; order == 0: caller's order)
 *   n_nonconvex   (optional) number of staged polygons that fail the reference's convexity test (it prints a warning)
 * n == 0 leaves the slots untouched and skips the dual side, like rda_step. */
int  rda_upload_scene(rda_handle *h, int n, const int32_t *kind, const int32_t *nvert, const double *geom,
                      const double *vel, const double *robot_xy, int order, int32_t *n_nonconvex);
/* rda_step with the obstacle conversion done on the device */
int  rda_step_scene(rda_handle *h, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed,
                    int n, const int32_t *kind, const int32_t *nvert, const double *geom, const double *vel,
                    const double *robot_xy, int order, double *out_u, double *out_s, rda_info *info);
/* test hook: staged slots A [N][nt][E][2], b [N][nt][E], cone [N]; *nt = 1 or T+1 (buffers sized for T+1) */
int  rda_get_obstacles(rda_handle *h, double *A, double *b, int32_t *cone, int32_t *nt);

/* Device-resident pipeline (what bench.py times): obstacles and a trace of K step inputs are
 * uploaded once; rda_enqueue_step queues the whole ADMM loop of step k on the handle's stream
 * without any host synchronisation; results land in device slot k and are fetched afterwards. */
int  rda_upload_obstacles(rda_handle *h, int n_obs, const double *A, const double *b, const int32_t *cone, int per_t);
int  rda_upload_trace(rda_handle *h, int K, const double *nom_s /*K*3*(T+1)*/, const double *nom_u /*K*2*T*/,
                      const double *ref_s /*K*3*(T+1)*/, const double *ref_speed /*K*/);
int  rda_enqueue_step(rda_handle *h, int k);
int  rda_enqueue_range(rda_handle *h, int k0, int k1);      /* steps k0 .. k1-1, one host call */
int  rda_sync(rda_handle *h);
int  rda_fetch_result(rda_handle *h, int k, double *out_u, double *out_s, rda_info *info);
/* elapsed GPU time (ms, hipEvent) of the launches named `which` (0 = LamMuZ, 1 = su, 2 = the shard all-gather) over the
 * steps enqueued since the last rda_timing_reset, and the number of launches */
int  rda_timing_reset(rda_handle *h, int enable);
int  rda_timing_read(rda_handle *h, int which, double *total_ms, int *launches);
/* the same per launch, in launch order: ms_out[i] for i < min(cap, *launches).  The early stop of rda_solver.py:594 is a
 * device flag, so launches queued behind it return at once; a caller that knows the executed ADMM iterations (rda_info.iters)
 * separates the two populations with this (bench.py: roofline per EXECUTED launch) */
int  rda_timing_launches(rda_handle *h, int which, double *ms_out, int cap, int *launches);
/* the form of the LamMuZ launch the library uses for this handle's shape and staged obstacles (kernel names joined by '+') */
const char *rda_lammuz_kernel(rda_handle *h);

/* ---- caller-side nominal roll-out + reference sampling on the device ------------------------------
 * What MPC.pre_process (mpc.py:251-291) with closest_point / inter_point / range_cir_seg / wraptopi and the three
 * motion_predict_model_* (mpc.py:293-433) computes per tick, as a device kernel in front of the ADMM loop: the caller
 * hands over the robot state, the signed reference speed and its path index instead of the 3x(T+1) nominal states
 * and reference.  rda_upload_path stores the polyline (L waypoints x, y, heading; row-major [L][3]).  The last
 * waypoint's heading is rewritten by a tick that reaches the end of the path exactly like the reference rewrites it
 * in place (quirk Q12); `end_heading` returns it so a host mirror of the path can follow.
 * nom_u: the nominal controls [2][T] (MPC.cur_vel_array), or NULL = the controls of the previous solve, which are
 * still resident (what cur_vel_array holds unless the caller replaced it).  min_index = the new MPC.cur_index.
 * nom_s_out / ref_out (may be NULL) return the 3x(T+1) nominal states / reference the solver was given. */
/* Number of polygons of the scene the LAST completed step ran on that fail the reference's convexity test (mpc.py:476-549 prints
 * "Warning: The polygon constructed by vertex is not convex" for each); 0 when the obstacles were staged as (A, b) slots.  Travels
 * with the step's result, costs no synchronisation (rda_upload_scene's n_nonconvex is the synchronous form). */
int  rda_last_nonconvex(rda_handle *h);
int  rda_upload_path(rda_handle *h, int L, const double *path /*L*3*/);
int  rda_step_tracked(rda_handle *h, const double *state /*3*/, double ref_speed, int cur_index, double threshold, int ind_range,
                      const double *nom_u, double *out_u, double *out_s, rda_info *info,
                      double *nom_s_out, double *ref_out, int32_t *min_index, double *end_heading);
/* The same tick in two halves, so that the caller's per-tick obstacle work overlaps the first su-problem.  The first
 * su-problem of a step reads the nominal trajectory and the lam'A / lam'b products of the PREVIOUS step (the reference never
 * refreshes them before rda_solver.py:591 - SURVEY quirk Q4) and nothing of the obstacles staged for this tick, so
 *   rda_tracked_begin        queues k_track, the step reset and su-problem 0 and returns at once;
 *   rda_upload_scene_async   (optional) stages this tick's raw scene like rda_upload_scene, without waiting; also usable on
 *                            its own when the caller's next call on the handle synchronises (rda_sync, a step, a fleet
 *                            step) - a second upload before that waits for the first one;
 *                            rda_upload_obstacles / rda_upload_scene may be used instead (they synchronise);
 *   rda_tracked_finish       queues the rest of the ADMM loop, waits, returns what rda_step_tracked returns.
 * Results are bit-identical to rda_upload_scene + rda_step_tracked (same kernels, same order of dependent work).
 * Between begin and finish only the upload calls may be used on the handle. */
int  rda_tracked_begin(rda_handle *h, const double *state /*3*/, double ref_speed, int cur_index, double threshold, int ind_range,
                       const double *nom_u);
int  rda_upload_scene_async(rda_handle *h, int n, const int32_t *kind, const int32_t *nvert, const double *geom,
                            const double *vel, const double *robot_xy, int order);
int  rda_tracked_finish(rda_handle *h, double *out_u, double *out_s, rda_info *info,
                        double *nom_s_out, double *ref_out, int32_t *min_index, double *end_heading);
/* The reference re-sorts the caller's obstacle list by distance to the robot on EVERY tick (MPC.convert_rda_obstacle with
 * obstacle_order=True, its default: mpc.py:205-206) and stages the first max_obs_num.  For a raw scene that is already resident
 * (rda_upload_scene / rda_upload_scene_async) rda_scene_resort re-ranks it about robot_xy and rebuilds the obstacle slots with the same
 * kernels (k_keys / k_rank / k_build / k_prepare): the position travels in the kernel arguments, nothing is copied.  Asynchronous like
 * rda_upload_scene_async and usable between rda_tracked_begin and rda_tracked_finish.  Slots come out bit-identical to uploading the
 * same scene again with order = 1.  A scene whose obstacles MOVE between ticks has to be uploaded again instead (the prediction over
 * the horizon starts from the uploaded geometry).  RDA_ERR_ARG without a resident raw scene. */
int  rda_scene_resort(rda_handle *h, const double *robot_xy /*2*/);

/* ---- Fleet: B independent egos advanced together (BASELINE config C5, "batched multi-ego") -------------------
 * The reference plans one robot per RDA_solver object (rda_solver.py:54-109) and a multi-robot user loops over
 * objects.  Here the members stay ordinary handles (own state, obstacles, trace, accessors); the fleet launches
 * their ADMM iterations as ONE grid with an ego dimension, which is what fills the 256 CUs.  Members must agree on
 * T, N, E, R and iter_num (weights, bounds, kinematics, robot polygons and obstacles may differ), must live on the
 * current device and must not be obstacle shards.  Results are identical to stepping every member with rda_step /
 * rda_enqueue_step.  Between rda_fleet_enqueue_range and rda_fleet_sync the members must not be used. */
typedef struct rda_fleet rda_fleet;
int  rda_fleet_create(rda_handle *const *egos, int B, rda_fleet **out);   /* does not take ownership of the members */
void rda_fleet_destroy(rda_fleet *f);
int  rda_fleet_size(rda_fleet *f);
/* one synchronous MPC step of every member with the obstacles each member has staged (rda_upload_obstacles /
 * rda_upload_scene): per-ego arrays of rda_step, concatenated ego-major */
int  rda_fleet_step(rda_fleet *f, const double *nom_s /*B*3*(T+1)*/, const double *nom_u /*B*2*T*/,
                    const double *ref_s /*B*3*(T+1)*/, const double *ref_speed /*B*/,
                    double *out_u /*B*2*T*/, double *out_s /*B*3*(T+1)*/, rda_info *info /*B, may be NULL*/);
/* rda_upload_scene_async for every member in one call: member i owns counts[i] consecutive entries of kind / nvert / geom /
 * vel, robot_xy [B][2], order [B].  No waiting: the next fleet step orders itself behind the staging and synchronises. */
int  rda_fleet_upload_scenes(rda_fleet *f, const int32_t *counts /*B*/, const int32_t *kind, const int32_t *nvert, const double *geom,
                             const double *vel, const double *robot_xy /*B*2*/, const int32_t *order /*B*/);
/* rda_step_tracked for every member (paths uploaded with rda_upload_path on the members); per-ego arrays ego-major,
 * nom_u NULL = every member's resident controls */
int  rda_fleet_step_tracked(rda_fleet *f, const double *states /*B*3*/, const double *ref_speed /*B*/, const int32_t *cur_index /*B*/,
                            double threshold, int ind_range, const double *nom_u /*B*2*T or NULL*/,
                            double *out_u, double *out_s, rda_info *info, double *ref_out /*B*3*(T+1) or NULL*/,
                            int32_t *min_index /*B*/, double *end_heading /*B*/);
/* rda_scene_resort for every member in ONE launch set: the members' resident raw scenes (rda_upload_scene*, obstacles that do not move between ticks) re-ranked
 * about states[i * stride + 0..1] (stride >= 2: the caller's [B][3] state array serves) and their slots rebuilt on the fleet's stream; the next fleet step runs
 * behind it.  Replaces one MPC.convert_rda_obstacle + distance sort per robot and tick (mpc.py:189-218, obstacle_order=True).  Slots bit-identical to
 * rda_scene_resort on each member.  Until the next fleet step has returned the members must not be used on their own. */
int  rda_fleet_scene_resort(rda_fleet *f, const double *states, int stride);
/* steps k0 .. k1-1 of every member's uploaded trace, asynchronous; read with rda_fetch_result after rda_fleet_sync */
int  rda_fleet_enqueue_range(rda_fleet *f, int k0, int k1);
int  rda_fleet_sync(rda_fleet *f);

/* State in the reference's shapes: lam [N][T+1][E], mu [N][T+1][R], z [N][T], xi [N][T+1][2],
 * zeta [N][T], dis [T], a_lam [N][T+1][2] (para_obsA_lam), b_lam [N][T+1] (para_obsb_lam).
 * NULL pointers are skipped. */
int  rda_get_state(rda_handle *h, double *lam, double *mu, double *z, double *xi, double *zeta,
                   double *dis, double *a_lam, double *b_lam);
int  rda_set_state(rda_handle *h, const double *lam, const double *mu, const double *z,
                   const double *xi, const double *zeta, const double *dis,
                   const double *a_lam, const double *b_lam);
/* Solver history of a handle: not reference-visible state, but it picks the START of the next su interior-point solve (easy /
 * moderate / cold), so two handles return bit-identical controls only if it agrees too.  hist[0] = interior-point
 * iterations of the last su-solve (99 = none), hist[1] = consecutive solves in the hard regime, hist[2] = the previous step ended above
 * iter_threshold, hist[3] = the last su-solve started far from its solution (the two keys of su_hard_warm); lam_keep [10*T] = the inequality
 * multipliers of the last converged su-solve.  rda_create and rda_reset set (99, 0, zeros).  NULL pointers are skipped. */
#define RDA_SU_HISTORY_INTS 8   /* entries of `hist` in THIS header (round 4: 2, round 5: 4 - the array grows with the start rules, an ABI break for callers of the count-less
                                   forms: use the _n forms below).  hist[4] (round 6) = the credit of the speculative landings, rda_opts::su_land_first = 2; hist[5] > 0: one of the last four landings took three or more
                                   rounds (the next solve's interior point runs to 1e-2 x su_land_tol before it is landed); hist[6] = su-solves in a row without an interior-point iteration and with
                                   one landing round, hist[7] = the gate of the blind landings (rda_opts::su_land_blind_from) */
int  rda_get_su_history(rda_handle *h, int32_t *hist /*RDA_SU_HISTORY_INTS*/, double *lam_keep /*10*T*/);
int  rda_set_su_history(rda_handle *h, const int32_t *hist /*RDA_SU_HISTORY_INTS*/, const double *lam_keep /*10*T*/);
/* ... with the caller's own count: get writes n_hist entries (those this library does not have read 0), set reads
 * min(n_hist, RDA_SU_HISTORY_INTS) and leaves the others as they are - a caller compiled against an older or newer header stays correct */
int  rda_get_su_history_n(rda_handle *h, int32_t *hist, int n_hist, double *lam_keep /*10*T*/);
int  rda_set_su_history_n(rda_handle *h, const int32_t *hist, int n_hist, const double *lam_keep /*10*T*/);
/* Interior-point LamMuZ mode: the central-path points the sub-problems last ended on ([T][N][5][16] doubles: x | s, z of the diagonal
 * rows | s, z of the general rows, csrc/lammuz_ip_device.h) and their validity flags [T][N] - where each sub-problem's next solve starts.
 * Solver history like the su history above: it moves a result only within the centring tolerance (1e-7 mu relative), rda_reset clears
 * it.  rda_lmz_history_doubles = 80 N T, or 0 when the handle keeps none (enumeration mode, per-thread kernel, lmz_ip_warm = 0). */
int  rda_lmz_history_doubles(rda_handle *h);
int  rda_get_lmz_history(rda_handle *h, double *points, int32_t *valid);
int  rda_set_lmz_history(rda_handle *h, const double *points, const int32_t *valid);
/* debug: accumulated clock64 phase counters of the su-solves of this handle since the last call (rda_opts::su_prof), 16 values */
int  rda_debug_su_prof(rda_handle *h, long long *out16);
/* -DSU_TRACE builds only (RDA_ERR_UNSUPPORTED otherwise): per-wave (event id, clock64) pairs of the LAST su launch, out[4][cap][2] (tools/su_trace.py) */
int  rda_debug_su_trace(rda_handle *h, long long *out, int cap, int *n_out);
int  rda_debug_su_land(rda_handle *h, int32_t *out4);      /* su_land: landings accepted, refused, rounds, passes spent on landings since the last call */
int  rda_debug_su_land_n(rda_handle *h, int32_t *out, int n);   /* ... the first n <= 20 counters: [4] speculative landings (su_land_first = 2) tried, [5] accepted, [6 + k] / [12 + k] tried / accepted by the decade k = 0 .. 5 of the start's relative dual residual (< 1e-4, < 1e-3, < 1e-2, < 1e-1, < 1, >= 1) */
int  rda_debug_flush_supports(rda_handle *h);             /* forget every remembered LamMuZ support (a cache: results must not depend on it) */
int  rda_debug_slot_src(rda_handle *h, int32_t *src /*N*/, int32_t *used); /* slot -> entry of the caller's raw scene (device pipeline; used = 0: host-staged slots) */
int  rda_debug_worklist(rda_handle *h, int *rows);        /* rows on the LamMuZ work list of the last executed iteration (split launch form) */

/* Obstacle sharding across the GPUs of one node (one process per GPU).  Rank r owns the obstacle slots
 * [r*ceil(N/world), (r+1)*ceil(N/world)): it solves their LamMuZ problems and keeps their duals.  What the su-problem needs of
 * them forms one contiguous chunk per rank - 3 doubles per (slot, stage): a = A'lam (2) and the hinge offset lam'b + mu'h + z - zeta;
 * plus, per (stage, 8-slot block), the 5 reduced sums (sum |a|^2, sum g.a, sum g x a, the two residual partials) and one 8-byte word
 * with the near mask the su set-up reads instead of passing over the terms - replicated to every rank by ONE all-gather per ADMM
 * iteration (replaces the pool.map scatter/gather of rda_solver.py:706-725); every rank then solves the identical su-problem.  The
 * other per-row records (lam'b, mu'h + z - zeta, G'mu + xi, residual records) stay on the owning rank; for that reason
 *   - rda_opts::su_pre = 0 (the su set-up evaluating raw terms) is overridden to 1 when world > 1, and
 *   - rda_reset / rda_set_state return RDA_ERR_UNSUPPORTED on a handle with world > 1 once it has stepped (the remote slots' terms
 *     cannot be rebuilt locally; reset / restore the state BEFORE the first step, or re-create the handles).
 * rda_shard_config must precede the first step.
 * N need not be divisible by world: the shards have ceil(N / world) slots, the slots past the last obstacle carry terms the
 * su-problem ignores - this relies on the hinge of the accelerated cost: with accelerated = 0 and N % world != 0
 * rda_shard_config returns RDA_ERR_UNSUPPORTED. */
int  rda_shard_config(rda_handle *h, int rank, int world);
int  rda_shard_chunk_doubles(rda_handle *h);                       /* 3*T*Nloc + 6*T*ceil(Nloc/8), Nloc = ceil(N/world) */
int  rda_shard_get_chunk(rda_handle *h, double *host_chunk);       /* this rank's chunk  */
int  rda_shard_set_chunks(rda_handle *h, const double *host_all);  /* all `world` chunks, rank-major */
/* RCCL exchange over xGMI: rank 0 calls rda_shard_unique_id and ships the 128 bytes to the other ranks (any
 * side channel); every rank then calls rda_shard_comm_init.  From then on rda_step / rda_enqueue_step issue
 * one in-place ncclAllGather per ADMM iteration on the handle's stream. */
int  rda_shard_unique_id(rda_handle *h, void *out128);
int  rda_shard_comm_init(rda_handle *h, const void *uid128);
int  rda_shard_comm_count(rda_handle *h);                          /* ncclCommCount of the handle's communicator; 0 without one */
/* Host-driven ADMM iteration for callers that do the exchange themselves (tests, gloo, MPI):
 *   rda_admm_begin; for it: rda_admm_su(it,&stopped); if stopped break; rda_admm_lammuz; <exchange chunks>; rda_admm_finish */
int  rda_admm_begin(rda_handle *h, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed);
int  rda_admm_su(rda_handle *h, int it, int *stopped);
int  rda_admm_lammuz(rda_handle *h);
int  rda_admm_finish(rda_handle *h, double *out_u, double *out_s, rda_info *info);

/* Pure-function hooks -----------------------------------------------------------------------*/
/* B independent (obstacle, stage) sub-problems in one launch, one per wavefront.
 * A [B][E][2], b [B][E], cone [B], p [B][2] (nominal position, column t+1), phi [B] (nominal heading,
 * column t), xi [B][2], zeta [B], dbar [B]; G [R][2], h [R].  Outputs lam [B][E], mu [B][R], z [B],
 * cmh [B][4] = (cost, m, H0, H1). */
int  rda_lammuz_batch(int B, int E, int R, const double *A, const double *b, const int32_t *cone,
                      const double *p, const double *phi, const double *G, const double *h,
                      const double *xi, const double *zeta, const double *dbar, double ro2, double delta,
                      int accelerated, double *lam, double *mu, double *z, double *cmh);

/* su-problem with condensed obstacle terms a [N][T][2], cc [N][T], g [N][T][2];
 * d0 [T] initial guess; outputs s 3x(T+1), u 2xT, d [T].  Returns 0 ok, 1 not converged. */
int  rda_su_solve(const rda_cfg *cfg, const double *nom_s, const double *nom_u, const double *ref_s,
                  double ref_speed, const double *a, const double *cc, const double *g,
                  const double *d0, double *s, double *u, double *d, int32_t *ipm_iters);
/* ... with the caller's rda_opts (NULL = rda_opts_init): su_tol, su_split, su_accept; su_prof = 1 prints the phase cycles, 2 also one line per
 * interior-point iteration, on stderr */
int  rda_su_solve_opts(const rda_cfg *cfg, const rda_opts *opts, const double *nom_s, const double *nom_u, const double *ref_s,
                       double ref_speed, const double *a, const double *cc, const double *g,
                       const double *d0, double *s, double *u, double *d, int32_t *ipm_iters);

#ifdef __cplusplus
}
#endif
#endif
