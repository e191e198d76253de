/*
 * ORACLE - test infrastructure only.  CPU (plain C, fp64) restatement of the RDA ADMM
 * inner solver of hanruihua/RDA-planner (RDA_planner/rda_solver.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (rda_planner_amd/) never does.
 *
 * PARITY (round 2): the reference solves its two convex sub-problems with CVXPY 1.5.2 -> ECOS (rda_solver.py:693,768,800),
 * neither installable in the build image, and ships no tests / golden vectors.  The restatement is pinned on the
 * UNMODIFIED reference modules executing in the build container on a cvxpy / pathos stand-in (oracle/refshim, ref_harness.py,
 * tests/test_reference_pinned.py, fixtures tests/golden/ref_*.npz): the ADMM plumbing per iteration (<= 1e-11) and both argmins
 * in everything that is unique (su: s, u, d; LamMuZ: value, min(Im, 0), Hm).  What stays unpinned BY CONSTRUCTION is the non-unique
 * part of the LamMuZ answer (lam, mu, z individually in the slack regime): no two interior-point solvers agree on it.
 * Also pinned by KKT certificates, scipy cross-checks and geometric known-answer tests (tests/test_oracle_*.py).
 */
#ifndef RDA_ORACLE_H
#define RDA_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int T;            /* receding                      rda_solver.py:34  */
    int N;            /* max_obs_num                   rda_solver.py:38  */
    int E;            /* max_edge_num                  rda_solver.py:39  */
    int R;            /* G.shape[0]                    rda_solver.py:99  */
    int dynamics;     /* 0 acker, 1 diff, 2 omni       rda_solver.py:446-451 */
    int accelerated;  /*                               rda_solver.py:47  */
    int iter_num;     /*                               rda_solver.py:42  */
    int robot_norm2;  /* car_tuple.cone_type=='norm2'  (interior-point LamMuZ mode only, orc_set_lmz_mode(1)) */
    double dt, L;
    double max_speed[2];   /* rda_solver.py:37 */
    double acce_bound[2];  /* max_acce*dt, rda_solver.py:44 */
    double iter_threshold;
    double ws, wu;                                  /* rda_solver.py:218-219 */
    double slack_gain, max_sd, min_sd, ro1, ro2;    /* rda_solver.py:196-201 */
    double delta;     /* tie-break (T1): clearance reward, default 1e-6 */
    double eps_u;     /* tie-break for an undetermined steering column, default 1e-8 */
} orc_cfg;

typedef struct {
    double resi_dual, resi_pri;
    int iters;          /* ADMM iterations executed */
    int su_status;      /* bit i set: su solve of iteration i did not converge (kept nominal) */
    int su_ipm_iters;   /* total IPM iterations */
    int lmz_fail;       /* sub-problems that kept their previous duals (non-finite data / not OPTIMAL), residual inf */
} orc_info;

typedef struct orc_handle orc_handle;

int  orc_create(const orc_cfg *cfg, const double *G /*R*2*/, const double *h /*R*/, orc_handle **out);
void orc_destroy(orc_handle *h);
int  orc_set_adjust(orc_handle *h, double slack_gain, double max_sd, double min_sd, double ro1, double ro2);
int  orc_reset(orc_handle *h);      /* rda_solver.py:1060-1068 */
void orc_set_threads(int n);
void orc_set_su_dump(const char *path);   /* debug: record every su-problem orc_admm_su solves (tools/su_replay.py) */
void orc_set_su_trace(int on);            /* debug: one stderr line per interior-point iteration */
void orc_set_lmz_mode(int mode);   /* LamMuZ sub-problems of orc_step: 0 support enumeration (tie-breaks T1-T3), 1 interior point (lmz_ipm.c) */
int  orc_lmz_failures(orc_handle *h);   /* sub-problems of the last step whose solve was not OPTIMAL (previous duals kept, residual inf) */
void orc_set_lmz_ipm_tol(double tol);
void orc_set_lmz_ipm_mu(double mu);
int  orc_lammuz_ipm_one(int E, int R, const double *A, const double *b, int cone_norm2, int robot_norm2,
                        const double *p, double phi, const double *G, const double *h,
                        const double *xi, double zeta, double dbar, double ro2, int accelerated,
                        double *lam, double *mu, double *z, double *cost_m_H /*4*/, int *iters);
void orc_set_centre(int on);   /* tie-break T1: 1 (default) central separating normal in the slack regime, 0 max clearance */

/* One MPC step == RDA_solver.iterative_solve (rda_solver.py:573-610).
 * nom_s 3x(T+1) row-major, nom_u 2xT, ref_s 3x(T+1); obstacles: n_obs entries,
 * A [n_obs][per_t? T+1 : 1][E][2], b [n_obs][per_t? T+1 : 1][E], cone [n_obs] (0 Rpositive, 1 norm2).
 * out_u 2xT, out_s 3x(T+1). */
int  orc_step(orc_handle *h, const double *nom_s, const double *nom_u, const double *ref_s,
              double ref_speed, int n_obs, const double *A, const double *b, const int *cone,
              int per_t, double *out_u, double *out_s, orc_info *info);

/* Obstacle sharding / host-driven ADMM pieces - same contract as rda_shard_* / rda_admm_* in include/rda_hip.h */
int  orc_upload_obstacles(orc_handle *h, int n_obs, const double *A, const double *b, const int *cone, int per_t);
int  orc_shard_config(orc_handle *h, int rank, int world);
int  orc_shard_chunk_doubles(orc_handle *h);
int  orc_shard_get_chunk(orc_handle *h, double *chunk);
int  orc_shard_set_chunks(orc_handle *h, const double *all);
int  orc_admm_begin(orc_handle *h, const double *nom_s, const double *nom_u, const double *ref_s, double ref_speed);
int  orc_admm_su(orc_handle *h, int it, int *stopped);
int  orc_admm_lammuz(orc_handle *h);
int  orc_admm_finish(orc_handle *h, double *out_u, double *out_s, orc_info *info);

/* State access in the reference's shapes: lam [N][T+1][E], mu [N][T+1][R], z [N][T],
 * xi [N][T+1][2], zeta [N][T], dis [T], a_lam [N][T+1][2], b_lam [N][T+1]. */
int  orc_get_state(orc_handle *h, double *lam, double *mu, double *z, double *xi, double *zeta,
                   double *dis, double *a_lam, double *b_lam);
int  orc_set_state(orc_handle *h, const double *lam, const double *mu, const double *z,
                   const double *xi, const double *zeta, const double *dis,
                   const double *a_lam, const double *b_lam);

/* Pure-function hooks (mirror solve_parallel, rda_solver.py:743-793) ---------------------*/
/* One (obstacle, stage) sub-problem.  A[E][2] b[E], p[2] nominal position (column t+1),
 * phi nominal heading (column t), xi[2].  Outputs lam[E], mu[R], z; returns candidate index. */
int  orc_lammuz_one(int E, int R, const double *A, const double *b, int cone_norm2,
                    const double *p, double phi, const double *G, const double *h,
                    const double *xi, double zeta, double dbar, double ro2, double delta,
                    int accelerated, double *lam, double *mu, double *z, double *cost_m_H /*4*/);

/* su-problem (rda_solver.py:216-231,313-387).  Condensed obstacle terms per (n,t):
 * a[N][T][2], cc[N][T] (= lam'b + mu'h + z - zeta), g[N][T][2] (= G'mu + xi).
 * nom_s 3x(T+1), nom_u 2xT linearisation point; d0 [T] initial guess.
 * Outputs s 3x(T+1), u 2xT, d [T]. returns 0 ok, 1 not converged. */
/* interior-point start of the su-problems of ADMM iterations >= 1 inside orc_step / orc_admm_su (defaults 1e-3, 1e-3, 30; 0,0,0 = cold);
 * orc_su_solve itself always starts cold */
void orc_set_su_warm(double wfl, double mu0, int cap);
void orc_set_su_warm_endgame(double tau_floor, double sigma_floor);
void orc_set_su_warm_clip(double margin);
void orc_set_su_easy(double wfl, double mu0, double clip, double tau, double sig, int max_iters);   /* mirror of RDA_SU_EASY */   /* warm attempts: relative margin of the start inside the boxes (cold: 0.01) */   /* warm attempts only; cold solves keep 0.995 / 1e-3 */
/* mirrors of the kernel's other start rules (csrc/rda_hip.hip su_body): rda_opts::su_hard_warm (default 1, 1e-3; 0, 0 = off; keys: the previous step
 * ended above iter_threshold AND the last solve's first iterate had a relative dual residual above 1e-2) and rda_opts::su_cold_from / su_cold_probe
 * (default 7, 8) */
void orc_set_su_hard_warm(double wfl, double mu0);
void orc_set_su_cold_from(int from, int probe);
void orc_set_su_first_attempt(int first);   /* test switch (= rda_opts::su_first_attempt): 1 = every su-solve starts with the last-resort attempt */
void orc_get_su_ipm_hist(const orc_handle *h, int *out, int n);   /* debug: interior-point iterations of the su-solve of each ADMM iteration of the last step */
/* interior-point stop of the su-problem: |r_dual| <= rd (1+|g|), |r_prim| <= rp, mean complementarity <= mu (1+|g|) */
void orc_set_su_tol(double rd, double rp, double mu);
int  orc_su_solve(const orc_cfg *cfg, const double *nom_s, const double *nom_u, const double *ref_s,
                  double ref_speed, const double *a, const double *cc, const double *g,
                  const double *d0, double *s, double *u, double *d, int *ipm_iters);

#ifdef __cplusplus
}
#endif
#endif
