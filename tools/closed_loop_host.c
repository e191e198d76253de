/* A caller of the C-ABI written in C: the closed MPC loop that bench.py times (BASELINE.md 2.4 - per step: robot state in,
 * control out, ONE host synchronisation, the host applies the first control to the kinematic model like ir-sim's env.step).
 * This is the loop a C / C++ user (a ROS node, the reference's simulator loop) runs around include/rda_hip.h; bench.py loads it
 * so that the timed region contains C-ABI calls and the caller's arithmetic only - no interpreter objects between two steps.
 * It links against nothing: the entry points are handed over as function pointers (bench.py takes them from librda_hip.so).
 *
 *     gcc -O2 -fPIC -shared -o tools/libclosed_loop_host.so tools/closed_loop_host.c -lm      (done by __graft_entry__.build())
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/rda_hip.h"

typedef int (*step_tracked_fn)(rda_handle *, const double *, double, int, double, int, const double *, double *, double *, rda_info *,
                               double *, double *, int32_t *, double *);
typedef int (*tracked_begin_fn)(rda_handle *, const double *, double, int, double, int, const double *);
typedef int (*upload_scene_async_fn)(rda_handle *, int, const int32_t *, const int32_t *, const double *, const double *, const double *, int);
typedef int (*tracked_finish_fn)(rda_handle *, double *, double *, rda_info *, double *, double *, int32_t *, double *);
typedef int (*scene_resort_fn)(rda_handle *, const double *);

struct closed_loop_api {
    step_tracked_fn step_tracked;
    tracked_begin_fn tracked_begin;
    upload_scene_async_fn upload_scene_async;
    tracked_finish_fn tracked_finish;
    scene_resort_fn scene_resort;
};

struct closed_loop_scene {          /* raw scene handed over on every tick (n == 0: the scene is resident, rda_step_tracked is used - or, with
                                     * order != 0, the resident scene is re-sorted about the robot on every tick like MPC.control does,
                                     * mpc.py:205-206: rda_tracked_begin + rda_scene_resort + rda_tracked_finish) */
    int32_t n, maxv, order, moving;
    const int32_t *kind, *nvert;
    double *geom;                   /* [n][maxv][2], advanced in place when `moving` */
    const double *geom0, *vel;      /* [n][maxv][2], [n][2] */
};

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* Runs steps k0 .. k0 + n_steps - 1 of one closed loop.  state[3], *cur_index are read and advanced; u_log [n_steps][2] receives the
 * applied controls, t_log [n_steps] the wall time of each step (call + kinematics), iters_log [n_steps] the executed ADMM
 * iterations, begin_log [n_steps] (may be NULL) the time the first half of a two-call tick took on the host, info_log [n_steps][3] (may be
 * NULL) the step's final ADMM residuals and its interior-point iterations (rda_info: resi_dual, resi_pri, su_ipm_iters).
 * dynamics: 0 acker, 1 diff, 2 omni (rda_cfg.dynamics).  Returns 0, a negative rda error code, or 1 when the path ended. */
int closed_loop_run(const struct closed_loop_api *api, rda_handle *h, const struct closed_loop_scene *sc, int T, int dynamics, double wheelbase,
                    double dt, double ref_speed, double threshold, int ind_range, int path_len, int k0, int n_steps, const double *nom_u_first,
                    double *state, int32_t *cur_index, double *u_log, double *t_log, int32_t *iters_log, double *begin_log, double *info_log)
{
    double out_u[2 * RDA_TMAX], out_s[3 * (RDA_TMAX + 1)], eh = 0;
    rda_info inf;
    int32_t mi = *cur_index;
    if (T > RDA_TMAX) return -1;
    for (int k = k0; k < k0 + n_steps; ++k) {
        const double t0 = now_s();
        const double *nu = (k == 0) ? nom_u_first : 0;     /* afterwards the previous controls are resident (MPC.cur_vel_array) */
        int rc;
        if (sc && sc->n > 0) {
            if (sc->moving)                                 /* obstacles advance every tick like in the dynamic_obs example */
                for (int i = 0; i < sc->n; ++i)
                    for (int v = 0; v < sc->nvert[i] && v < sc->maxv; ++v) {
                        const size_t o = ((size_t)i * sc->maxv + v) * 2;
                        sc->geom[o] = sc->geom0[o] + sc->vel[2 * i] * (dt * k);
                        sc->geom[o + 1] = sc->geom0[o + 1] + sc->vel[2 * i + 1] * (dt * k);
                    }
            rc = api->tracked_begin(h, state, ref_speed, *cur_index, threshold, ind_range, nu);
            if (begin_log) begin_log[k - k0] = now_s() - t0;
            if (rc >= 0) rc = api->upload_scene_async(h, sc->n, sc->kind, sc->nvert, sc->geom, sc->vel, state, sc->order);
            if (rc >= 0) rc = api->tracked_finish(h, out_u, out_s, &inf, 0, 0, &mi, &eh);
        } else if (sc && sc->order && api->scene_resort) {
            rc = api->tracked_begin(h, state, ref_speed, *cur_index, threshold, ind_range, nu);
            if (begin_log) begin_log[k - k0] = now_s() - t0;
            if (rc >= 0) rc = api->scene_resort(h, state);
            if (rc >= 0) rc = api->tracked_finish(h, out_u, out_s, &inf, 0, 0, &mi, &eh);
        } else {
            rc = api->step_tracked(h, state, ref_speed, *cur_index, threshold, ind_range, nu, out_u, out_s, &inf, 0, 0, &mi, &eh);
        }
        if (rc < 0) return rc;
        *cur_index = mi;
        if (mi >= path_len - 1) return 1;
        const double v = out_u[0], w = out_u[T], phi = state[2];
        if (dynamics == 0) { state[0] += dt * (v * cos(phi)); state[1] += dt * (v * sin(phi)); state[2] += dt * (v * tan(w) / wheelbase); }
        else if (dynamics == 1) { state[0] += dt * (v * cos(phi)); state[1] += dt * (v * sin(phi)); state[2] += dt * w; }
        else { state[0] += dt * (v * cos(w)); state[1] += dt * (v * sin(w)); }
        u_log[2 * (k - k0)] = v; u_log[2 * (k - k0) + 1] = w;
        iters_log[k - k0] = inf.iters;
        if (info_log) { info_log[3 * (k - k0)] = inf.resi_dual; info_log[3 * (k - k0) + 1] = inf.resi_pri; info_log[3 * (k - k0) + 2] = (double)inf.su_ipm_iters; }
        t_log[k - k0] = now_s() - t0;
    }
    return 0;
}

/* ---- the same loop for a FLEET (BASELINE config C5, "batched multi-ego"): B egos, one rda_fleet, ONE host synchronisation per fleet tick.  Per tick
 * rda_fleet_scene_resort (resort = 2; resort = 1: rda_scene_resort member by member) - the reference re-sorts every robot's obstacle list on every tick
 * (mpc.py:205-206) - then ONE
 * rda_fleet_step_tracked for all members (every member's MPC.pre_process, ADMM loop and result hand-over; it orders itself behind the members'
 * re-staging), then the host applies every member's first control to its kinematic model.  states [B][3], cur_index [B] are read and advanced;
 * u_log [n_steps][B][2], t_log [n_steps] (wall time of a fleet tick), iters_log [n_steps][B], ipm_log [n_steps][B] (may be NULL).  Members whose path
 * ends stop the run (return 1). */
typedef int (*fleet_step_tracked_fn)(rda_fleet *, const double *, const double *, const int32_t *, double, int, const double *, double *, double *,
                                     rda_info *, double *, int32_t *, double *);
typedef int (*fleet_scene_resort_fn)(rda_fleet *, const double *, int);
struct closed_loop_fleet_api { fleet_step_tracked_fn fleet_step_tracked; scene_resort_fn scene_resort; fleet_scene_resort_fn fleet_scene_resort; };

int closed_loop_fleet_run(const struct closed_loop_fleet_api *api, rda_fleet *f, rda_handle *const *egos, int B, int T, int dynamics, double wheelbase,
                          double dt, double ref_speed, double threshold, int ind_range, const int32_t *path_len, int resort, int k0, int n_steps,
                          const double *nom_u_first, double *states, int32_t *cur_index, double *u_log, double *t_log, int32_t *iters_log,
                          int32_t *ipm_log)
{
    if (T > RDA_TMAX || B < 1) return -1;
    const size_t nu = 2 * (size_t)T, ns = 3 * ((size_t)T + 1);
    double *out_u = (double *)malloc(sizeof(double) * B * nu), *out_s = (double *)malloc(sizeof(double) * B * ns);
    double *speed = (double *)malloc(sizeof(double) * B), *eh = (double *)malloc(sizeof(double) * B);
    rda_info *inf = (rda_info *)malloc(sizeof(rda_info) * B);
    int32_t *mi = (int32_t *)malloc(sizeof(int32_t) * B);
    int rc = 0;
    if (!out_u || !out_s || !speed || !eh || !inf || !mi) rc = -1;
    for (int i = 0; i < B && rc == 0; ++i) speed[i] = ref_speed;
    for (int k = k0; k < k0 + n_steps && rc == 0; ++k) {
        const double t0 = now_s();
        if (resort == 2 && api->fleet_scene_resort) rc = api->fleet_scene_resort(f, states, 3);       /* one launch set for all members */
        else if (resort)
            for (int i = 0; i < B && rc >= 0; ++i) rc = api->scene_resort(egos[i], states + 3 * i);      /* member by member (64 x 4 launches on 64 streams) */
        if (rc >= 0)
            rc = api->fleet_step_tracked(f, states, speed, cur_index, threshold, ind_range, k == 0 ? nom_u_first : 0, out_u, out_s, inf, 0, mi, eh);
        if (rc < 0) break;
        rc = 0;
        for (int i = 0; i < B; ++i) {
            double *st = states + 3 * i;
            const double v = out_u[i * nu], w = out_u[i * nu + T], phi = st[2];
            cur_index[i] = mi[i];
            if (mi[i] >= path_len[i] - 1) rc = 1;
            if (dynamics == 0) { st[0] += dt * (v * cos(phi)); st[1] += dt * (v * sin(phi)); st[2] += dt * (v * tan(w) / wheelbase); }
            else if (dynamics == 1) { st[0] += dt * (v * cos(phi)); st[1] += dt * (v * sin(phi)); st[2] += dt * w; }
            else { st[0] += dt * (v * cos(w)); st[1] += dt * (v * sin(w)); }
            u_log[((size_t)(k - k0) * B + i) * 2] = v; u_log[((size_t)(k - k0) * B + i) * 2 + 1] = w;
            iters_log[(size_t)(k - k0) * B + i] = inf[i].iters;
            if (ipm_log) ipm_log[(size_t)(k - k0) * B + i] = inf[i].su_ipm_iters;
        }
        t_log[k - k0] = now_s() - t0;
    }
    free(out_u); free(out_s); free(speed); free(eh); free(inf); free(mi);
    return rc;
}
