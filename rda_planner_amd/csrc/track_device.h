// Caller-side nominal roll-out and reference sampling on the device: what the reference's
// MPC.pre_process does per tick in Python before the solver is entered -
//   closest_point (windowed nearest waypoint with early exit)            mpc.py:338-353
//   motion_predict_model_acker / _diff / _omni (nominal roll-out)        mpc.py:293-336
//   inter_point / range_cir_seg (arc-length resampling of the polyline)  mpc.py:355-417
//   wraptopi, heading of the reference unwrapped against the prediction  mpc.py:283-284,425-433
// - as one thread per ego that writes the solver's step inputs (nominal states, reference, signed speed) where k_su
// reads them.  Products and sums are rounded separately like the Python expressions (no FMA contraction); sin / cos /
// tan come from the device maths library, so the values agree with the host code to the last bits, not bit for bit.
// Quirk Q12 is kept: past the end of the path the reference hands out the LAST WAYPOINT OBJECT itself and rewrites its
// heading in place, so (i) every reference column past the end carries the value of the last rewrite and (ii) the
// rewritten heading stays in the path for the next tick.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace track {

struct In {               // per ego, per tick
    double sx, sy, sth;   // robot state
    double speed;         // signed reference speed (gear * ref_speed)
    double threshold;     // closest_point early-exit distance (0.1)
    int cur_index, ind_range;
};
struct Out { int min_index; int pad; double end_heading; };

struct Ego {
    double *path; int L;                 // [L][3] x, y, heading (the last heading is rewritten, Q12)
    const double *nom_u;                 // [2][T] nominal controls
    double *nom_s, *ref, *speed;         // [3][T+1], [3][T+1], [1]: the step inputs of the solver
    int T, dynamics; double dt, wheelbase;
};

__device__ inline double wraptopi(double r)
{
    const double pi = 3.141592653589793;
    while (r > pi) r = r - 2 * pi;
    while (r < -pi) r = r + 2 * pi;
    return r;
}

__device__ inline void run(const Ego &e, const In &in, Out &out)
{
#pragma clang fp contract(off)
    const int T = e.T, L = e.L, C = T + 1;
    double *P = e.path;
    // ---- closest_point ------------------------------------------------------------------------------------------
    double min_dis = INFINITY; int min_ind = in.cur_index;
    for (int i = in.cur_index; i < in.cur_index + in.ind_range && i < L; ++i) {
        const double dx = in.sx - P[3 * i], dy = in.sy - P[3 * i + 1];
        const double dis = sqrt(dx * dx + dy * dy);
        if (dis < min_dis) { min_dis = dis; min_ind = i; if (dis < in.threshold) break; }
    }
    // ---- roll-out + reference sampling ------------------------------------------------------------------------------
    double cx = in.sx, cy = in.sy, cth = in.sth;                  // predicted state
    double tx = P[3 * min_ind], ty = P[3 * min_ind + 1], tth = P[3 * min_ind + 2];      // running reference point
    bool t_is_end = min_ind == L - 1;                             // ... is the last waypoint OBJECT
    unsigned long long end_mask = 0; bool end0 = t_is_end;        // columns that alias the last waypoint
    int cur = in.cur_index;                                       // the segment search restarts at the CALLER's index
    const double move = in.speed * e.dt;
    e.nom_s[0] = cx; e.nom_s[C] = cy; e.nom_s[2 * C] = cth;
    e.ref[0] = tx; e.ref[C] = ty; e.ref[2 * C] = tth;
    for (int t = 0; t < T; ++t) {
        const double v = e.nom_u[t], w = e.nom_u[T + t];
        if (e.dynamics == 0) {
            const double nx = cx + e.dt * (v * cos(cth)), ny = cy + e.dt * (v * sin(cth)), nth = cth + e.dt * (v * tan(w) / e.wheelbase);
            cx = nx; cy = ny; cth = nth;
        } else if (e.dynamics == 1) {
            const double nx = cx + e.dt * (v * cos(cth)), ny = cy + e.dt * (v * sin(cth)), nth = cth + e.dt * w;
            cx = nx; cy = ny; cth = nth;
        } else {
            cx = cx + e.dt * (v * cos(w)); cy = cy + e.dt * (v * sin(w)); cth = cth + e.dt * 0.0;
        }
        e.nom_s[t + 1] = cx; e.nom_s[C + t + 1] = cy; e.nom_s[2 * C + t + 1] = cth;
        // inter_point: first segment from `cur` on that the circle (centre = running point, radius = move) leaves
        const double ox = tx, oy = ty;
        bool hit = false;
        while (!hit) {
            if (cur + 1 > L - 1) {                                // end of the path: the last waypoint itself
                P[3 * (L - 1) + 2] = wraptopi(P[3 * (L - 1) + 2]);
                tx = P[3 * (L - 1)]; ty = P[3 * (L - 1) + 1]; tth = P[3 * (L - 1) + 2];
                t_is_end = true;
                break;
            }
            const double ax = P[3 * cur], ay = P[3 * cur + 1], bx = P[3 * cur + 3], by = P[3 * cur + 4];
            const double dx = bx - ax, dy = by - ay;
            bool found = false; double t2 = 0;
            if (!(dx == 0 && dy == 0)) {
                const double fx = ax - ox, fy = ay - oy;
                const double qa = dx * dx + dy * dy, qb = (2 * fx) * dx + (2 * fy) * dy, qc = (fx * fx + fy * fy) - move * move;
                const double disc = qb * qb - 4 * qa * qc;
                if (!(disc < 0)) {
                    t2 = (-qb + sqrt(disc)) / (2 * qa);
                    found = t2 >= 0 && t2 <= 1;
                }
            }
            if (!found) { cur = cur + 1; continue; }
            const double ha = P[3 * cur + 2], hb = P[3 * cur + 5];
            const double half = wraptopi(hb - ha) / 2;
            tx = ax + t2 * dx; ty = ay + t2 * dy; tth = wraptopi(ha + half);
            t_is_end = false; hit = true;
        }
        // heading of the reference unwrapped against the predicted heading (in place: on the path when at its end)
        tth = cth + wraptopi(tth - cth);
        if (t_is_end) { P[3 * (L - 1) + 2] = tth; end_mask |= 1ull << t; }
        e.ref[t + 1] = tx; e.ref[C + t + 1] = ty; e.ref[2 * C + t + 1] = tth;
    }
    // every column that is the last waypoint object shows the value of its last rewrite
    const double eh = P[3 * (L - 1) + 2];
    if (end0 && end_mask) e.ref[2 * C] = eh;
    for (int t = 0; t < T; ++t) if (end_mask >> t & 1) e.ref[2 * C + t + 1] = eh;
    e.speed[0] = in.speed;
    out.min_index = min_ind; out.pad = 0; out.end_heading = eh;
}

}  // namespace track
