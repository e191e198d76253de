// Caller-side nominal roll-out and reference sampling on the device: what the reference's
// MPC.pre_process does per tick in Python before the solver is entered -
//   closest_point (windowed nearest waypoint with early exit)            mpc.py:338-353
//   motion_predict_model_acker / _diff / _omni (nominal roll-out)        mpc.py:293-336
//   inter_point / range_cir_seg (arc-length resampling of the polyline)  mpc.py:355-417
//   wraptopi, heading of the reference unwrapped against the prediction  mpc.py:283-284,425-433
// - as one wave per ego (the lanes stage a window of the path in LDS, lane 0 does the serial walk) that writes the solver's step inputs (nominal states, reference, signed speed) where k_su
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

constexpr int WIN = 384;      // waypoints of the path kept in LDS around the caller's index (beyond: global memory)
constexpr int LDS_DOUBLES = 3 * WIN + 5 * 65 + WIN;        // window | roll-out scratch | closest_point distances

__device__ inline void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave per ego.  The three parts of pre_process are serial in the reference; here the lanes take what is
// independent - the window of the path, the distances of closest_point, the trigonometry of the roll-out, the
// circle / segment tests of inter_point (64 consecutive segments at a time, first hit wins like the sequential scan) -
// and lane 0 keeps the running sums in the reference's order, so the values do not depend on the lane count.
__device__ inline void run(const Ego &e, const In &in, Out &out, double *lds, int lane)
{
#pragma clang fp contract(off)
    const int T = e.T, L = e.L, C = T + 1;
    const int base = in.cur_index;
    double *win = lds, *inc = lds + 3 * WIN, *dist = inc + 5 * 65;
    for (int i = lane; i < 3 * WIN; i += 64) { const long long gi = 3ll * base + i; win[i] = gi < 3ll * L ? e.path[gi] : 0.0; }
    wsync();
    auto P = [&](int i, int c) -> double { const int k = i - base; return (k >= 0 && k < WIN) ? win[3 * k + c] : e.path[3 * i + c]; };
    double endh = P(L - 1, 2);                                    // heading of the last waypoint: rewritten below (Q12), stored at the end
    auto heading = [&](int i) -> double { return i == L - 1 ? endh : P(i, 2); };
    // ---- closest_point: the scan stops at the first waypoint closer than the threshold, else the first minimum -----------
    int min_ind = in.cur_index;
    {
        const int hi = (in.cur_index + in.ind_range < L) ? in.cur_index + in.ind_range : L;
        double min_dis = INFINITY; bool done = false;
        for (int c0 = in.cur_index; c0 < hi && !done; c0 += WIN) {
            const int n = hi - c0 < WIN ? hi - c0 : WIN;
            for (int k = lane; k < n; k += 64) {
                const double dx = in.sx - P(c0 + k, 0), dy = in.sy - P(c0 + k, 1);
                dist[k] = sqrt(dx * dx + dy * dy);
            }
            wsync();
            for (int k = 0; k < n; ++k) {                         // uniform: every lane walks the same few values
                const double dis = dist[k];
                if (dis < min_dis) { min_dis = dis; min_ind = c0 + k; if (dis < in.threshold) { done = true; break; } }
            }
            wsync();
        }
    }
    // ---- roll-out: heading increments and the trigonometry per lane, running sums in stage order --------------------------
    double *dth = inc, *ddx = inc + 65, *ddy = inc + 130, *hth = inc + 195;
    for (int t = lane; t < T; t += 64) {
        const double v = e.nom_u[t], w = e.nom_u[T + t];
        dth[t] = e.dynamics == 0 ? e.dt * (v * tan(w) / e.wheelbase) : (e.dynamics == 1 ? e.dt * w : e.dt * 0.0);
    }
    wsync();
    if (lane == 0) { double th = in.sth; hth[0] = th; for (int t = 0; t < T; ++t) { th = th + dth[t]; hth[t + 1] = th; } }
    wsync();
    for (int t = lane; t < T; t += 64) {
        const double v = e.nom_u[t], w = e.nom_u[T + t], ang = e.dynamics == 2 ? w : hth[t];
        ddx[t] = e.dt * (v * cos(ang)); ddy[t] = e.dt * (v * sin(ang));
    }
    wsync();
    if (lane == 0) {
        double x = in.sx, y = in.sy;
        e.nom_s[0] = x; e.nom_s[C] = y; e.nom_s[2 * C] = hth[0];
        for (int t = 0; t < T; ++t) { x = x + ddx[t]; y = y + ddy[t]; e.nom_s[t + 1] = x; e.nom_s[C + t + 1] = y; e.nom_s[2 * C + t + 1] = hth[t + 1]; }
    }
    // ---- reference sampling (every lane carries the running point; lane 0 stores) -----------------------------------------
    double tx = P(min_ind, 0), ty = P(min_ind, 1), tth = heading(min_ind);
    bool t_is_end = min_ind == L - 1;                             // the running point is the last waypoint OBJECT
    unsigned long long end_mask = 0; const bool end0 = t_is_end;  // columns that alias the last waypoint
    int cur = in.cur_index;                                       // the segment search restarts at the CALLER's index
    const double move = in.speed * e.dt;
    if (lane == 0) { e.ref[0] = tx; e.ref[C] = ty; e.ref[2 * C] = tth; }
    for (int t = 0; t < T; ++t) {
        const double cth = hth[t + 1];
        // inter_point: first segment from `cur` on that the circle (centre = running point, radius = move) leaves, or the end
        while (true) {
            const int sg = cur + lane;
            const bool isend = sg + 1 > L - 1;
            bool found = false; double t2 = 0, ax = 0, ay = 0, dx = 0, dy = 0;
            if (!isend) {
                ax = P(sg, 0); ay = P(sg, 1);
                const double bx = P(sg + 1, 0), by = P(sg + 1, 1);
                dx = bx - ax; dy = by - ay;
                if (!(dx == 0 && dy == 0)) {
                    const double fx = ax - tx, fy = ay - ty;
                    const double qa = dx * dx + dy * dy, qb = (2 * fx) * dx + (2 * fy) * dy, qc = (fx * fx + fy * fy) - move * move;
                    const double disc = qb * qb - 4 * qa * qc;
                    if (!(disc < 0)) { t2 = (-qb + sqrt(disc)) / (2 * qa); found = t2 >= 0 && t2 <= 1; }
                }
            }
            double hx = 0, hy = 0, th = 0;
            if (found) {
                const double ha = heading(sg), hb = heading(sg + 1);
                hx = ax + t2 * dx; hy = ay + t2 * dy; th = wraptopi(ha + wraptopi(hb - ha) / 2);
            }
            const unsigned long long m = __ballot(isend || found);
            if (!m) { cur += 64; continue; }
            const int first = __ffsll((long long)m) - 1;
            cur += first;
            if (__shfl((int)isend, first, 64)) {                  // end of the path: the last waypoint itself, heading wrapped in place
                endh = wraptopi(endh);
                tx = P(L - 1, 0); ty = P(L - 1, 1); tth = endh; t_is_end = true;
            } else {
                tx = __shfl(hx, first, 64); ty = __shfl(hy, first, 64); tth = __shfl(th, first, 64); t_is_end = false;
            }
            break;
        }
        // heading of the reference unwrapped against the predicted heading (in place: on the path when at its end)
        tth = cth + wraptopi(tth - cth);
        if (t_is_end) { endh = tth; end_mask |= 1ull << t; }
        if (lane == 0) { e.ref[t + 1] = tx; e.ref[C + t + 1] = ty; e.ref[2 * C + t + 1] = tth; }
    }
    if (lane != 0) return;
    // every column that is the last waypoint object shows the value of its last rewrite, which also stays in the path
    if (end0) e.ref[2 * C] = endh;
    for (int t = 0; t < T; ++t) if (end_mask >> t & 1) e.ref[2 * C + t + 1] = endh;
    e.path[3 * (L - 1) + 2] = endh;
    e.speed[0] = in.speed;
    out.min_index = min_ind; out.pad = 0; out.end_heading = endh;
}

}  // namespace track
