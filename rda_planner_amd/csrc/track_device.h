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
    long long *prof = nullptr;           // optional: clock64 ticks per phase (tools/track_micro.cpp)
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

// value of lane `l` (wave-uniform index): two v_readlane_b32 instead of the LDS round trip of a general shuffle
__device__ inline double readlane_f64(double v, int l)
{
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}

// One wave per ego.  The three parts of pre_process are serial in the reference; here the lanes take what is
// independent - the window of the path, the distances of closest_point, the trigonometry of the roll-out, the
// circle / segment tests of inter_point (64 consecutive segments at a time, first hit wins like the sequential scan) -
// and lane 0 keeps the running sums in the reference's order, so the values do not depend on the lane count.
// `part`: 0 = everything; 1 = the nominal roll-out only (nom_s, speed); 2 = everything but the stores of part 1.  Parts 1 and 2 may
// run in different workgroups at the same time (k_su_tracked): they share inputs only.
__device__ inline void run(const Ego &e, const In &in, Out &out, double *lds, int lane, int part = 0)
{
#pragma clang fp contract(off)
    const int T = e.T, L = e.L, C = T + 1;
    const int base = in.cur_index;
    long long tprev = e.prof ? clock64() : 0;
    auto mark = [&](int k) { if (e.prof) { const long long now = clock64(); if (lane == 0) e.prof[k] += now - tprev; tprev = now; } };
    double *win = lds, *inc = lds + 3 * WIN, *dist = inc + 5 * 65;
    if (part != 1) for (int i = lane; i < 3 * WIN; i += 64) { const long long gi = 3ll * base + i; win[i] = gi < 3ll * L ? e.path[gi] : 0.0; }
    wsync();
    mark(0);
    auto P = [&](int i, int c) -> double { const int k = i - base; return (k >= 0 && k < WIN) ? win[3 * k + c] : e.path[3 * i + c]; };
    double endh = P(L - 1, 2);                                    // heading of the last waypoint: rewritten below (Q12), stored at the end
    auto heading = [&](int i) -> double { return i == L - 1 ? endh : P(i, 2); };
    // ---- closest_point: the scan stops at the first waypoint closer than the threshold, else the first minimum -----------
    int min_ind = in.cur_index;
    if (part != 1) {
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
    mark(1);
    // ---- roll-out: heading increments and the trigonometry per lane, running sums in stage order --------------------------
    double *dth = inc, *ddx = inc + 65, *ddy = inc + 130, *hth = inc + 195;
    for (int t = lane; t < T; t += 64) {
        const double v = e.nom_u[t], w = e.nom_u[T + t];
        dth[t] = e.dynamics == 0 ? e.dt * (v * tan(w) / e.wheelbase) : (e.dynamics == 1 ? e.dt * w : e.dt * 0.0);
    }
    wsync();
    if (lane == 0) { double th = in.sth; hth[0] = th; for (int t = 0; t < T; ++t) { th = th + dth[t]; hth[t + 1] = th; } }
    wsync();
    if (part != 2) {
        for (int t = lane; t < T; t += 64) {
            const double v = e.nom_u[t], w = e.nom_u[T + t], ang = e.dynamics == 2 ? w : hth[t];
            ddx[t] = e.dt * (v * cos(ang)); ddy[t] = e.dt * (v * sin(ang));
        }
        wsync();
        if (lane == 0) {
            double x = in.sx, y = in.sy;
            e.nom_s[0] = x; e.nom_s[C] = y; e.nom_s[2 * C] = hth[0];
            for (int t = 0; t < T; ++t) { x = x + ddx[t]; y = y + ddy[t]; e.nom_s[t + 1] = x; e.nom_s[C + t + 1] = y; e.nom_s[2 * C + t + 1] = hth[t + 1]; }
            e.speed[0] = in.speed;
        }
    }
    if (part == 1) return;
    mark(2);
    // ---- reference sampling (every lane carries the running point; lane 0 stores) -----------------------------------------
    double tx = P(min_ind, 0), ty = P(min_ind, 1), tth = heading(min_ind);
    bool t_is_end = min_ind == L - 1;                             // the running point is the last waypoint OBJECT
    unsigned long long end_mask = 0; const bool end0 = t_is_end;  // columns that alias the last waypoint
    int cur = in.cur_index;                                       // the segment search restarts at the CALLER's index
    const double move = in.speed * e.dt;
    if (lane == 0) { e.ref[0] = tx; e.ref[C] = ty; e.ref[2 * C] = tth; }
    // Segment cache: the T searches walk forward over the same few segments (cur only grows), and everything of the circle / segment
    // test that does not involve the running point is a property of the segment.  Each lane keeps two segments, c0 + lane and
    // c0 + 64 + lane, in registers (end points, direction, 2 qa, 4 qa, mid heading); a search tests the cached segments >= cur in
    // one go and the cache is refilled further along the path only when a search runs off its end.  Same expressions, same
    // roundings and the same "first segment from cur on" rule as the scan it replaces.
    struct Seg { bool valid, nz, live_h; double ax, ay, dx, dy, qa2, qa4, ha, th; };
    Seg sg[2];
    int c0 = cur;
    auto fill = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            Seg &g = sg[j];
            const int i = c0 + 64 * j + lane;
            g.valid = i + 1 <= L - 1; g.nz = false; g.live_h = false;
            g.ax = g.ay = g.dx = g.dy = g.qa2 = g.qa4 = g.ha = g.th = 0;
            if (g.valid) {
                g.ax = P(i, 0); g.ay = P(i, 1);
                const double bx = P(i + 1, 0), by = P(i + 1, 1);
                g.dx = bx - g.ax; g.dy = by - g.ay;
                g.nz = !(g.dx == 0 && g.dy == 0);
                const double qa = g.dx * g.dx + g.dy * g.dy;
                g.qa2 = 2 * qa; g.qa4 = 4 * qa;
                g.ha = P(i, 2);
                g.live_h = i + 1 == L - 1;                         // the far end is the last waypoint: its heading is rewritten while we go (Q12)
                if (!g.live_h) { const double hb = P(i + 1, 2); g.th = wraptopi(g.ha + wraptopi(hb - g.ha) / 2); }
            }
        }
    };
    fill();
    const double move2 = move * move;
    for (int t = 0; t < T; ++t) {
        const double cth = hth[t + 1];
        // inter_point: first segment from `cur` on that the circle (centre = running point, radius = move) leaves, or the end
        while (true) {
            // block 0 (segments c0 .. c0 + 63) first; block 1 only when the circle leaves none of them
            bool isend = false; double hx = 0, hy = 0;
            auto test = [&](const Seg &g, int i) -> bool {
                const bool on = i >= cur;
                isend = on && !g.valid;
                bool found = false; double t2 = 0;
                if (on && g.valid && g.nz) {
                    const double fx = g.ax - tx, fy = g.ay - ty;
                    const double qb = (2 * fx) * g.dx + (2 * fy) * g.dy, qc = (fx * fx + fy * fy) - move2;
                    const double disc = qb * qb - g.qa4 * qc;
                    if (!(disc < 0)) { t2 = (-qb + sqrt(disc)) / g.qa2; found = t2 >= 0 && t2 <= 1; }
                }
                hx = g.ax + t2 * g.dx; hy = g.ay + t2 * g.dy;
                return isend || found;
            };
            int jb = 0;
            unsigned long long m = __ballot(test(sg[0], c0 + lane));
            if (!m) { jb = 1; m = __ballot(test(sg[1], c0 + 64 + lane)); }
            if (!m) { cur = c0 + 128; c0 = cur; fill(); continue; }              // ran off the cache: go on from its end
            const unsigned long long me = __ballot(isend);
            const int first = __ffsll((long long)m) - 1;                          // wave-uniform (scalar)
            cur = c0 + 64 * jb + first;
            if (me >> first & 1) {                                // end of the path: the last waypoint itself, heading wrapped in place
                endh = wraptopi(endh);
                tx = P(L - 1, 0); ty = P(L - 1, 1); tth = endh; t_is_end = true;
            } else {
                const Seg &g = jb ? sg[1] : sg[0];
                double th = g.th;
                if (__ballot(g.live_h) >> first & 1) th = wraptopi(g.ha + wraptopi(endh - g.ha) / 2);      // (uniform branch, rare)
                tx = readlane_f64(hx, first); ty = readlane_f64(hy, first); tth = readlane_f64(th, first); t_is_end = false;
            }
            break;
        }
        // heading of the reference unwrapped against the predicted heading (in place: on the path when at its end)
        tth = cth + wraptopi(tth - cth);
        if (t_is_end) { endh = tth; end_mask |= 1ull << t; }
        if (lane == 0) { e.ref[t + 1] = tx; e.ref[C + t + 1] = ty; e.ref[2 * C + t + 1] = tth; }
    }
    mark(3);
    if (lane != 0) return;
    // every column that is the last waypoint object shows the value of its last rewrite, which also stays in the path
    if (end0) e.ref[2 * C] = endh;
    for (int t = 0; t < T; ++t) if (end_mask >> t & 1) e.ref[2 * C + t + 1] = endh;
    e.path[3 * (L - 1) + 2] = endh;
    out.min_index = min_ind; out.pad = 0; out.end_heading = endh;
}

}  // namespace track
