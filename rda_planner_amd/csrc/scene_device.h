// Caller-side obstacle pipeline on the device: what the reference does per MPC tick in Python before
// the solver is entered -
//   MPC.convert_rda_obstacle / rda_obs_distance / sort          mpc.py:189-218
//   convert_inequal_circle / convert_inequal_polygon            mpc.py:440-472
//   gen_inequal_global, is_convex_and_ordered, cross_product    mpc.py:492-549
//   RDA_solver.assign_obstacle_parameter (truncate / pad / zero rows, per-t replication)   rda_solver.py:483-526
// - as three small kernels that write the solver's obstacle slots A [N][nt][E][2], b [N][nt][E], cone [N] directly.
// The arithmetic reproduces the numpy expressions operation by operation (FMA contraction is switched off in these
// kernels), so the slots are BIT-IDENTICAL to what the Python caller stages (tests/test_gpu_scene.py).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace scene {

struct Args {
    int n, N, E, T, nt, order;
    double dt;
    const int *kind;        // [n] 0 polygon (Rpositive), 1 circle (norm2)
    const int *nvert;       // [n] polygon vertex count (<= E)
    const double *geom;     // [n][E][2]: polygon vertices | circle: centre, (radius, -)
    const double *vel;      // [n][2]
    const double *robot;    // [2] robot position used for the distance ordering
    double rx, ry; int robot_val;   // robot_val != 0: the position travels in the kernel arguments (rda_scene_resort: a resident scene re-ranked without a copy)
    double *key; int *sel;  // [n] scratch
    double *A, *b; int *cone;
    int *nonconvex;         // count of polygons failing is_convex_and_ordered (the reference prints a warning)
};

// distance key of one obstacle: rda_obs_distance, mpc.py:214-218 (circle: centre distance; polygon: nearest vertex)
__device__ __forceinline__ void keys_body(const Args &a, const int i)
{
#pragma clang fp contract(off)        // numpy does not fuse: keep every product and sum separately rounded
    if (i >= a.n) return;
    if (!a.order) { a.key[i] = (double)i; return; }
    const double x = a.robot_val ? a.rx : a.robot[0], y = a.robot_val ? a.ry : a.robot[1];
    const double *g = a.geom + (size_t)i * a.E * 2;
    if (a.kind[i] == 1) {
        double dx = x - g[0], dy = y - g[1];
        a.key[i] = sqrt(dx * dx + dy * dy);
    } else {
        double best = INFINITY;
        for (int j = 0; j < a.nvert[i]; ++j) {
            double dx = x - g[2 * j], dy = y - g[2 * j + 1];
            double d = sqrt(dx * dx + dy * dy);
            if (d < best) best = d;
        }
        a.key[i] = best;
    }
}
__global__ void k_keys(Args a) { keys_body(a, blockIdx.x * blockDim.x + threadIdx.x); }

// stable rank by key (Python's list.sort is stable); the first min(n, N) survive (rda_solver.py:491-493)
// O(n^2) compares spread over 16 lanes per key (256-thread blocks = 16 keys; launch with (n + 15) / 16 blocks): a count, so the
// result does not depend on how the compares are split
__device__ __forceinline__ void rank_body(const Args &a, const int blk)
{
    const int i = blk * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    const bool live = i < a.n;
    const double ki = live ? a.key[i] : 0.0;
    int r = 0;
    if (live) for (int j = l; j < a.n; j += 16) { double kj = a.key[j]; r += (kj < ki) || (kj == ki && j < i); }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) r += __shfl_xor(r, off, 16);
    if (live && l == 0 && r < a.N) a.sel[r] = i;
}
__global__ void k_rank(Args a) { rank_body(a, blockIdx.x); }

// one thread per (slot, time slot): half-space form of the selected obstacle at time t
__device__ __forceinline__ void build_body(const Args &a, const int w)
{
#pragma clang fp contract(off)
    if (w >= a.N * a.nt) return;
    const int s = w / a.nt, t = w % a.nt;
    const int used = a.n < a.N ? a.n : a.N;
    const int i = a.sel[s < used ? s : used - 1];                 // quirk Q3: pad with copies of the last obstacle
    const int E = a.E;
    const double *g = a.geom + (size_t)i * E * 2;
    double *A = a.A + ((size_t)s * a.nt + t) * E * 2, *b = a.b + ((size_t)s * a.nt + t) * E;
    for (int e = 0; e < E; ++e) { A[2 * e] = 0; A[2 * e + 1] = 0; b[e] = 0; }
    const double vx = a.vel[2 * i], vy = a.vel[2 * i + 1];
    const bool moving = sqrt(vx * vx + vy * vy) > 0.01;     // mpc.py:443,462
    const double tt = (double)t * a.dt;
    const double sx = moving ? vx * tt : 0.0, sy = moving ? vy * tt : 0.0;
    if (a.kind[i] == 1) {
        if (t == 0) a.cone[s] = 1;
        A[0] = 1; A[3] = 1;                                        // [[1,0],[0,1],[0,0]]   mpc.py:441
        b[0] = moving ? g[0] + sx : g[0];
        b[1] = moving ? g[1] + sy : g[1];
        b[2] = -g[2];
        return;
    }
    if (t == 0) a.cone[s] = 0;
    const int k = a.nvert[i];
    double px[16], py[16];
    for (int j = 0; j < k; ++j) { px[j] = moving ? g[2 * j] + sx : g[2 * j]; py[j] = moving ? g[2 * j + 1] + sy : g[2 * j + 1]; }
    // is_convex_and_ordered, mpc.py:527-549
    int direction = 0; bool convex = k >= 3;
    for (int j = 0; j < k && convex; ++j) {
        int j1 = (j + 1) % k, j2 = (j + 2) % k;
        double cr = (px[j1] - px[j]) * (py[j2] - py[j]) - (py[j1] - py[j]) * (px[j2] - px[j]);
        if (cr != 0) {
            if (direction == 0) direction = cr > 0 ? 1 : -1;
            else if ((cr > 0) != (direction > 0)) convex = false;
        }
    }
    if (!convex && t == 0 && s < used) atomicAdd(a.nonconvex, 1);
    const bool cw = convex && direction <= 0;                      // order == 'CW' (direction 0 also reports CW), :500-501
    for (int j = 0; j < k; ++j) {
        int c0 = cw ? k - 1 - j : j, c1 = cw ? (k - 1 - ((j + 1) % k)) : (j + 1) % k;
        double ex = px[c1] - px[c0], ey = py[c1] - py[c0];
        double a0 = ey, a1 = -ex;                                  // A = [edge_y, -edge_x]   :505-506
        A[2 * j] = a0; A[2 * j + 1] = a1;
        b[j] = a0 * px[c0] + a1 * py[c0];   // sum(A * cur, axis=1)   :507
    }
}
__global__ void k_build(Args a) { build_body(a, blockIdx.x * blockDim.x + threadIdx.x); }

// The same three kernels for a FLEET (rda_fleet_scene_resort, round 6): blockIdx.y = the member, its Args in a device array, its robot position in
// rob [B][2] - one launch set re-sorts the resident scenes of all members about their robots (64 members x 4 launches on 64 streams cost the host
// 4 ms per fleet tick; this is 4 launches).  Same device code per member: the slots come out bit-identical to rda_scene_resort on the member.
__device__ __forceinline__ Args fleet_args(const Args *as, const double *rob)
{
    Args a = as[blockIdx.y];
    a.order = 1; a.robot_val = 1; a.rx = rob[2 * blockIdx.y]; a.ry = rob[2 * blockIdx.y + 1];
    return a;
}
__global__ void k_keys_fleet(const Args *as, const double *rob)
{
    const Args a = fleet_args(as, rob);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.nonconvex = 0;        // (k_build, later in the stream, counts into it)
    keys_body(a, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void k_rank_fleet(const Args *as, const double *rob) { const Args a = fleet_args(as, rob); rank_body(a, blockIdx.x); }
__global__ void k_build_fleet(const Args *as, const double *rob) { const Args a = fleet_args(as, rob); build_body(a, blockIdx.x * blockDim.x + threadIdx.x); }

}  // namespace scene
