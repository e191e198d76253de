// Micro-benchmark behind DESIGN.md 5 "no MFMA": the only dense contraction of the hot path is the pose product of
// k_lammuz (reference assign_combine_parameter_stateobs, rda_solver.py:544-568):  per (obstacle, stage)
//     M = A R(phi)  (E x 2 . 2 x 2),   q = A p - b  (E x 2 . 2 x 1)
// i.e. one [E x 3] . [3 x 3] product per sub-problem, E <= 8.  The packed kernel holds FOUR sub-problems per wavefront (one per
// 16-lane row), so one v_mfma_f64_16x16x4_f64 could take the 16 edge rows [ax ay b 0] of a wave against a 4 x 16 operand that
// carries the four sub-problems' [R | p ; -1] blocks side by side - 12 of the 256 outputs are wanted (the diagonal 4 x 3 blocks).
// This program times both forms on the data layout the kernel has (edge rows in per-row LDS slabs, pose in registers of the row)
// and checks that they agree:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_micro tools/mfma_micro.cpp && /tmp/mfma_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

struct Slab { double A[4][2]; double b[4]; double q[4]; double M[4][2]; };      // one 16-lane row's obstacle data (E = 4)

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k(const double *in, double *out, long long *cyc, int reps)
{
    __shared__ Slab sl[4];
    __shared__ double pose[4][4];         // cs, sn, px, py of the four rows
    const int lane = threadIdx.x, row = lane >> 4, gl = lane & 15;
    if (gl < 8) sl[row].A[gl >> 1][gl & 1] = in[row * 12 + gl];
    if (gl < 4) sl[row].b[gl] = in[row * 12 + 8 + gl];
    const double phi = in[48 + row], px = in[52 + row], py = in[56 + row];
    const double cs = cos(phi), sn = sin(phi);
    if (gl == 0) { pose[row][0] = cs; pose[row][1] = sn; pose[row][2] = px; pose[row][3] = py; }
    __syncthreads();
    double acc = 0;
    // ---- (a) VALU form: lane e < E of every row computes its edge's three numbers (what lmz::pose_products does) ----------
    long long t0 = clock64();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        if (gl < 4) {
            const double ax = sl[row].A[gl][0], ay = sl[row].A[gl][1];
            sl[row].q[gl] = ax * px + ay * py - sl[row].b[gl] + acc;
            sl[row].M[gl][0] = ax * cs + ay * sn;
            sl[row].M[gl][1] = -ax * sn + ay * cs;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        acc += sl[row].q[gl & 3] * 1e-300;          // consume: the next repetition depends on this one (a latency chain, like the kernel's)
    }
    long long t1 = clock64();
    if (lane == 0) cyc[0] = t1 - t0;
    double va[3] = {0, 0, 0};
    if (gl < 4) { va[0] = sl[row].q[gl]; va[1] = sl[row].M[gl][0]; va[2] = sl[row].M[gl][1]; }
    __syncthreads();
    // ---- (b) MFMA form: D(16x16) = A(16x4) . B(4x16);  A row 4g+e = [ax ay b 0] of edge e of sub-problem g;
    //      B column 4g+c = c == 0: [cs sn 0 0]', c == 1: [-sn cs 0 0]', c == 2: [px py -1 0]' of sub-problem g.
    //      Operand layout of v_mfma_f64_16x16x4_f64: A: lane l holds A[l % 16][l / 16]; B: lane l holds B[l / 16][l % 16];
    //      D: lane l holds D[(l / 16) + 4 i][l % 16], i = 0..3. ---------------------------------------------------------
    acc = 0;
    t0 = clock64();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        const int kk = lane >> 4, rr = lane & 15, g = rr >> 2, e = rr & 3;           // A operand: row rr = edge e of sub-problem g, column kk
        const double av = kk == 0 ? sl[g].A[e][0] : (kk == 1 ? sl[g].A[e][1] : (kk == 2 ? sl[g].b[e] : 0.0));
        const int c = rr & 3;                                                       // B operand: row kk, column rr = component c of sub-problem g
        const double pc = pose[g][0], ps = pose[g][1], ppx = pose[g][2], ppy = pose[g][3];
        double bv = 0.0;
        if (c == 0) bv = kk == 0 ? pc : (kk == 1 ? ps : 0.0);
        else if (c == 1) bv = kk == 0 ? -ps : (kk == 1 ? pc : 0.0);
        else if (c == 2) bv = kk == 0 ? ppx : (kk == 1 ? ppy : (kk == 2 ? -1.0 : 0.0));
        d4 d = {acc, acc, acc, acc};
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, d, 0, 0, 0);
        // wanted: D[4 g + e][4 g + comp]; register i of lane (q = l / 16, col = l % 16) is D[q + 4 i][col], so with g = col / 4 the lane's
        // register g is edge q of sub-problem g, component col % 4: 48 lanes hold one wanted number each
        {
            const int g2 = gl >> 2, comp = gl & 3;
            const double v = g2 == 0 ? d[0] : (g2 == 1 ? d[1] : (g2 == 2 ? d[2] : d[3]));
            if (comp == 2) sl[g2].q[row] = v; else if (comp < 2) sl[g2].M[row][comp] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        acc += sl[row].q[gl & 3] * 1e-300;
    }
    t1 = clock64();
    if (lane == 0) cyc[1] = t1 - t0;
    double vb[3] = {0, 0, 0};
    if (gl < 4) { vb[0] = sl[row].q[gl]; vb[1] = sl[row].M[gl][0]; vb[2] = sl[row].M[gl][1]; }
    if (gl < 4) for (int i = 0; i < 3; ++i) { out[(row * 4 + gl) * 6 + i] = va[i]; out[(row * 4 + gl) * 6 + 3 + i] = vb[i]; }
}

int main()
{
    double h_in[60], h_out[96];
    for (int i = 0; i < 60; ++i) h_in[i] = std::sin(1.0 + 0.37 * i) * 3.0;
    double *d_in, *d_out; long long *d_cyc, h_cyc[2];
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_cyc, sizeof(h_cyc));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, d_cyc, reps);
    hipDeviceSynchronize();
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost); hipMemcpy(h_cyc, d_cyc, sizeof(h_cyc), hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int c = 0; c < 3; ++c) err = std::fmax(err, std::fabs(h_out[i * 6 + c] - h_out[i * 6 + 3 + c]));
    printf("pose products of 4 sub-problems (16 edges) per repetition, clock64 ticks per repetition incl. the LDS hand-over:\n");
    printf("  VALU form (lmz::pose_products)        %8.1f\n", (double)h_cyc[0] / reps);
    printf("  v_mfma_f64_16x16x4_f64 form           %8.1f\n", (double)h_cyc[1] / reps);
    printf("  max |difference| of the results        %.3e\n", err);
    return err < 1e-12 ? 0 : 1;
}
