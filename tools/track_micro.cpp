// Phase timing of the device-side pre_process (track::run, csrc/track_device.h) on a straight 800-waypoint path:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/track_micro tools/track_micro.cpp && /tmp/track_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../rda_planner_amd/csrc/track_device.h"
#define OK(x) do { if ((x) != hipSuccess) { printf("HIP error at line %d\n", __LINE__); return 1; } } while (0)

__global__ void k(track::Ego e, track::In in, track::Out *out)
{
    __shared__ double win[track::LDS_DOUBLES];
    track::run(e, in, *out, win, threadIdx.x);
}

int main()
{
    const int L = 800, T = 20;
    std::vector<double> path(3 * L);
    for (int i = 0; i < L; ++i) { path[3 * i] = 0.1 * i; path[3 * i + 1] = 25.0; path[3 * i + 2] = 0.0; }
    std::vector<double> nu(2 * T);
    for (int t = 0; t < T; ++t) { nu[t] = 4.0; nu[T + t] = 0.01 * t; }
    double *dpath, *dnu, *dstep; track::Out *dout; long long *dprof;
    OK(hipMalloc(&dpath, path.size() * 8)); OK(hipMalloc(&dnu, nu.size() * 8)); OK(hipMalloc(&dstep, (6 * (T + 1) + 2 * T + 1) * 8));
    OK(hipMalloc(&dout, sizeof(track::Out))); OK(hipMalloc(&dprof, 8 * sizeof(long long)));
    OK(hipMemcpy(dpath, path.data(), path.size() * 8, hipMemcpyHostToDevice)); OK(hipMemcpy(dnu, nu.data(), nu.size() * 8, hipMemcpyHostToDevice));
    track::Ego e; e.path = dpath; e.L = L; e.nom_u = dnu; e.nom_s = dstep; e.ref = dstep + 3 * (T + 1) + 2 * T; e.speed = dstep + 6 * (T + 1) + 2 * T;
    e.T = T; e.dynamics = 0; e.dt = 0.1; e.wheelbase = 3.0; e.prof = dprof;
    track::In in; in.sx = 12.03; in.sy = 25.2; in.sth = 0.05; in.speed = 4.0; in.threshold = 0.1; in.cur_index = 118; in.ind_range = 10;
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; ++rep) {
        OK(hipMemset(dprof, 0, 8 * sizeof(long long)));
        OK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, e, in, dout);
        OK(hipEventRecord(b, 0));
        OK(hipDeviceSynchronize());
        float ms; OK(hipEventElapsedTime(&ms, a, b));
        long long hp[8]; OK(hipMemcpy(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost));
        track::Out o; OK(hipMemcpy(&o, dout, sizeof(o), hipMemcpyDeviceToHost));
        printf("rep %d: %.1f us by events; ticks window=%lld closest=%lld rollout=%lld sampling=%lld  (min_index %d)\n", rep, ms * 1e3, hp[0], hp[1], hp[2], hp[3], o.min_index);
    }
    return 0;
}
