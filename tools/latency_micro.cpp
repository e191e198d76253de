#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, long long *cyc, double a, double b)
{
    int lane = threadIdx.x;
    double x = a + lane * 1e-9, y = b;
    long long t0, t1;
    // 1. dependent fma chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fma(x, y, a);
    }
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
    // 2. independent fma (4 chains)
    double x1 = x + 1, x2 = x + 2, x3 = x + 3;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { x = __builtin_fma(x, y, a); x1 = __builtin_fma(x1, y, a); x2 = __builtin_fma(x2, y, a); x3 = __builtin_fma(x3, y, a); }
    }
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    x += x1 + x2 + x3;
    // 3. readlane -> fma dependent chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
            x = __builtin_fma(__hiloint2double(hi, lo), y, x * 0.5);
        }
    }
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
    // 4. bpermute dependent chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __shfl(x, (lane * 5 + 3) & 63, 64) + a;
    }
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
    // 5. rcp chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_amdgcn_rcp(x) + a;
    }
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
    // 6. division chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = a / x + b;
    }
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;
    // 7. LDS write->read roundtrip chain
    __shared__ double sm[64];
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sm[lane] = x; __builtin_amdgcn_wave_barrier(); x = sm[(lane + 1) & 63] + a; __builtin_amdgcn_wave_barrier(); }
    }
    t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
    // 8. DPP row_shr chain (f64 via two movs)
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, false);
            int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, false);
            x = __hiloint2double(hi, lo) + a;
        }
    }
    t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;
    // 9. wall clock vs cycle
    long long w0 = wall_clock64(); t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 20000; ++i) x = __builtin_fma(x, y, a);
    t1 = clock64(); long long w1 = wall_clock64(); if (lane == 0) { cyc[8] = t1 - t0; cyc[9] = w1 - w0; }
    out[lane] = x;
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001, 0.9999999);
    long long h[16]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[] = {"dep fma", "4 indep fma (per fma)", "readlane+fma+mul", "bpermute(f64)+add", "rcp+add", "div+add", "lds rt+add", "dpp+add"};
    for (int i = 0; i < 8; ++i) printf("%-24s %.1f cycles/iter\n", nm[i], h[i] / 16000.0);
    printf("clock64 ticks %lld wall ticks(100MHz) %lld -> clock64 freq %.1f MHz\n", h[8], h[9], h[8] * 100.0 / h[9]);
    return 0;
}
