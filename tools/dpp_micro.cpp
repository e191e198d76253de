// Checks v_fmac_f64_dpp row_newbcast semantics on gfx950 and times one affine-map sweep stage both ways.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double bcast(double v, int src)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double affine_rl(const double *r, double x)
{
    double e0 = r[5] + r[0] * bcast(x, 0), e1 = r[1] * bcast(x, 1);
    e0 += r[2] * bcast(x, 2); e1 += r[3] * bcast(x, 3); e0 += r[4] * bcast(x, 4);
    return e0 + e1;
}
__device__ __forceinline__ double affine_dpp(const double *r, double x)
{
    double e0 = r[5], e1 = 0.0;
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                 : "+v"(e0), "+v"(e1) : "v"(x), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]));
    return e0 + e1;
}
__global__ void k(double *out, long long *cyc)
{
    int lane = threadIdx.x;
    double r[6];
    for (int i = 0; i < 6; ++i) r[i] = 0.1 * ((lane * 7 + i * 3) % 11) - 0.4;
    double xa = 0.3 + 0.01 * lane, xb = xa;
    long long t0, t1;
    if (lane < 8) {
        t0 = clock64();
#pragma unroll 1
        for (int i = 0; i < 1000; ++i) { xa = affine_rl(r, xa); xa = affine_rl(r, xa); xa = affine_rl(r, xa); xa = affine_rl(r, xa); }
        t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
        t0 = clock64();
#pragma unroll 1
        for (int i = 0; i < 1000; ++i) { xb = affine_dpp(r, xb); xb = affine_dpp(r, xb); xb = affine_dpp(r, xb); xb = affine_dpp(r, xb); }
        t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
    }
    out[lane] = xa; out[64 + lane] = xb;
}
int main()
{
    double *o; long long *c; (void)hipMalloc(&o, 128 * 8); (void)hipMalloc(&c, 16 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c);
    long long h[2]; double ho[128];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    double md = 0; for (int i = 0; i < 8; ++i) { double d = ho[i] - ho[64 + i]; if (d < 0) d = -d; if (d > md) md = d; }
    printf("readlane stage %.1f ticks, dpp stage %.1f ticks, max |diff| over lanes 0..7 = %.3e (x0=%.6f)\n", h[0] / 4000.0, h[1] / 4000.0, md, ho[0]);
    return 0;
}
