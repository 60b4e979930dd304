// Latency of dependent double-precision operations on one wave (the pivot chain of the BA's diagonal factor):
// cycles per link of a chain of v_fma_f64, of v_rsq_f64, of a v_readlane pair feeding a v_fma_f64, and of
// one link of the factor's chain (scale, square, readlane, compare, rsq, third-order step).
//   hipcc --offload-arch=gfx950 -O3 -o f64_chain f64_chain.hip && ./f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readlane_d(double x, int l)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__global__ void k(double *out, long long *cyc, double seed)
{
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3;
    long long t0, t1;
    // 1. fma chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) { x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 0.9999999, 1e-9); }
    t1 = clock64(); if (lane == 0) cyc[0] = (t1 - t0) / 1024;
    // 2. rsq chain
    double y = x;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) { y = __builtin_amdgcn_rsq(y); y = __builtin_amdgcn_rsq(y); y = __builtin_amdgcn_rsq(y); y = __builtin_amdgcn_rsq(y); }
    t1 = clock64(); if (lane == 0) cyc[1] = (t1 - t0) / 1024;
    // 3. readlane pair + fma
    double z = y + x;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        z = fma(z, readlane_d(z, 3), 1e-9); z = fma(z, readlane_d(z, 5), 1e-9); z = fma(z, readlane_d(z, 7), 1e-9); z = fma(z, readlane_d(z, 9), 1e-9);
    }
    t1 = clock64(); if (lane == 0) cyc[2] = (t1 - t0) / 1024;
    // 4. one link of the factor: vj = v * rinv; dn = fma(-vj, vj, w); d = readlane; check; rinv = rsq + third-order step
    double v = 2.0 + lane * 1e-3, w = 9.0, rinv = 0.7;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double vj = v * rinv;
            const double dn = fma(-vj, vj, w);
            double d = readlane_d(dn, 5 + u);
            if (!(d > 0.0)) d = 1.0;
            const double s = __builtin_amdgcn_rsq(d);
            const double e = fma(-(d * s), s, 1.0);
            rinv = fma(s * e, fma(0.375, e, 0.5), s);
        }
    }
    t1 = clock64(); if (lane == 0) cyc[3] = (t1 - t0) / 1024;
    out[lane] = x + y + z + rinv;
}
int main()
{
    double *o; long long *c;
    hipMalloc(&o, 64 * 8); hipMalloc(&c, 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.5);
    long long h[4];
    hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    printf("cycles per link: v_fma_f64 %lld, v_rsq_f64 %lld, v_readlane pair + v_fma_f64 %lld, one link of the factor's pivot chain %lld\n", h[0], h[1], h[2], h[3]);
    return 0;
}
