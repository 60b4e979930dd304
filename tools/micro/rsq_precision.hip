// Relative error of v_rsq_f64 and of one / two Newton steps on it (the pivot chain of the BA's diagonal
// factor spends most of its latency in rsqrt_newton): hipcc --offload-arch=gfx950 -O3 -o rsq_precision rsq_precision.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *d, double *y0, double *y1, double *y2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i], h = 0.5 * x;
    double y = __builtin_amdgcn_rsq(x);
    y0[i] = y;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y1[i] = y;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y2[i] = y;
    {
        // the one third-order step of rsqrt_newton (ba_cholesky.hip), reported in place of y0
        const double s = __builtin_amdgcn_rsq(x);
        const double e = fma(-(x * s), s, 1.0);
        y0[i] = fma(s * e, fma(0.375, e, 0.5), s);
    }
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), r0(n), r1(n), r2(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = std::ldexp(1.0 + u, (int)(s % 41) - 20);
    }
    double *d, *a, *b, *c;
    hipMalloc(&d, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, n);
    hipMemcpy(r0.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), b, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), c, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)h[i]);
        e0 = fmax(e0, (double)fabsl((r0[i] - t) / t)); e1 = fmax(e1, (double)fabsl((r1[i] - t) / t)); e2 = fmax(e2, (double)fabsl((r2[i] - t) / t));
    }
    printf("max relative error: one third-order step on v_rsq_f64 %.3e (2^%.1f), one Newton step %.3e (2^%.1f), two %.3e (2^%.1f)\n",
        e0, log2(e0), e1, log2(e1), e2, log2(e2));
    return 0;
}
