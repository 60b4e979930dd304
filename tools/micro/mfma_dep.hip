// Issue rate of v_mfma_i32_32x32x32_i8 for chains of dependent accumulations:
// NACC independent accumulators updated round-robin, optionally with VPER
// independent VALU ops between consecutive MFMAs.  One wave per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NACC, int VPER>
__global__ __launch_bounds__(64) void k(long long *out, int iters, int *sink)
{
    v16i acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0;
    v4i A = {(int)threadIdx.x, 1, 2, 3}, B = {4, 5, 6, (int)threadIdx.x};
    int v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < VPER; ++j) {
                v[j & 7] = max(max(v[j & 7], v[(j + 3) & 7]), it);
                asm volatile("" : "+v"(v[j & 7]));
            }
        }
    }
    const long long t1 = clock64();
    int s = 0;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NACC, int VPER>
void run(int waves_per_simd)
{
    long long *d; int *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4 * 64 * 4096);
    const int iters = 20000;
    // blocks land on distinct CUs first; use 256 * 4 * w blocks to put w waves on every SIMD
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL((k<NACC, VPER>), dim3(blocks), dim3(64), 0, 0, d, iters, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, VPER>), dim3(blocks), dim3(64), 0, 0, d, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;
    printf("acc=%d valu/mfma=%2d waves/simd=%d : %.1f ns per MFMA per wave, clock64 ticks/MFMA %.1f, chip rate %.0f TOPS\n",
        NACC, VPER, waves_per_simd, ms * 1e6 / n, (double)c / n, blocks * n * 65536.0 / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(sink);
}

int main()
{
    for (int w = 1; w <= 4; ++w) {
        run<1, 0>(w); run<2, 0>(w); run<4, 0>(w);
        run<1, 4>(w); run<2, 4>(w);
        run<1, 6>(w); run<2, 6>(w); run<4, 6>(w);
        run<1, 7>(w); run<2, 7>(w);
        run<1, 8>(w); run<2, 8>(w);
    }
    return 0;
}
