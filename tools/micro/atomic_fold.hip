// How fast are the atomics the column folding would need?  Every wave issues, per "merge", one
// wave-wide returning atomicMax on 64 consecutive ColPart.best entries (stride 8 B) and one
// non-returning atomicMax on the .second entries, on the columns of a (pair, segment) region
// that ~60 other workgroups of the same XCD hit as well -- the access pattern of the tile kernel.
//   hipcc -O3 --offload-arch=gfx950 -o atomic_fold.bin atomic_fold.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void
fold_kernel(int *cols, int ncols_per_pair, int pairs, int merges, int mode, int spin, long long *lat_out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // workgroups of one "pair" are consecutive (64 per pair here)
    const int pair = (blockIdx.x / 64) % pairs;
    int *base = cols + (size_t)pair * ncols_per_pair * 2;
    unsigned x = blockIdx.x * 2654435761u + threadIdx.x;
    long long lat = 0;
    int pending_old = 0, pending_k = 0, pending_col = 0;
    for (int m = 0; m < merges; ++m) {
        // stand-in for four tiles of work
        for (int s = 0; s < spin; ++s) x = x * 1664525u + 1013904223u;
        const int col = ((m * 4 + wave) * 64 + lane) % ncols_per_pair;
        const int k = (int)(x >> 3);
        if (mode == 0) {                      // returning atomic, consumed at once
            const long long t0 = clock64();
            const int old = atomicMax(base + 2 * col, k);
            const int c = min(old, k);
            lat += clock64() - t0;
            __hip_atomic_fetch_max(base + 2 * col + 1, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (mode == 1) {               // returning atomic, consumed one merge later
            if (m > 0) __hip_atomic_fetch_max(base + 2 * pending_col + 1, min(pending_old, pending_k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending_old = atomicMax(base + 2 * col, k); pending_k = k; pending_col = col;
        } else if (mode == 2) {               // threshold load first, atomics only when it can matter
            const long long t0 = clock64();
            const int thr = __hip_atomic_load(base + 2 * col + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lat += clock64() - t0;
            if (k > thr) {
                const int old = atomicMax(base + 2 * col, k);
                const int c = min(old, k);
                if (c > thr) __hip_atomic_fetch_max(base + 2 * col + 1, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {                              // plain 8-byte store (what the tile kernel does now)
            reinterpret_cast<int2 *>(base)[col + (size_t)(blockIdx.x % 64) * 0] = make_int2(k, k);
        }
    }
    if (lane == 0 && lat_out) atomicAdd((unsigned long long *)lat_out, (unsigned long long)lat);
    if (x == 12345u) cols[0] = 1;
}
int main()
{
    const int ncols = 20480, pairs = 1225, merges = 32;          // 128 tiles per workgroup
    int *d; long long *dl;
    hipMalloc(&d, (size_t)pairs * ncols * 8); hipMalloc(&dl, 8);
    const int blocks = 1225 * 64;                                  // a quarter of the real launch
    for (int spin : {0, 3000}) for (int mode = 0; mode < 4; ++mode) {
        hipMemset(d, 0xC0, (size_t)pairs * ncols * 8); hipMemset(dl, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(fold_kernel, dim3(blocks), dim3(256), 0, 0, d, ncols, pairs, merges, mode, spin, dl);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long lat; hipMemcpy(&lat, dl, 8, hipMemcpyDeviceToHost);
        const double ops = (double)blocks * 256 * merges;
        printf("spin %4d mode %d: %8.3f ms, %7.1f G lane-merges/s, mean wait %6.0f cycles (per wave-merge)\n", spin, mode, ms,
            ops / ms * 1e-6, (double)lat / ((double)blocks * 4 * merges));
    }
    return 0;
}
