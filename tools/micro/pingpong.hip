// What a vector burst costs next to the partner wave's back-to-back MFMAs.
// Workgroup = 8 waves (2 per SIMD).  Role alternates every segment, barrier between:
//   M segment: 16 x v_mfma_i32_32x32x32_i8 on four independent accumulators
//   E segment: NV vector ops (v_max3_i32), either on plain registers (SRC = 0) or reading
//              the accumulators the wave's own M segment just wrote (SRC = 1)
// MODE 0: ping-pong (waves 4-7 one segment behind); MODE 1: all waves in step (no overlap
// possible between M and E: the serial reference); PRIO: s_setprio(3) around the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

template <int NV, int SRC, int MODE, int PRIO>
__global__ __launch_bounds__(512, 2) void k(int iters, int *sink, long long *cyc)
{
    v16i acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = threadIdx.x + a + i;
    v4i A = {(int)threadIdx.x, 1, 2, 3}, B = {4, 5, 6, (int)threadIdx.x};
    int v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 3 + i;
    const int G = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    if (MODE == 0 && G == 1) bar();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        bar();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (SRC == 0) v[j & 15] = max(max(v[j & 15], v[(j + 5) & 15]), it + j);
            else v[j & 15] = max(max(v[j & 15], acc[(j >> 4) & 3][j & 15]), acc[((j >> 4) + 1) & 3][(j + 3) & 15]);
            asm volatile("" : "+v"(v[j & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
        bar();
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    if (MODE == 0 && G == 0) bar();
    int s = 0;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    for (int j = 0; j < 16; ++j) s += v[j];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int SRC, int MODE, int PRIO>
void run()
{
    int *sink; long long *cyc;
    hipMalloc(&sink, 512 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipLaunchKernelGGL((k<NV, SRC, MODE, PRIO>), dim3(256), dim3(512), 0, 0, iters, sink, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, SRC, MODE, PRIO>), dim3(256), dim3(512), 0, 0, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per SIMD: 2 waves x 16 MFMAs per iteration = 1024 pipe cycles
    printf("NV=%3d src=%d mode=%d prio=%d : %7.1f cycles per iteration (2 segments), MFMA pipe %.0f %%, %.2f ms, clock %.2f GHz\n",
        NV, SRC, MODE, PRIO, (double)c / iters, 100.0 * 1024.0 / ((double)c / iters), ms, (double)c / (ms * 1e6));
    hipFree(sink); hipFree(cyc);
}

int main()
{
    run<0, 0, 0, 0>(); run<32, 0, 0, 0>(); run<64, 0, 0, 0>(); run<96, 0, 0, 0>(); run<128, 0, 0, 0>();
    run<64, 1, 0, 0>(); run<96, 1, 0, 0>();
    run<64, 0, 0, 1>(); run<96, 0, 0, 1>(); run<64, 1, 0, 1>();
    run<0, 0, 1, 0>(); run<64, 0, 1, 0>(); run<96, 0, 1, 0>(); run<64, 1, 1, 0>();
    return 0;
}
