// Where the cycles of the BA's diagonal factor go (orthosfm_amd/csrc/ba_cholesky.hip::factor_diag_block):
// the whole routine and ablations of it on one wave, and the latency / issue interval of v_mfma_f64_16x16x4_f64.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -w -o factor_bench.bin factor_bench.hip
#include "../../orthosfm_amd/csrc/ba_cholesky.hip"
#include <cstdio>
#include <vector>
namespace osfm { void set_error(const char *, ...) {} }
using namespace osfm;

__global__ __launch_bounds__(128) void bench_factor(const double *A, double *Ldiag, int *info, long long *cyc, int reps)
{
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Li[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) FactorComm fc;
    const int tid = threadIdx.x;
    long long total = 0, chain = 0;
    if (tid == 0) { fc.prog[0] = 0; fc.prog[1] = 0; }
    for (int r = 0; r < reps; ++r) {
        if (tid < NB) for (int c = 0; c < NB; ++c) M[tid][c] = A[tid * NB + c];
        __syncthreads();
        const long long t0 = clock64();
        factor_diag_block<false>(&M[0][0], NB + 1, 0, Ldiag, info, fc.ab, fc.l1, fc.prog, &Li[0][0], NB + 1);
        if (tid < 64) chain += clock64() - t0;           // the wave that runs the chain
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (tid >= 64 && tid < 128) total += clock64() - t0;
        __syncthreads();
    }
    if (tid == 64 && blockIdx.x == 0) cyc[0] = total / reps;
    if (tid == 0 && blockIdx.x == 0) cyc[1] = chain / reps;
}

__global__ __launch_bounds__(128) void bench_factor_fine(const double *A, double *Ldiag, int *info, long long *stamps)
{
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Li[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) FactorComm fc;
    const int tid = threadIdx.x;
    if (tid == 0) { fc.prog[0] = 0; fc.prog[1] = 0; }
    for (int r = 0; r < 3; ++r) {
        if (tid < NB) for (int c = 0; c < NB; ++c) M[tid][c] = A[tid * NB + c];
        __syncthreads();
        factor_diag_block<false, true>(&M[0][0], NB + 1, 0, Ldiag, info, fc.ab, fc.l1, fc.prog, &Li[0][0], NB + 1, NB, false, stamps);
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void bench_mfma(double *out, long long *cyc)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = clock64();
    if (lane == 0) cyc[0] = (t1 - t0) / 1024;          // dependent through C
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    t1 = clock64();
    if (lane == 0) cyc[1] = (t1 - t0) / 1024;          // four independent accumulators
    // dependent through the A operand: result register 0 feeds the next product (the panel -> update pattern)
    double x = a;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { v4d t = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b, c1, 0, 0, 0); x = t[0] * 1e-3; }
    }
    t1 = clock64();
    if (lane == 0) cyc[2] = (t1 - t0) / 1024;
    out[lane] = c0[0] + c1[1] + c2[2] + c3[3] + x;
}

int main()
{
    std::vector<double> A(32 * 32);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) A[i * 32 + j] = (i == j ? 3.0 : 0.0) + 1.0 / (1 + i + j);
    double *dA, *dD, *dout; int *dinfo; long long *dc;
    hipMalloc(&dA, 32 * 32 * 8); hipMalloc(&dD, 32 * 32 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dinfo, 16); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), 32 * 32 * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 16);
    long long h[4];
    if (0) for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench_mfma, dim3(1), dim3(64), 0, 0, dout, dc);
    if (0) printf("v_mfma_f64_16x16x4_f64, cycles per instruction: chained through C %lld, four accumulators %lld, result -> A operand (+ one v_mul_f64) %lld\n", h[0], h[1], h[2]);
    // cold: the first execution of the code on this CU (one pass); then warm: 20 passes in one launch
    hipLaunchKernelGGL(bench_factor, dim3(1), dim3(128), 0, 0, dA, dD, dinfo, dc, 1);
    hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
    printf("factor_diag_block (three waves), first pass of a process: %lld cycles until the inverse is stored, %lld until the chain's last panel is out\n", h[0], h[1]);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(bench_factor, dim3(1), dim3(128), 0, 0, dA, dD, dinfo, dc, 1);
        hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
        printf("factor_diag_block (three waves), one pass per launch: %lld / %lld\n", h[0], h[1]);
    }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench_factor, dim3(1), dim3(128), 0, 0, dA, dD, dinfo, dc, 20);
    hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
    printf("factor_diag_block (three waves), 20 passes in one launch: %lld cycles until the inverse is stored, %lld until the chain's last panel is out\n", h[0], h[1]);
    // 256 workgroups at once: every CU runs the code for the first time in this launch (block 0 reports)
    hipLaunchKernelGGL(bench_factor, dim3(256), dim3(128), 0, 0, dA, dD, dinfo, dc, 1);
    hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
    printf("factor_diag_block, 256 workgroups, one pass: %lld / %lld\n", h[0], h[1]);
    {
        long long *ds; hipMalloc(&ds, 128 * 8); hipMemset(ds, 0, 128 * 8);
        hipLaunchKernelGGL(bench_factor_fine, dim3(1), dim3(128), 0, 0, dA, dD, dinfo, ds);
        long long st[128]; hipMemcpy(st, ds, sizeof(st), hipMemcpyDeviceToHost);
        printf("panel: published | pivots done, A operand ready, panel out of the MFMA, update back (cycles since the wave entered)\n");
        for (int p = 0; p < 8; ++p) printf("  %d: %6lld | %6lld %6lld %6lld %6lld | inverse rows stored %6lld\n", p, st[15 + p], st[64 + 4 * p], st[64 + 4 * p + 1], st[64 + 4 * p + 2], st[64 + 4 * p + 3], st[23 + p]);
    }
    int info; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
    printf("info %d\n", info);
    return 0;
}
