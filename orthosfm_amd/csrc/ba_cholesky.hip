// Dense Cholesky solve of the Schur-reduced camera system S y = b (double).
//
// The reference hands this system to CHOLMOD through Ceres' SPARSE_SCHUR
// (bundle_adjustment.cpp:127-128); for a few hundred cameras with 5-6 free
// parameters each S is small (<= ~3000^2) and effectively dense, so it is
// factorised densely on the GPU.
//
// Layout: row-major, leading dimension ld = N = n rounded up to 32; the
// padding diagonal is 1.  The right-hand side rides along as an extra block
// row (row N): the panel/update steps applied to it perform the forward
// substitution for free, leaving y = L^-1 b in that row; a single-workgroup
// kernel then runs the backward substitution x = L^-T y.
//
// Right-looking blocked algorithm, block 32:
//   chol_panel(k):  every workgroup factors A_kk redundantly (32^3/3 flops) and
//                   solves A_ik <- A_ik L_kk^-T for 64 rows of the panel
//   chol_update(k): A_ij -= L_ik L_jk^T for k < j <= i (incl. the rhs row)
#include "ba_kernels.h"

namespace osfm {

constexpr int NB = 32;

__device__ __forceinline__ double readlane_d(double x, int l)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// 1 / sqrt(d) to double precision: v_rsq_f64 seed and two Newton steps.  The
// pivot chain of the factorisation is serial, so its length in dependent
// instructions (a sqrt and a divide are ~70) is what the panel kernel waits for.
__device__ __forceinline__ double rsqrt_newton(double d)
{
    const double h = 0.5 * d;
    double y = __builtin_amdgcn_rsq(d);
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}

// One wave per workgroup.  Every workgroup factors the 32x32 diagonal block
// with its rows held in registers (lane r = row r; the pivot column is
// broadcast with v_readlane; pivots enter as reciprocal square roots),
// workgroup 0 publishes the factor, the others solve X L_kk^T = A_ik for 64 rows
// of the panel (lane = row, X in registers, L broadcast from LDS, stored
// transposed so that two multipliers arrive per ds_read_b128).
__global__ __launch_bounds__(64, 1) void
chol_panel_kernel(double *A, int ld, int nblk, int k, double *Ldiag, int *info)
{
    __shared__ __attribute__((aligned(16))) double LsT[NB][NB];     // LsT[c][m] = L[m][c]
    const int lane = threadIdx.x, r = lane & 31;
    const double *Akk = A + (size_t)(k * NB + r) * ld + k * NB;
    double L[NB], dinv[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) L[c] = Akk[c];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double d = readlane_d(L[j], j);
        if (!(d > 0.0)) { d = 1.0; bad = j + 1; }
        const double rinv = rsqrt_newton(d);
        dinv[j] = rinv;
        const double lj = (r == j && bad == j + 1) ? 1.0 : L[j] * rinv;   // lane j: d / sqrt(d) = sqrt(d)
        L[j] = lj;
#pragma unroll
        for (int c = j + 1; c < NB; ++c) L[c] -= lj * readlane_d(lj, c);   // meaningful for r >= c
    }
    if (blockIdx.x == 0) {
        // Publishes inv(L_kk) for the backward substitution (a 32x32 product there
        // instead of a 32-step chain).  Lane j solves L x = e_j for column j of the
        // inverse: x stays in its own registers, the multipliers L[i][c] arrive as
        // LDS broadcasts -- no cross-lane traffic in the chain.  This workgroup has
        // no panel rows to solve, so the extra work hides behind the others.
        if (bad && lane == 0) atomicMax(info, k * NB + bad);
        if (lane < NB) {
#pragma unroll
            for (int c = 0; c < NB; ++c) LsT[r][c] = c <= r ? L[c] : 0.0;      // row-major here: LsT[i][c] = L[i][c]
        }
        __syncthreads();
        double x[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double acc = (i == r) ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < i; ++c) acc -= LsT[i][c] * x[c];
            x[i] = acc * dinv[i];
        }
        if (lane < NB) {
            double *Lk = Ldiag + (size_t)k * NB * NB;                          // [i][j] = inv(L)[i][j]
#pragma unroll
            for (int i = 0; i < NB; ++i) Lk[i * NB + r] = x[i];
        }
        return;
    }
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) LsT[c][r] = c <= r ? L[c] : 0.0;
    }
    __syncthreads();
    const int row = (k + 1) * NB + (blockIdx.x - 1) * 64 + lane;
    if (row >= (nblk + 1) * NB) return;
    double *Ar = A + (size_t)row * ld + k * NB;
    double X[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) X[c] = Ar[c];
    // right-looking forward substitution: once x_c is final, retire it from
    // the later columns (keeps only X live and bounds the loads in flight)
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        asm volatile("" ::: "memory");    // keep the LDS reads of step c behind step c-1
        const double xc = X[c] * dinv[c];
        X[c] = xc;
#pragma unroll
        for (int m = c + 1; m < NB; ++m) X[m] -= xc * LsT[c][m];
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) Ar[c] = X[c];
}

__global__ __launch_bounds__(256) void
chol_update_kernel(double *A, int ld, int nblk, int k)
{
    // tile (i, j), k < j <= i <= nblk (i == nblk is the rhs block row, j < nblk)
    const int j = k + 1 + blockIdx.x;
    const int i = k + 1 + blockIdx.y;
    if (j > i || j >= nblk) return;
    __shared__ double Li[NB][NB + 1];
    __shared__ double Lj[NB][NB + 1];
    const int tid = threadIdx.x;
    const int r = tid >> 3, c0 = (tid & 7) * 4;
    const double *Aik = A + (size_t)(i * NB) * ld + k * NB;
    const double *Ajk = A + (size_t)(j * NB) * ld + k * NB;
    for (int c = 0; c < 4; ++c) {
        Li[r][c0 + c] = Aik[(size_t)r * ld + c0 + c];
        Lj[r][c0 + c] = Ajk[(size_t)r * ld + c0 + c];
    }
    __syncthreads();
    double *Aij = A + (size_t)(i * NB) * ld + j * NB;
    double acc[4] = { 0, 0, 0, 0 };
    for (int m = 0; m < NB; ++m) {
        const double a = Li[r][m];
        for (int c = 0; c < 4; ++c) acc[c] += a * Lj[c0 + c][m];
    }
    for (int c = 0; c < 4; ++c) Aij[(size_t)r * ld + c0 + c] -= acc[c];
}

// x = L^-T y with y in row N (= nblk * NB) of A; result written to x[0..n).
// Ldiag holds inv(L_kk) of every diagonal block (chol_panel_kernel), so block k
// is x_k = inv(L_kk)^T y_k: 32 independent dot products.  One workgroup: the
// work is a chain of nblk small steps.
// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains the
// vector-memory counter, which here would expose the latency of every prefetch and
// of every result store once per step of the chain.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(1024) void
chol_backsolve_kernel(const double *A, int ld, int nblk, int n, const double *Ldiag, double *x)
{
    extern __shared__ double ys[];           // y [N], then x [N]
    __shared__ double xk[NB];
    __shared__ double Li[NB][NB];            // inv(L_kk), row-major
    const int N = nblk * NB;
    double *y = ys, *xs = ys + N;
    const int tid = threadIdx.x;
    for (int c = tid; c < N; c += blockDim.x) y[c] = A[(size_t)N * ld + c];
    double li_next = Ldiag[(size_t)(nblk - 1) * NB * NB + tid];       // 1024 threads = 32 x 32
    __syncthreads();
    for (int k = nblk - 1; k >= 0; --k) {
        Li[tid >> 5][tid & 31] = li_next;
        if (k > 0) li_next = Ldiag[(size_t)(k - 1) * NB * NB + tid];  // in flight during this step
        lds_barrier();
        if (tid < NB) {
            // x_k[m] = sum_{c >= m} inv(L)[c][m] * y_k[c]
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) acc += (c >= tid) ? Li[c][tid] * y[k * NB + c] : 0.0;
            xk[tid] = acc;
            xs[k * NB + tid] = acc;
        }
        lds_barrier();
        // y_j -= L[k-block rows][j]^T x_k for every column j left of the block
        const double *Lrow = A + (size_t)(k * NB) * ld;
        for (unsigned c = tid; c < (unsigned)(k * NB); c += blockDim.x) {
            // all 32 loads in flight at once: scalar row base + one shared vector offset
            // (left to itself the compiler issues load, wait, fma, load, ... -- 32 memory
            // latencies in a row per step of the chain)
            double l[NB];
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                const double *rowp = Lrow + (size_t)m * ld;       // uniform
                l[m] = rowp[c];
            }
            __builtin_amdgcn_sched_barrier(0);
            double v = y[c];
#pragma unroll
            for (int m = 0; m < NB; ++m) v -= l[m] * xk[m];
            y[c] = v;
        }
        lds_barrier();
    }
    for (int c = tid; c < n; c += blockDim.x) x[c] = xs[c];
}

int cholesky_padded_dim(int n) { return (n + NB - 1) / NB * NB; }

// A: (N + 32) x N row-major, rows/cols >= n padded with identity, rhs in row N.
// Ldiag: N * 32 doubles of scratch for the inverses of the factored diagonal blocks.
void launch_cholesky_solve(double *A, int n, double *Ldiag, double *x, int *info, hipStream_t s)
{
    const int N = cholesky_padded_dim(n);
    const int nblk = N / NB;
    for (int k = 0; k < nblk; ++k) {
        const int below = (nblk - k) * NB;      // rows under the diagonal block (incl. the rhs block row)
        hipLaunchKernelGGL(chol_panel_kernel, dim3(1 + (below + 63) / 64), dim3(64), 0, s, A, N, nblk, k, Ldiag, info);
        const int t = nblk - k - 1;
        if (t > 0)
            hipLaunchKernelGGL(chol_update_kernel, dim3(t, t + 1), dim3(256), 0, s, A, N, nblk, k);
    }
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(1024), (size_t)2 * N * sizeof(double), s, A, N,
        nblk, n, Ldiag, x);
}

}  // namespace osfm
