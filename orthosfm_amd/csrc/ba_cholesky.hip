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
//   chol_panel(k):  every row block i >= k factors A_kk in LDS (redundantly,
//                   32^3/3 flops) and solves its A_ik <- A_ik L_kk^-T
//   chol_update(k): A_ij -= L_ik L_jk^T for k < j <= i (incl. the rhs row)
#include "ba_kernels.h"

namespace osfm {

constexpr int NB = 32;

__global__ __launch_bounds__(256) void
chol_panel_kernel(double *A, int ld, int nblk, int k, double *Ldiag, int *info)
{
    __shared__ double L[NB][NB + 1];
    __shared__ double X[NB][NB + 1];
    const int i = k + blockIdx.x;          // row block (nblk = the rhs block row)
    const int tid = threadIdx.x;
    const int r = tid >> 3, c0 = (tid & 7) * 4;   // 32 rows x 8 threads x 4 columns
    double *Akk = A + (size_t)(k * NB) * ld + k * NB;
    for (int c = 0; c < 4; ++c) L[r][c0 + c] = Akk[(size_t)r * ld + c0 + c];
    __syncthreads();
    // factor the diagonal block in LDS (lower triangle)
    for (int j = 0; j < NB; ++j) {
        if (tid == 0) {
            const double d = L[j][j];
            if (!(d > 0.0)) { L[j][j] = 1.0; if (i == k) atomicMax(info, k * NB + j + 1); }
            else L[j][j] = sqrt(d);
        }
        __syncthreads();
        if (tid > j && tid < NB) L[tid][j] /= L[j][j];
        __syncthreads();
        // trailing update of the lower triangle: rows > j, cols in (j, row]
        for (int e = tid; e < NB * NB; e += 256) {
            const int rr = e / NB, cc = e % NB;
            if (rr > j && cc > j && cc <= rr) L[rr][cc] -= L[rr][j] * L[cc][j];
        }
        __syncthreads();
    }
    if (i == k) {
        // the factored diagonal block goes to a side buffer: A_kk itself is
        // still being read by the other workgroups of this launch
        double *Lk = Ldiag + (size_t)k * NB * NB;
        for (int c = 0; c < 4; ++c)
            Lk[r * NB + c0 + c] = (c0 + c <= r) ? L[r][c0 + c] : 0.0;
        return;
    }
    // panel solve: X L_kk^T = A_ik  (one thread per row of the block)
    double *Aik = A + (size_t)(i * NB) * ld + k * NB;
    for (int c = 0; c < 4; ++c) X[r][c0 + c] = Aik[(size_t)r * ld + c0 + c];
    __syncthreads();
    if (tid < NB) {
        for (int c = 0; c < NB; ++c) {
            double v = X[tid][c];
            for (int m = 0; m < c; ++m) v -= X[tid][m] * L[c][m];
            X[tid][c] = v / L[c][c];
        }
    }
    __syncthreads();
    for (int c = 0; c < 4; ++c) Aik[(size_t)r * ld + c0 + c] = X[r][c0 + c];
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

// x = L^-T y with y in row N (= nblk * NB) of A; result written to x[0..n)
__global__ __launch_bounds__(1024) void
chol_backsolve_kernel(const double *A, int ld, int nblk, int n, const double *Ldiag, double *x)
{
    extern __shared__ double y[];            // [nblk * NB]
    __shared__ double xk[NB];
    const int N = nblk * NB;
    const int tid = threadIdx.x;
    for (int c = tid; c < N; c += blockDim.x) y[c] = A[(size_t)N * ld + c];
    __syncthreads();
    for (int k = nblk - 1; k >= 0; --k) {
        const double *Lkk = Ldiag + (size_t)k * NB * NB;
        // L_kk^T x_k = y_k, backward substitution by one thread (32 unknowns)
        if (tid == 0) {
            for (int c = NB - 1; c >= 0; --c) {
                double v = y[k * NB + c];
                for (int m = c + 1; m < NB; ++m) v -= Lkk[m * NB + c] * xk[m];
                xk[c] = v / Lkk[c * NB + c];
            }
        }
        __syncthreads();
        // y_j -= L[k-block rows][j]^T x_k for every column j left of the block
        const double *Lk = A + (size_t)(k * NB) * ld;
        for (int c = tid; c < k * NB; c += blockDim.x) {
            double v = y[c];
            for (int m = 0; m < NB; ++m) v -= Lk[(size_t)m * ld + c] * xk[m];
            y[c] = v;
        }
        if (tid < NB && k * NB + tid < n) x[k * NB + tid] = xk[tid];
        __syncthreads();
    }
}

int cholesky_padded_dim(int n) { return (n + NB - 1) / NB * NB; }

// A: (N + 32) x N row-major, rows/cols >= n padded with identity, rhs in row N.
// Ldiag: N * 32 doubles of scratch for the factored diagonal blocks.
void launch_cholesky_solve(double *A, int n, double *Ldiag, double *x, int *info, hipStream_t s)
{
    const int N = cholesky_padded_dim(n);
    const int nblk = N / NB;
    for (int k = 0; k < nblk; ++k) {
        const int rows = nblk - k + 1;          // row blocks k..nblk (incl. rhs row)
        hipLaunchKernelGGL(chol_panel_kernel, dim3(rows), dim3(256), 0, s, A, N, nblk, k, Ldiag, info);
        const int t = nblk - k - 1;
        if (t > 0)
            hipLaunchKernelGGL(chol_update_kernel, dim3(t, t + 1), dim3(256), 0, s, A, N, nblk, k);
    }
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(1024), (size_t)N * sizeof(double), s, A, N,
        nblk, n, Ldiag, x);
}

}  // namespace osfm
