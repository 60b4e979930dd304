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

// Factor of one 32 x 32 diagonal block by ONE wave, then inv(L_kk) for the neighbours'
// triangular solves and the backward substitution.  The pivots form a serial chain, so
// what counts is the latency of one link and the instructions hanging off it.  Lane r (and
// its twin r + 32, which does the same work: no divergent code anywhere) holds row r in
// registers.  Link j: the pivot comes by v_readlane, enters as a reciprocal square root
// (v_rsq_f64 + two Newton steps; a sqrt and a divide are ~70 dependent instructions), the
// scaled column goes to LDS once and comes back as broadcast ds_read_b128s, two
// multipliers each (one v_readlane pair per multiplier was 2.5x the instructions).
// Msrc: the block, row-major with leading dimension lds_ld, in LDS.  All 64 lanes of one
// wave call this; workgroup barriers around it are the caller's.
__device__ __forceinline__ double swap_halves(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false);
    // lanes 0-31 receive what lanes 32-63 held and vice versa
    const int lane_hi = (int)(threadIdx.x & 63) >> 5;
    return __hiloint2double(lane_hi ? (int)b[0] : (int)b[1], lane_hi ? (int)a[0] : (int)a[1]);
}

__device__ __forceinline__ void
factor_diag_block(const double *Msrc, int lds_ld, double (*LsT)[NB], int kblock, double *Ldiag, int *info,
    double *linv_lds = nullptr, int linv_ld = 0, int nvalid = NB)
{
    // nvalid: rows / columns from there on are identity padding (wave-uniform); their pivot
    // steps and inverse rows change nothing and are skipped -- the chain is serial, so a
    // 17-unknown system (three cameras) is done in half the time of a full block
    __shared__ __attribute__((aligned(16))) double colbuf[2][2 * NB];  // [pivot parity][lane]: column j of L in [0, NB)
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    // Lanes 0..31: row r of the block, turning into row r of L.  Lanes 32..63: e_r, turning
    // into column r of inv(L) (L x = e_r, column-oriented: once x[j] is final every later
    // entry takes L[i][j] x[j]).  Both are the SAME instructions -- v[j] *= 1/L[j][j], then
    // v[c] -= v[j] * L[c][j] for c > j -- so the inverse costs nothing beyond the lanes that
    // used to mirror the factorisation, and its chain hangs off the pivot chain instead of
    // forming a second one of 32 links behind it.
    double v[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) v[c] = h == 0 ? Msrc[r * lds_ld + c] : (c == r ? 1.0 : 0.0);
    int bad = 0;
    // Pivot j: only column j + 1 has to be final before pivot j + 1 can start, so that
    // column takes its multiplier L[j+1][j] by v_readlane right away and the next pivot's
    // reciprocal square root (the longest link of the chain) is started at once; the other
    // columns take their multipliers from LDS (broadcast ds_read_b128, two each) -- a round
    // trip of > 100 cycles -- with the reads issued in front of that chain and consumed
    // behind it.
    double d = nvalid > 0 ? readlane_d(v[0], 0) : 1.0;
    if (!(d > 0.0)) { d = 1.0; bad = 1; }
    double rinv = rsqrt_newton(d);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j >= nvalid) continue;          // identity rows: nothing to eliminate, x[j] stays e_r[j]
        const double vj = (lane == j && bad == j + 1) ? 1.0 : v[j] * rinv;   // lane j: d / sqrt(d) = sqrt(d)
        v[j] = vj;
        if (j + 1 < NB) {
            colbuf[j & 1][lane] = vj;           // lanes 0..31: column j of L (the rest is not read)
            // The LDS executes one wave's operations in order, so no wait is needed -- but the
            // COMPILER must be told that other lanes wrote what this lane is about to read
            // (without the fence it re-used values a lane had loaded two pivots earlier)
            asm volatile("" ::: "memory");
            double col[NB];
#pragma unroll
            for (int c = j + 2; c < NB; ++c) col[c] = colbuf[j & 1][c];
            __builtin_amdgcn_sched_barrier(0);
            v[j + 1] = fma(-vj, readlane_d(vj, j + 1), v[j + 1]);
            if (j + 1 < nvalid) {
                d = readlane_d(v[j + 1], j + 1);
                if (!(d > 0.0)) { d = 1.0; bad = j + 2; }
                rinv = rsqrt_newton(d);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = j + 2; c < NB; ++c) v[c] = fma(-vj, col[c], v[c]);    // L: meaningful for r >= c
        }
    }
    if (bad && lane == 0) atomicMax(info, kblock * NB + bad);
    // L itself, transposed (LsT[c][i] = L[i][c], zero above the diagonal), for a caller that
    // wants to look at it: lane r writes element r of every row, consecutive addresses
    if (LsT && h == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) LsT[c][r] = c <= r ? v[c] : 0.0;
    }
    if (h == 1) {
        double *Lk = Ldiag + (size_t)kblock * NB * NB;                          // [i][j] = inv(L)[i][j]
#pragma unroll
        for (int i = 0; i < NB; ++i) Lk[i * NB + r] = v[i];
        // a copy in LDS for a caller that goes on to use it (Msrc itself may be the target:
        // the block was read into registers at the top)
        if (linv_lds) {
#pragma unroll
            for (int i = 0; i < NB; ++i) linv_lds[i * linv_ld + r] = v[i];
        }
    }
}

// Block 0: nothing to update, just the factor.
__global__ __launch_bounds__(64, 1) void
chol_first_kernel(const double *A, int ld, double *Ldiag, int *info, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed)) return;
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    const int lane = threadIdx.x, r = lane & 31;
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) M[r][c] = A[(size_t)r * ld + c];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    factor_diag_block(&M[0][0], NB + 1, nullptr, 0, Ldiag, info);
}

// A system of one block (n <= 32: the three-camera adjustments of the incremental
// reconstruction) start to finish in one launch of one wave: factor, y = inv(L) b,
// x = inv(L)^T y, and the candidate cameras Plus(x, -step) that the next kernel needs --
// four launches of a latency-bound chain in one.
__global__ __launch_bounds__(64, 1) void
chol_small_kernel(const double *A, int ld, int n, double *Ldiag, double *x, int *info, BaDev d,
    double *partials_cam)
{
    if (!lm_resolve(d)) return;
    if (d.lm->lin_failed) return;
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    __shared__ double ys[NB], xs[NB];
    const int lane = threadIdx.x, r = lane & 31;
    double b = 0.0;
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) M[r][c] = A[(size_t)r * ld + c];
        b = A[(size_t)NB * ld + r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    factor_diag_block(&M[0][0], NB + 1, nullptr, 0, Ldiag, info, &M[0][0], NB + 1, n);  // M := inv(L)
    if (lane < NB) ys[lane] = b;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double acc = 0.0;
    if (lane < NB) {
        // y_c = sum_{m <= c} b[m] inv(L)[c][m]
#pragma unroll
        for (int m = 0; m < NB; ++m) acc = fma(ys[m], m <= lane ? M[lane][m] : 0.0, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < NB) ys[lane] = acc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < NB) {
        // x_m = sum_{c >= m} inv(L)[c][m] y[c]
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) v += (c >= lane) ? M[c][lane] * ys[c] : 0.0;
        if (lane < n) x[lane] = v;          // the back pass reads it from global memory
        xs[lane] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // a failed factorisation is noticed by ba_lm_decide (chol_info); the candidates it then
    // ignores are written all the same, like the general path does
    double *cams_out = d.cams2[d.lm->cur ^ 1];
    for (int c = lane; c < d.C; c += 64) cam_update_one(d, xs, cams_out, partials_cam, c);
}

// Step k of the right-looking factorisation as ONE launch: tile (i, j), k < j <= i <= nblk
// (i == nblk: the right-hand side riding along), first turns its two panel blocks into L
// itself -- L_ik = A_ik inv(L_kk)^T, a 32^3 product with the inverse the previous step left
// in Ldiag, instead of waiting for a panel kernel to solve them -- then subtracts
// L_ik L_jk^T.  The tiles of column k + 1 store their L_ik (the finished panel, into Lout: the
// factor and the solved right-hand side live in a matrix of their own); tile
// (k+1, k+1) goes on to factor itself and publish inv(L_{k+1,k+1}), so the next step can
// start as soon as this launch ends: one kernel boundary per block column instead of two.
__global__ __launch_bounds__(256) void
chol_step_kernel(double *A, double *Lout, int ld, int nblk, int k, double *Ldiag, int *info, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed)) return;
    const int j = k + 1 + blockIdx.x;
    const int i = k + 1 + blockIdx.y;
    if (j > i || j >= nblk) return;
    __shared__ __attribute__((aligned(16))) double Li[NB][NB + 1];      // inv(L_kk)
    __shared__ __attribute__((aligned(16))) double Ai[NB][NB + 1];      // A_ik, then the updated diagonal tile
    __shared__ __attribute__((aligned(16))) double Aj[NB][NB + 1];      // A_jk
    __shared__ __attribute__((aligned(16))) double Xi[NB][NB + 1];      // L_ik
    __shared__ __attribute__((aligned(16))) double Xj[NB][NB + 1];      // L_jk
    const int tid = threadIdx.x;
    const int r = tid >> 3, c0 = (tid & 7) * 4;
    const double *Lk = Ldiag + (size_t)k * NB * NB;
    double *Aik = A + (size_t)(i * NB) * ld + k * NB;
    const double *Ajk = A + (size_t)(j * NB) * ld + k * NB;
    double *Aij = A + (size_t)(i * NB) * ld + j * NB;
    double cur[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Li[r][c0 + c] = Lk[r * NB + c0 + c];
        Ai[r][c0 + c] = Aik[(size_t)r * ld + c0 + c];
        Aj[r][c0 + c] = Ajk[(size_t)r * ld + c0 + c];
        cur[c] = Aij[(size_t)r * ld + c0 + c];
    }
    __syncthreads();
    // L_ik[r][c] = sum_{m <= c} A_ik[r][m] inv(L)[c][m]   (inv(L) is lower triangular)
    double xi[4] = {0, 0, 0, 0}, xj[4] = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const double ai = Ai[r][m], aj = Aj[r][m];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double l = Li[c0 + c][m];         // zero above the diagonal
            xi[c] = fma(ai, l, xi[c]);
            xj[c] = fma(aj, l, xj[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { Xi[r][c0 + c] = xi[c]; Xj[r][c0 + c] = xj[c]; }
    if (j == k + 1) {
        // the finished panel block goes to the factor's own matrix: A_ik itself is still being
        // read (unsolved) by the other tiles of row i in this launch
        double *Lik = Lout + (size_t)(i * NB) * ld + k * NB;
#pragma unroll
        for (int c = 0; c < 4; ++c) Lik[(size_t)r * ld + c0 + c] = xi[c];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const double a = Xi[r][m];
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = fma(-a, Xj[c0 + c][m], cur[c]);
    }
    const bool next_diag = i == k + 1 && j == k + 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Aij[(size_t)r * ld + c0 + c] = cur[c];
        if (next_diag) Ai[r][c0 + c] = cur[c];
    }
    if (!next_diag) return;
    __syncthreads();
    if (tid < 64) factor_diag_block(&Ai[0][0], NB + 1, nullptr, k + 1, Ldiag, info);
}

// The right-hand side's last block: y_k = b_k inv(L_kk)^T for k = nblk - 1 (every earlier
// block of the row is solved by the step that finishes its panel column).
__global__ __launch_bounds__(64) void
chol_rhs_tail_kernel(const double *A, double *Lout, int ld, int nblk, const double *Ldiag, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed)) return;
    const int k = nblk - 1, c = threadIdx.x;
    if (c >= NB) return;
    const double *y = A + (size_t)(nblk * NB) * ld + k * NB;
    const double *Lk = Ldiag + (size_t)k * NB * NB;
    double acc = 0.0;
    double b[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) b[m] = y[m];
#pragma unroll
    for (int m = 0; m < NB; ++m) acc = fma(b[m], m <= c ? Lk[c * NB + m] : 0.0, acc);
    Lout[(size_t)(nblk * NB) * ld + k * NB + c] = acc;
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
chol_backsolve_kernel(const double *A, int ld, int nblk, int n, const double *Ldiag, double *x, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed)) return;
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

void launch_small_solve(const double *A, int n, double *Ldiag, double *x, int *info, const BaDev &d,
    double *partials_cam, hipStream_t s)
{
    hipLaunchKernelGGL(chol_small_kernel, dim3(1), dim3(64), 0, s, A, NB, n, Ldiag, x, info, d, partials_cam);
}

// A: (N + 32) x N row-major, rows/cols >= n padded with identity, rhs in row N.
// Ldiag: N * 32 doubles of scratch for the inverses of the factored diagonal blocks.
void launch_cholesky_solve(double *A, double *Lmat, int n, double *Ldiag, double *x, int *info, const LmDev *lm, hipStream_t s)
{
    const int N = cholesky_padded_dim(n);
    const int nblk = N / NB;
    hipLaunchKernelGGL(chol_first_kernel, dim3(1), dim3(64), 0, s, A, N, Ldiag, info, lm);
    for (int k = 0; k < nblk; ++k) {
        // tiles (i, j), k < j <= i <= nblk, j < nblk: for k = nblk - 1 only the right-hand-side
        // row is left, and it has no tile with j < nblk: its forward substitution is the
        // triangular solve of the rhs block against every column, done by the steps before
        const int t = nblk - k - 1;
        if (t > 0)
            hipLaunchKernelGGL(chol_step_kernel, dim3(t, t + 1), dim3(256), 0, s, A, Lmat, N, nblk, k, Ldiag, info, lm);
    }
    hipLaunchKernelGGL(chol_rhs_tail_kernel, dim3(1), dim3(64), 0, s, A, Lmat, N, nblk, Ldiag, lm);
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(1024), (size_t)2 * N * sizeof(double), s, Lmat, N,
        nblk, n, Ldiag, x, lm);
}

}  // namespace osfm
