// The Schur complement's point part as a dense product, for problems whose tracks are seen by a large share of the
// cameras.
//
// The pair pass walks, for every camera pair, the tracks both cameras see: E = sum over tracks of len (len + 1) / 2
// entries, each a gather of two 144-byte records and 108 multiply-adds -- bound by the gathers, ~60 ps per entry
// (DESIGN.md 3.3).  With Z_a = Jc_a^T Q_a and W_a = Jc_a^T Jp_a (n_c x 3 per observation) an entry is Z_a W_b^T, so
//     S -= Zm Wm^T,   Zm, Wm: nc x 3M, row = camera unknown, column = track coordinate, zero where unobserved,
// and when every track is seen by 40 % of the cameras (the global adjustments of the end-to-end jobs: 50 M entries at
// 500 cameras, 3 ms per pass) the zeros cost less than the gathers: nc^2 x 3M / 2 multiply-adds for the lower half
// on v_mfma_f64_16x16x4_f64.  ba_solve_core takes this path when its cost model says so (schur_dense_wins); the pair
// pass then only does the diagonal pairs (U, the reduced right-hand side, the gradient) and ADDS them.
//
//   ba_dense_build_kernel    a lane per observation: its Z and W blocks from its record into the two matrices
//   ba_dense_gemm_kernel     128 x 128 tiles of the lower triangle, a K range each (split-K for small systems), four
//                            waves of 64 x 64, A / B tiles of 16 columns through LDS, the next tile's loads in flight
//   ba_dense_reduce_kernel   S = -(sum of the K ranges' partials), fixed order (split-K only)
#include <algorithm>

#include <cstdlib>

#include "ba_kernels.h"
#include "osfm_common.h"

namespace osfm {

typedef double dense_v4d __attribute__((ext_vector_type(4)));

constexpr int kDenseTile = 128;      // rows / columns of S per workgroup
constexpr int kDenseK = 16;          // columns of Zm / Wm per step
constexpr int kDenseLd = kDenseK + 1;

int schur_dense_rows(int nc) { return (nc + kDenseTile - 1) / kDenseTile * kDenseTile; }
int schur_dense_cols(int M) { return (3 * M + kDenseK - 1) / kDenseK * kDenseK; }

// Cost model (measured rates, MI355X): the entry lists at 60 ps per entry, the product at ~30 TFLOP/s of the f64
// matrix cores' 78 plus what building and clearing the matrices costs.
bool schur_dense_wins(int nc, int M, int64_t entries)
{
    if (nc < 64 || M <= 0) return false;
    const double rows = schur_dense_rows(nc), cols = schur_dense_cols(M);
    if (2.0 * rows * cols * 8.0 > 16e9) return false;          // the two matrices: at most 16 GB of the card
    const double t_sparse = 60e-12 * (double)entries;
    const double t_dense = rows * rows * cols / 30e12 + 2.0 * rows * cols * 8.0 / 3e12 + 60e-6;
    return t_dense < 0.8 * t_sparse;
}

__global__ __launch_bounds__(256) void
ba_dense_build_kernel(BaDev d, const double *obsrec, const int32_t *obs_lay, double *Zm, double *Wm, int ldk)
{
    if (!lm_resolve(d)) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d.O) return;
    const int lay = obs_lay[k], off = lay & 0xffffff, n = lay >> 24;
    if (n == 0) return;
    double r[kRecR];
    const double2 *src = reinterpret_cast<const double2 *>(obsrec + (size_t)k * kObsRec);
#pragma unroll
    for (int i = 0; i < kRecR / 2; ++i) { const double2 v = src[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
    const size_t col = (size_t)3 * d.obs_pt[k];
#pragma unroll
    for (int x = 0; x < 6; ++x) {
        if (x >= n) continue;
        const double j0 = r[kRecJc + x], j1 = r[kRecJc + 6 + x];
        double *z = Zm + (size_t)(off + x) * ldk + col, *w = Wm + (size_t)(off + x) * ldk + col;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            z[t] = j0 * r[kRecQ + t] + j1 * r[kRecQ + 3 + t];
            w[t] = j0 * r[kRecJp + t] + j1 * r[kRecJp + 3 + t];
        }
    }
}

// Tile (ti, tj), tj <= ti, K range `split` of `splits`: P = Zm[ti rows] Wm[tj rows]^T over that range.
// splits == 1: S = -P written in place (rows / columns < nc); else P into partial[split] (a dense rows x rows matrix).
__global__ __launch_bounds__(256, 2) void
ba_dense_gemm_kernel(const double *Zm, const double *Wm, int ldk, int ksteps_total, int splits, int ntile, double *S, int ldS, int nc,
    double *partial, int rows, const LmDev *lm)
{
    if (lm && (lm->stop || lm->flow_aborted)) return;
    __shared__ __attribute__((aligned(16))) double As[kDenseTile][kDenseLd];
    __shared__ __attribute__((aligned(16))) double Bs[kDenseTile][kDenseLd];
    // block -> (tile of the lower triangle, split)
    const int tile = blockIdx.x / splits, split = blockIdx.x - tile * splits;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    const int tj = tile - ti * (ti + 1) / 2;
    (void)ntile;
    const int ks0 = (int)((int64_t)ksteps_total * split / splits), ks1 = (int)((int64_t)ksteps_total * (split + 1) / splits);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;                 // the wave's 64 x 64 quadrant
    const int ar = lane & 15, ak = lane >> 4;
    // loads of a step: thread t brings columns 8 (t & 1) .. + 7 of row t / 2 of both tiles
    const int lrow = tid >> 1, lcol = (tid & 1) * 8;
    const double *ga = Zm + (size_t)(ti * kDenseTile + lrow) * ldk + lcol;
    const double *gb = Wm + (size_t)(tj * kDenseTile + lrow) * ldk + lcol;
    dense_v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = dense_v4d{0.0, 0.0, 0.0, 0.0};
    double2 pa[4], pb[4];
    auto fetch = [&](int ks) {
        const double2 *a2 = reinterpret_cast<const double2 *>(ga + (size_t)ks * kDenseK);
        const double2 *b2 = reinterpret_cast<const double2 *>(gb + (size_t)ks * kDenseK);
#pragma unroll
        for (int q = 0; q < 4; ++q) { pa[q] = a2[q]; pb[q] = b2[q]; }
    };
    auto put = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            As[lrow][lcol + 2 * q] = pa[q].x; As[lrow][lcol + 2 * q + 1] = pa[q].y;
            Bs[lrow][lcol + 2 * q] = pb[q].x; Bs[lrow][lcol + 2 * q + 1] = pb[q].y;
        }
    };
    if (ks0 < ks1) { fetch(ks0); put(); }
    __syncthreads();
    for (int ks = ks0; ks < ks1; ++ks) {
        // the next step's tiles are on their way (into registers) while this one's products run
        if (ks + 1 < ks1) fetch(ks + 1);
#pragma unroll
        for (int s4 = 0; s4 < kDenseK / 4; ++s4) {
            double a[4], b[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                a[f] = As[64 * wi + 16 * f + ar][4 * s4 + ak];
                b[f] = Bs[64 * wj + 16 * f + ar][4 * s4 + ak];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (ks + 1 < ks1) put();
        __syncthreads();
    }
    // element e of fragment (i, j): row 64 wi + 16 i + (lane >> 4) + 4 e, column 64 wj + 16 j + (lane & 15)
    const int r0 = ti * kDenseTile + 64 * wi + (lane >> 4), c0 = tj * kDenseTile + 64 * wj + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r0 + 16 * i + 4 * e, col = c0 + 16 * j;
                if (splits == 1) {
                    if (row < nc && col < nc) S[(size_t)row * ldS + col] = -acc[i][j][e];
                } else {
                    partial[((size_t)split * rows + row) * rows + col] = acc[i][j][e];
                }
            }
}

__global__ __launch_bounds__(256) void
ba_dense_reduce_kernel(const double *partial, int splits, int rows, double *S, int ldS, int nc, const LmDev *lm)
{
    if (lm && (lm->stop || lm->flow_aborted)) return;
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    // the tiles of the lower triangle only: the others were never written
    if (row >= nc || col >= nc || col / kDenseTile > row / kDenseTile) return;
    double v = 0.0;
    for (int s = 0; s < splits; ++s) v += partial[((size_t)s * rows + row) * rows + col];
    S[(size_t)row * ldS + col] = -v;
}

int schur_dense_splits(int nc, int M)
{
    const int nt = schur_dense_rows(nc) / kDenseTile, tiles = nt * (nt + 1) / 2;
    const int ksteps = schur_dense_cols(M) / kDenseK;
    int splits = std::max(1, std::min(16, (512 + tiles - 1) / tiles));
    splits = std::min(splits, std::max(1, ksteps / 8));
    // 192 .. 511 tiles (the 500-view job's last adjustments: 210): as one workgroup per CU nothing covers a workgroup's
    // barriers and its waits for the next tiles; split in two over K, two workgroups share a CU (OSFM_BA_DENSE_SPLIT2=0: off)
    static const bool split2 = !(getenv("OSFM_BA_DENSE_SPLIT2") && atoi(getenv("OSFM_BA_DENSE_SPLIT2")) == 0);
    if (tiles >= 192) return (split2 && tiles < 512 && ksteps >= 64) ? 2 : 1;
    return splits;
}

size_t schur_dense_partial_bytes(int nc, int M)
{
    const int splits = schur_dense_splits(nc, M);
    const size_t rows = schur_dense_rows(nc);
    return splits > 1 ? (size_t)splits * rows * rows * 8 : 16;
}

// Zm / Wm: schur_dense_rows(nc) x schur_dense_cols(M) doubles each, cleared (first: the first call of a solve) and filled here; S gets -Zm Wm^T in the
// 128 x 128 tiles of its lower triangle (rows / columns < nc)
void launch_schur_dense(const BaDev &d, const double *obsrec, const int32_t *obs_lay, double *Zm, double *Wm, double *partial,
    double *S, int ldS, bool first, hipStream_t s)
{
    const int nc = d.nc, rows = schur_dense_rows(nc), ldk = schur_dense_cols(d.M);
    // the blocks land at the same places in every linearisation of a solve: the zeros around them are written once
    if (first) {
        (void)hipMemsetAsync(Zm, 0, (size_t)rows * ldk * 8, s);
        (void)hipMemsetAsync(Wm, 0, (size_t)rows * ldk * 8, s);
    }
    if (d.O > 0) hipLaunchKernelGGL(ba_dense_build_kernel, dim3((d.O + 255) / 256), dim3(256), 0, s, d, obsrec, obs_lay, Zm, Wm, ldk);
    const int nt = rows / kDenseTile, tiles = nt * (nt + 1) / 2, splits = schur_dense_splits(nc, d.M);
    hipLaunchKernelGGL(ba_dense_gemm_kernel, dim3(tiles * splits), dim3(256), 0, s, Zm, Wm, ldk, ldk / kDenseK, splits, tiles, S, ldS, nc,
        partial, rows, d.lm);
    if (splits > 1)
        hipLaunchKernelGGL(ba_dense_reduce_kernel, dim3((nc + 255) / 256, nc), dim3(256), 0, s, partial, splits, rows, S, ldS, nc, d.lm);
}

}  // namespace osfm
