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
#include <algorithm>
#include <atomic>
#include <mutex>

#include "ba_kernels.h"

namespace osfm {

constexpr int NB = 32;
int cholesky_padded_dim(int n);

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
    // One third-order step instead of two Newton steps: with e = 1 - d y^2, y (1 + e / 2 + 3 e^2 / 8) leaves
    // an error of 5/16 e^3.  v_rsq_f64 is good to 2^-24 (tools/micro/rsq_precision.hip: 5.2e-8; one Newton
    // step 4.2e-15, two 1.4e-16), so the cubic step lands within an ulp as the two Newton steps did -- in a
    // chain of four dependent operations (t, e, {p, r}, result) instead of six.
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y), y, 1.0);
    const double p = fma(0.375, e, 0.5);
    const double r = y * e;
    return fma(r, p, y);
}

// 1 / d to double precision: v_rcp_f64 seed and two Newton steps (four dependent operations)
__device__ __forceinline__ double rcp_newton(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains the
// vector-memory counter, which would expose the latency of every prefetch and of every
// result store (2-3 us for a write-through store) once per step of a chain.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (store_sc1 / load_sc1: ba_device.h)

typedef double v4d __attribute__((ext_vector_type(4)));

// Factor of one 32 x 32 diagonal block as eight rank-4 panels on the f64 matrix cores, by TWO waves, with inv(L)
// -- what the neighbours' triangular solves and the backward substitution multiply by -- riding along.
//
// Round 3's form kept a row per lane of one wave and ran 32 scalar pivots, each with ~15 trailing FMAs and eight
// LDS reads hanging off it: 410 cycles per pivot for a dependency chain of ~106 (a wave issues in order).  Here the
// trailing matrix lives in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane = (c, g) = (lane & 15,
// lane >> 4), register e: element (g + 4 e, c) of a 16 x 16 quadrant), three quadrants C00, C01, C11 -- the
// matrix is symmetric, and that is what makes the scheme cheap: rows 4p .. 4p+3 of the quadrants are register
// p & 3 of every lane, and read as COLUMNS 4p .. 4p+3 they are the panel in the A/B OPERAND layout of the same
// instruction (lane (c, g) holds X[c][g]) without moving a byte.  Per panel:
//   1. the 4 x 4 diagonal block comes out by v_readlane (10 values) and is factored, and inverted, uniformly in
//      every lane -- the only serial part.  Its four pivots are taken as two PAIRS: with det = a00 a11 - a10^2
//      the second pivot of a pair is det / a00, so 1 / sqrt(d1) = sqrt(a00) * rsqrt(det) and the two reciprocal
//      square roots (v_rsq_f64 + a third-order step: five dependent operations each) run side by side instead
//      of one behind the other; det carries the rounding error a11 - l10^2 would (eps a10^2 in both).  18
//      dependent operations per panel instead of 36;
//   2. W = inv(L_pp), padded to 16 x 4, is the A operand of one MFMA per 16 panel rows whose B operand is the raw
//      panel: register 0 of the result is the finished panel L[:, 4p .. 4p+3], again in operand layout (TRSM as a product);
//   3. trailing update C -= L_p L_p^T: one MFMA per live quadrant with K = 4, the panel width.
// The inverse: E starts as the identity, is kept TRANSPOSED in the same layout (quadrants T00, T10, T11; E^T so
// that its panel, too, is a register in operand layout) and takes the same two steps -- Y_p = E_p W^T is rows
// 4p .. 4p+3 of inv(L), final at once and stored from there; E^T -= L_p Y_p^T.  Exact zeros stay exact (the
// upper triangle of inv(L) is read by the tile products as zeros).
// Why two waves: on gfx950 one v_mfma_f64_16x16x4_f64 occupies its SIMD's matrix pipe for ~64 cycles and returns
// after ~95 (tools/micro/factor_bench.hip), and a wave issues in order -- with the inverse in the same wave the
// 48 products of a block took 9.4k cycles of which the chain is 6.0k.  So wave 0 runs the chain -- pivots, the
// panel, the updates of A -- and publishes (W, L_p) per panel through LDS; wave 1, on another SIMD with a matrix
// pipe of its own, does the inverse and all the stores.  (A third wave for the rows 16 .. 31 of the first four
// panels, taking the chain over at panel 4, was measured: the hand-over costs more than the three products per
// panel it takes out of wave 0's stream, which fit behind the update the next panel waits for.)
// A pivot that is not positive raises info (kblock * 32 + the first row of its panel + 1, the largest such) and
// leaves NaNs behind: whoever reads info discards the factor.
// Msrc: the block (lower triangle valid), row-major with leading dimension lds_ld, in LDS.  The first two waves
// of the workgroup call this (128 threads); workgroup barriers around it are the caller's.  comm_ab (8 KB), comm_l1
// (2 KB): LDS scratch; prog: one LDS word, zero on entry, zero again on return.  nvalid: rows / columns from there
// on are identity padding (workgroup-uniform): their panels are skipped.  Msrc may be linv_lds: wave 0 has the
// block in registers before the first panel is published, and the inverse is stored behind that.
//
// The progress word lives in LDS; the pointer arrives as a generic one (through a struct, in the flow kernel), and a
// volatile access through a generic pointer is a flat instruction followed by s_waitcnt vmcnt(0) -- on the chain
// that waited for every global store in flight.  Hence the explicit address space.
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ void factor_post(int *word, int value)
{
    asm volatile("" ::: "memory");
    *(volatile lds_int_t *)(lds_int_t *)word = value;       // the LDS runs one wave's operations in order: data first, then this
}
__device__ __forceinline__ void factor_wait(const int *word, int want)
{
    while (*(const volatile lds_int_t *)(const lds_int_t *)word < want) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// FINE (tools/micro/factor_bench.hip only): dbg[64 + 4 p + i] = cycles at the pivots done / A operand ready / panel out / update back
template <bool SC1 = false, bool FINE = false>
__device__ __forceinline__ void
factor_diag_block(const double *Msrc, int lds_ld, int kblock, double *Ldiag, int *info, double *comm_ab, double *comm_l1, int *prog,
    double *linv_lds = nullptr, int linv_ld = 0, int nvalid = NB, bool ldiag_is_block = false, long long *dbg = nullptr)
{
    const long long dbg_t0 = dbg ? (long long)clock64() : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    if (wave == 0) {
        // ---- the chain ----
        __builtin_amdgcn_s_setprio(3);
        v4d C00, C01, C11;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = g + 4 * e, hi = max(i, c), lo = min(i, c);
            C00[e] = Msrc[hi * lds_ld + lo];
            C01[e] = Msrc[(16 + c) * lds_ld + i];
            C11[e] = Msrc[(16 + hi) * lds_ld + 16 + lo];
        }
        // weights of the ten entries of W in this lane's slot of the A operand: (row, column) at lane row + 16 column
        double mk[10];
        {
            const int at[10] = {0, 1, 17, 2, 18, 34, 3, 19, 35, 51};       // r0 w10 r1 w20 w21 r2 w30 w31 w32 r3
#pragma unroll
            for (int k = 0; k < 10; ++k) mk[k] = lane == at[k] ? 1.0 : 0.0;
        }
        int bad = 0;
#pragma unroll
        for (int p = 0; p < NB / 4; ++p) {
            const int qi = p >> 2, e = p & 3, b0 = 4 * e;
            if (4 * p >= nvalid) break;
            const double ad = qi ? C11[e] : C00[e];          // panel rows of the quadrant that holds its diagonal block
            // ---- the 4 x 4 diagonal block, the same in every lane ----
            const double a00 = readlane_d(ad, b0), a10 = readlane_d(ad, b0 + 1), a11 = readlane_d(ad, b0 + 17);
            const double a20 = readlane_d(ad, b0 + 2), a30 = readlane_d(ad, b0 + 3), a21 = readlane_d(ad, b0 + 18), a31 = readlane_d(ad, b0 + 19);
            const double a22 = readlane_d(ad, b0 + 34), a32 = readlane_d(ad, b0 + 35), a33 = readlane_d(ad, b0 + 51);
            // first pair of pivots: a00 and det / a00
            const double det1 = fma(a00, a11, -(a10 * a10));
            const double r0 = rsqrt_newton(a00), z1 = rsqrt_newton(det1);
            const double r1 = (a00 * r0) * z1;
            const double l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
            const double l21 = fma(-l20, l10, a21) * r1, l31 = fma(-l30, l10, a31) * r1;
            // Schur complement of the pair, second pair of pivots
            const double b22 = fma(-l21, l21, fma(-l20, l20, a22));
            const double b32 = fma(-l31, l21, fma(-l30, l20, a32));
            const double b33 = fma(-l31, l31, fma(-l30, l30, a33));
            const double det2 = fma(b22, b33, -(b32 * b32));
            const double r2 = rsqrt_newton(b22), z3 = rsqrt_newton(det2);
            const double r3 = (b22 * r2) * z3;
            const double l32 = b32 * r2;
            // all four pivots positive?  (the inputs are finite, so a NaN only follows a value that fails this test)
            if (!(fmin(fmin(a00, det1), fmin(b22, det2)) > 0.0)) bad = max(bad, 4 * p + 1);
            if (FINE) { asm volatile("" :: "v"(r3)); if (lane == 0) dbg[64 + 4 * p] = (long long)clock64() - dbg_t0; }
            // W = inv(L_pp): W[j][k] = -r_j sum_{k <= m < j} L[j][m] W[m][k]
            const double w10 = -(l10 * r0) * r1, w21 = -(l21 * r1) * r2, w32 = -(l32 * r2) * r3;
            const double w20 = -fma(l21, w10, l20 * r0) * r2, w31 = -fma(l32, w21, l31 * r1) * r3;
            const double w30 = -fma(l32, w20, fma(l31, w10, l30 * r0)) * r3;
            // lane (c, g) of the A operand: W[c][g] for g <= c < 4, else 0 -- as a sum over the ten entries with 0 / 1
            // weights per lane (at most one is 1): ten FMAs, the last row's four two operations deep, instead of ten
            // compares and twenty selects in a wave that is bound by its issue slots
            const double early = fma(mk[5], r2, fma(mk[4], w21, fma(mk[3], w20, fma(mk[2], r1, fma(mk[1], w10, mk[0] * r0)))));
            const double wop = fma(mk[6], w30, early) + fma(mk[9], r3, fma(mk[8], w32, mk[7] * w31));
            if (FINE) { asm volatile("" :: "v"(wop)); if (lane == 0) dbg[64 + 4 * p + 1] = (long long)clock64() - dbg_t0; }
            // ---- the panel: L[:, 4p .. 4p+3] = P W^T, register 0 of W_op x P^T; rows above the diagonal are zero ----
            double Ld = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, ad, zero4, 0, 0, 0)[0];
            double L1 = 0.0;
            if (qi == 0) L1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, C01[e], zero4, 0, 0, 0)[0];     // rows 16 .. 31
            Ld = c >= b0 + g ? Ld : 0.0;
            if (FINE) { asm volatile("" :: "v"(Ld)); if (lane == 0) dbg[64 + 4 * p + 2] = (long long)clock64() - dbg_t0; }
            // ---- trailing updates, the one the next panel waits for first ----
            if (qi == 0) {
                if (e < 3) C00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ld, Ld, C00, 0, 0, 0);
                C11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-L1, L1, C11, 0, 0, 0);
                if (e < 3) C01 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ld, L1, C01, 0, 0, 0);
                comm_l1[p * 64 + lane] = L1;
            } else if (e < 3) {
                C11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ld, Ld, C11, 0, 0, 0);
            }
            comm_ab[(2 * p) * 64 + lane] = wop;
            comm_ab[(2 * p + 1) * 64 + lane] = Ld;
            factor_post(prog, p + 1);                       // every lane writes the same word: no branch around it
            if (dbg && lane == 0) dbg[15 + p] = (long long)clock64() - dbg_t0;
            if (FINE && p < 7) { asm volatile("" :: "v"(p < 3 ? C00[(e + 1) & 3] : C11[(e + 1) & 3])); if (lane == 0) dbg[64 + 4 * p + 3] = (long long)clock64() - dbg_t0; }
        }
        __builtin_amdgcn_s_setprio(0);
        if (bad && lane == 0) atomicMax(info, kblock * NB + bad);
    } else if (wave == 1) {
        // ---- the inverse ----
        v4d T00, T10, T11;
#pragma unroll
        for (int e = 0; e < 4; ++e) { T00[e] = g + 4 * e == c ? 1.0 : 0.0; T11[e] = T00[e]; T10[e] = 0.0; }
        double *Lk = Ldiag + (ldiag_is_block ? 0 : (size_t)kblock * NB * NB);   // [i][j] = inv(L)[i][j]
        // (Msrc == linv_lds is safe: the first rows of the inverse are stored behind the wait for panel 0, which wave 0
        //  publishes with the whole block in its registers)
#pragma unroll
        for (int p = 0; p < NB / 4; ++p) {
            const int qi = p >> 2, e = p & 3;
            const double E0 = qi ? T10[e] : T00[e];          // panel of E, rows 0 .. 15
            double Y0 = E0, Y1 = qi ? T11[e] : 0.0;          // rows 4p .. 4p+3 of inv(L): columns 0 .. 15, 16 .. 31
            if (4 * p < nvalid) {
                factor_wait(prog, p + 1);
                const double wop = comm_ab[(2 * p) * 64 + lane], nLd = -comm_ab[(2 * p + 1) * 64 + lane];
                Y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, E0, zero4, 0, 0, 0)[0];
                if (qi == 0) {
                    if (e < 3) T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(nLd, Y0, T00, 0, 0, 0);
                    T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-comm_l1[p * 64 + lane], Y0, T10, 0, 0, 0);
                } else {
                    Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, T11[e], zero4, 0, 0, 0)[0];
                    if (e < 3) {
                        T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(nLd, Y0, T10, 0, 0, 0);
                        T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(nLd, Y1, T11, 0, 0, 0);
                    }
                }
            }
            // rows 4p .. 4p+3 of inv(L) are final: lane (c, g) holds [4p + g][c] and [4p + g][16 + c]
            double *row = Lk + (4 * p + g) * NB;
            if (SC1) { store_sc1(row + c, Y0); store_sc1(row + 16 + c, Y1); }
            else { row[c] = Y0; row[16 + c] = Y1; }
            if (linv_lds) { linv_lds[(4 * p + g) * linv_ld + c] = Y0; linv_lds[(4 * p + g) * linv_ld + 16 + c] = Y1; }
            if (dbg && lane == 0) dbg[23 + p] = (long long)clock64() - dbg_t0;
        }
        if (lane == 0) factor_post(prog, 0);
        if (dbg && lane == 0) dbg[0] = (long long)clock64() - dbg_t0;
    }
}

// LDS scratch of factor_diag_block: (W, L_p) of the eight panels, L_p's rows 16 .. 31 of the first four, progress words
struct FactorComm {
    double ab[8 * 2 * 64];
    double l1[4 * 64];
    int prog[2];
};

// Block 0: nothing to update, just the factor.
__global__ __launch_bounds__(128, 1) void
chol_first_kernel(const double *A, int ld, double *Ldiag, int *info, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed || lm->flow_aborted)) return;
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) FactorComm fc;
    const int lane = threadIdx.x, r = lane & 31;
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) M[r][c] = A[(size_t)r * ld + c];
    }
    if (lane == 64) { fc.prog[0] = 0; fc.prog[1] = 0; }
    __syncthreads();
    factor_diag_block(&M[0][0], NB + 1, 0, Ldiag, info, fc.ab, fc.l1, fc.prog);
}

// A system of one block (n <= 32: the three-camera adjustments of the incremental
// reconstruction) start to finish in one launch of one wave: factor, y = inv(L) b,
// x = inv(L)^T y, and the candidate cameras Plus(x, -step) that the next kernel needs --
// four launches of a latency-bound chain in one.  (Two waves for the factor, one for the rest.)
__global__ __launch_bounds__(128, 1) void
chol_small_kernel(const double *A, int ld, int n, double *Ldiag, double *x, int *info, BaDev d,
    double *partials_cam)
{
    if (!lm_resolve(d)) return;
    if (d.lm == nullptr || d.lm->lin_failed) return;      // this kernel writes the candidate cameras of an LM solve: no state, nothing to do
    __shared__ __attribute__((aligned(16))) double M[NB][NB + 1];
    __shared__ double ys[NB], xs[NB];
    __shared__ __attribute__((aligned(16))) FactorComm fc;
    const int lane = threadIdx.x, r = lane & 31;
    double b = 0.0;
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) M[r][c] = A[(size_t)r * ld + c];
        b = A[(size_t)NB * ld + r];
    }
    if (lane == 64) { fc.prog[0] = 0; fc.prog[1] = 0; }
    __syncthreads();
    factor_diag_block(&M[0][0], NB + 1, 0, Ldiag, info, fc.ab, fc.l1, fc.prog, &M[0][0], NB + 1, n);  // M := inv(L)
    __syncthreads();
    if (lane >= 64) return;                  // the substitutions and the camera update are one wave's work
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
    // (four lanes per camera: each a quarter of the candidate's table row)
    double *cams_out = d.lm->cur ? d.cams2[0] : d.cams2[1];
    double *table_out = d.lm->cur ? d.camder2[0] : d.camder2[1];
    for (int c = lane >> 2; c < d.C; c += 16) cam_update_part(d, xs, c, lane & 3, cams_out, partials_cam, table_out, kCamDer);
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
    if (lm && (lm->stop || lm->lin_failed || lm->flow_aborted)) return;
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
    __shared__ int prog[2];
    if (tid == 0) { prog[0] = 0; prog[1] = 0; }
    __syncthreads();
    // (Xi / Xj were last read in front of that barrier: they are the factor's scratch now)
    if (tid < 128) factor_diag_block(&Ai[0][0], NB + 1, k + 1, Ldiag, info, &Xi[0][0], &Xj[0][0], prog);
}


// ---------------------------------------------------------------------------
// The whole factorisation (and the forward substitution of the right-hand side riding along)
// as ONE launch of persistent workgroups that hand tiles to each other: the launch-per-
// block-column form above spends 15.6 us per column of which 2.7 are the pivot chain.
//
// Roles (all workgroups resident for the whole launch):
//   D workgroups own block rows: D_w takes rows w, w + num_d, w + 2 num_d, ... one after the other (every
//        system of BASELINE's configs has a D of its own per row; beyond ~80 rows they are shared, so that the
//        grid fits the device whatever the size).  For its row r a D holds the last kFlowW + 1 tiles
//        (r, r-kFlowW) .. (r, r) in registers -- the diagonal tile and its left neighbours -- and applies the
//        panels of earlier columns as they appear (right-looking).  For a column k it owns it turns its tile
//        into L_rk itself as soon as inv(L_kk) appears, applies it to its tiles right of k, and after the last
//        column factors its diagonal tile and publishes inv(L_rr).  Row nblk is the tail of the right-hand side.
//   P workgroups own the tiles (i, j), i - j > kFlowW (the right-hand side row i = nblk included): the
//        column-major list of them is dealt round robin, P_w takes tiles w, w + num_p, ...  A tile is done
//        left-looking, start to finish: all updates of the columns before j, then L_ij = tile * inv(L_jj)^T
//        once that inverse appears.  (Round 3 gave every tile a workgroup of its own, which stopped at ~38
//        block columns: the 500-view problem of BASELINE configs[4] has 78.)  A workgroup's tiles come in
//        column order and every dependency of a tile lies in an earlier column (or is a D product of the row's
//        own columns), so the earliest unfinished tile of the whole grid can always proceed: no cycle of waits.
//        Catching up on the columns that were finished while the workgroup was busy elsewhere is what a tile
//        mostly does, so it probes up to 16 steps ahead with one poll and fetches two steps per round.
// Why the band: POTRF(j) -> TRSM(j+1, j) -> SYRK -> POTRF(j+1) is the critical path, and row
// j+1 enters it with everything the columns before j did to it.  A tile handed from one
// workgroup to another costs a round trip through memory (sc1 store, drain, flag, poll, sc1
// load: ~5.5 us measured here), so the hand-offs on that path must be few and the others need
// slack: with the diagonal tile alone per workgroup the loop inverse(j-1) -> P computes
// L(j+1, j-1) -> D_(j+1) updates paced the whole thing at 10 us per column.  With the band, a P
// tile's result is needed kFlowW columns after the inverse it waited for.
// The D workgroups talk to each other through their XCD's L2 where they can: the blocks with
// blockIdx % 8 == 0 are the D's (observed placement: round robin over the XCDs -- speed only),
// every payload is published twice -- plain stores into a mailbox + a flag that carries the
// producer's XCC id (s_getreg HW_REG_XCC_ID), then sc1 stores into its place in the factor + a
// second flag --, and a D reads the mailbox only when the flag says the producer sits on its
// own XCD (one L2: the plain stores are there once their vmcnt has drained; the reads bypass
// L1).  Everybody else, and a D on another XCD, takes the sc1 copy as MI355X_MICROARCH.md
// ("Valid forms") prescribes: every byte stored sc1 and loaded sc1, each storing wave drains
// its stores before one lane stores the flag behind a workgroup barrier, the lanes of ONE wave
// poll, the others load behind the barrier it joins.  Flags hold the launch's epoch (no reset
// between factorisations).  All workgroups must be resident (the grid is sized by the occupancy
// query with a margin); a poll that outlasts spin_limit raises the abort word, which every poll
// loop watches: the launch drains, info carries kFlowAborted, and the CALLER repeats the
// factorisation in the launch-per-column form (ba_api.hip) -- a scheduling condition, e.g. a
// foreign kernel holding CUs, must not look like a matrix that is not positive definite.
// ---------------------------------------------------------------------------
constexpr int kFlowSpinLimitDefault = 1 << 21;
constexpr int kFlowW = 3;                 // left neighbours of the diagonal tile a D workgroup owns
constexpr int kFlowMaxD = 56;             // D workgroups of a launch (one XCD holds 64 workgroups of this kernel)
constexpr int kFlowMaxBlocks = 160;       // block columns the one-launch form takes (5120 unknowns)
constexpr int kFlowTraceStride = 32;      // int64 stamps per block row of the diagnostic trace
constexpr int kFlowProbe = 16;            // steps a P tile looks ahead with one poll

struct CholFlow {
    const double *A;      // (N + 32) x N reduced system, rhs in row N
    double *Lmat;         // factor + solved rhs (same layout)
    double *Ldiag;        // inv(L_kk), 32 x 32 each
    double *mailbox;      // [(nblk + 1)][kFlowW + 1][32 x 32]: slot 0 inv(L_rr), slot d = L(r, r - d); same-XCD copies
    int *flags;           // see flow_*_flag
    int *info;
    const LmDev *lm;
    double *x;            // solution of the reduced system (n entries used), written by the backward phase; null: factor only
    int ld, nblk, epoch;
    int num_d, num_p;     // D / P workgroups of the launch
    int num_tiles;        // P tiles
    int spin_limit;
    long long *trace;     // diagnostics (tools/chol_flow_trace.py): [nblk + 1][16] wall_clock64 stamps of the D's, or null
    // block pattern of the factor (ba_order.hip; null: dense).  A tile that is structurally zero is never waited
    // for, multiplied or published: skipping it changes no value (its products are exact zeros), it removes the
    // WAIT -- the arcs of an ordered ring are chains of their own.
    const unsigned long long *nz;     // [(nblk + 1)][kNzWords]
    const int32_t *ptiles;            // the P tiles as i << 16 | j, column-major (null: all of them, enumerated)
};

static_assert(160 + 1 <= 64 * kNzWords, "a row of the block pattern must hold every column");
// row i of the block pattern (all ones without one); bit k of a row
struct FlowRow { unsigned long long w0, w1, w2; };
__device__ __forceinline__ FlowRow flow_row(const CholFlow &f, int i)
{
    FlowRow r = {~0ull, ~0ull, ~0ull};
    if (f.nz) { r.w0 = f.nz[(size_t)i * kNzWords]; r.w1 = f.nz[(size_t)i * kNzWords + 1]; r.w2 = f.nz[(size_t)i * kNzWords + 2]; }
    return r;
}
// tile (i, k) of the factor exists -- read where it is asked for: rows kept in registers cost the D part 18 scalar
// registers it does not have (the kernel then spilled scalars into scratch memory, and its hand-offs raced)
__device__ __forceinline__ bool flow_nz(const CholFlow &f, int i, int k)
{
    return f.nz == nullptr || ((f.nz[(size_t)i * kNzWords + (k >> 6)] >> (k & 63)) & 1ull) != 0;
}
__device__ __forceinline__ FlowRow flow_row_and(const FlowRow &a, const FlowRow &b) { return FlowRow{a.w0 & b.w0, a.w1 & b.w1, a.w2 & b.w2}; }
__device__ __forceinline__ bool flow_bit(const FlowRow &r, int k)
{
    const unsigned long long w = k < 64 ? r.w0 : k < 128 ? r.w1 : r.w2;
    return (w >> (k & 63)) & 1ull;
}

// flag words: sc1 copy of tile (i, k) final | sc1 copy of inverse k | mailbox copies (r, slot) | abort
__device__ __forceinline__ int flow_tile_flag(const CholFlow &f, int i, int k) { return i * f.nblk + k; }
__device__ __forceinline__ int flow_inv_flag(const CholFlow &f, int k) { return (f.nblk + 1) * f.nblk + k; }
__device__ __forceinline__ int flow_box_flag(const CholFlow &f, int r, int slot) { return (f.nblk + 2) * f.nblk + r * (kFlowW + 1) + slot; }
__device__ __forceinline__ int flow_abort_flag(const CholFlow &f) { return (f.nblk + 2) * f.nblk + (f.nblk + 1) * (kFlowW + 1); }
__device__ __forceinline__ double *flow_box(const CholFlow &f, int r, int slot) { return f.mailbox + ((size_t)r * (kFlowW + 1) + slot) * NB * NB; }
// backward phase: block k of the solution, sc1 copy in f.x / same-XCD copy behind the tiles of the mailbox
__device__ __forceinline__ int flow_x_flag(const CholFlow &f, int k) { return flow_abort_flag(f) + 1 + k; }
__device__ __forceinline__ int flow_xbox_flag(const CholFlow &f, int k) { return flow_abort_flag(f) + 1 + f.nblk + k; }
__device__ __forceinline__ double *flow_xbox(const CholFlow &f, int k) { return f.mailbox + (size_t)(f.nblk + 1) * (kFlowW + 1) * NB * NB + (size_t)k * NB; }
__device__ __forceinline__ int flow_xcc() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }
// ... and as GRANULES in the flag block (zero at the start of a solve like the flags): 16 bytes { value, tag } per
// entry, the data is the flag (cdna_hip_programming.md R2; 16-byte stores and sc1 loads observed untorn on gfx950).
// Two copies per block: [0] stored plain, tag = epoch | producer's XCD -- a reader behind the same L2 sees it there;
// [1] stored sc1, tag = epoch, for everybody.  A reader asks for both side by side: ONE round trip per hand-off,
// where a flag poll followed by the load of the 32 doubles was two.
typedef unsigned int flow_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int flow_granule_base(const CholFlow &f) { return (flow_abort_flag(f) + 1 + 2 * f.nblk + 3) & ~3; }
__device__ __forceinline__ flow_u4 *flow_x_granules(const CholFlow &f, int k, int copy)
{
    return reinterpret_cast<flow_u4 *>(f.flags + flow_granule_base(f)) + ((size_t)k * 2 + copy) * NB;
}
__device__ __forceinline__ flow_u4 flow_granule(double v, int tag)
{
    flow_u4 g;
    g.x = (unsigned)__double2loint(v); g.y = (unsigned)__double2hiint(v); g.z = (unsigned)tag; g.w = 0u;
    return g;
}

// plain store / flag for a reader behind the same L2 (no cache-policy bits)
__device__ __forceinline__ void store_flag_plain(int *p, int v)
{
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

struct FlowWaiter {
    int *how;             // LDS, 2 x 8 words: verdict of lane l's payload -- 0 aborted, 1 sc1 copy, 2 mailbox copy.  Two sets, used
                          // alternately: a wave may still be reading the verdicts of one wait while wave 0 writes those of the next
                          // (every wait holds a barrier, so nobody lags two waits behind)
    int *word;            // LDS, one word (probe results)
    int *prog;            // LDS, two words, zero between uses: progress of the diagonal factor's waves
    int phase;            // verdict set of the most recent wait
    int pending_flag;     // sc1 flag to store once this workgroup's sc1 stores have drained (-1: none)
};
__device__ __forceinline__ int flow_how(const FlowWaiter &w, int lane) { return w.how[w.phase * 8 + lane]; }

// Waits until every payload that a lane of wave 0 names is published: (slow_flag) the sc1 copy, or -- D to D
// only -- (box_flag) the mailbox copy where the producer shares this workgroup's L2.  slow_flag < 0: the lane
// names nothing.  All threads call it; the arguments of the threads beyond wave 0 are ignored.  Every thread
// drains its own stores first (a pending sc1 publication of this workgroup becomes visible here, for free:
// the waves would idle at the barrier anyway).  Returns false when the launch was aborted.
__device__ __forceinline__ bool flow_wait_lanes(const CholFlow &f, FlowWaiter &w, int slow_flag, int box_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w.pending_flag >= 0) {
        lds_barrier();
        if (threadIdx.x == 0) __hip_atomic_store(f.flags + w.pending_flag, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w.pending_flag = -1;
    }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int verdict = slow_flag < 0 ? 1 : 0;
        bool ok = true;
        const int *ps = f.flags + max(slow_flag, 0), *pb = f.flags + max(box_flag, 0), *pab = f.flags + flow_abort_flag(f);
        const int want_box = f.epoch | ((flow_xcc() + 1) << 24);
        for (int spins = 0;; ++spins) {
            if (!verdict) {
                // both words asked for side by side: one round trip per poll, not two (a poll that found the mailbox
                // flag unset went on to the second load only after the first had returned)
                const int vb = __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int vs = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (box_flag >= 0 && vb == want_box) verdict = 2;
                else if (vs == f.epoch) verdict = 1;
            }
            if (__ballot(verdict == 0) == 0) break;
            if ((spins & 15) == 15 && __hip_atomic_load(pab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.epoch) { ok = false; break; }
            if (spins > f.spin_limit) {
                if (lane == 0) __hip_atomic_store(f.flags + flow_abort_flag(f), f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok && lane == 0) atomicMax(f.info, kFlowAborted);      // the caller repeats the factorisation launch by launch
        if (lane < 8) w.how[(w.phase ^ 1) * 8 + lane] = ok ? verdict : 0;
    }
    w.phase ^= 1;
    lds_barrier();
    return flow_how(w, 0) != 0;
}

// one payload, named by every thread alike; returns its verdict (0 aborted, 1 sc1 copy, 2 mailbox copy)
__device__ __forceinline__ int flow_wait(const CholFlow &f, FlowWaiter &w, int slow_flag, int box_flag)
{
    const bool first = threadIdx.x == 0;
    flow_wait_lanes(f, w, first ? slow_flag : -1, first ? box_flag : -1);
    return flow_how(w, 0);
}

// Thread <-> tile element map of the flow kernel: the accumulator layout of
// v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md: col = lane & 15, row = (lane >> 4) + 4 * reg),
// wave w holding the 16 x 16 quadrant (w >> 1, w & 1) of the 32 x 32 tile: element e of a thread
// is (tr0 + 4 e, tc).
struct FlowPos { int tr0, tc, ar, ak; };    // ar / ak: row inside a quadrant and k offset of the MFMA A / B operand
__device__ __forceinline__ FlowPos flow_pos()
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    FlowPos p;
    p.tr0 = 16 * (wave >> 1) + (lane >> 4);
    p.tc = 16 * (wave & 1) + (lane & 15);
    p.ar = lane & 15; p.ak = lane >> 4;
    return p;
}

// a published 32 x 32 tile (row-major, leading dimension ld): sc1 loads (L1 bypass) into registers, then into LDS --
// in two steps so that a caller can have the loads of several tiles in flight before the first LDS write waits
struct FlowTile { double v[4]; };
__device__ __forceinline__ FlowTile flow_fetch_tile(const double *G, int ld, const FlowPos &p)
{
    FlowTile t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t.v[e] = load_sc1(G + (size_t)(p.tr0 + 4 * e) * ld + p.tc);
    return t;
}
__device__ __forceinline__ void flow_put_tile(const FlowTile &t, double (*dst)[NB + 1], const FlowPos &p)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[p.tr0 + 4 * e][p.tc] = t.v[e];
}
__device__ __forceinline__ void flow_load_tile(const double *G, int ld, double (*dst)[NB + 1], const FlowPos &p)
{
    flow_put_tile(flow_fetch_tile(G, ld, p), dst, p);
}

// acc (+/-)= X Y^T on the matrix cores: D[i][j] += sum_m X[i][m] Y[j][m], eight k-steps of four.
// A operand: lane holds X[16 qi + (lane & 15)][4 s + (lane >> 4)], B operand Y[16 qj + (lane & 15)][same k].
// (As 128 vector FMAs per thread fed from LDS this took 2.6 us per tile, on the critical path twice
//  per block column.)
template <bool NEG>
__device__ __forceinline__ void flow_mma(double (&acc)[4], const double (*X)[NB + 1], const double (*Y)[NB + 1], const FlowPos &p)
{
    const int wave = threadIdx.x >> 6;
    const int xi = 16 * (wave >> 1) + p.ar, yj = 16 * (wave & 1) + p.ar;
    double a[8], bq[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) { a[s8] = X[xi][4 * s8 + p.ak]; bq[s8] = Y[yj][4 * s8 + p.ak]; }
    v4d c = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) c = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a[s8] : a[s8], bq[s8], c, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = c[e];
}

typedef double (*FlowLds)[NB + 1];

// One P tile (i, j), i - j > kFlowW, start to finish.  Returns false when the launch was aborted.
__device__ __forceinline__ bool
flow_p_tile(const CholFlow &f, FlowWaiter &w, const FlowPos &p, int i, int j, FlowLds Xr, FlowLds Xc, FlowLds Li, FlowLds Mt)
{
    const int ld = f.ld, tid = threadIdx.x;
    double acc[4];
    const double *Aij = f.A + (size_t)(i * NB) * ld + j * NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = Aij[(size_t)(p.tr0 + 4 * e) * ld + p.tc];
    // the steps that have both operands: L(i, k) and L(j, k) structurally nonzero
    const FlowRow need = flow_row_and(flow_row(f, i), flow_row(f, j));
    int k = 0;
    while (k < j) {
        if (!flow_bit(need, k)) { ++k; continue; }
        // how many of the steps k, k + 1, ... have both their operands -- L(i, .) and L(j, .) -- published already?
        // (lane 2 m: tile (i, k + m), lane 2 m + 1: tile (j, k + m); one load per lane, no waiting)
        if (tid < 64) {
            const int m = tid >> 1, kk = k + m;
            bool missing = false;
            if (m < kFlowProbe && kk < j && flow_bit(need, kk))
                missing = __hip_atomic_load(f.flags + flow_tile_flag(f, (tid & 1) ? j : i, kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != f.epoch;
            const unsigned long long mm = __ballot(missing);
            const int ready = mm ? (int)(__builtin_ctzll(mm) >> 1) : min(kFlowProbe, j - k);
            if (tid == 0) *w.word = ready;
        }
        lds_barrier();
        int ready = *w.word;
        if (ready == 0) {
            // the step the factorisation is at: wait for both operands (polled side by side)
            if (!flow_wait_lanes(f, w, tid == 0 ? flow_tile_flag(f, i, k) : tid == 1 ? flow_tile_flag(f, j, k) : -1, -1)) return false;
            ready = 1;
        }
        for (int m = 0; m < ready;) {
            // the next one or two steps that exist (uniform: the pattern is the launch's)
            while (m < ready && !flow_bit(need, k + m)) ++m;
            if (m >= ready) break;
            const int ka = k + m;
            ++m;
            while (m < ready && !flow_bit(need, k + m)) ++m;
            const bool two = m < ready;
            const int kb = k + m;
            if (two) ++m;
            const FlowTile t0 = flow_fetch_tile(f.Lmat + (size_t)(i * NB) * ld + ka * NB, ld, p);
            const FlowTile t1 = flow_fetch_tile(f.Lmat + (size_t)(j * NB) * ld + ka * NB, ld, p);
            if (two) {
                const FlowTile t2 = flow_fetch_tile(f.Lmat + (size_t)(i * NB) * ld + kb * NB, ld, p);
                const FlowTile t3 = flow_fetch_tile(f.Lmat + (size_t)(j * NB) * ld + kb * NB, ld, p);
                flow_put_tile(t0, Xr, p); flow_put_tile(t1, Xc, p); flow_put_tile(t2, Li, p); flow_put_tile(t3, Mt, p);
            } else {
                flow_put_tile(t0, Xr, p); flow_put_tile(t1, Xc, p);
            }
            lds_barrier();
            flow_mma<true>(acc, Xr, Xc, p);
            if (two) flow_mma<true>(acc, Li, Mt, p);
            lds_barrier();          // the buffers are rewritten by the next round
        }
        k += ready;
    }
    if (!flow_wait(f, w, flow_inv_flag(f, j), -1)) return false;
    flow_load_tile(f.Ldiag + (size_t)j * NB * NB, NB, Li, p);
#pragma unroll
    for (int e = 0; e < 4; ++e) Mt[p.tr0 + 4 * e][p.tc] = acc[e];
    lds_barrier();
    // L_ij = tile * inv(L_jj)^T (inv(L) is lower triangular: zeros above the diagonal)
    double x[4] = {0, 0, 0, 0};
    flow_mma<false>(x, Mt, Li, p);
    double *Lij = f.Lmat + (size_t)(i * NB) * ld + j * NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) store_sc1(Lij + (size_t)(p.tr0 + 4 * e) * ld + p.tc, x[e]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(f.flags + flow_tile_flag(f, i, j), f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Forward part of block row `row` by a D workgroup: updates, its own band of L, the factor of the diagonal tile
// and inv(L_rr) published (which is left in Li for the backward phase).  Returns false when the launch was aborted.
__device__ __forceinline__ bool
flow_d_forward(const CholFlow &f, FlowWaiter &w, const FlowPos &p, int row, FlowLds Xr, FlowLds Xc, FlowLds Li, FlowLds Mt)
{
    const int nblk = f.nblk, ld = f.ld, tid = threadIdx.x;
    const bool has_diag = row < nblk;
    const int lo = max(0, row - kFlowW), hi = min(row, nblk - 1);     // own columns
    const int my_tag = f.epoch | ((flow_xcc() + 1) << 24);
    auto stamp = [&](int slot) { if (f.trace && tid == 0) f.trace[row * kFlowTraceStride + slot] = (long long)wall_clock64(); };
    stamp(0);
    double acc[kFlowW + 1][4];
#pragma unroll
    for (int t = 0; t <= kFlowW; ++t) {
        const int c = lo + t;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.0;
        if (c <= hi) {
            const double *At = f.A + (size_t)(row * NB) * ld + c * NB;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] = At[(size_t)(p.tr0 + 4 * e) * ld + p.tc];
        }
    }
    // ---- columns left of the band: L(row, k) is a P tile, L(c, k) of the own columns c != row a P tile or a
    //      neighbour's band tile; all of a step's flags are polled side by side and its tiles fetched together ----
    for (int k = 0; k < lo; ++k) {
        if (!flow_nz(f, row, k)) continue;        // L(row, k) = 0: column k changes none of the own tiles
        bool has[kFlowW];
#pragma unroll
        for (int t = 0; t < kFlowW; ++t) has[t] = lo + t <= hi && lo + t != row && flow_nz(f, lo + t, k);
        int slow = -1, box = -1;
        if (tid == 0) slow = flow_tile_flag(f, row, k);
        else if (tid <= kFlowW) {
            const int c = lo + (int)tid - 1;
            const bool hc = tid == 1 ? has[0] : tid == 2 ? has[1] : has[2];
            if (hc) {
                slow = flow_tile_flag(f, c, k);
                if (c - k <= kFlowW) box = flow_box_flag(f, c, c - k);
            }
        }
        if (!flow_wait_lanes(f, w, slow, box)) return false;
        const FlowTile tr = flow_fetch_tile(f.Lmat + (size_t)(row * NB) * ld + k * NB, ld, p);
        FlowTile tc[kFlowW];
#pragma unroll
        for (int t = 0; t < kFlowW; ++t) {
            const int c = lo + t;
            if (has[t]) {
                if (flow_how(w, t + 1) == 2) tc[t] = flow_fetch_tile(flow_box(f, c, c - k), NB, p);
                else tc[t] = flow_fetch_tile(f.Lmat + (size_t)(c * NB) * ld + k * NB, ld, p);
            }
        }
        flow_put_tile(tr, Xr, p);
#pragma unroll
        for (int t = 0; t < kFlowW; ++t)
            if (has[t]) flow_put_tile(tc[t], t == 0 ? Xc : t == 1 ? Li : Mt, p);
        lds_barrier();
#pragma unroll
        for (int t = 0; t <= kFlowW; ++t) {
            const int c = lo + t;
            if (c <= hi && (c == row || (t < kFlowW && has[t < kFlowW ? t : 0])))
                flow_mma<true>(acc[t], Xr, c == row ? Xr : t == 0 ? Xc : t == 1 ? Li : Mt, p);
        }
        lds_barrier();
    }
    // ---- the band: columns lo .. row - 1, the critical path ----
    for (int k = lo; k < row; ++k) {
        // L(row, k) = 0 (another arc's column): nothing to solve, publish or apply -- and nothing to wait for
        if (!flow_nz(f, row, k)) continue;
        {
            const int how = flow_wait(f, w, flow_inv_flag(f, k), flow_box_flag(f, k, 0));
            if (!how) return false;
            if (k == row - 1 && f.trace && tid == 0) f.trace[row * kFlowTraceStride + 7] = how;
            flow_load_tile(how == 2 ? flow_box(f, k, 0) : f.Ldiag + (size_t)k * NB * NB, NB, Li, p);
            // own tile of column k (register index is compile-time under the unrolled select)
            double t4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t <= kFlowW; ++t)
                if (lo + t == k) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t4[e] = acc[t][e];
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) Mt[p.tr0 + 4 * e][p.tc] = t4[e];
            lds_barrier();
            double x[4] = {0, 0, 0, 0};
            flow_mma<false>(x, Mt, Li, p);
            // publish: mailbox (same-XCD D's), then the sc1 copy in the factor (flag at the next wait)
            double *box = flow_box(f, row, row - k);
            double *Lrk = f.Lmat + (size_t)(row * NB) * ld + k * NB;
#pragma unroll
            for (int e = 0; e < 4; ++e) { Xr[p.tr0 + 4 * e][p.tc] = x[e]; box[(p.tr0 + 4 * e) * NB + p.tc] = x[e]; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (tid == 0) store_flag_plain(f.flags + flow_box_flag(f, row, row - k), my_tag);
            if (k == row - 1) stamp(3);
#pragma unroll
            for (int e = 0; e < 4; ++e) store_sc1(Lrk + (size_t)(p.tr0 + 4 * e) * ld + p.tc, x[e]);
            w.pending_flag = flow_tile_flag(f, row, k);
        }
        // ---- apply column k to the own tiles right of it ----
        for (int c = max(k + 1, lo); c <= hi; ++c) {
            if (c != row && !flow_nz(f, c, k)) continue;       // L(c, k) = 0
            if (c == row) {
                lds_barrier();
            } else {
                // L(c, k): a D's if c - k <= kFlowW, else a P tile
                const bool from_d = c - k <= kFlowW;
                const int how = flow_wait(f, w, flow_tile_flag(f, c, k), from_d ? flow_box_flag(f, c, c - k) : -1);
                if (!how) return false;
                if (how == 2) flow_load_tile(flow_box(f, c, c - k), NB, Xc, p);
                else flow_load_tile(f.Lmat + (size_t)(c * NB) * ld + k * NB, ld, Xc, p);
                lds_barrier();
            }
#pragma unroll
            for (int t = 0; t <= kFlowW; ++t)
                if (lo + t == c) flow_mma<true>(acc[t], Xr, c == row ? Xr : Xc, p);
        }
        if (k == row - 1) stamp(9);
    }
    if (has_diag) {
        // the own-tile branch above ends on a barrier-free update: order it before Mt is rewritten
        lds_barrier();
#pragma unroll
        for (int t = 0; t <= kFlowW; ++t)
            if (lo + t == row) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Mt[p.tr0 + 4 * e][p.tc] = acc[t][e];
            }
        lds_barrier();
        stamp(4);
        const long long c_start = (long long)clock64();
        if (tid < 128) {
            // the critical path: the factor (two waves; Xr / Xc are its scratch), then its inverse to the mailbox
            // (D_(row+1) is polling) -- stored by the second wave, which therefore drains and flags
            factor_diag_block<false>(&Mt[0][0], NB + 1, row, flow_box(f, row, 0), f.info, &Xr[0][0], &Xc[0][0], w.prog, &Li[0][0], NB + 1, NB, true,
                f.trace ? f.trace + row * kFlowTraceStride + 1 : nullptr);
            if ((tid >> 6) == 1) {
                if (f.trace && tid == 64) f.trace[row * kFlowTraceStride + 2] = (long long)clock64() - c_start;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 64) store_flag_plain(f.flags + flow_box_flag(f, row, 0), my_tag);
                if (f.trace && tid == 64) { f.trace[row * kFlowTraceStride + 5] = (long long)wall_clock64(); f.trace[row * kFlowTraceStride + 6] = (long long)clock64() - c_start; }
            }
        }
        lds_barrier();
        // the sc1 copy of the inverse for everybody else (Li holds it)
        double *Lk = f.Ldiag + (size_t)row * NB * NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) store_sc1(Lk + (p.tr0 + 4 * e) * NB + p.tc, Li[p.tr0 + 4 * e][p.tc]);
    }
    // drain, then the flags of what is still pending
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (w.pending_flag >= 0) __hip_atomic_store(f.flags + w.pending_flag, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has_diag) __hip_atomic_store(f.flags + flow_inv_flag(f, row), f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    w.pending_flag = -1;
    return true;
}

// Backward substitution for block `row` of the solution by the D workgroup that owns the row:
// x_k = inv(L_kk)^T (y_k - sum_{i > k} L_ik^T x_i).  inv(L_kk) is in Li (li_resident) or fetched from the
// workgroup's own sc1 copy; y_k is row 0 of the right-hand side tile (nblk, k).  The blocks of the solution are a
// chain -- x_k cannot start before x_(k+1) is there -- and a link of it was a flag poll, the load of the 32
// doubles, two 32 x 32 matrix-vector products of 32 dependent multiply-adds each with a barrier between them,
// a store, a drain and a flag: 2.4 us, 77 of the 277 us of a 32-block system.  Now everything that does not
// need x_(k+1) is done before it arrives: the tiles L(i, k), i > k + 1, are applied as their x_i come in (each fetched
// while the workgroup waits), u = inv(L_kk)^T s and M = inv(L_kk)^T L(k+1, k)^T are formed, and the link itself is one
// wave's work: the granules of x_(k+1) (value and tag in one 16-byte word: one round trip), x_k = u - M x_(k+1) with
// four partial sums per lane, the granules of x_k.
__device__ __forceinline__ bool
flow_wait_x(const CholFlow &f, FlowWaiter &w, int i, double *xv, bool with_barrier)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w.pending_flag >= 0) {
        lds_barrier();
        if (threadIdx.x == 0) __hip_atomic_store(f.flags + w.pending_flag, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w.pending_flag = -1;
    }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool have = lane >= NB, ok = true;
        double val = 0.0;
        const flow_u4 *gp = flow_x_granules(f, i, 0) + (lane & (NB - 1)), *gs = flow_x_granules(f, i, 1) + (lane & (NB - 1));
        const int *pab = f.flags + flow_abort_flag(f);
        const unsigned want_box = (unsigned)(f.epoch | ((flow_xcc() + 1) << 24));
        for (int spins = 0;; ++spins) {
            if (!have) {
                flow_u4 a, b;
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                    : "=&v"(a), "=&v"(b) : "v"(gp), "v"(gs) : "memory");
                if (a.z == want_box) { val = __hiloint2double((int)a.y, (int)a.x); have = true; }
                else if (b.z == (unsigned)f.epoch) { val = __hiloint2double((int)b.y, (int)b.x); have = true; }
            }
            if (__ballot(!have) == 0) break;
            if ((spins & 15) == 15 && __hip_atomic_load(pab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.epoch) { ok = false; break; }
            if (spins > f.spin_limit) {
                if (lane == 0) __hip_atomic_store(f.flags + flow_abort_flag(f), f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok && lane == 0) atomicMax(f.info, kFlowAborted);      // the caller repeats the factorisation launch by launch
        if (lane < NB) xv[lane] = val;
        if (lane < 8) w.how[(w.phase ^ 1) * 8 + lane] = ok ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    w.phase ^= 1;
    if (with_barrier) lds_barrier();
    return true;
}

// the 32 lanes' dot products acc[lane] -= sum_r T[r][lane] x[r] (or T[lane][r]: ROWS), four partial sums each
template <bool ROWS>
__device__ __forceinline__ double flow_matvec_sub(double acc, const double (*T)[NB + 1], const double *xv, int lane)
{
    double a0 = acc, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int r = 0; r < NB; r += 4) {
        a0 = fma(-(ROWS ? T[lane][r] : T[r][lane]), xv[r], a0);
        a1 = fma(-(ROWS ? T[lane][r + 1] : T[r + 1][lane]), xv[r + 1], a1);
        a2 = fma(-(ROWS ? T[lane][r + 2] : T[r + 2][lane]), xv[r + 2], a2);
        a3 = fma(-(ROWS ? T[lane][r + 3] : T[r + 3][lane]), xv[r + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ bool
flow_d_backward(const CholFlow &f, FlowWaiter &w, const FlowPos &p, int row, bool li_resident, FlowLds Xr, FlowLds Xc, FlowLds Li,
    FlowLds Mx, double *sv, double *xv)
{
    const int nblk = f.nblk, ld = f.ld, tid = threadIdx.x;
    const int my_tag = f.epoch | ((flow_xcc() + 1) << 24);
    if (!li_resident) {
        lds_barrier();
        flow_load_tile(f.Ldiag + (size_t)row * NB * NB, NB, Li, p);
    }
    {
        const bool from_d = nblk - row <= kFlowW;
        const int how = flow_wait(f, w, flow_tile_flag(f, nblk, row), from_d ? flow_box_flag(f, nblk, nblk - row) : -1);
        if (!how) return false;
        if (tid < NB) sv[tid] = load_sc1(how == 2 ? flow_box(f, nblk, nblk - row) + tid : f.Lmat + (size_t)(nblk * NB) * ld + row * NB + tid);
    }
    // column `row` of the block pattern: bit i = tile (i, row) exists (gathered from the rows' words)
    auto in_col = [&](int i) { return f.nz == nullptr || ((f.nz[(size_t)i * kNzWords + (row >> 6)] >> (row & 63)) & 1ull) != 0; };
    const bool linked = row + 1 < nblk && in_col(row + 1);
    // every tile of the column is final by the time the first block of the solution exists; their flags are looked
    // at once, 64 per poll (the sc1 copies: no mailbox verdicts), and the tiles then fetched without waiting
    for (int i0 = row + 1; i0 < nblk; i0 += 64) {
        const int i = i0 + tid;
        if (!flow_wait_lanes(f, w, tid < 64 && i < nblk && in_col(i) ? flow_tile_flag(f, i, row) : -1, -1)) return false;
    }
    auto fetch = [&](int i) { return flow_fetch_tile(f.Lmat + (size_t)(i * NB) * ld + row * NB, ld, p); };
    if (linked) {
        // M[m][r] = sum_{c >= m} inv(L)[c][m] L(row + 1, row)[r][c]: the link's matrix, long before the link
        flow_put_tile(fetch(row + 1), Xr, p);
        lds_barrier();
        const int m = tid & (NB - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (tid >> 5) + 8 * q;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int c = 0; c < NB; c += 2) {
                a0 = fma(c >= m ? Li[c][m] : 0.0, Xr[r][c], a0);
                a1 = fma(c + 1 >= m ? Li[c + 1][m] : 0.0, Xr[r][c + 1], a1);
            }
            Mx[m][r] = a0 + a1;
        }
        lds_barrier();
    }
    // the tiles above the link, as their blocks of the solution come in: the next tile is on its way while this one's
    // x is awaited
    int cur = 0;
    // (the tiles of the column that exist, from the bottom up; -1: none left above the link)
    auto below = [&](int i) { for (--i; i > row + 1; --i) if (in_col(i)) return i; return -1; };
    int inext = below(nblk);
    if (inext >= 0) flow_put_tile(fetch(inext), Xr, p);
    while (inext >= 0) {
        const int i = inext;
        inext = below(i);
        FlowLds T = cur ? Xc : Xr;
        FlowTile next;
        const bool more = inext >= 0;
        if (more) next = fetch(inext);
        flow_wait_x(f, w, i, xv, true);
        if (!flow_how(w, 0)) return false;
        if (tid < 64) {
            // both halves of wave 0: lane c + 32 h adds the products of rows 16 h .. 16 h + 15
            const int c = tid & (NB - 1), r0 = (tid >> 5) * (NB / 2);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int r = 0; r < NB / 2; r += 4) {
                a0 = fma(T[r0 + r][c], xv[r0 + r], a0);
                a1 = fma(T[r0 + r + 1][c], xv[r0 + r + 1], a1);
                a2 = fma(T[r0 + r + 2][c], xv[r0 + r + 2], a2);
                a3 = fma(T[r0 + r + 3][c], xv[r0 + r + 3], a3);
            }
            const double part = (a0 + a1) + (a2 + a3);
            const double tot = part + __shfl_xor(part, 32);
            if (tid < NB) sv[tid] -= tot;
        }
        if (more) flow_put_tile(next, cur ? Xr : Xc, p);
        lds_barrier();
        cur ^= 1;
    }
    lds_barrier();
    // u = inv(L)^T s, then the link: one wave
    double u = 0.0;
    if (tid < NB) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int c = 0; c < NB; c += 2) {
            a0 = fma(c >= tid ? Li[c][tid] : 0.0, sv[c], a0);
            a1 = fma(c + 1 >= tid ? Li[c + 1][tid] : 0.0, sv[c + 1], a1);
        }
        u = a0 + a1;
    }
    if (linked) flow_wait_x(f, w, row + 1, xv, false);
    if (tid < 64) {
        // both halves of the wave: lane m + 32 h sums the products of entries 16 h .. 16 h + 15
        const bool ok = !linked || flow_how(w, 0) != 0;
        const int m = tid & (NB - 1), r0 = (tid >> 5) * (NB / 2);
        double v = u;
        if (linked) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int r = 0; r < NB / 2; r += 4) {
                a0 = fma(Mx[m][r0 + r], xv[r0 + r], a0);
                a1 = fma(Mx[m][r0 + r + 1], xv[r0 + r + 1], a1);
                a2 = fma(Mx[m][r0 + r + 2], xv[r0 + r + 2], a2);
                a3 = fma(Mx[m][r0 + r + 3], xv[r0 + r + 3], a3);
            }
            const double part = (a0 + a1) + (a2 + a3);
            v = u - (part + __shfl_xor(part, 32));
        }
        if (ok && tid < NB) {
            const flow_u4 gp = flow_granule(v, my_tag), gs = flow_granule(v, f.epoch);
            asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(flow_x_granules(f, row, 0) + tid), "v"(gp) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(flow_x_granules(f, row, 1) + tid), "v"(gs) : "memory");
            f.x[(size_t)row * NB + tid] = v;         // for the kernels behind this launch
        }
    }
    lds_barrier();
    return !linked || flow_how(w, 0) != 0;
}


// Two workgroups per CU (256 registers per lane): at three (168) the kernel spilled -- into scratch memory inside the
// diagonal factor, of all places -- and since a workgroup works through a list of tiles the grid no longer needs the room.
__global__ __launch_bounds__(256, 2) void
chol_flow_kernel(CholFlow f)
{
    if (f.lm && (f.lm->stop || f.lm->lin_failed || f.lm->flow_aborted)) return;
    __shared__ __attribute__((aligned(16))) double Xr[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Xc[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Li[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Mt[NB][NB + 1];
    __shared__ int lds_how[16];
    __shared__ int lds_word;
    __shared__ int lds_prog[2];
    __shared__ double sv[NB], xv[NB];
    const int nblk = f.nblk;
    const FlowPos p = flow_pos();
    FlowWaiter w;
    w.how = lds_how; w.word = &lds_word; w.prog = lds_prog; w.phase = 0; w.pending_flag = -1;
    if (threadIdx.x == 0) { lds_prog[0] = 0; lds_prog[1] = 0; }     // (a barrier stands between this and every use)

    // workgroup -> role: blocks 0, 8, 16, ... are the D's (one XCD under round-robin placement), the others the P's
    const int b = blockIdx.x;
    const bool is_d = (b & 7) == 0 && (b >> 3) < f.num_d;
    if (!is_d) {
        const int pidx = b - min((b + 7) >> 3, f.num_d);          // blocks below b that are not D's
        // column j of the list holds rows j + kFlowW + 1 .. nblk: nblk - kFlowW - j tiles
        int j = 0, base = 0;
        for (int t = pidx; t < f.num_tiles; t += f.num_p) {
            int ti, tj;
            if (f.ptiles) { const int32_t e = f.ptiles[t]; ti = e >> 16; tj = e & 0xffff; }
            else {
                while (t - base >= nblk - kFlowW - j) { base += nblk - kFlowW - j; ++j; }
                ti = j + kFlowW + 1 + (t - base); tj = j;
            }
            if (!flow_p_tile(f, w, p, ti, tj, Xr, Xc, Li, Mt)) return;
            lds_barrier();
        }
        return;
    }
    const int d = b >> 3;
    int last = -1;
    for (int row = d; row <= nblk; row += f.num_d) {
        if (!flow_d_forward(f, w, p, row, Xr, Xc, Li, Mt)) return;
        last = row;
        lds_barrier();
    }
    if (f.x == nullptr) return;
    for (int row = last; row >= 0; row -= f.num_d) {
        if (row >= nblk) continue;                                  // the right-hand side's row has no unknowns
        if (!flow_d_backward(f, w, p, row, row == last, Xr, Xc, Li, Mt, sv, xv)) return;
    }
}

// diagnostics: where the D workgroups of the most recent factorisation spent their time
static long long *g_flow_trace = nullptr;
constexpr size_t kFlowTraceBytes = (size_t)(kFlowMaxBlocks + 1) * kFlowTraceStride * 8;
long long *chol_flow_trace_buffer(int enable)
{
    if (enable && !g_flow_trace) { if (hipMalloc(reinterpret_cast<void **>(&g_flow_trace), kFlowTraceBytes) != hipSuccess) g_flow_trace = nullptr; else (void)hipMemset(g_flow_trace, 0, kFlowTraceBytes); }
    if (!enable && g_flow_trace) { (void)hipFree(g_flow_trace); g_flow_trace = nullptr; }
    return g_flow_trace;
}

// test hook (osfm_ba_debug_flow_spin_limit): polls a wait makes before it gives the launch up; <= 0: the default
static std::atomic<int> g_flow_spin_limit{kFlowSpinLimitDefault};
void chol_flow_set_spin_limit(int limit) { g_flow_spin_limit.store(limit > 0 ? limit : kFlowSpinLimitDefault); }

// residency check of the flow kernel, per device: workgroups that can be resident at once
int chol_flow_capacity()
{
    static int cap[64];
    static std::once_flag once[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::call_once(once[dev], [dev]() {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_flow_kernel, 256, 0) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess) { cap[dev] = 0; return; }
        // The query can promise one block per CU more than the hardware admits when the SGPR
        // count is what limits (MI355X_MICROARCH.md, residency: 7-8 blocks per CU); this kernel
        // is bound by its VGPRs and LDS (3-4 per CU), far from that.  A workgroup that is not resident
        // would be a hang, so: two fewer per CU where the answer is in the doubtful range, and a
        // tenth of the chip left free on top.
        if (per_cu >= 6) per_cu -= 2;
        cap[dev] = (int)(0.9 * per_cu * prop.multiProcessorCount);
    });
    return cap[dev];
}

int chol_flow_flag_count(int n)
{
    const int nblk = cholesky_padded_dim(n) / NB;
    // flags, then (16-byte aligned) the granules of the solution: 2 copies x 32 entries x 4 words per block
    return (nblk + 2) * nblk + (nblk + 1) * (kFlowW + 1) + 1 + 2 * nblk + 16 + nblk * 2 * NB * 4;
}

size_t chol_flow_mailbox_bytes(int n)
{
    const int nblk = cholesky_padded_dim(n) / NB;
    return ((size_t)(nblk + 1) * (kFlowW + 1) * NB * NB + (size_t)nblk * NB) * sizeof(double);
}

// The right-hand side's last block: y_k = b_k inv(L_kk)^T for k = nblk - 1 (every earlier
// block of the row is solved by the step that finishes its panel column).
__global__ __launch_bounds__(64) void
chol_rhs_tail_kernel(const double *A, double *Lout, int ld, int nblk, const double *Ldiag, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed || lm->flow_aborted)) return;
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

__global__ __launch_bounds__(1024) void
chol_backsolve_kernel(const double *A, int ld, int nblk, int n, const double *Ldiag, double *x, const LmDev *lm)
{
    if (lm && (lm->stop || lm->lin_failed || lm->flow_aborted)) return;
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

__global__ void
chol_padding_diagonal_kernel(double *S, int ld, const int32_t *__restrict__ pad, int npad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) S[(size_t)pad[i] * ld + pad[i]] = 1.0;
}

void launch_padding_diagonal(double *S, int ld, const int32_t *pad, int npad, hipStream_t s)
{
    if (npad > 0) hipLaunchKernelGGL(chol_padding_diagonal_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, S, ld, pad, npad);
}

void launch_small_solve(const double *A, int n, double *Ldiag, double *x, int *info, const BaDev &d,
    double *partials_cam, hipStream_t s)
{
    hipLaunchKernelGGL(chol_small_kernel, dim3(1), dim3(128), 0, s, A, NB, n, Ldiag, x, info, d, partials_cam);
}

// A: (N + 32) x N row-major, rows/cols >= n padded with identity, rhs in row N.
// Ldiag: N * 32 doubles of scratch for the inverses of the factored diagonal blocks.
// Returns the form that was launched: 1 the one-launch flow form, 0 launch per block column.
int launch_cholesky_solve(double *A, double *Lmat, int n, double *Ldiag, double *x, int *info, const LmDev *lm, hipStream_t s,
    int *flow_flags, int flow_epoch, double *flow_mailbox, FlowPattern pattern)
{
    static_assert(kFlowBand == kFlowW && kFlowOrderMaxBlocks == kFlowMaxBlocks, "ba_kernels.h mirrors these");
    const int N = cholesky_padded_dim(n);
    const int nblk = N / NB;
    // The D's sit at the blocks 0, 8, 16, ...; the P workgroups fill the blocks between and behind them.  One P
    // workgroup per tile (rows more than kFlowW below the diagonal, the right-hand side row included) while that
    // fits the device, else the tiles are dealt round robin to as many as do fit.
    int num_tiles = 0;
    for (int j = 0; j < nblk; ++j) num_tiles += std::max(nblk - kFlowW - j, 0);
    if (pattern.ptiles) num_tiles = pattern.num_ptiles;
    static const int exp_max_d = getenv("OSFM_FLOW_MAX_D") ? atoi(getenv("OSFM_FLOW_MAX_D")) : kFlowMaxD;
    static const int exp_max_groups = getenv("OSFM_FLOW_MAX_GROUPS") ? atoi(getenv("OSFM_FLOW_MAX_GROUPS")) : 1 << 30;
    const int num_d = std::min(nblk + 1, exp_max_d);
    const int d_span = 8 * (num_d - 1) + 1;                              // through the last D
    auto d_below = [&](int g) { return std::min((g + 7) >> 3, num_d); };  // D blocks among the first g
    int groups = d_span;
    while (groups - d_below(groups) < num_tiles) ++groups;
    const int cap = flow_flags && flow_mailbox ? chol_flow_capacity() : 0;
    groups = std::min(std::min(groups, cap), std::max(exp_max_groups, d_span + 1));
    const int num_p = groups - d_below(groups);
    if (flow_flags && flow_mailbox && nblk >= 2 && nblk <= kFlowMaxBlocks && groups >= d_span && (num_tiles == 0 || num_p >= 1)) {
        CholFlow f;
        f.mailbox = flow_mailbox;
        f.A = A; f.Lmat = Lmat; f.Ldiag = Ldiag; f.flags = flow_flags; f.info = info; f.lm = lm;
        f.ld = N; f.nblk = nblk; f.epoch = flow_epoch; f.trace = g_flow_trace;
        f.num_d = num_d; f.num_p = std::max(num_p, 1); f.num_tiles = num_tiles;
        f.nz = pattern.nz; f.ptiles = pattern.ptiles;
        f.spin_limit = g_flow_spin_limit.load();
        // the backward substitution runs inside the same launch; x has room for the padded system (N entries)
        f.x = getenv("OSFM_BA_FLOW_FACTOR_ONLY") ? nullptr : x;
        {
            // Two of these launches must never share the device: each needs ALL its workgroups resident, and two
            // half-resident grids would wait for each other until the spin limit fails both.  Solves on different
            // streams (host threads) are therefore chained on the device: a launch waits for the previous one's event.
            // (Launches of OTHER processes, or a foreign kernel that holds CUs for long, are not covered: the spin
            //  limit turns that into an aborted launch, which the caller repeats in the launch-per-column form.)
            static std::mutex flow_mu[64];
            static hipEvent_t flow_ev[64];
            int dev = 0;
            (void)hipGetDevice(&dev);
            dev = std::min(std::max(dev, 0), 63);
            std::lock_guard<std::mutex> lock(flow_mu[dev]);
            if (flow_ev[dev]) (void)hipStreamWaitEvent(s, flow_ev[dev], 0);
            else (void)hipEventCreateWithFlags(&flow_ev[dev], hipEventDisableTiming);
            hipLaunchKernelGGL(chol_flow_kernel, dim3(groups), dim3(256), 0, s, f);
            if (flow_ev[dev]) (void)hipEventRecord(flow_ev[dev], s);
        }
        if (!f.x)
            hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(1024), (size_t)2 * N * sizeof(double), s, Lmat, N,
                nblk, n, Ldiag, x, lm);
        return 1;
    }
    hipLaunchKernelGGL(chol_first_kernel, dim3(1), dim3(128), 0, s, A, N, Ldiag, info, lm);
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
    return 0;
}

}  // namespace osfm
