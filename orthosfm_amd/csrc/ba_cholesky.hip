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

// Workgroup barrier that only orders LDS traffic.  __syncthreads() also drains the
// vector-memory counter, which would expose the latency of every prefetch and of every
// result store (2-3 us for a write-through store) once per step of a chain.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (store_sc1 / load_sc1: ba_device.h)

template <bool SC1 = false>
__device__ __forceinline__ void
factor_diag_block(const double *Msrc, int lds_ld, double (*LsT)[NB], int kblock, double *Ldiag, int *info,
    double *linv_lds = nullptr, int linv_ld = 0, int nvalid = NB, bool ldiag_is_block = false, long long *dbg = nullptr)
{
    const long long dbg_t0 = dbg ? (long long)clock64() : 0;
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
    // (Measured and not kept: the elimination in LDL^T form on unscaled columns -- the column goes
    //  to LDS and lane j+1's entry comes by v_readlane before anything is computed from the pivot,
    //  the link is d -> 1/d (v_rcp_f64 + two Newton steps) -> t -> fma -> v_readlane, the Cholesky
    //  scale 1/sqrt(d) applied off the chain: 370 instead of 410 cycles per pivot, but the 32
    //  square-root iterations pile up behind the loop and the block takes 15.5k cycles instead of
    //  14.3k.  One wave issues in order: a pivot costs the SUM of its chain stalls, its ~15 trailing
    //  FMAs and its LDS reads, not the longest of them.)
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
    if (dbg && (threadIdx.x & 63) == 0) dbg[0] = (long long)clock64() - dbg_t0;
    if (bad && lane == 0) atomicMax(info, kblock * NB + bad);
    // L itself, transposed (LsT[c][i] = L[i][c], zero above the diagonal), for a caller that
    // wants to look at it: lane r writes element r of every row, consecutive addresses
    if (LsT && h == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) LsT[c][r] = c <= r ? v[c] : 0.0;
    }
    if (h == 1) {
        double *Lk = Ldiag + (ldiag_is_block ? 0 : (size_t)kblock * NB * NB);   // [i][j] = inv(L)[i][j]
#pragma unroll
        for (int i = 0; i < NB; ++i) { if (SC1) store_sc1(&Lk[i * NB + r], v[i]); else Lk[i * NB + r] = v[i]; }
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
    if (d.lm == nullptr || d.lm->lin_failed) return;      // this kernel writes the candidate cameras of an LM solve: no state, nothing to do
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
    double *cams_out = d.lm->cur ? d.cams2[0] : d.cams2[1];
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


// ---------------------------------------------------------------------------
// The whole factorisation (and the forward substitution of the right-hand side riding along)
// as ONE launch of persistent workgroups that hand tiles to each other: the launch-per-
// block-column form above spends 15.6 us per column of which 2.7 are the pivot chain.
//
// Every workgroup owns tiles of the lower triangle for the whole factorisation and keeps them
// in registers (right-looking updates as the panels of earlier columns appear):
//   D_r, r = 0 .. nblk: the last kFlowW + 1 tiles of block row r -- (r, r-kFlowW) .. (r, r) --,
//        i.e. the diagonal tile and its left neighbours.  For a column k it owns it turns its
//        tile into L_rk itself as soon as inv(L_kk) appears, applies it to its tiles right of k,
//        and after the last column factors its diagonal tile and publishes inv(L_rr).
//        D_nblk is the tail of the right-hand side row (no diagonal tile).
//   P_(i,j), i - j > kFlowW (the right-hand side row i = nblk included): updates, then
//        L_ij = tile * inv(L_jj)^T once that inverse appears.
// Why the band: POTRF(j) -> TRSM(j+1, j) -> SYRK -> POTRF(j+1) is the critical path, and row
// j+1 enters it with everything the columns before j did to it.  A tile handed from one
// workgroup to another costs a round trip through memory (sc1 store, drain, flag, poll, sc1
// load: ~5.5 us measured here, 2x the pivot chain of a block), so the hand-offs on that path
// must be few and the others need slack: with the diagonal tile alone per workgroup the loop
// inverse(j-1) -> P computes L(j+1, j-1) -> D_(j+1) updates paced the whole thing at 10 us per
// column.  With the band, a P tile's result is needed kFlowW columns after the inverse it
// waited for.
// The D workgroups talk to each other through their XCD's L2 where they can: the blocks with
// blockIdx % 8 == 0 are the D's (observed placement: round robin over the XCDs -- speed only),
// every payload is published twice -- plain stores into a mailbox + a flag that carries the
// producer's XCC id (s_getreg HW_REG_XCC_ID), then sc1 stores into its place in the factor + a
// second flag --, and a D reads the mailbox only when the flag says the producer sits on its
// own XCD (one L2: the plain stores are there once their vmcnt has drained; the reads bypass
// L1).  Everybody else, and a D on another XCD, takes the sc1 copy as MI355X_MICROARCH.md
// ("Valid forms") prescribes: every byte stored sc1 and loaded sc1, each storing wave drains
// its stores before one lane stores the flag behind a workgroup barrier, one lane polls, the
// others load behind the barrier it joins.  Flags hold the launch's epoch (no reset between
// factorisations).  All workgroups must be resident (checked against the occupancy query,
// else the launch-per-column form runs); a poll that outlasts kFlowSpinLimit raises the abort
// word, which every poll loop watches, and the factorisation is reported as failed (info)
// instead of hanging the device.
// ---------------------------------------------------------------------------
constexpr int kFlowSpinLimit = 1 << 21;
constexpr int kFlowW = 3;                 // left neighbours of the diagonal tile a D workgroup owns

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
    long long *trace;     // diagnostics (tools/chol_flow_trace.py): [nblk + 1][16] wall_clock64 stamps of the D's, or null
};

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

// plain store / flag for a reader behind the same L2 (no cache-policy bits)
__device__ __forceinline__ void store_flag_plain(int *p, int v)
{
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

struct FlowWaiter {
    int *lds_word;        // verdict: 0 aborted, 1 sc1 copy, 2 mailbox copy
    int pending_flag;     // sc1 flag to store once this workgroup's sc1 stores have drained (-1: none)
};

// Waits until the payload behind (slow_flag) or -- D to D only -- (box_flag) is published.
// One lane polls; every thread drains its own stores first (a pending sc1 publication of this
// workgroup becomes visible here, for free: the waves would idle at the barrier anyway).
__device__ __forceinline__ int flow_wait(const CholFlow &f, FlowWaiter &w, int slow_flag, int box_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w.pending_flag >= 0) {
        lds_barrier();
        if (threadIdx.x == 0) __hip_atomic_store(f.flags + w.pending_flag, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w.pending_flag = -1;
    }
    if (threadIdx.x == 0) {
        int verdict = 0;
        const int *ps = f.flags + slow_flag, *pb = f.flags + (box_flag >= 0 ? box_flag : slow_flag), *pab = f.flags + flow_abort_flag(f);
        const int want_box = f.epoch | ((flow_xcc() + 1) << 24);
        for (int spins = 0;; ++spins) {
            if (box_flag >= 0 && __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want_box) { verdict = 2; break; }
            if (__hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.epoch) { verdict = 1; break; }
            if ((spins & 15) == 15 && __hip_atomic_load(pab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.epoch) break;
            if (spins > kFlowSpinLimit) {
                __hip_atomic_store(f.flags + flow_abort_flag(f), f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!verdict) atomicMax(f.info, 1 << 20);      // reported as a failed factorisation
        *w.lds_word = verdict;
    }
    lds_barrier();
    return *w.lds_word;
}

// Thread <-> tile element map of the flow kernel: the accumulator layout of
// v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md: col = lane & 15, row = (lane >> 4) + 4 * reg),
// wave w holding the 16 x 16 quadrant (w >> 1, w & 1) of the 32 x 32 tile: element e of a thread
// is (tr0 + 4 e, tc).
typedef double v4d __attribute__((ext_vector_type(4)));
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

// a published 32 x 32 tile (row-major, leading dimension ld) into LDS; sc1 loads (L1 bypass)
__device__ __forceinline__ void flow_load_tile(const double *G, int ld, double (*dst)[NB + 1], const FlowPos &p)
{
    double v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = load_sc1(G + (size_t)(p.tr0 + 4 * e) * ld + p.tc);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[p.tr0 + 4 * e][p.tc] = v[e];
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

__global__ __launch_bounds__(256, 3) void       // three workgroups per CU: the whole grid has to be resident
chol_flow_kernel(CholFlow f)
{
    if (f.lm && (f.lm->stop || f.lm->lin_failed)) return;
    __shared__ __attribute__((aligned(16))) double Xr[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Xc[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Li[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Mt[NB][NB + 1];
    __shared__ int lds_word;
    __shared__ double sv[NB], xv[NB];
    const int nblk = f.nblk, ld = f.ld;
    const int tid = threadIdx.x;
    const FlowPos p = flow_pos();
    FlowWaiter w;
    w.lds_word = &lds_word; w.pending_flag = -1;

    // workgroup -> role: blocks 0, 8, 16, ... are D_0, D_1, ... (one XCD under round-robin placement);
    // the others take the P tiles, column by column
    const int b = blockIdx.x;
    const bool is_d = (b & 7) == 0 && (b >> 3) <= nblk;
    if (!is_d) {
        // ---- P_(i,j), i - j > kFlowW -------------------------------------------------
        int idx = b - min((b + 7) >> 3, nblk + 1);          // P index: blocks below b that are not D's
        int j = 0;
        // column j holds rows j + kFlowW + 1 .. nblk: nblk - j - kFlowW tiles
        while (j < nblk && idx >= nblk - j - kFlowW) { idx -= max(nblk - j - kFlowW, 0); ++j; }
        if (j >= nblk || nblk - j - kFlowW <= 0) return;    // surplus block of the grid
        const int i = j + kFlowW + 1 + idx;
        double acc[4];
        const double *Aij = f.A + (size_t)(i * NB) * ld + j * NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = Aij[(size_t)(p.tr0 + 4 * e) * ld + p.tc];
        for (int k = 0; k < j; ++k) {
            if (!flow_wait(f, w, flow_tile_flag(f, i, k), -1)) return;
            flow_load_tile(f.Lmat + (size_t)(i * NB) * ld + k * NB, ld, Xr, p);
            if (!flow_wait(f, w, flow_tile_flag(f, j, k), -1)) return;
            flow_load_tile(f.Lmat + (size_t)(j * NB) * ld + k * NB, ld, Xc, p);
            lds_barrier();
            flow_mma<true>(acc, Xr, Xc, p);
        }
        if (!flow_wait(f, w, flow_inv_flag(f, j), -1)) return;
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
        return;
    }

    // ---- D_row ---------------------------------------------------------------------------
    const int row = b >> 3;
    const bool has_diag = row < nblk;
    const int lo = max(0, row - kFlowW), hi = min(row, nblk - 1);     // own columns
    const int my_tag = f.epoch | ((flow_xcc() + 1) << 24);
    auto stamp = [&](int slot) { if (f.trace && tid == 0) f.trace[row * 16 + slot] = (long long)wall_clock64(); };
    stamp(0);
    // (A warm-up pass of the factor on an identity block -- one copy of the code, run twice -- was
    //  measured and dropped: the instruction fetches are not what the factor waits for, 12k of its
    //  14k cycles are the pivot loop itself warm or cold, and the loop form cost registers: 256
    //  VGPRs + scratch, two workgroups per CU instead of three.)
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
    for (int k = 0; k < row; ++k) {
        // ---- L(row, k) into Xr ----
        if (k < lo) {
            if (!flow_wait(f, w, flow_tile_flag(f, row, k), -1)) return;
            flow_load_tile(f.Lmat + (size_t)(row * NB) * ld + k * NB, ld, Xr, p);
        } else {
            const int how = flow_wait(f, w, flow_inv_flag(f, k), flow_box_flag(f, k, 0));
            if (!how) return;
            if (k == row - 1 && f.trace && tid == 0) f.trace[row * 16 + 7] = how;
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
            if (c == row) {
                lds_barrier();
            } else {
                // L(c, k): a D's if c - k <= kFlowW, else a P tile
                const bool from_d = c - k <= kFlowW;
                const int how = flow_wait(f, w, flow_tile_flag(f, c, k), from_d ? flow_box_flag(f, c, c - k) : -1);
                if (!how) return;
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
        if (tid < 64) {
            // the critical path: the factor, then its inverse to the mailbox (D_(row+1) is polling)
            factor_diag_block<false>(&Mt[0][0], NB + 1, nullptr, row, flow_box(f, row, 0), f.info, &Li[0][0], NB + 1, NB, true,
                f.trace ? f.trace + row * 16 + 1 : nullptr);
            if (f.trace && tid == 0) f.trace[row * 16 + 2] = (long long)clock64() - c_start;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) store_flag_plain(f.flags + flow_box_flag(f, row, 0), my_tag);
            stamp(5);
            if (f.trace && tid == 0) f.trace[row * 16 + 6] = (long long)clock64() - c_start;
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
    if (!has_diag || f.x == nullptr) return;

    // ---- backward substitution, same launch: x_k = inv(L_kk)^T (y_k - sum_{i > k} L_ik^T x_i) ----------------
    // D_k owns block k of the solution.  It still holds inv(L_kk) in LDS; y_k is row 0 of the right-hand side
    // tile (nblk, k); the tiles L(i, k) of ITS column come in from i = nblk - 1 downwards, each fetched while
    // the workgroup waits for x_i (two LDS buffers), so that a step costs the hand-off of 32 doubles and two
    // 32 x 32 matrix-vector products -- the single-workgroup kernel it replaces walked the same chain at
    // 3.7 us per step with a trip to L2 for every block row.
    auto tile_into = [&](int i, double (*dst)[NB + 1]) -> bool {
        const bool from_d = i - row <= kFlowW;
        const int how = flow_wait(f, w, flow_tile_flag(f, i, row), from_d ? flow_box_flag(f, i, i - row) : -1);
        if (!how) return false;
        if (how == 2) flow_load_tile(flow_box(f, i, i - row), NB, dst, p);
        else flow_load_tile(f.Lmat + (size_t)(i * NB) * ld + row * NB, ld, dst, p);
        return true;
    };
    {
        const bool from_d = nblk - row <= kFlowW;
        const int how = flow_wait(f, w, flow_tile_flag(f, nblk, row), from_d ? flow_box_flag(f, nblk, nblk - row) : -1);
        if (!how) return;
        if (tid < NB) sv[tid] = load_sc1(how == 2 ? flow_box(f, nblk, nblk - row) + tid : f.Lmat + (size_t)(nblk * NB) * ld + row * NB + tid);
    }
    int cur = 0;
    if (nblk - 1 > row && !tile_into(nblk - 1, Xr)) return;
    for (int i = nblk - 1; i > row; --i) {
        double (*T)[NB + 1] = cur ? Xc : Xr;
        if (i - 1 > row && !tile_into(i - 1, cur ? Xr : Xc)) return;
        const int how = flow_wait(f, w, flow_x_flag(f, i), flow_xbox_flag(f, i));
        if (!how) return;
        if (tid < NB) xv[tid] = load_sc1(how == 2 ? flow_xbox(f, i) + tid : f.x + (size_t)i * NB + tid);
        lds_barrier();
        if (tid < NB) {
            double a = sv[tid];
#pragma unroll
            for (int r = 0; r < NB; ++r) a = fma(-T[r][tid], xv[r], a);
            sv[tid] = a;
        }
        lds_barrier();
        cur ^= 1;
    }
    lds_barrier();
    if (tid < NB) {
        // x_k[m] = sum_{c >= m} inv(L)[c][m] * s[c]
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) v += (c >= tid) ? Li[c][tid] * sv[c] : 0.0;
        flow_xbox(f, row)[tid] = v;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) store_flag_plain(f.flags + flow_xbox_flag(f, row), my_tag);
        store_sc1(f.x + (size_t)row * NB + tid, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(f.flags + flow_x_flag(f, row), f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// diagnostics: where the D workgroups of the most recent factorisation spent their time
static long long *g_flow_trace = nullptr;
long long *chol_flow_trace_buffer(int enable)
{
    if (enable && !g_flow_trace) { if (hipMalloc(reinterpret_cast<void **>(&g_flow_trace), 65 * 16 * 8) != hipSuccess) g_flow_trace = nullptr; else (void)hipMemset(g_flow_trace, 0, 65 * 16 * 8); }
    if (!enable && g_flow_trace) { (void)hipFree(g_flow_trace); g_flow_trace = nullptr; }
    return g_flow_trace;
}

// residency check of the flow kernel, per device: workgroups that can be resident at once
int chol_flow_capacity()
{
    static int cap[64];
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!known[dev]) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_flow_kernel, 256, 0) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess) { cap[dev] = 0; }
        else {
            // The query can promise one block per CU more than the hardware admits when the SGPR
            // count is what limits (MI355X_MICROARCH.md, residency: 7-8 blocks per CU); this kernel
            // is bound by its 164 VGPRs (3 per CU), far from that.  A workgroup that is not resident
            // would be a hang, so: two fewer per CU where the answer is in the doubtful range, and a
            // tenth of the chip left free on top.
            if (per_cu >= 6) per_cu -= 2;
            cap[dev] = (int)(0.9 * per_cu * prop.multiProcessorCount);
        }
        known[dev] = true;
    }
    return cap[dev];
}

int chol_flow_flag_count(int n)
{
    const int nblk = cholesky_padded_dim(n) / NB;
    return (nblk + 2) * nblk + (nblk + 1) * (kFlowW + 1) + 1 + 2 * nblk + 16;
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
void launch_cholesky_solve(double *A, double *Lmat, int n, double *Ldiag, double *x, int *info, const LmDev *lm, hipStream_t s,
    int *flow_flags, int flow_epoch, double *flow_mailbox)
{
    const int N = cholesky_padded_dim(n);
    const int nblk = N / NB;
    // the D's sit at the blocks 0, 8, 16, ...; the P tiles (rows more than kFlowW below the diagonal, the
    // right-hand side row included) fill the blocks between and behind them
    int num_p = 0;
    for (int j = 0; j < nblk; ++j) num_p += std::max(nblk - j - kFlowW, 0);
    int flow_groups = 8 * nblk + 1;                                   // through D_nblk at block 8 * nblk
    {
        const int between = 7 * nblk;                                   // non-D blocks below 8 * nblk
        if (num_p > between) flow_groups += num_p - between;
    }
    if (flow_flags && flow_mailbox && nblk >= 2 && nblk <= 64 && flow_groups <= chol_flow_capacity()) {
        CholFlow f;
        f.mailbox = flow_mailbox;
        f.A = A; f.Lmat = Lmat; f.Ldiag = Ldiag; f.flags = flow_flags; f.info = info; f.lm = lm;
        f.ld = N; f.nblk = nblk; f.epoch = flow_epoch; f.trace = g_flow_trace;
        // the backward substitution runs inside the same launch; x has room for the padded system (N entries)
        f.x = getenv("OSFM_BA_FLOW_FACTOR_ONLY") ? nullptr : x;
        {
            // Two of these launches must never share the device: each needs ALL its workgroups resident, and two
            // half-resident grids would wait for each other until the spin limit fails both.  Solves on different
            // streams (host threads) are therefore chained on the device: a launch waits for the previous one's event.
            static std::mutex flow_mu[64];
            static hipEvent_t flow_ev[64];
            int dev = 0;
            (void)hipGetDevice(&dev);
            dev = std::min(std::max(dev, 0), 63);
            std::lock_guard<std::mutex> lock(flow_mu[dev]);
            if (flow_ev[dev]) (void)hipStreamWaitEvent(s, flow_ev[dev], 0);
            else (void)hipEventCreateWithFlags(&flow_ev[dev], hipEventDisableTiming);
            hipLaunchKernelGGL(chol_flow_kernel, dim3(flow_groups), dim3(256), 0, s, f);
            if (flow_ev[dev]) (void)hipEventRecord(flow_ev[dev], s);
        }
        if (!f.x)
            hipLaunchKernelGGL(chol_backsolve_kernel, dim3(1), dim3(1024), (size_t)2 * N * sizeof(double), s, Lmat, N,
                nblk, n, Ldiag, x, lm);
        return;
    }
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
