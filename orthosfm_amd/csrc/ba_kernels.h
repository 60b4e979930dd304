// Launchers and argument blocks of the bundle-adjustment kernels.
#pragma once
#include "ba_device.h"
#include "osfm_common.h"

namespace osfm {

enum { kPassNormal = 0, kPassScaleInit = 1 };

// Per-observation record written by the point pass and consumed by the pair,
// gradient and back-substitution passes of the same linearisation:
//   Jc[2][6] (Huber-corrected, Jacobi-scaled camera block, unused columns 0),
//   Jp[2][3], Q[2][3] = Jp (V_j + D^2)^-1, r[2]
constexpr int kObsRec = 26;
// Jp, Jc, Q, r: [0, 18) is what the b side of a Schur pair and the back pass read, [6, 24) what
// the a side of an off-diagonal pair reads (18 of the 26 doubles each)
constexpr int kRecJp = 0, kRecJc = 6, kRecQ = 18, kRecR = 24;


// The per-point kernels (point / back / cost pass) give every observation a lane: a workgroup of 256 threads takes
// the tracks whose first observation lies in a window of kWinObs consecutive observations -- at most 256
// observations unless a track runs more than 256 - kWinObs past the window's end; those windows (listed once per
// solve) go to kernels that give every track a wave instead.  Their partials: one slot per window.
constexpr int kWinObs = 224;
struct WinDesc { int32_t jf, jn, ka, kb; };     // tracks [jf, jn), observations [ka, kb)
struct ObsWindows {
    const WinDesc *desc;          // [num]
    const int32_t *over_list;     // [num_over] windows of more than 256 observations
    const int32_t *ok_list;       // [num - num_over] the others (launching a workgroup per window that returns at once
                                  // took 80 us where every window is of the first kind)
    const int32_t *obs_lay;       // [O] cam_off | cam_ldim << 24 of the observation's camera
    int32_t num, num_over;
};
// ---- ba_dense.hip: the Schur complement's point part as a dense product (dense visibility) ----
int schur_dense_rows(int nc);            // rows of Zm / Wm (nc rounded up to the tile)
int schur_dense_cols(int M);             // their leading dimension (3 M rounded up to the K step)
bool schur_dense_wins(int nc, int M, int64_t entries);      // cost model: the product against `entries` list entries
size_t schur_dense_partial_bytes(int nc, int M);            // split-K partials (16 when the product is not split)
void launch_schur_dense(const BaDev &d, const double *obsrec, const int32_t *obs_lay, double *Zm, double *Wm, double *partial,
    double *S, int ldS, bool first, hipStream_t s);

int obs_windows_count(int O);
// desc / over_list / ok_list: [num], counts[0] (over) and counts[1] (ok) zeroed by the caller, obs_lay: [O]
void launch_obs_windows(const BaDev &d, int num, WinDesc *desc, int32_t *over_list, int32_t *ok_list, int32_t *counts, int32_t *obs_lay,
    hipStream_t s);

struct PointPassArgs {
    int mode;                 // kPassScaleInit: only derive the Jacobi scaling
    int update_diag;          // recompute the LM diagonal (reuse_diagonal == false)
    int want_gradient;        // also reduce |Plus(x,-g) - x|_inf over the points
    double radius, min_diag, max_diag;
    double *diag_p;           // [3M] clamped diag(J^T J) of the point columns
    double *vinv;             // [9M] (V_j + D^2)^-1
    double *ge;               // [3M] Jp^T r
    double *scale_p_out;      // [3M] (scale-init mode)
    double *partials;         // [3][blocks]: cost, gradient max, not-PD flag
    double *obsrec;           // [O][kObsRec]: per-observation blocks of this linearisation
};

// Tolerances and limits of the LM loop (ceres::Solver::Options as set in
// bundle_adjustment.cpp:126-133 plus the defaults it leaves alone).
struct LmParams {
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double min_relative_decrease, max_radius, min_radius;
    int32_t max_iterations, max_invalid_steps;
};

// scratch the decision kernels reduce: per-block partials of the passes
struct LmScratch {
    const double *partA;     // [3][blocksM]  point pass: cost, gradient max, not-PD flag
    const double *partB;     // [3][blocksM]  back pass: model cost change, |dx|^2, |x|^2
    const double *partC;     // [blocksM]     cost pass: candidate cost
    const double *part_cam;  // [C][2]        camera update: |dx|^2, |x|^2
    const double *gmax_cam;  // [C]           camera gradient norms
    int32_t *chol_info;      // [1]           first non-positive pivot + 1 (reset after reading)
    int32_t blocksM, C;
    double *reset_S;         // one-block systems: the decide kernel clears S ((reset_N + 32) x reset_N,
    int32_t reset_n, reset_N;    // identity on the padding diagonal from reset_n on); null otherwise
};

// What the last workgroup of a launch needs to run the LM control in its tail (ba_lm_decide behind the back pass,
// ba_lm_post behind the pair pass) instead of a launch of one workgroup each: 6-14 us of kernel plus a launch gap,
// twice per iteration.  Every workgroup leaves its partials write-through and takes a ticket; the one that takes
// the last runs the fixed-order reductions over all of them, so the result does not depend on which one that is.
struct LmTail {
    LmDev *lm;
    LmParams prm;
    LmScratch sc;
    LmDev *host_out;          // page-locked host slot that receives the state (may be null)
    int32_t *ticket;          // lm_ticket_bytes() of them, zero between launches (sharded counter)
    int32_t initial;          // post: the linearisation in front of the first iteration
    int32_t enabled;
};

constexpr int kPairChunk = 512;   // entries of a camera pair's list per pair-pass wave; kPairChunkSmall below
constexpr int kPairChunkSmall = 256;       // kPairChunkSmallLimit entries (few cameras: more, shorter waves)
constexpr int kPairChunkSmallLimit = 1 << 20;
constexpr int kPairChunkTiny = 256;        // below kPairChunkTinyLimit entries (the 3-camera adjustments)
constexpr int kPairChunkTinyLimit = 1 << 16;
constexpr int kPairSums = 54;     // sums a pair-pass wave leaves per chunk: 6x6 block, 6 diagonal, 6 rhs, 6 gradient

// What a wave of the pair pass needs before its first entry: found through chunk_pair -> chunk_start /
// pair_start / pair_key it was three dependent memory latencies in front of the two the entries and
// their records cost anyway.  nchunks == 0: no work (the tail of the launch).
struct PairChunkDesc { int32_t pi, e0, e1, nchunks, c1, c2, first, pad1; };     // first: the pair's first chunk (= wave)

struct PairPassArgs {
    int mode, update_diag, want_gradient;
    double radius, min_diag, max_diag;
    int num_pairs;
    const uint32_t *pair_key;        // [num_pairs] c1 * C + c2, c1 >= c2
    const int32_t *pair_start;       // [num_pairs + 1]
    const uint64_t *entries;         // (obs a << 32) | obs b, grouped by pair, track order inside
    const int32_t *chunk_start;      // [num_pairs + 1] first chunk (= wave) of each pair
    const int32_t *chunk_pair;       // [chunk_start[num_pairs]] pair of each chunk
    const struct PairChunkDesc *chunk_desc;   // [max_chunks] everything a pair-pass wave needs to start, in ONE load
    int max_chunks;                  // waves to launch (upper bound of chunk_start[num_pairs])
    int chunk;                       // entries per chunk
    double *chunk_partials;          // [max_chunks][kPairSums] sums of the chunks of multi-chunk pairs
    int32_t *pair_ticket;            // [num_pairs] zero between launches: chunks of a multi-chunk pair that have left their sums
    double *gmax_out;                // [C] camera gradient norms (diagonal pairs; may be null)
    const double *vinv, *ge;
    const double *obsrec;     // [O][kObsRec]
    double *diag_c;           // [nc]
    double *scale_c_out;      // [nc] (scale-init mode)
    double *S;                // [.][ldS] row-major, lower triangle blocks written
    int ldS;
    double *rhs;              // [nc]
    LmTail post;              // enabled: the last workgroup runs ba_lm_post
    int dense;                // the lists hold the diagonal pairs only and -Z W^T is in S already (launch_schur_dense): blocks are ADDED
};

struct BackPassArgs {
    const double *y_c;        // [nc] solution of the reduced system
    const double *vinv, *ge;
    const double *obsrec;     // [O][kObsRec]
    double *points_out;       // [M][4] candidate points
    double *partials;         // [3][blocks]: model cost change, |dx|^2, |x|^2
    // fused form (LM solve): the launch also evaluates the candidate's cost from the
    // table rows ba_cam_update / chol_small left for the candidate cameras (the cost pass: a launch and a pass over
    // the observations less) and, in its last workgroup, decides
    int fused;
    double *cost_partials;    // [blocks]
    LmTail decide;
};

// host_out (may be null): page-locked host slot that receives the state the kernel leaves
void launch_lm_decide(LmDev *lm, const LmParams &prm, const LmScratch &sc, LmDev *host_out, hipStream_t s);
void launch_lm_post(LmDev *lm, const LmParams &prm, const LmScratch &sc, int initial, LmDev *host_out, hipStream_t s);
void launch_lm_clear_abort(LmDev *lm, hipStream_t s);

void launch_point_pass(const BaDev &d, const PointPassArgs &a, const ObsWindows &w, hipStream_t s);
void launch_pair_pass(const BaDev &d, const PairPassArgs &a, hipStream_t s);
// candidate cameras, their derived table rows (ba_device.h: cam_derive_part) and their share of the step norms;
// cams_out / table_out == nullptr (LM solve): into the iterate buffers that are not current
void launch_cam_update(const BaDev &d, const double *y_c, double *cams_out, double *table_out, double *partials_cam, hipStream_t s);
// the table rows of the cameras at `cams`
void launch_cam_derive(const BaDev &d, const double *cams, double *table_out, hipStream_t s);
void launch_back_pass(const BaDev &d, const BackPassArgs &a, const ObsWindows &w, hipStream_t s);
size_t lm_ticket_bytes();
// table: derived table rows of the cameras to evaluate at (ignored in an LM solve: the candidate's)
void launch_cost_pass(const BaDev &d, const double *table, const double *points, double *partials,
    const ObsWindows &w, hipStream_t s);
void launch_reduce(const double *partials, int n, int num_slots, unsigned max_mask, double *out,
    const double *extra, int extra_n, int extra_stride, int extra_slots, hipStream_t s);
void launch_max_reduce(const double *v, int n, double *out, hipStream_t s);
void launch_fill(double *v, size_t n, double value, hipStream_t s);
void launch_expand_points(const int32_t *pt_start, int M, int32_t *obs_pt, hipStream_t s);
void launch_reproj(const BaDev &d, double *err, double *residuals, hipStream_t s);
void launch_triangulate(const BaDev &d, double *points_out, uint8_t *valid, hipStream_t s);

// camera-pair lists built on the device (ba_pairs.hip)
struct PairListsDev {
    PooledBuffer counts, offsets, keys_in, keys, vals_in, entries, unique, runs, starts, scalars, temp;
    PooledBuffer chunk_start, chunk_pair, chunk_partials, chunk_desc, pair_ticket;
    int num_pairs = 0;
    bool dense = false;          // the lists hold the diagonal pairs only: the rest is ba_dense.hip's product
    int num_entries_all = 0;     // entries of the complete lists (what was counted before the choice)
    int num_entries = 0;
    int max_chunks = 0;
    int chunk = kPairChunk;
    int group = 1;               // the unique keys are pair_key_of(ca, cb, C, group)
    ~PairListsDev()
    {
        PooledBuffer *b[] = {&counts, &offsets, &keys_in, &keys, &vals_in, &entries, &unique, &runs, &starts, &scalars, &temp,
                             &chunk_start, &chunk_pair, &chunk_partials, &chunk_desc, &pair_ticket};
        for (auto *x : b) x->release();
    }
};
// dense_policy: -1 the cost model decides (schur_dense_wins), 0 never, 1 always (with points)
int pair_lists_build(const BaDev &d, bool with_points, int64_t max_entries, PairListsDev *out, hipStream_t s, int dense_policy = -1);

// dense Cholesky solve of the reduced camera system (ba_cholesky.hip)
int cholesky_padded_dim(int n);
// lm != nullptr: nothing happens when the solve has stopped or the last linearisation failed
// Lmat: (N + 32) x N scratch for the factor and the solved right-hand side (A keeps the
// reduced system's trailing updates)
// flow_flags (may be null: launch-per-column form): chol_flow_flag_count(n) ints, zeroed ONCE; flow_epoch: a
// value > 0 that differs from every earlier call on the same flags (the hand-off flags of the one-launch form)
// Returns 1 when the one-launch form ran, 0 for the launch-per-column form.  The one-launch form reports a launch it
// had to give up (a wait that outlasted its spin limit: not all workgroups were resident) as info >= kFlowAborted.
constexpr int kFlowAborted = 1 << 20;
// Block pattern of the factor for the one-launch form (ba_order.hip): nz[(nblk + 1)][kNzWords] bit k of row i = tile
// (i, k) of L is structurally nonzero (fill included; row nblk: the right-hand side); ptiles: the nonzero tiles more
// than kFlowBand below the diagonal as i << 16 | j, column-major.  Null: every tile is taken as nonzero.
constexpr int kNzWords = 3;
constexpr int kFlowBand = 3;
constexpr int kFlowOrderMaxBlocks = 160;
struct FlowPattern { const unsigned long long *nz = nullptr; const int32_t *ptiles = nullptr; int num_ptiles = 0; };
int launch_cholesky_solve(double *A, double *Lmat, int n, double *Ldiag, double *x, int *info, const LmDev *lm, hipStream_t s,
    int *flow_flags = nullptr, int flow_epoch = 0, double *flow_mailbox = nullptr, FlowPattern pattern = FlowPattern());
// S[i][i] = 1 for the listed unknowns (interior padding of an ordered layout: identity rows)
void launch_padding_diagonal(double *S, int ld, const int32_t *pad, int npad, hipStream_t s);
void chol_flow_set_spin_limit(int limit);   // test hook: polls before a wait gives the launch up (<= 0: default)
int chol_flow_flag_count(int n);
long long *chol_flow_trace_buffer(int enable);     // diagnostics, see osfm_ba_debug_chol_trace
size_t chol_flow_mailbox_bytes(int n);     // flow_mailbox: scratch of that size (no initialisation needed)
// n <= 32 (one block): factor, both substitutions and the candidate cameras in one launch
void launch_small_solve(const double *A, int n, double *Ldiag, double *x, int *info, const BaDev &d,
    double *partials_cam, hipStream_t s);
// zeroes the reduced system and puts the identity on its padding diagonal
void launch_reset_system(double *S, size_t elems, int ld, int n, int N, hipStream_t s);

}  // namespace osfm
