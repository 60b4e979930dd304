// Device-side data structures and launchers of the matching path (hot path A).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osfm {

constexpr int kRowsPerBlock = 256;     // rows of set 1 per workgroup (4 waves x 64)
constexpr int kTileCols = 64;          // columns of set 2 per LDS tile
constexpr int kSegCols = 8192;         // columns per workgroup (256 frag tiles -> 8 index bits)
constexpr int kSegColsMax = 32768;     // ... of a correction-free problem (512 tiles -> 9 index bits beside 21-bit products)
constexpr int kCycleCols = 1024;       // one cycle of the correction-free tile loop: 16 tiles
constexpr int kKeyNone = -(1 << 30);   // "no candidate" key

// One two-way matching problem: rows = descriptors of set 1, columns = set 2.
struct MatchProblem {
    const int8_t *A;        // [n1 padded to 256][D] int8 (SIFT: value-128, SURF: value)
    const int8_t *B;        // [n2 padded to 256][D]
    const int32_t *corrA;   // per row: 128*sum(a') + 2^20 (SIFT) or 0 (SURF)
    const int32_t *corrB;
    int32_t n1, n2;         // true counts (already limited for low-res matching)
    int32_t nrb, nseg;      // row blocks, column segments
    int32_t n2stride;       // n2 rounded up to 1024 (a whole cycle of the correction-free tile loop)
    int32_t seg_cols;       // columns per segment: kSegCols, or up to kSegColsMax for correction-free problems
    int32_t block_start;    // first workgroup of this problem in the launch
    int64_t rowpart_off;    // into RowPart[]: [nseg][nrb*256]
    int64_t colpart_off;    // into ColPart[]: [nrb][n2stride]
    int32_t *m12;           // [n1] result (set 1 -> set 2), device
    int32_t *m21;           // [n2]
    int32_t out_off12;      // combine_results offset added to valid m12 entries
    int32_t out_off21;
    int32_t force_exact;    // every query goes through the wrap-exact scan
    // Row blocks [0, nrb_main) run the RAW path: A_raw holds the descriptor
    // values themselves (rows with a value > 127 replaced by padding rows) and
    // corrA_raw = 128 * sum(a), so that  ip = sum(a * b') + corrA_raw  needs no
    // per-column correction.  The rows that do not fit ("special rows", SIFT
    // only) are gathered into A_special (value - 128 form, with corrA_special)
    // and handled by the row blocks [nrb_main, nrb) with the keyed epilogue.
    int32_t nrb_main;
    const int8_t *A_raw;
    const int32_t *corrA_raw;
    const int8_t *A_special;
    const int32_t *corrA_special;
    const int32_t *special_map;    // [n_special] original row of a special slot
    const int32_t *special_slot;   // [n1] slot of a row or -1 (null when n_special == 0)
    int32_t n_special;
    // c0 != 0: every descriptor of set 2 fits int8 as it is (SURF always; SIFT
    // when the view has no value > 127, the normal case: MVE's SIFT bytes stay
    // below 128).  Then the RAW row blocks take B_raw as column operand as well,
    // ip = sum(a * b) needs no correction at all and the MFMA C operand is 0.
    int32_t c0;
    const int8_t *B_raw;
    // sp != 0 (SIFT, both views with few special descriptors): the tile kernel runs the
    // correction-free form on A_raw x B_raw for EVERY pair -- special descriptors are blank
    // (zero) rows / columns there, which score 0 and change nothing (the reference's state
    // starts at (0, 0)) -- and match_special_kernel scores the special descriptors of either
    // view against all descriptors of the other one exactly:
    //   side 0: special rows of set 1 (A_special) x all of set 2;  side 1: special rows of
    //   set 2 (B_special) x all of set 1.
    // sp_row_off[side]: RowPart[chunks of the other set][ns rounded up to 32] -- the complete
    //   (best stream, its maximum, second largest stream maximum) of every special descriptor per
    //   4096-candidate chunk;
    // sp_col_off[side]: int32[n of the other set] in the column buffer -- for every descriptor of the
    //   other set its largest inner product over the special descriptors of this side.
    int32_t sp;
    int32_t nsA, nsB;                  // special descriptors of set 1 / set 2
    const int8_t *B_special;           // [nsB padded to 256][128], value - 128 form
    const int32_t *corrB_special;
    const int32_t *special_map_B;      // [nsB] original column of a special slot
    const int32_t *special_slot_B;     // [n2] slot of a column or -1
    int64_t sp_row_off[2];
    int64_t sp_col_off[2];
    // sp_wide[side] != 0: this side has more than kSpSlots units of special descriptors and went through
    // match_special_wide_kernel: chunks of kSpWideChunk candidates, streams at a stride of 32 (RowPart.pad = 4)
    int32_t sp_wide[2];
    // cascade hashing mode (cashash_kernels.h): hash data of set 1 / set 2
    const void *cas_rec[2];          // CasRecord[n]: hash words + packed bucket ids
    const int32_t *cas_start[2];
    const int32_t *cas_items[2];
    int64_t cas_state_off[2];        // first query of direction 0 / 1 in the candidate-key scratch
};

struct RowPart { int32_t ip_best, idx_best, ip_second, pad; };
struct ColPart { int32_t key_best, key_second; };

// Bijective XCD-aware remap (device code only): consecutive logical blocks (which share the
// column descriptors of one pair) land on the same XCD / L2.
__device__ __forceinline__ int xcd_remap(int bid, int total)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// Accept / reject tables and clamps of one descriptor type.
struct LoweTable {
    const int32_t *reject_from;  // [32768]: reject iff d1 >= reject_from[d2/2]
    int32_t max_d1;              // reject iff d1 > max_d1 (distance threshold)
    int32_t is_signed;           // SURF (short) vs SIFT (unsigned short)
};

// A query whose result must be recomputed with the reference's 16-bit
// wrap-around arithmetic (nearest_neighbor.cc:75-84 and the T-typed state).
struct ExactItem { int32_t problem; int32_t dir; int32_t query; };

// One workgroup of match_special_kernel: the special descriptors of one side of a problem
// against one chunk of kSpChunk descriptors of the other set.
constexpr int kSpChunk = 4096;
// count > 0: `count` entries (problem[e], side[e]) with ONE unit of special rows each, all streaming the same
// view (count <= kSpSlots); count == 0: entry 0 alone, with as many units as it has (passes of kSpSlots).
constexpr int kSpSlots = 2;             // units of 32 special rows a workgroup carries through one pass over a chunk
struct SpecialJob { int32_t problem[4], side[4]; int32_t count, chunk, pad0, pad1; };
void launch_match_special(const MatchProblem *d_problems, const SpecialJob *d_jobs, int num_jobs,
    RowPart *sp_parts, int32_t *sp_col, hipStream_t s);
// Entries with more than kSpSlots units (more than 64 special descriptors on a side): one workgroup per
// (entry, chunk of kSpWideChunk candidates), the chunk staged through LDS once and shared by four waves
// that carry two units each (jobs: problem[0], side[0], chunk).
constexpr int kSpWideChunk = 1024;
constexpr int kSpWideUnits = 8;         // units per pass of a workgroup
void launch_match_special_wide(const MatchProblem *d_problems, const SpecialJob *d_jobs, int num_jobs,
    RowPart *sp_parts, int32_t *sp_col, hipStream_t s);

// any_special: some problem has row blocks behind nrb_main (gathered special rows);
// any_c0 / any_corrected: some problem has / lacks the correction-free column operand
void launch_match_tiles(int ch, bool masked, bool any_special, bool any_c0, bool any_corrected,
    const MatchProblem *d_problems,
    int num_problems, int total_blocks, RowPart *rowparts, ColPart *colparts, hipStream_t s,
    unsigned long long *clock_probe = nullptr,       // [2]: shader cycles / 100 MHz ticks, added up by sampled workgroups of the C0 kernel
    const int8_t *zero_tile = nullptr);              // kTileCols blank descriptors (the correction-free kernel's filler tiles)

void launch_match_finish(const MatchProblem *d_problems, int num_problems,
    int max_n, const RowPart *rowparts, const ColPart *colparts, const RowPart *sp_parts, const int32_t *sp_col, LoweTable tab,
    int force_exact, ExactItem *exact_items, int32_t *exact_count, int exact_cap,
    hipStream_t s);

void launch_exact_scan(int dim, const MatchProblem *d_problems,
    const ExactItem *items, const int32_t *count, int exact_cap, LoweTable tab,
    hipStream_t s);

void launch_cross_check_mark(const MatchProblem *d_problems, int num_problems, int max_n,
    uint8_t *keep12, uint8_t *keep21, const int64_t *mark_off, int32_t *counts, hipStream_t s);
void launch_cross_check_apply(const MatchProblem *d_problems, int num_problems, int max_n,
    const uint8_t *keep12, const uint8_t *keep21, const int64_t *mark_off, hipStream_t s);

void launch_compact_pairs(int num_pairs, const int32_t *m12_all, const int64_t *m12_off,
    const int32_t *len12, const int64_t *corr_off, const uint8_t *keep, int32_t *corr,
    hipStream_t s);

void launch_gather_inliers(int num, const int32_t *corr, const int32_t *inl, const int64_t *src_off,
    const int64_t *dst_off, const int32_t *counts, int32_t *out, hipStream_t s);

void launch_prepare_sift(const uint16_t *src, int n, int npad, int8_t *dst,
    int32_t *corr, int8_t *dst_raw, int32_t *corr_raw, int32_t *range_err, hipStream_t s);
void launch_gather_rows(const int8_t *src, const int32_t *corr, const int32_t *map, int n, int npad,
    int dim, int32_t pad_corr, int8_t pad_byte, int8_t *dst, int32_t *dst_corr, hipStream_t s);
void launch_prepare_surf(const int16_t *src, int n, int npad, int8_t *dst,
    int32_t *corr, int32_t *norm2max, int32_t *range_err, hipStream_t s);

}  // namespace osfm
