// Cascade hashing (sfm::CascadeHashing, the application's default matcher):
// device data and launchers.  Semantics: src/mve/sfm/cascade_hashing.h:29-470,
// cascade_hashing.cc:20-227; CascadeHashing::Options defaults (6 bucket groups,
// 8 bucket bits, 6..10 candidates, cascade_hashing.h:35-47) are compiled in.
#pragma once
#include "match_kernels.h"

namespace osfm {

constexpr int kCasGroups = 6;        // num_bucket_groups
constexpr int kCasBits = 8;          // num_bucket_bits
constexpr int kCasBuckets = 1 << kCasBits;
constexpr int kCasMinCand = 6;       // min_num_candidates
constexpr int kCasMaxCand = 10;      // max_num_candidates
constexpr int kCasSecBits = kCasGroups * kCasBits;

// Per view and descriptor type.
struct CasView {
    const uint64_t *hashes;      // [n][dim / 64]   LocalData::comp_hash_data
    const uint8_t *bucket_ids;   // [groups][n]     bucket_grps_bucket_ids (ids < 256)
    const int32_t *start;        // [groups][257]   bucket_grps_feature_ids as CSR ...
    const int32_t *items;        // [groups][n]     ... ascending feature ids per bucket
};

// running float sums of compute_avg_descriptors (cascade_hashing.cc:128-163):
// sum[k] += value / div over the n descriptors of one view, in order
void launch_cashash_accumulate(const int8_t *desc, int n, int dim, int bias, float div, float *sum,
    hipStream_t s);
// avg[k] = sum[k] / float(count)
void launch_cashash_average(const float *sum, int dim, int64_t count, float *avg, hipStream_t s);
// zero-mean descriptors, hash words and bucket ids of one view
// (cascade_hashing.cc:165-183, cascade_hashing.h:258-310); projT = [dim][dim + 48]
// (primary vectors then the secondary ones, transposed so that threads read rows)
void launch_cashash_hash(const int8_t *desc, int n, int dim, int bias, float div, const float *avg,
    const float *projT, uint64_t *hashes, uint8_t *bucket_ids, hipStream_t s);
// One 32-byte record per feature for the candidate scan: hash words (the second
// one 0 for SURF), the six bucket ids packed into one 64-bit word (byte g = group g)
struct CasRecord { uint64_t h[2]; uint64_t buckets; uint64_t pad; };
void launch_cashash_pack(const uint64_t *hashes, const uint8_t *bucket_ids, int n, int words,
    CasRecord *rec, hipStream_t s);
// build_buckets (cascade_hashing.cc:187-209)
void launch_cashash_buckets(const uint8_t *bucket_ids, int n, int32_t *start, int32_t *items, hipStream_t s);

// CascadeHashing::twoway_match for a batch of problems (MatchProblem::cas_* set):
// m12 / m21 written, ready for the cross-check kernels.  state: scratch of
// kCasMaxCand ints per query of the batch (MatchProblem::cas_state_off)
void launch_cashash_match(int dim, const MatchProblem *d_problems, int num_problems, int max_n,
    int32_t *state, LoweTable tab, hipStream_t s);

}  // namespace osfm
