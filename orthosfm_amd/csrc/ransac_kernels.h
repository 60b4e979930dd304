// Batched RANSAC-F (geometric verification of matched view pairs).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace osfm {

struct RansacJob {
    const float *pos1;        // [n1][2] normalised feature positions of view 1 (device)
    const float *pos2;
    const int32_t *corr;      // [k][2] (feature in view 1, feature in view 2) (device)
    int32_t k;
    int32_t pad_;
    uint64_t pair_id;         // random stream of the pair
    int32_t *inliers_out;     // [k] ids into corr of the inliers of the best hypothesis
    int32_t *count_out;       // number of inliers (-1: fewer than 8 matches)
    double *F_out;            // [9] or null
};

constexpr int kRansacSplit = 2;     // workgroups per pair (each takes a share of the hypotheses)

// scratch: ransac_scratch_bytes(num_jobs) of device memory (best-of-part slots and the
// per-pair completion counters), 16-byte aligned
size_t ransac_scratch_bytes(int num_jobs);
void launch_ransac(const RansacJob *d_jobs, int num_jobs, int max_iterations, double threshold,
    uint64_t seed, void *scratch, hipStream_t s);
// see osfm_ransac_selfcheck (include/osfm_hip.h)
int ransac_set_mode(int mode, unsigned long long *counters_out);

}  // namespace osfm
