/*
 * osfm_hip.h -- C ABI of the MI355X (gfx950) backend for OrthoSfM's hot path:
 *   (A) exhaustive pairwise descriptor matching  (src/mve/sfm, src/matching)
 *   (B) orthographic bundle adjustment           (src/bundle_adjustment, src/algorithms)
 *
 * Plain C: pointers, sizes, POD structs; no C++/torch types.  Every entry
 * point names the reference interface it replaces (paths relative to the
 * OrthoSfM source tree).  All functions return OSFM_OK (0) or a negative
 * osfm_status; osfm_last_error() returns a thread-local description.
 * The library never hands out memory the caller must free and never falls
 * back to a CPU implementation: without a usable gfx950 device every
 * compute entry point fails with OSFM_E_DEVICE.
 */
#ifndef OSFM_HIP_H
#define OSFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSFM_API __attribute__((visibility("default")))

typedef enum osfm_status {
    OSFM_OK = 0,
    OSFM_E_ARG = -1,       /* null / out-of-range argument (MVE throws std::invalid_argument) */
    OSFM_E_DEVICE = -2,    /* no HIP device / HIP runtime error */
    OSFM_E_RANGE = -3,     /* descriptor value outside the quantised range */
    OSFM_E_CAPACITY = -4,  /* caller buffer too small; required size reported */
    OSFM_E_STATE = -5,     /* call order violated (e.g. view not set) */
    OSFM_E_NUMERIC = -6,   /* BA: linear solve failed / non-finite values */
    OSFM_E_IO = -7         /* text formats: file cannot be opened / malformed line */
} osfm_status;

OSFM_API const char *osfm_last_error(void);
/* ABI version: OSFM_ABI_VERSION of the header the library was built from.  The library writes its public structs
 * (osfm_match_stats, osfm_ba_summary, ...) in full, so a caller built against another header must not run: the
 * adapters (the headers under orthosfm_amd/host) and the Python mirror compare this with their own OSFM_ABI_VERSION at load time.
 * 100: rounds 1-3; 101: osfm_ba_summary.flow_fallbacks, osfm_match_stats.surf_*; 102: one observation per camera
 * and point enforced, osfm_scene_set_cameras,
 * osfm_ba_summary.order_arcs / chain_blocks*. */
#define OSFM_ABI_VERSION 102
OSFM_API int osfm_version(void);
/* Number of visible HIP devices (0 when there is none). */
OSFM_API int osfm_device_count(void);
/* Free / total bytes of a device's HBM (hipMemGetInfo): lets a caller size its
 * batches and check that handles give their memory back. */
OSFM_API int osfm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* What the library itself holds in this process, by kind -- the counterpart of osfm_device_memory that does not
 * depend on what the HIP runtime keeps in pools of its own: after osfm_match_destroy of the last matcher
 * device_buffer_bytes, live_matchers are zero and pinned_host_bytes / live_streams / live_events are back at what
 * the per-process stream sets of the BA entries hold (they are kept for reuse); pool_cached_bytes is what
 * osfm_trim_device_memory would hand back. */
typedef struct osfm_memory_report {
    int64_t device_buffer_bytes;   /* grow-only device buffers owned by live handles (banks, scratch, hashes) */
    int64_t pool_live_bytes;       /* work arrays of calls in flight (BA, triangulation, filters) */
    int64_t pool_cached_bytes;     /* work arrays kept for the next call */
    int64_t pinned_host_bytes;     /* page-locked staging of live handles and of the kept stream sets */
    int32_t live_matchers;
    int32_t live_streams;
    int32_t live_events;
    int32_t reserved;
} osfm_memory_report;
OSFM_API int osfm_library_memory(osfm_memory_report *out);

/* The work arrays of the bundle-adjustment / triangulation / filter calls come from a
 * per-device cache of device memory (hipMalloc and hipFree cost 50-100 us apiece, dozens per
 * call, hundreds of calls per reconstruction); at most 8 GiB per device are kept.  This
 * hands what is cached on `device` (-1: all devices) back to the driver; *released_bytes
 * (may be NULL) reports how much that was. */
OSFM_API int osfm_trim_device_memory(int device, uint64_t *released_bytes);
/* Diagnostic: the one-launch Cholesky of osfm_ba_solve leaves 16 stamps per diagonal workgroup of
 * its most recent factorisation: stamps[161][32] (one row per block row), read by tools/chol_flow_trace.py (100 MHz
 * counter: start, L of the last column published, factor started, inverse published; shader
 * cycles of the pivot loop and of the factor; how the inverse arrived: 1 = sc1 copy, 2 =
 * same-XCD mailbox).  enable != 0 switches the recording on
 * (current device) and returns what was recorded so far; 0 returns it and switches it off. */
OSFM_API int osfm_ba_debug_chol_trace(int enable, int64_t *stamps);

/* Test hook: the number of polls a wait of the one-launch Cholesky makes before it gives the launch up
 * (<= 0: the default, 2^21).  A launch given up is repeated by osfm_ba_solve in the launch-per-column
 * form and counted in osfm_ba_summary.flow_fallbacks; set small, every wait that is not satisfied at
 * once takes that path. */
OSFM_API int osfm_ba_debug_flow_spin_limit(int limit);
/* Diagnostic (host code only, no device): the elimination order osfm_ba_solve would choose for a reduced camera
 * system (ba_order.hip) -- cam_ldim[C] unknowns per camera, pairs[num_pairs][2] the camera pairs that share a track.
 * cam_off[C]: the cameras' offsets in the order chosen (the natural ones when none is); blocks (may be NULL) [nblk + 1]
 * rows of 3 x 64 bits: bit k of row i = tile (i, k) of the factor exists; info[8] = { ordered (0 / 1), arcs K,
 * cameras per separator, unknowns incl. interior padding, blocks of 32, chain of dependent diagonal blocks in the
 * cameras' own order, chain in the order chosen, padding unknowns }.  blocks_capacity: rows `blocks` has room for. */
OSFM_API int osfm_ba_debug_order(int num_cameras, const int32_t *cam_ldim, int num_pairs, const int32_t *pairs,
    int32_t *cam_off, uint64_t *blocks, int blocks_capacity, int32_t *info);

/* Diagnostic of the RANSAC-F scoring loop.  Its Sampson tests are pre-classified in packed
 * single precision; a test only counts when the float result is out of reach of its error
 * bound, everything else is redone in the reference's double arithmetic, so the inlier
 * counts are the double ones (orthosfm_amd/csrc/ransac_kernels.hip).  mode 0: double path
 * only; 1: pre-classification (default); 2: pre-classification, every decision checked
 * against the double path on the device.  counters (may be NULL) receives what the
 * launches since the previous call counted in mode 2: [0] wrong decisions (must be 0),
 * [1] undecided tests, [2] tests.  Process-wide; not for concurrent use with matching. */
OSFM_API int osfm_ransac_selfcheck(int mode, uint64_t *counters);

/* ====================================================================== */
/* (A) Matching                                                            */
/* ====================================================================== */

/*
 * Tunables of sfm::MatchingBase::Options (src/mve/sfm/matching_base.h:25-31)
 * and sfm::bundler::Matching::Options (src/mve/sfm/bundler_matching.h:58-77),
 * with the values the application hard-codes
 * (src/matching/matching_mve.cpp:393-408).
 */
/* Matcher behind pairwise_match (bundler_matching.cc:31-41).  Cascade hashing
 * is sfm::CascadeHashing with its default Options (6 bucket groups of 2^8
 * buckets, 6..10 candidates, cascade_hashing.h:35-47); its hashes are computed
 * from ALL views (the descriptor average, cascade_hashing.cc:128-163), lazily
 * at the first match call after a view was set.  pairwise_match_lowres stays
 * exhaustive in both (CascadeHashing inherits it). */
#define OSFM_MATCHER_EXHAUSTIVE 0
#define OSFM_MATCHER_CASCADE_HASHING 1

typedef struct osfm_match_options {
    float sift_lowe_ratio;          /* 0.8f   matching_base.h:27 */
    float sift_distance_threshold;  /* FLT_MAX                    */
    float surf_lowe_ratio;          /* 0.7f   matching_base.h:29 */
    float surf_distance_threshold;  /* FLT_MAX                    */
    int32_t use_lowres_matching;    /* 1      matching_mve.cpp:400 */
    int32_t num_lowres_features;    /* 500    bundler_matching.h:68 */
    int32_t min_lowres_matches;     /* 5      bundler_matching.h:70 */
    int32_t min_feature_matches;    /* 50     matching_mve.cpp:403 */
    int32_t pairs_per_batch;        /* pairs resident in one launch group (0 = auto) */
    /* geometric verification (RansacFundamental, bundler_matching.cc:194-219) */
    int32_t geometric_verification; /* 0: osfm_match_all stops before RANSAC (default) */
    int32_t ransac_max_iterations;  /* 1000   ransac_fundamental.h:42 */
    double ransac_threshold;        /* 0.0015 matching_mve.cpp:395 */
    int32_t min_matching_inliers;   /* 30     matching_mve.cpp:402 */
    int32_t matcher_type;      /* OSFM_MATCHER_EXHAUSTIVE (default) or OSFM_MATCHER_CASCADE_HASHING:
                                * bundler::Matching::MatcherType, bundler_matching.h:52-56 */
    uint64_t ransac_seed;           /* stream seed of the counter-based sampler */
    /* Cascade hashing only.  0 (default): the Result layout of sfm::CascadeHashing --
     * a descriptor type that EITHER view lacks contributes no entries at all
     * (CascadeHashing::oneway_match returns before sizing its list,
     * cascade_hashing.h:341-342, so combine_results sees empty lists and applies no
     * offset, matching.cc:74-86).  1: the exhaustive matcher's layout (the block of a
     * type view_1 has is always present, -1 filled): consistent combined indices. */
    int32_t cascade_keep_empty_blocks;
    /* SIFT descriptors with a byte > 127 ("special": MVE renormalises after the 0.2 clamp,
     * sift.cc:830-839, so a descriptor with its energy in a few bins holds bytes of 128-180)
     * do not fit the raw int8 operand of the correction-free score-tile kernel.  A view with
     * at most this many of them keeps all its other descriptors on that kernel and the
     * special ones are scored by a kernel of their own; a view with more takes the slower
     * per-view operand forms.  0: default (512, also the largest value taken); negative: always the
     * per-view forms.
     * Results are identical either way. */
    int32_t special_kernel_max;
} osfm_match_options;

OSFM_API int osfm_match_options_default(osfm_match_options *opts);

typedef struct osfm_matcher osfm_matcher;

/* Replaces the construction of an sfm::MatchingBase implementation
 * (bundler::Matching::Matching, src/mve/sfm/bundler_matching.cc:27-42). */
OSFM_API int osfm_match_create(int device, int num_views,
    const osfm_match_options *opts, osfm_matcher **out);
OSFM_API int osfm_match_destroy(osfm_matcher *m);

/*
 * The same matcher over several devices of one node, for the single C++ process that
 * drives the reference (bundler::Matching::compute, src/mve/sfm/bundler_matching.cc:58-136):
 * one worker thread and stream set per entry of device_ids for the duration of a call,
 * the descriptor bank on every device, the pairs of osfm_match_all dealt by work (N1*N2,
 * longest first), concurrent osfm_match_pair[_lowres] callers spread round robin.  No
 * data-path exchange between the devices; results (records, list order, bytes) are those
 * of osfm_match_create on one device.  A device id may appear more than once (logical
 * shards on one device).  Every other osfm_match_* entry takes the returned handle.
 */
OSFM_API int osfm_match_create_multi(const int *device_ids, int num_devices, int num_views,
    const osfm_match_options *opts, osfm_matcher **out);
/* The device of every shard (one entry for a single-device matcher). */
OSFM_API int osfm_match_get_devices(const osfm_matcher *m, int32_t *device_ids, int capacity,
    int32_t *num_devices);

/*
 * Host quantisation of float descriptors, i.e. convert_descriptor of
 * ExhaustiveMatching::init_sift / init_surf
 * (src/mve/sfm/exhaustive_matching.cc:17-38, 76-112).
 */
OSFM_API int osfm_quantize_sift(const float *src, int n, uint16_t *dst);
OSFM_API int osfm_quantize_surf(const float *src, int n, int16_t *dst);

/*
 * MatchingBase::init for one view (src/mve/sfm/matching_base.h:40,
 * ExhaustiveMatching::init, exhaustive_matching.cc:55-74).  The _float
 * variant takes Sift::Descriptor::data / Surf::Descriptor::data rows and
 * quantises them like the reference; the plain variant takes the already
 * quantised 16-bit lanes (0..255 / -127..127 [-128 accepted]).  Data is
 * copied to the device; the caller keeps ownership of its buffers (the
 * reference frees the float descriptors right after init,
 * bundler_matching.cc:54-55): values are range-checked and copied to page-locked
 * memory before the call returns, the transfer and the conversion kernels are
 * queued on a stream of their own and not waited for.  A view may be set while
 * matching calls that do not name it are in flight (a matching call waits for
 * the uploads of the views it names); setting a view that a running call names
 * is the caller's error.
 */
OSFM_API int osfm_match_set_view(osfm_matcher *m, int view,
    const uint16_t *sift, int n_sift, const int16_t *surf, int n_surf);
OSFM_API int osfm_match_set_view_float(osfm_matcher *m, int view,
    const float *sift, int n_sift, const float *surf, int n_surf);
/* A caller that feeds osfm_match_all in batches of growing size (a pipeline whose first batches name the few views
 * that are up already) says how many pairs its largest call will hold: the work arrays -- 13 MB of column partials
 * per 20 000-feature pair -- are then sized for that call by the first one, instead of being freed and allocated
 * again at every size on the way (0.7 s of a 200-view job).  A hint: calls of any size stay valid.  0 clears it.
 * (No reference counterpart: bundler::Matching::compute allocates per pair, bundler_matching.cc:86-160.) */
OSFM_API int osfm_match_expect_pairs(osfm_matcher *m, int32_t pairs_per_call);
OSFM_API int osfm_match_view_size(const osfm_matcher *m, int view,
    int *n_sift, int *n_surf);

/* FeatureSet::positions of a view (src/mve/sfm/feature_set.h:66): n = n_sift +
 * n_surf normalised (x, y) float pairs in the combined feature index space.
 * Needed only for geometric verification. */
OSFM_API int osfm_match_set_positions(osfm_matcher *m, int view, const float *xy, int n);

/*
 * MatchingBase::pairwise_match (matching_base.h:43-44;
 * ExhaustiveMatching::pairwise_match, exhaustive_matching.cc:114-144):
 * two-way SIFT + SURF matching, cross-check, combine.  m12 must hold
 * n_sift+n_surf ints of view_1, m21 those of view_2; unsuccessful = -1.
 * len12/len21 receive the lengths the reference's Result vectors would
 * have (a descriptor type that view_1 lacks contributes an EMPTY list).
 * Thread-safe (the reference calls it from an OpenMP loop).
 */
OSFM_API int osfm_match_pair(osfm_matcher *m, int view_1, int view_2,
    int32_t *m12, int32_t *len12, int32_t *m21, int32_t *len21);

/* MatchingBase::pairwise_match_lowres (matching_base.h:51-52;
 * exhaustive_matching.cc:146-180). */
OSFM_API int osfm_match_pair_lowres(osfm_matcher *m, int view_1, int view_2,
    int num_features, int32_t *count);

/*
 * sfm::Matching::twoway_match<T> (src/mve/sfm/matching.h:148-159) on the
 * stored descriptors of one type (0 = SIFT/unsigned short, 1 = SURF/short),
 * optionally restricted to the first num_features descriptors of each view
 * (0 = all): the two one-way lists BEFORE remove_inconsistent_matches.
 * m12 holds n1 ints, m21 n2 ints.
 */
OSFM_API int osfm_match_twoway(osfm_matcher *m, int view_1, int view_2,
    int descriptor_type, int num_features, int32_t *m12, int32_t *m21);

typedef struct osfm_pair {
    int32_t view_1;
    int32_t view_2;
} osfm_pair;

enum {
    OSFM_PAIR_MATCHED = 0,          /* survived both gates */
    OSFM_PAIR_REJECTED_LOWRES = 1,  /* bundler_matching.cc:146-158 */
    OSFM_PAIR_REJECTED_COUNT = 2,   /* bundler_matching.cc:163-172 */
    OSFM_PAIR_SKIPPED_EMPTY = 3,    /* a view without features, :96-99 */
    OSFM_PAIR_REJECTED_INLIERS = 4  /* bundler_matching.cc:203-210 (verification on) */
};

typedef struct osfm_pair_result {
    int32_t status;          /* OSFM_PAIR_* */
    int32_t lowres_matches;  /* -1 when the low-res gate did not apply */
    int32_t num_matches;     /* count_consistent_matches of the full match */
    int32_t num_inliers;     /* RANSAC inliers (-1 when verification is off / not reached) */
    int64_t offset;          /* first correspondence in `corr` (pairs of ints) */
} osfm_pair_result;

/*
 * Batched form of bundler::Matching::compute up to (not including) the
 * RANSAC stage (src/mve/sfm/bundler_matching.cc:58-192): low-res gate,
 * pairwise_match, count_consistent_matches, threshold, and the ordered
 * (feature_1, feature_2) correspondence list of bundler_matching.cc:176-192.
 * Results are returned in INPUT order (deterministic, unlike the
 * reference's thread-completion order).  corr receives 2 ints per
 * correspondence; if more than `capacity` correspondences are produced the
 * call fails with OSFM_E_CAPACITY and *total holds the required count.
 * With opts.geometric_verification the call continues through RANSAC-F
 * (bundler_matching.cc:194-219): the list of a pair then holds its num_inliers
 * inlier correspondences (TwoViewMatching::matches) and pairs with fewer than
 * max(8, min_matching_inliers) inliers are OSFM_PAIR_REJECTED_INLIERS.
 */
OSFM_API int osfm_match_all(osfm_matcher *m, const osfm_pair *pairs,
    int num_pairs, osfm_pair_result *results, int32_t *corr,
    int64_t capacity, int64_t *total);

/* (view_1 > view_2) of linear pair index i, bundler_matching.cc:92-93. */
OSFM_API int osfm_pair_from_index(int64_t index, int32_t *view_1, int32_t *view_2);

/* Statistics of the most recent osfm_match_all / osfm_match_pair call on
 * this matcher, for bench.py's live roofline: device time of the dominant
 * kernel (HIP events on the launch stream) and its launch count. */
typedef struct osfm_match_stats {
    double tile_kernel_ms;     /* sum over the full-matching launches of the score-tile kernel */
    int32_t tile_kernel_launches;
    int32_t exact_scan_queries;  /* queries re-done by the wrap-exact kernel */
    int64_t mac_count;           /* sum of n1*n2*D over those launches */
    int64_t algorithmic_bytes;   /* descriptors read once + results written */
    double lowres_kernel_ms;     /* same, launches of the low-res gate (limited variant) */
    int32_t lowres_kernel_launches;
    int32_t reserved;
    int64_t lowres_mac_count;
    double cashash_kernel_ms;    /* cascade hashing mode: candidate search + NN kernel */
    int32_t cashash_kernel_launches;
    int32_t special_kernel_launches;
    double special_kernel_ms;    /* special descriptors (a byte > 127) against the other view */
    /* sampled workgroups of the correction-free tile kernel: shader cycles and ticks of the 100 MHz
     * counter they took; cycles / ticks * 100 MHz = the clock the chip held under that kernel */
    double tile_shader_cycles;
    double tile_refclk_ticks;
    /* the SURF (D = 64) share of tile_kernel_ms / tile_kernel_launches / mac_count: a view that carries both
     * descriptor types (the application's FEATURE_ALL, matching_mve.cpp:333) runs one launch per type */
    double surf_tile_kernel_ms;
    int32_t surf_tile_kernel_launches;
    int32_t reserved2;
    int64_t surf_mac_count;
} osfm_match_stats;
OSFM_API int osfm_match_get_stats(const osfm_matcher *m, osfm_match_stats *out);
/* The same record of ONE shard of a multi-device matcher (osfm_match_create_multi; shard 0 of a single-device
 * one): what that device did in the most recent call -- the balance of the deal can be read off mac_count. */
OSFM_API int osfm_match_get_shard_stats(const osfm_matcher *m, int shard, osfm_match_stats *out);

/* CascadeHashing::LocalData of one view (cascade_hashing.h:146-168), computed on
 * demand: hashes [n][2] (SIFT, type 0) or [n][1] (SURF, type 1) 64-bit words,
 * bucket_ids [6][n] (ids < 256).  Either output may be NULL. */
OSFM_API int osfm_match_get_cascade_hashes(osfm_matcher *m, int view, int type,
    uint64_t *hashes, uint8_t *bucket_ids);

/* RansacFundamental::Options (src/mve/sfm/ransac_fundamental.h:33-54). */
typedef struct osfm_ransac_options {
    int32_t max_iterations;   /* 1000 */
    int32_t reserved;
    double threshold;         /* 0.0015 */
    uint64_t seed;
} osfm_ransac_options;
OSFM_API int osfm_ransac_options_default(osfm_ransac_options *o);

/* RansacFundamental::estimate for one pair (ransac_fundamental.cc:26-60):
 * pos1/pos2 normalised feature positions, corr k (feature_1, feature_2) pairs.
 * inliers receives ascending ids into corr (capacity k); *num_inliers = -1 for
 * k < 8 (the reference throws).  F (9 doubles, row major) may be NULL.  The
 * sample stream is (seed, pair_id): results do not depend on batching. */
OSFM_API int osfm_ransac_fundamental(int device, const float *pos1, int n1, const float *pos2, int n2,
    const int32_t *corr, int k, const osfm_ransac_options *opts, uint64_t pair_id,
    int32_t *inliers, int32_t *num_inliers, double *F);

/* ====================================================================== */
/* (B) Bundle adjustment                                                   */
/* ====================================================================== */

enum {
    OSFM_BA_MODEL_QUATERNION = 0,  /* solver 0: OrthoQuaternionCamera */
    OSFM_BA_MODEL_EULER = 1        /* solvers 1-3: OrthographicCamera */
};

/*
 * Flattened form of the arguments of orthosfm::runBundleAdjustment
 * (src/bundle_adjustment/bundle_adjustment.h:18-20): cameras and tracks
 * become structure-of-arrays.
 *
 * cam_params: num_cameras x 7 doubles.
 *   QUATERNION: (qx, qy, qz, qw, offsetX, offsetY, scale)  -- Eigen coeff
 *     order, OrthoQuaternionCamera.h:83-86
 *   EULER:      (phi, theta, roll, offsetX, offsetY, scale, unused)
 *     -- OrthographicCamera.h:122-127
 * cam_const: num_cameras x 7 bytes, non-zero = parameter block constant
 *   (camera fixed or per-block flag; OrthoQuaternionRecoAlgorithm.cpp:141-145,
 *   OrthographicReconstructionAlgorithm.cpp:170-176).  For QUATERNION the
 *   first byte covers the whole 4-vector rotation block.
 * points: num_points x 4 homogeneous (Track::m_point, track.h:103), in/out.
 * obs_*: one entry per residual block in the order of
 *   bundle_adjustment.cpp:103-123; obs_point MUST be non-decreasing
 *   (observations of a track are contiguous) and a point is observed AT MOST
 *   ONCE per camera (the reference's tracks hold one feature per view: a
 *   track with two is a conflict and dropped, bundler_tracks.cc:120-145) --
 *   a repeated (camera, point) is refused with OSFM_E_ARG.
 */
typedef struct osfm_ba_problem {
    int32_t model;
    int32_t num_cameras;
    int32_t num_points;
    int32_t num_observations;
    double *cam_params;
    const uint8_t *cam_const;
    const int32_t *img_width;
    const int32_t *img_height;
    double *points;
    const double *obs_xy;
    const int32_t *obs_camera;
    const int32_t *obs_point;
} osfm_ba_problem;

/* Solver settings of bundle_adjustment.cpp:61-64,126-133 plus the Ceres
 * defaults the reference leaves untouched. */
typedef struct osfm_ba_options {
    double huber_delta;               /* 1.0   HuberLoss(1.0) */
    double function_tolerance;        /* 1e-6  */
    double gradient_tolerance;        /* 1e-10 */
    double parameter_tolerance;       /* 1e-10 */
    int32_t max_num_iterations;       /* 100   */
    int32_t optimize_points;          /* runBundleAdjustment's optimizePoints */
    double initial_trust_region_radius;  /* 1e4  */
    double max_trust_region_radius;      /* 1e16 */
    double min_trust_region_radius;      /* 1e-32 */
    double min_relative_decrease;        /* 1e-3 */
    double min_lm_diagonal;              /* 1e-6 */
    double max_lm_diagonal;              /* 1e32 */
    int32_t jacobi_scaling;              /* 1 */
    int32_t max_consecutive_invalid_steps;  /* 5 */
    int32_t device;
    int32_t verbose;
    /* runBundleAdjustment's retriangulatePoints (bundle_adjustment.cpp:77-83): the points
     * are first replaced by the ray intersections of their observations under the given
     * cameras (triangulateOrthographicTracks with resetExistingPoints; a point with fewer
     * than two observations keeps its input value).  The reference does this on a filtered
     * copy of the tracks that it discards; whether to keep the returned points is the
     * caller's choice. */
    int32_t retriangulate_points;      /* 0 */
    int32_t reserved;
} osfm_ba_options;

enum {
    OSFM_BA_CONVERGENCE_FUNCTION = 1,
    OSFM_BA_CONVERGENCE_GRADIENT = 2,
    OSFM_BA_CONVERGENCE_PARAMETER = 3,
    OSFM_BA_CONVERGENCE_TRUST_REGION = 4,
    OSFM_BA_NO_CONVERGENCE = 5,       /* max iterations */
    OSFM_BA_FAILURE = 6
};

typedef struct osfm_ba_summary {
    double initial_cost;
    double final_cost;
    int32_t num_iterations;           /* LM iterations incl. rejected, like Ceres' iteration count */
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int32_t termination;              /* OSFM_BA_* */
    double mean_point_change;         /* bundle_adjustment.cpp:150-160 printout */
    double max_point_change;
    double solve_ms;                  /* device+host wall time of the whole call */
    double point_pass_ms;             /* summed device time (HIP events) per kernel family */
    double pair_pass_ms;
    double cholesky_ms;
    double back_pass_ms;
    int32_t linearizations;           /* number of point/pair pass executions timed */
    int32_t num_pair_entries;         /* observation pairs in the Schur complement lists */
    double lm_loop_ms;                /* wall time of the LM iterations alone (host control included;
                                       * problem upload, pair lists and the first linearisation are not) */
    int32_t flow_fallbacks;           /* factorisations whose one-launch form gave its launch up (its workgroups
                                       * were not all resident: the device was shared) and that were repeated in
                                       * the launch-per-column form; the results do not depend on it */
    int32_t order_arcs;               /* 0: the reduced camera system was factored in the cameras' own order; K > 0: the
                                       * cameras form a ring / strip (every track spans a short run of views) and the
                                       * system was laid out as K arcs + K separators, factored side by side */
    int32_t chain_blocks_natural;     /* longest chain of dependent 32-unknown diagonal blocks in the cameras' order ... */
    int32_t chain_blocks;             /* ... and in the order used */
} osfm_ba_summary;

OSFM_API int osfm_ba_options_default(osfm_ba_options *opts);

/* ceres::Solve on the problem runBundleAdjustment builds
 * (bundle_adjustment.cpp:61-145): cam_params and points updated in place. */
OSFM_API int osfm_ba_solve(const osfm_ba_problem *p, const osfm_ba_options *o,
    osfm_ba_summary *s);

/* Batched ReconstructionAlgorithm::evaluateReprojectionError
 * (OrthoQuaternionRecoAlgorithm.cpp:175-194,
 * OrthographicReconstructionAlgorithm.cpp:204-223): err[k] = ||r_k||_2 in
 * pixels; residuals (2 per observation) may be NULL. */
OSFM_API int osfm_ba_reprojection_errors(const osfm_ba_problem *p, int device,
    double *err, double *residuals);

/* triangulateOrthographicTracks (src/triangulation/triangulation.cpp:44-93)
 * as called from runBundleAdjustment (bundle_adjustment.cpp:77-83):
 * points[j] = least-squares ray intersection over the track's observations,
 * point_valid[j] = 0 when fewer than two rays. */
OSFM_API int osfm_ba_triangulate(const osfm_ba_problem *p, int device,
    uint8_t *point_valid);

/* ---- outlier filters around the bundle adjustment ---- */

/* orthosfm::getNearestNeighbourDistance
 * (src/triangulation/outlier_filtering.cpp:14-38): nn[i] = min over j != i of
 * the 2-norm of the difference of the HOMOGENEOUS 4-vectors, 1000000 when no
 * point is closer.  points: [num_points][4]. */
OSFM_API int osfm_nn_distances(int device, const double *points, int32_t num_points,
    double *nn);

typedef struct osfm_outlier_stats {
    double mean;               /* of the nearest-neighbour distances */
    double sigma;              /* as the reference computes it (:80-98), after the 1e-3 floor */
    int32_t num_with_point;
    int32_t num_kept;
} osfm_outlier_stats;

/* orthosfm::filterOutlierTracks (outlier_filtering.cpp:40-125) on flattened
 * tracks: points [num_tracks][4] (Track::getPoint), has_point [num_tracks]
 * (Track::hasPoint); keep[t] = 1 when the track is in the returned list.
 * stats may be NULL. */
OSFM_API int osfm_filter_outlier_tracks(int device, const double *points,
    const uint8_t *has_point, int32_t num_tracks, uint8_t *keep,
    osfm_outlier_stats *stats);

/* ---- the incremental reconstruction's scene, resident on the device (the caller of path B) ----
 *
 * runPoseEstimation (src/sfm/reconstruct.cpp:193-281) hands filtered copies of its std::vector<Track> to every step:
 * per camera group filterTracksWithReprojectionError + a 3-camera runBundleAdjustment on a re-triangulated copy,
 * triangulateTracks over everything, every third group a global runBundleAdjustment, filterOutlierTracks and the
 * reprojection filter again.  Through the per-call entries above each of those calls flattens and uploads its
 * tracks.  A scene holds the track table ONCE -- features in track order, alive flags per feature and track, point
 * and hasPoint() per track, the aligned cameras -- and every step selects its observations from the flags on the
 * device.  What a filter of the reference drops from its list is a cleared flag here.  The per-call entries stay
 * (orthosfm_amd/host/ba_hip_adapter.h uses them); results are identical, step by step.  A scene is used by one
 * thread at a time (its entries lock it). */
typedef struct osfm_scene osfm_scene;

/* track_offsets [num_tracks + 1] (features of track t: [offsets[t], offsets[t+1])), feat_view / feat_xy per feature
 * (Feature::viewID and the float pixel position Feature::x / y, track.h:26-27), image size per view.  Every flag
 * starts alive, no track has a point, no view has a camera.  A track holds at most one feature per view (OSFM_E_ARG
 * otherwise; bundler_tracks.cc:120-145 drops such tracks). */
OSFM_API int osfm_scene_create(int device, int model, int num_views, const int32_t *img_width, const int32_t *img_height,
    int32_t num_tracks, const int64_t *track_offsets, const int32_t *feat_view, const float *feat_xy, osfm_scene **out);
OSFM_API int osfm_scene_destroy(osfm_scene *s);
/* alive flags from the caller's table (either may be NULL: unchanged).  Once the scene has dropped dead entries
 * (it compacts its table when enough have died) a flag that would set one of THOSE alive is refused with
 * OSFM_E_STATE and nothing changes: clearing flags always works. */
OSFM_API int osfm_scene_set_flags(osfm_scene *s, const uint8_t *alive_track, const uint8_t *alive_feature);
/* mergeIntoGlobal (reconstruct.cpp:236-247): the views get cameras (params [n][7], const masks [n][7] as in
 * osfm_ba_problem), appended to the aligned cameras in this order.  OSFM_E_STATE when a view has one already. */
OSFM_API int osfm_scene_align_views(osfm_scene *s, int n, const int32_t *views, const double *params, const uint8_t *cam_const);
/* Replaces the parameters (params [n][7]) of views that ARE aligned -- a caller's own change to alignedCameras, e.g.
 * normalizeScene when camera 0 is not the identity -- so that the device copy follows; OSFM_E_STATE for a view without
 * a camera.  The next osfm_scene_triangulate redoes every track, whatever new_views says. */
OSFM_API int osfm_scene_set_cameras(osfm_scene *s, int n, const int32_t *views, const double *params);
/* the aligned cameras in the order they joined (views / params may be NULL) */
OSFM_API int osfm_scene_get_cameras(osfm_scene *s, int capacity, int32_t *views, double *params, int32_t *num_cameras);
/* algorithm->triangulateTracks(alignedCameras, tracks, true) (triangulation.cpp:44-93): every alive track with two or
 * more rays under the aligned cameras gets its intersection, the others lose their point.  new_views != NULL: only
 * the tracks those views see are redone -- identical to the full pass as long as the other cameras have not moved
 * since the last one; check_full != 0 repeats the pass in full and counts the tracks that differ (*mismatches). */
OSFM_API int osfm_scene_triangulate(osfm_scene *s, int num_new_views, const int32_t *new_views, int check_full, int32_t *mismatches);
/* filterTracksWithReprojectionError(tracks, alignedCameras) (outlier_filtering.cpp:127-192), permanent */
OSFM_API int osfm_scene_filter_reprojection(osfm_scene *s, double max_error);
/* One camera group (reconstruct.cpp:205-219): the reprojection filter under the cameras (views, params, cam_const)
 * on a copy, then runBundleAdjustment(localCameras, localTracks, algorithm, true, true) on what it leaves -- the
 * tracks seen by at least two of the cameras, re-triangulated first, their points discarded afterwards.  params is
 * updated in place; the scene is not changed. */
OSFM_API int osfm_scene_local_adjustment(osfm_scene *s, int n, const int32_t *views, double *params, const uint8_t *cam_const,
    double max_error, const osfm_ba_options *opt, osfm_ba_summary *sum, int32_t *num_points, int32_t *num_observations);
/* runBundleAdjustment(alignedCameras, tracks, algorithm, true, false) (bundle_adjustment.cpp:49-161): every alive
 * track with a point, every live feature of it whose view has a camera; cameras and points updated in the scene. */
OSFM_API int osfm_scene_global_adjustment(osfm_scene *s, const osfm_ba_options *opt, osfm_ba_summary *sum, int32_t *num_points,
    int32_t *num_observations);
/* filterOutlierTracks over the alive tracks (outlier_filtering.cpp:40-125); what it drops loses its flag */
OSFM_API int osfm_scene_filter_outliers(osfm_scene *s, osfm_outlier_stats *stats, int32_t *num_killed);
/* the scene's state back: alive flags per track / feature, hasPoint() and point [num_tracks][4] (any may be NULL) */
OSFM_API int osfm_scene_download(osfm_scene *s, uint8_t *alive_track, uint8_t *alive_feature, uint8_t *has_point, double *points);


/* The device part of orthosfm::filterTracksWithReprojectionError
 * (outlier_filtering.cpp:127-192).  p holds the FULL-SIZE tracks (the
 * caller's filterTracksToAvailableCameras(cameras, tracks, true, true)
 * selection, :131) flattened as for osfm_ba_solve.  They are re-triangulated
 * (triangulateTracks(cameras, fullSizeTracks, true), :134; p->points is
 * overwritten, point_valid[j] = 0 where fewer than two rays exist and the input
 * point is kept), every observation is evaluated against its track's point
 * (:158) and obs_keep[k] = err[k] < max_error (1.5 px in the reference, :140).
 * err and point_valid may be NULL. */
OSFM_API int osfm_filter_reprojection(const osfm_ba_problem *p, int device,
    double max_error, uint8_t *obs_keep, uint8_t *point_valid, double *err);

/* ---- track building ---- */

typedef struct osfm_tracks_summary {
    int32_t num_tracks;           /* tracks in the output */
    int32_t num_invalid_tracks;   /* removed for holding two features of one view (bundler_tracks.cc:166-171) */
    int64_t num_features;         /* feature references in the output */
} osfm_tracks_summary;

/* sfm::bundler::Tracks::compute (src/mve/sfm/bundler_tracks.cc:49-145, with
 * unify_tracks :23-45 and remove_invalid_tracks :149-203) on flat arrays; host
 * code, like the reference's.
 *   view_sizes   [num_views]          features per view (positions.size())
 *   colors       [sum sizes][3]       FeatureSet::colors, views concatenated (NULL: black)
 *   pairs        [num_pairs]          TwoViewMatching::view_1_id / view_2_id, PairwiseMatching order
 *   pair_offsets [num_pairs + 1]      match range of each pair in corr (osfm_match_all's
 *                                     results give them: offset .. offset + num_inliers)
 *   corr         [..][2]              (feature in view_1, feature in view_2)
 * out:
 *   track_ids    [sum sizes]          Viewport::track_ids, views concatenated, -1 = none
 *   track_offsets[num_tracks + 1], track_features [..][2] = (view_id, feature_id)
 *                                     in the reference's order, track_colors [num_tracks][3]
 * Capacities count tracks / feature references; a safe bound for both is the
 * number of matches resp. twice that.  OSFM_E_CAPACITY leaves summary filled in. */
OSFM_API int osfm_tracks_compute(int32_t num_views, const int32_t *view_sizes,
    const uint8_t *colors, int32_t num_pairs, const osfm_pair *pairs,
    const int64_t *pair_offsets, const int32_t *corr, int32_t *track_ids,
    int64_t track_capacity, int64_t feature_capacity, int64_t *track_offsets,
    int32_t *track_features, uint8_t *track_colors, osfm_tracks_summary *summary);

/* The same with one explicit match range per pair: pair p owns
 * corr[pair_starts[p] .. pair_starts[p] + pair_counts[p]).  For match lists that
 * are not packed back to back -- the per-rank slices of a multi-GPU run lie in
 * one shared host segment (orthosfm_amd/distributed.py) and are consumed where
 * they are. */
OSFM_API int osfm_tracks_compute_ranges(int32_t num_views, const int32_t *view_sizes,
    const uint8_t *colors, int32_t num_pairs, const osfm_pair *pairs,
    const int64_t *pair_starts, const int64_t *pair_counts, const int32_t *corr,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary);

/* Tracks::compute fed pair batch by pair batch (batches in the reference's pair order,
 * bundler_tracks.cc:66-119 is a sequential merge over the pairs): the host can merge the lists of
 * one batch of osfm_match_all while the device matches the next.  feed: pair p of the batch owns
 * corr[pair_starts[p] .. pair_starts[p] + pair_counts[p]) (2 ints per correspondence); finish
 * writes what osfm_tracks_compute writes (same arguments; track_ids may be NULL here) and leaves
 * the builder usable. */
typedef struct osfm_tracks_builder osfm_tracks_builder;
OSFM_API int osfm_tracks_builder_create(int32_t num_views, const int32_t *view_sizes, osfm_tracks_builder **out);
OSFM_API int osfm_tracks_builder_feed(osfm_tracks_builder *b, int32_t num_pairs, const osfm_pair *pairs,
    const int64_t *pair_starts, const int64_t *pair_counts, const int32_t *corr);
OSFM_API int osfm_tracks_builder_finish(const osfm_tracks_builder *b, const uint8_t *colors,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary);
OSFM_API int osfm_tracks_builder_destroy(osfm_tracks_builder *b);

/* The feature table the reconstruction works on, from the tracks (the conversion at the end of
 * calculateTracksUsingMVE, matching_mve.cpp:455-466): for feature i of the tracks (track order,
 * track_features = (view, feature) pairs as osfm_tracks_compute writes them) view_out[i], feat_out[i],
 * xy_out[2i..] = the pixel position  float(image_width * (double(normalised) + 0.5))  for BOTH axes
 * (stored as the double of that float: Feature::x/y are float, track.h:26-27), track_of_out[i]
 * (may be NULL), and -- both or neither -- by_view_out (the feature indices grouped by view,
 * ascending inside a view) with view_start_out[num_views + 1].  norm_positions[v] points to view v's
 * [view_sizes[v]][2] normalised positions.  OSFM_E_RANGE for a feature outside its view. */
OSFM_API int osfm_tracks_feature_table(int64_t num_tracks, const int64_t *track_offsets,
    const int32_t *track_features, int32_t num_views, const int32_t *view_sizes,
    const float *const *norm_positions, double image_width,
    int32_t *view_out, int32_t *feat_out, double *xy_out, int32_t *track_of_out,
    int64_t *by_view_out, int64_t *view_start_out);

/* The observation arrays of an osfm_ba_problem from a scene's tracks in one pass -- what
 * runBundleAdjustment / triangulateOrthographicTracks build residual by residual from their
 * std::vector<Track> (bundle_adjustment.cpp:86-123, triangulation.cpp:43-93): features are in
 * track order; feature i is taken when live[i] != 0, camera_of_feature[i] >= 0 (its view has a
 * camera) and (track_mask == NULL or track_mask[track_of[i]] != 0).  obs_point[k] = track_slot[track]
 * when track_slot is given (the caller's numbering of the point blocks), else the rank of the track
 * among the tracks that contributed a feature -- their ids then go to tracks_out (capacity rows as
 * well).  feature_ids (may be NULL) receives the taken feature indices.  *num_observations is always
 * set; OSFM_E_CAPACITY when it exceeds `capacity`. */
OSFM_API int osfm_tracks_select_observations(int64_t num_features, const int32_t *track_of,
    const int64_t *track_offsets /* NULL, or the first feature of every track, [largest track id + 2] entries (features
                                    are in track order): the pass then goes track by track and jumps over unselected ones */,
    const int32_t *camera_of_feature, const uint8_t *live, const uint8_t *track_mask,
    const int32_t *track_slot, const double *xy, int64_t capacity, int32_t *feature_ids,
    double *obs_xy, int32_t *obs_camera, int32_t *obs_point, int32_t *tracks_out,
    int64_t *num_observations, int64_t *num_tracks_out);

/* orthosfm::buildGroups (src/data_structures/group.cpp:13-88, completeGroup
 * :90-155): the order in which the incremental reconstruction adds views, as
 * groups of group_size views (3 in the reference's algorithms).
 *   view_ids [num_views]   View::getID() in the order of the `views` vector
 *                          (views 0 and 1 seed the first group)
 *   tracks as CSR: track_offsets [num_tracks + 1], track_views [..] = Feature::viewID
 * out: groups [max_groups][group_size] view ids, group_tracks [max_groups]
 * (ViewGroup::tracks), *num_groups.  A score is the number of tracks holding
 * every view of the group (bitset popcounts on the device); candidates are taken
 * in ascending id order on ties (the reference's OpenMP loop leaves ties to
 * thread timing).  OSFM_E_STATE when a remaining view shares no track with any
 * seed group (the reference never terminates there). */
OSFM_API int osfm_build_groups(int device, int32_t num_views, const int32_t *view_ids,
    int32_t num_tracks, const int64_t *track_offsets, const int32_t *track_views,
    int32_t group_size, int32_t max_groups, int32_t *groups, int32_t *group_tracks,
    int32_t *num_groups);

/* ---------------------------------------------------------------------------
 * On-disk text formats of the pipeline either side of the path (SURVEY 8f rank
 * 4): what `orthosfm-app --calculated-tracks` reads and the testbench parses.
 * Host code.  Numbers are written by the same C++ stream operations the
 * reference uses (`ostream << float/double/unsigned`, `std::to_string(double)`),
 * so the bytes agree by construction.  Files that cannot be opened and lines
 * that do not parse give OSFM_E_IO (the reference's std::stoi/stod throw there).
 * ------------------------------------------------------------------------- */

/* orthosfm::Feature (src/data_structures/track.h:21-31), one record per track feature */
typedef struct osfm_track_feature {
    uint32_t view_id, local_feature_id, global_feature_id;
    float x, y;                   /* pixel coordinates */
    uint32_t r, g, b;
} osfm_track_feature;

/* saveTracksToFile (src/matching/matching_io.cpp:16-48): one line per track,
 * "count;view;local;global;x;y;r;g;b;view;..." -- tracks as CSR over features. */
OSFM_API int osfm_tracks_file_write(const char *path, int64_t num_tracks,
    const int64_t *track_offsets, const osfm_track_feature *features);

/* loadTracksFromFile (matching_io.cpp:50-97).  *num_tracks / *num_features are
 * always set to what the file holds; with capacities below that the call fails
 * with OSFM_E_CAPACITY and writes nothing else (call once with 0 capacities to
 * size the buffers).  track_offsets takes num_tracks + 1 entries. */
OSFM_API int osfm_tracks_file_read(const char *path, int64_t track_capacity,
    int64_t feature_capacity, int64_t *track_offsets, osfm_track_feature *features,
    int64_t *num_tracks, int64_t *num_features);

/* saveTracksToPairwiseFiles (matching_io.cpp:99-141): for every pair i < j of
 * view_ids the tracks reduced to exactly those two views
 * (filterTracksToAvailableCameras(ids, tracks, true, false), common.cpp:85-138),
 * one line "x_i y_i x_j y_j" each, in `folder`/%03d_%03d.txt; no file for a pair
 * without such tracks.  *files_written (may be NULL) counts the files. */
OSFM_API int osfm_tracks_pairwise_files_write(const char *folder, int32_t num_views,
    const uint32_t *view_ids, int64_t num_tracks, const int64_t *track_offsets,
    const osfm_track_feature *features, int64_t *files_written);

/* MVE tracks -> orthosfm tracks (src/matching/matching_mve.cpp:455-466):
 * feature (v, f) of an MVE track becomes Feature(v, f, 32768 v + f,
 * width (pos.x + 0.5), width (pos.y + 0.5)) -- the reference scales BOTH axes by
 * the image width -- with the feature's colour.
 *   track_features [num_features][2]  (view, feature), osfm_tracks_compute's output
 *   view_starts    [num_views + 1]    first row of each view in positions / colors
 *   positions      [..][2] float      FeatureSet::positions, views concatenated
 *   colors         [..][3]            FeatureSet::colors (NULL: 0) */
OSFM_API int osfm_tracks_from_mve(int64_t num_features, const int32_t *track_features,
    int32_t num_views, const int64_t *view_starts, const float *positions,
    const uint8_t *colors, double image_width, osfm_track_feature *features);

/* exportCamerasToFile / importCameraFileAsMatrix
 * (src/data_structures/camera_io.cpp:15-40, :42-71): "name;m00,m01,...,m33" with
 * the row-major 4x4 camera-to-world matrix [x y z origin; 0 0 0 1] in
 * std::to_string format.  Reading: names come back NUL-separated in names_buf;
 * *num_cameras / *names_bytes are always set, OSFM_E_CAPACITY as above. */
OSFM_API int osfm_cameras_file_write(const char *path, int32_t num_cameras,
    const char *const *image_names, const double *matrices);
OSFM_API int osfm_cameras_file_read(const char *path, int32_t camera_capacity,
    int64_t names_capacity, char *names_buf, double *matrices, int32_t *num_cameras,
    int64_t *names_bytes);

/* savePointsToPLY (src/util/common.cpp:141-188): ASCII PLY of the tracks that
 * have a point (has_point[t] != 0), xyz of the homogeneous point as written by
 * `ostream << double`, colour of the track's first feature. */
OSFM_API int osfm_sparse_cloud_write(const char *path, int64_t num_tracks,
    const int64_t *track_offsets, const osfm_track_feature *features,
    const double *points, const uint8_t *has_point);

/* saveRuntimesToTxt / runtimesFromTxt (src/util/timing.cpp:18-53); seconds[4] =
 * initialisation, track building, pose estimation, total. */
OSFM_API int osfm_time_measurements_write(const char *path, const double *seconds);
OSFM_API int osfm_time_measurements_read(const char *path, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* OSFM_HIP_H */
