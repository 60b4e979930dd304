// C-ABI implementation of the matching path (include/osfm_hip.h, section A).
// Host orchestration only: every inner product, reduction, ratio test,
// cross-check and compaction runs in the kernels of match_kernels.hip.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "match_kernels.h"
#include "cashash_kernels.h"
#include <random>
#include "osfm_common.h"
#include "ransac_kernels.h"

namespace osfm {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct ViewData {
    bool set = false;
    int ns = 0, nu = 0;
    int ns_pad = 0, nu_pad = 0;
    DeviceBuffer sift, sift_corr, surf, surf_corr;
    // raw form of the SIFT rows (rows with a value > 127 blanked) and the
    // gathered "special" rows that need the value-128 form on the row side
    DeviceBuffer sift_raw, sift_raw_corr, special, special_corr, special_map, special_slot;
    int n_special = 0;
    int surf_norm2_max = 0;
    // cascade hashing data per descriptor type (0: SIFT, 1: SURF), see cashash_kernels.h
    DeviceBuffer cas_hash[2], cas_bucket[2], cas_start[2], cas_items[2], cas_rec[2];
    DeviceBuffer positions;      // [ns + nu][2] floats (geometric verification only)
    int n_positions = -1;
    // recorded on the upload stream behind the view's transfer and conversion kernels; the matching
    // stream waits for it before a batch that names the view (uploads do not wait for matching and
    // matching does not wait for uploads of views it does not use)
    hipEvent_t ready = nullptr;
    ~ViewData() { event_destroy(ready); }
    ViewData() = default;
    ViewData(const ViewData &) = delete;
    ViewData &operator=(const ViewData &) = delete;
};

// Host twin of matching.h:126-127,138-143 for every (d1, d2): the smallest
// (even) d1 that the float ratio test rejects for a given d2.
void build_lowe_table(float lowe, bool is_signed, std::vector<int32_t> *tab)
{
    const volatile float sq = lowe * lowe;     // MATH_POW2 in float
    const float sq_lowe = sq;
    const int dmax = is_signed ? 32258 : 65534;
    tab->assign(32768, 0x7fffffff);
    for (int d2 = 0; d2 <= dmax; d2 += 2) {
        // float(d1)/float(d2) is monotone in d1: binary search on even d1
        int lo = 0, hi = dmax / 2 + 1;   // in units of 2; hi = "never rejected"
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            const float ratio = static_cast<float>(2 * mid) / static_cast<float>(d2);
            if (ratio > sq_lowe) hi = mid; else lo = mid + 1;
        }
        (*tab)[d2 / 2] = lo > dmax / 2 ? 0x7fffffff : 2 * lo;
    }
}

int max_d1_for(float dist_thres, bool is_signed)
{
    const volatile float sq = dist_thres * dist_thres;
    const float sq_dist = sq;
    const int dmax = is_signed ? 32258 : 65534;
    int best = -1;
    // largest d with !(float(d) > sq_dist); monotone
    int lo = 0, hi = dmax;
    if (!(static_cast<float>(0) > sq_dist)) {
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (static_cast<float>(mid) > sq_dist) hi = mid - 1; else lo = mid;
        }
        best = lo;
    }
    return best;
}

}  // namespace
}  // namespace osfm

using namespace osfm;

struct osfm_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    osfm_match_options opts;
    std::vector<ViewData> views;
    DeviceBuffer lowe_sift, lowe_surf;
    LoweTable tab_sift, tab_surf;
    std::mutex mu;
    // uploads (osfm_match_set_view) have a stream and a lock of their own: a view can be uploaded while a
    // batch that does not name it is being matched
    hipStream_t up_stream = nullptr;
    std::mutex up_mu;

    // scratch (grow-only)
    DeviceBuffer d_problems[2], rowparts, colparts, out, keep, mark_off[2], counts[2];
    DeviceBuffer exact_items, exact_count, stage_in, flags;
    DeviceBuffer sp_parts, sp_col, d_spjobs;      // match_special_kernel: row results, column results, job list
    DeviceBuffer clock_probe;
    DeviceBuffer zero_tile;               // kTileCols blank descriptors: filler tiles of the correction-free tile loop
    int special_max = 512;                // views with more special descriptors take the per-view operand forms
    int expect_pairs = 0;                 // osfm_match_expect_pairs: the largest call to come (work arrays sized for it)
    DeviceBuffer d_m12_off, d_len12, d_corr_off, d_keep_pair, d_corr;
    // osfm_match_all without verification: the lists of chunk k leave the device on a copy stream while
    // chunk k + 1 is matched (two list buffers alternate)
    DeviceBuffer d_corr_alt;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_compact[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    bool copied_used[2] = {false, false};
    DeviceBuffer d_jobs, d_inl, d_inl_count, d_corr2, d_gather_off;
    hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipEvent_t ev_sp[2] = {nullptr, nullptr};
    // page-locked staging of the view uploads (two blocks alternate; the event says a block's transfer is done)
    char *pin_ptr[2] = {nullptr, nullptr};
    size_t pin_bytes[2] = {0, 0};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_used[2] = {false, false};
    unsigned pin_next = 0;

    // cascade hashing: projection matrices (transposed), running sums, average; the
    // hashes depend on the average over ALL views, hence the dirty flag
    DeviceBuffer cas_proj[2], cas_sum[2], cas_avg[2], cas_state;
    std::atomic<bool> cas_dirty{true};

    osfm_match_stats stats;

    // Concurrent callers of the per-pair entries (the reference drives MatchingBase from an
    // OpenMP loop, bundler_matching.cc:86-88) are combined: whoever finds no leader at work
    // takes every request that is waiting, runs them as ONE batch and hands the results out.
    // A pair alone costs 0.26 ms of launch and synchronisation latency for 60 us of GPU work.
    struct Staging { int32_t *ptr = nullptr; size_t ints = 0; int users = 0; };
    struct PairRequest {
        int kind;                       // 0: pairwise_match, 1: pairwise_match_lowres
        int v1, v2, num_features;
        int32_t *m12, *m21, *len12, *len21, *count;
        int status = OSFM_OK;
        bool done = false;
        std::string error;
        // where the leader left this request's lists: in a page-locked staging block that the
        // requester copies from itself (the copies of a batch run in parallel, one per thread)
        Staging *stage = nullptr;
        int64_t src12 = 0, src21 = 0;
        int32_t n12 = 0, n21 = 0;
    };
    std::vector<Staging *> comb_staging;      // guarded by comb_mu; blocks live as long as the matcher

    // Multi-device front (osfm_match_create_multi): no device state of its own, one complete
    // matcher per entry of device_ids (the same device may appear more than once: logical
    // shards).  Every view goes to every shard; osfm_match_all deals its pairs over them.
    std::vector<osfm_matcher *> shards;
    std::vector<Staging> shard_stage;         // page-locked list buffer per shard (grow-only)
    std::atomic<unsigned> next_shard{0};      // per-pair entries: round robin
    std::mutex multi_mu;                      // one osfm_match_all at a time on the front
    std::mutex comb_mu;
    std::condition_variable comb_cv;
    std::deque<PairRequest *> comb_queue;
    bool comb_leader = false;
};

namespace {

struct PairPlan {
    int v1, v2;
    int ns1, nu1, ns2, nu2;      // (possibly low-res limited) sizes used
    bool has_sift, has_surf;     // problems generated
    int64_t off12, off21;        // into the combined output buffer (ints)
    int len12, len21;
    int prob_index[2];           // index into the per-type problem list or -1
};

struct BatchResult {
    std::vector<PairPlan> plans;
    std::vector<int32_t> counts;     // mutual matches per pair (sift + surf)
    int64_t out_ints = 0;
};

int check_view(const osfm_matcher *m, int v, const char *what)
{
    if (v < 0 || v >= (int)m->views.size()) {
        set_error("%s: view id %d out of range [0, %zu)", what, v, m->views.size());
        return OSFM_E_ARG;
    }
    if (!m->views[v].set) {
        set_error("%s: view %d has not been set", what, v);
        return OSFM_E_STATE;
    }
    return OSFM_OK;
}

// Runs one batch of pairs through the device pipeline.
//   lowres_limit > 0 : pairwise_match_lowres semantics (SIFT if view_1 has
//                      SIFT, else SURF; first `limit` descriptors; count only)
//   lowres_limit == 0: pairwise_match semantics (both types, cross-check
//                      applied, combined lists left in m->out)
struct BatchMode {
    int limit = 0;            // > 0: only the first `limit` descriptors of each view
    bool lowres = false;      // pairwise_match_lowres: SIFT if view_1 has SIFT, else SURF
    bool apply = true;        // remove_inconsistent_matches + combine offsets
    int type_mask = 3;        // bit 0: SIFT, bit 1: SURF
    int expect = 0;           // pairs of the largest batch to come (osfm_match_expect_pairs): the work arrays are sized for it
};

int ensure_cashash(osfm_matcher *m);

int run_batch(osfm_matcher *m, const osfm_pair *pairs, int num_pairs, const BatchMode &mode,
    BatchResult *res)
{
    const int lowres_limit = mode.limit;
    // CascadeHashing overrides pairwise_match only (cascade_hashing.h:49-56): limited
    // (low-res / num_features) matching stays exhaustive
    const bool cascade = m->opts.matcher_type == OSFM_MATCHER_CASCADE_HASHING && mode.limit == 0;
    if (cascade) OSFM_RETURN_IF(ensure_cashash(m));
    res->plans.assign(num_pairs, PairPlan());
    res->counts.assign(num_pairs, 0);
    std::vector<MatchProblem> probs[2];
    std::vector<SpecialJob> spjobs, spjobs_wide;
    struct SpEntry { int problem, side, view, units, nchunk; };
    std::vector<SpEntry> spentries;
    int64_t sp_recs = 0, sp_cols = 0;
    std::vector<int64_t> mark_off[2];
    int64_t out_ints = 0, rowpart_recs = 0, colpart_recs = 0, keep_bytes = 0, total_queries = 0;
    int total_blocks[2] = {0, 0}, max_n[2] = {0, 0};
    bool needs_mask[2] = {false, false};   // some problem is limited below its view size
    bool any_special[2] = {false, false};  // some problem has gathered special rows
    bool any_c0[2] = {false, false}, any_corrected[2] = {false, false};
    int64_t macs = 0, alg_bytes = 0, macs_surf = 0;
    int64_t cas_queries[2] = {0, 0};

    // Segment length of the correction-free problems: a workgroup walks seg_cols columns for its 256 rows.
    // Every segment costs a prologue (A fragments, two tiles), a row merge and a set of row partials, so a
    // launch with enough row blocks to fill the chip several times over takes one segment per row block (up to
    // kSegColsMax columns); a small launch (the per-pair entries) is cut into more, shorter ones.
    int64_t rb_estimate = 0;
    for (int p = 0; p < num_pairs; ++p) {
        const int v1 = pairs[p].view_1;
        if (v1 >= 0 && v1 < (int)m->views.size())
            rb_estimate += (std::max(m->views[v1].ns, m->views[v1].nu) + kRowsPerBlock - 1) / kRowsPerBlock;
    }
    static const int forced_seg_tiles = [] { const char *e = getenv("OSFM_SEG_TILES"); return e ? atoi(e) : 0; }();
    auto choose_seg_cols = [&](int n2) {
        const int ncyc = (n2 + kCycleCols - 1) / kCycleCols;
        int cyc;                                                   // cycles per segment
        if (forced_seg_tiles > 0) cyc = std::max(1, forced_seg_tiles / 16);     // measurements (tools/seg_intercept.sh)
        else {
            constexpr int64_t kWantBlocks = 2048;                  // four rounds of the 512 resident workgroups
            const int64_t want = (kWantBlocks + std::max<int64_t>(rb_estimate, 1) - 1) / std::max<int64_t>(rb_estimate, 1);
            const int nseg = (int)std::min<int64_t>(std::max<int64_t>(want, 1), ncyc);
            cyc = (ncyc + nseg - 1) / nseg;
        }
        cyc = std::min(cyc, kSegColsMax / kCycleCols);
        return cyc * kCycleCols;
    };

    for (int p = 0; p < num_pairs; ++p) {
        PairPlan &pl = res->plans[p];
        pl.v1 = pairs[p].view_1; pl.v2 = pairs[p].view_2;
        OSFM_RETURN_IF(check_view(m, pl.v1, "match"));
        OSFM_RETURN_IF(check_view(m, pl.v2, "match"));
        const ViewData &a = m->views[pl.v1], &b = m->views[pl.v2];
        pl.ns1 = a.ns; pl.nu1 = a.nu; pl.ns2 = b.ns; pl.nu2 = b.nu;
        pl.has_sift = a.ns > 0 && (mode.type_mask & 1);
        pl.has_surf = a.nu > 0 && (mode.type_mask & 2);
        // sfm::CascadeHashing leaves a type out of the Result altogether when view_2 lacks it
        // (cascade_hashing.h:341-342; ExhaustiveMatching keeps the -1 filled block)
        if (cascade && !m->opts.cascade_keep_empty_blocks) {
            if (b.ns == 0) pl.has_sift = false;
            if (b.nu == 0) pl.has_surf = false;
        }
        // exhaustive_matching.cc:153-177: SIFT takes precedence
        if (mode.lowres && pl.has_sift) pl.has_surf = false;
        if (lowres_limit > 0) {
            pl.ns1 = std::min(pl.ns1, lowres_limit); pl.ns2 = std::min(pl.ns2, lowres_limit);
            pl.nu1 = std::min(pl.nu1, lowres_limit); pl.nu2 = std::min(pl.nu2, lowres_limit);
        }
        const int e1 = pl.has_sift ? pl.ns1 : 0;      // sift entries on side 1 / 2
        const int e2 = pl.has_sift ? pl.ns2 : 0;
        pl.len12 = e1 + (pl.has_surf ? pl.nu1 : 0);
        pl.len21 = e2 + (pl.has_surf ? pl.nu2 : 0);
        pl.off12 = out_ints; out_ints += round_up(pl.len12, 4);
        pl.off21 = out_ints; out_ints += round_up(pl.len21, 4);
        pl.prob_index[0] = pl.prob_index[1] = -1;
        for (int type = 0; type < 2; ++type) {
            if (type == 0 ? !pl.has_sift : !pl.has_surf) continue;
            MatchProblem pr;
            memset(&pr, 0, sizeof(pr));
            pr.n1 = type == 0 ? pl.ns1 : pl.nu1;
            pr.n2 = type == 0 ? pl.ns2 : pl.nu2;
            pr.A = (type == 0 ? a.sift : a.surf).as<int8_t>();
            pr.B = (type == 0 ? b.sift : b.surf).as<int8_t>();
            pr.corrA = (type == 0 ? a.sift_corr : a.surf_corr).as<int32_t>();
            pr.corrB = (type == 0 ? b.sift_corr : b.surf_corr).as<int32_t>();
            // raw row operand: SURF bytes are the values already
            pr.A_raw = (type == 0 ? a.sift_raw : a.surf).as<int8_t>();
            pr.corrA_raw = (type == 0 ? a.sift_raw_corr : a.surf_corr).as<int32_t>();
            // a num_features-limited batch runs the masked kernel, which keeps every
            // row on the keyed (value-128) path
            const bool limited = lowres_limit > 0;
            const int n_special = (type == 0 && !limited) ? a.n_special : 0;
            pr.n_special = n_special;
            pr.A_special = a.special.as<int8_t>();
            pr.corrA_special = a.special_corr.as<int32_t>();
            pr.special_map = a.special_map.as<int32_t>();
            pr.special_slot = n_special > 0 ? a.special_slot.as<int32_t>() : nullptr;
            const bool empty = pr.n1 == 0 || pr.n2 == 0;
            if (limited) needs_mask[type] = true;
            // correction-free column operand: SURF bytes are the values; SIFT views
            // without a value > 127 (the raw copy then holds every descriptor)
            pr.B_raw = (type == 0 ? b.sift_raw : b.surf).as<int8_t>();
            pr.c0 = (type == 1 || b.n_special == 0) ? 1 : 0;
            // Few special descriptors on either side (the case real SIFT data produces,
            // sift.cc:830-839): they go through match_special_kernel and every other
            // descriptor of both views stays on the correction-free form.
            if (type == 0 && !limited && !cascade && !empty && (a.n_special > 0 || b.n_special > 0) &&
                a.n_special <= m->special_max && b.n_special <= m->special_max) {
                pr.sp = 1; pr.c0 = 1;
                pr.nsA = a.n_special; pr.nsB = b.n_special;
                pr.n_special = 0;                          // no special row blocks in the tile launch
                pr.special_slot = a.n_special > 0 ? a.special_slot.as<int32_t>() : nullptr;
                pr.B_special = b.special.as<int8_t>();
                pr.corrB_special = b.special_corr.as<int32_t>();
                pr.special_map_B = b.special_map.as<int32_t>();
                pr.special_slot_B = b.n_special > 0 ? b.special_slot.as<int32_t>() : nullptr;
                for (int side = 0; side < 2; ++side) {
                    const int ns = side == 0 ? pr.nsA : pr.nsB, no = side == 0 ? pr.n2 : pr.n1;
                    if (ns == 0) continue;
                    // more than kSpSlots units: the wide kernel (smaller chunks, shared through LDS)
                    const int units = (ns + 31) / 32;
                    pr.sp_wide[side] = units > kSpSlots ? 1 : 0;
                    const int chunk_cols = pr.sp_wide[side] ? kSpWideChunk : kSpChunk;
                    const int nchunk = (no + chunk_cols - 1) / chunk_cols;
                    pr.sp_row_off[side] = sp_recs; sp_recs += (int64_t)nchunk * round_up(ns, 32);
                    pr.sp_col_off[side] = sp_cols; sp_cols += round_up(no, 32);
                    spentries.push_back({(int)probs[type].size(), side, side == 0 ? pl.v2 : pl.v1, units, nchunk});
                }
            }
            if (!empty && !limited) (pr.c0 ? any_c0 : any_corrected)[type] = true;
            if (!empty && pr.n_special > 0) any_special[type] = true;
            pr.nrb_main = empty ? 0 : (pr.n1 + kRowsPerBlock - 1) / kRowsPerBlock;
            pr.nrb = pr.nrb_main + (empty ? 0 : (pr.n_special + kRowsPerBlock - 1) / kRowsPerBlock);
            // (only the RAW row blocks of a c0 problem run the correction-free kernel; its special row blocks, if
            //  any, run the keyed kernel, whose keys hold kSegCols / 32 fragment indices: such a problem keeps kSegCols)
            pr.seg_cols = (pr.c0 && !limited && pr.n_special == 0 && !empty) ? choose_seg_cols(pr.n2) : kSegCols;
            pr.nseg = empty ? 0 : (pr.n2 + pr.seg_cols - 1) / pr.seg_cols;
            pr.n2stride = round_up(pr.n2, kCycleCols);
            pr.block_start = total_blocks[type];
            if (cascade) {
                // no score tiles: the candidate search works on the hash data of the two views
                const ViewData *vd[2] = {&a, &b};
                for (int side = 0; side < 2; ++side) {
                    pr.cas_rec[side] = vd[side]->cas_rec[type].ptr;
                    pr.cas_start[side] = vd[side]->cas_start[type].as<int32_t>();
                    pr.cas_items[side] = vd[side]->cas_items[type].as<int32_t>();
                }
                pr.cas_state_off[0] = cas_queries[type]; cas_queries[type] += pr.n1;
                pr.cas_state_off[1] = cas_queries[type]; cas_queries[type] += pr.n2;
            } else {
                total_blocks[type] += pr.nrb * pr.nseg;
                pr.rowpart_off = rowpart_recs;
                rowpart_recs += (int64_t)pr.nseg * pr.nrb * kRowsPerBlock;
                pr.colpart_off = colpart_recs;
                colpart_recs += (int64_t)pr.nrb * pr.n2stride;
            }
            // placed by offset into m->out after allocation (store offsets now)
            pr.m12 = reinterpret_cast<int32_t *>(pl.off12 + (type == 0 ? 0 : e1));
            pr.m21 = reinterpret_cast<int32_t *>(pl.off21 + (type == 0 ? 0 : e2));
            // combine_results offsets (matching.cc:74-86) -- SURF entries only
            pr.out_off12 = (type == 1 && mode.apply) ? e2 : 0;
            pr.out_off21 = (type == 1 && mode.apply) ? e1 : 0;
            if (type == 1) {
                // exactness of the int32 fast path needs every 16-bit lane sum
                // in range: guaranteed when |q|^2 * |c|^2 <= 32767^2
                const long long bound = (long long)a.surf_norm2_max * (long long)b.surf_norm2_max;
                pr.force_exact = bound > 32767LL * 32767LL ? 1 : 0;
            }
            pl.prob_index[type] = (int)probs[type].size();
            mark_off[type].push_back(keep_bytes); keep_bytes += round_up(pr.n1, 16);
            mark_off[type].push_back(keep_bytes); keep_bytes += round_up(pr.n2, 16);
            max_n[type] = std::max(max_n[type], std::max(pr.n1, pr.n2));
            total_queries += pr.n1 + pr.n2;
            probs[type].push_back(pr);
            if (!empty) {
                const int dim = type == 0 ? 128 : 64;
                macs += (int64_t)pr.n1 * pr.n2 * dim;
                if (type == 1) macs_surf += (int64_t)pr.n1 * pr.n2 * dim;
                alg_bytes += (int64_t)(pr.n1 + pr.n2) * (dim + 4);
            }
        }
    }
    res->out_ints = out_ints;

    // --- scratch ---------------------------------------------------------
    // (a smaller call than the largest the caller has announced: the arrays in proportion, once -- the parts of a
    //  large call are bounded by batch_size_for, and so is the proportion)
    double grow = 1.0;
    if (mode.expect > num_pairs && num_pairs > 0) grow = 1.02 * (double)mode.expect / num_pairs;
    auto grown = [&](int64_t n) { return (size_t)((double)std::max<int64_t>(n, 1) * std::max(grow, 1.0)); };
    OSFM_RETURN_IF(m->out.reserve(grown(std::max<int64_t>(out_ints, 4)) * 4));
    OSFM_RETURN_IF(m->rowparts.reserve(grown(rowpart_recs) * sizeof(RowPart)));
    OSFM_RETURN_IF(m->colparts.reserve(grown(colpart_recs) * sizeof(ColPart)));
    OSFM_RETURN_IF(m->keep.reserve(grown(std::max<int64_t>(keep_bytes, 16))));
    OSFM_RETURN_IF(m->exact_items.reserve(grown(total_queries) * sizeof(ExactItem)));
    OSFM_RETURN_IF(m->exact_count.reserve(16));
    OSFM_RETURN_IF(m->sp_parts.reserve((size_t)std::max<int64_t>(sp_recs, 1) * sizeof(RowPart)));
    hipStream_t s = m->stream;
    {
        // the views this batch names: their uploads (own stream) before anything of the batch
        std::vector<uint8_t> seen(m->views.size(), 0);
        for (int p = 0; p < num_pairs; ++p)
            for (int v : {res->plans[p].v1, res->plans[p].v2})
                if (!seen[v]) { seen[v] = 1; if (m->views[v].ready) OSFM_HIP_CHECK(hipStreamWaitEvent(s, m->views[v].ready, 0)); }
    }
    // Workgroups that stream the same 4096 descriptors of the same view run next to each other:
    // the chunk (512 KB) is then fetched from HBM once per L2 instead of once per pair (in
    // pair order the streamed view of side 0 changes with every pair: 3 GB of HBM reads per
    // 1225 pairs for 128 MB of distinct descriptors).
    // Jobs of match_special_kernel: the entries (one side of one problem) grouped by the view they
    // stream -- up to kSpSlots one-unit entries share a workgroup, an entry with more units has workgroups of
    // its own -- and, per group, one job per chunk; groups of one view next to each other, so that a chunk
    // (512 KB) is fetched from HBM once per L2 and not once per pair.
    std::stable_sort(spentries.begin(), spentries.end(), [](const SpEntry &x, const SpEntry &y) {
        return x.view != y.view ? x.view < y.view : x.units < y.units;
    });
    for (size_t e = 0; e < spentries.size();) {
        SpecialJob j;
        memset(&j, 0, sizeof(j));
        const SpEntry &f = spentries[e];
        if (f.units > kSpSlots) {
            // the wide kernel: one job per chunk of kSpWideChunk candidates
            j.problem[0] = f.problem; j.side[0] = f.side; j.count = 0;
            for (int c = 0; c < f.nchunk; ++c) { j.chunk = c; spjobs_wide.push_back(j); }
            ++e;
            continue;
        }
        if (f.units > 1) {
            j.problem[0] = f.problem; j.side[0] = f.side; j.count = 0;
            ++e;
        } else {
            int c = 0;
            while (c < kSpSlots && e < spentries.size() && spentries[e].view == f.view && spentries[e].units == 1) {
                j.problem[c] = spentries[e].problem; j.side[c] = spentries[e].side; ++c; ++e;
            }
            j.count = c;
        }
        for (int c = 0; c < f.nchunk; ++c) { j.chunk = c; spjobs.push_back(j); }
    }
    OSFM_RETURN_IF(m->sp_col.reserve((size_t)std::max<int64_t>(sp_cols, 1) * 4));
    OSFM_RETURN_IF(m->d_spjobs.reserve(std::max<size_t>(spjobs.size() + spjobs_wide.size(), 1) * sizeof(SpecialJob)));
    if (!spjobs.empty())
        OSFM_HIP_CHECK(hipMemcpyAsync(m->d_spjobs.ptr, spjobs.data(), spjobs.size() * sizeof(SpecialJob),
            hipMemcpyHostToDevice, s));
    if (!spjobs_wide.empty())
        OSFM_HIP_CHECK(hipMemcpyAsync(m->d_spjobs.as<SpecialJob>() + spjobs.size(), spjobs_wide.data(),
            spjobs_wide.size() * sizeof(SpecialJob), hipMemcpyHostToDevice, s));
    int32_t *d_out = m->out.as<int32_t>();
    OSFM_HIP_CHECK(hipMemsetAsync(d_out, 0xff, (size_t)std::max<int64_t>(out_ints, 4) * 4, s));

    OSFM_HIP_CHECK(hipMemsetAsync(m->exact_count.ptr, 0, 16, s));
    bool timed[2] = {false, false};
    bool probed = false;
    for (int type = 0; type < 2; ++type) {
        const int np = (int)probs[type].size();
        if (np == 0) continue;
        for (auto &pr : probs[type]) {
            pr.m12 = d_out + reinterpret_cast<intptr_t>(pr.m12);
            pr.m21 = d_out + reinterpret_cast<intptr_t>(pr.m21);
        }
        OSFM_RETURN_IF(m->d_problems[type].reserve(np * sizeof(MatchProblem)));
        OSFM_RETURN_IF(m->mark_off[type].reserve(np * 2 * sizeof(int64_t)));
        OSFM_RETURN_IF(m->counts[type].reserve(np * sizeof(int32_t)));
        OSFM_HIP_CHECK(hipMemcpyAsync(m->d_problems[type].ptr, probs[type].data(),
            np * sizeof(MatchProblem), hipMemcpyHostToDevice, s));
        OSFM_HIP_CHECK(hipMemcpyAsync(m->mark_off[type].ptr, mark_off[type].data(),
            np * 2 * sizeof(int64_t), hipMemcpyHostToDevice, s));
        OSFM_HIP_CHECK(hipMemsetAsync(m->counts[type].ptr, 0, np * sizeof(int32_t), s));
        const MatchProblem *dp = m->d_problems[type].as<MatchProblem>();
        const LoweTable tab = type == 0 ? m->tab_sift : m->tab_surf;
        const int ecap = (int)std::min<int64_t>(total_queries, 0x7fffffff);
        int32_t *ecount = m->exact_count.as<int32_t>() + type;

        if (cascade) {
            OSFM_HIP_CHECK(hipEventRecord(m->ev[type][0], s));
            timed[type] = true;
            OSFM_RETURN_IF(m->cas_state.reserve((size_t)std::max<int64_t>(cas_queries[type], 1) * kCasMaxCand * 4));
            launch_cashash_match(type == 0 ? 128 : 64, dp, np, max_n[type], m->cas_state.as<int32_t>(), tab, s);
            OSFM_HIP_CHECK(hipEventRecord(m->ev[type][1], s));
        } else {
            if (total_blocks[type] > 0) { OSFM_HIP_CHECK(hipEventRecord(m->ev[type][0], s)); timed[type] = true; }
            unsigned long long *probe = nullptr;
            if (type == 0 && mode.limit == 0) {
                OSFM_RETURN_IF(m->clock_probe.reserve(16));
                OSFM_HIP_CHECK(hipMemsetAsync(m->clock_probe.ptr, 0, 16, s));
                probe = m->clock_probe.as<unsigned long long>();
                probed = true;
            }
            if (!m->zero_tile.ptr) {
                OSFM_RETURN_IF(m->zero_tile.reserve((size_t)kTileCols * 128));
                OSFM_HIP_CHECK(hipMemsetAsync(m->zero_tile.ptr, 0, (size_t)kTileCols * 128, s));
            }
            launch_match_tiles(type == 0 ? 8 : 4, needs_mask[type], any_special[type], any_c0[type],
                any_corrected[type], dp, np, total_blocks[type],
                m->rowparts.as<RowPart>(), m->colparts.as<ColPart>(), s, probe, m->zero_tile.as<int8_t>());
            if (timed[type]) OSFM_HIP_CHECK(hipEventRecord(m->ev[type][1], s));
            if (type == 0 && !(spjobs.empty() && spjobs_wide.empty())) {
                OSFM_HIP_CHECK(hipEventRecord(m->ev_sp[0], s));
                launch_match_special(dp, m->d_spjobs.as<SpecialJob>(), (int)spjobs.size(), m->sp_parts.as<RowPart>(),
                    m->sp_col.as<int32_t>(), s);
                launch_match_special_wide(dp, m->d_spjobs.as<SpecialJob>() + spjobs.size(), (int)spjobs_wide.size(),
                    m->sp_parts.as<RowPart>(), m->sp_col.as<int32_t>(), s);
                OSFM_HIP_CHECK(hipEventRecord(m->ev_sp[1], s));
            }
            launch_match_finish(dp, np, max_n[type], m->rowparts.as<RowPart>(),
                m->colparts.as<ColPart>(), m->sp_parts.as<RowPart>(), m->sp_col.as<int32_t>(), tab, 0,
                m->exact_items.as<ExactItem>(), ecount, ecap, s);
            launch_exact_scan(type == 0 ? 128 : 64, dp, m->exact_items.as<ExactItem>(), ecount, ecap,
                tab, s);
        }
        launch_cross_check_mark(dp, np, max_n[type], m->keep.as<uint8_t>(), m->keep.as<uint8_t>(),
            m->mark_off[type].as<int64_t>(), m->counts[type].as<int32_t>(), s);
        if (mode.apply)
            launch_cross_check_apply(dp, np, max_n[type], m->keep.as<uint8_t>(),
                m->keep.as<uint8_t>(), m->mark_off[type].as<int64_t>(), s);
        OSFM_HIP_CHECK(hipGetLastError());
    }

    // --- counts back -------------------------------------------------------
    std::vector<int32_t> hc[2];
    int32_t hexact[4] = {0, 0, 0, 0};
    for (int type = 0; type < 2; ++type) {
        hc[type].resize(probs[type].size());
        if (!hc[type].empty())
            OSFM_HIP_CHECK(hipMemcpyAsync(hc[type].data(), m->counts[type].ptr,
                hc[type].size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    OSFM_HIP_CHECK(hipMemcpyAsync(hexact, m->exact_count.ptr, 16, hipMemcpyDeviceToHost, s));
    unsigned long long hprobe[2] = {0, 0};
    if (probed) OSFM_HIP_CHECK(hipMemcpyAsync(hprobe, m->clock_probe.ptr, 16, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    m->stats.tile_shader_cycles += (double)hprobe[0];
    m->stats.tile_refclk_ticks += (double)hprobe[1];
    for (int p = 0; p < num_pairs; ++p) {
        int c = 0;
        for (int type = 0; type < 2; ++type)
            if (res->plans[p].prob_index[type] >= 0) c += hc[type][res->plans[p].prob_index[type]];
        res->counts[p] = c;
    }
    for (int type = 0; type < 2; ++type) {
        if (!timed[type]) continue;
        float ms = 0.f;
        OSFM_HIP_CHECK(hipEventElapsedTime(&ms, m->ev[type][0], m->ev[type][1]));
        if (cascade) {
            m->stats.cashash_kernel_ms += ms;
            m->stats.cashash_kernel_launches += 1;
        } else if (mode.limit > 0) {
            m->stats.lowres_kernel_ms += ms;
            m->stats.lowres_kernel_launches += 1;
        } else {
            m->stats.tile_kernel_ms += ms;
            m->stats.tile_kernel_launches += 1;
            if (type == 1) { m->stats.surf_tile_kernel_ms += ms; m->stats.surf_tile_kernel_launches += 1; }
        }
    }
    if (!(spjobs.empty() && spjobs_wide.empty())) {
        float ms = 0.f;
        OSFM_HIP_CHECK(hipEventElapsedTime(&ms, m->ev_sp[0], m->ev_sp[1]));
        m->stats.special_kernel_ms += ms;
        m->stats.special_kernel_launches += 1;
    }
    m->stats.exact_scan_queries += hexact[0] + hexact[1];
    if (cascade) {
        // no dense products in this mode
    } else if (mode.limit > 0) {
        m->stats.lowres_mac_count += macs;
    } else {
        m->stats.mac_count += macs;
        m->stats.surf_mac_count += macs_surf;
        m->stats.algorithmic_bytes += alg_bytes;
    }
    return OSFM_OK;
}

int upload_view(osfm_matcher *m, int view, const uint16_t *sift, int n_sift, const int16_t *surf,
    int n_surf)
{
    if (view < 0 || view >= (int)m->views.size()) {
        set_error("set_view: view id %d out of range", view);
        return OSFM_E_ARG;
    }
    if (n_sift < 0 || n_surf < 0 || (n_sift > 0 && !sift) || (n_surf > 0 && !surf)) {
        set_error("set_view: null descriptors / negative count");
        return OSFM_E_ARG;
    }
    ViewData &v = m->views[view];
    hipStream_t s = m->up_stream;
    v.set = false;
    v.ns = n_sift; v.nu = n_surf;
    v.ns_pad = round_up(std::max(n_sift, 1), kRowsPerBlock);
    v.nu_pad = round_up(std::max(n_surf, 1), kRowsPerBlock);
    OSFM_RETURN_IF(v.sift.reserve((size_t)v.ns_pad * 128));
    OSFM_RETURN_IF(v.sift_corr.reserve((size_t)v.ns_pad * 4));
    OSFM_RETURN_IF(v.sift_raw.reserve((size_t)v.ns_pad * 128));
    OSFM_RETURN_IF(v.sift_raw_corr.reserve((size_t)v.ns_pad * 4));
    OSFM_RETURN_IF(v.surf.reserve((size_t)v.nu_pad * 64));
    OSFM_RETURN_IF(v.surf_corr.reserve((size_t)v.nu_pad * 4));
    // One pass over the caller's descriptors on the host: range check (an error leaves the view
    // unset before anything is queued), the list of special rows, the largest SURF norm -- while
    // copying them into a page-locked block (two alternate, guarded by events).  Everything behind
    // that is queued on the stream and NOT waited for: the next view's host pass runs while this
    // view's transfer and conversion kernels do (200 views x 20k: 0.29 -> 0.12 s).
    const size_t b_sift = (size_t)n_sift * 128 * 2, b_surf = (size_t)n_surf * 64 * 2;
    const size_t b_ints = (size_t)n_sift * 4 * 2;                       // special list + slot map behind the descriptors
    const size_t need = std::max<size_t>(b_sift + b_surf + b_ints, 64);
    const int ps = m->pin_next++ & 1;
    if (m->pin_used[ps]) OSFM_HIP_CHECK(hipEventSynchronize(m->pin_ev[ps]));
    if (m->pin_bytes[ps] < need) {
        pinned_free(m->pin_ptr[ps], m->pin_bytes[ps]);
        m->pin_ptr[ps] = nullptr; m->pin_bytes[ps] = 0;
        OSFM_HIP_CHECK(pinned_alloc(reinterpret_cast<void **>(&m->pin_ptr[ps]), need + need / 4, hipHostMallocDefault));
        m->pin_bytes[ps] = need + need / 4;
    }
    char *pin = m->pin_ptr[ps];
    uint16_t *p_sift = reinterpret_cast<uint16_t *>(pin);
    int16_t *p_surf = reinterpret_cast<int16_t *>(pin + b_sift);
    int32_t *p_special = reinterpret_cast<int32_t *>(pin + b_sift + b_surf), *p_slot = p_special + n_sift;
    int n_special = 0;
    bool bad = false;
    {
        // copy + row maxima on a few threads (the pass is memory bound: 5 MB per 20k-feature view), then the
        // special rows numbered in ascending order: the same gathered set on every run
        auto rows = [&](int lo, int hi, int *bad_out) {
            int b = 0;
            for (int i = lo; i < hi; ++i) {
                const uint16_t *d = sift + (size_t)i * 128;
                uint16_t *o = p_sift + (size_t)i * 128;
                unsigned mx = 0;
                for (int k = 0; k < 128; ++k) { o[k] = d[k]; mx = std::max<unsigned>(mx, d[k]); }
                b |= mx > 255;
                p_slot[i] = mx > 127 ? 1 : 0;
            }
            *bad_out = b;
        };
        static const int max_thr = getenv("OSFM_UPLOAD_THREADS") ? std::max(1, std::min(4, atoi(getenv("OSFM_UPLOAD_THREADS")))) : 4;
        const int nthr = n_sift >= 8192 ? max_thr : 1;
        int bad_t[4] = {0, 0, 0, 0};
        std::vector<std::thread> th;
        for (int t = 1; t < nthr; ++t)
            th.emplace_back(rows, (int)((int64_t)n_sift * t / nthr), (int)((int64_t)n_sift * (t + 1) / nthr), &bad_t[t]);
        rows(0, (int)((int64_t)n_sift / nthr), &bad_t[0]);
        for (auto &t : th) t.join();
        for (int t = 0; t < nthr; ++t) bad |= bad_t[t] != 0;
        for (int i = 0; i < n_sift; ++i) {
            if (p_slot[i]) { p_slot[i] = n_special; p_special[n_special++] = i; }
            else p_slot[i] = -1;
        }
    }
    long long norm2_max = 0;
    for (int i = 0; i < n_surf; ++i) {
        const int16_t *d = surf + (size_t)i * 64;
        int16_t *o = p_surf + (size_t)i * 64;
        long long n2 = 0;
        for (int k = 0; k < 64; ++k) { o[k] = d[k]; bad |= d[k] > 127 || d[k] < -128; n2 += (long long)d[k] * d[k]; }
        norm2_max = std::max(norm2_max, n2);
    }
    if (bad) {
        set_error("set_view: descriptor value outside the quantised range "
                  "(SIFT 0..255, SURF -128..127) in view %d", view);
        return OSFM_E_RANGE;
    }
    OSFM_RETURN_IF(m->stage_in.reserve(std::max<size_t>(b_sift + b_surf, 16)));
    OSFM_RETURN_IF(m->flags.reserve(16));
    char *stage = m->stage_in.as<char>();
    if (b_sift + b_surf) OSFM_HIP_CHECK(hipMemcpyAsync(stage, pin, b_sift + b_surf, hipMemcpyHostToDevice, s));
    int32_t *flags = m->flags.as<int32_t>();       // written by the kernels' own range checks; the host pass above has decided
    launch_prepare_sift(reinterpret_cast<const uint16_t *>(stage), n_sift, v.ns_pad,
        v.sift.as<int8_t>(), v.sift_corr.as<int32_t>(), v.sift_raw.as<int8_t>(),
        v.sift_raw_corr.as<int32_t>(), flags + 0, s);
    v.n_special = n_special;
    if (v.n_special > 0) {
        const int sp_pad = round_up(v.n_special, kRowsPerBlock);
        OSFM_RETURN_IF(v.special.reserve((size_t)sp_pad * 128));
        OSFM_RETURN_IF(v.special_corr.reserve((size_t)sp_pad * 4));
        OSFM_RETURN_IF(v.special_map.reserve((size_t)v.n_special * 4));
        OSFM_RETURN_IF(v.special_slot.reserve((size_t)n_sift * 4));
        OSFM_HIP_CHECK(hipMemcpyAsync(v.special_map.ptr, p_special, (size_t)v.n_special * 4, hipMemcpyHostToDevice, s));
        OSFM_HIP_CHECK(hipMemcpyAsync(v.special_slot.ptr, p_slot, (size_t)n_sift * 4, hipMemcpyHostToDevice, s));
        // zero-vector padding in the value-128 form: bytes -128, correction -2^20 - 2^22
        launch_gather_rows(v.sift.as<int8_t>(), v.sift_corr.as<int32_t>(), v.special_map.as<int32_t>(),
            v.n_special, sp_pad, 128, -(1 << 20) - (1 << 22), (int8_t)-128, v.special.as<int8_t>(),
            v.special_corr.as<int32_t>(), s);
    }
    launch_prepare_surf(reinterpret_cast<const int16_t *>(stage + b_sift), n_surf, v.nu_pad,
        v.surf.as<int8_t>(), v.surf_corr.as<int32_t>(), flags + 1, flags + 2, s);
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipEventRecord(m->pin_ev[ps], s));
    m->pin_used[ps] = true;
    if (!v.ready) OSFM_HIP_CHECK(event_create(&v.ready, false));
    OSFM_HIP_CHECK(hipEventRecord(v.ready, s));
    v.surf_norm2_max = (int)std::min<long long>(norm2_max, 0x7fffffff);
    v.set = true;
    m->cas_dirty = true;           // the cascade hashes depend on the average over all views
    return OSFM_OK;
}

// CascadeHashing::init (cascade_hashing.cc:33-70) for the views set so far.
// The projection matrices are drawn on the host with the same standard-library
// facilities the reference uses (std::mt19937(0) + std::normal_distribution<>,
// cascade_hashing.h:225-254): the distribution's algorithm is the library's,
// so the values equal the reference's when both are built against libstdc++.
static int build_cashash(osfm_matcher *m);

// The average descriptor -- and with it every hash -- is taken over ALL views (CascadeHashing::init receives the whole
// viewport list, cascade_hashing.cc:33-70), so a cascade batch needs the complete bank: a slot that was never set
// is an error, not an empty view, and no upload may run beside the build (uploads mutate the views under up_mu; the
// flag is cleared BEFORE the build so that an upload that follows raises it again and is not lost).
int ensure_cashash(osfm_matcher *m)
{
    std::lock_guard<std::mutex> lock(m->up_mu);
    for (size_t v = 0; v < m->views.size(); ++v)
        if (!m->views[v].set) {
            set_error("cascade hashing: view %d of %d has not been set (the hashes depend on the average descriptor of ALL views)",
                (int)v, (int)m->views.size());
            return OSFM_E_STATE;
        }
    if (!m->cas_dirty.exchange(false)) return OSFM_OK;
    const int rc = build_cashash(m);
    if (rc != OSFM_OK) m->cas_dirty = true;
    return rc;
}

static int build_cashash(osfm_matcher *m)
{
    OSFM_HIP_CHECK(hipStreamSynchronize(m->up_stream));       // the hashes are built from every view
    hipStream_t s = m->stream;
    for (int type = 0; type < 2; ++type) {
        const int dim = type == 0 ? 128 : 64;
        const int np = dim + kCasSecBits;
        if (!m->cas_proj[type].ptr) {
            std::mt19937 prng(0);
            std::normal_distribution<> dis(0, 1);
            std::vector<float> projT((size_t)dim * np);
            for (int i = 0; i < dim; ++i)                    // primary: prim_hash[i][j]
                for (int j = 0; j < dim; ++j) projT[(size_t)j * np + i] = (float)dis(prng);
            for (int g = 0; g < kCasGroups; ++g)             // secondary: sec_hash[g][i][j]
                for (int i = 0; i < kCasBits; ++i)
                    for (int j = 0; j < dim; ++j) projT[(size_t)j * np + dim + g * kCasBits + i] = (float)dis(prng);
            OSFM_RETURN_IF(m->cas_proj[type].reserve(projT.size() * 4));
            OSFM_HIP_CHECK(hipMemcpy(m->cas_proj[type].ptr, projT.data(), projT.size() * 4, hipMemcpyHostToDevice));
            OSFM_RETURN_IF(m->cas_sum[type].reserve(128 * 4));
            OSFM_RETURN_IF(m->cas_avg[type].reserve(128 * 4));
        }
        // compute_avg_descriptors: one running float sum per dimension over all views in order
        OSFM_HIP_CHECK(hipMemsetAsync(m->cas_sum[type].ptr, 0, 128 * 4, s));
        int64_t total = 0;
        for (auto &v : m->views) {
            if (!v.set) continue;
            const int n = type == 0 ? v.ns : v.nu;
            total += n;
            launch_cashash_accumulate((type == 0 ? v.sift : v.surf).as<int8_t>(), n, dim, type == 0 ? 128 : 0,
                type == 0 ? 255.0f : 127.0f, m->cas_sum[type].as<float>(), s);
        }
        launch_cashash_average(m->cas_sum[type].as<float>(), dim, total, m->cas_avg[type].as<float>(), s);
        for (auto &v : m->views) {
            if (!v.set) continue;
            const int n = type == 0 ? v.ns : v.nu;
            if (n >= (1 << 17)) {
                // candidate keys hold the position inside a bucket list in 17 bits
                set_error("cascade hashing: %d descriptors of one type in a view, at most %d supported", n, (1 << 17) - 1);
                return OSFM_E_RANGE;
            }
            OSFM_RETURN_IF(v.cas_hash[type].reserve((size_t)std::max(n, 1) * (dim / 64) * 8));
            OSFM_RETURN_IF(v.cas_bucket[type].reserve((size_t)std::max(n, 1) * kCasGroups));
            OSFM_RETURN_IF(v.cas_start[type].reserve((size_t)kCasGroups * (kCasBuckets + 1) * 4));
            OSFM_RETURN_IF(v.cas_items[type].reserve((size_t)std::max(n, 1) * kCasGroups * 4));
            launch_cashash_hash((type == 0 ? v.sift : v.surf).as<int8_t>(), n, dim, type == 0 ? 128 : 0,
                type == 0 ? 255.0f : 127.0f, m->cas_avg[type].as<float>(), m->cas_proj[type].as<float>(),
                v.cas_hash[type].as<uint64_t>(), v.cas_bucket[type].as<uint8_t>(), s);
            launch_cashash_buckets(v.cas_bucket[type].as<uint8_t>(), n, v.cas_start[type].as<int32_t>(),
                v.cas_items[type].as<int32_t>(), s);
            OSFM_RETURN_IF(v.cas_rec[type].reserve((size_t)std::max(n, 1) * sizeof(CasRecord)));
            launch_cashash_pack(v.cas_hash[type].as<uint64_t>(), v.cas_bucket[type].as<uint8_t>(), n, dim / 64,
                static_cast<CasRecord *>(v.cas_rec[type].ptr), s);
        }
    }
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    return OSFM_OK;
}

void reset_stats(osfm_matcher *m) { memset(&m->stats, 0, sizeof(m->stats)); }

}  // namespace

// ---------------------------------------------------------------------------
// Multi-device front.  The reference's caller is ONE C++ process
// (bundler::Matching::compute, bundler_matching.cc:58-136), so the sharding over the GPUs of
// a node has to live behind the C ABI: one worker thread per shard for the duration of a
// call, every shard a complete single-device matcher holding the full descriptor bank, the
// pairs of osfm_match_all dealt by work (N1 * N2, longest first, to the least loaded shard:
// the rule of orthosfm_amd/distributed.py::deal_pairs), no data-path exchange between
// devices.  Each shard leaves its lists in its own page-locked block; once every count is
// known the offsets follow in pair order and the shard threads copy their lists to their
// places in the caller's buffer -- records and bytes are those of a single-device call.
// ---------------------------------------------------------------------------
namespace {

enum { kPhaseBoth = 0, kPhaseGate = 1, kPhaseFull = 2 };
int match_all_single(osfm_matcher *m, const osfm_pair *pairs, int num_pairs, osfm_pair_result *results,
    int32_t *corr, int64_t capacity, int64_t *total, int phase);

template <class F>
int for_each_shard(osfm_matcher *m, F &&fn)
{
    const size_t n = m->shards.size();
    std::vector<int> st(n, OSFM_OK);
    std::vector<std::string> msg(n);
    auto run = [&](size_t k) {
        st[k] = fn(k);
        if (st[k] != OSFM_OK) msg[k] = osfm_last_error();      // the error text is per thread
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; ++k) th.emplace_back(run, k);
    run(0);
    for (auto &t : th) t.join();
    // a failure is reported once: the first failing shard's status and text
    for (size_t k = 0; k < n; ++k)
        if (st[k] != OSFM_OK) {
            set_error("device %d (shard %zu of %zu): %s", m->shards[k]->device, k, n, msg[k].c_str());
            return st[k];
        }
    return OSFM_OK;
}

// Longest processing time first on N1 * N2 over the pairs `cand` (ascending pair indices); ties go to the lower
// shard.  Equal work everywhere degenerates to round robin, as in distributed.py.
void deal_pairs_lpt(const osfm_matcher *m, const osfm_pair *pairs, const std::vector<int> &cand, std::vector<std::vector<int>> *owner)
{
    const int n = (int)m->shards.size();
    owner->assign(n, {});
    const auto &views = m->shards[0]->views;
    const int nc = (int)cand.size();
    std::vector<int64_t> w(nc);
    bool uniform = true;
    for (int i = 0; i < nc; ++i) {
        const ViewData &a = views[pairs[cand[i]].view_1], &b = views[pairs[cand[i]].view_2];
        w[i] = (int64_t)(a.ns + a.nu) * (int64_t)(b.ns + b.nu);
        uniform &= w[i] == w[0];
    }
    if (uniform) {
        for (int i = 0; i < nc; ++i) (*owner)[i % n].push_back(cand[i]);
        return;
    }
    std::vector<int> order(nc);
    for (int i = 0; i < nc; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return w[x] > w[y]; });
    typedef std::pair<int64_t, int> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int k = 0; k < n; ++k) heap.push({0, k});
    std::vector<int> own(nc);
    for (int i : order) {
        Load l = heap.top(); heap.pop();
        own[i] = l.second;
        heap.push({l.first + w[i], l.second});
    }
    for (int i = 0; i < nc; ++i) (*owner)[own[i]].push_back(cand[i]);      // ascending inside a shard
}

// Two rounds, each dealt by what it costs.  The low-res gate (bundler_matching.cc:146-158: 500 x 500 features per
// pair, 1 / 1600 of a full match) goes round robin over ALL pairs; what it lets through -- on real image sets the
// survivors follow the scene, not the pair index: views of another scene reject whole rows of the pair matrix --
// is then dealt longest-first on N1 * N2 for the full matching.  (One deal before the gate balanced the weights
// of pairs of which a structured third was never matched.)  The shard threads live through both rounds and the
// copy of the lists: a barrier between the rounds (the survivors of all shards are dealt together) and one behind
// the counts (a list's place in the caller's buffer follows from the counts of every pair before it).
int multi_match_all(osfm_matcher *m, const osfm_pair *pairs, int num_pairs, osfm_pair_result *results,
    int32_t *corr, int64_t capacity, int64_t *total)
{
    std::lock_guard<std::mutex> lock(m->multi_mu);
    const int n = (int)m->shards.size();
    for (int p = 0; p < num_pairs; ++p) {
        OSFM_RETURN_IF(check_view(m->shards[0], pairs[p].view_1, "match_all"));
        OSFM_RETURN_IF(check_view(m->shards[0], pairs[p].view_2, "match_all"));
    }
    const auto &views = m->shards[0]->views;
    const bool verify = m->opts.geometric_verification != 0;
    std::vector<std::vector<int>> owner(n);                   // round 2: pairs of each shard, ascending
    std::vector<std::vector<osfm_pair>> sub_pairs(n);
    std::vector<std::vector<osfm_pair_result>> sub_res(n);
    std::vector<int64_t> sub_total(n, 0), sub_cap(n, 0);
    std::vector<int> sub_status(n, OSFM_OK);
    std::vector<int64_t> goff(num_pairs, 0);                  // a pair's list inside its shard's block
    int64_t run = 0;
    bool overflow = false;

    // a barrier for the n shard threads; the last one to arrive runs `serial` before anyone leaves
    std::mutex bmu;
    std::condition_variable bcv;
    int waiting = 0, generation = 0;
    bool failed = false;                                      // a shard has failed: the others stop at the next barrier
    auto barrier = [&](bool ok, const std::function<void()> &serial) -> bool {
        std::unique_lock<std::mutex> lk(bmu);
        if (!ok) failed = true;
        if (++waiting == n) {
            if (!failed) serial();
            waiting = 0; ++generation;
            bcv.notify_all();
        } else {
            const int gen = generation;
            bcv.wait(lk, [&] { return generation != gen; });
        }
        return !failed;
    };

    const int st = for_each_shard(m, [&](size_t k) -> int {
        // ---- round 1: the gate of every n-th pair ----
        std::vector<osfm_pair> gp;
        std::vector<int> gidx;
        for (int p = (int)k; p < num_pairs; p += n) { gp.push_back(pairs[p]); gidx.push_back(p); }
        std::vector<osfm_pair_result> gres(gp.size());
        int r = match_all_single(m->shards[k], gp.data(), (int)gp.size(), gres.data(), nullptr, 0, nullptr, kPhaseGate);
        if (r == OSFM_OK) for (size_t i = 0; i < gidx.size(); ++i) results[gidx[i]] = gres[i];
        if (!barrier(r == OSFM_OK, [&] {
                // the survivors of all shards, dealt by the work they are
                std::vector<int> cand;
                for (int p = 0; p < num_pairs; ++p) if (results[p].status == OSFM_PAIR_MATCHED) cand.push_back(p);
                deal_pairs_lpt(m, pairs, cand, &owner);
                for (int q = 0; q < n; ++q) {
                    int64_t bound = 0;      // a pair has at most min(features of either view) mutual matches
                    for (int p : owner[q]) {
                        sub_pairs[q].push_back(pairs[p]);
                        sub_res[q].push_back(results[p]);
                        const ViewData &a = views[pairs[p].view_1], &b = views[pairs[p].view_2];
                        bound += std::min(a.ns + a.nu, b.ns + b.nu);
                    }
                    sub_cap[q] = std::min(bound, capacity);
                }
            }))
            return r;
        // ---- round 2: full matching (and RANSAC-F) of this shard's survivors into its own page-locked block ----
        osfm_matcher::Staging &sg = m->shard_stage[k];
        const size_t need = (size_t)std::max<int64_t>(sub_cap[k], 1) * 2;
        r = OSFM_OK;
        if (sg.ints < need) {
            pinned_free(sg.ptr, sg.ints * 4);
            sg.ptr = nullptr; sg.ints = 0;
            if (pinned_alloc(reinterpret_cast<void **>(&sg.ptr), (need + need / 8) * 4, hipHostMallocPortable) != hipSuccess) {
                set_error("match_all: no page-locked memory for the lists of device %d", m->shards[k]->device);
                r = OSFM_E_DEVICE;
            } else sg.ints = need + need / 8;
        }
        if (r == OSFM_OK) {
            r = match_all_single(m->shards[k], sub_pairs[k].data(), (int)sub_pairs[k].size(), sub_res[k].data(),
                sg.ptr, sub_cap[k], &sub_total[k], kPhaseFull);
            // a shard that runs out of room has still classified and counted every pair
            sub_status[k] = r;
            if (r == OSFM_E_CAPACITY) r = OSFM_OK;
        }
        if (!barrier(r == OSFM_OK, [&] {
                // every count is known: the lists' places in the caller's buffer, in pair order
                std::vector<size_t> pos(n, 0);
                std::vector<int> own(num_pairs, -1);
                for (int q = 0; q < n; ++q) for (int p : owner[q]) own[p] = q;
                for (int p = 0; p < num_pairs; ++p) {
                    const int q = own[p];
                    if (q < 0) continue;                       // stopped at the gate (or empty): as round 1 left it
                    results[p] = sub_res[q][pos[q]++];
                    const int64_t local = results[p].offset;
                    results[p].offset = 0;
                    if (results[p].status == OSFM_PAIR_MATCHED) {
                        results[p].offset = run;
                        goff[p] = local;
                        run += verify ? results[p].num_inliers : results[p].num_matches;
                    }
                }
                overflow = run > capacity;
                for (int q = 0; q < n; ++q) overflow |= sub_status[q] == OSFM_E_CAPACITY;
            }))
            return r;
        if (overflow) return OSFM_OK;
        // ---- this shard's lists to their places ----
        const int32_t *src = sg.ptr;
        for (int p : owner[k]) {
            if (results[p].status != OSFM_PAIR_MATCHED) continue;
            const int64_t cnt = verify ? results[p].num_inliers : results[p].num_matches;
            if (cnt > 0) memcpy(corr + 2 * results[p].offset, src + 2 * goff[p], (size_t)cnt * 8);
        }
        return OSFM_OK;
    });
    if (st != OSFM_OK) return st;
    if (total) {
        // on overflow every shard reports what it needs: the required total
        int64_t need = 0;
        for (int k = 0; k < n; ++k) need += sub_total[k];
        *total = overflow ? need : run;
    }
    if (overflow) {
        set_error("match_all: %lld correspondences need more than the given capacity %lld",
            (long long)(total ? *total : run), (long long)capacity);
        return OSFM_E_CAPACITY;
    }
    return OSFM_OK;
}

}  // namespace

extern "C" {

const char *osfm_last_error(void) { return g_last_error.c_str(); }
int osfm_version(void) { return OSFM_ABI_VERSION; }

int osfm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int osfm_ransac_selfcheck(int mode, uint64_t *counters)
{
    if (mode < 0 || mode > 2) { set_error("ransac_selfcheck: mode %d", mode); return OSFM_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("ransac_selfcheck: no HIP device available");
        return OSFM_E_DEVICE;
    }
    unsigned long long c[3] = {0, 0, 0};
    OSFM_RETURN_IF(ransac_set_mode(mode, c));
    if (counters) for (int i = 0; i < 3; ++i) counters[i] = c[i];
    return OSFM_OK;
}

int osfm_trim_device_memory(int device, uint64_t *released_bytes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("trim_device_memory: no HIP device available");
        return OSFM_E_DEVICE;
    }
    if (device < -1 || device >= ndev) { set_error("trim_device_memory: device %d out of range [-1,%d)", device, ndev); return OSFM_E_ARG; }
    const size_t b = g_device_pool.trim(device);
    if (released_bytes) *released_bytes = b;
    return OSFM_OK;
}

int osfm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("device_memory: no HIP device available");
        return OSFM_E_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device_memory: device %d out of range [0,%d)", device, ndev); return OSFM_E_ARG; }
    OSFM_HIP_CHECK(hipSetDevice(device));
    size_t f = 0, t = 0;
    OSFM_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return OSFM_OK;
}

int osfm_library_memory(osfm_memory_report *out)
{
    if (!out) { set_error("library_memory: null"); return OSFM_E_ARG; }
    out->device_buffer_bytes = g_ledger.device_buffer_bytes.load();
    out->pool_live_bytes = g_ledger.pool_live_bytes.load();
    out->pool_cached_bytes = g_ledger.pool_cached_bytes.load();
    out->pinned_host_bytes = g_ledger.pinned_host_bytes.load();
    out->live_matchers = g_ledger.live_matchers.load();
    out->live_streams = g_ledger.live_streams.load();
    out->live_events = g_ledger.live_events.load();
    out->reserved = 0;
    return OSFM_OK;
}

int osfm_match_options_default(osfm_match_options *o)
{
    if (!o) { set_error("options_default: null"); return OSFM_E_ARG; }
    o->sift_lowe_ratio = 0.8f;
    o->sift_distance_threshold = FLT_MAX;
    o->surf_lowe_ratio = 0.7f;
    o->surf_distance_threshold = FLT_MAX;
    o->use_lowres_matching = 1;
    o->num_lowres_features = 500;
    o->min_lowres_matches = 5;
    o->min_feature_matches = 50;
    o->pairs_per_batch = 0;
    o->geometric_verification = 0;
    o->ransac_max_iterations = 1000;
    o->ransac_threshold = 0.0015;
    o->min_matching_inliers = 30;
    o->matcher_type = OSFM_MATCHER_EXHAUSTIVE;
    o->ransac_seed = 0;
    o->cascade_keep_empty_blocks = 0;
    o->special_kernel_max = 0;
    return OSFM_OK;
}

int osfm_match_create(int device, int num_views, const osfm_match_options *opts, osfm_matcher **out)
{
    if (!out || num_views < 0) { set_error("match_create: bad arguments"); return OSFM_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("match_create: no HIP device available (this backend has no CPU fallback)");
        return OSFM_E_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("match_create: device %d out of range [0,%d)", device, ndev);
        return OSFM_E_ARG;
    }
    OSFM_HIP_CHECK(hipSetDevice(device));
    // owned until the last step succeeded: an error return frees everything made so far
    struct Owner { osfm_matcher *p; ~Owner() { if (p) osfm_match_destroy(p); } } owner{(g_ledger.live_matchers++, new osfm_matcher())};
    osfm_matcher *m = owner.p;
    m->device = device;
    if (opts) m->opts = *opts; else osfm_match_options_default(&m->opts);
    m->views = std::vector<ViewData>(num_views);
    reset_stats(m);
    OSFM_HIP_CHECK(stream_create(&m->stream));
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) OSFM_HIP_CHECK(event_create(&m->ev[i][j], true));
    for (int j = 0; j < 2; ++j) OSFM_HIP_CHECK(event_create(&m->ev_sp[j], true));
    for (int j = 0; j < 2; ++j) OSFM_HIP_CHECK(event_create(&m->pin_ev[j], false));
    OSFM_HIP_CHECK(stream_create(&m->copy_stream));
    OSFM_HIP_CHECK(stream_create(&m->up_stream));
    for (int j = 0; j < 2; ++j) {
        OSFM_HIP_CHECK(event_create(&m->ev_compact[j], false));
        OSFM_HIP_CHECK(event_create(&m->ev_copied[j], false));
    }
    // at most 512: the finish kernel's scan of the special rows keys them with nine bits
    if (m->opts.special_kernel_max != 0) m->special_max = std::min(std::max(m->opts.special_kernel_max, 0), 512);
    std::vector<int32_t> t;
    build_lowe_table(m->opts.sift_lowe_ratio, false, &t);
    OSFM_RETURN_IF(m->lowe_sift.reserve(t.size() * 4));
    OSFM_HIP_CHECK(hipMemcpy(m->lowe_sift.ptr, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    build_lowe_table(m->opts.surf_lowe_ratio, true, &t);
    OSFM_RETURN_IF(m->lowe_surf.reserve(t.size() * 4));
    OSFM_HIP_CHECK(hipMemcpy(m->lowe_surf.ptr, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    m->tab_sift.reject_from = m->lowe_sift.as<int32_t>();
    m->tab_sift.max_d1 = max_d1_for(m->opts.sift_distance_threshold, false);
    m->tab_sift.is_signed = 0;
    m->tab_surf.reject_from = m->lowe_surf.as<int32_t>();
    m->tab_surf.max_d1 = max_d1_for(m->opts.surf_distance_threshold, true);
    m->tab_surf.is_signed = 1;
    owner.p = nullptr;
    *out = m;
    return OSFM_OK;
}

int osfm_match_create_multi(const int *device_ids, int num_devices, int num_views,
    const osfm_match_options *opts, osfm_matcher **out)
{
    if (!out || !device_ids || num_devices < 1 || num_views < 0) { set_error("match_create_multi: bad arguments"); return OSFM_E_ARG; }
    *out = nullptr;
    struct Owner { osfm_matcher *p; ~Owner() { if (p) osfm_match_destroy(p); } } owner{(g_ledger.live_matchers++, new osfm_matcher())};
    osfm_matcher *m = owner.p;
    m->device = device_ids[0];
    if (opts) m->opts = *opts; else osfm_match_options_default(&m->opts);
    reset_stats(m);
    for (int k = 0; k < num_devices; ++k) {
        osfm_matcher *sh = nullptr;
        OSFM_RETURN_IF(osfm_match_create(device_ids[k], num_views, &m->opts, &sh));
        m->shards.push_back(sh);
    }
    m->shard_stage.resize(num_devices);
    owner.p = nullptr;
    *out = m;
    return OSFM_OK;
}

int osfm_match_get_devices(const osfm_matcher *m, int32_t *device_ids, int capacity, int32_t *num_devices)
{
    if (!m || !num_devices || capacity < 0 || (capacity > 0 && !device_ids)) { set_error("match_get_devices: bad arguments"); return OSFM_E_ARG; }
    const int n = m->shards.empty() ? 1 : (int)m->shards.size();
    *num_devices = n;
    for (int k = 0; k < std::min(n, capacity); ++k) device_ids[k] = m->shards.empty() ? m->device : m->shards[k]->device;
    return OSFM_OK;
}

int osfm_match_destroy(osfm_matcher *m)
{
    if (!m) return OSFM_OK;
    if (!m->shards.empty() || !m->shard_stage.empty()) {
        for (auto *sh : m->shards) osfm_match_destroy(sh);
        for (auto &sg : m->shard_stage) pinned_free(sg.ptr, sg.ints * 4);
        g_ledger.live_matchers--;
        delete m;
        return OSFM_OK;
    }
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) event_destroy(m->ev[i][j]);
    for (int j = 0; j < 2; ++j) event_destroy(m->ev_sp[j]);
    for (int j = 0; j < 2; ++j) {
        event_destroy(m->pin_ev[j]);
        pinned_free(m->pin_ptr[j], m->pin_bytes[j]);
    }
    stream_destroy(m->copy_stream);
    stream_destroy(m->up_stream);
    for (int j = 0; j < 2; ++j) {
        event_destroy(m->ev_compact[j]);
        event_destroy(m->ev_copied[j]);
    }
    stream_destroy(m->stream);
    for (auto *sg : m->comb_staging) { pinned_free(sg->ptr, sg->ints * 4); delete sg; }
    g_ledger.live_matchers--;
    delete m;       // every DeviceBuffer (views, scratch, cascade-hashing data) frees itself
    return OSFM_OK;
}

// exhaustive_matching.cc:17-27 with math::clamp / math::round in float.
int osfm_quantize_sift(const float *src, int n, uint16_t *dst)
{
    if (n < 0 || (n > 0 && (!src || !dst))) { set_error("quantize_sift: bad arguments"); return OSFM_E_ARG; }
    for (size_t i = 0; i < (size_t)n * 128; ++i) {
        float v = src[i];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        v *= 255.0f;
        v = v > 0.0f ? floorf(v + 0.5f) : ceilf(v - 0.5f);
        dst[i] = (uint16_t)(unsigned char)v;
    }
    return OSFM_OK;
}

// exhaustive_matching.cc:29-38.
int osfm_quantize_surf(const float *src, int n, int16_t *dst)
{
    if (n < 0 || (n > 0 && (!src || !dst))) { set_error("quantize_surf: bad arguments"); return OSFM_E_ARG; }
    for (size_t i = 0; i < (size_t)n * 64; ++i) {
        float v = src[i];
        v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        v *= 127.0f;
        v = v > 0.0f ? floorf(v + 0.5f) : ceilf(v - 0.5f);
        dst[i] = (int16_t)(signed char)v;
    }
    return OSFM_OK;
}

int osfm_match_set_view(osfm_matcher *m, int view, const uint16_t *sift, int n_sift,
    const int16_t *surf, int n_surf)
{
    if (!m) { set_error("set_view: null matcher"); return OSFM_E_ARG; }
    if (!m->shards.empty())        // every shard holds the full bank: one host-to-device copy per device, side by side
        return for_each_shard(m, [&](size_t k) { return osfm_match_set_view(m->shards[k], view, sift, n_sift, surf, n_surf); });
    std::lock_guard<std::mutex> lock(m->up_mu);
    OSFM_HIP_CHECK(hipSetDevice(m->device));
    return upload_view(m, view, sift, n_sift, surf, n_surf);
}

int osfm_match_set_view_float(osfm_matcher *m, int view, const float *sift, int n_sift,
    const float *surf, int n_surf)
{
    if (!m) { set_error("set_view_float: null matcher"); return OSFM_E_ARG; }
    if (n_sift < 0 || n_surf < 0) { set_error("set_view_float: negative count"); return OSFM_E_ARG; }
    std::vector<uint16_t> qs((size_t)n_sift * 128);
    std::vector<int16_t> qu((size_t)n_surf * 64);
    OSFM_RETURN_IF(osfm_quantize_sift(sift, n_sift, qs.data()));
    OSFM_RETURN_IF(osfm_quantize_surf(surf, n_surf, qu.data()));
    return osfm_match_set_view(m, view, qs.data(), n_sift, qu.data(), n_surf);
}

int osfm_match_expect_pairs(osfm_matcher *m, int32_t pairs_per_call)
{
    if (!m || pairs_per_call < 0) { set_error("match_expect_pairs: bad arguments"); return OSFM_E_ARG; }
    if (!m->shards.empty()) {
        for (auto *sh : m->shards) sh->expect_pairs = (pairs_per_call + (int)m->shards.size() - 1) / (int)m->shards.size();
        return OSFM_OK;
    }
    std::lock_guard<std::mutex> lock(m->mu);
    m->expect_pairs = pairs_per_call;
    return OSFM_OK;
}

int osfm_match_view_size(const osfm_matcher *m, int view, int *n_sift, int *n_surf)
{
    if (!m) { set_error("view_size: null matcher"); return OSFM_E_ARG; }
    if (!m->shards.empty()) return osfm_match_view_size(m->shards[0], view, n_sift, n_surf);
    OSFM_RETURN_IF(check_view(m, view, "view_size"));
    if (n_sift) *n_sift = m->views[view].ns;
    if (n_surf) *n_surf = m->views[view].nu;
    return OSFM_OK;
}

int osfm_match_set_positions(osfm_matcher *m, int view, const float *xy, int n)
{
    if (!m || n < 0 || (n > 0 && !xy)) { set_error("set_positions: bad argument"); return OSFM_E_ARG; }
    if (!m->shards.empty())
        return for_each_shard(m, [&](size_t k) { return osfm_match_set_positions(m->shards[k], view, xy, n); });
    std::lock_guard<std::mutex> lock(m->mu);
    OSFM_HIP_CHECK(hipSetDevice(m->device));
    OSFM_RETURN_IF(check_view(m, view, "set_positions"));
    ViewData &v = m->views[view];
    if (n != v.ns + v.nu) {
        set_error("set_positions: view %d has %d features, got %d positions", view, v.ns + v.nu, n);
        return OSFM_E_ARG;
    }
    OSFM_RETURN_IF(v.positions.reserve((size_t)std::max(n, 1) * 8));
    if (n) OSFM_HIP_CHECK(hipMemcpy(v.positions.ptr, xy, (size_t)n * 8, hipMemcpyHostToDevice));
    v.n_positions = n;
    return OSFM_OK;
}

int osfm_ransac_options_default(osfm_ransac_options *o)
{
    if (!o) { set_error("ransac_options_default: null"); return OSFM_E_ARG; }
    o->max_iterations = 1000; o->reserved = 0; o->threshold = 0.0015; o->seed = 0;
    return OSFM_OK;
}

int osfm_ransac_fundamental(int device, const float *pos1, int n1, const float *pos2, int n2,
    const int32_t *corr, int k, const osfm_ransac_options *opts, uint64_t pair_id,
    int32_t *inliers, int32_t *num_inliers, double *F)
{
    if (n1 < 0 || n2 < 0 || k < 0 || !num_inliers || (k > 0 && (!pos1 || !pos2 || !corr || !inliers))) {
        set_error("ransac_fundamental: bad argument"); return OSFM_E_ARG;
    }
    for (int i = 0; i < k; ++i)
        if (corr[2 * i] < 0 || corr[2 * i] >= n1 || corr[2 * i + 1] < 0 || corr[2 * i + 1] >= n2) {
            set_error("ransac_fundamental: correspondence %d out of range", i); return OSFM_E_ARG;
        }
    osfm_ransac_options o;
    if (opts) o = *opts; else osfm_ransac_options_default(&o);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("ransac_fundamental: no HIP device available (this backend has no CPU fallback)");
        return OSFM_E_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("ransac_fundamental: bad device"); return OSFM_E_ARG; }
    OSFM_HIP_CHECK(hipSetDevice(device));
    if (k < 8) { *num_inliers = -1; return OSFM_OK; }
    DeviceBuffer d1, d2, dc, di, dn, df, dj;
    struct Rel { DeviceBuffer *b[7]; ~Rel() { for (auto *x : b) x->release(); } } rel{{&d1, &d2, &dc, &di, &dn, &df, &dj}};
    OSFM_RETURN_IF(d1.reserve((size_t)n1 * 8)); OSFM_RETURN_IF(d2.reserve((size_t)n2 * 8));
    OSFM_RETURN_IF(dc.reserve((size_t)k * 8)); OSFM_RETURN_IF(di.reserve((size_t)k * 4));
    OSFM_RETURN_IF(dn.reserve(16)); OSFM_RETURN_IF(df.reserve(72));
    OSFM_RETURN_IF(dj.reserve(256 + ransac_scratch_bytes(1)));       // the job, then the kernel's scratch
    OSFM_HIP_CHECK(hipMemcpy(d1.ptr, pos1, (size_t)n1 * 8, hipMemcpyHostToDevice));
    OSFM_HIP_CHECK(hipMemcpy(d2.ptr, pos2, (size_t)n2 * 8, hipMemcpyHostToDevice));
    OSFM_HIP_CHECK(hipMemcpy(dc.ptr, corr, (size_t)k * 8, hipMemcpyHostToDevice));
    RansacJob job;
    memset(&job, 0, sizeof(job));
    job.pos1 = d1.as<float>(); job.pos2 = d2.as<float>(); job.corr = dc.as<int32_t>(); job.k = k;
    job.pair_id = pair_id; job.inliers_out = di.as<int32_t>(); job.count_out = dn.as<int32_t>();
    job.F_out = df.as<double>();
    OSFM_HIP_CHECK(hipMemcpy(dj.ptr, &job, sizeof(job), hipMemcpyHostToDevice));
    launch_ransac(dj.as<RansacJob>(), 1, o.max_iterations, o.threshold, o.seed, static_cast<char *>(dj.ptr) + 256, nullptr);
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipDeviceSynchronize());
    int32_t cnt = 0;
    OSFM_HIP_CHECK(hipMemcpy(&cnt, dn.ptr, 4, hipMemcpyDeviceToHost));
    *num_inliers = cnt;
    if (cnt > 0) OSFM_HIP_CHECK(hipMemcpy(inliers, di.ptr, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    if (F) OSFM_HIP_CHECK(hipMemcpy(F, df.ptr, 72, hipMemcpyDeviceToHost));
    return OSFM_OK;
}

namespace {

constexpr size_t kCombineMax = 64;       // requests per combined batch

// One combined batch: requests of one kind (and, for the low-res kind, one feature limit).
void serve_requests(osfm_matcher *m, const std::vector<osfm_matcher::PairRequest *> &batch)
{
    auto fail_all = [&](int status) {
        const std::string msg = osfm_last_error();
        for (auto *r : batch) { r->status = status; r->error = msg; }
    };
    std::lock_guard<std::mutex> lock(m->mu);
    if (hipSetDevice(m->device) != hipSuccess) { set_error("match_pair: hipSetDevice failed"); fail_all(OSFM_E_DEVICE); return; }
    reset_stats(m);
    std::vector<osfm_pair> pairs(batch.size());
    for (size_t k = 0; k < batch.size(); ++k) pairs[k] = {batch[k]->v1, batch[k]->v2};
    BatchResult res;
    BatchMode mode;
    if (batch[0]->kind == 1) { mode.limit = batch[0]->num_features; mode.lowres = true; mode.apply = false; }
    int st = run_batch(m, pairs.data(), (int)pairs.size(), mode, &res);
    if (st != OSFM_OK) { fail_all(st); return; }
    if (batch[0]->kind == 1) {
        for (size_t k = 0; k < batch.size(); ++k) *batch[k]->count = res.counts[k];
        return;
    }
    // all lists of the batch lie back to back in m->out: ONE copy into a page-locked block (two
    // pageable copies per pair were most of a batch's time); every requester then takes its own
    osfm_matcher::Staging *stage = nullptr;
    {
        std::lock_guard<std::mutex> q(m->comb_mu);
        for (auto *sg : m->comb_staging)
            if (sg->users == 0 && (!stage || sg->ints > stage->ints)) stage = sg;
        if (!stage) { stage = new osfm_matcher::Staging; m->comb_staging.push_back(stage); }
        stage->users = (int)batch.size();
    }
    const size_t need = (size_t)std::max<int64_t>(res.out_ints, 4);
    hipError_t e = hipSuccess;
    if (stage->ints < need) {
        pinned_free(stage->ptr, stage->ints * 4);
        stage->ptr = nullptr; stage->ints = 0;
        e = pinned_alloc(reinterpret_cast<void **>(&stage->ptr), (need + need / 4) * 4, hipHostMallocDefault);
        if (e == hipSuccess) stage->ints = need + need / 4;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(stage->ptr, m->out.ptr, need * 4, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
        set_error("match_pair: copy of the lists failed: %s", hipGetErrorString(e));
        fail_all(OSFM_E_DEVICE);
        std::lock_guard<std::mutex> q(m->comb_mu);
        stage->users = 0;
        return;
    }
    for (size_t k = 0; k < batch.size(); ++k) {
        const PairPlan &pl = res.plans[k];
        batch[k]->stage = stage;
        batch[k]->src12 = pl.off12; batch[k]->n12 = pl.len12;
        batch[k]->src21 = pl.off21; batch[k]->n21 = pl.len21;
    }
}

int combine(osfm_matcher *m, osfm_matcher::PairRequest &rq)
{
    std::unique_lock<std::mutex> lk(m->comb_mu);
    m->comb_queue.push_back(&rq);
    while (!rq.done) {
        if (m->comb_leader) { m->comb_cv.wait(lk); continue; }
        m->comb_leader = true;
        while (!rq.done && !m->comb_queue.empty()) {
            // everything of the kind at the head of the queue, in arrival order
            std::vector<osfm_matcher::PairRequest *> batch;
            const osfm_matcher::PairRequest *head = m->comb_queue.front();
            for (auto it = m->comb_queue.begin(); it != m->comb_queue.end() && batch.size() < kCombineMax;) {
                if ((*it)->kind == head->kind && (*it)->num_features == head->num_features) {
                    batch.push_back(*it);
                    it = m->comb_queue.erase(it);
                } else {
                    ++it;
                }
            }
            lk.unlock();
            // The leader must not leave the others waiting for ever, and no exception may cross the C ABI into
            // the caller's OpenMP loop: a failure inside the batch (an allocation, say) fails the batch.
            try {
                serve_requests(m, batch);
            } catch (const std::exception &e) {
                for (auto *r : batch) { r->status = OSFM_E_STATE; r->error = std::string("match_pair: ") + e.what(); r->stage = nullptr; }
            } catch (...) {
                for (auto *r : batch) { r->status = OSFM_E_STATE; r->error = "match_pair: unknown failure in the combined batch"; r->stage = nullptr; }
            }
            lk.lock();
            for (auto *r : batch) r->done = true;
            m->comb_cv.notify_all();
        }
        m->comb_leader = false;
        m->comb_cv.notify_all();             // a waiter whose request is still queued takes over
    }
    lk.unlock();
    if (rq.stage) {
        if (rq.n12) memcpy(rq.m12, rq.stage->ptr + rq.src12, (size_t)rq.n12 * 4);
        if (rq.n21) memcpy(rq.m21, rq.stage->ptr + rq.src21, (size_t)rq.n21 * 4);
        if (rq.len12) *rq.len12 = rq.n12;
        if (rq.len21) *rq.len21 = rq.n21;
        lk.lock();
        rq.stage->users--;
        lk.unlock();
    }
    if (rq.status != OSFM_OK) set_error("%s", rq.error.c_str());
    return rq.status;
}

}  // namespace

int osfm_match_pair(osfm_matcher *m, int view_1, int view_2, int32_t *m12, int32_t *len12,
    int32_t *m21, int32_t *len21)
{
    if (!m || !m12 || !m21) { set_error("match_pair: null argument"); return OSFM_E_ARG; }
    if (!m->shards.empty())        // concurrent callers spread over the devices; each shard combines its own arrivals
        return osfm_match_pair(m->shards[m->next_shard++ % m->shards.size()], view_1, view_2, m12, len12, m21, len21);
    // a bad view id fails its own call, not the batch it would have joined
    OSFM_RETURN_IF(check_view(m, view_1, "match"));
    OSFM_RETURN_IF(check_view(m, view_2, "match"));
    osfm_matcher::PairRequest rq;
    rq.kind = 0; rq.v1 = view_1; rq.v2 = view_2; rq.num_features = 0;
    rq.m12 = m12; rq.m21 = m21; rq.len12 = len12; rq.len21 = len21; rq.count = nullptr;
    return combine(m, rq);
}

int osfm_match_pair_lowres(osfm_matcher *m, int view_1, int view_2, int num_features, int32_t *count)
{
    if (!m || !count) { set_error("match_pair_lowres: null argument"); return OSFM_E_ARG; }
    if (!m->shards.empty())
        return osfm_match_pair_lowres(m->shards[m->next_shard++ % m->shards.size()], view_1, view_2, num_features, count);
    if (num_features <= 0) { *count = 0; return OSFM_OK; }
    OSFM_RETURN_IF(check_view(m, view_1, "match"));
    OSFM_RETURN_IF(check_view(m, view_2, "match"));
    osfm_matcher::PairRequest rq;
    rq.kind = 1; rq.v1 = view_1; rq.v2 = view_2; rq.num_features = num_features;
    rq.m12 = rq.m21 = rq.len12 = rq.len21 = nullptr; rq.count = count;
    return combine(m, rq);
}

int osfm_match_twoway(osfm_matcher *m, int view_1, int view_2, int descriptor_type, int num_features,
    int32_t *m12, int32_t *m21)
{
    if (!m || !m12 || !m21 || (descriptor_type != 0 && descriptor_type != 1) || num_features < 0) {
        set_error("match_twoway: bad argument");
        return OSFM_E_ARG;
    }
    if (!m->shards.empty()) return osfm_match_twoway(m->shards[0], view_1, view_2, descriptor_type, num_features, m12, m21);
    std::lock_guard<std::mutex> lock(m->mu);
    OSFM_HIP_CHECK(hipSetDevice(m->device));
    reset_stats(m);
    osfm_pair pr = {view_1, view_2};
    BatchResult res;
    BatchMode mode;
    mode.limit = num_features; mode.apply = false; mode.type_mask = 1 << descriptor_type;
    OSFM_RETURN_IF(run_batch(m, &pr, 1, mode, &res));
    const PairPlan &pl = res.plans[0];
    // With an empty set 1 Matching::twoway_match yields matches_1_2 of size 0
    // and matches_2_1 of size n2 filled with -1 (matching.h:121-124).
    const ViewData &a = m->views[view_1], &b = m->views[view_2];
    int n1 = descriptor_type == 0 ? a.ns : a.nu, n2 = descriptor_type == 0 ? b.ns : b.nu;
    if (num_features > 0) { n1 = std::min(n1, num_features); n2 = std::min(n2, num_features); }
    if (n1 == 0) { for (int i = 0; i < n2; ++i) m21[i] = -1; return OSFM_OK; }
    const int32_t *d_out = m->out.as<int32_t>();
    if (pl.len12) OSFM_HIP_CHECK(hipMemcpy(m12, d_out + pl.off12, (size_t)pl.len12 * 4, hipMemcpyDeviceToHost));
    if (pl.len21) OSFM_HIP_CHECK(hipMemcpy(m21, d_out + pl.off21, (size_t)pl.len21 * 4, hipMemcpyDeviceToHost));
    return OSFM_OK;
}

int osfm_pair_from_index(int64_t index, int32_t *view_1, int32_t *view_2)
{
    if (index < 0 || !view_1 || !view_2) { set_error("pair_from_index: bad arguments"); return OSFM_E_ARG; }
    // bundler_matching.cc:92-93
    const int a = (int)(0.5 + std::sqrt(0.25 + 2.0 * (double)index));
    *view_1 = a;
    *view_2 = (int)index - a * (a - 1) / 2;
    return OSFM_OK;
}

int osfm_match_all(osfm_matcher *m, const osfm_pair *pairs, int num_pairs, osfm_pair_result *results,
    int32_t *corr, int64_t capacity, int64_t *total)
{
    if (!m || (num_pairs > 0 && (!pairs || !results)) || num_pairs < 0 || capacity < 0 ||
        (capacity > 0 && !corr)) {
        set_error("match_all: bad arguments");
        return OSFM_E_ARG;
    }
    if (!m->shards.empty()) return multi_match_all(m, pairs, num_pairs, results, corr, capacity, total);
    return match_all_single(m, pairs, num_pairs, results, corr, capacity, total, kPhaseBoth);
}

}  // extern "C"

namespace {

// bundler::Matching::compute on one device.  phase: kPhaseBoth -- the whole of it; kPhaseGate -- classification and
// the low-res gate only (a pair that goes on to full matching is left at OSFM_PAIR_MATCHED with its lowres_matches);
// kPhaseFull -- full matching (and RANSAC-F) of the pairs whose result arrives as OSFM_PAIR_MATCHED, everything else
// is left as it is.  The multi-device front runs the two halves as two rounds with a deal of their own each.
int match_all_single(osfm_matcher *m, const osfm_pair *pairs, int num_pairs, osfm_pair_result *results,
    int32_t *corr, int64_t capacity, int64_t *total, int phase)
{
    std::lock_guard<std::mutex> lock(m->mu);
    OSFM_HIP_CHECK(hipSetDevice(m->device));
    if (phase != kPhaseFull) reset_stats(m);
    const osfm_match_options &o = m->opts;
    // whatever way this call ends, no copy into the caller's buffer is still in flight afterwards
    struct CopyDrain { hipStream_t s; ~CopyDrain() { if (s) (void)hipStreamSynchronize(s); } } copy_drain{m->copy_stream};

    // ---- classify (bundler_matching.cc:96-99, 146-158) ---------------------
    std::vector<int> lowres_idx, full_idx;
    for (int p = 0; p < num_pairs; ++p) {
        OSFM_RETURN_IF(check_view(m, pairs[p].view_1, "match_all"));
        OSFM_RETURN_IF(check_view(m, pairs[p].view_2, "match_all"));
        const ViewData &a = m->views[pairs[p].view_1], &b = m->views[pairs[p].view_2];
        osfm_pair_result &r = results[p];
        if (phase == kPhaseFull) {
            if (r.status == OSFM_PAIR_MATCHED) full_idx.push_back(p);
            continue;
        }
        r.status = OSFM_PAIR_MATCHED; r.lowres_matches = -1; r.num_matches = 0; r.num_inliers = -1; r.offset = 0;
        const size_t np1 = (size_t)a.ns + a.nu, np2 = (size_t)b.ns + b.nu;
        if (np1 == 0 || np2 == 0) { r.status = OSFM_PAIR_SKIPPED_EMPTY; continue; }
        if (o.use_lowres_matching && np1 * np2 > 1000000) lowres_idx.push_back(p);
        else full_idx.push_back(p);
    }

    // per-pair workspace estimate -> batch size
    // pairs per launch group: bounded by the partial-result workspace
    // (column partials dominate: nrb * n2 * 8 B per pair) -- 24 GB of the 288 GB
    auto batch_size_for = [&](bool lowres) {
        if (o.pairs_per_batch > 0) return o.pairs_per_batch;
        if (lowres) return 16384;
        size_t worst = 1;
        for (const auto &v : m->views) {
            const size_t n = (size_t)std::max(v.ns, v.nu) + 256;
            worst = std::max(worst, (n / kRowsPerBlock + 1) * n * sizeof(ColPart) + n * 64);
        }
        const size_t budget = (size_t)24 << 30;
        return (int)std::max<size_t>(1, std::min<size_t>(budget / worst, 4096));
    };
    // A call that fits one batch is still cut in three when it is large: the lists of one part then leave
    // the device (63 MB per 1225 pairs: 2.5 ms of an otherwise idle device) while the next part is matched.
    auto full_batch_size = [&](size_t n_full) {
        int bs = batch_size_for(false);
        // (not in cascade-hashing mode: its bucket kernels want the large launch)
        if (o.pairs_per_batch <= 0 && !o.geometric_verification && o.matcher_type != OSFM_MATCHER_CASCADE_HASHING && n_full >= 768)
            bs = std::min<int>(bs, (int)((n_full + 2) / 3));
        // parts of equal size: a call of 3898 pairs at 1875 per part ran 1875 + 1875 + 148, and the short part took
        // 17.5 ms where its pairs are worth 8 (a launch's fixed costs and tail): 0.3 s of a 500-view job
        if (o.pairs_per_batch <= 0 && n_full > (size_t)bs) {
            const size_t parts = (n_full + bs - 1) / bs;
            bs = (int)((n_full + parts - 1) / parts);
        }
        return bs;
    };

    // ---- low-res gate -------------------------------------------------------
    {
        const int bs = batch_size_for(true);
        std::vector<osfm_pair> chunk;
        for (size_t start = 0; start < lowres_idx.size(); start += bs) {
            const size_t end = std::min(lowres_idx.size(), start + bs);
            chunk.clear();
            for (size_t k = start; k < end; ++k) chunk.push_back(pairs[lowres_idx[k]]);
            BatchResult res;
            BatchMode mode;
            mode.limit = o.num_lowres_features; mode.lowres = true; mode.apply = false;
            OSFM_RETURN_IF(run_batch(m, chunk.data(), (int)chunk.size(), mode, &res));
            for (size_t k = start; k < end; ++k) {
                const int p = lowres_idx[k];
                results[p].lowres_matches = res.counts[k - start];
                if (res.counts[k - start] < o.min_lowres_matches)
                    results[p].status = OSFM_PAIR_REJECTED_LOWRES;
                else
                    full_idx.push_back(p);
            }
        }
        std::sort(full_idx.begin(), full_idx.end());
    }
    if (phase == kPhaseGate) { if (total) *total = 0; return OSFM_OK; }

    // ---- full matching + ordered correspondence lists -----------------------
    const int min_matches = std::max(8, o.min_feature_matches);
    int64_t written = 0, written_out = 0;     // pre-RANSAC / post-RANSAC correspondence counts
    bool overflow = false;
    {
        const int bs = full_batch_size(full_idx.size());
        int chunk_no = 0;
        std::vector<osfm_pair> chunk;
        std::vector<int64_t> h_m12_off, h_corr_off;
        std::vector<int32_t> h_len12;
        std::vector<uint8_t> h_keep;
        for (size_t start = 0; start < full_idx.size(); start += bs) {
            const size_t end = std::min(full_idx.size(), start + bs);
            const int n = (int)(end - start);
            chunk.clear();
            for (size_t k = start; k < end; ++k) chunk.push_back(pairs[full_idx[k]]);
            static const bool trace = getenv("OSFM_MATCH_TRACE") != nullptr;
            const auto t_chunk = std::chrono::steady_clock::now();
            auto lap = [&](const char *what) {
                if (trace) fprintf(stderr, "[osfm match] %-18s %9.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_chunk).count());
            };
            BatchResult res;
            BatchMode full_mode;
            // (the largest part of the largest call announced: what the work arrays are sized for from the first call on)
            if (m->expect_pairs > 0) full_mode.expect = std::min<int>(full_batch_size((size_t)m->expect_pairs), m->expect_pairs);
            OSFM_RETURN_IF(run_batch(m, chunk.data(), n, full_mode, &res));
            lap("matched");
            h_m12_off.assign(n, 0); h_corr_off.assign(n, 0); h_len12.assign(n, 0); h_keep.assign(n, 0);
            int64_t chunk_corr = 0;
            for (int k = 0; k < n; ++k) {
                const int p = full_idx[start + k];
                osfm_pair_result &r = results[p];
                r.num_matches = res.counts[k];
                if (res.counts[k] < min_matches) { r.status = OSFM_PAIR_REJECTED_COUNT; continue; }
                r.status = OSFM_PAIR_MATCHED;
                r.offset = written + chunk_corr;
                h_keep[k] = 1;
                h_m12_off[k] = res.plans[k].off12;
                h_len12[k] = res.plans[k].len12;
                h_corr_off[k] = chunk_corr;
                chunk_corr += res.counts[k];
            }
            if (!o.geometric_verification && written + chunk_corr > capacity) overflow = true;
            // With verification the pre-RANSAC lists are device scratch (d_corr) that the
            // RANSAC jobs of THIS chunk read: they are built whatever the state of the
            // caller's buffer; only the copies to the host stop after an overflow.
            if (chunk_corr > 0 && (o.geometric_verification || !overflow)) {
                hipStream_t s = m->stream;
                OSFM_RETURN_IF(m->d_m12_off.reserve(n * 8));
                OSFM_RETURN_IF(m->d_corr_off.reserve(n * 8));
                OSFM_RETURN_IF(m->d_len12.reserve(n * 4));
                OSFM_RETURN_IF(m->d_keep_pair.reserve(n));
                // without verification two list buffers alternate: the copy of the one is in flight on the
                // copy stream while the next chunk is matched and compacted into the other
                const int cb = o.geometric_verification ? 0 : (chunk_no++ & 1);
                DeviceBuffer &dcorr = cb ? m->d_corr_alt : m->d_corr;
                if (m->copied_used[cb]) OSFM_HIP_CHECK(hipEventSynchronize(m->ev_copied[cb]));      // before a reserve may free it
                OSFM_RETURN_IF(dcorr.reserve((size_t)chunk_corr * 8));
                OSFM_HIP_CHECK(hipMemcpyAsync(m->d_m12_off.ptr, h_m12_off.data(), n * 8, hipMemcpyHostToDevice, s));
                OSFM_HIP_CHECK(hipMemcpyAsync(m->d_corr_off.ptr, h_corr_off.data(), n * 8, hipMemcpyHostToDevice, s));
                OSFM_HIP_CHECK(hipMemcpyAsync(m->d_len12.ptr, h_len12.data(), n * 4, hipMemcpyHostToDevice, s));
                OSFM_HIP_CHECK(hipMemcpyAsync(m->d_keep_pair.ptr, h_keep.data(), n, hipMemcpyHostToDevice, s));
                launch_compact_pairs(n, m->out.as<int32_t>(), m->d_m12_off.as<int64_t>(),
                    m->d_len12.as<int32_t>(), m->d_corr_off.as<int64_t>(), m->d_keep_pair.as<uint8_t>(),
                    dcorr.as<int32_t>(), s);
                OSFM_HIP_CHECK(hipGetLastError());
                if (!o.geometric_verification) {
                    OSFM_HIP_CHECK(hipEventRecord(m->ev_compact[cb], s));
                    OSFM_HIP_CHECK(hipStreamWaitEvent(m->copy_stream, m->ev_compact[cb], 0));
                    OSFM_HIP_CHECK(hipMemcpyAsync(corr + 2 * written, dcorr.ptr, (size_t)chunk_corr * 8,
                        hipMemcpyDeviceToHost, m->copy_stream));
                    OSFM_HIP_CHECK(hipEventRecord(m->ev_copied[cb], m->copy_stream));
                    m->copied_used[cb] = true;
                }
            }
            if (o.geometric_verification && chunk_corr > 0) {
                // ---- RANSAC-F on the device-resident lists (bundler_matching.cc:194-219) ----
                hipStream_t s = m->stream;
                std::vector<RansacJob> jobs;
                std::vector<int> job_pair;
                OSFM_RETURN_IF(m->d_inl.reserve((size_t)chunk_corr * 4));
                OSFM_RETURN_IF(m->d_inl_count.reserve((size_t)n * 4));
                for (int k = 0; k < n; ++k) {
                    if (!h_keep[k]) continue;
                    const int p = full_idx[start + k];
                    const ViewData &a = m->views[pairs[p].view_1], &b = m->views[pairs[p].view_2];
                    if (a.n_positions < 0 || b.n_positions < 0) {
                        set_error("match_all: geometric verification needs osfm_match_set_positions for views %d and %d",
                            pairs[p].view_1, pairs[p].view_2);
                        return OSFM_E_STATE;
                    }
                    RansacJob j;
                    memset(&j, 0, sizeof(j));
                    j.pos1 = a.positions.as<float>(); j.pos2 = b.positions.as<float>();
                    j.corr = m->d_corr.as<int32_t>() + 2 * h_corr_off[k];
                    j.k = res.counts[k];
                    // stream id = linear index of the pair in the reference's enumeration
                    const int64_t hi = std::max(pairs[p].view_1, pairs[p].view_2), lo = std::min(pairs[p].view_1, pairs[p].view_2);
                    j.pair_id = (uint64_t)(hi * (hi - 1) / 2 + lo);
                    j.inliers_out = m->d_inl.as<int32_t>() + h_corr_off[k];
                    j.count_out = m->d_inl_count.as<int32_t>() + (int)jobs.size();
                    j.F_out = nullptr;
                    jobs.push_back(j); job_pair.push_back(k);
                }
                const size_t jobs_bytes = (jobs.size() * sizeof(RansacJob) + 255) / 256 * 256;
                OSFM_RETURN_IF(m->d_jobs.reserve(jobs_bytes + ransac_scratch_bytes((int)jobs.size())));
                OSFM_HIP_CHECK(hipMemcpyAsync(m->d_jobs.ptr, jobs.data(), jobs.size() * sizeof(RansacJob), hipMemcpyHostToDevice, s));
                launch_ransac(m->d_jobs.as<RansacJob>(), (int)jobs.size(), o.ransac_max_iterations, o.ransac_threshold,
                    o.ransac_seed, static_cast<char *>(m->d_jobs.ptr) + jobs_bytes, s);
                OSFM_HIP_CHECK(hipGetLastError());
                std::vector<int32_t> h_cnt(jobs.size());
                OSFM_HIP_CHECK(hipMemcpyAsync(h_cnt.data(), m->d_inl_count.ptr, jobs.size() * 4, hipMemcpyDeviceToHost, s));
                lap("ransac queued");
                OSFM_HIP_CHECK(hipStreamSynchronize(s));
                lap("ransac done");
                // keep pairs with enough inliers; gather their inlier correspondences
                const int min_inl = std::max(8, o.min_matching_inliers);
                std::vector<int64_t> g_src, g_dst;     // per job: source offset (corr / inlier ids), destination
                std::vector<int32_t> g_cnt;
                int64_t chunk_out = 0;
                for (size_t jx = 0; jx < jobs.size(); ++jx) {
                    const int k = job_pair[jx];
                    osfm_pair_result &r = results[full_idx[start + k]];
                    r.num_inliers = h_cnt[jx];
                    if (h_cnt[jx] < min_inl) { r.status = OSFM_PAIR_REJECTED_INLIERS; r.offset = 0; continue; }
                    r.offset = written_out + chunk_out;
                    g_src.push_back(h_corr_off[k]); g_dst.push_back(chunk_out); g_cnt.push_back(h_cnt[jx]);
                    chunk_out += h_cnt[jx];
                }
                if (written_out + chunk_out > capacity) overflow = true;
                if (!overflow && chunk_out > 0) {
                    const int ng = (int)g_cnt.size();
                    OSFM_RETURN_IF(m->d_gather_off.reserve((size_t)ng * 20));
                    OSFM_RETURN_IF(m->d_corr2.reserve((size_t)chunk_out * 8));
                    char *gb = m->d_gather_off.as<char>();
                    OSFM_HIP_CHECK(hipMemcpyAsync(gb, g_src.data(), (size_t)ng * 8, hipMemcpyHostToDevice, s));
                    OSFM_HIP_CHECK(hipMemcpyAsync(gb + (size_t)ng * 8, g_dst.data(), (size_t)ng * 8, hipMemcpyHostToDevice, s));
                    OSFM_HIP_CHECK(hipMemcpyAsync(gb + (size_t)ng * 16, g_cnt.data(), (size_t)ng * 4, hipMemcpyHostToDevice, s));
                    launch_gather_inliers(ng, m->d_corr.as<int32_t>(), m->d_inl.as<int32_t>(),
                        reinterpret_cast<const int64_t *>(gb), reinterpret_cast<const int64_t *>(gb + (size_t)ng * 8),
                        reinterpret_cast<const int32_t *>(gb + (size_t)ng * 16), m->d_corr2.as<int32_t>(), s);
                    OSFM_HIP_CHECK(hipGetLastError());
                    OSFM_HIP_CHECK(hipMemcpyAsync(corr + 2 * written_out, m->d_corr2.ptr, (size_t)chunk_out * 8,
                        hipMemcpyDeviceToHost, s));
                    OSFM_HIP_CHECK(hipStreamSynchronize(s));
                    lap("lists on the host");
                }
                written_out += chunk_out;
            }
            written += chunk_corr;
        }
    }
    if (!o.geometric_verification) OSFM_HIP_CHECK(hipStreamSynchronize(m->copy_stream));      // every list has arrived
    if (o.geometric_verification) written = written_out;
    if (total) *total = written;
    if (overflow) {
        set_error("match_all: %lld correspondences need more than the given capacity %lld",
            (long long)written, (long long)capacity);
        return OSFM_E_CAPACITY;
    }
    return OSFM_OK;
}

}  // namespace

extern "C" {

int osfm_match_get_cascade_hashes(osfm_matcher *m, int view, int type, uint64_t *hashes,
    uint8_t *bucket_ids)
{
    if (!m || type < 0 || type > 1) { set_error("get_cascade_hashes: bad arguments"); return OSFM_E_ARG; }
    if (!m->shards.empty()) return osfm_match_get_cascade_hashes(m->shards[0], view, type, hashes, bucket_ids);
    std::lock_guard<std::mutex> lock(m->mu);
    OSFM_RETURN_IF(check_view(m, view, "get_cascade_hashes"));
    OSFM_HIP_CHECK(hipSetDevice(m->device));
    OSFM_RETURN_IF(ensure_cashash(m));
    const ViewData &v = m->views[view];
    const int n = type == 0 ? v.ns : v.nu;
    const int words = type == 0 ? 2 : 1;
    if (n > 0 && hashes)
        OSFM_HIP_CHECK(hipMemcpy(hashes, v.cas_hash[type].ptr, (size_t)n * words * 8, hipMemcpyDeviceToHost));
    if (n > 0 && bucket_ids)
        OSFM_HIP_CHECK(hipMemcpy(bucket_ids, v.cas_bucket[type].ptr, (size_t)n * kCasGroups, hipMemcpyDeviceToHost));
    return OSFM_OK;
}

int osfm_match_get_shard_stats(const osfm_matcher *m, int shard, osfm_match_stats *out)
{
    if (!m || !out) { set_error("get_shard_stats: null argument"); return OSFM_E_ARG; }
    const int n = m->shards.empty() ? 1 : (int)m->shards.size();
    if (shard < 0 || shard >= n) { set_error("get_shard_stats: shard %d of %d", shard, n); return OSFM_E_ARG; }
    *out = m->shards.empty() ? m->stats : m->shards[shard]->stats;
    return OSFM_OK;
}

int osfm_match_get_stats(const osfm_matcher *m, osfm_match_stats *out)
{
    if (!m || !out) { set_error("get_stats: null argument"); return OSFM_E_ARG; }
    if (!m->shards.empty()) {
        // sums over the shards' most recent calls (times are device times: they ran side by side)
        memset(out, 0, sizeof(*out));
        for (const auto *sh : m->shards) {
            const osfm_match_stats &t = sh->stats;
            out->tile_kernel_ms += t.tile_kernel_ms; out->tile_kernel_launches += t.tile_kernel_launches;
            out->exact_scan_queries += t.exact_scan_queries; out->mac_count += t.mac_count;
            out->algorithmic_bytes += t.algorithmic_bytes; out->lowres_kernel_ms += t.lowres_kernel_ms;
            out->lowres_kernel_launches += t.lowres_kernel_launches; out->lowres_mac_count += t.lowres_mac_count;
            out->cashash_kernel_ms += t.cashash_kernel_ms; out->cashash_kernel_launches += t.cashash_kernel_launches;
            out->special_kernel_launches += t.special_kernel_launches; out->special_kernel_ms += t.special_kernel_ms;
            out->tile_shader_cycles += t.tile_shader_cycles; out->tile_refclk_ticks += t.tile_refclk_ticks;
            out->surf_tile_kernel_ms += t.surf_tile_kernel_ms; out->surf_tile_kernel_launches += t.surf_tile_kernel_launches;
            out->surf_mac_count += t.surf_mac_count;
        }
        return OSFM_OK;
    }
    *out = m->stats;
    return OSFM_OK;
}

}  // extern "C"
