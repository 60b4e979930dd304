// Track building: sfm::bundler::Tracks::compute
// (src/mve/sfm/bundler_tracks.cc:49-145) on flat arrays.  Host code, as in the
// reference -- the merge is a sequential, order-defining union over the match
// lists (the feature order inside a track and the track order are part of the
// result), so it is restated with cheaper data structures rather than moved to
// the device: every feature is a node of a singly linked list per track
// (head / tail / size), unify_tracks (:23-45) splices the smaller list behind
// the larger one in O(1) after relabelling its nodes, and nothing is
// reallocated.  The output order is the reference's, element for element.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "osfm_common.h"

using namespace osfm;

// The merge as a state that takes the match lists pair batch by pair batch (in the reference's
// pair order): the host can then build tracks while the device matches the next batch.
struct osfm_tracks_builder {
    int32_t num_views = 0;
    std::vector<int32_t> view_sizes;
    std::vector<int64_t> voff;
    // node g = voff[view] + feature; nxt[g] = next feature of the same track
    std::vector<int32_t> tid;
    std::vector<int64_t> nxt;
    std::vector<int64_t> head, tail;
    std::vector<int32_t> size;
    // a self-match (view_1 == view_2, feature matched to itself) makes the reference push
    // the same FeatureReference twice (:80-86): the track is invalid for good (two features
    // of one view) and counts one feature more than it has nodes
    std::vector<uint8_t> twice;
};

static int builder_init(osfm_tracks_builder *b, int32_t num_views, const int32_t *view_sizes)
{
    if (num_views < 0 || (num_views > 0 && !view_sizes)) { set_error("tracks: null array / negative count"); return OSFM_E_ARG; }
    b->num_views = num_views;
    b->view_sizes.assign(view_sizes, view_sizes + num_views);
    b->voff.assign((size_t)num_views + 1, 0);
    for (int v = 0; v < num_views; ++v) {
        if (view_sizes[v] < 0) { set_error("tracks_compute: negative view size"); return OSFM_E_ARG; }
        b->voff[v + 1] = b->voff[v] + view_sizes[v];
    }
    const int64_t G = b->voff[num_views];
    b->tid.assign((size_t)G, -1);
    b->nxt.assign((size_t)G, -1);
    return OSFM_OK;
}

// pair p owns corr[pair_begin[p] .. pair_end[p])
static int builder_feed(osfm_tracks_builder *b, int32_t num_pairs, const osfm_pair *pairs, const int64_t *pair_begin,
    const int64_t *pair_end, const int32_t *corr)
{
    if (num_pairs < 0 || (num_pairs > 0 && (!pairs || !pair_begin || !pair_end))) {
        set_error("tracks_compute: null array / negative count");
        return OSFM_E_ARG;
    }
    int64_t total_matches = 0;
    for (int p = 0; p < num_pairs; ++p) {
        if (pair_begin[p] < 0 || pair_end[p] < pair_begin[p]) {
            set_error("tracks_compute: pair %d has the match range [%lld, %lld)", p,
                (long long)pair_begin[p], (long long)pair_end[p]);
            return OSFM_E_ARG;
        }
        total_matches += pair_end[p] - pair_begin[p];
    }
    if (total_matches > 0 && !corr) { set_error("tracks_compute: corr is null"); return OSFM_E_ARG; }
    const int num_views = b->num_views;
    const int32_t *view_sizes = b->view_sizes.data();
    auto &voff = b->voff; auto &tid = b->tid; auto &nxt = b->nxt; auto &head = b->head; auto &tail = b->tail;
    auto &size = b->size; auto &twice = b->twice;
    for (int p = 0; p < num_pairs; ++p) {                                       // :66-119
        const int v1 = pairs[p].view_1, v2 = pairs[p].view_2;
        if (v1 < 0 || v1 >= num_views || v2 < 0 || v2 >= num_views) {
            set_error("tracks_compute: pair %d names view %d / %d of %d", p, v1, v2, num_views);
            return OSFM_E_ARG;
        }
        for (int64_t k = pair_begin[p]; k < pair_end[p]; ++k) {
            const int f1 = corr[2 * k], f2 = corr[2 * k + 1];
            if (f1 < 0 || f1 >= view_sizes[v1] || f2 < 0 || f2 >= view_sizes[v2]) {
                set_error("tracks_compute: match %lld of pair %d out of range", (long long)k, p);
                return OSFM_E_RANGE;
            }
            const int64_t g1 = voff[v1] + f1, g2 = voff[v2] + f2;
            const int32_t t1 = tid[g1], t2 = tid[g2];
            if (t1 == -1 && t2 == -1) {
                const int32_t t = (int32_t)head.size();
                head.push_back(g1); tail.push_back(g2); size.push_back(2);
                twice.push_back(g1 == g2 ? 1 : 0);
                if (g1 != g2) nxt[g1] = g2;
                tid[g1] = t; tid[g2] = t;
            } else if (t1 == -1) {
                tid[g1] = t2; nxt[tail[t2]] = g1; tail[t2] = g1; size[t2]++;
            } else if (t2 == -1) {
                tid[g2] = t1; nxt[tail[t1]] = g2; tail[t1] = g2; size[t1]++;
            } else if (t1 != t2) {
                // unify into the larger track, the first one on a draw (:28-31)
                int32_t a = t1, bq = t2;
                if (size[a] < size[bq]) { a = t2; bq = t1; }
                for (int64_t g = head[bq]; g >= 0; g = nxt[g]) tid[g] = a;
                nxt[tail[a]] = head[bq]; tail[a] = tail[bq]; size[a] += size[bq];
                twice[a] |= twice[bq];
                head[bq] = -1; tail[bq] = -1; size[bq] = 0;
            }
        }
    }
    return OSFM_OK;
}

static int builder_finish(const osfm_tracks_builder *b, const uint8_t *colors,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary)
{
    if (!track_offsets || (track_capacity > 0 && !track_colors) || (feature_capacity > 0 && !track_features)) {
        set_error("tracks_compute: null array / negative count");
        return OSFM_E_ARG;
    }
    const int num_views = b->num_views;
    const auto &voff = b->voff; const auto &tid = b->tid; const auto &nxt = b->nxt; const auto &head = b->head;
    const auto &size = b->size; const auto &twice = b->twice;
    const int64_t G = voff[num_views];

    // remove_invalid_tracks (:149-203: empty tracks, tracks with two features of one view) and the
    // output in ONE walk over each track's list: the walk is a chain of cache misses (the nodes of a
    // track lie all over the feature array), and two walks -- validate, then write -- were most of
    // this function's time.  A track is written where it would go and discarded (the write position
    // rolls back) when a second feature of one view turns up.
    //
    // The walks of different tracks are independent, and each is bound by memory latency, so the
    // tracks are cut into ranges of equal node counts, one per host thread: a thread walks its range
    // into buffers of its own, and the ranges are then laid end to end (the order of the reference:
    // tracks by id).
    const int64_t nt = (int64_t)head.size();
    std::vector<int32_t> map((size_t)nt, -1);
    int64_t total_nodes = 0;
    for (int64_t t = 0; t < nt; ++t) total_nodes += size[t];
    int workers = 1;
    if (const char *e = getenv("OSFM_TRACKS_THREADS")) workers = std::max(1, std::min(64, atoi(e)));
    else if (total_nodes >= (int64_t)1 << 16)
        workers = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    struct Range {
        int64_t t0 = 0, t1 = 0;
        std::vector<int32_t> feats;          // (view, feature) of the kept tracks, in order
        std::vector<int64_t> starts;         // per kept track: first feature in feats
        std::vector<uint8_t> cols;
        std::vector<int64_t> kept;           // ids of the kept tracks
        int32_t invalid = 0;
    };
    std::vector<Range> ranges((size_t)workers);
    {
        int64_t t = 0, acc = 0;
        for (int w = 0; w < workers; ++w) {
            ranges[w].t0 = t;
            const int64_t want = total_nodes * (w + 1) / workers;
            while (t < nt && (acc < want || w == workers - 1)) acc += size[t++];
            ranges[w].t1 = t;
        }
    }
    auto walk = [&](Range &R) {
        std::vector<int64_t> seen((size_t)num_views, -1);
        int64_t cap = 0;
        for (int64_t t = R.t0; t < R.t1; ++t) cap += size[t];
        R.feats.resize((size_t)2 * cap);
        int64_t nf = 0;
        for (int64_t t = R.t0; t < R.t1; ++t) {
            if (size[t] == 0) continue;
            bool bad = twice[t] != 0;
            const int64_t start = nf;
            float col[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                // :133-143
            for (int64_t g = head[t]; g >= 0 && !bad; g = nxt[g]) {
                // view of node g: the last voff entry <= g (views are few: binary search)
                int lo = 0, hi = num_views - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (voff[mid] <= g) lo = mid; else hi = mid - 1; }
                if (seen[lo] == t) { bad = true; break; }
                seen[lo] = t;
                R.feats[2 * nf] = lo;
                R.feats[2 * nf + 1] = (int32_t)(g - voff[lo]);
                ++nf;
                for (int c = 0; c < 3; ++c) col[c] += colors ? (float)colors[3 * g + c] : 0.0f;
                col[3] += 1.0f;
            }
            if (bad) { R.invalid++; nf = start; continue; }
            R.starts.push_back(start);
            for (int c = 0; c < 3; ++c) R.cols.push_back((uint8_t)(col[c] / col[3] + 0.5f));
            R.kept.push_back(t);
        }
        R.feats.resize((size_t)2 * nf);
    };
    // (an allocation failure inside a thread must come back as a status, not end the process)
    std::atomic<bool> failed{false};
    auto guarded = [&](auto &&fn) { try { fn(); } catch (...) { failed = true; } };
    if (workers == 1) guarded([&] { walk(ranges[0]); });
    else {
        std::vector<std::thread> th;
        for (int w = 1; w < workers; ++w) th.emplace_back([&, w] { guarded([&] { walk(ranges[w]); }); });
        guarded([&] { walk(ranges[0]); });
        for (auto &x : th) x.join();
    }
    if (failed) { set_error("tracks_compute: out of host memory while writing the tracks"); return OSFM_E_STATE; }
    int32_t invalid = 0, valid = 0;
    int64_t nf = 0;
    std::vector<int64_t> base_f((size_t)workers), base_t((size_t)workers);
    for (int w = 0; w < workers; ++w) {
        base_f[w] = nf; base_t[w] = valid;
        nf += (int64_t)ranges[w].feats.size() / 2;
        valid += (int32_t)ranges[w].kept.size();
        invalid += ranges[w].invalid;
    }
    const bool overflow = valid > track_capacity || nf > feature_capacity;
    if (!overflow) {
        auto place = [&](int w) {
            const Range &R = ranges[w];
            if (!R.feats.empty()) memcpy(track_features + 2 * base_f[w], R.feats.data(), R.feats.size() * sizeof(int32_t));
            for (size_t k = 0; k < R.kept.size(); ++k) {
                track_offsets[base_t[w] + (int64_t)k] = base_f[w] + R.starts[k];
                map[(size_t)R.kept[k]] = (int32_t)(base_t[w] + (int64_t)k);
            }
            if (!R.cols.empty()) memcpy(track_colors + 3 * base_t[w], R.cols.data(), R.cols.size());
        };
        if (workers == 1) place(0);
        else {
            std::vector<std::thread> th;
            for (int w = 1; w < workers; ++w) th.emplace_back([&, w] { place(w); });
            place(0);
            for (auto &x : th) x.join();
        }
    }
    const int64_t kept_features = nf;
    if (track_ids)          // Viewport::track_ids (the builder's finish may leave them out)
        for (int64_t g = 0; g < G; ++g) track_ids[g] = tid[g] >= 0 ? map[tid[g]] : -1;
    if (summary) {
        summary->num_tracks = valid;
        summary->num_invalid_tracks = invalid;
        summary->num_features = kept_features;
    }
    if (overflow || valid > track_capacity || kept_features > feature_capacity) {
        set_error("tracks_compute: %d tracks / %lld features exceed the output capacity (%lld / %lld)",
            valid, (long long)kept_features, (long long)track_capacity, (long long)feature_capacity);
        return OSFM_E_CAPACITY;
    }
    track_offsets[valid] = nf;
    return OSFM_OK;
}

extern "C" {

static int tracks_compute_impl(int32_t num_views, const int32_t *view_sizes, const uint8_t *colors,
    int32_t num_pairs, const osfm_pair *pairs, const int64_t *pair_begin, const int64_t *pair_end,
    const int32_t *corr,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary)
{
    if (num_views < 0 || num_pairs < 0 || (num_views > 0 && !view_sizes) ||
        (num_pairs > 0 && (!pairs || !pair_begin || !pair_end)) || !track_offsets ||
        (track_capacity > 0 && !track_colors) || (feature_capacity > 0 && !track_features)) {
        set_error("tracks_compute: null array / negative count");
        return OSFM_E_ARG;
    }
    osfm_tracks_builder b;
    OSFM_RETURN_IF(builder_init(&b, num_views, view_sizes));
    if (b.voff[num_views] > 0 && !track_ids) { set_error("tracks_compute: track_ids is null"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(builder_feed(&b, num_pairs, pairs, pair_begin, pair_end, corr));
    return builder_finish(&b, colors, track_ids, track_capacity, feature_capacity, track_offsets, track_features,
        track_colors, summary);
}

int osfm_tracks_builder_create(int32_t num_views, const int32_t *view_sizes, osfm_tracks_builder **out)
{
    if (!out) { set_error("tracks_builder_create: null argument"); return OSFM_E_ARG; }
    *out = nullptr;
    osfm_tracks_builder *b = new osfm_tracks_builder();
    const int st = builder_init(b, num_views, view_sizes);
    if (st != OSFM_OK) { delete b; return st; }
    *out = b;
    return OSFM_OK;
}

int osfm_tracks_builder_feed(osfm_tracks_builder *b, int32_t num_pairs, const osfm_pair *pairs,
    const int64_t *pair_starts, const int64_t *pair_counts, const int32_t *corr)
{
    if (!b || num_pairs < 0 || (num_pairs > 0 && (!pair_starts || !pair_counts))) { set_error("tracks_builder_feed: bad arguments"); return OSFM_E_ARG; }
    std::vector<int64_t> ends((size_t)num_pairs);
    for (int p = 0; p < num_pairs; ++p) ends[p] = pair_starts[p] + pair_counts[p];
    return builder_feed(b, num_pairs, pairs, pair_starts, ends.data(), corr);
}

int osfm_tracks_builder_finish(const osfm_tracks_builder *b, const uint8_t *colors,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary)
{
    if (!b) { set_error("tracks_builder_finish: null builder"); return OSFM_E_ARG; }
    return builder_finish(b, colors, track_ids, track_capacity, feature_capacity, track_offsets, track_features,
        track_colors, summary);
}

int osfm_tracks_builder_destroy(osfm_tracks_builder *b)
{
    delete b;
    return OSFM_OK;
}

// The observation arrays of an osfm_ba_problem from the tracks of a scene in one pass: what the
// reference builds residual block by residual block from its std::vector<Track>
// (bundle_adjustment.cpp:86-123: every feature of a selected track whose view has a camera).
// Feature i (features in track order) is taken when live[i] != 0, camera_of_feature[i] >= 0 and
// (track_mask == NULL or track_mask[track_of[i]] != 0).  obs_point[k] = track_slot[track] when
// track_slot is given (the caller's numbering of the parameter blocks), else the rank of the track
// among the tracks that contributed a feature, whose ids then go to tracks_out.
int osfm_tracks_select_observations(int64_t num_features, const int32_t *track_of, const int64_t *track_offsets,
    const int32_t *camera_of_feature,
    const uint8_t *live, const uint8_t *track_mask, const int32_t *track_slot, const double *xy,
    int64_t capacity, int32_t *feature_ids, double *obs_xy, int32_t *obs_camera, int32_t *obs_point,
    int32_t *tracks_out, int64_t *num_observations, int64_t *num_tracks_out)
{
    if (num_features < 0 || capacity < 0 || !num_observations ||
        (num_features > 0 && (!track_of || !camera_of_feature || !live || !xy)) ||
        (capacity > 0 && (!obs_xy || !obs_camera || !obs_point))) {
        set_error("tracks_select_observations: bad arguments");
        return OSFM_E_ARG;
    }
    int64_t n = 0, nt = 0;
    int32_t last = -1;
    if (track_offsets && capacity >= num_features && num_features > 0) {
        // Track by track, without a branch per feature: whether a feature's view has a camera is as good
        // as random along a track while half the views are aligned, and a mispredicted branch per feature
        // was most of this pass.  Every feature is written to the next free slot and the slot is kept or
        // not (n <= i < capacity, so the write is always inside the buffers).
        const int32_t num_tracks = track_of[num_features - 1] + 1;
        for (int32_t t = 0; t < num_tracks; ++t) {
            if (track_mask && !track_mask[t]) continue;
            const int64_t n0 = n;
            const int32_t pt = track_slot ? track_slot[t] : (int32_t)nt;
            for (int64_t i = track_offsets[t]; i < track_offsets[t + 1]; ++i) {
                const int32_t c = camera_of_feature[i];
                if (feature_ids) feature_ids[n] = (int32_t)i;
                obs_xy[2 * n] = xy[2 * i]; obs_xy[2 * n + 1] = xy[2 * i + 1];
                obs_camera[n] = c;
                obs_point[n] = pt;
                n += (int64_t)((live[i] != 0) & (c >= 0));
            }
            if (n != n0) { if (!track_slot && tracks_out) tracks_out[nt] = t; ++nt; }
        }
        *num_observations = n;
        if (num_tracks_out) *num_tracks_out = nt;
        return OSFM_OK;
    }
    for (int64_t i = 0; i < num_features; ++i) {
        const int32_t t = track_of[i];
        if (track_mask && !track_mask[t]) {
            // features are in track order: skip the rest of an unselected track in one step when the
            // caller's offsets are at hand, else feature by feature
            if (track_offsets) i = track_offsets[t + 1] - 1;
            continue;
        }
        if (!live[i] || camera_of_feature[i] < 0) continue;
        if (n < capacity) {
            if (feature_ids) feature_ids[n] = (int32_t)i;
            obs_xy[2 * n] = xy[2 * i]; obs_xy[2 * n + 1] = xy[2 * i + 1];
            obs_camera[n] = camera_of_feature[i];
            if (t != last) { if (!track_slot && tracks_out) tracks_out[nt] = t; ++nt; last = t; }
            obs_point[n] = track_slot ? track_slot[t] : (int32_t)(nt - 1);
        } else if (t != last) { ++nt; last = t; }
        ++n;
    }
    *num_observations = n;
    if (num_tracks_out) *num_tracks_out = nt;
    if (n > capacity) {
        set_error("tracks_select_observations: %lld observations exceed the capacity %lld", (long long)n, (long long)capacity);
        return OSFM_E_CAPACITY;
    }
    return OSFM_OK;
}

// The feature table of the reconstruction from the tracks (calculateTracksUsingMVE's conversion,
// matching_mve.cpp:455-466): per feature of every track its view, its index in the view, its pixel
// position  float(imageWidth * (double(normalised) + 0.5))  for both axes, the track it belongs to, and
// the features grouped by view (ascending inside a view).  One pass on several host threads; the
// grouping is a stable counting sort with per-thread histograms.
int osfm_tracks_feature_table(int64_t num_tracks, const int64_t *track_offsets, const int32_t *track_features,
    int32_t num_views, const int32_t *view_sizes, const float *const *norm_positions, double image_width,
    int32_t *view_out, int32_t *feat_out, double *xy_out, int32_t *track_of_out,
    int64_t *by_view_out, int64_t *view_start_out)
{
    if (num_tracks < 0 || num_views < 0 || !track_offsets || (num_views > 0 && (!view_sizes || !norm_positions))) {
        set_error("tracks_feature_table: bad arguments");
        return OSFM_E_ARG;
    }
    const int64_t nf = track_offsets[num_tracks];
    if (nf < 0 || (nf > 0 && (!track_features || !view_out || !feat_out || !xy_out))) {
        set_error("tracks_feature_table: null array");
        return OSFM_E_ARG;
    }
    if (track_offsets[0] != 0) { set_error("tracks_feature_table: track_offsets[0] must be 0"); return OSFM_E_ARG; }
    for (int64_t t = 0; t < num_tracks; ++t)
        if (track_offsets[t + 1] < track_offsets[t]) {
            set_error("tracks_feature_table: track_offsets decrease at track %lld", (long long)t);
            return OSFM_E_ARG;
        }
    if ((by_view_out && !view_start_out) || (view_start_out && !by_view_out && nf > 0)) {
        set_error("tracks_feature_table: by_view and view_start go together");
        return OSFM_E_ARG;
    }
    int workers = 1;
    if (const char *e = getenv("OSFM_TRACKS_THREADS")) workers = std::max(1, std::min(64, atoi(e)));
    else if (nf >= (int64_t)1 << 16)
        workers = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::vector<int64_t>> hist((size_t)workers, std::vector<int64_t>((size_t)num_views + 1, 0));
    std::vector<int64_t> bad((size_t)workers, -1);
    auto run = [&](auto &&fn) {
        if (workers == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int w = 1; w < workers; ++w) th.emplace_back([&, w] { fn(w); });
        fn(0);
        for (auto &x : th) x.join();
    };
    run([&](int w) {
        const int64_t i0 = nf * w / workers, i1 = nf * (w + 1) / workers;
        // the track of feature i0: the last offset <= i0
        int64_t t = 0;
        if (track_of_out && num_tracks > 0) {
            int64_t lo = 0, hi = num_tracks - 1;
            while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (track_offsets[mid] <= i0) lo = mid; else hi = mid - 1; }
            t = lo;
        }
        auto &h = hist[(size_t)w];
        for (int64_t i = i0; i < i1; ++i) {
            const int32_t v = track_features[2 * i], f = track_features[2 * i + 1];
            if (v < 0 || v >= num_views || f < 0 || f >= view_sizes[v] || !norm_positions[v]) { bad[(size_t)w] = i; return; }
            view_out[i] = v; feat_out[i] = f;
            const float *p = norm_positions[v] + 2 * (size_t)f;
            xy_out[2 * i] = (double)(float)(image_width * ((double)p[0] + 0.5));
            xy_out[2 * i + 1] = (double)(float)(image_width * ((double)p[1] + 0.5));
            if (track_of_out) {
                while (track_offsets[t + 1] <= i) ++t;
                track_of_out[i] = (int32_t)t;
            }
            h[(size_t)v]++;
        }
    });
    for (int w = 0; w < workers; ++w)
        if (bad[(size_t)w] >= 0) {
            set_error("tracks_feature_table: feature %lld names view %d / feature %d out of range", (long long)bad[(size_t)w],
                track_features[2 * bad[(size_t)w]], track_features[2 * bad[(size_t)w] + 1]);
            return OSFM_E_RANGE;
        }
    if (view_start_out) {
        // first slot of (view, thread): views in order, threads in order inside a view
        int64_t pos = 0;
        for (int v = 0; v < num_views; ++v) {
            view_start_out[v] = pos;
            for (int w = 0; w < workers; ++w) { const int64_t c = hist[(size_t)w][(size_t)v]; hist[(size_t)w][(size_t)v] = pos; pos += c; }
        }
        view_start_out[num_views] = pos;
        run([&](int w) {
            const int64_t i0 = nf * w / workers, i1 = nf * (w + 1) / workers;
            auto &h = hist[(size_t)w];
            for (int64_t i = i0; i < i1; ++i) by_view_out[h[(size_t)view_out[i]]++] = i;
        });
    }
    return OSFM_OK;
}

int osfm_tracks_compute(int32_t num_views, const int32_t *view_sizes, const uint8_t *colors,
    int32_t num_pairs, const osfm_pair *pairs, const int64_t *pair_offsets, const int32_t *corr,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary)
{
    return tracks_compute_impl(num_views, view_sizes, colors, num_pairs, pairs, pair_offsets,
        pair_offsets ? pair_offsets + 1 : nullptr, corr, track_ids, track_capacity, feature_capacity,
        track_offsets, track_features, track_colors, summary);
}

int osfm_tracks_compute_ranges(int32_t num_views, const int32_t *view_sizes, const uint8_t *colors,
    int32_t num_pairs, const osfm_pair *pairs, const int64_t *pair_starts, const int64_t *pair_counts,
    const int32_t *corr,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors,
    osfm_tracks_summary *summary)
{
    if (num_pairs > 0 && (!pair_starts || !pair_counts)) {
        set_error("tracks_compute_ranges: null range arrays");
        return OSFM_E_ARG;
    }
    std::vector<int64_t> ends((size_t)std::max(num_pairs, 0));
    for (int p = 0; p < num_pairs; ++p) ends[p] = pair_starts[p] + pair_counts[p];
    return tracks_compute_impl(num_views, view_sizes, colors, num_pairs, pairs, pair_starts, ends.data(),
        corr, track_ids, track_capacity, feature_capacity, track_offsets, track_features, track_colors,
        summary);
}

}  // extern "C"
