/*
 * ref_shim_tracks.cc -- TEST INFRASTRUCTURE.  extern "C" shim over the
 * REFERENCE's own track builder (src/mve/sfm/bundler_tracks.cc, compiled where
 * it lies by oracle/Makefile into oracle/_ref/libref_tracks.so) so that the
 * restatement in tracks_oracle.c can be pinned against it.  No algorithm here:
 * flat arrays are marshalled into the reference's types and back.
 */
#include <cstdint>
#include <vector>

#include "sfm/bundler_common.h"
#include "sfm/bundler_tracks.h"

extern "C" {

/* Layout of all arrays: see oracle_tracks_compute in tracks_oracle.c.
 * Returns the number of tracks, or -1 when an output capacity is too small. */
__attribute__((visibility("default"))) int
ref_tracks_compute(int num_views, const int32_t *view_sizes, const uint8_t *colors,
    int num_pairs, const int32_t *pairs, const int64_t *pair_offsets, const int32_t *corr,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors)
{
    sfm::bundler::ViewportList viewports(num_views);
    std::size_t g = 0;
    for (int v = 0; v < num_views; ++v) {
        viewports[v].features.positions.resize(view_sizes[v]);
        viewports[v].features.colors.resize(view_sizes[v]);
        for (int f = 0; f < view_sizes[v]; ++f, ++g)
            for (int c = 0; c < 3; ++c)
                viewports[v].features.colors[f][c] = colors ? colors[3 * g + c] : 0;
    }
    sfm::bundler::PairwiseMatching matching(num_pairs);
    for (int p = 0; p < num_pairs; ++p) {
        matching[p].view_1_id = pairs[2 * p];
        matching[p].view_2_id = pairs[2 * p + 1];
        for (int64_t k = pair_offsets[p]; k < pair_offsets[p + 1]; ++k)
            matching[p].matches.push_back(sfm::CorrespondenceIndex(corr[2 * k], corr[2 * k + 1]));
    }
    sfm::bundler::TrackList tracks;
    sfm::bundler::Tracks::Options opts;
    sfm::bundler::Tracks builder(opts);
    builder.compute(matching, &viewports, &tracks);

    g = 0;
    for (int v = 0; v < num_views; ++v)
        for (int f = 0; f < view_sizes[v]; ++f, ++g)
            track_ids[g] = viewports[v].track_ids[f];
    if ((int64_t)tracks.size() > track_capacity) return -1;
    int64_t nf = 0;
    for (std::size_t t = 0; t < tracks.size(); ++t) {
        track_offsets[t] = nf;
        for (std::size_t k = 0; k < tracks[t].features.size(); ++k) {
            if (nf >= feature_capacity) return -1;
            track_features[2 * nf] = tracks[t].features[k].view_id;
            track_features[2 * nf + 1] = tracks[t].features[k].feature_id;
            ++nf;
        }
        for (int c = 0; c < 3; ++c) track_colors[3 * t + c] = tracks[t].color[c];
    }
    track_offsets[tracks.size()] = nf;
    return (int)tracks.size();
}

}  /* extern "C" */
