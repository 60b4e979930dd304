/*
 * tracks_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of sfm::bundler::Tracks::compute
 * (src/mve/sfm/bundler_tracks.cc:49-145) with unify_tracks (:23-45) and
 * remove_invalid_tracks (:149-203), kept literal: one growable feature array
 * per track, the same branch order, the same "unify into the larger track"
 * rule, the same clean-up.  Pinned: compared with the reference's own file
 * (compiled from /root/reference into oracle/_ref/libref_tracks.so) on
 * randomised matchings in tests/test_oracle_tracks.py.
 *
 * Flat layout (shared with the product's osfm_tracks_compute):
 *   view_sizes   [num_views]            features per view (positions.size())
 *   colors       [sum view_sizes][3]    FeatureSet::colors, views concatenated
 *   pairs        [num_pairs][2]         TwoViewMatching::view_1_id / view_2_id
 *   pair_offsets [num_pairs + 1]        match range of a pair in corr
 *   corr         [..][2]                CorrespondenceIndex (first, second)
 *   track_ids    [sum view_sizes]       Viewport::track_ids, views concatenated
 *   track_offsets[num_tracks + 1], track_features [..][2] (view_id, feature_id),
 *   track_colors [num_tracks][3]
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

typedef struct { int32_t *v; int64_t n, cap; } ivec;     /* (view, feature) pairs, n = pair count */

static void push(ivec *a, int view, int feat)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 4;
        a->v = (int32_t *)realloc(a->v, sizeof(int32_t) * 2 * a->cap);
    }
    a->v[2 * a->n] = view; a->v[2 * a->n + 1] = feat; a->n++;
}

ORACLE_API int
oracle_tracks_compute(int num_views, const int32_t *view_sizes, const uint8_t *colors,
    int num_pairs, const int32_t *pairs, const int64_t *pair_offsets, const int32_t *corr,
    int32_t *track_ids, int64_t track_capacity, int64_t feature_capacity,
    int64_t *track_offsets, int32_t *track_features, uint8_t *track_colors, int32_t *num_invalid)
{
    int64_t *voff = (int64_t *)malloc(sizeof(int64_t) * (num_views + 1));
    voff[0] = 0;
    for (int v = 0; v < num_views; ++v) voff[v + 1] = voff[v] + view_sizes[v];
    for (int64_t g = 0; g < voff[num_views]; ++g) track_ids[g] = -1;          /* :54-58 */

    ivec *tracks = NULL;
    int64_t nt = 0, tcap = 0;
    for (int p = 0; p < num_pairs; ++p) {                                      /* :66-119 */
        const int v1 = pairs[2 * p], v2 = pairs[2 * p + 1];
        for (int64_t k = pair_offsets[p]; k < pair_offsets[p + 1]; ++k) {
            const int f1 = corr[2 * k], f2 = corr[2 * k + 1];
            int32_t *t1 = &track_ids[voff[v1] + f1], *t2 = &track_ids[voff[v2] + f2];
            if (*t1 == -1 && *t2 == -1) {
                if (nt == tcap) {
                    tcap = tcap ? 2 * tcap : 1024;
                    tracks = (ivec *)realloc(tracks, sizeof(ivec) * tcap);
                }
                memset(&tracks[nt], 0, sizeof(ivec));
                *t1 = (int32_t)nt; *t2 = (int32_t)nt;
                push(&tracks[nt], v1, f1);
                push(&tracks[nt], v2, f2);
                nt++;
            } else if (*t1 == -1 && *t2 != -1) {
                *t1 = *t2;
                push(&tracks[*t2], v1, f1);
            } else if (*t1 != -1 && *t2 == -1) {
                *t2 = *t1;
                push(&tracks[*t1], v2, f2);
            } else if (*t1 == *t2) {
                /* already propagated */
            } else {
                /* unify_tracks (:23-45): into the larger one, the first on a draw */
                int a = *t1, b = *t2;
                if (tracks[a].n < tracks[b].n) { const int s = a; a = b; b = s; }
                for (int64_t q = 0; q < tracks[b].n; ++q)
                    track_ids[voff[tracks[b].v[2 * q]] + tracks[b].v[2 * q + 1]] = a;
                for (int64_t q = 0; q < tracks[b].n; ++q)
                    push(&tracks[a], tracks[b].v[2 * q], tracks[b].v[2 * q + 1]);
                free(tracks[b].v);
                memset(&tracks[b], 0, sizeof(ivec));
            }
        }
    }

    /* remove_invalid_tracks (:149-203) */
    uint8_t *del = (uint8_t *)calloc(nt > 0 ? nt : 1, 1);
    int64_t *seen = (int64_t *)malloc(sizeof(int64_t) * (num_views > 0 ? num_views : 1));
    for (int v = 0; v < num_views; ++v) seen[v] = -1;
    int invalid = 0;
    for (int64_t t = 0; t < nt; ++t) {
        if (tracks[t].n == 0) { del[t] = 1; continue; }
        for (int64_t q = 0; q < tracks[t].n; ++q) {
            const int v = tracks[t].v[2 * q];
            if (seen[v] == t) { invalid++; del[t] = 1; break; }
            seen[v] = t;
        }
    }
    int32_t *map = (int32_t *)malloc(sizeof(int32_t) * (nt > 0 ? nt : 1));
    int32_t valid = 0;
    for (int64_t t = 0; t < nt; ++t) map[t] = del[t] ? -1 : valid++;
    for (int64_t g = 0; g < voff[num_views]; ++g)
        if (track_ids[g] >= 0) track_ids[g] = map[track_ids[g]];

    /* colours (:130-144) and output */
    int rc = valid;
    int64_t nf = 0;
    if (valid > track_capacity) rc = -1;
    for (int64_t t = 0; t < nt && rc >= 0; ++t) {
        if (del[t]) continue;
        const int32_t o = map[t];
        track_offsets[o] = nf;
        float col[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int64_t q = 0; q < tracks[t].n; ++q) {
            if (nf >= feature_capacity) { rc = -1; break; }
            const int v = tracks[t].v[2 * q], f = tracks[t].v[2 * q + 1];
            track_features[2 * nf] = v; track_features[2 * nf + 1] = f; nf++;
            for (int c = 0; c < 3; ++c) col[c] += colors ? (float)colors[3 * (voff[v] + f) + c] : 0.0f;
            col[3] += 1.0f;
        }
        for (int c = 0; c < 3; ++c) track_colors[3 * o + c] = (uint8_t)(col[c] / col[3] + 0.5f);
    }
    if (rc >= 0) track_offsets[valid] = nf;
    if (num_invalid) *num_invalid = invalid;
    for (int64_t t = 0; t < nt; ++t) free(tracks[t].v);
    free(tracks); free(del); free(seen); free(map); free(voff);
    return rc;
}
