/*
 * groups_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of orthosfm::buildGroups / completeGroup /
 * getAllPossibleCombinations (src/data_structures/group.cpp:13-88, 90-155,
 * 157-210) with filterTracksToAvailableCameras (src/util/common.cpp:85-139)
 * kept literal (every score is a fresh pass over the tracks).
 *
 * PARITY UNPINNED: the reference file needs OpenCV / Eigen / Boost types and
 * cannot be built here, and the reference holds no vectors for it.  One
 * deliberate choice: completeGroup scores the candidates inside an OpenMP
 * parallel loop whose critical section keeps the FIRST strictly better score
 * it happens to see (group.cpp:118-146), so ties are broken by thread timing in
 * the reference; here (and in the product) candidates are visited in ascending
 * id order, which is what the reference does with one thread.
 *
 * Flat layout: view_ids [num_views] in the order of the `views` vector; tracks
 * as CSR (track_offsets [num_tracks + 1], track_views [..] = Feature::viewID).
 * Output: groups [max_groups][group_size] ids, group_tracks [max_groups]
 * (ViewGroup::tracks); returns the number of groups, or -1 when the loop cannot
 * make progress (the reference would spin forever: a remaining view shares no
 * track with any seed) or max_groups is too small.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

typedef struct { const int64_t *off; const int32_t *views; int n; } tracks_t;

static int in_ids(const int32_t *ids, int n, int32_t v)
{
    for (int i = 0; i < n; ++i) if (ids[i] == v) return 1;
    return 0;
}

/* number of features of track t whose view is in ids */
static int count_in(const tracks_t *tr, int t, const int32_t *ids, int n)
{
    int c = 0;
    for (int64_t k = tr->off[t]; k < tr->off[t + 1]; ++k) c += in_ids(ids, n, tr->views[k]);
    return c;
}

/* completeGroup (group.cpp:90-155); ids: seed in, completed group out */
static int complete_group(const tracks_t *tr, int32_t *ids, int n_seed, const int32_t *remaining,
    int n_remaining, int group_size)
{
    int added = 0;
    uint8_t *seed_pre = (uint8_t *)malloc(tr->n > 0 ? tr->n : 1), *pre = (uint8_t *)malloc(tr->n > 0 ? tr->n : 1);
    for (int t = 0; t < tr->n; ++t) seed_pre[t] = count_in(tr, t, ids, n_seed) > 1;        /* :103 */
    int n = n_seed;
    while (n < group_size) {
        for (int t = 0; t < tr->n; ++t) pre[t] = seed_pre[t] && count_in(tr, t, ids, n) > 1;   /* :110 */
        unsigned best = 0;
        int32_t best_id = 0;
        for (int i = 0; i < n_remaining; ++i) {
            const int32_t id = remaining[i];
            if (in_ids(ids, n, id)) continue;
            ids[n] = id;
            unsigned score = 0;
            for (int t = 0; t < tr->n; ++t) score += pre[t] && count_in(tr, t, ids, n + 1) == n + 1;   /* :131-134 */
            if (score > best) { best = score; best_id = id; }
        }
        ids[n++] = best_id;
        added = (int)best;
    }
    free(seed_pre); free(pre);
    return added;
}

static int cmp_i32(const void *a, const void *b)
{
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

ORACLE_API int
oracle_build_groups(int num_views, const int32_t *view_ids, int num_tracks, const int64_t *track_offsets,
    const int32_t *track_views, int group_size, int max_groups, int32_t *groups, int32_t *group_tracks)
{
    if (num_views < 2 || group_size < 2 || group_size > 8) return -1;
    tracks_t tr = { track_offsets, track_views, num_tracks };
    int32_t *to_assign = (int32_t *)malloc(sizeof(int32_t) * num_views), *used = (int32_t *)malloc(sizeof(int32_t) * num_views);
    int n_assign = 0, n_used = 0, ng = 0, rc = 0;
    for (int i = 2; i < num_views; ++i) to_assign[n_assign++] = view_ids[i];
    qsort(to_assign, n_assign, sizeof(int32_t), cmp_i32);                /* std::set order */
    int32_t g[8];
    g[0] = view_ids[0]; g[1] = view_ids[1];
    int added = complete_group(&tr, g, 2, to_assign, n_assign, group_size);
    if (max_groups < 1) rc = -1;
    while (rc == 0) {
        memcpy(groups + (size_t)ng * group_size, g, sizeof(int32_t) * group_size);
        group_tracks[ng++] = added;
        int progress = 0;
        for (int i = 0; i < group_size; ++i) {
            if (in_ids(used, n_used, g[i])) continue;
            for (int k = 0; k < n_assign; ++k)
                if (to_assign[k] == g[i]) { memmove(to_assign + k, to_assign + k + 1, sizeof(int32_t) * (n_assign - k - 1)); n_assign--; progress = 1; break; }
            used[n_used++] = g[i];
        }
        if (ng == 1) progress = 1;
        if (n_assign == 0) break;
        if (!progress || ng >= max_groups) { rc = -1; break; }
        /* all (group_size - 1)-combinations of the used ids in lexicographic order (:157-210) */
        qsort(used, n_used, sizeof(int32_t), cmp_i32);
        const int k = group_size - 1;
        if (n_used < k) { rc = -1; break; }
        int idx[8];
        for (int i = 0; i < k; ++i) idx[i] = i;
        int best_added = -1;
        int32_t best[8], cand[8];
        for (;;) {
            for (int i = 0; i < k; ++i) cand[i] = used[idx[i]];
            const int a = complete_group(&tr, cand, k, to_assign, n_assign, group_size);
            if (a > best_added) { best_added = a; memcpy(best, cand, sizeof(int32_t) * group_size); }
            int p = k - 1;
            while (p >= 0 && idx[p] == n_used - k + p) --p;
            if (p < 0) break;
            ++idx[p];
            for (int q = p + 1; q < k; ++q) idx[q] = idx[q - 1] + 1;
        }
        memcpy(g, best, sizeof(int32_t) * group_size);
        added = best_added;
    }
    free(to_assign); free(used);
    return rc == 0 ? ng : -1;
}
