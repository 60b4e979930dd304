// Group ordering at scale: orthosfm::buildGroups
// (src/data_structures/group.cpp:13-88) with completeGroup (:90-155) and
// getAllPossibleCombinations (:157-210).
//
// The reference scores every (seed, candidate view) by re-filtering the whole
// track list (filterTracksToAvailableCameras, src/util/common.cpp:85-139): the
// score is the number of tracks that contain every view of the group plus the
// candidate.  Here every view owns a bitset over the tracks; a score is
// popcount(AND of the group's bitsets AND the candidate's), evaluated on the
// device for all (seed, candidate) pairs of an iteration at once, and the
// greedy loop stays on the host.  Seeds are evaluated lazily: a seed's best
// candidate stays valid until that candidate is assigned (scores do not depend
// on the remaining set), so an iteration re-scores only the new seeds and the
// invalidated ones.
//
// Candidates are visited in ascending id order (ties: smallest id), which is the
// reference's single-threaded behaviour; its OpenMP loop breaks ties by thread
// timing (group.cpp:118-146).
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "osfm_common.h"

namespace osfm {

constexpr int kMaxGroup = 8;
constexpr int kGroupThreads = 256;

// One workgroup per seed.  bt: transposed bitsets [W][V] (word w of view v at
// bt[w * V + v]); seeds: [S][n_seed] view indices; cand: [R] view indices in
// ascending id order.  out: per seed (best score, index into cand of the first
// candidate reaching it); score 0 -> index -1.
__global__ __launch_bounds__(kGroupThreads) void
group_score_kernel(const uint64_t *__restrict__ bt, int V, int W, const int32_t *__restrict__ seeds,
    int n_seed, const int32_t *__restrict__ cand, int R, uint32_t *__restrict__ best_score,
    int32_t *__restrict__ best_idx)
{
    __shared__ uint32_t s_score[kGroupThreads];
    __shared__ int32_t s_idx[kGroupThreads];
    const int s = blockIdx.x, tid = threadIdx.x;
    int sv[kMaxGroup];
    for (int j = 0; j < n_seed; ++j) sv[j] = seeds[(size_t)s * n_seed + j];
    uint32_t my_best = 0;
    int32_t my_idx = -1;
    for (int r0 = 0; r0 < R; r0 += kGroupThreads) {
        const int r = r0 + tid;
        const int c = r < R ? cand[r] : 0;
        uint32_t acc = 0;
        for (int w = 0; w < W; ++w) {
            const uint64_t *row = bt + (size_t)w * V;
            uint64_t m = row[sv[0]];
            for (int j = 1; j < n_seed; ++j) m &= row[sv[j]];
            acc += (uint32_t)__popcll(m & row[c]);
        }
        bool in_seed = false;
        for (int j = 0; j < n_seed; ++j) in_seed |= sv[j] == c;        // group.cpp:123-125
        if (r < R && !in_seed && acc > my_best) { my_best = acc; my_idx = r; }   // strictly better, ascending r
    }
    s_score[tid] = my_best; s_idx[tid] = my_idx;
    __syncthreads();
    for (int st = kGroupThreads / 2; st >= 1; st >>= 1) {
        if (tid < st) {
            const uint32_t o = s_score[tid + st];
            const int32_t oi = s_idx[tid + st];
            // higher score wins; equal scores: the smaller candidate index (seen first)
            if (o > s_score[tid] || (o == s_score[tid] && o > 0 && oi < s_idx[tid])) { s_score[tid] = o; s_idx[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) { best_score[s] = s_score[0]; best_idx[s] = s_score[0] > 0 ? s_idx[0] : -1; }
}

}  // namespace osfm

using namespace osfm;

extern "C" {

int osfm_build_groups(int device, int32_t num_views, const int32_t *view_ids, int32_t num_tracks,
    const int64_t *track_offsets, const int32_t *track_views, int32_t group_size, int32_t max_groups,
    int32_t *groups, int32_t *group_tracks, int32_t *num_groups)
{
    if (!view_ids || !num_groups || !groups || !group_tracks || num_views < 2 || num_tracks < 0 ||
        (num_tracks > 0 && (!track_offsets || !track_views))) {
        set_error("build_groups: null array / bad counts");
        return OSFM_E_ARG;
    }
    if (group_size < 3 || group_size > kMaxGroup) {
        set_error("build_groups: group size %d outside [3, %d] (the reference's algorithms use 3)", group_size, kMaxGroup);
        return OSFM_E_ARG;
    }
    *num_groups = 0;
    const int V = num_views;
    // ranks by ascending id: internal index order == id order (std::set / std::sort in the reference)
    std::vector<int> by_id(V);
    for (int i = 0; i < V; ++i) by_id[i] = i;
    std::sort(by_id.begin(), by_id.end(), [&](int a, int b) { return view_ids[a] < view_ids[b]; });
    std::map<int32_t, int> rank_of;
    for (int r = 0; r < V; ++r) {
        if (!rank_of.insert({view_ids[by_id[r]], r}).second) { set_error("build_groups: duplicate view id"); return OSFM_E_ARG; }
    }
    // bitsets over the tracks, transposed
    const int W = std::max(1, (num_tracks + 63) / 64);
    std::vector<uint64_t> bt((size_t)W * V, 0);
    for (int t = 0; t < num_tracks; ++t)
        for (int64_t k = track_offsets[t]; k < track_offsets[t + 1]; ++k) {
            auto it = rank_of.find(track_views[k]);
            if (it != rank_of.end()) bt[(size_t)(t >> 6) * V + it->second] |= 1ull << (t & 63);
        }
    OSFM_HIP_CHECK(hipSetDevice(device));
    DeviceBuffer d_bt, d_seeds, d_cand, d_score, d_idx;
    struct Cleanup { DeviceBuffer *b[5]; ~Cleanup() { for (auto *x : b) x->release(); } } cleanup{{&d_bt, &d_seeds, &d_cand, &d_score, &d_idx}};
    OSFM_RETURN_IF(d_bt.reserve(bt.size() * 8));
    OSFM_HIP_CHECK(hipMemcpy(d_bt.ptr, bt.data(), bt.size() * 8, hipMemcpyHostToDevice));

    std::vector<int> remaining;                         // ranks, ascending
    std::vector<uint8_t> is_used(V, 0), is_remaining(V, 0);
    for (int i = 2; i < V; ++i) is_remaining[rank_of[view_ids[i]]] = 1;
    auto rebuild_remaining = [&]() { remaining.clear(); for (int r = 0; r < V; ++r) if (is_remaining[r]) remaining.push_back(r); };
    rebuild_remaining();

    // scores (seeds x remaining) on the device; returns per seed (score, rank of the best candidate or -1)
    auto score_seeds = [&](const std::vector<int32_t> &seeds, int n_seed, std::vector<uint32_t> *sc, std::vector<int32_t> *best) -> int {
        const int S = (int)(seeds.size() / n_seed), R = (int)remaining.size();
        sc->assign(S, 0); best->assign(S, -1);
        if (S == 0 || R == 0) return OSFM_OK;
        OSFM_RETURN_IF(d_seeds.reserve(seeds.size() * 4));
        OSFM_RETURN_IF(d_cand.reserve((size_t)R * 4));
        OSFM_RETURN_IF(d_score.reserve((size_t)S * 4));
        OSFM_RETURN_IF(d_idx.reserve((size_t)S * 4));
        OSFM_HIP_CHECK(hipMemcpy(d_seeds.ptr, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice));
        OSFM_HIP_CHECK(hipMemcpy(d_cand.ptr, remaining.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(group_score_kernel, dim3(S), dim3(kGroupThreads), 0, 0, d_bt.as<uint64_t>(), V, W,
            d_seeds.as<int32_t>(), n_seed, d_cand.as<int32_t>(), R, d_score.as<uint32_t>(), d_idx.as<int32_t>());
        OSFM_HIP_CHECK(hipGetLastError());
        std::vector<int32_t> idx(S);
        OSFM_HIP_CHECK(hipMemcpy(sc->data(), d_score.ptr, (size_t)S * 4, hipMemcpyDeviceToHost));
        OSFM_HIP_CHECK(hipMemcpy(idx.data(), d_idx.ptr, (size_t)S * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < S; ++i) (*best)[i] = idx[i] >= 0 ? remaining[idx[i]] : -1;
        return OSFM_OK;
    };
    // completeGroup for a batch of seeds of n_seed ranks each: greedy, one view at a time
    auto complete = [&](std::vector<int32_t> seeds, int n_seed, std::vector<int32_t> *full, std::vector<int32_t> *added) -> int {
        const int S = (int)(seeds.size() / n_seed);
        std::vector<int32_t> cur = seeds;
        int n = n_seed;
        added->assign(S, 0);
        while (n < group_size) {
            std::vector<uint32_t> sc;
            std::vector<int32_t> best;
            OSFM_RETURN_IF(score_seeds(cur, n, &sc, &best));
            std::vector<int32_t> nxt((size_t)S * (n + 1));
            for (int i = 0; i < S; ++i) {
                memcpy(&nxt[(size_t)i * (n + 1)], &cur[(size_t)i * n], sizeof(int32_t) * n);
                // no candidate shares a track: the reference appends view id 0 (bestViewID's start value)
                nxt[(size_t)i * (n + 1) + n] = best[i];
                (*added)[i] = (int32_t)sc[i];
            }
            cur.swap(nxt);
            ++n;
        }
        full->swap(cur);
        return OSFM_OK;
    };
    auto emit = [&](const int32_t *g, int added) -> int {
        if (*num_groups >= max_groups) { set_error("build_groups: more than %d groups", max_groups); return OSFM_E_CAPACITY; }
        for (int i = 0; i < group_size; ++i) {
            if (g[i] < 0) {
                set_error("build_groups: a remaining view shares no track with any seed group "
                          "(the reference loops forever here, group.cpp:64-66)");
                return OSFM_E_STATE;
            }
            groups[(size_t)(*num_groups) * group_size + i] = view_ids[by_id[g[i]]];
        }
        group_tracks[*num_groups] = added;
        ++*num_groups;
        for (int i = 0; i < group_size; ++i) { is_used[g[i]] = 1; is_remaining[g[i]] = 0; }
        rebuild_remaining();
        return OSFM_OK;
    };

    // first group: views 0 and 1 (group.cpp:27-38)
    {
        std::vector<int32_t> seed = {rank_of[view_ids[0]], rank_of[view_ids[1]]}, full, added;
        OSFM_RETURN_IF(complete(seed, 2, &full, &added));
        OSFM_RETURN_IF(emit(full.data(), added[0]));
    }
    // seeds = all (group_size - 1)-combinations of the used views in lexicographic order;
    // cache: completed group and score per seed, valid while its added views are still unassigned
    struct Done { std::vector<int32_t> full; int32_t added; };
    std::map<std::vector<int32_t>, Done> cache;
    const int k = group_size - 1;
    while (!remaining.empty()) {
        std::vector<int> used;
        for (int r = 0; r < V; ++r) if (is_used[r]) used.push_back(r);
        if ((int)used.size() < k) { set_error("build_groups: fewer used views than a seed needs"); return OSFM_E_STATE; }
        // enumerate the combinations; collect those that need (re)scoring
        std::vector<std::vector<int32_t>> combos;
        std::vector<int> idx(k);
        for (int i = 0; i < k; ++i) idx[i] = i;
        for (;;) {
            std::vector<int32_t> c(k);
            for (int i = 0; i < k; ++i) c[i] = used[idx[i]];
            combos.push_back(c);
            int p = k - 1;
            while (p >= 0 && idx[p] == (int)used.size() - k + p) --p;
            if (p < 0) break;
            ++idx[p];
            for (int q = p + 1; q < k; ++q) idx[q] = idx[q - 1] + 1;
        }
        std::vector<int32_t> todo;
        std::vector<size_t> todo_of;
        for (size_t ci = 0; ci < combos.size(); ++ci) {
            auto it = cache.find(combos[ci]);
            bool valid = it != cache.end();
            if (valid)
                for (int j = k; j < group_size; ++j) {
                    const int32_t a = it->second.full[j];
                    if (a >= 0 && !is_remaining[a]) valid = false;       // its pick has been assigned meanwhile
                }
            if (valid && it->second.added == 0 && it->second.full[k] < 0) valid = false;   // re-try empty results
            if (!valid) { todo.insert(todo.end(), combos[ci].begin(), combos[ci].end()); todo_of.push_back(ci); }
        }
        if (!todo_of.empty()) {
            std::vector<int32_t> full, added;
            OSFM_RETURN_IF(complete(todo, k, &full, &added));
            for (size_t i = 0; i < todo_of.size(); ++i) {
                Done d;
                d.full.assign(full.begin() + (ptrdiff_t)i * group_size, full.begin() + (ptrdiff_t)(i + 1) * group_size);
                d.added = added[i];
                cache[combos[todo_of[i]]] = d;
            }
        }
        int best_added = -1;
        const Done *best = nullptr;
        for (auto const &c : combos) {                                      // group.cpp:55-62: first maximum
            const Done &d = cache[c];
            if (d.added > best_added) { best_added = d.added; best = &d; }
        }
        OSFM_RETURN_IF(emit(best->full.data(), best->added));
    }
    return OSFM_OK;
}

}  // extern "C"
