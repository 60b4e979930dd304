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
#include <cmath>
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


// ---------------------------------------------------------------------------
// Group size 3 (what the reference's algorithms use): incremental form.
//
// A seed is a pair (a, b) of used views, its score row row[c] = |T_a & T_b & T_c|
// over all views c.  Scores never change, so a row is computed ONCE, when the
// second view of the pair gets used (a new view c* makes |used| new seeds), and
// kept in HBM (V(V-1)/2 rows of V ints: 250 MB at V = 500).  Every seed caches its
// best remaining candidate; assigning c* invalidates only the seeds whose pick
// was c*, and those re-scan their stored row (V ints), not the bitsets.  One
// iteration = score the new rows, refresh, arg-max over all seeds (a 64-bit
// atomicMax of (score, ~seed order): order independent, hence deterministic).
// ---------------------------------------------------------------------------
__device__ __forceinline__ size_t pair_index(int a, int b) { return (size_t)b * (b - 1) / 2 + a; }   // a < b

// grid (new seeds, candidate chunks); a wave scores one candidate at a time:
// lanes stride over the W words of the three bitsets (row-major [V][W])
__global__ __launch_bounds__(256) void
group3_score_rows_kernel(const uint64_t *__restrict__ bits, int V, int W, const int32_t *__restrict__ new_seeds,
    int32_t *__restrict__ rows)
{
    const int a = new_seeds[2 * blockIdx.x], b = new_seeds[2 * blockIdx.x + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t *Ba = bits + (size_t)a * W, *Bb = bits + (size_t)b * W;
    int32_t *row = rows + pair_index(a, b) * V;
    constexpr int kCandPerBlock = 32;
    const int c0 = blockIdx.y * kCandPerBlock;
    for (int ci = wave; ci < kCandPerBlock; ci += 4) {
        const int c = c0 + ci;
        if (c >= V) break;
        const uint64_t *Bc = bits + (size_t)c * W;
        uint32_t acc0 = 0, acc1 = 0;
        int w = lane;
        for (; w + 64 < W; w += 128) {
            acc0 += (uint32_t)__popcll(Ba[w] & Bb[w] & Bc[w]);
            acc1 += (uint32_t)__popcll(Ba[w + 64] & Bb[w + 64] & Bc[w + 64]);
        }
        if (w < W) acc0 += (uint32_t)__popcll(Ba[w] & Bb[w] & Bc[w]);
        uint32_t acc = acc0 + acc1;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) row[c] = (c == a || c == b) ? 0 : (int32_t)acc;      // group.cpp:123-125
    }
}

// One thread per pair (a < b) of ranks; pairs with both views used are seeds.
// state[v]: 0 remaining, 1 used, 2 not (yet) part of anything.
// best[pair] = (score << 32) | candidate (or 0xffffffff when no candidate shares a
// track); a seed is refreshed when it is new (best == ~0) or its pick was assigned.
__global__ __launch_bounds__(256) void
group3_select_kernel(int V, const uint8_t *__restrict__ state, const int32_t *__restrict__ rows,
    uint64_t *__restrict__ best, int assigned, unsigned long long *__restrict__ result)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t P = (size_t)V * (V - 1) / 2;
    unsigned long long key = 0;
    if (p < P) {
        // invert the triangular index: b = largest with b (b - 1) / 2 <= p
        int b = (int)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
        while ((size_t)b * (b - 1) / 2 > p) --b;
        while ((size_t)(b + 1) * b / 2 <= p) ++b;
        const int a = (int)(p - (size_t)b * (b - 1) / 2);
        if (state[a] == 1 && state[b] == 1) {
            uint64_t cur = best[p];
            const uint32_t cand = (uint32_t)(cur & 0xffffffffu);
            if (cur == ~0ull || (cand != 0xffffffffu && (int)cand == assigned)) {
                const int32_t *row = rows + p * V;
                int bs = 0, bc = -1;
                for (int c = 0; c < V; ++c)
                    if (state[c] == 0 && row[c] > bs) { bs = row[c]; bc = c; }     // first maximum, ascending id
                cur = ((uint64_t)(uint32_t)bs << 32) | (uint32_t)bc;
                best[p] = cur;
            }
            // first maximum in the lexicographic (a, b) order of the sorted used views
            const uint32_t order = (uint32_t)a * (uint32_t)V + (uint32_t)b;
            key = ((cur >> 32) << 32) | (uint32_t)(~order);
            key += 1ull << 63;                       // any seed beats "no seed"
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(key, m);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(result, key);
}

__global__ void
group3_init_kernel(uint64_t *best, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) best[i] = ~0ull;
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
    if (group_size == 3) {
        // ---- incremental form (see the kernels above) --------------------------
        const size_t P = (size_t)V * (V - 1) / 2;
        const size_t rows_bytes = P * (size_t)V * 4;
        size_t free_b = 0, total_b = 0;
        OSFM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        if (rows_bytes + ((size_t)1 << 30) < free_b) {
            // row-major bitsets [V][W]
            std::vector<uint64_t> bits((size_t)V * W, 0);
            for (int w = 0; w < W; ++w)
                for (int v = 0; v < V; ++v) bits[(size_t)v * W + w] = bt[(size_t)w * V + v];
            DeviceBuffer d_bits, d_rows, d_best, d_state, d_new, d_result;
            OSFM_RETURN_IF(d_bits.reserve(bits.size() * 8));
            OSFM_RETURN_IF(d_rows.reserve(rows_bytes));
            OSFM_RETURN_IF(d_best.reserve(P * 8));
            OSFM_RETURN_IF(d_state.reserve((size_t)V));
            OSFM_RETURN_IF(d_new.reserve((size_t)V * 8));
            OSFM_RETURN_IF(d_result.reserve(8));
            OSFM_HIP_CHECK(hipMemcpy(d_bits.ptr, bits.data(), bits.size() * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(group3_init_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, d_best.as<uint64_t>(), P);
            std::vector<uint8_t> state(V, 2);
            for (int i = 2; i < V; ++i) state[rank_of[view_ids[i]]] = 0;
            const int r0 = rank_of[view_ids[0]], r1 = rank_of[view_ids[1]];
            if (r0 == r1) { set_error("build_groups: views 0 and 1 are the same view"); return OSFM_E_ARG; }
            // the first seed (views 0 and 1) counts as used for the selection
            state[r0] = 1; state[r1] = 1;
            std::vector<int32_t> new_seeds = {std::min(r0, r1), std::max(r0, r1)};
            int assigned = -1, remaining_n = V - 2;
            bool first = true;
            while (first || remaining_n > 0) {
                const int nn = (int)(new_seeds.size() / 2);
                OSFM_HIP_CHECK(hipMemcpyAsync(d_state.ptr, state.data(), (size_t)V, hipMemcpyHostToDevice, 0));
                if (nn > 0) {
                    OSFM_HIP_CHECK(hipMemcpyAsync(d_new.ptr, new_seeds.data(), new_seeds.size() * 4, hipMemcpyHostToDevice, 0));
                    hipLaunchKernelGGL(group3_score_rows_kernel, dim3(nn, (V + 31) / 32), dim3(256), 0, 0, d_bits.as<uint64_t>(), V, W,
                        d_new.as<int32_t>(), d_rows.as<int32_t>());
                }
                OSFM_HIP_CHECK(hipMemsetAsync(d_result.ptr, 0, 8, 0));
                hipLaunchKernelGGL(group3_select_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, V, d_state.as<uint8_t>(),
                    d_rows.as<int32_t>(), d_best.as<uint64_t>(), assigned, d_result.as<unsigned long long>());
                OSFM_HIP_CHECK(hipGetLastError());
                unsigned long long key = 0;
                OSFM_HIP_CHECK(hipMemcpy(&key, d_result.ptr, 8, hipMemcpyDeviceToHost));
                if (!key) { set_error("build_groups: no seed group"); return OSFM_E_STATE; }
                const uint32_t order = ~(uint32_t)(key & 0xffffffffu);
                const int a = (int)(order / (uint32_t)V), b = (int)(order % (uint32_t)V);
                const int added = (int)((key >> 32) & 0x7fffffffu);
                // the winning seed's pick
                uint64_t cur = 0;
                OSFM_HIP_CHECK(hipMemcpy(&cur, d_best.as<uint64_t>() + ((size_t)b * (b - 1) / 2 + a), 8, hipMemcpyDeviceToHost));
                const int c = (int)(int32_t)(uint32_t)(cur & 0xffffffffu);
                if (*num_groups >= max_groups) { set_error("build_groups: more than %d groups", max_groups); return OSFM_E_CAPACITY; }
                if (c < 0) {
                    set_error("build_groups: a remaining view shares no track with any seed group "
                              "(the reference loops forever here, group.cpp:64-66)");
                    return OSFM_E_STATE;
                }
                int32_t *g = groups + (size_t)(*num_groups) * 3;
                if (first) { g[0] = view_ids[0]; g[1] = view_ids[1]; }       // group.cpp:27-30: views 0 and 1 in that order
                else { g[0] = view_ids[by_id[a]]; g[1] = view_ids[by_id[b]]; }
                g[2] = view_ids[by_id[c]];
                group_tracks[*num_groups] = added;
                ++*num_groups;
                // the new view joins the used set: its pairs with every used view are the new seeds
                new_seeds.clear();
                for (int u = 0; u < V; ++u)
                    if (state[u] == 1) { new_seeds.push_back(std::min(u, c)); new_seeds.push_back(std::max(u, c)); }
                state[c] = 1;
                assigned = c;
                --remaining_n;
                first = false;
            }
            return OSFM_OK;
        }
        // not enough device memory for the score rows: the general path below
    }
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
