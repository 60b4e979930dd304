// Camera-pair lists of the Schur complement, built on the device.
//
// For every track, every ordered pair of its observations (a, b) with
// cam(a) >= cam(b) (both cameras having free parameters) contributes a block to
// S[cam(a)][cam(b)].  The pair pass wants these grouped by camera pair, in a
// reproducible order: keys cam(a) * C + cam(b) are generated track by track at
// offsets given by a prefix sum, then sorted with a STABLE radix sort, so the
// order inside one camera pair is the track order on every run.
// (Host-side construction of the same lists cost ~14 ms for 3.6 M entries;
// here it is a few launches.)
#include <cstdlib>
#include <vector>
#include <hipcub/hipcub.hpp>

#include "ba_kernels.h"
#include "osfm_common.h"

namespace osfm {

// One thread per observation a: the entries (a, b) of its track with cam(a) >= cam(b), both free.  The
// observations are in track order, so an exclusive sum over these counts is the position of a's
// first entry in the track-by-track, a-major, b-minor order the stable sort starts from.  (One
// thread per TRACK, as this was first written, left a 200-camera global adjustment -- 2500 tracks of
// 80 observations, 8 M entries -- with 2500 threads of 3000+ serial stores each: 4.7 of the call's
// 12.5 ms.)
__global__ void
pair_count_kernel(BaDev d, int with_points, int32_t *counts)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= d.O) return;
    const int ca = d.obs_cam[a];
    int n = 0;
    if (d.cam_ldim[ca] != 0) {
        if (with_points != 1) n = 1;             // 0: no points (no Schur term); 2: the diagonal pairs only
        else {
            const int j = d.obs_pt[a];
            const int k0 = d.pt_start[j], k1 = d.pt_start[j + 1];
            for (int b = k0; b < k1; ++b) {
                const int cb = d.obs_cam[b];
                n += (d.cam_ldim[cb] != 0 && ca >= cb) ? 1 : 0;
            }
        }
    }
    counts[a] = n;
}

// Key of the camera pair (ca, cb), ca >= cb: the sorted order is the order the pair pass's waves take the pairs
// in, an XCD a contiguous range each.  Rows of `group` cameras ca go through the cameras cb together -- (ca / group,
// cb, ca % group) -- so that the records of a camera cb, fetched into an XCD's L2 for one pair, serve the
// other pairs of the group while they are there: in plain (ca, cb) order the b side of every pair came from
// beyond the L2 (a global adjustment of 500 cameras streams 7 GB of them per pass).
__host__ __device__ __forceinline__ uint32_t pair_key_of(uint32_t ca, uint32_t cb, uint32_t C, uint32_t group)
{
    return ((ca / group) * C + cb) * group + ca % group;
}

__global__ void
pair_fill_kernel(BaDev d, int with_points, int group, const int32_t *offsets, uint32_t *keys, uint64_t *vals)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= d.O) return;
    const int ca = d.obs_cam[a];
    if (d.cam_ldim[ca] == 0) return;
    int pos = offsets[a];
    int k0 = a, k1 = a + 1;
    if (with_points == 1) { const int j = d.obs_pt[a]; k0 = d.pt_start[j]; k1 = d.pt_start[j + 1]; }
    for (int b = k0; b < k1; ++b) {
        const int cb = d.obs_cam[b];
        if (d.cam_ldim[cb] == 0 || ca < cb) continue;
        keys[pos] = pair_key_of((uint32_t)ca, (uint32_t)cb, (uint32_t)d.C, (uint32_t)group);
        vals[pos] = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
        pos++;
    }
}

// chunks (= pair-pass waves) per camera pair; entry num_pairs is the trailing zero of the scan
__global__ void
pair_chunk_count_kernel(const int32_t *runs, int num_pairs, int chunk, int32_t *chunks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > num_pairs) return;
    chunks[i] = i < num_pairs ? (runs[i] + chunk - 1) / chunk : 0;
}

// chunk (= wave) -> pair, so that a pair-pass wave finds its work with one load
__global__ void
pair_chunk_fill_kernel(const int32_t *chunk_start, int num_pairs, int32_t *chunk_pair, const int32_t *pair_start,
    const uint32_t *pair_key, int num_cameras, int group, int chunk, PairChunkDesc *desc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_pairs) return;
    const int c0 = chunk_start[i], c1 = chunk_start[i + 1];
    const int p0 = pair_start[i], p1 = pair_start[i + 1];
    const uint32_t key = pair_key[i];
    for (int c = c0; c < c1; ++c) {
        chunk_pair[c] = i;
        PairChunkDesc dsc;
        dsc.pi = i; dsc.e0 = p0 + (c - c0) * chunk; dsc.e1 = min(p1, dsc.e0 + chunk); dsc.nchunks = c1 - c0;
        dsc.c1 = (int)((key / ((uint32_t)num_cameras * (uint32_t)group)) * (uint32_t)group + key % (uint32_t)group);
        dsc.c2 = (int)((key / (uint32_t)group) % (uint32_t)num_cameras);
        dsc.first = c0; dsc.pad1 = 0;
        desc[c] = dsc;
    }
}

// ---------------------------------------------------------------------------
// A handful of cameras (the 3-camera adjustments of the incremental reconstruction: BASELINE configs[0] and every
// group of the larger jobs).  The general build sorts the entries by camera pair -- rocPRIM's merge sort for sizes
// like these: two dozen launches, and with the run-length encoding, its scans and two read-backs 0.19 ms of a call
// whose LM loop takes 0.4.  With at most kSmallCams cameras there are at most 36 pairs: a thread per track flags, for
// every pair, whether the track sees both cameras; ONE exclusive sum over the flags laid out pair-major is the
// position of every entry in the sorted lists (pairs in key order, tracks in order inside a pair: the same lists
// to the byte); the per-pair totals come back in one copy and the chunk descriptors are made on the host.
// ---------------------------------------------------------------------------
// entries of a camera pair's list per pair-pass wave (OSFM_BA_PAIR_CHUNK: experiments)
static int pair_chunk_for(int64_t entries)
{
    static const int forced = getenv("OSFM_BA_PAIR_CHUNK") ? atoi(getenv("OSFM_BA_PAIR_CHUNK")) : 0;
    if (forced >= 64) return forced / 64 * 64;
    return entries < kPairChunkTinyLimit ? kPairChunkTiny : entries < kPairChunkSmallLimit ? kPairChunkSmall : kPairChunk;
}

constexpr int kSmallCams = 8;
constexpr int kSmallPairs = kSmallCams * (kSmallCams + 1) / 2;

// the observation of track j in every camera (-1: none), from its at most C observations
__device__ __forceinline__ void small_track_slots(const BaDev &d, int j, int (&slot)[kSmallCams])
{
#pragma unroll
    for (int c = 0; c < kSmallCams; ++c) slot[c] = -1;
    for (int k = d.pt_start[j]; k < d.pt_start[j + 1]; ++k) {
        const int c = d.obs_cam[k];
#pragma unroll
        for (int cc = 0; cc < kSmallCams; ++cc) slot[cc] = (cc == c && d.cam_ldim[c] != 0) ? k : slot[cc];
    }
}

// pair p of the order (c2 major, c1 >= c2 minor) -- the order of the general build's keys for C <= its group of 8
__host__ __device__ __forceinline__ void small_pair_cameras(int p, int C, int &c1, int &c2)
{
    c2 = 0;
    while (p >= C - c2) { p -= C - c2; ++c2; }
    c1 = c2 + p;
}

__global__ __launch_bounds__(256) void
pair_small_flags_kernel(BaDev d, int num_pairs_all, int32_t *flags)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) flags[(size_t)num_pairs_all * d.M] = 0;           // the scan's last entry is the total
    if (j >= d.M) return;
    int slot[kSmallCams];
    small_track_slots(d, j, slot);
    int p = 0;
    for (int c2 = 0; c2 < d.C; ++c2)
        for (int c1 = c2; c1 < d.C; ++c1, ++p) {
            int a = -1, b = -1;
#pragma unroll
            for (int cc = 0; cc < kSmallCams; ++cc) { a = cc == c1 ? slot[cc] : a; b = cc == c2 ? slot[cc] : b; }
            flags[(size_t)p * d.M + j] = (a >= 0 && b >= 0) ? 1 : 0;
        }
}

__global__ void pair_small_totals_kernel(const int32_t *scan, int num_pairs_all, int M, int32_t *starts)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= num_pairs_all) starts[p] = scan[(size_t)p * M];
}

__global__ __launch_bounds__(256) void
pair_small_fill_kernel(BaDev d, int num_pairs_all, const int32_t *flags, const int32_t *scan, uint64_t *entries)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d.M) return;
    int slot[kSmallCams];
    small_track_slots(d, j, slot);
    int p = 0;
    for (int c2 = 0; c2 < d.C; ++c2)
        for (int c1 = c2; c1 < d.C; ++c1, ++p) {
            if (!flags[(size_t)p * d.M + j]) continue;
            int a = -1, b = -1;
#pragma unroll
            for (int cc = 0; cc < kSmallCams; ++cc) { a = cc == c1 ? slot[cc] : a; b = cc == c2 ? slot[cc] : b; }
            entries[scan[(size_t)p * d.M + j]] = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
        }
}

static int pair_lists_build_small(const BaDev &d, int64_t max_entries, PairListsDev *out, hipStream_t s)
{
    const int C = d.C, M = d.M, P = C * (C + 1) / 2;
    const size_t nflag = (size_t)P * M + 1;
    OSFM_RETURN_IF(out->counts.reserve(nflag * 4));
    OSFM_RETURN_IF(out->offsets.reserve(nflag * 4));
    OSFM_RETURN_IF(out->starts.reserve((size_t)(P + 1) * 4 + 16));
    int32_t *flags = out->counts.as<int32_t>(), *scan = out->offsets.as<int32_t>();
    hipLaunchKernelGGL(pair_small_flags_kernel, dim3((M + 255) / 256), dim3(256), 0, s, d, P, flags);
    size_t t1 = 0;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, flags, scan, (int)nflag, s));
    OSFM_RETURN_IF(out->temp.reserve(t1 + 256));
    size_t tb = out->temp.bytes;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(out->temp.ptr, tb, flags, scan, (int)nflag, s));
    hipLaunchKernelGGL(pair_small_totals_kernel, dim3(1), dim3(64), 0, s, scan, P, M, out->starts.as<int32_t>());
    int32_t h_starts[kSmallPairs + 1];
    OSFM_HIP_CHECK(hipMemcpyAsync(h_starts, out->starts.ptr, (size_t)(P + 1) * 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    const int E = h_starts[P];
    if (E > max_entries) { set_error("pair lists: %d entries exceed the bound %lld", E, (long long)max_entries); return OSFM_E_ARG; }
    out->num_entries = E; out->num_entries_all = E;
    if (E == 0) return OSFM_OK;
    OSFM_RETURN_IF(out->entries.reserve((size_t)E * 8));
    hipLaunchKernelGGL(pair_small_fill_kernel, dim3((M + 255) / 256), dim3(256), 0, s, d, P, flags, scan, out->entries.as<uint64_t>());
    // chunk descriptors of the pairs that have entries, on the host
    out->chunk = pair_chunk_for(E);
    std::vector<PairChunkDesc> desc;
    int np = 0;
    for (int p = 0; p < P; ++p) {
        const int p0 = h_starts[p], p1 = h_starts[p + 1];
        if (p1 == p0) continue;
        int c1, c2;
        small_pair_cameras(p, C, c1, c2);
        const int nch = (p1 - p0 + out->chunk - 1) / out->chunk, first = (int)desc.size();
        for (int c = 0; c < nch; ++c) {
            PairChunkDesc dsc;
            dsc.pi = np; dsc.e0 = p0 + c * out->chunk; dsc.e1 = std::min(p1, dsc.e0 + out->chunk); dsc.nchunks = nch;
            dsc.c1 = c1; dsc.c2 = c2; dsc.first = first; dsc.pad1 = 0;
            desc.push_back(dsc);
        }
        ++np;
    }
    out->num_pairs = np;
    out->max_chunks = (int)desc.size();
    OSFM_RETURN_IF(out->chunk_desc.reserve((desc.size() + 4) * sizeof(PairChunkDesc)));
    OSFM_HIP_CHECK(hipMemcpyAsync(out->chunk_desc.ptr, desc.data(), desc.size() * sizeof(PairChunkDesc), hipMemcpyHostToDevice, s));
    OSFM_RETURN_IF(out->chunk_partials.reserve((size_t)out->max_chunks * kPairSums * sizeof(double)));
    OSFM_RETURN_IF(out->pair_ticket.reserve((size_t)(np + 1) * 4));
    OSFM_HIP_CHECK(hipMemsetAsync(out->pair_ticket.ptr, 0, (size_t)(np + 1) * 4, s));
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipStreamSynchronize(s));            // desc leaves scope
    return OSFM_OK;
}

int pair_lists_build(const BaDev &d, bool with_points, int64_t max_entries, PairListsDev *out, hipStream_t s, int dense_policy)
{
    const int M = d.O;          // the lists are generated per observation
    out->num_pairs = 0; out->num_entries = 0; out->dense = false; out->num_entries_all = 0;
    if (M == 0 || d.M == 0) return OSFM_OK;
    const bool no_small = getenv("OSFM_BA_PAIR_LISTS_GENERAL") != nullptr;      // (A/B runs and tests)
    if (with_points && d.C <= kSmallCams && dense_policy != 1 && !no_small && (int64_t)d.C * (d.C + 1) / 2 * d.M < (1ll << 30))
        return pair_lists_build_small(d, max_entries, out, s);
    // per-observation counts, offsets and the hipCUB item counts are 32-bit, and the lists take
    // about 36 bytes per entry up front: long tracks (sum of squared track lengths) are
    // refused here instead of wrapping the offsets
    if (max_entries > (int64_t)0x7fffffff) {
        set_error("ba_solve: %lld observation pairs in the Schur complement lists (sum of squared track "
                  "lengths) exceed the supported 2^31 - 1", (long long)max_entries);
        return OSFM_E_RANGE;
    }
    // Count first, allocate what the count says: the bound (sum of squared track lengths) is twice the number
    // of entries (pairs with cam(a) >= cam(b) only), and the arrays it would size are the gigabytes of this
    // build on a 500-view job -- reallocated whenever the growing problem crossed a size class of the block
    // cache (0.65 s per event: a dozen of them were half of that job's global adjustments).  The runs of the
    // sorted keys are camera pairs: at most C (C + 1) / 2 of them, however many entries there are.
    const int64_t max_runs = std::min<int64_t>(max_entries, (int64_t)d.C * ((int64_t)d.C + 1) / 2) + 1;
    OSFM_RETURN_IF(out->counts.reserve((size_t)(std::max<int64_t>(M, max_runs) + 1) * 4));   // later: chunk counts per pair
    OSFM_RETURN_IF(out->offsets.reserve((size_t)(M + 1) * 4));
    OSFM_RETURN_IF(out->unique.reserve((size_t)max_runs * 4 + 16));
    OSFM_RETURN_IF(out->runs.reserve((size_t)max_runs * 4 + 16));
    OSFM_RETURN_IF(out->starts.reserve((size_t)max_runs * 4 + 16));
    OSFM_RETURN_IF(out->scalars.reserve(64));

    const int blocks = (M + 255) / 256;
    int list_mode = with_points ? 1 : 0;
    hipLaunchKernelGGL(pair_count_kernel, dim3(blocks), dim3(256), 0, s, d, list_mode,
        out->counts.as<int32_t>());
    size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    int32_t *counts = out->counts.as<int32_t>(), *offsets = out->offsets.as<int32_t>();
    int bits = 1;
    static const int group = getenv("OSFM_BA_PAIR_GROUP") ? std::max(1, atoi(getenv("OSFM_BA_PAIR_GROUP"))) : 8;
    out->group = group;
    if (((unsigned long long)d.C + group) * (unsigned long long)d.C >= (1ull << 32)) { set_error("ba_solve: too many cameras for 32-bit pair keys"); return OSFM_E_RANGE; }
    while ((1ull << bits) < ((unsigned long long)d.C + group) * (unsigned long long)d.C) ++bits;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, counts, offsets, M, s));
    OSFM_RETURN_IF(out->temp.reserve(t1 + 256));
    size_t tb = out->temp.bytes;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(out->temp.ptr, tb, counts, offsets, M, s));
    // total = offsets[M-1] + counts[M-1]
    int32_t h_last[2] = { 0, 0 };
    OSFM_HIP_CHECK(hipMemcpyAsync(&h_last[0], offsets + (M - 1), 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipMemcpyAsync(&h_last[1], counts + (M - 1), 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    int E = h_last[0] + h_last[1];
    if (E > max_entries) { set_error("pair lists: %d entries exceed the bound %lld", E, (long long)max_entries); return OSFM_E_ARG; }
    out->num_entries_all = E;
    // Dense visibility: the point part of the Schur complement as a product of two dense matrices (ba_dense.hip), and
    // lists of the diagonal pairs only -- counted again, nothing of the full lists has been built yet
    if (with_points && E > 0 && (dense_policy == 1 || (dense_policy < 0 && schur_dense_wins(d.nc, d.M, E)))) {
        out->dense = true;
        list_mode = 2;
        hipLaunchKernelGGL(pair_count_kernel, dim3(blocks), dim3(256), 0, s, d, list_mode, out->counts.as<int32_t>());
        tb = out->temp.bytes;
        OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(out->temp.ptr, tb, counts, offsets, M, s));
        OSFM_HIP_CHECK(hipMemcpyAsync(&h_last[0], offsets + (M - 1), 4, hipMemcpyDeviceToHost, s));
        OSFM_HIP_CHECK(hipMemcpyAsync(&h_last[1], counts + (M - 1), 4, hipMemcpyDeviceToHost, s));
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
        E = h_last[0] + h_last[1];
    }
    out->num_entries = E;
    if (E == 0) return OSFM_OK;
    OSFM_RETURN_IF(out->keys_in.reserve((size_t)E * 4));
    OSFM_RETURN_IF(out->keys.reserve((size_t)E * 4));
    OSFM_RETURN_IF(out->vals_in.reserve((size_t)E * 8));
    OSFM_RETURN_IF(out->entries.reserve((size_t)E * 8));
    uint32_t *kin = out->keys_in.as<uint32_t>(), *kout = out->keys.as<uint32_t>();
    uint64_t *vin = out->vals_in.as<uint64_t>(), *vout = out->entries.as<uint64_t>();
    OSFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, t2, kin, kout, vin, vout, E, 0, bits, s));
    OSFM_HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(nullptr, t3, kout, out->unique.as<uint32_t>(),
        out->runs.as<int32_t>(), out->scalars.as<int32_t>(), E, s));
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t4, out->runs.as<int32_t>(), out->starts.as<int32_t>(),
        (int)max_runs, s));
    OSFM_RETURN_IF(out->temp.reserve(std::max(std::max(t1, t2), std::max(t3, t4)) + 256));
    hipLaunchKernelGGL(pair_fill_kernel, dim3(blocks), dim3(256), 0, s, d, list_mode, group, offsets, kin, vin);
    tb = out->temp.bytes;
    OSFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(out->temp.ptr, tb, kin, kout, vin, vout, E, 0, bits, s));
    tb = out->temp.bytes;
    OSFM_HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(out->temp.ptr, tb, kout, out->unique.as<uint32_t>(),
        out->runs.as<int32_t>(), out->scalars.as<int32_t>(), E, s));
    int32_t h_runs = 0;
    OSFM_HIP_CHECK(hipMemcpyAsync(&h_runs, out->scalars.ptr, 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    out->num_pairs = h_runs;
    tb = out->temp.bytes;
    // starts[num_pairs] must hold the total: scan over num_pairs + 1 with a trailing zero run
    OSFM_HIP_CHECK(hipMemsetAsync(out->runs.as<int32_t>() + h_runs, 0, 4, s));
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(out->temp.ptr, tb, out->runs.as<int32_t>(),
        out->starts.as<int32_t>(), h_runs + 1, s));
    // chunks of the pair pass: chunk_start = exclusive scan of ceil(run / kPairChunk); the number
    // of waves to launch is bounded without reading anything back
    out->chunk = pair_chunk_for(E);
    out->max_chunks = h_runs + E / out->chunk;
    OSFM_RETURN_IF(out->chunk_start.reserve((size_t)(h_runs + 1) * 4));
    OSFM_RETURN_IF(out->chunk_partials.reserve((size_t)out->max_chunks * kPairSums * sizeof(double)));
    // the run lengths are not needed after this: the chunk counts take their place
    hipLaunchKernelGGL(pair_chunk_count_kernel, dim3((h_runs + 256) / 256), dim3(256), 0, s,
        out->runs.as<int32_t>(), h_runs, out->chunk, out->counts.as<int32_t>());
    tb = out->temp.bytes;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(out->temp.ptr, tb, out->counts.as<int32_t>(),
        out->chunk_start.as<int32_t>(), h_runs + 1, s));
    OSFM_RETURN_IF(out->chunk_pair.reserve((size_t)(out->max_chunks + 1) * 4));
    // the launch has max_chunks waves (an upper bound known without a read-back): descriptors past the real
    // chunks stay zero, nchunks == 0 = nothing to do
    OSFM_RETURN_IF(out->pair_ticket.reserve((size_t)(h_runs + 1) * 4));
    OSFM_HIP_CHECK(hipMemsetAsync(out->pair_ticket.ptr, 0, (size_t)(h_runs + 1) * 4, s));
    OSFM_RETURN_IF(out->chunk_desc.reserve((size_t)(out->max_chunks + 4) * sizeof(PairChunkDesc)));
    OSFM_HIP_CHECK(hipMemsetAsync(out->chunk_desc.ptr, 0, (size_t)(out->max_chunks + 4) * sizeof(PairChunkDesc), s));
    hipLaunchKernelGGL(pair_chunk_fill_kernel, dim3((h_runs + 255) / 256), dim3(256), 0, s,
        out->chunk_start.as<int32_t>(), h_runs, out->chunk_pair.as<int32_t>(), out->starts.as<int32_t>(),
        out->unique.as<uint32_t>(), d.C, group, out->chunk, out->chunk_desc.as<PairChunkDesc>());
    // (nothing more to read back: the pairs with several chunks finish themselves, by ticket, inside the pair pass)
    OSFM_HIP_CHECK(hipGetLastError());
    return OSFM_OK;
}

}  // namespace osfm
