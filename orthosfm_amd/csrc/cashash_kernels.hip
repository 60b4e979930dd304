// Cascade hashing on the device: the application's default,
// approximate matcher, reproduced decision for decision.
//
// What has to agree with the reference bit for bit, and how:
//   * descriptor average: float sums in descriptor order (order dependent) --
//     one thread per dimension walks the descriptors of the views in order;
//   * hash bits: signs of float dot products accumulated in element order with
//     separate multiply and add (no FMA contraction, as the reference's x86-64
//     build) -- one thread per (descriptor, projection vector);
//   * candidate ranking: Hamming distance, then order of first appearance over
//     the bucket groups; "already seen" is decided from the bucket ids of the
//     earlier groups instead of a per-query bitmap;
//   * the final nearest-neighbour search over <= 10 candidates repeats the
//     16-bit lane arithmetic and T-typed state of NearestNeighbor<T>::find.
// This first version favours exactness and simplicity over speed (one thread per
// query, candidates gathered from L2).
#include "cashash_kernels.h"

namespace osfm {

// One thread per dimension; the descriptors pass through LDS in chunks that all
// threads fetch together (many loads in flight), then every thread adds its
// dimension of the chunk's rows in order: the sum order is the reference's, the
// memory latency is paid once per chunk instead of once per descriptor.
constexpr int kAccChunk = 64;

constexpr int kAccStage = 4 * kAccChunk;       // rows fetched per round (32 KB of LDS for SIFT)

__global__ __launch_bounds__(128, 1) void      // one workgroup: all registers are its own
cashash_accumulate_kernel(const int8_t *__restrict__ desc, int n, int dim, int bias, float div,
    float *__restrict__ sum)
{
    __shared__ __attribute__((aligned(16))) int8_t rows[kAccStage * 128];
    __shared__ float quot[256];          // (float)(byte + bias) / div for every stored byte value
    const int k = threadIdx.x;
    // A stored byte has 256 values, so the reference's division (IEEE, not a multiply by
    // a reciprocal) is done 256 times here instead of once per element: 14 instructions
    // per element were what this one-workgroup kernel spent its time on.
    for (int v = k; v < 256; v += 128) quot[v] = (float)((int)(int8_t)v + bias) / div;
    float s = k < dim ? sum[k] : 0.0f;
    constexpr int kPer = kAccStage * 128 / 16 / 128;        // 16-byte pieces per thread and round (16)
    // One workgroup walks the whole view, so what it waits for is memory latency: a round
    // of 256 rows is fetched (16 loads per thread in flight) while the round before is summed.
    // (written out rather than as lambdas taking the array: a reference to it puts it in
    // scratch memory and every load is then waited for at once)
#define OSFM_ACC_FETCH(BASE)                                                                       \
    {                                                                                              \
        const int vecs_ = min(kAccStage, n - (BASE)) * dim / 16;                                   \
        const int4 *src_ = reinterpret_cast<const int4 *>(desc + (size_t)(BASE) * dim);            \
        _Pragma("unroll") for (int u = 0; u < kPer; ++u) v[u] = src_[min(u * 128 + k, max(vecs_ - 1, 0))]; \
    }
    // every piece is stored, also the clamped repeats behind the round's end (rows that are
    // never read): a store under a per-piece condition becomes a branch per piece with v[]
    // parked in scratch memory
#define OSFM_ACC_DEPOSIT(BASE)                                                                     \
    {                                                                                              \
        _Pragma("unroll") for (int u = 0; u < kPer; ++u) reinterpret_cast<int4 *>(rows)[u * 128 + k] = v[u]; \
    }
    int4 v[kPer];                        // the only copy in registers: no ping-pong arrays
    if (n <= 0) return;
    OSFM_ACC_FETCH(0);
    OSFM_ACC_DEPOSIT(0);
    __syncthreads();
    for (int base = 0; base < n; base += kAccStage) {
        const int cnt = min(kAccStage, n - base);
        const bool more = base + kAccStage < n;
        const int nb = more ? base + kAccStage : base;       // unconditional (a conditional definition of v[] sends it to scratch)
        OSFM_ACC_FETCH(nb);                                  // in flight while this round is summed
        if (k < dim) {
            for (int j0 = 0; j0 < cnt; j0 += kAccChunk) {
                if (cnt - j0 >= kAccChunk) {
                    // quotients looked up first (independent), then the additions in the
                    // reference's order: only the additions form a chain
                    float q[kAccChunk];
#pragma unroll
                    for (int j = 0; j < kAccChunk; ++j) q[j] = quot[(uint8_t)rows[(j0 + j) * dim + k]];   // sift_descr[j][k] / 255.0f
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < kAccChunk; ++j) s += q[j];
                } else {
                    for (int j = j0; j < cnt; ++j) s += quot[(uint8_t)rows[j * dim + k]];
                }
            }
        }
        __syncthreads();                                     // this round has been consumed
        OSFM_ACC_DEPOSIT(nb);                                // the last round re-deposits itself: harmless
        __syncthreads();
    }
#undef OSFM_ACC_FETCH
#undef OSFM_ACC_DEPOSIT
    if (k < dim) sum[k] = s;
}

__global__ void
cashash_average_kernel(const float *__restrict__ sum, int dim, float count, float *__restrict__ avg)
{
    const int k = threadIdx.x;
    if (k < dim) avg[k] = sum[k] / count;
}

// Block = kDescPerBlock descriptors x (dim + 48) projection vectors (one thread
// each, rounded up to whole waves).  Zero-mean descriptors in LDS, projection
// rows read coalesced from the transposed matrix.
constexpr int kDescPerBlock = 8;

template <int DIM>
__global__ __launch_bounds__(256) void
cashash_hash_kernel(const int8_t *__restrict__ desc, int n, int bias, float div,
    const float *__restrict__ avg, const float *__restrict__ projT, uint64_t *__restrict__ hashes,
    uint8_t *__restrict__ bucket_ids)
{
    constexpr int NP = DIM + kCasSecBits;             // projection vectors
    __shared__ float zm[kDescPerBlock][DIM];
    const int tid = threadIdx.x;
    const int d0 = blockIdx.x * kDescPerBlock;
    for (int e = tid; e < kDescPerBlock * DIM; e += blockDim.x) {
        const int d = e / DIM, k = e % DIM;
        const int i = d0 + d;
        zm[d][k] = i < n ? (float)((int)desc[(size_t)i * DIM + k] + bias) / div - avg[k] : 0.0f;
    }
    __syncthreads();
    const int p = tid;                                // projection vector of this thread
    float sum[kDescPerBlock];
#pragma unroll
    for (int d = 0; d < kDescPerBlock; ++d) sum[d] = 0.0f;
    if (p < NP) {
        for (int e = 0; e < DIM; ++e) {
            const float w = projT[(size_t)e * NP + p];
#pragma unroll
            for (int d = 0; d < kDescPerBlock; ++d) sum[d] = sum[d] + zm[d][e] * w;
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 0; d < kDescPerBlock; ++d) {
        const unsigned long long bal = __ballot(p < NP && sum[d] > 0.0f);
        const int i = d0 + d;
        if (i >= n || lane != 0) continue;
        if (wave < DIM / 64) {
            // comp_hash = (comp_hash << 1) | bit for k ascending: vector 64 w + l -> bit 63 - l
            hashes[(size_t)i * (DIM / 64) + wave] = __brevll(bal);
        } else if (wave == DIM / 64) {
            // 48 secondary bits: group g = l / 8, bit b = l % 8 -> bit 7 - b of the id
            for (int g = 0; g < kCasGroups; ++g) {
                const unsigned byte = (unsigned)((bal >> (8 * g)) & 0xffu);
                bucket_ids[(size_t)g * n + i] = (uint8_t)(__brev(byte) >> 24);
            }
        }
    }
}

// One block per bucket group; thread b collects the features of bucket b in
// ascending order (two passes over the ids: count, then fill).  The ids pass
// through LDS in chunks fetched by all threads, so that the per-thread scans
// read LDS broadcasts instead of waiting on one global load per id.
// build_buckets (cascade_hashing.cc:187-209): per group the CSR of feature ids by
// bucket, ascending inside a bucket -- a stable counting sort.  One wave per group:
// histogram with LDS atomics, then the ids go through 64 at a time; lanes holding
// the same bucket find each other with eight ballots (one per id bit), their rank
// among equals orders them, the last of them advances the bucket's cursor.
__global__ __launch_bounds__(64) void
cashash_buckets_kernel(const uint8_t *__restrict__ bucket_ids, int n, int32_t *__restrict__ start,
    int32_t *__restrict__ items)
{
    __shared__ int32_t cnt[kCasBuckets + 1];
    __shared__ int32_t cur[kCasBuckets];
    const int g = blockIdx.x, lane = threadIdx.x;
    const uint8_t *ids = bucket_ids + (size_t)g * n;
    int32_t *out = items + (size_t)g * n;
    for (int b = lane; b <= kCasBuckets; b += 64) cnt[b] = 0;
    __syncthreads();
    for (int i = lane; i < n; i += 64) atomicAdd(&cnt[ids[i] + 1], 1);
    __syncthreads();
    if (lane == 0) for (int k = 0; k < kCasBuckets; ++k) cnt[k + 1] += cnt[k];
    __syncthreads();
    for (int b = lane; b <= kCasBuckets; b += 64) {
        start[(size_t)g * (kCasBuckets + 1) + b] = cnt[b];
        if (b < kCasBuckets) cur[b] = cnt[b];
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1;
    int id_next = lane < n ? ids[lane] : 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool act = i < n;
        const int id = id_next;
        if (base + 64 + lane < n) id_next = ids[base + 64 + lane];
        // lanes with the same id
        unsigned long long same = __ballot(act);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((id >> bit) & 1);
            same &= ((id >> bit) & 1) ? m : ~m;
        }
        if (act) {
            const int rank = __popcll(same & below);
            out[cur[id] + rank] = i;
        }
        __syncthreads();
        // the highest lane of every set moves the cursor past the set
        if (act && (same >> lane) == 1ull) cur[id] += __popcll(same);
        __syncthreads();
    }
}

__global__ void
cashash_pack_kernel(const uint64_t *__restrict__ hashes, const uint8_t *__restrict__ bucket_ids, int n,
    int words, CasRecord *__restrict__ rec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    CasRecord r;
    r.h[0] = hashes[(size_t)i * words];
    r.h[1] = words > 1 ? hashes[(size_t)i * words + 1] : 0ull;
    uint64_t b = 0;
    for (int g = 0; g < kCasGroups; ++g) b |= (uint64_t)bucket_ids[(size_t)g * n + i] << (8 * g);
    r.buckets = b;
    r.pad = 0;
    rec[i] = r;
}

// ---------------------------------------------------------------------------
// CascadeHashing::oneway_match (cascade_hashing.h:328-412) in two steps.
//
// Scan (one launch per bucket group g): a workgroup takes one bucket; the
// candidates of that bucket (set 2) go through LDS once and are scored by all
// the queries of the same bucket (set 1), one query per thread -- the records
// are read once per bucket instead of once per (query, candidate).  A query
// keeps its ten best candidates so far as ten sorted keys
//     Hamming distance << 20 | group << 17 | position in the bucket list,
// which order exactly like (distance, order of first appearance) because
// groups are visited in order and positions ascend (collect_features_from_buckets
// h:414-444 appends in that order; duplicates are dropped at their later
// appearance, decided from the packed bucket ids of the earlier groups).  The
// keys live in global memory between the launches (the thread <-> query mapping
// changes with the group).
//
// Finish (one thread per query): candidate ids back from the keys, the 6..10
// rule (h:446-468), NearestNeighbor<T>::find over the survivors and the ratio
// test.
// ---------------------------------------------------------------------------
constexpr int kCasPosBits = 17, kCasGroupShift = kCasPosBits, kCasDistShift = 20;
constexpr int kCasKeyNone = 0x7fffffff;

__device__ __forceinline__ int cas_med3(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int DIM, int G>
__global__ __launch_bounds__(128) void
cashash_scan_kernel(const MatchProblem *__restrict__ problems, int32_t *__restrict__ state, int buckets_per_block)
{
    constexpr int g = G;
    const MatchProblem &pd = problems[blockIdx.y];
    const int dir = blockIdx.z;
    const int nq = dir == 0 ? pd.n1 : pd.n2;
    const int nc = dir == 0 ? pd.n2 : pd.n1;
    if (nq == 0 || nc == 0) return;
    const int s1 = dir, s2 = dir ^ 1;
    const int32_t *st1 = pd.cas_start[s1] + (size_t)g * (kCasBuckets + 1);
    const int32_t *it1 = pd.cas_items[s1] + (size_t)g * nq;
    const int32_t *st2 = pd.cas_start[s2] + (size_t)g * (kCasBuckets + 1);
    const int32_t *it2 = pd.cas_items[s2] + (size_t)g * nc;
    const CasRecord *rec1 = static_cast<const CasRecord *>(pd.cas_rec[s1]);
    const CasRecord *rec2 = static_cast<const CasRecord *>(pd.cas_rec[s2]);
    int32_t *st = state + pd.cas_state_off[dir] * kCasMaxCand;
    __shared__ CasRecord crec[128];

    // Several buckets per workgroup, one after the other: a bucket is a few microseconds
    // of work, and a launch of 256 x pairs x 2 workgroups that small is paced by the
    // dispatcher (1225 pairs: 67 ms with one bucket per workgroup, 46 ms with 64)
    for (int beta = blockIdx.x * buckets_per_block; beta < (int)(blockIdx.x + 1) * buckets_per_block; ++beta) {
    const int qb = st1[beta], qe = st1[beta + 1];
    const int cb = st2[beta], ce = st2[beta + 1];
    if (qb == qe) continue;
    for (int q0 = qb; q0 < qe; q0 += 128) {
        const int qi = q0 + (int)threadIdx.x;
        const bool act = qi < qe;
        const int q = act ? it1[qi] : 0;
        const CasRecord me = rec1[q];
        int key[kCasMaxCand];
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) key[j] = (act && g > 0) ? st[(size_t)q * kCasMaxCand + j] : kCasKeyNone;
        for (int c0 = cb; c0 < ce; c0 += 128) {
            __syncthreads();
            if (c0 + (int)threadIdx.x < ce) crec[threadIdx.x] = rec2[it2[c0 + threadIdx.x]];
            __syncthreads();
            const int cnt = min(128, ce - c0);
            if (!act) continue;
            // Branch-free and two candidates per round: a duplicate or a candidate that
            // does not make the list goes through the insertion network as "infinity" and
            // leaves the keys as they are.  (With 64 queries per wave some lane inserts at
            // almost every candidate, so the network ran anyway -- behind two `continue`s
            // whose exec-mask handling was half of the loop's instructions.)
            auto cand_key = [&](uint64_t h0, uint64_t h1, uint64_t bk, int pos) {
                bool dup = false;
                if (G > 0) {
                    // seen in an earlier group iff one of the first G bucket ids equals the
                    // query's: a zero byte in the xor (bytes >= G forced non-zero)
                    const uint32_t x = ((uint32_t)bk ^ (uint32_t)me.buckets) | (G < 4 ? ~0u << (8 * (G & 3)) : 0u);
                    dup = ((x - 0x01010101u) & ~x & 0x80808080u) != 0;
                }
                if (G == 5) dup |= (((uint32_t)(bk >> 32) ^ (uint32_t)(me.buckets >> 32)) & 0xffu) == 0;
                const int hd = __popcll(me.h[0] ^ h0) + (DIM > 64 ? __popcll(me.h[1] ^ h1) : 0);
                const int k = (hd << kCasDistShift) | (g << kCasGroupShift) | pos;
                return dup ? kCasKeyNone : k;
            };
            for (int j = 0; j < cnt; j += 2) {
                const int j1 = min(j + 1, cnt - 1);
                const uint64_t a0 = crec[j].h[0], a1 = crec[j].h[1], ab = crec[j].buckets;
                const uint64_t b0 = crec[j1].h[0], b1 = crec[j1].h[1], bb = crec[j1].buckets;
                int k0 = cand_key(a0, a1, ab, c0 - cb + j);
                int k1 = j + 1 < cnt ? cand_key(b0, b1, bb, c0 - cb + j + 1) : kCasKeyNone;
                // sorted insertion, one op per slot: slot e of the new list is the median of
                // (old slot e-1, old slot e, x) -- the old neighbour if x lies below it, x if
                // it falls between the two, the old value otherwise
#pragma unroll
                for (int e = kCasMaxCand - 1; e >= 1; --e) key[e] = cas_med3(key[e - 1], key[e], k0);
                key[0] = min(key[0], k0);
#pragma unroll
                for (int e = kCasMaxCand - 1; e >= 1; --e) key[e] = cas_med3(key[e - 1], key[e], k1);
                key[0] = min(key[0], k1);
            }
        }
        if (act) {
#pragma unroll
            for (int j = 0; j < kCasMaxCand; ++j) st[(size_t)q * kCasMaxCand + j] = key[j];
        }
    }
    }
}

// DPP moves inside a lane group (no LDS traffic)
template <int CTRL>
__device__ __forceinline__ int cas_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int LPC>
__device__ __forceinline__ int group_sum(int v)      // sum over the LPC lanes of a group, in every lane
{
    v += cas_dpp<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += cas_dpp<0x4E>(v);                 // quad_perm [2,3,0,1]
    if (LPC == 8) v += cas_dpp<0x141>(v);  // row_half_mirror
    return v;
}

// The final stage of CascadeHashing::oneway_match for one query per thread:
// candidate ids from the ranked keys, the 6..10 rule, then NearestNeighbor<T>::find
// over the chosen candidates.  The nearest-neighbour part runs with DIM/16 lanes per
// query (8 queries of a wave at a time for SIFT): a candidate row is one coalesced
// 128-byte read per lane group instead of eight 16-byte reads scattered over 64
// rows per wave instruction, and the inner product is four v_dot4 per lane plus
// three DPP adds.
//   Exactness: the reference sums eight 16-bit lanes that wrap (nearest_neighbor.cc:
//   60-129).  SIFT products are non-negative, so no lane can wrap unless the whole
//   inner product exceeds 65535: below that the dot-product value IS the wrapped
//   sum.  Otherwise -- and always for SURF, whose signed lanes can wrap at any
//   total -- the eight lane sums are formed as such (mod 2^16 is a ring
//   homomorphism, so they may be reduced across lanes before they are truncated).
template <int DIM, bool SIGNED>
__global__ __launch_bounds__(128) void
cashash_finish_kernel(const MatchProblem *__restrict__ problems, const int32_t *__restrict__ state,
    LoweTable tab, int blocks_per_dir, int total_blocks)
{
    constexpr int LPC = DIM / 16, NG = 64 / LPC;
    // every XCD works through a contiguous run of problems (candidate rows stay in one L2)
    const int lin = xcd_remap(blockIdx.x, total_blocks);
    const int problem = lin / (2 * blocks_per_dir), within = lin - problem * 2 * blocks_per_dir;
    const MatchProblem &pd = problems[problem];
    const int dir = within / blocks_per_dir;
    const int q = (within - dir * blocks_per_dir) * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int nq = dir == 0 ? pd.n1 : pd.n2;
    const int nc = dir == 0 ? pd.n2 : pd.n1;
    if ((within - dir * blocks_per_dir) * (int)blockDim.x >= nq) return;     // whole block out of range
    int32_t *out = dir == 0 ? pd.m12 : pd.m21;
    if (nc == 0) {                          // uniform: the other view has nothing of this type
        if (q < nq) out[q] = -1;
        return;
    }
    const bool active = q < nq;
    const int s1 = dir, s2 = dir ^ 1;
    const int8_t *Q = dir == 0 ? pd.A : pd.B;
    const int8_t *Cm = dir == 0 ? pd.B : pd.A;
    const int32_t *corrQ = dir == 0 ? pd.corrA : pd.corrB;
    const int32_t *corrC = dir == 0 ? pd.corrB : pd.corrA;

    int key[kCasMaxCand], cid[kCasMaxCand];
    int nt = 0;
    {
        const int qs = active ? q : 0;
        const uint64_t my_buckets = active ? static_cast<const CasRecord *>(pd.cas_rec[s1])[qs].buckets : 0;
        const int32_t *st = state + (pd.cas_state_off[dir] + qs) * kCasMaxCand;
        // all loads of a stage issued together: keys, then bucket starts, then ids
        // (invalid entries read index 0 and are discarded)
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) key[j] = active ? st[j] : kCasKeyNone;
        int base[kCasMaxCand];
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) {
            const bool ok = key[j] != kCasKeyNone;
            const int g = ok ? (key[j] >> kCasGroupShift) & 7 : 0;
            const int myb = (int)((my_buckets >> (8 * g)) & 0xffu);
            base[j] = pd.cas_start[s2][(size_t)g * (kCasBuckets + 1) + myb];
        }
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) {
            const bool ok = key[j] != kCasKeyNone;
            const int g = ok ? (key[j] >> kCasGroupShift) & 7 : 0;
            const int pos = ok ? key[j] & ((1 << kCasPosBits) - 1) : 0;
            const int v = pd.cas_items[s2][(size_t)g * nc + (ok ? base[j] + pos : 0)];
            cid[j] = ok ? v : -1;
        }
        // collect_top_ranked_candidates (h:446-468): whole distance levels until at
        // least 6 are in, never more than 10
        bool open = true;
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) {
            if (cid[j] < 0) open = false;
            if (j >= kCasMinCand && (key[j] >> kCasDistShift) > (key[j - 1] >> kCasDistShift)) open = false;
            if (open) nt = j + 1;
        }
    }

    // NearestNeighbor<T>::find over the candidates in that order, one lane group per query
    const int g = lane / LPC, c = lane % LPC;
    int my_best = 0, my_second = 0, my_i1 = 0;
    for (int round = 0; round < LPC; ++round) {
        const int src = (round * NG + g) * 4;                    // byte address of the owner lane
        const int gq = __builtin_amdgcn_ds_bpermute(src, active ? q : 0);
        const int gnt = __builtin_amdgcn_ds_bpermute(src, nt);
        int gcid[kCasMaxCand];
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) gcid[j] = max(__builtin_amdgcn_ds_bpermute(src, cid[j]), 0);
        // candidates behind the cut re-read the first row (same cache line, no extra traffic)
#pragma unroll
        for (int j = 1; j < kCasMaxCand; ++j) gcid[j] = j < gnt ? gcid[j] : gcid[0];
        if (__ballot(gnt > 0) == 0) continue;                    // uniform: nothing to do in this round
        const int4 qv = *reinterpret_cast<const int4 *>(Q + (size_t)gq * DIM + c * 16);
        const int cq = corrQ[gq];
        int4 cv[kCasMaxCand];
        int cc[kCasMaxCand];
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) {
            cv[j] = *reinterpret_cast<const int4 *>(Cm + (size_t)gcid[j] * DIM + c * 16);
            cc[j] = corrC[gcid[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
        int best = 0, second = 0, i1 = 0;
#pragma unroll
        for (int j = 0; j < kCasMaxCand; ++j) {
            int ip;
            bool wrapped = SIGNED;
            if (!SIGNED) {
                int acc = __builtin_amdgcn_sdot4(qv.x, cv[j].x, 0, false);
                acc = __builtin_amdgcn_sdot4(qv.y, cv[j].y, acc, false);
                acc = __builtin_amdgcn_sdot4(qv.z, cv[j].z, acc, false);
                acc = __builtin_amdgcn_sdot4(qv.w, cv[j].w, acc, false);
                ip = group_sum<LPC>(acc) + cq + cc[j];
                wrapped = ip > 65535 && j < gnt;
            }
            if (__ballot(wrapped)) {
                // the eight 16-bit lanes as such: element 16 c + 4 w + b is SSE lane (4 w + b) % 8
                const int qa[4] = {qv.x, qv.y, qv.z, qv.w};
                const int ca[4] = {cv[j].x, cv[j].y, cv[j].z, cv[j].w};
                int part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int qb = (int)(int8_t)(qa[w] >> (8 * b)), cb = (int)(int8_t)(ca[w] >> (8 * b));
                        part[(4 * w + b) & 7] += (SIGNED ? qb : qb + 128) * (SIGNED ? cb : cb + 128);
                    }
                int ipw = 0;
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const int t = group_sum<LPC>(part[l]);
                    ipw += SIGNED ? (int)(short)(t & 0xffff) : (t & 0xffff);
                }
                if (SIGNED) ip = ipw; else ip = wrapped ? ipw : ip;
            }
            if (j < gnt && ip >= second) {
                if (ip >= best) {
                    second = best;
                    best = SIGNED ? (int)(short)ip : (int)(unsigned short)ip;
                    i1 = j;
                } else {
                    second = SIGNED ? (int)(short)ip : (int)(unsigned short)ip;
                }
            }
        }
        // back to the owner lane: lane L was served in round L / NG by group L % NG
        const int from = ((lane % NG) * LPC) * 4;
        const int rb = __builtin_amdgcn_ds_bpermute(from, best);
        const int rs = __builtin_amdgcn_ds_bpermute(from, second);
        const int ri = __builtin_amdgcn_ds_bpermute(from, i1);
        if (lane / NG == round) { my_best = rb; my_second = rs; my_i1 = ri; }
    }
    if (q >= nq) return;
    int d1, d2;
    if (SIGNED) {
        const int b = min(16129, max(0, my_best)), s = min(16129, max(0, my_second));
        d1 = (int)(short)(32258 - 2 * b); d2 = (int)(short)(32258 - 2 * s);
    } else {
        const int b = min(65025, my_best), s = min(65025, my_second);
        d1 = min(32767, 65025 - b) * 2; d2 = min(32767, 65025 - s) * 2;
    }
    // my_i1 indexes the candidate list; the cid[] array is indexed dynamically only here
    int res = -1;
#pragma unroll
    for (int j = 0; j < kCasMaxCand; ++j) if (j == my_i1 && j < nt) res = cid[j];
    if (d1 > tab.max_d1) res = -1;
    else if (d1 >= tab.reject_from[d2 >> 1]) res = -1;
    out[q] = res;
}

void launch_cashash_accumulate(const int8_t *desc, int n, int dim, int bias, float div, float *sum,
    hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(cashash_accumulate_kernel, dim3(1), dim3(128), 0, s, desc, n, dim, bias, div, sum);
}

void launch_cashash_average(const float *sum, int dim, int64_t count, float *avg, hipStream_t s)
{
    hipLaunchKernelGGL(cashash_average_kernel, dim3(1), dim3(128), 0, s, sum, dim, (float)count, avg);
}

void launch_cashash_hash(const int8_t *desc, int n, int dim, int bias, float div, const float *avg,
    const float *projT, uint64_t *hashes, uint8_t *bucket_ids, hipStream_t s)
{
    if (n <= 0) return;
    const dim3 grid((n + kDescPerBlock - 1) / kDescPerBlock);
    if (dim == 128)
        hipLaunchKernelGGL((cashash_hash_kernel<128>), grid, dim3(192), 0, s, desc, n, bias, div, avg, projT, hashes, bucket_ids);
    else
        hipLaunchKernelGGL((cashash_hash_kernel<64>), grid, dim3(128), 0, s, desc, n, bias, div, avg, projT, hashes, bucket_ids);
}

void launch_cashash_pack(const uint64_t *hashes, const uint8_t *bucket_ids, int n, int words,
    CasRecord *rec, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(cashash_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, hashes, bucket_ids, n, words, rec);
}

void launch_cashash_buckets(const uint8_t *bucket_ids, int n, int32_t *start, int32_t *items, hipStream_t s)
{
    hipLaunchKernelGGL(cashash_buckets_kernel, dim3(kCasGroups), dim3(64), 0, s, bucket_ids, n, start, items);
}

void launch_cashash_match(int dim, const MatchProblem *d_problems, int num_problems, int max_n,
    int32_t *state, LoweTable tab, hipStream_t s)
{
    if (num_problems <= 0 || max_n <= 0) return;
    // buckets per workgroup: as many as leave about 8k workgroups in the launch (a power
    // of two between 1 and 64; a single pair keeps one bucket per workgroup)
    int bpb = 1;
    while (bpb < 64 && (int64_t)(kCasBuckets / (2 * bpb)) * num_problems * 2 >= 8192) bpb *= 2;
    const dim3 sgrid(kCasBuckets / bpb, num_problems, 2);
#define OSFM_CAS_SCAN(G) \
    if (dim == 128) hipLaunchKernelGGL((cashash_scan_kernel<128, G>), sgrid, dim3(128), 0, s, d_problems, state, bpb); \
    else hipLaunchKernelGGL((cashash_scan_kernel<64, G>), sgrid, dim3(128), 0, s, d_problems, state, bpb)
    static_assert(kCasGroups == 6, "one instantiation per bucket group");
    OSFM_CAS_SCAN(0); OSFM_CAS_SCAN(1); OSFM_CAS_SCAN(2); OSFM_CAS_SCAN(3); OSFM_CAS_SCAN(4); OSFM_CAS_SCAN(5);
#undef OSFM_CAS_SCAN
    const int blocks_per_dir = (max_n + 127) / 128;
    const int64_t total = (int64_t)blocks_per_dir * 2 * num_problems;
    if (total <= 0 || total > INT_MAX) return;
    const dim3 grid((unsigned)total);
    if (dim == 128)
        hipLaunchKernelGGL((cashash_finish_kernel<128, false>), grid, dim3(128), 0, s, d_problems, state, tab,
            blocks_per_dir, (int)total);
    else
        hipLaunchKernelGGL((cashash_finish_kernel<64, true>), grid, dim3(128), 0, s, d_problems, state, tab,
            blocks_per_dir, (int)total);
}

}  // namespace osfm
