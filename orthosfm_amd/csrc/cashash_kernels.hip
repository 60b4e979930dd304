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

__global__ __launch_bounds__(128) void
cashash_accumulate_kernel(const int8_t *__restrict__ desc, int n, int dim, int bias, float div,
    float *__restrict__ sum)
{
    __shared__ int8_t rows[kAccChunk * 128];
    const int k = threadIdx.x;
    float s = k < dim ? sum[k] : 0.0f;
    for (int base = 0; base < n; base += kAccChunk) {
        const int cnt = min(kAccChunk, n - base);
        const int bytes = cnt * dim;
        const int4 *src = reinterpret_cast<const int4 *>(desc + (size_t)base * dim);
        for (int e = threadIdx.x; e * 16 < bytes; e += blockDim.x)
            reinterpret_cast<int4 *>(rows)[e] = src[e];
        __syncthreads();
        if (k < dim)
            for (int j = 0; j < cnt; ++j)
                s += (float)((int)rows[j * dim + k] + bias) / div;      // sift_descr[j][k] / 255.0f
        __syncthreads();
    }
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
// ascending order (two passes over the ids: count, then fill).
__global__ __launch_bounds__(kCasBuckets) void
cashash_buckets_kernel(const uint8_t *__restrict__ bucket_ids, int n, int32_t *__restrict__ start,
    int32_t *__restrict__ items)
{
    __shared__ int32_t cnt[kCasBuckets + 1];
    const int g = blockIdx.x, b = threadIdx.x;
    const uint8_t *ids = bucket_ids + (size_t)g * n;
    int c = 0;
    for (int i = 0; i < n; ++i) c += ids[i] == b;
    cnt[b + 1] = c;
    if (b == 0) cnt[0] = 0;
    __syncthreads();
    if (b == 0) for (int k = 0; k < kCasBuckets; ++k) cnt[k + 1] += cnt[k];
    __syncthreads();
    start[(size_t)g * (kCasBuckets + 1) + b] = cnt[b];
    if (b == kCasBuckets - 1) start[(size_t)g * (kCasBuckets + 1) + kCasBuckets] = cnt[kCasBuckets];
    int pos = cnt[b];
    int32_t *out = items + (size_t)g * n;
    for (int i = 0; i < n; ++i)
        if (ids[i] == b) out[pos++] = i;
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

// CascadeHashing::oneway_match (cascade_hashing.h:328-412) for query q of set
// `dir` against the other set; one thread per query.
template <int DIM, bool SIGNED>
__global__ __launch_bounds__(128) void
cashash_match_kernel(const MatchProblem *__restrict__ problems, LoweTable tab)
{
    const MatchProblem &pd = problems[blockIdx.y];
    const int dir = blockIdx.z;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = dir == 0 ? pd.n1 : pd.n2;
    const int nc = dir == 0 ? pd.n2 : pd.n1;
    if (q >= nq) return;
    int32_t *out = dir == 0 ? pd.m12 : pd.m21;
    if (nc == 0) { out[q] = -1; return; }
    const int s1 = dir, s2 = dir ^ 1;                    // hash data sets: query side, candidate side
    const int8_t *Q = dir == 0 ? pd.A : pd.B;
    const int8_t *Cm = dir == 0 ? pd.B : pd.A;
    const CasRecord *rec2 = static_cast<const CasRecord *>(pd.cas_rec[s2]);
    const CasRecord me = static_cast<const CasRecord *>(pd.cas_rec[s1])[q];

    // the ten best (Hamming distance, order of first appearance), ascending
    int key[kCasMaxCand], cid[kCasMaxCand];
#pragma unroll
    for (int j = 0; j < kCasMaxCand; ++j) { key[j] = 0x7fffffff; cid[j] = -1; }
    int order = 0;
#pragma unroll 1
    for (int g = 0; g < kCasGroups; ++g) {
        const int32_t *st = pd.cas_start[s2] + (size_t)g * (kCasBuckets + 1);
        const int32_t *it = pd.cas_items[s2] + (size_t)g * nc;
        const int myb = (int)((me.buckets >> (8 * g)) & 0xffu);
        const int pb = st[myb], pe = st[myb + 1];
        // bytes of the earlier groups; the others are forced non-zero below
        const uint64_t later = g == 0 ? ~0ull : ~0ull << (8 * g);
        for (int p = pb; p < pe; ++p) {
            const int c = it[p];
            const CasRecord r = rec2[c];
            // seen before iff it shares the query's bucket in an earlier group
            // (data_index_used, cascade_hashing.h:432-433,441): a zero byte among
            // the first g bytes of the xor of the packed bucket ids
            const uint64_t x = (r.buckets ^ me.buckets) | later;
            if ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) continue;
            const int hd = __popcll(me.h[0] ^ r.h[0]) + (DIM > 64 ? __popcll(me.h[1] ^ r.h[1]) : 0);
            int k = (hd << 20) | order;
            ++order;
            if (k >= key[kCasMaxCand - 1]) continue;
            int ci = c;
            // insertion into the sorted list (compare-exchange from the front)
#pragma unroll
            for (int j = 0; j < kCasMaxCand; ++j) {
                const bool sw = k < key[j];
                const int tk = key[j], tc = cid[j];
                key[j] = sw ? k : tk; cid[j] = sw ? ci : tc;
                k = sw ? tk : k; ci = sw ? tc : ci;
            }
        }
    }
    // collect_top_ranked_candidates (h:446-468): whole distance levels until at
    // least 6 are in, never more than 10
    int nt = 0;
#pragma unroll
    for (int j = 0; j < kCasMaxCand; ++j) {
        if (cid[j] < 0) break;
        if (nt >= kCasMinCand && (key[j] >> 20) > (key[nt - 1] >> 20)) break;
        nt = j + 1;
    }
    // NearestNeighbor<T>::find over the candidates in that order
    // (nearest_neighbor.cc:60-129,214-268): 8 lanes of 16-bit wrap-around sums,
    // state held in T
    // query descriptor once into registers, 16 bytes at a time
    const int4 *qrow = reinterpret_cast<const int4 *>(Q + (size_t)q * DIM);
    int4 qv[DIM / 16];
#pragma unroll
    for (int e = 0; e < DIM / 16; ++e) qv[e] = qrow[e];
    int best = 0, second = 0, i1 = 0;
    for (int j = 0; j < nt; ++j) {
        const int4 *crow = reinterpret_cast<const int4 *>(Cm + (size_t)cid[j] * DIM);
        unsigned lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < DIM / 16; ++e) {
            const int4 cv = crow[e];
            const int qa[4] = {qv[e].x, qv[e].y, qv[e].z, qv[e].w};
            const int ca[4] = {cv.x, cv.y, cv.z, cv.w};
            // element 16 e + 4 w + b sits in byte b of word w; its SSE lane is (4 w + b) % 8
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int qb = (int)(int8_t)(qa[w] >> (8 * b)), cb = (int)(int8_t)(ca[w] >> (8 * b));
                    lanes[(4 * w + b) & 7] += (unsigned)((SIGNED ? qb : qb + 128) * (SIGNED ? cb : cb + 128));
                }
        }
        int ip = 0;
#pragma unroll
        for (int l = 0; l < 8; ++l) ip += SIGNED ? (int)(short)(lanes[l] & 0xffffu) : (int)(lanes[l] & 0xffffu);
        if (ip >= second) {
            if (ip >= best) {
                second = best;
                best = SIGNED ? (int)(short)ip : (int)(unsigned short)ip;
                i1 = j;
            } else {
                second = SIGNED ? (int)(short)ip : (int)(unsigned short)ip;
            }
        }
    }
    int d1, d2;
    if (SIGNED) {
        const int b = min(16129, max(0, best)), s = min(16129, max(0, second));
        d1 = (int)(short)(32258 - 2 * b); d2 = (int)(short)(32258 - 2 * s);
    } else {
        const int b = min(65025, best), s = min(65025, second);
        d1 = min(32767, 65025 - b) * 2; d2 = min(32767, 65025 - s) * 2;
    }
    int res = nt > 0 ? cid[i1] : -1;
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
    hipLaunchKernelGGL(cashash_buckets_kernel, dim3(kCasGroups), dim3(kCasBuckets), 0, s, bucket_ids, n, start, items);
}

void launch_cashash_match(int dim, const MatchProblem *d_problems, int num_problems, int max_n,
    LoweTable tab, hipStream_t s)
{
    if (num_problems <= 0 || max_n <= 0) return;
    const dim3 grid((max_n + 127) / 128, num_problems, 2);
    if (dim == 128)
        hipLaunchKernelGGL((cashash_match_kernel<128, false>), grid, dim3(128), 0, s, d_problems, tab);
    else
        hipLaunchKernelGGL((cashash_match_kernel<64, true>), grid, dim3(128), 0, s, d_problems, tab);
}

}  // namespace osfm
