// Geometric verification on the GPU: RANSAC over 8-point fundamental matrices
// with Sampson-distance inliers, one workgroup per view pair
// (sfm::RansacFundamental::estimate, src/mve/sfm/ransac_fundamental.cc:26-105;
// fundamental_8_point / enforce_fundamental_constraints / sampson_distance,
// src/mve/sfm/fundamental.cc:78-127,225-246), as called per pair from
// bundler::Matching::two_view_matching (bundler_matching.cc:194-219).
//
// The reference draws samples from std::rand() shared between OpenMP threads,
// so its inlier sets change from run to run; here sample d of iteration i of
// pair p is splitmix64(seed, p, i, d): results are reproducible and independent
// of how pairs are batched or sharded.  All arithmetic is double precision in
// the operation order of the CPU oracle, so both agree bit for bit.
#include <atomic>
#include <mutex>

#include "ransac_kernels.h"
#include "osfm_common.h"

#include <algorithm>

namespace osfm {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint64_t ransac_rand(uint64_t seed, uint64_t pair, uint64_t it, uint64_t draw)
{
    return splitmix64(splitmix64(seed ^ (pair * 0xD1342543DE82EF95ull)) + it * 0x2545F4914F6CDD1Dull + draw);
}

// fundamental.cc:225-246
__device__ __forceinline__ double
sampson(const double *F, double x1, double y1, double x2, double y2)
{
    double n = 0.0;
    n += x2 * (x1 * F[0] + y1 * F[1] + F[2]);
    n += y2 * (x1 * F[3] + y1 * F[4] + F[5]);
    n += 1.0 * (x1 * F[6] + y1 * F[7] + F[8]);
    n *= n;
    double sum = 0.0, t;
    t = x1 * F[0] + y1 * F[1] + F[2]; sum += t * t;
    t = x1 * F[3] + y1 * F[4] + F[5]; sum += t * t;
    t = x2 * F[0] + y2 * F[3] + F[6]; sum += t * t;
    t = x2 * F[1] + y2 * F[4] + F[7]; sum += t * t;
    return n / sum;
}

// sampson(...) < thr2 decided without the division wherever a product test is
// safe: n < sum * thr2 (1 - 2^-49) implies fl(n / sum) < thr2 and
// n >= sum * thr2 (1 + 2^-49) implies fl(n / sum) >= thr2 (the roundings of the
// products and of the quotient, 2^-53 each, cannot bridge 2^-49); only the
// sliver in between takes the reference's division (fundamental.cc:245).  Same
// decisions as the oracle, bit for bit.  thr_lo / thr_hi are those two bounds.
__device__ __forceinline__ bool
sampson_below(const double *F, double x1, double y1, double x2, double y2, double thr2,
    double thr_lo, double thr_hi)
{
    const double a = x1 * F[0] + y1 * F[1] + F[2];
    const double b = x1 * F[3] + y1 * F[4] + F[5];
    const double c = x1 * F[6] + y1 * F[7] + F[8];
    double n = 0.0;
    n += x2 * a;
    n += y2 * b;
    n += c;                        // the reference's 1.0 * c, an exact identity
    n *= n;
    double sum = 0.0, t;
    sum += a * a;
    sum += b * b;
    t = x2 * F[0] + y2 * F[3] + F[6]; sum += t * t;
    t = x2 * F[1] + y2 * F[4] + F[7]; sum += t * t;
    if (n < sum * thr_lo) return true;                    // strict: sum == 0 never passes here
    if (n >= sum * thr_hi) return false;                  // 0/0 and x/0 of the reference: not below
    return n / sum < thr2;                                // the sliver, and NaN operands
}

__device__ void eig3_fixed(double A[3][3], double V[3][3], double w[3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

// 8-point fundamental matrix of one sample: null vector of the 8x9 system by
// Gauss-Jordan elimination with full pivoting, then the rank-2 projection.
__device__ bool
eight_point(const double p1[8][2], const double p2[8][2], double F[9])
{
    double A[8][9];
    for (int i = 0; i < 8; ++i) {
        const double x1 = p1[i][0], y1 = p1[i][1], x2 = p2[i][0], y2 = p2[i][1];
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2;
        A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
        A[i][6] = x1; A[i][7] = y1; A[i][8] = 1.0;
    }
    int colperm[9];
    for (int c = 0; c < 9; ++c) colperm[c] = c;
    for (int r = 0; r < 8; ++r) {
        int pr = r, pc = r;
        double best = -1.0;
        for (int i = r; i < 8; ++i)
            for (int j = r; j < 9; ++j) {
                const double v = fabs(A[i][j]);
                if (v > best) { best = v; pr = i; pc = j; }
            }
        if (!(best > 0.0)) { for (int i = 0; i < 9; ++i) F[i] = 0.0; return false; }
        if (pr != r) for (int j = 0; j < 9; ++j) { const double t = A[r][j]; A[r][j] = A[pr][j]; A[pr][j] = t; }
        if (pc != r) {
            for (int i = 0; i < 8; ++i) { const double t = A[i][r]; A[i][r] = A[i][pc]; A[i][pc] = t; }
            const int t = colperm[r]; colperm[r] = colperm[pc]; colperm[pc] = t;
        }
        const double inv = 1.0 / A[r][r];
        for (int j = r; j < 9; ++j) A[r][j] *= inv;
        for (int i = 0; i < 8; ++i) {
            if (i == r) continue;
            const double fct = A[i][r];
            for (int j = r; j < 9; ++j) A[i][j] -= fct * A[r][j];
        }
    }
    double f[9], n2 = 1.0;
    for (int r = 0; r < 8; ++r) { f[colperm[r]] = -A[r][8]; n2 += A[r][8] * A[r][8]; }
    f[colperm[8]] = 1.0;
    const double invn = 1.0 / sqrt(n2);
    for (int i = 0; i < 9; ++i) f[i] *= invn;
    double M[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            M[i][j] = f[0 + i] * f[0 + j] + f[3 + i] * f[3 + j] + f[6 + i] * f[6 + j];
    eig3_fixed(M, V, w);
    int m = 0;
    if (w[1] < w[m]) m = 1;
    if (w[2] < w[m]) m = 2;
    const double v3[3] = { V[0][m], V[1][m], V[2][m] };
    for (int r = 0; r < 3; ++r) {
        const double fv = f[3 * r] * v3[0] + f[3 * r + 1] * v3[1] + f[3 * r + 2] * v3[2];
        for (int c = 0; c < 3; ++c) F[3 * r + c] = f[3 * r + c] - fv * v3[c];
    }
    return true;
}

// ---------------------------------------------------------------------------
// Packed single-precision pre-classification of the scoring loop.
//
// 7.8e9 Sampson tests per 1225 pairs are 85 % of this kernel, and the reference's double
// arithmetic (no FMA, 41 operations per test) runs at the same rate as float here -- but
// v_pk_fma_f32 does two floats per lane and instruction, and a thread scores two
// hypotheses, so one packed operation serves both.  The float result only DECIDES when
// it is out of reach of every error it can carry; the rest (about one test in 10^4) takes
// the double path above, so the counts are those of the double arithmetic, bit for bit.
//
// Bound, for |x1|, |y1|, |x2|, |y2| <= 1 (checked per chunk; a chunk with a larger
// coordinate is scored in double), u = 2^-24, S = sum |F_ij|, exact quantities unmarked,
// computed ones with ~ (F rounded to float: one u per entry; every FMA one rounding):
//   a = x1 F0 + y1 F1 + F2 (b, c, t3, t4 alike):   |a~ - a| <= 4u (|F0| + |F1| + |F2|)
//   n = x2 a + y2 b + c:                           |n~ - n| <= 4uS + 2u S (1 + 4u) <= 7uS
//   s = a^2 + b^2 + t3^2 + t4^2:  |s~ - s| <= 2 sqrt(s) 4u 2S + 64 u^2 S^2 + 4u s
//                                           <= 8u (s + S^2) + 4u s + ...  <= 14u s + 9u S^2
//     (sqrt(s) <= (s / S + S) / 2)
// so with N = |n~|, E = 10uS, and m = 1e-5 standing for the roundings of the test itself
// and of the double reference (1e-15):
//   (N + E)^2 < thr2 (1 - m) / (1 + 14u) (s~ - 12u S^2)   =>  n^2 < thr2 s      inlier for sure
//   (max(N - E, 0))^2 > thr2 (1 + m) / (1 - 14u) (s~ + 12u S^2)  =>  n^2 > thr2 s   not one
// NaN or infinity anywhere fails both comparisons and falls through to the double path.
// ---------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

struct PrefilterConst { v2f F[9], E, K1, K2, K3, K4; };

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// next float above |v| (constants of the bound must not be rounded down)
__device__ __forceinline__ float float_up(double v)
{
    float f = (float)v;
    if ((double)f < v) f = __uint_as_float(__float_as_uint(f) + 1u);
    return f;
}
__device__ __forceinline__ float float_down(double v)
{
    float f = (float)v;
    if ((double)f > v && f > 0.f) f = __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

__device__ __forceinline__ void
prefilter_setup(const double (&F)[2][9], double thr2, PrefilterConst &pc)
{
    const double u = 5.9604644775390625e-08;      // 2^-24
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double S = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { S += fabs(F[h][i]); pc.F[i][h] = (float)F[h][i]; }
        const double k1 = thr2 * (1.0 - 1e-5) / (1.0 + 14.0 * u), k3 = thr2 * (1.0 + 1e-5) / (1.0 - 14.0 * u);
        pc.E[h] = float_up(10.0 * u * S);
        pc.K1[h] = float_down(k1);
        pc.K2[h] = float_up(k1 * 12.0 * u * S * S * 1.001);
        pc.K3[h] = float_up(k3);
        pc.K4[h] = float_up(k3 * 12.0 * u * S * S * 1.001);
    }
}

// bit 0 / 1: hypothesis 0 / 1 certainly below the threshold; bit 2 / 3: certainly not
__device__ __forceinline__ unsigned
prefilter_test(const PrefilterConst &pc, float x1, float y1, float x2, float y2)
{
    const v2f X1 = {x1, x1}, Y1 = {y1, y1}, X2 = {x2, x2}, Y2 = {y2, y2};
    const v2f a = pk_fma(X1, pc.F[0], pk_fma(Y1, pc.F[1], pc.F[2]));
    const v2f b = pk_fma(X1, pc.F[3], pk_fma(Y1, pc.F[4], pc.F[5]));
    const v2f c = pk_fma(X1, pc.F[6], pk_fma(Y1, pc.F[7], pc.F[8]));
    const v2f n = pk_fma(X2, a, pk_fma(Y2, b, c));
    const v2f t3 = pk_fma(X2, pc.F[0], pk_fma(Y2, pc.F[3], pc.F[6]));
    const v2f t4 = pk_fma(X2, pc.F[1], pk_fma(Y2, pc.F[4], pc.F[7]));
    const v2f sm = pk_fma(t4, t4, pk_fma(t3, t3, pk_fma(b, b, a * a)));
    const v2f N = __builtin_elementwise_abs(n);
    const v2f p = N + pc.E;
    const v2f zero = {0.f, 0.f};
    const v2f q = __builtin_elementwise_max(N - pc.E, zero);
    const v2f lo = pk_fma(pc.K1, sm, -pc.K2), hi = pk_fma(pc.K3, sm, pc.K4);
    const v2f p2 = p * p, q2 = q * q;
    unsigned r = 0;
    r |= p2[0] < lo[0] ? 1u : 0u;
    r |= p2[1] < lo[1] ? 2u : 0u;
    r |= q2[0] > hi[0] ? 4u : 0u;
    r |= q2[1] > hi[1] ? 8u : 0u;
    return r;
}

constexpr int kHyp = 2;            // hypotheses per thread and pass
constexpr int kChunk = 1024;       // matches staged in LDS at a time
// The hypotheses of a pair are split over kRansacSplit workgroups: one workgroup per pair
// of the bench (1225 workgroups, two resident per CU) ran as 2.4 waves of workgroups, the
// last one 40 % full.  Every part keeps its best (count, iteration, F) in a slot; the part
// that finishes last (a counter per pair) picks the overall best and lists its inliers.
struct RansacSlot { int32_t count, iter; double F[9]; };

size_t ransac_scratch_bytes(int num_jobs)
{
    return (size_t)std::max(num_jobs, 1) * (kRansacSplit * sizeof(RansacSlot) + 16);
}

__global__ __launch_bounds__(256) void
ransac_kernel(const RansacJob *__restrict__ jobs, int num_jobs, int max_iterations, double thr2, uint64_t seed,
    RansacSlot *__restrict__ slots, int32_t *__restrict__ done, int prefilter, unsigned long long *check)
{
    __shared__ double4 mpt[kChunk];           // (x1, y1, x2, y2) of the staged matches, widened once
    __shared__ float4 mpf[kChunk];            // the same as the floats they are (pre-classification)
    __shared__ int s_wide;                    // a staged coordinate lies outside [-1, 1]
    __shared__ int s_count[256], s_iter[256];
    __shared__ double s_F[9];
    __shared__ int s_wave[4], s_run, s_last;

    const int jid = blockIdx.x / kRansacSplit, part = blockIdx.x % kRansacSplit;
    const RansacJob job = jobs[jid];
    const int k = job.k, tid = threadIdx.x;
    if (k < 8) {                      // the reference throws (ransac_fundamental.cc:66-67)
        if (tid == 0 && part == 0) *job.count_out = -1;
        return;
    }
    // this part's iterations: whole passes of 256 * kHyp
    const int per_part = ((max_iterations + kRansacSplit - 1) / kRansacSplit + 256 * kHyp - 1) / (256 * kHyp) * (256 * kHyp);
    const int it_begin = part * per_part, it_end = min(max_iterations, it_begin + per_part);
    const double thr_lo = thr2 * (1.0 - 1.7763568394002505e-15);     // 2^-49
    const double thr_hi = thr2 * (1.0 + 1.7763568394002505e-15);
    int best_count = 0, best_iter = 0x7fffffff;
    double bestF[9];
    for (int i = 0; i < 9; ++i) bestF[i] = 0.0;

    for (int base = it_begin; base < it_end; base += 256 * kHyp) {
        double F[kHyp][9];
        bool valid[kHyp];
        int cnt[kHyp];
#pragma unroll
        for (int h = 0; h < kHyp; ++h) {
            const int it = base + h * 256 + tid;
            valid[h] = false; cnt[h] = 0;
            for (int i = 0; i < 9; ++i) F[h][i] = 0.0;
            if (it >= it_end) continue;
            // 8 distinct match ids, ascending (std::set order, :69-76)
            int idx[8], n = 0;
            for (uint64_t d = 0; n < 8; ++d) {
                const int v = (int)(ransac_rand(seed, job.pair_id, (uint64_t)it, d) % (uint64_t)k);
                bool dup = false;
                for (int i = 0; i < n; ++i) dup |= idx[i] == v;
                if (!dup) idx[n++] = v;
            }
            for (int i = 1; i < 8; ++i) {
                const int v = idx[i];
                int j = i - 1;
                while (j >= 0 && idx[j] > v) { idx[j + 1] = idx[j]; --j; }
                idx[j + 1] = v;
            }
            double p1[8][2], p2[8][2];
            for (int i = 0; i < 8; ++i) {
                const int a = job.corr[2 * idx[i]], b = job.corr[2 * idx[i] + 1];
                p1[i][0] = job.pos1[2 * a]; p1[i][1] = job.pos1[2 * a + 1];
                p2[i][0] = job.pos2[2 * b]; p2[i][1] = job.pos2[2 * b + 1];
            }
            valid[h] = eight_point(p1, p2, F[h]);
        }
        static_assert(kHyp == 2, "the pre-classification packs the two hypotheses of a thread");
        PrefilterConst pc;
        prefilter_setup(F, thr2, pc);
        // inlier counts of this thread's hypotheses over all matches
        for (int c0 = 0; c0 < k; c0 += kChunk) {
            __syncthreads();
            if (tid == 0) s_wide = 0;
            __syncthreads();
            bool wide = false;
            for (int i = tid; i < kChunk && c0 + i < k; i += 256) {
                const int a = job.corr[2 * (c0 + i)], b = job.corr[2 * (c0 + i) + 1];
                const float4 f = make_float4(job.pos1[2 * a], job.pos1[2 * a + 1], job.pos2[2 * b], job.pos2[2 * b + 1]);
                mpf[i] = f;
                mpt[i] = make_double4(f.x, f.y, f.z, f.w);
                wide |= !(fabsf(f.x) <= 1.f && fabsf(f.y) <= 1.f && fabsf(f.z) <= 1.f && fabsf(f.w) <= 1.f);
            }
            if (wide) s_wide = 1;
            __syncthreads();
            const int lim = min(kChunk, k - c0);
            if (s_wide || !prefilter) {
                for (int i = 0; i < lim; ++i) {
                    const double4 m = mpt[i];
                    // invalid hypotheses (all-zero F) are evaluated too and discarded below:
                    // no divergence inside the loop
#pragma unroll
                    for (int h = 0; h < kHyp; ++h)
                        cnt[h] += sampson_below(F[h], m.x, m.y, m.z, m.w, thr2, thr_lo, thr_hi) ? 1 : 0;
                }
                continue;
            }
            // four matches per round: their (broadcast) LDS reads are issued together, the
            // first use waits once -- one read per round trip left the loop latency-bound
            const unsigned vmask = (valid[0] ? 1u : 0u) | (valid[1] ? 2u : 0u);
            auto score = [&](int i, const float4 &f) {
                const unsigned r = prefilter_test(pc, f.x, f.y, f.z, f.w);
                cnt[0] += r & 1u;
                cnt[1] += (r >> 1) & 1u;
                // undecided (neither bit of a hypothesis set): the double path; counts of an
                // invalid hypothesis are discarded below, it never goes there
                const unsigned und = ~(r | (r >> 2)) & vmask;
                if (und) {
                    const double4 m = mpt[i];
                    if (und & 1u) cnt[0] += sampson_below(F[0], m.x, m.y, m.z, m.w, thr2, thr_lo, thr_hi) ? 1 : 0;
                    if (und & 2u) cnt[1] += sampson_below(F[1], m.x, m.y, m.z, m.w, thr2, thr_lo, thr_hi) ? 1 : 0;
                }
                if (check) {
                    // self-check: every decision the floats took against the double path
                    const double4 m = mpt[i];
#pragma unroll
                    for (int h = 0; h < kHyp; ++h) {
                        const bool ref = sampson_below(F[h], m.x, m.y, m.z, m.w, thr2, thr_lo, thr_hi);
                        const bool in = (r >> h) & 1u, out = (r >> (2 + h)) & 1u;
                        if (valid[h] && ((in && !ref) || (out && ref) || (in && out))) atomicAdd(&check[0], 1ull);
                        if (valid[h] && !in && !out) atomicAdd(&check[1], 1ull);
                        if (valid[h]) atomicAdd(&check[2], 1ull);
                    }
                }
            };
            int i = 0;
            for (; i + 4 <= lim; i += 4) {
                const float4 f0 = mpf[i], f1 = mpf[i + 1], f2 = mpf[i + 2], f3 = mpf[i + 3];
                score(i, f0); score(i + 1, f1); score(i + 2, f2); score(i + 3, f3);
            }
            for (; i < lim; ++i) score(i, mpf[i]);
        }
#pragma unroll
        for (int h = 0; h < kHyp; ++h) {
            const int it = base + h * 256 + tid;
            // strictly more inliers wins; equal counts keep the earlier iteration (:47)
            if (valid[h] && (cnt[h] > best_count || (cnt[h] == best_count && cnt[h] > 0 && it < best_iter))) {
                best_count = cnt[h]; best_iter = it;
                for (int i = 0; i < 9; ++i) bestF[i] = F[h][i];
            }
        }
    }
    // block argmax: (count desc, iteration asc)
    s_count[tid] = best_count; s_iter[tid] = best_iter;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) {
            const int c2 = s_count[tid + st], i2 = s_iter[tid + st];
            if (c2 > s_count[tid] || (c2 == s_count[tid] && i2 < s_iter[tid])) { s_count[tid] = c2; s_iter[tid] = i2; }
        }
        __syncthreads();
    }
    int win_count = s_count[0], win_iter = s_iter[0];
    if (best_count == win_count && best_iter == win_iter && win_count > 0)
        for (int i = 0; i < 9; ++i) s_F[i] = bestF[i];
    __syncthreads();
    // this part's best into its slot; the last part to get here goes on with all of them
    if (tid == 0) {
        RansacSlot &mine = slots[(size_t)jid * kRansacSplit + part];
        mine.count = win_count; mine.iter = win_iter;
        for (int i = 0; i < 9; ++i) mine.F[i] = win_count > 0 ? s_F[i] : 0.0;
        __threadfence();
        s_last = atomicAdd(&done[jid], 1) == kRansacSplit - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
        __threadfence();
        int bc = 0, bi = 0x7fffffff, bp = -1;
        for (int p = 0; p < kRansacSplit; ++p) {
            const RansacSlot &sl = slots[(size_t)jid * kRansacSplit + p];
            // strictly more inliers wins; equal counts keep the earlier iteration (:47)
            if (sl.count > bc || (sl.count == bc && sl.count > 0 && sl.iter < bi)) { bc = sl.count; bi = sl.iter; bp = p; }
        }
        s_count[0] = bc; s_iter[0] = bi;
        if (bp >= 0) for (int i = 0; i < 9; ++i) s_F[i] = slots[(size_t)jid * kRansacSplit + bp].F[i];
        done[jid] = 0;                 // ready for the next launch
        s_run = 0;
    }
    __syncthreads();
    win_count = s_count[0]; win_iter = s_iter[0];
    if (win_count == 0) {
        if (tid == 0) { *job.count_out = 0; if (job.F_out) for (int i = 0; i < 9; ++i) job.F_out[i] = 0.0; }
        return;
    }
    double Fw[9];
    for (int i = 0; i < 9; ++i) Fw[i] = s_F[i];
    if (tid == 0 && job.F_out) for (int i = 0; i < 9; ++i) job.F_out[i] = Fw[i];
    // ordered list of the inlier ids of the winning hypothesis
    const int lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < k; c0 += 256) {
        const int i = c0 + tid;
        bool in = false;
        if (i < k) {
            const int a = job.corr[2 * i], b = job.corr[2 * i + 1];
            in = sampson(Fw, job.pos1[2 * a], job.pos1[2 * a + 1], job.pos2[2 * b], job.pos2[2 * b + 1]) < thr2;
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_run;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (in) job.inliers_out[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) *job.count_out = s_run;
}

// Scoring mode of the process and, per device, the counters of mode 2 (every pre-classified decision compared with
// the double path): a launch counts into the block of the device it runs on -- one block for "the current device"
// let device k add into device 0's memory once a matcher had shards on several devices.
static std::atomic<int> g_ransac_mode{1};
constexpr int kRansacMaxDevices = 64;
static std::mutex g_ransac_check_mu;
static unsigned long long *g_ransac_check[kRansacMaxDevices];

static unsigned long long *ransac_check_block(bool create)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kRansacMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_ransac_check_mu);
    if (!g_ransac_check[dev] && create) {
        if (hipMalloc(&g_ransac_check[dev], 3 * sizeof(unsigned long long)) != hipSuccess) { g_ransac_check[dev] = nullptr; return nullptr; }
        (void)hipMemset(g_ransac_check[dev], 0, 3 * sizeof(unsigned long long));
    }
    return g_ransac_check[dev];
}

void launch_ransac(const RansacJob *d_jobs, int num_jobs, int max_iterations, double threshold,
    uint64_t seed, void *scratch, hipStream_t s)
{
    if (num_jobs <= 0) return;
    RansacSlot *slots = static_cast<RansacSlot *>(scratch);
    int32_t *done = reinterpret_cast<int32_t *>(static_cast<char *>(scratch) + (size_t)num_jobs * kRansacSplit * sizeof(RansacSlot));
    (void)hipMemsetAsync(done, 0, (size_t)num_jobs * sizeof(int32_t), s);
    const int mode = g_ransac_mode.load();
    hipLaunchKernelGGL(ransac_kernel, dim3(num_jobs * kRansacSplit), dim3(256), 0, s, d_jobs, num_jobs, max_iterations,
        threshold * threshold, seed, slots, done, mode != 0 ? 1 : 0, mode == 2 ? ransac_check_block(true) : nullptr);
}

// diagnostics (osfm_ransac_selfcheck): mode 0 = double path only, 1 = pre-classification (default),
// 2 = pre-classification with every decision compared against the double path; counters[3] =
// wrong decisions, undecided tests, tests -- summed over the devices that ran checked launches
int ransac_set_mode(int mode, unsigned long long *counters_out)
{
    int cur = 0;
    OSFM_HIP_CHECK(hipGetDevice(&cur));
    unsigned long long sum[3] = {0, 0, 0};
    for (int dev = 0; dev < kRansacMaxDevices; ++dev) {
        unsigned long long *blk;
        { std::lock_guard<std::mutex> lock(g_ransac_check_mu); blk = g_ransac_check[dev]; }
        if (!blk) continue;
        OSFM_HIP_CHECK(hipSetDevice(dev));
        OSFM_HIP_CHECK(hipDeviceSynchronize());
        unsigned long long c[3] = {0, 0, 0};
        OSFM_HIP_CHECK(hipMemcpy(c, blk, sizeof(c), hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) sum[i] += c[i];
        if (mode == 2) OSFM_HIP_CHECK(hipMemset(blk, 0, sizeof(c)));
        else {
            (void)hipFree(blk);
            std::lock_guard<std::mutex> lock(g_ransac_check_mu);
            g_ransac_check[dev] = nullptr;
        }
    }
    OSFM_HIP_CHECK(hipSetDevice(cur));
    if (counters_out) for (int i = 0; i < 3; ++i) counters_out[i] = sum[i];
    g_ransac_mode.store(mode);
    return OSFM_OK;
}

}  // namespace osfm
