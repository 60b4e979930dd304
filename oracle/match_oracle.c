/*
 * match_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of OrthoSfM's exhaustive descriptor matcher (vendored MVE),
 * written from scratch in plain C.  It exists only so tests/, smoke() and
 * bench.py's cpu_baseline leg can check / time the HIP path against it.
 * Nothing under orthosfm_amd/ may include, link or call this file.
 *
 * Parity: pinned.  tests/test_oracle_match.py checks every function below
 * bit-for-bit against the reference's own code compiled from
 * /root/reference (oracle/_ref/libref_match.so, built by oracle/Makefile)
 * and against the committed vectors in tests/golden/.
 *
 * Each function cites the reference file:line it restates
 * (paths relative to /root/reference/src/mve/sfm/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* A1: descriptor quantisation. exhaustive_matching.cc:17-38 with
 * math::clamp and math::round (../math/functions.h:70-73):
 *   round(x) = x > 0 ? floor(x + 0.5) : ceil(x - 0.5), all in float.   */

static float oracle_roundf(float x)
{
    return x > 0.0f ? floorf(x + 0.5f) : ceilf(x - 0.5f);
}

ORACLE_API void
oracle_convert_sift(const float *src, int n, uint16_t *dst)
{
    for (long i = 0; i < (long)n * 128; ++i) {
        float v = src[i];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        v = oracle_roundf(v * 255.0f);
        dst[i] = (uint16_t)(unsigned char)v;
    }
}

ORACLE_API void
oracle_convert_surf(const float *src, int n, int16_t *dst)
{
    for (long i = 0; i < (long)n * 64; ++i) {
        float v = src[i];
        v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        v = oracle_roundf(v * 127.0f);
        dst[i] = (int16_t)(signed char)v;
    }
}

/* ------------------------------------------------------------------ */
/* A2: nearest / second nearest neighbour.  nearest_neighbor.cc:60-129
 * (SSE2 branch) + :214-268.
 *
 * The reference accumulates 8 independent 16-bit lanes (lane l sums the
 * products of elements 8*i + l) with wrap-around (`_mm_mullo_epi16`,
 * `_mm_add_epi16`), reads the lanes back as T (sign- or zero-extended),
 * adds them as int, and keeps the running best / second best in fields of
 * type T (so the int is truncated when stored).  The comparisons are
 * `>=` so a later candidate wins ties, and the state starts at
 * (0, 0, idx 0, idx 0).                                                */

typedef struct {
    int dist_1st, dist_2nd;     /* values held in T-typed fields */
    int idx_1st, idx_2nd;
} oracle_nn_result;

/* 8 x 16-bit lanes with wrap-around, written with GCC vector extensions so
 * the compiler emits the packed 16-bit multiply/add the semantics call for. */
typedef uint16_t v8u16 __attribute__((vector_size(16), aligned(2)));
typedef int16_t v8s16 __attribute__((vector_size(16), aligned(2)));

static inline int
ip_u16(const uint16_t *q, const uint16_t *c, int dim)
{
    v8u16 lane = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i + 8 <= dim; i += 8) {
        v8u16 a, b;
        memcpy(&a, q + i, 16);
        memcpy(&b, c + i, 16);
        lane += a * b;                 /* per-lane mod 2^16 */
    }
    int s = 0;
    for (int l = 0; l < 8; ++l) s += lane[l];
    return s;
}

static inline int
ip_s16(const int16_t *q, const int16_t *c, int dim)
{
    v8u16 lane = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i + 8 <= dim; i += 8) {
        v8u16 a, b;
        memcpy(&a, q + i, 16);
        memcpy(&b, c + i, 16);
        lane += a * b;                 /* low 16 bits of the signed product */
    }
    int s = 0;
    for (int l = 0; l < 8; ++l) s += (int16_t)lane[l];
    return s;
}

ORACLE_API void
oracle_nn_find_u16(const uint16_t *query, const uint16_t *elements, int n,
    int dim, oracle_nn_result *out)
{
    uint16_t best = 0, second = 0;
    int i1 = 0, i2 = 0;
    for (int j = 0; j < n; ++j) {
        int ip = ip_u16(query, elements + (long)j * dim, dim);
        if (ip >= (int)second) {
            if (ip >= (int)best) {
                i2 = i1; second = best;
                i1 = j; best = (uint16_t)ip;
            } else {
                i2 = j; second = (uint16_t)ip;
            }
        }
    }
    /* nearest_neighbor.cc:262-267 */
    int b = (int)best < 65025 ? (int)best : 65025;
    int s = (int)second < 65025 ? (int)second : 65025;
    b = 65025 - b; s = 65025 - s;
    b = (b < 32767 ? b : 32767) * 2;
    s = (s < 32767 ? s : 32767) * 2;
    out->dist_1st = (uint16_t)b; out->dist_2nd = (uint16_t)s;
    out->idx_1st = i1; out->idx_2nd = i2;
}

ORACLE_API void
oracle_nn_find_s16(const int16_t *query, const int16_t *elements, int n,
    int dim, oracle_nn_result *out)
{
    int16_t best = 0, second = 0;
    int i1 = 0, i2 = 0;
    for (int j = 0; j < n; ++j) {
        int ip = ip_s16(query, elements + (long)j * dim, dim);
        if (ip >= (int)second) {
            if (ip >= (int)best) {
                i2 = i1; second = best;
                i1 = j; best = (int16_t)ip;
            } else {
                i2 = j; second = (int16_t)ip;
            }
        }
    }
    /* nearest_neighbor.cc:234-237 */
    int b = (int)best, s = (int)second;
    b = b < 0 ? 0 : (b > 16129 ? 16129 : b);
    s = s < 0 ? 0 : (s > 16129 ? 16129 : s);
    out->dist_1st = (int16_t)(32258 - 2 * b);
    out->dist_2nd = (int16_t)(32258 - 2 * s);
    out->idx_1st = i1; out->idx_2nd = i2;
}

/* ------------------------------------------------------------------ */
/* A3: one-way matching.  matching.h:114-146.  Thresholds are squared in
 * float (MATH_POW2, ../math/defines.h:68); FLT_MAX^2 = +inf disables the
 * distance test; 0/0 = NaN compares false so such a match is ACCEPTED.   */

static void
oneway_u16(const uint16_t *s1, int n1, const uint16_t *s2, int n2, int dim,
    float lowe, float dist_thres, int *result)
{
    for (int i = 0; i < n1; ++i) result[i] = -1;
    if (n1 == 0 || n2 == 0) return;
    const float sq_lowe = lowe * lowe;
    const float sq_dist = dist_thres * dist_thres;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n1; ++i) {
        oracle_nn_result r;
        oracle_nn_find_u16(s1 + (long)i * dim, s2, n2, dim, &r);
        if ((float)(uint16_t)r.dist_1st > sq_dist) continue;
        if ((float)(uint16_t)r.dist_1st / (float)(uint16_t)r.dist_2nd > sq_lowe)
            continue;
        result[i] = r.idx_1st;
    }
}

static void
oneway_s16(const int16_t *s1, int n1, const int16_t *s2, int n2, int dim,
    float lowe, float dist_thres, int *result)
{
    for (int i = 0; i < n1; ++i) result[i] = -1;
    if (n1 == 0 || n2 == 0) return;
    const float sq_lowe = lowe * lowe;
    const float sq_dist = dist_thres * dist_thres;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n1; ++i) {
        oracle_nn_result r;
        oracle_nn_find_s16(s1 + (long)i * dim, s2, n2, dim, &r);
        if ((float)(int16_t)r.dist_1st > sq_dist) continue;
        if ((float)(int16_t)r.dist_1st / (float)(int16_t)r.dist_2nd > sq_lowe)
            continue;
        result[i] = r.idx_1st;
    }
}

/* A4: two-way matching, matching.h:148-159 (both directions recomputed). */
ORACLE_API void
oracle_twoway_match_u16(const uint16_t *s1, int n1, const uint16_t *s2, int n2,
    int dim, float lowe, float dist_thres, int *m12, int *m21)
{
    oneway_u16(s1, n1, s2, n2, dim, lowe, dist_thres, m12);
    oneway_u16(s2, n2, s1, n1, dim, lowe, dist_thres, m21);
}

ORACLE_API void
oracle_twoway_match_s16(const int16_t *s1, int n1, const int16_t *s2, int n2,
    int dim, float lowe, float dist_thres, int *m12, int *m21)
{
    oneway_s16(s1, n1, s2, n2, dim, lowe, dist_thres, m12);
    oneway_s16(s2, n2, s1, n1, dim, lowe, dist_thres, m21);
}

/* A5: cross-check.  matching.cc:18-36 and :38-47. */
ORACLE_API void
oracle_remove_inconsistent(int *m12, int n1, int *m21, int n2)
{
    for (int i = 0; i < n1; ++i) {
        if (m12[i] < 0) continue;
        if (m21[m12[i]] != i) m12[i] = -1;
    }
    for (int i = 0; i < n2; ++i) {
        if (m21[i] < 0) continue;
        if (m12[m21[i]] != i) m21[i] = -1;
    }
}

ORACLE_API int
oracle_count_consistent(const int *m12, int n1, const int *m21, int n2)
{
    (void)n2;
    int c = 0;
    for (int i = 0; i < n1; ++i)
        if (m12[i] != -1 && m21[m12[i]] == i) c++;
    return c;
}

/* A6: combine SIFT and SURF results.  matching.cc:49-88.  The SURF
 * entries are shifted by the number of SIFT entries on the OTHER side,
 * and only when that number is non-zero.  Outputs hold ns1+nu1 / ns2+nu2
 * ints.                                                                 */
ORACLE_API void
oracle_combine_results(const int *sift12, int ns1, const int *sift21, int ns2,
    const int *surf12, int nu1, const int *surf21, int nu2,
    int *out12, int *out21)
{
    memcpy(out12, sift12, sizeof(int) * (size_t)ns1);
    memcpy(out12 + ns1, surf12, sizeof(int) * (size_t)nu1);
    memcpy(out21, sift21, sizeof(int) * (size_t)ns2);
    memcpy(out21 + ns2, surf21, sizeof(int) * (size_t)nu2);
    if (ns2 > 0)
        for (int i = ns1; i < ns1 + nu1; ++i)
            if (out12[i] >= 0) out12[i] += ns2;
    if (ns1 > 0)
        for (int i = ns2; i < ns2 + nu2; ++i)
            if (out21[i] >= 0) out21[i] += ns1;
}

/* A7: ExhaustiveMatching::pairwise_match, exhaustive_matching.cc:114-144.
 * SIFT matching happens only when view 1 has SIFT descriptors, SURF only
 * when view 1 has SURF descriptors; a skipped type contributes EMPTY
 * lists to combine_results (not lists of -1).  The caller sizes out12 /
 * out21 for the worst case (ns1+nu1, ns2+nu2); the actual lengths are
 * returned through len12 / len21.                                        */
ORACLE_API void
oracle_pairwise_match(
    const uint16_t *sift1, int ns1, const int16_t *surf1, int nu1,
    const uint16_t *sift2, int ns2, const int16_t *surf2, int nu2,
    float sift_lowe, float sift_dist, float surf_lowe, float surf_dist,
    int *out12, int *len12, int *out21, int *len21)
{
    int *a12 = NULL, *a21 = NULL, *b12 = NULL, *b21 = NULL;
    int la1 = 0, la2 = 0, lb1 = 0, lb2 = 0;
    if (ns1 > 0) {
        la1 = ns1; la2 = ns2;
        a12 = (int *)malloc(sizeof(int) * (size_t)(la1 + 1));
        a21 = (int *)malloc(sizeof(int) * (size_t)(la2 + 1));
        oracle_twoway_match_u16(sift1, ns1, sift2, ns2, 128, sift_lowe,
            sift_dist, a12, a21);
        oracle_remove_inconsistent(a12, la1, a21, la2);
    }
    if (nu1 > 0) {
        lb1 = nu1; lb2 = nu2;
        b12 = (int *)malloc(sizeof(int) * (size_t)(lb1 + 1));
        b21 = (int *)malloc(sizeof(int) * (size_t)(lb2 + 1));
        oracle_twoway_match_s16(surf1, nu1, surf2, nu2, 64, surf_lowe,
            surf_dist, b12, b21);
        oracle_remove_inconsistent(b12, lb1, b21, lb2);
    }
    oracle_combine_results(a12, la1, a21, la2, b12, lb1, b21, lb2, out12, out21);
    *len12 = la1 + lb1;
    *len21 = la2 + lb2;
    free(a12); free(a21); free(b12); free(b21);
}

/* A7: ExhaustiveMatching::pairwise_match_lowres,
 * exhaustive_matching.cc:146-180: two-way match on the first
 * min(num_features, n) descriptors, NO remove_inconsistent, count mutual
 * matches; SIFT if view 1 has any SIFT, else SURF, else 0.               */
ORACLE_API int
oracle_pairwise_match_lowres(
    const uint16_t *sift1, int ns1, const int16_t *surf1, int nu1,
    const uint16_t *sift2, int ns2, const int16_t *surf2, int nu2,
    float sift_lowe, float sift_dist, float surf_lowe, float surf_dist,
    int num_features)
{
    if (ns1 > 0) {
        int n1 = ns1 < num_features ? ns1 : num_features;
        int n2 = ns2 < num_features ? ns2 : num_features;
        int *m12 = (int *)malloc(sizeof(int) * (size_t)(n1 + 1));
        int *m21 = (int *)malloc(sizeof(int) * (size_t)(n2 + 1));
        oracle_twoway_match_u16(sift1, n1, sift2, n2, 128, sift_lowe,
            sift_dist, m12, m21);
        int c = oracle_count_consistent(m12, n1, m21, n2);
        free(m12); free(m21);
        return c;
    }
    if (nu1 > 0) {
        int n1 = nu1 < num_features ? nu1 : num_features;
        int n2 = nu2 < num_features ? nu2 : num_features;
        int *m12 = (int *)malloc(sizeof(int) * (size_t)(n1 + 1));
        int *m21 = (int *)malloc(sizeof(int) * (size_t)(n2 + 1));
        oracle_twoway_match_s16(surf1, n1, surf2, n2, 64, surf_lowe,
            surf_dist, m12, m21);
        int c = oracle_count_consistent(m12, n1, m21, n2);
        free(m12); free(m21);
        return c;
    }
    return 0;
}

/* A8 (pair enumeration + low-res gate), bundler_matching.cc:92-93 and
 * :146-172.  Returns for linear pair index i the view ids (v1 > v2).     */
ORACLE_API void
oracle_pair_from_index(long i, int *v1, int *v2)
{
    int a = (int)(0.5 + sqrt(0.25 + 2.0 * (double)i));
    *v1 = a;
    *v2 = (int)i - a * (a - 1) / 2;
}

ORACLE_API int
oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
