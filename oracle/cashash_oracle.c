/*
 * cashash_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's cascade-hashing matcher, the application's
 * default (src/mve/sfm/cascade_hashing.h:29-221,225-470, cascade_hashing.cc:20-227):
 *   projection matrices   GlobalData::generate_proj_matrices (h:225-254)
 *   descriptor average    compute_avg_descriptors (cc:128-163)
 *   zero mean + hashes    compute_zero_mean_descs (cc:165-183), compute_cascade_hashes (h:258-310)
 *   buckets               build_buckets (cc:187-209)
 *   one-way matching      oneway_match (h:328-412) with collect_features_from_buckets
 *                         (h:414-444) and collect_top_ranked_candidates (h:446-468)
 *   pairwise_match        cc:73-104 (two-way, cross-check, combine)
 *
 * Pinned: every stage is compared bit for bit with the reference's own files
 * compiled into oracle/_ref/libref_cashash.so (tests/test_oracle_cashash.py).
 *
 * The projection matrices come from std::mt19937(0) through
 * std::normal_distribution<>, whose algorithm the C++ standard leaves to the
 * library; restated here is libstdc++'s (Marsaglia polar method over
 * generate_canonical<double, 53>, second value cached) with glibc's log/sqrt --
 * what the reference does when built with GCC.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---- std::mt19937 ---- */
typedef struct { uint32_t mt[624]; int idx; } mt19937;

static void mt_seed(mt19937 *g, uint32_t s)
{
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937 *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* libstdc++ generate_canonical<double, 53>(mt19937): two draws, low word first */
static double canonical(mt19937 *g)
{
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) { sum += (double)mt_next(g) * tmp; tmp *= 4294967296.0; }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

typedef struct { int have; double saved; } normal_state;

static double normal(mt19937 *g, normal_state *st)
{
    if (st->have) { st->have = 0; return st->saved; }
    double x, y, r2;
    do {
        x = 2.0 * canonical(g) - 1.0;
        y = 2.0 * canonical(g) - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = sqrt(-2 * log(r2) / r2);
    st->saved = x * mult; st->have = 1;
    return y * mult;
}

/* prim [dim][dim], sec [groups][bits][dim] (h:225-254); one generator per descriptor type */
ORACLE_API void
oracle_cashash_proj_matrices(int dim, int groups, int bits, float *prim, float *sec)
{
    mt19937 g;
    mt_seed(&g, 0);
    normal_state st = { 0, 0.0 };
    for (int i = 0; i < dim; ++i)
        for (int j = 0; j < dim; ++j) prim[i * dim + j] = (float)normal(&g, &st);
    for (int grp = 0; grp < groups; ++grp)
        for (int i = 0; i < bits; ++i)
            for (int j = 0; j < dim; ++j) sec[(grp * bits + i) * dim + j] = (float)normal(&g, &st);
}

/* cc:128-163 for one descriptor type: descs = all views concatenated in view
 * order, values as held in the u16 / s16 arrays; div = 255 (SIFT) or 127 (SURF) */
ORACLE_API void
oracle_cashash_avg(const int32_t *descs, int64_t n_total, int dim, float div, float *avg)
{
    for (int k = 0; k < dim; ++k) avg[k] = 0.0f;
    for (int64_t j = 0; j < n_total; ++j)
        for (int k = 0; k < dim; ++k) avg[k] += (float)descs[j * dim + k] / div;
    for (int k = 0; k < dim; ++k) avg[k] /= (float)n_total;
}

/* zero mean (cc:165-183) + hashes and bucket ids (h:258-310) of one view.
 * hashes [n][dim/64], bucket_ids [groups][n] */
ORACLE_API void
oracle_cashash_hashes(const int32_t *descs, int n, int dim, float div, const float *avg,
    const float *prim, const float *sec, int groups, int bits, uint64_t *hashes, uint16_t *bucket_ids)
{
    const int words = dim / 64;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float zm[128];
        for (int k = 0; k < dim; ++k) zm[k] = (float)descs[(int64_t)i * dim + k] / div - avg[k];
        for (int w = 0; w < words; ++w) {
            uint64_t h = 0;
            for (int k = w * 64; k < (w + 1) * 64; ++k) {
                float sum = 0.0f;
                for (int e = 0; e < dim; ++e) sum = sum + zm[e] * prim[k * dim + e];
                h = (h << 1) | (uint64_t)(sum > 0.0f);
            }
            hashes[(int64_t)i * words + w] = h;
        }
        for (int g = 0; g < groups; ++g) {
            uint16_t id = 0;
            for (int b = 0; b < bits; ++b) {
                float sum = 0.0f;
                const float *pv = sec + (int64_t)(g * bits + b) * dim;
                for (int e = 0; e < dim; ++e) sum = sum + zm[e] * pv[e];
                id = (uint16_t)((id << 1) | (sum > 0.0f));
            }
            bucket_ids[(int64_t)g * n + i] = id;
        }
    }
}

typedef struct { int32_t dist_1st, dist_2nd, idx_1st, idx_2nd; } nn_result;
void oracle_nn_find_u16(const uint16_t *q, const uint16_t *el, int n, int dim, nn_result *out);
void oracle_nn_find_s16(const int16_t *q, const int16_t *el, int n, int dim, nn_result *out);

/* h:328-412.  is_signed: SURF (s16, dim 64) else SIFT (u16, dim 128); descriptors
 * passed as 16-bit arrays.  result[n1] (-1 = no match); an empty set leaves
 * everything -1 (the reference returns an EMPTY vector there, h:341-342). */
ORACLE_API void
oracle_cashash_oneway(int is_signed, int dim, int groups, int bits,
    const void *d1, int n1, const uint64_t *h1, const uint16_t *b1,
    const void *d2, int n2, const uint64_t *h2, const uint16_t *b2,
    float lowe, float dist_thres, int min_cand, int max_cand, int32_t *result)
{
    for (int i = 0; i < n1; ++i) result[i] = -1;
    if (n1 == 0 || n2 == 0) return;
    const int words = dim / 64;
    const int nb = 1 << bits;
    /* build_buckets (cc:187-209) of set 2: ids in ascending order per bucket */
    int32_t *start = (int32_t *)calloc((size_t)groups * (nb + 1), sizeof(int32_t));
    int32_t *items = (int32_t *)malloc(sizeof(int32_t) * (size_t)groups * n2);
    for (int g = 0; g < groups; ++g) {
        int32_t *s = start + (size_t)g * (nb + 1);
        for (int i = 0; i < n2; ++i) s[b2[(size_t)g * n2 + i] + 1]++;
        for (int b = 0; b < nb; ++b) s[b + 1] += s[b];
        int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * nb);
        memcpy(fill, s, sizeof(int32_t) * nb);
        for (int i = 0; i < n2; ++i) items[(size_t)g * n2 + fill[b2[(size_t)g * n2 + i]]++] = i;
        free(fill);
    }
    const float sq_lowe = lowe * lowe, sq_dist = dist_thres * dist_thres;
#pragma omp parallel
    {
        uint8_t *used = (uint8_t *)malloc((size_t)n2);
        int32_t *grp_items = (int32_t *)malloc(sizeof(int32_t) * (size_t)(dim + 1) * n2);   /* grouped_features */
        int32_t *grp_n = (int32_t *)malloc(sizeof(int32_t) * (dim + 1));
        void *tmp = malloc((size_t)max_cand * dim * 2);
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n1; ++i) {
            memset(used, 0, (size_t)n2);
            memset(grp_n, 0, sizeof(int32_t) * (dim + 1));
            /* collect_features_from_buckets (h:414-444) */
            for (int g = 0; g < groups; ++g) {
                const int bucket = (uint8_t)b1[(size_t)g * n1 + i];             /* uint8_t bucket_id (h:427) */
                const int32_t *s = start + (size_t)g * (nb + 1);
                for (int32_t p = s[bucket]; p < s[bucket + 1]; ++p) {
                    const int c = items[(size_t)g * n2 + p];
                    if (used[c]) continue;
                    int hd = 0;
                    for (int w = 0; w < words; ++w)
                        hd += __builtin_popcountll(h1[(size_t)i * words + w] ^ h2[(size_t)c * words + w]);
                    grp_items[(size_t)hd * n2 + grp_n[hd]++] = c;
                    used[c] = 1;
                }
            }
            /* collect_top_ranked_candidates (h:446-468) */
            int32_t top[64];
            int nt = 0;
            for (int hd = 0; hd <= dim; ++hd) {
                for (int j = 0; j < grp_n[hd]; ++j) {
                    top[nt++] = grp_items[(size_t)hd * n2 + j];
                    if (nt >= max_cand) break;
                }
                if (nt >= min_cand) break;
            }
            nn_result r;
            if (is_signed) {
                for (int j = 0; j < nt; ++j)
                    memcpy((int16_t *)tmp + (size_t)j * dim, (const int16_t *)d2 + (size_t)top[j] * dim, (size_t)dim * 2);
                oracle_nn_find_s16((const int16_t *)d1 + (size_t)i * dim, (const int16_t *)tmp, nt, dim, &r);
                if ((float)(int16_t)r.dist_1st > sq_dist) continue;
                if ((float)(int16_t)r.dist_1st / (float)(int16_t)r.dist_2nd > sq_lowe) continue;
            } else {
                for (int j = 0; j < nt; ++j)
                    memcpy((uint16_t *)tmp + (size_t)j * dim, (const uint16_t *)d2 + (size_t)top[j] * dim, (size_t)dim * 2);
                oracle_nn_find_u16((const uint16_t *)d1 + (size_t)i * dim, (const uint16_t *)tmp, nt, dim, &r);
                if ((float)(uint16_t)r.dist_1st > sq_dist) continue;
                if ((float)(uint16_t)r.dist_1st / (float)(uint16_t)r.dist_2nd > sq_lowe) continue;
            }
            /* with no candidate at all the reference indexes an empty vector here; the
             * ratio test above has rejected that case for every ratio < 1 */
            result[i] = nt > 0 ? top[r.idx_1st] : -1;
        }
        free(used); free(grp_items); free(grp_n); free(tmp);
    }
    free(start); free(items);
}
