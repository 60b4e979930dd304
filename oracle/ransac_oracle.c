/*
 * ransac_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the geometric verification step that follows descriptor
 * matching in sfm::bundler::Matching::two_view_matching
 * (src/mve/sfm/bundler_matching.cc:194-219): RANSAC over 8-point fundamental
 * matrices with Sampson-distance inliers
 * (src/mve/sfm/ransac_fundamental.cc:26-105, fundamental.cc:78-127,225-246).
 *
 * Parity: PARTLY pinned.  oracle_sampson_distance follows fundamental.cc:225-246
 * operation by operation and is checked bit-for-bit against the reference build
 * (oracle/_ref); the 8-point solve returns the same null space as the
 * reference's SVD (checked to 1e-9 up to sign/scale).  The RANSAC loop itself
 * cannot be pinned run-for-run: the reference draws its samples from
 * std::rand() (util/system.h:118-122), shared across OpenMP threads, so its
 * inlier sets differ from run to run; here the samples come from a
 * counter-based generator (splitmix64 of seed, pair id, iteration, draw) and
 * the comparison with the reference is statistical (tests/test_oracle_ransac.py).
 *
 * Numerical route (same subspaces as the reference's SVDs, different
 * arithmetic): the null vector of the 8x9 system comes from Gauss-Jordan
 * elimination with full pivoting (exact for rank 8, where the SVD's smallest
 * singular value is 0 as well), and the rank-2 projection
 * removes the smallest singular direction found by a 3x3 Jacobi
 * eigen-decomposition of F^T F instead of recomposing U S V^T.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* fundamental.cc:225-246, F row-major 9 doubles */
ORACLE_API double
oracle_sampson_distance(const double *F, const double *p1, const double *p2)
{
    double p2_F_p1 = 0.0;
    p2_F_p1 += p2[0] * (p1[0] * F[0] + p1[1] * F[1] + F[2]);
    p2_F_p1 += p2[1] * (p1[0] * F[3] + p1[1] * F[4] + F[5]);
    p2_F_p1 += 1.0 * (p1[0] * F[6] + p1[1] * F[7] + F[8]);
    p2_F_p1 *= p2_F_p1;
    double sum = 0.0, t;
    t = p1[0] * F[0] + p1[1] * F[1] + F[2]; sum += t * t;
    t = p1[0] * F[3] + p1[1] * F[4] + F[5]; sum += t * t;
    t = p2[0] * F[0] + p2[1] * F[3] + F[6]; sum += t * t;
    t = p2[0] * F[1] + p2[1] * F[4] + F[7]; sum += t * t;
    return p2_F_p1 / sum;
}

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* draw `draw` of iteration `it` of pair `pair` */
static uint64_t ransac_rand(uint64_t seed, uint64_t pair, uint64_t it, uint64_t draw)
{
    return splitmix64(splitmix64(seed ^ (pair * 0xD1342543DE82EF95ull)) + it * 0x2545F4914F6CDD1Dull + draw);
}

/* 8 distinct indices in [0, k), ascending (the reference collects them in a
 * std::set, ransac_fundamental.cc:69-76) */
static void sample8(uint64_t seed, uint64_t pair, uint64_t it, int k, int idx[8])
{
    int n = 0;
    for (uint64_t d = 0; n < 8; ++d) {
        const int v = (int)(ransac_rand(seed, pair, it, d) % (uint64_t)k);
        int dup = 0;
        for (int i = 0; i < n; ++i) dup |= idx[i] == v;
        if (!dup) idx[n++] = v;
    }
    for (int i = 1; i < 8; ++i) {      /* insertion sort */
        const int v = idx[i];
        int j = i - 1;
        while (j >= 0 && idx[j] > v) { idx[j + 1] = idx[j]; --j; }
        idx[j + 1] = v;
    }
}

/* symmetric 3x3 eigen decomposition, cyclic Jacobi, fixed schedule */
static void eig3(double A[3][3], double V[3][3], double w[3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j;
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

/* fundamental_8_point + enforce_fundamental_constraints
 * (fundamental.cc:78-127): rows of A are (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2,
 * x1, y1, 1); F = null vector (unit norm), then the smallest singular
 * direction is removed.  Returns 0 on a numerically rank-deficient sample. */
ORACLE_API int
oracle_fundamental_8_point(const double p1[8][2], const double p2[8][2], double F[9])
{
    double A[8][9];
    for (int i = 0; i < 8; ++i) {
        const double x1 = p1[i][0], y1 = p1[i][1], x2 = p2[i][0], y2 = p2[i][1];
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2;
        A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
        A[i][6] = x1; A[i][7] = y1; A[i][8] = 1.0;
    }
    /* An 8x9 system always has a null vector, and for rank 8 it is unique: the
     * right singular vector of sigma_9 = 0 that the reference reads off its SVD.
     * Gauss-Jordan elimination with full pivoting finds it directly. */
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
        if (!(best > 0.0)) { memset(F, 0, 9 * sizeof(double)); return 0; }   /* rank < 8 */
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
    /* reduced form [I | a]: null vector = (-a, 1) in permuted order, unit norm */
    double f[9], n2 = 1.0;
    for (int r = 0; r < 8; ++r) { f[colperm[r]] = -A[r][8]; n2 += A[r][8] * A[r][8]; }
    f[colperm[8]] = 1.0;
    const double invn = 1.0 / sqrt(n2);
    for (int i = 0; i < 9; ++i) f[i] *= invn;
    /* rank 2: F <- F - (F v3) v3^T, v3 = eigenvector of F^T F with the smallest eigenvalue */
    double M[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            M[i][j] = f[0 + i] * f[0 + j] + f[3 + i] * f[3 + j] + f[6 + i] * f[6 + j];
    eig3(M, V, w);
    int m = 0;
    if (w[1] < w[m]) m = 1;
    if (w[2] < w[m]) m = 2;
    const double v3[3] = { V[0][m], V[1][m], V[2][m] };
    for (int r = 0; r < 3; ++r) {
        const double fv = f[3 * r] * v3[0] + f[3 * r + 1] * v3[1] + f[3 * r + 2] * v3[2];
        for (int c = 0; c < 3; ++c) F[3 * r + c] = f[3 * r + c] - fv * v3[c];
    }
    return 1;
}

/*
 * RansacFundamental::estimate (ransac_fundamental.cc:26-60) for one view pair.
 * pos1 / pos2: normalised feature positions (FeatureSet::positions, float x,y
 * per feature); corr: k pairs (feature in view 1, feature in view 2) in the
 * order of bundler_matching.cc:176-192.  Writes the inlier ids (indices into
 * corr, ascending) and returns their count.  A later hypothesis replaces the
 * current best only when it has strictly more inliers (:47).
 */
ORACLE_API int
oracle_ransac_fundamental(const float *pos1, const float *pos2, const int32_t *corr, int k,
    int max_iterations, double threshold, uint64_t seed, uint64_t pair_id,
    int32_t *inliers, double *F_out)
{
    if (k < 8) return -1;                 /* the reference throws (:66-67) */
    const double thr2 = threshold * threshold;
    int best_count = 0;
    double bestF[9] = { 0 };
    for (int it = 0; it < max_iterations; ++it) {
        int idx[8];
        sample8(seed, pair_id, (uint64_t)it, k, idx);
        double p1[8][2], p2[8][2], F[9];
        for (int i = 0; i < 8; ++i) {
            const int a = corr[2 * idx[i]], b = corr[2 * idx[i] + 1];
            p1[i][0] = pos1[2 * a]; p1[i][1] = pos1[2 * a + 1];
            p2[i][0] = pos2[2 * b]; p2[i][1] = pos2[2 * b + 1];
        }
        if (!oracle_fundamental_8_point(p1, p2, F)) continue;
        int count = 0;
        for (int i = 0; i < k; ++i) {
            const double q1[2] = { pos1[2 * corr[2 * i]], pos1[2 * corr[2 * i] + 1] };
            const double q2[2] = { pos2[2 * corr[2 * i + 1]], pos2[2 * corr[2 * i + 1] + 1] };
            if (oracle_sampson_distance(F, q1, q2) < thr2) count++;
        }
        if (count > best_count) { best_count = count; memcpy(bestF, F, sizeof bestF); }
    }
    int n = 0;
    if (best_count > 0)
        for (int i = 0; i < k; ++i) {
            const double q1[2] = { pos1[2 * corr[2 * i]], pos1[2 * corr[2 * i] + 1] };
            const double q2[2] = { pos2[2 * corr[2 * i + 1]], pos2[2 * corr[2 * i + 1] + 1] };
            if (oracle_sampson_distance(bestF, q1, q2) < thr2) inliers[n++] = i;
        }
    if (F_out) memcpy(F_out, bestF, sizeof bestF);
    return n;
}
