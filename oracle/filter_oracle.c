/*
 * filter_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the two outlier filters that bracket the bundle
 * adjustment calls (src/sfm/reconstruct.cpp:212,264-265):
 *   getNearestNeighbourDistance / filterOutlierTracks
 *       (src/triangulation/outlier_filtering.cpp:14-38, 40-125)
 *   the per-feature decision of filterTracksWithReprojectionError (:127-192)
 *
 * PARITY UNPINNED: the reference file needs Eigen (absent from this image), so
 * it cannot be compiled here, and its tests hold no vectors for these
 * functions.  The arithmetic is restated operation by operation:
 *   - distances are norms of the difference of the HOMOGENEOUS 4-vectors (w
 *     included), strict '<' against a start value of 1000000 (:22-31);
 *   - Eigen evaluates squaredNorm() of a Vector4d with two 2-lane packets:
 *     (d0^2 + d2^2) + (d1^2 + d3^2); that order is kept (an un-vectorised
 *     build would sum left to right and may differ in the last bit);
 *   - mean over the tracks with a point, in track order; the standard
 *     deviation divides by the counter that KEEPS counting in the second
 *     loop (:64-94), i.e. by twice the number of points;
 *   - sigma = fmax(sigma, 1e-3), threshold mean + 1.6 sigma, tracks without a
 *     point are kept, points with a 4-vector norm > 10 dropped (:97-118).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ORACLE_API __attribute__((visibility("default")))

static double norm4(const double *d)
{
    return sqrt((d[0] * d[0] + d[2] * d[2]) + (d[1] * d[1] + d[3] * d[3]));
}

/* outlier_filtering.cpp:14-38 */
ORACLE_API void
oracle_nn_distances(const double *points, int n, double *nn)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) {
        double min_dist = 1000000;
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            double d[4];
            for (int k = 0; k < 4; ++k) d[k] = points[4 * i + k] - points[4 * j + k];
            const double dist = norm4(d);
            if (dist < min_dist) min_dist = dist;
        }
        nn[i] = min_dist;
    }
}

/* outlier_filtering.cpp:40-125; keep[t] = 1 when track t survives.
 * stats: mean, sigma (after the 1e-3 floor). */
ORACLE_API void
oracle_filter_outlier_tracks(const double *points, const uint8_t *has_point, int num_tracks,
    uint8_t *keep, double *stats)
{
    /* reduced list of the tracks with a point (:46-52) */
    int n = 0;
    int *ids = (int *)malloc(sizeof(int) * (num_tracks > 0 ? num_tracks : 1));
    double *red = (double *)calloc(4 * (size_t)(num_tracks > 0 ? num_tracks : 1), sizeof(double));
    for (int t = 0; t < num_tracks; ++t)
        if (has_point[t]) {
            for (int k = 0; k < 4; ++k) red[4 * n + k] = points[4 * t + k];
            ids[n++] = t;
        }
    double *nn = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    oracle_nn_distances(red, n, nn);
    double *dist = (double *)calloc(num_tracks > 0 ? num_tracks : 1, sizeof(double));
    for (int i = 0; i < n; ++i) dist[ids[i]] = nn[i];

    double sum = 0;
    int counter = 0;
    for (int t = 0; t < num_tracks; ++t)
        if (has_point[t]) { sum += dist[t]; counter++; }
    const double mean = sum / (double)counter;
    double sq = 0;
    for (int t = 0; t < num_tracks; ++t)
        if (has_point[t]) { sq += pow(dist[t] - mean, 2); counter++; }   /* counter keeps counting */
    double sigma = sqrt(sq / (double)counter);
    sigma = fmax(sigma, 1e-3);
    const double thr = 1.6;
    for (int t = 0; t < num_tracks; ++t) {
        if (!has_point[t]) { keep[t] = 1; continue; }
        if (norm4(points + 4 * t) > 10) { keep[t] = 0; continue; }
        keep[t] = dist[t] < mean + thr * sigma ? 1 : 0;
    }
    if (stats) { stats[0] = mean; stats[1] = sigma; }
    free(ids); free(red); free(nn); free(dist);
}
