// Outlier filters that bracket the bundle adjustment calls:
// the O(P^2) nearest-neighbour distance of getNearestNeighbourDistance
// (src/triangulation/outlier_filtering.cpp:14-38) as a kernel, the O(P)
// statistics of filterOutlierTracks (:40-125) on the host in the reference's
// own sequential order (their sums are order dependent).
#include <algorithm>
#include <cmath>
#include <vector>

#include "osfm_common.h"

namespace osfm {

constexpr int kNnThreads = 256;

// Thread i keeps point i in registers and walks all points through LDS tiles.
// Distances are compared squared (sqrt is monotone and correctly rounded, so
// sqrt(min d^2) equals the reference's min over sqrt(d^2) bit for bit); the sum
// of squares follows Eigen's two-lane packet order (d0^2 + d2^2) + (d1^2 + d3^2).
__global__ __launch_bounds__(kNnThreads) void
nn_distance_kernel(const double4 *__restrict__ points, int n, double *__restrict__ nn)
{
    __shared__ double4 tile[kNnThreads];
    const int i = blockIdx.x * kNnThreads + threadIdx.x;
    const double4 p = points[min(i, n - 1)];
    double best = 1000000.0 * 1000000.0;          // start value 1000000 of the reference, squared
    for (int base = 0; base < n; base += kNnThreads) {
        const int j0 = base + threadIdx.x;
        tile[threadIdx.x] = points[min(j0, n - 1)];
        __syncthreads();
        const int cnt = min(kNnThreads, n - base);
        if (base != (int)(blockIdx.x * kNnThreads)) {
            // a tile that cannot hold the thread's own point (uniform per workgroup: the
            // threads of a workgroup are the points of ONE tile): no index test
#pragma unroll 8
            for (int t = 0; t < cnt; ++t) {
                const double4 q = tile[t];
                const double d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
                const double s = (d0 * d0 + d2 * d2) + (d1 * d1 + d3 * d3);
                best = s < best ? s : best;                  // strict '<' (:29)
            }
        } else {
#pragma unroll 8
            for (int t = 0; t < cnt; ++t) {
                const double4 q = tile[t];
                const double d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
                const double s = (d0 * d0 + d2 * d2) + (d1 * d1 + d3 * d3);
                // the point itself is skipped by index, duplicates are not
                if (base + t != i && s < best) best = s;
            }
        }
        __syncthreads();
    }
    if (i < n) nn[i] = sqrt(best);
}

int nn_distances_device(int device, const double *points, int n, double *nn_out)
{
    if (n <= 0) return OSFM_OK;
    OSFM_HIP_CHECK(hipSetDevice(device));
    StreamLease lease;                       // pooled: creating a stream costs ~1 ms
    OSFM_RETURN_IF(lease.acquire());
    hipStream_t s = lease.s;
    PooledBuffer d_pts, d_nn;
    int rc = d_pts.reserve((size_t)n * 32);
    if (rc == OSFM_OK) rc = d_nn.reserve((size_t)n * 8);
    if (rc == OSFM_OK) {
        hipError_t e = hipMemcpyAsync(d_pts.ptr, points, (size_t)n * 32, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(nn_distance_kernel, dim3((n + kNnThreads - 1) / kNnThreads), dim3(kNnThreads), 0, s,
                d_pts.as<double4>(), n, d_nn.as<double>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(nn_out, d_nn.ptr, (size_t)n * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("nn_distances: %s", hipGetErrorString(e)); rc = OSFM_E_DEVICE; }
    }
    d_pts.release(); d_nn.release();
    return rc;
}

static double norm4(const double *d)
{
    return std::sqrt((d[0] * d[0] + d[2] * d[2]) + (d[1] * d[1] + d[3] * d[3]));
}

}  // namespace osfm

using namespace osfm;

extern "C" {

int osfm_nn_distances(int device, const double *points, int32_t num_points, double *nn)
{
    if (num_points < 0 || (num_points > 0 && (!points || !nn))) {
        set_error("nn_distances: null array / negative count");
        return OSFM_E_ARG;
    }
    return nn_distances_device(device, points, num_points, nn);
}

int osfm_filter_outlier_tracks(int device, const double *points, const uint8_t *has_point,
    int32_t num_tracks, uint8_t *keep, osfm_outlier_stats *stats)
{
    if (num_tracks < 0 || (num_tracks > 0 && (!points || !has_point || !keep))) {
        set_error("filter_outlier_tracks: null array / negative count");
        return OSFM_E_ARG;
    }
    // reduced list of the tracks that have a point (:46-52)
    std::vector<double> red;
    std::vector<int> ids;
    for (int t = 0; t < num_tracks; ++t)
        if (has_point[t]) {
            red.insert(red.end(), points + 4 * (size_t)t, points + 4 * (size_t)t + 4);
            ids.push_back(t);
        }
    const int n = (int)ids.size();
    std::vector<double> nn((size_t)std::max(n, 1));
    OSFM_RETURN_IF(nn_distances_device(device, red.data(), n, nn.data()));
    std::vector<double> dist((size_t)std::max(num_tracks, 1), 0.0);
    for (int i = 0; i < n; ++i) dist[ids[i]] = nn[i];

    // mean and sigma exactly as written (:60-98): sequential sums in track order,
    // the counter keeps counting in the second loop
    double sum = 0;
    int counter = 0;
    for (int t = 0; t < num_tracks; ++t)
        if (has_point[t]) { sum += dist[t]; counter++; }
    const double mean = sum / (double)counter;
    double sq = 0;
    for (int t = 0; t < num_tracks; ++t)
        if (has_point[t]) { const double d = dist[t] - mean; sq += d * d; counter++; }
    double sigma = std::sqrt(sq / (double)counter);
    sigma = std::fmax(sigma, 1e-3);
    const double sigma_threshold = 1.6;
    int kept = 0;
    for (int t = 0; t < num_tracks; ++t) {
        if (!has_point[t]) keep[t] = 1;
        else if (norm4(points + 4 * (size_t)t) > 10) keep[t] = 0;
        else keep[t] = dist[t] < mean + sigma_threshold * sigma ? 1 : 0;
        kept += keep[t];
    }
    if (stats) {
        stats->mean = mean;
        stats->sigma = sigma;
        stats->num_with_point = n;
        stats->num_kept = kept;
    }
    return OSFM_OK;
}

}  // extern "C"
