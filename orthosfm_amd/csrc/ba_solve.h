// The bundle-adjustment solve on device-resident arrays: what osfm_ba_solve (host arrays in, host arrays out)
// and the device-resident scene (scene_api.hip: the arrays never leave the device) both run.
#pragma once
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "ba_kernels.h"
#include "osfm_common.h"

namespace osfm {

// a work array of one call, from the pool of osfm_common.h
struct DevArray {
    void *ptr = nullptr;
    DevArray() = default;
    DevArray(const DevArray &) = delete;
    DevArray &operator=(const DevArray &) = delete;
    ~DevArray() { if (ptr) pool_release(ptr); }
    int alloc(size_t bytes)
    {
        if (ptr) { pool_release(ptr); ptr = nullptr; }
        return g_device_pool.alloc(&ptr, std::max<size_t>(bytes, 16));
    }
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

template <typename T>
int upload(DevArray &d, const T *src, size_t n, hipStream_t s)
{
    OSFM_RETURN_IF(d.alloc(n * sizeof(T)));
    if (n) OSFM_HIP_CHECK(hipMemcpyAsync(d.ptr, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return OSFM_OK;
}

// tangent layout of the camera blocks (host side; pt_start only where the observations are the caller's)
struct Layout {
    std::vector<int32_t> cam_ldim, cam_off, pt_start;
    std::vector<int8_t> colmap;
    int nc = 0;
};

// An elimination order of the reduced camera system for banded (ring / strip) visibility: ba_order.hip
struct ReducedOrder {
    bool active = false;
    int span = 0;                             // unknowns of the laid-out system, interior padding included
    int nblk = 0;                             // blocks of 32 of it
    int arcs = 0, sep_cams = 0;               // K arcs, separators of that many cameras
    int chain_natural = 0, chain_ordered = 0; // longest chain of dependent diagonal blocks, before / after
    std::vector<int32_t> cam_off;             // [C]
    std::vector<int32_t> pad;                 // padding unknowns below span (identity rows)
    std::vector<unsigned long long> nz;       // FlowPattern::nz
    std::vector<int32_t> ptiles;              // FlowPattern::ptiles
};
// pairs: the camera pairs (a >= b, a == b included or not) that share a track; ldim: unknowns per camera.  False
// (out->active == false, chain_natural filled in): the natural order stands.
bool choose_reduced_order(int C, const int32_t *ldim, const std::vector<std::pair<int, int>> &pairs, ReducedOrder *out);

struct DeviceProblem {
    DevArray cams[2], points[2], obs_xy, obs_cam, obs_pt, pt_start, img_w, img_h;
    DevArray cam_ldim, cam_off, colmap, scale_c, scale_p;
    DevArray camder[2];       // the cameras' derived table rows, per iterate buffer (ba_solve_core)
    BaDev dev;
};

// which columns of a camera block are free (SetupParameterBlocks, OrthoQuaternionRecoAlgorithm.cpp:121-148,
// OrthographicReconstructionAlgorithm.cpp:148-178), from the constancy masks [C][7]
void build_camera_layout(int model, int C, const uint8_t *cam_const, Layout *L);
// D->cam_ldim / cam_off / colmap / scale_c from L (L outlives the queued copies: the caller's concern)
int upload_camera_layout(const Layout &L, int C, hipStream_t s, DeviceProblem *D);
// D->dev from the arrays D holds (cams[0], points[0], obs_*, pt_start, img_*, the camera layout); allocates the
// candidate buffers, the point scales and the per-observation point index
int finish_device_problem(int model, int C, int M, int O, int nc, double huber, int pdim, hipStream_t s, DeviceProblem *D);

// The Levenberg-Marquardt solve on a finished DeviceProblem (start values in cams[0] / points[0]).  pair_bound: an
// upper bound of the Schur pair entries (sum of squared track lengths), refused beyond 2^31 - 1.  On return
// *cur names the buffer pair (cams[cur], points[cur]) that holds the result; the stream is synchronised.
int ba_solve_core(DeviceProblem &D, const osfm_ba_options &o, StreamLease &sg, int64_t pair_bound, osfm_ba_summary *sum, int *cur);

int select_device(int device);

}  // namespace osfm
