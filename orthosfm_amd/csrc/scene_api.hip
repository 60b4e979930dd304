// Device-resident scene of the incremental reconstruction (runPoseEstimation, src/sfm/reconstruct.cpp:193-281).
//
// The reference keeps a std::vector<Track> and hands filtered copies of it to every step of its loop: per camera
// group a reprojection filter and a 3-camera bundle adjustment on a re-triangulated copy, a triangulation of all
// tracks, every third group a global adjustment and two outlier filters.  Through the per-call entries of the C
// ABI (osfm_ba_solve, osfm_ba_triangulate, osfm_filter_reprojection) every one of those ~530 calls of a 200-view
// job flattens its tracks on the host and uploads them -- the device worked for a quarter of the pose
// estimation's wall time.  Here the track table is uploaded ONCE: features in track order (view, pixel position),
// alive flags per feature and per track, point and hasPoint() per track, the aligned cameras; a step is a handful
// of kernels that select its observations from the flags (a scan over the features and a scatter), run the same
// triangulation / reprojection / LM kernels on the compacted arrays and write flags, points and cameras back.
// Nothing but a few counters, the cameras of a step and the LM summary crosses PCIe.  What a filter of the
// reference removes from its list is a cleared flag here.  The table is compacted when a filter has left fewer than
// half of its tracks alive (scene_compact): every step passes over ALL features a dozen times, and after the first
// global round ~5 % of a large job's table is alive (200 views: 3.2 M features, 500 views: 8 M).  Maps back to the
// caller's numbering serve osfm_scene_download / osfm_scene_set_flags.
//
// Semantics are those of orthosfm_amd/pipeline.py's per-call form (which stays, behind use_scene = False, and is
// what tests/test_e2e_gpu.py compares this against bit for bit): see the entry points below.
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "ba_solve.h"

using namespace osfm;

struct osfm_scene {
    int device = 0, model = 0, V = 0, T = 0;
    int64_t F = 0;
    std::vector<int32_t> img_w, img_h;                    // per view
    DeviceBuffer feat_view, feat_xy, track_of, offsets;   // int32 [F], double [F][2], int32 [F], int32 [T + 1]
    DeviceBuffer alive_f, alive_t, has_point, point;      // uint8 [F], uint8 [T], uint8 [T], double [T][4]
    DeviceBuffer cam_of_view, cams;                       // int32 [V] (-1: no camera), double [V][7] by camera index
    std::vector<int32_t> aligned;                         // view of every camera, in the order they joined
    std::vector<double> h_cams;                           // [V][7] by camera index (mirror of cams)
    std::vector<uint8_t> h_const;                         // [V][7] by camera index
    // scratch, grow-only
    DeviceBuffer cam_map, sel, scan, cnt, tflag, tslot, aux_f, aux_t, aux_t2, cub_temp, counters, tmp_hp, tmp_point;
    // after a compaction: the caller's numbering (T0 tracks, F0 features) and where every kept track / feature was
    int T0 = 0;
    int64_t F0 = 0;
    bool compacted = false;
    bool cams_moved = false;                              // osfm_scene_set_cameras since the last full triangulation
    DeviceBuffer orig_t, orig_f;                          // int32 [T], int32 [F] (valid when compacted)
    std::mutex mu;
};

namespace {

constexpr int kThreads = 256;
inline int blocks_for(int64_t n) { return (int)std::max<int64_t>(1, (n + kThreads - 1) / kThreads); }

// scene_create: the features' positions widened, the track of every feature
__global__ void scene_widen_xy_kernel(int64_t n, const float *__restrict__ in, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
__global__ void scene_expand_tracks_kernel(int T, const int32_t *__restrict__ offsets, int32_t *__restrict__ track_of)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    for (int f = offsets[t]; f < offsets[t + 1]; ++f) track_of[f] = t;
}

// sel[f] = 1: a live feature (alive, of an alive track), of a track of the mask (if any), whose view maps to a camera
__global__ void scene_select_kernel(int64_t F, const int32_t *__restrict__ view, const int32_t *__restrict__ track_of,
    const uint8_t *__restrict__ alive_f, const uint8_t *__restrict__ alive_t, const int32_t *__restrict__ cam_map,
    const uint8_t *__restrict__ tmask, int32_t *__restrict__ sel)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f > F) return;
    if (f == F) { sel[F] = 0; return; }                   // the scan's last entry is the total
    const int t = track_of[f];
    sel[f] = (alive_f[f] && alive_t[t] && (!tmask || tmask[t]) && (!cam_map || cam_map[view[f]] >= 0)) ? 1 : 0;
}

// cnt[t] = selected features of track t (features of a track are neighbours: a difference of the scan)
__global__ void scene_track_count_kernel(int T, const int32_t *__restrict__ offsets, const int32_t *__restrict__ scan,
    int32_t *__restrict__ cnt)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) cnt[t] = scan[offsets[t + 1]] - scan[offsets[t]];
}

// tracks that see a view of the set (live features only): tmask[t] = 1 (all writers store 1)
__global__ void scene_touch_kernel(int64_t F, const int32_t *__restrict__ view, const int32_t *__restrict__ track_of,
    const uint8_t *__restrict__ alive_f, const uint8_t *__restrict__ alive_t, const int32_t *__restrict__ view_flag,
    uint8_t *__restrict__ tmask)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int t = track_of[f];
    if (alive_f[f] && alive_t[t] && view_flag[view[f]] >= 0) tmask[t] = 1;
}

// tflag[t] from the counts / masks of an operation (see the callers), with the scan's extra last entry
enum { kFlagCountPositive = 0, kFlagCountEquals = 1, kFlagAliveWithPoint = 2, kFlagAlive = 3 };
__global__ void scene_track_flag_kernel(int T, int mode, int n, const int32_t *__restrict__ cnt, const uint8_t *__restrict__ alive_t,
    const uint8_t *__restrict__ has_point, int32_t *__restrict__ tflag)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > T) return;
    if (t == T) { tflag[T] = 0; return; }
    int v = 0;
    if (mode == kFlagCountPositive) v = cnt[t] > 0;
    else if (mode == kFlagCountEquals) v = cnt[t] == n;
    else if (mode == kFlagAliveWithPoint) v = alive_t[t] && has_point[t];
    else v = alive_t[t];
    tflag[t] = v;
}

// observations of the selected features in feature (= track) order
__global__ void scene_scatter_obs_kernel(int64_t F, const int32_t *__restrict__ sel, const int32_t *__restrict__ scan,
    const int32_t *__restrict__ view, const double2 *__restrict__ xy, const int32_t *__restrict__ cam_map,
    double2 *__restrict__ obs_xy, int32_t *__restrict__ obs_cam)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F || !sel[f]) return;
    const int k = scan[f];
    obs_xy[k] = xy[f];
    obs_cam[k] = cam_map[view[f]];
}

// points of the flagged tracks: pt_start (CSR over the observations), the track of every point, its start value
// (the scene's point, or (0, 0, 0, 1))
__global__ void scene_points_kernel(int T, int M, int O, const int32_t *__restrict__ offsets, const int32_t *__restrict__ scan,
    const int32_t *__restrict__ tflag, const int32_t *__restrict__ tslot, const double *__restrict__ point_in,
    int32_t *__restrict__ pt_start, int32_t *__restrict__ point_track, double *__restrict__ points)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) pt_start[M] = O;
    if (t >= T || !tflag[t]) return;
    const int j = tslot[t];
    pt_start[j] = scan[offsets[t]];
    point_track[j] = t;
    if (point_in) { for (int i = 0; i < 4; ++i) points[4 * j + i] = point_in[4 * (size_t)t + i]; }
    else { points[4 * j] = 0.0; points[4 * j + 1] = 0.0; points[4 * j + 2] = 0.0; points[4 * j + 3] = 1.0; }
}

// sum of squared track lengths (the bound of the Schur pair lists)
__global__ void scene_pair_bound_kernel(int M, const int32_t *__restrict__ pt_start, unsigned long long *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (j < M) { const unsigned long long l = (unsigned long long)(pt_start[j + 1] - pt_start[j]); v = l * l; }
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// triangulation results back into the scene: tracks of the mask (all tracks when clear_all) lose their point, the
// valid intersections set it (triangulation.cpp:76-91)
__global__ void scene_store_points_kernel(int T, const uint8_t *__restrict__ tmask, int clear_all, const int32_t *__restrict__ tflag,
    const int32_t *__restrict__ tslot, const uint8_t *__restrict__ valid, const double *__restrict__ pts,
    uint8_t *__restrict__ has_point, double *__restrict__ point)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    if (!clear_all && !(tmask && tmask[t])) return;
    bool hp = false;
    if (tflag[t]) {
        const int j = tslot[t];
        if (valid[j]) {
            hp = true;
            for (int i = 0; i < 4; ++i) point[4 * (size_t)t + i] = pts[4 * (size_t)j + i];
        }
    }
    has_point[t] = hp ? 1 : 0;
}

// points of an adjustment back to their tracks
__global__ void scene_scatter_points_kernel(int M, const int32_t *__restrict__ point_track, const double *__restrict__ pts,
    double *__restrict__ point)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int t = point_track[j];
    for (int i = 0; i < 4; ++i) point[4 * (size_t)t + i] = pts[4 * (size_t)j + i];
}

// reprojection filter, per feature of the camera set: keep unless its track is seen by all cameras and the feature
// reprojects max_error pixels or more away from the re-triangulated point (outlier_filtering.cpp:140-176)
__global__ void scene_keep_kernel(int64_t F, const int32_t *__restrict__ in_set, const int32_t *__restrict__ sel_full,
    const int32_t *__restrict__ scan_full, const double *__restrict__ err, double max_error, int32_t *__restrict__ keep)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f > F) return;
    if (f == F) { keep[F] = 0; return; }
    int k = 0;
    if (in_set[f]) k = sel_full[f] ? (err[scan_full[f]] < max_error ? 1 : 0) : 1;
    keep[f] = k;
}

// per track of the camera set: does it stay?  Only tracks seen by ALL cameras are judged: what they keep inside
// the set plus what they have outside it must be more than one feature (:147-149, 187-189)
__global__ void scene_track_ok_kernel(int T, int n, const int32_t *__restrict__ cnt_set, const int32_t *__restrict__ cnt_keep,
    const int32_t *__restrict__ cnt_live, uint8_t *__restrict__ ok)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    ok[t] = cnt_set[t] == n ? (cnt_live[t] - cnt_set[t] + cnt_keep[t] > 1 ? 1 : 0) : 1;
}

// the filter made permanent: judged tracks that do not stay and features that were dropped lose their flags
__global__ void scene_kill_tracks_kernel(int T, const int32_t *__restrict__ cnt_set, const uint8_t *__restrict__ ok, uint8_t *__restrict__ alive_t)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && cnt_set[t] > 0 && !ok[t]) alive_t[t] = 0;
}
__global__ void scene_kill_features_kernel(int64_t F, const int32_t *__restrict__ in_set, const int32_t *__restrict__ keep, uint8_t *__restrict__ alive_f)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F && in_set[f] && !keep[f]) alive_f[f] = 0;
}

// what the adjustment behind the filter works on: kept features of tracks that stay and keep more than one
// feature inside the camera set
__global__ void scene_select_ba_kernel(int64_t F, const int32_t *__restrict__ in_set, const int32_t *__restrict__ keep,
    const int32_t *__restrict__ track_of, const uint8_t *__restrict__ ok, const int32_t *__restrict__ cnt_keep, int32_t *__restrict__ sel)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f > F) return;
    if (f == F) { sel[F] = 0; return; }
    const int t = track_of[f];
    sel[f] = (in_set[f] && keep[f] && ok[t] && cnt_keep[t] > 1) ? 1 : 0;
}

__global__ void scene_to_byte_kernel(int T, const int32_t *__restrict__ in, uint8_t *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = in[t] ? 1 : 0;
}

__global__ void scene_fill_i32_kernel(int64_t n, int32_t value, int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = value;
}
__global__ void scene_set_map_kernel(int n, const int32_t *__restrict__ views, int32_t first, int32_t *__restrict__ map)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) map[views[i]] = first + i;
}

// the incremental triangulation against a full pass: hasPoint() of every track, the point of every track that has one
__global__ void scene_compare_kernel(int T, const uint8_t *__restrict__ hp_a, const double *__restrict__ pt_a,
    const uint8_t *__restrict__ hp_b, const double *__restrict__ pt_b, int32_t *__restrict__ mismatches)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    bool bad = (hp_a[t] != 0) != (hp_b[t] != 0);
    if (!bad && hp_b[t])
        for (int i = 0; i < 4; ++i)
            bad |= __double_as_longlong(pt_a[4 * (size_t)t + i]) != __double_as_longlong(pt_b[4 * (size_t)t + i]);
    if (bad) atomicAdd(mismatches, 1);
}

int exclusive_scan(osfm_scene *sc, const int32_t *in, int32_t *out, int64_t n, hipStream_t s)
{
    size_t need = 0;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)n, s));
    OSFM_RETURN_IF(sc->cub_temp.reserve(need + 256));
    size_t have = sc->cub_temp.bytes;
    OSFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(sc->cub_temp.ptr, have, in, out, (int)n, s));
    return OSFM_OK;
}

// last entry of a scan over n + 1 flags = number of flags set
int read_total(const int32_t *scan, int64_t n, hipStream_t s, int *out)
{
    int32_t v = 0;
    OSFM_HIP_CHECK(hipMemcpyAsync(&v, scan + n, 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    *out = v;
    return OSFM_OK;
}

// two of them behind ONE synchronisation
int read_totals2(const int32_t *scan_a, int64_t na, const int32_t *scan_b, int64_t nb, hipStream_t s, int *out_a, int *out_b)
{
    int32_t v[2] = { 0, 0 };
    OSFM_HIP_CHECK(hipMemcpyAsync(&v[0], scan_a + na, 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipMemcpyAsync(&v[1], scan_b + nb, 4, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    *out_a = v[0]; *out_b = v[1];
    return OSFM_OK;
}

int reserve_scratch(osfm_scene *sc)
{
    const size_t F1 = (size_t)sc->F + 1, T1 = (size_t)sc->T + 1;
    OSFM_RETURN_IF(sc->cam_map.reserve((size_t)std::max(sc->V, 1) * 4));
    OSFM_RETURN_IF(sc->sel.reserve(F1 * 4));
    OSFM_RETURN_IF(sc->scan.reserve(F1 * 4));
    OSFM_RETURN_IF(sc->aux_f.reserve(F1 * 4 * 4));          // four more per-feature int32 arrays (the filter)
    OSFM_RETURN_IF(sc->cnt.reserve(T1 * 4));
    OSFM_RETURN_IF(sc->tflag.reserve(T1 * 4));
    OSFM_RETURN_IF(sc->tslot.reserve(T1 * 4));
    OSFM_RETURN_IF(sc->aux_t.reserve(T1 * 4 * 3));
    OSFM_RETURN_IF(sc->aux_t2.reserve(T1));
    OSFM_RETURN_IF(sc->counters.reserve(64));
    return OSFM_OK;
}

// ---- a compacted problem: observations of sel (scan = its exclusive sum, O of them), points = flagged tracks ----
struct Compact {
    DeviceProblem D;
    DevArray point_track;
    int O = 0, M = 0, C = 0;
    DevArray cam_block;              // image sizes and tangent layout of the cameras: one block, one copy
    std::vector<int32_t> w;          // ... and its host image: the source of a copy that may still be queued
};

// sel / scan: per-feature flags and their scan; tflag: per-track point flags (T + 1 entries); point_in: start
// values by track (nullptr: (0, 0, 0, 1)); cameras: C x 7 on the host (h_cams) or on the device (d_cams).
// known_M >= 0: the caller has scanned tflag into sc->tslot and read its total already (with another count,
// behind one synchronisation).  Nothing is synchronised here: *out, *L, h_cams and h_const have to outlive the
// stream's next synchronisation.
int build_problem(osfm_scene *sc, const int32_t *sel, const int32_t *scan, int O, const int32_t *tflag,
    const double *point_in, int C, const double *h_cams, const double *d_cams, const uint8_t *h_const,
    const int32_t *cam_views, const int32_t *cam_map, double huber, int pdim, Layout *L, hipStream_t s, Compact *out,
    int known_M = -1)
{
    int32_t *tslot = sc->tslot.as<int32_t>();
    int M = known_M;
    if (M < 0) {
        OSFM_RETURN_IF(exclusive_scan(sc, tflag, tslot, (int64_t)sc->T + 1, s));
        OSFM_RETURN_IF(read_total(tslot, sc->T, s, &M));
    }
    out->O = O; out->M = M; out->C = C;
    DeviceProblem &D = out->D;
    OSFM_RETURN_IF(D.obs_xy.alloc((size_t)std::max(O, 1) * 16));
    OSFM_RETURN_IF(D.obs_cam.alloc((size_t)std::max(O, 1) * 4));
    OSFM_RETURN_IF(D.pt_start.alloc((size_t)(M + 1) * 4));
    OSFM_RETURN_IF(D.points[0].alloc((size_t)std::max(M, 1) * 32));
    OSFM_RETURN_IF(out->point_track.alloc((size_t)std::max(M, 1) * 4));
    OSFM_RETURN_IF(D.cams[0].alloc((size_t)std::max(C, 1) * 56));
    if (O) hipLaunchKernelGGL(scene_scatter_obs_kernel, dim3(blocks_for(sc->F)), dim3(kThreads), 0, s, sc->F, sel, scan,
        sc->feat_view.as<int32_t>(), sc->feat_xy.as<double2>(), cam_map, D.obs_xy.as<double2>(), D.obs_cam.as<int32_t>());
    hipLaunchKernelGGL(scene_points_kernel, dim3(blocks_for(sc->T)), dim3(kThreads), 0, s, sc->T, M, O, sc->offsets.as<int32_t>(),
        scan, tflag, tslot, point_in, D.pt_start.as<int32_t>(), out->point_track.as<int32_t>(), D.points[0].as<double>());
    if (h_cams) OSFM_HIP_CHECK(hipMemcpyAsync(D.cams[0].ptr, h_cams, (size_t)C * 56, hipMemcpyHostToDevice, s));
    else if (C) OSFM_HIP_CHECK(hipMemcpyAsync(D.cams[0].ptr, d_cams, (size_t)C * 56, hipMemcpyDeviceToDevice, s));
    // the cameras' small tables -- image sizes, tangent layout -- in ONE block and one copy (five uploads of a few
    // dozen bytes each were 50 us of host time per problem, two problems per local adjustment)
    build_camera_layout(sc->model, C, h_const, L);
    std::vector<int32_t> &pack = out->w;
    const size_t Cz = (size_t)std::max(C, 1);
    pack.assign(4 * Cz + (6 * Cz + 3) / 4, 0);
    for (int c = 0; c < C; ++c) {
        pack[c] = sc->img_w[cam_views[c]]; pack[Cz + c] = sc->img_h[cam_views[c]];
        pack[2 * Cz + c] = L->cam_ldim[c]; pack[3 * Cz + c] = L->cam_off[c];
    }
    memcpy(reinterpret_cast<int8_t *>(pack.data() + 4 * Cz), L->colmap.data(), (size_t)6 * C);
    OSFM_RETURN_IF(upload(out->cam_block, pack.data(), pack.size(), s));
    OSFM_RETURN_IF(D.scale_c.alloc((size_t)L->nc * 8));
    launch_fill(D.scale_c.as<double>(), (size_t)L->nc, 1.0, s);
    OSFM_RETURN_IF(finish_device_problem(sc->model, C, M, O, L->nc, huber, pdim, s, &D));
    const int32_t *blk = out->cam_block.as<int32_t>();
    D.dev.img_w = blk; D.dev.img_h = blk + Cz; D.dev.cam_ldim = blk + 2 * Cz; D.dev.cam_off = blk + 3 * Cz;
    D.dev.cam_colmap = reinterpret_cast<const int8_t *>(blk + 4 * Cz);
    return OSFM_OK;
}

int check_views(const osfm_scene *sc, const int32_t *views, int n, const char *what)
{
    if (n < 0 || (n > 0 && !views)) { set_error("%s: bad view list", what); return OSFM_E_ARG; }
    for (int i = 0; i < n; ++i)
        if (views[i] < 0 || views[i] >= sc->V) { set_error("%s: view %d out of range [0,%d)", what, views[i], sc->V); return OSFM_E_ARG; }
    return OSFM_OK;
}

// cam_map[v] = i for views[i], -1 elsewhere
int set_cam_map(osfm_scene *sc, const int32_t *views, int n, hipStream_t s, DevArray *d_views)
{
    hipLaunchKernelGGL(scene_fill_i32_kernel, dim3(blocks_for(sc->V)), dim3(kThreads), 0, s, (int64_t)sc->V, -1, sc->cam_map.as<int32_t>());
    if (n == 0) return OSFM_OK;
    OSFM_RETURN_IF(upload(*d_views, views, (size_t)n, s));
    hipLaunchKernelGGL(scene_set_map_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, n, d_views->as<int32_t>(), 0, sc->cam_map.as<int32_t>());
    return OSFM_OK;
}

// one triangulation pass over the tracks of tmask (nullptr: all alive tracks) into (hp_out, pt_out)
int triangulate_pass(osfm_scene *sc, const uint8_t *tmask, bool clear_all, uint8_t *hp_out, double *pt_out, hipStream_t s)
{
    const int64_t F = sc->F;
    const int T = sc->T, C = (int)sc->aligned.size();
    int32_t *sel = sc->sel.as<int32_t>(), *scan = sc->scan.as<int32_t>(), *cnt = sc->cnt.as<int32_t>(), *tflag = sc->tflag.as<int32_t>();
    hipLaunchKernelGGL(scene_select_kernel, dim3(blocks_for(F + 1)), dim3(kThreads), 0, s, F, sc->feat_view.as<int32_t>(),
        sc->track_of.as<int32_t>(), sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), sc->cam_of_view.as<int32_t>(), tmask, sel);
    OSFM_RETURN_IF(exclusive_scan(sc, sel, scan, F + 1, s));
    hipLaunchKernelGGL(scene_track_count_kernel, dim3(blocks_for(T)), dim3(kThreads), 0, s, T, sc->offsets.as<int32_t>(), scan, cnt);
    hipLaunchKernelGGL(scene_track_flag_kernel, dim3(blocks_for(T + 1)), dim3(kThreads), 0, s, T, (int)kFlagCountPositive, 0, cnt,
        nullptr, nullptr, tflag);
    int O = 0, M = 0;
    OSFM_RETURN_IF(exclusive_scan(sc, tflag, sc->tslot.as<int32_t>(), (int64_t)T + 1, s));
    OSFM_RETURN_IF(read_totals2(scan, F, sc->tslot.as<int32_t>(), T, s, &O, &M));
    Compact P;
    Layout L;
    OSFM_RETURN_IF(build_problem(sc, sel, scan, O, tflag, nullptr, C, nullptr, sc->cams.as<double>(), sc->h_const.data(),
        sc->aligned.data(), sc->cam_of_view.as<int32_t>(), 1.0, 3, &L, s, &P, M));
    DevArray valid;
    OSFM_RETURN_IF(valid.alloc((size_t)std::max(P.M, 1)));
    if (P.M) {
        // tracks with fewer than two rays keep their start value and are flagged invalid
        OSFM_HIP_CHECK(hipMemcpyAsync(P.D.points[1].ptr, P.D.points[0].ptr, (size_t)P.M * 32, hipMemcpyDeviceToDevice, s));
        launch_triangulate(P.D.dev, P.D.points[1].as<double>(), valid.as<uint8_t>(), s);
    }
    hipLaunchKernelGGL(scene_store_points_kernel, dim3(blocks_for(T)), dim3(kThreads), 0, s, T, tmask, clear_all ? 1 : 0, tflag,
        sc->tslot.as<int32_t>(), valid.as<uint8_t>(), P.D.points[1].as<double>(), hp_out, pt_out);
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    return OSFM_OK;
}

// filterTracksWithReprojectionError on the scene for the cameras (views, params, const masks); permanent: the flags
// are cleared.  Leaves sel / scan (the features the adjustment behind it works on) and *num_selected.
int reprojection_filter(osfm_scene *sc, int n, const int32_t *views, const double *params, const uint8_t *cconst,
    double max_error, bool permanent, hipStream_t s, int *num_selected, int *num_tracks_selected = nullptr)
{
    const int64_t F = sc->F;
    const int T = sc->T;
    int32_t *in_set = sc->aux_f.as<int32_t>(), *sel_full = in_set + (F + 1), *keep = sel_full + (F + 1), *scan2 = keep + (F + 1);
    int32_t *sel = sc->sel.as<int32_t>(), *scan = sc->scan.as<int32_t>();
    int32_t *cnt_set = sc->cnt.as<int32_t>(), *tflag = sc->tflag.as<int32_t>();
    int32_t *cnt_keep = sc->aux_t.as<int32_t>(), *cnt_live = cnt_keep + (T + 1), *cnt_tmp = cnt_live + (T + 1);
    uint8_t *ok = sc->aux_t2.as<uint8_t>();
    DevArray d_views;
    OSFM_RETURN_IF(set_cam_map(sc, views, n, s, &d_views));
    const int32_t *cam_map = sc->cam_map.as<int32_t>();
    const dim3 gF(blocks_for(F + 1)), gT(blocks_for(T + 1)), b(kThreads);
    // live features of the camera set, per track
    hipLaunchKernelGGL(scene_select_kernel, gF, b, 0, s, F, sc->feat_view.as<int32_t>(), sc->track_of.as<int32_t>(),
        sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), cam_map, nullptr, in_set);
    OSFM_RETURN_IF(exclusive_scan(sc, in_set, scan, F + 1, s));
    hipLaunchKernelGGL(scene_track_count_kernel, gT, b, 0, s, T, sc->offsets.as<int32_t>(), scan, cnt_set);
    // all live features, per track (what a track has outside the set counts towards "more than one left")
    hipLaunchKernelGGL(scene_select_kernel, gF, b, 0, s, F, sc->feat_view.as<int32_t>(), sc->track_of.as<int32_t>(),
        sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), nullptr, nullptr, sel);
    OSFM_RETURN_IF(exclusive_scan(sc, sel, scan2, F + 1, s));
    hipLaunchKernelGGL(scene_track_count_kernel, gT, b, 0, s, T, sc->offsets.as<int32_t>(), scan2, cnt_live);
    // tracks seen by ALL cameras: re-triangulated from them, every feature's reprojection error
    hipLaunchKernelGGL(scene_track_flag_kernel, gT, b, 0, s, T, (int)kFlagCountEquals, n, cnt_set, nullptr, nullptr, tflag);
    {
        // sel_full[f] = in_set[f] && full(track): the select kernel with the track flags as a byte mask
        uint8_t *fullmask = ok;           // (ok is written further down)
        (void)cnt_tmp;
        hipLaunchKernelGGL(scene_to_byte_kernel, gT, b, 0, s, T, tflag, fullmask);
        hipLaunchKernelGGL(scene_select_kernel, gF, b, 0, s, F, sc->feat_view.as<int32_t>(), sc->track_of.as<int32_t>(),
            sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), cam_map, fullmask, sel_full);
    }
    OSFM_RETURN_IF(exclusive_scan(sc, sel_full, scan2, F + 1, s));
    int O1 = 0, M1 = 0;
    OSFM_RETURN_IF(exclusive_scan(sc, tflag, sc->tslot.as<int32_t>(), (int64_t)T + 1, s));
    OSFM_RETURN_IF(read_totals2(scan2, F, sc->tslot.as<int32_t>(), T, s, &O1, &M1));
    DevArray err;
    OSFM_RETURN_IF(err.alloc((size_t)std::max(O1, 1) * 8));
    // (P's arrays go back to the pool when this function returns: behind the synchronisation at its end)
    Compact P;
    Layout L;
    if (O1 > 0) {
        OSFM_RETURN_IF(build_problem(sc, sel_full, scan2, O1, tflag, sc->point.as<double>(), n, params, nullptr, cconst, views, cam_map,
            1.0, 3, &L, s, &P, M1));
        // triangulateTracks(cameras, fullSizeTracks, true) + evaluateReprojectionError per feature (osfm_filter_reprojection)
        OSFM_HIP_CHECK(hipMemcpyAsync(P.D.points[1].ptr, P.D.points[0].ptr, (size_t)P.M * 32, hipMemcpyDeviceToDevice, s));
        launch_triangulate(P.D.dev, P.D.points[1].as<double>(), nullptr, s);
        OSFM_HIP_CHECK(hipMemcpyAsync(P.D.points[0].ptr, P.D.points[1].ptr, (size_t)P.M * 32, hipMemcpyDeviceToDevice, s));
        launch_reproj(P.D.dev, err.as<double>(), nullptr, s);
        OSFM_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(scene_keep_kernel, gF, b, 0, s, F, in_set, sel_full, scan2, err.as<double>(), max_error, keep);
    OSFM_RETURN_IF(exclusive_scan(sc, keep, scan, F + 1, s));
    hipLaunchKernelGGL(scene_track_count_kernel, gT, b, 0, s, T, sc->offsets.as<int32_t>(), scan, cnt_keep);
    hipLaunchKernelGGL(scene_track_ok_kernel, gT, b, 0, s, T, n, cnt_set, cnt_keep, cnt_live, ok);
    // the adjustment's selection (taken before a permanent filter clears flags: it reads none of them)
    hipLaunchKernelGGL(scene_select_ba_kernel, gF, b, 0, s, F, in_set, keep, sc->track_of.as<int32_t>(), ok, cnt_keep, sel);
    OSFM_RETURN_IF(exclusive_scan(sc, sel, scan, F + 1, s));
    if (permanent) {
        hipLaunchKernelGGL(scene_kill_tracks_kernel, gT, b, 0, s, T, cnt_set, ok, sc->alive_t.as<uint8_t>());
        hipLaunchKernelGGL(scene_kill_features_kernel, gF, b, 0, s, F, in_set, keep, sc->alive_f.as<uint8_t>());
    }
    OSFM_HIP_CHECK(hipGetLastError());
    if (num_tracks_selected) {
        // the tracks of the selection, numbered as they appear (what the adjustment behind this builds its points
        // from): flags in sc->tflag, their scan in sc->tslot, both totals behind one synchronisation
        hipLaunchKernelGGL(scene_track_count_kernel, gT, b, 0, s, T, sc->offsets.as<int32_t>(), scan, cnt_set);
        hipLaunchKernelGGL(scene_track_flag_kernel, gT, b, 0, s, T, (int)kFlagCountPositive, 0, cnt_set, nullptr, nullptr, tflag);
        OSFM_RETURN_IF(exclusive_scan(sc, tflag, sc->tslot.as<int32_t>(), (int64_t)T + 1, s));
        OSFM_RETURN_IF(read_totals2(scan, F, sc->tslot.as<int32_t>(), T, s, num_selected, num_tracks_selected));
    } else {
        OSFM_RETURN_IF(read_total(scan, F, s, num_selected));
    }
    return OSFM_OK;
}

// ---- compaction: the alive features of the alive tracks, in order ----
__global__ void scene_compact_features_kernel(int64_t F, const int32_t *__restrict__ sel, const int32_t *__restrict__ scan,
    const int32_t *__restrict__ tslot, const int32_t *__restrict__ view, const double2 *__restrict__ xy,
    const int32_t *__restrict__ track_of, const int32_t *__restrict__ orig_f, int32_t *__restrict__ view2,
    double2 *__restrict__ xy2, int32_t *__restrict__ track_of2, int32_t *__restrict__ orig_f2)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F || !sel[f]) return;
    const int nf = scan[f];
    view2[nf] = view[f];
    xy2[nf] = xy[f];
    track_of2[nf] = tslot[track_of[f]];
    orig_f2[nf] = orig_f ? orig_f[f] : (int32_t)f;
}

__global__ void scene_compact_tracks_kernel(int T, int T2, int F2, const int32_t *__restrict__ tflag, const int32_t *__restrict__ tslot,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ scan, const uint8_t *__restrict__ has_point,
    const double *__restrict__ point, const int32_t *__restrict__ orig_t, int32_t *__restrict__ offsets2,
    uint8_t *__restrict__ has_point2, double *__restrict__ point2, int32_t *__restrict__ orig_t2)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) offsets2[T2] = F2;
    if (t >= T || !tflag[t]) return;
    const int nt = tslot[t];
    offsets2[nt] = scan[offsets[t]];
    has_point2[nt] = has_point[t];
    for (int i = 0; i < 4; ++i) point2[4 * (size_t)nt + i] = point[4 * (size_t)t + i];
    orig_t2[nt] = orig_t ? orig_t[t] : t;
}

// out[map[i]] = in[i] (bytes / 32-byte points): the compact table back into the caller's numbering
__global__ void scene_scatter_u8_kernel(int64_t n, const int32_t *__restrict__ map, const uint8_t *__restrict__ in, uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[map[i]] = in[i];
}
__global__ void scene_scatter_pt_kernel(int n, const int32_t *__restrict__ map, const double *__restrict__ in, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 4; ++k) out[4 * (size_t)map[i] + k] = in[4 * (size_t)i + k];
}
__global__ void scene_gather_u8_kernel(int64_t n, const int32_t *__restrict__ map, const uint8_t *__restrict__ in, uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[map[i]];
}

// how many of the entries a compacted table still holds are alive in the caller's (uncompacted) flags
__global__ void scene_count_mapped_u8_kernel(int64_t n, const int32_t *__restrict__ map, const uint8_t *__restrict__ in, int32_t *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < n && in[map[i]] != 0;
    const unsigned long long b = __ballot(on);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (int32_t)__popcll(b));
}

// Drops the dead tracks and the dead features of the others (the caller has decided that it pays).
int scene_compact(osfm_scene *sc, hipStream_t s)
{
    const int64_t F = sc->F;
    const int T = sc->T;
    int32_t *sel = sc->sel.as<int32_t>(), *scan = sc->scan.as<int32_t>(), *tflag = sc->tflag.as<int32_t>(), *tslot = sc->tslot.as<int32_t>();
    hipLaunchKernelGGL(scene_track_flag_kernel, dim3(blocks_for(T + 1)), dim3(kThreads), 0, s, T, (int)kFlagAlive, 0, nullptr,
        sc->alive_t.as<uint8_t>(), nullptr, tflag);
    OSFM_RETURN_IF(exclusive_scan(sc, tflag, tslot, (int64_t)T + 1, s));
    hipLaunchKernelGGL(scene_select_kernel, dim3(blocks_for(F + 1)), dim3(kThreads), 0, s, F, sc->feat_view.as<int32_t>(),
        sc->track_of.as<int32_t>(), sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), nullptr, nullptr, sel);
    OSFM_RETURN_IF(exclusive_scan(sc, sel, scan, F + 1, s));
    int F2 = 0, T2 = 0;
    OSFM_RETURN_IF(read_totals2(scan, F, tslot, T, s, &F2, &T2));
    if (T2 == T && F2 == F) return OSFM_OK;
    if (!sc->compacted) { sc->T0 = T; sc->F0 = F; }
    DeviceBuffer view2, xy2, tof2, off2, af2, at2, hp2, pt2, ot2, of2;
    const size_t Fz = (size_t)std::max(F2, 1), Tz = (size_t)std::max(T2, 1);
    OSFM_RETURN_IF(view2.reserve(Fz * 4)); OSFM_RETURN_IF(xy2.reserve(Fz * 16)); OSFM_RETURN_IF(tof2.reserve(Fz * 4));
    OSFM_RETURN_IF(off2.reserve((Tz + 1) * 4)); OSFM_RETURN_IF(af2.reserve(Fz)); OSFM_RETURN_IF(at2.reserve(Tz));
    OSFM_RETURN_IF(hp2.reserve(Tz)); OSFM_RETURN_IF(pt2.reserve(Tz * 32)); OSFM_RETURN_IF(ot2.reserve(Tz * 4)); OSFM_RETURN_IF(of2.reserve(Fz * 4));
    const int32_t *orig_t = sc->compacted ? sc->orig_t.as<int32_t>() : nullptr, *orig_f = sc->compacted ? sc->orig_f.as<int32_t>() : nullptr;
    hipLaunchKernelGGL(scene_compact_features_kernel, dim3(blocks_for(F)), dim3(kThreads), 0, s, F, sel, scan, tslot, sc->feat_view.as<int32_t>(),
        sc->feat_xy.as<double2>(), sc->track_of.as<int32_t>(), orig_f, view2.as<int32_t>(), xy2.as<double2>(), tof2.as<int32_t>(), of2.as<int32_t>());
    hipLaunchKernelGGL(scene_compact_tracks_kernel, dim3(blocks_for(T)), dim3(kThreads), 0, s, T, T2, F2, tflag, tslot, sc->offsets.as<int32_t>(),
        scan, sc->has_point.as<uint8_t>(), sc->point.as<double>(), orig_t, off2.as<int32_t>(), hp2.as<uint8_t>(), pt2.as<double>(), ot2.as<int32_t>());
    OSFM_HIP_CHECK(hipMemsetAsync(af2.ptr, 1, Fz, s));
    OSFM_HIP_CHECK(hipMemsetAsync(at2.ptr, 1, Tz, s));
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipStreamSynchronize(s));            // the old arrays are released below
    sc->feat_view = std::move(view2); sc->feat_xy = std::move(xy2); sc->track_of = std::move(tof2); sc->offsets = std::move(off2);
    sc->alive_f = std::move(af2); sc->alive_t = std::move(at2); sc->has_point = std::move(hp2); sc->point = std::move(pt2);
    sc->orig_t = std::move(ot2); sc->orig_f = std::move(of2);
    sc->T = T2; sc->F = F2; sc->compacted = true;
    return OSFM_OK;
}

// the pair-list bound of a finished problem
int pair_bound_of(osfm_scene *sc, const Compact &P, hipStream_t s, int64_t *bound)
{
    unsigned long long *c = sc->counters.as<unsigned long long>();
    OSFM_HIP_CHECK(hipMemsetAsync(c, 0, 8, s));
    if (P.M) hipLaunchKernelGGL(scene_pair_bound_kernel, dim3(blocks_for(P.M)), dim3(kThreads), 0, s, P.M, P.D.pt_start.as<int32_t>(), c);
    unsigned long long v = 0;
    OSFM_HIP_CHECK(hipMemcpyAsync(&v, c, 8, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    *bound = (int64_t)v;
    return OSFM_OK;
}

}  // namespace

extern "C" {

int osfm_scene_create(int device, int model, int num_views, const int32_t *img_width, const int32_t *img_height,
    int32_t num_tracks, const int64_t *track_offsets, const int32_t *feat_view, const float *feat_xy, osfm_scene **out)
{
    if (!out || num_views <= 0 || num_tracks < 0 || !img_width || !img_height || !track_offsets ||
        (model != OSFM_BA_MODEL_QUATERNION && model != OSFM_BA_MODEL_EULER)) {
        set_error("scene_create: bad arguments");
        return OSFM_E_ARG;
    }
    const int64_t F = track_offsets[num_tracks];
    if (track_offsets[0] != 0 || F < 0 || F >= (int64_t)0x7fffffff || (F > 0 && (!feat_view || !feat_xy))) {
        set_error("scene_create: bad track table (%lld features; fewer than 2^31 - 1 supported)", (long long)F);
        return OSFM_E_ARG;
    }
    for (int t = 0; t < num_tracks; ++t)
        if (track_offsets[t + 1] < track_offsets[t]) { set_error("scene_create: track_offsets must be non-decreasing"); return OSFM_E_ARG; }
    for (int64_t f = 0; f < F; ++f)
        if (feat_view[f] < 0 || feat_view[f] >= num_views) { set_error("scene_create: feature %lld names view %d", (long long)f, feat_view[f]); return OSFM_E_ARG; }
    {
        // one feature per view and track (bundler_tracks.cc:120-145 drops tracks with two): the adjustments'
        // pair lists and the analytic size bounds of the scene rest on it
        std::vector<int32_t> seen_in((size_t)num_views, -1);
        for (int t = 0; t < num_tracks; ++t)
            for (int64_t f = track_offsets[t]; f < track_offsets[t + 1]; ++f) {
                if (seen_in[feat_view[f]] == t) {
                    set_error("scene_create: track %d holds two features of view %d; one feature per view and track", t, feat_view[f]);
                    return OSFM_E_ARG;
                }
                seen_in[feat_view[f]] = t;
            }
    }
    OSFM_RETURN_IF(select_device(device));
    struct Owner { osfm_scene *p; ~Owner() { delete p; } } owner{new osfm_scene()};
    osfm_scene *sc = owner.p;
    sc->device = device; sc->model = model; sc->V = num_views; sc->T = num_tracks; sc->F = F;
    sc->img_w.assign(img_width, img_width + num_views);
    sc->img_h.assign(img_height, img_height + num_views);
    sc->h_cams.assign((size_t)num_views * 7, 0.0);
    sc->h_const.assign((size_t)num_views * 7, 0);
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;
    const size_t Fz = (size_t)std::max<int64_t>(F, 1), Tz = (size_t)std::max(num_tracks, 1);
    OSFM_RETURN_IF(sc->feat_view.reserve(Fz * 4));
    OSFM_RETURN_IF(sc->feat_xy.reserve(Fz * 16));
    OSFM_RETURN_IF(sc->track_of.reserve(Fz * 4));
    OSFM_RETURN_IF(sc->offsets.reserve((Tz + 1) * 4));
    OSFM_RETURN_IF(sc->alive_f.reserve(Fz));
    OSFM_RETURN_IF(sc->alive_t.reserve(Tz));
    OSFM_RETURN_IF(sc->has_point.reserve(Tz));
    OSFM_RETURN_IF(sc->point.reserve(Tz * 32));
    OSFM_RETURN_IF(sc->cam_of_view.reserve((size_t)num_views * 4));
    OSFM_RETURN_IF(sc->cams.reserve((size_t)num_views * 56));
    OSFM_RETURN_IF(sc->tmp_hp.reserve(Tz));
    OSFM_RETURN_IF(sc->tmp_point.reserve(Tz * 32));
    OSFM_RETURN_IF(reserve_scratch(sc));
    // Feature::x / y are floats (track.h:26-27): the double of each is what the residuals see.  The floats go up as
    // they are and are widened there, the track of every feature is expanded from the offsets there (on the host the
    // two loops were 15 ms of a 200-view job's 3.2 M features)
    std::vector<int32_t> off32((size_t)num_tracks + 1);
    for (int t = 0; t < num_tracks; ++t) off32[t] = (int32_t)track_offsets[t];
    off32[num_tracks] = (int32_t)F;
    OSFM_HIP_CHECK(hipMemcpyAsync(sc->offsets.ptr, off32.data(), ((size_t)num_tracks + 1) * 4, hipMemcpyHostToDevice, s));
    if (F) {
        OSFM_HIP_CHECK(hipMemcpyAsync(sc->feat_view.ptr, feat_view, (size_t)F * 4, hipMemcpyHostToDevice, s));
        // (the scan array is scratch of F + 1 ints: the floats' landing place)
        OSFM_RETURN_IF(sc->aux_f.reserve((size_t)F * 8));
        OSFM_HIP_CHECK(hipMemcpyAsync(sc->aux_f.ptr, feat_xy, (size_t)F * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(scene_widen_xy_kernel, dim3(blocks_for(2 * F)), dim3(kThreads), 0, s, 2 * F, sc->aux_f.as<float>(), sc->feat_xy.as<double>());
        hipLaunchKernelGGL(scene_expand_tracks_kernel, dim3(blocks_for(num_tracks)), dim3(kThreads), 0, s, num_tracks, sc->offsets.as<int32_t>(),
            sc->track_of.as<int32_t>());
    }
    OSFM_HIP_CHECK(hipMemsetAsync(sc->alive_f.ptr, 1, Fz, s));
    OSFM_HIP_CHECK(hipMemsetAsync(sc->alive_t.ptr, 1, Tz, s));
    OSFM_HIP_CHECK(hipMemsetAsync(sc->has_point.ptr, 0, Tz, s));
    OSFM_HIP_CHECK(hipMemsetAsync(sc->point.ptr, 0, Tz * 32, s));
    hipLaunchKernelGGL(scene_fill_i32_kernel, dim3(blocks_for(num_views)), dim3(kThreads), 0, s, (int64_t)num_views, -1, sc->cam_of_view.as<int32_t>());
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    *out = sc;
    owner.p = nullptr;
    return OSFM_OK;
}

int osfm_scene_destroy(osfm_scene *sc)
{
    if (!sc) return OSFM_OK;
    (void)hipSetDevice(sc->device);
    delete sc;
    return OSFM_OK;
}

int osfm_scene_set_flags(osfm_scene *sc, const uint8_t *alive_track, const uint8_t *alive_feature)
{
    if (!sc) { set_error("scene_set_flags: null scene"); return OSFM_E_ARG; }
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    if (!sc->compacted) {
        if (alive_track && sc->T) OSFM_HIP_CHECK(hipMemcpy(sc->alive_t.ptr, alive_track, (size_t)sc->T, hipMemcpyHostToDevice));
        if (alive_feature && sc->F) OSFM_HIP_CHECK(hipMemcpy(sc->alive_f.ptr, alive_feature, (size_t)sc->F, hipMemcpyHostToDevice));
        return OSFM_OK;
    }
    // a compacted table: the flags of what is still in it (what was dropped stays dropped)
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;
    DevArray tmp;
    OSFM_RETURN_IF(tmp.alloc(std::max<size_t>(std::max((size_t)sc->F0, (size_t)sc->T0), 16)));
    // A flag set for an entry the compaction dropped cannot be honoured (its data is gone): refuse the call
    // before anything changes instead of leaving the caller's table and the scene's apart.
    for (int kind = 0; kind < 2; ++kind) {
        const uint8_t *flags = kind == 0 ? alive_track : alive_feature;
        const int64_t n0 = kind == 0 ? (int64_t)sc->T0 : sc->F0, n = kind == 0 ? (int64_t)sc->T : sc->F;
        if (!flags || n0 == 0) continue;
        int64_t want = 0;
        for (int64_t i = 0; i < n0; ++i) want += flags[i] != 0;
        int32_t have = 0;
        int32_t *c = sc->counters.as<int32_t>();
        OSFM_HIP_CHECK(hipMemsetAsync(c, 0, 4, s));
        if (n) {
            OSFM_HIP_CHECK(hipMemcpyAsync(tmp.ptr, flags, (size_t)n0, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(scene_count_mapped_u8_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, n,
                (kind == 0 ? sc->orig_t : sc->orig_f).as<int32_t>(), tmp.as<uint8_t>(), c);
        }
        OSFM_HIP_CHECK(hipMemcpyAsync(&have, c, 4, hipMemcpyDeviceToHost, s));
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
        if (want != have) {
            set_error("scene_set_flags: %lld %s flagged alive that the scene has dropped (a compacted scene cannot revive them)",
                (long long)(want - have), kind == 0 ? "track(s)" : "feature(s)");
            return OSFM_E_STATE;
        }
    }
    if (alive_track && sc->T) {
        OSFM_HIP_CHECK(hipMemcpyAsync(tmp.ptr, alive_track, (size_t)sc->T0, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(scene_gather_u8_kernel, dim3(blocks_for(sc->T)), dim3(kThreads), 0, s, (int64_t)sc->T, sc->orig_t.as<int32_t>(),
            tmp.as<uint8_t>(), sc->alive_t.as<uint8_t>());
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (alive_feature && sc->F) {
        OSFM_HIP_CHECK(hipMemcpyAsync(tmp.ptr, alive_feature, (size_t)sc->F0, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(scene_gather_u8_kernel, dim3(blocks_for(sc->F)), dim3(kThreads), 0, s, sc->F, sc->orig_f.as<int32_t>(),
            tmp.as<uint8_t>(), sc->alive_f.as<uint8_t>());
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return OSFM_OK;
}

int osfm_scene_align_views(osfm_scene *sc, int n, const int32_t *views, const double *params, const uint8_t *cam_const)
{
    if (!sc || n < 0 || (n > 0 && (!params || !cam_const))) { set_error("scene_align_views: bad arguments"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(check_views(sc, views, n, "scene_align_views"));
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    std::vector<bool> seen(sc->V, false);
    for (int v : sc->aligned) seen[v] = true;
    for (int i = 0; i < n; ++i) {
        if (seen[views[i]]) { set_error("scene_align_views: view %d has a camera already", views[i]); return OSFM_E_STATE; }
        seen[views[i]] = true;
    }
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    const int first = (int)sc->aligned.size();
    DevArray d_views;
    if (n) {
        OSFM_RETURN_IF(upload(d_views, views, (size_t)n, sg.s));
        hipLaunchKernelGGL(scene_set_map_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, sg.s, n, d_views.as<int32_t>(), first,
            sc->cam_of_view.as<int32_t>());
        OSFM_HIP_CHECK(hipMemcpyAsync(sc->cams.as<double>() + (size_t)first * 7, params, (size_t)n * 56, hipMemcpyHostToDevice, sg.s));
        OSFM_HIP_CHECK(hipStreamSynchronize(sg.s));
    }
    for (int i = 0; i < n; ++i) {
        sc->aligned.push_back(views[i]);
        memcpy(&sc->h_cams[(size_t)(first + i) * 7], params + (size_t)i * 7, 56);
        memcpy(&sc->h_const[(size_t)(first + i) * 7], cam_const + (size_t)i * 7, 7);
    }
    return OSFM_OK;
}

int osfm_scene_set_cameras(osfm_scene *sc, int n, const int32_t *views, const double *params)
{
    if (!sc || n < 0 || (n > 0 && !params)) { set_error("scene_set_cameras: bad arguments"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(check_views(sc, views, n, "scene_set_cameras"));
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    std::vector<int> slot_of(sc->V, -1);
    for (size_t c = 0; c < sc->aligned.size(); ++c) slot_of[sc->aligned[c]] = (int)c;
    for (int i = 0; i < n; ++i)
        if (slot_of[views[i]] < 0) { set_error("scene_set_cameras: view %d has no camera yet (osfm_scene_align_views)", views[i]); return OSFM_E_STATE; }
    if (n == 0) return OSFM_OK;
    for (int i = 0; i < n; ++i) memcpy(&sc->h_cams[(size_t)slot_of[views[i]] * 7], params + (size_t)i * 7, 56);
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    OSFM_HIP_CHECK(hipMemcpyAsync(sc->cams.ptr, sc->h_cams.data(), sc->aligned.size() * 56, hipMemcpyHostToDevice, sg.s));
    OSFM_HIP_CHECK(hipStreamSynchronize(sg.s));
    // intersections made from the old parameters are stale: the next triangulation redoes every track
    sc->cams_moved = true;
    return OSFM_OK;
}

int osfm_scene_get_cameras(osfm_scene *sc, int capacity, int32_t *views, double *params, int32_t *num_cameras)
{
    if (!sc) { set_error("scene_get_cameras: null scene"); return OSFM_E_ARG; }
    std::lock_guard<std::mutex> lock(sc->mu);
    const int C = (int)sc->aligned.size();
    if (num_cameras) *num_cameras = C;
    if (capacity < C && (views || params)) { set_error("scene_get_cameras: %d cameras, room for %d", C, capacity); return OSFM_E_CAPACITY; }
    for (int c = 0; c < C; ++c) {
        if (views) views[c] = sc->aligned[c];
        if (params) memcpy(params + (size_t)c * 7, &sc->h_cams[(size_t)c * 7], 56);
    }
    return OSFM_OK;
}

int osfm_scene_triangulate(osfm_scene *sc, int num_new_views, const int32_t *new_views, int check_full, int32_t *mismatches)
{
    if (!sc) { set_error("scene_triangulate: null scene"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(check_views(sc, new_views, num_new_views, "scene_triangulate"));
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;
    if (mismatches) *mismatches = 0;
    if (sc->T == 0) return OSFM_OK;
    // cameras replaced from outside since the last pass (osfm_scene_set_cameras): nothing may be kept
    const bool incremental = new_views != nullptr && !sc->cams_moved;
    sc->cams_moved = false;
    uint8_t *tmask = nullptr;
    DevArray d_views;
    if (incremental) {
        // the tracks the new views see: everything else keeps the intersection it has (the cameras it was made
        // from have not moved)
        tmask = sc->aux_t2.as<uint8_t>();
        OSFM_HIP_CHECK(hipMemsetAsync(tmask, 0, (size_t)sc->T, s));
        OSFM_RETURN_IF(set_cam_map(sc, new_views, num_new_views, s, &d_views));
        hipLaunchKernelGGL(scene_touch_kernel, dim3(blocks_for(sc->F)), dim3(kThreads), 0, s, sc->F, sc->feat_view.as<int32_t>(),
            sc->track_of.as<int32_t>(), sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), sc->cam_map.as<int32_t>(), tmask);
    }
    OSFM_RETURN_IF(triangulate_pass(sc, tmask, !incremental, sc->has_point.as<uint8_t>(), sc->point.as<double>(), s));
    if (incremental && check_full) {
        OSFM_HIP_CHECK(hipMemcpyAsync(sc->tmp_point.ptr, sc->point.ptr, (size_t)sc->T * 32, hipMemcpyDeviceToDevice, s));
        OSFM_RETURN_IF(triangulate_pass(sc, nullptr, true, sc->tmp_hp.as<uint8_t>(), sc->tmp_point.as<double>(), s));
        int32_t *c = sc->counters.as<int32_t>();
        OSFM_HIP_CHECK(hipMemsetAsync(c, 0, 4, s));
        hipLaunchKernelGGL(scene_compare_kernel, dim3(blocks_for(sc->T)), dim3(kThreads), 0, s, sc->T, sc->has_point.as<uint8_t>(),
            sc->point.as<double>(), sc->tmp_hp.as<uint8_t>(), sc->tmp_point.as<double>(), c);
        int32_t bad = 0;
        OSFM_HIP_CHECK(hipMemcpyAsync(&bad, c, 4, hipMemcpyDeviceToHost, s));
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
        if (mismatches) *mismatches = bad;
    }
    return OSFM_OK;
}

int osfm_scene_filter_reprojection(osfm_scene *sc, double max_error)
{
    if (!sc) { set_error("scene_filter_reprojection: null scene"); return OSFM_E_ARG; }
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    const int C = (int)sc->aligned.size();
    if (C == 0 || sc->T == 0) return OSFM_OK;
    int nsel = 0;
    return reprojection_filter(sc, C, sc->aligned.data(), sc->h_cams.data(), sc->h_const.data(), max_error, true, sg.s, &nsel);
}

int osfm_scene_local_adjustment(osfm_scene *sc, int n, const int32_t *views, double *params, const uint8_t *cam_const,
    double max_error, const osfm_ba_options *opt, osfm_ba_summary *sum, int32_t *num_points, int32_t *num_observations)
{
    if (!sc || n <= 0 || !params || !cam_const || !sum) { set_error("scene_local_adjustment: bad arguments"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(check_views(sc, views, n, "scene_local_adjustment"));
    memset(sum, 0, sizeof(*sum));
    osfm_ba_options o;
    if (opt) o = *opt; else osfm_ba_options_default(&o);
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    const auto t_begin = std::chrono::steady_clock::now();
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;
    static const bool trace = getenv("OSFM_SCENE_TRACE") != nullptr;
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[osfm scene] %-22s %8.3f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    int O = 0, M = 0;
    OSFM_RETURN_IF(reprojection_filter(sc, n, views, params, cam_const, max_error, false, s, &O, &M));
    lap("filter");
    // the features the filter leaves, their tracks numbered as they appear (the filter has flagged and counted them),
    // every point at (0, 0, 0, 1): the adjustment re-triangulates its copy first (runBundleAdjustment(..., true, true),
    // reconstruct.cpp:212-219)
    int32_t *sel = sc->sel.as<int32_t>(), *scan = sc->scan.as<int32_t>(), *tflag = sc->tflag.as<int32_t>();
    Compact P;
    Layout L;
    const int pdim = o.optimize_points ? 3 : 0;
    OSFM_RETURN_IF(build_problem(sc, sel, scan, O, tflag, nullptr, n, params, nullptr, cam_const, views, sc->cam_map.as<int32_t>(),
        o.huber_delta, pdim, &L, s, &P, M));
    if (num_points) *num_points = P.M;
    if (num_observations) *num_observations = P.O;
    lap("problem");
    if (o.retriangulate_points && P.M > 0) {
        OSFM_HIP_CHECK(hipMemcpyAsync(P.D.points[1].ptr, P.D.points[0].ptr, (size_t)P.M * 32, hipMemcpyDeviceToDevice, s));
        launch_triangulate(P.D.dev, P.D.points[1].as<double>(), nullptr, s);
        OSFM_HIP_CHECK(hipMemcpyAsync(P.D.points[0].ptr, P.D.points[1].ptr, (size_t)P.M * 32, hipMemcpyDeviceToDevice, s));
    }
    // pair-list bound: a track has at most one feature per view, i.e. at most n observations here
    const int64_t bound = pdim ? (int64_t)P.O * n : P.O;
    int cur = 0;
    lap("bound");
    OSFM_RETURN_IF(ba_solve_core(P.D, o, sg, bound, sum, &cur));
    lap("solve");
    OSFM_HIP_CHECK(hipMemcpyAsync(params, P.D.cams[cur].ptr, (size_t)n * 56, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return OSFM_OK;
}

int osfm_scene_global_adjustment(osfm_scene *sc, const osfm_ba_options *opt, osfm_ba_summary *sum, int32_t *num_points,
    int32_t *num_observations)
{
    if (!sc || !sum) { set_error("scene_global_adjustment: bad arguments"); return OSFM_E_ARG; }
    memset(sum, 0, sizeof(*sum));
    osfm_ba_options o;
    if (opt) o = *opt; else osfm_ba_options_default(&o);
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    const auto t_begin = std::chrono::steady_clock::now();
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;
    const int64_t F = sc->F;
    const int T = sc->T, C = (int)sc->aligned.size();
    // every alive track with a point is a parameter block, every live feature of such a track whose view has a
    // camera a residual (bundle_adjustment.cpp:86-123)
    int32_t *sel = sc->sel.as<int32_t>(), *scan = sc->scan.as<int32_t>(), *tflag = sc->tflag.as<int32_t>();
    uint8_t *tmask = sc->aux_t2.as<uint8_t>();
    hipLaunchKernelGGL(scene_track_flag_kernel, dim3(blocks_for(T + 1)), dim3(kThreads), 0, s, T, (int)kFlagAliveWithPoint, 0, nullptr,
        sc->alive_t.as<uint8_t>(), sc->has_point.as<uint8_t>(), tflag);
    hipLaunchKernelGGL(scene_to_byte_kernel, dim3(blocks_for(T)), dim3(kThreads), 0, s, T, tflag, tmask);
    hipLaunchKernelGGL(scene_select_kernel, dim3(blocks_for(F + 1)), dim3(kThreads), 0, s, F, sc->feat_view.as<int32_t>(),
        sc->track_of.as<int32_t>(), sc->alive_f.as<uint8_t>(), sc->alive_t.as<uint8_t>(), sc->cam_of_view.as<int32_t>(), tmask, sel);
    OSFM_RETURN_IF(exclusive_scan(sc, sel, scan, F + 1, s));
    int O = 0, M = 0;
    OSFM_RETURN_IF(exclusive_scan(sc, tflag, sc->tslot.as<int32_t>(), (int64_t)T + 1, s));
    OSFM_RETURN_IF(read_totals2(scan, F, sc->tslot.as<int32_t>(), T, s, &O, &M));
    Compact P;
    Layout L;
    const int pdim = o.optimize_points ? 3 : 0;
    OSFM_RETURN_IF(build_problem(sc, sel, scan, O, tflag, sc->point.as<double>(), C, nullptr, sc->cams.as<double>(), sc->h_const.data(),
        sc->aligned.data(), sc->cam_of_view.as<int32_t>(), o.huber_delta, pdim, &L, s, &P, M));
    if (num_points) *num_points = P.M;
    if (num_observations) *num_observations = P.O;
    int64_t bound = 0;
    OSFM_RETURN_IF(pair_bound_of(sc, P, s, &bound));
    if (!pdim) bound = P.O;
    int cur = 0;
    OSFM_RETURN_IF(ba_solve_core(P.D, o, sg, bound, sum, &cur));
    // cameras and points updated in place
    if (C) {
        OSFM_HIP_CHECK(hipMemcpyAsync(sc->cams.ptr, P.D.cams[cur].ptr, (size_t)C * 56, hipMemcpyDeviceToDevice, s));
        OSFM_HIP_CHECK(hipMemcpyAsync(sc->h_cams.data(), P.D.cams[cur].ptr, (size_t)C * 56, hipMemcpyDeviceToHost, s));
    }
    if (P.M) hipLaunchKernelGGL(scene_scatter_points_kernel, dim3(blocks_for(P.M)), dim3(kThreads), 0, s, P.M,
        P.point_track.as<int32_t>(), P.D.points[cur].as<double>(), sc->point.as<double>());
    OSFM_HIP_CHECK(hipGetLastError());
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return OSFM_OK;
}

int osfm_scene_filter_outliers(osfm_scene *sc, osfm_outlier_stats *stats, int32_t *num_killed)
{
    if (!sc) { set_error("scene_filter_outliers: null scene"); return OSFM_E_ARG; }
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    const int T = sc->T;
    if (num_killed) *num_killed = 0;
    if (T == 0) return OSFM_OK;
    // filterOutlierTracks on the alive tracks (outlier_filtering.cpp:40-125): the O(P^2) distance search runs on
    // the device, its statistics on the host in the reference's sequential order; what it drops loses its flag
    std::vector<uint8_t> alive(T), hp(T);
    std::vector<double> pt((size_t)T * 4);
    OSFM_HIP_CHECK(hipMemcpy(alive.data(), sc->alive_t.ptr, (size_t)T, hipMemcpyDeviceToHost));
    OSFM_HIP_CHECK(hipMemcpy(hp.data(), sc->has_point.ptr, (size_t)T, hipMemcpyDeviceToHost));
    OSFM_HIP_CHECK(hipMemcpy(pt.data(), sc->point.ptr, (size_t)T * 32, hipMemcpyDeviceToHost));
    std::vector<int> at;
    for (int t = 0; t < T; ++t) if (alive[t]) at.push_back(t);
    const int n = (int)at.size();
    std::vector<double> cp((size_t)std::max(n, 1) * 4);
    std::vector<uint8_t> chp(std::max(n, 1)), keep(std::max(n, 1));
    for (int i = 0; i < n; ++i) { memcpy(&cp[(size_t)i * 4], &pt[(size_t)at[i] * 4], 32); chp[i] = hp[at[i]]; }
    OSFM_RETURN_IF(osfm_filter_outlier_tracks(sc->device, cp.data(), chp.data(), n, keep.data(), stats));
    int killed = 0;
    for (int i = 0; i < n; ++i) if (!keep[i]) { alive[at[i]] = 0; ++killed; }
    if (killed) OSFM_HIP_CHECK(hipMemcpy(sc->alive_t.ptr, alive.data(), (size_t)T, hipMemcpyHostToDevice));
    if (num_killed) *num_killed = killed;
    // fewer than half of a large table's tracks are left (the reprojection filter in front of this one clears flags
    // too): drop the rest, every step passes over all features.  OSFM_SCENE_COMPACT = 0: never; 2: whenever anything is
    // dead (tests: small sets whose tracks mostly survive)
    const int policy = getenv("OSFM_SCENE_COMPACT") ? atoi(getenv("OSFM_SCENE_COMPACT")) : 1;
    const bool pays = T >= 4096 && 2 * (int64_t)(n - killed) <= T;
    if (policy == 2 || (policy == 1 && pays)) {
        StreamLease sg;
        OSFM_RETURN_IF(sg.acquire());
        OSFM_RETURN_IF(scene_compact(sc, sg.s));
    }
    return OSFM_OK;
}

int osfm_scene_download(osfm_scene *sc, uint8_t *alive_track, uint8_t *alive_feature, uint8_t *has_point, double *points)
{
    if (!sc) { set_error("scene_download: null scene"); return OSFM_E_ARG; }
    std::lock_guard<std::mutex> lock(sc->mu);
    OSFM_RETURN_IF(select_device(sc->device));
    if (!sc->compacted) {
        if (alive_track && sc->T) OSFM_HIP_CHECK(hipMemcpy(alive_track, sc->alive_t.ptr, (size_t)sc->T, hipMemcpyDeviceToHost));
        if (alive_feature && sc->F) OSFM_HIP_CHECK(hipMemcpy(alive_feature, sc->alive_f.ptr, (size_t)sc->F, hipMemcpyDeviceToHost));
        if (has_point && sc->T) OSFM_HIP_CHECK(hipMemcpy(has_point, sc->has_point.ptr, (size_t)sc->T, hipMemcpyDeviceToHost));
        if (points && sc->T) OSFM_HIP_CHECK(hipMemcpy(points, sc->point.ptr, (size_t)sc->T * 32, hipMemcpyDeviceToHost));
        return OSFM_OK;
    }
    // the compacted table back into the caller's numbering: what was dropped is dead, without a point, at (0, 0, 0, 0)
    StreamLease sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;
    DevArray tmp;
    const size_t T0 = (size_t)sc->T0, F0 = (size_t)sc->F0;
    OSFM_RETURN_IF(tmp.alloc(std::max<size_t>(std::max(F0, T0 * 32), 16)));
    const int32_t *ot = sc->orig_t.as<int32_t>(), *of = sc->orig_f.as<int32_t>();
    auto bytes_out = [&](const uint8_t *in, int64_t n, const int32_t *map, size_t n0, uint8_t *host) -> int {
        OSFM_HIP_CHECK(hipMemsetAsync(tmp.ptr, 0, std::max<size_t>(n0, 1), s));
        if (n) hipLaunchKernelGGL(scene_scatter_u8_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, n, map, in, tmp.as<uint8_t>());
        OSFM_HIP_CHECK(hipMemcpyAsync(host, tmp.ptr, n0, hipMemcpyDeviceToHost, s));
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
        return OSFM_OK;
    };
    if (alive_track && T0) OSFM_RETURN_IF(bytes_out(sc->alive_t.as<uint8_t>(), sc->T, ot, T0, alive_track));
    if (alive_feature && F0) OSFM_RETURN_IF(bytes_out(sc->alive_f.as<uint8_t>(), sc->F, of, F0, alive_feature));
    if (has_point && T0) OSFM_RETURN_IF(bytes_out(sc->has_point.as<uint8_t>(), sc->T, ot, T0, has_point));
    if (points && T0) {
        OSFM_HIP_CHECK(hipMemsetAsync(tmp.ptr, 0, T0 * 32, s));
        if (sc->T) hipLaunchKernelGGL(scene_scatter_pt_kernel, dim3(blocks_for(sc->T)), dim3(kThreads), 0, s, sc->T, ot, sc->point.as<double>(), tmp.as<double>());
        OSFM_HIP_CHECK(hipMemcpyAsync(points, tmp.ptr, T0 * 32, hipMemcpyDeviceToHost, s));
        OSFM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return OSFM_OK;
}

}  // extern "C"
