// C-ABI implementation of the bundle-adjustment path (include/osfm_hip.h,
// section B): the Levenberg-Marquardt control loop of ceres::Solve as
// configured by runBundleAdjustment (bundle_adjustment.cpp:126-145), driving
// the kernels of ba_kernels.hip / ba_cholesky.hip.  The host only moves a few
// scalars per iteration and takes the accept/reject decisions.
//
// Trust-region logic restated from the published Ceres 2.0/2.1 algorithm
// (TrustRegionMinimizer, LevenbergMarquardtStrategy); see DESIGN.md for the
// list of behaviours and the "parity unpinned" note.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <string>
#include <vector>

#include "ba_solve.h"

using namespace osfm;

namespace {

int validate_observations(const osfm_ba_problem *p, const char *what, int32_t *pt_start);

// sizes, model, null pointers: what has to hold before any array is touched
int validate_header(const osfm_ba_problem *p, const char *what)
{
    if (!p) { set_error("%s: null problem", what); return OSFM_E_ARG; }
    if (p->model != OSFM_BA_MODEL_QUATERNION && p->model != OSFM_BA_MODEL_EULER) {
        set_error("%s: unknown camera model %d", what, p->model); return OSFM_E_ARG;
    }
    if (p->num_cameras < 0 || p->num_points < 0 || p->num_observations < 0) {
        set_error("%s: negative size", what); return OSFM_E_ARG;
    }
    if ((p->num_cameras && (!p->cam_params || !p->cam_const || !p->img_width || !p->img_height)) ||
        (p->num_points && !p->points) ||
        (p->num_observations && (!p->obs_xy || !p->obs_camera || !p->obs_point))) {
        set_error("%s: null array", what); return OSFM_E_ARG;
    }
    return OSFM_OK;
}

int validate_problem(const osfm_ba_problem *p, const char *what)
{
    OSFM_RETURN_IF(validate_header(p, what));
    return validate_observations(p, what, nullptr);
}

// The per-observation half of the checks; pt_start (when given, sized M + 2, zeroed) takes the observation count of
// point j in entry j + 1 -- the counting pass of build_layout, in the same sweep over the caller's arrays.
int validate_observations(const osfm_ba_problem *p, const char *what, int32_t *pt_start)
{
    int prev = 0;
    // A point is observed at most once per camera: the reference's tracks hold one feature per view (a track
    // with two is a conflict and dropped, bundler_tracks.cc:120-145), and the Schur-complement fast paths
    // (one slot per camera and track, ba_pairs.hip; the dense product, ba_dense.hip) rest on it.
    std::vector<int> seen_in((size_t)p->num_cameras, -1);
    for (int k = 0; k < p->num_observations; ++k) {
        const int c = p->obs_camera[k], j = p->obs_point[k];
        if (c < 0 || c >= p->num_cameras || j < 0 || j >= p->num_points) {
            set_error("%s: observation %d references camera %d / point %d out of range", what, k, c, j);
            return OSFM_E_ARG;
        }
        if (j < prev) {
            set_error("%s: obs_point must be non-decreasing (observation %d)", what, k);
            return OSFM_E_ARG;
        }
        if (seen_in[c] == j) {
            set_error("%s: point %d is observed twice by camera %d (observation %d); one observation per camera and point", what, j, c, k);
            return OSFM_E_ARG;
        }
        seen_in[c] = j;
        prev = j;
        if (pt_start) pt_start[j + 1]++;
    }
    return OSFM_OK;
}

// pt_start of the caller's observations (validated non-decreasing in obs_point)
void build_layout(const osfm_ba_problem *p, Layout *L)
{
    const int M = p->num_points, O = p->num_observations;
    build_camera_layout(p->model, p->num_cameras, p->cam_const, L);
    L->pt_start.assign(M + 2, 0);
    for (int k = 0; k < O; ++k) L->pt_start[p->obs_point[k] + 1]++;
    for (int j = 0; j < M; ++j) L->pt_start[j + 1] += L->pt_start[j];
}

// the caller's arrays: queued before the layout is derived on the host, so that the 24 bytes per observation
// cross PCIe while the host counts them
int upload_caller_arrays(const osfm_ba_problem *p, hipStream_t s, DeviceProblem *D)
{
    const int C = p->num_cameras, M = p->num_points, O = p->num_observations;
    OSFM_RETURN_IF(upload(D->obs_xy, p->obs_xy, (size_t)2 * O, s));
    OSFM_RETURN_IF(upload(D->obs_cam, p->obs_camera, (size_t)O, s));
    OSFM_RETURN_IF(upload(D->points[0], p->points, (size_t)4 * M, s));
    OSFM_RETURN_IF(upload(D->cams[0], p->cam_params, (size_t)7 * C, s));
    OSFM_RETURN_IF(upload(D->img_w, p->img_width, (size_t)C, s));
    OSFM_RETURN_IF(upload(D->img_h, p->img_height, (size_t)C, s));
    return OSFM_OK;
}

int upload_problem(const osfm_ba_problem *p, const Layout &L, double huber, int pdim, hipStream_t s,
    DeviceProblem *D, bool caller_arrays_queued = false)
{
    const int C = p->num_cameras, M = p->num_points, O = p->num_observations;
    if (!caller_arrays_queued) OSFM_RETURN_IF(upload_caller_arrays(p, s, D));
    OSFM_RETURN_IF(upload(D->pt_start, L.pt_start.data(), (size_t)M + 1, s));
    OSFM_RETURN_IF(upload_camera_layout(L, C, s, D));
    // No synchronisation here: every source array (the caller's and the Layout's) outlives
    // the call, and what follows is ordered behind the copies on the same stream.
    return finish_device_problem(p->model, C, M, O, L.nc, huber, pdim, s, D);
}

using StreamGuard = StreamLease;

}  // namespace

namespace osfm {

int select_device(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (this backend has no CPU fallback)");
        return OSFM_E_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range [0,%d)", device, ndev); return OSFM_E_ARG; }
    OSFM_HIP_CHECK(hipSetDevice(device));
    return OSFM_OK;
}


void build_camera_layout(int model, int C, const uint8_t *cam_const, Layout *L)
{
    L->cam_ldim.assign(C + 1, 0); L->cam_off.assign(C + 1, 0); L->colmap.assign((size_t)6 * (C + 1), 0);
    int tot = 0;
    for (int c = 0; c < C; ++c) {
        const uint8_t *cc = cam_const + 7 * c;
        int n = 0;
        int8_t *cm = L->colmap.data() + 6 * c;
        if (model == OSFM_BA_MODEL_QUATERNION) {
            if (!cc[0]) { cm[n++] = 0; cm[n++] = 1; cm[n++] = 2; }
            for (int s = 4; s < 7; ++s) if (!cc[s]) cm[n++] = (int8_t)(s - 1);   // full cols 3,4,5
        } else {
            for (int s = 0; s < 6; ++s) if (!cc[s]) cm[n++] = (int8_t)s;
        }
        L->cam_ldim[c] = n; L->cam_off[c] = tot; tot += n;
    }
    L->nc = tot;
}

int upload_camera_layout(const Layout &L, int C, hipStream_t s, DeviceProblem *D)
{
    OSFM_RETURN_IF(upload(D->cam_ldim, L.cam_ldim.data(), (size_t)C, s));
    OSFM_RETURN_IF(upload(D->cam_off, L.cam_off.data(), (size_t)C, s));
    OSFM_RETURN_IF(upload(D->colmap, L.colmap.data(), (size_t)6 * C, s));
    OSFM_RETURN_IF(D->scale_c.alloc((size_t)L.nc * 8));
    launch_fill(D->scale_c.as<double>(), (size_t)L.nc, 1.0, s);
    return OSFM_OK;
}

int finish_device_problem(int model, int C, int M, int O, int nc, double huber, int pdim, hipStream_t s, DeviceProblem *D)
{
    OSFM_RETURN_IF(D->cams[1].alloc((size_t)7 * C * 8));
    OSFM_RETURN_IF(D->points[1].alloc((size_t)4 * M * 8));
    // obs_point is non-decreasing, i.e. it is the expansion of pt_start
    OSFM_RETURN_IF(D->obs_pt.alloc((size_t)O * sizeof(int32_t)));
    launch_expand_points(D->pt_start.as<int32_t>(), M, D->obs_pt.as<int32_t>(), s);
    OSFM_RETURN_IF(D->scale_p.alloc((size_t)3 * M * 8));
    launch_fill(D->scale_p.as<double>(), (size_t)3 * M, 1.0, s);
    BaDev &d = D->dev;
    memset(&d, 0, sizeof(d));          // lm == nullptr: the plain pointers below are used as they are
    d.model = model; d.C = C; d.M = M; d.O = O; d.nc = nc; d.pdim = pdim;
    d.cams = D->cams[0].as<double>(); d.points = D->points[0].as<double>();
    d.obs_xy = D->obs_xy.as<double>(); d.obs_cam = D->obs_cam.as<int32_t>(); d.obs_pt = D->obs_pt.as<int32_t>();
    d.pt_start = D->pt_start.as<int32_t>(); d.img_w = D->img_w.as<int32_t>(); d.img_h = D->img_h.as<int32_t>();
    d.cam_ldim = D->cam_ldim.as<int32_t>(); d.cam_off = D->cam_off.as<int32_t>();
    d.cam_colmap = D->colmap.as<int8_t>();
    d.scale_c = D->scale_c.as<double>(); d.scale_p = D->scale_p.as<double>();
    d.huber = huber;
    return OSFM_OK;
}

namespace { struct SubArray { void *ptr; template <typename T> T *as() const { return static_cast<T *>(ptr); } }; }    // a piece of a DevArray

int ba_solve_core(DeviceProblem &D, const osfm_ba_options &o, StreamLease &sg, int64_t pair_bound, osfm_ba_summary *sum, int *cur_out)
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (o.verbose >= 2)
            fprintf(stderr, "[osfm ba]   %-16s %8.3f ms\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    hipStream_t s = sg.s;
    BaDev &d = D.dev;
    const int C = d.C, M = d.M, pdim = d.pdim;
    int nc = d.nc;
    // observation windows of the per-point passes (ba_kernels.h): one workgroup, one partial slot each.  Laid out
    // by a kernel queued in front of the pair lists; the count of windows that need the other kernels comes back
    // with the pair lists' own synchronisation
    if (nc >= (1 << 24)) { set_error("ba_solve: more than 2^24 camera unknowns"); return OSFM_E_ARG; }
    const int max_slots = o.max_num_iterations + 3;
    OSFM_RETURN_IF(sg.set->ensure_pinned((size_t)max_slots * sizeof(LmDev)));
    LmDev *h_state = static_cast<LmDev *>(sg.set->pinned);
    ObsWindows win;
    win.num = obs_windows_count(d.O);
    DevArray win_desc, win_over, win_ok, win_count, obs_lay;
    OSFM_RETURN_IF(win_desc.alloc((size_t)win.num * sizeof(WinDesc)));
    OSFM_RETURN_IF(win_over.alloc((size_t)win.num * 4));
    OSFM_RETURN_IF(win_ok.alloc((size_t)win.num * 4));
    OSFM_RETURN_IF(win_count.alloc(16));
    OSFM_RETURN_IF(obs_lay.alloc((size_t)std::max(d.O, 1) * 4));
    OSFM_HIP_CHECK(hipMemsetAsync(win_count.ptr, 0, 16, s));
    int32_t *h_over = reinterpret_cast<int32_t *>(&h_state[max_slots - 1]);
    auto lay_out_windows = [&]() -> int {
        // (obs_lay holds the cameras' offsets: behind the choice of the elimination order where there is one)
        launch_obs_windows(d, win.num, win_desc.as<WinDesc>(), win_over.as<int32_t>(), win_ok.as<int32_t>(), win_count.as<int32_t>(),
            obs_lay.as<int32_t>(), s);
        OSFM_HIP_CHECK(hipMemcpyAsync(h_over, win_count.ptr, 4, hipMemcpyDeviceToHost, s));
        return OSFM_OK;
    };
    // An elimination order for the reduced camera system is looked for where the system has at least eight block
    // columns (ba_order.hip; OSFM_BA_ORDER=0: never): it needs the camera pairs, so the pair lists come first then.
    const int order_policy = getenv("OSFM_BA_ORDER") ? atoi(getenv("OSFM_BA_ORDER")) : 1;        // (read per solve: A/B runs and tests)
    const bool may_order = order_policy != 0 && pdim != 0 && cholesky_padded_dim(std::max(nc, 1)) / 32 >= 8 &&
        getenv("OSFM_BA_CHOLESKY_STEPS") == nullptr;
    if (!may_order) OSFM_RETURN_IF(lay_out_windows());
    win.desc = win_desc.as<WinDesc>(); win.over_list = win_over.as<int32_t>(); win.ok_list = win_ok.as<int32_t>();
    win.obs_lay = obs_lay.as<int32_t>();
    // camera-pair lists of the Schur complement, built on the device
    PairListsDev PL;
    const int dense_policy = getenv("OSFM_BA_DENSE_SCHUR") ? atoi(getenv("OSFM_BA_DENSE_SCHUR")) : -1;     // (0 / 1: A/B runs and tests)
    OSFM_RETURN_IF(pair_lists_build(d, pdim != 0, std::max<int64_t>(pair_bound, 1), &PL, s, dense_policy));
    const int num_pairs = PL.num_pairs;
    sum->num_pair_entries = PL.dense ? PL.num_entries_all : PL.num_entries;
    ReducedOrder ord;
    DevArray ord_nz, ord_ptiles, ord_pad;
    std::vector<uint32_t> h_keys;
    std::vector<int32_t> h_ldim;
    if (may_order) {
        if (!PL.dense && num_pairs > 0) {
            // the camera pairs that share a track (the unique keys of the lists) and the cameras' block sizes
            h_keys.resize((size_t)num_pairs); h_ldim.resize((size_t)C);
            OSFM_HIP_CHECK(hipMemcpyAsync(h_keys.data(), PL.unique.ptr, (size_t)num_pairs * 4, hipMemcpyDeviceToHost, s));
            OSFM_HIP_CHECK(hipMemcpyAsync(h_ldim.data(), d.cam_ldim, (size_t)C * 4, hipMemcpyDeviceToHost, s));
            OSFM_HIP_CHECK(hipStreamSynchronize(s));
            std::vector<std::pair<int, int>> cpairs((size_t)num_pairs);
            const uint32_t g = (uint32_t)PL.group, Cu = (uint32_t)C;
            for (int i = 0; i < num_pairs; ++i) {
                const uint32_t key = h_keys[i];
                cpairs[i] = {(int)((key / (Cu * g)) * g + key % g), (int)((key / g) % Cu)};
            }
            if (choose_reduced_order(C, h_ldim.data(), cpairs, &ord)) {
                OSFM_HIP_CHECK(hipMemcpyAsync(const_cast<int32_t *>(d.cam_off), ord.cam_off.data(), (size_t)C * 4, hipMemcpyHostToDevice, s));
                nc = ord.span;
                d.nc = nc;
                OSFM_RETURN_IF(D.scale_c.alloc((size_t)nc * 8));
                launch_fill(D.scale_c.as<double>(), (size_t)nc, 1.0, s);
                d.scale_c = D.scale_c.as<double>();
                OSFM_RETURN_IF(upload(ord_nz, ord.nz.data(), ord.nz.size(), s));
                OSFM_RETURN_IF(upload(ord_ptiles, ord.ptiles.data(), ord.ptiles.size(), s));
                OSFM_RETURN_IF(upload(ord_pad, ord.pad.data(), ord.pad.size(), s));
            }
        }
        OSFM_RETURN_IF(lay_out_windows());
    }
    sum->order_arcs = ord.active ? ord.arcs : 0;
    sum->chain_blocks_natural = ord.chain_natural;
    sum->chain_blocks = ord.active ? ord.chain_ordered : ord.chain_natural;
    FlowPattern pattern;
    if (ord.active) { pattern.nz = ord_nz.as<unsigned long long>(); pattern.ptiles = ord_ptiles.as<int32_t>(); pattern.num_ptiles = (int)ord.ptiles.size(); }
    // dense visibility: the point part of the Schur complement is a product of two dense matrices (ba_dense.hip)
    DevArray dense_z, dense_w, dense_partial;
    bool dense_first = true;
    if (PL.dense) {
        const size_t zw = (size_t)schur_dense_rows(nc) * schur_dense_cols(M) * 8;
        OSFM_RETURN_IF(dense_z.alloc(zw));
        OSFM_RETURN_IF(dense_w.alloc(zw));
        OSFM_RETURN_IF(dense_partial.alloc(schur_dense_partial_bytes(nc, M)));
    }
    OSFM_HIP_CHECK(hipStreamSynchronize(s));     // (pair_lists_build has synchronised: this returns at once)
    win.num_over = *h_over;
    lap("pair lists (device)");

    const int blocksM = win.num;
    const int N = cholesky_padded_dim(std::max(nc, 1));
    DevArray Lmat;
    DevArray obsrec, diag_c, diag_p, vinv, ge, S, Ldiag, y_c, partA, partB, partC, scalars;
    OSFM_RETURN_IF(obsrec.alloc((size_t)std::max(d.O, 1) * kObsRec * 8));
    OSFM_RETURN_IF(diag_c.alloc((size_t)nc * 8));
    OSFM_RETURN_IF(diag_p.alloc((size_t)3 * M * 8));
    OSFM_RETURN_IF(vinv.alloc((size_t)9 * M * 8));
    OSFM_RETURN_IF(ge.alloc((size_t)3 * M * 8));
    const size_t s_elems = (size_t)(N + 32) * N;
    OSFM_RETURN_IF(S.alloc(s_elems * 8));
    OSFM_RETURN_IF(Lmat.alloc(s_elems * 8));
    OSFM_RETURN_IF(Ldiag.alloc((size_t)N * 32 * 8));
    OSFM_RETURN_IF(y_c.alloc((size_t)N * 8));
    OSFM_RETURN_IF(partA.alloc((size_t)3 * blocksM * 8));
    OSFM_RETURN_IF(partB.alloc((size_t)3 * blocksM * 8));
    OSFM_RETURN_IF(partC.alloc((size_t)blocksM * 8));
    OSFM_RETURN_IF(scalars.alloc(16 * 8));
    // what has to start at zero -- the cameras' partials and gradient norms, the Cholesky's info word, the tickets of
    // the fused tails -- is one block and one memset (four of them were 30 us of a 3-camera adjustment's set-up)
    auto r256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t z_part = 0, z_gmax = z_part + r256((size_t)2 * std::max(C, 1) * 8), z_info = z_gmax + r256((size_t)std::max(C, 1) * 8),
                 z_tick = z_info + 256, z_end = z_tick + r256(2 * lm_ticket_bytes());
    DevArray zeros;
    OSFM_RETURN_IF(zeros.alloc(z_end));
    OSFM_HIP_CHECK(hipMemsetAsync(zeros.ptr, 0, z_end, s));
    const SubArray part_cam{zeros.as<char>() + z_part}, gmax_cam{zeros.as<char>() + z_gmax}, info{zeros.as<char>() + z_info},
              tickets{zeros.as<char>() + z_tick};
    DevArray flow_flags, flow_mailbox;        // hand-off flags of the one-launch Cholesky, zeroed once per solve
    // (a system of one block never takes the one-launch form: chol_small_kernel)
    const bool use_flow = getenv("OSFM_BA_CHOLESKY_STEPS") == nullptr && N > 32;
    if (use_flow) {
        OSFM_RETURN_IF(flow_flags.alloc((size_t)chol_flow_flag_count(std::max(nc, 1)) * 4));
        OSFM_HIP_CHECK(hipMemsetAsync(flow_flags.ptr, 0, (size_t)chol_flow_flag_count(std::max(nc, 1)) * 4, s));
        OSFM_RETURN_IF(flow_mailbox.alloc(chol_flow_mailbox_bytes(std::max(nc, 1))));
    }
    int flow_epoch = 0;
    // the cameras' derived tables, one per iterate buffer: whoever writes cameras writes their rows
    OSFM_RETURN_IF(D.camder[0].alloc((size_t)std::max(C, 1) * kCamDer * 8));
    OSFM_RETURN_IF(D.camder[1].alloc((size_t)std::max(C, 1) * kCamDer * 8));
    d.camder2[0] = D.camder[0].as<double>(); d.camder2[1] = D.camder[1].as<double>();
    d.camder = d.camder2[0];
    launch_cam_derive(d, D.cams[0].as<double>(), d.camder2[0], s);

    PointPassArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.min_diag = o.min_lm_diagonal; pa.max_diag = o.max_lm_diagonal;
    pa.diag_p = diag_p.as<double>(); pa.vinv = vinv.as<double>(); pa.ge = ge.as<double>();
    pa.scale_p_out = D.scale_p.as<double>(); pa.partials = partA.as<double>();
    pa.obsrec = obsrec.as<double>();
    PairPassArgs qa;
    memset(&qa, 0, sizeof(qa));
    qa.min_diag = o.min_lm_diagonal; qa.max_diag = o.max_lm_diagonal;
    qa.num_pairs = num_pairs;
    qa.dense = PL.dense ? 1 : 0;
    qa.pair_key = PL.unique.as<uint32_t>(); qa.pair_start = PL.starts.as<int32_t>();
    qa.entries = PL.entries.as<uint64_t>();
    qa.chunk_start = PL.chunk_start.as<int32_t>(); qa.max_chunks = PL.max_chunks; qa.chunk = PL.chunk;
    qa.chunk_pair = PL.chunk_pair.as<int32_t>();
    qa.chunk_desc = PL.chunk_desc.as<PairChunkDesc>();
    qa.pair_ticket = PL.pair_ticket.as<int32_t>();
    qa.chunk_partials = PL.chunk_partials.as<double>();
    qa.gmax_out = gmax_cam.as<double>();
    qa.vinv = vinv.as<double>(); qa.ge = ge.as<double>(); qa.obsrec = obsrec.as<double>();
    qa.diag_c = diag_c.as<double>(); qa.scale_c_out = D.scale_c.as<double>();
    qa.S = S.as<double>(); qa.ldS = N; qa.rhs = S.as<double>() + (size_t)N * N;

    // ---- Levenberg-Marquardt, control on the device ---------------------------------
    // Every iteration is the same fixed sequence of launches; what they do (which iterate
    // is current, the radius, refresh of the LM diagonal, nothing at all once the solve has
    // stopped) is read from the LmDev state that ba_lm_decide / ba_lm_post keep.  The host
    // enqueues iteration i + 1 before it looks at the state iteration i left behind, so
    // the stream never waits for it; the one iteration enqueued past the end is a row of
    // kernels that return at once.
    DevArray lmdev;
    OSFM_RETURN_IF(lmdev.alloc(sizeof(LmDev)));
    LmDev init;
    memset(&init, 0, sizeof(init));
    init.radius = o.initial_trust_region_radius; init.decrease_factor = 2.0;
    init.update_diag = 1; init.want_gradient = 1; init.term = OSFM_BA_NO_CONVERGENCE;
    LmDev *lm = lmdev.as<LmDev>();
    // verbose: up to 8 timing events per iteration slot, plus what a given-up Cholesky launch adds (one more
    // linearisation and up to two iterations repeated launch by launch: 20 events) -- three slots of slack
    OSFM_RETURN_IF(sg.set->ensure_events((size_t)max_slots + (o.verbose ? 8 * ((size_t)max_slots + 3) : 0)));
    memcpy(&h_state[0], &init, sizeof(init));
    OSFM_HIP_CHECK(hipMemcpyAsync(lm, &h_state[0], sizeof(LmDev), hipMemcpyHostToDevice, s));
    d.cams2[0] = D.cams[0].as<double>(); d.cams2[1] = D.cams[1].as<double>();
    d.points2[0] = D.points[0].as<double>(); d.points2[1] = D.points[1].as<double>();
    // iteration 0 (Jacobi scaling) runs on the plain pointers, the state comes after it
    LmParams prm;
    prm.function_tolerance = o.function_tolerance; prm.gradient_tolerance = o.gradient_tolerance;
    prm.parameter_tolerance = o.parameter_tolerance; prm.min_relative_decrease = o.min_relative_decrease;
    prm.max_radius = o.max_trust_region_radius; prm.min_radius = o.min_trust_region_radius;
    prm.max_iterations = o.max_num_iterations; prm.max_invalid_steps = o.max_consecutive_invalid_steps;
    LmScratch sc;
    sc.partA = partA.as<double>(); sc.partB = partB.as<double>(); sc.partC = partC.as<double>();
    sc.part_cam = part_cam.as<double>(); sc.gmax_cam = gmax_cam.as<double>();
    sc.chol_info = info.as<int32_t>(); sc.blocksM = blocksM; sc.C = std::max(C, 1);
    // one block of unknowns: the solve and the candidate cameras are one launch, and the
    // decide kernel clears the system for the next linearisation
    const bool small = nc > 0 && N == 32;
    sc.reset_S = small ? S.as<double>() : nullptr; sc.reset_n = nc; sc.reset_N = N;

    // kernel-family timing (o.verbose): event pairs per iteration, read after the loop
    std::vector<hipEvent_t> &evs = sg.set->events;
    size_t ev_next = (size_t)max_slots;
    std::vector<std::pair<size_t, int>> ev_pairs;     // (first event of the pair, family)
    auto tic = [&](int family) -> int {
        if (!o.verbose || ev_next + 1 >= evs.size()) return OSFM_OK;      // out of events: the span goes untimed
        OSFM_HIP_CHECK(hipEventRecord(evs[ev_next], s));
        ev_pairs.push_back({ev_next, family});
        return OSFM_OK;
    };
    auto toc = [&]() -> int {
        if (!o.verbose || ev_next + 1 >= evs.size()) return OSFM_OK;
        OSFM_HIP_CHECK(hipEventRecord(evs[ev_next + 1], s));
        ev_next += 2;
        return OSFM_OK;
    };
    int n_lin = 0;
    // The LM control rides in the tails of the passes (the last workgroup of the back pass decides, the last one of
    // the pair pass finalises the iteration): two launches of one workgroup less per iteration; and while the camera
    // tables with the candidates fit LDS the back pass also makes the candidate cameras and evaluates the
    // candidate's cost -- two more launches and a pass over the observations less.
    const bool post_fused = num_pairs > 0 && getenv("OSFM_BA_SEPARATE_POST") == nullptr;
    enum { kPostNone = 0, kPostInitial = 1, kPostLoop = 2 };

    auto linearize = [&](bool reset, int post, LmDev *host_out) -> int {
        pa.mode = kPassNormal; qa.mode = kPassNormal;
        OSFM_RETURN_IF(tic(0));
        launch_point_pass(d, pa, win, s);
        OSFM_RETURN_IF(toc());
        if (reset) {
            launch_reset_system(S.as<double>(), s_elems, N, nc, N, s);
            if (ord.active) launch_padding_diagonal(S.as<double>(), N, ord_pad.as<int32_t>(), (int)ord.pad.size(), s);
        }
        if (PL.dense) {
            launch_schur_dense(d, obsrec.as<double>(), win.obs_lay, dense_z.as<double>(), dense_w.as<double>(), dense_partial.as<double>(),
                S.as<double>(), N, dense_first, s);
            dense_first = false;
        }
        memset(&qa.post, 0, sizeof(qa.post));
        if (post != kPostNone && post_fused) {
            qa.post.lm = lm; qa.post.prm = prm; qa.post.sc = sc; qa.post.host_out = host_out;
            qa.post.ticket = tickets.as<int32_t>() + lm_ticket_bytes() / 4; qa.post.initial = post == kPostInitial; qa.post.enabled = 1;
        }
        OSFM_RETURN_IF(tic(1));
        launch_pair_pass(d, qa, s);
        OSFM_RETURN_IF(toc());
        if (post != kPostNone && !post_fused) launch_lm_post(lm, prm, sc, post == kPostInitial, host_out, s);
        n_lin++;
        return OSFM_OK;
    };

    // ---- iteration 0: Jacobi scaling from the unscaled column norms ---------
    if (o.jacobi_scaling) {
        pa.mode = kPassScaleInit; qa.mode = kPassScaleInit;
        pa.radius = qa.radius = o.initial_trust_region_radius;
        launch_point_pass(d, pa, win, s);
        launch_pair_pass(d, qa, s);
        OSFM_HIP_CHECK(hipGetLastError());
    }
    d.lm = lm;
    const bool fused = C > 0 && getenv("OSFM_BA_SEPARATE_BACK") == nullptr;
    lap("alloc + lists up");
    OSFM_RETURN_IF(linearize(true, kPostInitial, nullptr));
    OSFM_HIP_CHECK(hipGetLastError());
    lap("first linearize");
    const auto t_loop = std::chrono::steady_clock::now();

    // Small systems (a handful of cameras: the local adjustments of the incremental
    // reconstruction) are bound by launch latency: the host runs a whole iteration ahead of
    // what it knows and pays one row of do-nothing kernels at the end.  Large ones are bound
    // by the device: there the host waits for the decision of iteration i (it has the
    // linearisation of i still queued behind it, so the device does not idle) and never
    // enqueues the Cholesky of an iteration that will not happen.
    const bool eager = N / 32 <= 4;
    LmDev fin;
    memset(&fin, 0, sizeof(fin));
    bool flow_now = use_flow;            // the one-launch Cholesky, until a launch of it had to be given up
    int restarts = 0;
    for (int it = 0; it < max_slots - 2; ++it) {
        const int slot = it + 1;          // h_state[slot]: the state this iteration leaves
        OSFM_RETURN_IF(tic(2));
        // the launch-per-column form works in place: the system has to be cleared before it is accumulated again
        // (the one-launch form only reads it, and the pair pass overwrites every block it owns)
        bool consumed = false;
        if (small) launch_small_solve(S.as<double>(), nc, Ldiag.as<double>(), y_c.as<double>(), info.as<int>(), d, part_cam.as<double>(), s);
        else if (nc > 0) consumed = launch_cholesky_solve(S.as<double>(), Lmat.as<double>(), nc, Ldiag.as<double>(), y_c.as<double>(), info.as<int>(), lm, s,
            flow_now ? flow_flags.as<int>() : nullptr, ++flow_epoch, flow_now ? flow_mailbox.as<double>() : nullptr, pattern) == 0;
        OSFM_RETURN_IF(toc());
        OSFM_RETURN_IF(tic(3));
        BackPassArgs ba;
        memset(&ba, 0, sizeof(ba));
        ba.y_c = y_c.as<double>(); ba.vinv = vinv.as<double>(); ba.ge = ge.as<double>(); ba.obsrec = obsrec.as<double>();
        ba.points_out = nullptr; ba.partials = partB.as<double>();
        if (fused) {
            ba.fused = 1; ba.cost_partials = partC.as<double>();
            ba.decide.lm = lm; ba.decide.prm = prm; ba.decide.sc = sc; ba.decide.host_out = eager ? nullptr : &h_state[slot];
            ba.decide.ticket = tickets.as<int32_t>(); ba.decide.enabled = 1;
            if (!small) launch_cam_update(d, y_c.as<double>(), nullptr, nullptr, part_cam.as<double>(), s);
            launch_back_pass(d, ba, win, s);
            OSFM_RETURN_IF(toc());
        } else {
            if (!small) launch_cam_update(d, y_c.as<double>(), nullptr, nullptr, part_cam.as<double>(), s);
            launch_back_pass(d, ba, win, s);
            OSFM_RETURN_IF(toc());
            launch_cost_pass(d, nullptr, nullptr, partC.as<double>(), win, s);
            // the kernels write the state they leave straight into the host's slot
            launch_lm_decide(lm, prm, sc, eager ? nullptr : &h_state[slot], s);
        }
        if (!eager) OSFM_HIP_CHECK(hipEventRecord(evs[slot], s));
        OSFM_RETURN_IF(linearize(consumed, kPostLoop, eager ? &h_state[slot] : nullptr));
        OSFM_HIP_CHECK(hipGetLastError());
        int seen = -1;                    // the slot whose state the host has read in this round
        if (eager) {
            OSFM_HIP_CHECK(hipEventRecord(evs[slot], s));
            if (it >= 1) {
                // what iteration it - 1 left behind (this iteration is already queued after it)
                OSFM_HIP_CHECK(hipEventSynchronize(evs[slot - 1]));
                seen = slot - 1;
            }
        } else {
            OSFM_HIP_CHECK(hipEventSynchronize(evs[slot]));
            seen = slot;
        }
        if (seen >= 0 && h_state[seen].flow_aborted) {
            // The one-launch factorisation of iteration seen - 1 gave up (a wait outlasted its spin limit: its
            // workgroups were not all resident).  Nothing was decided from it -- the decision left the state as
            // it was and every kernel behind it returned at once -- so the system is linearised again at the same
            // iterate, with the same diagonal (no second finalisation: that is the iteration's, still to come),
            // and the iteration is repeated in the launch-per-column form, like the rest of the solve.
            if (!flow_now || ++restarts > 1) { set_error("ba_solve: the Cholesky launch was given up twice (device busy?)"); return OSFM_E_DEVICE; }
            flow_now = false;
            sum->flow_fallbacks++;
            OSFM_HIP_CHECK(hipStreamSynchronize(s));
            launch_lm_clear_abort(lm, s);
            OSFM_RETURN_IF(linearize(!small, kPostNone, nullptr));
            it = seen - 2;                // the loop's increment makes it seen - 1: that iteration again
            continue;
        }
        if (seen >= 0 && h_state[seen].stop) break;
    }
    OSFM_HIP_CHECK(hipMemcpyAsync(&h_state[max_slots - 1], lm, sizeof(LmDev), hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    fin = h_state[max_slots - 1];
    if (fin.nonfinite) { set_error("ba_solve: non-finite initial cost"); return OSFM_E_NUMERIC; }
    const int cur = fin.cur;
    const double x_cost = fin.x_cost;
    const int iteration = fin.iteration, term = fin.term;
    sum->initial_cost = fin.initial_cost;
    sum->num_successful_steps = fin.num_success;
    sum->num_unsuccessful_steps = fin.num_unsuccess;
    double t_point = 0, t_pair = 0, t_chol = 0, t_back = 0;
    for (auto &pr : ev_pairs) {
        float ms = 0.f;
        if (pr.first + 1 >= ev_next + 1) continue;
        OSFM_HIP_CHECK(hipEventElapsedTime(&ms, evs[pr.first], evs[pr.first + 1]));
        (pr.second == 0 ? t_point : pr.second == 1 ? t_pair : pr.second == 2 ? t_chol : t_back) += ms;
    }

    sum->lm_loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
    lap("LM loop");
    *cur_out = cur;
    sum->final_cost = x_cost;
    sum->num_iterations = iteration;
    sum->termination = term;
    sum->point_pass_ms = t_point; sum->pair_pass_ms = t_pair; sum->cholesky_ms = t_chol; sum->back_pass_ms = t_back;
    sum->linearizations = fin.num_success + fin.num_unsuccess + 1;   // the speculative ones past the end do nothing
    (void)n_lin;
    return OSFM_OK;
}

}  // namespace osfm

extern "C" {

int osfm_ba_debug_chol_trace(int enable, int64_t *stamps)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("ba_debug_chol_trace: no HIP device available"); return OSFM_E_DEVICE; }
    OSFM_HIP_CHECK(hipDeviceSynchronize());
    long long *buf = chol_flow_trace_buffer(1);
    if (!buf) { set_error("ba_debug_chol_trace: no memory for the trace"); return OSFM_E_DEVICE; }
    if (stamps) OSFM_HIP_CHECK(hipMemcpy(stamps, buf, 161 * 32 * 8, hipMemcpyDeviceToHost));
    if (!enable) chol_flow_trace_buffer(0);
    return OSFM_OK;
}

int osfm_ba_debug_flow_spin_limit(int limit)
{
    chol_flow_set_spin_limit(limit);
    return OSFM_OK;
}

int osfm_ba_debug_order(int num_cameras, const int32_t *cam_ldim, int num_pairs, const int32_t *pairs, int32_t *cam_off,
    uint64_t *blocks, int blocks_capacity, int32_t *info)
{
    if (num_cameras <= 0 || !cam_ldim || num_pairs < 0 || (num_pairs && !pairs) || !cam_off || !info) { set_error("ba_debug_order: bad arguments"); return OSFM_E_ARG; }
    std::vector<std::pair<int, int>> cp((size_t)num_pairs);
    for (int i = 0; i < num_pairs; ++i) {
        const int a = pairs[2 * i], b = pairs[2 * i + 1];
        if (a < 0 || a >= num_cameras || b < 0 || b >= num_cameras) { set_error("ba_debug_order: pair %d names camera %d / %d", i, a, b); return OSFM_E_ARG; }
        cp[i] = {std::max(a, b), std::min(a, b)};
    }
    ReducedOrder ord;
    const bool on = choose_reduced_order(num_cameras, cam_ldim, cp, &ord);
    int tot = 0;
    for (int c = 0; c < num_cameras; ++c) { cam_off[c] = on ? ord.cam_off[c] : tot; tot += cam_ldim[c]; }
    info[0] = on ? 1 : 0; info[1] = on ? ord.arcs : 0; info[2] = on ? ord.sep_cams : 0; info[3] = on ? ord.span : tot;
    info[4] = on ? ord.nblk : (tot + 31) / 32; info[5] = ord.chain_natural; info[6] = on ? ord.chain_ordered : ord.chain_natural;
    info[7] = on ? (int)ord.pad.size() : 0;
    if (blocks && on) {
        if (blocks_capacity < ord.nblk + 1) { set_error("ba_debug_order: %d rows of blocks, room for %d", ord.nblk + 1, blocks_capacity); return OSFM_E_CAPACITY; }
        for (size_t i = 0; i < ord.nz.size(); ++i) blocks[i] = ord.nz[i];
    }
    return OSFM_OK;
}

int osfm_ba_options_default(osfm_ba_options *o)
{
    if (!o) { set_error("ba_options_default: null"); return OSFM_E_ARG; }
    o->huber_delta = 1.0;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-10;
    o->max_num_iterations = 100;
    o->optimize_points = 1;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->jacobi_scaling = 1;
    o->max_consecutive_invalid_steps = 5;
    o->device = 0;
    o->verbose = 0;
    o->retriangulate_points = 0;
    o->reserved = 0;
    return OSFM_OK;
}

int osfm_ba_solve(const osfm_ba_problem *p, const osfm_ba_options *opt, osfm_ba_summary *sum)
{
    if (!sum) { set_error("ba_solve: null summary"); return OSFM_E_ARG; }
    memset(sum, 0, sizeof(*sum));
    const auto t_begin = std::chrono::steady_clock::now();
    OSFM_RETURN_IF(validate_header(p, "ba_solve"));
    osfm_ba_options o;
    if (opt) o = *opt; else osfm_ba_options_default(&o);
    // The sweep over the caller's observations (range / order / one-observation-per-camera checks, the points'
    // observation counts, the copy of the start points) takes as long as queueing the uploads does (0.4 ms each for
    // BASELINE config 4: pageable arrays are staged by the calling thread): for problems of that size it runs on a
    // thread of its own beside them.  What it finds is looked at before anything is computed.
    Layout L;
    const int C = p->num_cameras, M = p->num_points;
    std::vector<double> pts0;
    std::string sweep_error;
    auto sweep = [&]() -> int {
        L.pt_start.assign((size_t)M + 2, 0);
        const int rc = validate_observations(p, "ba_solve", L.pt_start.data());
        if (rc != OSFM_OK) { sweep_error = osfm_last_error(); return rc; }      // (the message is per thread)
        for (int j = 0; j < M; ++j) L.pt_start[j + 1] += L.pt_start[j];
        // the points the optimisation starts from (tracksBackup, bundle_adjustment.cpp:99)
        pts0.resize((size_t)4 * M);
        if (M) memcpy(pts0.data(), p->points, (size_t)4 * M * 8);
        return OSFM_OK;
    };
    bool beside = p->num_observations >= 100000;
    std::future<int> swept;
    if (beside) {
        // (no thread to be had: the sweep runs here, in front of the uploads, as for small problems)
        try { swept = std::async(std::launch::async, sweep); } catch (const std::exception &) { beside = false; }
    }
    if (!beside) OSFM_RETURN_IF(sweep());
    struct Joiner { std::future<int> &f; ~Joiner() { if (f.valid()) f.wait(); } } joiner{swept};     // never leave it running
    OSFM_RETURN_IF(select_device(o.device));

    auto lap = [&](const char *what) {
        if (o.verbose >= 2)
            fprintf(stderr, "[osfm ba] %-18s %8.3f ms\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    StreamGuard sg;
    OSFM_RETURN_IF(sg.acquire());
    hipStream_t s = sg.s;

    // (D is declared before L: the transfers queued from L's vectors are waited for before either goes)
    DeviceProblem D;
    OSFM_RETURN_IF(upload_caller_arrays(p, s, &D));
    lap("caller arrays queued");
    if (beside) {
        const int rc = swept.get();
        if (rc != OSFM_OK) {
            OSFM_HIP_CHECK(hipStreamSynchronize(s));       // the queued uploads, before their buffers go back to the pool
            set_error("%s", sweep_error.c_str());
            return rc;
        }
    }
    build_camera_layout(p->model, C, p->cam_const, &L);
    const int pdim = o.optimize_points ? 3 : 0;
    lap("layout");
    OSFM_RETURN_IF(upload_problem(p, L, o.huber_delta, pdim, s, &D, true));
    BaDev &d = D.dev;
    if (o.retriangulate_points && M > 0) {
        // triangulateTracks(cameras, localTracks, true) in front of the solve (bundle_adjustment.cpp:77-83)
        OSFM_HIP_CHECK(hipMemcpyAsync(D.points[1].ptr, D.points[0].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToDevice, s));
        launch_triangulate(d, D.points[1].as<double>(), nullptr, s);
        OSFM_HIP_CHECK(hipMemcpyAsync(D.points[0].ptr, D.points[1].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToDevice, s));
        OSFM_HIP_CHECK(hipMemcpyAsync(pts0.data(), D.points[1].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToHost, s));
    }
    lap("upload problem");

    int64_t bound = 0;
    for (int j = 0; j < M; ++j) {
        const int64_t l = L.pt_start[j + 1] - L.pt_start[j];
        bound += pdim ? l * l : l;
    }
    int cur = 0;
    OSFM_RETURN_IF(ba_solve_core(D, o, sg, bound, sum, &cur));
    // ---- write back the current iterate --------------------------------------
    if (C) OSFM_HIP_CHECK(hipMemcpyAsync(p->cam_params, D.cams[cur].ptr, (size_t)7 * C * 8, hipMemcpyDeviceToHost, s));
    if (M) OSFM_HIP_CHECK(hipMemcpyAsync(p->points, D.points[cur].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToHost, s));
    OSFM_HIP_CHECK(hipStreamSynchronize(s));
    double mx = 0.0, acc = 0.0;
    for (int j = 0; j < M; ++j) {       // bundle_adjustment.cpp:150-160
        double dd = 0.0;
        for (int i = 0; i < 4; ++i) { const double e = pts0[4 * j + i] - p->points[4 * j + i]; dd += e * e; }
        dd = std::sqrt(dd); mx = std::max(mx, dd); acc += dd;
    }
    sum->mean_point_change = M ? acc / M : 0.0;
    sum->max_point_change = mx;
    sum->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return OSFM_OK;
}

int osfm_ba_reprojection_errors(const osfm_ba_problem *p, int device, double *err, double *residuals)
{
    OSFM_RETURN_IF(validate_problem(p, "ba_reprojection_errors"));
    if (!err && !residuals) { set_error("ba_reprojection_errors: no output"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(select_device(device));
    Layout L;                 // declared before the stream lease: it must outlive the copies that read it
    build_layout(p, &L);
    StreamGuard sg;
    OSFM_RETURN_IF(sg.acquire());
    DeviceProblem D;
    OSFM_RETURN_IF(upload_problem(p, L, 1.0, 3, sg.s, &D));
    const int O = p->num_observations;
    DevArray d_err, d_res;
    OSFM_RETURN_IF(d_err.alloc((size_t)O * 8));
    OSFM_RETURN_IF(d_res.alloc((size_t)2 * O * 8));
    launch_reproj(D.dev, d_err.as<double>(), d_res.as<double>(), sg.s);
    OSFM_HIP_CHECK(hipGetLastError());
    if (err && O) OSFM_HIP_CHECK(hipMemcpyAsync(err, d_err.ptr, (size_t)O * 8, hipMemcpyDeviceToHost, sg.s));
    if (residuals && O) OSFM_HIP_CHECK(hipMemcpyAsync(residuals, d_res.ptr, (size_t)2 * O * 8, hipMemcpyDeviceToHost, sg.s));
    OSFM_HIP_CHECK(hipStreamSynchronize(sg.s));
    return OSFM_OK;
}

int osfm_ba_triangulate(const osfm_ba_problem *p, int device, uint8_t *point_valid)
{
    OSFM_RETURN_IF(validate_problem(p, "ba_triangulate"));
    OSFM_RETURN_IF(select_device(device));
    Layout L;                 // declared before the stream lease: it must outlive the copies that read it
    build_layout(p, &L);
    StreamGuard sg;
    OSFM_RETURN_IF(sg.acquire());
    DeviceProblem D;
    OSFM_RETURN_IF(upload_problem(p, L, 1.0, 3, sg.s, &D));
    const int M = p->num_points;
    DevArray d_valid;
    OSFM_RETURN_IF(d_valid.alloc((size_t)std::max(M, 1)));
    // points of tracks with fewer than two rays keep their input value
    OSFM_HIP_CHECK(hipMemcpyAsync(D.points[1].ptr, D.points[0].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToDevice, sg.s));
    launch_triangulate(D.dev, D.points[1].as<double>(), d_valid.as<uint8_t>(), sg.s);
    OSFM_HIP_CHECK(hipGetLastError());
    if (M) OSFM_HIP_CHECK(hipMemcpyAsync(p->points, D.points[1].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToHost, sg.s));
    if (point_valid && M) OSFM_HIP_CHECK(hipMemcpyAsync(point_valid, d_valid.ptr, (size_t)M, hipMemcpyDeviceToHost, sg.s));
    OSFM_HIP_CHECK(hipStreamSynchronize(sg.s));
    return OSFM_OK;
}

int osfm_filter_reprojection(const osfm_ba_problem *p, int device, double max_error,
    uint8_t *obs_keep, uint8_t *point_valid, double *err)
{
    OSFM_RETURN_IF(validate_problem(p, "filter_reprojection"));
    if (!obs_keep && p->num_observations > 0) { set_error("filter_reprojection: obs_keep is null"); return OSFM_E_ARG; }
    OSFM_RETURN_IF(select_device(device));
    Layout L;                 // declared before the stream lease: it must outlive the copies that read it
    build_layout(p, &L);
    StreamGuard sg;
    OSFM_RETURN_IF(sg.acquire());
    DeviceProblem D;
    OSFM_RETURN_IF(upload_problem(p, L, 1.0, 3, sg.s, &D));
    const int M = p->num_points, O = p->num_observations;
    DevArray d_valid, d_err, d_res;
    OSFM_RETURN_IF(d_valid.alloc((size_t)std::max(M, 1)));
    OSFM_RETURN_IF(d_err.alloc((size_t)std::max(O, 1) * 8));
    OSFM_RETURN_IF(d_res.alloc((size_t)2 * std::max(O, 1) * 8));
    // triangulate into the spare point array (tracks with fewer than two rays
    // keep their input point), then evaluate against the triangulated points
    OSFM_HIP_CHECK(hipMemcpyAsync(D.points[1].ptr, D.points[0].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToDevice, sg.s));
    launch_triangulate(D.dev, D.points[1].as<double>(), d_valid.as<uint8_t>(), sg.s);
    OSFM_HIP_CHECK(hipMemcpyAsync(D.points[0].ptr, D.points[1].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToDevice, sg.s));
    launch_reproj(D.dev, d_err.as<double>(), d_res.as<double>(), sg.s);
    OSFM_HIP_CHECK(hipGetLastError());
    std::vector<double> herr((size_t)std::max(O, 1));
    if (O) OSFM_HIP_CHECK(hipMemcpyAsync(herr.data(), d_err.ptr, (size_t)O * 8, hipMemcpyDeviceToHost, sg.s));
    if (M) OSFM_HIP_CHECK(hipMemcpyAsync(p->points, D.points[1].ptr, (size_t)4 * M * 8, hipMemcpyDeviceToHost, sg.s));
    if (point_valid && M) OSFM_HIP_CHECK(hipMemcpyAsync(point_valid, d_valid.ptr, (size_t)M, hipMemcpyDeviceToHost, sg.s));
    OSFM_HIP_CHECK(hipStreamSynchronize(sg.s));
    for (int k = 0; k < O; ++k) {
        obs_keep[k] = herr[k] < max_error ? 1 : 0;
        if (err) err[k] = herr[k];
    }
    return OSFM_OK;
}

}  // extern "C"
