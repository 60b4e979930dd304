"""GPU tests of hot path B through the C ABI, against the CPU oracle
(oracle/ba_oracle.c, PARITY UNPINNED w.r.t. Ceres -- see its header).

Tolerances (double precision on both sides; the GPU uses hand-derived
analytic Jacobians and a different summation order than the oracle's jets):
  residuals / reprojection errors   |diff| <= 1e-9 px
  triangulated points               |diff| <= 1e-9 (scene scale ~1)
  LM: identical iteration / accept counts and termination;
      final cost rel. diff <= 1e-9; cameras |diff| <= 1e-8; points <= 1e-7
"""
import numpy as np
import pytest

import oracle_lib
from orthosfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba():
    from orthosfm_amd import ba as m
    from orthosfm_amd import capi
    assert capi.device_count() >= 1
    return m


@pytest.mark.parametrize("model", [0, 1])
def test_reprojection_errors(ba, model):
    sc = synth.make_ba_scene(model, 11, 500, config_id=31)
    sc.points[:, 3] = 1.0 + 0.2 * np.cos(np.arange(500))
    sc.points[:, :3] *= sc.points[:, 3:4]
    fp = ba.FlatProblem.from_scene(sc)
    err, res = ba.reprojection_errors(fp)
    eres, eerr = oracle_lib.oracle_ba_residuals(sc)
    assert np.abs(res - eres).max() <= 1e-9
    assert np.abs(err - eerr).max() <= 1e-9


@pytest.mark.parametrize("model", [0, 1])
def test_triangulation(ba, model):
    sc = synth.make_ba_scene(model, 9, 400, config_id=32, noise_px=0.3)
    ref = sc.copy()
    evalid = oracle_lib.oracle_ba_triangulate(ref)
    fp = ba.FlatProblem.from_scene(sc)
    valid = ba.triangulate(fp)
    assert np.array_equal(valid, evalid)
    assert np.abs(fp.points - ref.points).max() <= 1e-9
    # tracks with a single ray are flagged invalid and keep their point
    sc2 = synth.make_ba_scene(model, 5, 50, config_id=33, min_len=1, max_len=2)
    p0 = sc2.points.copy()
    fp2 = ba.FlatProblem.from_scene(sc2)
    v2 = ba.triangulate(fp2)
    single = np.bincount(sc2.obs_point, minlength=50) < 2
    assert np.array_equal(v2 == 0, single)
    assert np.array_equal(fp2.points[single], p0[single])


def _compare_solve(ba, sc, **opt):
    ref = sc.copy()
    so = oracle_lib.oracle_ba_solve(ref, **opt)
    fp = ba.FlatProblem.from_scene(sc)
    s = ba.solve(fp, **opt)
    assert np.isclose(s.initial_cost, so.initial_cost, rtol=1e-11)
    assert s.num_iterations == so.num_iterations
    assert s.num_successful_steps == so.num_successful_steps
    assert s.num_unsuccessful_steps == so.num_unsuccessful_steps
    assert s.termination == so.termination
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * max(1.0, abs(so.final_cost))
    assert np.abs(fp.cam_params - ref.cam_params).max() <= 1e-8
    assert np.abs(fp.points - ref.points).max() <= 1e-7
    assert np.isclose(s.mean_point_change, so.mean_point_change, rtol=1e-6, atol=1e-9)
    return s, fp, ref


@pytest.mark.parametrize("model", [0, 1])
def test_lm_matches_oracle_small(ba, model):
    sc = synth.make_ba_scene(model, 8, 300, config_id=34)
    s, fp, _ = _compare_solve(ba, sc)
    assert s.final_cost < s.initial_cost
    assert np.array_equal(fp.cam_params[0], sc.gt_cams[0])         # fixed camera untouched


@pytest.mark.parametrize("model", [0, 1])
def test_lm_matches_oracle_medium(ba, model):
    """More cameras than one Cholesky block (nc > 32) and outliers (Huber active)."""
    sc = synth.make_ba_scene(model, 40, 4000, config_id=35)
    sc.obs_xy[::41] += 25.0
    _compare_solve(ba, sc)


@pytest.mark.parametrize("model", [0, 1])
def test_config1_suzanne_three_cameras(ba, model):
    """BASELINE configs[0] on the data it names: the reference's synthetic dataset
    (src/testbench/dataset_generation.cpp:40-93) -- one track per vertex of resources/Suzanne.ply seen
    by every camera, the first three of the test bench's 16 cameras (tests/golden/cfg1_suzanne.npz) --
    as one adjustment from a perturbed start: --solver=0 (quaternion cameras) and the Euler model.
    Same iteration counts / termination / cost as the oracle, cameras back at the ground truth."""
    sc = synth.make_suzanne_ba_scene(model, 3)
    assert sc.points.shape[0] == 7872 and sc.obs_xy.shape[0] == 3 * 7872
    s, fp, _ = _compare_solve(ba, sc)
    assert s.final_cost < 1e-3 * s.initial_cost
    assert np.array_equal(fp.cam_params[0], sc.gt_cams[0])
    # gauge: camera 0 is fixed at its true pose, so the others come back to theirs (the points are free)
    if model == 0:
        for c in range(3):
            q, g = fp.cam_params[c, :4], sc.gt_cams[c, :4]
            assert min(np.abs(q - g).max(), np.abs(q + g).max()) < 1e-4, c
    else:
        assert np.abs(fp.cam_params[:, :3] - sc.gt_cams[:, :3]).max() < 1e-4


def test_lm_constant_points_and_three_cameras(ba):
    """optimize_points = 0 (no Schur elimination) and the 3-camera local BA
    shape of the incremental pipeline (reconstruct.cpp:219)."""
    sc = synth.make_ba_scene(0, 6, 200, config_id=36, point_perturb=0.0, noise_px=0.2)
    s, fp, _ = _compare_solve(ba, sc, optimize_points=0)
    assert np.array_equal(fp.points, sc.points)
    sc3 = synth.make_ba_scene(0, 3, 500, config_id=37, min_len=3, max_len=3)
    _compare_solve(ba, sc3)


def test_euler_solver_dof_masks(ba):
    """setSolverType: solver 1 -> phi only, solver 2 -> phi+theta, solver 3 -> 5 DoF
    (OrthographicReconstructionAlgorithm.cpp:15-34)."""
    for free in (1, 2, 5):
        sc = synth.make_ba_scene(1, 7, 250, config_id=38 + free, euler_free=free)
        s, fp, _ = _compare_solve(ba, sc)
        const = sc.cam_const.astype(bool)
        assert np.array_equal(fp.cam_params[const], sc.cam_params[const])


def test_many_cameras(ba):
    """900 Euler cameras with one free angle each: a 900-unknown system, 216 KB of derived camera tables read
    through the caches."""
    sc = synth.make_ba_scene(1, 900, 3000, config_id=47, euler_free=1, min_len=3, max_len=6)
    _compare_solve(ba, sc, max_num_iterations=4)


@pytest.mark.parametrize("model", [0, 1])
def test_long_tracks_take_the_kernels_for_windows_of_any_size(ba, model):
    """The per-point passes give every observation a lane, 224 observations (plus what the last track adds) per
    workgroup; a window that ends in a track of more than 33 observations holds more than 256 and goes to the
    kernels with four lanes per track (ba_kernels.h: ObsWindows).  Tracks of 3..70 observations over 72 cameras:
    most windows do; the solve still is the oracle's."""
    sc = synth.make_ba_scene(model, 72, 1200, config_id=61 + model, min_len=3, max_len=70)
    ln = np.bincount(sc.obs_point, minlength=sc.points.shape[0])
    assert ln.max() > 60 and ln.min() <= 5
    _compare_solve(ba, sc, max_num_iterations=6)


def test_tracks_without_observations_keep_their_points(ba):
    """Tracks between others that have lost all observations: their points are parameters nobody moves."""
    sc = synth.make_ba_scene(0, 10, 900, config_id=63)
    drop = np.isin(sc.obs_point, np.arange(5, 900, 37))
    keep = ~drop
    sc.obs_xy, sc.obs_camera, sc.obs_point = sc.obs_xy[keep], sc.obs_camera[keep], sc.obs_point[keep]
    p0 = sc.points.copy()
    s, fp, _ = _compare_solve(ba, sc, max_num_iterations=5)
    assert np.array_equal(fp.points[5::37], p0[5::37])
    assert not np.array_equal(fp.points[6], p0[6])


def test_run_bundle_adjustment_semantics(ba):
    """The adapter reproduces runBundleAdjustment's quirks: viewID lookup,
    tracks without a point are skipped, in-place point update only without
    retriangulation (bundle_adjustment.cpp:71-83,103-123)."""
    sc = synth.make_ba_scene(0, 6, 120, config_id=45)
    cams = [ba.QuatCamera(view_id=10 + c, width=2048, height=2048, rotation=sc.cam_params[c, :4].copy(),
                          offset_x=sc.cam_params[c, 4], offset_y=sc.cam_params[c, 5], fixed=(c == 0))
            for c in range(6)]
    tracks = []
    for j in range(120):
        sel = np.nonzero(sc.obs_point == j)[0]
        fs = [ba.Feature(10 + int(sc.obs_camera[k]), k, sc.obs_xy[k, 0], sc.obs_xy[k, 1]) for k in sel]
        fs.append(ba.Feature(999, 0, 1.0, 2.0))            # view without a camera: skipped
        tracks.append(ba.Track(fs, sc.points[j].copy(), has_point=(j % 10 != 0)))
    p_before = [t.point.copy() for t in tracks]
    s = ba.run_bundle_adjustment(cams, tracks, None, True, False, verbose=False)
    assert s.final_cost < s.initial_cost
    for j, t in enumerate(tracks):
        moved = not np.array_equal(t.point, p_before[j])
        assert moved == t.has_point
    # equivalent flat problem through the oracle
    keep = np.array([j % 10 != 0 for j in range(120)])
    remap = -np.ones(120, dtype=np.int64)
    remap[keep] = np.arange(keep.sum())
    sel = keep[sc.obs_point]
    ref = sc.copy()
    ref.points = np.ascontiguousarray(sc.points[keep])
    ref.obs_xy = np.ascontiguousarray(sc.obs_xy[sel])
    ref.obs_camera = np.ascontiguousarray(sc.obs_camera[sel])
    ref.obs_point = np.ascontiguousarray(remap[sc.obs_point[sel]].astype(np.int32))
    so = oracle_lib.oracle_ba_solve(ref)
    assert s.num_iterations == so.num_iterations
    for c in range(6):
        assert np.abs(cams[c].params() - ref.cam_params[c]).max() <= 1e-8
    # retriangulate = True: cameras move, the caller's points do not
    cams2 = [ba.QuatCamera(view_id=10 + c, width=2048, height=2048, rotation=sc.cam_params[c, :4].copy(),
                           offset_x=sc.cam_params[c, 4], offset_y=sc.cam_params[c, 5], fixed=(c == 0))
             for c in range(6)]
    tracks2 = [ba.Track(list(t.features), p.copy(), t.has_point) for t, p in zip(tracks, p_before)]
    ba.run_bundle_adjustment(cams2, tracks2, None, True, True, verbose=False)
    assert all(np.array_equal(t.point, p) for t, p in zip(tracks2, p_before))
    assert any(not np.allclose(c.params(), sc.cam_params[i]) for i, c in enumerate(cams2) if i)


def test_global_ba_properties_config4_full_size(ba):
    """BASELINE config 4 at its stated size (200 quaternion cameras, 100k tracks,
    ~750k observations): too big for a quick oracle run, checked through properties
    -- convergence, cost far below the start, reprojection error near the noise
    level, ground truth recovered, bit-identical repeat (no floating-point atomics
    anywhere in the solve)."""
    sc = synth.make_ba_scene(0, 200, 100000, config_id=4)
    fp = ba.FlatProblem.from_scene(sc)
    s = ba.solve(fp)
    assert s.termination in (1, 2, 3)
    assert s.final_cost < 0.05 * s.initial_cost
    err, _ = ba.reprojection_errors(fp)
    assert np.median(err) < 1.5            # 0.5 px noise per axis
    # camera 0 is fixed at its ground-truth pose, so the gauge is pinned: the other
    # cameras come back to the ground truth (2 degrees / 0.01 off at the start)
    q, g = fp.cam_params[:, :4], sc.gt_cams[:, :4]
    ang = 2.0 * np.arccos(np.clip(np.abs((q * g).sum(1)) / np.linalg.norm(q, axis=1), 0, 1))
    assert np.rad2deg(ang).max() < 0.05
    assert np.abs(fp.cam_params[:, 4:6] - sc.gt_cams[:, 4:6]).max() < 1e-3
    fp2 = ba.FlatProblem.from_scene(sc)
    s2 = ba.solve(fp2)
    assert s2.final_cost == s.final_cost and s2.num_iterations == s.num_iterations
    assert np.array_equal(fp2.cam_params, fp.cam_params) and np.array_equal(fp2.points, fp.points)


def test_pair_list_bound_is_refused(ba):
    """The Schur pair lists are 32-bit: a problem whose sum of squared track lengths
    passes 2^31 - 1 is refused with OSFM_E_RANGE before anything is allocated
    (2100 tracks seen by each of 1024 cameras: 2100 * 1024^2 = 2^31 + 54.5 M; a point is
    observed at most once per camera, so long tracks need as many cameras)."""
    from orthosfm_amd import capi
    C, M = 1024, 2100
    cams = np.zeros((C, 7)); cams[:, 3] = 1.0; cams[:, 6] = 1.0
    const = np.zeros((C, 7), np.uint8); const[0] = 1; const[:, 6] = 1
    wh = np.full(C, 2048, np.int32)
    pts = np.zeros((M, 4)); pts[:, 3] = 1.0
    cam = np.tile(np.arange(C, dtype=np.int32), M)
    pt = np.repeat(np.arange(M, dtype=np.int32), C)
    fp = ba.FlatProblem(0, cams, const, wh, wh, pts, np.zeros((C * M, 2)), cam, pt)
    with pytest.raises(capi.OsfmError) as e:
        ba.solve(fp)
    assert e.value.status == capi.E_RANGE


def test_ba_error_behaviour(ba):
    from orthosfm_amd import capi
    sc = synth.make_ba_scene(0, 4, 20, config_id=46)
    fp = ba.FlatProblem.from_scene(sc)
    fp.obs_point[3], fp.obs_point[4] = fp.obs_point[4] + 5, fp.obs_point[3]     # not sorted
    with pytest.raises(capi.OsfmError) as e:
        ba.solve(fp)
    assert e.value.status == capi.E_ARG
    fp = ba.FlatProblem.from_scene(sc)
    fp.obs_camera[0] = 77
    with pytest.raises(capi.OsfmError):
        ba.solve(fp)


@pytest.mark.parametrize("model,retri", [(0, False), (1, True), (0, True)])
def test_cpp_adapter_equals_python_mirror(ba, model, retri, tmp_path):
    """orthosfm_amd/host/ba_hip_adapter.h (the C++ body of orthosfm::runBundleAdjustment,
    compiled in tests/host with test doubles of the reference classes) against the Python
    mirror on the same cameras and tracks: identical cameras and points, bit for bit."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "host", "ba_adapter_check")
    assert os.path.exists(exe), "tests/host/ba_adapter_check missing: run __graft_entry__.build()"
    sc = synth.make_ba_scene(model, 7, 160, config_id=81 + model)
    C = 7
    def cams():
        out = []
        for c in range(C):
            if model == 0:
                out.append(ba.QuatCamera(view_id=20 + c, width=2048, height=2048, rotation=sc.cam_params[c, :4].copy(),
                                         offset_x=sc.cam_params[c, 4], offset_y=sc.cam_params[c, 5], fixed=(c == 0)))
            else:
                out.append(ba.EulerCamera(20 + c, 2048, 2048, *sc.cam_params[c, :5], fixed=(c == 0)))
        return out
    starts = np.concatenate([[0], np.cumsum(np.bincount(sc.obs_point, minlength=160))])
    def tracks():
        out = []
        for j in range(160):
            feats = [ba.Feature(20 + int(sc.obs_camera[k]), int(k), float(np.float32(sc.obs_xy[k, 0])), float(np.float32(sc.obs_xy[k, 1])))
                     for k in range(starts[j], starts[j + 1])]
            if j % 9 == 4:
                feats.append(ba.Feature(999, 0, 5.0, 6.0))            # a view without camera
            out.append(ba.Track(feats, sc.points[j].copy(), j % 11 != 3))     # some tracks without a point
        return out
    py_c, py_t = cams(), tracks()
    ba.run_bundle_adjustment(py_c, py_t, None, True, retri, verbose=False)
    lines = [f"{model} {C} 160 {int(retri)}"]
    for c in cams():
        lines.append(f"{c.view_id} {c.width} {c.height} {int(c.fixed)} " + " ".join(repr(float(x)) for x in c.params()))
    for t in tracks():
        lines.append(f"{len(t.features)} {int(t.has_point)} " + " ".join(repr(float(x)) for x in t.point))
        for f in t.features:
            lines.append(f"{f.viewID} {f.localFeatureID} {float(np.float32(f.x))!r} {float(np.float32(f.y))!r}")
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cam_rows = [np.array(l.split()[1:], dtype=np.float64) for l in out.stdout.splitlines() if l.startswith("CAM")]
    pt_rows = [l.split()[1:] for l in out.stdout.splitlines() if l.startswith("PT")]
    assert len(cam_rows) == C and len(pt_rows) == 160
    for c, row in zip(py_c, cam_rows):
        assert np.array_equal(c.params()[:6], row[:6])
    for t, row in zip(py_t, pt_rows):
        assert bool(int(row[0])) == bool(t.has_point)
        if t.has_point:           # a track without a point has no value to compare (the double holds zeros)
            assert np.array_equal(np.asarray(t.point, dtype=np.float64), np.array(row[1:], dtype=np.float64))
    assert "Average point change" in out.stdout


def test_work_arrays_are_cached_and_can_be_handed_back(ba):
    """The per-call work arrays come from a per-device cache (osfm_common.h: DevicePool):
    a second identical solve allocates nothing new, results do not depend on it, and
    osfm_trim_device_memory returns what is cached to the driver."""
    from orthosfm_amd import capi, pipeline
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, 12, 4000, config_id=1)
    pipeline.join_background()               # a matcher an earlier job released on a thread is gone by now
    capi.trim_device_memory()
    free0, _ = capi.device_memory(0)
    a = ba.FlatProblem.from_scene(sc)
    s1 = ba.solve(a, max_num_iterations=6)
    free1, _ = capi.device_memory(0)
    b = ba.FlatProblem.from_scene(sc)
    s2 = ba.solve(b, max_num_iterations=6)
    free2, _ = capi.device_memory(0)
    assert free1 < free0                     # the first call's arrays stay cached ...
    assert free2 == free1                    # ... and serve the second call
    assert s1.final_cost == s2.final_cost and np.array_equal(a.cam_params, b.cam_params)
    released = capi.trim_device_memory(0)
    free3, _ = capi.device_memory(0)
    # (the driver accounts whole pages, so its numbers and the pool's byte count differ slightly)
    assert released > 0 and free3 > free1 and free3 >= free0 - (1 << 22)
    assert capi.trim_device_memory() == 0    # nothing left


def test_concurrent_solves_share_the_device(ba):
    """Two host threads solving at once (their own streams): the one-launch Cholesky needs all its workgroups
    resident, so such launches are chained on the device instead of interleaving -- every solve finishes and
    equals the serial one bit for bit."""
    import threading
    sc = synth.make_ba_scene(0, 60, 6000, config_id=36)          # 300 camera unknowns: ten block columns
    ref = ba.FlatProblem.from_scene(sc.copy())
    s0 = ba.solve(ref)
    out = [None] * 6
    def work(k):
        fp = ba.FlatProblem.from_scene(sc.copy())
        s = ba.solve(fp)
        out[k] = (s.num_iterations, s.final_cost, fp.cam_params.copy(), s.termination)
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(6):
        assert out[k] is not None
        assert out[k][0] == s0.num_iterations and out[k][1] == s0.final_cost and out[k][3] == s0.termination
        assert np.array_equal(out[k][2], ref.cam_params)


def _solve_with_env(ba, sc, env=None):
    import os
    fp = ba.FlatProblem.from_scene(sc)
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        s = ba.solve(fp)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return s, fp


@pytest.mark.parametrize("model,cams,pts", [(0, 61, 6000), (0, 200, 30000), (1, 500, 60000)])
def test_one_launch_cholesky_equals_launch_per_column(ba, model, cams, pts):
    """The same 300-, 995- and 2495-unknown reduced systems (60 / 199 free quaternion cameras, 499 Euler
    cameras with every angle free: BASELINE configs[3] and configs[4]) solved with the one-launch Cholesky
    (workgroups handing tiles to each other; 78 block columns share the workgroups of the device) and with
    one launch per block column: same LM trajectory -- iteration / accept counts, termination -- and the
    final cost to 1e-12 (the two forms round differently: matrix-core tile products against FMA chains)."""
    sc = synth.make_ba_scene(model, cams, pts, config_id=41)
    s1, fp1 = _solve_with_env(ba, sc)
    s2, fp2 = _solve_with_env(ba, sc, {"OSFM_BA_CHOLESKY_STEPS": "1"})
    assert s1.flow_fallbacks == 0 and s2.flow_fallbacks == 0
    assert (s1.num_iterations, s1.num_successful_steps, s1.num_unsuccessful_steps, s1.termination) == \
        (s2.num_iterations, s2.num_successful_steps, s2.num_unsuccessful_steps, s2.termination)
    assert s1.num_iterations >= 3
    assert abs(s1.final_cost - s2.final_cost) <= 1e-12 * s2.final_cost
    assert np.abs(fp1.cam_params - fp2.cam_params).max() <= 1e-10
    assert np.abs(fp1.points - fp2.points).max() <= 1e-9


@pytest.mark.parametrize("model,cams,pts,lo,hi", [(0, 40, 900, 12, 24), (1, 150, 1200, 50, 90), (1, 30, 5000, 3, 8),
                                                  (1, 500, 2500, 150, 250)])
def test_dense_schur_product_equals_the_entry_lists(ba, model, cams, pts, lo, hi):
    """Tracks seen by a large share of the cameras: the point part of the Schur complement as a product of two dense
    matrices on the f64 matrix cores (ba_dense.hip: three tiles split over K; a 745-unknown system; a sparse problem
    forced through it; the 500-view job's last global adjustments -- 2495 unknowns, 210 tiles, no split) against the
    camera-pair entry lists: same LM trajectory, the final cost to 1e-12."""
    sc = synth.make_ba_scene(model, cams, pts, config_id=71, min_len=lo, max_len=hi)
    s1, fp1 = _solve_with_env(ba, sc, {"OSFM_BA_DENSE_SCHUR": "1"})
    s2, fp2 = _solve_with_env(ba, sc, {"OSFM_BA_DENSE_SCHUR": "0"})
    assert (s1.num_iterations, s1.num_successful_steps, s1.num_unsuccessful_steps, s1.termination) == \
        (s2.num_iterations, s2.num_successful_steps, s2.num_unsuccessful_steps, s2.termination)
    assert s1.num_iterations >= 3
    assert s1.num_pair_entries == s2.num_pair_entries
    assert abs(s1.final_cost - s2.final_cost) <= 1e-12 * s2.final_cost
    assert np.abs(fp1.cam_params - fp2.cam_params).max() <= 1e-10
    assert np.abs(fp1.points - fp2.points).max() <= 1e-9
    if cams == 40:
        # the product in front of the launch-per-column Cholesky (which consumes the system: cleared, product, added to)
        s4, fp4 = _solve_with_env(ba, sc, {"OSFM_BA_DENSE_SCHUR": "1", "OSFM_BA_CHOLESKY_STEPS": "1"})
        assert (s4.num_iterations, s4.termination) == (s1.num_iterations, s1.termination)
        assert abs(s4.final_cost - s1.final_cost) <= 1e-12 * s1.final_cost
    if cams == 150:
        # whichever of the two the cost model takes here: the oracle's solve
        s3, fp3 = _solve_with_env(ba, sc)
        assert np.array_equal(fp3.cam_params, fp1.cam_params) or np.array_equal(fp3.cam_params, fp2.cam_params)
        _compare_solve(ba, sc)


@pytest.mark.parametrize("model,cams,pts,lo,hi", [(0, 3, 2500, 3, 3), (1, 3, 400, 2, 3), (0, 8, 3000, 2, 8), (0, 6, 50, 1, 4)])
def test_pair_lists_of_a_handful_of_cameras_without_the_sort(ba, model, cams, pts, lo, hi):
    """Up to eight cameras: the Schur pair lists from one scan over per-pair flags (ba_pairs.hip:
    pair_lists_build_small) instead of the general build's sort -- the same lists, hence the same solve to the bit
    (3-camera local adjustments with every track in every camera; tracks of two views; eight cameras; tracks of one
    view, which have no off-diagonal entries)."""
    sc = synth.make_ba_scene(model, cams, pts, config_id=81, min_len=lo, max_len=hi)
    s1, fp1 = _solve_with_env(ba, sc)
    s2, fp2 = _solve_with_env(ba, sc, {"OSFM_BA_PAIR_LISTS_GENERAL": "1"})
    assert s1.num_pair_entries == s2.num_pair_entries and s1.num_pair_entries > 0
    assert (s1.num_iterations, s1.termination, s1.final_cost) == (s2.num_iterations, s2.termination, s2.final_cost)
    assert np.array_equal(fp1.cam_params, fp2.cam_params) and np.array_equal(fp1.points, fp2.points)


@pytest.mark.parametrize("cams,pts", [(12, 1500), (61, 6000)])
def test_a_cholesky_launch_given_up_is_repeated_launch_by_launch(ba, cams, pts):
    """A wait of the one-launch Cholesky that outlasts its spin limit (its workgroups were not all resident:
    another process or a long foreign kernel on the device) gives the launch up.  That is a scheduling
    condition, not a matrix that is not positive definite: the solve repeats the factorisation in the
    launch-per-column form, keeps that form, counts it -- and lands on the bits of a launch-per-column solve.
    The test hook makes every wait that is not satisfied at once give up (12 cameras: the host runs an
    iteration ahead of the device's decisions; 61: it reads every decision)."""
    from orthosfm_amd import capi
    sc = synth.make_ba_scene(0, cams, pts, config_id=42)
    s_ref, fp_ref = _solve_with_env(ba, sc, {"OSFM_BA_CHOLESKY_STEPS": "1"})
    capi.check(capi.lib.osfm_ba_debug_flow_spin_limit(1))
    try:
        # (the cameras' own order, as the launch-per-column reference has it: bits are compared; the ordered layout
        #  through a given-up launch is test_ordered_elimination_survives_a_given_up_launch)
        s, fp = _solve_with_env(ba, sc, {"OSFM_BA_ORDER": "0"})
    finally:
        capi.check(capi.lib.osfm_ba_debug_flow_spin_limit(0))
    assert s.flow_fallbacks == 1
    assert (s.num_iterations, s.num_successful_steps, s.num_unsuccessful_steps, s.termination) == \
        (s_ref.num_iterations, s_ref.num_successful_steps, s_ref.num_unsuccessful_steps, s_ref.termination)
    assert s.final_cost == s_ref.final_cost
    assert np.array_equal(fp.cam_params, fp_ref.cam_params) and np.array_equal(fp.points, fp_ref.points)
    # and the next solve is back on the one-launch form
    s2, _ = _solve_with_env(ba, sc)
    assert s2.flow_fallbacks == 0 and s2.num_iterations == s_ref.num_iterations


@pytest.mark.parametrize("model,cams,pts,maxlen,arcs", [(0, 200, 20000, 12, True), (1, 120, 8000, 12, True), (0, 500, 20000, 12, True),
                                                        (0, 64, 4000, 30, False), (0, 60, 3000, 6, True)])
def test_ordered_elimination_of_a_ring_equals_the_natural_order(ba, model, cams, pts, maxlen, arcs):
    """The reference factors the reduced camera system behind a fill-reducing ordering (SPARSE_SCHUR + SUITE_SPARSE,
    bundle_adjustment.cpp:126-133).  Where every track spans a short run of neighbouring views the cameras are laid
    out as K arcs + K separators (ba_order.hip) and the arcs are factored side by side: same LM trajectory as in the
    cameras' own order, costs and parameters to rounding, a shorter chain of dependent blocks; a set whose tracks
    span half the ring keeps its order; both forms repeat bit for bit."""
    sc = synth.make_ba_scene(model, cams, pts, config_id=81, max_len=maxlen)
    s1, fp1 = _solve_with_env(ba, sc, {"OSFM_BA_ORDER": "1"})
    s0, fp0 = _solve_with_env(ba, sc, {"OSFM_BA_ORDER": "0"})
    s2, fp2 = _solve_with_env(ba, sc, {"OSFM_BA_ORDER": "1"})
    assert (s1.order_arcs > 0) == arcs and s0.order_arcs == 0
    if arcs:
        assert s1.chain_blocks * 4 <= s1.chain_blocks_natural * 3 and s1.chain_blocks_natural >= 8
    assert (s1.num_iterations, s1.num_successful_steps, s1.num_unsuccessful_steps, s1.termination) == \
        (s0.num_iterations, s0.num_successful_steps, s0.num_unsuccessful_steps, s0.termination)
    assert s1.num_iterations >= 3 and s1.flow_fallbacks == 0
    assert abs(s1.final_cost - s0.final_cost) <= 1e-11 * s0.final_cost
    assert np.abs(fp1.cam_params - fp0.cam_params).max() <= 1e-9
    assert np.abs(fp1.points - fp0.points).max() <= 1e-8
    assert s1.final_cost == s2.final_cost and np.array_equal(fp1.cam_params, fp2.cam_params) and np.array_equal(fp1.points, fp2.points)


def test_ordered_elimination_survives_a_given_up_launch(ba):
    """The launch-per-column fallback works on the same laid-out system (interior padding rows are identity rows that
    it clears and sets like the tail's): a Cholesky launch given up in the ordered form is repeated there."""
    from orthosfm_amd import capi
    sc = synth.make_ba_scene(0, 200, 20000, config_id=82)
    s0, fp0 = _solve_with_env(ba, sc, {})
    assert s0.order_arcs > 0
    capi.check(capi.lib.osfm_ba_debug_flow_spin_limit(1))
    try:
        s1, fp1 = _solve_with_env(ba, sc, {})
    finally:
        capi.check(capi.lib.osfm_ba_debug_flow_spin_limit(0))
    assert s1.flow_fallbacks == 1 and s1.order_arcs > 0
    assert (s1.num_iterations, s1.termination) == (s0.num_iterations, s0.termination)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost
