"""Host-side mirror of the reference's bundle-adjustment entry point on top of
the C ABI: orthosfm::runBundleAdjustment(cameras, tracks, algorithm,
optimizePoints, retriangulatePoints) (src/bundle_adjustment/bundle_adjustment.h:
18-20, .cpp:49-161) plus the two calls the pipeline makes around it,
ReconstructionAlgorithm::evaluateReprojectionError and ::triangulateTracks.

The classes below carry exactly the fields of the reference types that cross
the boundary (Feature, Track: src/data_structures/track.h:21-107; camera
parameter blocks: OrthoQuaternionCamera.h:83-91, OrthographicCamera.h:122-134).
All arithmetic happens in libosfm_hip.so; this module only flattens AoS to the
SoA arrays of osfm_ba_problem and scatters the results back.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi

MODEL_QUATERNION = 0
MODEL_EULER = 1
TERMINATION = {1: "CONVERGENCE (function tolerance)", 2: "CONVERGENCE (gradient tolerance)",
               3: "CONVERGENCE (parameter tolerance)", 4: "CONVERGENCE (trust region)",
               5: "NO_CONVERGENCE (max iterations)", 6: "FAILURE"}


@dataclass
class Feature:
    """orthosfm::Feature (track.h:21-31): x/y are float32 pixel coordinates."""
    viewID: int
    localFeatureID: int
    x: float
    y: float


@dataclass
class Track:
    """orthosfm::Track (track.h:69-107)."""
    features: list
    point: np.ndarray = field(default_factory=lambda: np.zeros(4))
    has_point: bool = False


@dataclass
class QuatCamera:
    """OrthoQuaternionCamera: rotation (x, y, z, w), offsets, scale; per-block
    fixed flags default to the reference's (rotation/offset free, scale fixed)."""
    view_id: int
    width: int
    height: int
    rotation: np.ndarray
    offset_x: float = 0.0
    offset_y: float = 0.0
    scale: float = 1.0
    fixed: bool = False
    fix_rotation: bool = False
    fix_offset: bool = False
    fix_scale: bool = True
    model = MODEL_QUATERNION

    def params(self):
        return np.array([*self.rotation, self.offset_x, self.offset_y, self.scale])

    def const_mask(self):
        f = self.fixed
        return np.array([f or self.fix_rotation] * 4 + [f or self.fix_offset] * 2 + [f or self.fix_scale],
                        dtype=np.uint8)

    def set_params(self, v):
        self.rotation = np.array(v[:4])
        self.offset_x, self.offset_y, self.scale = float(v[4]), float(v[5]), float(v[6])


@dataclass
class EulerCamera:
    """OrthographicCamera: phi, theta, roll, offsets, scale with the DoF mask of
    setDegreesOfFreedom (OrthographicCamera.cpp:195-207)."""
    view_id: int
    width: int
    height: int
    phi: float = 0.0
    theta: float = 0.0
    roll: float = 0.0
    offset_x: float = 0.0
    offset_y: float = 0.0
    scale: float = 1.0
    fixed: bool = False
    dof: int = 4            # solver 3 (ORTHO_EULER_ALL_DOF) -> 4
    model = MODEL_EULER

    def params(self):
        return np.array([self.phi, self.theta, self.roll, self.offset_x, self.offset_y, self.scale, 0.0])

    def const_mask(self):
        d = self.dof
        flags = [d < 1, d < 2, d < 3, d < 4, d < 4, d < 5, True]
        return np.array([self.fixed or f for f in flags], dtype=np.uint8)

    def set_params(self, v):
        self.phi, self.theta, self.roll = float(v[0]), float(v[1]), float(v[2])
        self.offset_x, self.offset_y, self.scale = float(v[3]), float(v[4]), float(v[5])


class FlatProblem:
    """The arrays of osfm_ba_problem (kept alive while the struct is in use)."""

    def __init__(self, model, cam_params, cam_const, img_w, img_h, points, obs_xy, obs_camera, obs_point):
        self.model = int(model)
        # always private copies: the solver updates cam_params / points in place
        self.cam_params = np.array(cam_params, dtype=np.float64, order="C").reshape(-1, 7)
        self.cam_const = np.array(cam_const, dtype=np.uint8, order="C").reshape(-1, 7)
        self.img_w = np.array(img_w, dtype=np.int32, order="C")
        self.img_h = np.array(img_h, dtype=np.int32, order="C")
        self.points = np.array(points, dtype=np.float64, order="C").reshape(-1, 4)
        self.obs_xy = np.array(obs_xy, dtype=np.float64, order="C").reshape(-1, 2)
        self.obs_camera = np.array(obs_camera, dtype=np.int32, order="C")
        self.obs_point = np.array(obs_point, dtype=np.int32, order="C")

    @classmethod
    def from_scene(cls, sc):
        return cls(sc.model, sc.cam_params, sc.cam_const, sc.img_w, sc.img_h, sc.points, sc.obs_xy,
                   sc.obs_camera, sc.obs_point)

    def struct(self):
        p = capi.BaProblem()
        p.model = self.model
        p.num_cameras = self.cam_params.shape[0]
        p.num_points = self.points.shape[0]
        p.num_observations = self.obs_camera.shape[0]
        p.cam_params = self.cam_params.ctypes.data
        p.cam_const = self.cam_const.ctypes.data
        p.img_width = self.img_w.ctypes.data
        p.img_height = self.img_h.ctypes.data
        p.points = self.points.ctypes.data
        p.obs_xy = self.obs_xy.ctypes.data
        p.obs_camera = self.obs_camera.ctypes.data
        p.obs_point = self.obs_point.ctypes.data
        return p


def default_options(**kw) -> capi.BaOptions:
    o = capi.BaOptions()
    capi.check(capi.lib.osfm_ba_options_default(C.byref(o)))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def solve(problem: FlatProblem, options: capi.BaOptions | None = None, **kw) -> capi.BaSummary:
    """osfm_ba_solve: in-place update of problem.cam_params / problem.points."""
    o = options if options is not None else default_options(**kw)
    s = capi.BaSummary()
    st = problem.struct()
    capi.check(capi.lib.osfm_ba_solve(C.byref(st), C.byref(o), C.byref(s)))
    return s


def reprojection_errors(problem: FlatProblem, device: int = 0):
    """Batched ReconstructionAlgorithm::evaluateReprojectionError."""
    O = problem.obs_camera.shape[0]
    err = np.zeros(O)
    res = np.zeros((O, 2))
    st = problem.struct()
    capi.check(capi.lib.osfm_ba_reprojection_errors(C.byref(st), device, capi._ptr(err, C.c_double),
                                                    capi._ptr(res, C.c_double)))
    return err, res


def triangulate(problem: FlatProblem, device: int = 0) -> np.ndarray:
    """triangulateOrthographicTracks on the flattened tracks (in place);
    returns the per-point validity mask."""
    valid = np.zeros(max(problem.points.shape[0], 1), dtype=np.uint8)
    st = problem.struct()
    capi.check(capi.lib.osfm_ba_triangulate(C.byref(st), device, capi._ptr(valid, C.c_uint8)))
    return valid[:problem.points.shape[0]]


def _flatten(cameras, tracks):
    """bundle_adjustment.cpp:54-57,103-123: residual blocks for every feature
    of every track with a point whose view has a camera."""
    cam_index = {}
    for i, cam in enumerate(cameras):
        cam_index.setdefault(cam.view_id, i)          # std::map::insert keeps the first
    obs_xy, obs_cam, obs_pt, used, pts = [], [], [], [], []
    for ti, t in enumerate(tracks):
        if not t.has_point:
            continue
        j = len(used)
        n_before = len(obs_cam)
        for f in t.features:
            ci = cam_index.get(f.viewID)
            if ci is None:
                continue
            obs_xy.append((float(np.float32(f.x)), float(np.float32(f.y))))
            obs_cam.append(ci)
            obs_pt.append(j)
        # the parameter block exists even without residuals (AddParameterBlock, :88)
        used.append(ti)
        pts.append(np.asarray(t.point, dtype=np.float64))
        if len(obs_cam) == n_before:
            pass
    model = cameras[0].model if cameras else MODEL_QUATERNION
    fp = FlatProblem(model,
                     np.array([c.params() for c in cameras]).reshape(-1, 7),
                     np.array([c.const_mask() for c in cameras], dtype=np.uint8).reshape(-1, 7),
                     [c.width for c in cameras], [c.height for c in cameras],
                     np.array(pts).reshape(-1, 4), np.array(obs_xy).reshape(-1, 2), obs_cam, obs_pt)
    return fp, used


def filter_tracks_to_available_cameras(cameras, tracks):
    """filterTracksToAvailableCameras(cameras, tracks, false, false)
    (src/util/common.cpp:85-139): keeps tracks with >= 2 features in the given
    cameras, restricted to those features; the copies carry NO point."""
    ids = {c.view_id for c in cameras}
    out = []
    for t in tracks:
        fs = [f for f in t.features if f.viewID in ids]
        if len(fs) > 1:
            out.append(Track(fs))
    return out


def triangulate_tracks(cameras, tracks, reset_existing=True, device=0):
    """ReconstructionAlgorithm::triangulateTracks ->
    triangulateOrthographicTracks (triangulation.cpp:44-93)."""
    tmp = [Track(t.features, np.array([0.0, 0, 0, 1]), True) for t in tracks]
    fp, used = _flatten(cameras, tmp)
    valid = triangulate(fp, device) if fp.obs_camera.size else np.zeros(len(used), np.uint8)
    for j, ti in enumerate(used):
        t = tracks[ti]
        if valid[j]:
            if (not t.has_point) or reset_existing:
                t.point = fp.points[j].copy()
                t.has_point = True
        elif reset_existing:
            t.has_point = False


def run_bundle_adjustment(cameras, tracks, algorithm=None, optimize_points=True,
                          retriangulate_points=False, device=0, verbose=True):
    """orthosfm::runBundleAdjustment.  Cameras are updated in place; track points
    are updated in place only when retriangulate_points is False -- with True
    the reference optimises a filtered, re-triangulated COPY that it discards
    (bundle_adjustment.cpp:71-83,161), and so does this function."""
    work = tracks
    if retriangulate_points:
        work = filter_tracks_to_available_cameras(cameras, tracks)
        triangulate_tracks(cameras, work, True, device)
    fp, used = _flatten(cameras, work)
    s = solve(fp, optimize_points=1 if optimize_points else 0, device=device)
    for cam, v in zip(cameras, fp.cam_params):
        cam.set_params(v)
    for j, ti in enumerate(used):
        work[ti].point = fp.points[j].copy()
    if verbose:
        # summary.BriefReport() + the point-motion line (bundle_adjustment.cpp:148-160)
        print(f"Ceres-style Solver Report: Iterations: {s.num_iterations + 1}, Initial cost: {s.initial_cost:e}, "
              f"Final cost: {s.final_cost:e}, Termination: {TERMINATION.get(s.termination, '?')}")
        n = max(len(work), 1)
        print(f"Average point change: {s.mean_point_change * len(used) / n} (maximum change: {s.max_point_change})")
    return s


# ---------------------------------------------------------------------------
# bench helper
# ---------------------------------------------------------------------------

def bench_global_ba(num_cameras=200, num_points=100000, max_iterations=25, device=0):
    """BASELINE configs[3]: global BA, quaternion solver, 200 cameras, ~100k
    tracks.  Returns LM iterations/s of one solve from a perturbed start."""
    from . import synth
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, num_cameras, num_points, config_id=4)
    fp = FlatProblem.from_scene(sc)
    warm = FlatProblem.from_scene(sc)
    solve(warm, max_num_iterations=2, device=device)          # warm-up (allocator, code objects)
    # the call in its steady state: five identical calls, the MEDIAN one reported (by call time; the extremes beside it --
    # the CPU side of the comparison is a single measurement, so neither is the best of several)
    runs = [solve(FlatProblem.from_scene(sc), max_num_iterations=max_iterations, device=device) for _ in range(5)]
    runs.sort(key=lambda r: r.solve_ms)
    s = runs[len(runs) // 2]
    its = s.num_iterations
    # the per-family device times come from a second, instrumented solve (verbose = 1 records
    # an event pair around every kernel family of every iteration)
    prof = FlatProblem.from_scene(sc)
    sp = solve(prof, max_num_iterations=max_iterations, device=device, verbose=1)
    assert sp.num_iterations == its and sp.final_cost == s.final_cost
    s.point_pass_ms, s.pair_pass_ms, s.cholesky_ms, s.back_pass_ms = (sp.point_pass_ms, sp.pair_pass_ms,
                                                                      sp.cholesky_ms, sp.back_pass_ms)
    lm_ms = s.point_pass_ms + s.pair_pass_ms + s.cholesky_ms + s.back_pass_ms
    return {"workload": f"{num_cameras} quaternion cameras, {num_points} tracks, "
                        f"{fp.obs_camera.size} observations, Schur + dense Cholesky",
            "iterations": int(its), "iterations_per_s": its / (s.solve_ms * 1e-3), "calls_timed": len(runs), "reported": "median call",
            "iterations_per_s_min_max": [its / (runs[-1].solve_ms * 1e-3), its / (runs[0].solve_ms * 1e-3)],
            "lm_loop_iterations_per_s": its / max(s.lm_loop_ms * 1e-3, 1e-9),
            "lm_loop_iterations_per_s_min_max": [its / max(max(r.lm_loop_ms for r in runs) * 1e-3, 1e-9),
                                                 its / max(min(r.lm_loop_ms for r in runs) * 1e-3, 1e-9)],
            # elimination order of the reduced camera system (ba_order.hip): arcs factored side by side, the
            # chain of dependent diagonal blocks in the cameras' order and in the order used
            "order_arcs": int(s.order_arcs), "chain_blocks": [int(s.chain_blocks_natural), int(s.chain_blocks)],
            "solve_ms": s.solve_ms, "lm_loop_ms": s.lm_loop_ms, "initial_cost": s.initial_cost, "final_cost": s.final_cost,
            "termination": TERMINATION.get(s.termination, "?"),
            "kernel_ms": {"point_pass": s.point_pass_ms, "pair_pass": s.pair_pass_ms,
                          "cholesky": s.cholesky_ms, "back_pass": s.back_pass_ms,
                          "sum": lm_ms, "linearizations": int(s.linearizations)},
            "pair_entries": int(s.num_pair_entries),
            "observations": int(fp.obs_camera.size), "points": int(num_points),
            # tangent size of the free camera blocks: 3 per free rotation, 1 per free scalar
            "camera_unknowns": int(sum((0 if c[0] else 3) + int((c[4:] == 0).sum()) for c in fp.cam_const))}


def bench_local_ba(num_points=3000, num_cameras=3, device=0, repeats=5):
    """BASELINE configs[0] / the per-group call of the incremental reconstruction
    (reconstruct.cpp:219): a 3-camera quaternion BA.  Latency of one call and of one LM
    iteration inside it (the median of `repeats` calls)."""
    import time
    from . import synth
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, num_cameras, num_points, config_id=1)
    solve(FlatProblem.from_scene(sc), max_num_iterations=2, device=device)
    calls = []
    for _ in range(repeats):
        fp = FlatProblem.from_scene(sc)
        t0 = time.perf_counter()
        s = solve(fp, max_num_iterations=50, device=device)
        calls.append(((time.perf_counter() - t0) * 1e3, s))
    calls.sort(key=lambda c: c[0])
    call_ms, s = calls[len(calls) // 2]              # the median call (its extremes: call_ms_min_max)
    return {"workload": f"{num_cameras} quaternion cameras, {num_points} tracks, {fp.obs_camera.size} observations "
                        "(the local adjustment of one camera group)",
            "iterations": int(s.num_iterations), "call_ms": call_ms, "call_ms_min_max": [calls[0][0], calls[-1][0]], "reported": "median call",
            "lm_loop_ms": s.lm_loop_ms,
            "us_per_iteration": 1e3 * s.lm_loop_ms / max(s.num_iterations, 1),
            "iterations_per_s": s.num_iterations / max(s.lm_loop_ms * 1e-3, 1e-9),
            "final_cost": s.final_cost, "termination": TERMINATION.get(s.termination, "?")}
