"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when it was
built, to the reference's own matcher (oracle/_ref/libref_match.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by orthosfm_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_match.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_s16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _load(path):
    if not os.path.exists(path):
        if path == ORACLE_SO:
            build_oracle()
        else:
            return None
    return C.CDLL(path)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = _load(ORACLE_SO)
    return _oracle


def ref():
    """The reference matcher, or None when oracle/_ref was not built."""
    global _ref
    if _ref is None:
        _ref = _load(REF_SO)
    return _ref


def have_ref():
    return ref() is not None


def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def _s16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


FLT_MAX = float(np.finfo(np.float32).max)


class _Matcher:
    """Same call surface for the oracle restatement and the reference shim."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.p = prefix

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    def nn_find(self, q, el):
        out = np.zeros(4, dtype=np.int32)
        if el.dtype == np.uint16:
            f = self._f("nn_find_u16")
            f.argtypes = [_u16p, _u16p, C.c_int, C.c_int, _i32p]
        else:
            f = self._f("nn_find_s16")
            f.argtypes = [_s16p, _s16p, C.c_int, C.c_int, _i32p]
        f.restype = None
        el = np.ascontiguousarray(el)
        q = np.ascontiguousarray(q)
        f(q, el if el.size else np.zeros(1, el.dtype), el.shape[0], q.shape[0], out)
        return out

    def twoway(self, s1, s2, lowe, dist=FLT_MAX):
        n1, n2 = s1.shape[0], s2.shape[0]
        dim = s1.shape[1] if s1.ndim == 2 else s2.shape[1]
        m12 = np.full(max(n1, 1), -7, dtype=np.int32)
        m21 = np.full(max(n2, 1), -7, dtype=np.int32)
        if s1.dtype == np.uint16:
            f = self._f("twoway_match_u16")
            f.argtypes = [_u16p, C.c_int, _u16p, C.c_int, C.c_int, C.c_float, C.c_float, _i32p, _i32p]
        else:
            f = self._f("twoway_match_s16")
            f.argtypes = [_s16p, C.c_int, _s16p, C.c_int, C.c_int, C.c_float, C.c_float, _i32p, _i32p]
        f.restype = None
        a = np.ascontiguousarray(s1).reshape(-1)
        b = np.ascontiguousarray(s2).reshape(-1)
        if a.size == 0:
            a = np.zeros(1, s1.dtype)
        if b.size == 0:
            b = np.zeros(1, s2.dtype)
        f(a, n1, b, n2, dim, lowe, dist, m12, m21)
        return m12[:n1].copy(), m21[:n2].copy()

    def remove_inconsistent(self, m12, m21):
        a = np.ascontiguousarray(m12, dtype=np.int32).copy()
        b = np.ascontiguousarray(m21, dtype=np.int32).copy()
        f = self._f("remove_inconsistent")
        f.argtypes = [_i32p, C.c_int, _i32p, C.c_int]
        f.restype = None
        aa = a if a.size else np.zeros(1, np.int32)
        bb = b if b.size else np.zeros(1, np.int32)
        f(aa, a.size, bb, b.size)
        return aa[:a.size], bb[:b.size]

    def count_consistent(self, m12, m21):
        a = np.ascontiguousarray(m12, dtype=np.int32)
        b = np.ascontiguousarray(m21, dtype=np.int32)
        f = self._f("count_consistent")
        f.argtypes = [_i32p, C.c_int, _i32p, C.c_int]
        f.restype = C.c_int
        aa = a if a.size else np.zeros(1, np.int32)
        bb = b if b.size else np.zeros(1, np.int32)
        return f(aa, a.size, bb, b.size)

    def combine(self, s12, s21, u12, u21):
        arrs = [np.ascontiguousarray(x, dtype=np.int32) for x in (s12, s21, u12, u21)]
        n = [x.size for x in arrs]
        arrs = [x if x.size else np.zeros(1, np.int32) for x in arrs]
        o12 = np.zeros(max(n[0] + n[2], 1), dtype=np.int32)
        o21 = np.zeros(max(n[1] + n[3], 1), dtype=np.int32)
        f = self._f("combine_results")
        f.argtypes = [_i32p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, _i32p, _i32p]
        f.restype = None
        f(arrs[0], n[0], arrs[1], n[1], arrs[2], n[2], arrs[3], n[3], o12, o21)
        return o12[:n[0] + n[2]].copy(), o21[:n[1] + n[3]].copy()


def oracle_matcher():
    return _Matcher(oracle(), "oracle_")


def ref_matcher():
    r = ref()
    return None if r is None else _Matcher(r, "ref_")


# --- whole-view functions (A1, A7) -----------------------------------------

def oracle_convert_sift(f):
    f = np.ascontiguousarray(f, dtype=np.float32)
    out = np.zeros(f.shape, dtype=np.uint16)
    fn = oracle().oracle_convert_sift
    fn.argtypes = [_f32p, C.c_int, _u16p]
    fn.restype = None
    if f.size:
        fn(f.reshape(-1), f.shape[0], out.reshape(-1))
    return out


def oracle_convert_surf(f):
    f = np.ascontiguousarray(f, dtype=np.float32)
    out = np.zeros(f.shape, dtype=np.int16)
    fn = oracle().oracle_convert_surf
    fn.argtypes = [_f32p, C.c_int, _s16p]
    fn.restype = None
    if f.size:
        fn(f.reshape(-1), f.shape[0], out.reshape(-1))
    return out


def _nz(a, dt):
    a = np.ascontiguousarray(a, dtype=dt).reshape(-1)
    return a if a.size else np.zeros(1, dt)


def oracle_pairwise_match(sift1, surf1, sift2, surf2, sift_lowe=0.8, surf_lowe=0.7,
                          sift_dist=FLT_MAX, surf_dist=FLT_MAX):
    ns1, nu1, ns2, nu2 = sift1.shape[0], surf1.shape[0], sift2.shape[0], surf2.shape[0]
    o12 = np.zeros(max(ns1 + nu1, 1), dtype=np.int32)
    o21 = np.zeros(max(ns2 + nu2, 1), dtype=np.int32)
    l12 = C.c_int(0)
    l21 = C.c_int(0)
    fn = oracle().oracle_pairwise_match
    fn.argtypes = [_u16p, C.c_int, _s16p, C.c_int, _u16p, C.c_int, _s16p, C.c_int,
                   C.c_float, C.c_float, C.c_float, C.c_float,
                   _i32p, C.POINTER(C.c_int), _i32p, C.POINTER(C.c_int)]
    fn.restype = None
    fn(_nz(sift1, np.uint16), ns1, _nz(surf1, np.int16), nu1,
       _nz(sift2, np.uint16), ns2, _nz(surf2, np.int16), nu2,
       sift_lowe, sift_dist, surf_lowe, surf_dist, o12, C.byref(l12), o21, C.byref(l21))
    return o12[:l12.value].copy(), o21[:l21.value].copy()


def oracle_pairwise_match_lowres(sift1, surf1, sift2, surf2, num_features=500,
                                 sift_lowe=0.8, surf_lowe=0.7,
                                 sift_dist=FLT_MAX, surf_dist=FLT_MAX):
    fn = oracle().oracle_pairwise_match_lowres
    fn.argtypes = [_u16p, C.c_int, _s16p, C.c_int, _u16p, C.c_int, _s16p, C.c_int,
                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
    fn.restype = C.c_int
    return fn(_nz(sift1, np.uint16), sift1.shape[0], _nz(surf1, np.int16), surf1.shape[0],
              _nz(sift2, np.uint16), sift2.shape[0], _nz(surf2, np.int16), surf2.shape[0],
              sift_lowe, sift_dist, surf_lowe, surf_dist, num_features)


class RefExhaustive:
    """The reference's ExhaustiveMatching fed with FLOAT descriptors
    (init() quantises them itself)."""

    def __init__(self, views):
        r = ref()
        self.r = r
        r.ref_matcher_create.restype = C.c_void_p
        r.ref_matcher_create.argtypes = [C.c_int]
        self.h = C.c_void_p(r.ref_matcher_create(len(views)))
        r.ref_matcher_set_view.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_int, _f32p, C.c_int]
        r.ref_matcher_set_view.restype = None
        self.sizes = []
        for v, (sift, surf) in enumerate(views):
            sift = np.ascontiguousarray(sift, dtype=np.float32)
            surf = np.ascontiguousarray(surf, dtype=np.float32)
            self.sizes.append((sift.shape[0], surf.shape[0]))
            r.ref_matcher_set_view(self.h, v, _nz(sift, np.float32), sift.shape[0],
                                   _nz(surf, np.float32), surf.shape[0])
        r.ref_matcher_init.argtypes = [C.c_void_p]
        r.ref_matcher_init.restype = None
        r.ref_matcher_init(self.h)

    def pairwise_match(self, v1, v2):
        n1 = sum(self.sizes[v1])
        n2 = sum(self.sizes[v2])
        o12 = np.zeros(max(n1, 1), dtype=np.int32)
        o21 = np.zeros(max(n2, 1), dtype=np.int32)
        l12, l21 = C.c_int(0), C.c_int(0)
        f = self.r.ref_matcher_pairwise_match
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.POINTER(C.c_int), _i32p, C.POINTER(C.c_int)]
        f.restype = None
        f(self.h, v1, v2, o12, C.byref(l12), o21, C.byref(l21))
        return o12[:l12.value].copy(), o21[:l21.value].copy()

    def pairwise_match_lowres(self, v1, v2, n):
        f = self.r.ref_matcher_pairwise_match_lowres
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        f.restype = C.c_int
        return f(self.h, v1, v2, n)

    def __del__(self):
        try:
            self.r.ref_matcher_destroy.argtypes = [C.c_void_p]
            self.r.ref_matcher_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------------------
# bundle adjustment oracle (oracle/ba_oracle.c)
# ---------------------------------------------------------------------------

class OBaProblem(C.Structure):
    _fields_ = [("model", C.c_int32), ("num_cameras", C.c_int32), ("num_points", C.c_int32),
                ("num_observations", C.c_int32),
                ("cam_params", C.c_void_p), ("cam_const", C.c_void_p),
                ("img_width", C.c_void_p), ("img_height", C.c_void_p),
                ("points", C.c_void_p), ("obs_xy", C.c_void_p),
                ("obs_camera", C.c_void_p), ("obs_point", C.c_void_p)]


class OBaOptions(C.Structure):
    _fields_ = [("huber_delta", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("max_num_iterations", C.c_int32), ("optimize_points", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
                ("device", C.c_int32), ("verbose", C.c_int32)]


class OBaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("termination", C.c_int32),
                ("mean_point_change", C.c_double), ("max_point_change", C.c_double),
                ("solve_ms", C.c_double), ("point_pass_ms", C.c_double),
                ("pair_pass_ms", C.c_double), ("cholesky_ms", C.c_double),
                ("back_pass_ms", C.c_double),
                ("linearizations", C.c_int32), ("num_pair_entries", C.c_int32)]


def ba_default_options(**kw):
    o = OBaOptions(1.0, 1e-6, 1e-10, 1e-10, 100, 1, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 1, 5, 0, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def ba_problem_struct(scene, cls=OBaProblem):
    """Struct over the scene's arrays (which must stay alive and contiguous)."""
    for name, dt in (("cam_params", np.float64), ("points", np.float64), ("obs_xy", np.float64)):
        a = getattr(scene, name)
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
    for name, dt in (("img_w", np.int32), ("img_h", np.int32), ("obs_camera", np.int32),
                     ("obs_point", np.int32)):
        a = getattr(scene, name)
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"], name
    assert scene.cam_const.dtype == np.uint8 and scene.cam_const.flags["C_CONTIGUOUS"]
    p = cls()
    p.model = scene.model
    p.num_cameras = scene.cam_params.shape[0]
    p.num_points = scene.points.shape[0]
    p.num_observations = scene.obs_camera.shape[0]
    p.cam_params = scene.cam_params.ctypes.data
    p.cam_const = scene.cam_const.ctypes.data
    p.img_width = scene.img_w.ctypes.data
    p.img_height = scene.img_h.ctypes.data
    p.points = scene.points.ctypes.data
    p.obs_xy = scene.obs_xy.ctypes.data
    p.obs_camera = scene.obs_camera.ctypes.data
    p.obs_point = scene.obs_point.ctypes.data
    return p


def oracle_ba_solve(scene, **opt_kw):
    """Runs the oracle LM in place on scene.cam_params / scene.points."""
    lib = oracle()
    p = ba_problem_struct(scene)
    o = ba_default_options(**opt_kw)
    s = OBaSummary()
    lib.oracle_ba_solve.argtypes = [C.POINTER(OBaProblem), C.POINTER(OBaOptions), C.POINTER(OBaSummary)]
    lib.oracle_ba_solve.restype = C.c_int
    rc = lib.oracle_ba_solve(C.byref(p), C.byref(o), C.byref(s))
    assert rc == 0
    return s


def oracle_ba_residuals(scene):
    lib = oracle()
    p = ba_problem_struct(scene)
    O = scene.obs_camera.shape[0]
    res = np.zeros((O, 2))
    err = np.zeros(O)
    lib.oracle_ba_residuals.argtypes = [C.POINTER(OBaProblem), _f64p, _f64p]
    lib.oracle_ba_residuals.restype = None
    lib.oracle_ba_residuals(C.byref(p), res.reshape(-1), err)
    return res, err


def oracle_ba_jacobian(scene, k):
    lib = oracle()
    p = ba_problem_struct(scene)
    r = np.zeros(2)
    jc = np.zeros((2, 7))
    jp = np.zeros((2, 4))
    lib.oracle_ba_jacobian.argtypes = [C.POINTER(OBaProblem), C.c_int, _f64p, _f64p, _f64p]
    lib.oracle_ba_jacobian.restype = None
    lib.oracle_ba_jacobian(C.byref(p), k, r, jc.reshape(-1), jp.reshape(-1))
    return r, jc, jp


def oracle_ba_triangulate(scene):
    lib = oracle()
    p = ba_problem_struct(scene)
    valid = np.zeros(scene.points.shape[0], dtype=np.uint8)
    lib.oracle_ba_triangulate.argtypes = [C.POINTER(OBaProblem), np.ctypeslib.ndpointer(np.uint8)]
    lib.oracle_ba_triangulate.restype = C.c_int
    lib.oracle_ba_triangulate(C.byref(p), valid)
    return valid


def ba_cost(scene, huber=1.0):
    """1/2 sum rho(|r|^2) with Huber(delta) per 2-D block, from oracle residuals."""
    res, _ = oracle_ba_residuals(scene)
    s = (res ** 2).sum(1)
    b = huber * huber
    rho = np.where(s > b, 2.0 * huber * np.sqrt(s) - b, s)
    return 0.5 * rho.sum()


# ---------------------------------------------------------------------------
# geometric verification oracle (oracle/ransac_oracle.c) and reference shim
# ---------------------------------------------------------------------------
REF_RANSAC_SO = os.path.join(ORACLE_DIR, "_ref", "libref_ransac.so")
_ref_ransac = None


def ref_ransac():
    global _ref_ransac
    if _ref_ransac is None and os.path.exists(REF_RANSAC_SO):
        _ref_ransac = C.CDLL(REF_RANSAC_SO)
    return _ref_ransac


def oracle_sampson(F, p1, p2):
    f = oracle().oracle_sampson_distance
    f.argtypes = [_f64p, _f64p, _f64p]
    f.restype = C.c_double
    return f(np.ascontiguousarray(F, np.float64).reshape(-1), np.ascontiguousarray(p1, np.float64),
             np.ascontiguousarray(p2, np.float64))


def oracle_fundamental_8_point(p1, p2):
    f = oracle().oracle_fundamental_8_point
    f.argtypes = [_f64p, _f64p, _f64p]
    f.restype = C.c_int
    F = np.zeros(9)
    ok = f(np.ascontiguousarray(p1, np.float64).reshape(-1), np.ascontiguousarray(p2, np.float64).reshape(-1), F)
    return ok, F.reshape(3, 3)


def oracle_ransac(pos1, pos2, corr, max_iterations=1000, threshold=0.0015, seed=0, pair_id=0):
    f = oracle().oracle_ransac_fundamental
    f.argtypes = [_f32p, _f32p, _i32p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64, _i32p, _f64p]
    f.restype = C.c_int
    corr = np.ascontiguousarray(corr, np.int32).reshape(-1, 2)
    inl = np.zeros(max(corr.shape[0], 1), np.int32)
    F = np.zeros(9)
    n = f(np.ascontiguousarray(pos1, np.float32).reshape(-1), np.ascontiguousarray(pos2, np.float32).reshape(-1),
          corr.reshape(-1) if corr.size else np.zeros(2, np.int32), corr.shape[0], max_iterations, threshold,
          seed, pair_id, inl, F)
    return n, inl[:max(n, 0)].copy(), F.reshape(3, 3)


# ---------------------------------------------------------------------------
# outlier filter oracle (oracle/filter_oracle.c)
# ---------------------------------------------------------------------------
def oracle_nn_distances(points):
    lib = oracle()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
    nn = np.zeros(max(pts.shape[0], 1))
    lib.oracle_nn_distances.argtypes = [_f64p, C.c_int, _f64p]
    lib.oracle_nn_distances.restype = None
    lib.oracle_nn_distances(pts.reshape(-1), pts.shape[0], nn)
    return nn[:pts.shape[0]]


def oracle_filter_outlier_tracks(points, has_point):
    lib = oracle()
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
    hp = np.ascontiguousarray(has_point, dtype=np.uint8).reshape(-1)
    keep = np.zeros(max(hp.shape[0], 1), dtype=np.uint8)
    stats = np.zeros(2)
    lib.oracle_filter_outlier_tracks.argtypes = [_f64p, np.ctypeslib.ndpointer(np.uint8), C.c_int,
                                                 np.ctypeslib.ndpointer(np.uint8), _f64p]
    lib.oracle_filter_outlier_tracks.restype = None
    lib.oracle_filter_outlier_tracks(pts.reshape(-1), hp, hp.shape[0], keep, stats)
    return keep[:hp.shape[0]].astype(bool), stats[0], stats[1]


# ---------------------------------------------------------------------------
# track building: oracle (oracle/tracks_oracle.c) and the reference's own
# bundler_tracks.cc behind oracle/_ref/libref_tracks.so
# ---------------------------------------------------------------------------
REF_TRACKS_SO = os.path.join(ORACLE_DIR, "_ref", "libref_tracks.so")
_ref_tracks = None
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def ref_tracks():
    global _ref_tracks
    if _ref_tracks is None and os.path.exists(REF_TRACKS_SO):
        _ref_tracks = C.CDLL(REF_TRACKS_SO)
    return _ref_tracks


def _tracks_call(fn, view_sizes, colors, pairs, pair_offsets, corr, with_invalid):
    view_sizes = np.ascontiguousarray(view_sizes, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    pair_offsets = np.ascontiguousarray(pair_offsets, dtype=np.int64)
    corr = np.ascontiguousarray(corr, dtype=np.int32).reshape(-1, 2)
    total = int(view_sizes.sum())
    nm = corr.shape[0]
    ids = np.zeros(max(total, 1), dtype=np.int32)
    toff = np.zeros(max(nm, 1) + 1, dtype=np.int64)
    tfeat = np.zeros((max(2 * nm, 1), 2), dtype=np.int32)
    tcol = np.zeros((max(nm, 1), 3), dtype=np.uint8)
    col = None if colors is None else np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 3)
    args = [len(view_sizes), view_sizes.ctypes.data_as(C.c_void_p),
            None if col is None else col.ctypes.data_as(C.c_void_p),
            pairs.shape[0], pairs.ctypes.data_as(C.c_void_p), pair_offsets.ctypes.data_as(C.c_void_p),
            corr.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
            C.c_int64(max(nm, 1)), C.c_int64(max(2 * nm, 1)), toff.ctypes.data_as(C.c_void_p),
            tfeat.ctypes.data_as(C.c_void_p), tcol.ctypes.data_as(C.c_void_p)]
    inv = C.c_int32(-1)
    if with_invalid:
        args.append(C.byref(inv))
    fn.restype = C.c_int
    nt = fn(*args)
    assert nt >= 0
    nf = int(toff[nt])
    return {"track_ids": ids[:total], "track_offsets": toff[:nt + 1], "track_features": tfeat[:nf],
            "track_colors": tcol[:nt], "num_invalid": inv.value}


def oracle_tracks(view_sizes, colors, pairs, pair_offsets, corr):
    return _tracks_call(oracle().oracle_tracks_compute, view_sizes, colors, pairs, pair_offsets, corr, True)


def ref_tracks_compute(view_sizes, colors, pairs, pair_offsets, corr):
    return _tracks_call(ref_tracks().ref_tracks_compute, view_sizes, colors, pairs, pair_offsets, corr, False)


# ---------------------------------------------------------------------------
# cascade hashing: oracle (oracle/cashash_oracle.c) and the reference's own
# cascade_hashing.{h,cc} behind oracle/_ref/libref_cashash.so
# ---------------------------------------------------------------------------
REF_CASHASH_SO = os.path.join(ORACLE_DIR, "_ref", "libref_cashash.so")
_ref_cashash = None
CASHASH_GROUPS, CASHASH_BITS, CASHASH_MIN, CASHASH_MAX = 6, 8, 6, 10   # CascadeHashing::Options defaults


def ref_cashash():
    global _ref_cashash
    if _ref_cashash is None and os.path.exists(REF_CASHASH_SO):
        _ref_cashash = C.CDLL(REF_CASHASH_SO)
        _ref_cashash.ref_cashash_create.restype = C.c_void_p
    return _ref_cashash


class RefCasHash:
    """The reference's sfm::CascadeHashing on u16 / s16 descriptor arrays."""

    def __init__(self, sifts, surfs):
        lib = ref_cashash()
        self.lib = lib
        self.n_sift = [s.shape[0] for s in sifts]
        self.n_surf = [s.shape[0] for s in surfs]
        self.h = C.c_void_p(lib.ref_cashash_create(len(sifts)))
        for v, (s, u) in enumerate(zip(sifts, surfs)):
            # exact inverse of the reference's quantisation (exhaustive_matching.cc:17-38)
            sf = np.ascontiguousarray(s.astype(np.float32) / np.float32(255.0))
            uf = np.ascontiguousarray(u.astype(np.float32) / np.float32(127.0))
            lib.ref_cashash_set_view(self.h, v, sf.ctypes.data_as(C.c_void_p), s.shape[0],
                                     uf.ctypes.data_as(C.c_void_p), u.shape[0])
        lib.ref_cashash_init(self.h)

    def proj(self, type_):
        dim = 128 if type_ == 0 else 64
        prim = np.zeros((dim, dim), np.float32)
        sec = np.zeros((CASHASH_GROUPS, CASHASH_BITS, dim), np.float32)
        self.lib.ref_cashash_get_proj(self.h, type_, prim.ctypes.data_as(C.c_void_p), sec.ctypes.data_as(C.c_void_p))
        return prim, sec

    def local(self, type_, view):
        n = (self.n_sift if type_ == 0 else self.n_surf)[view]
        words = 2 if type_ == 0 else 1
        hashes = np.zeros((max(n, 1), words), np.uint64)
        ids = np.zeros((CASHASH_GROUPS, max(n, 1)), np.uint16)
        if n:
            tmp = np.zeros((CASHASH_GROUPS, n), np.uint16)
            self.lib.ref_cashash_get_local(self.h, type_, view, hashes.ctypes.data_as(C.c_void_p),
                                           tmp.ctypes.data_as(C.c_void_p))
            ids = tmp
        return hashes[:n], ids[:, :n]

    def pairwise_match(self, v1, v2):
        n1 = self.n_sift[v1] + self.n_surf[v1]
        n2 = self.n_sift[v2] + self.n_surf[v2]
        o12 = np.full(max(n1, 1), -7, np.int32)
        o21 = np.full(max(n2, 1), -7, np.int32)
        l12, l21 = C.c_int(), C.c_int()
        self.lib.ref_cashash_pairwise_match(self.h, v1, v2, o12.ctypes.data_as(C.c_void_p), C.byref(l12),
                                            o21.ctypes.data_as(C.c_void_p), C.byref(l21))
        return o12[:l12.value].copy(), o21[:l21.value].copy()

    def close(self):
        if self.h:
            self.lib.ref_cashash_destroy(self.h)
            self.h = None


def oracle_cashash_proj(dim):
    lib = oracle()
    prim = np.zeros((dim, dim), np.float32)
    sec = np.zeros((CASHASH_GROUPS, CASHASH_BITS, dim), np.float32)
    lib.oracle_cashash_proj_matrices(dim, CASHASH_GROUPS, CASHASH_BITS, prim.ctypes.data_as(C.c_void_p),
                                     sec.ctypes.data_as(C.c_void_p))
    return prim, sec


def oracle_cashash_avg(desc_list, dim, div):
    lib = oracle()
    cat = np.ascontiguousarray(np.concatenate([d.reshape(-1, dim) for d in desc_list]).astype(np.int32))
    avg = np.zeros(dim, np.float32)
    lib.oracle_cashash_avg(cat.ctypes.data_as(C.c_void_p), C.c_int64(cat.shape[0]), dim, C.c_float(div),
                           avg.ctypes.data_as(C.c_void_p))
    return avg


def oracle_cashash_hashes(desc, dim, div, avg, prim, sec):
    lib = oracle()
    d = np.ascontiguousarray(desc.reshape(-1, dim).astype(np.int32))
    n = d.shape[0]
    hashes = np.zeros((max(n, 1), dim // 64), np.uint64)
    ids = np.zeros((CASHASH_GROUPS, max(n, 1)), np.uint16)
    if n:
        ids = np.zeros((CASHASH_GROUPS, n), np.uint16)
        lib.oracle_cashash_hashes(d.ctypes.data_as(C.c_void_p), n, dim, C.c_float(div), avg.ctypes.data_as(C.c_void_p),
                                  prim.ctypes.data_as(C.c_void_p), sec.ctypes.data_as(C.c_void_p), CASHASH_GROUPS,
                                  CASHASH_BITS, hashes.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p))
    return hashes[:n], ids[:, :n]


def oracle_cashash_oneway(is_signed, d1, h1, b1, d2, h2, b2, lowe, dist=np.finfo(np.float32).max):
    lib = oracle()
    dim = 64 if is_signed else 128
    dt = np.int16 if is_signed else np.uint16
    d1 = np.ascontiguousarray(d1.reshape(-1, dim).astype(dt))
    d2 = np.ascontiguousarray(d2.reshape(-1, dim).astype(dt))
    h1, h2 = np.ascontiguousarray(h1), np.ascontiguousarray(h2)
    b1, b2 = np.ascontiguousarray(b1), np.ascontiguousarray(b2)
    res = np.full(max(d1.shape[0], 1), -1, np.int32)
    lib.oracle_cashash_oneway(int(is_signed), dim, CASHASH_GROUPS, CASHASH_BITS,
                              d1.ctypes.data_as(C.c_void_p), d1.shape[0], h1.ctypes.data_as(C.c_void_p),
                              b1.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), d2.shape[0],
                              h2.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), C.c_float(lowe),
                              C.c_float(dist), CASHASH_MIN, CASHASH_MAX, res.ctypes.data_as(C.c_void_p))
    return res[:d1.shape[0]]


class OracleCasHash:
    """All stages of the oracle chained like CascadeHashing::init / pairwise_match."""

    def __init__(self, sifts, surfs, sift_lowe=0.8, surf_lowe=0.7):
        self.sifts, self.surfs = sifts, surfs
        self.lowe = (sift_lowe, surf_lowe)
        self.proj = [oracle_cashash_proj(128), oracle_cashash_proj(64)]
        self.avg = [oracle_cashash_avg(sifts, 128, 255.0), oracle_cashash_avg(surfs, 64, 127.0)]
        self.local = [[oracle_cashash_hashes(s, 128, 255.0, self.avg[0], *self.proj[0]) for s in sifts],
                      [oracle_cashash_hashes(u, 64, 127.0, self.avg[1], *self.proj[1]) for u in surfs]]

    def pairwise_match(self, v1, v2, keep_empty_blocks=False):
        """cascade_hashing.cc:73-104.  A descriptor type takes part iff view 1 has
        descriptors of it (:82,95); with an empty set on side 2 the reference
        leaves that part out of its vectors (oneway_match returns before resizing,
        cascade_hashing.h:341-342), and so does this unless keep_empty_blocks asks
        for the exhaustive matcher's layout (the block stays in as -1)."""
        om = oracle_matcher()
        parts = []
        for t, descs in enumerate((self.sifts, self.surfs)):
            a, b = descs[v1], descs[v2]
            if a.shape[0] == 0 or (b.shape[0] == 0 and not keep_empty_blocks):
                parts.append((np.zeros(0, np.int32), np.zeros(0, np.int32)))
                continue
            (h1, b1), (h2, b2) = self.local[t][v1], self.local[t][v2]
            m12 = oracle_cashash_oneway(t, a, h1, b1, b, h2, b2, self.lowe[t])
            m21 = oracle_cashash_oneway(t, b, h2, b2, a, h1, b1, self.lowe[t])
            parts.append(om.remove_inconsistent(m12, m21))
        return om.combine(parts[0][0], parts[0][1], parts[1][0], parts[1][1])


# ---------------------------------------------------------------------------
# group ordering oracle (oracle/groups_oracle.c)
# ---------------------------------------------------------------------------
def oracle_build_groups(view_ids, track_offsets, track_views, group_size=3):
    lib = oracle()
    view_ids = np.ascontiguousarray(view_ids, dtype=np.int32)
    track_offsets = np.ascontiguousarray(track_offsets, dtype=np.int64)
    track_views = np.ascontiguousarray(track_views, dtype=np.int32)
    cap = max(len(view_ids), 1)
    groups = np.zeros((cap, group_size), np.int32)
    gtracks = np.zeros(cap, np.int32)
    lib.oracle_build_groups.restype = C.c_int
    n = lib.oracle_build_groups(len(view_ids), view_ids.ctypes.data_as(C.c_void_p), len(track_offsets) - 1,
                                track_offsets.ctypes.data_as(C.c_void_p), track_views.ctypes.data_as(C.c_void_p),
                                group_size, cap, groups.ctypes.data_as(C.c_void_p), gtracks.ctypes.data_as(C.c_void_p))
    if n < 0:
        return None
    return groups[:n].copy(), gtracks[:n].copy()
