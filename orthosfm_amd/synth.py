"""Synthetic orthographic image sets for the parity tests and bench.py.

Follows SURVEY.md section 8(d): a counter-based generator (splitmix64 ->
uniform, Box-Muller -> normal) so every process derives identical inputs
from (seed, stream, counter) without shared state.  The scene mirrors the
reference testbench's synthetic datasets (src/testbench/dataset_generation.cpp:
14-93: cameras on a ring, every landmark projected orthographically) and the
descriptor statistics of MVE SIFT (unit L2, clamp 0.2, renormalise --
src/mve/sfm/sift.cc:832-839) and SURF (signed, unit L2).

This module is host-side input generation only; it computes nothing on the
matching / bundle-adjustment path.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 0x05F30001
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per input counter (vectorised, uint64)."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, stream: int) -> np.uint64:
    s = splitmix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    t = splitmix64(np.array([(int(s) ^ (stream * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF],
                            dtype=np.uint64))[0]
    return t


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in [0, 1) from counters offset..offset+n-1 of a stream."""
    base = _stream_base(seed, stream)
    with np.errstate(over="ignore"):
        ctr = (np.arange(offset, offset + n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D)) ^ base
    bits = splitmix64(ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n standard normals (Box-Muller on two uniforms per pair)."""
    m = (n + 1) // 2
    u = uniform(seed, stream, 2 * m, 2 * offset)
    u1 = 1.0 - u[0::2]            # (0, 1]
    u2 = u[1::2]
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.empty(2 * m)
    z[0::2] = r * np.cos(2.0 * np.pi * u2)
    z[1::2] = r * np.sin(2.0 * np.pi * u2)
    return z[:n]


# ---------------------------------------------------------------------------
# descriptors
# ---------------------------------------------------------------------------

def _unit_rows(x: np.ndarray) -> np.ndarray:
    n = np.linalg.norm(x, axis=1, keepdims=True)
    n[n == 0] = 1.0
    return x / n


def sift_like(g: np.ndarray) -> np.ndarray:
    """|g| -> unit -> clamp 0.2 -> unit (float32, in [0, 1])."""
    v = _unit_rows(np.abs(g))
    v = np.minimum(v, 0.2)
    return _unit_rows(v).astype(np.float32)


def surf_like(g: np.ndarray) -> np.ndarray:
    return _unit_rows(g).astype(np.float32)


def quantize_sift(f: np.ndarray) -> np.ndarray:
    """Host-side twin of exhaustive_matching.cc:17-27 for input generation
    (the product quantiser is osfm_quantize_sift; the oracle one is
    oracle_convert_sift).  float32 arithmetic throughout."""
    v = np.clip(f.astype(np.float32), np.float32(0), np.float32(1))
    v = np.floor(v * np.float32(255.0) + np.float32(0.5))
    return v.astype(np.uint8).astype(np.uint16)


def quantize_surf(f: np.ndarray) -> np.ndarray:
    v = np.clip(f.astype(np.float32), np.float32(-1), np.float32(1)) * np.float32(127.0)
    r = np.where(v > 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5)))
    return r.astype(np.int8).astype(np.int16)


class ImageSet:
    """V views of one synthetic scene.

    sift[v]: (n_v, 128) uint16 in 0..255, surf[v]: (m_v, 64) int16 in -127..127,
    landmark[v]: (n_v,) landmark id or -1 for a distractor, pos[v]: (n_v, 2)
    pixel positions (float32) from the orthographic projection.
    """

    def __init__(self, sift, surf, landmark, pos, cams, points, width, height):
        self.sift = sift
        self.surf = surf
        self.landmark = landmark
        self.pos = pos
        self.cams = cams
        self.points = points
        self.width = width
        self.height = height

    @property
    def num_views(self):
        return len(self.sift)


def make_image_set(num_views: int, feats_per_view: int, *, seed: int = BASE_SEED,
                   config_id: int = 2, n_surf: int = 0, visibility: float = 0.4,
                   distractor_frac: float = 0.2, noise: float = 0.03,
                   width: int = 2048, height: int = 2048, twin_frac: float = 0.0,
                   unrelated_views: int = 0, landmarks=None, cameras=None) -> ImageSet:
    """L landmarks in the ball |p| <= 0.5 with a base descriptor
    each; view v sees a random subset; descriptor = unit(base + noise*N(0,1)),
    quantised as A1; per-view feature order = descending keypoint scale.

    twin_frac > 0: repeated structure -- that share of the landmarks comes in
    pairs with the SAME base descriptor at different 3-D positions, so a view
    that sees one twin matches it to the other twin of a view that sees only
    that one: mutual matches that pass ratio test and cross-check and are
    geometrically wrong (work for RANSAC-F).  unrelated_views > 0: the last
    views show a different scene (landmarks of their own), so their pairs with
    the others are rejected by the low-res gate / the match-count threshold."""
    n_real = int(round(feats_per_view * (1.0 - distractor_frac)))
    n_dis = feats_per_view - n_real
    L = max(int(round(n_real / visibility)), n_real)
    st = config_id << 32

    # landmarks (given: e.g. the vertices of the reference's test model, tests/golden/cfg1_suzanne.npz --
    # every view then sees a random n_real of them)
    if landmarks is not None:
        points = np.asarray(landmarks, dtype=np.float64).reshape(-1, 3)
        L = points.shape[0]
        assert L >= n_real, (L, n_real)
    else:
        d = normal(seed, st | 1, 3 * L).reshape(L, 3)
        d = _unit_rows(d)
        rad = 0.5 * np.cbrt(uniform(seed, st | 2, L))
        points = d * rad[:, None]
    base_sift = normal(seed, st | 3, 128 * L).reshape(L, 128)
    base_sift_u = _unit_rows(np.minimum(_unit_rows(np.abs(base_sift)), 0.2))
    if twin_frac > 0.0:
        n_tw = int(L * twin_frac) // 2
        base_sift_u[n_tw:2 * n_tw] = base_sift_u[:n_tw]          # landmark n_tw + i is the twin of i
    if unrelated_views > 0:
        # a second scene: descriptors of its own (positions may coincide, they never match)
        other_sift = normal(seed, st | 6, 128 * L).reshape(L, 128)
        other_sift_u = _unit_rows(np.minimum(_unit_rows(np.abs(other_sift)), 0.2))
    base_surf = _unit_rows(normal(seed, st | 4, 64 * L).reshape(L, 64)) if n_surf else None
    # intrinsic keypoint scale of a landmark: MVE sorts features by scale,
    # largest first (feature_set.cc:70), so the low-res prefix of two views
    # holds largely the same landmarks
    lm_scale = uniform(seed, st | 5, L)

    sift, surf, landmark, pos, cams = [], [], [], [], []
    for v in range(num_views):
        sv = st | (0x1000 + 16 * v)
        # camera: phi on a ring, theta/rho in +-30 deg (dataset_generation.cpp:17-30)
        ang = uniform(seed, sv | 0, 2)
        phi = 2.0 * np.pi * v / num_views
        theta = np.deg2rad(-30.0 + 60.0 * ang[0])
        rho = np.deg2rad(-30.0 + 60.0 * ang[1])
        if cameras is not None:
            phi, theta, rho = (float(x) for x in cameras[v])
        cams.append((phi, theta, rho))
        # visible subset: n_real landmarks with the smallest random keys
        keys = uniform(seed, sv | 1, L)
        vis = np.argsort(keys, kind="stable")[:n_real]
        g = normal(seed, sv | 2, 128 * n_real).reshape(n_real, 128)
        scene = other_sift_u if v >= num_views - unrelated_views and unrelated_views > 0 else base_sift_u
        f_real = _unit_rows(np.abs(scene[vis] + noise * g))
        f_real = _unit_rows(np.minimum(f_real, 0.2)).astype(np.float32)
        f_dis = sift_like(normal(seed, sv | 3, 128 * n_dis).reshape(n_dis, 128))
        f = np.concatenate([f_real, f_dis], axis=0)
        lm = np.concatenate([vis, -np.ones(n_dis, dtype=np.int64)])
        if v >= num_views - unrelated_views and unrelated_views > 0:
            lm = np.concatenate([vis + L, -np.ones(n_dis, dtype=np.int64)])      # ids of the other scene
        # per-view order: descending scale (landmark scale + detection jitter;
        # distractors get a random scale)
        jit = uniform(seed, sv | 4, feats_per_view)
        sc = np.concatenate([lm_scale[vis] + 0.02 * (jit[:n_real] - 0.5), jit[n_real:]])
        order = np.argsort(-sc, kind="stable")
        sift.append(quantize_sift(f[order]))
        landmark.append(lm[order])
        # positions
        xy = project_euler(points[vis], phi, theta, rho, 0.0, 0.0, 1.0, width, height)
        xy_d = uniform(seed, sv | 5, 2 * n_dis).reshape(n_dis, 2) * [width, height]
        pos.append(np.concatenate([xy, xy_d], axis=0)[order].astype(np.float32))
        if n_surf:
            m_real = min(n_surf, n_real)
            gs = normal(seed, sv | 6, 64 * m_real).reshape(m_real, 64)
            fs = surf_like(base_surf[vis[:m_real]] + noise * gs)
            fd = surf_like(normal(seed, sv | 7, 64 * (n_surf - m_real)).reshape(n_surf - m_real, 64))
            o2 = np.argsort(uniform(seed, sv | 8, n_surf), kind="stable")
            surf.append(quantize_surf(np.concatenate([fs, fd], axis=0)[o2]))
        else:
            surf.append(np.zeros((0, 64), dtype=np.int16))
    return ImageSet(sift, surf, landmark, pos, cams, points, width, height)


def suzanne_scene(num_cameras: int = 3):
    """BASELINE configs[0]: the landmarks of the reference's test model and the first cameras its test
    bench draws (tests/golden/cfg1_suzanne.npz, made by tests/golden/make_cfg1_suzanne.py from
    /root/reference/resources/Suzanne.ply and src/testbench/dataset_generation.cpp:14-38).
    Returns (points (7872, 3), cameras [(phi, theta, rho) in radians], width, height)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cfg1_suzanne.npz"))
    cams = [tuple(np.deg2rad(g["cams_deg"][c])) for c in range(num_cameras)]
    return g["points"], cams, int(g["width"]), int(g["height"])


def make_suzanne_ba_scene(model: int = 0, num_cameras: int = 3, *, seed: int = BASE_SEED, noise_px: float = 0.0,
                          rot_perturb_deg: float = 2.0, off_perturb: float = 0.01, point_perturb: float = 0.01,
                          euler_free: int = 5):
    """The reference's synthetic dataset (dataset_generation.cpp:40-93): one track per Suzanne vertex,
    seen by every camera, observations = exact orthographic projections (stored through float32 as
    Feature::x / y are) -- as a bundle-adjustment problem from a perturbed start, camera 0 fixed."""
    pts, cams, width, height = suzanne_scene(num_cameras)
    C, M = num_cameras, pts.shape[0]
    gt = np.zeros((C, 7))
    for c, (phi, theta, rho) in enumerate(cams):
        if model == MODEL_QUATERNION:
            gt[c, :4] = euler_to_quat(phi, theta, rho)
            gt[c, 4:] = (0.0, 0.0, 1.0)
        else:
            gt[c, :6] = (phi, theta, rho, 0.0, 0.0, 1.0)
    obs_point = np.repeat(np.arange(M), C)
    obs_camera = np.tile(np.arange(C), M)
    xy = np.zeros((M * C, 2))
    for c in range(C):
        sel = np.nonzero(obs_camera == c)[0]
        xy[sel] = _project(model, gt[c], pts[obs_point[sel]], width, height)
    st = 0x5A << 32
    if noise_px > 0:
        xy += noise_px * normal(seed, st | 1, 2 * M * C).reshape(-1, 2)
    xy = xy.astype(np.float32).astype(np.float64)
    camsp = gt.copy()
    pr = normal(seed, st | 2, 3 * C).reshape(C, 3)
    po = normal(seed, st | 3, 2 * C).reshape(C, 2)
    for c in range(1, C):
        if model == MODEL_QUATERNION:
            axis = pr[c] / np.linalg.norm(pr[c])
            a = np.deg2rad(rot_perturb_deg)
            dq = np.array([*(np.sin(a / 2) * axis), np.cos(a / 2)])
            camsp[c, :4] = quat_mul(dq, gt[c, :4])
            camsp[c, 4:6] = gt[c, 4:6] + off_perturb * po[c]
        else:
            camsp[c, :3] = gt[c, :3] + np.deg2rad(rot_perturb_deg) * pr[c] / np.sqrt(3.0)
            camsp[c, 3:5] = gt[c, 3:5] + off_perturb * po[c]
    P = np.ones((M, 4))
    P[:, :3] = pts + point_perturb * normal(seed, st | 4, 3 * M).reshape(M, 3)
    const = np.zeros((C, 7), dtype=np.uint8)
    if model == MODEL_QUATERNION:
        const[:, 6] = 1
    else:
        free = [0, 1, 2, 3, 4][:euler_free] if euler_free <= 5 else [0, 1, 2, 3, 4, 5]
        const[:, :] = 1
        for s_ in free:
            const[:, s_] = 0
    const[0, :] = 1
    return BaScene(model, camsp, const, np.full(C, width, np.int32), np.full(C, height, np.int32), P, xy,
                   obs_camera.astype(np.int32), obs_point.astype(np.int32), gt, pts)


def add_peaky_rows(iset: "ImageSet", k: int, *, seed: int = BASE_SEED, stream: int = 0x7EA) -> list:
    """Replaces k SIFT rows of every view by descriptors with their energy in one to three
    bins, built the way MVE builds them (normalise, clamp at 0.2, normalise again,
    src/mve/sfm/sift.cc:830-839): the renormalisation lifts the clamped bins to 0.5-1.0,
    i.e. bytes 128..255 after convert_descriptor -- the rows real images produce at corners
    and line ends, and the ones a plain Gaussian generator never draws.  Half of the k rows
    of a view show landmarks shared by all views (the same peaky base descriptor plus
    noise, so they are each other's best matches), the rest are unique to the view.
    Returns the replaced row ids per view."""
    rng = np.random.default_rng([seed & 0xFFFFFFFF, stream, k])
    n_shared = (k + 1) // 2

    def peaky(n):
        f = np.abs(rng.normal(size=(n, 128))) * 0.01
        for i in range(n):
            nb = int(rng.integers(1, 4))
            f[i, rng.choice(128, nb, replace=False)] = rng.uniform(0.6, 1.0, nb)
        return f

    base = peaky(n_shared)
    rows_out = []
    for v in range(iset.num_views):
        n = iset.sift[v].shape[0]
        rows = np.sort(rng.choice(n, min(k, n), replace=False))
        f = np.concatenate([base + np.abs(rng.normal(size=base.shape)) * 0.004, peaky(k - n_shared)], axis=0)[:len(rows)]
        q = quantize_sift(sift_like(f))
        assert (q.max(axis=1) > 127).all()
        iset.sift[v][rows] = q[rng.permutation(len(rows))]
        rows_out.append(rows)
    return rows_out


# ---------------------------------------------------------------------------
# camera models used to PLACE synthetic observations (generation only)
# ---------------------------------------------------------------------------

def euler_matrix(phi, theta, rho):
    """S = Rz(phi) * Rx(theta + pi/2) * Rz-form(rho); reference
    OrthographicCamera.cpp:78-96 (getSphericalProjectionMatrix)."""
    om = theta + 0.5 * np.pi
    Ry = np.array([[np.cos(rho), -np.sin(rho), 0], [np.sin(rho), np.cos(rho), 0], [0, 0, 1.0]])
    Rx = np.array([[1.0, 0, 0], [0, np.cos(om), -np.sin(om)], [0, np.sin(om), np.cos(om)]])
    Rz = np.array([[np.cos(phi), -np.sin(phi), 0], [np.sin(phi), np.cos(phi), 0], [0, 0, 1.0]])
    return (Rz @ Rx) @ Ry


_T = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])


def project_euler(p, phi, theta, rho, off_x, off_y, scale, width, height):
    """OrthographicCamera::projectPointOntoImagePlane (OrthographicCamera.cpp:63-76)."""
    S = euler_matrix(phi, theta, rho)
    loc = (S.T @ _T @ p.T).T / scale
    x = width * ((loc[:, 0] - off_x) / -2.0 + 0.5)
    y = height * ((loc[:, 1] - off_y) / -2.0 + 0.5)
    return np.stack([x, y], axis=1)


def quat_rotate(q, v):
    """Rotate rows of v by unit quaternion q = (x, y, z, w)."""
    u = q[:3]
    w = q[3]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def project_quat(p, q, off_x, off_y, scale, width, height):
    """Pixel projection of the quaternion model (residual functor
    OrthographicQuaternionReprojectorError.h:24-67 without the observation)."""
    qi = np.array([-q[0], -q[1], -q[2], q[3]]) / np.dot(q, q)
    loc = quat_rotate(qi, p)
    x = width * (((loc[:, 0] / scale) - off_x) / -2.0 + 0.5)
    y = height * (((loc[:, 1] / scale) - off_y) / -2.0 + 0.5)
    return np.stack([x, y], axis=1)


def euler_to_quat(phi, theta, rho):
    """Quaternion (x, y, z, w) whose inverse rotation equals S^T * T, i.e. the
    same world->local map as the Euler model, so both models can share scenes."""
    R = (euler_matrix(phi, theta, rho).T @ _T).T     # local -> world
    return mat_to_quat(R)


def mat_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    return np.array([x, y, z, w])


# ---------------------------------------------------------------------------
# bundle-adjustment scenes
# ---------------------------------------------------------------------------

MODEL_QUATERNION = 0
MODEL_EULER = 1


class BaScene:
    """Flattened BA problem (the arrays of osfm_ba_problem) plus ground truth."""

    def __init__(self, model, cam_params, cam_const, img_w, img_h, points, obs_xy, obs_camera,
                 obs_point, gt_cams, gt_points):
        self.model = model
        self.cam_params = cam_params
        self.cam_const = cam_const
        self.img_w = img_w
        self.img_h = img_h
        self.points = points
        self.obs_xy = obs_xy
        self.obs_camera = obs_camera
        self.obs_point = obs_point
        self.gt_cams = gt_cams
        self.gt_points = gt_points

    def copy(self):
        return BaScene(self.model, self.cam_params.copy(), self.cam_const.copy(), self.img_w.copy(),
                       self.img_h.copy(), self.points.copy(), self.obs_xy.copy(),
                       self.obs_camera.copy(), self.obs_point.copy(), self.gt_cams.copy(),
                       self.gt_points.copy())


def _project(model, cam, p, w, h):
    if model == MODEL_QUATERNION:
        return project_quat(p, cam[:4], cam[4], cam[5], cam[6], w, h)
    return project_euler(p, cam[0], cam[1], cam[2], cam[3], cam[4], cam[5], w, h)


def make_ba_scene(model: int, num_cameras: int, num_points: int, *, seed: int = BASE_SEED,
                  config_id: int = 4, min_len: int = 3, max_len: int = 12, noise_px: float = 0.5,
                  rot_perturb_deg: float = 2.0, off_perturb: float = 0.01,
                  point_perturb: float = 0.01, width: int = 2048, height: int = 2048,
                  euler_free: int = 5) -> BaScene:
    """C cameras on a ring (phi = 360 v / C, theta/rho in +-30 deg), M points
    in the ball |p| <= 0.5, each seen by a contiguous arc of U{min_len..max_len}
    cameras; observations = exact projection + N(0, noise_px) stored through
    float32 (Feature::x/y are float, track.h:26-27).  Start = ground truth
    perturbed (cameras by rot_perturb_deg / off_perturb, points by
    point_perturb); camera 0 fixed; scale fixed (OrthoQuaternionCamera.h:89-91
    / setDegreesOfFreedom(4), OrthographicCamera.cpp:195-207)."""
    st = config_id << 32
    C, M = num_cameras, num_points
    ang = uniform(seed, st | 0x21, 2 * C).reshape(C, 2)
    gt = np.zeros((C, 7))
    for c in range(C):
        phi = 2.0 * np.pi * c / C
        theta = np.deg2rad(-30.0 + 60.0 * ang[c, 0])
        rho = np.deg2rad(-30.0 + 60.0 * ang[c, 1])
        if model == MODEL_QUATERNION:
            gt[c, :4] = euler_to_quat(phi, theta, rho)
            gt[c, 4:] = (0.0, 0.0, 1.0)
        else:
            gt[c, :6] = (phi, theta, rho, 0.0, 0.0, 1.0)
    d = _unit_rows(normal(seed, st | 0x22, 3 * M).reshape(M, 3))
    pts = d * (0.5 * np.cbrt(uniform(seed, st | 0x23, M)))[:, None]
    ln = min_len + np.floor(uniform(seed, st | 0x24, M) * (max_len - min_len + 1)).astype(np.int64)
    ln = np.minimum(ln, C)
    first = np.floor(uniform(seed, st | 0x25, M) * C).astype(np.int64)
    obs_point = np.repeat(np.arange(M), ln)
    k_in = np.arange(obs_point.size) - np.repeat(np.cumsum(ln) - ln, ln)
    obs_camera = (first[obs_point] + k_in) % C
    O = obs_point.size
    xy = np.zeros((O, 2))
    for c in range(C):
        sel = np.nonzero(obs_camera == c)[0]
        if sel.size:
            xy[sel] = _project(model, gt[c], pts[obs_point[sel]], width, height)
    xy += noise_px * normal(seed, st | 0x26, 2 * O).reshape(O, 2)
    xy = xy.astype(np.float32).astype(np.float64)

    # perturbed start
    cams = gt.copy()
    pr = normal(seed, st | 0x27, 3 * C).reshape(C, 3)
    po = normal(seed, st | 0x28, 2 * C).reshape(C, 2)
    for c in range(1, C):
        if model == MODEL_QUATERNION:
            axis = pr[c] / np.linalg.norm(pr[c])
            a = np.deg2rad(rot_perturb_deg)
            dq = np.array([*(np.sin(a / 2) * axis), np.cos(a / 2)])
            cams[c, :4] = quat_mul(dq, gt[c, :4])
            cams[c, 4:6] = gt[c, 4:6] + off_perturb * po[c]
        else:
            cams[c, :3] = gt[c, :3] + np.deg2rad(rot_perturb_deg) * pr[c] / np.sqrt(3.0)
            cams[c, 3:5] = gt[c, 3:5] + off_perturb * po[c]
    P = np.ones((M, 4))
    P[:, :3] = pts + point_perturb * normal(seed, st | 0x29, 3 * M).reshape(M, 3)

    const = np.zeros((C, 7), dtype=np.uint8)
    if model == MODEL_QUATERNION:
        const[:, 6] = 1                      # m_fixScale = true
    else:
        free = [0, 1, 2, 3, 4][:euler_free] if euler_free <= 5 else [0, 1, 2, 3, 4, 5]
        const[:, :] = 1
        for s_ in free:
            const[:, s_] = 0
    const[0, :] = 1                          # camera 0 fixed (reconstruct.cpp:215)
    wh = np.full(C, width, dtype=np.int32), np.full(C, height, dtype=np.int32)
    return BaScene(model, cams, const, wh[0], wh[1], P, xy, obs_camera.astype(np.int32),
                   obs_point.astype(np.int32), gt, pts)


def quat_mul(a, b):
    """Hamilton product of quaternions stored (x, y, z, w)."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])
