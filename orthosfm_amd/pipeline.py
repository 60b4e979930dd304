"""End-to-end driver over the C ABI: everything the backend replaces, in the order
the reference calls it, timed as one job.

    orthosfm::reconstruct                     src/sfm/reconstruct.cpp:32-172
      calculateTracksUsingMVE                 src/matching/matching_mve.cpp:247-473
        bundler::Matching::init / compute       (low-res gate, two-way match, cross-check, RANSAC-F)
        bundler::Tracks::compute
        MVE track -> orthosfm::Track            :455-466
      runPoseEstimation                       src/sfm/reconstruct.cpp:174-295
        buildGroups                             :181-184
        per group (:193-275): initial alignment, filterTracksWithReprojectionError,
          local bundle adjustment (retriangulated copy), merge, triangulateTracks,
          every 3rd group: global bundle adjustment + filterOutlierTracks +
          filterTracksWithReprojectionError; final bundle adjustment (:281)

What is NOT here, because SURVEY section 8 puts it out of scope: image IO / feature
extraction (the synthetic ImageSet stands in for the views with their
descriptors) and the Tomasi-Kanade initial alignment of a group
(ReconstructionAlgorithm::calculateInitialAlignment).  The pose a new camera
starts its local bundle adjustment from is its ground-truth pose perturbed by a
given rotation / offset -- the role TK's estimate plays in the reference; the
scene is expressed in the frame in which camera 0 is canonical, which is the
frame normalizeScene (reconstruct.cpp:228,268) keeps the reconstruction in.

The reference drags std::vector<Track> copies through this loop (one copy of all
tracks per group, :205).  Here the tracks are ONE structure of arrays (features
in track order) plus alive flags; a per-view inverted index serves the
three-camera steps, so a group costs what its cameras see, not what the scene
holds.  All arithmetic -- matching, RANSAC, track merge, group scores,
triangulation, reprojection errors, LM -- happens behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
from dataclasses import dataclass, field

import numpy as np

from . import ba as B
from . import capi, groups as G, synth, tracks as T
from .matching import HipCascadeHashing, HipExhaustiveMatching

MAX_REPROJECTION_ERROR = 1.5            # outlier_filtering.cpp:140
GLOBAL_BA_INTERVAL = 3                  # reconstruct.cpp:187


# ---------------------------------------------------------------------------
# tracks as a structure of arrays
# ---------------------------------------------------------------------------
def _stable_order_by_view(view, num_views):
    """Stable argsort of the view ids: on 16-bit keys numpy's stable sort is a radix sort (a few ms
    for a million features; the merge sort it uses on int32 keys takes ten times as long)."""
    key = view.astype(np.uint16) if num_views <= 65535 else view
    return np.argsort(key, kind="stable").astype(np.int64)


class TrackTable:
    """All tracks of the scene: features in track order (the order of
    Tracks::compute), pixel positions as the float32 values Feature::x/y hold
    (track.h:26-27), the homogeneous point and hasPoint() per track, and the
    alive flags the filters clear (a filtered std::vector<Track> in the
    reference is the subset with both flags set here)."""

    def __init__(self, offsets, view, feat, xy, num_views, _table=None):
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.num_views = int(num_views)
        T_ = self.offsets.shape[0] - 1
        if _table is not None:           # from_mve: everything per feature comes from one pass of the library
            self.view, self.feat, self.xy, self.track_of, self.by_view, self.view_start = _table
        else:
            self.view = np.ascontiguousarray(view, dtype=np.int32)
            self.feat = np.ascontiguousarray(feat, dtype=np.int32)
            self.xy = np.ascontiguousarray(xy, dtype=np.float32).astype(np.float64)
            self.track_of = np.repeat(np.arange(T_, dtype=np.int32), np.diff(self.offsets))
        self.point = np.zeros((T_, 4))
        self.has_point = np.zeros(T_, dtype=bool)
        self.alive_t = np.ones(T_, dtype=bool)
        self.alive_f = np.ones(self.view.shape[0], dtype=bool)
        # kept up to date by kill() / align_view(), so that the per-call selections are one pass:
        self.live_f = np.ones(self.view.shape[0], dtype=bool)        # alive_f & alive_t[track_of]
        self.cam_f = np.full(self.view.shape[0], -1, dtype=np.int32)  # camera index of the feature's view, -1: not aligned
        self._lengths = np.diff(self.offsets).astype(np.int64)        # alive features per alive track
        if _table is None:
            self.by_view = _stable_order_by_view(self.view, self.num_views)       # feature ids grouped by view, ascending inside
            self.view_start = np.concatenate([[0], np.cumsum(np.bincount(self.view, minlength=self.num_views))]).astype(np.int64)

    @classmethod
    def from_mve(cls, track_offsets, track_features, norm_positions, image_width, num_views):
        """matching_mve.cpp:455-466: pixel = imageWidth * (normalised + 0.5) for BOTH axes
        (double arithmetic, stored as float)."""
        if len(norm_positions) != num_views:
            raise ValueError("from_mve: one position array per view")
        table = T.feature_table(track_offsets, track_features, norm_positions, image_width)
        return cls(track_offsets, None, None, None, num_views, _table=table)

    @property
    def num_tracks(self):
        return int(self.alive_t.sum())

    def compacted(self):
        """A working table that holds the live tracks and features only, in the same order, with
        orig_track / orig_feat naming their rows in the table the job started from.  The filters
        of the reference return ever smaller std::vector<Track> copies (reconstruct.cpp:205,264-265);
        here flags are cleared instead, and after the first global round most of the scene is flags
        (200 views x 20k features: 3.2 M features of which ~5 % stay live) -- every later selection
        would still scan all of it."""
        sel_t = np.flatnonzero(self.alive_t)
        sel_f = np.flatnonzero(self.live_f)
        w = TrackTable.__new__(TrackTable)
        lens = self._lengths[sel_t]
        w.offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        w.view, w.feat, w.xy = self.view[sel_f], self.feat[sel_f], self.xy[sel_f]
        w.num_views = self.num_views
        nt = sel_t.shape[0]
        w.track_of = np.repeat(np.arange(nt, dtype=np.int32), lens)
        w.point, w.has_point = self.point[sel_t], self.has_point[sel_t]
        w.alive_t = np.ones(nt, dtype=bool)
        w.alive_f = np.ones(sel_f.shape[0], dtype=bool)
        w.live_f = np.ones(sel_f.shape[0], dtype=bool)
        w.cam_f = self.cam_f[sel_f]
        w._lengths = lens.astype(np.int64)
        order = _stable_order_by_view(w.view, w.num_views)
        w.by_view = order
        w.view_start = np.concatenate([[0], np.cumsum(np.bincount(w.view, minlength=w.num_views))]).astype(np.int64)
        w.orig_track = getattr(self, "orig_track", np.arange(self.alive_t.shape[0]))[sel_t]
        w.orig_feat = getattr(self, "orig_feat", np.arange(self.alive_f.shape[0]))[sel_f]
        return w

    def write_back(self, work):
        """The state of a working table (compacted(), possibly several times) into this one."""
        if work is self:
            return
        ot, of = work.orig_track, work.orig_feat
        self.alive_t[:] = False
        self.alive_t[ot[work.alive_t]] = True
        self.alive_f[:] = False
        self.alive_f[of[work.alive_f]] = True
        self.live_f[:] = False
        self.live_f[of[work.live_f]] = True
        self.has_point[:] = False
        self.has_point[ot] = work.has_point
        self.point[ot] = work.point
        self.cam_f[of] = work.cam_f
        self._lengths[:] = 0
        self._lengths[ot] = work._lengths

    def features_of_views(self, views):
        """Alive features of alive tracks seen by the given views, in track order."""
        idx = np.concatenate([self.by_view[self.view_start[v]:self.view_start[v + 1]] for v in views])
        idx = idx[self.live_f[idx]]
        idx.sort()
        return idx

    def alive_lengths(self):
        """Features left per track (0 for dead tracks)."""
        return self._lengths

    def align_view(self, v, cam_index):
        """The view got a camera: its features select it from now on."""
        self.cam_f[self.by_view[self.view_start[v]:self.view_start[v + 1]]] = cam_index

    def kill(self, tracks=None, features=None):
        """Clears alive flags (what a filter's smaller output list means here)."""
        if features is not None and len(features):
            f = np.asarray(features)
            f = f[self.alive_f[f]]
            self.alive_f[f] = False
            was_live = self.live_f[f]
            self.live_f[f] = False
            np.subtract.at(self._lengths, self.track_of[f[was_live]], 1)
        if tracks is not None and len(tracks):
            t = np.asarray(tracks)
            t = t[self.alive_t[t]]
            self.alive_t[t] = False
            self._lengths[t] = 0
            lo, n = self.offsets[t], self.offsets[t + 1] - self.offsets[t]
            if t.size:
                fid = np.repeat(lo - np.concatenate([[0], np.cumsum(n)[:-1]]), n) + np.arange(int(n.sum()))
                self.live_f[fid] = False


def _runs(sorted_ids):
    """For a non-decreasing id array: (unique ids, first index of each run, run lengths,
    run number of every element) -- np.unique without the sort."""
    n = sorted_ids.shape[0]
    if n == 0:
        z = np.zeros(0, dtype=np.int64)
        return sorted_ids[:0], z, z, z
    flag = np.empty(n, dtype=bool)
    flag[0] = True
    np.not_equal(sorted_ids[1:], sorted_ids[:-1], out=flag[1:])
    first = np.flatnonzero(flag)
    cnt = np.diff(np.append(first, n))
    run = np.cumsum(flag) - 1
    return sorted_ids[first], first, cnt, run


# ---------------------------------------------------------------------------
# cameras
# ---------------------------------------------------------------------------
_Tm = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])


def _euler_from_S(S):
    """(phi, theta, rho) with S = Rz(phi) Rx(theta + pi/2) Rz(rho) (synth.euler_matrix)."""
    om = np.arccos(np.clip(S[2, 2], -1.0, 1.0))
    if abs(np.sin(om)) < 1e-12:
        return np.arctan2(S[1, 0], S[0, 0]), om - 0.5 * np.pi, 0.0
    phi = np.arctan2(S[0, 2], -S[1, 2])
    rho = np.arctan2(S[2, 0], S[2, 1])
    return phi, om - 0.5 * np.pi, rho


def _quat_to_mat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _cam_rotation(model, p):
    """local -> world rotation of a camera parameter vector."""
    if model == B.MODEL_QUATERNION:
        return _quat_to_mat(p[:4])
    return _Tm.T @ synth.euler_matrix(p[0], p[1], p[2])


def _set_cam_rotation(model, p, R):
    if model == B.MODEL_QUATERNION:
        q = synth.mat_to_quat(R)
        if np.dot(q, p[:4]) < 0:
            q = -q
        p[:4] = q
    else:
        p[0], p[1], p[2] = _euler_from_S(_Tm @ R)


def canonical_ground_truth(iset, model):
    """Ground-truth cameras and landmarks in the frame in which camera 0 is canonical
    (identity rotation / zero angles), i.e. after normalizeScene.  The half-pixel of
    the MVE -> pixel conversion (matching_mve.cpp:463) goes into the offsets, so the
    cameras project the landmarks onto the feature positions the tracks carry."""
    V = iset.num_views
    R = [_Tm.T @ synth.euler_matrix(*iset.cams[v]) for v in range(V)]
    A = R[0].T
    gt = np.zeros((V, 7))
    for v in range(V):
        Rv = A @ R[v]
        if model == B.MODEL_QUATERNION:
            gt[v, :4] = synth.mat_to_quat(Rv)
            gt[v, 4:] = (1.0 / iset.width, 1.0 / iset.height, 1.0)
        else:
            gt[v, :3] = _euler_from_S(_Tm @ Rv)
            gt[v, 3:6] = (1.0 / iset.width, 1.0 / iset.height, 1.0)
    pts = (A @ iset.points.T).T
    return gt, pts


def euler_dof_of_solver(solver):
    """setSolverType (OrthographicReconstructionAlgorithm.cpp:15-34): solver 1 -> 1 (phi),
    2 -> 2 (phi, theta), 3 -> 4 (phi, theta, roll, offsets)."""
    return {1: 1, 2: 2, 3: 4}.get(int(solver), 4)


def default_const_mask(model, fixed=False, euler_dof=4):
    """SetupParameterBlocks' constancy per parameter: quaternion model rotation/offset
    free, scale fixed (OrthoQuaternionCamera.h:89-91); Euler model by degrees of freedom
    (OrthographicCamera.cpp:195-207; solver 3 = 4)."""
    if model == B.MODEL_QUATERNION:
        m = np.array([0, 0, 0, 0, 0, 0, 1], dtype=np.uint8)
    else:
        d = euler_dof
        m = np.array([d < 1, d < 2, d < 3, d < 4, d < 4, d < 5, True], dtype=np.uint8)
    if fixed:
        m[:] = 1
    return m


def align_to_global(model, local, global_):
    """alignToGlobalCameras: the rotation that carries the local copies of the shared
    cameras onto their global poses, applied to all local cameras.  Both models use the
    least-squares rotation between the camera frames (origin and axes of the shared
    cameras, as OrthographicReconstructionAlgorithm.cpp:98-139 feeds to umeyama)."""
    src, dst = [], []
    for pl, pg in zip(local, global_):
        if pg is None:
            continue
        Rl, Rg = _cam_rotation(model, pl), _cam_rotation(model, pg)
        for k in range(3):
            src.append(Rl[:, k]); dst.append(Rg[:, k])
        src.append(-10.0 * Rl[:, 2]); dst.append(-10.0 * Rg[:, 2])
    if not src:
        return
    S, D = np.array(src).T, np.array(dst).T
    S = S - S.mean(1, keepdims=True)
    D = D - D.mean(1, keepdims=True)
    U, _, Vt = np.linalg.svd(D @ S.T)
    d = np.sign(np.linalg.det(U @ Vt))
    Rot = U @ np.diag([1.0, 1.0, d]) @ Vt
    for pl in local:
        _set_cam_rotation(model, pl, Rot @ _cam_rotation(model, pl))


# ---------------------------------------------------------------------------
# the job
# ---------------------------------------------------------------------------
@dataclass
class Timings:
    upload_s: float = 0.0
    setup_s: float = 0.0          # inside matching_s: creating the matcher, waiting for the page-locked list buffers
    matching_s: float = 0.0
    tracks_s: float = 0.0
    tracks_busy_s: float = 0.0
    convert_s: float = 0.0
    groups_s: float = 0.0
    local_ba_s: float = 0.0
    local_filter_s: float = 0.0
    triangulate_s: float = 0.0
    global_ba_s: float = 0.0
    outlier_filter_s: float = 0.0
    pose_host_s: float = 0.0
    pose_s: float = 0.0
    total_s: float = 0.0


@dataclass
class BaCall:
    kind: str               # "local" / "global" / "final"
    cameras: int
    points: int
    observations: int
    iterations: int
    ms: float
    lm_ms: float


@dataclass
class Result:
    cam_params: np.ndarray
    aligned_views: list
    tracks: TrackTable
    groups: list
    timings: Timings
    ba_calls: list
    num_pairs: int = 0
    matched_pairs: int = 0
    correspondences: int = 0
    num_mve_tracks: int = 0
    invalid_mve_tracks: int = 0
    captured: dict = field(default_factory=dict)
    pair_status: np.ndarray | None = None


def _problem(model, cams, const, width, height, points, xy, obs_cam, obs_pt):
    return B.FlatProblem(model, cams, const, np.full(cams.shape[0], width, np.int32),
                         np.full(cams.shape[0], height, np.int32), points, xy, obs_cam, obs_pt)


class _ListBufferPool:
    """Page-locked (rows, 2) int32 list buffers, kept between jobs (page-locked memory is expensive to allocate --
    0.1 to 0.5 s for the 200 MB of a 200-view job, depending on the host).  A job CHECKS its buffers OUT and hands
    them back when it is done, so two reconstruct() calls on two threads never share one; an idle buffer that is too
    small for the job that asks is freed before the larger one is made."""

    def __init__(self):
        self._lock = threading.Lock()
        self._idle = []

    def acquire(self, rows):
        stale = None
        with self._lock:
            fit = [b for b in self._idle if b.shape[0] >= rows]
            if fit:
                b = min(fit, key=lambda x: x.shape[0])
                self._idle = [x for x in self._idle if x is not b]
                return b
            if self._idle:
                stale = max(self._idle, key=lambda x: x.shape[0])
                self._idle = [x for x in self._idle if x is not stale]
        if stale is not None:
            capi.pinned_free(stale)
        return capi.pinned_rows(rows)

    def release(self, buf):
        with self._lock:
            self._idle.append(buf)

    def idle_rows(self):
        with self._lock:
            return sorted(int(b.shape[0]) for b in self._idle)


_list_buffers = _ListBufferPool()


def match_and_build_tracks(iset, matcher="exhaustive", device=0, verify=True, timings=None, pairs=None):
    """calculateTracksUsingMVE up to (and including) the track conversion."""
    tm = timings if timings is not None else Timings()
    V = iset.num_views
    o = capi.default_match_options()
    o.geometric_verification = 1 if verify else 0
    cls = HipExhaustiveMatching if matcher == "exhaustive" else HipCascadeHashing
    import queue
    import threading
    t0 = time.perf_counter()
    m = cls(V, device=device, options=o, copy_results=False)
    tm.setup_s = time.perf_counter() - t0
    norm = [None] * V
    W, H = iset.width, iset.height
    # The views go up on a thread of their own (the library gives uploads their own stream and lock): the
    # first batch of pairs -- in the reference's order pair i names views up to sqrt(2 i) -- starts as soon
    # as the views it names are there, the rest of the bank follows beside the matching.
    uploaded = [0]
    up_cv = threading.Condition()
    up_err = []

    def upload_worker():
        try:
            for v in range(V):
                m.set_view(v, iset.sift[v], iset.surf[v] if iset.surf[v].shape[0] else None)
                xy = ((iset.pos[v] + 0.5 - np.array([W / 2, H / 2])) / max(W, H)).astype(np.float32)   # feature_set.cc:42-55
                if iset.surf[v].shape[0]:
                    xy = np.concatenate([xy, np.zeros((iset.surf[v].shape[0], 2), np.float32)])
                norm[v] = xy
                if verify:
                    m.set_positions(v, xy)
                with up_cv:
                    uploaded[0] = v + 1
                    up_cv.notify_all()
        except Exception as e:
            with up_cv:
                up_err.append(e)
                up_cv.notify_all()

    def wait_for_views(n_views):
        with up_cv:
            while uploaded[0] < n_views and not up_err:
                up_cv.wait()
        if up_err:
            raise up_err[0]

    up_thread = threading.Thread(target=upload_worker, daemon=True)
    up_thread.start()
    upload_wait = 0.0
    if pairs is None:
        pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
    pf = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    sizes = np.array([iset.sift[v].shape[0] + iset.surf[v].shape[0] for v in range(V)], dtype=np.int32)
    # Matching and track building side by side: the pair list goes through the matcher in batches
    # (in pair order), and while the device works on batch k + 1 a second thread merges the lists of
    # batch k into the tracks (Tracks::compute is a sequential merge over the pairs in that order, so
    # it takes them as they come; both are C calls that release the interpreter lock).  Two list
    # buffers alternate: a batch's lists stay where they are until the merge has read them.
    builder = T.TracksBuilder(sizes)
    n_batches = max(1, min(32, pf.shape[0] // 512))
    bounds = np.linspace(0, pf.shape[0], n_batches + 1).astype(np.int64)
    if n_batches > 1 and matcher == "exhaustive" and os.environ.get("OSFM_PIPELINE_EARLY_BATCHES", "1") != "0":
        # The first batch of 32 equal ones names views up to sqrt(2 pairs / 32) -- 35 of 200 -- and the device idles
        # until they are all up (2 - 3 ms each: 0.1 s of a 2 s job).  In the reference's pair order the first
        # v (v - 1) / 2 pairs name v views: short batches (6, 12, 20, 32, 48 and 64 views, as far as they stay below
        # the first regular one) let the matching start 15 ms in and keep it fed while the rest goes up.
        early = [v * (v - 1) // 2 for v in (6, 12, 20, 32, 48, 64) if v * (v - 1) // 2 < int(bounds[1]) * 3 // 4]
        bounds = np.array(sorted(set([0] + early + [int(b) for b in bounds[1:]])), dtype=np.int64)
        n_batches = bounds.shape[0] - 1
    m.expect_pairs(int(np.diff(bounds).max()) if n_batches > 0 else 0)      # the short first batches do not size the work arrays
    pair_cap = np.minimum(sizes[pf[:, 0]], sizes[pf[:, 1]]).astype(np.int64) if pf.shape[0] else np.zeros(0, np.int64)
    max_cap = max(int(max((pair_cap[bounds[k]:bounds[k + 1]].sum() for k in range(n_batches)), default=1)), 1)
    # The two list buffers are page-locked on a thread of their own: the first is there when the first views
    # are (the uploads run meanwhile), the second while the first batch is matched.
    free = queue.Queue()

    checked_out = []

    def buffer_worker():
        for k in range(2 if n_batches > 1 else 1):
            try:
                b = _list_buffers.acquire(max_cap)
                checked_out.append(b)
                free.put(b[:max_cap])
            except Exception as e:      # raised by the main thread when it takes the item
                free.put(e)

    buf_thread = threading.Thread(target=buffer_worker, daemon=True)
    buf_thread.start()
    work = queue.Queue()
    tracks_busy = [0.0]
    err = []

    def merge_worker():
        while True:
            item = work.get()
            if item is None:
                return
            sub, ra, corr_buf = item
            t1 = time.perf_counter()
            try:
                builder.feed(sub, ra, corr_buf)
            except Exception as e:      # reported by the main thread
                err.append(e)
            tracks_busy[0] += time.perf_counter() - t1
            free.put(corr_buf)

    th = threading.Thread(target=merge_worker, daemon=True)
    th.start()
    status = np.zeros(pf.shape[0], dtype=np.int32)
    for k in range(n_batches):
        sub = pf[bounds[k]:bounds[k + 1]]
        t1 = time.perf_counter()
        # Cascade hashing hashes every descriptor against the average over ALL views (CascadeHashing::init,
        # cascade_hashing.cc:33-70): its first batch needs the complete bank, not just the views it names --
        # the library refuses a cascade batch while a view is missing.  Exhaustive matching takes what it names.
        wait_for_views(V if matcher != "exhaustive" else (int(sub.max()) + 1 if sub.size else 0))
        upload_wait += time.perf_counter() - t1
        t1 = time.perf_counter()
        buf = free.get()
        if isinstance(buf, Exception):
            raise buf
        if k < 2:
            tm.setup_s += time.perf_counter() - t1        # waiting for page-locked memory (later waits: for the merge)
        m.use_result_buffer(buf)
        ra, corr_buf = m.compute_arrays(sub, capacity=max_cap)
        status[bounds[k]:bounds[k + 1]] = ra["status"]
        work.put((sub, ra.copy(), buf))
    wait_for_views(V)
    up_thread.join()
    # upload_s: what the job waited for views (before the first batch and between batches); the rest of
    # the uploads ran beside the matching
    tm.upload_s = upload_wait
    tm.matching_s = time.perf_counter() - t0 - upload_wait
    t0 = time.perf_counter()
    work.put(None)
    th.join()
    if err:
        raise err[0]
    # every batch is merged and the matcher is idle: the list buffers go back to the pool (on an error they do not --
    # a copy into one of them may still be under way)
    buf_thread.join()
    for b in checked_out:
        _list_buffers.release(b)
    ids, toff, tfeat, tcol, summary = builder.finish(want_track_ids=False)      # Viewport::track_ids are not used downstream
    # tracks_s: what the job waited for after the last batch was matched (the merge of the earlier
    # batches ran beside the matching: tracks_busy_s of it in all)
    tm.tracks_s = time.perf_counter() - t0
    tm.tracks_busy_s = tracks_busy[0]
    t0 = time.perf_counter()
    tt = TrackTable.from_mve(toff, tfeat, norm, W, V)
    tm.convert_s = time.perf_counter() - t0
    info = {"num_pairs": int(pf.shape[0]), "matched_pairs": int(builder.num_pairs), "correspondences": int(builder.num_matches),
            "num_mve_tracks": int(summary.num_tracks), "invalid_mve_tracks": int(summary.num_invalid_tracks),
            "pair_status": status}
    try:
        st = m.stats()
        info["match_stats"] = {f[0]: getattr(st, f[0]) for f in st._fields_ if not f[0].startswith("reserved")}
    except Exception:
        pass
    builder.close()
    # the matcher's device memory (the bank, 16 GB of partial-result scratch) goes back on a thread of its
    # own: freeing it takes 0.1 s that the pose estimation need not wait for
    closer = threading.Thread(target=m.close, daemon=True)
    closer.start()
    _background.append(closer)
    return tt, info


_background = []


def join_background():
    """Wait for what earlier jobs left running beside the caller (the release of a matcher's device memory)."""
    while _background:
        _background.pop().join()


def run_pose_estimation(tt: TrackTable, iset, model, device=0, rot_perturb_deg=2.0, off_perturb=0.01,
                        seed=7, timings=None, capture=(), max_groups=None, verbose=False, view_ids=None,
                        euler_dof=4, check_incremental=False, use_scene=True):
    """runPoseEstimation (reconstruct.cpp:174-295) on the track table.  use_scene: the table lives on the device
    for the whole loop (osfm_scene_*: every step selects its observations there); False: every step flattens its
    tracks on the host and goes through the per-call entries of the C ABI (what a caller without the scene does;
    the two forms produce the same cameras, flags and points to the bit: tests/test_e2e_gpu.py)."""
    tm = timings if timings is not None else Timings()
    V = iset.num_views
    W, H = iset.width, iset.height
    gt, _ = canonical_ground_truth(iset, model)
    rng = np.random.default_rng(seed)
    t_pose = time.perf_counter()
    full_table = tt          # the caller's table; `tt` becomes a compacted working copy once the filters have thinned it

    # ---- buildGroups ----------------------------------------------------------
    t0 = time.perf_counter()
    fmask = tt.alive_f & tt.alive_t[tt.track_of]
    offs = np.concatenate([[0], np.cumsum(tt.alive_lengths()[tt.alive_t])]).astype(np.int64)
    views_flat = tt.view[fmask]
    vids = np.arange(V, dtype=np.int32) if view_ids is None else np.asarray(view_ids, np.int32)
    groups = G.build_groups_flat(vids, offs, views_flat, 3, device)
    tm.groups_s = time.perf_counter() - t0
    if max_groups is not None:
        groups = groups[:max_groups]

    cams = np.zeros((V, 7))
    const = np.zeros((V, 7), dtype=np.uint8)
    aligned = []                       # view ids in the order they joined (alignedCameras)
    is_aligned = np.zeros(V, dtype=bool)
    calls, captured = [], {}
    opt = B.default_options(device=device)
    opt_local = B.default_options(device=device, retriangulate_points=1)
    scene = None
    if use_scene and not capture:
        from .scene import Scene
        t0 = time.perf_counter()
        scene = Scene(model, np.full(V, W, np.int32), np.full(V, H, np.int32), tt.offsets, tt.view, tt.xy.astype(np.float32), device)
        if not (tt.alive_t.all() and tt.alive_f.all()):
            scene.set_flags(tt.alive_t, tt.alive_f)
        tm.pose_host_s += time.perf_counter() - t0

    def start_pose(v):
        p = gt[v].copy()
        if v == groups[0].ids[0] and not aligned:
            return p                                     # the first camera: canonical, fixed for the whole run
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        a = np.deg2rad(rot_perturb_deg)
        if model == B.MODEL_QUATERNION:
            dq = np.array([*(np.sin(a / 2) * axis), np.cos(a / 2)])
            p[:4] = synth.quat_mul(dq, p[:4])
            p[4:6] += off_perturb * rng.normal(size=2)
        else:
            p[:3] += a * axis / np.sqrt(3.0)
            p[3:5] += off_perturb * rng.normal(size=2)
        return p

    def solve(kind, prob, options):
        t0 = time.perf_counter()
        if len(calls) in capture:
            captured[len(calls)] = (kind, B.FlatProblem(prob.model, prob.cam_params, prob.cam_const, prob.img_w, prob.img_h,
                                                        prob.points, prob.obs_xy, prob.obs_camera, prob.obs_point),
                                    int(options.retriangulate_points))
        s = B.solve(prob, options)
        dt = time.perf_counter() - t0
        calls.append(BaCall(kind, prob.cam_params.shape[0], prob.points.shape[0], prob.obs_camera.shape[0],
                            int(s.num_iterations), dt * 1e3, s.lm_loop_ms))
        return s, dt

    # A track's intersection depends on its alive features under the aligned cameras and on
    # those cameras alone.  While the cameras already aligned stay as they are (every step but
    # a global adjustment, which also overwrites the points), adding views changes only the
    # tracks those views see: the full pass of the reference then recomputes everything else
    # to the same bits, and is skipped here.
    tri = {"full": True}

    def _triangulate(tracks_subset):
        # the observation arrays in one pass over the table (osfm_tracks_select_observations): the live
        # features of the selected tracks whose view has a camera, the tracks numbered as they appear
        mask = None
        if tracks_subset is not None:
            mask = np.zeros(tt.alive_t.shape[0], dtype=np.uint8)
            mask[tracks_subset] = 1
        obs_xy, obs_cam, obs_pt, uniq, _ = T.select_observations(tt.track_of, tt.cam_f, tt.live_f, tt.xy, track_mask=mask, track_offsets=tt.offsets)
        uniq = uniq.astype(np.int64)                    # a copy: the selector's buffers are reused
        start = np.zeros((uniq.size, 4))
        start[:, 3] = 1.0
        prob = _problem(model, cams[aligned], const[aligned], W, H, start, obs_xy, obs_cam, obs_pt)
        valid = (B.triangulate(prob, device) if uniq.size else np.zeros(0, np.uint8)).astype(bool)
        return uniq, valid, prob.points

    def triangulate_all(new_views=None):
        """algorithm->triangulateTracks(alignedCameras, tracks, true): every track with two
        or more rays under the aligned cameras gets its intersection, the others lose their
        point (triangulation.cpp:76-91)."""
        t0 = time.perf_counter()
        subset = None
        if not tri["full"] and new_views is not None:
            idx = tt.features_of_views(new_views)
            subset = np.unique(tt.track_of[idx]).astype(np.int64)
        uniq, valid, pts = _triangulate(subset)
        if subset is None:
            tt.has_point[:] = False
        else:
            tt.has_point[subset] = False
        ok = uniq[valid]
        tt.has_point[ok] = True
        tt.point[ok] = pts[valid]
        if check_incremental and subset is not None:
            hp, pt = tt.has_point.copy(), tt.point.copy()
            uniq, valid, pts = _triangulate(None)
            full_hp = np.zeros_like(hp)
            full_hp[uniq[valid]] = True
            assert np.array_equal(full_hp, hp) and np.array_equal(pts[valid], pt[uniq[valid]]), "incremental triangulation"
        tri["full"] = False
        tm.triangulate_s += time.perf_counter() - t0

    def reprojection_filter(view_list, cam_p, cam_c, permanent):
        """filterTracksWithReprojectionError: tracks seen by ALL the cameras are
        re-triangulated from them and lose the features that reproject 1.5 px or more
        away; a track left with fewer than two features goes.  Returns the features (in
        track order) that the tracks seen by at least two of the cameras keep inside the
        camera set -- what the bundle adjustment that follows works on."""
        t0 = time.perf_counter()
        n = len(view_list)
        idx = tt.features_of_views(view_list)
        if idx.size == 0:
            return idx
        tr = tt.track_of[idx]
        uniq, first, cnt, run = _runs(tr)
        cnt_f = cnt[run]
        keep_f = np.ones(idx.shape[0], dtype=bool)
        full_f = cnt_f == n
        if full_f.any():
            fi = idx[full_f]
            fu = uniq[cnt == n]
            cam_of = np.full(V, -1, dtype=np.int32)
            cam_of[view_list] = np.arange(n, dtype=np.int32)
            prob = _problem(model, cam_p, cam_c, W, H, tt.point[fu].copy(), tt.xy[fi], cam_of[tt.view[fi]],
                            np.repeat(np.arange(fu.shape[0], dtype=np.int32), n))
            st = prob.struct()
            ok = np.zeros(fi.shape[0], dtype=np.uint8)
            capi.check(capi.lib.osfm_filter_reprojection(C.byref(st), device, C.c_double(MAX_REPROJECTION_ERROR),
                                                         capi._ptr(ok, C.c_uint8), None, None))
            keep_f[full_f] = ok.astype(bool)
        # features a track keeps outside the camera set count towards "more than one left"
        total = tt.alive_lengths()[uniq] - cnt + np.add.reduceat(keep_f.astype(np.int64), first)
        track_ok = np.where(cnt == n, total > 1, True)              # only full-size tracks are judged (:147-149,187-189)
        if permanent:
            tt.kill(tracks=uniq[~track_ok], features=idx[~keep_f])
        inside = np.add.reduceat(keep_f.astype(np.int64), first)
        sel = keep_f & (track_ok & (inside > 1))[run]
        tm.local_filter_s += time.perf_counter() - t0
        return idx[sel]

    if scene is not None:
        # ---- the loop on the device-resident scene ----------------------------------------------------------
        tri_full = True
        processed = 0
        for g in groups:
            processed += 1
            ids = list(g.ids)
            first_group = not aligned
            lp = np.array([cams[v].copy() if is_aligned[v] else start_pose(v) for v in ids])
            lc = np.array([default_const_mask(model, euler_dof=euler_dof) for _ in ids])
            if first_group:
                lc[0] = default_const_mask(model, fixed=True, euler_dof=euler_dof)
            t0 = time.perf_counter()
            s, M, O = scene.local_adjustment(ids, lp, lc, MAX_REPROJECTION_ERROR, opt_local)
            dt = time.perf_counter() - t0
            tm.local_ba_s += dt
            calls.append(BaCall("local", len(ids), M, O, int(s.num_iterations), dt * 1e3, s.lm_loop_ms))
            new_views = []
            if first_group:
                for k, v in enumerate(ids):
                    cams[v] = lp[k]; const[v] = lc[k]
                    aligned.append(v); is_aligned[v] = True; new_views.append(v)
                scene.align_views(new_views, cams[new_views], const[new_views])
            else:
                align_to_global(model, lp, [cams[v] if is_aligned[v] else None for v in ids])
                for k, v in enumerate(ids):
                    if not is_aligned[v]:
                        cams[v] = lp[k]; const[v] = default_const_mask(model, euler_dof=euler_dof)
                        aligned.append(v); is_aligned[v] = True; new_views.append(v)
                scene.align_views(new_views, cams[new_views], const[new_views])
            t0 = time.perf_counter()
            bad = scene.triangulate(None if tri_full else new_views, check_full=check_incremental)
            assert bad == 0, "incremental triangulation"
            tri_full = False
            tm.triangulate_s += time.perf_counter() - t0
            if not first_group and processed % GLOBAL_BA_INTERVAL == 0:
                t0 = time.perf_counter()
                s, M, O = scene.global_adjustment(opt)
                dt = time.perf_counter() - t0
                tm.global_ba_s += dt
                calls.append(BaCall("global", len(aligned), M, O, int(s.num_iterations), dt * 1e3, s.lm_loop_ms))
                cams[aligned] = scene.cameras()[1]
                tri_full = True
                t0 = time.perf_counter()
                scene.filter_outliers()
                tm.outlier_filter_s += time.perf_counter() - t0
                t0 = time.perf_counter()
                scene.filter_reprojection(MAX_REPROJECTION_ERROR)      # no track seen by every camera: nothing is judged
                tm.local_filter_s += time.perf_counter() - t0
            if verbose:
                print(f"group {processed}/{len(groups)} {ids}: {len(aligned)} cameras")
        t0 = time.perf_counter()
        s, M, O = scene.global_adjustment(opt)
        dt = time.perf_counter() - t0
        tm.global_ba_s += dt
        calls.append(BaCall("final", len(aligned), M, O, int(s.num_iterations), dt * 1e3, s.lm_loop_ms))
        cams[aligned] = scene.cameras()[1]
        # the scene's state into the caller's table
        t0 = time.perf_counter()
        at, af, hp, pt = scene.download()
        scene.close()
        tt.alive_t[:] = at
        tt.alive_f[:] = af
        tt.live_f[:] = af & at[tt.track_of]
        tt.has_point[:] = hp
        tt.point[:] = pt
        tt._lengths[:] = np.add.reduceat(tt.live_f.astype(np.int64), tt.offsets[:-1]) if tt.live_f.size else 0
        tt._lengths[np.diff(tt.offsets) == 0] = 0
        for k, v in enumerate(aligned):
            tt.align_view(v, k)
        tm.pose_host_s += time.perf_counter() - t0
        tm.pose_s = time.perf_counter() - t_pose
        return cams, aligned, groups, calls, captured

    processed = 0
    for g in groups:
        processed += 1
        ids = list(g.ids)
        first_group = not aligned
        # ---- calculateInitialAlignment (stand-in, see the module docstring) -------
        lp = np.array([cams[v].copy() if is_aligned[v] else start_pose(v) for v in ids])
        lc = np.array([default_const_mask(model, euler_dof=euler_dof) for _ in ids])
        if first_group:
            lc[0] = default_const_mask(model, fixed=True, euler_dof=euler_dof)          # localCameras[0]->setFixed(true), :215
        # ---- filterTracksWithReprojectionError(localTracks, localCameras) + local BA on a
        # filtered, re-triangulated copy (runBundleAdjustment(..., true, true), :212-219) ----
        fsel = reprojection_filter(ids, lp, lc, permanent=False)
        t0 = time.perf_counter()
        uniq, _, _, run = _runs(tt.track_of[fsel])
        cam_of = np.full(V, -1, dtype=np.int32)
        cam_of[ids] = np.arange(len(ids), dtype=np.int32)
        prob = _problem(model, lp, lc, W, H, np.tile([0.0, 0, 0, 1], (uniq.size, 1)), tt.xy[fsel],
                        cam_of[tt.view[fsel]], run.astype(np.int32))
        tm.pose_host_s += time.perf_counter() - t0
        s, dt = solve("local", prob, opt_local)
        tm.local_ba_s += dt
        lp = prob.cam_params
        if first_group:
            # normalizeScene is the identity here: camera 0 is canonical and fixed
            for k, v in enumerate(ids):
                cams[v] = lp[k]; const[v] = lc[k]
                aligned.append(v); is_aligned[v] = True; tt.align_view(v, len(aligned) - 1)
            triangulate_all()
        else:
            align_to_global(model, lp, [cams[v] if is_aligned[v] else None for v in ids])
            new_views = []
            for k, v in enumerate(ids):                             # mergeIntoGlobal: only the new cameras
                if not is_aligned[v]:
                    cams[v] = lp[k]; const[v] = default_const_mask(model, euler_dof=euler_dof)
                    aligned.append(v); is_aligned[v] = True; new_views.append(v); tt.align_view(v, len(aligned) - 1)
            triangulate_all(new_views)
            if processed % GLOBAL_BA_INTERVAL == 0:
                _global_ba(tt, model, cams, const, aligned, W, H, V, solve, "global", opt, tm)
                tri["full"] = True                      # old cameras moved, points overwritten
                # filterOutlierTracks + filterTracksWithReprojectionError (:264-265)
                t0 = time.perf_counter()
                at = np.nonzero(tt.alive_t)[0]
                keep, _ = _outlier_flags(tt.point[at], tt.has_point[at], device)
                tt.kill(tracks=at[~keep])
                tm.outlier_filter_s += time.perf_counter() - t0
                if len(aligned) <= int(tt.alive_lengths().max(initial=0)):
                    reprojection_filter(list(aligned), cams[aligned], const[aligned], permanent=True)
                # drop what the filters have cleared once it is most of the table
                t0 = time.perf_counter()
                if int(tt.live_f.sum()) < 0.6 * tt.live_f.shape[0]:
                    tt = tt.compacted()
                tm.pose_host_s += time.perf_counter() - t0
        if verbose:
            print(f"group {processed}/{len(groups)} {ids}: {len(aligned)} cameras, {tt.num_tracks} tracks, "
                  f"{int(tt.has_point.sum())} points")
    _global_ba(tt, model, cams, const, aligned, W, H, V, solve, "final", opt, tm)      # :281
    full_table.write_back(tt)
    tm.pose_s = time.perf_counter() - t_pose
    return cams, aligned, groups, calls, captured


def _outlier_flags(points, has_point, device):
    from . import filters
    return filters.outlier_track_flags(points, has_point, device)


def _global_ba(tt, model, cams, const, aligned, W, H, V, solve, kind, opt, tm):
    """runBundleAdjustment(alignedCameras, tracks, algorithm, true, false): every track
    with a point is a parameter block, every feature of such a track whose view has a
    camera a residual (bundle_adjustment.cpp:86-123); cameras and points updated in place."""
    t0 = time.perf_counter()
    with_point = tt.alive_t & tt.has_point
    tsel = np.flatnonzero(with_point)
    slot = (np.cumsum(with_point) - 1).astype(np.int32)                  # track -> row of tsel
    obs_xy, obs_cam, obs_pt, _, _ = T.select_observations(tt.track_of, tt.cam_f, tt.live_f, tt.xy,
                                                          track_mask=with_point, track_slot=slot, track_offsets=tt.offsets)
    prob = _problem(model, cams[aligned], const[aligned], W, H, tt.point[tsel], obs_xy, obs_cam, obs_pt)
    tm.pose_host_s += time.perf_counter() - t0
    s, dt = solve(kind, prob, opt)
    tm.global_ba_s += dt
    t0 = time.perf_counter()
    cams[aligned] = prob.cam_params
    tt.point[tsel] = prob.points
    tm.pose_host_s += time.perf_counter() - t0
    return s


def reconstruct(iset, solver=0, matcher="exhaustive", device=0, verify=True, rot_perturb_deg=2.0,
                off_perturb=0.01, seed=7, capture=(), max_groups=None, verbose=False,
                check_incremental=False, use_scene=True) -> Result:
    """orthosfm::reconstruct from the views' descriptors on: one wall clock over matching,
    track building, group ordering and the incremental pose estimation.
    solver 0: quaternion cameras (ORTHO_QUATERNION); 1..3: Euler cameras with that many
    free angle blocks (3 = ORTHO_EULER_ALL_DOF: phi, theta, roll and the offsets)."""
    tm = Timings()
    t_all = time.perf_counter()
    tt, info = match_and_build_tracks(iset, matcher, device, verify, tm)
    model = B.MODEL_QUATERNION if solver == 0 else B.MODEL_EULER
    cams, aligned, groups, calls, captured = run_pose_estimation(
        tt, iset, model, device, rot_perturb_deg, off_perturb, seed, tm, capture, max_groups, verbose,
        euler_dof=euler_dof_of_solver(solver), check_incremental=check_incremental, use_scene=use_scene)
    join_background()            # inside the clock: the matcher's memory is back when the job is done
    tm.total_s = time.perf_counter() - t_all
    info.pop("match_stats", None)
    return Result(cams, aligned, tt, groups, tm, calls, captured=captured, **info)


CAMERA_DISTANCE = 10.0                  # OrthoQuaternionCamera.cpp:70, OrthographicCamera.h:119


def camera_to_world_matrices(model, cam_params):
    """The 4x4 matrices exportCamerasToFile writes (camera_io.cpp:24-29):
    columns x axis, y axis, z axis, origin = R (0, 0, -10)."""
    out = np.zeros((len(cam_params), 4, 4))
    for i, p in enumerate(cam_params):
        R = _cam_rotation(model, p)
        out[i, :3, :3] = R
        out[i, :3, 3] = R @ np.array([0.0, 0.0, -CAMERA_DISTANCE])
        out[i, 3, 3] = 1.0
    return out


def save_project(res: Result, model, folder, image_names=None):
    """The files orthosfm::reconstruct leaves in the project folder (reconstruct.cpp:125,
    :160, :168, :290), written through the C ABI of the text formats: tracks.txt (all
    tracks, as built), cameras.txt (aligned cameras), sparse_cloud.ply (surviving tracks
    with a point), time_measurements.txt."""
    from . import formats as F
    os.makedirs(folder, exist_ok=True)
    tt = res.tracks
    feats = np.zeros(tt.view.shape[0], dtype=capi.TRACK_FEATURE)
    feats["view_id"] = tt.view
    feats["local_feature_id"] = tt.feat
    feats["global_feature_id"] = 32768 * tt.view.astype(np.int64) + tt.feat
    feats["x"] = tt.xy[:, 0]
    feats["y"] = tt.xy[:, 1]
    F.save_tracks_to_file_native(tt.offsets, feats, os.path.join(folder, "tracks.txt"))
    names = image_names or ["view_%04d" % v for v in range(tt.num_views)]
    al = list(res.aligned_views)
    F.export_cameras_to_file_native([names[v] for v in al], camera_to_world_matrices(model, res.cam_params[al]),
                                    os.path.join(folder, "cameras.txt"))
    F.save_points_to_ply_native(os.path.join(folder, "sparse_cloud.ply"), tt.offsets, feats, tt.point,
                                tt.has_point & tt.alive_t)
    tm = res.timings
    F.save_runtimes_to_txt_native(os.path.join(folder, "time_measurements.txt"), tm.upload_s,
                                  tm.matching_s + tm.tracks_s + tm.convert_s, tm.groups_s + tm.pose_s, tm.total_s)
