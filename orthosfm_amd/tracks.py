"""Host-side mirror of sfm::bundler::Tracks (src/mve/sfm/bundler_tracks.h:23-66,
bundler_tracks.cc:49-203): compute(matching, viewports) -> tracks, with the
per-feature track ids written back to the viewports.  The work is done by
osfm_tracks_compute behind the C ABI."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import threading

import numpy as np

from . import capi


@dataclass
class Viewport:
    """The part of sfm::bundler::Viewport (bundler_common.h:37-59) that track
    building touches: the feature count, FeatureSet::colors and track_ids."""
    num_features: int
    colors: np.ndarray | None = None            # (n, 3) uint8
    track_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))


@dataclass
class MveTrack:
    """sfm::bundler::Track (bundler_common.h:79-88) without the 3-D position."""
    features: np.ndarray                         # (k, 2) int32: view_id, feature_id
    color: np.ndarray                            # (3,) uint8


def flatten_matching(matching):
    """matching: sequence of objects with view_1_id, view_2_id, matches ((k, 2)
    int32) -- e.g. what HipExhaustiveMatching.compute returns.  Pairs without
    matches are kept (they are no-ops for the merge)."""
    n = len(matching)
    pairs = (capi.Pair * max(n, 1))()
    offsets = np.zeros(n + 1, dtype=np.int64)
    chunks = []
    for i, tvm in enumerate(matching):
        pairs[i].view_1, pairs[i].view_2 = int(tvm.view_1_id), int(tvm.view_2_id)
        m = np.ascontiguousarray(tvm.matches, dtype=np.int32).reshape(-1, 2)
        chunks.append(m)
        offsets[i + 1] = offsets[i] + m.shape[0]
    corr = np.concatenate(chunks) if chunks else np.zeros((0, 2), np.int32)
    return pairs, offsets, np.ascontiguousarray(corr, dtype=np.int32)


def compute_flat(view_sizes, colors, pairs, pair_offsets, corr):
    """Thin wrapper of osfm_tracks_compute; returns (track_ids, track_offsets,
    track_features, track_colors, summary)."""
    view_sizes = np.ascontiguousarray(view_sizes, dtype=np.int32)
    total = int(view_sizes.sum())
    n_matches = int(pair_offsets[-1]) if len(pair_offsets) else 0
    track_ids = np.full(max(total, 1), -1, dtype=np.int32)
    tcap, fcap = max(n_matches, 1), max(2 * n_matches, 1)
    track_offsets = np.zeros(tcap + 1, dtype=np.int64)
    track_features = np.zeros((fcap, 2), dtype=np.int32)
    track_colors = np.zeros((tcap, 3), dtype=np.uint8)
    summary = capi.TracksSummary()
    col_ptr = None
    if colors is not None:
        colors = np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 3)
        assert colors.shape[0] == total
        col_ptr = capi._ptr(colors, C.c_uint8)
    pair_offsets = np.ascontiguousarray(pair_offsets, dtype=np.int64)
    capi.check(capi.lib.osfm_tracks_compute(
        len(view_sizes), capi._ptr(view_sizes, C.c_int32), col_ptr, len(pair_offsets) - 1, pairs,
        capi._ptr(pair_offsets, C.c_int64), capi._ptr(corr, C.c_int32), capi._ptr(track_ids, C.c_int32),
        C.c_int64(tcap), C.c_int64(fcap), capi._ptr(track_offsets, C.c_int64),
        capi._ptr(track_features, C.c_int32), capi._ptr(track_colors, C.c_uint8), C.byref(summary)))
    nt = summary.num_tracks
    return (track_ids[:total], track_offsets[:nt + 1], track_features[:summary.num_features],
            track_colors[:nt], summary)


def compute_flat_ranges(view_sizes, colors, pairs, pair_starts, pair_counts, corr):
    """osfm_tracks_compute_ranges: like compute_flat, but pair p owns
    corr[pair_starts[p] : pair_starts[p] + pair_counts[p]] -- the lists need not
    be packed (distributed.SharedMatchStore hands them over in place)."""
    view_sizes = np.ascontiguousarray(view_sizes, dtype=np.int32)
    total = int(view_sizes.sum())
    pair_starts = np.ascontiguousarray(pair_starts, dtype=np.int64)
    pair_counts = np.ascontiguousarray(pair_counts, dtype=np.int64)
    assert pair_starts.shape == pair_counts.shape
    n_matches = int(pair_counts.sum())
    track_ids = np.full(max(total, 1), -1, dtype=np.int32)
    tcap, fcap = max(n_matches, 1), max(2 * n_matches, 1)
    track_offsets = np.zeros(tcap + 1, dtype=np.int64)
    track_features = np.zeros((fcap, 2), dtype=np.int32)
    track_colors = np.zeros((tcap, 3), dtype=np.uint8)
    summary = capi.TracksSummary()
    col_ptr = None
    if colors is not None:
        colors = np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 3)
        assert colors.shape[0] == total
        col_ptr = capi._ptr(colors, C.c_uint8)
    assert corr.dtype == np.int32 and corr.flags.c_contiguous
    capi.check(capi.lib.osfm_tracks_compute_ranges(
        len(view_sizes), capi._ptr(view_sizes, C.c_int32), col_ptr, len(pair_starts), pairs,
        capi._ptr(pair_starts, C.c_int64), capi._ptr(pair_counts, C.c_int64), capi._ptr(corr, C.c_int32),
        capi._ptr(track_ids, C.c_int32), C.c_int64(tcap), C.c_int64(fcap),
        capi._ptr(track_offsets, C.c_int64), capi._ptr(track_features, C.c_int32),
        capi._ptr(track_colors, C.c_uint8), C.byref(summary)))
    nt = summary.num_tracks
    return (track_ids[:total], track_offsets[:nt + 1], track_features[:summary.num_features],
            track_colors[:nt], summary)


def feature_table(track_offsets, track_features, norm_positions, image_width, want_order=True):
    """osfm_tracks_feature_table: (view, feat, xy, track_of, by_view, view_start) of the features of the tracks;
    norm_positions[v] = view v's float32 [n_v, 2] normalised positions."""
    offs = np.ascontiguousarray(track_offsets, dtype=np.int64)
    tf = np.ascontiguousarray(track_features, dtype=np.int32).reshape(-1, 2)
    V = len(norm_positions)
    pos = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 2) for p in norm_positions]
    sizes = np.array([p.shape[0] for p in pos], dtype=np.int32)
    ptrs = (C.c_void_p * max(V, 1))(*[p.ctypes.data for p in pos])
    nt = offs.shape[0] - 1
    nf = int(offs[-1]) if offs.size else 0
    view = np.empty(nf, dtype=np.int32)
    feat = np.empty(nf, dtype=np.int32)
    xy = np.empty((nf, 2), dtype=np.float64)
    track_of = np.empty(nf, dtype=np.int32)
    by_view = np.empty(nf, dtype=np.int64) if want_order else None
    view_start = np.empty(V + 1, dtype=np.int64) if want_order else None
    capi.check(capi.lib.osfm_tracks_feature_table(
        C.c_int64(nt), capi._ptr(offs, C.c_int64), capi._ptr(tf, C.c_int32), C.c_int32(V), capi._ptr(sizes, C.c_int32),
        C.cast(ptrs, C.POINTER(C.POINTER(C.c_float))), C.c_double(float(image_width)),
        capi._ptr(view, C.c_int32), capi._ptr(feat, C.c_int32), capi._ptr(xy, C.c_double), capi._ptr(track_of, C.c_int32),
        capi._ptr(by_view, C.c_int64) if want_order else None, capi._ptr(view_start, C.c_int64) if want_order else None))
    return view, feat, xy, track_of, by_view, view_start


_SELECT_BUFFERS = threading.local()      # per thread: the slices a call returns are valid until the SAME thread's next call


def select_observations(track_of, cam_f, live, xy, track_mask=None, track_slot=None, want_features=False, track_offsets=None):
    """osfm_tracks_select_observations: (obs_xy, obs_camera, obs_point, tracks, feature_ids) of the live
    features whose view has a camera, optionally restricted to the tracks of a mask; obs_point numbers the
    tracks by track_slot when given, else densely in order of appearance (tracks = their ids)."""
    n = int(track_of.shape[0])
    # output buffers are kept between calls (fresh arrays of the table's size cost more in page faults than
    # the pass itself); the slices returned are overwritten by the next call -- callers copy what they keep
    # (FlatProblem does)
    cap = n
    if not hasattr(_SELECT_BUFFERS, "buf"):
        _SELECT_BUFFERS.buf = {}
    buf = _SELECT_BUFFERS.buf
    if buf.get("cap", -1) < cap:
        buf["cap"] = cap
        buf["xy"] = np.empty((cap, 2), dtype=np.float64)
        buf["cam"] = np.empty(cap, dtype=np.int32)
        buf["pt"] = np.empty(cap, dtype=np.int32)
        buf["tracks"] = np.empty(cap, dtype=np.int32)
        buf["fids"] = np.empty(cap, dtype=np.int32)
    obs_xy, obs_cam, obs_pt = buf["xy"], buf["cam"], buf["pt"]
    tracks = buf["tracks"] if track_slot is None else None
    fids = buf["fids"] if want_features else None
    nobs, nt = C.c_int64(), C.c_int64()
    live8 = live.view(np.uint8) if live.dtype == np.bool_ else live
    mask8 = None if track_mask is None else (track_mask.view(np.uint8) if track_mask.dtype == np.bool_ else track_mask)
    capi.check(capi.lib.osfm_tracks_select_observations(
        C.c_int64(n), capi._ptr(track_of, C.c_int32),
        capi._ptr(track_offsets, C.c_int64) if track_offsets is not None else None,
        capi._ptr(cam_f, C.c_int32), capi._ptr(live8, C.c_uint8),
        capi._ptr(mask8, C.c_uint8) if mask8 is not None else None,
        capi._ptr(track_slot, C.c_int32) if track_slot is not None else None,
        capi._ptr(xy, C.c_double), C.c_int64(cap), capi._ptr(fids, C.c_int32) if fids is not None else None,
        capi._ptr(obs_xy, C.c_double), capi._ptr(obs_cam, C.c_int32), capi._ptr(obs_pt, C.c_int32),
        capi._ptr(tracks, C.c_int32) if tracks is not None else None, C.byref(nobs), C.byref(nt)))
    k, t = int(nobs.value), int(nt.value)
    return (obs_xy[:k], obs_cam[:k], obs_pt[:k], None if tracks is None else tracks[:t],
            None if fids is None else fids[:k])


class TracksBuilder:
    """Tracks::compute fed pair batch by pair batch (osfm_tracks_builder_*): feed() takes what
    HipExhaustiveMatching.compute_arrays returned for one batch of pairs -- in the reference's pair
    order over the batches -- so the merge of one batch runs on the host while the device matches
    the next (the C calls release the interpreter lock)."""

    def __init__(self, view_sizes):
        self.view_sizes = np.ascontiguousarray(view_sizes, dtype=np.int32)
        self._h = C.c_void_p()
        capi.check(capi.lib.osfm_tracks_builder_create(len(self.view_sizes), capi._ptr(self.view_sizes, C.c_int32),
                                                       C.byref(self._h)))
        self.num_matches = 0
        self.num_pairs = 0

    def feed(self, pairs_flat, records, corr):
        """pairs_flat (n, 2) int32, records: osfm_pair_result records of the batch (numpy), corr: the
        (rows, 2) int32 list buffer the offsets of the records point into."""
        matched = records["status"] == capi.PAIR_MATCHED
        cnt = np.where(records["num_inliers"] >= 0, records["num_inliers"], records["num_matches"]).astype(np.int64)
        cnt = np.where(matched, cnt, 0)
        starts = np.ascontiguousarray(np.where(matched, records["offset"], 0), dtype=np.int64)
        cnt = np.ascontiguousarray(cnt, dtype=np.int64)
        pf = np.ascontiguousarray(pairs_flat, dtype=np.int32).reshape(-1, 2)
        assert corr.dtype == np.int32 and corr.flags.c_contiguous
        capi.check(capi.lib.osfm_tracks_builder_feed(self._h, pf.shape[0], pf.ctypes.data_as(C.POINTER(capi.Pair)),
                                                     capi._ptr(starts, C.c_int64), capi._ptr(cnt, C.c_int64),
                                                     capi._ptr(corr, C.c_int32)))
        self.num_matches += int(cnt.sum())
        self.num_pairs += int(matched.sum())

    def finish(self, colors=None, want_track_ids=True):
        total = int(self.view_sizes.sum())
        track_ids = np.full(max(total, 1), -1, dtype=np.int32) if want_track_ids else None
        tcap, fcap = max(self.num_matches, 1), max(2 * self.num_matches, 1)
        # a track has at least two features and every feature is in one track at most
        tcap, fcap = min(tcap, max(total // 2, 1)), min(fcap, max(total, 1))
        track_offsets = np.zeros(tcap + 1, dtype=np.int64)
        track_features = np.zeros((fcap, 2), dtype=np.int32)
        track_colors = np.zeros((tcap, 3), dtype=np.uint8)
        summary = capi.TracksSummary()
        col_ptr = None
        if colors is not None:
            colors = np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 3)
            col_ptr = capi._ptr(colors, C.c_uint8)
        capi.check(capi.lib.osfm_tracks_builder_finish(
            self._h, col_ptr, capi._ptr(track_ids, C.c_int32) if track_ids is not None else None, C.c_int64(tcap), C.c_int64(fcap),
            capi._ptr(track_offsets, C.c_int64), capi._ptr(track_features, C.c_int32),
            capi._ptr(track_colors, C.c_uint8), C.byref(summary)))
        nt = summary.num_tracks
        return (track_ids[:total] if track_ids is not None else None, track_offsets[:nt + 1],
                track_features[:summary.num_features], track_colors[:nt], summary)

    def close(self):
        if self._h:
            capi.lib.osfm_tracks_builder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tracks:
    """sfm::bundler::Tracks."""

    def __init__(self, verbose_output: bool = False):
        self.verbose_output = verbose_output

    def compute(self, matching, viewports):
        """Tracks::compute(matching, &viewports, &tracks): returns the track
        list; viewports[i].track_ids is (re)written."""
        sizes = [vp.num_features for vp in viewports]
        colors = None
        if any(vp.colors is not None for vp in viewports):
            colors = np.concatenate([
                np.zeros((vp.num_features, 3), np.uint8) if vp.colors is None
                else np.asarray(vp.colors, dtype=np.uint8).reshape(-1, 3) for vp in viewports])
        pairs, offsets, corr = flatten_matching(matching)
        if self.verbose_output:
            print("Propagating track IDs...")
        ids, toff, tfeat, tcol, summary = compute_flat(sizes, colors, pairs, offsets, corr)
        if self.verbose_output:
            print(f"Removing tracks with conflicts... deleted {summary.num_invalid_tracks} tracks.")
            print("Colorizing tracks...")
        start = 0
        for vp in viewports:
            vp.track_ids = ids[start:start + vp.num_features].copy()
            start += vp.num_features
        return [MveTrack(tfeat[toff[t]:toff[t + 1]], tcol[t]) for t in range(summary.num_tracks)]
