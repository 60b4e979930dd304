"""Host-side mirror of the two outlier filters that bracket the bundle
adjustment calls in the reference pipeline (src/sfm/reconstruct.cpp:212,264-265):

    filter_outlier_tracks                  <- orthosfm::filterOutlierTracks
    filter_tracks_with_reprojection_error  <- orthosfm::filterTracksWithReprojectionError
    nearest_neighbour_distance             <- orthosfm::getNearestNeighbourDistance
        (src/triangulation/outlier_filtering.cpp:14-38, 40-125, 127-192)

Same names, argument meaning and outputs; the O(P^2) distance search, the
triangulation and the reprojection errors run on the device through the C ABI
(no CPU fallback).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .ba import Track, _flatten

MAX_ALLOWED_REPROJECTION_ERROR = 1.5     # pixels, outlier_filtering.cpp:140


def nearest_neighbour_distance(points, device: int = 0) -> np.ndarray:
    """points: (n, 4) homogeneous; returns the (n,) distances."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
    nn = np.zeros(max(pts.shape[0], 1))
    capi.check(capi.lib.osfm_nn_distances(device, capi._ptr(pts, C.c_double), pts.shape[0],
                                          capi._ptr(nn, C.c_double)))
    return nn[:pts.shape[0]]


def outlier_track_flags(points, has_point, device: int = 0):
    """keep flags + (mean, sigma) of filterOutlierTracks on flattened tracks."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
    hp = np.ascontiguousarray(has_point, dtype=np.uint8).reshape(-1)
    keep = np.zeros(max(hp.shape[0], 1), dtype=np.uint8)
    st = capi.OutlierStats()
    capi.check(capi.lib.osfm_filter_outlier_tracks(device, capi._ptr(pts, C.c_double), capi._ptr(hp, C.c_uint8),
                                                   hp.shape[0], capi._ptr(keep, C.c_uint8), C.byref(st)))
    return keep[:hp.shape[0]].astype(bool), st


def filter_outlier_tracks(tracks, cameras=None, device: int = 0, verbose: bool = True):
    """orthosfm::filterOutlierTracks(tracks, cameras): the cameras argument is
    unused by the reference as well."""
    pts = np.array([np.asarray(t.point, dtype=np.float64) for t in tracks]).reshape(-1, 4)
    keep, _ = outlier_track_flags(pts, [t.has_point for t in tracks], device)
    out = [t for t, k in zip(tracks, keep) if k]
    if verbose:
        print(f"{len(tracks) - len(out)} outlier tracks discarded ({len(out)} tracks remaining)")
    return out


def _track_key(t: Track):
    """What Track::operator== compares (track.cpp:86-99): the globalFeatureIDs
    in order, globalFeatureID = 32768 * view + feature (matching.h:24,
    matching_mve.cpp:461-463 -- it collides above 32768 features per view, and
    so does this key)."""
    return tuple(32768 * f.viewID + f.localFeatureID for f in t.features)


def filter_tracks_with_reprojection_error(tracks, cameras, algorithm=None, device: int = 0,
                                          verbose: bool = True,
                                          max_error: float = MAX_ALLOWED_REPROJECTION_ERROR):
    """orthosfm::filterTracksWithReprojectionError.  Tracks observed by ALL
    cameras are re-triangulated; their features with a reprojection error of
    max_error pixels or more are dropped, and the track with them when fewer
    than two features remain.  All other tracks pass unchanged."""
    ids = []
    for c in cameras:
        ids.append(c.view_id)
    avail = set(ids)
    # filterTracksToAvailableCameras(cameras, tracks, true, true) (common.cpp:85-139):
    # the ORIGINAL track is kept when the number of its features inside the
    # camera set equals the number of cameras
    full = [i for i, t in enumerate(tracks)
            if sum(1 for f in t.features if f.viewID in avail) == len(ids)]
    if not full:
        return list(tracks)
    # triangulateTracks(cameras, fullSizeTracks, true) + evaluateReprojectionError per feature
    work = [Track(tracks[i].features, np.asarray(tracks[i].point, dtype=np.float64).copy(), True) for i in full]
    fp, used = _flatten(cameras, work)
    st = fp.struct()
    obs_keep = np.zeros(max(fp.obs_camera.shape[0], 1), dtype=np.uint8)
    capi.check(capi.lib.osfm_filter_reprojection(C.byref(st), device, C.c_double(max_error),
                                                 capi._ptr(obs_keep, C.c_uint8), None, None))
    cam_ids = {}
    for c in cameras:
        cam_ids.setdefault(c.view_id, True)
    # std::find over fullSizeTracks returns the FIRST equal track (:147)
    first_equal = {}
    for pos, i in enumerate(full):
        first_equal.setdefault(_track_key(tracks[i]), pos)
    # observation ranges of the flattened full-size tracks
    starts = np.zeros(len(work) + 1, dtype=np.int64)
    np.add.at(starts, np.asarray(fp.obs_point, dtype=np.int64) + 1, 1)
    starts = np.cumsum(starts)
    out = []
    for i, t in enumerate(tracks):
        pos = first_equal.get(_track_key(t))
        if pos is None:
            out.append(t)
            continue
        k = int(starts[pos])
        feats = []
        for f in t.features:
            if f.viewID in cam_ids:
                if obs_keep[k]:
                    feats.append(f)
                k += 1
            else:
                feats.append(f)           # no camera: no judgement, keep (:170-173)
        if len(feats) > 1:
            out.append(Track(feats, np.asarray(t.point, dtype=np.float64).copy() if t.has_point else np.zeros(4),
                             t.has_point))
    if verbose:
        print(f"{len(out)} out of {len(tracks)} tracks remaining after outlier filtering")
    return out
