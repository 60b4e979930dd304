"""ctypes mirror of the device-resident scene (include/osfm_hip.h, osfm_scene_*): the track table of the incremental
reconstruction uploaded once, every step of runPoseEstimation (src/sfm/reconstruct.cpp:193-281) a call that
selects its observations from the flags on the device.  Used by orthosfm_amd/pipeline.py; a C++ caller uses the
same entries (tests/host/scene_check.cc)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class Scene:
    def __init__(self, model: int, img_w, img_h, track_offsets, feat_view, feat_xy, device: int = 0):
        self.num_views = len(img_w)
        self._w = np.ascontiguousarray(img_w, dtype=np.int32)
        self._h = np.ascontiguousarray(img_h, dtype=np.int32)
        off = np.ascontiguousarray(track_offsets, dtype=np.int64)
        view = np.ascontiguousarray(feat_view, dtype=np.int32)
        xy = np.ascontiguousarray(feat_xy, dtype=np.float32).reshape(-1, 2)
        self.num_tracks = int(off.shape[0] - 1)
        self.num_features = int(view.shape[0])
        self._h_scene = C.c_void_p()
        capi.check(capi.lib.osfm_scene_create(C.c_int(device), C.c_int(model), C.c_int(self.num_views),
                                              capi._ptr(self._w, C.c_int32), capi._ptr(self._h, C.c_int32),
                                              C.c_int32(self.num_tracks), capi._ptr(off, C.c_int64),
                                              capi._ptr(view, C.c_int32), capi._ptr(xy, C.c_float), C.byref(self._h_scene)))

    def close(self):
        if self._h_scene:
            capi.lib.osfm_scene_destroy(self._h_scene)
            self._h_scene = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_flags(self, alive_track=None, alive_feature=None):
        at = None if alive_track is None else np.ascontiguousarray(alive_track, dtype=np.uint8)
        af = None if alive_feature is None else np.ascontiguousarray(alive_feature, dtype=np.uint8)
        capi.check(capi.lib.osfm_scene_set_flags(self._h_scene, None if at is None else capi._ptr(at, C.c_uint8),
                                                 None if af is None else capi._ptr(af, C.c_uint8)))

    def align_views(self, views, params, cam_const):
        v = np.ascontiguousarray(views, dtype=np.int32)
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 7)
        c = np.ascontiguousarray(cam_const, dtype=np.uint8).reshape(-1, 7)
        capi.check(capi.lib.osfm_scene_align_views(self._h_scene, C.c_int(v.shape[0]), capi._ptr(v, C.c_int32),
                                                   capi._ptr(p, C.c_double), capi._ptr(c, C.c_uint8)))

    def set_cameras(self, views, params):
        """Replaces the parameters of views that are aligned already (the next triangulation is a full pass)."""
        v = np.ascontiguousarray(views, dtype=np.int32)
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 7)
        capi.check(capi.lib.osfm_scene_set_cameras(self._h_scene, C.c_int(v.shape[0]), capi._ptr(v, C.c_int32),
                                                   capi._ptr(p, C.c_double)))

    def cameras(self):
        n = C.c_int32()
        capi.check(capi.lib.osfm_scene_get_cameras(self._h_scene, 0, None, None, C.byref(n)))
        views = np.zeros(max(n.value, 1), dtype=np.int32)
        params = np.zeros((max(n.value, 1), 7))
        capi.check(capi.lib.osfm_scene_get_cameras(self._h_scene, n.value, capi._ptr(views, C.c_int32),
                                                   capi._ptr(params, C.c_double), C.byref(n)))
        return views[:n.value], params[:n.value]

    def triangulate(self, new_views=None, check_full=False) -> int:
        """Returns the number of tracks on which the incremental pass differs from a full one (check_full)."""
        bad = C.c_int32()
        if new_views is None:
            capi.check(capi.lib.osfm_scene_triangulate(self._h_scene, 0, None, 0, C.byref(bad)))
        else:
            v = np.ascontiguousarray(new_views, dtype=np.int32)
            capi.check(capi.lib.osfm_scene_triangulate(self._h_scene, C.c_int(v.shape[0]), capi._ptr(v, C.c_int32),
                                                       C.c_int(1 if check_full else 0), C.byref(bad)))
        return bad.value

    def filter_reprojection(self, max_error: float):
        capi.check(capi.lib.osfm_scene_filter_reprojection(self._h_scene, C.c_double(max_error)))

    def local_adjustment(self, views, params, cam_const, max_error: float, options):
        """params is updated in place; returns (summary, points, observations)."""
        v = np.ascontiguousarray(views, dtype=np.int32)
        c = np.ascontiguousarray(cam_const, dtype=np.uint8).reshape(-1, 7)
        assert params.dtype == np.float64 and params.flags["C_CONTIGUOUS"]
        s = capi.BaSummary()
        m, o = C.c_int32(), C.c_int32()
        capi.check(capi.lib.osfm_scene_local_adjustment(self._h_scene, C.c_int(v.shape[0]), capi._ptr(v, C.c_int32),
                                                        capi._ptr(params, C.c_double), capi._ptr(c, C.c_uint8),
                                                        C.c_double(max_error), C.byref(options), C.byref(s),
                                                        C.byref(m), C.byref(o)))
        return s, m.value, o.value

    def global_adjustment(self, options):
        s = capi.BaSummary()
        m, o = C.c_int32(), C.c_int32()
        capi.check(capi.lib.osfm_scene_global_adjustment(self._h_scene, C.byref(options), C.byref(s), C.byref(m), C.byref(o)))
        return s, m.value, o.value

    def filter_outliers(self) -> int:
        k = C.c_int32()
        capi.check(capi.lib.osfm_scene_filter_outliers(self._h_scene, None, C.byref(k)))
        return k.value

    def download(self):
        """(alive_track, alive_feature, has_point, points) as the scene holds them."""
        at = np.zeros(max(self.num_tracks, 1), dtype=np.uint8)
        af = np.zeros(max(self.num_features, 1), dtype=np.uint8)
        hp = np.zeros(max(self.num_tracks, 1), dtype=np.uint8)
        pt = np.zeros((max(self.num_tracks, 1), 4))
        capi.check(capi.lib.osfm_scene_download(self._h_scene, capi._ptr(at, C.c_uint8), capi._ptr(af, C.c_uint8),
                                                capi._ptr(hp, C.c_uint8), capi._ptr(pt, C.c_double)))
        return (at[:self.num_tracks].astype(bool), af[:self.num_features].astype(bool),
                hp[:self.num_tracks].astype(bool), pt[:self.num_tracks])
