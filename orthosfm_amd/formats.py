"""On-disk text formats of the reference pipeline, so that
GPU-produced tracks / cameras / timings can be fed to `orthosfm-app
--calculated-tracks` and parsed by the testbench unchanged:

    tracks.txt            src/matching/matching_io.cpp:16-48 (write), :50-97 (read)
    pairwise %03d_%03d    src/matching/matching_io.cpp:99-141
    cameras.txt           src/data_structures/camera_io.cpp:15-40 (write), :42-71 (read)
    sparse_cloud.ply      src/util/common.cpp:141-188
    time_measurements.txt src/util/timing.cpp:18-28

plus the conversion of MVE tracks to orthosfm tracks
(src/matching/matching_mve.cpp:455-466).  Number formatting follows the C++
streams the reference uses: `ostream << float/double` is "%g" with 6
significant digits, `std::to_string(double)` is "%f".

PARITY UNPINNED: the reference files need Boost / OpenCV / Eigen and cannot be
built in this image, and the reference holds no sample files; the formats are
restated from the source and checked by round trips.

Two forms of everything: the C ABI (`osfm_tracks_file_write`, ... in
include/osfm_hip.h, csrc/formats_api.hip -- what a C++ caller links; it formats
with the very stream operations the reference uses) wrapped by the `*_native`
functions on flat arrays, and a pure-Python restatement on `Track` objects that
the tests hold against it byte for byte.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Feature:
    """orthosfm::Feature (src/data_structures/track.h:21-31)."""
    viewID: int
    localFeatureID: int
    globalFeatureID: int
    x: float
    y: float
    r: int = 0
    g: int = 0
    b: int = 0


@dataclass
class Track:
    features: list = field(default_factory=list)
    point: np.ndarray | None = None          # homogeneous (4,), None = hasPoint() false


def _g(v) -> str:
    """ostream << v for float / double: general format, precision 6."""
    return "%g" % float(v)


def mve_tracks_to_orthosfm(track_offsets, track_features, positions, image_width, colors=None):
    """matching_mve.cpp:455-466.  positions: per view (n, 2) float32 normalised
    feature positions; image_width: the (single, last written) width the
    reference uses for BOTH axes; globalFeatureID = 32768 * view + feature."""
    out = []
    w = float(image_width)
    for t in range(len(track_offsets) - 1):
        feats = []
        for k in range(int(track_offsets[t]), int(track_offsets[t + 1])):
            v, f = int(track_features[k][0]), int(track_features[k][1])
            px, py = positions[v][f]
            # double arithmetic (imageWidth is a double, 0.5 a double), stored as float
            x = np.float32(w * (float(np.float32(px)) + 0.5))
            y = np.float32(w * (float(np.float32(py)) + 0.5))
            ft = Feature(v, f, 32768 * v + f, float(x), float(y))
            if colors is not None:
                ft.r, ft.g, ft.b = (int(c) for c in colors[v][f])
            feats.append(ft)
        out.append(Track(feats))
    return out


def save_tracks_to_file(tracks, path):
    with open(path, "w") as fh:
        for t in tracks:
            parts = []
            for f in t.features:
                parts += [str(f.viewID), str(f.localFeatureID), str(f.globalFeatureID), _g(np.float32(f.x)),
                          _g(np.float32(f.y)), str(f.r), str(f.g), str(f.b)]
            # the count is always followed by ";" (:25), so an empty track is the line "0;"
            fh.write(str(len(t.features)) + ";" + ";".join(parts) + "\n")


def load_tracks_from_file(path):
    tracks = []
    with open(path) as fh:
        for line in fh:
            s = line.rstrip("\n").split(";")
            n = int(s[0])
            feats = []
            for i in range(n):
                o = 1 + 8 * i
                ft = Feature(int(s[o]), int(s[o + 1]), int(s[o + 2]), float(np.float32(float(s[o + 3]))),
                             float(np.float32(float(s[o + 4]))), int(s[o + 5]), int(s[o + 6]), int(s[o + 7]))
                feats.append(ft)
            tracks.append(Track(feats))
    return tracks


def save_tracks_to_pairwise_files(tracks, view_ids, folder):
    """matching_io.cpp:99-141: for every pair of views the tracks holding BOTH
    (filterTracksToAvailableCameras(ids, tracks, true, false)), one line
    "x_i y_i x_j y_j" each."""
    import os
    written = []
    for a in range(len(view_ids)):
        for b in range(a + 1, len(view_ids)):
            ids = (view_ids[a], view_ids[b])
            rows = []
            for t in tracks:
                cur = [f for f in t.features if f.viewID in ids]
                if len(cur) != 2:
                    continue
                line = ""
                for k, vid in enumerate(ids):
                    for f in cur:
                        if f.viewID == vid:
                            line += _g(np.float32(f.x)) + " " + _g(np.float32(f.y)) + (" " if k == 0 else "\n")
                rows.append(line)
            if not rows:
                continue
            path = os.path.join(folder, "%03d_%03d.txt" % ids)
            with open(path, "w") as fh:
                fh.write("".join(rows))
            written.append(path)
    return written


def export_cameras_to_file(names, matrices, path):
    """camera_io.cpp:15-40: name;16 comma-separated row-major entries of the
    4x4 camera-to-world matrix [x y z origin; 0 0 0 1], std::to_string format."""
    with open(path, "w") as fh:
        for name, m in zip(names, matrices):
            m = np.asarray(m, dtype=np.float64).reshape(4, 4)
            fh.write(name + ";" + ",".join("%f" % v for v in m.reshape(-1)) + "\n")


def import_camera_file_as_matrix(path):
    out = []
    with open(path) as fh:
        for line in fh:
            name, rest = line.rstrip("\n").split(";")[:2]
            out.append((name, np.array([float(v) for v in rest.split(",")]).reshape(4, 4)))
    return out


def save_points_to_ply(path, tracks):
    """common.cpp:141-188: ASCII PLY of the tracks that have a point, coloured
    by their first feature."""
    pts = [t for t in tracks if t.point is not None]
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\n" % len(pts))
        fh.write("property float x\nproperty float y\nproperty float z\n")
        fh.write("property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        for t in pts:
            p = np.asarray(t.point, dtype=np.float64)
            f0 = t.features[0]
            fh.write("%s %s %s %d %d %d\n" % (_g(p[0]), _g(p[1]), _g(p[2]), f0.r, f0.g, f0.b))


def save_runtimes_to_txt(path, init, track, pose, total):
    """timing.cpp:18-28."""
    with open(path, "w") as fh:
        fh.write("Initialization Time [s] = %s\n" % _g(init))
        fh.write("Track Building Time [s] = %s\n" % _g(track))
        fh.write("Pose Estimation Time [s] = %s\n" % _g(pose))
        fh.write("Total Time [s] = %s\n" % _g(total))


def runtimes_from_txt(path):
    vals = []
    with open(path) as fh:
        for line in fh:
            vals.append(float(line.split("=")[1]))
    return dict(zip(("init", "track", "pose", "total"), vals))


# ---------------------------------------------------------------------------
# The C ABI (csrc/formats_api.hip) on flat arrays: tracks are CSR -- `offsets`
# int64 [T + 1] over `features`, a record array of capi.TRACK_FEATURE.
# ---------------------------------------------------------------------------
def _native():
    from . import capi
    return capi


def tracks_to_flat(tracks):
    """Track objects -> (offsets, features, points [T][4], has_point [T])."""
    capi = _native()
    offsets = np.zeros(len(tracks) + 1, dtype=np.int64)
    for t, tr in enumerate(tracks):
        offsets[t + 1] = offsets[t] + len(tr.features)
    feats = np.zeros(int(offsets[-1]), dtype=capi.TRACK_FEATURE)
    k = 0
    for tr in tracks:
        for f in tr.features:
            feats[k] = (f.viewID, f.localFeatureID, f.globalFeatureID, f.x, f.y, f.r, f.g, f.b)
            k += 1
    points = np.zeros((len(tracks), 4))
    has_point = np.zeros(len(tracks), dtype=np.uint8)
    for t, tr in enumerate(tracks):
        if tr.point is not None:
            points[t] = tr.point
            has_point[t] = 1
    return offsets, feats, points, has_point


def flat_to_tracks(offsets, feats):
    out = []
    for t in range(len(offsets) - 1):
        out.append(Track([Feature(int(f["view_id"]), int(f["local_feature_id"]), int(f["global_feature_id"]),
                                  float(f["x"]), float(f["y"]), int(f["r"]), int(f["g"]), int(f["b"]))
                          for f in feats[int(offsets[t]):int(offsets[t + 1])]]))
    return out


def _csr_args(capi, offsets, feats):
    import ctypes as C
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    feats = np.ascontiguousarray(feats, dtype=capi.TRACK_FEATURE)
    if offsets.size < 1 or (offsets.size > 1 and int(offsets[-1]) > feats.size):
        raise ValueError("offsets do not fit the feature array")
    return (offsets, feats, C.c_int64(offsets.size - 1), offsets.ctypes.data_as(C.c_void_p),
            feats.ctypes.data_as(C.c_void_p) if feats.size else None)


def save_tracks_to_file_native(offsets, feats, path):
    capi = _native()
    offsets, feats, n, po, pf = _csr_args(capi, offsets, feats)
    capi.check(capi.lib.osfm_tracks_file_write(os.fsencode(path), n, po, pf))


def load_tracks_from_file_native(path):
    """-> (offsets, features)"""
    import ctypes as C
    capi = _native()
    nt, nf = C.c_int64(), C.c_int64()
    st = capi.lib.osfm_tracks_file_read(os.fsencode(path), C.c_int64(0), C.c_int64(0), None, None,
                                        C.byref(nt), C.byref(nf))
    if st not in (capi.OK, capi.E_CAPACITY):
        capi.check(st)
    offsets = np.zeros(nt.value + 1, dtype=np.int64)
    feats = np.zeros(nf.value, dtype=capi.TRACK_FEATURE)
    capi.check(capi.lib.osfm_tracks_file_read(
        os.fsencode(path), C.c_int64(nt.value), C.c_int64(nf.value), offsets.ctypes.data_as(C.c_void_p),
        feats.ctypes.data_as(C.c_void_p) if feats.size else None, C.byref(nt), C.byref(nf)))
    return offsets, feats


def save_tracks_to_pairwise_files_native(offsets, feats, view_ids, folder):
    """-> number of files written"""
    import ctypes as C
    capi = _native()
    offsets, feats, n, po, pf = _csr_args(capi, offsets, feats)
    ids = np.ascontiguousarray(view_ids, dtype=np.uint32)
    written = C.c_int64()
    capi.check(capi.lib.osfm_tracks_pairwise_files_write(
        os.fsencode(folder), C.c_int32(ids.size), ids.ctypes.data_as(C.c_void_p) if ids.size else None, n, po, pf,
        C.byref(written)))
    return written.value


def mve_tracks_to_flat_native(track_features, positions, image_width, colors=None):
    """matching_mve.cpp:455-466 on the output of osfm_tracks_compute: track_features
    [F][2] (view, feature); positions / colors: per-view arrays."""
    import ctypes as C
    capi = _native()
    tf = np.ascontiguousarray(track_features, dtype=np.int32).reshape(-1, 2)
    starts = np.zeros(len(positions) + 1, dtype=np.int64)
    for v, p in enumerate(positions):
        starts[v + 1] = starts[v] + len(p)
    pos = (np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1, 2) for p in positions])
           if len(positions) else np.zeros((0, 2), np.float32))
    pos = np.ascontiguousarray(pos)
    col = None
    if colors is not None:
        col = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint8).reshape(-1, 3) for c in colors]))
    feats = np.zeros(tf.shape[0], dtype=capi.TRACK_FEATURE)
    capi.check(capi.lib.osfm_tracks_from_mve(
        C.c_int64(tf.shape[0]), tf.ctypes.data_as(C.c_void_p) if tf.size else None, C.c_int32(len(positions)),
        starts.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p) if pos.size else None,
        col.ctypes.data_as(C.c_void_p) if col is not None and col.size else None, C.c_double(image_width),
        feats.ctypes.data_as(C.c_void_p) if feats.size else None))
    return feats


def export_cameras_to_file_native(names, matrices, path):
    import ctypes as C
    capi = _native()
    m = np.ascontiguousarray(np.asarray(matrices, dtype=np.float64).reshape(-1, 16))
    if m.shape[0] != len(names):
        raise ValueError("one matrix per name")
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    capi.check(capi.lib.osfm_cameras_file_write(os.fsencode(path), C.c_int32(len(names)), arr,
                                                m.ctypes.data_as(C.c_void_p) if m.size else None))


def import_camera_file_as_matrix_native(path):
    import ctypes as C
    capi = _native()
    nc, nb = C.c_int32(), C.c_int64()
    st = capi.lib.osfm_cameras_file_read(os.fsencode(path), C.c_int32(0), C.c_int64(0), None, None,
                                         C.byref(nc), C.byref(nb))
    if st not in (capi.OK, capi.E_CAPACITY):
        capi.check(st)
    buf = C.create_string_buffer(max(nb.value, 1))
    m = np.zeros((nc.value, 16))
    capi.check(capi.lib.osfm_cameras_file_read(os.fsencode(path), nc, nb, buf,
                                               m.ctypes.data_as(C.c_void_p) if m.size else None,
                                               C.byref(nc), C.byref(nb)))
    names = buf.raw[:nb.value].split(b"\0")[:nc.value]
    return [(n.decode(), m[i].reshape(4, 4)) for i, n in enumerate(names)]


def save_points_to_ply_native(path, offsets, feats, points, has_point):
    import ctypes as C
    capi = _native()
    offsets, feats, n, po, pf = _csr_args(capi, offsets, feats)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
    hp = np.ascontiguousarray(has_point, dtype=np.uint8)
    if pts.shape[0] != offsets.size - 1 or hp.size != offsets.size - 1:
        raise ValueError("one point / flag per track")
    capi.check(capi.lib.osfm_sparse_cloud_write(os.fsencode(path), n, po, pf,
                                                pts.ctypes.data_as(C.c_void_p) if pts.size else None,
                                                hp.ctypes.data_as(C.c_void_p) if hp.size else None))


def save_runtimes_to_txt_native(path, init, track, pose, total):
    import ctypes as C
    capi = _native()
    v = (C.c_double * 4)(init, track, pose, total)
    capi.check(capi.lib.osfm_time_measurements_write(os.fsencode(path), v))


def runtimes_from_txt_native(path):
    import ctypes as C
    capi = _native()
    v = (C.c_double * 4)()
    capi.check(capi.lib.osfm_time_measurements_read(os.fsencode(path), v))
    return dict(zip(("init", "track", "pose", "total"), list(v)))
