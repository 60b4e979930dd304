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
"""
from __future__ import annotations

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
            parts = [str(len(t.features))]
            for f in t.features:
                parts += [str(f.viewID), str(f.localFeatureID), str(f.globalFeatureID), _g(np.float32(f.x)),
                          _g(np.float32(f.y)), str(f.r), str(f.g), str(f.b)]
            fh.write(";".join(parts) + "\n")


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
