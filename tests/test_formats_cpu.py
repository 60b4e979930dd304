"""Text formats of the reference pipeline (orthosfm_amd/formats.py; PARITY
UNPINNED, see its header): exact strings for known values (C++ stream / to_string
formatting) and round trips."""
import os

import numpy as np
import pytest

from orthosfm_amd import formats as F


def test_tracks_file_strings_and_round_trip(tmp_path):
    t = [F.Track([F.Feature(0, 12, 12, 1023.5, 7.25, 10, 20, 30), F.Feature(3, 4, 98308, 0.1, 123456.789)]),
         F.Track([F.Feature(1, 0, 32768, 1e-5, 2048.0), F.Feature(2, 5, 65541, 3.14159274, 1e7)])]
    p = tmp_path / "tracks.txt"
    F.save_tracks_to_file(t, p)
    lines = p.read_text().split("\n")
    # ostream << float: 6 significant digits, general format (123456.789f -> 123457, 1e7 -> 1e+07)
    assert lines[0] == "2;0;12;12;1023.5;7.25;10;20;30;3;4;98308;0.1;123457;0;0;0"
    assert lines[1] == "2;1;0;32768;1e-05;2048;0;0;0;2;5;65541;3.14159;1e+07;0;0;0"
    back = F.load_tracks_from_file(p)
    assert [len(x.features) for x in back] == [2, 2]
    assert back[0].features[1].globalFeatureID == 98308 and back[0].features[0].b == 30
    assert back[1].features[1].x == float(np.float32(3.14159))      # what the 6 digits keep
    # a second save of the loaded tracks is a fixed point
    F.save_tracks_to_file(back, tmp_path / "again.txt")
    assert (tmp_path / "again.txt").read_text() == p.read_text()


def test_mve_track_conversion():
    pos = [np.array([[-0.25, 0.125], [0.0, 0.0]], np.float32), np.array([[0.4999, -0.5]], np.float32)]
    tr = F.mve_tracks_to_orthosfm([0, 2], [[0, 0], [1, 0]], pos, 2048)
    f0, f1 = tr[0].features
    assert (f0.viewID, f0.localFeatureID, f0.globalFeatureID) == (0, 0, 0)
    assert f1.globalFeatureID == 32768
    assert f0.x == 512.0 and f0.y == 1280.0                         # width * (pos + 0.5), both axes
    assert f1.x == float(np.float32(2048.0 * (float(np.float32(0.4999)) + 0.5))) and f1.y == 0.0


def test_pairwise_files(tmp_path):
    t = [F.Track([F.Feature(0, 0, 0, 1.5, 2.5), F.Feature(1, 0, 1, 3.5, 4.5), F.Feature(2, 0, 2, 5.5, 6.5)]),
         F.Track([F.Feature(2, 1, 3, 9.0, 8.0), F.Feature(0, 1, 4, 7.0, 6.0)])]
    files = F.save_tracks_to_pairwise_files(t, [0, 1, 2], str(tmp_path))
    assert sorted(os.path.basename(f) for f in files) == ["000_001.txt", "000_002.txt", "001_002.txt"]
    # view i first, then view j, whatever the order inside the track
    assert (tmp_path / "000_002.txt").read_text() == "1.5 2.5 5.5 6.5\n7 6 9 8\n"
    assert (tmp_path / "000_001.txt").read_text() == "1.5 2.5 3.5 4.5\n"


def test_cameras_ply_timings(tmp_path):
    m = np.eye(4)
    m[0, 3], m[1, 1] = 1.23456789, -0.5
    F.export_cameras_to_file(["img_000.png"], [m], tmp_path / "cameras.txt")
    line = (tmp_path / "cameras.txt").read_text()
    assert line.startswith("img_000.png;1.000000,0.000000,0.000000,1.234568,0.000000,-0.500000,")
    (name, back), = F.import_camera_file_as_matrix(tmp_path / "cameras.txt")
    assert name == "img_000.png" and np.allclose(back, m, atol=5e-7)

    tr = [F.Track([F.Feature(0, 0, 0, 1, 1, 255, 128, 0)], np.array([0.5, -1.25, 1e-7, 1.0])),
          F.Track([F.Feature(0, 1, 1, 2, 2)], None)]
    F.save_points_to_ply(tmp_path / "c.ply", tr)
    ply = (tmp_path / "c.ply").read_text().split("\n")
    assert ply[2] == "element vertex 1" and ply[10] == "0.5 -1.25 1e-07 255 128 0"

    F.save_runtimes_to_txt(tmp_path / "t.txt", 1.5, 120.25, 33.0, 154.75)
    assert (tmp_path / "t.txt").read_text().split("\n")[1] == "Track Building Time [s] = 120.25"
    assert F.runtimes_from_txt(tmp_path / "t.txt") == {"init": 1.5, "track": 120.25, "pose": 33.0, "total": 154.75}


# ---------------------------------------------------------------------------
# The C ABI (csrc/formats_api.hip) against the Python restatement, byte for byte.
# The C++ side formats with ostream / std::to_string themselves, so agreement here
# also pins the "%g" / "%f" reading of those operations the Python side rests on.
# ---------------------------------------------------------------------------
def _random_tracks(rng, n_tracks, n_views, with_points=True):
    specials = [0.0, -0.0, 1e-5, 123456.789, 1e7, 999999.5, 9999995.0, 0.1, 3.4e38, 100000.0, 1234567.0,
                0.000123456789, 2048.0, 1023.5]
    tracks = []
    for t in range(n_tracks):
        views = rng.choice(n_views, size=int(rng.integers(1, min(n_views, 6) + 1)), replace=False)
        feats = []
        for v in views:
            x = specials[int(rng.integers(len(specials)))] if rng.random() < 0.3 else float(rng.uniform(0, 4096))
            y = float(rng.uniform(-1, 1) * 10 ** rng.uniform(-6, 6))
            f = int(rng.integers(0, 30000))
            feats.append(F.Feature(int(v), f, 32768 * int(v) + f, float(np.float32(x)), float(np.float32(y)),
                                   int(rng.integers(256)), int(rng.integers(256)), int(rng.integers(256))))
        point = None
        if with_points and rng.random() < 0.7:
            point = np.array([rng.normal() * 10 ** rng.uniform(-8, 3), rng.normal(), specials[t % len(specials)], 1.0])
        tracks.append(F.Track(feats, point))
    return tracks


def test_native_tracks_file_equals_python(tmp_path):
    rng = np.random.default_rng(5)
    tracks = _random_tracks(rng, 300, 9) + [F.Track([])]        # an empty track is the line "0;"
    off, feats, _, _ = F.tracks_to_flat(tracks)
    F.save_tracks_to_file(tracks, tmp_path / "py.txt")
    F.save_tracks_to_file_native(off, feats, tmp_path / "c.txt")
    assert (tmp_path / "c.txt").read_bytes() == (tmp_path / "py.txt").read_bytes()
    # reading: the C ABI and the Python reader agree, and re-writing is a fixed point
    off2, feats2 = F.load_tracks_from_file_native(tmp_path / "py.txt")
    back = F.load_tracks_from_file(tmp_path / "py.txt")
    assert np.array_equal(off2, off)
    o3, f3, _, _ = F.tracks_to_flat(back)
    assert np.array_equal(o3, off2) and f3.tobytes() == feats2.tobytes()
    F.save_tracks_to_file_native(off2, feats2, tmp_path / "again.txt")
    assert (tmp_path / "again.txt").read_bytes() == (tmp_path / "py.txt").read_bytes()
    assert [len(t.features) for t in F.flat_to_tracks(off2, feats2)] == [len(t.features) for t in tracks]


def test_native_pairwise_files_equal_python(tmp_path):
    rng = np.random.default_rng(6)
    tracks = _random_tracks(rng, 200, 7, with_points=False)
    # a track with two features of one view (the reference's filter counts features, not views)
    tracks.append(F.Track([F.Feature(2, 1, 65537, 5.0, 6.0), F.Feature(2, 2, 65538, 7.0, 8.0)]))
    ids = [0, 1, 2, 3, 4, 5, 6, 11]                              # view 11 has no track: no file
    (tmp_path / "py").mkdir(); (tmp_path / "c").mkdir()
    files = F.save_tracks_to_pairwise_files(tracks, ids, str(tmp_path / "py"))
    off, feats, _, _ = F.tracks_to_flat(tracks)
    n = F.save_tracks_to_pairwise_files_native(off, feats, ids, str(tmp_path / "c"))
    assert n == len(files) > 10
    assert sorted(os.listdir(tmp_path / "c")) == sorted(os.listdir(tmp_path / "py"))
    for name in os.listdir(tmp_path / "py"):
        assert (tmp_path / "c" / name).read_bytes() == (tmp_path / "py" / name).read_bytes(), name


def test_native_mve_conversion_equals_python():
    rng = np.random.default_rng(7)
    positions = [rng.uniform(-0.5, 0.5, size=(n, 2)).astype(np.float32) for n in (40, 0, 25, 33)]
    colors = [rng.integers(0, 256, size=(len(p), 3)).astype(np.uint8) for p in positions]
    tf, offs = [], [0]
    for _ in range(30):
        for v in rng.choice([0, 2, 3], size=int(rng.integers(2, 4)), replace=False):
            tf.append([int(v), int(rng.integers(len(positions[v])))])
        offs.append(len(tf))
    for width, col in ((2048, None), (1936.0, colors)):
        py = F.mve_tracks_to_orthosfm(offs, tf, positions, width, col)
        _, want, _, _ = F.tracks_to_flat(py)
        got = F.mve_tracks_to_flat_native(tf, positions, width, col)
        assert got.tobytes() == want.tobytes()
    from orthosfm_amd import capi
    with pytest.raises(capi.OsfmError) as e:
        F.mve_tracks_to_flat_native([[1, 0]], positions, 2048)   # view 1 has no features
    assert e.value.status == capi.E_RANGE


def test_native_cameras_ply_timings_equal_python(tmp_path):
    rng = np.random.default_rng(8)
    names = ["img_%03d.png" % i for i in range(12)] + ["with space.jpg"]
    mats = []
    for i in range(len(names)):
        m = np.eye(4)
        m[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        m[:3, 3] = rng.normal(size=3) * 10 ** rng.uniform(-7, 6)
        mats.append(m)
    mats[3][0, 3] = -0.0000004          # "%f" rounds to -0.000000
    mats[4][1, 3] = 1e15
    F.export_cameras_to_file(names, mats, tmp_path / "py.txt")
    F.export_cameras_to_file_native(names, mats, tmp_path / "c.txt")
    assert (tmp_path / "c.txt").read_bytes() == (tmp_path / "py.txt").read_bytes()
    a = F.import_camera_file_as_matrix(tmp_path / "py.txt")
    b = F.import_camera_file_as_matrix_native(tmp_path / "py.txt")
    assert [n for n, _ in a] == [n for n, _ in b] == names
    assert all(np.array_equal(x, y) for (_, x), (_, y) in zip(a, b))

    tracks = _random_tracks(rng, 150, 5)
    off, feats, pts, has = F.tracks_to_flat(tracks)
    F.save_points_to_ply(tmp_path / "py.ply", tracks)
    F.save_points_to_ply_native(tmp_path / "c.ply", off, feats, pts, has)
    assert (tmp_path / "c.ply").read_bytes() == (tmp_path / "py.ply").read_bytes()
    assert 50 < int(has.sum()) < 150

    for vals in ((0.0123456789, 12.5, 3600.123456, 3612.6358), (1e-7, 0.0, 123456789.0, 1e21)):
        F.save_runtimes_to_txt(tmp_path / "py_t.txt", *vals)
        F.save_runtimes_to_txt_native(tmp_path / "c_t.txt", *vals)
        assert (tmp_path / "c_t.txt").read_bytes() == (tmp_path / "py_t.txt").read_bytes()
        assert F.runtimes_from_txt_native(tmp_path / "py_t.txt") == F.runtimes_from_txt(tmp_path / "py_t.txt")


def test_native_format_errors(tmp_path):
    from orthosfm_amd import capi
    with pytest.raises(capi.OsfmError) as e:
        F.load_tracks_from_file_native(tmp_path / "missing.txt")
    assert e.value.status == capi.E_IO
    (tmp_path / "bad.txt").write_text("2;0;1;2;3.5;4.5;0;0;0;1;2\n")          # second feature cut short
    with pytest.raises(capi.OsfmError) as e:
        F.load_tracks_from_file_native(tmp_path / "bad.txt")
    assert e.value.status == capi.E_IO and "line 1" in str(e.value)
    (tmp_path / "bad2.txt").write_text("1;0;1;x;3.5;4.5;0;0;0\n")               # std::stoi throws in the reference
    with pytest.raises(capi.OsfmError):
        F.load_tracks_from_file_native(tmp_path / "bad2.txt")
    with pytest.raises(capi.OsfmError) as e:
        F.save_tracks_to_file_native(np.array([0, 1]), np.zeros(1, capi.TRACK_FEATURE), tmp_path / "nodir" / "t.txt")
    assert e.value.status == capi.E_IO
    with pytest.raises(capi.OsfmError) as e:                                    # offsets must not decrease
        F.save_tracks_to_file_native(np.array([0, 2, 1]), np.zeros(2, capi.TRACK_FEATURE), tmp_path / "t.txt")
    assert e.value.status == capi.E_ARG
    # a subnormal coordinate prints fine and then fails to load: std::stof reports ERANGE as
    # std::out_of_range in the reference too (pixel coordinates never get there)
    F.save_tracks_to_file_native(np.array([0, 1]), np.array([(0, 1, 1, 1e-39, 2.0, 0, 0, 0)], capi.TRACK_FEATURE),
                                 tmp_path / "sub.txt")
    assert (tmp_path / "sub.txt").read_text() == "1;0;1;1;1e-39;2;0;0;0\n"
    with pytest.raises(capi.OsfmError) as e:
        F.load_tracks_from_file_native(tmp_path / "sub.txt")
    assert e.value.status == capi.E_IO
    (tmp_path / "cams.txt").write_text("a.png;1,2,3\n")
    with pytest.raises(capi.OsfmError) as e:
        F.import_camera_file_as_matrix_native(tmp_path / "cams.txt")
    assert e.value.status == capi.E_IO
    # capacity protocol of the reader: counts are reported, nothing is written
    (tmp_path / "ok.txt").write_text("1;0;1;2;3.5;4.5;7;8;9\n0;\n")
    import ctypes as C
    nt, nf = C.c_int64(), C.c_int64()
    st = capi.lib.osfm_tracks_file_read(os.fsencode(tmp_path / "ok.txt"), C.c_int64(0), C.c_int64(0), None, None,
                                        C.byref(nt), C.byref(nf))
    assert st == capi.E_CAPACITY and (nt.value, nf.value) == (2, 1)
    off, feats = F.load_tracks_from_file_native(tmp_path / "ok.txt")
    assert off.tolist() == [0, 1, 1] and feats[0]["b"] == 9 and feats[0]["x"] == 3.5
