"""GPU tests of the outlier filters through the C ABI,
against oracle/filter_oracle.c (PARITY UNPINNED w.r.t. the reference, whose
file needs Eigen -- see the oracle header).

nearest-neighbour distances: bit-exact (same subtraction, the same order of the
sum of squares, sqrt correctly rounded on both sides);
filterOutlierTracks: identical keep flags, mean / sigma bit-exact (the library
computes them on the host in the reference's sequential order);
reprojection filter: keep flags identical to the oracle's triangulation +
residuals wherever the error is not within 1e-9 px of the threshold."""
import numpy as np
import pytest

import oracle_lib
from orthosfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def flt():
    from orthosfm_amd import capi, filters
    assert capi.device_count() >= 1
    return filters


def cloud(n, seed, outliers=0):
    rng = np.random.default_rng(seed)
    p = np.concatenate([rng.normal(0, 1.0, (n, 3)), 1.0 + 0.1 * rng.normal(size=(n, 1))], axis=1)
    if outliers:
        p[:outliers, :3] *= 8.0
    return p


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 5000])
def test_nn_distances_bit_exact(flt, n):
    p = cloud(n, 10 + n)
    if n >= 256:
        p[7] = p[200]                      # a duplicate: distance 0 for both
    got = flt.nearest_neighbour_distance(p)
    want = oracle_lib.oracle_nn_distances(p)
    np.testing.assert_array_equal(got, want)


def test_nn_distances_empty_and_far(flt):
    assert flt.nearest_neighbour_distance(np.zeros((0, 4))).shape == (0,)
    far = np.array([[0.0, 0, 0, 1], [3e6, 0, 0, 1]])
    np.testing.assert_array_equal(flt.nearest_neighbour_distance(far), [1e6, 1e6])


def test_filter_outlier_tracks_matches_oracle(flt):
    p = cloud(6000, 3, outliers=100)
    p[5] = [30.0, 0, 0, 1]                 # outside the 10-unit box
    has = np.ones(6000, dtype=bool)
    has[::11] = False
    keep, st = flt.outlier_track_flags(p, has)
    ekeep, emean, esigma = oracle_lib.oracle_filter_outlier_tracks(p, has)
    np.testing.assert_array_equal(keep, ekeep)
    assert st.mean == emean and st.sigma == esigma
    assert st.num_with_point == int(has.sum()) and st.num_kept == int(ekeep.sum())
    assert not keep[5] and keep[~has].all()


def test_filter_outlier_tracks_mirror(flt):
    from orthosfm_amd.ba import Feature, Track
    p = cloud(800, 4, outliers=30)
    tracks = [Track([Feature(0, i, 0.0, 0.0), Feature(1, i, 0.0, 0.0)], p[i].copy(), i % 9 != 0) for i in range(800)]
    out = flt.filter_outlier_tracks(tracks, cameras=None, verbose=False)
    ekeep, _, _ = oracle_lib.oracle_filter_outlier_tracks(p, [t.has_point for t in tracks])
    assert [t.features[0].localFeatureID for t in out] == list(np.nonzero(ekeep)[0])


def test_full_size_nn_distances_properties(flt):
    """100k points (the global-BA size): distances are symmetric minima."""
    p = cloud(100000, 5)
    nn = flt.nearest_neighbour_distance(p)
    from scipy.spatial import cKDTree
    d, idx = cKDTree(p).query(p, k=2)
    np.testing.assert_allclose(nn, d[:, 1], rtol=1e-12)
    # the nearest neighbour's own nearest distance cannot be larger
    assert (nn[idx[:, 1]] <= nn * (1 + 1e-12)).all()


@pytest.mark.parametrize("model", [0, 1])
def test_reprojection_filter_flags(flt, model):
    import ctypes as C
    from orthosfm_amd import ba, capi
    sc = synth.make_ba_scene(model, 6, 600, config_id=41, min_len=6, max_len=6, noise_px=0.4,
                             rot_perturb_deg=0.0, off_perturb=0.0, point_perturb=0.0)
    rng = np.random.default_rng(9)
    bad = rng.choice(sc.obs_xy.shape[0], 150, replace=False)
    sc.obs_xy[bad] += rng.normal(0, 4.0, (150, 2)).astype(np.float32)     # gross outliers
    ref = sc.copy()
    evalid = oracle_lib.oracle_ba_triangulate(ref)            # ref.points <- triangulated
    _, eerr = oracle_lib.oracle_ba_residuals(ref)
    fp = ba.FlatProblem.from_scene(sc)
    st = fp.struct()
    keep = np.zeros(fp.obs_camera.shape[0], dtype=np.uint8)
    valid = np.zeros(fp.points.shape[0], dtype=np.uint8)
    err = np.zeros(fp.obs_camera.shape[0])
    capi.check(capi.lib.osfm_filter_reprojection(C.byref(st), 0, C.c_double(1.5), capi._ptr(keep, C.c_uint8),
                                                 capi._ptr(valid, C.c_uint8), capi._ptr(err, C.c_double)))
    assert np.array_equal(valid, evalid)
    assert np.abs(fp.points - ref.points).max() <= 1e-9
    assert np.abs(err - eerr).max() <= 1e-9
    clear = np.abs(eerr - 1.5) > 1e-9
    assert np.array_equal(keep[clear].astype(bool), (eerr < 1.5)[clear])
    assert 50 < (keep == 0).sum() < 600


def test_reprojection_filter_mirror(flt):
    """filterTracksWithReprojectionError: only tracks seen by every camera are
    judged; features of other views and all other tracks pass through."""
    from orthosfm_amd.ba import Feature, QuatCamera, Track
    sc = synth.make_ba_scene(0, 4, 300, config_id=42, min_len=4, max_len=4, noise_px=0.2, rot_perturb_deg=0.0,
                             off_perturb=0.0, point_perturb=0.0)
    cams = [QuatCamera(10 + c, 2048, 2048, sc.cam_params[c, :4].copy(), *sc.cam_params[c, 4:7]) for c in range(4)]
    tracks = []
    for j in range(300):
        ks = np.nonzero(sc.obs_point == j)[0]
        feats = [Feature(10 + int(sc.obs_camera[k]), j, float(sc.obs_xy[k, 0]), float(sc.obs_xy[k, 1])) for k in ks]
        tracks.append(Track(feats, sc.points[j].copy(), True))
    tracks[3].features[1].x += 4.0                       # one outlier feature
    tracks[5].features = tracks[5].features[:3]         # not full size: passes unchanged
    tracks[7].features.append(Feature(99, 7, 1.0, 2.0))  # extra view without a camera: kept
    for f in tracks[9].features[1:]:
        f.x += 40.0                                      # inconsistent track
    out = flt.filter_tracks_with_reprojection_error(tracks, cams, verbose=False)

    # expectation from the oracle: triangulate the full-size tracks, judge every feature
    from orthosfm_amd import ba
    full = [i for i, t in enumerate(tracks) if sum(f.viewID in range(10, 14) for f in t.features) == 4]
    work = [Track(tracks[i].features, tracks[i].point.copy(), True) for i in full]
    fp, _ = ba._flatten(cams, work)
    oracle_lib.oracle_ba_triangulate(fp)
    _, eerr = oracle_lib.oracle_ba_residuals(fp)
    assert np.abs(eerr - 1.5).min() > 1e-6               # no borderline feature in this scene
    expect, k = {}, 0
    for i, t in enumerate(tracks):
        if i not in full:
            expect[i] = [f.viewID for f in t.features]
            continue
        kept = []
        for f in t.features:
            if 10 <= f.viewID < 14:
                if eerr[k] < 1.5:
                    kept.append(f.viewID)
                k += 1
            else:
                kept.append(f.viewID)
        if len(kept) > 1:
            expect[i] = kept
    got = {t.features[0].localFeatureID if t.features[0].viewID != 99 else -1: [f.viewID for f in t.features]
           for t in out}
    got_by_track = {}
    for t in out:
        got_by_track[t.features[0].localFeatureID] = [f.viewID for f in t.features]
    # local ids equal the track index in this scene; a track whose first feature was dropped keeps its id
    assert got_by_track == {i: v for i, v in expect.items()}
    assert len(got_by_track[5]) == 3 and 99 in got_by_track[7]
    assert len(got_by_track[3]) < 4                      # the shifted feature (and what it drags along) went
    assert sum(len(v) == 4 for v in got_by_track.values()) >= 290
    del got
