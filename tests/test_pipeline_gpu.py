"""End-to-end run of everything the backend replaces, in the order the reference
pipeline calls it (src/matching/matching_mve.cpp:247-473, src/sfm/reconstruct.cpp:174-295):

    matching + geometric verification  ->  track building  ->  track conversion
    ->  triangulation  ->  reprojection filter  ->  bundle adjustment
    ->  outlier filter  ->  tracks.txt round trip

on one synthetic scene with known cameras and landmarks.  No oracle here (each
stage has its own parity tests); this checks that the stages fit together:
conventions of indices, pixel coordinates and camera models."""
import numpy as np
import pytest

from orthosfm_amd import synth

pytestmark = pytest.mark.gpu

V, F = 6, 2500


@pytest.fixture(scope="module")
def scene():
    return synth.make_image_set(V, F, config_id=31)


@pytest.mark.parametrize("matcher", ["exhaustive", "cascade"])
def test_pipeline(scene, matcher, tmp_path):
    from orthosfm_amd import ba, capi, filters, formats
    from orthosfm_amd.matching import HipCascadeHashing, HipExhaustiveMatching
    from orthosfm_amd.tracks import Tracks, Viewport

    W, H = scene.width, scene.height
    o = capi.default_match_options()
    o.geometric_verification = 1
    cls = HipExhaustiveMatching if matcher == "exhaustive" else HipCascadeHashing
    m = cls(V, options=o)
    norm = []
    for v in range(V):
        m.set_view(v, scene.sift[v])
        # normalize_feature_positions (feature_set.cc:42-55)
        xy = ((scene.pos[v] + 0.5 - np.array([W / 2, H / 2])) / max(W, H)).astype(np.float32)
        norm.append(xy)
        m.set_positions(v, xy)
    matching = [tv for tv in m.compute() if tv.status == capi.PAIR_MATCHED]
    m.close()
    assert len(matching) >= 10
    # verified matches connect features of the same landmark
    good = tot = 0
    for tv in matching:
        la = scene.landmark[tv.view_1_id][tv.matches[:, 0]]
        lb = scene.landmark[tv.view_2_id][tv.matches[:, 1]]
        good += int(((la == lb) & (la >= 0)).sum())
        tot += tv.matches.shape[0]
    assert good / tot > 0.995

    # track building
    viewports = [Viewport(F) for _ in range(V)]
    mve_tracks = Tracks().compute(matching, viewports)
    assert len(mve_tracks) > 800
    pure = sum(len({int(scene.landmark[v][f]) for v, f in t.features}) == 1 for t in mve_tracks)
    assert pure / len(mve_tracks) > 0.99

    # MVE tracks -> orthosfm tracks (pixel coordinates), tracks.txt round trip
    offs = np.concatenate([[0], np.cumsum([len(t.features) for t in mve_tracks])])
    feats = np.concatenate([t.features for t in mve_tracks])
    otr = formats.mve_tracks_to_orthosfm(offs, feats, [n - 0.0 for n in norm], W)
    formats.save_tracks_to_file(otr, tmp_path / "tracks.txt")
    back = formats.load_tracks_from_file(tmp_path / "tracks.txt")
    assert len(back) == len(otr) and back[0].features[0].globalFeatureID == otr[0].features[0].globalFeatureID
    # the conversion undoes the normalisation up to the half-pixel convention (matching_mve.cpp:463)
    f0 = otr[0].features[0]
    px = scene.pos[f0.viewID][f0.localFeatureID]
    assert abs(f0.x - (px[0] + 0.5 - W / 2 + W * 0.5)) < 1e-2

    # cameras: ground truth, perturbed except the first (reconstruct.cpp:215 keeps camera 0 fixed)
    rng = np.random.default_rng(5)
    cams = []
    for v, (phi, theta, rho) in enumerate(scene.cams):
        d = np.deg2rad(rng.normal(0, 0.5, 3)) if v else np.zeros(3)
        cams.append(ba.EulerCamera(v, W, H, phi + d[0], theta + d[1], rho + d[2], fixed=(v == 0)))
    # the reference pixel (x + 0.5 shift) is what the features carry; observations for BA
    tracks = [ba.Track([ba.Feature(f.viewID, f.localFeatureID, f.x - 0.5, f.y - 0.5) for f in t.features])
              for t in otr]
    ba.triangulate_tracks(cams, tracks, True)
    assert sum(t.has_point for t in tracks) == len(tracks)
    tracks = filters.filter_tracks_with_reprojection_error(tracks, cams, verbose=False, max_error=50.0)
    s = ba.run_bundle_adjustment(cams, tracks, optimize_points=True, verbose=False)
    assert s.final_cost < 1e-3 * s.initial_cost
    # cameras are back at the ground truth (gauge fixed by camera 0)
    for cam, (phi, theta, rho) in zip(cams, scene.cams):
        assert abs(cam.phi - phi) < 2e-4 and abs(cam.theta - theta) < 2e-4 and abs(cam.roll - rho) < 2e-4
    kept = filters.filter_outlier_tracks(tracks, cams, verbose=False)
    assert 0.7 * len(tracks) < len(kept) <= len(tracks)
    # reconstructed points sit on their landmarks
    err = []
    for t in kept[:300]:
        lm = int(scene.landmark[t.features[0].viewID][t.features[0].localFeatureID])
        p = np.asarray(t.point)
        err.append(np.linalg.norm(p[:3] / p[3] - scene.points[lm]))
    assert np.median(err) < 1e-3
