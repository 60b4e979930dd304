"""Host logic of the end-to-end driver (orthosfm_amd/pipeline.py) that needs no GPU:
the canonical ground-truth frame, the camera-model conversions, the alignment of a
local camera triple to the global cameras and the track table."""
import numpy as np
import pytest

from orthosfm_amd import ba as B
from orthosfm_amd import pipeline as P
from orthosfm_amd import synth


@pytest.fixture(scope="module")
def iset():
    return synth.make_image_set(5, 300, config_id=61)


@pytest.mark.parametrize("model", [0, 1])
def test_canonical_ground_truth_projects_onto_the_track_pixels(iset, model):
    """Camera 0 is canonical (identity / zero angles) and every camera projects the
    landmarks onto feature position + 0.5 px -- the pixel the converted tracks carry
    (matching_mve.cpp:463)."""
    gt, pts = P.canonical_ground_truth(iset, model)
    if model == 0:
        assert np.allclose(np.abs(gt[0, :4]), [0, 0, 0, 1], atol=1e-12)
    else:
        assert np.allclose(gt[0, :3], 0, atol=1e-12)
    for v in range(iset.num_views):
        lm = iset.landmark[v]
        sel = lm >= 0
        if model == 0:
            xy = synth.project_quat(pts[lm[sel]], gt[v, :4], gt[v, 4], gt[v, 5], gt[v, 6], iset.width, iset.height)
        else:
            xy = synth.project_euler(pts[lm[sel]], gt[v, 0], gt[v, 1], gt[v, 2], gt[v, 3], gt[v, 4], gt[v, 5],
                                     iset.width, iset.height)
        assert np.abs(xy - (iset.pos[v][sel].astype(np.float64) + 0.5)).max() < 2e-3     # float32 positions


def test_euler_angles_round_trip():
    rng = np.random.default_rng(3)
    for _ in range(50):
        phi, theta, rho = rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3)
        S = synth.euler_matrix(phi, theta, rho)
        a = P._euler_from_S(S)
        assert np.allclose(synth.euler_matrix(*a), S, atol=1e-12)


@pytest.mark.parametrize("model", [0, 1])
def test_align_to_global_undoes_a_frame_rotation(iset, model):
    """Rotate the three local cameras by one common rotation (the gauge freedom of a local
    bundle adjustment with all cameras free): aligning to the two shared global cameras
    brings the third one back."""
    gt, _ = P.canonical_ground_truth(iset, model)
    glob = [gt[1].copy(), gt[2].copy(), None]
    axis = np.array([0.3, -0.5, 0.8]); axis /= np.linalg.norm(axis)
    ang = 0.2
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rot = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    local = [gt[v].copy() for v in (1, 2, 3)]
    for p in local:
        P._set_cam_rotation(model, p, Rot @ P._cam_rotation(model, p))
    P.align_to_global(model, local, glob)
    assert np.allclose(P._cam_rotation(model, local[2]), P._cam_rotation(model, gt[3]), atol=1e-10)


def test_track_table_indexing():
    offs = np.array([0, 2, 5, 7])
    view = np.array([0, 2, 1, 2, 3, 0, 3])
    feat = np.array([5, 1, 2, 0, 4, 6, 3])
    tt = P.TrackTable(offs, view, feat, np.zeros((7, 2)), 4)
    assert tt.track_of.tolist() == [0, 0, 1, 1, 1, 2, 2]
    assert tt.features_of_views([2, 3]).tolist() == [1, 3, 4, 6]
    tt.kill(features=[3])
    assert tt.features_of_views([2, 3]).tolist() == [1, 4, 6]
    tt.kill(tracks=[2])
    assert tt.features_of_views([0, 3]).tolist() == [0, 4]
    assert tt.alive_lengths().tolist() == [2, 2, 0] and tt.num_tracks == 2
    # the flags kill() keeps beside the two alive arrays
    assert tt.live_f.tolist() == [True, True, True, False, True, False, False]
    tt.kill(features=[3, 4], tracks=[2])                       # killing the dead again changes nothing
    assert tt.alive_lengths().tolist() == [2, 1, 0]
    tt.align_view(2, 0); tt.align_view(0, 1)
    assert tt.cam_f.tolist() == [1, 0, -1, 0, -1, 1, -1]
    norm = [np.array([[0.1, -0.2]] * 7, np.float32)] * 4
    t2 = P.TrackTable.from_mve(offs, np.stack([view, feat], 1), norm, 100, 4)
    assert np.allclose(t2.xy[0], [np.float32(100 * (np.float64(np.float32(0.1)) + 0.5)),
                                  np.float32(100 * (np.float64(np.float32(-0.2)) + 0.5))])


def test_track_table_compaction_round_trip():
    """compacted() keeps the live tracks / features in order; selections on the working table name
    the same features (through orig_feat) as on the full one; write_back() restores the flags,
    points and lengths -- also through two compactions."""
    r = np.random.default_rng(5)
    lens = r.integers(2, 6, 40)
    offs = np.concatenate([[0], np.cumsum(lens)])
    n = int(offs[-1])
    view = np.concatenate([r.choice(8, l, replace=False) for l in lens])
    tt = P.TrackTable(offs, view, r.integers(0, 100, n), r.normal(size=(n, 2)), 8)
    ref = P.TrackTable(offs, view, tt.feat.copy(), tt.xy.copy(), 8)
    for v in (1, 4, 6):
        tt.align_view(v, v); ref.align_view(v, v)
    work = tt
    for round_ in range(2):
        kt = r.choice(np.flatnonzero(work.alive_t), 8, replace=False)
        kf = r.choice(np.flatnonzero(work.live_f), 10, replace=False)
        ot = getattr(work, "orig_track", np.arange(40))
        of = getattr(work, "orig_feat", np.arange(n))
        work.kill(tracks=kt, features=kf)
        ref.kill(tracks=ot[kt], features=of[kf])
        work = work.compacted()
        assert work.alive_t.all() and work.live_f.all()
        assert np.array_equal(work.orig_feat, np.flatnonzero(ref.live_f))
        assert np.array_equal(work.orig_track, np.flatnonzero(ref.alive_t))
        assert np.array_equal(work.offsets[1:] - work.offsets[:-1], ref.alive_lengths()[ref.alive_t])
        for views in ([1, 4], [6], [0, 2, 7]):
            assert np.array_equal(work.orig_feat[work.features_of_views(views)], ref.features_of_views(views))
        assert np.array_equal(work.cam_f, ref.cam_f[work.orig_feat])
    work.point[:] = r.normal(size=work.point.shape)
    work.has_point[::2] = True
    tt.write_back(work)
    assert np.array_equal(tt.alive_t, ref.alive_t) and np.array_equal(tt.live_f, ref.live_f)
    assert np.array_equal(tt.alive_lengths(), ref.alive_lengths())
    assert np.array_equal(tt.point[work.orig_track], work.point) and tt.has_point.sum() == work.has_point.sum()
    assert not tt.has_point[~tt.alive_t].any()


def test_const_masks_follow_the_solver():
    assert P.default_const_mask(B.MODEL_QUATERNION).tolist() == [0, 0, 0, 0, 0, 0, 1]
    assert P.default_const_mask(B.MODEL_EULER, euler_dof=P.euler_dof_of_solver(3)).tolist() == [0, 0, 0, 0, 0, 1, 1]
    assert P.default_const_mask(B.MODEL_EULER, euler_dof=P.euler_dof_of_solver(1)).tolist() == [0, 1, 1, 1, 1, 1, 1]
    assert P.default_const_mask(B.MODEL_EULER, fixed=True).all()
