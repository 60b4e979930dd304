"""Track building: sfm::bundler::Tracks::compute.

Host code on both sides, so everything here runs without a GPU:
  * the oracle (oracle/tracks_oracle.c) is PINNED against the reference's own
    bundler_tracks.cc (oracle/_ref/libref_tracks.so, live and randomised, when
    /root/reference was present at build time) and against golden vectors the
    reference produced (tests/golden/tracks_*.npz, generator committed);
  * the product (osfm_tracks_compute, linked lists instead of per-track
    vectors) is compared with goldens and oracle element for element: track
    order, feature order inside tracks, per-feature track ids, colours."""
import os

import numpy as np
import pytest

import oracle_lib
import track_cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("track_ids", "track_offsets", "track_features", "track_colors")


def product(m):
    from orthosfm_amd import capi, tracks
    pairs = (capi.Pair * max(len(m["pairs"]), 1))()
    for i, (a, b) in enumerate(m["pairs"]):
        pairs[i].view_1, pairs[i].view_2 = int(a), int(b)
    ids, toff, tfeat, tcol, summary = tracks.compute_flat(m["view_sizes"], m["colors"], pairs, m["pair_offsets"],
                                                           np.ascontiguousarray(m["corr"], dtype=np.int32))
    return {"track_ids": ids, "track_offsets": toff, "track_features": tfeat, "track_colors": tcol,
            "num_invalid": summary.num_invalid_tracks}


def same(a, b):
    for k in KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("name", sorted(track_cases.CASES))
def test_oracle_and_product_match_reference_golden(name):
    g = np.load(os.path.join(GOLDEN, f"tracks_{name}.npz"))
    m = {k: g[k] for k in ("view_sizes", "colors", "pairs", "pair_offsets", "corr")}
    want = {k: g["out_" + k] for k in KEYS}
    # the golden inputs are what the generator builds today
    regen = track_cases.random_matching(**track_cases.CASES[name])
    for k in m:
        np.testing.assert_array_equal(m[k], regen[k])
    same(oracle_lib.oracle_tracks(**m), want)
    same(product(m), want)


@pytest.mark.skipif(oracle_lib.ref_tracks() is None, reason="oracle/_ref/libref_tracks.so not built")
@pytest.mark.parametrize("seed", range(12))
def test_oracle_pinned_against_reference_live(seed):
    rng = np.random.default_rng(100 + seed)
    m = track_cases.random_matching(num_views=int(rng.integers(2, 12)), feats_per_view=int(rng.integers(5, 150)),
                                    num_scene_points=int(rng.integers(5, 250)), p_seen=float(rng.uniform(0.2, 0.9)),
                                    p_false=float(rng.choice([0.0, 0.05, 0.3])), seed=seed)
    ref = oracle_lib.ref_tracks_compute(**m)
    orc = oracle_lib.oracle_tracks(**m)
    same(orc, ref)
    got = product(m)
    same(got, ref)
    assert got["num_invalid"] == orc["num_invalid"]


def test_edge_cases():
    from orthosfm_amd import capi
    empty = {"view_sizes": np.array([3, 4], np.int32), "colors": None, "pairs": np.zeros((0, 2), np.int32),
             "pair_offsets": np.zeros(1, np.int64), "corr": np.zeros((0, 2), np.int32)}
    out = product(empty)
    assert (out["track_ids"] == -1).all() and len(out["track_offsets"]) == 1
    # a chain 0-1, 1-2 and a conflicting second feature of view 0: the merged track is invalid
    m = {"view_sizes": np.array([2, 1, 1], np.int32), "colors": np.arange(12, dtype=np.uint8).reshape(4, 3),
         "pairs": np.array([[1, 0], [2, 1], [2, 0]], np.int32), "pair_offsets": np.array([0, 1, 2, 3], np.int64),
         "corr": np.array([[0, 0], [0, 0], [0, 1]], np.int32)}
    out, orc = product(m), oracle_lib.oracle_tracks(**m)
    same(out, orc)
    assert out["num_invalid"] == 1 and len(out["track_offsets"]) == 1 and (out["track_ids"] == -1).all()
    # a draw in unify_tracks keeps the FIRST track's features in front (bundler_tracks.cc:28-31)
    m = {"view_sizes": np.array([1, 1, 1, 1], np.int32), "colors": None,
         "pairs": np.array([[1, 0], [3, 2], [2, 1]], np.int32), "pair_offsets": np.array([0, 1, 2, 3], np.int64),
         "corr": np.zeros((3, 2), np.int32)}
    out = product(m)
    same(out, oracle_lib.oracle_tracks(**m))
    np.testing.assert_array_equal(out["track_features"][:, 0], [3, 2, 1, 0])   # view1_tid = track of view 2
    # out-of-range input is refused, not read
    bad = dict(m, corr=np.array([[0, 0], [0, 0], [5, 0]], np.int32))
    with pytest.raises(capi.OsfmError):
        product(bad)


def test_mirror_class_and_scale():
    from orthosfm_amd.matching import TwoViewMatching
    from orthosfm_amd.tracks import Tracks, Viewport
    m = track_cases.random_matching(num_views=30, feats_per_view=4000, num_scene_points=6000, p_seen=0.4, seed=9)
    viewports = [Viewport(4000, m["colors"][v * 4000:(v + 1) * 4000]) for v in range(30)]
    matching = [TwoViewMatching(int(a), int(b), m["corr"][m["pair_offsets"][i]:m["pair_offsets"][i + 1]])
                for i, (a, b) in enumerate(m["pairs"])]
    tracks = Tracks().compute(matching, viewports)
    orc = oracle_lib.oracle_tracks(**m)
    assert len(tracks) == len(orc["track_offsets"]) - 1 > 1000
    np.testing.assert_array_equal(np.concatenate([vp.track_ids for vp in viewports]), orc["track_ids"])
    np.testing.assert_array_equal(np.concatenate([t.features for t in tracks]), orc["track_features"])
    np.testing.assert_array_equal(np.stack([t.color for t in tracks]), orc["track_colors"])


def test_ranges_entry_matches_packed_entry():
    """osfm_tracks_compute_ranges on lists scattered through a larger buffer gives
    the result of osfm_tracks_compute on the packed lists."""
    from orthosfm_amd import capi, tracks
    m = track_cases.random_matching(7, 60, 150, seed=11)
    pairs = (capi.Pair * len(m["pairs"]))()
    for i, (a, b) in enumerate(m["pairs"]):
        pairs[i].view_1, pairs[i].view_2 = int(a), int(b)
    m["corr"] = np.ascontiguousarray(m["corr"], dtype=np.int32)
    ref = tracks.compute_flat(m["view_sizes"], m["colors"], pairs, m["pair_offsets"], m["corr"])
    offs = np.asarray(m["pair_offsets"], dtype=np.int64)
    counts = np.diff(offs)
    # lay the lists out in reverse pair order with gaps of garbage between them
    big = np.full((int(counts.sum()) + 5 * len(counts) + 3, 2), -77, dtype=np.int32)
    starts = np.zeros(len(counts), dtype=np.int64)
    pos = 3
    for p in range(len(counts) - 1, -1, -1):
        starts[p] = pos
        big[pos:pos + counts[p]] = m["corr"][offs[p]:offs[p + 1]]
        pos += counts[p] + 5
    got = tracks.compute_flat_ranges(m["view_sizes"], m["colors"], pairs, starts, counts, big)
    for a, b in zip(ref[:4], got[:4]):
        assert np.array_equal(a, b)
    assert (ref[4].num_tracks, ref[4].num_features) == (got[4].num_tracks, got[4].num_features)
    # a negative count is an argument error
    bad = counts.copy()
    bad[0] = -1
    with pytest.raises(capi.OsfmError):
        tracks.compute_flat_ranges(m["view_sizes"], m["colors"], pairs, starts, bad, big)


def test_self_match_is_an_invalid_track_not_a_hang():
    """A pair with view_1 == view_2 whose list matches a feature to itself makes the
    reference push the same FeatureReference twice (bundler_tracks.cc:80-86): the
    track is dropped as conflicting, whatever is merged into it later.  Checked
    against the oracle (and the reference build when present)."""
    m = {"view_sizes": np.array([6, 5, 4], np.int32), "colors": None,
         "pairs": np.array([[1, 1], [1, 0], [2, 1], [2, 0]], np.int32),
         "pair_offsets": np.array([0, 2, 4, 6, 7], np.int64),
         "corr": np.array([[3, 3], [0, 1],          # (1,1): self-match of feature 3; 0 <-> 1 inside view 1
                           [3, 2], [4, 4],          # (1,0): the poisoned track grows; a clean track
                           [1, 3], [2, 4],          # (2,1): joins the poisoned track; joins the clean one
                           [0, 5]], np.int32)}
    got = product(m)
    want = oracle_lib.oracle_tracks(**m)
    same(got, want)
    assert got["track_ids"][6 + 3] == -1 and got["num_invalid"] == 2
    assert got["track_offsets"].shape[0] - 1 == 2
    if oracle_lib.ref_tracks() is not None:
        same(oracle_lib.ref_tracks_compute(**m), want)


def test_tracks_builder_equals_one_shot():
    """osfm_tracks_builder_*: the merge fed in several batches (pair order kept) yields exactly what
    osfm_tracks_compute yields on the whole list; finish() may be called between feeds."""
    from orthosfm_amd import capi, tracks as T
    r = np.random.default_rng(9)
    V = 7
    sizes = r.integers(30, 60, V).astype(np.int32)
    pairs, lists = [], []
    for a in range(1, V):
        for b in range(a):
            k = int(r.integers(0, 25))
            f1 = r.choice(sizes[a], k, replace=False)
            f2 = r.choice(sizes[b], k, replace=False)
            pairs.append((a, b)); lists.append(np.stack([f1, f2], 1).astype(np.int32))
    offs = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
    corr = np.concatenate(lists)
    parr = (capi.Pair * len(pairs))()
    for i, (a, b) in enumerate(pairs):
        parr[i].view_1, parr[i].view_2 = a, b
    want = T.compute_flat(sizes, None, parr, offs, corr)
    rec = np.zeros(len(pairs), dtype=[("status", np.int32), ("lowres_matches", np.int32), ("num_matches", np.int32),
                                      ("num_inliers", np.int32), ("offset", np.int64)])
    rec["status"] = capi.PAIR_MATCHED
    rec["num_inliers"] = -1
    rec["num_matches"] = np.diff(offs)
    b = T.TracksBuilder(sizes)
    cuts = [0, 5, 6, 14, len(pairs)]
    pf = np.array(pairs, np.int32)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sub = rec[lo:hi].copy()
        base = int(offs[lo])
        sub["offset"] = offs[lo:hi] - base                      # a batch's offsets are relative to its own buffer
        b.feed(pf[lo:hi], sub, np.ascontiguousarray(corr[base:int(offs[hi])]))
        if hi == 6:
            b.finish()                                            # a look at the state changes nothing
    got = b.finish()
    for x, y in zip(got[:4], want[:4]):
        assert np.array_equal(x, y)
    assert got[4].num_tracks == want[4].num_tracks and got[4].num_invalid_tracks == want[4].num_invalid_tracks
    assert b.num_pairs == len(pairs) and b.num_matches == int(offs[-1])
    b.close()


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_finish_is_the_same_on_any_number_of_threads(threads, monkeypatch):
    """The output walk of osfm_tracks_compute runs on several host threads (ranges of tracks laid end to
    end): same arrays as the oracle for 1, 3 and 8 of them, invalid tracks and colours included."""
    monkeypatch.setenv("OSFM_TRACKS_THREADS", str(threads))
    m = track_cases.random_matching(num_views=11, feats_per_view=140, num_scene_points=220, p_seen=0.6, p_false=0.2, seed=23)
    got, want = product(m), oracle_lib.oracle_tracks(**m)
    same(got, want)
    assert got["num_invalid"] == want["num_invalid"] and want["num_invalid"] > 0


@pytest.mark.parametrize("threads", [1, 5])
def test_feature_table_equals_numpy(threads, monkeypatch):
    """osfm_tracks_feature_table against the array arithmetic it replaces (matching_mve.cpp:455-466: pixel =
    float(imageWidth * (double(normalised) + 0.5)) for both axes), the features grouped by view in a stable
    order; a feature outside its view is refused."""
    from orthosfm_amd import capi, tracks as T
    monkeypatch.setenv("OSFM_TRACKS_THREADS", str(threads))
    r = np.random.default_rng(4)
    V = 6
    sizes = r.integers(5, 40, V)
    pos = [r.uniform(-0.5, 0.5, (n, 2)).astype(np.float32) for n in sizes]
    lens = r.integers(2, V + 1, 200)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tf = np.zeros((int(offs[-1]), 2), np.int32)
    for t in range(200):
        vs = r.choice(V, lens[t], replace=False)
        tf[offs[t]:offs[t + 1], 0] = vs
        tf[offs[t]:offs[t + 1], 1] = [r.integers(0, sizes[v]) for v in vs]
    view, feat, xy, track_of, by_view, view_start = T.feature_table(offs, tf.reshape(-1), pos, 2048)
    voff = np.concatenate([[0], np.cumsum(sizes)])
    p = np.concatenate(pos)[voff[tf[:, 0]] + tf[:, 1]].astype(np.float64)
    want = (2048.0 * (p + 0.5)).astype(np.float32).astype(np.float64)
    assert np.array_equal(view, tf[:, 0]) and np.array_equal(feat, tf[:, 1]) and np.array_equal(xy, want)
    assert np.array_equal(track_of, np.repeat(np.arange(200), lens))
    assert np.array_equal(by_view, np.argsort(tf[:, 0], kind="stable"))
    assert np.array_equal(view_start, np.concatenate([[0], np.cumsum(np.bincount(tf[:, 0], minlength=V))]))
    tf[7, 1] = sizes[tf[7, 0]]
    with pytest.raises(capi.OsfmError):
        T.feature_table(offs, tf.reshape(-1), pos, 2048)
    tf[7, 1] = 0
    bad = offs.copy(); bad[3] = bad[5] + 1                     # offsets that go down again
    with pytest.raises(capi.OsfmError):
        T.feature_table(bad, tf.reshape(-1), pos, 2048)
    # no tracks at all
    view, feat, xy, track_of, by_view, view_start = T.feature_table(np.zeros(1, np.int64), np.zeros(0, np.int32), pos, 2048)
    assert view.size == 0 and np.array_equal(view_start, np.zeros(V + 1, np.int64))


def test_select_observations_equals_numpy():
    """osfm_tracks_select_observations against the index arithmetic it replaces: live features whose view has
    a camera, optionally of masked tracks; points numbered by a slot table or densely in order of appearance."""
    from orthosfm_amd import tracks as T
    r = np.random.default_rng(3)
    lens = r.integers(1, 7, 300)
    track_of = np.repeat(np.arange(300, dtype=np.int32), lens)
    n = track_of.shape[0]
    cam_f = r.integers(-1, 5, n).astype(np.int32)
    live = r.random(n) < 0.8
    xy = r.normal(size=(n, 2))
    mask = r.random(300) < 0.5
    # dense numbering, no mask / with mask
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    # (with the track offsets the pass goes track by track, without them feature by feature)
    for m in (None, mask):
        for o in (None, offs):
            sel = live & (cam_f >= 0) & (True if m is None else m[track_of])
            ids = np.flatnonzero(sel)
            uniq, inv = np.unique(track_of[ids], return_inverse=True)
            oxy, ocam, opt, tracks, fids = T.select_observations(track_of, cam_f, live, xy, track_mask=m, want_features=True,
                                                                 track_offsets=o)
            assert np.array_equal(fids, ids) and np.array_equal(oxy, xy[ids]) and np.array_equal(ocam, cam_f[ids])
            assert np.array_equal(tracks, uniq) and np.array_equal(opt, inv)
    # the caller's numbering
    slot = (np.cumsum(mask) - 1).astype(np.int32)
    sel = live & (cam_f >= 0) & mask[track_of]
    for o in (None, offs):
        oxy, ocam, opt, tracks, _ = T.select_observations(track_of, cam_f, live, xy, track_mask=mask, track_slot=slot, track_offsets=o)
        assert tracks is None and np.array_equal(opt, slot[track_of[sel]]) and np.array_equal(oxy, xy[sel])
    # nothing selected
    oxy, ocam, opt, tracks, _ = T.select_observations(track_of, np.full(n, -1, np.int32), live, xy)
    assert oxy.shape == (0, 2) and tracks.size == 0
