"""GPU tests of the group ordering (orthosfm::buildGroups)
against oracle/groups_oracle.c -- a literal restatement that re-filters the
track list for every score, PARITY UNPINNED w.r.t. the reference (see its
header).  Integer work: identical groups and track counts."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def random_tracks(num_views, num_tracks, seed, ring=True, view_ids=None):
    """Tracks over contiguous arcs of a camera ring (like the synthetic scenes),
    plus a few random ones."""
    rng = np.random.default_rng(seed)
    ids = np.arange(num_views) if view_ids is None else np.asarray(view_ids)
    tracks = []
    for _ in range(num_tracks):
        if ring and rng.random() < 0.85:
            ln = min(int(rng.integers(2, 7)), num_views)      # no view twice in a track (such tracks
                                                               # are removed by Tracks::remove_invalid_tracks)
            st = int(rng.integers(0, num_views))
            vs = [(st + k) % num_views for k in range(ln)]
        else:
            vs = list(rng.choice(num_views, int(rng.integers(2, 5)), replace=False))
        tracks.append([int(ids[v]) for v in vs])
    return tracks


@pytest.mark.parametrize("num_views,num_tracks,gs,seed", [(6, 300, 3, 1), (12, 2500, 3, 2), (9, 1500, 4, 3),
                                                            (20, 6000, 3, 4)])
def test_groups_match_oracle(num_views, num_tracks, gs, seed):
    from orthosfm_amd import groups as G
    view_ids = np.arange(num_views)
    tracks = random_tracks(num_views, num_tracks, seed)
    offs, views = G.flatten_tracks(tracks)
    want = oracle_lib.oracle_build_groups(view_ids, offs, views, gs)
    assert want is not None
    got = G.build_groups(view_ids, tracks, gs)
    assert [g.ids for g in got] == [list(map(int, r)) for r in want[0]]
    assert [g.tracks for g in got] == list(map(int, want[1]))
    # every view ends up in a group, the first group starts with views 0 and 1
    assert got[0].ids[:2] == [0, 1]
    assert sorted({v for g in got for v in g.ids}) == list(range(num_views))


def test_shuffled_view_ids_and_mirror_types():
    from orthosfm_amd import groups as G
    from orthosfm_amd.ba import Feature, Track
    ids = np.array([40, 7, 23, 5, 99, 12, 64, 3])
    raw = random_tracks(8, 1200, 7, view_ids=ids)
    tracks = [Track([Feature(v, i, 0.0, 0.0) for v in t]) for i, t in enumerate(raw)]
    offs, views = G.flatten_tracks(tracks)
    want = oracle_lib.oracle_build_groups(ids, offs, views, 3)
    got = G.build_groups(ids, tracks, 3)
    assert [g.ids for g in got] == [list(map(int, r)) for r in want[0]]
    assert got[0].ids[:2] == [40, 7]


def test_unreachable_view_is_reported():
    """A view without any shared track: the reference never terminates; here an error."""
    from orthosfm_amd import capi, groups as G
    tracks = random_tracks(5, 400, 11)
    assert oracle_lib.oracle_build_groups(np.arange(6), *G.flatten_tracks(tracks), 3) is None
    with pytest.raises(capi.OsfmError):
        G.build_groups(np.arange(6), tracks, 3)              # view 5 appears in no track


def test_scale_500_views():
    """BASELINE config 5 size: 500 views, 100k tracks -- seconds, not hours."""
    import time
    from orthosfm_amd import groups as G
    tracks = random_tracks(500, 100000, 21)
    t0 = time.perf_counter()
    got = G.build_groups(np.arange(500), tracks, 3)
    dt = time.perf_counter() - t0
    assert sorted({v for g in got for v in g.ids}) == list(range(500))
    assert len(got) == 498 and dt < 120.0
    print(f"500 views / 100k tracks: {len(got)} groups in {dt:.1f} s")
