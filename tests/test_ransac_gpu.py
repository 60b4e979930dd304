"""GPU tests of the geometric verification row (RANSAC-F) through the C ABI,
against the CPU oracle: same counter-based sample stream and the same double
arithmetic in the same order, so inlier sets and F agree bit for bit."""
import numpy as np
import pytest

import oracle_lib
from test_oracle_ransac import two_view_scene
from orthosfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hm():
    from orthosfm_amd import capi
    from orthosfm_amd.matching import HipExhaustiveMatching
    assert capi.device_count() >= 1
    return HipExhaustiveMatching


@pytest.mark.parametrize("n,frac,seed", [(1500, 0.35, 4), (300, 0.1, 5), (9, 0.0, 6), (5000, 0.6, 7)])
def test_ransac_bit_exact_vs_oracle(hm, n, frac, seed):
    pos1, pos2, corr, _ = two_view_scene(n, frac, seed)
    for pair_id in (0, 17):
        en, einl, eF = oracle_lib.oracle_ransac(pos1, pos2, corr, seed=11, pair_id=pair_id)
        gn, ginl, gF = hm.ransac_fundamental(pos1, pos2, corr, seed=11, pair_id=pair_id)
        assert gn == en
        assert np.array_equal(ginl, einl)
        assert np.array_equal(gF, eF)


def test_ransac_edge_cases(hm):
    pos1, pos2, corr, _ = two_view_scene(50, 0.0, 8)
    n, inl, _ = hm.ransac_fundamental(pos1, pos2, corr[:7])
    assert n == -1                                  # the reference throws below 8 matches
    n, inl, _ = hm.ransac_fundamental(pos1, pos2, corr[:8], max_iterations=16)
    en, einl, _ = oracle_lib.oracle_ransac(pos1, pos2, corr[:8], max_iterations=16)
    assert n == en and np.array_equal(inl, einl)
    # more iterations than one pass of the kernel (1024)
    pos1, pos2, corr, _ = two_view_scene(400, 0.5, 9)
    n, inl, F = hm.ransac_fundamental(pos1, pos2, corr, max_iterations=2500, seed=3, pair_id=5)
    en, einl, eF = oracle_lib.oracle_ransac(pos1, pos2, corr, max_iterations=2500, seed=3, pair_id=5)
    assert n == en and np.array_equal(inl, einl) and np.array_equal(F, eF)


def test_match_all_with_geometric_verification(hm):
    """bundler::Matching::compute end to end on a small synthetic set: low-res
    gate, matching, thresholds, RANSAC-F, inlier threshold -- every stage
    against the oracle chain."""
    from orthosfm_amd import capi
    V, F = 5, 1500
    iset = synth.make_image_set(V, F, config_id=19)
    o = capi.default_match_options()
    o.geometric_verification = 1
    o.ransac_seed = 42
    m = hm(V, options=o)
    norm = []
    for v in range(V):
        m.set_view(v, iset.sift[v])
        xy = ((iset.pos[v] + 0.5 - np.array([iset.width / 2, iset.height / 2])) / max(iset.width, iset.height))
        norm.append(xy.astype(np.float32))
        m.set_positions(v, norm[v])
    out = m.compute()
    empty = np.zeros((0, 64), np.int16)
    n_matched = 0
    for tv in out:
        a, b = tv.view_1_id, tv.view_2_id
        e12, _ = oracle_lib.oracle_pairwise_match(iset.sift[a], empty, iset.sift[b], empty)
        idx = np.nonzero(e12 >= 0)[0]
        corr = np.stack([idx, e12[idx]], axis=1).astype(np.int32)
        assert tv.num_matches == corr.shape[0]
        if corr.shape[0] < 50:
            assert tv.status == capi.PAIR_REJECTED_COUNT
            continue
        en, einl, _ = oracle_lib.oracle_ransac(norm[a], norm[b], corr, seed=42, pair_id=a * (a - 1) // 2 + b)
        assert tv.num_inliers == en
        if en < 30:
            assert tv.status == capi.PAIR_REJECTED_INLIERS
        else:
            assert tv.status == capi.PAIR_MATCHED
            assert np.array_equal(tv.matches, corr[einl])
            # the planted geometry: inliers are true landmark correspondences
            lm_a, lm_b = iset.landmark[a][tv.matches[:, 0]], iset.landmark[b][tv.matches[:, 1]]
            assert (lm_a == lm_b).mean() > 0.99
            n_matched += 1
    assert n_matched >= 5
    # without positions the verification must refuse to run
    m2 = hm(V, options=o)
    for v in range(V):
        m2.set_view(v, iset.sift[v])
    with pytest.raises(capi.OsfmError) as e:
        m2.compute()
    assert e.value.status == capi.E_STATE
    m.close()
    m2.close()


def test_prefilter_decisions_are_the_double_ones(hm):
    """The scoring loop pre-classifies its Sampson tests in packed single precision and only
    lets a float result count when it is out of reach of its error bound.  Mode 2 compares
    every such decision with the double path on the device: none may differ, on benign scenes
    and on ones built to sit at the edges of the bound (coordinates at +-1, at 1e-4 scale, a
    threshold at the noise level, and coordinates above 1, which must switch the chunk to the
    double path).  Results equal those of the double-only mode, bit for bit."""
    from orthosfm_amd import capi
    scenes = []
    for n, frac, seed in ((3000, 0.3, 21), (800, 0.0, 22), (6000, 0.7, 23)):
        scenes.append((*two_view_scene(n, frac, seed)[:3], 0.0015, f"scene {seed}"))
    p1, p2, corr, _ = two_view_scene(2000, 0.2, 24)
    s = 1.0 / max(np.abs(p1).max(), np.abs(p2).max())
    scenes.append(((p1 * s).astype(np.float32), (p2 * s).astype(np.float32), corr, 0.0015 * s, "stretched to +-1"))
    scenes.append(((p1 * 1e-4).astype(np.float32), (p2 * 1e-4).astype(np.float32), corr, 0.0015e-4, "scale 1e-4"))
    scenes.append((p1, p2, corr, 2e-5, "threshold at the noise level"))
    scenes.append(((p1 * 3.0).astype(np.float32), (p2 * 3.0).astype(np.float32), corr, 0.0045, "coordinates above 1"))
    try:
        for pos1, pos2, c, thr, what in scenes:
            capi.ransac_selfcheck(0)
            ref = hm.ransac_fundamental(pos1, pos2, c, seed=5, pair_id=3, threshold=thr)
            capi.ransac_selfcheck(2)
            got = hm.ransac_fundamental(pos1, pos2, c, seed=5, pair_id=3, threshold=thr)
            wrong, undecided, tests = capi.ransac_selfcheck(1)
            assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), what
            assert wrong == 0, (what, wrong, tests)
            if what == "coordinates above 1":
                assert tests == 0                      # every chunk went the double way
            elif what == "scale 1e-4":
                assert tests > 0                       # the bound (built for |x| <= 1) is loose here: mostly undecided
            else:
                assert tests > 1000 * 0.5 * len(c) and undecided < 0.02 * tests, (what, undecided, tests)
            en, einl, eF = oracle_lib.oracle_ransac(pos1, pos2, c, seed=5, pair_id=3, threshold=thr)
            assert got[0] == en and np.array_equal(got[1], einl) and np.array_equal(got[2], eF), what
    finally:
        capi.ransac_selfcheck(1)
