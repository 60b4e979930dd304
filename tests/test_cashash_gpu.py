"""GPU tests of the cascade-hashing mode (sfm::CascadeHashing, the
application's default matcher) through the C ABI, against the oracle
(oracle/cashash_oracle.c, pinned bit-exact to the reference's own
cascade_hashing.{h,cc} in tests/test_oracle_cashash.py).  Everything is integer
or sign-of-float work with a fixed evaluation order: bit-exact."""
import numpy as np
import pytest

import oracle_lib
from orthosfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from orthosfm_amd import capi
    from orthosfm_amd.matching import HipCascadeHashing
    assert capi.device_count() >= 1
    iset = synth.make_image_set(4, 2200, n_surf=500, config_id=23)
    iset.sift[3] = iset.sift[3][:1501]
    iset.surf[2] = iset.surf[2][:0]
    orc = oracle_lib.OracleCasHash(iset.sift, iset.surf)
    m = HipCascadeHashing(4)
    for v in range(4):
        m.set_view(v, iset.sift[v], iset.surf[v])
    yield iset, orc, m, capi
    m.close()


def test_hashes_and_bucket_ids_bit_exact(setup):
    iset, orc, m, _ = setup
    for t in (0, 1):
        for v in range(4):
            h, b = m.cascade_hashes(v, t)
            oh, ob = orc.local[t][v]
            assert np.array_equal(h, oh), (t, v)
            assert np.array_equal(b, ob.astype(np.uint8)), (t, v)


def test_pairwise_match_bit_exact(setup):
    iset, orc, m, _ = setup
    for a in range(4):
        for b in range(4):
            if a == b:
                continue
            got = m.pairwise_match(a, b)
            o12, o21 = orc.pairwise_match(a, b)
            assert np.array_equal(got.matches_1_2, o12), (a, b)
            assert np.array_equal(got.matches_2_1, o21), (a, b)
    assert (orc.pairwise_match(1, 0)[0] >= 0).sum() > 100


def test_keep_empty_blocks_option(setup):
    """View 2 has no SURF: by default the lists follow sfm::CascadeHashing (the SURF part
    of a pair with view 2 on side 2 is absent); with cascade_keep_empty_blocks the
    exhaustive matcher's layout (-1 filled block) comes back."""
    iset, orc, _, capi = setup
    from orthosfm_amd.matching import HipCascadeHashing
    o = capi.default_match_options()
    o.cascade_keep_empty_blocks = 1
    m = HipCascadeHashing(4, options=o)
    for v in range(4):
        m.set_view(v, iset.sift[v], iset.surf[v])
    for a, b in ((0, 2), (2, 0), (1, 3)):
        got = m.pairwise_match(a, b)
        k12, k21 = orc.pairwise_match(a, b, keep_empty_blocks=True)
        assert np.array_equal(got.matches_1_2, k12) and np.array_equal(got.matches_2_1, k21), (a, b)
    assert m.pairwise_match(0, 2).matches_1_2.shape[0] == iset.sift[0].shape[0] + iset.surf[0].shape[0]
    r12, _ = orc.pairwise_match(0, 2)
    assert r12.shape[0] == iset.sift[0].shape[0]
    m.close()


def test_lowres_stays_exhaustive_and_compute_runs_the_gates(setup):
    iset, orc, m, capi = setup
    low = oracle_lib.oracle_pairwise_match_lowres(iset.sift[1], iset.surf[1], iset.sift[0], iset.surf[0], 500)
    assert m.pairwise_match_lowres(1, 0, 500) == low
    out = m.compute()
    assert len(out) == 6
    for tv in out:
        a, b = tv.view_1_id, tv.view_2_id
        o12, _ = orc.pairwise_match(a, b)
        cnt = int((o12 >= 0).sum())
        if tv.status == capi.PAIR_MATCHED:
            idx = np.nonzero(o12 >= 0)[0]
            assert np.array_equal(tv.matches, np.stack([idx, o12[idx]], axis=1).astype(np.int32))
        elif tv.status == capi.PAIR_REJECTED_COUNT:
            assert cnt < 50
    assert any(tv.status == capi.PAIR_MATCHED for tv in out)


def test_hashes_follow_the_set_of_views(setup):
    """The descriptor average runs over ALL views: replacing one view changes
    the hashes of the others, and the library notices."""
    iset, orc, m, _ = setup
    from orthosfm_amd.matching import HipCascadeHashing
    m2 = HipCascadeHashing(2)
    m2.set_view(0, iset.sift[0], iset.surf[0])
    m2.set_view(1, iset.sift[1], iset.surf[1])
    o2 = oracle_lib.OracleCasHash(iset.sift[:2], iset.surf[:2])
    h, _ = m2.cascade_hashes(0, 0)
    assert np.array_equal(h, o2.local[0][0][0])
    m2.set_view(1, iset.sift[2], iset.surf[1])                 # another view 1: new average
    o3 = oracle_lib.OracleCasHash([iset.sift[0], iset.sift[2]], [iset.surf[0], iset.surf[1]])
    h3, _ = m2.cascade_hashes(0, 0)
    assert np.array_equal(h3, o3.local[0][0][0])
    got = m2.pairwise_match(0, 1)
    o12, o21 = o3.pairwise_match(0, 1)
    assert np.array_equal(got.matches_1_2, o12) and np.array_equal(got.matches_2_1, o21)
    m2.close()
