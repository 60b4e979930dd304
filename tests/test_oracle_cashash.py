"""Cascade hashing (the application's default matcher):
the oracle (oracle/cashash_oracle.c) is PINNED stage by stage against the
reference's own cascade_hashing.{h,cc} (oracle/_ref/libref_cashash.so):
projection matrices, descriptor averages via the hashes, hashes, bucket ids and
the two-way match results, all bit for bit."""
import numpy as np
import pytest

import oracle_lib
from orthosfm_amd import synth

pytestmark = pytest.mark.skipif(oracle_lib.ref_cashash() is None, reason="oracle/_ref/libref_cashash.so not built")


@pytest.fixture(scope="module")
def views():
    iset = synth.make_image_set(4, 900, n_surf=300, config_id=21)
    # one view with a different size and one empty SURF set
    iset.sift[3] = iset.sift[3][:611]
    iset.surf[2] = iset.surf[2][:0]
    ref = oracle_lib.RefCasHash(iset.sift, iset.surf)
    orc = oracle_lib.OracleCasHash(iset.sift, iset.surf)
    yield iset, ref, orc
    ref.close()


def test_projection_matrices_bit_exact(views):
    _, ref, orc = views
    for t in (0, 1):
        rp, rs = ref.proj(t)
        op, os_ = orc.proj[t]
        assert np.array_equal(rp.view(np.uint32), op.view(np.uint32))
        assert np.array_equal(rs.view(np.uint32), os_.view(np.uint32))
    # sanity: standard normal
    assert abs(orc.proj[0][0].mean()) < 0.03 and abs(orc.proj[0][0].std() - 1.0) < 0.03


def test_hashes_and_buckets_bit_exact(views):
    iset, ref, orc = views
    for t in (0, 1):
        for v in range(4):
            rh, rb = ref.local(t, v)
            oh, ob = orc.local[t][v]
            assert np.array_equal(rh, oh), (t, v)
            assert np.array_equal(rb, ob), (t, v)
    assert orc.local[0][0][1].max() < 256


def test_pairwise_match_bit_exact(views):
    iset, ref, orc = views
    nonempty = 0
    for a in range(4):
        for b in range(4):
            if a == b:
                continue
            r12, r21 = ref.pairwise_match(a, b)
            o12, o21 = orc.pairwise_match(a, b)
            # including the layout with an empty SURF set on one side: the reference leaves
            # that part out of its result vectors (cascade_hashing.h:341-342)
            assert np.array_equal(r12, o12) and np.array_equal(r21, o21), (a, b)
            k12, k21 = orc.pairwise_match(a, b, keep_empty_blocks=True)
            assert np.array_equal(r12, k12[:len(r12)]) and (k12[len(r12):] == -1).all(), (a, b)
            nonempty += int((o12 >= 0).sum() > 20)
    assert nonempty >= 6


def test_cascade_differs_from_exhaustive_but_mostly_agrees(views):
    """It is an approximate matcher: most exhaustive matches are found, not all."""
    iset, _, orc = views
    e12, _ = oracle_lib.oracle_pairwise_match(iset.sift[1], iset.surf[1], iset.sift[0], iset.surf[0])
    c12, _ = orc.pairwise_match(1, 0)
    both = (e12 >= 0) & (c12 >= 0)
    assert (e12[both] == c12[both]).mean() > 0.99
    assert 0.5 < (c12 >= 0).sum() / max((e12 >= 0).sum(), 1) <= 1.2
