"""CPU checks of the outlier-filter oracle (oracle/filter_oracle.c).  The
reference file needs Eigen and cannot be built here (parity unpinned, see the
oracle header); the oracle is checked against an independent nearest-neighbour
search (scipy k-d tree) and against the statistics written out in numpy,
including the reference's counter quirk."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle_lib


def cloud(n, seed, outliers=0):
    rng = np.random.default_rng(seed)
    p = np.concatenate([rng.normal(0, 1.0, (n, 3)), np.ones((n, 1))], axis=1)
    if outliers:
        p[:outliers, :3] *= 8.0
    return p


def test_nn_distances_match_kdtree():
    p = cloud(3000, 1)
    nn = oracle_lib.oracle_nn_distances(p)
    d, _ = cKDTree(p).query(p, k=2)
    np.testing.assert_allclose(nn, d[:, 1], rtol=1e-13, atol=0)


def test_nn_distances_edge_cases():
    assert oracle_lib.oracle_nn_distances(np.zeros((0, 4))).shape == (0,)
    # a single point has no neighbour: the start value survives (outlier_filtering.cpp:22)
    assert oracle_lib.oracle_nn_distances(np.array([[1.0, 2, 3, 1]]))[0] == 1000000
    # duplicates are neighbours at distance 0; w takes part in the distance
    p = np.array([[0.0, 0, 0, 1], [0, 0, 0, 1], [0, 0, 0, 3]])
    np.testing.assert_array_equal(oracle_lib.oracle_nn_distances(p), [0.0, 0.0, 2.0])
    # nothing closer than 1e6: the start value again
    far = np.array([[0.0, 0, 0, 1], [3e6, 0, 0, 1]])
    np.testing.assert_array_equal(oracle_lib.oracle_nn_distances(far), [1e6, 1e6])


def test_filter_statistics_follow_the_reference_quirks():
    p = cloud(2000, 2, outliers=40)
    has = np.ones(2000, dtype=bool)
    has[::7] = False
    keep, mean, sigma = oracle_lib.oracle_filter_outlier_tracks(p, has)
    nn = oracle_lib.oracle_nn_distances(p[has])
    m = np.sum(nn) / nn.size
    # sigma divides by TWICE the point count (the counter keeps counting, :80-94)
    s = max(np.sqrt(np.sum((nn - m) ** 2) / (2 * nn.size)), 1e-3)
    assert mean == pytest.approx(m, rel=1e-12)
    assert sigma == pytest.approx(s, rel=1e-12)
    dist = np.zeros(2000)
    dist[has] = nn
    expect = ~has | ((np.linalg.norm(p, axis=1) <= 10) & (dist < mean + 1.6 * sigma))
    np.testing.assert_array_equal(keep, expect)
    assert keep[~has].all()                       # tracks without a point always stay
    assert 0 < (~keep).sum() < 400


def test_filter_sigma_floor_and_bounding_box():
    # a regular grid: all distances equal, sigma = 0 -> floored to 1e-3, everything inside stays
    g = np.stack(np.meshgrid(np.arange(6.0), np.arange(6.0), np.arange(6.0)), -1).reshape(-1, 3) * 0.5
    p = np.concatenate([g, np.ones((g.shape[0], 1))], axis=1)
    keep, mean, sigma = oracle_lib.oracle_filter_outlier_tracks(p, np.ones(len(p), bool))
    assert sigma == 1e-3 and mean == 0.5 and keep.all()
    # a point whose 4-vector norm exceeds 10 goes even though its neighbour is close
    q = np.concatenate([p, [[9.0, 5.0, 0, 1], [9.0, 5.0, 0.5, 1]]])
    keep, _, _ = oracle_lib.oracle_filter_outlier_tracks(q, np.ones(len(q), bool))
    assert not keep[-1] and not keep[-2] and keep[:-2].all()
