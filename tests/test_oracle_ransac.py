"""Pins the geometric-verification oracle (oracle/ransac_oracle.c):
  * Sampson distance bit-for-bit against the reference's sampson_distance,
  * the 8-point solve against the reference's fundamental_8_point +
    enforce_fundamental_constraints (same matrix up to sign/scale),
  * the RANSAC loop statistically against RansacFundamental::estimate (the
    reference samples with std::rand, so runs differ by construction)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from orthosfm_amd import synth

needs_ref = pytest.mark.skipif(oracle_lib.ref_ransac() is None, reason="oracle/_ref/libref_ransac.so not built here")


def two_view_scene(n, outlier_frac, seed, noise=2e-4):
    """Normalised positions of n correspondences between two orthographic views
    (MVE normalisation: (x + 0.5 - W/2) / max(W, H), feature_set.cc:42-55)."""
    r = np.random.default_rng(seed)
    P = r.uniform(-0.5, 0.5, (n, 3))
    xy1 = synth.project_euler(P, 0.3, 0.1, -0.05, 0.0, 0.0, 1.0, 2048, 2048)
    xy2 = synth.project_euler(P, 0.9, -0.2, 0.1, 0.0, 0.0, 1.0, 2048, 2048)
    p1 = ((xy1 + 0.5 - 1024) / 2048).astype(np.float32)
    p2 = ((xy2 + 0.5 - 1024) / 2048 + noise * r.standard_normal((n, 2))).astype(np.float32)
    n_out = int(outlier_frac * n)
    out = r.choice(n, n_out, replace=False)
    p2[out] = r.uniform(-0.5, 0.5, (n_out, 2)).astype(np.float32)
    inlier = np.ones(n, bool)
    inlier[out] = False
    perm1, perm2 = r.permutation(n), r.permutation(n)
    pos1 = np.zeros((n, 2), np.float32)
    pos2 = np.zeros((n, 2), np.float32)
    pos1[perm1] = p1
    pos2[perm2] = p2
    corr = np.stack([perm1, perm2], axis=1).astype(np.int32)
    order = np.argsort(corr[:, 0], kind="stable")
    return pos1, pos2, corr[order], inlier[order]


@needs_ref
@pytest.mark.ref
def test_sampson_distance_bit_exact():
    ref = oracle_lib.ref_ransac()
    f64 = np.ctypeslib.ndpointer(np.float64)
    ref.ref_sampson_distance.argtypes = [f64, f64, f64]
    ref.ref_sampson_distance.restype = C.c_double
    r = np.random.default_rng(1)
    for _ in range(300):
        F = r.standard_normal(9)
        p1, p2 = r.uniform(-0.5, 0.5, 2), r.uniform(-0.5, 0.5, 2)
        assert oracle_lib.oracle_sampson(F, p1, p2) == ref.ref_sampson_distance(F, p1, p2)


def _nd(a, b):
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max())


def _sample_systems(n_samples):
    pos1, pos2, corr, _ = two_view_scene(400, 0.0, 2, noise=1e-3)
    r = np.random.default_rng(3)
    for _ in range(n_samples):
        idx = np.sort(r.choice(400, 8, replace=False))
        yield pos1[corr[idx, 0]].astype(np.float64), pos2[corr[idx, 1]].astype(np.float64)


def test_eight_point_is_the_exact_null_vector():
    """fundamental_8_point + enforce_fundamental_constraints (fundamental.cc:78-127):
    smallest right singular vector of the 8x9 system, smallest singular value of
    the 3x3 removed -- checked against LAPACK (numpy) to 1e-9."""
    for p1, p2 in _sample_systems(60):
        A = np.stack([p2[:, 0] * p1[:, 0], p2[:, 0] * p1[:, 1], p2[:, 0], p2[:, 1] * p1[:, 0],
                      p2[:, 1] * p1[:, 1], p2[:, 1], p1[:, 0], p1[:, 1], np.ones(8)], 1)
        Fn = np.linalg.svd(A)[2][-1].reshape(3, 3)
        U, S, Vt = np.linalg.svd(Fn)
        S[2] = 0.0
        ok, F = oracle_lib.oracle_fundamental_8_point(p1, p2)
        assert ok and _nd(F, (U * S) @ Vt) < 1e-9
        assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-12


@needs_ref
@pytest.mark.ref
def test_eight_point_vs_reference_svd():
    """The reference's own SVD (math/matrix_svd.h) returns the same matrix for
    most samples; on a minority of near-degenerate systems it is off by 1e-3..1e-2
    from the LAPACK answer (measured here), which is why the 8-point solve is
    pinned to the exact null vector rather than to the reference's digits."""
    ref = oracle_lib.ref_ransac()
    f64 = np.ctypeslib.ndpointer(np.float64)
    ref.ref_fundamental_8_point.argtypes = [f64, f64, f64]
    ref.ref_fundamental_8_point.restype = None
    diffs = []
    for p1, p2 in _sample_systems(60):
        ok, F = oracle_lib.oracle_fundamental_8_point(p1, p2)
        Fr = np.zeros(9)
        ref.ref_fundamental_8_point(p1.reshape(-1), p2.reshape(-1), Fr)
        diffs.append(_nd(F, Fr.reshape(3, 3)))
    diffs = np.array(diffs)
    assert np.median(diffs) < 1e-9
    assert (diffs < 1e-6).mean() > 0.7 and diffs.max() < 0.1


def test_ransac_recovers_planted_inliers():
    pos1, pos2, corr, inlier = two_view_scene(1500, 0.35, 4)
    n, inl, F = oracle_lib.oracle_ransac(pos1, pos2, corr, seed=7, pair_id=3)
    got = np.zeros(corr.shape[0], bool)
    got[inl] = True
    assert n == inl.size and np.all(np.diff(inl) > 0)
    assert (got & inlier).sum() > 0.97 * inlier.sum()
    assert (got & ~inlier).sum() < 0.05 * (~inlier).sum() + 5
    # deterministic in (seed, pair id); a different stream still finds the structure
    n2, inl2, _ = oracle_lib.oracle_ransac(pos1, pos2, corr, seed=7, pair_id=3)
    assert n2 == n and np.array_equal(inl, inl2)
    n3, _, _ = oracle_lib.oracle_ransac(pos1, pos2, corr, seed=8, pair_id=3)
    assert abs(n3 - n) < 0.03 * n
    assert oracle_lib.oracle_ransac(pos1, pos2, corr[:7])[0] == -1      # < 8 matches


@needs_ref
@pytest.mark.ref
def test_ransac_statistics_vs_reference():
    ref = oracle_lib.ref_ransac()
    f64 = np.ctypeslib.ndpointer(np.float64)
    i32 = np.ctypeslib.ndpointer(np.int32)
    ref.ref_ransac_fundamental.argtypes = [f64, f64, C.c_int, C.c_int, C.c_double, C.c_uint, i32, f64]
    ref.ref_ransac_fundamental.restype = C.c_int
    for seed, frac in ((5, 0.2), (6, 0.5)):
        pos1, pos2, corr, inlier = two_view_scene(800, frac, seed)
        p1 = pos1[corr[:, 0]].astype(np.float64)
        p2 = pos2[corr[:, 1]].astype(np.float64)
        counts_ref, counts_or = [], []
        for s in range(5):
            inl = np.zeros(800, np.int32)
            F = np.zeros(9)
            counts_ref.append(ref.ref_ransac_fundamental(p1.reshape(-1), p2.reshape(-1), 800, 1000, 0.0015, s, inl, F))
            counts_or.append(oracle_lib.oracle_ransac(pos1, pos2, corr, seed=s, pair_id=1)[0])
        # with 50 % outliers only ~4 of the 1000 samples are outlier-free, so single
        # runs of either implementation scatter; compare the best runs and require
        # the oracle not to be systematically worse
        assert abs(max(counts_ref) - max(counts_or)) < 0.02 * inlier.sum() + 3, (counts_ref, counts_or)
        assert np.mean(counts_or) > 0.9 * np.mean(counts_ref), (counts_ref, counts_or)
        if frac <= 0.2:
            assert abs(np.mean(counts_ref) - np.mean(counts_or)) < 0.01 * inlier.sum() + 2


@needs_ref
@pytest.mark.ref
def test_hypothesis_inlier_sets_vs_reference_on_identical_samples():
    """The one place where the oracle does not follow the reference's digits: the 8-point
    solve (exact null vector vs MVE's one-sided Jacobi SVD).  Same 8-index samples into
    both, the resulting F scored with the (bit-identical) Sampson distance over all
    matches of a realistic pair: how often does a hypothesis' INLIER SET differ, and does
    it change what RANSAC returns -- the best hypothesis and the pair's accept decision?
    (Numbers quoted in DESIGN.md section 2.5.)"""
    ref = oracle_lib.ref_ransac()
    f64 = np.ctypeslib.ndpointer(np.float64)
    ref.ref_fundamental_8_point.argtypes = [f64, f64, f64]
    ref.ref_fundamental_8_point.restype = None
    n, thr = 1200, 0.0015
    stats = []
    for frac, seed in ((0.15, 11), (0.35, 12)):
        pos1, pos2, corr, inlier = two_view_scene(n, frac, seed)
        p1 = pos1[corr[:, 0]].astype(np.float64)
        p2 = pos2[corr[:, 1]].astype(np.float64)
        h = np.concatenate([p1, np.ones((n, 1))], 1), np.concatenate([p2, np.ones((n, 1))], 1)

        def inlier_set(F):
            # sampson_distance (fundamental.cc:225-246), vectorised: (p2^T F p1)^2 / (|F p1|_xy^2 + |F^T p2|_xy^2)
            F = F.reshape(3, 3)
            Fp1 = h[0] @ F.T
            Ftp2 = h[1] @ F
            num = (h[1] * Fp1).sum(1) ** 2
            den = Fp1[:, 0] ** 2 + Fp1[:, 1] ** 2 + Ftp2[:, 0] ** 2 + Ftp2[:, 1] ** 2
            return num / den < thr ** 2

        r = np.random.default_rng(seed)
        differ = 0
        best_ref = best_or = 0
        total = 10000
        size_gap = []
        for _ in range(total):
            idx = r.choice(n, 8, replace=False)
            a, b = np.ascontiguousarray(p1[idx]), np.ascontiguousarray(p2[idx])
            ok, Fo = oracle_lib.oracle_fundamental_8_point(a, b)
            Fr = np.zeros(9)
            ref.ref_fundamental_8_point(a.reshape(-1), b.reshape(-1), Fr)
            so, sr = inlier_set(Fo), inlier_set(Fr)
            if not np.array_equal(so, sr):
                differ += 1
                size_gap.append(int(so.sum()) - int(sr.sum()))
            best_ref, best_or = max(best_ref, int(sr.sum())), max(best_or, int(so.sum()))
        stats.append((frac, differ / total, best_ref, best_or, int(inlier.sum()),
                      float(np.mean(np.abs(size_gap))) if size_gap else 0.0))
    for frac, share, best_ref, best_or, planted, gap in stats:
        print(f"outliers {frac:.2f}: {100 * share:.2f} % of 10000 hypotheses differ in their inlier set "
              f"(mean |size gap| {gap:.1f}); best hypothesis: reference {best_ref}, oracle {best_or} of {planted} planted")
        # the sets differ for a minority of (near-degenerate) samples, and never where it counts:
        # the best hypotheses agree to a few matches, far above the accept threshold of 30
        assert share < 0.25
        assert abs(best_ref - best_or) <= max(3, 0.01 * planted)
        assert best_or >= 0.95 * planted
