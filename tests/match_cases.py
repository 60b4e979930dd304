"""Seeded small / adversarial input cases for the matching parity tests.

Shared by the golden generator (tests/golden/make_match_golden.py), the
oracle tests and the GPU parity tests so all three see the same bytes.
"""
import numpy as np

from orthosfm_amd import synth


def _rng(seed):
    return np.random.default_rng(seed)


def sift_pair(n1, n2, shared, seed, noise=0.04):
    """Two SIFT-like u16 sets that share `shared` landmarks."""
    r = _rng(seed)
    base = synth.sift_like(r.standard_normal((max(n1, n2) + shared, 128)))
    a = base[r.permutation(base.shape[0])[:n1]].copy() if n1 else base[:0]
    b = base[r.permutation(base.shape[0])[:n2]].copy() if n2 else base[:0]
    k = min(shared, n1, n2)
    if k:
        ia = r.permutation(n1)[:k]
        ib = r.permutation(n2)[:k]
        b[ib] = a[ia]
    a = synth.sift_like(a + noise * np.abs(r.standard_normal(a.shape)) / 11.3) if n1 else a
    b = synth.sift_like(b + noise * np.abs(r.standard_normal(b.shape)) / 11.3) if n2 else b
    return synth.quantize_sift(a).reshape(n1, 128), synth.quantize_sift(b).reshape(n2, 128)


def surf_pair(n1, n2, shared, seed, noise=0.05):
    r = _rng(seed)
    a = synth.surf_like(r.standard_normal((n1, 64))) if n1 else np.zeros((0, 64), np.float32)
    b = synth.surf_like(r.standard_normal((n2, 64))) if n2 else np.zeros((0, 64), np.float32)
    k = min(shared, n1, n2)
    if k:
        ia = r.permutation(n1)[:k]
        ib = r.permutation(n2)[:k]
        b[ib] = synth.surf_like(a[ia] + noise * r.standard_normal((k, 64)))
    return synth.quantize_surf(a).reshape(n1, 64), synth.quantize_surf(b).reshape(n2, 64)


def u16_cases():
    """name -> (set1 u16 [n1,128], set2 u16 [n2,128], lowe)."""
    c = {}
    c["sift_37x53"] = (*sift_pair(37, 53, 20, 1), 0.8)
    c["sift_200x130"] = (*sift_pair(200, 130, 90, 2), 0.8)
    c["sift_1x1"] = (*sift_pair(1, 1, 1, 3), 0.8)
    c["sift_1x40"] = (*sift_pair(1, 40, 1, 4), 0.8)
    c["sift_40x1"] = (*sift_pair(40, 1, 1, 5), 0.8)
    c["sift_0x9"] = (*sift_pair(0, 9, 0, 6), 0.8)
    c["sift_9x0"] = (*sift_pair(9, 0, 0, 7), 0.8)
    c["sift_lowe1"] = (*sift_pair(64, 64, 30, 8), 1.0)
    c["sift_lowe05"] = (*sift_pair(64, 96, 40, 9), 0.5)
    # ties: duplicated candidates -> the LAST index must win, second == best
    a, b = sift_pair(48, 40, 30, 10)
    b = np.concatenate([b, b[:17], b[5:9]], axis=0)
    a = np.concatenate([a, a[:11]], axis=0)
    c["sift_dups"] = (a, b, 0.8)
    # zero vectors on both sides (ip == 0 everywhere for those rows)
    a, b = sift_pair(33, 35, 10, 11)
    a[3] = 0
    a[17] = 0
    b[0] = 0
    b[34] = 0
    c["sift_zero_rows"] = (a, b, 0.8)
    z = np.zeros((5, 128), np.uint16)
    c["all_zero"] = (z, np.zeros((7, 128), np.uint16), 0.8)
    # exact duplicates of maximal norm: best and second both clamp to
    # distance 0 -> 0/0 = NaN -> accepted (matching.h:140-143)
    one = np.zeros((1, 128), np.uint16)
    one[0, 5] = 255
    c["nan_accept"] = (np.repeat(one, 3, 0), np.repeat(one, 4, 0), 0.8)
    # 16-bit lane wrap and truncation (nearest_neighbor.cc:75-84, result
    # fields are unsigned short): un-normalised large values
    r = _rng(12)
    big1 = r.integers(0, 256, (21, 128)).astype(np.uint16)
    big2 = r.integers(0, 256, (19, 128)).astype(np.uint16)
    c["wrap_random255"] = (big1, big2, 0.8)
    c["wrap_all255"] = (np.full((4, 128), 255, np.uint16), np.full((6, 128), 255, np.uint16), 0.8)
    mix1, mix2 = sift_pair(40, 44, 20, 13)
    mix1[7] = big1[0]
    mix2[9] = big2[0]
    mix2[30] = big2[1]
    c["wrap_mixed"] = (mix1, mix2, 0.8)
    # descriptors with entries >= 128 (single dominant bins)
    a, b = sift_pair(70, 66, 30, 14)
    for k in range(0, 70, 7):
        a[k] = 0
        a[k, (3 * k) % 128] = 255
        a[k, (5 * k + 1) % 128] = 20
    for k in range(0, 66, 6):
        b[k] = 0
        b[k, (3 * k) % 128] = 250
        b[k, (7 * k + 2) % 128] = 60
    c["sift_hi_values"] = (a, b, 0.8)
    return c


def s16_cases():
    """name -> (set1 s16 [n1,64], set2 s16 [n2,64], lowe)."""
    c = {}
    c["surf_41x29"] = (*surf_pair(41, 29, 15, 21), 0.7)
    c["surf_150x170"] = (*surf_pair(150, 170, 80, 22), 0.7)
    c["surf_1x1"] = (*surf_pair(1, 1, 1, 23), 0.7)
    c["surf_0x5"] = (*surf_pair(0, 5, 0, 24), 0.7)
    c["surf_5x0"] = (*surf_pair(5, 0, 0, 25), 0.7)
    a, b = surf_pair(30, 30, 12, 26)
    b = np.concatenate([b, b[:9]], axis=0)
    c["surf_dups"] = (a, b, 0.7)
    # every inner product negative for some queries: state stays (0,0,idx 0)
    a, b = surf_pair(20, 24, 8, 27)
    a[4] = -np.abs(a[4]) - 1
    b[:] = np.abs(b)
    a[11] = 0
    c["surf_negative"] = (a, b, 0.7)
    r = _rng(28)
    c["surf_wrap_random"] = (r.integers(-127, 128, (17, 64)).astype(np.int16),
                             r.integers(-127, 128, (23, 64)).astype(np.int16), 0.7)
    c["surf_wrap_all127"] = (np.full((3, 64), 127, np.int16), np.full((5, 64), 127, np.int16), 0.7)
    m = np.full((3, 64), 127, np.int16)
    m[1, ::2] = -127
    c["surf_wrap_signs"] = (m, np.full((5, 64), -127, np.int16), 0.7)
    return c


def exhaustive_views(seed=31):
    """Float descriptor views for ExhaustiveMatching (A1 + A7): mixed
    SIFT/SURF counts including a view with no SIFT and one with no SURF."""
    r = _rng(seed)
    L = 260
    bs = synth.sift_like(r.standard_normal((L, 128)))
    bu = synth.surf_like(r.standard_normal((L, 64)))
    counts = [(120, 60), (90, 0), (0, 70), (150, 80), (1, 1), (0, 0)]
    views = []
    for ns, nu in counts:
        ids = r.permutation(L)[:ns]
        s = synth.sift_like(bs[ids] + 0.03 * r.standard_normal((ns, 128)) / 11.3 * 4) if ns else np.zeros((0, 128), np.float32)
        idu = r.permutation(L)[:nu]
        u = synth.surf_like(bu[idu] + 0.04 * r.standard_normal((nu, 64))) if nu else np.zeros((0, 64), np.float32)
        # out-of-range floats exercise the clamps of convert_descriptor
        if ns > 5:
            s[2, 7] = 1.7
            s[3, 9] = -0.3
        if nu > 5:
            u[1, 3] = 1.4
            u[2, 5] = -1.9
        views.append((s.astype(np.float32), u.astype(np.float32)))
    return views
