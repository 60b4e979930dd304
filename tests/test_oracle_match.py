"""Pins the CPU oracle (oracle/match_oracle.c) for hot path A.

* against the committed golden vectors produced by the reference's own
  matcher (tests/golden/match_*.npz, made by make_match_golden.py), and
* live against oracle/_ref/libref_match.so when that build is present
  (this container), on extra randomised inputs.
Bit-exact everywhere: the path is integer arithmetic plus one float ratio.
"""
import os

import numpy as np
import pytest

import match_cases
import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "match_twoway.npz"))


@pytest.fixture(scope="module")
def gold_ex():
    return np.load(os.path.join(GOLD, "match_exhaustive.npz"))


def _case_names(g):
    return sorted({k.rsplit("/", 1)[0] for k in g.files})


def test_golden_inputs_match_case_generator(gold):
    """The seeded generators still produce the bytes the goldens were made from."""
    for kind, cases in (("u16", match_cases.u16_cases()), ("s16", match_cases.s16_cases())):
        for name, (s1, s2, lowe) in cases.items():
            k = f"{kind}/{name}/"
            assert np.array_equal(gold[k + "s1"], s1), k
            assert np.array_equal(gold[k + "s2"], s2), k


def test_oracle_twoway_vs_golden(gold):
    om = oracle_lib.oracle_matcher()
    names = _case_names(gold)
    assert len(names) >= 25
    for k in names:
        s1, s2, lowe = gold[k + "/s1"], gold[k + "/s2"], float(gold[k + "/lowe"])
        m12, m21 = om.twoway(s1, s2, lowe)
        assert np.array_equal(m12, gold[k + "/m12"]), k
        assert np.array_equal(m21, gold[k + "/m21"]), k
        c12, c21 = om.remove_inconsistent(m12, m21)
        assert np.array_equal(c12, gold[k + "/c12"]), k
        assert np.array_equal(c21, gold[k + "/c21"]), k
        assert om.count_consistent(m12, m21) == int(gold[k + "/count"]), k
        nn = gold[k + "/nn"]
        for i in range(nn.shape[0]):
            assert np.array_equal(om.nn_find(s1[i], s2), nn[i]), (k, i)


def test_oracle_exhaustive_vs_golden(gold_ex):
    """A1 quantisation + A7 pairwise_match / pairwise_match_lowres + A6 combine."""
    views = []
    v = 0
    while f"view{v}/sift" in gold_ex.files:
        views.append((gold_ex[f"view{v}/sift"], gold_ex[f"view{v}/surf"]))
        v += 1
    q = [(oracle_lib.oracle_convert_sift(s), oracle_lib.oracle_convert_surf(u)) for s, u in views]
    n = 0
    for a in range(len(views)):
        for b in range(len(views)):
            if a == b:
                continue
            m12, m21 = oracle_lib.oracle_pairwise_match(q[a][0], q[a][1], q[b][0], q[b][1])
            assert np.array_equal(m12, gold_ex[f"pair{a}_{b}/m12"]), (a, b)
            assert np.array_equal(m21, gold_ex[f"pair{a}_{b}/m21"]), (a, b)
            for nf in (40, 500):
                c = oracle_lib.oracle_pairwise_match_lowres(q[a][0], q[a][1], q[b][0], q[b][1], nf)
                assert c == int(gold_ex[f"pair{a}_{b}/lowres{nf}"]), (a, b, nf)
            n += 1
    assert n == 30


def test_combine_results_offsets():
    om = oracle_lib.oracle_matcher()
    s12 = np.array([1, -1, 0], np.int32)
    s21 = np.array([2, 0], np.int32)
    u12 = np.array([-1, 1], np.int32)
    u21 = np.array([-1, 1, -1], np.int32)
    o12, o21 = om.combine(s12, s21, u12, u21)
    assert o12.tolist() == [1, -1, 0, -1, 1 + 2]
    assert o21.tolist() == [2, 0, -1, 1 + 3, -1]
    # no SIFT on side 2 -> SURF indices of side 1 are NOT shifted (matching.cc:78-81)
    o12, o21 = om.combine(np.array([-1, -1], np.int32), np.zeros(0, np.int32), u12, u21)
    assert o12.tolist() == [-1, -1, -1, 1]
    assert o21.tolist() == [-1, 1 + 2, -1]


def test_pair_enumeration():
    import ctypes as C
    f = oracle_lib.oracle().oracle_pair_from_index
    f.argtypes = [C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    seen = set()
    V = 23
    for i in range(V * (V - 1) // 2):
        a, b = C.c_int(), C.c_int()
        f(i, C.byref(a), C.byref(b))
        assert 0 <= b.value < a.value < V
        seen.add((a.value, b.value))
    assert len(seen) == V * (V - 1) // 2


# ---------------------------------------------------------------------------
# live comparison with the reference build (only where oracle/_ref exists)
# ---------------------------------------------------------------------------

needs_ref = pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built here")


@needs_ref
@pytest.mark.ref
@pytest.mark.parametrize("seed", range(6))
def test_oracle_vs_reference_random_u16(seed):
    om, rm = oracle_lib.oracle_matcher(), oracle_lib.ref_matcher()
    r = np.random.default_rng(100 + seed)
    n1, n2 = int(r.integers(1, 300)), int(r.integers(1, 300))
    if seed % 2:
        s1, s2 = match_cases.sift_pair(n1, n2, min(n1, n2) // 2, 200 + seed)
    else:   # arbitrary 0..255 data: exercises lane wrap + u16 truncation
        s1 = r.integers(0, 256, (n1, 128)).astype(np.uint16)
        s2 = r.integers(0, 256, (n2, 128)).astype(np.uint16)
    for lowe in (0.8, 0.95):
        a = om.twoway(s1, s2, lowe)
        b = rm.twoway(s1, s2, lowe)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for i in range(min(n1, 16)):
        assert np.array_equal(om.nn_find(s1[i], s2), rm.nn_find(s1[i], s2))


@needs_ref
@pytest.mark.ref
@pytest.mark.parametrize("seed", range(6))
def test_oracle_vs_reference_random_s16(seed):
    om, rm = oracle_lib.oracle_matcher(), oracle_lib.ref_matcher()
    r = np.random.default_rng(300 + seed)
    n1, n2 = int(r.integers(1, 300)), int(r.integers(1, 300))
    if seed % 2:
        s1, s2 = match_cases.surf_pair(n1, n2, min(n1, n2) // 2, 400 + seed)
    else:
        s1 = r.integers(-127, 128, (n1, 64)).astype(np.int16)
        s2 = r.integers(-127, 128, (n2, 64)).astype(np.int16)
    for lowe in (0.7, 0.9):
        a = om.twoway(s1, s2, lowe)
        b = rm.twoway(s1, s2, lowe)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for i in range(min(n1, 16)):
        assert np.array_equal(om.nn_find(s1[i], s2), rm.nn_find(s1[i], s2))


@needs_ref
@pytest.mark.ref
def test_oracle_vs_reference_cross_check_and_combine():
    om, rm = oracle_lib.oracle_matcher(), oracle_lib.ref_matcher()
    r = np.random.default_rng(7)
    for _ in range(20):
        n1, n2 = int(r.integers(0, 40)), int(r.integers(0, 40))
        m12 = r.integers(-1, max(n2, 1), n1).astype(np.int32) if n2 else -np.ones(n1, np.int32)
        m21 = r.integers(-1, max(n1, 1), n2).astype(np.int32) if n1 else -np.ones(n2, np.int32)
        a, b = om.remove_inconsistent(m12, m21), rm.remove_inconsistent(m12, m21)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert om.count_consistent(m12, m21) == rm.count_consistent(m12, m21)
        u1, u2 = int(r.integers(0, 9)), int(r.integers(0, 9))
        u12 = r.integers(-1, max(u2, 1), u1).astype(np.int32) if u2 else -np.ones(u1, np.int32)
        u21 = r.integers(-1, max(u1, 1), u2).astype(np.int32) if u1 else -np.ones(u2, np.int32)
        a, b = om.combine(m12, m21, u12, u21), rm.combine(m12, m21, u12, u21)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
