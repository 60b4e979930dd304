"""GPU parity tests of hot path A, through the C ABI (libosfm_hip.so).

Every expected value is either a committed golden vector produced by the
reference's own matcher (tests/golden) or the CPU oracle on the same seeded
inputs.  Bit-exact: the path is integer work plus one float ratio test.
"""
import os

import numpy as np
import pytest

import match_cases
import oracle_lib
from orthosfm_amd import synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hm():
    from orthosfm_amd import capi
    from orthosfm_amd.matching import HipExhaustiveMatching
    assert capi.device_count() >= 1, "no HIP device"
    return HipExhaustiveMatching


def _names(g):
    return sorted({k.rsplit("/", 1)[0] for k in g.files})


def test_golden_twoway_all_cases(hm):
    """Matching::twoway_match / remove_inconsistent / count vs the reference's
    outputs for every small + adversarial case (ties, zeros, n=0/1, NaN accept,
    16-bit wrap, values >= 128, negative SURF products)."""
    g = np.load(os.path.join(GOLD, "match_twoway.npz"))
    names = _names(g)
    assert len(names) >= 25
    for k in names:
        s1, s2, lowe = g[k + "/s1"], g[k + "/s2"], float(g[k + "/lowe"])
        from orthosfm_amd import capi
        o = capi.default_match_options()
        is_u16 = k.startswith("u16")
        if is_u16:
            o.sift_lowe_ratio = lowe
        else:
            o.surf_lowe_ratio = lowe
        m = hm(2, options=o)
        if is_u16:
            m.set_view(0, s1)
            m.set_view(1, s2)
        else:
            m.set_view(0, np.zeros((0, 128), np.uint16), s1)
            m.set_view(1, np.zeros((0, 128), np.uint16), s2)
        t = 0 if is_u16 else 1
        r = m.twoway_match(0, 1, t)
        assert np.array_equal(r.matches_1_2, g[k + "/m12"]), k
        assert np.array_equal(r.matches_2_1, g[k + "/m21"]), k
        r = m.pairwise_match(0, 1)
        if s1.shape[0] > 0:
            assert np.array_equal(r.matches_1_2, g[k + "/c12"]), k
            assert np.array_equal(r.matches_2_1, g[k + "/c21"]), k
        else:   # view 1 has no descriptors: both lists are EMPTY (exhaustive_matching.cc:122,134)
            assert r.matches_1_2.size == 0 and r.matches_2_1.size == 0, k
        assert m.pairwise_match_lowres(0, 1, 100000) == (int(g[k + "/count"]) if s1.shape[0] else 0), k
        m.close()


def test_golden_exhaustive_views(hm):
    """Float descriptors -> init (quantisation) -> pairwise_match /
    pairwise_match_lowres with mixed SIFT/SURF counts, vs the reference."""
    g = np.load(os.path.join(GOLD, "match_exhaustive.npz"))
    views = []
    v = 0
    while f"view{v}/sift" in g.files:
        views.append((g[f"view{v}/sift"], g[f"view{v}/surf"]))
        v += 1
    m = hm(len(views))
    m.init(views)
    for a in range(len(views)):
        for b in range(len(views)):
            if a == b:
                continue
            r = m.pairwise_match(a, b)
            assert np.array_equal(r.matches_1_2, g[f"pair{a}_{b}/m12"]), (a, b)
            assert np.array_equal(r.matches_2_1, g[f"pair{a}_{b}/m21"]), (a, b)
            for nf in (40, 500):
                assert m.pairwise_match_lowres(a, b, nf) == int(g[f"pair{a}_{b}/lowres{nf}"]), (a, b, nf)
    m.close()


@pytest.mark.parametrize("n1,n2,seed", [(300, 257, 1), (1000, 777, 2), (2500, 9000, 3), (8300, 700, 4), (2049, 4097, 5),
                                          (2048, 4096, 6)])
def test_sift_vs_oracle_multi_block(hm, n1, n2, seed):
    """Several row blocks / column segments / partial tiles, vs the oracle."""
    s1, s2 = match_cases.sift_pair(n1, n2, min(n1, n2) // 2, 500 + seed)
    om = oracle_lib.oracle_matcher()
    e12, e21 = om.twoway(s1, s2, 0.8)
    m = hm(2)
    m.set_view(0, s1)
    m.set_view(1, s2)
    r = m.twoway_match(0, 1, 0)
    assert np.array_equal(r.matches_1_2, e12)
    assert np.array_equal(r.matches_2_1, e21)
    c12, c21 = om.remove_inconsistent(e12, e21)
    r = m.pairwise_match(0, 1)
    assert np.array_equal(r.matches_1_2, c12) and np.array_equal(r.matches_2_1, c21)
    assert m.stats().exact_scan_queries <= (n1 + n2) // 100      # only ties for best go to the sequential-scan kernel
    m.close()


@pytest.mark.parametrize("n1,n2,seed", [(513, 300, 1), (2000, 3100, 2)])
def test_surf_vs_oracle_multi_block(hm, n1, n2, seed):
    u1, u2 = match_cases.surf_pair(n1, n2, min(n1, n2) // 2, 600 + seed)
    om = oracle_lib.oracle_matcher()
    e12, e21 = om.twoway(u1, u2, 0.7)
    m = hm(2)
    z = np.zeros((0, 128), np.uint16)
    m.set_view(0, z, u1)
    m.set_view(1, z, u2)
    r = m.twoway_match(0, 1, 1)
    assert np.array_equal(r.matches_1_2, e12) and np.array_equal(r.matches_2_1, e21)
    m.close()


def test_wrap_exact_path_large(hm):
    """Un-normalised 0..255 data: most inner products exceed 65535, so the
    reference's 16-bit lane wrap and truncated state decide every match;
    the wrap-exact kernel must reproduce them."""
    r = np.random.default_rng(5)
    s1 = r.integers(0, 256, (600, 128)).astype(np.uint16)
    s2 = r.integers(0, 256, (700, 128)).astype(np.uint16)
    s1[::7] = (s1[::7] // 9)           # mix in rows that stay in range
    om = oracle_lib.oracle_matcher()
    e12, e21 = om.twoway(s1, s2, 0.8)
    m = hm(2)
    m.set_view(0, s1)
    m.set_view(1, s2)
    got = m.twoway_match(0, 1, 0)
    assert np.array_equal(got.matches_1_2, e12) and np.array_equal(got.matches_2_1, e21)
    assert m.stats().exact_scan_queries > 0
    # SURF with un-normalised rows: the norm bound fails -> every query exact-scanned
    u1 = r.integers(-127, 128, (300, 64)).astype(np.int16)
    u2 = r.integers(-127, 128, (280, 64)).astype(np.int16)
    e12, e21 = om.twoway(u1, u2, 0.7)
    z = np.zeros((0, 128), np.uint16)
    m.set_view(0, z, u1)
    m.set_view(1, z, u2)
    got = m.twoway_match(0, 1, 1)
    assert np.array_equal(got.matches_1_2, e12) and np.array_equal(got.matches_2_1, e21)
    m.close()


def test_match_all_small_image_set(hm):
    """bundler::Matching::compute up to RANSAC on a 6-view synthetic set:
    low-res gate, thresholds and the ordered correspondence lists, each pair
    checked against the oracle."""
    from orthosfm_amd import capi
    iset = synth.make_image_set(6, 1500, n_surf=200, config_id=9)
    # make one view unrelated (distractors only) so some pairs get rejected
    rr = np.random.default_rng(3)
    iset.sift[5] = synth.quantize_sift(synth.sift_like(rr.standard_normal((1500, 128))))
    iset.surf[5] = synth.quantize_surf(synth.surf_like(rr.standard_normal((200, 64))))
    m = hm(6)
    for v in range(6):
        m.set_view(v, iset.sift[v], iset.surf[v])
    out = m.compute()
    assert len(out) == 15
    statuses = set()
    for tv in out:
        a, b = tv.view_1_id, tv.view_2_id
        assert a > b
        low = oracle_lib.oracle_pairwise_match_lowres(iset.sift[a], iset.surf[a], iset.sift[b], iset.surf[b], 500)
        assert tv.lowres_matches == low
        if low < 5:
            assert tv.status == capi.PAIR_REJECTED_LOWRES
            statuses.add(tv.status)
            continue
        e12, e21 = oracle_lib.oracle_pairwise_match(iset.sift[a], iset.surf[a], iset.sift[b], iset.surf[b])
        cnt = int((e12 >= 0).sum())
        assert tv.num_matches == cnt
        if cnt < 50:
            assert tv.status == capi.PAIR_REJECTED_COUNT
        else:
            assert tv.status == capi.PAIR_MATCHED
            idx = np.nonzero(e12 >= 0)[0]
            exp = np.stack([idx, e12[idx]], axis=1).astype(np.int32)
            assert np.array_equal(tv.matches, exp)
        statuses.add(tv.status)
    assert capi.PAIR_MATCHED in statuses and len(statuses) >= 2
    m.close()


def test_full_size_properties(hm):
    """BASELINE-size views (20k features): properties that need no oracle --
    swapping the views swaps the lists; the result is a mutual matching; the
    planted landmark correspondences are recovered; matching a view with a
    permuted copy of itself returns the permutation."""
    iset = synth.make_image_set(2, 20000, config_id=2)
    m = hm(3)
    m.set_view(0, iset.sift[0])
    m.set_view(1, iset.sift[1])
    r01 = m.pairwise_match(0, 1)
    r10 = m.pairwise_match(1, 0)
    assert np.array_equal(r01.matches_1_2, r10.matches_2_1)
    assert np.array_equal(r01.matches_2_1, r10.matches_1_2)
    i = np.nonzero(r01.matches_1_2 >= 0)[0]
    assert np.array_equal(r01.matches_2_1[r01.matches_1_2[i]], i)
    # recovered correspondences agree with the planted landmarks
    lm0, lm1 = iset.landmark[0], iset.landmark[1]
    ok = lm0[i] == lm1[r01.matches_1_2[i]]
    shared = np.intersect1d(lm0[lm0 >= 0], lm1[lm1 >= 0]).size
    assert i.size > 0.5 * shared and ok.mean() > 0.99
    # permutation round trip
    perm = np.random.default_rng(1).permutation(20000)
    m.set_view(2, iset.sift[0][perm])
    r = m.pairwise_match(0, 2)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(20000)
    acc = r.matches_1_2 >= 0
    assert acc.mean() > 0.95
    assert np.array_equal(r.matches_1_2[acc], inv[acc])
    m.close()


def test_error_behaviour(hm):
    from orthosfm_amd import capi
    m = hm(2)
    with pytest.raises(capi.OsfmError) as e:
        m.pairwise_match(0, 1)          # views not set
    assert e.value.status == capi.E_STATE
    with pytest.raises(capi.OsfmError) as e:
        m.set_view(5, np.zeros((1, 128), np.uint16))
    assert e.value.status == capi.E_ARG
    with pytest.raises(capi.OsfmError) as e:
        m.set_view(0, np.full((3, 128), 300, np.uint16))
    assert e.value.status == capi.E_RANGE
    m.close()


def test_cpp_adapter_vs_reference_side_by_side():
    """oracle/_ref/adapter_check (built in the build container from the
    reference's own sources + the product's sfm::MatchingBase adapter) drives
    both matchers through the same C++ virtual interface."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "adapter_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/adapter_check was not built (reference sources absent)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter_check ok: 20 pairs identical to sfm::ExhaustiveMatching" in out.stdout
    assert "adapter_check ok: 20 pairs identical to sfm::CascadeHashing" in out.stdout
    assert "from 8 OpenMP threads identical to the serial calls" in out.stdout
    assert "20 pairs on 2 logical devices identical to sfm::ExhaustiveMatching" in out.stdout
    assert "20 pairs on 3 logical devices identical to sfm::ExhaustiveMatching" in out.stdout


def test_special_rows_large_values(hm):
    """Descriptors with entries > 127 cannot use the raw int8 row form: they are
    gathered into extra row blocks on the keyed path.  Mix them into both views
    (several hundred, spread over many row blocks) and compare with the oracle."""
    s1, s2 = match_cases.sift_pair(3000, 2600, 1500, 900)
    r = np.random.default_rng(11)
    for s in (s1, s2):
        rows = r.choice(s.shape[0], 300, replace=False)
        for k in rows:
            d = np.zeros(128, np.uint16)
            pos = r.choice(128, 3, replace=False)
            d[pos] = [int(r.integers(128, 256)), int(r.integers(0, 90)), int(r.integers(0, 60))]
            s[k] = d
    # duplicates of special rows: ties between special and ordinary blocks
    s1[10] = s1[2990]
    s2[5] = s2[2000]
    from orthosfm_amd import capi
    om = oracle_lib.oracle_matcher()
    e12, e21 = om.twoway(s1, s2, 0.8)
    c12, c21 = om.remove_inconsistent(e12, e21)
    for smax in (-1, 0):                    # the gathered row blocks this test was written for, then the special kernel
        o = capi.default_match_options()
        o.special_kernel_max = smax
        m = hm(2, options=o)
        m.set_view(0, s1)
        m.set_view(1, s2)
        got = m.twoway_match(0, 1, 0)       # masked (keyed) kernel
        assert np.array_equal(got.matches_1_2, e12) and np.array_equal(got.matches_2_1, e21)
        got = m.pairwise_match(0, 1)        # raw + special-row blocks / correction-free + special kernel
        assert np.array_equal(got.matches_1_2, c12) and np.array_equal(got.matches_2_1, c21), smax
        got = m.pairwise_match(1, 0)
        assert np.array_equal(got.matches_1_2, c21) and np.array_equal(got.matches_2_1, c12), smax
        m.close()


@pytest.mark.parametrize("k", [1, 20, 65, 200, 300, 500, 600])
def test_special_descriptors_few_per_view(hm, k):
    """The case real SIFT data produces (sift.cc:830-839 renormalises after the clamp): k
    descriptors per view with bytes > 127, built like MVE builds them.  Up to
    special_kernel_max of them (512) go through match_special_kernel while every other
    descriptor stays on the correction-free tile kernel; more (600), or a negative option,
    select the per-view operand forms.  Both against the oracle, all three pair
    orientations (special rows only in set 1, only in set 2, in both), 5000 features per
    view = two candidate chunks of the special kernel.  k <= 64 (two units of 32) runs
    match_special_kernel, 65 .. 512 match_special_wide_kernel (five chunks of 1024 candidates,
    the last one ragged; 65 = three units, one wave with a dead unit; 300 = ten units = two
    passes, the second with idle waves; 500 (+ 2 doubled rows) = sixteen units, two full passes)."""
    from orthosfm_amd import capi
    iset = synth.make_image_set(3, 5000, config_id=31)
    rows = synth.add_peaky_rows(iset, k)
    iset.sift[2][rows[2]] = synth.make_image_set(1, 5000, config_id=32).sift[0][rows[2]]   # view 2 stays plain
    # a special and an ordinary descriptor of view 1 doubled in view 0: ties across the two kernels
    iset.sift[0][7] = iset.sift[1][rows[1][0]]
    iset.sift[0][8] = iset.sift[1][rows[1][0]]
    om = oracle_lib.oracle_matcher()
    expect = {}
    for a, b in ((0, 1), (2, 1), (0, 2), (1, 0)):
        e12, e21 = om.twoway(iset.sift[a], iset.sift[b], 0.8)
        expect[(a, b)] = om.remove_inconsistent(e12, e21)
    for smax in (0, -1):
        o = capi.default_match_options()
        o.special_kernel_max = smax
        m = hm(3, options=o)
        for v in range(3):
            m.set_view(v, iset.sift[v])
        for (a, b), (c12, c21) in expect.items():
            got = m.pairwise_match(a, b)
            assert np.array_equal(got.matches_1_2, c12) and np.array_equal(got.matches_2_1, c21), (k, smax, a, b)
            assert (m.stats().special_kernel_launches > 0) == (smax == 0 and k <= 512), (k, smax, a, b)
        m.close()
    assert int((expect[(0, 1)][0] >= 0).sum()) > 800


def test_special_descriptors_edge_shapes(hm):
    """match_special_kernel at its edges: views where EVERY descriptor is special (the tile
    kernel then sees blanks only), 33 special rows (a unit of 32 plus one), a one-descriptor
    view, sizes off every multiple, exact duplicates (accepted ties go to the scan kernel),
    lowe ratio 1 (everything accepted, including ties at 0)."""
    from orthosfm_amd import capi
    r = np.random.default_rng(77)
    om = oracle_lib.oracle_matcher()

    def rand_special(n):
        d = np.zeros((n, 128), np.uint16)
        for i in range(n):
            pos = r.choice(128, 4, replace=False)
            d[i, pos] = [int(r.integers(128, 256)), int(r.integers(0, 200)), int(r.integers(0, 90)), int(r.integers(0, 40))]
        return d

    plain = synth.make_image_set(2, 700, config_id=33)
    cases = []
    cases.append((rand_special(97), rand_special(61)))                     # all special on both sides
    cases.append((rand_special(33), plain.sift[0][:515].copy()))           # 33 special rows vs a plain view
    cases.append((plain.sift[1][:1].copy(), rand_special(40)))             # one-descriptor view
    a = plain.sift[0].copy(); b = plain.sift[1].copy()
    a[[3, 100, 699]] = rand_special(3); b[[0, 650]] = rand_special(2)
    b[5] = a[3]; b[6] = a[3]; a[50] = b[650]                               # duplicates of special rows
    cases.append((a, b))
    z = plain.sift[0][:300].copy(); z[10:20] = 0                           # zero descriptors: inner products of 0
    cases.append((z, rand_special(12)))
    for lowe in (0.8, 1.0):
        o = capi.default_match_options()
        o.sift_lowe_ratio = lowe
        for ci, (s1, s2) in enumerate(cases):
            e12, e21 = om.twoway(s1, s2, lowe)
            c12, c21 = om.remove_inconsistent(e12, e21)
            m = hm(2, options=o)
            m.set_view(0, s1)
            m.set_view(1, s2)
            got = m.pairwise_match(0, 1)
            assert np.array_equal(got.matches_1_2, c12) and np.array_equal(got.matches_2_1, c21), (lowe, ci)
            assert m.stats().special_kernel_launches > 0
            got = m.pairwise_match(1, 0)
            assert np.array_equal(got.matches_1_2, c21) and np.array_equal(got.matches_2_1, c12), (lowe, ci)
            m.close()


def test_special_descriptors_random_sweep(hm):
    """Thirty random pairs: sizes 1..2500 (off every multiple), 0..70 special rows per view of several kinds
    (peaky rows as MVE makes them, rows with one byte just over 127, near-duplicates of ordinary rows lifted over
    the limit, all-255 rows whose products leave the 16-bit range), ties planted between special and ordinary rows,
    three ratios -- the special-descriptor kernel and the per-view forms both against the oracle."""
    from orthosfm_amd import capi
    om = oracle_lib.oracle_matcher()
    base = synth.make_image_set(2, 2500, config_id=37)
    r = np.random.default_rng(123)

    def mutate(s, k):
        n = s.shape[0]
        for _ in range(k):
            i = int(r.integers(n))
            kind = int(r.integers(5))
            if kind == 0:
                d = np.zeros(128, np.uint16); d[r.choice(128, 2, replace=False)] = [int(r.integers(128, 256)), int(r.integers(0, 128))]
            elif kind == 1:
                d = s[i].copy(); d[int(r.integers(128))] = 128
            elif kind == 2:
                d = s[int(r.integers(n))].copy(); d[int(r.integers(128))] = int(r.integers(128, 200))
            elif kind == 3:
                d = np.full(128, 255, np.uint16)
            else:
                d = s[i].copy(); d[:4] = 200
            s[i] = d
        return s

    for case in range(30):
        n1, n2 = int(r.integers(1, 2500)), int(r.integers(1, 2500))
        s1 = mutate(base.sift[0][r.permutation(2500)[:n1]].copy(), int(r.integers(0, 71)) if case % 5 else 0)
        s2 = mutate(base.sift[1][r.permutation(2500)[:n2]].copy(), int(r.integers(0, 71)) if case % 7 else 0)
        if n1 > 3 and n2 > 3:                # a row of the one view planted twice in the other: ties
            s2[0] = s1[1]; s2[2] = s1[1]; s1[3] = s2[1]
        lowe = (0.6, 0.8, 1.0)[case % 3]
        e12, e21 = om.twoway(s1, s2, lowe)
        c12, c21 = om.remove_inconsistent(e12, e21)
        for smax in (0, -1):
            o = capi.default_match_options()
            o.sift_lowe_ratio = lowe
            o.special_kernel_max = smax
            m = hm(2, options=o)
            m.set_view(0, s1)
            m.set_view(1, s2)
            got = m.pairwise_match(0, 1)
            assert np.array_equal(got.matches_1_2, c12) and np.array_equal(got.matches_2_1, c21), (case, smax, n1, n2, lowe)
            m.close()


def test_mixed_operand_forms_in_one_batch(hm):
    """One compute() over views with and without entries > 127: the launch then
    holds problems for all three kernel kinds (correction-free raw operands,
    raw rows against corrected columns, gathered special rows), and SURF rides
    along.  Every pair against the oracle."""
    from orthosfm_amd import capi
    iset = synth.make_image_set(4, 1200, n_surf=150, config_id=12)
    r = np.random.default_rng(21)
    for v in (1, 3):                                   # views 1 and 3 get large values, 0 and 2 stay plain
        rows = r.choice(1200, 150, replace=False)
        for k in rows:
            d = iset.sift[v][k].copy()
            d[r.choice(128, 2, replace=False)] = [int(r.integers(128, 256)), int(r.integers(128, 200))]
            iset.sift[v][k] = d
    expect = {}
    for smax in (-1, 0):       # the per-view forms (what the name says), then the special kernel on the same set
        o = capi.default_match_options()
        o.use_lowres_matching = 0
        o.min_feature_matches = 0
        o.special_kernel_max = smax
        m = hm(4, options=o)
        for v in range(4):
            m.set_view(v, iset.sift[v], iset.surf[v])
        out = m.compute()
        assert len(out) == 6
        assert (m.stats().special_kernel_launches > 0) == (smax == 0)
        for tv in out:
            a, b = tv.view_1_id, tv.view_2_id
            if (a, b) not in expect:
                e12, _ = oracle_lib.oracle_pairwise_match(iset.sift[a], iset.surf[a], iset.sift[b], iset.surf[b])
                idx = np.nonzero(e12 >= 0)[0]
                expect[(a, b)] = np.stack([idx, e12[idx]], axis=1).astype(np.int32)
            assert tv.status == capi.PAIR_MATCHED and np.array_equal(tv.matches, expect[(a, b)]), (smax, a, b)
        m.close()


def test_gather_reorder_on_device():
    """Rank 0's device-side reordering of gathered match lists (the part of the
    multi-GPU path that runs on the GPU under NCCL), fed with fabricated shards.
    In a fresh interpreter: torch brings its own ROCm runtime and has to be
    imported before libosfm_hip.so is loaded, as bench.py does for N > 1."""
    import subprocess
    import sys
    code = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
from orthosfm_amd import distributed as D
if not torch.cuda.is_available():
    print("NOGPU"); raise SystemExit(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(4)
world, num_pairs = 3, 47
counts = rng.integers(0, 900, num_pairs)
lists = [rng.integers(0, 20000, (c, 2)).astype(np.int32) for c in counts]
max_local = len(range(0, num_pairs, world))
heads = torch.zeros((world, max_local), dtype=torch.int64)
shards = []
for r in range(world):
    mine = list(range(r, num_pairs, world))
    heads[r, :len(mine)] = torch.from_numpy(counts[mine])
    shards.append(np.concatenate([lists[g] for g in mine] + [np.zeros((0, 2), np.int32)]))
width = max(s.shape[0] for s in shards)
bufs = []
for s in shards:
    b = torch.zeros((width, 2), dtype=torch.int32, device=dev)
    b[:s.shape[0]] = torch.from_numpy(s).to(dev)
    bufs.append(b)
c, off, corr = D.assemble_global_order(heads.to(dev), bufs, num_pairs, world, dev)
assert np.array_equal(c, counts) and off[-1] == counts.sum()
assert np.array_equal(corr, np.concatenate(lists))
print("OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    if "NOGPU" in out.stdout:
        pytest.skip("torch sees no GPU")
    assert "OK" in out.stdout


def test_lists_written_into_the_shared_segment():
    """The single-node exchange on the GPU side: the matcher's device-to-host copy
    lands in this rank's page-locked slice of the shared segment (world 1 here;
    the cross-rank part is covered by the gloo tests) and rank 0's view of it
    equals the lists compute() returns.  Fresh interpreter: torch first."""
    import subprocess
    import sys
    code = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
from orthosfm_amd import capi, synth, distributed as D
from orthosfm_amd.matching import HipExhaustiveMatching
if not torch.cuda.is_available():
    print("NOGPU"); raise SystemExit(0)
dev = torch.device("cuda:0")
V, F = 5, 1500
iset = synth.make_image_set(V, F, seed=3)
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
m = HipExhaustiveMatching(V, device=0)
for v in range(V):
    m.set_view(v, iset.sift[v])
ref = m.compute(pairs, capacity=F * len(pairs))
ref_lists = [np.array(tv.matches) if tv.status == capi.PAIR_MATCHED else np.zeros((0, 2), np.int32) for tv in ref]
store = D.SharedMatchStore(F * len(pairs), 0, 1, dev)
assert store._registered is not None
store.slice[:] = -5
m.use_result_buffer(store.slice)
out = m.compute(pairs, capacity=F * len(pairs))
counts = np.array([tv.num_matches if tv.status == capi.PAIR_MATCHED else 0 for tv in out], dtype=np.int64)
c, starts, corr = store.collect(counts, len(pairs))
assert counts.sum() > 0 and np.array_equal(c, counts)
for i in range(len(pairs)):
    assert np.array_equal(corr[starts[i]:starts[i] + c[i]], ref_lists[i]), i
assert np.all(np.asarray(corr[int(counts.sum()):]) == -5)
m.close(); store.close()
print("OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    if "NOGPU" in out.stdout:
        pytest.skip("torch sees no GPU")
    assert "OK" in out.stdout


def test_full_size_bit_parity_20k(hm):
    """BASELINE cfg2's per-pair shape at its stated size -- 20000 x 20000 SIFT, 79 row
    blocks x 20 column segments of one 16-tile cycle (a single pair is cut short so that
    it fills the chip), the correction-free kernel every bench number comes
    from -- compared with the oracle list by list (two-way lists, then the
    cross-checked ones), and once more with a few hundred rows holding values > 127
    in both views (special-row blocks + the corrected column operand)."""
    from orthosfm_amd import capi
    iset = synth.make_image_set(2, 20000, config_id=2)
    om = oracle_lib.oracle_matcher()
    m = hm(2)
    o_forms = capi.default_match_options()
    o_forms.special_kernel_max = -1            # "special-forms": the per-view operand forms on the same input
    m_forms = hm(2, options=o_forms)
    expect = None
    for variant in ("plain", "special", "special-forms"):
        s1, s2 = iset.sift[0].copy(), iset.sift[1].copy()
        if variant != "plain":
            r = np.random.default_rng(41)
            for s in (s1, s2):
                for k in r.choice(20000, 400, replace=False):
                    d = s[k].copy()
                    d[r.choice(128, 2, replace=False)] = [int(r.integers(128, 256)), int(r.integers(128, 180))]
                    s[k] = d
        if variant == "special-forms":
            e12, e21, c12, c21 = expect        # same input as "special"
            mm = m_forms
        else:
            e12, e21 = om.twoway(s1, s2, 0.8)
            c12, c21 = om.remove_inconsistent(e12, e21)
            expect = (e12, e21, c12, c21)
            mm = m
        mm.set_view(0, s1)
        mm.set_view(1, s2)
        got = mm.pairwise_match(0, 1)
        assert np.array_equal(got.matches_1_2, c12) and np.array_equal(got.matches_2_1, c21), variant
        assert (mm.stats().special_kernel_launches > 0) == (variant == "special"), variant
        if variant == "plain":
            assert int((c12 >= 0).sum()) > 5000
        else:       # the over-long rows out-score the true partners of most queries: few matches survive
            assert 0 < int((c12 >= 0).sum()) < 5000
        two = mm.twoway_match(0, 1, 0)           # the pre-cross-check seam (masked kernel)
        assert np.array_equal(two.matches_1_2, e12) and np.array_equal(two.matches_2_1, e21), variant
    m.close()
    m_forms.close()


def test_capacity_overflow_with_verification(hm):
    """osfm_match_all with geometric verification, chunks of two pairs and a caller
    buffer that holds only the first chunk: OSFM_E_CAPACITY with the required total,
    no device fault (the later chunks' RANSAC jobs read THEIR chunk's lists), and the
    same call with enough room afterwards gives the full result."""
    from orthosfm_amd import capi
    V, F = 5, 1500
    iset = synth.make_image_set(V, F, config_id=13)
    o = capi.default_match_options()
    o.geometric_verification = 1
    o.pairs_per_batch = 2
    m = hm(V, options=o)
    for v in range(V):
        m.set_view(v, iset.sift[v])
        xy = (iset.pos[v] + 0.5 - np.array([iset.width / 2, iset.height / 2])) / max(iset.width, iset.height)
        m.set_positions(v, xy.astype(np.float32))
    full = m.compute()
    counts = [tv.num_inliers for tv in full if tv.status == capi.PAIR_MATCHED]
    assert len(counts) == 10 and min(counts) > 100
    need = sum(counts)
    small = counts[0] + counts[1] + 3          # room for the first chunk only; later chunks are larger than the slack
    with pytest.raises(capi.OsfmError) as e:
        m.compute(capacity=small)
    assert e.value.status == capi.E_CAPACITY and str(need) in str(e.value)
    again = m.compute(capacity=need)
    for a, b in zip(full, again):
        assert a.status == b.status and np.array_equal(a.matches, b.matches)
    # same without verification (lists are copied chunk by chunk)
    o2 = capi.default_match_options()
    o2.pairs_per_batch = 2
    m2 = hm(V, options=o2)
    for v in range(V):
        m2.set_view(v, iset.sift[v])
    with pytest.raises(capi.OsfmError) as e:
        m2.compute(capacity=100)
    assert e.value.status == capi.E_CAPACITY
    m.close(); m2.close()


def test_matcher_handles_give_their_memory_back():
    """Creating and destroying matchers (cascade-hashing mode included: its per-view
    hash tables and candidate scratch are the largest buffers) leaves the device's
    free memory where it was."""
    from orthosfm_amd import capi
    from orthosfm_amd.matching import HipCascadeHashing, HipExhaustiveMatching
    iset = synth.make_image_set(3, 4000, config_id=14)

    def cycle(cls):
        m = cls(3)
        for v in range(3):
            m.set_view(v, iset.sift[v])
        out = m.compute()
        assert len(out) == 3
        m.close()

    # The library's own books first (osfm_library_memory): after close() nothing of a matcher may be left -- device
    # buffers, page-locked staging, streams, events -- and that is exact, whatever the HIP runtime keeps in pools
    # of its own.  ONE warm-up cycle (code objects, the first stream of the process).
    cycle(HipCascadeHashing); cycle(HipExhaustiveMatching)
    base = capi.library_memory()
    assert base.live_matchers == 0 and base.device_buffer_bytes == 0
    free0, _ = capi.device_memory(0)
    frees = []
    for _ in range(6):
        cycle(HipCascadeHashing)
        cycle(HipExhaustiveMatching)
        r = capi.library_memory()
        assert (r.live_matchers, r.device_buffer_bytes, r.pool_live_bytes) == (0, 0, 0)
        assert (r.pinned_host_bytes, r.live_streams, r.live_events) == (base.pinned_host_bytes, base.live_streams, base.live_events)
        frees.append(capi.device_memory(0)[0])
    # Secondary bound, the free memory of the device: what is missing after the cycles is the runtime's (seen:
    # 67 - 142 MB that appear during the first create / destroy cycles of a process -- with the library's books at
    # zero they are not the matcher's -- and then stay); a leak outside the books would GROW with every cycle.
    # The driver hands freed blocks back with a delay now and then: read a few times.
    import time
    for attempt in range(5):
        free1, _ = capi.device_memory(0)
        if frees[2] - free1 < (8 << 20):
            break
        time.sleep(0.5)
    assert frees[2] - free1 < (8 << 20), (free0, frees, free1)      # no growth over the last three cycles
    assert free0 - free1 < (256 << 20), (free0, frees, free1)       # and the one-time share stays what it is
    print(f"runtime-held after the first cycle: {(free0 - free1) / 2**20:.1f} MB; last three cycles: {(frees[2] - free1) / 2**20:.1f} MB")


def test_multi_device_front_balances_what_the_gate_lets_through(hm):
    """Two scenes in one image set (14 views of one, 10 of another): every pair across the scenes -- 140 of the
    276, a block of the pair matrix, not a sprinkle -- stops at the low-res gate, which costs 1 / 1600 of a full
    match.  The front deals the gate round robin and the SURVIVORS by work: bytes identical to one device, and the
    full-matching work (mac_count) of the logical shards within 5 % of each other (a deal of all pairs by N1 * N2
    in front of the gate left them 20-30 % apart on this set)."""
    from orthosfm_amd import capi
    a = synth.make_image_set(14, 1600, config_id=15)
    b = synth.make_image_set(10, 1600, config_id=16)
    sift = list(a.sift) + list(b.sift)
    V = len(sift)
    single = hm(V, device=0)
    for v in range(V):
        single.set_view(v, sift[v])
    ra1, corr1 = single.compute_arrays()
    ra1, corr1 = ra1.copy(), corr1.copy()
    rejected = int((ra1["status"] == capi.PAIR_REJECTED_LOWRES).sum())
    assert rejected >= 0.3 * ra1.shape[0], (rejected, ra1.shape[0])
    assert int((ra1["status"] == capi.PAIR_MATCHED).sum()) >= 100
    for dev in ([0, 0], [0, 0, 0]):
        m = hm(V, device=dev)
        for v in range(V):
            m.set_view(v, sift[v])
        ra, corr = m.compute_arrays()
        assert ra.tobytes() == ra1.tobytes() and corr.tobytes() == corr1.tobytes(), dev
        macs = [m.shard_stats(k).mac_count for k in range(len(dev))]
        gate = [m.shard_stats(k).lowres_kernel_launches for k in range(len(dev))]
        assert min(gate) >= 1                                  # every shard took its share of the gate
        assert max(macs) <= 1.05 * min(macs), (dev, macs)
        assert sum(macs) == single.stats().mac_count
        m.close()
    single.close()


def test_concurrent_pair_calls_are_combined(hm):
    """The per-pair entries called from many threads at once (the reference's OpenMP loop,
    bundler_matching.cc:86-88): whoever finds no leader at work runs everything that is
    waiting as one batch.  Results equal the serial calls; a call with a bad view id fails
    alone, the calls that would have shared its batch do not."""
    import threading
    from orthosfm_amd import capi
    V = 7
    iset = synth.make_image_set(V, 2500, config_id=23)
    m = hm(V)
    for v in range(V):
        m.set_view(v, iset.sift[v])
    pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
    serial = {}
    for a, b in pairs:
        r = m.pairwise_match(a, b)
        serial[(a, b)] = (m.pairwise_match_lowres(a, b, 500), r.matches_1_2.copy(), r.matches_2_1.copy())
    got, errors = {}, []

    def worker(k):
        try:
            for rep in range(3):
                for a, b in pairs[k::12]:
                    low = m.pairwise_match_lowres(a, b, 500)
                    r = m.pairwise_match(a, b)
                    got[(a, b, rep)] = (low, r.matches_1_2.copy(), r.matches_2_1.copy())
                if k == 0:
                    try:
                        m.pairwise_match(0, V + 3)
                        errors.append("bad view id accepted")
                    except capi.OsfmError as e:
                        if e.status != capi.E_ARG:
                            errors.append(f"bad view id: status {e.status}")
        except Exception as e:       # noqa: BLE001 -- reported below, a thread must not die silently
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(got) == 3 * len(pairs)
    for (a, b, _), (low, m12, m21) in got.items():
        s_low, s12, s21 = serial[(a, b)]
        assert low == s_low and np.array_equal(m12, s12) and np.array_equal(m21, s21), (a, b)
    m.close()


def test_multi_device_matcher_logical_shards(hm):
    """osfm_match_create_multi with device_ids = {0, 0} and {0, 0, 0} (logical shards on the one
    device of the box): every view on every shard, the pairs of compute() dealt by work --
    ragged deals: six views of 400..2600 features --, records and list bytes identical to the
    single-device matcher, with and without RANSAC-F; capacity overflow reports the same
    required total; a failing shard is reported once; the per-pair entries and twoway_match go
    through the front as well."""
    from orthosfm_amd import capi
    sizes = [2600, 400, 1900, 800, 2200, 1300]
    big = synth.make_image_set(6, 2600, n_surf=120, config_id=14)
    sift = [big.sift[v][:sizes[v]] for v in range(6)]
    pos = [big.pos[v][:sizes[v]] for v in range(6)]

    def build(dev, verify):
        o = capi.default_match_options()
        o.geometric_verification = verify
        o.pairs_per_batch = 4                       # several batches per shard
        m = hm(6, device=dev, options=o)
        for v in range(6):
            m.set_view(v, sift[v], big.surf[v] if v % 2 == 0 else None)
            npos = sizes[v] + (120 if v % 2 == 0 else 0)
            xy = np.zeros((npos, 2), np.float32)
            xy[:sizes[v]] = (pos[v] + 0.5 - np.array([big.width / 2, big.height / 2])) / max(big.width, big.height)
            m.set_positions(v, xy)
        return m

    for verify in (0, 1):
        single = build(0, verify)
        ra1, corr1 = single.compute_arrays()
        ra1, corr1 = ra1.copy(), corr1.copy()
        assert int((ra1["status"] == capi.PAIR_MATCHED).sum()) >= 10
        for dev in ([0, 0], [0, 0, 0]):
            m = build(dev, verify)
            assert m.devices() == dev
            ra, corr = m.compute_arrays()
            assert ra.tobytes() == ra1.tobytes(), (verify, dev)
            assert corr.tobytes() == corr1.tobytes(), (verify, dev)
            st = m.stats()
            assert st.tile_kernel_launches >= len(dev)          # every shard worked
            # an arbitrary pair list (a subset, out of order)
            sub = [(4, 1), (2, 0), (5, 3), (3, 0)]
            ra_s, corr_s = m.compute_arrays(sub)
            ra_1, corr_1 = single.compute_arrays(sub)
            assert ra_s.tobytes() == ra_1.tobytes() and corr_s.tobytes() == corr_1.tobytes()
            # capacity: the same error and required total as on one device
            need = corr1.shape[0]
            for mm in (single, m):
                with pytest.raises(capi.OsfmError) as e:
                    mm.compute(capacity=need - 5)
                assert e.value.status == capi.E_CAPACITY and str(need) in str(e.value), str(e.value)
            # per-pair entries and the two-way seam through the front
            a = m.pairwise_match(4, 2)
            b = single.pairwise_match(4, 2)
            assert np.array_equal(a.matches_1_2, b.matches_1_2) and np.array_equal(a.matches_2_1, b.matches_2_1)
            assert m.pairwise_match_lowres(4, 2, 500) == single.pairwise_match_lowres(4, 2, 500)
            t1, t2 = m.twoway_match(0, 2, 1), single.twoway_match(0, 2, 1)
            assert np.array_equal(t1.matches_1_2, t2.matches_1_2) and np.array_equal(t1.matches_2_1, t2.matches_2_1)
            # a failing shard is reported once, with its device
            bad = sift[1].copy()
            bad[0, 0] = 300
            with pytest.raises(capi.OsfmError) as e:
                m.set_view(1, bad)
            assert e.value.status == capi.E_RANGE and str(e.value).count("outside the quantised range") == 1
            assert "shard 0 of %d" % len(dev) in str(e.value)
            with pytest.raises(capi.OsfmError) as e:
                m.compute_arrays([(1, 0)])
            assert e.value.status == capi.E_STATE                     # view 1 is unset on every shard now
            m.set_view(1, sift[1])
            m.set_positions(1, (pos[1] + 0.5 - np.array([big.width / 2, big.height / 2])) / max(big.width, big.height))
            ra2, corr2 = m.compute_arrays()
            assert ra2.tobytes() == ra1.tobytes() and corr2.tobytes() == corr1.tobytes()
            m.close()
        single.close()
    with pytest.raises(capi.OsfmError):
        hm(2, device=[0, 99])



def test_bench_batch_at_full_size_equals_the_oracle(hm):
    """What bench.py times, as a test: a batch of full-size pairs (9 views x 20000 SIFT descriptors, 36
    pairs in one osfm_match_all -- enough row blocks that every pair runs ONE column segment of 313 tiles per row
    block, 320 with the blank filler tiles of the last cycle) through low-res gate, two-way match, cross-check and
    list compaction, eight of its pairs compared list by list with the CPU oracle (nearest_neighbor.cc:60-129,
    matching.cc:18-88, exhaustive_matching.cc:114-180)."""
    from orthosfm_amd import capi
    V, F = 9, 20000
    iset = synth.make_image_set(V, F, config_id=2)
    m = hm(V)
    for v in range(V):
        m.set_view(v, iset.sift[v])
    pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
    out = m.compute(pairs, capacity=F * len(pairs))
    assert m.stats().tile_kernel_launches >= 1
    om = oracle_lib.oracle_matcher()
    empty = np.zeros((0, 64), np.int16)
    checked = 0
    for k in np.linspace(0, len(pairs) - 1, 8).astype(int):
        a, b = pairs[k]
        tv = out[k]
        assert (tv.view_1_id, tv.view_2_id) == (a, b)
        gate = oracle_lib.oracle_pairwise_match_lowres(iset.sift[a], empty, iset.sift[b], empty, 500)
        if gate < 5:
            assert tv.status != capi.PAIR_MATCHED
            continue
        e12, _ = oracle_lib.oracle_pairwise_match(iset.sift[a], empty, iset.sift[b], empty)
        idx = np.nonzero(e12 >= 0)[0]
        if idx.size < 50:
            assert tv.status != capi.PAIR_MATCHED
            continue
        assert tv.status == capi.PAIR_MATCHED
        assert np.array_equal(np.asarray(tv.matches), np.stack([idx, e12[idx]], 1)), (a, b)
        checked += 1
    assert checked >= 6
    # and against the reference's own matcher, compiled from /root/reference into oracle/_ref, where it travelled
    rm = oracle_lib.ref_matcher()
    if rm is not None:
        for k in (3, len(pairs) - 2):
            a, b = pairs[k]
            e12, e21 = rm.twoway(iset.sift[a], iset.sift[b], 0.8)
            c12 = rm.remove_inconsistent(e12, e21)[0]
            idx = np.nonzero(c12 >= 0)[0]
            if out[k].status == capi.PAIR_MATCHED:
                assert np.array_equal(np.asarray(out[k].matches), np.stack([idx, c12[idx]], 1)), (a, b)


def test_ragged_views_in_one_batch_each_with_its_own_segments(hm):
    """Round 5: a correction-free problem's column segment length is chosen per problem and launch (`seg_cols`: as few
    16-tile cycles' worth of segments as still fill the chip), and the tile loop runs whole cycles with blank filler
    tiles behind a segment's last real one.  One batch over views of 1 .. 9000 descriptors -- a single column (fifteen
    and a half filler tiles), sizes just below / at / above a tile, a cycle and a row block, several cycles -- every
    pair, both directions and the cross-check, against the oracle; then the same views pair by pair through the
    per-pair entry, which cuts a pair into many short segments."""
    from orthosfm_amd import capi
    sizes = [1, 63, 64, 65, 255, 257, 1023, 1024, 1025, 2049, 5000, 9000]
    base = synth.make_image_set(len(sizes), 9000, config_id=14, twin_frac=0.3)
    sift = [base.sift[v][:n].copy() for v, n in enumerate(sizes)]
    o = capi.default_match_options()
    o.use_lowres_matching = 0
    o.min_feature_matches = 0
    m = hm(len(sizes), options=o)
    for v, s in enumerate(sift):
        m.set_view(v, s)
    out = m.compute()
    assert len(out) == len(sizes) * (len(sizes) - 1) // 2
    empty = np.zeros((0, 64), np.int16)
    expect = {}
    for tv in out:
        a, b = tv.view_1_id, tv.view_2_id
        e12, _ = oracle_lib.oracle_pairwise_match(sift[a], empty, sift[b], empty)
        idx = np.nonzero(e12 >= 0)[0]
        expect[(a, b)] = np.stack([idx, e12[idx]], axis=1).astype(np.int32)
        if expect[(a, b)].shape[0] < 8:            # bundler_matching.cc:150-153: fewer than eight matches are no pair
            assert tv.status != capi.PAIR_MATCHED, (a, b)
            continue
        assert tv.status == capi.PAIR_MATCHED and np.array_equal(np.asarray(tv.matches).reshape(-1, 2), expect[(a, b)]), \
            (a, b, sizes[a], sizes[b])
    assert sum(1 for e in expect.values() if e.shape[0] >= 8) >= 30
    # the per-pair entry returns the lists whatever their length: every pair with a small view, and some large ones
    small = [(a, b) for (a, b) in expect if min(sizes[a], sizes[b]) <= 257]
    for a, b in small + [(11, 10), (9, 5), (10, 8)]:
        got = m.pairwise_match(a, b)
        c12 = np.full(sizes[a], -1, np.int32)
        c12[expect[(a, b)][:, 0]] = expect[(a, b)][:, 1]
        assert np.array_equal(got.matches_1_2, c12), (a, b)
    m.close()
