"""CPU-only checks of the drop-in boundary: libosfm_hip.so loads, exports
every symbol include/osfm_hip.h declares, and -- there being no CPU fallback
-- refuses to compute without a HIP device.  No kernel runs here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "osfm_hip.h")).read()
    return sorted(set(re.findall(r"OSFM_API\s+[\w\s\*]+?\b(osfm_\w+)\s*\(", text)))


def test_header_symbols_exported():
    from orthosfm_amd import capi
    names = _declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(capi.lib, n)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == names


def test_defaults_match_reference_constants():
    """matching_base.h:27-30, matching_mve.cpp:395-404, bundle_adjustment.cpp:126-133."""
    from orthosfm_amd import capi
    from orthosfm_amd import ba
    o = capi.default_match_options()
    assert np.float32(o.sift_lowe_ratio) == np.float32(0.8) and np.float32(o.surf_lowe_ratio) == np.float32(0.7)
    assert o.sift_distance_threshold == np.finfo(np.float32).max
    assert (o.use_lowres_matching, o.num_lowres_features, o.min_lowres_matches, o.min_feature_matches) == (1, 500, 5, 50)
    b = ba.default_options()
    assert (b.huber_delta, b.function_tolerance, b.gradient_tolerance, b.parameter_tolerance) == (1.0, 1e-6, 1e-10, 1e-10)
    assert (b.max_num_iterations, b.initial_trust_region_radius, b.min_relative_decrease) == (100, 1e4, 1e-3)


def test_host_quantisation_equals_oracle():
    """osfm_quantize_* (host part of MatchingBase::init) vs the oracle's A1."""
    import oracle_lib
    from orthosfm_amd import capi
    r = np.random.default_rng(0)
    f = (r.standard_normal((300, 128)) * 0.3).astype(np.float32)
    f[0, :8] = [0.0, 1.0, 0.5, 0.0019607844, 0.00196, 0.998, 2.0, -1.0]
    assert np.array_equal(capi.quantize_sift(f), oracle_lib.oracle_convert_sift(f))
    g = (r.standard_normal((200, 64)) * 0.5).astype(np.float32)
    g[0, :6] = [-1.0, 1.0, 0.5 / 127, -0.5 / 127, 1.5 / 127, -3.0]
    assert np.array_equal(capi.quantize_surf(g), oracle_lib.oracle_convert_surf(g))


def test_pair_enumeration_matches_reference_formula():
    from orthosfm_amd import capi
    seen = set()
    V = 40
    for i in range(V * (V - 1) // 2):
        a, b = capi.pair_from_index(i)
        assert 0 <= b < a < V
        seen.add((a, b))
    assert len(seen) == V * (V - 1) // 2


def test_no_device_is_a_loud_error():
    """On a box without a GPU every compute entry point must fail, never fall
    back.  (Skipped where a device exists.)"""
    from orthosfm_amd import capi
    if capi.device_count() > 0:
        pytest.skip("HIP device present")
    from orthosfm_amd.matching import HipExhaustiveMatching
    with pytest.raises(capi.OsfmError) as e:
        HipExhaustiveMatching(2)
    assert e.value.status == capi.E_DEVICE
    from orthosfm_amd import ba, synth
    sc = synth.make_ba_scene(0, 3, 10, config_id=50)
    with pytest.raises(capi.OsfmError) as e:
        ba.solve(ba.FlatProblem.from_scene(sc))
    assert e.value.status == capi.E_DEVICE


def test_product_does_not_touch_the_oracle():
    """The product package must not import / link the test oracle."""
    pkg = os.path.join(ROOT, "orthosfm_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_lib" not in src and "liboracle" not in src and "oracle/" not in src, fn


def test_m0_is_written_only_by_hand_in_the_pipe_kernels():
    """The correction-free tile kernels and match_special_wide_kernel set m0 inside an asm statement (hipcc rejects
    m0 as a clobber: a reserved register).  That is sound only while nothing else in those kernels uses m0;
    tools/check_m0.sh disassembles match_kernels.hip and match_special.hip for gfx950 and fails when one of them
    holds any other m0 instruction."""
    import os
    import shutil
    import subprocess
    import pytest
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([os.path.join(root, "tools", "check_m0.sh")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("[hand-written m0]") >= 3 and "0 other than s_mov_b32 m0 [hand-written m0]" in out.stdout
    assert "match_special_wide_kernel" in out.stdout


def test_a_point_observed_twice_by_one_camera_is_refused():
    """the Schur fast paths keep one slot per camera and track, so a repeated (camera, point)
    observation would drop entries silently.  The reference never builds one (a track with two features of a view
    is a conflict and dropped, bundler_tracks.cc:120-145): the boundary refuses it, before any device work."""
    from orthosfm_amd import ba, capi, synth
    sc = synth.make_ba_scene(0, 3, 10, config_id=50)
    fp = ba.FlatProblem.from_scene(sc)
    k = int(np.flatnonzero(fp.obs_point == fp.obs_point[0])[1])       # second observation of the first track
    fp.obs_camera[k] = fp.obs_camera[0]
    with pytest.raises(capi.OsfmError) as e:
        ba.solve(fp)
    assert e.value.status == capi.E_ARG and "observed twice" in str(e.value)
    # the scene entry: a track with two features of one view
    scn = C.c_void_p()
    offs = np.array([0, 2, 4], dtype=np.int64)
    view = np.array([0, 1, 1, 1], dtype=np.int32)
    xy = np.zeros((4, 2), dtype=np.float32)
    wh = np.array([100, 100], dtype=np.int32)
    st = capi.lib.osfm_scene_create(0, 0, 2, wh.ctypes.data_as(C.c_void_p), wh.ctypes.data_as(C.c_void_p), 2,
                                    offs.ctypes.data_as(C.c_void_p), view.ctypes.data_as(C.c_void_p),
                                    xy.ctypes.data_as(C.c_void_p), C.byref(scn))
    assert st == capi.E_ARG and "two features of view 1" in capi.last_error()


def test_abi_version_of_library_header_and_mirror_agree():
    """public structs grew; a caller built against another header must be able to tell."""
    from orthosfm_amd import capi
    text = open(os.path.join(ROOT, "include", "osfm_hip.h")).read()
    v = int(re.search(r"#define\s+OSFM_ABI_VERSION\s+(\d+)", text).group(1))
    assert capi.lib.osfm_version() == v == capi.ABI_VERSION
    for fn in ("ba_hip_adapter.h", "mve_hip_matching.h"):
        assert "OSFM_ABI_VERSION" in open(os.path.join(ROOT, "orthosfm_amd", "host", fn)).read()


def test_handoff_order_of_the_ba_kernels_in_the_isa():
    """The tickets of the pair / back passes and the flags of the one-launch Cholesky are relaxed agent-scope
    atomics ordered by the instruction sequence (sc1 stores, drained vmcnt, barrier, signal; sc1 loads behind the
    poll).  tools/check_handoff.sh disassembles ba_kernels.hip / ba_cholesky.hip for gfx950 and checks that
    sequence -- and that chol_flow_kernel spills nothing into scratch memory, which once made its hand-offs race."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this machine")
    out = subprocess.run([os.path.join(ROOT, "tools", "check_handoff.sh")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "chol_flow_kernel             scratch bytes 0" in out.stdout
    assert out.stdout.count("0 not behind a drained vmcnt") == 4 and "FAIL" not in out.stdout
