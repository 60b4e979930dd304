"""ctypes binding of the C ABI in include/osfm_hip.h (libosfm_hip.so).

There is no fallback: if the HIP library is not built the import of this
module raises, and every compute entry point raises OsfmError when the
library reports an error (e.g. no gfx950 device).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSFM_HIP_LIBRARY: another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("OSFM_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libosfm_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). orthosfm_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)


class OsfmError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"osfm status {status}: {msg}")
        self.status = status


OK, E_ARG, E_DEVICE, E_RANGE, E_CAPACITY, E_STATE, E_NUMERIC, E_IO = 0, -1, -2, -3, -4, -5, -6, -7
MATCHER_EXHAUSTIVE, MATCHER_CASCADE_HASHING = 0, 1
PAIR_MATCHED, PAIR_REJECTED_LOWRES, PAIR_REJECTED_COUNT, PAIR_SKIPPED_EMPTY, PAIR_REJECTED_INLIERS = 0, 1, 2, 3, 4


class MatchOptions(C.Structure):
    _fields_ = [("sift_lowe_ratio", C.c_float), ("sift_distance_threshold", C.c_float),
                ("surf_lowe_ratio", C.c_float), ("surf_distance_threshold", C.c_float),
                ("use_lowres_matching", C.c_int32), ("num_lowres_features", C.c_int32),
                ("min_lowres_matches", C.c_int32), ("min_feature_matches", C.c_int32),
                ("pairs_per_batch", C.c_int32),
                ("geometric_verification", C.c_int32), ("ransac_max_iterations", C.c_int32),
                ("ransac_threshold", C.c_double), ("min_matching_inliers", C.c_int32),
                ("matcher_type", C.c_int32), ("ransac_seed", C.c_uint64),
                ("cascade_keep_empty_blocks", C.c_int32), ("special_kernel_max", C.c_int32)]


class RansacOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("reserved", C.c_int32),
                ("threshold", C.c_double), ("seed", C.c_uint64)]


class Pair(C.Structure):
    _fields_ = [("view_1", C.c_int32), ("view_2", C.c_int32)]


class PairResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("lowres_matches", C.c_int32),
                ("num_matches", C.c_int32), ("num_inliers", C.c_int32), ("offset", C.c_int64)]


class MatchStats(C.Structure):
    _fields_ = [("tile_kernel_ms", C.c_double), ("tile_kernel_launches", C.c_int32),
                ("exact_scan_queries", C.c_int32), ("mac_count", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("lowres_kernel_ms", C.c_double),
                ("lowres_kernel_launches", C.c_int32), ("reserved", C.c_int32),
                ("lowres_mac_count", C.c_int64), ("cashash_kernel_ms", C.c_double),
                ("cashash_kernel_launches", C.c_int32), ("special_kernel_launches", C.c_int32),
                ("special_kernel_ms", C.c_double), ("tile_shader_cycles", C.c_double), ("tile_refclk_ticks", C.c_double),
                ("surf_tile_kernel_ms", C.c_double), ("surf_tile_kernel_launches", C.c_int32), ("reserved2", C.c_int32),
                ("surf_mac_count", C.c_int64)]


class MemoryReport(C.Structure):
    _fields_ = [("device_buffer_bytes", C.c_int64), ("pool_live_bytes", C.c_int64), ("pool_cached_bytes", C.c_int64),
                ("pinned_host_bytes", C.c_int64), ("live_matchers", C.c_int32), ("live_streams", C.c_int32),
                ("live_events", C.c_int32), ("reserved", C.c_int32)]


class BaProblem(C.Structure):
    _fields_ = [("model", C.c_int32), ("num_cameras", C.c_int32), ("num_points", C.c_int32),
                ("num_observations", C.c_int32),
                ("cam_params", C.c_void_p), ("cam_const", C.c_void_p),
                ("img_width", C.c_void_p), ("img_height", C.c_void_p),
                ("points", C.c_void_p), ("obs_xy", C.c_void_p),
                ("obs_camera", C.c_void_p), ("obs_point", C.c_void_p)]


class BaOptions(C.Structure):
    _fields_ = [("huber_delta", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("max_num_iterations", C.c_int32), ("optimize_points", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int32), ("max_consecutive_invalid_steps", C.c_int32),
                ("device", C.c_int32), ("verbose", C.c_int32),
                ("retriangulate_points", C.c_int32), ("reserved", C.c_int32)]


class TracksSummary(C.Structure):
    _fields_ = [("num_tracks", C.c_int32), ("num_invalid_tracks", C.c_int32), ("num_features", C.c_int64)]


class OutlierStats(C.Structure):
    _fields_ = [("mean", C.c_double), ("sigma", C.c_double),
                ("num_with_point", C.c_int32), ("num_kept", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("termination", C.c_int32),
                ("mean_point_change", C.c_double), ("max_point_change", C.c_double),
                ("solve_ms", C.c_double), ("point_pass_ms", C.c_double),
                ("pair_pass_ms", C.c_double), ("cholesky_ms", C.c_double),
                ("back_pass_ms", C.c_double),
                ("linearizations", C.c_int32), ("num_pair_entries", C.c_int32),
                ("lm_loop_ms", C.c_double),
                ("flow_fallbacks", C.c_int32), ("order_arcs", C.c_int32),
                ("chain_blocks_natural", C.c_int32), ("chain_blocks", C.c_int32)]


# every symbol include/osfm_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "osfm_last_error", "osfm_version", "osfm_device_count", "osfm_device_memory", "osfm_library_memory", "osfm_trim_device_memory", "osfm_ransac_selfcheck", "osfm_ba_debug_chol_trace", "osfm_ba_debug_flow_spin_limit", "osfm_ba_debug_order",
    "osfm_match_options_default", "osfm_match_create", "osfm_match_create_multi", "osfm_match_get_devices", "osfm_match_destroy",
    "osfm_quantize_sift", "osfm_quantize_surf",
    "osfm_match_set_view", "osfm_match_set_view_float", "osfm_match_view_size", "osfm_match_expect_pairs", "osfm_match_set_positions",
    "osfm_ransac_options_default", "osfm_ransac_fundamental",
    "osfm_match_pair", "osfm_match_pair_lowres", "osfm_match_twoway", "osfm_match_all",
    "osfm_pair_from_index", "osfm_match_get_stats", "osfm_match_get_shard_stats", "osfm_match_get_cascade_hashes",
    "osfm_ba_options_default", "osfm_ba_solve", "osfm_ba_reprojection_errors",
    "osfm_ba_triangulate",
    "osfm_nn_distances", "osfm_filter_outlier_tracks", "osfm_filter_reprojection",
    "osfm_scene_create", "osfm_scene_destroy", "osfm_scene_set_flags", "osfm_scene_align_views", "osfm_scene_set_cameras", "osfm_scene_get_cameras",
    "osfm_scene_triangulate", "osfm_scene_filter_reprojection", "osfm_scene_local_adjustment", "osfm_scene_global_adjustment",
    "osfm_scene_filter_outliers", "osfm_scene_download",
    "osfm_tracks_compute", "osfm_tracks_compute_ranges", "osfm_build_groups",
    "osfm_tracks_builder_create", "osfm_tracks_builder_feed", "osfm_tracks_builder_finish", "osfm_tracks_builder_destroy",
    "osfm_tracks_select_observations",
    "osfm_tracks_feature_table",
    "osfm_tracks_file_write", "osfm_tracks_file_read", "osfm_tracks_pairwise_files_write",
    "osfm_tracks_from_mve", "osfm_cameras_file_write", "osfm_cameras_file_read",
    "osfm_sparse_cloud_write", "osfm_time_measurements_write", "osfm_time_measurements_read",
]

# osfm_track_feature as a numpy record (32 bytes, no padding)
TRACK_FEATURE = np.dtype([("view_id", "<u4"), ("local_feature_id", "<u4"), ("global_feature_id", "<u4"),
                          ("x", "<f4"), ("y", "<f4"), ("r", "<u4"), ("g", "<u4"), ("b", "<u4")])

lib.osfm_last_error.restype = C.c_char_p
lib.osfm_version.restype = C.c_int
# include/osfm_hip.h OSFM_ABI_VERSION: the structs below mirror THAT header, and the library writes them in full
ABI_VERSION = 102
if lib.osfm_version() != ABI_VERSION and os.environ.get("OSFM_ALLOW_ABI_MISMATCH") != "1":
    raise ImportError(f"libosfm_hip.so reports ABI version {lib.osfm_version()}, orthosfm_amd/capi.py mirrors {ABI_VERSION} "
                      "(rebuild: make -C orthosfm_amd/csrc; experiments with an older build: OSFM_ALLOW_ABI_MISMATCH=1)")
lib.osfm_device_count.restype = C.c_int


def last_error() -> str:
    return lib.osfm_last_error().decode()


def check(status):
    if status != OK:
        raise OsfmError(status, last_error())


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None and a.size else None


def device_count() -> int:
    return lib.osfm_device_count()


def ransac_selfcheck(mode: int):
    """Sets the scoring mode of the RANSAC kernel (0 double, 1 pre-classified, 2 checked) and
    returns (wrong, undecided, tests) counted in mode 2 since the previous call."""
    c = (C.c_uint64 * 3)()
    check(lib.osfm_ransac_selfcheck(C.c_int(mode), c))
    return int(c[0]), int(c[1]), int(c[2])


_pinned_blocks = {}        # address -> the ctypes view that keeps the block reachable


def pinned_rows(rows: int):
    """A (rows, 2) int32 array in page-locked host memory (hipHostMalloc of the HIP runtime the
    library is linked against), e.g. as the result buffer of HipExhaustiveMatching.compute: the
    match lists then leave the device at full PCIe rate instead of through pageable staging.
    Lives until pinned_free (or as long as the process)."""
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    nbytes = max(int(rows), 1) * 8
    rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(nbytes), C.c_uint(0))
    if rc != 0 or not p.value:
        raise OsfmError(E_DEVICE, f"hipHostMalloc({nbytes}) failed with {rc}")
    buf = (C.c_int32 * (max(int(rows), 1) * 2)).from_address(p.value)
    _pinned_blocks[p.value] = buf
    return np.frombuffer(buf, dtype=np.int32).reshape(-1, 2)[:rows]


def pinned_free(arr) -> bool:
    """Gives a pinned_rows block back (the caller drops its views of it); False when it is not one."""
    addr = int(arr.ctypes.data)
    if _pinned_blocks.pop(addr, None) is None:
        return False
    C.CDLL("libamdhip64.so").hipHostFree(C.c_void_p(addr))
    return True


def trim_device_memory(device: int = -1) -> int:
    """Hands the cached work-array memory back to the driver; returns the bytes released."""
    b = C.c_uint64()
    check(lib.osfm_trim_device_memory(C.c_int(device), C.byref(b)))
    return b.value


def device_memory(device: int = 0):
    """(free, total) bytes of the device's HBM."""
    f, t = C.c_uint64(), C.c_uint64()
    check(lib.osfm_device_memory(device, C.byref(f), C.byref(t)))
    return f.value, t.value


def library_memory() -> MemoryReport:
    """What the library itself holds in this process (osfm_library_memory)."""
    r = MemoryReport()
    check(lib.osfm_library_memory(C.byref(r)))
    return r


def default_match_options() -> MatchOptions:
    o = MatchOptions()
    check(lib.osfm_match_options_default(C.byref(o)))
    return o


def quantize_sift(f: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(f, dtype=np.float32).reshape(-1, 128)
    out = np.zeros(f.shape, dtype=np.uint16)
    check(lib.osfm_quantize_sift(_ptr(f, C.c_float), f.shape[0], _ptr(out, C.c_uint16)))
    return out


def quantize_surf(f: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(f, dtype=np.float32).reshape(-1, 64)
    out = np.zeros(f.shape, dtype=np.int16)
    check(lib.osfm_quantize_surf(_ptr(f, C.c_float), f.shape[0], _ptr(out, C.c_int16)))
    return out


def pair_from_index(i: int):
    a, b = C.c_int32(), C.c_int32()
    check(lib.osfm_pair_from_index(C.c_int64(i), C.byref(a), C.byref(b)))
    return a.value, b.value
