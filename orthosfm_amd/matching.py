"""Host-side mirror of the reference's matcher plug-in interface on top of
the C ABI: sfm::MatchingBase { init, pairwise_match, pairwise_match_lowres }
(src/mve/sfm/matching_base.h:22-55) and the all-pairs driver
sfm::bundler::Matching::compute (src/mve/sfm/bundler_matching.cc:58-136,
up to the RANSAC stage).  Same names, argument meaning and error behaviour
(exceptions for invalid arguments, -1 for unsuccessful matches).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class MatchResult:
    """sfm::Matching::Result (src/mve/sfm/matching.h:56-62)."""
    matches_1_2: np.ndarray
    matches_2_1: np.ndarray


_PAIR_RESULT_DTYPE = np.dtype([("status", np.int32), ("lowres_matches", np.int32), ("num_matches", np.int32),
                               ("num_inliers", np.int32), ("offset", np.int64)])
assert _PAIR_RESULT_DTYPE.itemsize == C.sizeof(capi.PairResult)


@dataclass
class TwoViewMatching:
    """sfm::bundler::TwoViewMatching (bundler_common.h:118-125) before RANSAC."""
    view_1_id: int
    view_2_id: int
    matches: np.ndarray          # (k, 2) int32 (feature in view 1, feature in view 2)
    status: int = 0
    lowres_matches: int = -1
    num_matches: int = 0
    num_inliers: int = -1


class HipExhaustiveMatching:
    """Drop-in for sfm::ExhaustiveMatching (src/mve/sfm/exhaustive_matching.h:26-66)
    backed by the gfx950 kernels."""

    def __init__(self, num_views: int, device=0, options: capi.MatchOptions | None = None,
                 copy_results: bool = True):
        """copy_results=False: the lists compute() returns are views into a buffer
        the matcher reuses -- valid until the next compute() (no second pass over
        the results on the host).  device: one device id, or a sequence of them
        (osfm_match_create_multi: the bank on every device, the pairs of compute() dealt
        over them by work; an id may repeat -- logical shards on one device)."""
        self.opts = options if options is not None else capi.default_match_options()
        self._copy_results = bool(copy_results)
        self._h = C.c_void_p()
        if isinstance(device, (list, tuple, np.ndarray)):
            ids = (C.c_int * len(device))(*[int(d) for d in device])
            capi.check(capi.lib.osfm_match_create_multi(ids, len(device), num_views, C.byref(self.opts), C.byref(self._h)))
        else:
            capi.check(capi.lib.osfm_match_create(int(device), num_views, C.byref(self.opts), C.byref(self._h)))
        self.num_views = num_views

    def devices(self):
        """The device of every shard (osfm_match_get_devices)."""
        ids = (C.c_int32 * 64)()
        n = C.c_int32()
        capi.check(capi.lib.osfm_match_get_devices(self._h, ids, 64, C.byref(n)))
        return [int(ids[k]) for k in range(min(n.value, 64))]

    # --- MatchingBase::init ---------------------------------------------------
    def init(self, viewports):
        """viewports: sequence of (sift_float[n,128], surf_float[m,64]) -- the
        FeatureSet::sift_descriptors / surf_descriptors data rows."""
        if viewports is None:
            raise ValueError("Viewports must not be null")   # bundler_matching.cc:47-48
        for v, (sift, surf) in enumerate(viewports):
            self.set_view_float(v, sift, surf)

    def set_view_float(self, view, sift, surf):
        sift = np.ascontiguousarray(sift, dtype=np.float32).reshape(-1, 128)
        surf = np.ascontiguousarray(surf, dtype=np.float32).reshape(-1, 64)
        capi.check(capi.lib.osfm_match_set_view_float(
            self._h, view, capi._ptr(sift, C.c_float), sift.shape[0],
            capi._ptr(surf, C.c_float), surf.shape[0]))

    def set_view(self, view, sift_u16, surf_s16=None):
        sift = np.ascontiguousarray(sift_u16, dtype=np.uint16).reshape(-1, 128)
        surf = (np.zeros((0, 64), np.int16) if surf_s16 is None
                else np.ascontiguousarray(surf_s16, dtype=np.int16).reshape(-1, 64))
        capi.check(capi.lib.osfm_match_set_view(
            self._h, view, capi._ptr(sift, C.c_uint16), sift.shape[0],
            capi._ptr(surf, C.c_int16), surf.shape[0]))

    def expect_pairs(self, pairs_per_call: int):
        """The largest compute() call to come: the work arrays are sized for it at once (osfm_match_expect_pairs)."""
        capi.check(capi.lib.osfm_match_expect_pairs(self._h, C.c_int32(int(pairs_per_call))))

    def set_positions(self, view, xy):
        """FeatureSet::positions (normalised x, y per feature) for RANSAC-F."""
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        capi.check(capi.lib.osfm_match_set_positions(self._h, view, capi._ptr(xy, C.c_float), xy.shape[0]))

    def view_size(self, view):
        a, b = C.c_int(), C.c_int()
        capi.check(capi.lib.osfm_match_view_size(self._h, view, C.byref(a), C.byref(b)))
        return a.value, b.value

    # --- MatchingBase::pairwise_match -----------------------------------------
    def pairwise_match(self, view_1_id, view_2_id) -> MatchResult:
        n1 = sum(self.view_size(view_1_id))
        n2 = sum(self.view_size(view_2_id))
        m12 = np.full(max(n1, 1), -7, dtype=np.int32)
        m21 = np.full(max(n2, 1), -7, dtype=np.int32)
        l12, l21 = C.c_int32(), C.c_int32()
        capi.check(capi.lib.osfm_match_pair(
            self._h, view_1_id, view_2_id, capi._ptr(m12, C.c_int32), C.byref(l12),
            capi._ptr(m21, C.c_int32), C.byref(l21)))
        return MatchResult(m12[:l12.value].copy(), m21[:l21.value].copy())

    # --- MatchingBase::pairwise_match_lowres ----------------------------------
    def pairwise_match_lowres(self, view_1_id, view_2_id, num_features) -> int:
        c = C.c_int32()
        capi.check(capi.lib.osfm_match_pair_lowres(self._h, view_1_id, view_2_id, num_features, C.byref(c)))
        return c.value

    # --- Matching::twoway_match (matching.h:148-159), one descriptor type -------
    def twoway_match(self, view_1_id, view_2_id, descriptor_type=0, num_features=0) -> MatchResult:
        n1 = self.view_size(view_1_id)[descriptor_type]
        n2 = self.view_size(view_2_id)[descriptor_type]
        if num_features > 0:
            n1, n2 = min(n1, num_features), min(n2, num_features)
        m12 = np.full(max(n1, 1), -7, dtype=np.int32)
        m21 = np.full(max(n2, 1), -7, dtype=np.int32)
        capi.check(capi.lib.osfm_match_twoway(self._h, view_1_id, view_2_id, descriptor_type,
                                              num_features, capi._ptr(m12, C.c_int32),
                                              capi._ptr(m21, C.c_int32)))
        return MatchResult(m12[:n1].copy(), m21[:n2].copy())

    # --- bundler::Matching::compute (pre-RANSAC part), batched ----------------
    def compute_arrays(self, pairs=None, capacity=None):
        """osfm_match_all as it is: returns (records, corr) -- one record per input pair (status,
        lowres_matches, num_matches, num_inliers, offset: numpy fields of osfm_pair_result) and the
        (rows, 2) buffer all lists lie in, pair k at corr[offset : offset + count].  `pairs` may be
        an (n, 2) int32 array; the marshalled form of the last pair list is kept and reused when
        the CONTENT is the same (a list mutated in place is marshalled again).  The records are a
        fresh array per call; as_objects() takes the pair list from the call it is given."""
        if pairs is None:
            pairs = [capi.pair_from_index(i) for i in range(self.num_views * (self.num_views - 1) // 2)]
        flat = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        cached = getattr(self, "_pairs_cache", None)
        if cached is not None and cached[1].shape == flat.shape and np.array_equal(cached[1], flat):
            arr, flat = cached
        else:
            n = flat.shape[0]
            arr = (capi.Pair * max(n, 1))()
            flat = flat.copy()
            if n:
                np.ctypeslib.as_array(C.cast(arr, C.POINTER(C.c_int32)), shape=(n, 2))[:] = flat
            self._pairs_cache = (arr, flat)
        n = flat.shape[0]
        res = (capi.PairResult * max(n, 1))()
        if capacity is None:
            capacity = sum(min(sum(self.view_size(a)), sum(self.view_size(b))) for a, b in flat.tolist())
        # One result buffer per call, taken from a pool the matcher keeps (its pages
        # stay mapped, so the device-to-host copy does not fault them in again);
        # the per-pair lists are views into it.
        corr = self._take_buffer(max(capacity, 1))
        total = C.c_int64()
        capi.check(capi.lib.osfm_match_all(self._h, arr, n, res, capi._ptr(corr, C.c_int32),
                                           C.c_int64(capacity), C.byref(total)))
        corr = corr[:max(int(total.value), 0)].copy() if self._copy_results else corr
        # all lists of the call, concatenated in pair order (what the per-pair views point into)
        self.last_flat = corr[:max(int(total.value), 0)]
        ra = np.frombuffer(res, dtype=_PAIR_RESULT_DTYPE, count=n)      # keeps `res` alive; one per call
        self._last_pairs_flat = flat
        return ra, corr

    def compute(self, pairs=None, capacity=None):
        """Matches `pairs` (default: all V(V-1)/2 pairs in the reference's
        triangular order, view_1 > view_2).  Returns one TwoViewMatching per
        input pair, in input order."""
        ra, corr = self.compute_arrays(pairs, capacity)
        return self.as_objects(ra, corr)

    def as_objects(self, ra, corr, pairs=None):
        """The TwoViewMatching list of what compute_arrays returned (pairs: the list that call
        was given; default: the most recent call's)."""
        flat = self._last_pairs_flat if pairs is None else np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        assert flat.shape[0] == len(ra), "as_objects: records and pair list of different calls"
        n = flat.shape[0]
        empty = np.zeros((0, 2), np.int32)
        verify = bool(self.opts.geometric_verification)
        # the result records as columns (one pass each) instead of 7 ctypes field reads per pair
        status, lowres = ra["status"].tolist(), ra["lowres_matches"].tolist()
        nm, ni, off = ra["num_matches"].tolist(), ra["num_inliers"].tolist(), ra["offset"].tolist()
        v1, v2 = flat[:, 0].tolist(), flat[:, 1].tolist()
        cnt = ni if verify else nm
        out = [TwoViewMatching(v1[k], v2[k],
                               corr[off[k]:off[k] + cnt[k]] if status[k] == capi.PAIR_MATCHED else empty,
                               status[k], lowres[k], nm[k], ni[k]) for k in range(n)]
        return out

    @staticmethod
    def ransac_fundamental(pos1, pos2, corr, max_iterations=1000, threshold=0.0015, seed=0, pair_id=0, device=0):
        """sfm::RansacFundamental::estimate for one pair; returns (inlier ids, F)."""
        pos1 = np.ascontiguousarray(pos1, dtype=np.float32).reshape(-1, 2)
        pos2 = np.ascontiguousarray(pos2, dtype=np.float32).reshape(-1, 2)
        corr = np.ascontiguousarray(corr, dtype=np.int32).reshape(-1, 2)
        o = capi.RansacOptions(max_iterations, 0, threshold, seed)
        inl = np.zeros(max(corr.shape[0], 1), dtype=np.int32)
        n = C.c_int32()
        F = np.zeros(9)
        capi.check(capi.lib.osfm_ransac_fundamental(
            device, capi._ptr(pos1, C.c_float), pos1.shape[0], capi._ptr(pos2, C.c_float), pos2.shape[0],
            capi._ptr(corr, C.c_int32), corr.shape[0], C.byref(o), C.c_uint64(pair_id),
            capi._ptr(inl, C.c_int32), C.byref(n), capi._ptr(F, C.c_double)))
        return n.value, inl[:max(n.value, 0)].copy(), F.reshape(3, 3)

    def use_result_buffer(self, buf):
        """Lets the caller provide the (rows, 2) int32 array compute() writes the match
        lists into, e.g. page-locked memory (orthosfm_amd.distributed.pinned_array)."""
        assert buf.dtype == np.int32 and buf.ndim == 2 and buf.shape[1] == 2 and buf.flags.c_contiguous
        self._corr_buf = buf

    def _take_buffer(self, rows):
        buf = getattr(self, "_corr_buf", None)
        if buf is None or buf.shape[0] < rows:
            buf = np.empty((rows, 2), dtype=np.int32)
            self._corr_buf = buf
        return buf[:rows]

    def cascade_hashes(self, view, type_=0):
        """CascadeHashing::LocalData of a view: (hashes (n, words) uint64, bucket ids (6, n) uint8)."""
        n = self.view_size(view)[type_]
        words = 2 if type_ == 0 else 1
        hashes = np.zeros((max(n, 1), words), np.uint64)
        ids = np.zeros((6, max(n, 1)), np.uint8)
        tmp = np.zeros(6 * max(n, 1), np.uint8)
        capi.check(capi.lib.osfm_match_get_cascade_hashes(self._h, view, type_, capi._ptr(hashes, C.c_uint64),
                                                          capi._ptr(tmp, C.c_uint8)))
        if n:
            ids = tmp[:6 * n].reshape(6, n)
        return hashes[:n], ids[:, :n]

    def stats(self) -> capi.MatchStats:
        s = capi.MatchStats()
        capi.check(capi.lib.osfm_match_get_stats(self._h, C.byref(s)))
        return s

    def shard_stats(self, shard: int) -> capi.MatchStats:
        """What one shard of a multi-device matcher did in the most recent call."""
        s = capi.MatchStats()
        capi.check(capi.lib.osfm_match_get_shard_stats(self._h, C.c_int(shard), C.byref(s)))
        return s

    def close(self):
        if self._h:
            capi.lib.osfm_match_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipCascadeHashing(HipExhaustiveMatching):
    """Drop-in for sfm::CascadeHashing (src/mve/sfm/cascade_hashing.h:29-221), the
    matcher the application selects (matching_mve.cpp:406-408): pairwise_match
    is the approximate cascade-hashing search, pairwise_match_lowres the
    exhaustive one it inherits."""

    def __init__(self, num_views: int, device: int = 0, options: capi.MatchOptions | None = None,
                 copy_results: bool = True):
        opts = options if options is not None else capi.default_match_options()
        opts.matcher_type = capi.MATCHER_CASCADE_HASHING
        super().__init__(num_views, device, opts, copy_results)
