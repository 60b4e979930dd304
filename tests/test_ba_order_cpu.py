"""The elimination order of the reduced camera system (orthosfm_amd/csrc/ba_order.hip) is host code: checked here
without a device through osfm_ba_debug_order.  The reference leaves this to CHOLMOD's ordering behind Ceres'
SPARSE_SCHUR (bundle_adjustment.cpp:126-133)."""
import ctypes as C

import numpy as np
import pytest


def _order(ldim, pairs):
    from orthosfm_amd import capi
    ldim = np.ascontiguousarray(ldim, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    off = np.zeros(ldim.shape[0], dtype=np.int32)
    blocks = np.zeros((200, 3), dtype=np.uint64)
    info = np.zeros(8, dtype=np.int32)
    capi.check(capi.lib.osfm_ba_debug_order(int(ldim.shape[0]), ldim.ctypes.data_as(C.c_void_p), int(pairs.shape[0]),
                                            pairs.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                            blocks.ctypes.data_as(C.c_void_p), 200, info.ctypes.data_as(C.c_void_p)))
    return off, blocks, dict(zip(("ordered", "arcs", "sep", "span", "nblk", "chain_natural", "chain", "pad"), info.tolist()))


def _ring_pairs(C_, w, closed=True):
    out = []
    for a in range(C_):
        for b in range(a + 1):
            d = a - b
            if (min(d, C_ - d) if closed else d) <= w:
                out.append((a, b))
    return out


def _bit(blocks, i, k):
    i, k = int(i), int(k)
    return (int(blocks[i, k >> 6]) >> (k & 63)) & 1


@pytest.mark.parametrize("cams,w,closed", [(200, 11, True), (500, 11, True), (120, 11, True), (96, 7, False), (64, 5, True)])
def test_rings_and_strips_become_arcs_and_separators(cams, w, closed):
    ldim = np.full(cams, 5, np.int32)
    ldim[0] = 0                                     # the gauge camera has no unknowns
    pairs = _ring_pairs(cams, w, closed)
    off, blocks, info = _order(ldim, pairs)
    assert info["ordered"] == 1 and info["arcs"] >= 2 and info["sep"] == w
    assert 4 * info["chain"] <= 3 * info["chain_natural"]
    # a layout: every camera's unknowns inside the span, no two cameras overlap, padding fills the rest
    used = np.zeros(info["span"], dtype=np.int32)
    for c in range(cams):
        used[off[c]:off[c] + ldim[c]] += 1
    assert used.max() == 1 and int((used == 0).sum()) == info["pad"] and info["nblk"] == (info["span"] + 31) // 32
    # every coupled camera pair lies on a tile the pattern has (fill can only add)
    for a, b in pairs:
        if ldim[a] == 0 or ldim[b] == 0:
            continue
        for x in {off[a] // 32, (off[a] + ldim[a] - 1) // 32}:
            for y in {off[b] // 32, (off[b] + ldim[b] - 1) // 32}:
                assert _bit(blocks, max(x, y), min(x, y)), (a, b)
    # the pattern is closed under elimination (no tile appears that the kernel would not know of)
    n = info["nblk"]
    for k in range(n):
        rows = [i for i in range(k + 1, n) if _bit(blocks, i, k)]
        for x in rows:
            for y in rows:
                if y < x:
                    assert _bit(blocks, x, y)
    # the chain the library reports is the pattern's longest path of dependent diagonal blocks
    depth = [1] * n
    for j in range(n):
        for k in range(j):
            if _bit(blocks, j, k):
                depth[j] = max(depth[j], depth[k] + 1)
    assert max(depth) == info["chain"]


def test_dense_visibility_and_small_systems_keep_their_order():
    ldim = np.full(200, 5, np.int32)
    off, _, info = _order(ldim, _ring_pairs(200, 90))          # tracks that span almost half the ring: no band
    assert info["ordered"] == 0 and np.array_equal(off, np.arange(200) * 5)
    ldim = np.full(30, 5, np.int32)                              # 150 unknowns: five blocks, not worth an order
    assert _order(ldim, _ring_pairs(30, 3))[2]["ordered"] == 0
