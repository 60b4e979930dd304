"""Splits one matching step into library time and Python wrapper time."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orthosfm_amd import capi, synth
from orthosfm_amd.matching import HipExhaustiveMatching

V, F = 50, 20000
iset = synth.make_image_set(V, F, seed=1)
m = HipExhaustiveMatching(V, device=0)
for v in range(V):
    m.set_view(v, iset.sift[v])
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
n = len(pairs)
arr = (capi.Pair * n)()
for k, (a, b) in enumerate(pairs):
    arr[k].view_1, arr[k].view_2 = a, b
res = (capi.PairResult * n)()
cap = n * F
corr = np.zeros((cap, 2), dtype=np.int32)
total = C.c_int64()
for it in range(3):
    t0 = time.perf_counter()
    capi.check(capi.lib.osfm_match_all(m._h, arr, n, res, capi._ptr(corr, C.c_int32), C.c_int64(cap), C.byref(total)))
    t1 = time.perf_counter()
    st = m.stats()
    print(f"osfm_match_all {1e3 * (t1 - t0):.2f} ms  (tile kernel {st.tile_kernel_ms:.2f} ms, total corr {total.value})")
for it in range(2):
    t0 = time.perf_counter()
    m._copy_results = False
    out = m.compute(pairs, capacity=cap)
    t1 = time.perf_counter()
    print(f"compute() {1e3 * (t1 - t0):.2f} ms")
os.environ["OSFM_VERBOSE"] = "1"
