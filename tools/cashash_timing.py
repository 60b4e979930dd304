"""Times the cascade-hashing mode on the bench image set."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orthosfm_amd import capi, synth
from orthosfm_amd.matching import HipCascadeHashing

V, F = int(sys.argv[1]) if len(sys.argv) > 1 else 50, 20000
iset = synth.make_image_set(V, F, seed=1)
m = HipCascadeHashing(V, copy_results=False)
t0 = time.perf_counter()
for v in range(V):
    m.set_view(v, iset.sift[v])
t1 = time.perf_counter()
m.cascade_hashes(0, 0)
t2 = time.perf_counter()
print(f"upload {1e3 * (t1 - t0):.1f} ms, init (average, hashes, buckets) {1e3 * (t2 - t1):.1f} ms")
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
for it in range(3):
    t0 = time.perf_counter()
    out = m.compute(pairs, capacity=len(pairs) * F)
    dt = time.perf_counter() - t0
    st = m.stats()
    print(f"compute {1e3 * dt:.1f} ms -> {len(pairs) / dt:.0f} pairs/s; cascade kernel {st.cashash_kernel_ms:.1f} ms; "
          f"matched {sum(tv.status == 0 for tv in out)}, correspondences {sum(tv.num_matches for tv in out if tv.status == 0)}")
