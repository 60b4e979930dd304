"""One rank's side of the two exchange forms at the bench size (50 views, 1225 pairs):
step time with the lists going (a) to a pageable array, (b) to a torch page-locked
array (what the RCCL gather uploads from), (c) into the rank's registered slice of the
shared segment -- after which rank 0 needs nothing but the counts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orthosfm_amd import capi, synth, distributed as D
from orthosfm_amd.matching import HipExhaustiveMatching

dev = torch.device("cuda:0")
V, F = 50, 20000
iset = synth.make_image_set(V, F, config_id=2)
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
m = HipExhaustiveMatching(V, device=0, copy_results=False)
for v in range(V):
    m.set_view(v, iset.sift[v])
cap = F * len(pairs)


def timed(tag):
    m.compute(pairs, capacity=cap)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = m.compute(pairs, capacity=cap)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag:28s} {min(ts):7.2f} ms/step  ({sum(tv.num_matches for tv in out) / 1e6:.2f} M correspondences)")
    return out


timed("pageable result array")
m.use_result_buffer(D.pinned_array("local", cap, dev))
timed("torch page-locked array")
t0 = time.perf_counter()
store = D.SharedMatchStore(cap, 0, 1, dev)
print(f"segment of {store.slice.nbytes / 1e6:.0f} MB created + registered in {1e3 * (time.perf_counter() - t0):.1f} ms")
m.use_result_buffer(store.slice)
out = timed("slice of the shared segment")
counts = np.array([tv.num_matches if tv.status == capi.PAIR_MATCHED else 0 for tv in out], dtype=np.int64)
t0 = time.perf_counter()
c, starts, corr = store.collect(counts, len(pairs))
print(f"collect (counts -> starts) {1e3 * (time.perf_counter() - t0):.3f} ms")
store.close()
