import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from orthosfm_amd import capi, synth
from orthosfm_amd.matching import HipExhaustiveMatching, HipCascadeHashing
V, F = 120, 20000
iset = synth.make_image_set(V, F, config_id=2)
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
def probe(tag):
    m = HipCascadeHashing(V, device=0, copy_results=False)
    t0 = time.perf_counter()
    for v in range(V):
        m.set_view(v, iset.sift[v])
    t1 = time.perf_counter()
    m.cascade_hashes(0, 0)
    t2 = time.perf_counter()
    print(f"{tag}: set_view {1e3*(t1-t0):.1f} ms, cascade_hashes {1e3*(t2-t1):.1f} ms")
    m.close()
probe("fresh process")
probe("second matcher")
e = HipExhaustiveMatching(V, device=0, copy_results=False)
for v in range(V):
    e.set_view(v, iset.sift[v])
e.compute(pairs[:2000], capacity=F * 2000)
probe("with an exhaustive matcher alive")
e.close()
probe("after closing it")
