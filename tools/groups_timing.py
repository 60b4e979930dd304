#!/usr/bin/env python3
"""osfm_build_groups at the cfg5 size: V views, T tracks of 2..12 random views."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orthosfm_amd import groups as G
V = int(sys.argv[1]) if len(sys.argv) > 1 else 500
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
rng = np.random.default_rng(1)
lens = rng.integers(2, 13, Tn)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
# arcs of neighbouring views, as a turntable sees them, plus some random far views
first = rng.integers(0, V, Tn)
views = np.concatenate([np.sort((f + np.arange(l)) % V) for f, l in zip(first, lens)]).astype(np.int32)
ids = np.arange(V, dtype=np.int32)
G.build_groups_flat(ids, offs, views, 3)
t0 = time.perf_counter()
g = G.build_groups_flat(ids, offs, views, 3)
dt = time.perf_counter() - t0
print(json.dumps({"views": V, "tracks": Tn, "groups": len(g), "seconds": dt}))
