#!/usr/bin/env python3
"""Tile-kernel time of one matching step on the bench set, split into its passes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orthosfm_amd import capi, synth
from orthosfm_amd.matching import HipExhaustiveMatching
V = int(sys.argv[1]) if len(sys.argv) > 1 else 50
F = 20000
iset = synth.make_image_set(V, F, config_id=2)
o = capi.default_match_options()
m = HipExhaustiveMatching(V, options=o, copy_results=False)
for v in range(V):
    m.set_view(v, iset.sift[v])
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
for _ in range(3):
    out = m.compute(pairs, capacity=F * len(pairs))
    st = m.stats()
    macs = st.mac_count
    print(os.environ.get("OSFM_HIP_LIBRARY","default")[-24:], json.dumps({"tile_ms": st.tile_kernel_ms, 
                      "TOPS": 2 * macs / (st.tile_kernel_ms * 1e-3) / 1e12,
                      "corr": int(sum(tv.num_matches for tv in out if tv.status == 0))}))
