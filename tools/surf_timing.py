"""Both descriptor types at scale (FEATURE_ALL, matching_mve.cpp:333): SIFT + SURF per view."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orthosfm_amd import capi, synth
from orthosfm_amd.matching import HipExhaustiveMatching

V, F, U = 24, 20000, 8000
iset = synth.make_image_set(V, F, n_surf=U, seed=1)
m = HipExhaustiveMatching(V, copy_results=False)
for v in range(V):
    m.set_view(v, iset.sift[v], iset.surf[v])
pairs = [capi.pair_from_index(i) for i in range(V * (V - 1) // 2)]
for it in range(3):
    t0 = time.perf_counter()
    out = m.compute(pairs, capacity=len(pairs) * (F + U))
    dt = time.perf_counter() - t0
    st = m.stats()
    print(f"{len(pairs)} pairs SIFT {F} + SURF {U}: {1e3 * dt:.1f} ms -> {len(pairs) / dt:.0f} pairs/s; tile kernels {st.tile_kernel_ms:.1f} ms "
          f"({st.tile_kernel_launches} launches), {2e-12 * st.mac_count / (st.tile_kernel_ms * 1e-3):.0f} TOP/s, "
          f"correspondences {sum(tv.num_matches for tv in out if tv.status == 0)}")
