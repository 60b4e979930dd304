"""Reads the in-kernel segment timers of an OSFM_EXP=32 build (kernel experiments)."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orthosfm_amd import capi, synth
from orthosfm_amd.matching import HipExhaustiveMatching

V, F = 12, 20000
iset = synth.make_image_set(V, F, seed=1)
o = capi.default_match_options()
o.use_lowres_matching = 0
m = HipExhaustiveMatching(V, device=0, options=o)
for v in range(V):
    m.set_view(v, iset.sift[v])
pairs = [(a, b) for a in range(V) for b in range(a)]
m.compute(pairs, capacity=len(pairs) * F)
out = (C.c_ulonglong * 8)()
capi.lib.osfm_debug_read_prof(out)
n = out[4] or 1
names = ["compute (top .. before vmcnt wait)", "vmcnt(0) wait (DMA)", "barrier", "merge + loop end"]
tot = sum(out[i] for i in range(4))
for i in range(4):
    print(f"{names[i]:40s} {out[i] / n:9.1f} ticks/tile  {100.0 * out[i] / tot:5.1f} %")
print("tiles", n, "total ticks/tile", tot / n)
