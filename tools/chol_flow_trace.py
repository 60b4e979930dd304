"""Timeline of the one-launch Cholesky (chol_flow_kernel) on BASELINE config 4: per diagonal workgroup,
microseconds since the kernel started (100 MHz device counter)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orthosfm_amd import ba, capi, synth

cams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
sc = synth.make_ba_scene(synth.MODEL_QUATERNION, cams, pts, config_id=4)
fp = ba.FlatProblem.from_scene(sc)
buf = np.zeros((161, 32), np.int64)
capi.check(capi.lib.osfm_ba_debug_chol_trace(1, buf.ctypes.data_as(C.POINTER(C.c_int64))))
s = ba.solve(fp)
capi.check(capi.lib.osfm_ba_debug_chol_trace(0, buf.ctypes.data_as(C.POINTER(C.c_int64))))
t0 = buf[0, 0]
names = ["start", "piv cyc", "fac+st cyc", "L out", "factor", "inv out", "fac cyc", "how"]
print(f"{s.num_iterations} LM iterations; last factorisation, us since D_0 started")
print("row " + " ".join(f"{n:>9s}" for n in names) + "   period")
prev = None
for r in range(161):
    if buf[r, 0] == 0:
        continue
    v = [(buf[r, k] - t0) / 100.0 if buf[r, k] else float("nan") for k in range(6)]
    v[1], v[2] = float(buf[r, 1]), float(buf[r, 2])        # cycles: pivot loop, factor incl. its stores issued
    per = (v[5] - prev) if prev is not None else float("nan")
    prev = v[5]
    extra = [(buf[r, k] - t0) / 100.0 if buf[r, k] else float("nan") for k in (8, 9)]
    print(f"{r:3d} " + " ".join(f"{x:9.2f}" for x in v) + f" {int(buf[r, 6]):9d} {int(buf[r, 7]):9d}   {per:6.2f}   sc1 issued {extra[0]:8.2f} updated {extra[1]:8.2f}")
print("per panel, shader cycles since the wave entered the factor: chain (panel published) | inverse (rows stored)")
for r in range(161):
    if buf[r, 0] == 0 or buf[r, 16] == 0:
        continue
    print(f"{r:3d} " + " ".join(f"{int(x):6d}" for x in buf[r, 16:24]) + " | " + " ".join(f"{int(x):6d}" for x in buf[r, 24:32]))
    if r >= 6:
        break
