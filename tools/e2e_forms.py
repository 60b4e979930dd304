"""The 200-view job in both forms (track table on the device / per-call): are flags and points the same?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import pipeline as P, synth
V = int(sys.argv[1]) if len(sys.argv) > 1 else 200
iset = synth.make_image_set(V, 20000, config_id=3)
a = P.reconstruct(iset, solver=0)
b = P.reconstruct(iset, solver=0, use_scene=False)
ta, tb = a.tracks, b.tracks
print("total_s", a.timings.total_s, b.timings.total_s, "pose_s", a.timings.pose_s, b.timings.pose_s)
print("cams", np.array_equal(a.cam_params, b.cam_params))
for name in ("alive_t", "alive_f", "live_f", "has_point"):
    x, y = getattr(ta, name), getattr(tb, name)
    print(name, np.array_equal(x, y), int((x != y).sum()))
sa, sb = ta.alive_t & ta.has_point, tb.alive_t & tb.has_point
print("alive&has_point", np.array_equal(sa, sb), int(sa.sum()), "points", np.array_equal(ta.point[sa], tb.point[sb]) if sa.sum() == sb.sum() else "shape")
print("lengths", np.array_equal(ta.alive_lengths(), tb.alive_lengths()))
