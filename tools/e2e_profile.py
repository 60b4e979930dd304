#!/usr/bin/env python3
"""cProfile of the pose-estimation half of the end-to-end job (host time: where does it go?)."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orthosfm_amd import pipeline as P, synth, ba as B
V = int(sys.argv[1]) if len(sys.argv) > 1 else 200
iset = synth.make_image_set(V, 20000, config_id=3)
tm = P.Timings()
tt, info = P.match_and_build_tracks(iset, "exhaustive", 0, True, tm)
pr = cProfile.Profile()
pr.enable()
P.run_pose_estimation(tt, iset, B.MODEL_QUATERNION, 0, 2.0, 0.01, 7, tm)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
print({k: round(getattr(tm, k), 3) for k in ("local_ba_s", "local_filter_s", "triangulate_s", "global_ba_s", "pose_host_s", "pose_s")})
