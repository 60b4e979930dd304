#!/usr/bin/env python3
"""The matching half of the end-to-end job: wall time against the time its kernels were timed at.
    python tools/e2e_match_stats.py 500"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orthosfm_amd import pipeline as P, synth
V = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DEV = [0] * int(os.environ["SHARDS"]) if os.environ.get("SHARDS") else 0
iset = synth.make_image_set(V, 20000, config_id=3)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    tm = P.Timings()
    t0 = time.perf_counter()
    tt, info = P.match_and_build_tracks(iset, "exhaustive", DEV, True, tm)
    wall = time.perf_counter() - t0
    P.join_background()
    st = info.get("match_stats", {})
    print(json.dumps({"views": V, "rep": rep, "wall_s": round(wall, 3), "upload_s": round(tm.upload_s, 3), "setup_s": round(tm.setup_s, 3),
                      "matching_s": round(tm.matching_s, 3), "tracks_s": round(tm.tracks_s, 3), "tracks_busy_s": round(tm.tracks_busy_s, 3),
                      "pairs": info["num_pairs"], "stats": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}}))
