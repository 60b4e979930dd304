#!/usr/bin/env python3
"""Latency of the 3-camera local BA (the per-group call of the incremental reconstruction,
reconstruct.cpp:219): wall time per call, time inside the LM loop, time per LM iteration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orthosfm_amd import ba, synth

out = []
cases = ((3, 300), (3, 3000), (3, 20000), (6, 3000), (12, 5000))
if len(sys.argv) > 2:
    cases = ((int(sys.argv[1]), int(sys.argv[2])),)
for cams, pts in cases:
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, cams, pts, config_id=1)
    ba.solve(ba.FlatProblem.from_scene(sc), max_num_iterations=2)
    wall, loop, its = [], [], []
    for rep in range(5):
        fp = ba.FlatProblem.from_scene(sc)
        t0 = time.perf_counter()
        s = ba.solve(fp, max_num_iterations=50)
        wall.append((time.perf_counter() - t0) * 1e3)
        loop.append(s.lm_loop_ms); its.append(s.num_iterations)
    k = int(np.argmin(wall))
    out.append({"cameras": cams, "points": pts, "observations": int(fp.obs_camera.size), "iterations": int(its[k]),
                "call_ms": round(wall[k], 3), "solve_ms": round(float(s.solve_ms), 3), "lm_loop_ms": round(loop[k], 3),
                "us_per_iteration": round(1e3 * loop[k] / max(its[k], 1), 1),
                "iterations_per_s": round(its[k] / (loop[k] * 1e-3))})
    print(json.dumps(out[-1]))
