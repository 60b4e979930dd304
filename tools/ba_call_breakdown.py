#!/usr/bin/env python3
"""Where a BA call spends its time outside the LM loop (verbose = 2 prints the host-side laps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orthosfm_amd import ba, synth
for cams, pts, cfg in ((3, 3000, 1), (200, 100000, 4)):
    sc = synth.make_ba_scene(synth.MODEL_QUATERNION, cams, pts, config_id=cfg)
    ba.solve(ba.FlatProblem.from_scene(sc), max_num_iterations=2)
    for rep in range(2):
        print(f"--- {cams} cameras, {pts} points, call {rep}", file=sys.stderr)
        s = ba.solve(ba.FlatProblem.from_scene(sc), max_num_iterations=50, verbose=2)
        print(f"solve_ms {s.solve_ms:.3f} lm_loop_ms {s.lm_loop_ms:.3f} iterations {s.num_iterations}", file=sys.stderr)
