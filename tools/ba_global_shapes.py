#!/usr/bin/env python3
"""The global adjustments of the end-to-end jobs have few, long tracks (2500 tracks seen by 60..250 of 200 / 500
cameras): one solve each, for a kernel trace.
    rocprofv3 --kernel-trace --stats -- python3 tools/ba_global_shapes.py [200|500]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
which = sys.argv[1] if len(sys.argv) > 1 else "200"
C, lo, hi = (200, 60, 100) if which == "200" else (500, 150, 250)
sc = synth.make_ba_scene(synth.MODEL_EULER, C, 2500, config_id=4, min_len=lo, max_len=hi)
for rep in range(3):
    s = ba.solve(ba.FlatProblem.from_scene(sc), ba.default_options(max_num_iterations=10))
    print(rep, s.num_iterations, round(s.lm_loop_ms, 3), round(s.solve_ms, 3), file=sys.stderr)
