#!/usr/bin/env python3
"""BASELINE config 4 (200 quaternion cameras, 100k tracks) solved three times: for a kernel trace
(rocprofv3 --kernel-trace --stats -- python3 tools/ba_config4_once.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
sc = synth.make_ba_scene(synth.MODEL_QUATERNION, 200, 100000, config_id=4)
for rep in range(3):
    s = ba.solve(ba.FlatProblem.from_scene(sc))
    print(rep, s.num_iterations, round(s.lm_loop_ms, 3), round(s.solve_ms, 3), file=sys.stderr)
