#!/usr/bin/env python3
"""BASELINE config 5's largest global adjustment (500 Euler cameras, every angle free: 2495 unknowns) solved three
times: for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/ba_prof500.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
sc = synth.make_ba_scene(synth.MODEL_EULER, 500, 60000, config_id=4)
for rep in range(3):
    s = ba.solve(ba.FlatProblem.from_scene(sc))
    print(rep, s.num_iterations, round(s.lm_loop_ms, 3), round(s.solve_ms, 3), file=sys.stderr)
