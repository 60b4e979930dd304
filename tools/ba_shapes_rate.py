#!/usr/bin/env python3
"""LM loop time per iteration of three adjustment shapes: BASELINE config 4 (200 cameras, 100k tracks of 3..12
views) and the global adjustments of the 200- / 500-view end-to-end jobs (2500 tracks of 60..100 / 150..250 views)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
cases = (("config4", synth.MODEL_QUATERNION, 200, 100000, 3, 12), ("global200", synth.MODEL_EULER, 200, 2500, 60, 100),
         ("global500", synth.MODEL_EULER, 500, 2500, 150, 250))
for name, model, C, M, lo, hi in cases:
    sc = synth.make_ba_scene(model, C, M, config_id=4, min_len=lo, max_len=hi)
    best = None
    for rep in range(3):
        s = ba.solve(ba.FlatProblem.from_scene(sc), ba.default_options(max_num_iterations=10))
        if best is None or s.lm_loop_ms < best.lm_loop_ms: best = s
    print(f"{name}: {best.num_iterations} iterations, loop {best.lm_loop_ms:.3f} ms = {1e3 * best.lm_loop_ms / best.num_iterations:.0f} us per iteration, solve {best.solve_ms:.2f} ms")
