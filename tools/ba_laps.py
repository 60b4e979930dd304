#!/usr/bin/env python3
"""Where a bundle adjustment call spends its wall time outside the LM loop (osfm_ba_options.verbose = 2
prints a lap per stage to stderr): BASELINE config 4 (200 cameras, 100k tracks of 3..12 views), a
global-adjustment shape of the 200-view job (200 cameras, 2500 tracks of 60..100 views) and one of the 500-view job (500 cameras, 2500
tracks of 150..250 views) and a local one (3 cameras)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
cases = (("config4", synth.MODEL_QUATERNION, 200, 100000, 3, 12), ("global", synth.MODEL_EULER, 200, 2500, 60, 100),
         ("global500", synth.MODEL_EULER, 500, 2500, 150, 250), ("local", synth.MODEL_EULER, 3, 2500, 3, 3))
for name, model, C, M, lo, hi in cases:
    sc = synth.make_ba_scene(model, C, M, config_id=4, min_len=lo, max_len=hi)
    for rep in range(3):
        fp = ba.FlatProblem.from_scene(sc)
        t0 = time.perf_counter()
        s = ba.solve(fp, ba.default_options(verbose=2 if rep == 2 else 0, max_num_iterations=10))
        dt = (time.perf_counter() - t0) * 1e3
        print(f"{name} rep {rep}: call {dt:.2f} ms, solve_ms {s.solve_ms:.2f}, lm_loop_ms {s.lm_loop_ms:.2f}, it {s.num_iterations}, obs {fp.obs_camera.shape[0]}", file=sys.stderr)
