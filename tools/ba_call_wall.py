"""Wall time of a whole osfm_ba_solve call on BASELINE config 4 as the caller sees it, next to what the library reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
sc = synth.make_ba_scene(0, 200, 100000, config_id=4)
ba.solve(ba.FlatProblem.from_scene(sc))
for rep in range(5):
    fp = ba.FlatProblem.from_scene(sc)
    st = fp.struct()
    t0 = time.perf_counter()
    s = ba.solve(fp)
    w = (time.perf_counter() - t0) * 1e3
    print(f"wall {w:.3f} ms, solve_ms {s.solve_ms:.3f}, loop {s.lm_loop_ms:.3f}, iterations {s.num_iterations}: per call {1e3 * s.num_iterations / w:.0f} it/s, in loop {1e3 * s.num_iterations / s.lm_loop_ms:.0f}")
