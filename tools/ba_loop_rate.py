"""LM iterations per second of BASELINE config 4 (200 cameras, 100k tracks), best of a few solves: in the loop / per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
cams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
model = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = synth.make_ba_scene(model, cams, pts, config_id=4)
best = None
for rep in range(6):
    s = ba.solve(ba.FlatProblem.from_scene(sc), verbose=1 if rep == 5 else 0)
    if best is None or s.lm_loop_ms < best.lm_loop_ms:
        best = s
n = best.num_iterations
print(f"{n} iterations, loop {best.lm_loop_ms:.3f} ms = {1e3 * n / best.lm_loop_ms:.0f} it/s ({best.lm_loop_ms / n * 1e3:.0f} us per iteration), call {best.solve_ms:.3f} ms = {1e3 * n / best.solve_ms:.0f} it/s;"
      f" last solve per iteration: cholesky {s.cholesky_ms / n * 1e3:.0f} us, pair {s.pair_pass_ms / s.linearizations * 1e3:.0f}, point {s.point_pass_ms / s.linearizations * 1e3:.0f}, back {s.back_pass_ms / n * 1e3:.0f}")
