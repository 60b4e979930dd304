"""Does the order of the tracks matter to the passes?  BASELINE config 4 as generated (tracks in random order around the
ring) and with the tracks renumbered by the first camera that sees them (the caller's arrays permuted on the host: a
probe, not a feature)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orthosfm_amd import ba, synth
sc = synth.make_ba_scene(0, 200, 100000, config_id=4)


def permuted(sc, key):
    M = sc.points.shape[0]
    start = np.searchsorted(sc.obs_point, np.arange(M))
    end = np.searchsorted(sc.obs_point, np.arange(M), side="right")
    order = np.argsort(key, kind="stable")
    idx = np.concatenate([np.arange(start[j], end[j]) for j in order])
    out = sc.copy()
    out.points = sc.points[order].copy()
    out.obs_xy = sc.obs_xy[idx].copy(); out.obs_camera = sc.obs_camera[idx].copy()
    out.obs_point = np.repeat(np.arange(M), (end - start)[order]).astype(np.int32)
    return out


M = sc.points.shape[0]
first = np.full(M, 1 << 30)
np.minimum.at(first, sc.obs_point, sc.obs_camera)
# the arc's first camera on the ring (an arc that wraps has camera 0 in it: its start is the camera behind the gap)
cams = [sc.obs_camera[sc.obs_point == j] for j in range(0)]
for name, scene in (("as generated", sc), ("tracks by smallest camera", permuted(sc, first))):
    best = None
    for rep in range(5):
        s = ba.solve(ba.FlatProblem.from_scene(scene), verbose=1)
        if best is None or s.lm_loop_ms < best.lm_loop_ms:
            best = s
    n = best.num_iterations
    print(f"{name:28s} {n} iterations, cost {best.final_cost:.6f}, loop {1e3 * n / best.lm_loop_ms:.0f} it/s; per iteration: cholesky {best.cholesky_ms / n * 1e3:.0f} us, "
          f"pair {best.pair_pass_ms / best.linearizations * 1e3:.0f}, point {best.point_pass_ms / best.linearizations * 1e3:.0f}, back {best.back_pass_ms / n * 1e3:.0f}")
