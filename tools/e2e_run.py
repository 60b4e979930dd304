#!/usr/bin/env python3
"""One end-to-end job (orthosfm_amd/pipeline.py) on a synthetic set; prints one JSON line.
    python tools/e2e_run.py --views 200 --features 20000 --solver 0"""
import argparse
import dataclasses
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=200)
    ap.add_argument("--features", type=int, default=20000)
    ap.add_argument("--solver", type=int, default=0)
    ap.add_argument("--matcher", default="exhaustive")
    ap.add_argument("--twin-frac", type=float, default=0.0)
    ap.add_argument("--config-id", type=int, default=3)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--max-groups", type=int, default=0)
    a = ap.parse_args()
    from orthosfm_amd import pipeline as P
    from orthosfm_amd import synth
    t0 = time.perf_counter()
    iset = synth.make_image_set(a.views, a.features, config_id=a.config_id, twin_frac=a.twin_frac)
    gen_s = time.perf_counter() - t0
    res = P.reconstruct(iset, solver=a.solver, matcher=a.matcher, verbose=a.verbose,
                        max_groups=a.max_groups or None)
    model = 0 if a.solver == 0 else 1
    gt, pts = P.canonical_ground_truth(iset, model)
    ang = []
    for v in res.aligned_views:
        Rg, Rc = P._cam_rotation(model, gt[v]), P._cam_rotation(model, res.cam_params[v])
        ang.append(float(np.degrees(np.arccos(np.clip((np.trace(Rg.T @ Rc) - 1) / 2, -1, 1)))))
    calls = res.ba_calls
    out = {"views": a.views, "features": a.features, "solver": a.solver, "matcher": a.matcher,
           "generate_s": gen_s, "timings": dataclasses.asdict(res.timings),
           "pairs": res.num_pairs, "matched_pairs": res.matched_pairs, "correspondences": res.correspondences,
           "mve_tracks": res.num_mve_tracks, "invalid_mve_tracks": res.invalid_mve_tracks,
           "final_tracks": res.tracks.num_tracks, "points": int((res.tracks.alive_t & res.tracks.has_point).sum()),
           "groups": len(res.groups), "cameras": len(res.aligned_views),
           "max_rotation_error_deg": max(ang), "median_rotation_error_deg": float(np.median(ang)),
           "ba_calls": {k: {"n": sum(1 for c in calls if c.kind == k),
                            "iterations": sum(c.iterations for c in calls if c.kind == k),
                            "ms": sum(c.ms for c in calls if c.kind == k),
                            "lm_ms": sum(c.lm_ms for c in calls if c.kind == k),
                            "max_observations": max([c.observations for c in calls if c.kind == k] + [0])}
                        for k in ("local", "global", "final")}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
