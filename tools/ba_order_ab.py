"""BASELINE config 4 with and without the elimination order of the reduced camera system (OSFM_BA_ORDER=0 / 1,
one process per arm: the policy is read once): iterations, final cost, cameras, rates."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "arm":
    sys.path.insert(0, ROOT)
    import numpy as np
    from orthosfm_amd import ba, synth
    cams, pts, model = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sc = synth.make_ba_scene(model, cams, pts, config_id=4)
    best, fp, calls = None, None, []
    for rep in range(6):
        fp = ba.FlatProblem.from_scene(sc)
        s = ba.solve(fp, verbose=1)
        calls.append(s.solve_ms)
        if best is None or s.lm_loop_ms < best.lm_loop_ms:
            best = s
    n = best.num_iterations
    np.save(sys.argv[5], np.concatenate([fp.cam_params.ravel(), fp.points.ravel()[:3000]]))
    print(json.dumps({"order": os.environ.get("OSFM_BA_ORDER"), "arcs": best.order_arcs, "chain": [best.chain_blocks_natural, best.chain_blocks],
                      "iterations": n, "final_cost": best.final_cost, "loop_it_s": round(1e3 * n / best.lm_loop_ms), "call_it_s": round(1e3 * n / min(calls)), "call_ms": [round(c, 2) for c in calls],
                      "us": {"cholesky": round(best.cholesky_ms / n * 1e3), "pair": round(best.pair_pass_ms / best.linearizations * 1e3),
                             "point": round(best.point_pass_ms / best.linearizations * 1e3), "back": round(best.back_pass_ms / n * 1e3)},
                      "fallbacks": best.flow_fallbacks}))
    sys.exit(0)
import numpy as np
cases = [(200, 100000, 0), (200, 100000, 1), (120, 20000, 0), (500, 60000, 0)]
for cams, pts, model in cases:
    outs = []
    for arm in ("0", "1"):
        env = dict(os.environ, OSFM_BA_ORDER=arm)
        path = f"/tmp/ba_order_{arm}.npy"
        r = subprocess.run([sys.executable, __file__, "arm", str(cams), str(pts), str(model), path], env=env, capture_output=True, text=True, timeout=600)
        print(cams, pts, model, (r.stdout.strip().splitlines() or [r.stderr[-800:]])[-1], flush=True)
        outs.append(np.load(path) if os.path.exists(path) else None)
    if outs[0] is not None and outs[1] is not None:
        print("   max |difference| of cameras / points between the arms:", float(np.max(np.abs(outs[0] - outs[1]))), flush=True)
