import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import ba, synth
if len(sys.argv) > 1 and sys.argv[1] == "timing":
    sc = synth.make_ba_scene(0, 200, 100000, config_id=4)
    fp = ba.FlatProblem.from_scene(sc)
    ba.solve(ba.FlatProblem.from_scene(sc), max_num_iterations=2)
    s = ba.solve(fp, max_num_iterations=25, verbose=2)
    print(s.num_iterations, s.solve_ms)
else:
    print(json.dumps(ba.bench_global_ba()))
