import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib
from orthosfm_amd import ba, synth
for model in (0, 1):
    sc = synth.make_ba_scene(model, 40, 4000, config_id=35)
    sc.obs_xy[::41] += 25.0
    ref = sc.copy()
    so = oracle_lib.oracle_ba_solve(ref)
    for rep in range(3):
        fp = ba.FlatProblem.from_scene(sc)
        s = ba.solve(fp)
        print(model, s.num_iterations, so.num_iterations, repr(s.final_cost), repr(so.final_cost), abs(s.final_cost - so.final_cost) / so.final_cost,
              np.abs(fp.cam_params - ref.cam_params).max(), s.order_arcs, s.chain_blocks_natural, s.chain_blocks, s.flow_fallbacks)
