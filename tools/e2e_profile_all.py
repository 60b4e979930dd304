#!/usr/bin/env python3
"""cProfile of the whole end-to-end job (host time by function), 200 views by default."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orthosfm_amd import pipeline as P, synth
V = int(sys.argv[1]) if len(sys.argv) > 1 else 200
iset = synth.make_image_set(V, 20000, config_id=3)
P.reconstruct(synth.make_image_set(4, 2000, config_id=5), solver=0)      # code objects, pools
pr = cProfile.Profile()
pr.enable()
res = P.reconstruct(iset, solver=0)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue())
print({k: round(v, 3) for k, v in res.timings.__dict__.items()})
