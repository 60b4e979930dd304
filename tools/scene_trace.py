"""Where a local adjustment on the device-resident scene spends its time (OSFM_SCENE_TRACE laps of the last groups)."""
import os, sys
os.environ["OSFM_SCENE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orthosfm_amd import pipeline as P, synth
iset = synth.make_image_set(int(sys.argv[1]) if len(sys.argv) > 1 else 40, 20000, config_id=3)
res = P.reconstruct(iset, solver=0, max_groups=12)
print({k: round(v, 4) for k, v in res.timings.__dict__.items()}, file=sys.stderr)
