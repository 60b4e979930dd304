#!/bin/bash
# laps of every osfm_match_all part of the 200-view job's matching half (OSFM_MATCH_TRACE) and its stage times
cd "$(dirname "$0")/.."
OSFM_MATCH_TRACE=1 python tools/e2e_run.py --views 200 > gpurun_out/r05_e2e_trace.json 2> gpurun_out/r05_e2e_trace.err
grep "osfm match" gpurun_out/r05_e2e_trace.err | tail -60
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_e2e_trace.json"))
print({k: round(v, 3) for k, v in d["timings"].items()})
PY
