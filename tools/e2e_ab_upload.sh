#!/bin/bash
# A/B of the host threads of the view upload on one box: e2e_run.py --views 200, alternating
cd "$(dirname "$0")/.."
for t in 4 1 4 1; do
  OSFM_UPLOAD_THREADS=$t python tools/e2e_run.py --views 200 2>/dev/null > /tmp/e2e_ab.json
  python - "$t" <<'PY'
import json, sys
d = json.load(open("/tmp/e2e_ab.json")); t = d["timings"]
print("threads", sys.argv[1], round(t["upload_s"], 3), round(t["matching_s"], 3), round(t["pose_s"], 3), round(t["total_s"], 3))
PY
done
