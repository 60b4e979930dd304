#!/bin/bash
# whole-step time (bench headline only) of the product and of experimental builds under orthosfm_amd/lib/exp
cd "$(dirname "$0")/.."
run() { OSFM_HIP_LIBRARY=${1:+$PWD/$1} python bench.py --steps 10 --warmup 2 --no-ba --no-verify --no-e2e --no-cpu-baseline --no-realistic 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${1:-product}', round(r['ms_per_step'],3), 'ms per step, tile launch', round(r['roofline']['avg_launch_ms'],3))"; }
run; for lib in "$@"; do run $lib; done; run
