#!/bin/bash
# times the tile kernel of every experimental build under orthosfm_amd/lib/exp
for lib in "" orthosfm_amd/lib/exp/*.so; do
    OSFM_HIP_LIBRARY=${lib:+$PWD/$lib} python bench.py --views 24 --no-ba --no-verify --no-cpu-baseline --no-lowres-gate --steps 2 --warmup 1 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${lib:-default}', round(d['roofline']['avg_launch_ms'],3), 'ms', round(d['roofline']['achieved'],1), 'TOPS', round(d['ms_per_step'],2), 'ms/step', d['correspondences_rank0'])"
done
