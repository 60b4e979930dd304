#!/bin/bash
# A/B of the 512-row tile workgroups against the 256-row ones, alternating processes on one box:
# tile-kernel ms, whole-step ms and the finish kernel from a kernel trace.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-r02}_ab_wide.txt
: > $out
for rep in 1 2 3; do
  for mode in 0 1; do
    OSFM_NARROW_TILES=$mode python bench.py --steps 5 --warmup 2 --no-ba --no-verify --no-cpu-baseline --no-e2e 2>/dev/null |
      python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('narrow=$mode', 'ms_per_step', round(d['ms_per_step'], 2), 'tile_ms', round(d['roofline']['avg_launch_ms'], 2), 'pairs/s', round(d['value']))" >> $out
  done
done
cat $out
