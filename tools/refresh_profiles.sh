#!/bin/bash
# Round evidence: bench line + rocprofv3 kernel summary of the same command.
#   gpurun -- 'tools/refresh_profiles.sh r02 v1'   ->  gpurun_out/<round>_bench_<tag>.json,
#   gpurun_out/<round>_bench_kernel_stats_<tag>.csv  (copy both into profiles/)
set -e
round=${1:-r02}
tag=${2:-vX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/bench_$tag.log 2>&1
grep "^{" $R/gpurun_out/bench_$tag.log | tail -1 > $R/gpurun_out/${round}_bench_$tag.json
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -- python $R/bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
# (one database per traced process: bench.py's drop-in leg runs a host program of its own -- all of them are summed)
db=$(find $R/gpurun_out/prof_$tag -name "*.db" | tr '\n' ' ')
if [ -n "$db" ]; then
  python $R/tools/rocpd_stats.py $db > $R/gpurun_out/${round}_bench_kernel_stats_$tag.csv
else
  f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/${round}_bench_kernel_stats_$tag.csv
fi
rm -rf $R/gpurun_out/prof_$tag        # the trace itself is large; the summary is what is kept
# Second summary: a run with ONLY the headline launches (no BA, no RANSAC passes, no end-to-end job, no CPU
# baseline, no drop-in / realistic-operand legs), so that the AverageNs of match_tile_kernel<8, false, true, true>
# IS roofline.avg_launch_ms of the line printed by the same command.
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_h_$tag -- python $R/bench.py --steps 10 --warmup 2 --no-ba --no-verify --no-e2e --no-cpu-baseline --no-realistic > $R/gpurun_out/prof_h_$tag.log 2>&1
grep "^{" $R/gpurun_out/prof_h_$tag.log | tail -1 > $R/gpurun_out/${round}_bench_headline_only_$tag.json
db=$(find $R/gpurun_out/prof_h_$tag -name "*.db" | tr '\n' ' ')
if [ -n "$db" ]; then
  python $R/tools/rocpd_stats.py $db > $R/gpurun_out/${round}_bench_headline_only_kernel_stats_$tag.csv
else
  f=$(find $R/gpurun_out/prof_h_$tag -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/${round}_bench_headline_only_kernel_stats_$tag.csv
fi
rm -rf $R/gpurun_out/prof_h_$tag
head -6 $R/gpurun_out/${round}_bench_headline_only_kernel_stats_$tag.csv
head -8 $R/gpurun_out/${round}_bench_kernel_stats_$tag.csv
python - <<PY
import json
d = json.load(open("$R/gpurun_out/${round}_bench_$tag.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
