#!/bin/bash
# HBM traffic of the tile kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (MI355X_MICROARCH.md, HBM: both do not fit one pass), then tools/pmc_traffic.py.  Two warm-up passes:
# the counters of the LAST (third, warmed) full-matching launch are the ones kept.
#   gpurun -- 'tools/pmc_traffic.sh r02'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
round=${1:-r02}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-include-regex "match_tile_kernel" --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 1 --warmup 2 --no-ba --no-verify --no-cpu-baseline --no-e2e --no-realistic > $R/gpurun_out/pmc_$c.log 2>&1 || echo "$c pass failed"
  f=$(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  cp $f $R/gpurun_out/${round}_match_pmc_$(echo $c | tr A-Z a-z).csv
  rm -rf $R/gpurun_out/pmc_$c
done
python $R/tools/pmc_traffic.py $R/gpurun_out/${round}_match_pmc_fetch_size.csv $R/gpurun_out/${round}_match_pmc_write_size.csv auto $R/gpurun_out/${round}_match_traffic_pmc.json
