#!/bin/bash
# Counters of the BA kernels on config 4 (separate rocprofv3 --pmc passes).
#   gpurun -- 'tools/pmc_ba.sh'   ->  gpurun_out/<round>_pmc_ba.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-r02}_pmc_ba.txt
: > $out
cat > /tmp/ba_once.py <<PY
import sys
sys.path.insert(0, "$R")
from orthosfm_amd import ba
print(ba.bench_global_ba()["lm_loop_iterations_per_s"])
PY
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "ba_pair_pass|ba_point_pass|ba_back_pass" --output-format csv -d $R/gpurun_out/pmcba_$i -- python3 /tmp/ba_once.py > $R/gpurun_out/pmcba_$i.log 2>&1 || echo "set $i failed" >> $out
  f=$(find $R/gpurun_out/pmcba_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    for k in ba_pair_pass_kernel ba_point_pass_kernel ba_back_pass_kernel; do python3 $R/tools/pmc_summary.py $f "$k" >> $out 2>&1; done
  fi
  rm -rf $R/gpurun_out/pmcba_$i
done
cat $out
