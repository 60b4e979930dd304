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
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "ba_pair_pass|ba_point_win|ba_back_win|chol_flow" --output-format csv -d $R/gpurun_out/pmcba_$i -- python3 /tmp/ba_once.py > $R/gpurun_out/pmcba_$i.log 2>&1 || echo "set $i failed" >> $out
  f=$(find $R/gpurun_out/pmcba_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    for k in ba_pair_pass_kernel ba_point_win_kernel ba_back_win_kernel chol_flow_kernel; do python3 $R/tools/pmc_summary.py $f "$k" >> $out 2>&1; done
  fi
  rm -rf $R/gpurun_out/pmcba_$i
done
python3 $R/tools/pmc_ba_traffic.py $out $R/gpurun_out/${1:-r02}_ba_traffic_pmc.json
cat $out
