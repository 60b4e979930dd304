#!/bin/bash
# Counters of one kernel (regex) in a bench run without the BA / end-to-end / CPU legs (separate rocprofv3 --pmc passes).
#   gpurun -- 'bash tools/pmc_kernel.sh ransac_kernel r04'  ->  gpurun_out/<round>_pmc_<regex>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
k=${1:-ransac_kernel}
out=$R/gpurun_out/${2:-r04}_pmc_$k.txt
: > $out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "$k" --output-format csv -d $R/gpurun_out/pmck_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-ba --no-e2e --no-cpu-baseline --no-realistic > $R/gpurun_out/pmck_$i.log 2>&1 || echo "set $i failed" >> $out
  f=$(find $R/gpurun_out/pmck_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py $f "$k" >> $out 2>&1; fi
  rm -rf $R/gpurun_out/pmck_$i
done
cat $out
