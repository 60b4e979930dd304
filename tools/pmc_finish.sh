#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" orthosfm_amd/lib/exp/lib_new8x.so; do
  tag=$( [ -z "$lib" ] && echo base || echo new )
  export OSFM_HIP_LIBRARY=${lib:+$R/$lib}
  [ -z "$lib" ] && unset OSFM_HIP_LIBRARY
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-include-regex "match_finish_kernel" --output-format csv -d $R/gpurun_out/pmcf_${tag}_$i -- python $R/bench.py --views 24 --no-ba --no-verify --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/pmcf_${tag}_$i.log 2>&1 || echo "set $i failed"
    f=$(find $R/gpurun_out/pmcf_${tag}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f match_finish_kernel
  done
done
