#!/bin/bash
# SQ / LDS counters of the tile kernel (separate rocprofv3 --pmc passes, 8 SQ slots each).
#   gpurun -- 'tools/pmc_tile.sh tag [library]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-base}
[ -n "$2" ] && export OSFM_HIP_LIBRARY=$R/$2
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "match_tile" --output-format csv -d $R/gpurun_out/pmct_${tag}_$i -- python $R/bench.py --no-ba --no-verify --no-cpu-baseline --no-e2e --steps 1 --warmup 0 > $R/gpurun_out/pmct_${tag}_$i.log 2>&1 || echo "set $i failed"
  f=$(find $R/gpurun_out/pmct_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f "${KERNEL:-match_tile_kernel<8, false, true, true>}"
  rm -rf $R/gpurun_out/pmct_${tag}_$i
done
