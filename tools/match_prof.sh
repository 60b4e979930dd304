# per-kernel durations of the matching half of the end-to-end job: gpurun_out/prof_match_<views>.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${1:-200}
rm -rf $R/gpurun_out/prof_m
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m -- python3 $R/tools/e2e_match_stats.py $n ${2:-2} > $R/gpurun_out/prof_m_$n.log 2>&1
db=$(find $R/gpurun_out/prof_m -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db --out $R/gpurun_out/prof_match_$n.csv | cut -c1-160 | head -8
python3 $R/tools/rocpd_overlap.py $db ransac_kernel match_tile_kernel
rm -rf $R/gpurun_out/prof_m
tail -3 $R/gpurun_out/prof_m_$n.log
