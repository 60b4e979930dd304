cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ba
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ba -- python $R/tools/ba_bench.py > $R/gpurun_out/prof_ba.log 2>&1
db=$(find $R/gpurun_out/prof_ba -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db | grep -i "chol" | cut -c1-140
rm -rf $R/gpurun_out/prof_ba
