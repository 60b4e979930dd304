cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_v
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v -- python $R/bench.py --no-ba --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/prof_v.log 2>&1
db=$(find $R/gpurun_out/prof_v -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db | grep -i "ransac\|gather_inl" | cut -c1-130
rm -rf $R/gpurun_out/prof_v
