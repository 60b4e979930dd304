cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_cas
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cas -- python $R/tools/cashash_timing.py ${CAS_VIEWS:-50} > $R/gpurun_out/prof_cas.log 2>&1
db=$(find $R/gpurun_out/prof_cas -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db | head -14 | cut -c1-150
rm -rf $R/gpurun_out/prof_cas
