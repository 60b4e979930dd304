# per-kernel durations of BASELINE config 4's bundle adjustment (three solves): gpurun_out/prof_ba_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ba
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ba -- python3 $R/tools/ba_config4_once.py > $R/gpurun_out/prof_ba.log 2>&1
db=$(find $R/gpurun_out/prof_ba -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db --out $R/gpurun_out/prof_ba_stats.csv | grep "osfm::" | cut -c1-140
rm -rf $R/gpurun_out/prof_ba
