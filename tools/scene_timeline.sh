# device timeline of the last steps of a small end-to-end job (local adjustments on the device-resident scene)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_sc
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_sc -- python3 $R/tools/scene_trace.py ${1:-40} > $R/gpurun_out/prof_sc.log 2>&1
db=$(find $R/gpurun_out/prof_sc -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $db ${2:-260} > $R/gpurun_out/scene_timeline.txt
rm -rf $R/gpurun_out/prof_sc
tail -5 $R/gpurun_out/prof_sc.log
