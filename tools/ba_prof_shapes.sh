# per-kernel durations of the global-adjustment shapes of the end-to-end jobs: gpurun_out/prof_ba_shape_<n>.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 200 500; do
  rm -rf $R/gpurun_out/prof_bas
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bas -- python3 $R/tools/ba_global_shapes.py $n > $R/gpurun_out/prof_bas_$n.log 2>&1
  db=$(find $R/gpurun_out/prof_bas -name "*.db" | head -1)
  echo "== $n cameras"
  python3 $R/tools/rocpd_stats.py $db --out $R/gpurun_out/prof_ba_shape_$n.csv | grep "osfm::" | cut -c1-150
  rm -rf $R/gpurun_out/prof_bas
done
