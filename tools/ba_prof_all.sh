# Round evidence of the bundle adjustment: per-kernel durations of config 4 (three solves), of 500 Euler cameras /
# 60k tracks (config 5's largest adjustment) and of the end-to-end jobs' global-adjustment shapes; the loop rates.
#   gpurun -- 'bash tools/ba_prof_all.sh r04'  ->  gpurun_out/<round>_ba_*.csv / .txt (copy into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
round=${1:-r04}
prof() {   # name, script, args
  rm -rf $R/gpurun_out/prof_tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tmp -- python3 $R/tools/$2 $3 > $R/gpurun_out/prof_tmp.log 2>&1
  db=$(find $R/gpurun_out/prof_tmp -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $db --out $R/gpurun_out/${round}_ba_$1_kernel_stats.csv | grep "osfm::" | cut -c1-150 | head -8
  rm -rf $R/gpurun_out/prof_tmp
}
echo "== config 4"; prof config4 ba_config4_once.py
echo "== 500 cameras, 60k tracks"; prof 500cameras ba_prof500.py
echo "== global adjustment shape, 200 cameras"; prof shape200 ba_global_shapes.py 200
echo "== global adjustment shape, 500 cameras"; prof shape500 ba_global_shapes.py 500
python3 $R/tools/ba_shapes_rate.py > $R/gpurun_out/${round}_ba_rates.txt 2>&1
python3 $R/tools/ba_loop_rate.py >> $R/gpurun_out/${round}_ba_rates.txt 2>&1
python3 $R/tools/small_ba_timing.py >> $R/gpurun_out/${round}_ba_rates.txt 2>&1
cat $R/gpurun_out/${round}_ba_rates.txt
python3 $R/tools/chol_flow_trace.py > $R/gpurun_out/${round}_chol_flow_trace.txt 2>&1
head -8 $R/gpurun_out/${round}_chol_flow_trace.txt
