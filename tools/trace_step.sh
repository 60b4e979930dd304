cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace1 -- python $R/bench.py --no-ba --no-verify --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/trace1.log 2>&1
find $R/gpurun_out/trace1 -name "*.csv" | head
