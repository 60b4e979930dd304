#!/bin/bash
# Kernel trace of a 3-camera local BA: per-kernel durations of the launch chain.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_small
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_small -o small -- python3 tools/small_ba_timing.py 3 ${1:-3000} > gpurun_out/prof_small.log 2>&1
f=$(find gpurun_out/prof_small -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${2:-r02}_small_ba_kernel_stats.csv
head -30 gpurun_out/${2:-r02}_small_ba_kernel_stats.csv | cut -c1-150
tail -3 gpurun_out/prof_small.log
