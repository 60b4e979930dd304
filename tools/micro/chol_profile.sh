#!/bin/bash
# per-kernel durations of the stand-alone Cholesky check
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_chol
timeout -k 10 120 tools/micro/chol_test.bin > gpurun_out/prof_chol_plain.log 2>&1 || { echo 'chol_test failed or hung'; tail -3 gpurun_out/prof_chol_plain.log; exit 1; }
timeout -k 10 180 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_chol -o chol -- tools/micro/chol_test.bin > gpurun_out/prof_chol.log 2>&1
python3 tools/rocpd_stats.py $(find gpurun_out/prof_chol -name "*.db" | head -1) | cut -c1-160
