#!/bin/bash
# One 3-camera local adjustment (the per-call entry, osfm_ba_solve) as the host sees it (verbose laps) and as the
# device sees it (kernel + copy timeline of the last call).   gpurun -- 'tools/local_ba_timeline.sh [tag]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r05}
python3 - > $R/gpurun_out/${tag}_local_ba_laps.txt 2>&1 <<PY
import sys, time
sys.path.insert(0, "$R")
from orthosfm_amd import ba, synth
sc = synth.make_ba_scene(0, 3, 3000, config_id=1)
ba.solve(ba.FlatProblem.from_scene(sc), max_num_iterations=2)
for v in (0, 2):
    for rep in range(3):
        fp = ba.FlatProblem.from_scene(sc)
        t0 = time.perf_counter()
        s = ba.solve(fp, max_num_iterations=50, verbose=v)
        print("verbose", v, "call_ms", round((time.perf_counter() - t0) * 1e3, 3), "loop_ms", round(s.lm_loop_ms, 3), "iterations", s.num_iterations, flush=True)
PY
rm -rf $R/gpurun_out/prof_lba
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_lba -- python3 $R/tools/small_ba_timing.py 3 3000 > $R/gpurun_out/prof_lba.log 2>&1
db=$(find $R/gpurun_out/prof_lba -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $db ${2:-70} > $R/gpurun_out/${tag}_local_ba_timeline.txt
rm -rf $R/gpurun_out/prof_lba
cat $R/gpurun_out/${tag}_local_ba_laps.txt | tail -40
