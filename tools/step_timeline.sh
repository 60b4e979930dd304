#!/bin/bash
# Timeline of one matching step of the bench (kernels and copies in order, with gaps).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_step
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_step -o step -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-ba --no-verify > gpurun_out/prof_step.log 2>&1
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/prof_step/*.db")[0])
cur = db.cursor()
rows = [(s, e, n[:70]) for n, s, e in cur.execute("select name, start, end from kernels")]
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "memory_copies" in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
    size_col = "size" if "size" in cols else cols[-1]
    rows += [(s, e, f"COPY {n} {sz} B") for n, s, e, sz in cur.execute(f"select name, start, end, {size_col} from memory_copies")]
rows.sort()
# the last step: from the last big tile kernel backwards to the previous one
tiles = [i for i, r in enumerate(rows) if "match_tile_kernel<8, false, true, true>" in r[2] and r[1] - r[0] > 2e7]
lo = tiles[-2] + 1 if len(tiles) > 1 else 0
prev = rows[lo - 1][1] if lo > 0 else rows[lo][0]
print("timeline from the end of one tile kernel to the end of the next step's last event (us):")
for s, e, n in rows[lo - 1:]:
    print(f"  +{(s - prev) / 1e3:9.1f} gap  {(e - s) / 1e3:9.1f} dur  {n}")
    prev = e
PY
