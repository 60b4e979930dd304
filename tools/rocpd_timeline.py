"""The last N kernel / memory-copy records of a rocprofv3 rocpd database as a timeline (start since the first shown,
duration, gap to the record before): what a short host-driven sequence looks like on the device.
usage: python tools/rocpd_timeline.py results.db [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(db.execute("select name, start, end from kernels order by start"))
try:
    rows += [("[copy] " + str(r[0]), r[1], r[2]) for r in db.execute("select name, start, end from memory_copies")]
except Exception:
    pass
rows.sort(key=lambda r: r[1])
rows = rows[-n:]
t0 = rows[0][1]
prev = None
for name, s, e in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:8.1f}  {name[:90]}")
    prev = e
