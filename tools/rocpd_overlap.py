"""How much of the time kernels matching pattern A ran did kernels matching pattern B run too (rocpd database of
rocprofv3 --kernel-trace)?  usage: python tools/rocpd_overlap.py results.db ransac_kernel match_tile_kernel"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pa, pb = sys.argv[2], sys.argv[3]
rows = list(db.execute("select name, start, end from kernels order by start"))
A = [(s, e) for n, s, e in rows if pa in n]
B = [(s, e) for n, s, e in rows if pb in n]
ov = 0
j = 0
for s, e in A:
    for bs, be in B:
        if be <= s: continue
        if bs >= e: break
        ov += min(e, be) - max(s, bs)
ta = sum(e - s for s, e in A)
span = rows[-1][2] - rows[0][1]
busy = 0
cur_s, cur_e = None, None
for n, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{pa}: {len(A)} launches, {ta / 1e6:.1f} ms; {ov / 1e6:.1f} ms of it beside {pb} ({100.0 * ov / max(ta, 1):.0f} %)")
print(f"trace span {span / 1e6:.1f} ms, some kernel running {busy / 1e6:.1f} ms, sum of kernel times {sum(e - s for _, s, e in rows) / 1e6:.1f} ms")
