"""Per-kernel summary (calls, total, average, share) of rocprofv3 rocpd databases -- the same table
`--stats` prints, for runs whose output format was left at the default.  A traced command that starts child
processes (bench.py's drop-in leg runs a C++ host program) leaves one database per process: all of them are
read and summed.  usage: python tools/rocpd_stats.py results.db [more.db ...] [--out out.csv]"""
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    out = None
    if "--out" in args:
        i = args.index("--out")
        out = args[i + 1]
        del args[i:i + 2]
    acc = {}
    for path in args:
        db = sqlite3.connect(path)
        tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
        if "kernels" not in tables:
            continue
        for name, calls, tot, mn, mx in db.execute(
                "select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name"):
            a = acc.setdefault(name, [0, 0, None, 0])
            a[0] += calls
            a[1] += tot
            a[2] = mn if a[2] is None else min(a[2], mn)
            a[3] = max(a[3], mx)
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for name, (calls, tot, mn, mx) in rows:
        lines.append(f'"{name}",{calls},{tot},{tot / calls:.1f},{100.0 * tot / total:.2f},{mn},{mx}')
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
