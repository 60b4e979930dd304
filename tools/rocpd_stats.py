"""Per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd
database -- the same table `--stats` prints, for runs whose output format was
left at the default.  usage: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for name, calls, tot, avg, mn, mx in rows:
        lines.append(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}')
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
