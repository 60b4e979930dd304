"""Sums rocprofv3 --pmc counter values per kernel name.
usage: python tools/pmc_summary.py counter_collection.csv [kernel-substring]"""
import csv
import collections
import sys


def main():
    sel = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for row in csv.DictReader(open(sys.argv[1])):
        name = row["Kernel_Name"]
        if sel not in name:
            continue
        short = name.split("(")[0][-60:]
        acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[short].add(row["Dispatch_Id"])
    for k, d in acc.items():
        print(k, "dispatches", len(calls[k]))
        for c, v in sorted(d.items()):
            print(f"   {c:32s} {v:.4g}")


if __name__ == "__main__":
    main()
