"""HBM traffic of the dominant matching kernel from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md
prescribes) -> profiles/<round>_match_traffic_pmc.json, read by bench.py.
usage: python tools/pmc_traffic.py fetch.csv write.csv pairs_in_launch out.json"""
import csv
import json
import sys

KERNEL = "match_tile_kernel<8, false, true, true>"


def total(path, counter):
    # the LAST launch of the kernel in the trace: the timed pass behind the warm-up passes
    # (rows come in dispatch order; several rows of one dispatch -- one per XCD / shader engine -- add up)
    per_dispatch = {}
    for row in csv.DictReader(open(path)):
        if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
            per_dispatch[int(row["Dispatch_Id"])] = per_dispatch.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
            GRID[int(row["Dispatch_Id"])] = int(row["Grid_Size"])
    return per_dispatch[max(per_dispatch)] if per_dispatch else 0.0


GRID = {}


def pairs_of_last_launch(features=20000):
    """Pairs in the launch the counters come from, from its grid: a pair of two `features`-descriptor views
    is ceil(n / 256) row blocks of 256 threads, one column segment each (osfm_match_all cuts a large
    call into three launches, so this is not the pair count of the call)."""
    # (round 5: a launch with this many row blocks runs ONE column segment per row block, match_api.hip::choose_seg_cols)
    nrb, nseg = (features + 255) // 256, 1
    return round(GRID[max(GRID)] / 256 / (nrb * nseg)) if GRID else 0


def main():
    fetch_kb = total(sys.argv[1], "FETCH_SIZE")
    write_kb = total(sys.argv[2], "WRITE_SIZE")
    pairs = pairs_of_last_launch() if sys.argv[3] == "auto" else int(sys.argv[3])
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    rec = {
        "kernel": KERNEL,
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py "
                   "--steps 1 --warmup 2 --no-ba --no-verify --no-cpu-baseline --no-e2e --no-realistic; the last (warmed) launch",
        "pairs_in_launch": pairs,
        "FETCH_SIZE_KB": fetch_kb,
        "WRITE_SIZE_KB": write_kb,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled "
                      "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        "hbm_bytes_per_launch": hbm,
        "hbm_bytes_per_pair": hbm / pairs,
    }
    json.dump(rec, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
