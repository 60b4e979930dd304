"""HBM traffic of one LM iteration of BASELINE config 4 from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_ba.sh
(separate rocprofv3 --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950's wide reads,
WRITE_SIZE exact; both in KB) -> profiles/<round>_ba_traffic_pmc.json, read by bench.py (ba.roofline.traffic).
usage: python tools/pmc_ba_traffic.py pmc_ba.txt out.json"""
import json
import re
import sys

KERNELS = ("ba_pair_pass_kernel", "ba_point_win_kernel", "ba_back_win_kernel", "chol_flow_kernel")


def main():
    cur, disp = None, {}
    vals = {k: {} for k in KERNELS}
    for line in open(sys.argv[1]):
        m = re.match(r"^(\S+) dispatches (\d+)", line)
        if m:
            cur = next((k for k in KERNELS if k in m.group(1)), None)
            if cur:
                disp[cur] = int(m.group(2))
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+(\S+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(2))
    per_kernel, total = {}, 0.0
    for k in KERNELS:
        if "FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k] and disp.get(k):
            b = (2.0 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024.0 / disp[k]
            per_kernel[k] = {"dispatches": disp[k], "FETCH_SIZE_KB": vals[k]["FETCH_SIZE"], "WRITE_SIZE_KB": vals[k]["WRITE_SIZE"],
                             "hbm_bytes_per_launch": b}
            total += b
    rec = {"workload": "BASELINE config 4 (200 cameras, 100k tracks): ba.bench_global_ba()", "per_kernel": per_kernel,
           "hbm_bytes_per_iteration": total,
           "note": "one launch of each kernel per LM iteration (the Cholesky's reads of the 8 MB reduced system stay in the L2 / Infinity Cache); "
                   "FETCH_SIZE doubled (gfx950: 64 B tallied per 128-B request), WRITE_SIZE exact"}
    json.dump(rec, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
