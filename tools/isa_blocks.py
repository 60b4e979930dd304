#!/usr/bin/env python3
"""Instruction census of one kernel's ISA (hipcc -S output): per basic block, how many vector,
matrix, LDS, vector-memory and scalar instructions it holds, so that the share of the vector
instructions inside and outside the tile loop can be read off the code itself.
    tools/isa_blocks.py file.s '<substring of the mangled kernel name>' [min instructions per block]"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return op.split()[0]
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    blocks = OrderedDict()
    cur, inside = None, False
    for line in open(path):
        if re.match(r"^_Z\w+:", line):
            inside = key in line.split(":")[0]
            cur = "entry"
            if inside:
                blocks[cur] = Counter()
            continue
        if not inside:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        blocks[cur][classify(op)] += 1
        blocks[cur]["op:" + op] += 1
    tot = Counter()
    for name, c in blocks.items():
        n = sum(v for k, v in c.items() if not k.startswith("op:"))
        for k, v in c.items():
            tot[k] += v
        if n >= minsz:
            print(f"{name:14s} n={n:5d} " + " ".join(f"{k}={c[k]}" for k in ("valu", "mfma", "lds", "vmem", "salu", "s_waitcnt", "s_nop", "s_barrier") if c[k]))
    print("TOTAL".ljust(14), " ".join(f"{k}={tot[k]}" for k in ("valu", "mfma", "lds", "vmem", "salu", "s_waitcnt", "s_nop", "s_barrier")))
    if len(sys.argv) > 4:
        blk = blocks[sys.argv[4]]
        for k, v in sorted(((k, v) for k, v in blk.items() if k.startswith("op:")), key=lambda kv: -kv[1]):
            print(f"   {k[3:]:28s} {v}")


if __name__ == "__main__":
    main()
