#!/bin/bash
# The cross-workgroup hand-offs of the bundle-adjustment kernels (the tickets of ba_pair_pass_kernel / ba_back_*_kernel,
# the flags of chol_flow_kernel) use relaxed agent-scope atomics; what orders them is the instruction sequence
# MI355X_MICROARCH.md prescribes: write-through (sc1) stores, `s_waitcnt vmcnt(0)` in every storing wave, a workgroup
# barrier, THEN one lane's atomic / flag store; readers load sc1 behind the barrier the polling lane joins.  The
# compiler is not told about that order by the memory model, so this script reads it off the ISA:
#   1. chol_flow_kernel uses no scratch memory (round 5: eighteen more live scalars made the compiler spill scalar
#      registers into scratch memory there, and its hand-offs raced -- solutions differed run to run; the pair pass
#      keeps a small array indexed at run time in scratch, which is data, not spills);
#   2. between the last data store (64 bits or wider) before a ticket add / flag store and that instruction stands
#      an `s_waitcnt` that drains vmcnt to 0 (read off the linear instruction stream: a guard, not a proof);
#   3. the kernels that read another workgroup's partials do so with sc1 loads.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
for f in ba_kernels ba_cholesky; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off \
        -Iinclude -Iorthosfm_amd/csrc -S --cuda-device-only -o $tmp/$f.s orthosfm_amd/csrc/$f.hip 2>/dev/null
done
python3 - $tmp/ba_kernels.s $tmp/ba_cholesky.s <<'PY'
import re, subprocess, sys
WATCH = ("ba_pair_pass_kernel", "ba_back_win_kernel", "ba_back_over_kernel", "chol_flow_kernel")
NO_SCRATCH = ("chol_flow_kernel",)      # (the pair pass keeps a 7-entry array indexed at run time there: data, not spills)
bad = 0
demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
for path in sys.argv[1:]:
    text = open(path).read()
    # 1. scratch (metadata: the fields of a kernel entry are sorted, .symbol comes behind .private_segment_fixed_size)
    for size, sym in re.findall(r"\.private_segment_fixed_size:\s+(\d+)\n(?:\s+\.(?!symbol).*\n)*\s+\.symbol:\s+(\S+)\.kd", text):
        name = demangle(sym)
        if any(w in name for w in NO_SCRATCH):
            ok = int(size) == 0
            print(f"{name.split('(')[0]:34s} scratch bytes {size}" + ("" if ok else "   <-- FAIL"))
            bad |= not ok
    # 2. / 3. instruction order per kernel
    cur, body = None, {}
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            n = demangle(m.group(1))
            cur = n if any(w in n for w in WATCH) else None
            if cur: body[cur] = []
            continue
        if line.startswith(".Lfunc_end"): cur = None
        t = line.strip()
        if cur and t and not t.startswith((";", ".", "//")): body[cur].append(t)
    is_data_store = lambda t: re.match(r"(global|buffer|flat)_store_dwordx[234]\b", t) is not None      # doubles / tiles / granules
    # tickets: returning agent-scope adds; flags / tags (chol_flow_kernel only): 32-bit stores -- relaxed agent-scope
    # atomic stores compile to `global_store_dword ... sc1`, the mailbox flags to plain ones.  (atomic max on the info /
    # abort words reports errors and orders nothing.)
    is_signal = lambda t, flow: re.match(r"global_atomic_add\b.* sc0", t) is not None or (flow and re.match(r"global_store_dword\b", t) is not None)
    for name, ins in body.items():
        flow = "chol_flow_kernel" in name
        signals = [i for i, t in enumerate(ins) if is_signal(t, flow)]
        n_bad = 0
        for a in signals:
            # walk back to the nearest data store: a wait that drains vmcnt must stand between it and the signal
            drained = False
            for j in range(a - 1, max(a - 6000, -1), -1):
                t = ins[j]
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t: drained = True
                if is_data_store(t):
                    if not drained:
                        n_bad += 1
                        print(f"      {ins[a]}   <- last data store {a - j} instructions before: {t}")
                    break
        sc1_loads = sum(1 for t in ins if t.startswith(("global_load", "buffer_load")) and " sc1" in t)
        sc1_stores = sum(1 for t in ins if is_data_store(t) and " sc1" in t)
        ok = n_bad == 0 and sc1_loads > 0 and sc1_stores > 0
        print(f"{name.split('(')[0]:34s} {len(signals)} flag / ticket instructions, {n_bad} not behind a drained vmcnt after the last data store; "
              f"sc1 data stores {sc1_stores}, sc1 loads {sc1_loads}" + ("" if ok else "   <-- FAIL"))
        bad |= not ok
sys.exit(1 if bad else 0)
PY
rc=$?
rm -rf $tmp
exit $rc
