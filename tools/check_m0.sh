#!/bin/bash
# The correction-free tile kernels and match_special_wide_kernel write m0 inside an asm statement the
# compiler cannot be told about (m0 is a reserved register: not accepted as a clobber).  This is only
# sound while nothing else in those kernels touches m0.  Disassemble match_kernels.hip and
# match_special.hip and list, per kernel, every instruction that mentions m0; for the <.., true, true>
# (C0) tile kernels and the wide special kernel only the hand-written "s_mov_b32 m0" in front of
# global_load_lds_dwordx4 may appear.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off \
    -Iinclude -S --cuda-device-only -o $tmp/mk.s orthosfm_amd/csrc/match_kernels.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off \
    -Iinclude -S --cuda-device-only -o $tmp/ms.s orthosfm_amd/csrc/match_special.hip 2>/dev/null
cat $tmp/ms.s >> $tmp/mk.s
python3 - $tmp/mk.s <<'PY'
import re, sys, subprocess
cur = None; bad = 0; seen = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_ZN4osfm(?:17match_tile_kernel|25match_special_wide_kernel|20match_special_kernel)\w+):', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        seen[cur] = []
        continue
    if line.startswith('.Lfunc_end'): cur = None
    if cur and re.search(r'\bm0\b', line) and not line.strip().startswith(';'):
        seen[cur].append(line.strip())
for k, v in seen.items():
    c0 = 'true, true>' in k or 'match_special_wide_kernel' in k
    others = [x for x in v if not x.startswith('s_mov_b32 m0')]
    print(f"{k.split('(')[0]}: {len(v)} m0 instructions, {len(others)} other than s_mov_b32 m0" + (" [hand-written m0]" if c0 else ""))
    if c0 and others:
        bad = 1
        for x in others[:5]: print("   ", x)
sys.exit(bad)
PY
rm -rf $tmp
