#!/bin/bash
# tile-kernel time (bench step, 50 views x 20k) of the product library and of every experimental build under
# orthosfm_amd/lib/exp (or the ones named):   gpurun -- 'tools/exp_variants.sh [lib.so ...]'
cd "$(dirname "$0")/.."
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(orthosfm_amd/lib/exp/*.so)
echo "product"; python tools/tile_timing.py 2>/dev/null | tail -1
for lib in "${libs[@]}"; do
    echo "$lib"; OSFM_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python tools/tile_timing.py 2>/dev/null | tail -1
done
echo "product again"; python tools/tile_timing.py 2>/dev/null | tail -1
