#!/bin/bash
# Tile-kernel time of the bench step with the segment length of the correction-free problems forced
# (OSFM_SEG_TILES, whole cycles of 16 tiles): the per-workgroup fixed cost (prologue, row merge, partials)
# is the intercept of time against workgroups per row block.   gpurun -- 'tools/seg_intercept.sh [base.so]'
cd "$(dirname "$0")/.."
for t in 16 32 64 128 160 320; do
    echo "seg_tiles=$t"
    OSFM_SEG_TILES=$t python tools/tile_timing.py 2>/dev/null | tail -1
done
echo "auto"
python tools/tile_timing.py 2>/dev/null | tail -1
for lib in "$@"; do
    echo "library $lib"
    OSFM_HIP_LIBRARY=$PWD/$lib python tools/tile_timing.py 2>/dev/null | tail -1
done
