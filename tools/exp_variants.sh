#!/bin/bash
# tile-kernel time (both passes) of every experimental build under orthosfm_amd/lib/exp
for lib in "" orthosfm_amd/lib/exp/*.so; do
    OSFM_HIP_LIBRARY=${lib:+$PWD/$lib} python tools/tile_timing.py 2>/dev/null | tail -1
done
