#!/bin/bash
# BASELINE config 4's loop rate under the knobs of the ordered factorisation and the pair pass (one process per setting).
cd "$(dirname "$0")/.."
for k in 2 3 4 5 6; do echo "arcs $k"; OSFM_BA_ORDER_ARCS=$k python tools/ba_loop_rate.py 2>/dev/null | tail -1; done
for c in 256 512 1024; do echo "pair chunk $c"; OSFM_BA_PAIR_CHUNK=$c python tools/ba_loop_rate.py 2>/dev/null | tail -1; done
for g in 1 4 8 16; do echo "pair group $g"; OSFM_BA_PAIR_GROUP=$g python tools/ba_loop_rate.py 2>/dev/null | tail -1; done
