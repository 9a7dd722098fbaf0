#!/bin/bash
# us per step of the resident kernel by poll sleep (LBM_RESIDENT_SLEEP) on the wide grids
for s in 1 2 4 8 16; do echo "== sleep $s"; LBM_RESIDENT_SLEEP=$s python3 tools/resident_bench.py 4000 2>&1 | grep -E "^ *(512|768|1024) x" ; done
