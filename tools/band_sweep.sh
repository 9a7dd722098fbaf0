#!/bin/bash
# ms per step of a single periodic slab by band height (and cells per lane) of the two-step kernel
# usage: GRIDS="8192x8192 4096x4096" BANDS="4 5 6 7 8" CELLS="4" MATH=exact tools/band_sweep.sh
export LBM_BENCH_ALSO=0 LBM_FUSE2=1
for g in ${GRIDS:-8192x8192}; do for c in ${CELLS:-4}; do for b in ${BANDS:-4 5 6 7 8}; do
  LBM_LANE_CELLS=$c LBM_BAND_ROWS=$b python bench.py --grid $g --steps ${STEPS:-300} --warmup 30 --math ${MATH:-exact} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$g ${MATH:-exact} cells/lane $c band $b: ms/step %.4f' % d['ms_per_step'])"
done; done; done
