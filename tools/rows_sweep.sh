#!/bin/bash
# us per two-step pass of a single periodic slab 8192 x R for several R (fixed-overhead fit)
export LBM_BENCH_ALSO=0 LBM_FUSE2=1 LBM_LANE_CELLS=4 LBM_BAND_ROWS=${BAND:-6}
for kv in "$@"; do export "$kv"; done
for r in ${ROWS:-192 384 768 1152 1536 2304 3072 4608 6144}; do
  python bench.py --grid 8192x$r --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$* rows $r  us/pass %.2f  kernel us/pass %.2f' % (2e3*d['ms_per_step'], 2e3*d['roofline']['kernel_ms_per_step']))"
done
