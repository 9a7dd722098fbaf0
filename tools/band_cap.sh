#!/bin/bash
# the band-height model with heights up to 64 rows (round 2) and up to 160 (round 3), size by size
for grid in 2048x2048 3072x3072 4096x4096 6144x6144 8192x8192 8192x4096 8192x2048; do
  for cap in 64 160; do
    LBM_BAND_MAX=$cap LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=3 python3 bench.py --grid $grid --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$grid cap $cap: ms/step %.4f' % l['ms_per_step'], 'MLUPS %.0f' % l['value'], l['roofline']['geometry'])"
  done
done
