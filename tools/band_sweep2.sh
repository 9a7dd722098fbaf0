#!/bin/bash
# ms per step of the default large-grid kernel by band height: tools/band_sweep2.sh <NXxNY> <steps> <band> [<band> ...]
grid=$1; steps=$2; shift 2
for b in "$@"; do
  LBM_BAND_ROWS=$b LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=3 python3 bench.py --grid $grid --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('band', $b, 'ms/step %.4f' % l['ms_per_step'], 'kernel %.4f' % l['roofline']['kernel_ms_per_step'], l['roofline']['geometry'])"
done
