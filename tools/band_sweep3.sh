#!/bin/bash
# ms per step by band height and XCD chunk: tools/band_sweep3.sh <NXxNY> <steps> "<band>:<chunk>" ...
grid=$1; steps=$2; shift 2
for bc in "$@"; do
  b=${bc%%:*}; c=${bc##*:}
  LBM_BAND_ROWS=$b LBM_XCD_CHUNK=$c LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=3 python3 bench.py --grid $grid --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$grid band $b chunk $c: ms/step %.4f' % l['ms_per_step'], 'MLUPS %.0f' % l['value'])"
done
