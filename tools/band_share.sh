#!/bin/bash
# one rank's share (8192x1024 through the rank pipeline, RCCL self-exchange) under band heights / steps per pass:
#   tools/band_share.sh "<env assignments>" ...     e.g. tools/band_share.sh "" "LBM_BAND_ROWS=34" "LBM_PASS_STEPS=3"
for v in "$@"; do
  env $v LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=3 LBM_BENCH_RANK_API=1 LBM_FORCE_HALO=1 python3 bench.py --grid ${GRID:-8192x1024} --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']; print('${v:-default}', '-> band', r['geometry']['band_rows'], 'steps/launch', r['steps_per_launch'], 'ms/step %.5f' % l['ms_per_step'], 'kernel %.5f' % r['kernel_ms_per_step'], 'MLUPS %.0f' % l['value'])"
done
