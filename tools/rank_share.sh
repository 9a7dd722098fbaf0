#!/bin/bash
# One rank's share of a strong-scaling run, through the one-process-per-GPU pipeline (RCCL
# self-exchange, interior/boundary split): ms per step for the given slab sizes.
# usage: GRIDS="8192x2048 8192x1024" tools/rank_share.sh [ENV=VAL ...]
export LBM_BENCH_RANK_API=1 LBM_FORCE_HALO=1 LBM_BENCH_ALSO=0
for kv in "$@"; do export "$kv"; done
for g in ${GRIDS:-8192x4096 8192x2048 8192x1024}; do
  python bench.py --grid $g --steps ${STEPS:-400} --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$g $*', 'ms/step %.4f kernel %.4f  MLUPS %.0f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['value']))"
done
