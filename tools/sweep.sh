#!/bin/bash
# ms per step for combinations of engine knobs.  usage:
#   GRID=8192x8192 STEPS=300 tools/sweep.sh "LBM_PASS_STEPS=3 LBM_BAND_ROWS=7" "LBM_PASS_STEPS=3 LBM_BAND_ROWS=14 LBM_PREFETCH=1" ...
export LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=${REPEATS:-3}
for combo in "$@"; do
  env $combo python bench.py --grid ${GRID:-8192x8192} --steps ${STEPS:-300} --warmup 30 --math ${MATH:-exact} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${GRID:-8192x8192} ${MATH:-exact} [$combo]: ms/step %.4f  kernel %.4f  spl %d' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['steps_per_launch']))"
done
