#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <variant.so> <NXxNY> <steps> [rounds]
v=$1; grid=$2; steps=$3; rounds=${4:-3}
for i in $(seq $rounds); do
  for lib in "" "$v"; do
    LBM_LIB=$lib LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=3 python3 bench.py --grid $grid --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('${lib:-tree}'.split('/')[-1], '$grid ms/step %.4f' % l['ms_per_step'], 'kernel %.4f' % l['roofline']['kernel_ms_per_step'])"
  done
done
