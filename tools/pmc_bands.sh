#!/bin/bash
# PMC counters of the two-step kernel at several band heights (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp LBM_BENCH_ALSO=0
for b in ${BANDS:-7 32}; do
  export LBM_BAND_ROWS=$b
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
    out=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$b
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    for c in $grp; do python3 $GRAFT_REPO_ROOT/tools/pmc_mean.py $out/p_counter_collection.csv $c | grep step2 | sed "s/^/band $b: /"; done
    rm -rf $out
  done
done
