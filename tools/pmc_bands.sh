#!/bin/bash
# PMC counters of the step kernel at several band heights (one rocprofv3 --pmc pass per counter group).
# usage: BANDS="7 32" MATH=exact PMC_GROUPS="A B C;D E" GRID=8192x8192 tools/pmc_bands.sh     (groups separated by ';')
cd /tmp && export TMPDIR=/tmp LBM_BENCH_ALSO=0
DEFAULT="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE;SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
IFS=';' read -ra GRPS <<< "${PMC_GROUPS:-$DEFAULT}"
for b in ${BANDS:-7 32}; do
  export LBM_BAND_ROWS=$b
  for grp in "${GRPS[@]}"; do
    out=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$b
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --grid ${GRID:-8192x8192} --math ${MATH:-exact} --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    for c in $grp; do python3 $GRAFT_REPO_ROOT/tools/pmc_mean.py $out/p_counter_collection.csv $c | grep -E "step2|stepk|step_vec4" | sed "s/^/band $b ${MATH:-exact}: /"; done
    rm -rf $out
  done
done
