#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, as MI355X_MICROARCH.md prescribes) of bench.py on one grid:
#   tools/pmc_grid.sh <tag> <NXxNY> <steps> <warmup> [kernel-name regex] [min launches]   (on the GPU box; env passes through)
# appends "<tag>,<kernel>,<counter>,<launches>,<mean>,<min>,<max>" lines to gpurun_out/prof/<tag>.csv
tag=$1; grid=$2; steps=$3; warm=$4; pat=${5:-step}; minl=${6:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LBM_BENCH_ALSO=0 LBM_BENCH_REPEATS=1 LBM_BENCH_PREWARM_S=0
pmc() {
  rm -rf $O/tmp_pmc
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/tmp_pmc -o p -- python3 $R/bench.py --grid $grid --steps $steps --warmup $warm --no-cpu-baseline > /dev/null 2> $O/tmp_pmc.err
  f=$(find $O/tmp_pmc -name 'p_counter_collection.csv' | head -1)
  for c in "$@"; do python3 $R/tools/pmc_mean.py $f $c $minl | grep -E "$pat" | sed "s/^/$tag,/" >> $O/$tag.csv; done
  rm -rf $O/tmp_pmc
}
rm -f $O/$tag.csv
pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc FETCH_SIZE
pmc WRITE_SIZE
cat $O/$tag.csv
