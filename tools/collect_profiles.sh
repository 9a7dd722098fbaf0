#!/bin/bash
# Round profiles: rocprofv3 --kernel-trace --stats of bench.py for every configuration the bench line quotes, plus the
# PMC passes behind roofline.traffic / limiter (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes).  Output under gpurun_out/prof/ (copy what is to be judged into profiles/).
#   usage (on the GPU box): tools/collect_profiles.sh [tag]
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # name, then bench.py arguments; environment from the caller
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/${tag}_${name}_bench_profiled.json 2> $O/tmp_$name.err
  cp $O/tmp_$name/p_kernel_stats.csv $O/${tag}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/tmp_$name $O/tmp_$name.err
  echo "== $name"; head -4 $O/${tag}_${name}_kernel_stats.csv
}
export LBM_BENCH_ALSO=0
stats 8192 --steps 20 --warmup 5
stats 1024 --grid 1024x1024 --steps 2000 --warmup 200
stats 16384 --grid 16384x16384 --steps 100 --warmup 10
LBM_FUSE2=0 stats 8192_one_step --steps 20 --warmup 5
LBM_BENCH_RANK_API=1 LBM_FORCE_HALO=1 stats rank_share_8192x1024 --grid 8192x1024 --steps 400 --warmup 40
LBM_BENCH_RANK_API=1 LBM_FORCE_HALO=1 stats rank_share_16384x2048 --grid 16384x2048 --steps 200 --warmup 20
# PMC of the dominant kernel at the default geometry (8192^2) and of the one-step kernel
pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/tmp_pmc -o p -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  for c in "$@"; do python3 $R/tools/pmc_mean.py $O/tmp_pmc/p_counter_collection.csv $c | grep -E "stepk|step2|step_vec4" | sed "s/^/$name,/" >> $O/${tag}_pmc_8192.csv; done
  rm -rf $O/tmp_pmc
}
rm -f $O/${tag}_pmc_8192.csv
pmc default FETCH_SIZE
pmc default WRITE_SIZE
pmc default SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc default SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
export LBM_FUSE2=0
pmc one_step FETCH_SIZE
pmc one_step WRITE_SIZE
pmc one_step SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
cat $O/${tag}_pmc_8192.csv
