#!/bin/bash
# Round profiles: rocprofv3 --kernel-trace --stats of bench.py for every configuration the bench line quotes, plus the
# PMC passes behind roofline.traffic / limiter (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes).  Output under gpurun_out/prof/ (copy what is to be judged into profiles/).
#   usage (on the GPU box): tools/collect_profiles.sh [tag]
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # name, then bench.py arguments; environment from the caller
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -o p -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/${tag}_${name}_bench_profiled.json 2> $O/tmp_$name.err
  f=$(find $O/tmp_$name -name 'p_kernel_stats.csv' | head -1)
  cp $f $O/${tag}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/tmp_$name $O/tmp_$name.err
  echo "== $name"; head -4 $O/${tag}_${name}_kernel_stats.csv
}
export LBM_BENCH_ALSO=0
stats 8192 --steps 20 --warmup 5
stats 1024_resident --grid 1024x1024 --steps 2000 --warmup 200
LBM_RESIDENT=0 stats 1024_per_pass --grid 1024x1024 --steps 2000 --warmup 200
stats 256_resident --grid 256x256 --steps 4000 --warmup 400
stats 128_resident --grid 128x128 --steps 4000 --warmup 400
stats 16384 --grid 16384x16384 --steps 100 --warmup 10
LBM_FUSE2=0 stats 8192_one_step --steps 20 --warmup 5
LBM_BENCH_RANK_API=1 LBM_FORCE_HALO=1 stats rank_share_8192x1024 --grid 8192x1024 --steps 400 --warmup 40
LBM_BENCH_RANK_API=1 LBM_FORCE_HALO=1 stats rank_share_16384x2048 --grid 16384x2048 --steps 200 --warmup 20
# PMC of the dominant kernels: 8192^2 (stepk_pk) and the resident kernel at 1024^2
cd $R
tools/pmc_grid.sh ${tag}_pmc_8192 8192x8192 12 2 stepk 3
tools/pmc_grid.sh ${tag}_pmc_1024_resident 1024x1024 2000 200 resident 1
LBM_FUSE2=0 tools/pmc_grid.sh ${tag}_pmc_8192_one_step 8192x8192 12 2 step_vec4 3
