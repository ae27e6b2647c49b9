#!/bin/bash
# Runs on the GPU box: the first measurement of the exact tier's round-4 kernels (written after the round's GPU budget
# was spent). Kernel-trace stats of the BA-1 solve with SPARSE_SCHUR (6 exact Newton steps, scripts/ba_probe.py --lst 3),
# once with the pair-major formation (default) and once with the point-major one (COLMAP_AMD_BA_FORM_PAIRS=0), then the
# iterative tier (10 LM iterations) for the Gram / schur_g / finalize / tail kernels. ~1 minute of box time.
# Outputs: gpurun_out/prof_ba_exact_$TAG/{exact_pairs,exact_points,iterative}_kernel_stats.csv + the probe logs; copy
# the summaries into profiles/ afterwards (profiles/r05_ba_exact_tier_kernel_stats.csv ...).
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ba_exact_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
run() {  # name, env assignment (or "X=0"), probe arguments
  local name=$1 envv=$2; shift 2
  env $envv timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o ba -- \
      python $ROOT/scripts/ba_probe.py --frames 1000 --points 200000 --track 10 "$@" > $OUT/${name}_probe.log 2>&1
  tail -2 $OUT/${name}_probe.log
  f=$(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" | cut -c1-220 > $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/stats_$name
  cut -c1-150 $OUT/${name}_kernel_stats.csv | head -16
}
echo "== exact tier, pair-major formation"
run exact_pairs X=0 --iters 6 --lst 3
echo "== exact tier, point-major formation"
run exact_points X=0 --iters 6 --lst 3 --switch COLMAP_AMD_BA_FORM_PAIRS=0
echo "== iterative tier"
run iterative X=0 --iters 10
du -sh $OUT
