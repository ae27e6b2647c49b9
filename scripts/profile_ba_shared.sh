#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the BA-1 solve with ONE camera shared by all images
# (scripts/ba_probe.py --shared 1). Output: gpurun_out/prof_ba_shared_$TAG/ba_kernel_stats.csv
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ba_shared_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10 --shared 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ba -- $PROBE > $OUT/probe.log 2>&1
grep "HIP rep" $OUT/probe.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220 > $OUT/ba_kernel_stats.csv
rm -rf $OUT/stats
