#!/bin/bash
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ba_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ba -- python $ROOT/scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10 > $OUT/probe.log 2>&1
tail -2 $OUT/probe.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -30 "$f" | cut -c1-200 > $OUT/ba_kernel_stats.csv
cat $OUT/ba_kernel_stats.csv
rm -rf $OUT/stats
