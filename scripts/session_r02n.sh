#!/bin/bash
# GPU session: BA tests incl. position priors; BA-1 probe (regression check).
TAG=${1:-r02n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -q -x -k "prior or error_behaviour" > $OUT/prior_tests.log 2>&1; echo "prior rc=$?"; tail -25 $OUT/prior_tests.log | cut -c1-250
timeout 300 python scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10 > $OUT/ba_probe.log 2>&1; tail -2 $OUT/ba_probe.log
