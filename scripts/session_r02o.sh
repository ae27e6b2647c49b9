#!/bin/bash
TAG=${1:-r02o}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -k "prior or dense" > $OUT/prior_tests.log 2>&1; echo "prior rc=$?"; grep -E "^E  |passed|failed" $OUT/prior_tests.log | head -30 | cut -c1-250
timeout 1200 python -m pytest tests/test_ba_gpu.py tests/test_cpp_host.py tests/test_bundle_adjuster_cli.py -m gpu -q -k "not prior and not dense" > $OUT/ba_rest.log 2>&1; echo "rest rc=$?"; grep -E "^E  |passed|failed" $OUT/ba_rest.log | head -30 | cut -c1-250
