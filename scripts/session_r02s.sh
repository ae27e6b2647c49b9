#!/bin/bash
TAG=${1:-r02s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
for i in 1 2 3; do timeout 300 python scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10 2>&1 | tail -1; done
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_cpp_host.py tests/test_bundle_adjuster_cli.py -m gpu -q > $OUT/ba_tests.log 2>&1; echo "ba rc=$?"; grep -E "^E  |passed|failed|FAILED" $OUT/ba_tests.log | head -20 | cut -c1-250
