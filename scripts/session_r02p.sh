#!/bin/bash
TAG=${1:-r02p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 900 python -m pytest tests/test_fusion.py tests/test_cpp_host.py -m gpu -x -q > $OUT/fusion_tests.log 2>&1; echo "fusion rc=$?"; tail -4 $OUT/fusion_tests.log
timeout 600 python scripts/fusion_probe.py > $OUT/fusion_probe.log 2>&1; echo "probe rc=$?"; tail -4 $OUT/fusion_probe.log
