#!/bin/bash
# GPU session: HIP fusion against the oracle, the two BA tests that failed in r02i, fusion timing.
TAG=${1:-r02j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 900 python -m pytest tests/test_fusion.py -m gpu -x -q > $OUT/fusion_tests.log 2>&1; echo "fusion rc=$?"
tail -15 $OUT/fusion_tests.log
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -q -k "sharded or more_camera_models" > $OUT/ba_tests.log 2>&1; echo "ba rc=$?"
tail -8 $OUT/ba_tests.log
timeout 600 python scripts/fusion_probe.py > $OUT/fusion_probe.log 2>&1; echo "probe rc=$?"
tail -12 $OUT/fusion_probe.log
