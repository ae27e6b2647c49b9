#!/bin/bash
TAG=${1:-r02r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_cpp_host.py -m gpu -q -k "more_camera_models or pose_prior or cpp" > $OUT/ba_tests.log 2>&1; echo "ba rc=$?"; grep -E "^E  |passed|failed|FAILED" $OUT/ba_tests.log | head -30 | cut -c1-250
