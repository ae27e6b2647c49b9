#!/bin/bash
TAG=${1:-r02i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_tests_full.log 2>&1; grep -v "rccl\|HIP version\|ROCm version\|Hostname\|RCCL" $OUT/gpu_tests_full.log | grep -i "passed\|failed\|fault\|error" | tail -12 | tee $OUT/gpu_tests.log
bash scripts/profile_ba.sh $TAG 2>&1 | tail -45
