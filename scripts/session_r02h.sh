#!/bin/bash
# Round-2 consolidated run: all GPU tests, smoke, bench line, PatchMatch + BA profiles.
TAG=${1:-r02h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests_full.log 2>&1; grep -v "rccl\|HIP version\|ROCm version\|Hostname\|RCCL" $OUT/gpu_tests_full.log | grep -i "passed\|failed\|fault\|error" | tail -8 | tee $OUT/gpu_tests.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -i "smoke" | tee $OUT/smoke.log
timeout 600 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; tail -c 4000 $OUT/bench.json
bash scripts/profile_pm.sh $TAG 16 2>&1 | tail -25
bash scripts/profile_ba.sh $TAG 2>&1 | tail -60
