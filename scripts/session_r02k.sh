#!/bin/bash
# GPU session: full GPU suite, fusion timing probe.
TAG=${1:-r02k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests_full.log 2>&1; echo "suite rc=$?"
grep -v "rccl\|HIP version\|ROCm version\|Hostname\|RCCL" $OUT/gpu_tests_full.log | grep -i "passed\|failed\|fault\|error" | tail -12 | tee $OUT/gpu_tests.log
timeout 600 python scripts/fusion_probe.py > $OUT/fusion_probe.log 2>&1; echo "probe rc=$?"
tail -6 $OUT/fusion_probe.log
