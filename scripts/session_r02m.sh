#!/bin/bash
# GPU session: BA tests (new 12-parameter models), C++ host tests, BA bench sanity (register tiers unchanged).
TAG=${1:-r02m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_cpp_host.py -m gpu -q > $OUT/ba_tests.log 2>&1; echo "ba rc=$?"; tail -12 $OUT/ba_tests.log | cut -c1-300
timeout 300 python scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10 > $OUT/ba_probe.log 2>&1; tail -3 $OUT/ba_probe.log
