#!/bin/bash
TAG=${1:-r02f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
timeout 600 python -m pytest tests/test_pm_gpu.py -m gpu -x -q > $OUT/pm_tests_full.log 2>&1; grep -v "rccl\|HIP version\|ROCm version\|Hostname\|RCCL" $OUT/pm_tests_full.log | grep -i "passed\|failed\|fault\|error" | tail -8 | tee $OUT/pm_tests.log
COLMAP_AMD_PM_PIPE=1 timeout 300 python -m pytest tests/test_pm_gpu.py -m gpu -x -q -k "sweep_direction or full_photometric or baseline_source or geometric or golden" 2>&1 | grep -v "rccl\|HIP version\|ROCm version\|Hostname\|RCCL" | tail -4 | tee $OUT/pm_tests_pipe.log
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1"
run() { echo "== $1" | tee -a $OUT/diag.log; shift; env "$@" 2>&1 | grep -E "sweep kernel|phase profile|Error|error|rror|LM" | tee -a $OUT/diag.log; }
run "plain C=2 conc=16 4 sweeps"      A=1 $PROBE --conc 16 --sweeps 4
run "plain C=3 conc=16 4 sweeps"      A=1 $PROBE --conc 16 --sweeps 4 --cols 3
run "plain C=2 conc=32 4 sweeps"      A=1 $PROBE --conc 32 --sweeps 4
run "plain C=2 conc=16 20 sweeps"     A=1 $PROBE --conc 16 --sweeps 20
run "pipe C=3 conc=16 4 sweeps"       COLMAP_AMD_PM_PIPE=1 $PROBE --conc 16 --sweeps 4 --cols 3
