#!/bin/bash
# GPU session: fusion (new schedule) tests + probe, controller exchange tests, PatchMatch phase ablation.
TAG=${1:-r02l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
ulimit -c 0
cd $ROOT
timeout 900 python -m pytest tests/test_fusion.py -m gpu -x -q > $OUT/fusion_tests.log 2>&1; echo "fusion rc=$?"; tail -4 $OUT/fusion_tests.log
timeout 600 python scripts/fusion_probe.py > $OUT/fusion_probe.log 2>&1; echo "probe rc=$?"; tail -4 $OUT/fusion_probe.log
timeout 900 python -m pytest tests/test_pm_gpu.py -m gpu -q -k "two_rank or cli_on_a_workspace or controller" > $OUT/pm_ctl_tests.log 2>&1; echo "ctl rc=$?"; tail -4 $OUT/pm_ctl_tests.log
for ab in 0 1 3 7; do
  for cols in 2 4; do
    COLMAP_AMD_PM_ABLATE=$ab timeout 300 python bench.py --steps 1 --warmup 1 --no-ba --no-cpu-baseline --cols $cols > $OUT/ablate_${ab}_c${cols}.json 2> $OUT/ablate_${ab}_c${cols}.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/ablate_${ab}_c${cols}.json").read().strip().splitlines()[-1])
    print("ablate=$ab cols=$cols avg_launch_ms=%.1f value=%.2f" % (d["roofline"]["avg_launch_ms"], d["value"]))
except Exception as e:
    print("ablate=$ab cols=$cols failed", e)
PY
  done
done
