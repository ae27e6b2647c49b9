#!/bin/bash
TAG=${1:-r02c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
python -m pytest tests/test_pm_gpu.py -m gpu -x -q 2>&1 | grep -v "rccl\|HIP version\|ROCm version\|Hostname" | tail -8 | tee $OUT/pm_tests.log
python -m pytest tests/test_ba_gpu.py tests/test_cpp_host.py tests/test_bundle_adjuster_cli.py -m gpu -x -q 2>&1 | grep -v "rccl\|HIP version\|ROCm version\|Hostname" | tail -8 | tee $OUT/ba_tests.log
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1"
run() { echo "== $1" | tee -a $OUT/diag.log; shift; env "$@" 2>&1 | grep -E "sweep kernel|phase profile|Error|error|rror|LM" | tee -a $OUT/diag.log; }
run "wave kernel C=3 conc=16 4 sweeps"      A=1 $PROBE --conc 16 --sweeps 4
run "wave kernel C=3 conc=32 4 sweeps"      A=1 $PROBE --conc 32 --sweeps 4
run "wave kernel C=4 conc=16 4 sweeps"      A=1 $PROBE --conc 16 --sweeps 4 --cols 4
run "wave kernel C=2 conc=16 4 sweeps"      A=1 $PROBE --conc 16 --sweeps 4 --cols 2
run "wave kernel C=3 conc=16 8 sweeps"      A=1 $PROBE --conc 16 --sweeps 8
run "BA probe"                              A=1 python $ROOT/scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" | tee -a $OUT/diag.log
  timeout 300 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep" --pmc $ctrs \
    -d $OUT/pmc_p$i -o pmc -- $PROBE --conc 16 --sweeps 4 > $OUT/pmc_p$i.log 2>&1 || tail -5 $OUT/pmc_p$i.log
  python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
  find $OUT/pmc_p$i -type f -size +1M -delete
done
rm -rf $OUT/pmc_p*/
python - <<PY
import json
d=json.load(open("$OUT/pmc_per_dispatch.json"))
for k,v in d.items():
    print(k)
    for r in v:
        print({a:(f"{b:.3e}" if isinstance(b,float) else b) for a,b in r.items()})
PY
