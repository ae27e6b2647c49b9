#!/bin/bash
TAG=${1:-r02e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
COLMAP_AMD_PM_PIPE=0 python -m pytest tests/test_pm_gpu.py -m gpu -x -q 2>&1 | grep -v "rccl\|HIP version\|ROCm version\|Hostname\|RCCL" | tail -6 | tee $OUT/pm_tests_plain.log
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1"
run() { echo "== $1" | tee -a $OUT/diag.log; shift; env "$@" 2>&1 | grep -E "sweep kernel|phase profile|Error|error|rror|LM" | tee -a $OUT/diag.log; }
run "plain C=3 conc=16 4 sweeps"      COLMAP_AMD_PM_PIPE=0 $PROBE --conc 16 --sweeps 4
run "plain C=2 conc=16 4 sweeps"      COLMAP_AMD_PM_PIPE=0 $PROBE --conc 16 --sweeps 4 --cols 2
run "plain C=4 conc=16 4 sweeps"      COLMAP_AMD_PM_PIPE=0 $PROBE --conc 16 --sweeps 4 --cols 4
run "plain C=3 conc=32 4 sweeps"      COLMAP_AMD_PM_PIPE=0 $PROBE --conc 32 --sweeps 4
run "plain C=3 conc=16 lds_pad=1400 (12 WG/CU)" COLMAP_AMD_PM_PIPE=0 COLMAP_AMD_PM_LDS_PAD=1400 $PROBE --conc 16 --sweeps 4
run "pipe C=3 conc=16 4 sweeps"       A=1 $PROBE --conc 16 --sweeps 4
i=0
for ctrs in "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" | tee -a $OUT/diag.log
  COLMAP_AMD_PM_PIPE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep" --pmc $ctrs \
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
