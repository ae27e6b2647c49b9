#!/bin/bash
# Round-2 diagnostics, part b: single-wave sweep kernel vs the two-wave kernel, launch sizes that fit
# the resident set (lockstep), workgroup -> XCD mappings, counters. Outputs: gpurun_out/diag_$TAG.
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1"
run() { echo "== $1" | tee -a $OUT/diag.log; shift; env "$@" 2>&1 | grep -E "sweep kernel|phase profile|Error|error|rror" | tee -a $OUT/diag.log; }
run "two-wave kernel conc=16 4 sweeps"  COLMAP_AMD_PM_WAVE=0 $PROBE --conc 16 --sweeps 4
run "wave kernel conc=16 4 sweeps"      A=1 $PROBE --conc 16 --sweeps 4
run "wave kernel conc=16 xcd_map=2"     COLMAP_AMD_PM_XCD_MAP=2 $PROBE --conc 16 --sweeps 4
run "wave kernel conc=4"                A=1 $PROBE --conc 4 --sweeps 4
run "wave kernel conc=4 xcd_map=2"      COLMAP_AMD_PM_XCD_MAP=2 $PROBE --conc 4 --sweeps 4
run "wave kernel conc=5 xcd_map=2"      COLMAP_AMD_PM_XCD_MAP=2 $PROBE --conc 5 --sweeps 4
run "wave kernel conc=6 xcd_map=2"      COLMAP_AMD_PM_XCD_MAP=2 $PROBE --conc 6 --sweeps 4
run "wave kernel conc=8 xcd_map=2"      COLMAP_AMD_PM_XCD_MAP=2 $PROBE --conc 8 --sweeps 4
run "wave kernel conc=32"               A=1 $PROBE --conc 32 --sweeps 4
run "wave kernel conc=16 lds_pad=2600 (10 WG/CU)" COLMAP_AMD_PM_LDS_PAD=2600 $PROBE --conc 16 --sweeps 2
run "wave kernel conc=16 lds_pad=7000 (8 WG/CU)"  COLMAP_AMD_PM_LDS_PAD=7000 $PROBE --conc 16 --sweeps 2
run "wave kernel conc=16 C=2"           A=1 $PROBE --conc 16 --sweeps 2 --cols 2
run "wave kernel conc=16 C=3"           A=1 $PROBE --conc 16 --sweeps 2 --cols 3
run "wave kernel conc=16 C=6"           A=1 $PROBE --conc 16 --sweeps 2 --cols 6
i=0
for spec in "16 0" "4 2"; do
  set -- $spec
  for ctrs in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
              "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    echo "== pmc pass $i (conc=$1 xcd_map=$2): $ctrs" | tee -a $OUT/diag.log
    COLMAP_AMD_PM_XCD_MAP=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep" --pmc $ctrs \
      -d $OUT/pmc_p$i -o pmc -- $PROBE --conc $1 --sweeps 6 > $OUT/pmc_p$i.log 2>&1 || tail -5 $OUT/pmc_p$i.log
    python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
    find $OUT/pmc_p$i -type f -size +1M -delete
  done
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
du -sh $OUT
