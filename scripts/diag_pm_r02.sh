#!/bin/bash
# Round-2 diagnostics of pm_sweep_kernel on the GPU box (outputs under gpurun_out/diag_$TAG):
# occupancy curve (LDS padding), workgroup->XCD mapping, workgroup shapes, per-sweep cache counters.
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc 16"
run() { echo "== $1" | tee -a $OUT/diag.log; shift; env "$@" 2>&1 | grep -E "sweep kernel|phase profile|Error|error" | tee -a $OUT/diag.log; }
run "baseline 4 sweeps"            A=1 $PROBE --sweeps 4
run "xcd_map=1 4 sweeps"           COLMAP_AMD_PM_XCD_MAP=1 $PROBE --sweeps 4
for pad in 6000 12000 20000 33000; do
  run "lds_pad=$pad (occupancy) 2 sweeps" COLMAP_AMD_PM_LDS_PAD=$pad $PROBE --sweeps 2
done
run "C=4 T=64 2 sweeps"            A=1 $PROBE --sweeps 2 --cols 4 --threads 64
run "C=2 T=64 2 sweeps"            A=1 $PROBE --sweeps 2 --cols 2 --threads 64
run "C=2 T=64 xcd_map=1 2 sweeps"  COLMAP_AMD_PM_XCD_MAP=1 $PROBE --sweeps 2 --cols 2 --threads 64
run "conc=32 xcd_map=0 2 sweeps"   A=1 $PROBE --sweeps 2 --conc 32
run "conc=32 xcd_map=1 2 sweeps"   COLMAP_AMD_PM_XCD_MAP=1 $PROBE --sweeps 2 --conc 32
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -c . $OUT/counters_avail.txt
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs" | tee -a $OUT/diag.log
  timeout 300 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep_kernel" --pmc $ctrs \
    -d $OUT/pmc_p$i -o pmc -- $PROBE --sweeps 8 > $OUT/pmc_p$i.log 2>&1 || tail -5 $OUT/pmc_p$i.log
  python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
  find $OUT/pmc_p$i -type f -size +1M -delete
done
rm -rf $OUT/pmc_p*/
cat $OUT/pmc_per_dispatch.json | head -150
du -sh $OUT
