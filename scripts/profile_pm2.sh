#!/bin/bash
# Focused PMC passes of the PatchMatch sweep kernel (run on the GPU box); summaries -> gpurun_out/prof_$TAG
TAG=${1:-r01b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>&1 | grep -o -E "\b(SQ|TCP|TCC|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u > $OUT/counter_names.txt
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --sweeps 2 --nofilter 1 --conc 8"
$PROBE --prof 1 > $OUT/phase_profile.log 2>&1
pass() { name=$1; shift; echo "== pmc $name: $*"; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $PROBE > $OUT/pmc_$name.log 2>&1 || tail -3 $OUT/pmc_$name.log; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass sq3 SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
pass fetch FETCH_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python $ROOT/scripts/summarize_prof.py $OUT > /dev/null
rm -rf $OUT/stats $OUT/pmc_*/
cat $OUT/phase_profile.log | tail -3
du -sh $OUT
