#!/bin/bash
# The profiling recipe of scripts/profile_pm.sh on the torch-free probe: kernel trace + the five PMC passes of the sweep
# kernel on bench.py's launch shape (16 reference images of 2560x1920 per launch, S = 20) WITHOUT importing torch on the
# box (each of the six runs of profile_pm.sh pays 1-2 minutes for it on a fresh box; here a run is ~15 s).
# Needs scripts/tmp/pm_probe_views.npz (python scripts/pm_probe_lite.py make, in the container, once per session).
# Counter passes are separate runs with --kernel-trace only (no other trace domain), restricted to the sweep kernel.
# Outputs under gpurun_out/prof_$TAG; copy the summaries into profiles/ afterwards.
TAG=${1:-lite}
SWEEPS=${2:-4}   # sweep launches per counter pass (the kernel trace runs the full 20)
BATCH=${3:-16}   # reference images per step (pm_run_batch runs 16 as two concurrent launches of 8)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe_lite.py run --reps 1 --batch $BATCH"
echo "== kernel trace / stats"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o probe -- $PROBE \
  > $OUT/probe_under_rocprof.log 2> $OUT/probe_under_rocprof.err
tail -4 $OUT/probe_under_rocprof.log
python $ROOT/scripts/summarize_prof.py $OUT > /dev/null 2>&1
rm -rf $OUT/stats
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
            "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs"
  timeout 90 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep" --pmc $ctrs \
    -d $OUT/pmc_p$i -o pmc -- $PROBE --sweeps $SWEEPS > $OUT/pmc_p$i.log 2>&1 || tail -3 $OUT/pmc_p$i.log
  python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
  find $OUT/pmc_p$i -type f -size +1M -delete
done
python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
rm -rf $OUT/pmc_p*/ $OUT/pmc_p*.log
cat $OUT/kernel_stats_summary.csv 2>/dev/null | cut -c1-160
du -sh $OUT
