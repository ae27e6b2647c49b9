#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of bench.py (PatchMatch part) + PMC passes of the
# sweep kernel on the same launch shape (16 reference images per launch). Counter passes are separate
# runs with --kernel-trace only (no other trace domain), restricted to the sweep kernel.
# The kernel trace is of the primary leg only (--no-geom --no-ba): every sweep launch in it covers $BATCH images.
# Outputs under gpurun_out/prof_$TAG; copy the summaries into profiles/ afterwards.
TAG=${1:-r02}
BATCH=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace / stats of bench.py"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python $ROOT/bench.py --steps 1 --warmup 1 --batch $BATCH --no-cpu-baseline --no-ba --no-geom \
  > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
tail -c 600 $OUT/bench_under_rocprof.json
python $ROOT/scripts/summarize_prof.py $OUT > /dev/null 2>&1
rm -rf $OUT/stats
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc $BATCH"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
            "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs"
  timeout 120 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep" --pmc $ctrs \
    -d $OUT/pmc_p$i -o pmc -- $PROBE --sweeps 4 > $OUT/pmc_p$i.log 2>&1 || tail -3 $OUT/pmc_p$i.log
  python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
  find $OUT/pmc_p$i -type f -size +1M -delete
done
python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
rm -rf $OUT/pmc_p*/ $OUT/pmc_p*.log
cat $OUT/kernel_stats_summary.csv | cut -c1-160
du -sh $OUT
