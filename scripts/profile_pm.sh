#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of bench.py + PMC passes of a short probe.
# Outputs under gpurun_out/prof_$TAG (copy the summaries into profiles/ afterwards).
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
echo "== kernel trace / stats of bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
tail -1 $OUT/bench_under_rocprof.json
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --sweeps 2 --nofilter 1 --conc 8"
pass() { name=$1; shift; echo "== pmc $name: $*"; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $PROBE > $OUT/pmc_$name.log 2>&1 || tail -3 $OUT/pmc_$name.log; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
python $ROOT/scripts/summarize_prof.py $OUT
cat $OUT/kernel_stats_summary.csv
# keep only the small summaries (gpurun_out is capped at 64 MiB)
rm -rf $OUT/stats $OUT/pmc_*/ $OUT/counters_list.txt
du -sh $OUT
