#!/bin/bash
# Runs on the GPU box: PMC passes over the memory side of the sweep kernel (address translation, L1 stall reasons,
# L2 -> fabric requests) on the launch shape of bench.py (16 reference images per launch, four sweeps). Separate
# rocprofv3 runs per counter group, --kernel-trace only. Summary -> gpurun_out/prof_$TAG/pmc_summary.json.
TAG=${1:-mem}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc 16 --sweeps 4"
i=0
# At most four counters of one hardware block per pass: a pass with six TCP counters made rocprofv3 abort and then
# sit until the timeout (round 3: 3 x 240 s of GPU budget lost) -- hence also the short timeout.
for ctrs in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_GUI_ACTIVE" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum" \
            "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  echo "== pmc pass $i: $ctrs"
  timeout 90 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "pm_sweep" --pmc $ctrs \
    -d $OUT/pmc_p$i -o pmc -- $PROBE > $OUT/pmc_p$i.log 2>&1 || tail -3 $OUT/pmc_p$i.log
  grep -E "sweep kernel" $OUT/pmc_p$i.log
  python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
  find $OUT/pmc_p$i -type f -size +1M -delete
done
python $ROOT/scripts/summarize_prof.py $OUT --per-dispatch > /dev/null 2>&1
rm -rf $OUT/pmc_p*/ $OUT/pmc_p*.log
ls $OUT; cat $OUT/pmc_summary.json 2>/dev/null | head -80
