#!/bin/bash
# One GPU-box session = a list of stages, run in order (replaces the per-session scripts of round 2):
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_session.sh TAG stage[:arg] ...'
#
# stages:  tests:<pytest -k expression or file>   pytest -m gpu on that selection (log -> gpurun_out/TAG/)
#          alltests                                the whole `pytest tests -m gpu`
#          smoke                                   __graft_entry__.smoke()
#          bench[:extra bench.py flags]            python bench.py ... (JSON line -> gpurun_out/TAG/bench.json)
#          ba[:iters]                              scripts/ba_probe.py at BA-1, three repetitions
#          pm[:extra pm_probe flags]               scripts/pm_probe.py 2560x1920, 16 concurrent, 4 sweeps
#          prof-pm / prof-ba                       scripts/profile_{pm,ba}.sh TAG (rocprofv3 stats + PMC passes)
#          fusion                                  scripts/fusion_probe.py
#          sh:<command>                            anything else
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
ulimit -c 0   # a GPU memory fault must not fill /tmp with a core dump
cd $ROOT
export TMPDIR=/tmp
quiet() { grep -v "rccl\|HIP version\|ROCm version\|Hostname\|^$"; }
for st in "$@"; do
  name=${st%%:*}; arg=""; [[ "$st" == *:* ]] && arg=${st#*:}
  echo "=== stage $name $arg"
  t0=$(date +%s)
  case $name in
    tests)    if [[ "$arg" == *.py* ]]; then sel="$arg"; else sel="tests -k \"$arg\""; fi
              eval timeout 1500 python -m pytest $sel -m gpu -q --timeout 900 2>&1 | quiet | tail -25 | tee -a $OUT/tests.log ;;
    alltests) timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=40 2>&1 | quiet | tail -80 | tee $OUT/alltests.log ;;
    smoke)    timeout 600 python __graft_entry__.py smoke 2>&1 | quiet | tail -8 | tee $OUT/smoke.log ;;
    bench)    timeout 1200 python bench.py $arg > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    ba)       for i in 1 2 3; do timeout 300 python scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters ${arg:-10} 2>&1 | tail -1; done | tee -a $OUT/ba_probe.log ;;
    pm)       timeout 600 python scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc 16 --sweeps 4 $arg 2>&1 | grep -E "sweep kernel|phase|rror" | tee -a $OUT/pm_probe.log ;;
    prof-pm)  bash scripts/profile_pm.sh $TAG 2>&1 | tail -40 ;;
    prof-ba)  bash scripts/profile_ba.sh $TAG 2>&1 | tail -60 ;;
    fusion)   timeout 600 python scripts/fusion_probe.py $arg 2>&1 | quiet | tail -20 | tee -a $OUT/fusion_probe.log ;;
    sh)       eval "$arg" 2>&1 | tail -40 ;;
    *)        echo "unknown stage $name" ;;
  esac
  echo "=== stage $name took $(( $(date +%s) - t0 )) s"
done
