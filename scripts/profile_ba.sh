#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the BA-1 solve (1000 cameras x 200 k points,
# 10 LM iterations, scripts/ba_probe.py) + PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs,
# --kernel-trace only) of the kernels of the implicit Schur product. Outputs: gpurun_out/prof_ba_$TAG.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ba_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ba -- $PROBE > $OUT/probe.log 2>&1
tail -2 $OUT/probe.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220 > $OUT/ba_kernel_stats.csv
cat $OUT/ba_kernel_stats.csv | cut -c1-150
rm -rf $OUT/stats
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "ba_obs_jx|ba_point_pass|ba_block_jtv|ba_block_gram|ba_linearize" \
    --pmc $ctrs -d $OUT/pmc_p$i -o pmc -- $PROBE > $OUT/pmc_p$i.log 2>&1 || tail -3 $OUT/pmc_p$i.log
done
python - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/pmc_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        a = agg[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: {c: {"avg_per_launch": v[0] / max(v[1], 1), "launches": v[1]} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open("$OUT/ba_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $OUT/pmc_p*/ $OUT/pmc_p*.log
du -sh $OUT
