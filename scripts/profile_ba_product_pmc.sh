#!/bin/bash
# Runs on the GPU box: PMC passes (separate runs, --kernel-trace only) of the three streaming kernels of the implicit
# Schur product at BA-1 -- wave / wait cycles, L2 hit rate and request latency, texture-address busy -- to say what
# bounds each of them. Output: gpurun_out/prof_ba_product_$TAG/ba_product_pmc.json
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ba_product_$TAG
mkdir -p $OUT
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
PROBE="python $ROOT/scripts/ba_probe.py --frames 1000 --points 200000 --track 10 --iters 10"
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
            "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "ba_obs_jx|ba_point_pass_tiled_kernel<0|ba_block_jtv_kernel<false" \
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
json.dump(out, open("$OUT/ba_product_pmc.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(v["avg_per_launch"]) for c, v in d.items()})
PY
rm -rf $OUT/pmc_p*/ $OUT/pmc_p*.log
