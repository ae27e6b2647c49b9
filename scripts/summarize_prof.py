"""Summarise rocprofv3 CSV output into small files (run on the GPU box)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
per_dispatch = "--per-dispatch" in sys.argv
summary = {}
dispatch_rows = {}
# kernel stats
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "pm_" in r.get("Name", "")] + rows[:5]
    with open(os.path.join(out, "kernel_stats_summary.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys())
        w.writeheader()
        seen = set()
        for r in keep:
            if r["Name"] in seen:
                continue
            seen.add(r["Name"])
            w.writerow(r)
# PMC passes
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    agg = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "pm_sweep" in name:
                k = "pm_sweep_kernel"
            elif "pm_initial_cost" in name:
                k = "pm_initial_cost_kernel"
            else:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[k].add(r.get("Dispatch_Id"))
            if per_dispatch and k == "pm_sweep_kernel":
                row = dispatch_rows.setdefault(os.path.basename(d), {}).setdefault(int(r.get("Dispatch_Id", 0)), {})
                row[r["Counter_Name"]] = row.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    summary[os.path.basename(d)] = {k: dict(v, dispatches=len(ndisp[k])) for k, v in agg.items()}
prev_path = os.path.join(out, "pmc_summary.json")
if os.path.exists(prev_path):  # passes summarised earlier whose bulk output is already deleted
    prev = json.load(open(prev_path))
    for k, v in prev.items():
        if k not in summary or not summary[k]:
            summary[k] = v
if per_dispatch:
    pd_path = os.path.join(out, "pmc_per_dispatch.json")
    prev_pd = json.load(open(pd_path)) if os.path.exists(pd_path) else {}
    for k, v in dispatch_rows.items():
        prev_pd[k] = [dict(v[i], dispatch=i) for i in sorted(v)]
    json.dump(prev_pd, open(pd_path, "w"), indent=1)
json.dump(summary, open(prev_path, "w"), indent=1)
print(json.dumps(summary, indent=1))
