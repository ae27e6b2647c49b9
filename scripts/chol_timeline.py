"""Timeline of ONE blocked Cholesky factorisation out of a rocprofv3 kernel trace (CSV with start / end timestamps):
which kernels sit on the critical chain, how long the chain idles between them, how much of the bulk trailing update
runs beside it. Usage (on the GPU box, see scripts/profile_ba_exact.sh):

    rocprofv3 --kernel-trace --output-format csv -d DIR -o ba -- python scripts/ba_probe.py ... --lst 3
    python scripts/chol_timeline.py DIR > gpurun_out/.../chol_timeline.txt
"""
import csv, glob, sys, collections

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
# the last complete factorisation: from the last form_pairs / form_kernel to the last solve_backward_panel_kernel after it
idx_form = max(i for i, r in enumerate(rows) if r[2] in ("form_pairs_kernel", "form_kernel"))
# (the very last formation may belong to a rejected step without a following factorisation: walk back until one has diag kernels after it)
while not any(r[2].startswith("chol_diag") or r[2] == "chol_outer_kernel" for r in rows[idx_form:]):
    idx_form = max(i for i, r in enumerate(rows[:idx_form]) if r[2] in ("form_pairs_kernel", "form_kernel"))
seg = [r for r in rows[idx_form:] if r[2].startswith("chol_") or r[2].startswith("solve_")]
t0 = seg[0][0]
chain_names = {"chol_diag_kernel", "chol_diag_mfma_kernel", "chol_outer_kernel", "chol_panel_kernel", "chol_strip_kernel"}
end = max(r[1] for r in seg if r[2].startswith("chol_"))
print(f"factorisation: {len(seg)} kernels, {(end - t0) / 1e3:.1f} us from the first diagonal block to the last update")
tot = collections.Counter()
for s, e, n, q, st in seg:
    tot[n] += e - s
for n, v in tot.most_common():
    print(f"  {n:32s} {v / 1e3:10.1f} us")
# per outer panel of 4 diagonal blocks: wall time, kernel time on the chain, idle time of the chain
diag = [r for r in seg if r[2].startswith("chol_diag") or r[2] == "chol_outer_kernel"]
step = 1 if any(r[2] == "chol_outer_kernel" for r in diag) else 4
print("panel  start_us  wall_us  diag_us  chain_busy_us  chain_idle_us  bulk_overlap_us")
bulk = [(s, e) for s, e, n, q, st in seg if n == "chol_update128_kernel"]
for p in range(0, len(diag), step):
    a = diag[p][0]
    b = diag[p + step][0] if p + step < len(diag) else end
    inside = [r for r in seg if a <= r[0] < b and r[2] != "chol_update128_kernel"]
    # union of the intervals of the non-bulk kernels
    iv = sorted((r[0], r[1]) for r in inside)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    dsum = sum(r[1] - r[0] for r in inside if r[2].startswith("chol_diag") or r[2] == "chol_outer_kernel")
    bo = sum(max(0, min(e, b) - max(s, a)) for s, e in bulk)
    print(f"{p // step:5d} {(a - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f} {dsum / 1e3:8.1f} {busy / 1e3:14.1f} {(b - a - busy) / 1e3:14.1f} {bo / 1e3:15.1f}")
# the first panels in detail
print("first 60 kernels: start_us dur_us name")
for s, e, n, q, st in seg[:60]:
    print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {n} q{q}")
