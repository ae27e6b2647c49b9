"""profiles/ba_schur_traffic.json from a PMC summary of scripts/profile_ba.sh (MEASUREMENT INFRASTRUCTURE).

    python scripts/update_ba_traffic.py gpurun_out/prof_ba_r06/ba_pmc_summary.json r06

FETCH_SIZE / WRITE_SIZE are reported in KiB per launch; FETCH_SIZE is doubled for gfx950 (MI355X_MICROARCH.md: the counter
tallies 128-byte read requests at 64 bytes), WRITE_SIZE taken as reported. One implicit Schur product = one launch each of
ba_obs_jx_kernel, ba_point_pass_tiled_kernel<0, ..> and ba_block_jtv_kernel<false, ..>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
d = json.load(open(src))
pick = {}
for name, c in d.items():
    if name.startswith("ba_obs_jx_kernel") or name.startswith("ba_point_pass_tiled_kernel<0") or name.startswith("ba_block_jtv_kernel<false"):
        pick[name] = {k: v["avg_per_launch"] for k, v in c.items()}
assert len(pick) == 3, list(d)
fetch_raw = sum(v.get("FETCH_SIZE", 0.0) for v in pick.values()) * 1024.0
write = sum(v.get("WRITE_SIZE", 0.0) for v in pick.values()) * 1024.0
out = {"observations": 2000000, "fetch_bytes_per_product": 2.0 * fetch_raw, "write_bytes_per_product": write,
       "fetch_bytes_raw_counter": fetch_raw, "per_kernel_KiB": pick,
       "source": f"profiles/{tag}_ba_pmc_summary.json (scripts/profile_ba.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate "
                 "passes, BA-1 probe 1000 frames x 200 k points x track 10, averages over the launches of 10 LM iterations; KiB -> bytes). "
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported."}
json.dump(out, open(os.path.join(ROOT, "profiles", "ba_schur_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fetch_bytes_per_product", "write_bytes_per_product")}))
