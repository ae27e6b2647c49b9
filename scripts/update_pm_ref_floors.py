"""Copies the noise floors a slow run of tests/test_pm_ref.py measured (COLMAP_AMD_TEST_SLOW=1 on an MI355X:
the reference's own build against its -ffp-contract=fast build, gpurun_out/pm_ref_parity.json) into
tests/golden/pm_ref_floors.json, the committed floors the default run of that file takes its bars from.
TEST INFRASTRUCTURE."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pm_ref_parity.json")
d = json.load(open(src))
floors = {k: v for k, v in d.get("noise_floor", {}).items() if "source" not in v}
assert floors, "no measured floors in " + src
out = {"note": "agreement of oracle/_ref/libref_pm.so with libref_pm_fast.so (the reference's sources, -ffp-contract=fast) "
               "on the problems of tests/test_pm_ref.py, measured on an MI355X by COLMAP_AMD_TEST_SLOW=1; "
               "scripts/update_pm_ref_floors.py", "floors": floors}
path = os.path.join(ROOT, "tests", "golden", "pm_ref_floors.json")
old = {}
if os.path.exists(path):
    old = json.load(open(path)).get("floors", {})
old.update(floors)
out["floors"] = old
json.dump(out, open(path, "w"), indent=1)
print(path, sorted(old))
