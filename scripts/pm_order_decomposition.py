"""Where the HIP kernel's ComputeInitialCost differs from the reference's evaluation order, ingredient by ingredient.

The HIP sweep kernel equals the oracle in "device order" (order = 1) bit for bit (tests/test_pm_gpu.py), and the
oracle in the reference's order (order = 0) is what tests/test_pm_ref.py compares with the reference build. Both run
on the CPU, so the difference between the two orders can be taken apart here without a GPU: PMO_DEVICE_MIX (see
oracle/pm_oracle.c: ncc_cost_device_mixed) reverts one ingredient of the device order at a time to the reference's
form. Output: profiles/r04_pm_initial_cost_decomposition.json. MEASUREMENT INFRASTRUCTURE (calls the checker)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIXES = {0: "device order (= the HIP kernel)", 1: "sums tap by tap (no 16-lane dealing / tree)",
         2: "running-sum coordinates", 4: "one division per tap", 8: "reference bilinear sample",
         14: "everything but the summation order reverted", 13: "everything but the coordinates reverted",
         11: "everything but the shared division reverted", 7: "everything but the bilinear form reverted",
         15: "all four reverted (= order 0)"}

WORKER = r"""
import sys, os, json
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'oracle'), os.path.join(%(root)r, 'tests')]
import numpy as np
import pm_oracle
from colmap_amd import synthetic as syn
from pm_common import scene, oracle_inputs
out = {}
for shape in ("96x72_S4", "96x72_S20"):
    if shape == "96x72_S4":
        views, r, src = scene(), 2, [0, 1, 3, 4]
    else:
        views = scene(22, 96, 72, 3.6 * 21)
        r, src = 10, [i for i in range(21) if i != 10]
    imgs = oracle_inputs(views)
    dmin, dmax = syn.depth_range(views, r)
    o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=0, num_iterations=5)
    o.max_sweeps = 0
    o.order = int(sys.argv[1])
    out[shape] = pm_oracle.run(o, imgs, r, src, want_cost=True)["cost"].astype(np.float64).tolist()
json.dump(out, sys.stdout)
"""


def costs(order, mix):
    env = dict(os.environ, PMO_DEVICE_MIX=str(mix))
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, str(order)], env=env, capture_output=True, text=True,
                       check=True)
    return json.loads(r.stdout)


def main():
    import numpy as np
    base = costs(0, 0)
    rows = {}
    for mix, what in MIXES.items():
        c = costs(1, mix)
        rows[str(mix)] = {"what": what}
        for shape in base:
            d = np.abs(np.array(c[shape]) - np.array(base[shape]))
            rows[str(mix)][shape] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "p999": float(np.quantile(d, 0.999)),
                                     "pixels_different": float((d > 0).mean())}
        print(mix, what, {k: (f"{v['max_abs']:.3g}", f"{v['mean_abs']:.3g}") for k, v in rows[str(mix)].items() if k != "what"},
              flush=True)
    # both float orders against the same costs with every intermediate in double (the common yardstick)
    exact = costs(1, 16)
    dev = costs(1, 0)
    yard = {}
    for shape in base:
        e = np.array(exact[shape])
        sel = (e > 0.0) & (e < 2.0)   # (clamped costs carry no rounding information)
        d0, d1 = np.abs(np.array(base[shape]) - e)[sel], np.abs(np.array(dev[shape]) - e)[sel]
        yard[shape] = {"reference_order_vs_double": {"max_abs": float(d0.max()), "mean_abs": float(d0.mean()), "p999": float(np.quantile(d0, 0.999))},
                       "device_order_vs_double": {"max_abs": float(d1.max()), "mean_abs": float(d1.mean()), "p999": float(np.quantile(d1, 0.999))}}
        print(shape, yard[shape], flush=True)
    out = {"against_double": yard, "what": "ComputeInitialCost: |oracle order 1 with PMO_DEVICE_MIX = bits  -  oracle order 0| per pixel and source "
                   "(the HIP kernel == order 1, mix 0, bit for bit); bits: 1 summation order, 2 coordinates, 4 division, 8 "
                   "bilinear form", "rows": rows}
    path = os.path.join(ROOT, "profiles", "r04_pm_initial_cost_decomposition.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path)


if __name__ == "__main__":
    main()
