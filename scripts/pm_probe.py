"""Quick GPU timing probe for the PatchMatch path (not part of the test suite)."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("PM_PROBE_LIB"):   # a diagnostic build of the library (scripts/profile_pm_gather_diag.sh)
    from colmap_amd import build as _b
    _b.LIB_PATH = os.path.abspath(os.environ["PM_PROBE_LIB"])
from colmap_amd import mvs, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=640); ap.add_argument("--h", type=int, default=480)
ap.add_argument("--views", type=int, default=9); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--cols", type=int, default=0); ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--conc", type=int, default=1); ap.add_argument("--arc", type=float, default=30.0)
ap.add_argument("--sweeps", type=int, default=0); ap.add_argument("--prof", type=int, default=0); ap.add_argument("--batched", type=int, default=1); ap.add_argument("--nofilter", type=int, default=0)
ap.add_argument("--trace", type=int, default=0)  # progress trace of the last sweep launch: how far the waves drift apart
ap.add_argument("--geom", type=int, default=0)  # geometric consistency against the scene's ground-truth maps
a = ap.parse_args()
t = time.time()
views = syn.make_scene(a.views, a.w, a.h, arc_deg=a.arc, device="cuda")
print(f"render {time.time()-t:.2f}s")
ref = a.views // 2
src = [i for i in range(a.views) if i != ref]
dmin, dmax = syn.depth_range(views, ref)
opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=dmin, depth_max=dmax, sigma_spatial=5.0, geom_consistency=bool(a.geom),
                            filter=not a.nofilter, num_iterations=a.iters, columns_per_group=a.cols, threads_per_group=a.threads,
                            max_sweeps=a.sweeps)
images = [mvs.Image(v.K, v.R, v.T, torch.from_numpy(v.gray).cuda()) for v in views]
dm = [torch.from_numpy(v.depth).cuda() for v in views] if a.geom else None
nm = [torch.from_numpy(v.normal).cuda() for v in views] if a.geom else None
pms = [mvs.PatchMatch(opt, mvs.PatchMatch.Problem(ref, src, images, dm, nm)) for _ in range(a.conc)]
t = time.time()
for pm in pms: pm.Create()
torch.cuda.synchronize(); tc = time.time() - t
if a.prof:
    for pm in pms: pm.EnablePhaseProfile()
if a.trace:
    for pm in pms: pm.EnableProgressTrace()
t = time.time()
if a.batched:
    mvs.run_batch(pms)
else:
    for pm in pms: pm.RunAsync()
    for pm in pms: pm.Synchronize()
tr = time.time() - t
ms, n = pms[0].GetSweepTiming()
mpix = a.w * a.h * a.conc / 1e6
print(f"W={a.w} H={a.h} S={len(src)} conc={a.conc} C={a.cols} T={a.threads}: create {tc:.3f}s run {tr:.3f}s "
      f"-> {mpix/tr:.3f} Mpix/s (run only), {mpix/(tr+tc):.3f} incl create; sweep kernel avg {ms/max(n,1):.2f} ms x{n}")
print("per launch ms: " + " ".join(f"{v:.0f}" for v in pms[0].GetSweepTimes()))
if a.prof:
    pr = pms[0].GetPhaseProfile(); tot = sum(pr)
    names = ["setup+backward", "P0 tile", "P1 hyp+weights", "P2 priors", "P3 cdf+draws", "P4 ncc", "P5 argmin", "P6 ncc-winner", "P7-8 update", "-"]
    print("phase profile: " + ", ".join(f"{n} {100*v/max(tot,1):.1f}%" for n, v in zip(names, pr)))
if a.trace:
    # rows of the LAST sweep launch; stamps in 10 ns units. For every sample row: when the first / median / last wave
    # of the launch (all problems) got there, and the same for the first generation of waves (the column groups that
    # start together at launch: the first 4096 / conc column groups of every problem) -- the spread of a generation is
    # the band of rows in use at a time.
    tr = np.stack([pm.GetProgressTrace() for pm in pms]).astype(np.float64)   # (problem, group, sample)
    tr[tr == 0] = np.nan
    tr = tr[:, ~np.isnan(tr[0, :, 0]), :]       # the column groups the last launch had (its frame may be the narrow one)
    t0 = np.nanmin(tr)
    us = (tr - t0) / 100.0
    gen0 = max(1, min(us.shape[1], 4096 // len(pms)))
    print(f"progress trace: {us.shape[1]} column groups x {us.shape[2]} samples, first generation = groups 0..{gen0 - 1}")
    for k in range(us.shape[2]):
        col = us[:, :, k]
        if np.all(np.isnan(col)): continue
        g0 = us[:, :gen0, k]
        print(f"  row {128 * k:5d}: all groups {np.nanmin(col):9.0f} / {np.nanmedian(col):9.0f} / {np.nanmax(col):9.0f} us"
              f"   first generation {np.nanmin(g0):9.0f} / {np.nanmedian(g0):9.0f} / {np.nanmax(g0):9.0f} us")
    dt = np.diff(us[:, :gen0, :], axis=2)
    print(f"  first generation: {np.nanmedian(dt) / 128:.1f} us per row (median), fastest / slowest wave over the whole sweep "
          f"{np.nanmin(np.nansum(dt, axis=2)):.0f} / {np.nanmax(np.nansum(dt, axis=2)):.0f} us")
d = pms[0].GetDepthMap(); gt = views[ref].depth
ok = d > 0
rel = np.abs(d[ok] - gt[ok]) / gt[ok]
print(f"kept {ok.mean():.3f} median rel err {np.median(rel):.5f} frac<1% {(rel<0.01).mean():.3f}")
