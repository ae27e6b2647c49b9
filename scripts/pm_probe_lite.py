"""PatchMatch timing probe that costs a GPU box seconds instead of minutes: no torch on the box.

A fresh box pays 1-2 minutes for `import torch` (and bench.py renders its views with it); this probe renders the views
HERE, once per session, and ships them with the snapshot:

    python scripts/pm_probe_lite.py make [--w 2560 --h 1920 --batch 16 --S 20]   # here (CPU torch, ~minutes): scripts/tmp/pm_probe_views.npz
    gpurun -- 'python scripts/pm_probe_lite.py run [--reps 2 --cols 0 --threads 0 --sweeps 0]'   # on the box: ctypes + numpy only

`run` is bench.py's step -- `batch` reference images of the camera ring, S = 20 sources each (S/2 neighbours either side),
photometric + filter, packed sources shared through the image cache, one launch per sweep for the whole batch -- and prints
the per-launch sweep times, Mpix/s, NCC evaluations per pixel and the kernel that ran. PM_PROBE_LIB=<path> times another
build of the library. MEASUREMENT INFRASTRUCTURE."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NPZ = os.path.join(ROOT, "scripts", "tmp", "pm_probe_views.npz")


def make(a):
    from colmap_amd import synthetic as syn
    half = a.S // 2
    n = a.batch + 2 * half
    step_deg = 360.0 / a.ring
    cams = syn.ring_cameras(n, a.w, a.h, 2400.0 * a.w / 2560.0, arc_deg=step_deg * (n - 1), start_deg=-half * step_deg)
    gray = np.empty((n, a.h, a.w), np.uint8)
    K, R, T, rng = [], [], [], []
    t = time.time()
    for i, (k, r, tt) in enumerate(cams):
        g, d, _ = syn.render_view(k, r, tt, a.w, a.h, seed=0, device="cpu")
        gray[i] = g.numpy()
        K.append(k); R.append(r); T.append(tt)
        rng.append((float(d.min()), float(d.max())))
        print(f"view {i + 1}/{n} rendered ({time.time() - t:.0f}s)", flush=True)
    os.makedirs(os.path.dirname(NPZ), exist_ok=True)
    np.savez_compressed(NPZ, gray=gray, K=np.stack(K), R=np.stack(R), T=np.stack(T), range=np.array(rng), S=a.S, batch=a.batch)
    print(NPZ, os.path.getsize(NPZ) / 1e6, "MB")


def run(a):
    if os.environ.get("PM_PROBE_LIB"):
        from colmap_amd import build as _b
        _b.LIB_PATH = os.path.abspath(os.environ["PM_PROBE_LIB"])
    from colmap_amd import mvs
    for kv in a.switch:   # development switches of the library (csrc/switches.h), e.g. COLMAP_AMD_PM_WAVE=0
        k, v = kv.split("=", 1)
        mvs.lib().colmap_amd_set_switch(k.encode(), v.encode())
    z = np.load(NPZ)
    S, batch = int(z["S"]), int(z["batch"]) if a.batch <= 0 else a.batch
    half = S // 2
    gray = z["gray"]
    h, w = gray.shape[1:]
    images = [mvs.Image(z["K"][i], z["R"][i], z["T"][i], np.ascontiguousarray(gray[i])) for i in range(len(gray))]
    cache = mvs.ImageCache(0)

    def problems():
        pms = []
        for j in range(batch):
            ref = half + j
            src = [ref + o for o in range(-half, half + 1) if o != 0][:S]
            dmin, dmax = z["range"][ref][0] * 0.9, z["range"][ref][1] * 1.1
            opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=float(dmin), depth_max=float(dmax), sigma_spatial=5.0,
                                        geom_consistency=False, filter=True, columns_per_group=a.cols,
                                        threads_per_group=a.threads, max_sweeps=a.sweeps)
            pms.append(mvs.PatchMatch(opt, mvs.PatchMatch.Problem(ref, src, images), cache))
        return pms

    for rep in range(a.reps):
        pms = problems()
        t = time.time()
        for pm in pms:
            pm.Create()
        tc = time.time() - t
        if a.profile:
            for pm in pms:
                pm.EnablePhaseProfile()
        t = time.time()
        if a.split > 1:
            # sub-batches on their own streams (each run_batch enqueues on its first handle's stream), issued together:
            # the tail of one sub-batch's sweep launch overlaps with the other sub-batches' launches
            import threading
            per = (len(pms) + a.split - 1) // a.split
            th = [threading.Thread(target=mvs.run_batch, args=(pms[i:i + per],), kwargs={"wait": True})
                  for i in range(0, len(pms), per)]
            for x in th:
                x.start()
            for x in th:
                x.join()
        else:
            mvs.run_batch(pms, wait=True)
        tr = time.time() - t
        ms, n = pms[0].GetSweepTiming()
        ev = [pm.GetEvaluationCount() for pm in pms]
        mpix = batch * w * h / 1e6
        ipl, conc = pms[0].GetLaunchShape()
        print(f"rep {rep}: {w}x{h} S={S} batch={batch} ({ipl} images per launch, {conc} launches in flight) kernel {pms[0].GetSweepKernelName()}: create {tc:.2f}s run {tr:.3f}s -> "
              f"{mpix / tr:.3f} Mpix/s (run), {mpix / (tr + tc):.3f} incl. create; sweep launch avg {ms / max(n, 1):.2f} ms x {n}; "
              f"NCC evaluations per pixel per sweep {sum(e[0] for e in ev) / (batch * w * h * max(n, 1)):.2f}", flush=True)
        print("  per launch ms: " + " ".join(f"{v:.0f}" for v in pms[0].GetSweepTimes()), flush=True)
        if a.profile:
            import json
            tot = {}
            for pm in pms:
                for k, v in pm.GetPhaseProfileSlots().items():
                    tot[k] = tot.get(k, 0) + v
            waves = tot.pop("waves")
            cyc = sum(tot.values())
            out = {"kernel": pms[0].GetSweepKernelName(), "shape": f"{batch} x {w}x{h} S={S}", "sweep_launches": n,
                   "sweep_launch_ms_avg": ms / max(n, 1), "waves_reported": waves, "cycles_total": cyc,
                   "cycles_per_wave_row": cyc / max(waves, 1) / (h if True else w),
                   "share": {k: round(v / cyc, 4) for k, v in tot.items()}, "cycles": tot}
            print("PHASE_PROFILE " + json.dumps(out), flush=True)
            if a.profile_out:
                os.makedirs(os.path.dirname(a.profile_out), exist_ok=True)
                with open(a.profile_out, "w") as f:
                    json.dump(out, f, indent=1)
        if rep == a.reps - 1:
            d = pms[0].GetDepthMap()
            print(f"  kept by the filter: {(d > 0).mean():.3f}")
        for pm in pms:
            pm.close()
    cache.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["make", "run"])
    ap.add_argument("--w", type=int, default=2560); ap.add_argument("--h", type=int, default=1920)
    ap.add_argument("--batch", type=int, default=-1); ap.add_argument("--S", type=int, default=20)
    ap.add_argument("--ring", type=int, default=100); ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--cols", type=int, default=0); ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--sweeps", type=int, default=0)
    ap.add_argument("--switch", action="append", default=[], help="development switch NAME=VALUE (repeatable)")
    ap.add_argument("--split", type=int, default=1, help="run the batch as N concurrent sub-batches (one stream each)")
    ap.add_argument("--profile", action="store_true", help="launch the phase-profiling build of the sweep kernel")
    ap.add_argument("--profile-out", default="")
    a = ap.parse_args()
    if a.mode == "make":
        if a.batch <= 0:
            a.batch = 16
        make(a)
    else:
        run(a)
