"""The two long problems of tests/test_pm_ref.py -- BASELINE config[0] and the benchmark crop -- as named cases, and the
machinery that solves them through the reference's own build IN BACKGROUND PROCESSES while the rest of the GPU suite runs.

Why: the reference's kernel (one thread per image column, 32-thread blocks) occupies a few dozen CUs for minutes; run
inside the test that needs it, it is minutes of a suite in which the GPU idles. tests/conftest.py starts one worker
process per (case, build) as soon as the collection shows that the tests are selected and orders those tests last;
the tests wait for the workers' .npz files. Run alone (or with the workers failing to start) a test solves in-process.

TEST INFRASTRUCTURE ONLY. `python tests/ref_pm_cases.py <case> <out.npz> [fast] [oracle0]` is the worker."""
from __future__ import annotations

import os
import subprocess
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_PROCS = {}   # (case, fast) -> (Popen, npz path, log path)


def build_case(name):
    """(views, ref index, source indices, depth range or None, ground-truth depth of the reference view, option keywords)"""
    from colmap_amd import synthetic as syn
    if name == "config0_photometric":
        views = syn.make_scene(3, 640, 480, focal=600.0, arc_deg=8.0)
        return views, 1, [0, 2], None, views[1].depth, dict(geom_consistency=0, filter=1)
    if name == "bench_crop_384x288":
        from pm_common import bench_crop_problem
        mixed, r, src, crop, rng = bench_crop_problem(cw=384, ch=288)
        return mixed, r, src, rng, crop.depth, dict(geom_consistency=0, filter=1)
    raise KeyError(name)


def solve_reference(name, fast=False, oracle0=False):
    """The case through oracle/_ref (or its -ffp-contract=fast build); with oracle0 also through the oracle in the
    reference's order (host cores)."""
    import pm_oracle
    import ref_pm
    from colmap_amd import synthetic as syn
    from pm_common import oracle_inputs
    pm_oracle.build()
    views, r, src, rng, _, kw = build_case(name)
    imgs = oracle_inputs(views)
    dmin, dmax = rng if rng else syn.depth_range(views, r)
    o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, **kw)
    ref = ref_pm.RefPatchMatch(o, imgs, r, src, fast=fast)
    out = ref.run()
    ref.close()
    res = {"depth": out["depth"], "normal": out["normal"], "mask": out["mask"]}
    if oracle0:
        o.order = 0
        w = pm_oracle.run(o, imgs, r, src)
        res.update(o0_depth=w["depth"], o0_normal=w["normal"], o0_mask=w["mask"])
    return res


def start(name, fast=False, oracle0=False):
    """Launch the worker of (name, fast) unless it is running already."""
    key = (name, bool(fast))
    if key in _PROCS:
        return
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", _ROOT), "gpurun_out", "ref_pm_workers")
    os.makedirs(out_dir, exist_ok=True)
    tag = f"{name}{'_fast' if fast else ''}_{os.getpid()}"
    import tempfile
    npz, log = os.path.join(tempfile.gettempdir(), "ref_pm_" + tag + ".npz"), os.path.join(out_dir, tag + ".log")
    cmd = [sys.executable, os.path.abspath(__file__), name, npz] + (["fast"] if fast else []) + (["oracle0"] if oracle0 else [])
    _PROCS[key] = (subprocess.Popen(cmd, stdout=open(log, "w"), stderr=subprocess.STDOUT, cwd=_ROOT), npz, log, time.time())


def result(name, fast=False, oracle0=False, timeout=1500.0):
    """The worker's arrays (waits for it); solves in-process when no worker was started or it failed."""
    key = (name, bool(fast))
    if key in _PROCS:
        proc, npz, log, t0 = _PROCS[key]
        try:
            proc.wait(timeout=max(1.0, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            proc.kill()
        if proc.returncode == 0 and os.path.exists(npz):
            z = np.load(npz)
            out = {k: z[k] for k in z.files}
            if not oracle0 or "o0_depth" in out:
                return out
        sys.stderr.write(f"ref_pm_cases: worker of {key} did not deliver (rc {proc.returncode}, log {log}); solving in-process\n")
    return solve_reference(name, fast=fast, oracle0=oracle0)


if __name__ == "__main__":
    sys.path.insert(0, _ROOT)
    sys.path.insert(0, os.path.join(_ROOT, "oracle"))
    sys.path.insert(0, _HERE)
    t = time.time()
    r = solve_reference(sys.argv[1], fast="fast" in sys.argv[3:], oracle0="oracle0" in sys.argv[3:])
    np.savez(sys.argv[2], **r)
    print(f"{sys.argv[1:]} solved in {time.time() - t:.1f} s")
