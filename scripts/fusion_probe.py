"""Times fusion_run (HIP) on a synthetic workspace and prints the round statistics (runs on the GPU box)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from colmap_amd import fusion
from colmap_amd._lib import lib
from pm_common import scene


def run(n, w, h, **kw):
    views = scene(n, w, h)
    images = []
    for v in views:
        rgb = np.stack([v.gray, v.gray, v.gray], -1)
        images.append(fusion.FusionImage(w, h, v.K, v.R, v.T, rgb, v.depth.copy(), v.normal.copy()))
    overlap = [[j for j in range(n) if j != i] for i in range(n)]
    opt = fusion.StereoFusionOptions(**kw)
    fusion.fuse(opt, images[:2], [[1], [0]])  # warm-up (module load)
    t = time.time()
    pts = fusion.fuse(opt, images, overlap)
    dt = time.time() - t
    st = [C.c_int64() for _ in range(4)]
    lib().fusion_last_stats(*[C.byref(x) for x in st])
    images_, seeds, rounds, walks = [x.value for x in st]
    print(json.dumps(dict(images=n, width=w, height=h, points=len(pts.xyz), seconds=round(dt, 3),
                          mpix_per_s=round(seeds / dt / 1e6, 3), rounds=rounds, rounds_per_image=round(rounds / images_, 2),
                          walks_per_seed=round(walks / seeds, 3))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # fusion_probe.py N W H
        run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        run(5, 320, 240)
        run(8, 640, 480)
        run(8, 1280, 960)
