"""One short GPU call for the fusion kernels (no torch import: a fresh box pays minutes for it).

    python scripts/fusion_gpu_check.py make     # here: writes scripts/tmp/fusion_cases.pkl (inputs of the GPU tests)
    python scripts/fusion_gpu_check.py run      # on the GPU box: fusion_run against oracle mode 1, stage by stage

Stages run in order of cost and every line is flushed to gpurun_out/fusion_gpu_check.log, so a call that is cut off
still says how far it got. TEST / MEASUREMENT INFRASTRUCTURE (it calls the checker in oracle/)."""
import ctypes as C
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
PKL = os.path.join(ROOT, "scripts", "tmp", "fusion_cases.pkl")


def plane_scene(n, w, h, sigma, seed=0, holes=0.03):
    """n cameras side by side (R = I) in front of the plane z = 0 at distance 10: constant depth, numpy only."""
    from colmap_amd import fusion
    rng = np.random.default_rng(seed)
    f = 1.2 * w
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    ims = []
    for i in range(n):
        T = np.array([-(i - (n - 1) / 2) * 0.25, 0.0, 10.0], np.float32)
        d = (10.0 * (1.0 + sigma * rng.standard_normal((h, w)))).astype(np.float32)
        d[rng.random((h, w)) < holes] = 0.0
        nm = np.zeros((3, h, w), np.float32)
        nm[2] = -1.0
        ims.append(fusion.FusionImage(w, h, K, np.eye(3, dtype=np.float32), T, None, d, nm))
    return ims


def make():
    import test_fusion as TF
    from pm_common import scene
    cases = []
    for name in sorted(TF._CASES):
        opt, images, overlap = TF._case(name)
        cases.append((name, opt, images, overlap))
    rng = np.random.default_rng(1)
    im = TF._images(scene(4, 24, 160))
    for a in im:
        a.depth_map = (a.depth_map * (1 + 0.01 * rng.standard_normal(a.depth_map.shape))).astype(np.float32)
    from colmap_amd import fusion
    loose = dict(min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05, max_normal_error=30.0)
    cases.append(("noisy_tall_4x24x160", fusion.StereoFusionOptions(**loose), im, TF._overlap(4)))
    cases.append(("many_seeds_3x320x240", fusion.StereoFusionOptions(min_num_pixels=2), TF._images(scene(3, 320, 240)),
                  TF._overlap(3)))
    os.makedirs(os.path.dirname(PKL), exist_ok=True)
    with open(PKL, "wb") as fh:
        pickle.dump(cases, fh)
    print(PKL, os.path.getsize(PKL) / 1e6, "MB")


def run():
    import fusion_oracle
    from colmap_amd import fusion
    from colmap_amd._lib import lib
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "fusion_gpu_check.log"), "w")

    def say(*a):
        s = " ".join(str(x) for x in a)
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()
        os.fsync(log.fileno())

    def same(a, b):
        return (len(a.xyz) == len(b.xyz) and np.array_equal(a.xyz, b.xyz) and np.array_equal(a.normal, b.normal) and
                np.array_equal(a.rgb, b.rgb) and all(np.array_equal(x, y) for x, y in zip(a.visibility, b.visibility)))

    def one(name, opt, images, overlap, repeat=1):
        t0 = time.time()
        want = fusion_oracle.fuse(opt, images, overlap, mode=1)
        t1 = time.time()
        for _ in range(repeat):
            got = fusion.fuse(opt, images, overlap)
        t2 = time.time()
        st = [C.c_int64() for _ in range(4)]
        lib().fusion_last_stats(*[C.byref(x) for x in st])
        up, dev = C.c_double(), C.c_double()
        lib().fusion_last_timing(C.byref(up), C.byref(dev))
        mpix = sum(im.depth_map.size for im in images if im.used and im.depth_map is not None) / 1e6
        say(f"{name}: equal={same(got, want)} points={len(got.xyz)}/{len(want.xyz)} passes={st[2].value} walks={st[3].value} "
            f"oracle={t1 - t0:.2f}s hip_wall={(t2 - t1) / repeat:.3f}s device={dev.value:.4f}s upload={up.value:.4f}s "
            f"Mpix/s(device)={mpix / max(dev.value, 1e-9):.2f}")

    t = time.time()
    lib()
    say(f"library loaded in {time.time() - t:.1f}s")
    with open(PKL, "rb") as fh:
        cases = pickle.load(fh)
    for name, opt, images, overlap in cases:
        try:
            one(name, opt, images, overlap)
        except Exception as e:  # keep going: later stages still tell something
            say(f"{name}: FAILED {type(e).__name__}: {e}")
    loose = dict(min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05, max_normal_error=30.0)
    sizes = [(4, 640, 480, 0.002), (4, 1280, 960, 0.002), (8, 1280, 960, 0.002), (4, 2560, 1920, 0.002)]
    if "--quick" in sys.argv:  # a call with seconds of budget: the two sizes whose checker answers within a second first
        sizes = [(4, 640, 480, 0.002), (4, 1280, 960, 0.002), (4, 2560, 1920, 0.002)]
    for (n, w, h, sigma) in sizes:
        try:
            ims = plane_scene(n, w, h, sigma)
            one(f"plane_{n}x{w}x{h}_sigma{sigma}", fusion.StereoFusionOptions(**loose), ims,
                [[j for j in range(n) if j != i] for i in range(n)])
        except Exception as e:
            say(f"plane {n}x{w}x{h}: FAILED {type(e).__name__}: {e}")
    say("done")


if __name__ == "__main__":
    make() if sys.argv[1:2] == ["make"] else run()
