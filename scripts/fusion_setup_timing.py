"""Where fusion_run's set-up time goes at the bench's size (torch-free): 8 planar views of 2560x1920 with colour bitmaps,
development switch COLMAP_AMD_FUSION_TIMING (stderr marks of colmap_amd/csrc/fusion.hip). MEASUREMENT INFRASTRUCTURE."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
from colmap_amd import fusion
from colmap_amd._lib import lib
import fusion_gpu_check as F
n, w, h = 8, 2560, 1920
ims = F.plane_scene(n, w, h, 0.002)
rgb = np.zeros((h, w, 3), np.uint8)
for im in ims:
    im.rgb = rgb
loose = dict(min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05, max_normal_error=30.0)
lib().colmap_amd_set_switch(b"COLMAP_AMD_FUSION_TIMING", b"1")
for rep in range(2):
    t = time.time()
    pts = fusion.fuse(fusion.StereoFusionOptions(**loose), ims, [[j for j in range(n) if j != i] for i in range(n)])
    print(f"rep {rep}: {len(pts.xyz)} points, end to end {time.time() - t:.3f} s", flush=True)
