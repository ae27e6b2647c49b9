"""smoke(): one small depth-map fusion on cuda:0 through fusion_run, checked bit for bit against the sequential
checker in oracle/ (test infrastructure; imported only from __graft_entry__.smoke())."""
import numpy as np


def run():
    import fusion_oracle
    from colmap_amd import fusion
    from pm_common import scene
    views = scene(4, 48, 36)
    images = []
    for v in views:
        rgb = np.stack([v.gray, 255 - v.gray, v.gray // 2], -1)
        images.append(fusion.FusionImage(48, 36, v.K, v.R, v.T, rgb, v.depth.copy(), v.normal.copy()))
    overlap = [[j for j in range(4) if j != i] for i in range(4)]
    opt = fusion.StereoFusionOptions(min_num_pixels=3)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    got = fusion.fuse(opt, images, overlap)
    assert len(got.xyz) == len(want.xyz) > 100
    assert np.array_equal(got.xyz, want.xyz) and np.array_equal(got.normal, want.normal) and np.array_equal(got.rgb, want.rgb)
    assert all(np.array_equal(a, b) for a, b in zip(got.visibility, want.visibility))
    print(f"smoke: fusion HIP == sequential oracle (bit-exact), {len(got.xyz)} points from 4 x 48x36 maps")
