"""Shared helpers: one seeded scene -> inputs for both the oracle and the HIP path."""
import functools

import numpy as np

from colmap_amd import synthetic as syn


@functools.lru_cache(maxsize=8)
def scene(num_views=5, width=96, height=72, arc_deg=24.0, seed=0):
    return syn.make_scene(num_views, width, height, arc_deg=arc_deg, seed=seed)


def oracle_inputs(views, with_maps=False, maps=None):
    imgs = syn.as_image_dicts(views, with_maps=False)
    if with_maps:
        for d, m in zip(imgs, maps):
            d["depth"], d["normal"] = m
    return imgs


def hip_problem(views, ref_idx, src_idxs, maps=None):
    from colmap_amd import mvs
    images = [mvs.Image(v.K, v.R, v.T, v.gray) for v in views]
    prob = mvs.PatchMatch.Problem(ref_image_idx=ref_idx, src_image_idxs=list(src_idxs), images=images)
    if maps is not None:
        prob.depth_maps = [m[0] for m in maps]
        prob.normal_maps = [m[1] for m in maps]
    return prob


OPTION_FIELDS = ["depth_min", "depth_max", "sigma_spatial", "sigma_color", "ncc_sigma",
                 "min_triangulation_angle", "incident_angle_sigma", "geom_consistency_regularizer",
                 "geom_consistency_max_cost", "filter_min_ncc", "filter_min_triangulation_angle",
                 "filter_geom_consistency_max_cost", "window_radius", "window_step", "num_samples",
                 "num_iterations", "filter_min_num_consistent", "geom_consistency", "filter"]


def paired_options(pm_oracle, **kw):
    """Same option values for the oracle struct and the host-mirror dataclass."""
    from colmap_amd import mvs
    max_sweeps = kw.pop("max_sweeps", -1)
    tuning = {k: kw.pop(k) for k in ("columns_per_group", "threads_per_group") if k in kw}
    o = pm_oracle.default_options(**kw)
    o.max_sweeps = max_sweeps
    o.order = 1  # the HIP kernel's evaluation order (oracle/pm_oracle.c: ncc_cost_device)
    h = mvs.PatchMatchOptions(gpu_index="0", **tuning)
    for f in OPTION_FIELDS:
        v = getattr(o, f)
        if f in ("geom_consistency", "filter"):
            v = bool(v)
        setattr(h, f, v)
    h.max_sweeps = 0 if max_sweeps < 0 else (-1 if max_sweeps == 0 else max_sweeps)
    return o, h


def write_dense_workspace(path, views, num_points=400, seed=0, cfg_spec="__auto__, 20"):
    """An undistorted COLMAP dense workspace on disk from rendered views: images/*.png, a binary
    sparse model whose points are back-projected ground-truth pixels with visibility-checked
    tracks, stereo/patch-match.cfg. Returns the image names."""
    import os
    from PIL import Image as PILImage
    from colmap_amd import workspace as W
    rng = np.random.default_rng(seed)
    h, w = views[0].gray.shape
    sm = W.SparseModel()
    names = [f"view{i:03d}.png" for i in range(len(views))]
    os.makedirs(os.path.join(path, "images"), exist_ok=True)
    os.makedirs(os.path.join(path, "stereo"), exist_ok=True)
    for i, v in enumerate(views):
        K = np.asarray(v.K, np.float64)
        sm.cameras[i + 1] = W.SparseCamera(i + 1, 1, w, h, np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]))
        R = np.asarray(v.R, np.float64)
        qw = np.sqrt(max(1e-12, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
        sm.images[i + 1] = W.SparseImage(i + 1, q, np.asarray(v.T, np.float64), i + 1, names[i])
        PILImage.fromarray(v.gray).save(os.path.join(path, "images", names[i]))
    pid = 0
    for _ in range(num_points):
        i = int(rng.integers(len(views)))
        v = views[i]
        x, y = int(rng.integers(2, w - 2)), int(rng.integers(2, h - 2))
        d = float(v.depth[y, x])
        if not d > 0:
            continue
        K, R, T = (np.asarray(a, np.float64) for a in (v.K, v.R, v.T))
        X = R.T @ (d * np.linalg.inv(K) @ np.array([x, y, 1.0]) - T)
        track = []
        for j, u in enumerate(views):
            Kj, Rj, Tj = (np.asarray(a, np.float64) for a in (u.K, u.R, u.T))
            pc = Rj @ X + Tj
            if pc[2] <= 0:
                continue
            px = Kj @ (pc / pc[2])
            cx, cy = int(round(px[0])), int(round(px[1]))
            if 0 <= cx < w and 0 <= cy < h and abs(float(u.depth[cy, cx]) - pc[2]) < 0.02 * pc[2]:
                im = sm.images[j + 1]
                track.append((j + 1, len(im.xys)))
                im.xys = np.vstack([im.xys, px[:2]])
                im.point3D_ids = np.append(im.point3D_ids, pid + 1)
        if len(track) >= 2:
            pid += 1
            sm.points3D[pid] = W.SparsePoint3D(pid, X, (128, 128, 128), 0.1, track)
        else:
            for iid, idx in track:
                sm.images[iid].xys = sm.images[iid].xys[:-1]
                sm.images[iid].point3D_ids = sm.images[iid].point3D_ids[:-1]
    W.write_model_binary(sm, os.path.join(path, "sparse"))
    W.write_patch_match_config(os.path.join(path, "stereo", "patch-match.cfg"), names, cfg_spec)
    return names


def bench_crop_problem(device=None, cw=512, ch=384):
    """The problem bench.py hands to the oracle for `cpu_baseline`: a 512 x 384 centre crop of a 2560 x 1920
    reference image against its 20 full-resolution sources (S = 20, M = 15). Returns (views with the crop in the
    reference's place, ref index, source indices, the crop view, depth range)."""
    W, H, S = 2560, 1920, 20
    views = syn.make_scene(S + 1, W, H, arc_deg=3.6 * S, **({"device": device} if device else {}))
    ref = S // 2
    src = [i for i in range(S + 1) if i != ref]
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    v = views[ref]
    K = v.K.copy()
    K[0, 2] -= x0
    K[1, 2] -= y0
    crop = syn.View(K, v.R, v.T, np.ascontiguousarray(v.gray[y0:y0 + ch, x0:x0 + cw]),
                    np.ascontiguousarray(v.depth[y0:y0 + ch, x0:x0 + cw]), None)
    mixed = [crop if i == ref else u for i, u in enumerate(views)]
    return mixed, ref, src, crop, (float(v.depth.min() * 0.9), float(v.depth.max() * 1.1))
