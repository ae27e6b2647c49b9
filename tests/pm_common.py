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
