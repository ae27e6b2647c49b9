"""Generates tests/golden/*.npz from the CPU oracles (run once; the outputs are committed).

The reference holds no golden vectors for PatchMatch and cannot be built here (DESIGN.md section
1.2), so these fixtures pin OUR oracles against accidental change: any edit of oracle/*.c that moves
a bit of these outputs shows up as a failing CPU test. Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pm_case():
    import pm_oracle
    from pm_common import scene, oracle_inputs
    from colmap_amd import synthetic as syn
    views = scene(4, 48, 36)
    dmin, dmax = syn.depth_range(views, 1)
    out = {}
    for order in (0, 1):
        o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1,
                                      num_iterations=1, order=order)
        r = pm_oracle.run(o, oracle_inputs(views), 1, [0, 2, 3], want_cost=True)
        for k, v in r.items():
            out[f"order{order}_{k}"] = v
    out["gray"] = np.stack([v.gray for v in views])
    out["K"] = np.stack([v.K for v in views]); out["R"] = np.stack([v.R for v in views]); out["T"] = np.stack([v.T for v in views])
    out["depth_range"] = np.array([dmin, dmax])
    raw, uni = pm_oracle.rng_stream(12345, 16)
    out["xorwow_raw"], out["xorwow_uniform"] = raw, uni
    return out


def ba_case():
    import ba_oracle
    from colmap_amd import estimators as est, scene
    d = scene.synthesize_flat(6, 40, 4, seed=3, mixed_models=True, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
    fp = est.FlatProblem.from_arrays(d)
    est.fix_gauge_two_cams(fp)
    inp = {k: getattr(fp, k).copy() for k in ("poses", "cams", "cam_model", "points", "obs_pose", "obs_cam", "obs_point",
                                              "obs_xy", "pose_const", "pose_fixed_t", "cam_const", "point_const")}
    s = est.solve_flat(fp, est.SolverOptions(gradient_tolerance=1e-10, max_num_iterations=200), solve_fn=ba_oracle.solve_fn)
    out = {f"in_{k}": v for k, v in inp.items()}
    out.update(out_poses=fp.poses, out_cams=fp.cams, out_points=fp.points,
               costs=np.array([s.initial_cost, s.final_cost]), counts=np.array([s.num_residuals, s.num_effective_parameters]))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "pm_48x36.npz"), **pm_case())
    np.savez_compressed(os.path.join(HERE, "ba_6x40.npz"), **ba_case())
    print("written", os.listdir(HERE))
