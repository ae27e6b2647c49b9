"""The oracles against the committed golden fixtures (tests/golden/make_golden.py)."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pm_oracle_reproduces_golden(pm_oracle):
    g = np.load(os.path.join(HERE, "pm_48x36.npz"))
    imgs = [dict(K=g["K"][i], R=g["R"][i], T=g["T"][i], gray=g["gray"][i]) for i in range(len(g["gray"]))]
    dmin, dmax = g["depth_range"]
    for order in (0, 1):
        o = pm_oracle.default_options(depth_min=float(dmin), depth_max=float(dmax), geom_consistency=0, filter=1,
                                      num_iterations=1, order=order)
        r = pm_oracle.run(o, imgs, 1, [0, 2, 3], want_cost=True)
        for k, v in r.items():
            assert np.array_equal(v, g[f"order{order}_{k}"]), (order, k)
    raw, uni = pm_oracle.rng_stream(12345, 16)
    assert np.array_equal(raw, g["xorwow_raw"]) and np.array_equal(uni, g["xorwow_uniform"])


def test_ba_oracle_reproduces_golden():
    import ba_oracle
    from colmap_amd import estimators as est
    g = np.load(os.path.join(HERE, "ba_6x40.npz"))
    fp = est.FlatProblem(**{k: g[f"in_{k}"].copy() for k in ("poses", "cams", "cam_model", "points", "obs_pose", "obs_cam",
                                                           "obs_point", "obs_xy", "pose_const", "pose_fixed_t",
                                                           "cam_const", "point_const")})
    # the fixture was written with 12 doubles per camera block; BA_CAM_STRIDE is 16 now (unused tail: zero / constant)
    pad = est.CAM_STRIDE - fp.cams.shape[1]
    fp.cams = np.ascontiguousarray(np.pad(fp.cams, ((0, 0), (0, pad))))
    fp.cam_const = np.ascontiguousarray(np.pad(fp.cam_const, ((0, 0), (0, pad)), constant_values=1))
    s = est.solve_flat(fp, est.SolverOptions(gradient_tolerance=1e-10, max_num_iterations=200), solve_fn=ba_oracle.solve_fn)
    assert [s.num_residuals, s.num_effective_parameters] == list(g["counts"])
    assert s.initial_cost == g["costs"][0]
    # the final iterate depends on the OpenMP thread count only through round-off
    assert abs(s.final_cost - g["costs"][1]) <= 1e-10 * g["costs"][1]
    np.testing.assert_allclose(fp.points, g["out_points"], atol=1e-8)
    np.testing.assert_allclose(fp.poses, g["out_poses"], atol=1e-8)
