"""CPU tests of the bundle-adjustment oracle and of the host adapter (flattening rules).

Known answers restated from the reference's own tests:
  estimators/cost_functions/reprojection_error_test.cc:41-72   (Nominal residual values)
  estimators/cost_functions/reprojection_error_test.cc:211-325 (analytic vs numeric Jacobians)
  estimators/bundle_adjustment_test.cc:303-412                 (Nominal / 594 / 80 + constant points)
  estimators/bundle_adjustment_ceres_test.cc:244-256           (TwoView: 400 residuals, 309 parameters)
"""
import numpy as np
import pytest
import scipy.optimize

import ba_oracle
from colmap_amd import estimators as est
from colmap_amd import scene


def test_reproj_nominal_known_answers():
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    cases = [([0, 0, 1], [1, 0, 0], (0, 0)), ([0, 1, 1], [1, 0, 0], (0, 1)), ([0, 1, 1], [2, 0, 0], (0, 2)),
             ([-1, 1, 1], [2, 0, 0], (-2, 2)), ([-1, 1, -1], [2, 0, 0], (0, 0))]
    for pt, prm, want in cases:
        r, *_ = ba_oracle.reproj_error(scene.SIMPLE_PINHOLE, pt, pose, prm, [0, 0], want_jac=False)
        assert tuple(r) == want


@pytest.mark.parametrize("model,params", [(scene.SIMPLE_PINHOLE, [700.0, 320, 240]),
                                          (scene.PINHOLE, [700.0, 720, 320, 240]),
                                          (scene.SIMPLE_RADIAL, [700.0, 320, 240, 0.05])])
def test_analytic_vs_numeric_jacobians(model, params):
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        pose = np.concatenate([q, rng.normal(size=3) * 0.3 + [0, 0, 4]])
        pt = rng.normal(size=3) * 0.5
        xy = rng.normal(size=2) * 100
        r0, Jpt, Jpose, Jpar = ba_oracle.reproj_error(model, pt, pose, params, xy)
        def f(v):
            return ba_oracle.reproj_error(model, v[:3], v[3:10], v[10:], xy, want_jac=False)[0]
        x0 = np.concatenate([pt, pose, params])
        J = np.zeros((2, len(x0)))
        for i in range(len(x0)):
            h = 1e-6 * max(1.0, abs(x0[i]))
            e = np.zeros(len(x0)); e[i] = h
            J[:, i] = (f(x0 + e) - f(x0 - e)) / (2 * h)
        np.testing.assert_allclose(Jpt, J[:, :3], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(Jpose, J[:, 3:10], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(Jpar, J[:, 10:], rtol=1e-5, atol=1e-5)


def test_behind_camera_gives_zero_residual_and_jacobian():
    # reprojection_error.h:96-116
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    r, Jpt, Jpose, Jpar = ba_oracle.reproj_error(scene.SIMPLE_RADIAL, [0.1, 0.2, -1.0], pose, [700, 320, 240, 0.1], [5, 6])
    assert not r.any() and not Jpt.any() and not Jpose.any() and not Jpar.any()


def test_quaternion_plus_is_a_rotation_update():
    q = np.array([0.1, -0.2, 0.3, 0.9]); q /= np.linalg.norm(q)
    d = np.array([0.01, -0.02, 0.03])
    out = ba_oracle.quat_plus(q, d)
    assert abs(np.linalg.norm(out) - 1) < 1e-14
    assert np.array_equal(ba_oracle.quat_plus(q, np.zeros(3)), q)
    # rotation by angle 2|d| about d, applied on the left
    R = scene.quat_to_rot(out) @ scene.quat_to_rot(q).T
    ang = np.arccos((np.trace(R) - 1) / 2)
    assert abs(ang - 2 * np.linalg.norm(d)) < 1e-12


def _dataset(num_rigs, frames, points, noise, seed=0, **kw):
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=num_rigs, num_frames_per_rig=frames,
                                                               num_points3D=points, **kw), seed=seed)
    gt = rec.copy()
    scene.SynthesizeNoise(noise, rec, seed=seed + 1)
    return gt, rec


def _config(rec, gauge=est.BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD):
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    cfg.FixGauge(gauge)
    return cfg


def _recon_near(gt, rec, max_rot_deg, max_center):
    """ReconstructionNear (scene/reconstruction_matchers.h:87-217) without the Sim3 alignment:
    the gauge holds frame 1, so both live in the same coordinate frame up to the noise on it."""
    for i in gt.RegImageIds():
        a, b = gt.images[i].cam_from_world, rec.images[i].cam_from_world
        R = scene.quat_to_rot(a[:4]).T @ scene.quat_to_rot(b[:4])
        ang = np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))
        ca = -scene.quat_to_rot(a[:4]).T @ a[4:]
        cb = -scene.quat_to_rot(b[:4]).T @ b[4:]
        assert ang < max_rot_deg and np.linalg.norm(ca - cb) < max_center, (i, ang, np.linalg.norm(ca - cb))


def test_backend_nominal():
    # bundle_adjustment_test.cc:303-348 (noise only on what the gauge leaves free to compare)
    gt, rec = _dataset(1, 10, 200, scene.SyntheticNoiseOptions(point2D_stddev=0.5, point3D_stddev=0.1))
    # perturb all poses except the two gauge frames' fixed parts
    rng = np.random.default_rng(5)
    ids = rec.RegImageIds()
    for i in ids[2:]:
        rec.images[i].cam_from_world[4:] += rng.normal(0, 0.1, 3)
        ang = np.deg2rad(rng.normal(0, 0.5))
        rec.images[i].cam_from_world[:4] = scene.quat_mul(rec.images[i].cam_from_world[:4],
                                                           np.array([0, 0, np.sin(ang / 2), np.cos(ang / 2)]))
    ba = est.BundleAdjuster(est.BundleAdjustmentOptions(), _config(rec), rec, solve_fn=ba_oracle.solve_fn)
    assert ba.Config().NumImages() == 10
    summary = ba.Solve()
    assert summary.IsSolutionUsable() and summary.num_residuals > 0
    assert summary.final_cost < 0.5 * summary.initial_cost
    _recon_near(gt, rec, 0.1, 0.1)


def test_backend_minimum_track_length():
    # bundle_adjustment_test.cc:350-381: 99 points x 3 observations x 2 = 594
    _, rec = _dataset(3, 1, 100, scene.SyntheticNoiseOptions(point2D_stddev=1), num_points2D_without_point3D=0)
    img3 = rec.images[3]
    idx = next(i for i, p in enumerate(img3.points2D) if p.HasPoint3D())
    rec.DeleteObservation(3, idx)
    opts = est.BundleAdjustmentOptions(min_track_length=3)
    summary = est.BundleAdjuster(opts, _config(rec), rec, solve_fn=ba_oracle.solve_fn).Solve()
    assert summary.IsSolutionUsable()
    assert summary.num_residuals == 594


def test_backend_constant_points3D():
    # bundle_adjustment_test.cc:383-412: 20 points x 2 images x 2 = 80, points bit-identical
    _, rec = _dataset(2, 1, 20, scene.SyntheticNoiseOptions(point2D_stddev=1))
    orig = rec.copy()
    cfg = _config(rec, est.BundleAdjustmentGauge.UNSPECIFIED)
    opts = est.BundleAdjustmentOptions(refine_points3D=False)
    summary = est.BundleAdjuster(opts, cfg, rec, solve_fn=ba_oracle.solve_fn).Solve()
    assert summary.IsSolutionUsable()
    assert summary.num_residuals == 80
    for pid, pt in rec.points3D.items():
        assert np.array_equal(pt.xyz, orig.points3D[pid].xyz)


def test_two_view_parameter_count():
    # bundle_adjustment_ceres_test.cc:244-256: 400 residuals, 309 effective parameters
    # (100 x 3 points + 5 pose DoF of image 2 + 2 x 2 camera parameters)
    _, rec = _dataset(2, 1, 100, scene.SyntheticNoiseOptions(point2D_stddev=1), num_points2D_without_point3D=0)
    ba = est.BundleAdjuster(est.BundleAdjustmentOptions(), _config(rec), rec, solve_fn=ba_oracle.solve_fn)
    summary = ba.Solve()
    assert summary.num_residuals == 400
    assert summary.num_effective_parameters == 309
    fp = ba.problem_
    assert fp.pose_const.sum() == 1 and (fp.pose_fixed_t >= 0).sum() == 1


def test_constant_pose_and_intrinsics_untouched():
    _, rec = _dataset(1, 4, 60, scene.SyntheticNoiseOptions(point2D_stddev=0.5, point3D_stddev=0.05))
    orig = rec.copy()
    cfg = _config(rec)
    cfg.SetConstantRigFromWorldPose(3)
    cfg.SetConstantCamIntrinsics(1)
    est.BundleAdjuster(est.BundleAdjustmentOptions(), cfg, rec, solve_fn=ba_oracle.solve_fn).Solve()
    assert np.array_equal(rec.images[3].cam_from_world, orig.images[3].cam_from_world)
    # gauge (bundle_adjustment_ceres.cc:343-416): the already-constant frame 3 is "image1"; the first
    # variable frame (image 1) only gets its largest-baseline translation coordinate held
    same = rec.images[1].cam_from_world[4:] == orig.images[1].cam_from_world[4:]
    assert same.sum() == 1 and not np.array_equal(rec.images[1].cam_from_world[:4], orig.images[1].cam_from_world[:4])
    assert np.array_equal(rec.cameras[1].params, orig.cameras[1].params)
    assert not np.array_equal(rec.images[4].cam_from_world, orig.images[4].cam_from_world)
    # principal point is never refined by default (bundle_adjustment.h:177-178)
    cfg2 = _config(orig)
    rec2 = orig.copy()
    est.BundleAdjuster(est.BundleAdjustmentOptions(), cfg2, rec2, solve_fn=ba_oracle.solve_fn).Solve()
    assert np.array_equal(rec2.cameras[1].params[1:3], orig.cameras[1].params[1:3])
    assert rec2.cameras[1].params[0] != orig.cameras[1].params[0]


def _scipy_reference(fp):
    """Same least-squares problem minimised by scipy (trust-region reflective, dense-ish)."""
    n_c, n_k, n_p = len(fp.poses), len(fp.cams), len(fp.points)
    q0 = fp.poses[:, :4].copy()
    var_cam = [np.nonzero(fp.cam_const[k, : scene.MODEL_NUM_PARAMS[int(fp.cam_model[k])]] == 0)[0] for k in range(n_k)]
    def unpack(x):
        poses, cams, pts = fp.poses.copy(), fp.cams.copy(), fp.points.copy()
        o = 0
        for i in range(n_c):
            if fp.pose_const[i]:
                continue
            poses[i, :4] = ba_oracle.quat_plus(q0[i], x[o:o + 3]); o += 3
            for c in range(3):
                if c != fp.pose_fixed_t[i]:
                    poses[i, 4 + c] = x[o]; o += 1
        for k in range(n_k):
            for j in var_cam[k]:
                cams[k, j] = x[o]; o += 1
        for j in range(n_p):
            if not fp.point_const[j]:
                pts[j] = x[o:o + 3]; o += 3
        return poses, cams, pts
    def pack():
        x = []
        for i in range(n_c):
            if fp.pose_const[i]:
                continue
            x += [0, 0, 0] + [fp.poses[i, 4 + c] for c in range(3) if c != fp.pose_fixed_t[i]]
        for k in range(n_k):
            x += [fp.cams[k, j] for j in var_cam[k]]
        for j in range(n_p):
            if not fp.point_const[j]:
                x += list(fp.points[j])
        return np.array(x, float)
    def fun(x):
        poses, cams, pts = unpack(x)
        r = np.zeros(2 * len(fp.obs_pose))
        for o in range(len(fp.obs_pose)):
            r[2 * o:2 * o + 2] = ba_oracle.reproj_error(int(fp.cam_model[fp.obs_cam[o]]), pts[fp.obs_point[o]],
                                                        poses[fp.obs_pose[o]], cams[fp.obs_cam[o]][:4],
                                                        fp.obs_xy[o], want_jac=False)[0]
        return r
    sol = scipy.optimize.least_squares(fun, pack(), method="trf", x_scale="jac", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    return 0.5 * float(sol.fun @ sol.fun)


def test_final_cost_matches_scipy():
    d = scene.synthesize_flat(6, 40, 4, seed=3, mixed_models=True,
                              noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
    fp = est.FlatProblem.from_arrays(d)
    est.fix_gauge_two_cams(fp)
    want = _scipy_reference(fp.copy())
    so = est.SolverOptions(gradient_tolerance=1e-10, max_num_iterations=200)
    summary = est.solve_flat(fp, so, solve_fn=ba_oracle.solve_fn)
    assert summary.IsSolutionUsable()
    assert abs(summary.final_cost - want) <= 1e-7 * want, (summary.final_cost, want)
    q = fp.poses[:, :4]
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)  # renormalised (:491,508)


def test_three_point_gauge_and_no_gauge_also_converge():
    d = scene.synthesize_flat(5, 30, 4, seed=4, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.02, 0.5))
    base = est.FlatProblem.from_arrays(d)
    costs = []
    for mode in ("two_cams", "three_points", "none"):
        fp = base.copy()
        if mode == "two_cams":
            assert est.fix_gauge_two_cams(fp)
        elif mode == "three_points":
            assert est.fix_gauge_three_points(fp) and fp.point_const.sum() == 3
        s = est.solve_flat(fp, est.SolverOptions(gradient_tolerance=1e-9, max_num_iterations=300),
                           solve_fn=ba_oracle.solve_fn)
        assert s.IsSolutionUsable()
        costs.append(s.final_cost)
    # fixing 7 gauge freedoms with two cameras does not change the attainable cost; three held
    # (noisy) points are 9 constraints on a 7-dimensional gauge, so that minimum is higher
    assert abs(costs[0] - costs[2]) <= 1e-9 * costs[2]
    assert costs[1] >= costs[2]


# ------------------------------------------------------------------------------------------------
# RADIAL model, rigs with a constant sensor_from_rig, robust losses
# ------------------------------------------------------------------------------------------------

def test_radial_model_values_and_jacobians():
    # models_jacobian.h:323-398; forward model sensor/models.h (RadialCameraModel::ImgFromCam)
    rng = np.random.default_rng(3)
    params = [650.0, 320, 240, 0.08, -0.02]
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for _ in range(20):
        pt = np.array([rng.normal() * 0.6, rng.normal() * 0.6, 3 + rng.random()])
        r0, Jpt, Jpose, Jpar = ba_oracle.reproj_error(scene.RADIAL, pt, pose, params, [0, 0])
        np.testing.assert_allclose(r0, scene.img_from_cam(scene.RADIAL, np.array(params), pt[None])[0], rtol=1e-13)
        def f(v):
            return ba_oracle.reproj_error(scene.RADIAL, v[:3], pose, v[3:], [0, 0], want_jac=False)[0]
        x0 = np.concatenate([pt, params])
        J = np.zeros((2, len(x0)))
        for i in range(len(x0)):
            h = 1e-6 * max(1.0, abs(x0[i]))
            e = np.zeros(len(x0)); e[i] = h
            J[:, i] = (f(x0 + e) - f(x0 - e)) / (2 * h)
        np.testing.assert_allclose(Jpt, J[:, :3], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(Jpar, J[:, 3:], rtol=1e-5, atol=1e-5)


def test_rig_residual_equals_composed_pose_and_jacobians_are_exact():
    # RigReprojErrorConstantRigCostFunctor (reprojection_error.h:386-417): same residual as the plain
    # functor on cam_from_world = sensor_from_rig * rig_from_world; derivatives w.r.t. rig_from_world
    rng = np.random.default_rng(4)
    params = [700.0, 320, 240, 0.05]
    for _ in range(20):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        rig = np.concatenate([q, rng.normal(size=3) * 0.3 + [0, 0, 4]])
        ang = rng.normal() * 0.2
        sens = np.array([0, 0, np.sin(ang / 2), np.cos(ang / 2), *(rng.normal(size=3) * 0.1)])
        pt = rng.normal(size=3) * 0.5
        xy = rng.normal(size=2) * 50
        r, Jpt, Jpose, Jpar = ba_oracle.rig_reproj_error(scene.SIMPLE_RADIAL, pt, rig, sens, params, xy)
        r2, *_ = ba_oracle.reproj_error(scene.SIMPLE_RADIAL, pt, scene.rigid_compose(sens, rig), params, xy,
                                        want_jac=False)
        np.testing.assert_allclose(r, r2, rtol=0, atol=1e-9)
        def f(v):
            return ba_oracle.rig_reproj_error(scene.SIMPLE_RADIAL, v[:3], v[3:10], sens, v[10:], xy, want_jac=False)[0]
        x0 = np.concatenate([pt, rig, params])
        J = np.zeros((2, len(x0)))
        for i in range(len(x0)):
            h = 1e-6 * max(1.0, abs(x0[i]))
            e = np.zeros(len(x0)); e[i] = h
            J[:, i] = (f(x0 + e) - f(x0 - e)) / (2 * h)
        np.testing.assert_allclose(Jpt, J[:, :3], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(Jpose, J[:, 3:10], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(Jpar, J[:, 10:], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("loss,scale", [(est.LossFunctionType.SOFT_L1, 1.5), (est.LossFunctionType.CAUCHY, 2.0),
                                        (est.LossFunctionType.HUBER, 1.2), (est.LossFunctionType.TRIVIAL, 1.0)])
def test_loss_functions_closed_forms(loss, scale):
    """ceres::SoftLOneLoss / CauchyLoss / HuberLoss / TrivialLoss: rho(s) in closed form, rho' and
    rho'' by differentiation, rho(0) = 0, rho'(0) = 1 (Ceres' normalisation), rho'' <= 0."""
    a = scale
    closed = {
        est.LossFunctionType.TRIVIAL: lambda s: s,
        est.LossFunctionType.SOFT_L1: lambda s: 2 * a * a * (np.sqrt(1 + s / (a * a)) - 1),
        est.LossFunctionType.CAUCHY: lambda s: a * a * np.log1p(s / (a * a)),
        est.LossFunctionType.HUBER: lambda s: s if s <= a * a else 2 * a * np.sqrt(s) - a * a,
    }[loss]
    rho0 = ba_oracle.loss(loss, a, 0.0)
    assert rho0[0] == 0.0 and rho0[1] == 1.0
    for s in [1e-3, 0.3, 1.0, a * a * 0.999, a * a * 1.001, 7.0, 250.0]:
        rho = ba_oracle.loss(loss, a, s)
        assert abs(rho[0] - closed(s)) <= 1e-12 * max(1.0, closed(s))
        h = 1e-5 * s
        d1 = (closed(s + h) - closed(s - h)) / (2 * h)
        assert abs(rho[1] - d1) <= 1e-6 * max(1.0, abs(d1))
        assert rho[2] <= 0.0
        if loss != est.LossFunctionType.HUBER or abs(s - a * a) > 0.01:
            d2 = (ba_oracle.loss(loss, a, s + h)[1] - ba_oracle.loss(loss, a, s - h)[1]) / (2 * h)
            assert abs(rho[2] - d2) <= 1e-5 * max(1e-3, abs(d2))


def _rig_dataset(num_rigs, cams, frames, points, noise, seed=0):
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=num_rigs, num_cameras_per_rig=cams, num_frames_per_rig=frames, num_points3D=points,
        num_points2D_without_point3D=0), seed=seed)
    gt = rec.copy()
    scene.SynthesizeNoise(noise, rec, seed=seed + 1)
    return gt, rec


def test_two_view_rig_counts():
    # bundle_adjustment_ceres_test.cc:323-376 (TwoViewRig) with the sensor_from_rig block held constant:
    # 800 residuals; 97 x 3 points + 2 x 6 rig_from_world + 2 x 2 camera parameters = 307 (the
    # reference's 313 minus the 6 sensor_from_rig parameters)
    _, rec = _rig_dataset(1, 2, 2, 100, scene.SyntheticNoiseOptions(point2D_stddev=1))
    opts = est.BundleAdjustmentOptions(refine_sensor_from_rig=False)
    ba = est.BundleAdjuster(opts, _config(rec, est.BundleAdjustmentGauge.THREE_POINTS), rec,
                            solve_fn=ba_oracle.solve_fn)
    summary = ba.Solve()
    assert summary.IsSolutionUsable()
    assert summary.num_residuals == 800
    assert summary.num_effective_parameters == 307
    fp = ba.problem_
    assert len(fp.poses) == 2 and fp.sensors.shape == (1, 7)          # two frames, one non-reference sensor
    assert (fp.obs_sensor >= 0).sum() == 200 * 2 // 2                  # the observations of camera 2
    # the config can hold that sensor constant (ManyViewRigConstantSensorFromRig, :434-489)
    cfg = _config(rec, est.BundleAdjustmentGauge.THREE_POINTS)
    cfg.SetConstantSensorFromRigPose(2)
    held = est.BundleAdjuster(est.BundleAdjustmentOptions(), cfg, rec.copy(), solve_fn=ba_oracle.solve_fn)
    assert held.problem_.sensor_const.all()
    # the reference's own TwoViewRig (:323-376): refine_sensor_from_rig (the default) makes the
    # sensor_from_rig of camera 2 a parameter block: 800 residuals, 313 effective parameters, and it moves
    rec2 = rec.copy()
    before = rec2.rigs[1].sensors[2].copy()
    ba2 = est.BundleAdjuster(est.BundleAdjustmentOptions(), _config(rec2, est.BundleAdjustmentGauge.THREE_POINTS),
                             rec2, solve_fn=ba_oracle.solve_fn)
    assert ba2.problem_.sensor_const.tolist() == [0]
    s2 = ba2.Solve()
    assert s2.IsSolutionUsable() and s2.num_residuals == 800 and s2.num_effective_parameters == 313
    assert not np.array_equal(rec2.rigs[1].sensors[2], before)
    assert abs(np.linalg.norm(rec2.rigs[1].sensors[2][:4]) - 1.0) < 1e-12
    assert s2.final_cost <= summary.final_cost * (1 + 1e-9)      # six more degrees of freedom


def test_rig_residual_sensor_jacobian():
    """RigReprojErrorCostFunctor (reprojection_error.h:344-384): analytic Jacobian w.r.t. the
    sensor_from_rig block against central differences; the value equals the constant-rig residual."""
    rng = np.random.default_rng(5)
    for model, params in ((scene.SIMPLE_RADIAL, [600.0, 320, 240, 0.05]), (scene.OPENCV, [600.0, 610, 320, 240, 0.02, -0.01, 1e-3, -1e-3])):
        for _ in range(5):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            rig = np.concatenate([q, rng.normal(size=3) * 0.2])
            qs = np.array([0.05, -0.03, 0.02, 1.0]) + rng.normal(size=4) * 0.01; qs /= np.linalg.norm(qs)
            sens = np.concatenate([qs, [0.3, -0.1, 0.05]])
            X = scene.quat_to_rot(q).T @ (np.array([rng.normal() * 0.3, rng.normal() * 0.3, 4.0]) - rig[4:])
            r, Jpt, Jpose, Jpar, Jsens = ba_oracle.rig_reproj_error_sensor(model, X, rig, sens, params, [1.0, 2.0])
            r0, Jpt0, Jpose0, Jpar0 = ba_oracle.rig_reproj_error(model, X, rig, sens, params, [1.0, 2.0])
            assert np.array_equal(r, r0) and np.array_equal(Jpose, Jpose0) and np.array_equal(Jpt, Jpt0)
            J = np.zeros((2, 7))
            for i in range(7):
                h = 1e-6
                e = np.zeros(7); e[i] = h
                f = lambda sv: ba_oracle.rig_reproj_error(model, X, rig, sv, params, [1.0, 2.0], want_jac=False)[0]
                J[:, i] = (f(sens + e) - f(sens - e)) / (2 * h)
            np.testing.assert_allclose(Jsens, J, rtol=1e-5, atol=1e-5)


def test_many_view_rig_refines_sensor_from_rig():
    """bundle_adjustment_ceres_test.cc ManyViewRig (:378-432): with noise on the sensor_from_rig of
    the non-reference camera the adjustment brings it back to the ground truth."""
    gt, rec = _rig_dataset(1, 2, 6, 200, scene.SyntheticNoiseOptions(point2D_stddev=0.3))
    rig = rec.rigs[1]
    cid = next(iter(rig.sensors))
    rig.sensors[cid] = rig.sensors[cid] + np.array([0, 0, 0, 0, 0.05, -0.04, 0.03])
    rec.UpdateCamFromWorld()
    ba = est.BundleAdjuster(est.BundleAdjustmentOptions(), _config(rec), rec, solve_fn=ba_oracle.solve_fn)
    s = ba.Solve()
    assert s.IsSolutionUsable() and s.final_cost < 0.05 * s.initial_cost
    np.testing.assert_allclose(rec.rigs[1].sensors[cid][4:], gt.rigs[1].sensors[cid][4:], atol=0.01)


def test_many_view_rig_recovers_ground_truth():
    # bundle_adjustment_ceres_test.cc:183-220 (NominalMultiCameraRig), constant sensor_from_rig
    gt, rec = _rig_dataset(2, 3, 5, 100, scene.SyntheticNoiseOptions(
        point2D_stddev=0.3, point3D_stddev=0.05, rig_from_world_translation_stddev=0.02,
        rig_from_world_rotation_stddev=0.5))
    # the gauge frames keep their (noisy) poses: restore the first two frames to ground truth
    for fid in (1, 2):
        rec.frames[fid].rig_from_world = gt.frames[fid].rig_from_world.copy()
    rec.UpdateCamFromWorld()
    opts = est.BundleAdjustmentOptions(refine_sensor_from_rig=False)
    ba = est.BundleAdjuster(opts, _config(rec), rec, solve_fn=ba_oracle.solve_fn)
    summary = ba.Solve()
    assert summary.IsSolutionUsable()
    assert summary.num_residuals == 2 * sum(len(p.track) for p in rec.points3D.values())
    # 10 frames: one constant, one with a fixed translation coordinate
    fp = ba.problem_
    assert len(fp.poses) == 10 and fp.pose_const.sum() == 1 and (fp.pose_fixed_t >= 0).sum() == 1
    assert summary.final_cost < 0.2 * summary.initial_cost
    _recon_near(gt, rec, 0.1, 0.05)
    # every image of a frame moved rigidly with its frame
    for fr in rec.frames.values():
        for im in fr.image_ids:
            img = rec.images[im]
            if not rec.IsRefInFrame(im):
                want = scene.rigid_compose(rec.SensorFromRig(im), fr.rig_from_world)
                np.testing.assert_allclose(img.cam_from_world, want, atol=1e-15)


def test_constant_frame_of_a_rig_uses_composed_constant_poses():
    # ManyViewRigConstantRigFromWorld (:491-553): frame 1 constant -> its three images enter with
    # ReprojErrorConstantPoseCostFunctor on sensor_from_rig * rig_from_world (:769-772,797-803)
    _, rec = _rig_dataset(2, 3, 5, 60, scene.SyntheticNoiseOptions(point2D_stddev=1))
    orig = rec.copy()
    cfg = _config(rec, est.BundleAdjustmentGauge.THREE_POINTS)
    cfg.SetConstantRigFromWorldPose(1)
    opts = est.BundleAdjustmentOptions(refine_sensor_from_rig=False)
    ba = est.BundleAdjuster(opts, cfg, rec, solve_fn=ba_oracle.solve_fn)
    fp = ba.problem_
    # 9 variable frame blocks + 3 constant blocks (frame 1: rig pose for the reference image, two compositions)
    assert len(fp.poses) == 12 and fp.pose_const.sum() == 3
    summary = ba.Solve()
    assert summary.IsSolutionUsable()
    assert np.array_equal(rec.frames[1].rig_from_world, orig.frames[1].rig_from_world)
    for im in orig.frames[1].image_ids:
        assert np.array_equal(rec.images[im].cam_from_world, orig.images[im].cam_from_world)
    assert not np.array_equal(rec.frames[2].rig_from_world, orig.frames[2].rig_from_world)


def test_robust_loss_downweights_outliers():
    """Gross outliers in 5 % of the observations: with the trivial loss they drag the solution
    away, with CAUCHY (the loss COLMAP's mapper selects for local BA) the ground truth is recovered."""
    gt, rec = _dataset(1, 8, 150, scene.SyntheticNoiseOptions(point2D_stddev=0.3, point3D_stddev=0.05))
    rng = np.random.default_rng(11)
    for i in rec.RegImageIds():
        for p2 in rec.images[i].points2D:
            if p2.HasPoint3D() and rng.random() < 0.05:
                p2.xy = p2.xy + rng.normal(0, 80, 2)
    def solve(loss):
        r = rec.copy()
        so = est.SolverOptions(loss_type=int(loss), loss_scale=1.0)
        opts = est.BundleAdjustmentOptions(solver_options=so)
        s = est.BundleAdjuster(opts, _config(r), r, solve_fn=ba_oracle.solve_fn).Solve()
        assert s.IsSolutionUsable()
        err = np.mean([np.linalg.norm(r.points3D[k].xyz - gt.points3D[k].xyz) for k in gt.points3D])
        return s, err
    s_l2, err_l2 = solve(est.LossFunctionType.TRIVIAL)
    s_c, err_c = solve(est.LossFunctionType.CAUCHY)
    assert err_c < 0.25 * err_l2, (err_c, err_l2)
    assert err_c < 0.02
    assert s_c.final_cost < s_l2.final_cost   # 1/2 sum rho(s) <= 1/2 sum s


def test_robust_cost_matches_scipy_least_squares():
    """The LM solve on the corrected system minimises 1/2 sum rho(|r|^2): scipy's soft_l1 / cauchy /
    huber losses use the same rho (f_scale = a) and must reach the same optimum."""
    _, rec = _dataset(1, 4, 30, scene.SyntheticNoiseOptions(point2D_stddev=2.0, point3D_stddev=0.05))
    cfg = _config(rec)
    for i in rec.RegImageIds():
        cfg.SetConstantRigFromWorldPose(i)
    for loss, name in [(est.LossFunctionType.SOFT_L1, "soft_l1"), (est.LossFunctionType.CAUCHY, "cauchy"),
                       (est.LossFunctionType.HUBER, "huber")]:
        r = rec.copy()
        so = est.SolverOptions(loss_type=int(loss), loss_scale=2.0, gradient_tolerance=1e-10, max_num_iterations=200)
        opts = est.BundleAdjustmentOptions(refine_focal_length=False, refine_extra_params=False, solver_options=so)
        ba = est.BundleAdjuster(opts, cfg, r, solve_fn=ba_oracle.solve_fn)
        fp0 = ba.problem_.copy()
        s = ba.Solve()
        assert s.IsSolutionUsable()
        # scipy on the points only (poses and intrinsics constant)
        def resid(x):
            pts = x.reshape(-1, 3)
            out = np.zeros(2 * len(fp0.obs_pose))
            for o in range(len(fp0.obs_pose)):
                rr, *_ = ba_oracle.reproj_error(int(fp0.cam_model[fp0.obs_cam[o]]), pts[fp0.obs_point[o]],
                                                fp0.poses[fp0.obs_pose[o]], fp0.cams[fp0.obs_cam[o]],
                                                fp0.obs_xy[o], want_jac=False)
                out[2 * o: 2 * o + 2] = rr
            return out
        def total(x):
            rr = resid(x).reshape(-1, 2)
            return 0.5 * sum(ba_oracle.loss(loss, 2.0, float(v @ v))[0] for v in rr)
        assert abs(total(ba.problem_.points.ravel()) - s.final_cost) <= 1e-9 * s.final_cost
        # local optimality of the robust cost: no better point nearby along the gradient
        x = ba.problem_.points.ravel().copy()
        g = scipy.optimize.approx_fprime(x, total, 1e-7)
        assert np.abs(g).max() < 1e-3 * max(1.0, s.final_cost)


def test_opencv_model_values_and_jacobians():
    # models_jacobian.h:401-496; forward model = OpenCVCameraModel::ImgFromCam (sensor/models.h)
    rng = np.random.default_rng(5)
    params = [640.0, 655.0, 320, 240, 0.08, -0.03, 0.002, -0.001]
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for _ in range(20):
        pt = np.array([rng.normal() * 0.6, rng.normal() * 0.6, 3 + rng.random()])
        r0, Jpt, Jpose, Jpar = ba_oracle.reproj_error(scene.OPENCV, pt, pose, params, [0, 0])
        np.testing.assert_allclose(r0, scene.img_from_cam(scene.OPENCV, np.array(params), pt[None])[0], rtol=1e-13)
        def f(v):
            return ba_oracle.reproj_error(scene.OPENCV, v[:3], pose, v[3:], [0, 0], want_jac=False)[0]
        x0 = np.concatenate([pt, params])
        J = np.zeros((2, len(x0)))
        for i in range(len(x0)):
            h = 1e-6 * max(1.0, abs(x0[i]))
            e = np.zeros(len(x0)); e[i] = h
            J[:, i] = (f(x0 + e) - f(x0 - e)) / (2 * h)
        np.testing.assert_allclose(Jpt, J[:, :3], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(Jpar, J[:, 3:], rtol=1e-5, atol=2e-5)


def test_opencv_cameras_six_variable_intrinsics():
    """OPENCV with COLMAP's default refinement (focal + extra, principal point fixed): six variable
    intrinsics per camera (fx fy k1 k2 p1 p2) -- wider than the pose blocks' neighbours in the tangent vector."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=3, num_frames_per_rig=4, num_points3D=200, camera_model_id=scene.OPENCV,
        camera_params=(1280.0, 1290.0, 512.0, 384.0, 0.05, -0.01, 0.001, -0.002)), seed=2)
    gt = rec.copy()
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(point2D_stddev=0.3, point3D_stddev=0.05), rec, seed=3)
    for c in rec.cameras.values():
        c.params = c.params * np.array([1.01, 0.99, 1, 1, 1.2, 0.8, 1.3, 0.7])
    ba = est.BundleAdjuster(est.BundleAdjustmentOptions(), _config(rec), rec, solve_fn=ba_oracle.solve_fn)
    fp = ba.problem_
    assert (fp.cam_const[:, :8] == [0, 0, 1, 1, 0, 0, 0, 0]).all()
    s = ba.Solve()
    assert s.IsSolutionUsable() and s.final_cost < 0.05 * s.initial_cost
    # 200 points x 3 + 12 frames (6 x 10 + 5, one constant) + 3 cameras x 6
    assert s.num_effective_parameters == 3 * len(fp.points) + 6 * 10 + 5 + 18
    for cid, c in rec.cameras.items():
        np.testing.assert_allclose(c.params[:2], gt.cameras[cid].params[:2], rtol=2e-3)
        assert not np.array_equal(c.params[4:], gt.cameras[cid].params[4:] * [1.2, 0.8, 1.3, 0.7])  # distortion moved


@pytest.mark.parametrize("model,params", [
    (scene.SIMPLE_RADIAL_FISHEYE, [500.0, 320, 240, 0.03]),
    (scene.RADIAL_FISHEYE, [500.0, 320, 240, 0.03, -0.004]),
    (scene.OPENCV_FISHEYE, [500.0, 510.0, 320, 240, 0.03, -0.004, 0.001, -0.0002])])
def test_fisheye_models_values_and_jacobians(model, params):
    # models_jacobian.h:726-942 on top of internal::FisheyeProjectionWithJac (:51-80)
    rng = np.random.default_rng(int(model))
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    pts = [np.array([rng.normal() * 1.5, rng.normal() * 1.5, 1 + rng.random()]) for _ in range(20)]
    pts.append(np.array([0.0, 0.0, 2.0]))          # on the optical axis: the r -> 0 branch
    for pt in pts:
        r0, Jpt, Jpose, Jpar = ba_oracle.reproj_error(model, pt, pose, params, [0, 0])
        np.testing.assert_allclose(r0, scene.img_from_cam(model, np.array(params), pt[None])[0], rtol=1e-12, atol=1e-10)
        def f(v):
            return ba_oracle.reproj_error(model, v[:3], pose, v[3:], [0, 0], want_jac=False)[0]
        x0 = np.concatenate([pt, params])
        J = np.zeros((2, len(x0)))
        for i in range(len(x0)):
            h = 1e-6 * max(1.0, abs(x0[i]))
            e = np.zeros(len(x0)); e[i] = h
            J[:, i] = (f(x0 + e) - f(x0 - e)) / (2 * h)
        np.testing.assert_allclose(Jpt, J[:, :3], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(Jpar, J[:, 3:], rtol=2e-5, atol=5e-5)


# Parameter vectors and the (u, v, w) grid of the reference's own model tests (sensor/models_test.cc:186-195,
# 333-417); analytic Jacobians against central differences as models_jacobian_test.cc does against autodiff.
_REF_MODEL_PARAMS = [
    (scene.FOV, [651.123, 655.123, 386.123, 511.123, 0.0]),
    (scene.FOV, [651.123, 655.123, 386.123, 511.123, 0.9]),
    (scene.FOV, [651.123, 655.123, 386.123, 511.123, 1e-6]),
    (scene.FOV, [651.123, 655.123, 386.123, 511.123, 1e-2]),
    (scene.SIMPLE_DIVISION, [651.123, 386.123, 511.123, 0.0]),
    (scene.SIMPLE_DIVISION, [651.123, 386.123, 511.123, 0.1]),
    (scene.SIMPLE_DIVISION, [651.123, 386.123, 511.123, -0.1]),
    (scene.DIVISION, [651.123, 655.123, 386.123, 511.123, 0.0]),
    (scene.DIVISION, [651.123, 655.123, 386.123, 511.123, 0.1]),
    (scene.DIVISION, [651.123, 655.123, 386.123, 511.123, -0.1]),
    (scene.SIMPLE_FISHEYE, [651.123, 386.123, 511.123]),
    (scene.FISHEYE, [651.123, 655.123, 386.123, 511.123]),
    (scene.EUCM, [651.123, 655.123, 386.123, 511.123, 0.56, 0.87]),
    (scene.EUCM, [400.0, 400.0, 400.0, 400.0, 0.88, 0.64]),
    (scene.EUCM, [651.123, 655.123, 386.123, 511.123, 0.0, 1.0]),
    (scene.EUCM, [651.123, 655.123, 386.123, 511.123, 0.5, 1.0]),
    # the parameter vectors of the reference's own model tests (sensor/models_test.cc:318-369)
    (scene.FULL_OPENCV, [651.123, 655.123, 386.123, 511.123, -0.471, 0.223, -0.001, 0.001, 0.001, 0.02, -0.02, 0.001]),
    (scene.THIN_PRISM_FISHEYE, [651.123, 655.123, 386.123, 511.123, -0.471, 0.223, -0.001, 0.001, 0.001, 0.02, -0.02, 0.001]),
    (scene.RAD_TAN_THIN_PRISM_FISHEYE, [651.123, 655.123, 386.123, 511.123, -0.0232, 0.0924, -0.0591, 0.003, 0.0048, -0.0009,
                                        0.0002, 0.0005, -0.0009, -0.0001, 0.00007, -0.00017]),   # models_test.cc:371-390
    (scene.EQUIRECTANGULAR, [1000.0, 500.0]),
]


@pytest.mark.parametrize("model,params", _REF_MODEL_PARAMS)
def test_more_camera_models_values_and_jacobians(model, params):
    """FOV (models_jacobian.h:627-724, all three branches of the distortion), SIMPLE_DIVISION / DIVISION
    (:88-113, 1291-1411), SIMPLE_FISHEYE / FISHEYE (:1190-1288), EUCM (:1413-1500), FULL_OPENCV (:498-625),
    THIN_PRISM_FISHEYE (:944-1047), RAD_TAN_THIN_PRISM_FISHEYE (:1049-1188), EQUIRECTANGULAR (:1502-1565)."""
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    params = np.array(params, np.float64)
    grid = [np.array([u, v, w]) for u in np.arange(-0.5, 0.51, 0.1) for v in np.arange(-0.5, 0.51, 0.1)
            for w in (0.5, 1.0, 2.0)]
    for pt in grid[::3] + [np.array([0.0, 0.0, 1.0])]:
        r0, Jpt, Jpose, Jpar = ba_oracle.reproj_error(model, pt, pose, params, [0, 0])
        if model == scene.FOV and params[4] ** 2 < 1e-4:   # small-omega series branch of the distortion
            f = params[4] ** 2 * (pt[0] ** 2 + pt[1] ** 2) / pt[2] ** 2 / 3.0 - params[4] ** 2 / 12.0 + 1.0
            want = np.array([params[0] * pt[0] / pt[2] * f + params[2], params[1] * pt[1] / pt[2] * f + params[3]])
        else:
            want = scene.img_from_cam(model, params, pt[None])[0]
        np.testing.assert_allclose(r0, want, rtol=1e-12, atol=1e-9)

        def f_(v):
            return ba_oracle.reproj_error(model, v[:3], pose, v[3:], [0, 0], want_jac=False)[0]
        x0 = np.concatenate([pt, params])
        J = np.zeros((2, len(x0)))
        for i in range(len(x0)):
            h = 1e-6 * max(1.0, abs(x0[i]))
            e = np.zeros(len(x0)); e[i] = h
            if model == scene.FOV and i == 7 and abs(abs(x0[i]) - 1e-2) < 1e-9:
                # omega^2 = 1e-4 is the branch point between the series and the closed form: stay on
                # the closed-form side (one-sided second-order difference)
                J[:, i] = (-3 * f_(x0) + 4 * f_(x0 + e) - f_(x0 + 2 * e)) / (2 * h)
                continue
            J[:, i] = (f_(x0 + e) - f_(x0 - e)) / (2 * h)
        np.testing.assert_allclose(Jpt, J[:, :3], rtol=5e-5, atol=5e-5)
        np.testing.assert_allclose(Jpar, J[:, 3:], rtol=5e-5, atol=2e-4)
    # the parameter layout the adapters use (FocalLengthIdxs / PrincipalPointIdxs / ExtraParamsIdxs)
    n = scene.MODEL_NUM_PARAMS[model]
    refinable = [] if model == scene.EQUIRECTANGULAR else list(range(n))  # (width, height) are metadata, never refined
    assert sorted(scene.MODEL_FOCAL_IDXS[model] + scene.MODEL_PP_IDXS[model] + scene.MODEL_EXTRA_IDXS[model]) == refinable


def test_division_and_eucm_reject_points_outside_their_domain():
    """No cheirality test for the division model: the discriminant decides (models_jacobian.h:97-101);
    EUCM rejects a non-positive denominator (:1437-1446). A rejected projection gives a zero residual
    and zero Jacobians (reprojection_error.h:96-116)."""
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    r, Jpt, _, _ = ba_oracle.reproj_error(scene.DIVISION, np.array([3.0, 3.0, 1.0]), pose,
                                          np.array([600.0, 600, 320, 240, 0.2]), [10.0, 20.0])
    assert np.all(r == 0) and np.all(Jpt == 0)          # w^2 - 4 rho^2 k < 0
    r, _, _, _ = ba_oracle.reproj_error(scene.DIVISION, np.array([0.1, 0.1, -1.0]), pose,
                                        np.array([600.0, 600, 320, 240, -0.1]), [0.0, 0.0])
    assert np.all(np.isfinite(r)) and np.any(r != 0)    # behind the camera but projectable: not rejected
    r, _, _, _ = ba_oracle.reproj_error(scene.EUCM, np.array([0.1, 0.1, -1.0]), pose,
                                        np.array([600.0, 600, 320, 240, 0.3, 1.0]), [5.0, 5.0])
    assert np.all(r == 0)


def test_constant_rig_from_world_rotation():
    """options.constant_rig_from_world_rotation (bundle_adjustment_ceres.cc:404-408,513-516): every
    variable rig_from_world keeps its rotation bit for bit (SubsetManifold over the quaternion), only
    translations -- minus the gauge coordinate of the second gauge frame -- points and intrinsics move."""
    gt, rec = _dataset(2, 5, 150, scene.SyntheticNoiseOptions(point2D_stddev=0.3, point3D_stddev=0.05,
                                                               rig_from_world_translation_stddev=0.03), seed=5)
    before = {i: img.cam_from_world.copy() for i, img in rec.images.items()}
    opt = est.BundleAdjustmentOptions(constant_rig_from_world_rotation=True)
    ba = est.BundleAdjuster(opt, _config(rec), rec, solve_fn=ba_oracle.solve_fn)
    fp = ba.problem_
    var = fp.pose_const == 0
    assert var.sum() == len(fp.poses) - 1
    codes = sorted(fp.pose_fixed_t[var].tolist())
    assert codes[-1] == 7 and codes.count(7) == var.sum() - 1 and 4 <= codes[0] <= 6   # one frame also holds a coordinate
    s = ba.Solve()
    assert s.IsSolutionUsable() and s.final_cost < 0.2 * s.initial_cost
    # tangent size: 3 per free frame, 2 for the second gauge frame, + points + 2 intrinsics per camera
    n_cam_params = int((fp.cam_const[:, :4] == 0).sum())
    assert s.num_effective_parameters == 3 * (var.sum() - 1) + 2 + 3 * int((fp.point_const == 0).sum()) + n_cam_params
    moved = 0
    for i, img in rec.images.items():
        assert np.array_equal(img.cam_from_world[:4], before[i][:4])       # rotations untouched
        moved += int(not np.array_equal(img.cam_from_world[4:], before[i][4:]))
    assert moved == len(rec.images) - 1


# ------------------------------------------------------------------------------------------------
# position priors (cost_functions/pose_prior.h:76-129, PosePriorBundleAdjuster)
# ------------------------------------------------------------------------------------------------

def _rand_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([q, rng.normal(size=3)])


def test_position_prior_residual_and_jacobians():
    """residual = prior + R^-1 t = prior - projection centre (pose_prior.h:88-92); rig variant on the composed
    pose (:112-124); Jacobians against differences along the manifold (quaternion Plus, translation)."""
    rng = np.random.default_rng(5)
    for _ in range(5):
        pose, sens, pos = _rand_pose(rng), _rand_pose(rng), rng.normal(size=3)
        r, Jp, _ = ba_oracle.position_prior(pos, pose)
        centre = -scene.quat_to_rot(pose[:4]).T @ pose[4:]
        np.testing.assert_allclose(r, pos - centre, atol=1e-13)
        comp = scene.rigid_compose(sens, pose)
        r2, Jr, Js = ba_oracle.position_prior(pos, pose, sens)
        np.testing.assert_allclose(r2, pos + scene.quat_to_rot(comp[:4]).T @ comp[4:], atol=1e-13)

        def plus(x, d):  # manifold Plus of a pose block: quaternion (x) R^3
            return np.concatenate([ba_oracle.quat_plus(x[:4], d[:3]), x[4:] + d[3:]])

        def tangent(J, x):  # ambient 3 x 7 -> tangent 3 x 6
            q = x[:4]
            PJ = np.array([[q[3], q[2], -q[1]], [-q[2], q[3], q[0]], [q[1], -q[0], q[3]], [-q[0], -q[1], -q[2]]])
            return np.concatenate([J[:, :4] @ PJ, J[:, 4:]], 1)
        h = 1e-6
        for which, Jamb in (("pose", Jr), ("sens", Js)):
            num = np.zeros((3, 6))
            for i in range(6):
                d = np.zeros(6)
                d[i] = h
                if which == "pose":
                    a = ba_oracle.position_prior(pos, plus(pose, d), sens, want_jac=False)[0]
                    b = ba_oracle.position_prior(pos, plus(pose, -d), sens, want_jac=False)[0]
                else:
                    a = ba_oracle.position_prior(pos, pose, plus(sens, d), want_jac=False)[0]
                    b = ba_oracle.position_prior(pos, pose, plus(sens, -d), want_jac=False)[0]
                num[:, i] = (a - b) / (2 * h)
            np.testing.assert_allclose(tangent(Jamb, pose if which == "pose" else sens), num, atol=1e-7)
        np.testing.assert_allclose(tangent(Jp, pose),
                                   tangent(ba_oracle.position_prior(pos, pose, np.array([0, 0, 0, 1, 0, 0, 0.0]))[1], pose),
                                   atol=1e-13)


def _prior_problem(seed=3, n_img=8, sigma=0.05, with_gauge=False):
    fp = est.FlatProblem.from_arrays(scene.synthesize_flat(n_img, 120, 4, seed=seed))
    rng = np.random.default_rng(seed)
    centres = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in fp.poses])
    fp.prior_pose = np.arange(n_img, dtype=np.int32)
    fp.prior_position = np.ascontiguousarray(centres + sigma * rng.normal(size=centres.shape))
    cov = np.diag([0.01, 0.02, 0.04]) + 0.002
    L = np.linalg.cholesky(np.linalg.inv(cov))          # cov^-1 = L L^T ; left sqrt information = L^T
    fp.prior_sqrt_info = np.ascontiguousarray(np.repeat(L.T[None], n_img, 0))
    if with_gauge:
        assert est.fix_gauge_two_cams(fp)
    return fp


# the gauge directions are held by a handful of priors only: weak curvature, so the inexact-Newton PCG
# (eta = 0.1) crawls; the reference solves problems of this size with DENSE_SCHUR, i.e. exactly
_EXACT = dict(eta=1e-10, max_linear_solver_iterations=500)


def test_position_priors_fix_the_gauge_and_pull_the_centres():
    """No gauge fixing: the seven gauge freedoms are constrained by the priors alone
    (bundle_adjustment_ceres.cc:925-936). 3 residuals per prior; the solve converges and every projection
    centre ends up close to its prior; the prior cost is part of the reported cost."""
    fp = _prior_problem()
    no_prior = _prior_problem()
    no_prior.prior_pose = None
    a = fp.copy()
    s = est.solve_flat(a, est.SolverOptions(max_num_iterations=60, gradient_tolerance=1e-10, **_EXACT), solve_fn=ba_oracle.solve_fn)
    assert s.num_residuals == 2 * len(fp.obs_pose) + 3 * 8
    assert s.IsSolutionUsable() and s.final_cost < s.initial_cost
    centres = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in a.poses])
    assert np.abs(centres - fp.prior_position).max() < 0.2
    # moving all priors by a common offset moves the whole solution (the priors own the gauge)
    b = fp.copy()
    b.prior_position = np.ascontiguousarray(b.prior_position + np.array([0.5, -0.25, 0.125]))
    est.solve_flat(b, est.SolverOptions(max_num_iterations=60, gradient_tolerance=1e-10, **_EXACT), solve_fn=ba_oracle.solve_fn)
    cb = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in b.poses])
    np.testing.assert_allclose(cb - centres, np.tile([0.5, -0.25, 0.125], (8, 1)), atol=1e-5)
    # a robust loss on the priors with one gross outlier: the outlier is down-weighted
    c = fp.copy()
    c.prior_position = c.prior_position.copy()
    c.prior_position[3] += 5.0
    c.prior_loss_type, c.prior_loss_scale = int(est.LossFunctionType.CAUCHY), 1.0
    est.solve_flat(c, est.SolverOptions(max_num_iterations=60, gradient_tolerance=1e-10, **_EXACT), solve_fn=ba_oracle.solve_fn)
    cc = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in c.poses])
    assert np.linalg.norm(cc[3] - c.prior_position[3]) > 4.0 and np.abs(np.delete(cc - c.prior_position, 3, 0)).max() < 0.3


def test_position_prior_cost_against_scipy():
    """The minimum of reprojection + prior cost, cross-checked with scipy.optimize.least_squares on the
    same residual vector (quaternions re-normalised inside the residual)."""
    from scipy.optimize import least_squares
    fp = _prior_problem(seed=9, n_img=5, sigma=0.02)
    a = fp.copy()
    s = est.solve_flat(a, est.SolverOptions(max_num_iterations=100, gradient_tolerance=1e-12, function_tolerance=0.0, **_EXACT),
                       solve_fn=ba_oracle.solve_fn)
    n_pose, n_pt = len(fp.poses), len(fp.points)
    f_idx = scene.MODEL_FOCAL_IDXS[int(fp.cam_model[0])] + scene.MODEL_EXTRA_IDXS[int(fp.cam_model[0])]

    def unpack(x):
        poses = x[:7 * n_pose].reshape(n_pose, 7).copy()
        poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)
        pts = x[7 * n_pose:7 * n_pose + 3 * n_pt].reshape(n_pt, 3)
        cams = fp.cams.copy()
        cams[:, f_idx] = x[7 * n_pose + 3 * n_pt:].reshape(len(cams), len(f_idx))
        return poses, pts, cams

    def resid(x):
        poses, pts, cams = unpack(x)
        out = []
        for o in range(len(fp.obs_pose)):
            pz = poses[fp.obs_pose[o]]
            pc = scene.quat_to_rot(pz[:4]) @ pts[fp.obs_point[o]] + pz[4:]
            ci = fp.obs_cam[o]
            out.append(scene.img_from_cam(int(fp.cam_model[ci]), cams[ci][:scene.MODEL_NUM_PARAMS[int(fp.cam_model[ci])]],
                                          pc[None])[0] - fp.obs_xy[o])
        for k in range(len(fp.prior_pose)):
            pz = poses[fp.prior_pose[k]]
            out.append(fp.prior_sqrt_info[k] @ (fp.prior_position[k] + scene.quat_to_rot(pz[:4]).T @ pz[4:]))
        return np.concatenate(out)
    x0 = np.concatenate([a.poses.ravel(), a.points.ravel(), a.cams[:, f_idx].ravel()])
    assert abs(0.5 * np.sum(resid(x0) ** 2) - s.final_cost) <= 1e-9 * s.final_cost
    sol = least_squares(resid, x0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=200)
    assert abs(sol.cost - s.final_cost) <= 1e-6 * s.final_cost


def test_pose_prior_bundle_adjuster_adapter():
    """CreatePosePriorBundleAdjuster (bundle_adjustment.cc:373-395, bundle_adjustment_ceres.cc:900-1085) through
    the oracle: a reconstruction in an arbitrary similarity frame + priors in the metric frame -> the result
    is in the metric frame (aligned, normalised, solved without a fixed gauge, de-normalised)."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_frames_per_rig=4, num_points3D=150), seed=41)
    gt = rec.copy()
    rng = np.random.default_rng(42)
    priors = [est.PosePrior(i, gt.ProjectionCenter(i) + 0.01 * rng.normal(size=3), 1e-4 * np.eye(3)) for i in gt.RegImageIds()]
    priors.append(est.PosePrior(gt.RegImageIds()[0]))           # no position: dropped
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.3, 0.02, 0.2), rec, seed=43)
    Rw = scene.quat_to_rot(np.array([0.1, -0.2, 0.3, 0.9]) / np.linalg.norm([0.1, -0.2, 0.3, 0.9]))
    rec.Transform(1.7, Rw, np.array([3.0, -2.0, 1.0]))        # away from the metric frame
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    opt = est.BundleAdjustmentOptions()
    opt.solver_options = est.SolverOptions(max_num_iterations=60, gradient_tolerance=1e-10, **_EXACT)
    ba = est.CreatePosePriorBundleAdjuster(opt, est.PosePriorBundleAdjustmentOptions(), cfg, priors, rec,
                                           solve_fn=ba_oracle.solve_fn)
    assert ba.use_prior_position_ and len(ba.problem_.prior_pose) == len(gt.RegImageIds())
    assert (ba.problem_.pose_fixed_t == -1).all()               # no gauge fixed: the priors own it
    s = ba.Solve()
    assert s.IsSolutionUsable() and s.num_residuals == 2 * len(ba.problem_.obs_pose) + 3 * len(gt.RegImageIds())
    err = [np.linalg.norm(rec.ProjectionCenter(i) - gt.ProjectionCenter(i)) for i in gt.RegImageIds()]
    assert max(err) < 0.05, max(err)
    pts = np.array([np.linalg.norm(rec.points3D[j].xyz - gt.points3D[j].xyz) for j in gt.points3D])
    assert np.median(pts) < 0.05
    # fewer than three priors: two-camera gauge, priors unused (:925-936)
    rec2 = gt.copy()
    ba2 = est.CreatePosePriorBundleAdjuster(opt, est.PosePriorBundleAdjustmentOptions(), cfg, priors[:2], rec2,
                                            solve_fn=ba_oracle.solve_fn)
    assert not ba2.use_prior_position_ and ba2.problem_.prior_pose is None
    assert (ba2.problem_.pose_const == 1).sum() >= 1


def test_dense_schur_tier_equals_tight_pcg():
    """linear_solver_type DENSE_SCHUR (reduced camera system formed + Cholesky, the reference's choice up to
    50 images, bundle_adjustment_ceres.cc:203-213) reaches the minimum the iterative tier reaches with
    near-exact linear solves, in far fewer LM iterations than the default inexact PCG needs on a problem
    whose gauge is held by priors only; AUTO picks it by image count."""
    fp = _prior_problem()
    a, b, c = fp.copy(), fp.copy(), fp.copy()
    so = dict(max_num_iterations=80, gradient_tolerance=1e-9)
    sa = est.solve_flat(a, est.SolverOptions(linear_solver_type=est.SOLVER_DENSE_SCHUR, **so), solve_fn=ba_oracle.solve_fn)
    sb = est.solve_flat(b, est.SolverOptions(**so, **_EXACT), solve_fn=ba_oracle.solve_fn)
    sc = est.solve_flat(c, est.SolverOptions(linear_solver_type=est.SOLVER_AUTO, **so), solve_fn=ba_oracle.solve_fn)
    assert sa.termination_type == est.BundleAdjustmentTerminationType.CONVERGENCE
    assert abs(sa.final_cost - sb.final_cost) <= 1e-9 * sb.final_cost
    np.testing.assert_allclose(a.poses, b.poses, atol=1e-6)
    assert sc.final_cost == sa.final_cost and sc.num_iterations == sa.num_iterations   # 8 images: AUTO = dense
    assert (sa.log_linear_iters[:sa.num_iterations] == 1).all()
    d = fp.copy()
    sd = est.solve_flat(d, est.SolverOptions(**so), solve_fn=ba_oracle.solve_fn)          # default: inexact PCG
    assert sd.num_iterations > 2 * sa.num_iterations


def test_reference_position_prior_functor_cases():
    """AbsolutePosePositionPriorCostFunctor.Nominal, AbsoluteRigPosePositionPriorCostFunctor.Nominal and
    CovarianceWeightedCostFunctor<...> (cost_functions/pose_prior_test.cc:42-107,200-222), restated."""
    rng = np.random.default_rng(0)
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    assert np.abs(ba_oracle.position_prior(np.zeros(3), ident)[0]).max() < 1e-6
    for _ in range(3):
        pose = _rand_pose(rng)
        position_in_world = -scene.quat_to_rot(pose[:4]).T @ pose[4:]          # Inverse(sensor_from_world).translation()
        np.testing.assert_allclose(ba_oracle.position_prior(np.zeros(3), pose)[0], -position_in_world, atol=1e-6)
        assert np.abs(ba_oracle.position_prior(position_in_world, pose)[0]).max() < 1e-6
        # rig variant: sensor_from_world = sensor_from_rig * rig_from_world
        sens, rig = _rand_pose(rng), _rand_pose(rng)
        assert np.abs(ba_oracle.position_prior(np.zeros(3), ident, ident)[0]).max() < 1e-6
        comp = scene.rigid_compose(sens, rig)
        pos = -scene.quat_to_rot(comp[:4]).T @ comp[4:]
        np.testing.assert_allclose(ba_oracle.position_prior(np.zeros(3), rig, sens)[0], -pos, atol=1e-6)
        assert np.abs(ba_oracle.position_prior(pos, rig, sens)[0]).max() < 1e-6
        # covariance 2 I: residual = -0.5 sqrt(2) * world_from_cam.translation() -- through the solver's weighting:
        # one prior, cost = 1/2 |L^T r0|^2 with cov^-1 = L L^T
        L = np.linalg.cholesky(np.linalg.inv(2 * np.eye(3)))
        np.testing.assert_allclose(L.T @ ba_oracle.position_prior(np.zeros(3), pose)[0],
                                   -0.5 * np.sqrt(2) * position_in_world, atol=1e-6)


def _reference_pose_prior_backend_case(solve_fn, gpu_index="-1"):
    """PosePriorBundleAdjusterBackendTest.Nominal (bundle_adjustment_test.cc:423-481): 1 rig x 1 camera x 7 frames,
    100 points, priors = ground-truth positions + N(0, 0.05), noise 0.5 px / 0.1 / 0.5 deg / 0.1."""
    gt = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=1, num_cameras_per_rig=1, num_frames_per_rig=7,
                                                               num_points3D=100), seed=0)
    rec = gt.copy()
    rng = np.random.default_rng(1)
    priors = [est.PosePrior(i, gt.ProjectionCenter(i) + 0.05 * rng.normal(size=3)) for i in gt.RegImageIds()]
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.1, 0.5, 0.1, 0.5), rec, seed=2)
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    opt = est.BundleAdjustmentOptions()
    opt.gpu_index = gpu_index
    ba = est.CreatePosePriorBundleAdjuster(opt, est.PosePriorBundleAdjustmentOptions(), cfg, priors, rec, solve_fn=solve_fn)
    assert ba.Options().backend == est.BundleAdjustmentBackend.MI355X and ba.Config().NumImages() == 7
    summary = ba.Solve()
    assert summary.IsSolutionUsable() and summary.num_residuals > 0
    # ReconstructionNear(gt, max_rotation_error_deg 0.1, max_proj_center_error 0.1): the matcher first aligns
    # the two worlds through the projection centres (scene/reconstruction_matchers.h:149-169)
    ids = gt.RegImageIds()
    tf = est.align_to_positions(np.array([rec.ProjectionCenter(i) for i in ids]), np.array([gt.ProjectionCenter(i) for i in ids]))
    rec = rec.copy()
    rec.Transform(*tf)
    for i in gt.RegImageIds():
        a, b = gt.images[i].cam_from_world, rec.images[i].cam_from_world
        R = scene.quat_to_rot(a[:4]).T @ scene.quat_to_rot(b[:4])
        assert np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))) < 0.1
        assert np.linalg.norm(gt.ProjectionCenter(i) - rec.ProjectionCenter(i)) < 0.1
    return summary


def test_reference_pose_prior_backend_case_oracle():
    _reference_pose_prior_backend_case(ba_oracle.solve_fn)


def test_explicit_reduced_camera_system_equals_operator_products():
    """ba_oracle's exact tier forms S = B + D^2 - E C^-1 E^T explicitly (per point, row blocks owned by one
    thread) and factors it with a blocked Cholesky; the round-2 formation (n_c implicit operator products,
    BAO_DENSE_BY_PRODUCTS=1) is the independent cross-check: mixed models, priors, the SPARSE_SCHUR name."""
    import os
    import ba_oracle
    for mixed, priors in ((False, False), ("three", False), (True, True)):
        d = scene.synthesize_flat(30, 1500, 6, seed=3, mixed_models=mixed, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
        fp = est.FlatProblem.from_arrays(d)
        assert est.fix_gauge_two_cams(fp)
        if priors:
            rng = np.random.default_rng(77)
            centres = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in fp.poses])
            fp.prior_pose = np.arange(len(fp.poses), dtype=np.int32)
            fp.prior_position = np.ascontiguousarray(centres + 0.02 * rng.normal(size=centres.shape))
            fp.prior_sqrt_info = np.ascontiguousarray(np.repeat((5.0 * np.eye(3))[None], len(fp.poses), 0))
        out = {}
        for name, env, lst in (("products", "1", est.SOLVER_DENSE_SCHUR), ("explicit", "0", est.SOLVER_DENSE_SCHUR),
                               ("sparse", "0", est.SOLVER_SPARSE_SCHUR)):
            os.environ["BAO_DENSE_BY_PRODUCTS"] = env
            try:
                a = fp.copy()
                s = est.solve_flat(a, est.SolverOptions(max_num_iterations=6, linear_solver_type=lst), solve_fn=ba_oracle.solve_fn)
            finally:
                os.environ.pop("BAO_DENSE_BY_PRODUCTS", None)
            out[name] = (s, a)
        np.testing.assert_allclose(out["explicit"][0].log_cost, out["products"][0].log_cost, rtol=1e-12)
        np.testing.assert_allclose(out["explicit"][1].poses, out["products"][1].poses, atol=1e-12)
        assert np.array_equal(out["sparse"][0].log_cost, out["explicit"][0].log_cost)
        assert out["sparse"][0].linear_solver_used == est.SOLVER_SPARSE_SCHUR
        assert out["explicit"][0].linear_solver_used == est.SOLVER_DENSE_SCHUR


def test_robust_alignment_to_pose_priors_rejects_an_outlier():
    """AlignReconstructionToPosePriors runs RANSAC (estimators/alignment.cc:240-299): one wild GPS prior among
    twelve must not skew the similarity the reconstruction is aligned with (round 2 used plain least squares
    over all priors). The outlier is rejected, the other priors are inliers, the frame is the inliers' frame."""
    rng = np.random.default_rng(5)
    src = rng.normal(size=(12, 3))
    Rz = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    dst = 2.0 * src @ Rz.T + np.array([1.0, -2.0, 0.5]) + 0.01 * rng.normal(size=(12, 3))
    dst[4] += np.array([30.0, -20.0, 10.0])  # the outlier
    plain = est.align_to_positions(src, dst)
    robust = est.align_to_positions_robust(src, dst, max_error=0.1)
    assert robust is not None
    s, R, t = robust
    assert abs(s - 2.0) < 0.02 and np.allclose(R, Rz, atol=0.02) and np.allclose(t, [1.0, -2.0, 0.5], atol=0.05)
    assert abs(plain[0] - 2.0) > 0.2 or not np.allclose(plain[2], [1.0, -2.0, 0.5], atol=0.5)  # least squares IS skewed
    clean = np.delete(np.arange(12), 4)
    ref = est.align_to_positions(src[clean], dst[clean])  # = the refit on the inlier set
    assert np.allclose(s, ref[0]) and np.allclose(R, ref[1]) and np.allclose(t, ref[2])
    # without outliers (noise well inside max_error) the robust estimate is the least-squares one
    dst2 = dst.copy()
    dst2[4] = 2.0 * src[4] @ Rz.T + np.array([1.0, -2.0, 0.5])
    a, b = est.align_to_positions(src, dst2), est.align_to_positions_robust(src, dst2, max_error=0.1)
    assert np.allclose(a[0], b[0]) and np.allclose(a[1], b[1]) and np.allclose(a[2], b[2])
    assert est.align_to_positions_robust(src[:2], dst[:2], max_error=0.1) is None


def test_adapter_auto_rule_uses_the_measured_gpu_thresholds():
    """CreateSolverOptions' rule (bundle_adjustment_ceres.cc:203-213) in the adapters, with this backend's GPU pair
    (the reference keeps one pair per device class, bundle_adjustment_ceres.h:68-71): DENSE_SCHUR up to 200 images,
    SPARSE_SCHUR up to 4000, ITERATIVE_SCHUR beyond -- options of BundleAdjustmentOptions like the reference's, resolved
    by the adapter on config.NumImages() before the flat C interface is called (the oracle stands in for it here)."""
    assert est.resolve_linear_solver(1) == est.resolve_linear_solver(200) == est.SOLVER_DENSE_SCHUR
    assert est.resolve_linear_solver(201) == est.resolve_linear_solver(1000) == est.resolve_linear_solver(4000) == est.SOLVER_SPARSE_SCHUR
    assert est.resolve_linear_solver(4001) == est.resolve_linear_solver(100000) == est.SOLVER_ITERATIVE_SCHUR
    tiers = [est.resolve_linear_solver(n) for n in range(1, 6000, 7)]
    assert tiers == sorted(tiers, key=[est.SOLVER_DENSE_SCHUR, est.SOLVER_SPARSE_SCHUR, est.SOLVER_ITERATIVE_SCHUR].index)   # monotone in the image count
    o = est.BundleAdjustmentOptions()
    assert (o.max_num_images_direct_dense_gpu_solver, o.max_num_images_direct_sparse_gpu_solver) == (200, 4000)
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_frames_per_rig=4, num_points3D=60), seed=5)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.5, 0.03, 0.5), rec, seed=6)
    used = []
    for dense, sparse, want in ((200, 4000, est.SOLVER_DENSE_SCHUR), (4, 4000, est.SOLVER_SPARSE_SCHUR), (4, 6, est.SOLVER_ITERATIVE_SCHUR)):
        cfg = est.BundleAdjustmentConfig()
        for i in rec.RegImageIds():
            cfg.AddImage(i)
        cfg.FixGauge(est.BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD)
        opt = est.BundleAdjustmentOptions(max_num_images_direct_dense_gpu_solver=dense, max_num_images_direct_sparse_gpu_solver=sparse)
        import copy
        ba = est.BundleAdjuster(opt, cfg, copy.deepcopy(rec), solve_fn=ba_oracle.solve_fn)
        s = ba.Solve()
        assert s.IsSolutionUsable() and ba.linear_solver_requested_ == want
        used.append(ba.linear_solver_used_)
    assert used == [est.SOLVER_DENSE_SCHUR, est.SOLVER_SPARSE_SCHUR, est.SOLVER_ITERATIVE_SCHUR]
