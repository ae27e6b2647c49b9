"""GPU parity tests: the HIP bundle-adjustment solve (through the C ABI) against the fp64
CPU oracle on the same flattened problems.

Tolerances (fp64 on both sides; the only differences are summation orders): the first LM
iterations follow the oracle's trajectory to 1e-9 relative cost, the final cost agrees to 1e-8
relative under a tight gradient tolerance, parameters to 1e-6; residual / parameter counts are
exact; constant blocks are bit-identical.
"""
import numpy as np
import pytest

import os

import ba_compare
import ba_oracle
from colmap_amd import estimators as est
from colmap_amd import scene

pytestmark = pytest.mark.gpu

TIGHT = dict(gradient_tolerance=1e-10, max_num_iterations=200)


FLOOR_MAX_OBS = 60000   # problems up to this size also run the perturbed oracle (tests/ba_compare.py)


def _both(fp, **so_kw):
    so = est.SolverOptions(**so_kw)
    a, b = fp.copy(), fp.copy()
    want = est.solve_flat(a, so, solve_fn=ba_oracle.solve_fn)
    got = est.solve_flat(b, so, gpu_index=0)
    if len(fp.obs_pose) <= FLOOR_MAX_OBS and so.max_num_iterations > 1:
        # the noise floor of THIS problem: the oracle against itself under a few-ulp perturbation of its arithmetic --
        # solved only when a comparison misses a base bar (tests/ba_compare.py)
        def floor():
            c = fp.copy()
            return ba_compare.diff(a, want, c, est.solve_flat(c, so, solve_fn=ba_oracle.solve_fn_fast))
        a.floor = floor
    return (a, want), (b, got)


def _assert_close(a, want, b, got, cost_rtol=1e-8, param_atol=1e-6, traj_rtol=1e-7, proj_atol=5e-5):
    """Cost, trajectory, poses, points; the intrinsics through the projections of the observed points (pixels). Every bar
    is max(base value, 10 x the measured floor of the problem): see tests/ba_compare.py."""
    d, bars = ba_compare.assert_solutions_close(a, want, b, got, getattr(a, "floor", None), cost_rtol=cost_rtol,
                                                param_atol=param_atol, traj_rtol=traj_rtol, proj_atol=proj_atol)
    if os.environ.get("COLMAP_AMD_TEST_PRINT_DIFFS"):
        print("\nDIFF", d, "\nBARS", bars)


def _flat(frames, points, track, seed, mixed=False, noise=None):
    noise = noise or scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0)
    return est.FlatProblem.from_arrays(scene.synthesize_flat(frames, points, track, seed=seed, mixed_models=mixed,
                                                            noise=noise))


@pytest.mark.parametrize("frames,points,track,mixed", [(6, 40, 4, True), (12, 300, 5, False), (40, 2000, 8, True)])
def test_solution_matches_oracle(frames, points, track, mixed):
    fp = _flat(frames, points, track, seed=frames, mixed=mixed)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    _assert_close(a, want, b, got)
    assert got.final_cost < 0.2 * got.initial_cost


def test_run_to_run_determinism():
    """All reductions use fixed trees (per-workgroup partials + single-workgroup finals); parameter
    blocks whose observation list fits one chunk accumulate without atomics."""
    fp = _flat(40, 3000, 8, seed=5)
    assert est.fix_gauge_two_cams(fp)
    runs = []
    for _ in range(2):
        b = fp.copy()
        s = est.solve_flat(b, est.SolverOptions(max_num_iterations=15), gpu_index=0)
        runs.append((b, s))
    assert runs[0][1].final_cost == runs[1][1].final_cost
    assert np.array_equal(runs[0][0].poses, runs[1][0].poses) and np.array_equal(runs[0][0].points, runs[1][0].points)
    assert np.array_equal(runs[0][1].log_linear_iters, runs[1][1].log_linear_iters)


def test_default_options_benchmark_noise():
    """COLMAP's default tolerances (gradient 1e-4, <= 100 iterations) on the benchmark's noise model
    (benchmark/runtime/bundle_adjustment.cc:76-81)."""
    fp = _flat(30, 1500, 6, seed=42)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp)
    assert got.termination_type == want.termination_type
    assert abs(got.final_cost - want.final_cost) <= 1e-6 * want.final_cost
    # iteration counts are not compared: near convergence the accept/reject decisions of an
    # inexact-Newton LM flip on round-off (the oracle itself changes count with its thread count)
    np.testing.assert_allclose(got.log_cost[:4], want.log_cost[:4], rtol=1e-7)


def test_constant_blocks_are_untouched_bitwise():
    fp = _flat(8, 120, 4, seed=7)
    assert est.fix_gauge_two_cams(fp)
    fp.pose_const[5] = 1
    fp.point_const[::7] = 1
    fp.cam_const[3, :] = 1
    orig = fp.copy()
    (a, want), (b, got) = _both(fp, **TIGHT)
    _assert_close(a, want, b, got)
    assert np.array_equal(b.poses[5], orig.poses[5]) and np.array_equal(b.poses[fp.pose_const == 1], orig.poses[fp.pose_const == 1])
    assert np.array_equal(b.points[::7], orig.points[::7])
    assert np.array_equal(b.cams[3], orig.cams[3])
    assert np.array_equal(b.cams[:, 1:3], orig.cams[:, 1:3])          # principal points never refined
    k = int(np.nonzero(fp.pose_fixed_t >= 0)[0][0])
    assert b.poses[k, 4 + fp.pose_fixed_t[k]] == orig.poses[k, 4 + fp.pose_fixed_t[k]]
    np.testing.assert_allclose(np.linalg.norm(b.poses[:, :4], axis=1), 1.0, atol=1e-12)


def test_shared_intrinsics_and_three_point_gauge():
    """One camera shared by every image: the intrinsics block of the Schur-Jacobi preconditioner
    couples all observations of a point (the FAQ's dense case, doc/faq.rst:626-632)."""
    d = scene.synthesize_flat(10, 200, 5, seed=11, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.03, 0.5))
    d["obs_cam"] = np.zeros_like(d["obs_cam"])
    d["cams"] = d["cams"][:1].copy()
    d["cam_model"] = d["cam_model"][:1].copy()
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_three_points(fp)
    # this configuration converges slowly for both solvers (inexact steps, eta = 0.1), so the
    # comparison is on the early trajectory, not on a converged minimum
    (a, want), (b, got) = _both(fp, max_num_iterations=25)
    assert got.num_residuals == want.num_residuals == 2000
    assert got.num_effective_parameters == want.num_effective_parameters
    np.testing.assert_allclose(got.log_cost[:5], want.log_cost[:5], rtol=1e-7)
    np.testing.assert_array_equal(got.log_linear_iters[:3], want.log_linear_iters[:3])
    assert abs(got.final_cost - want.final_cost) <= 0.02 * want.final_cost
    assert got.final_cost < 0.05 * got.initial_cost


def test_heavy_blocks_reduce_their_chunks_first():
    """A camera shared by many images is ONE block with thousands of chunk partials; such heavy blocks are summed by
    ba_cpart_heavy_reduce_kernel (a fixed tree over all threads of a workgroup) before the finalize kernels read them,
    instead of one thread walking them all. Forced here on a small shared-intrinsics problem (64 observations per chunk,
    every block with more than one chunk heavy): the default path's trajectory to rounding, the oracle's to 1e-7, the
    same PCG iteration counts -- for the iterative tier, whose every product ends in a finalize, and the exact tier."""
    d = scene.synthesize_flat(10, 200, 5, seed=11, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.03, 0.5))
    d["obs_cam"] = np.zeros_like(d["obs_cam"])
    d["cams"] = d["cams"][:1].copy()
    d["cam_model"] = d["cam_model"][:1].copy()
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_two_cams(fp)
    for tier in (est.SOLVER_ITERATIVE_SCHUR, est.SOLVER_DENSE_SCHUR):
        so = dict(max_num_iterations=8, linear_solver_type=tier)
        b0, s0 = _solve_env(fp, {}, **so)
        b1, s1 = _solve_env(fp, {"COLMAP_AMD_BA_CHUNK": "64", "COLMAP_AMD_BA_HEAVY_CHUNKS": "1"}, **so)
        a = fp.copy()
        want = est.solve_flat(a, est.SolverOptions(**so), solve_fn=ba_oracle.solve_fn)
        np.testing.assert_allclose(s1.log_cost, s0.log_cost, rtol=1e-11)
        np.testing.assert_array_equal(s1.log_linear_iters[:4], s0.log_linear_iters[:4])
        np.testing.assert_allclose(s1.log_cost[:5], want.log_cost[:5], rtol=1e-7)
        np.testing.assert_allclose(b1.points, b0.points, atol=1e-8)


def test_pair_terms_per_incidence_equal_per_observation():
    """The Schur-Jacobi pair terms of observations that share a block (shared intrinsics) computed per (point, block)
    incidence -- -(Ws C^-1 Ws^T - sum_o W_o C^-1 W_o^T), ba_pair_cross_kernel, what a single-GPU solve runs -- against the
    per-observation walk over all partners (ba_block_schur_cross_kernel, what sharded solves keep for their local
    pairs): the same blocks to rounding, the same PCG iteration counts. Three cameras shared by twelve images, some
    points constant, Cauchy loss."""
    d = scene.synthesize_flat(12, 400, 6, seed=71, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.03, 0.5))
    d["obs_cam"] = (d["obs_cam"] % 3).astype(np.int32)
    d["cams"] = d["cams"][:3].copy()
    d["cam_model"] = d["cam_model"][:3].copy()
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_two_cams(fp)
    fp.point_const[::9] = 1
    so = dict(max_num_iterations=10, loss_type=int(est.LossFunctionType.CAUCHY), loss_scale=2.0)
    b0, s0 = _solve_env(fp, {"COLMAP_AMD_BA_PAIR_INCIDENCES": "0"}, **so)
    b1, s1 = _solve_env(fp, {}, **so)
    np.testing.assert_allclose(s1.log_cost, s0.log_cost, rtol=1e-11)
    np.testing.assert_array_equal(s1.log_linear_iters, s0.log_linear_iters)
    np.testing.assert_allclose(b1.points, b0.points, atol=1e-9)
    np.testing.assert_allclose(b1.cams, b0.cams, rtol=1e-9, atol=1e-9)


def test_only_points_variable_and_only_cameras_variable():
    fp = _flat(6, 80, 4, seed=13)
    pts_only = fp.copy()
    pts_only.pose_const[:] = 1
    pts_only.cam_const[:] = 1
    (a, want), (b, got) = _both(pts_only, **TIGHT)
    _assert_close(a, want, b, got)
    cams_only = fp.copy()
    cams_only.point_const[:] = 1
    (a, want), (b, got) = _both(cams_only, **TIGHT)
    _assert_close(a, want, b, got)


def _config(rec, gauge=est.BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD):
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    cfg.FixGauge(gauge)
    return cfg


def test_backend_interface_reference_cases():
    """bundle_adjustment_test.cc:303-412 through CreateDefaultBundleAdjuster(backend=MI355X)."""
    opt = est.BundleAdjustmentOptions(gpu_index="0")
    assert opt.backend == est.BundleAdjustmentBackend.MI355X
    # MinimumTrackLength: 594 residuals
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=3, num_frames_per_rig=1, num_points3D=100,
                                                               num_points2D_without_point3D=0))
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(point2D_stddev=1), rec)
    idx = next(i for i, p in enumerate(rec.images[3].points2D) if p.HasPoint3D())
    rec.DeleteObservation(3, idx)
    o = est.BundleAdjustmentOptions(gpu_index="0", min_track_length=3)
    s = est.CreateDefaultBundleAdjuster(o, _config(rec), rec).Solve()
    assert s.IsSolutionUsable() and s.num_residuals == 594
    # ConstantPoints3D: 80 residuals, points bit-identical
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_frames_per_rig=1, num_points3D=20))
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(point2D_stddev=1), rec)
    orig = rec.copy()
    o = est.BundleAdjustmentOptions(gpu_index="0", refine_points3D=False)
    s = est.CreateDefaultBundleAdjuster(o, _config(rec, est.BundleAdjustmentGauge.UNSPECIFIED), rec).Solve()
    assert s.IsSolutionUsable() and s.num_residuals == 80
    for pid, pt in rec.points3D.items():
        assert np.array_equal(pt.xyz, orig.points3D[pid].xyz)
    # Nominal: usable solution near ground truth
    gt = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=1, num_frames_per_rig=10, num_points3D=200))
    rec = gt.copy()
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(point2D_stddev=0.5, point3D_stddev=0.1), rec)
    rng = np.random.default_rng(5)
    for i in rec.RegImageIds()[2:]:
        rec.images[i].cam_from_world[4:] += rng.normal(0, 0.1, 3)
    ba = est.CreateDefaultBundleAdjuster(opt, _config(rec), rec)
    assert ba.Config().NumImages() == 10 and ba.Options().backend == est.BundleAdjustmentBackend.MI355X
    s = ba.Solve()
    assert s.IsSolutionUsable() and s.num_residuals > 0
    for i in gt.RegImageIds():
        a, b = gt.images[i].cam_from_world, rec.images[i].cam_from_world
        R = scene.quat_to_rot(a[:4]).T @ scene.quat_to_rot(b[:4])
        assert np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))) < 0.1
        ca, cb = -scene.quat_to_rot(a[:4]).T @ a[4:], -scene.quat_to_rot(b[:4]).T @ b[4:]
        assert np.linalg.norm(ca - cb) < 0.1


def test_error_behaviour():
    fp = _flat(4, 20, 3, seed=1)
    fp.cam_model[0] = 18  # unknown model id
    with pytest.raises(RuntimeError, match="unsupported camera model"):
        est.solve_flat(fp, gpu_index=0)
    fp = _flat(4, 20, 3, seed=1)
    with pytest.raises(RuntimeError, match="gpu_index"):
        est.solve_flat(fp, gpu_index=99)
    with pytest.raises(ValueError):
        est.CreateDefaultBundleAdjuster(est.BundleAdjustmentOptions(backend=est.BundleAdjustmentBackend.CERES),
                                        est.BundleAdjustmentConfig(), scene.Reconstruction())


def test_iteration_callback_stops_with_the_last_accepted_state():
    """ba_options.iteration_callback = the reference's BundleAdjustmentIterationCallback
    (controllers/bundle_adjustment.cc:40-57,84-86): a callback that asks to terminate after iteration 3 ends the
    solve with USER_SUCCESS (estimators/bundle_adjustment.h:50-57: a usable solution), the parameter blocks hold
    iteration 3's accepted state -- the same bits a solve limited to three iterations leaves -- and the callback saw
    iterations 0..3 with the logged costs; BA_CALLBACK_ABORT gives USER_FAILURE."""
    seen = []

    def stop_at_3(sm):
        seen.append((sm.iteration, sm.cost, sm.step_is_successful, sm.linear_solver_iterations))
        return est.CALLBACK_TERMINATE if sm.iteration >= 3 else est.CALLBACK_CONTINUE

    a = _flat(12, 300, 5, seed=3)
    b = _flat(12, 300, 5, seed=3)
    sa = est.solve_flat(a, est.SolverOptions(iteration_callback=stop_at_3, **TIGHT), gpu_index=0)
    sb = est.solve_flat(b, est.SolverOptions(gradient_tolerance=1e-10, max_num_iterations=3), gpu_index=0)
    assert sa.termination_type == est.BundleAdjustmentTerminationType.USER_SUCCESS
    assert sa.num_iterations == 3 and [it for it, *_ in seen] == [0, 1, 2, 3]
    assert seen[0][1] == sa.initial_cost
    assert [c for _, c, *_ in seen[1:]] == list(sa.log_cost) and list(sa.log_cost) == list(sb.log_cost)
    assert sa.final_cost == sb.final_cost == sa.log_cost[-1]
    for name in ("poses", "cams", "points"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert sa.setup_seconds > 0.0 and sa.lm_seconds > 0.0
    c = _flat(12, 300, 5, seed=3)
    sc = est.solve_flat(c, est.SolverOptions(iteration_callback=lambda sm: est.CALLBACK_ABORT, **TIGHT), gpu_index=0)
    assert sc.termination_type == est.BundleAdjustmentTerminationType.USER_FAILURE and sc.num_iterations == 0
    assert np.array_equal(c.points, _flat(12, 300, 5, seed=3).points)   # nothing was accepted


def test_against_committed_golden_fixture():
    """tests/golden/ba_6x40.npz (oracle solution, committed) reproduced by the HIP solver."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_6x40.npz"))
    fp = est.FlatProblem(**{k: g[f"in_{k}"].copy() for k in ("poses", "cams", "cam_model", "points", "obs_pose", "obs_cam",
                                                           "obs_point", "obs_xy", "pose_const", "pose_fixed_t",
                                                           "cam_const", "point_const")})
    pad = est.CAM_STRIDE - fp.cams.shape[1]  # fixture written with 12 doubles per camera block
    fp.cams = np.ascontiguousarray(np.pad(fp.cams, ((0, 0), (0, pad))))
    fp.cam_const = np.ascontiguousarray(np.pad(fp.cam_const, ((0, 0), (0, pad)), constant_values=1))
    s = est.solve_flat(fp, est.SolverOptions(gradient_tolerance=1e-10, max_num_iterations=200), gpu_index=0)
    assert [s.num_residuals, s.num_effective_parameters] == list(g["counts"])
    assert abs(s.initial_cost - g["costs"][0]) <= 1e-12 * g["costs"][0]
    assert abs(s.final_cost - g["costs"][1]) <= 1e-8 * g["costs"][1]
    np.testing.assert_allclose(fp.points, g["out_points"], atol=1e-6)
    np.testing.assert_allclose(fp.poses, g["out_poses"], atol=1e-6)
    np.testing.assert_allclose(fp.cams[:, :g["out_cams"].shape[1]], g["out_cams"], rtol=1e-7, atol=1e-6)


def test_full_size_properties():
    """BASELINE.json config[3] shape (1000 cameras x 200k points, 2 M observations) through
    size-independent properties."""
    fp = _flat(1000, 200000, 10, seed=42)
    assert est.fix_gauge_two_cams(fp)
    fp.point_const[::1000] = 1
    orig = fp.copy()
    s = est.solve_flat(fp, est.SolverOptions(max_num_iterations=12), gpu_index=0)
    assert s.IsSolutionUsable()
    assert s.num_residuals == 2 * len(fp.obs_pose) == 4000000
    assert s.num_effective_parameters == 6 * 998 + 5 + 2 * 1000 + 3 * (200000 - 200)
    # accepted steps never increase the cost (monotonic trust region); the noise floor is reached
    assert np.all(np.diff(s.log_cost) <= 1e-9 * s.log_cost[:-1])
    assert s.final_cost < 0.01 * s.initial_cost
    # final cost ~ N_residuals/2 * sigma^2 for 1 px observation noise
    assert 0.3 < s.final_cost / (0.5 * s.num_residuals) < 1.2
    np.testing.assert_allclose(np.linalg.norm(fp.poses[:, :4], axis=1), 1.0, atol=1e-12)
    assert np.array_equal(fp.points[::1000], orig.points[::1000])           # constant points bit-identical
    assert np.array_equal(fp.poses[fp.pose_const == 1], orig.poses[fp.pose_const == 1])
    assert np.array_equal(fp.cams[:, 1:3], orig.cams[:, 1:3])               # principal points
    # idempotence: a second identical solve reproduces the same bits
    again = orig.copy()
    s2 = est.solve_flat(again, est.SolverOptions(max_num_iterations=12), gpu_index=0)
    assert s2.final_cost == s.final_cost and np.array_equal(again.poses, fp.poses)


def test_baseline_config3_cost_log_matches_oracle():
    """BASELINE.json config[3] itself (1000 cameras x 200k points, SIMPLE_RADIAL, 2 M observations): the
    HIP solve against the oracle on the first 9 LM iterations (what bench.py times the oracle on) --
    every logged cost to 1e-7 relative, the same PCG iteration counts, the same step decisions."""
    fp = _flat(1000, 200000, 10, seed=42)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, max_num_iterations=9)
    assert got.num_residuals == want.num_residuals == 4000000
    assert got.num_iterations == want.num_iterations and got.num_successful_steps == want.num_successful_steps
    assert abs(got.initial_cost - want.initial_cost) <= 1e-12 * want.initial_cost
    np.testing.assert_allclose(got.log_cost, want.log_cost, rtol=1e-7)
    np.testing.assert_array_equal(got.log_linear_iters[:4], want.log_linear_iters[:4])
    np.testing.assert_allclose(b.points, a.points, atol=1e-6)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)


def test_baseline_config3_shared_intrinsics_matches_oracle():
    """SURVEY 8(d)'s second variant of BASELINE.json config[3]: the same 1000 cameras x 200k points with ONE camera shared
    by all images (the case the reference's FAQ warns about, doc/faq.rst:626-632, and the common single-camera capture).
    The one intrinsics block is reached from all 2 M observations -- 4 000 chunk partials (ba_cpart_heavy_reduce_kernel),
    1.8 M observation pairs inside the block (ba_obs_w_kernel + ba_block_schur_cross_kernel) -- at the size where those
    paths carry real load: first 6 LM iterations against the oracle, costs 1e-7, the same PCG iteration counts."""
    d = scene.synthesize_flat(1000, 200000, 10, seed=42, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
    d["obs_cam"] = np.zeros_like(d["obs_cam"])
    d["cams"] = d["cams"][:1].copy()
    d["cam_model"] = d["cam_model"][:1].copy()
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, max_num_iterations=6)
    assert got.num_residuals == want.num_residuals == 4000000
    assert got.num_effective_parameters == want.num_effective_parameters == 3 * 200000 + 6 * 998 + 5 + 2
    assert got.num_iterations == want.num_iterations and got.num_successful_steps == want.num_successful_steps
    np.testing.assert_allclose(got.log_cost, want.log_cost, rtol=1e-7)
    np.testing.assert_array_equal(got.log_linear_iters[:4], want.log_linear_iters[:4])
    np.testing.assert_allclose(b.points, a.points, atol=1e-6)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)
    assert ba_compare.projection_diff_px(a, b) <= 5e-5


def test_baseline_config4_mixed_models_matches_oracle():
    """BASELINE.json config[4]'s ingredients at a tenth of its size (500 cameras x 200k points, track 10,
    thirds of SIMPLE_RADIAL / PINHOLE / OPENCV: the <8, 8> camera-block tier with three models in one
    launch) against the oracle, first 8 LM iterations."""
    fp = _flat(500, 200000, 10, seed=43, mixed="three")
    assert est.fix_gauge_two_cams(fp)
    assert set(np.unique(fp.cam_model)) == {scene.SIMPLE_RADIAL, scene.PINHOLE, scene.OPENCV}
    (a, want), (b, got) = _both(fp, max_num_iterations=8)
    assert got.num_residuals == want.num_residuals == 4000000
    assert got.num_effective_parameters == want.num_effective_parameters
    np.testing.assert_allclose(got.log_cost, want.log_cost, rtol=1e-7)
    np.testing.assert_allclose(b.points, a.points, atol=1e-6)
    assert ba_compare.projection_diff_px(a, b) <= 5e-5   # the intrinsics, through what they project (tests/ba_compare.py)


def test_baseline_config4_full_size_properties():
    """BASELINE.json config[4] at full size on ONE GPU (5000 cameras x 2 M points, 20 M observations,
    mixed SIMPLE_RADIAL / PINHOLE / OPENCV; ~7 GB of Jacobian in HBM) through size-independent
    properties: counts, monotone cost, the 1 px noise floor, constant blocks bit-identical."""
    fp = _flat(5000, 2000000, 10, seed=42, mixed="three")
    assert est.fix_gauge_two_cams(fp)
    fp.point_const[::10000] = 1
    orig_points, orig_cams = fp.points[::10000].copy(), fp.cams[:, 2:4].copy()
    s = est.solve_flat(fp, est.SolverOptions(max_num_iterations=10), gpu_index=0)
    assert s.IsSolutionUsable()
    assert s.num_residuals == 2 * len(fp.obs_pose) == 40000000
    n_sr, n_ph, n_cv = [(fp.cam_model == m).sum() for m in (scene.SIMPLE_RADIAL, scene.PINHOLE, scene.OPENCV)]
    assert s.num_effective_parameters == 6 * 4998 + 5 + 2 * n_sr + 2 * n_ph + 6 * n_cv + 3 * (2000000 - 200)
    assert np.all(np.diff(s.log_cost) <= 1e-9 * s.log_cost[:-1])
    assert s.final_cost < 0.01 * s.initial_cost
    assert 0.3 < s.final_cost / (0.5 * s.num_residuals) < 1.2
    np.testing.assert_allclose(np.linalg.norm(fp.poses[:, :4], axis=1), 1.0, atol=1e-12)
    assert np.array_equal(fp.points[::10000], orig_points)
    pp = np.where((fp.cam_model == scene.SIMPLE_RADIAL)[:, None], fp.cams[:, 1:3], fp.cams[:, 2:4])
    pp0 = np.where((fp.cam_model == scene.SIMPLE_RADIAL)[:, None], [[512.0, 384.0]], orig_cams)
    assert np.array_equal(pp, pp0)  # principal points never refined


def _solve_env(fp, env, **so_kw):
    """One solve with development switches of the library set (tests/switches.py; the library reads no environment)."""
    from switches import switches
    with switches(est.lib(), **env):
        b = fp.copy()
        return b, est.solve_flat(b, est.SolverOptions(**so_kw), gpu_index=0)


@pytest.mark.parametrize("mixed", [False, "three"])
def test_split_linearisation_is_bit_identical(mixed):
    """Point-side columns from their own p-order pass (ba_linearize_point_kernel, coalesced stores) against the
    c-order kernel scattering them through c2a: the same expressions, so the whole solve is bit-identical."""
    fp = _flat(40, 3000, 8, seed=5, mixed=mixed)
    assert est.fix_gauge_two_cams(fp)
    b0, s0 = _solve_env(fp, {"COLMAP_AMD_BA_SPLIT_LINEARIZE": "0"}, max_num_iterations=12)
    b1, s1 = _solve_env(fp, {"COLMAP_AMD_BA_SPLIT_LINEARIZE": "1"}, max_num_iterations=12)
    assert np.array_equal(s0.log_cost, s1.log_cost) and s0.final_cost == s1.final_cost
    assert np.array_equal(b0.poses, b1.poses) and np.array_equal(b0.points, b1.points) and np.array_equal(b0.cams, b1.cams)


@pytest.mark.parametrize("model", ["SIMPLE_RADIAL", "PINHOLE", "SIMPLE_PINHOLE"])
def test_plain_linearisation_is_bit_identical(model):
    """One camera model, no rig observations, trivial loss: the linearisation kernels instantiated for that model (the
    model switch, the rig branch and the loss corrector folded away at compile time -- 92 instead of 168 registers)
    against the generic kernels: the same expressions, so the whole solve is bit-identical."""
    fp = _flat(40, 3000, 8, seed=7)
    if model == "PINHOLE":
        fp.cam_model[:] = scene.PINHOLE
        fp.cams[:, :4] = [1280.0, 1280.0, 512.0, 384.0]
    elif model == "SIMPLE_PINHOLE":
        fp.cam_model[:] = scene.SIMPLE_PINHOLE
        fp.cams[:, :4] = [1280.0, 512.0, 384.0, 0.0]
    assert est.fix_gauge_two_cams(fp)
    b0, s0 = _solve_env(fp, {"COLMAP_AMD_BA_PLAIN_LINEARIZE": "0"}, max_num_iterations=12)
    b1, s1 = _solve_env(fp, {"COLMAP_AMD_BA_PLAIN_LINEARIZE": "1"}, max_num_iterations=12)
    assert s1.final_cost < 0.1 * s1.initial_cost
    assert np.array_equal(s0.log_cost, s1.log_cost) and s0.final_cost == s1.final_cost
    assert np.array_equal(b0.poses, b1.poses) and np.array_equal(b0.points, b1.points) and np.array_equal(b0.cams, b1.cams)


def test_fused_pcg_kernel_matches_separate_kernels():
    """ba_pcg_fused_kernel (one single-workgroup kernel around the three streaming kernels of a product) against
    the separate precondition / direction / finalize / update kernels: same algorithm, different summation
    trees -- cost log to 1e-12, identical PCG iteration counts."""
    fp = _flat(40, 3000, 8, seed=6)
    assert est.fix_gauge_two_cams(fp)
    b0, s0 = _solve_env(fp, {"COLMAP_AMD_BA_PCG_FUSED": "0"}, max_num_iterations=12)
    b1, s1 = _solve_env(fp, {"COLMAP_AMD_BA_PCG_FUSED": "1"}, max_num_iterations=12)
    np.testing.assert_allclose(s1.log_cost, s0.log_cost, rtol=1e-12)
    np.testing.assert_array_equal(s1.log_linear_iters, s0.log_linear_iters)
    np.testing.assert_allclose(b1.points, b0.points, atol=1e-10)


@pytest.mark.parametrize("frames,points,track,mixed", [(40, 2000, 8, True), (200, 20000, 10, False)])
def test_fp32_operator_reaches_the_fp64_solution(frames, points, track, mixed):
    """ba_options.operator_precision = BA_OPERATOR_F32: the inexact inner CG solve streams fp32 copies of the
    Jacobian columns. Contract (include/colmap_amd_ba.h): the converged solution is the fp64 oracle's -- final
    cost to 1e-8 relative, parameters to 1e-6 under a tight gradient tolerance -- and the early trajectory
    follows it to 1e-5 relative (instead of 1e-7 with the fp64 operator)."""
    fp = _flat(frames, points, track, seed=frames, mixed=mixed)
    assert est.fix_gauge_two_cams(fp)
    a, b = fp.copy(), fp.copy()
    want = est.solve_flat(a, est.SolverOptions(**TIGHT), solve_fn=ba_oracle.solve_fn)
    got = est.solve_flat(b, est.SolverOptions(operator_precision=est.OPERATOR_F32, **TIGHT), gpu_index=0)
    assert got.termination_type == want.termination_type
    assert abs(got.final_cost - want.final_cost) <= 1e-8 * want.final_cost, (got.final_cost, want.final_cost)
    n = min(4, len(want.log_cost), len(got.log_cost))
    np.testing.assert_allclose(got.log_cost[:n], want.log_cost[:n], rtol=1e-5)
    np.testing.assert_allclose(b.points, a.points, atol=1e-6)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)
    assert ba_compare.projection_diff_px(a, b) <= 5e-5   # the intrinsics, through what they project (tests/ba_compare.py)


def test_tracks_longer_than_a_tile():
    """Tracks with more observations than the LDS tile (512) take the untiled point pass."""
    fp = _flat(600, 6, 600, seed=9, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.02, 0.5))
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, max_num_iterations=8)
    assert got.num_residuals == want.num_residuals == 2 * 6 * 600
    np.testing.assert_allclose(got.log_cost[:4], want.log_cost[:4], rtol=1e-7)
    assert abs(got.final_cost - want.final_cost) <= 1e-5 * want.final_cost


def _with_priors(fp):
    """position priors on every pose (on top of the gauge): exercises the rank-0 prior terms of a sharded solve"""
    rng = np.random.default_rng(77)
    centres = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in fp.poses])
    fp.prior_pose = np.arange(len(fp.poses), dtype=np.int32)
    fp.prior_position = np.ascontiguousarray(centres + 0.02 * rng.normal(size=centres.shape))
    fp.prior_sqrt_info = np.ascontiguousarray(np.repeat((5.0 * np.eye(3))[None], len(fp.poses), 0))
    return fp


def _sharded_problem(priors=False, shared=False):
    if shared:  # three cameras shared by twelve images: observation pairs of a point inside one intrinsics block,
        # on DIFFERENT ranks under image sharding (the cross terms a rank cannot see drop out of its preconditioner)
        d = scene.synthesize_flat(12, 300, 5, seed=21, noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
        d["obs_cam"] = (d["obs_cam"] % 3).astype(np.int32)
        d["cams"] = d["cams"][:3].copy()
        d["cam_model"] = d["cam_model"][:3].copy()
        fp = est.FlatProblem.from_arrays(d)
    else:
        fp = _flat(12, 300, 5, seed=21, mixed=True)
    assert est.fix_gauge_two_cams(fp)
    if priors:
        _with_priors(fp)
    return fp


def _sharded_worker(rank, world, port, backend, q, sharding=0, priors=False, shared=False, solver=0):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fp = _sharded_problem(priors, shared)
        comm = est.Communicator(backend, gpu_index=0, sharding=sharding)
        s = est.solve_flat(fp, est.SolverOptions(linear_solver_type=solver, **TIGHT), gpu_index=0, comm=comm)
        import ctypes as C
        pl, sw = C.c_int64(), C.c_int64()
        est.lib().ba_last_pcg_loops(C.byref(pl), C.byref(sw))
        q.put((rank, s.final_cost, s.num_residuals, s.num_iterations, fp.poses.copy(), fp.points.copy(), comm.calls,
               None if s.log_linear_iters is None else np.asarray(s.log_linear_iters).copy(), (pl.value, sw.value)))
        comm.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sharding,priors", [(est.SHARD_BY_IMAGE, False), (est.SHARD_BY_POINT, False),
                                             (est.SHARD_BY_IMAGE, True), (est.SHARD_BY_POINT, True)])
def test_two_rank_sharded_solve_matches_single_gpu(sharding, priors):
    """Image sharding / point sharding with the sum-over-ranks callback (gloo): two processes share
    GPU 0, each linearises only its own observations; the solution equals the single-rank solve.
    Point sharding moves only camera-space vectors (fewer, smaller all-reduces)."""
    import socket
    import torch.multiprocessing as mp
    fp = _flat(12, 300, 5, seed=21, mixed=True)
    assert est.fix_gauge_two_cams(fp)
    if priors:
        _with_priors(fp)
    single = fp.copy()
    s1 = est.solve_flat(single, est.SolverOptions(**TIGHT), gpu_index=0)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, "callback", q, sharding, priors)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, n0, it0, poses0, pts0, calls0, lin0, loops0), (r1, c1, n1, it1, poses1, pts1, calls1, lin1, loops1) = res
    assert c0 == c1 and np.array_equal(poses0, poses1) and np.array_equal(pts0, pts1)   # ranks agree bitwise
    assert n0 == n1 == s1.num_residuals
    assert calls0 == calls1 > 0
    if sharding == est.SHARD_BY_POINT and not priors:
        # the pipelined PCG loop (host one iteration behind, stopping test on the device; the one collective per
        # iteration is the all-reduce of the camera-space vector) on every rank, and the single-rank iteration counts
        assert loops0 == loops1 and loops0[0] > 0 and loops0[1] == 0, (loops0, loops1)
        # (the first dozen LM iterations: near the 1e-10 gradient the counts differ by summation order)
        k = min(12, s1.num_iterations, it0)
        assert np.array_equal(lin0, lin1) and np.array_equal(lin0[:k], np.asarray(s1.log_linear_iters)[:k]), (lin0, s1.log_linear_iters)
    else:
        assert loops0 == loops1 and loops0[0] == 0
    assert abs(c0 - s1.final_cost) <= 1e-9 * s1.final_cost
    np.testing.assert_allclose(pts0, single.points, atol=1e-7)
    np.testing.assert_allclose(poses0, single.poses, atol=1e-7)


def _run_sharded(world, sharding, priors=False, shared=False, solver=0):
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, "callback", q, sharding, priors, shared, solver))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("sharding", [est.SHARD_BY_IMAGE, est.SHARD_BY_POINT])
def test_three_rank_sharded_solve_with_shared_intrinsics(sharding):
    """World size 3 and intrinsics blocks shared across ranks (12 images, 3 cameras): under image sharding the
    observation pairs of a point that sit on different ranks are invisible to any one rank; their part of the
    Schur-Jacobi blocks comes from the all-reduced W = J_b^T J_p products (ba_inc_* kernels). The sharded solve then has
    the single-GPU preconditioner: the same CG iteration counts in every LM iteration, the same solution; all ranks
    hold identical bits."""
    single = _sharded_problem(shared=True)
    s1 = est.solve_flat(single, est.SolverOptions(**TIGHT), gpu_index=0)
    res = _run_sharded(3, sharding, shared=True)
    for r in res[1:]:
        assert r[1] == res[0][1] and np.array_equal(r[4], res[0][4]) and np.array_equal(r[5], res[0][5])
    assert res[0][2] == s1.num_residuals
    assert abs(res[0][1] - s1.final_cost) <= 1e-8 * s1.final_cost
    np.testing.assert_allclose(res[0][5], single.points, atol=1e-6)
    np.testing.assert_allclose(res[0][4], single.poses, atol=1e-6)
    # same preconditioner => same CG trajectory: the iteration counts of the LM iterations equal the single-rank
    # solve's while the iterates are away from the rounding floor (the last LM iterations of a solve driven to a
    # 1e-10 gradient differ by summation order, for point sharding too)
    k = min(12, s1.num_iterations, res[0][3])
    print("CG iterations per LM iteration, single:", np.asarray(s1.log_linear_iters)[:s1.num_iterations].tolist(),
          "sharded:", res[0][7][:res[0][3]].tolist())
    assert np.array_equal(res[0][7][:k], np.asarray(s1.log_linear_iters)[:k])


@pytest.mark.parametrize("sharding", [est.SHARD_BY_IMAGE, est.SHARD_BY_POINT])
def test_two_rank_sharded_exact_tier(sharding):
    """DENSE_SCHUR in a sharded solve: point sharding sums the ranks' explicitly formed partial systems (every
    observation pair of a point is on one rank), image sharding falls back to the operator-product formation
    (every product is all-reduced; camera-side dimension <= 1024). Both reproduce the single-rank exact solve."""
    single = _sharded_problem()
    s1 = est.solve_flat(single, est.SolverOptions(linear_solver_type=est.SOLVER_DENSE_SCHUR, **TIGHT), gpu_index=0)
    assert s1.linear_solver_used == est.SOLVER_DENSE_SCHUR
    res = _run_sharded(2, sharding, solver=est.SOLVER_DENSE_SCHUR)
    assert res[0][1] == res[1][1] and np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])
    assert abs(res[0][1] - s1.final_cost) <= 1e-9 * s1.final_cost
    np.testing.assert_allclose(res[0][5], single.points, atol=1e-7)
    np.testing.assert_allclose(res[0][4], single.poses, atol=1e-7)


def test_rccl_transport_world_size_one():
    """RCCL communicator plumbing on the one GPU available here (all-reduce over one rank is the
    identity): bit-identical to the plain solve."""
    fp = _flat(10, 200, 5, seed=23)
    assert est.fix_gauge_two_cams(fp)
    a, b = fp.copy(), fp.copy()
    s0 = est.solve_flat(a, est.SolverOptions(max_num_iterations=20), gpu_index=0)
    comm = est.Communicator("rccl", gpu_index=0)
    try:
        s1 = est.solve_flat(b, est.SolverOptions(max_num_iterations=20), gpu_index=0, comm=comm)
    finally:
        comm.close()
    assert s0.final_cost == s1.final_cost and np.array_equal(a.poses, b.poses) and np.array_equal(a.points, b.points)


# ------------------------------------------------------------------------------------------------
# robust losses, rigs with a constant sensor_from_rig, RADIAL
# ------------------------------------------------------------------------------------------------

def _adapter_problem(rec, gauge=est.BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD, **opt_kw):
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    cfg.FixGauge(gauge)
    return est.flatten(est.BundleAdjustmentOptions(**opt_kw), cfg, rec)


@pytest.mark.parametrize("loss,scale", [(est.LossFunctionType.SOFT_L1, 1.0), (est.LossFunctionType.CAUCHY, 1.0),
                                        (est.LossFunctionType.HUBER, 2.0)])
def test_robust_losses_match_oracle(loss, scale):
    """ceres::LossFunction + Corrector on every residual block (bundle_adjustment_ceres.cc:66-80):
    same trajectory and optimum as the oracle, with 5 % gross outliers in the observations."""
    fp = _flat(12, 300, 5, seed=3)
    rng = np.random.default_rng(0)
    bad = rng.random(len(fp.obs_xy)) < 0.05
    fp.obs_xy[bad] += rng.normal(0, 60, (int(bad.sum()), 2))
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, loss_type=int(loss), loss_scale=scale, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    _assert_close(a, want, b, got)
    # and the loss is actually in effect: the robust cost is far below the squared cost
    (_, l2), _ = _both(fp, max_num_iterations=1)
    assert got.initial_cost < 0.5 * l2.initial_cost


def test_rig_frames_match_oracle():
    """Two rigs of three cameras, five frames each: a frame's pose block is observed through three
    cameras (three c-order runs per block) and two of them through a constant sensor_from_rig."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=2, num_cameras_per_rig=3, num_frames_per_rig=5, num_points3D=200,
        num_points2D_without_point3D=0), seed=5)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.5, 0.05, 0.5), rec, seed=6)
    fp = _adapter_problem(rec, refine_sensor_from_rig=False)
    assert fp.sensors is not None and len(fp.sensors) == 4 and len(fp.poses) == 10
    (a, want), (b, got) = _both(fp, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    assert want.num_residuals == 2 * len(fp.obs_pose)
    _assert_close(a, want, b, got)
    assert got.final_cost < 0.2 * got.initial_cost
    assert np.array_equal(b.sensors, fp.sensors)       # constant input


def test_variable_sensor_from_rig_matches_oracle():
    """refine_sensor_from_rig (the reference's default): every non-reference sensor_from_rig is a
    6-dimensional block of its own (RigReprojErrorCostFunctor, reprojection_error.h:344-384; block
    kind 2: own Jacobian columns, Gram block, PCG vector entries). Same optimum and trajectory as the
    oracle, the sensor blocks move and stay unit quaternions, plus a robust loss on top."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=2, num_cameras_per_rig=3, num_frames_per_rig=5, num_points3D=200,
        num_points2D_without_point3D=0), seed=5)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.5, 0.05, 0.5), rec, seed=6)
    for rig in rec.rigs.values():
        for cid in rig.sensors:
            rig.sensors[cid] = rig.sensors[cid] + np.array([0, 0, 0, 0, 0.03, -0.02, 0.01])
    rec.UpdateCamFromWorld()
    fp = _adapter_problem(rec)
    assert fp.sensor_const.tolist() == [0, 0, 0, 0]
    (a, want), (b, got) = _both(fp, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    assert got.num_effective_parameters == want.num_effective_parameters
    _assert_close(a, want, b, got)
    np.testing.assert_allclose(b.sensors, a.sensors, atol=1e-6)
    assert not np.array_equal(b.sensors, fp.sensors)
    np.testing.assert_allclose(np.linalg.norm(b.sensors[:, :4], axis=1), 1.0, atol=1e-12)
    (a, want), (b, got) = _both(fp, loss_type=int(est.LossFunctionType.CAUCHY), loss_scale=1.0, **TIGHT)
    _assert_close(a, want, b, got, cost_rtol=1e-7, param_atol=1e-5, traj_rtol=1e-5)
    np.testing.assert_allclose(b.sensors, a.sensors, atol=1e-5)
    # a constant frame observed through a variable sensor (the "rare" case of :792-795)
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    cfg.FixGauge(est.BundleAdjustmentGauge.THREE_POINTS)
    cfg.SetConstantRigFromWorldPose(next(iter(rec.frames)))
    fp2 = est.flatten(est.BundleAdjustmentOptions(), cfg, rec)
    (a, want), (b, got) = _both(fp2, **TIGHT)
    _assert_close(a, want, b, got)
    np.testing.assert_allclose(b.sensors, a.sensors, atol=1e-6)


def test_radial_model_matches_oracle():
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=3, num_frames_per_rig=4, num_points3D=250, camera_model_id=scene.RADIAL,
        camera_params=(1280.0, 512.0, 384.0, 0.05, -0.01)), seed=7)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.5, 0.05, 0.5), rec, seed=8)
    fp = _adapter_problem(rec)
    assert (fp.cam_model == scene.RADIAL).all() and (fp.cam_const[:, :5] == [0, 1, 1, 0, 0]).all()
    (a, want), (b, got) = _both(fp, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    _assert_close(a, want, b, got)
    # five variable intrinsics (principal point refined too): the wide <KD, BD> = <8, 8> kernels
    fp5 = _adapter_problem(rec, refine_principal_point=True)
    assert (fp5.cam_const[:, :5] == 0).all()
    (a, want), (b, got) = _both(fp5, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    assert got.num_effective_parameters == want.num_effective_parameters
    # principal point + two radial terms are weakly observable here: the inexact-Newton steps of the
    # two implementations may stop PCG one iteration apart, so only the optimum is compared tightly
    _assert_close(a, want, b, got, cost_rtol=1e-7, param_atol=1e-5, traj_rtol=1e-3)


def test_opencv_model_matches_oracle():
    """OPENCV (8 parameters; fx fy k1 k2 p1 p2 variable by default, all 8 with the principal point):
    camera blocks wider than the pose blocks, mixed with SIMPLE_RADIAL cameras in one problem."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=4, num_frames_per_rig=4, num_points3D=300, camera_model_id=scene.OPENCV,
        camera_params=(1280.0, 1290.0, 512.0, 384.0, 0.05, -0.01, 0.001, -0.002)), seed=9)
    # two of the four cameras become SIMPLE_RADIAL: blocks of width 6 and 2 side by side
    for cid in (2, 4):
        rec.cameras[cid].model_id = scene.SIMPLE_RADIAL
        rec.cameras[cid].params = np.array([1280.0, 512.0, 384.0, 0.05])
    for iid, img in rec.images.items():   # re-project the observations of the changed cameras
        cam = rec.cameras[img.camera_id]
        if cam.model_id != scene.SIMPLE_RADIAL:
            continue
        R, t = scene.quat_to_rot(img.cam_from_world[:4]), img.cam_from_world[4:]
        for p2 in img.points2D:
            if p2.HasPoint3D():
                p2.xy = scene.img_from_cam(cam.model_id, cam.params, (R @ rec.points3D[p2.point3D_id].xyz + t)[None])[0]
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.5, 0.05, 0.5), rec, seed=10)
    for pp in (False, True):
        fp = _adapter_problem(rec, refine_principal_point=pp)
        nvar = (fp.cam_const[:, :8] == 0).sum(1)
        assert sorted(nvar.tolist()) == ([2, 2, 6, 6] if not pp else [4, 4, 8, 8])
        (a, want), (b, got) = _both(fp, **TIGHT)
        assert want.IsSolutionUsable() and got.IsSolutionUsable()
        assert got.num_effective_parameters == want.num_effective_parameters
        _assert_close(a, want, b, got, cost_rtol=1e-7, param_atol=1e-5, traj_rtol=1e-3)
        assert got.final_cost < 0.2 * got.initial_cost


@pytest.mark.parametrize("model,params", [
    (scene.SIMPLE_RADIAL_FISHEYE, (900.0, 512.0, 384.0, 0.03)),
    (scene.RADIAL_FISHEYE, (900.0, 512.0, 384.0, 0.03, -0.004)),
    (scene.OPENCV_FISHEYE, (900.0, 910.0, 512.0, 384.0, 0.03, -0.004, 0.001, -0.0002))])
def test_fisheye_models_match_oracle(model, params):
    _model_matches_oracle(model, params)


@pytest.mark.parametrize("model,params", [
    (scene.FOV, (900.0, 910.0, 512.0, 384.0, 0.6)),
    (scene.SIMPLE_DIVISION, (900.0, 512.0, 384.0, -0.05)),
    (scene.DIVISION, (900.0, 910.0, 512.0, 384.0, -0.05)),
    (scene.SIMPLE_FISHEYE, (900.0, 512.0, 384.0)),
    (scene.FISHEYE, (900.0, 910.0, 512.0, 384.0)),
    (scene.EUCM, (900.0, 910.0, 512.0, 384.0, 0.56, 0.87)),
    (scene.FULL_OPENCV, (900.0, 910.0, 512.0, 384.0, -0.05, 0.02, -0.001, 0.001, 0.001, 0.02, -0.02, 0.001)),
    (scene.THIN_PRISM_FISHEYE, (900.0, 910.0, 512.0, 384.0, -0.05, 0.02, -0.001, 0.001, 0.001, 0.02, -0.02, 0.001)),
    (scene.RAD_TAN_THIN_PRISM_FISHEYE, (900.0, 910.0, 512.0, 384.0, -0.0232, 0.0924, -0.0591, 0.003, 0.0048, -0.0009,
                                        0.0002, 0.0005, -0.0009, -0.0001, 0.00007, -0.00017)),
    (scene.EQUIRECTANGULAR, (1024.0, 768.0))])
def test_more_camera_models_match_oracle(model, params):
    """FOV, SIMPLE_DIVISION / DIVISION, SIMPLE_FISHEYE / FISHEYE, EUCM (models_jacobian.h:627-724,
    1190-1500), and the two 12-parameter models FULL_OPENCV / THIN_PRISM_FISHEYE (:498-625, 944-1047), RAD_TAN_THIN_PRISM_FISHEYE (:1049-1188), EQUIRECTANGULAR (:1502-1565; third
    <KD, BD> = <12, 12> kernel tier): the HIP linearisation against the oracle's through full solves."""
    _model_matches_oracle(model, params)


def test_constant_rig_from_world_rotation_matches_oracle():
    """Pose blocks whose rotation is held (3- and 2-dimensional translation tangents): the HIP solve
    follows the oracle and leaves every quaternion bit-identical."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_frames_per_rig=5, num_points3D=200),
                                  seed=31)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.03, 0.0, 0.05, 0.5), rec, seed=32)
    fp = _adapter_problem(rec, constant_rig_from_world_rotation=True)
    assert (fp.pose_fixed_t[fp.pose_const == 0] >= est.POSE_ROT_CONST).all()
    (a, want), (b, got) = _both(fp, **TIGHT)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    _assert_close(a, want, b, got, cost_rtol=1e-8, param_atol=1e-6)
    assert np.array_equal(b.poses[:, :4], fp.poses[:, :4])
    assert not np.array_equal(b.poses[:, 4:], fp.poses[:, 4:])


def _model_matches_oracle(model, params):
    """Equidistant fisheye projection + radial polynomial (models_jacobian.h:51-80,726-942)."""
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(
        num_rigs=3, num_frames_per_rig=4, num_points3D=250, camera_model_id=model, camera_params=params), seed=21)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.5, 0.05, 0.5), rec, seed=22)
    fp = _adapter_problem(rec)
    assert (fp.cam_model == model).all()
    (a, want), (b, got) = _both(fp, gradient_tolerance=1e-10, max_num_iterations=60)
    assert want.IsSolutionUsable() and got.IsSolutionUsable()
    assert got.num_effective_parameters == want.num_effective_parameters
    # EUCM: alpha and beta trade off along a nearly flat valley at this field of view (both solvers
    # walk it for all 60 iterations), so the intrinsics agree to fewer digits than the cost does
    # (the same for the rational / high-order coefficients of the two 12-parameter models)
    atol = 2e-4 if model in (scene.EUCM, scene.FULL_OPENCV, scene.THIN_PRISM_FISHEYE, scene.RAD_TAN_THIN_PRISM_FISHEYE) else 1e-5
    # the high-order coefficients of the 12- / 16-parameter models are unobservable at this field of view and
    # drift to 1e3 .. 1e8: compared relatively
    _assert_close(a, want, b, got, cost_rtol=1e-7, param_atol=atol, traj_rtol=1e-5)
    assert got.final_cost < 0.2 * got.initial_cost


# ------------------------------------------------------------------------------------------------
# position priors (PosePriorBundleAdjuster, cost_functions/pose_prior.h:76-129)
# ------------------------------------------------------------------------------------------------

_EXACT = dict(eta=1e-10, max_linear_solver_iterations=500)  # see tests/test_ba_oracle.py


def _flat_prior_problem(seed=3, n_img=8, sigma=0.05):
    fp = _flat(n_img, 120, 4, seed=seed, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.02, 0.5))
    rng = np.random.default_rng(seed)
    centres = np.stack([-scene.quat_to_rot(p[:4]).T @ p[4:] for p in fp.poses])
    fp.prior_pose = np.arange(n_img, dtype=np.int32)
    fp.prior_position = np.ascontiguousarray(centres + sigma * rng.normal(size=centres.shape))
    cov = np.diag([0.01, 0.02, 0.04]) + 0.002
    L = np.linalg.cholesky(np.linalg.inv(cov))
    fp.prior_sqrt_info = np.ascontiguousarray(np.repeat(L.T[None], n_img, 0))
    return fp


@pytest.mark.parametrize("loss", [est.LossFunctionType.TRIVIAL, est.LossFunctionType.CAUCHY])
def test_position_priors_match_oracle(loss):
    """Priors instead of a gauge: 3 residuals per prior on the pose blocks, covariance weighted, with their own
    loss; the HIP solve follows the oracle (cost 1e-8, parameters 1e-6) and is bit-reproducible."""
    fp = _flat_prior_problem()
    fp.prior_loss_type, fp.prior_loss_scale = int(loss), 1.5
    if loss != est.LossFunctionType.TRIVIAL:
        fp.prior_position = fp.prior_position.copy()
        fp.prior_position[2] += 3.0  # an outlier for the robust loss
    # (stop on the relative cost change: at the floor of the projected gradient accept / reject decisions
    # are rounding noise and the two sides would end with different termination types)
    (a, want), (b, got) = _both(fp, max_num_iterations=60, gradient_tolerance=1e-7, function_tolerance=1e-12, **_EXACT)
    assert want.IsSolutionUsable() and got.num_residuals == 2 * len(fp.obs_pose) + 24
    _assert_close(a, want, b, got, cost_rtol=1e-8, param_atol=1e-6)
    c = fp.copy()
    again = est.solve_flat(c, est.SolverOptions(max_num_iterations=60, gradient_tolerance=1e-7, function_tolerance=1e-12, **_EXACT),
                           gpu_index=0)
    assert again.final_cost == got.final_cost and np.array_equal(c.poses, b.poses)
    # a gauge-fixed problem with priors on top (priors on constant blocks are ignored)
    g = fp.copy()
    assert est.fix_gauge_two_cams(g)
    (a2, want2), (b2, got2) = _both(g, max_num_iterations=40, gradient_tolerance=1e-7, function_tolerance=1e-12, **_EXACT)
    assert got2.num_residuals == want2.num_residuals == 2 * len(fp.obs_pose) + 3 * int((g.pose_const == 0).sum())
    _assert_close(a2, want2, b2, got2, cost_rtol=1e-8, param_atol=1e-6)


def test_pose_prior_adjuster_on_rigs_matches_oracle():
    """CreatePosePriorBundleAdjuster on two-camera rigs with refine_sensor_from_rig: priors of the reference
    sensors sit on rig_from_world, priors of the other sensors on (sensor_from_rig, rig_from_world)
    (AbsoluteRigPosePositionPriorCostFunctor); HIP and oracle produce the same reconstruction."""
    def run(solve_fn):
        rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_cameras_per_rig=2, num_frames_per_rig=5,
                                                                    num_points3D=200), seed=51)
        gt = rec.copy()
        rng = np.random.default_rng(52)
        priors = [est.PosePrior(i, gt.ProjectionCenter(i) + 0.01 * rng.normal(size=3),
                                np.diag([1e-4, 2e-4, 4e-4])) for i in gt.RegImageIds()]
        scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.3, 0.02, 0.2), rec, seed=53)
        cfg = est.BundleAdjustmentConfig()
        for i in rec.RegImageIds():
            cfg.AddImage(i)
        opt = est.BundleAdjustmentOptions(refine_sensor_from_rig=True)
        opt.solver_options = est.SolverOptions(max_num_iterations=60, gradient_tolerance=1e-7, function_tolerance=1e-12,
                                               linear_solver_type=est.SOLVER_DENSE_SCHUR)
        opt.gpu_index = "0"
        ba = est.CreatePosePriorBundleAdjuster(opt, est.PosePriorBundleAdjustmentOptions(), cfg, priors, rec, solve_fn=solve_fn)
        assert ba.use_prior_position_ and (ba.problem_.prior_sensor >= 0).any() and (ba.problem_.prior_sensor < 0).any()
        assert ba.problem_.sensor_const is not None and not ba.problem_.sensor_const.all()
        return rec, gt, ba.Solve()
    rec_o, gt, s_o = run(ba_oracle.solve_fn)
    rec_h, _, s_h = run(None)
    assert s_h.num_residuals == s_o.num_residuals and s_h.termination_type == s_o.termination_type
    assert abs(s_h.final_cost - s_o.final_cost) <= 1e-8 * s_o.final_cost
    for i in gt.RegImageIds():
        np.testing.assert_allclose(rec_h.images[i].cam_from_world, rec_o.images[i].cam_from_world, atol=1e-6)
        assert np.linalg.norm(rec_h.ProjectionCenter(i) - gt.ProjectionCenter(i)) < 0.05
    for rid, rig in rec_o.rigs.items():
        for cid, sfr in rig.sensors.items():
            np.testing.assert_allclose(rec_h.rigs[rid].sensors[cid], sfr, atol=1e-6)


def test_dense_schur_tier_matches_oracle():
    """DENSE_SCHUR (the reduced camera system formed explicitly, blocked Cholesky on the f64 matrix cores:
    colmap_amd/csrc/ba_schur_explicit.hip) against the oracle's exact tier, and AUTO = dense for <= 50 images
    (bundle_adjustment_ceres.cc:203-213)."""
    fp = _flat(10, 250, 5, seed=61, mixed=True)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, max_num_iterations=30, gradient_tolerance=1e-8, function_tolerance=1e-12,
                                linear_solver_type=est.SOLVER_DENSE_SCHUR)
    assert want.IsSolutionUsable() and (got.log_linear_iters[:got.num_iterations] == 1).all()
    _assert_close(a, want, b, got, cost_rtol=1e-9, param_atol=1e-7)
    c = fp.copy()
    auto = est.solve_flat(c, est.SolverOptions(max_num_iterations=30, gradient_tolerance=1e-8, function_tolerance=1e-12,
                                               linear_solver_type=est.SOLVER_AUTO), gpu_index=0)
    # (the explicit formation accumulates S with INTEGER atomics on 2^-60 fixed-point terms: bit-reproducible)
    assert auto.linear_solver_used == est.SOLVER_DENSE_SCHUR
    assert auto.final_cost == got.final_cost and np.array_equal(c.poses, b.poses)
    # the priors-only gauge: the exact tier converges where the default inexact PCG crawls
    p = _flat_prior_problem()
    (a2, want2), (b2, got2) = _both(p, max_num_iterations=60, gradient_tolerance=1e-7, function_tolerance=1e-12,
                                    linear_solver_type=est.SOLVER_DENSE_SCHUR)
    assert got2.termination_type == est.BundleAdjustmentTerminationType.CONVERGENCE
    _assert_close(a2, want2, b2, got2, cost_rtol=1e-8, param_atol=1e-6)


@pytest.mark.parametrize("frames,points,track,mixed,iters", [(120, 12000, 8, "three", 8), (300, 40000, 10, False, 6),
                                                             (1000, 200000, 10, False, 4)])
def test_sparse_schur_tier_matches_oracle(frames, points, track, mixed, iters):
    """SPARSE_SCHUR -- the reference's default tier for 51..1000 images, hence at BASELINE config[3]'s own
    size (bundle_adjustment_ceres.cc:203-213): exact Newton steps from the explicitly formed reduced camera
    system (n_c = 960 ... 7 995; one wave per point scatters the blocks of the camera pairs it connects, the
    factorisation runs 64-wide panels on v_mfma_f64_16x16x4_f64) against the oracle's explicit formation +
    blocked CPU Cholesky: every logged cost to 1e-7 relative, parameters to 1e-6."""
    fp = _flat(frames, points, track, seed=frames, mixed=mixed)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = _both(fp, max_num_iterations=iters, linear_solver_type=est.SOLVER_AUTO)
    assert got.linear_solver_used == want.linear_solver_used == est.SOLVER_SPARSE_SCHUR
    assert (got.log_linear_iters[:got.num_iterations] == 1).all()
    assert got.num_iterations == want.num_iterations and got.num_successful_steps == want.num_successful_steps
    np.testing.assert_allclose(got.log_cost, want.log_cost, rtol=1e-7)
    np.testing.assert_allclose(b.points, a.points, atol=1e-6)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)
    assert ba_compare.projection_diff_px(a, b) <= 5e-5   # the intrinsics, through what they project (tests/ba_compare.py)
    assert got.factor_seconds > 0.0


def test_exact_tier_is_bit_reproducible():
    """The exact tiers accumulate the reduced camera system with integer atomics (fixed point, 2^-60): the order the
    hardware serves them in cannot change a bit. Two solves of a 200-image problem (SPARSE_SCHUR via AUTO)."""
    fp = _flat(200, 20000, 8, seed=13)
    assert est.fix_gauge_two_cams(fp)
    runs = []
    for _ in range(2):
        b = fp.copy()
        runs.append((b, est.solve_flat(b, est.SolverOptions(max_num_iterations=6, linear_solver_type=est.SOLVER_AUTO), gpu_index=0)))
    assert runs[0][1].linear_solver_used == est.SOLVER_SPARSE_SCHUR
    assert np.array_equal(runs[0][1].log_cost, runs[1][1].log_cost)
    assert np.array_equal(runs[0][0].poses, runs[1][0].poses) and np.array_equal(runs[0][0].points, runs[1][0].points)


@pytest.mark.parametrize("frames,points,track,shared", [(120, 6000, 8, False), (60, 3000, 6, True)])
def test_exact_tier_pair_major_formation_equals_point_major(frames, points, track, shared):
    """The two formations of the reduced camera system (ba_schur_explicit.h: form) inside whole solves: pair-major
    (records + sorted incidence lists, one wave per 64 incidences, runs accumulated in registers; the default) against
    point-major (one wave per point, one atomic per term; COLMAP_AMD_BA_FORM_PAIRS=0). Different summation orders of
    the same terms: the exact Newton steps agree to rounding. With cameras shared between images the intrinsics blocks
    are reached from every pair of images (the atomics of the pair-major flush) and pairs of observations of one shared
    block meet its diagonal twice."""
    d = scene.synthesize_flat(frames, points, track, seed=frames, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.03, 0.5))
    if shared:
        d["obs_cam"] = (d["obs_cam"] % 3).astype(np.int32)
        d["cams"] = d["cams"][:3].copy()
        d["cam_model"] = d["cam_model"][:3].copy()
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_two_cams(fp)
    fp.point_const[::11] = 1
    so = dict(max_num_iterations=4, linear_solver_type=est.SOLVER_SPARSE_SCHUR)
    b0, s0 = _solve_env(fp, {"COLMAP_AMD_BA_FORM_PAIRS": "0"}, **so)
    b1, s1 = _solve_env(fp, {"COLMAP_AMD_BA_FORM_PAIRS": "1"}, **so)
    assert s0.linear_solver_used == s1.linear_solver_used == est.SOLVER_SPARSE_SCHUR
    assert len(s1.log_cost) == len(s0.log_cost)
    np.testing.assert_allclose(s1.log_cost, s0.log_cost, rtol=1e-11)
    np.testing.assert_allclose(b1.points, b0.points, atol=1e-9)
    np.testing.assert_allclose(b1.poses, b0.poses, atol=1e-9)
    np.testing.assert_allclose(b1.cams, b0.cams, rtol=1e-9, atol=1e-9)


def test_exact_tier_explicit_formation_equals_operator_products():
    """The explicit formation against the round-2 formation of the same matrix (n_c implicit operator products,
    one-workgroup Cholesky; kept behind COLMAP_AMD_BA_DENSE_BY_PRODUCTS for image-sharded solves): shared
    intrinsics (observation pairs of a point inside one block), constant points, a held translation coordinate."""
    d = scene.synthesize_flat(12, 400, 5, seed=71, noise=scene.SyntheticNoiseOptions(0.01, 0.5, 0.03, 0.5))
    d["obs_cam"] = (d["obs_cam"] % 3).astype(np.int32)   # three cameras shared by twelve images
    d["cams"] = d["cams"][:3].copy()
    d["cam_model"] = d["cam_model"][:3].copy()
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_two_cams(fp)
    fp.point_const[::9] = 1
    # four exact Newton steps (at the floor of the projected gradient the two formations stop on different
    # iterations: the accept / reject decisions there are rounding noise)
    so = dict(max_num_iterations=4, linear_solver_type=est.SOLVER_DENSE_SCHUR)
    b0, s0 = _solve_env(fp, {"COLMAP_AMD_BA_DENSE_BY_PRODUCTS": "1"}, **so)
    b1, s1 = _solve_env(fp, {"COLMAP_AMD_BA_DENSE_BY_PRODUCTS": "0"}, **so)
    assert len(s1.log_cost) == len(s0.log_cost)
    np.testing.assert_allclose(s1.log_cost, s0.log_cost, rtol=1e-10)
    np.testing.assert_allclose(b1.points, b0.points, atol=1e-9)
    np.testing.assert_allclose(b1.poses, b0.poses, atol=1e-9)


def test_reference_pose_prior_backend_case():
    """PosePriorBundleAdjusterBackendTest.Nominal (bundle_adjustment_test.cc:423-481) with backend MI355X."""
    from test_ba_oracle import _reference_pose_prior_backend_case
    s_hip = _reference_pose_prior_backend_case(None, gpu_index="0")
    s_cpu = _reference_pose_prior_backend_case(ba_oracle.solve_fn)
    assert s_hip.num_residuals == s_cpu.num_residuals and s_hip.num_residuals % 2 == 1  # 2 per observation + 3 x 7 priors
    assert abs(s_hip.final_cost - s_cpu.final_cost) <= 1e-6 * s_cpu.final_cost
