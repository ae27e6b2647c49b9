"""The C++ host side (include/colmap_amd/*.hpp) above the C ABI: compiled with g++ against
libcolmap_amd.so and driven from here. CPU tests cover the host logic (checks, file formats, the
BundleAdjuster adapter's flattening rules); GPU tests run the same problems through the C++ classes
and through the Python mirror and require bit-identical outputs (both sit on the same C ABI)."""
import os
import subprocess

import numpy as np
import pytest

from colmap_amd import build as _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(name, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / name)
    _build.build()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", name + ".cc"), "-L", _build.LIB_DIR, "-lcolmap_amd",
           "-Wl,-rpath," + _build.LIB_DIR, "-o", out]
    subprocess.check_call(cmd)
    return out


@pytest.fixture(scope="session")
def mvs_host(tmp_path_factory):
    return _compile("test_mvs_host", tmp_path_factory)


def test_mvs_host_checks(mvs_host, tmp_path):
    r = subprocess.run([mvs_host, "check", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "host checks OK" in r.stdout


@pytest.mark.gpu
def test_cpp_stereo_fusion(mvs_host):
    """colmap_amd::mvs::StereoFusion (C++ class over fusion_run, HIP): two fronto-parallel views of a plane."""
    r = subprocess.run([mvs_host, "fusion"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "fusion checks OK" in r.stdout


def _write_problem(d, views, ref, src, geom, filt, iters, dmin, dmax, maps=None):
    from colmap_amd import mvs
    h, w = views[0].gray.shape
    with open(os.path.join(d, "problem.txt"), "w") as f:
        f.write(f"{len(views)} {w} {h} {ref} {len(src)} {int(geom)} {int(filt)} {iters} {dmin!r} {dmax!r}\n")
        f.write(" ".join(str(s) for s in src) + "\n")
        for v in views:
            for a in (v.K, v.R, v.T):
                f.write(" ".join(repr(float(x)) for x in np.asarray(a, np.float32).ravel()) + "\n")
    for i, v in enumerate(views):
        v.gray.tofile(os.path.join(d, f"img{i}.gray"))
        if maps is not None:
            mvs.write_mat(os.path.join(d, f"in_depth{i}.bin"), maps[i][0])
            mvs.write_mat(os.path.join(d, f"in_normal{i}.bin"), maps[i][1])


@pytest.mark.gpu
def test_cpp_patch_match_equals_python_mirror(mvs_host, tmp_path):
    """colmap_amd::mvs::PatchMatch (C++) vs colmap_amd.mvs.PatchMatch (Python): same C ABI, same bits;
    photometric pass, then the geometric pass fed with the photometric maps through Mat files."""
    from colmap_amd import mvs, synthetic as syn, workspace as W
    from pm_common import scene, hip_problem
    views = scene(4, 80, 60)
    ref, src = 1, [0, 2, 3]
    dmin, dmax = syn.depth_range(views, ref)
    # photometric maps of all images (the inputs of the geometric pass)
    maps = []
    for r in range(4):
        lo, hi = syn.depth_range(views, r)
        o = mvs.PatchMatchOptions(gpu_index="0", depth_min=lo, depth_max=hi, sigma_spatial=5.0,
                                  geom_consistency=False, filter=False, num_iterations=2)
        pm = mvs.PatchMatch(o, hip_problem(views, r, [i for i in range(4) if i != r]))
        pm.Run()
        maps.append((pm.GetDepthMap(), pm.GetNormalMap()))
    for geom in (False, True):
        d = str(tmp_path / ("geom" if geom else "photo"))
        os.makedirs(d)
        _write_problem(d, views, ref, src, geom, True, 2, dmin, dmax, maps if geom else None)
        r = subprocess.run([mvs_host, "run", d], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        o = mvs.PatchMatchOptions(gpu_index="0", depth_min=dmin, depth_max=dmax, sigma_spatial=5.0,
                                  geom_consistency=geom, filter=True, num_iterations=2)
        pm = mvs.PatchMatch(o, hip_problem(views, ref, src, maps if geom else None))
        pm.Run()
        assert np.array_equal(mvs.read_mat(os.path.join(d, "depth.bin")), pm.GetDepthMap())
        assert np.array_equal(mvs.read_mat(os.path.join(d, "normal.bin")), pm.GetNormalMap())
        assert np.array_equal(mvs.read_mat(os.path.join(d, "sel_prob.bin")), pm.GetSelProbMap())
        gw, gh, graph = W.read_consistency_graph(os.path.join(d, "graph.bin"))
        flat = pm.GetConsistentImageIdxs()
        assert (gw, gh) == (80, 60)
        assert np.array_equal(np.fromfile(os.path.join(d, "graph.bin"), "<i4", offset=len(b"80&60&1&")), flat)
        assert (pm.GetDepthMap() > 0).sum() == len(graph)


# ------------------------------------------------------------------------------------------------
# bundle adjustment: C++ BundleAdjuster adapter vs the Python mirror
# ------------------------------------------------------------------------------------------------

@pytest.fixture(scope="session")
def ba_host(tmp_path_factory):
    return _compile("test_ba_host", tmp_path_factory)


def _write_ba_spec(path, rec, cfg, opts):
    """The reconstruction + config + options in the whitespace format tests/cpp/test_ba_host.cc reads."""
    r = repr
    with open(path, "w") as f:
        f.write(f"cameras {len(rec.cameras)}\n")
        for cid in sorted(rec.cameras):
            c = rec.cameras[cid]
            f.write(f"{cid} {c.model_id} {c.width} {c.height} {len(c.params)} " + " ".join(r(float(v)) for v in c.params) + "\n")
        f.write(f"rigs {len(rec.rigs)}\n")
        for rid in sorted(rec.rigs):
            g = rec.rigs[rid]
            f.write(f"{rid} {g.ref_camera_id} {len(g.sensors)}")
            for cid in sorted(g.sensors):
                f.write(f" {cid} " + " ".join(r(float(v)) for v in g.sensors[cid]))
            f.write("\n")
        f.write(f"frames {len(rec.frames)}\n")
        for fid in sorted(rec.frames):
            fr = rec.frames[fid]
            f.write(f"{fid} {fr.rig_id} " + " ".join(r(float(v)) for v in fr.rig_from_world) +
                    f" {len(fr.image_ids)} " + " ".join(str(i) for i in fr.image_ids) + "\n")
        f.write(f"images {len(rec.images)}\n")
        for iid in sorted(rec.images):
            im = rec.images[iid]
            f.write(f"{iid} {im.camera_id} {-1 if im.frame_id_ is None else im.frame_id_} " +
                    " ".join(r(float(v)) for v in im.cam_from_world) + f" {len(im.points2D)}")
            for p in im.points2D:
                f.write(f" {r(float(p.xy[0]))} {r(float(p.xy[1]))} {p.point3D_id}")
            f.write("\n")
        f.write(f"points {len(rec.points3D)}\n")
        for pid in sorted(rec.points3D):
            p = rec.points3D[pid]
            f.write(f"{pid} " + " ".join(r(float(v)) for v in p.xyz) + f" {len(p.track)} " +
                    " ".join(f"{a} {b}" for a, b in p.track) + "\n")
        f.write(f"gauge {int(cfg.FixedGauge())}\n")
        for name, ids in [("images", cfg.Images()), ("const_cams", sorted(cfg.constant_cam_intrinsics_)),
                          ("const_frames", sorted(cfg.constant_rig_from_world_poses_)),
                          ("const_sensors", sorted(cfg.constant_sensor_from_rig_)),
                          ("var_points", cfg.VariablePoints()), ("const_points", cfg.ConstantPoints()),
                          ("ignored", sorted(cfg.ignored_point3D_ids_))]:
            f.write(f"{name} {len(ids)} " + " ".join(str(i) for i in ids) + "\n")
        so = opts.solver_options
        f.write("options " + " ".join(str(int(b)) for b in [
            opts.refine_focal_length, opts.refine_principal_point, opts.refine_extra_params,
            opts.refine_sensor_from_rig, opts.refine_rig_from_world, opts.refine_points3D,
            opts.constant_rig_from_world_rotation]) +
            f" {opts.min_track_length} {int(so.loss_type)} {so.loss_scale!r} {so.max_num_iterations} "
            f"{so.gradient_tolerance!r}\n")


def _ba_cases():
    from colmap_amd import estimators as est, scene
    G = est.BundleAdjustmentGauge
    cases = []

    def config(rec, gauge):
        cfg = est.BundleAdjustmentConfig()
        for i in rec.RegImageIds():
            cfg.AddImage(i)
        cfg.FixGauge(gauge)
        return cfg

    # TwoView (bundle_adjustment_ceres_test.cc:222-269): 400 residuals, 309 parameters
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_frames_per_rig=1, num_points3D=100,
                                                                num_points2D_without_point3D=0), seed=0)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(point2D_stddev=1), rec, seed=1)
    cases.append(("two_view", rec, config(rec, G.TWO_CAMS_FROM_WORLD), est.BundleAdjustmentOptions(), (400, 309)))
    # TwoViewRig (:323-376), sensor_from_rig constant: 800 residuals, 313 - 6 parameters
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=1, num_cameras_per_rig=2, num_frames_per_rig=2,
                                                                num_points3D=100, num_points2D_without_point3D=0), seed=0)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(point2D_stddev=1), rec, seed=1)
    cases.append(("two_view_rig", rec, config(rec, G.THREE_POINTS),
                  est.BundleAdjustmentOptions(refine_sensor_from_rig=False), (800, 307)))
    # TwoViewRig as the reference runs it (:323-376): sensor_from_rig refined: 800 residuals, 313 parameters
    cases.append(("two_view_rig_variable_sensor", rec.copy(), config(rec, G.THREE_POINTS),
                  est.BundleAdjustmentOptions(), (800, 313)))
    # constant frame + constant camera + partially contained tracks + ignored / constant points, CAUCHY loss
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_cameras_per_rig=3, num_frames_per_rig=3,
                                                                num_points3D=80), seed=3)
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.01, 0.3, 0.03, 0.5), rec, seed=4)
    cfg = est.BundleAdjustmentConfig()
    ids = rec.RegImageIds()
    for i in ids[:-4]:
        cfg.AddImage(i)
    cfg.FixGauge(G.TWO_CAMS_FROM_WORLD)
    cfg.SetConstantRigFromWorldPose(2)
    cfg.SetConstantCamIntrinsics(2)
    cfg.SetConstantSensorFromRigPose(5)
    pids = sorted(rec.points3D)
    cfg.AddVariablePoint(pids[0]); cfg.AddVariablePoint(pids[1]); cfg.AddConstantPoint(pids[2]); cfg.IgnorePoint(pids[3])
    so = est.SolverOptions(loss_type=int(est.LossFunctionType.CAUCHY), loss_scale=2.0, max_num_iterations=30,
                           gradient_tolerance=1e-8, linear_solver_type=est.SOLVER_AUTO)  # the adapters' default
    cases.append(("rigs_mixed_config", rec, cfg,
                  est.BundleAdjustmentOptions(refine_sensor_from_rig=False, min_track_length=3, solver_options=so), None))
    return cases


def test_ba_host_api_and_flattening_counts(ba_host, tmp_path):
    from colmap_amd import estimators as est
    r = subprocess.run([ba_host, "api"], capture_output=True, text=True)
    assert r.returncode == 0 and "api OK" in r.stdout, r.stderr
    for name, rec, cfg, opts, known in _ba_cases():
        spec = str(tmp_path / f"{name}.txt")
        _write_ba_spec(spec, rec, cfg, opts)
        r = subprocess.run([ba_host, "counts", spec], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        residuals, params, poses, const_poses, config_residuals = (int(v) for v in r.stdout.split())
        fp = est.flatten(opts, cfg, rec.copy())
        # the Python mirror flattens to the same problem
        assert poses == len(fp.poses) and const_poses == int(fp.pose_const.sum())
        assert residuals == 2 * est.shard_num_observations(fp, 0, 1)
        if known is not None:
            assert (residuals, params) == known
            assert config_residuals == residuals          # config.NumResiduals == problem.NumResiduals


@pytest.mark.gpu
def test_cpp_bundle_adjuster_equals_python_mirror(ba_host, tmp_path):
    """Mi355xBundleAdjuster (C++) and colmap_amd.estimators.BundleAdjuster (Python) flatten to the same
    ba_problem, so the GPU solve returns the same bits; only variable blocks are written back."""
    from colmap_amd import estimators as est
    for name, rec, cfg, opts, known in _ba_cases():
        spec, out = str(tmp_path / f"{name}.txt"), str(tmp_path / f"{name}.out")
        _write_ba_spec(spec, rec, cfg, opts)
        r = subprocess.run([ba_host, "solve", spec, out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        mine = rec.copy()
        opts.gpu_index = "0"
        summary = est.BundleAdjuster(opts, cfg, mine).Solve()
        lines = open(out).read().splitlines()
        head = lines[0].split()
        assert int(head[0]) == int(summary.termination_type) and int(head[1]) == summary.num_residuals
        assert int(head[2]) == summary.num_effective_parameters and int(head[3]) == summary.num_iterations
        assert float(head[4]) == summary.initial_cost and float(head[5]) == summary.final_cost
        if known is not None:
            assert (summary.num_residuals, summary.num_effective_parameters) == known
        got = {"image": {}, "frame": {}, "camera": {}, "point": {}}
        for ln in lines[1:]:
            t = ln.split()
            got[t[0]][int(t[1])] = np.array(t[2:], float)
        for iid, im in mine.images.items():
            if mine.IsRefInFrame(iid):
                assert np.array_equal(got["image"][iid], im.cam_from_world), (name, iid)
            else:   # derived: sensor_from_rig * rig_from_world, composed by two different host codes
                np.testing.assert_allclose(got["image"][iid], im.cam_from_world, rtol=0, atol=1e-14)
        for fid, fr in mine.frames.items():
            assert np.array_equal(got["frame"][fid], fr.rig_from_world)
        for cid, c in mine.cameras.items():
            assert np.array_equal(got["camera"][cid], c.params)
        for pid, p in mine.points3D.items():
            assert np.array_equal(got["point"][pid], p.xyz)
        assert summary.final_cost < summary.initial_cost


@pytest.mark.gpu
def test_cpp_pose_prior_adjuster_equals_python_mirror(ba_host, tmp_path):
    """CreatePosePriorBundleAdjuster in C++ (Horn alignment, fixed-scale normalisation, covariance weighting) and
    in Python (Umeyama alignment): the alignments agree to round-off, so both solves end in the same metric
    reconstruction; two-camera rigs with refined sensor_from_rig exercise both prior functors."""
    from colmap_amd import estimators as est, scene
    rec = scene.SynthesizeDataset(scene.SyntheticDatasetOptions(num_rigs=2, num_cameras_per_rig=2, num_frames_per_rig=4,
                                                                num_points3D=150), seed=71)
    gt = rec.copy()
    rng = np.random.default_rng(72)
    priors = [est.PosePrior(i, gt.ProjectionCenter(i) + 0.01 * rng.normal(size=3),
                            np.diag([1e-4, 2e-4, 4e-4]) if i % 2 else None) for i in gt.RegImageIds()]
    scene.SynthesizeNoise(scene.SyntheticNoiseOptions(0.02, 0.3, 0.02, 0.2), rec, seed=73)
    Rw = scene.quat_to_rot(np.array([0.1, -0.2, 0.3, 0.9]) / np.linalg.norm([0.1, -0.2, 0.3, 0.9]))
    rec.Transform(1.3, Rw, np.array([2.0, -1.0, 0.5]))
    cfg = est.BundleAdjustmentConfig()
    for i in rec.RegImageIds():
        cfg.AddImage(i)
    so = est.SolverOptions(max_num_iterations=40, gradient_tolerance=1e-8, function_tolerance=1e-12,
                           linear_solver_type=est.SOLVER_AUTO)
    opts = est.BundleAdjustmentOptions(refine_sensor_from_rig=True, solver_options=so)
    spec, out, pfile = str(tmp_path / "prior.txt"), str(tmp_path / "prior.out"), str(tmp_path / "priors.txt")
    _write_ba_spec(spec, rec, cfg, opts)
    with open(pfile, "w") as f:
        for p in priors:
            cov = p.position_covariance if p.position_covariance is not None else np.full((3, 3), np.nan)
            f.write(f"{p.image_id} " + " ".join(repr(float(v)) for v in list(p.position) + list(np.ravel(cov))) + "\n")
    r = subprocess.run([ba_host, "prior", spec, out, pfile], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"priors used 1 count {len(priors)}" in r.stdout
    mine = rec.copy()
    opts.gpu_index = "0"
    ba = est.CreatePosePriorBundleAdjuster(opts, est.PosePriorBundleAdjustmentOptions(), cfg, priors, mine)
    summary = ba.Solve()
    lines = open(out).read().splitlines()
    head = lines[0].split()
    assert int(head[1]) == summary.num_residuals and int(head[2]) == summary.num_effective_parameters
    assert abs(float(head[5]) - summary.final_cost) <= 1e-6 * summary.final_cost
    got = {"image": {}, "point": {}}
    for ln in lines[1:]:
        t = ln.split()
        if t[0] in got:
            got[t[0]][int(t[1])] = np.array(t[2:], float)
    for iid, im in mine.images.items():
        q_sign = np.sign(got["image"][iid][:4] @ im.cam_from_world[:4])
        np.testing.assert_allclose(got["image"][iid] * np.r_[np.full(4, q_sign), np.ones(3)], im.cam_from_world, atol=1e-5)
        assert np.linalg.norm(mine.ProjectionCenter(iid) - gt.ProjectionCenter(iid)) < 0.05
    for pid, p in mine.points3D.items():
        np.testing.assert_allclose(got["point"][pid], p.xyz, atol=1e-5)
