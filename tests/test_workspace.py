"""CPU tests of the data formats either side of the PatchMatch path (colmap_amd/workspace.py):
sparse model files, mvs::Model statistics, patch-match.cfg, workspace layout, consistency graphs."""
import os

import numpy as np
import pytest

from colmap_amd import mvs, synthetic as syn, workspace as W


def _sparse_model(n_img=5, n_pts=60, seed=0, w=96, h=72):
    """A small ring reconstruction: cameras on an arc looking at points near the origin."""
    rng = np.random.default_rng(seed)
    sm = W.SparseModel()
    sm.cameras[1] = W.SparseCamera(1, 1, w, h, np.array([90.0, 91.0, w / 2, h / 2]))
    cams = syn.ring_cameras(n_img, w, h, 90.0, arc_deg=40.0)
    pts = rng.uniform(-0.4, 0.4, (n_pts, 3))
    for i, (K, R, T) in enumerate(cams):
        R = np.asarray(R, np.float64); T = np.asarray(T, np.float64)
        # rotation matrix -> quaternion (w x y z)
        qw = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        q = np.array([qw, (R[2, 1] - R[1, 2]) / (4 * qw), (R[0, 2] - R[2, 0]) / (4 * qw), (R[1, 0] - R[0, 1]) / (4 * qw)])
        sm.images[i + 1] = W.SparseImage(i + 1, q, T, 1, f"img{i:02d}.png")
    for j in range(n_pts):
        track = []
        for iid, im in sm.images.items():
            if rng.random() < 0.7:
                pc = im.RotationMatrix() @ pts[j] + im.tvec
                xy = sm.cameras[1].CalibrationMatrix() @ (pc / pc[2])
                idx = len(im.xys)
                im.xys = np.vstack([im.xys, xy[:2]])
                im.point3D_ids = np.append(im.point3D_ids, j + 1)
                track.append((iid, idx))
        if len(track) >= 2:
            sm.points3D[j + 1] = W.SparsePoint3D(j + 1, pts[j], (10, 20, 30), 0.5, track)
        else:
            for iid, idx in track:
                sm.images[iid].point3D_ids[idx] = -1
    return sm


def _assert_models_equal(a, b):
    assert sorted(a.cameras) == sorted(b.cameras) and sorted(a.images) == sorted(b.images)
    assert sorted(a.points3D) == sorted(b.points3D)
    for k in a.cameras:
        assert (a.cameras[k].model_id, a.cameras[k].width, a.cameras[k].height) == \
               (b.cameras[k].model_id, b.cameras[k].width, b.cameras[k].height)
        assert np.array_equal(a.cameras[k].params, b.cameras[k].params)
    for k in a.images:
        x, y = a.images[k], b.images[k]
        assert x.name == y.name and x.camera_id == y.camera_id
        assert np.array_equal(x.qvec, y.qvec) and np.array_equal(x.tvec, y.tvec)
        assert np.array_equal(x.xys, y.xys) and np.array_equal(x.point3D_ids, y.point3D_ids)
    for k in a.points3D:
        x, y = a.points3D[k], b.points3D[k]
        assert np.array_equal(x.xyz, y.xyz) and tuple(x.rgb) == tuple(y.rgb) and x.error == y.error
        assert x.track == y.track


def test_sparse_model_binary_and_text_round_trip(tmp_path):
    sm = _sparse_model()
    W.write_model_binary(sm, str(tmp_path / "bin"))
    W.write_model_text(sm, str(tmp_path / "txt"))
    _assert_models_equal(sm, W.read_sparse_model(str(tmp_path / "bin")))
    _assert_models_equal(sm, W.read_sparse_model(str(tmp_path / "txt")))
    # byte layout of the binary files (reconstruction_io_binary.cc): counts and record sizes
    raw = (tmp_path / "bin" / "cameras.bin").read_bytes()
    assert len(raw) == 8 + 1 * (4 + 4 + 8 + 8 + 4 * 8)
    raw = (tmp_path / "bin" / "images.bin").read_bytes()
    want = 8 + sum(4 + 56 + 4 + len(im.name) + 1 + 8 + 24 * len(im.xys) for im in sm.images.values())
    assert len(raw) == want
    raw = (tmp_path / "bin" / "points3D.bin").read_bytes()
    assert len(raw) == 8 + sum(8 + 24 + 3 + 8 + 8 + 8 * len(p.track) for p in sm.points3D.values())
    # an observation without a 3-D point is stored as 2^64 - 1 (kInvalidPoint3DId)
    im = next(i for i in sm.images.values() if (i.point3D_ids == -1).any())
    k = int(np.nonzero(im.point3D_ids == -1)[0][0])
    off = (tmp_path / "bin" / "images.bin").read_bytes().find(im.name.encode() + b"\0") + len(im.name) + 1 + 8
    rec = (tmp_path / "bin" / "images.bin").read_bytes()[off + 24 * k + 16: off + 24 * k + 24]
    assert rec == b"\xff" * 8
    with pytest.raises(FileNotFoundError):
        W.read_sparse_model(str(tmp_path / "nothing"))


def test_model_statistics_match_their_definitions():
    sm = _sparse_model(6, 120, seed=2)
    m = W.Model.FromSparseModel(sm, "/images")
    assert [im.path for im in m.images] == [f"/images/img{i:02d}.png" for i in range(6)]
    assert m.GetImageIdx("img03.png") == 3 and m.GetImageName(3) == "img03.png"
    with pytest.raises(KeyError):
        m.GetImageIdx("missing.png")
    # shared points: symmetric, equal to the track co-occurrence counts
    shared = m.ComputeSharedPoints()
    for a in range(6):
        for b, c in shared[a].items():
            assert shared[b][a] == c
            assert c == sum(1 for p in m.points if a in p.track and b in p.track)
    # depth ranges: stretched 1st / 99th percentile elements of the positive depths (model.cc:178-218)
    ranges = m.ComputeDepthRanges()
    for idx, (lo, hi) in enumerate(ranges):
        d = sorted(float(np.dot(m.images[idx].R[2], np.array([p.x, p.y, p.z], np.float32)) + m.images[idx].T[2])
                   for p in m.points if idx in p.track)
        assert abs(lo - 0.75 * d[int(len(d) * 0.01)]) < 1e-5 and abs(hi - 1.25 * d[int(len(d) * 0.99)]) < 1e-5
        assert 0 < lo < hi
    # triangulation angles: symmetric, within (0, pi/2], larger for wider baselines on average
    ang = m.ComputeTriangulationAngles(75.0)
    for a in range(6):
        for b, v in ang[a].items():
            assert abs(ang[b][a] - v) < 1e-7 and 0 < v <= np.pi / 2 + 1e-6
    assert ang[0][5] > ang[0][1]
    # overlapping images: ordered by shared points, filtered by the angle
    over = m.GetMaxOverlappingImages(3, 0.0)
    for idx, lst in enumerate(over):
        counts = [shared[idx][o] for o in lst]
        assert counts == sorted(counts, reverse=True) and len(lst) == 3
        assert min(counts) >= max([c for o, c in shared[idx].items() if o not in lst] or [0])
    assert all(len(l) == 0 for l in m.GetMaxOverlappingImages(3, 89.0))
    assert W.percentile([3.0, 1.0, 2.0, 4.0], 75) == np.percentile([1, 2, 3, 4], 75)
    # an image without points gets (-1, -1)
    sm.images[7] = W.SparseImage(7, np.array([1.0, 0, 0, 0]), np.zeros(3), 1, "lonely.png")
    assert W.Model.FromSparseModel(sm, "")\
        .ComputeDepthRanges()[6] == (-1.0, -1.0)


def test_patch_match_config_parsing():
    # patch_match.cc:239-359
    m = W.Model.FromSparseModel(_sparse_model(5, 80, seed=1), "")
    cfg = """
# comment
img00.png
__all__

img01.png
  img00.png, img02.png ,img04.png
img02.png
__auto__, 2
img03.png
__auto__, 50
""".splitlines()
    probs = W.read_patch_match_config(cfg, m, min_triangulation_angle=0.0)
    assert probs[0] == (0, [1, 2, 3, 4])
    assert probs[1] == (1, [0, 2, 4])
    shared = m.ComputeSharedPoints()
    assert probs[2][0] == 2 and len(probs[2][1]) == 2
    assert set(probs[2][1]) <= set(shared[2]) and \
        min(shared[2][o] for o in probs[2][1]) >= max(c for o, c in shared[2].items() if o not in probs[2][1])
    assert sorted(probs[3][1]) == sorted(shared[3])          # fewer candidates than requested: all of them
    # a reference image whose candidates all fail the triangulation-angle test is dropped with a warning
    warned = []
    probs = W.read_patch_match_config(["img02.png", "__auto__, 2"], m, min_triangulation_angle=80.0, warn=warned.append)
    assert probs == [] and "img02.png" in warned[0]
    with pytest.raises(KeyError):
        W.read_patch_match_config(["img00.png", "nope.png"], m)


def test_grey_conversion_and_workspace_layout(tmp_path):
    from PIL import Image as PILImage
    ws = tmp_path / "dense"
    sm = _sparse_model(4, 60, seed=3, w=40, h=30)
    W.write_model_binary(sm, str(ws / "sparse"))
    os.makedirs(ws / "images"); os.makedirs(ws / "stereo")
    rng = np.random.default_rng(0)
    rgbs = {}
    for i, im in sm.images.items():
        rgb = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
        rgbs[im.name] = rgb
        PILImage.fromarray(rgb).save(ws / "images" / im.name)
    W.write_patch_match_config(str(ws / "stereo" / "patch-match.cfg"), [im.name for im in sm.images.values()],
                               "__all__")
    w = W.Workspace(str(ws))
    g = w.GetBitmap(2)
    rgb = rgbs["img02.png"].astype(np.float32)
    want = np.floor(np.float32(.2126) * rgb[..., 0] + np.float32(.7152) * rgb[..., 1] +
                    np.float32(.0722) * rgb[..., 2] + np.float32(0.5)).astype(np.uint8)   # Bitmap::CloneAsGrey
    assert np.array_equal(g, want) and g.dtype == np.uint8 and g.flags["C_CONTIGUOUS"]
    assert w.GetBitmap(2) is g                                  # cached: stable address for the device cache
    assert w.GetDepthMapPath(1, "photometric") == str(ws / "stereo" / "depth_maps" / "img01.png.photometric.bin")
    assert w.GetNormalMapPath(1, "geometric") == str(ws / "stereo" / "normal_maps" / "img01.png.geometric.bin")
    # grey files are taken as they are
    PILImage.fromarray(want).save(ws / "images" / "grey.png")
    assert np.array_equal(W.read_bitmap_grey(str(ws / "images" / "grey.png")), want)
    # max_image_size: mvs::Image::Downsize scales K with the realised size ratios (image.cc:66-95)
    w2 = W.Workspace(str(ws), max_image_size=20)
    im = w2.GetModel().images[0]
    assert (im.width, im.height) == (20, 15)
    assert abs(im.K[0, 0] - 90.0 * 20 / 40) < 1e-4 and abs(im.K[1, 2] - 15.0 * 15 / 30) < 1e-4
    assert w2.GetBitmap(0).shape == (15, 20)
    # the controller built from the workspace: problems, depth ranges, no GPU touched yet
    ctl = mvs.PatchMatchController.FromWorkspace(mvs.PatchMatchOptions(gpu_index="0"), str(ws))
    assert len(ctl.problems_) == 4 and ctl.problems_[0] == (0, [1, 2, 3])
    assert all(im.depth_range is not None and 0 < im.depth_range[0] < im.depth_range[1] for im in ctl.images_)
    # a missing source image is an error unless allow_missing_files
    os.remove(ws / "images" / "img03.png")
    with pytest.raises(mvs.PatchMatchError, match="Missing image"):
        mvs.PatchMatchController.FromWorkspace(mvs.PatchMatchOptions(gpu_index="0"), str(ws))
    ctl = mvs.PatchMatchController.FromWorkspace(mvs.PatchMatchOptions(gpu_index="0", allow_missing_files=True), str(ws))
    assert [p for p in ctl.problems_] == [(0, [1, 2]), (1, [0, 2]), (2, [0, 1])]


def test_consistency_graph_files(tmp_path):
    # consistency_graph.cc:69-139 + GetConsistentImageIdxs (patch_match_cuda.cu:1367-1391)
    data = np.array([3, 0, 2, 7, 9,   1, 2, 1, 4], np.int32)     # (col,row,n,idxs...)
    p = str(tmp_path / "g.bin")
    W.write_consistency_graph(p, 5, 4, data)
    assert open(p, "rb").read()[:6] == b"5&4&1&"
    w, h, g = W.read_consistency_graph(p)
    assert (w, h) == (5, 4) and g == {(0, 3): [7, 9], (2, 1): [4]}
    W.write_consistency_graph(p, 5, 4, np.array([9, 0, 1, 2], np.int32))
    with pytest.raises(ValueError):
        W.read_consistency_graph(p)                                # column out of range


def test_cli_option_surface():
    from colmap_amd import patch_match_stereo as cli
    a = cli.build_parser().parse_args(["--workspace_path", "/x", "--PatchMatchStereo.geom_consistency", "false",
                                       "--PatchMatchStereo.window_radius", "7", "--PatchMatchStereo.depth_min", "0.5"])
    o = cli.options_from_args(a)
    assert o.geom_consistency is False and o.window_radius == 7 and o.depth_min == 0.5 and o.filter is True
    # every option of controllers/option_manager.cc:932-980 is accepted
    for name in ["max_image_size", "gpu_index", "depth_min", "depth_max", "window_radius", "window_step",
                 "sigma_spatial", "sigma_color", "num_samples", "ncc_sigma", "min_triangulation_angle",
                 "incident_angle_sigma", "num_iterations", "geom_consistency", "geom_consistency_regularizer",
                 "geom_consistency_max_cost", "filter", "filter_min_ncc", "filter_min_triangulation_angle",
                 "filter_min_num_consistent", "filter_geom_consistency_max_cost", "cache_size",
                 "allow_missing_files", "write_consistency_graph", "num_threads"]:
        assert hasattr(a, "pm_" + name), name


def test_text_model_edge_cases(tmp_path):
    """images.txt: an image without observations has an EMPTY second line; names may contain spaces;
    comment / blank lines are skipped (scene/reconstruction_io_text.cc)."""
    d = tmp_path / "m"
    os.makedirs(d)
    (d / "cameras.txt").write_text("# Camera list\n\n1 SIMPLE_RADIAL 100 80 90.5 50 40 0.01\n2 OPENCV 64 48 70 71 32 24 0.1 0.01 0 0\n")
    (d / "images.txt").write_text(
        "# Image list with two lines of data per image:\n"
        "1 1 0 0 0 0.5 0 2 1 my image 01.png\n"
        "10.5 20.25 7 30 40 -1\n"
        "2 0.5 0.5 0.5 0.5 0 0 0 2 b.png\n"
        "\n"
        "3 1 0 0 0 0 0 1 1 c.png\n"
        "1 2 7\n")
    (d / "points3D.txt").write_text("# points\n7 0.1 0.2 0.3 255 0 10 0.75 1 0 3 0\n")
    sm = W.read_sparse_model(str(d))
    assert sm.cameras[2].model_id == 4 and len(sm.cameras[2].params) == 8
    assert sm.images[1].name == "my image 01.png" and sm.images[1].xys.shape == (2, 2)
    assert list(sm.images[1].point3D_ids) == [7, -1]
    assert sm.images[2].xys.shape == (0, 2) and sm.images[2].name == "b.png"
    assert sm.images[3].name == "c.png" and list(sm.images[3].point3D_ids) == [7]
    assert sm.points3D[7].track == [(1, 0), (3, 0)] and sm.points3D[7].rgb == (255, 0, 10) and sm.points3D[7].error == 0.75
    K = sm.cameras[2].CalibrationMatrix()
    assert (K[0, 0], K[1, 1], K[0, 2], K[1, 2]) == (70, 71, 32, 24)
    # rotation of quaternion (w x y z) = (.5 .5 .5 .5): cyclic permutation of the axes
    np.testing.assert_allclose(sm.images[2].RotationMatrix(), [[0, 0, 1], [1, 0, 0], [0, 1, 0]], atol=1e-15)
    # text -> binary -> text keeps everything
    W.write_model_binary(sm, str(tmp_path / "b"))
    back = W.read_sparse_model(str(tmp_path / "b"))
    _assert_models_equal(sm, back)
    # patch-match.cfg with CRLF line endings and trailing blanks
    m = W.Model.FromSparseModel(sm, "")
    probs = W.read_patch_match_config("my image 01.png\r\n__all__\r\n\r\nb.png  \r\nc.png , my image 01.png\r\n".splitlines(), m)
    assert probs == [(0, [1, 2]), (1, [2, 0])]
    # a dangling reference image line without a source line is ignored like the reference's parser does
    assert W.read_patch_match_config(["b.png"], m) == []


def test_pmvs_workspace_import(tmp_path):
    """workspace_format PMVS (mvs/model.cc:358-454, workspace.cc:250-322): projection matrices in
    txt/%08d.txt, vis.dat, option file -> model + patch-match.cfg / fusion.cfg under stereo-<option>."""
    from PIL import Image as PILImage
    cams = syn.ring_cameras(4, 64, 48, 60.0, arc_deg=30.0)
    root = tmp_path / "pmvs"
    os.makedirs(root / "visualize"); os.makedirs(root / "txt")
    rng = np.random.default_rng(0)
    for i, (K, R, T) in enumerate(cams):
        PILImage.fromarray(rng.integers(0, 255, (48, 64), dtype=np.uint8)).save(root / "visualize" / f"{i:08d}.jpg")
        P = np.asarray(K, np.float64) @ np.concatenate([np.asarray(R, np.float64), np.asarray(T, np.float64)[:, None]], 1)
        (root / "txt" / f"{i:08d}.txt").write_text("CONTOUR\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in P) + "\n")
    (root / "vis.dat").write_text("VISDATA\n4\n0 2 1 2\n1 3 0 2 3\n2 3 0 1 3\n3 2 1 2\n")
    (root / "option-all").write_text("# comment\nlevel 1\ntimages -1 0 4\noimages 0\n")
    ws = W.Workspace(str(root), "PMVS", stereo_folder="stereo-option-all")
    m = ws.GetModel()
    assert [os.path.basename(im.path) for im in m.images] == [f"{i:08d}.jpg" for i in range(4)]
    for im, (K, R, T) in zip(m.images, cams):
        assert (im.width, im.height) == (64, 48)
        np.testing.assert_allclose(im.K, np.asarray(K, np.float32), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(im.R, np.asarray(R, np.float32), atol=1e-5)
        np.testing.assert_allclose(im.T, np.asarray(T, np.float32), atol=1e-4)
        assert im.K[0, 1] == 0 and im.K[2, 2] == 1                 # skew dropped (:396-402)
    assert m.GetMaxOverlappingImagesFromPMVS() == [[1, 2], [0, 2, 3], [0, 1, 3], [1, 2]]
    W.import_pmvs_workspace(ws, "option-all")
    base = root / "stereo-option-all"
    assert (base / "depth_maps").is_dir() and (base / "consistency_graphs").is_dir()
    cfg = (base / "patch-match.cfg").read_text().splitlines()
    assert cfg[0] == "00000000.jpg" and cfg[1] == "00000001.jpg, 00000002.jpg, "
    assert (base / "fusion.cfg").read_text().split() == [f"{i:08d}.jpg" for i in range(4)]
    # the written configuration parses back into the visibility lists
    probs = W.read_patch_match_config(cfg, m)
    assert probs == [(0, [1, 2]), (1, [0, 2, 3]), (2, [0, 1, 3]), (3, [1, 2])]
    # explicit image list in the option file; malformed line
    (root / "option-some").write_text("timages 2 1 3\n")
    W.import_pmvs_workspace(ws, "option-some")
    assert (base / "fusion.cfg").read_text().split() == ["00000001.jpg", "00000003.jpg"]
    (root / "option-bad").write_text("timages 3 1 3\n")
    with pytest.raises(ValueError):
        W.import_pmvs_workspace(ws, "option-bad")
    with pytest.raises(ValueError):
        W.Workspace(str(tmp_path), "PMVS")                          # neither bundle.rd.out nor vis.dat
    with pytest.raises(ValueError):
        W.Workspace(str(root), "NVM")
    # the controller sets the workspace up the same way (patch_match.cc:212-233)
    ctl = mvs.PatchMatchController.FromWorkspace(
        mvs.PatchMatchOptions(gpu_index="0", depth_min=1.0, depth_max=10.0), str(root), "PMVS", "option-all")
    assert ctl.problems_ == [(0, [1, 2]), (1, [0, 2, 3]), (2, [0, 1, 3]), (3, [1, 2])]
    assert ctl._paths(0, "photometric")[0] == str(base / "depth_maps" / "00000000.jpg.photometric.bin")
