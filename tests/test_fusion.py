"""Depth-map fusion.

CPU tests: the checker oracle/fusion_oracle.cpp -- mode 0 (the reference's sequential walk,
mvs/fusion.cc:401-524, pixels row-major = the reference with one thread) is pinned against a pure-Python
float32 restatement; mode 1 (the same walk, turns in the order of the reference's own pool schedule,
mvs/fusion.cc:253-269: stripes of ten rows, its threads advancing in step) against the ground truth of the
synthetic renderer and mode 0; mode 2 (a simulation of how the HIP kernels execute that order: speculative
passes, tentative marks, rank cut) against mode 1. GPU tests: colmap_amd/csrc/fusion.hip through fusion_run
equals mode 1 bit for bit. tests/test_fusion_emul.py runs the same kernels on the CPU."""
import os

import numpy as np
import pytest

import fusion_oracle
from colmap_amd import fusion, mvs, workspace as W
from pm_common import scene, write_dense_workspace

f32 = np.float32


def _images(views, with_rgb=True):
    out = []
    for v in views:
        h, w = v.gray.shape
        rgb = np.stack([v.gray, 255 - v.gray, (v.gray // 2)], -1) if with_rgb else None
        out.append(fusion.FusionImage(w, h, v.K, v.R, v.T, rgb, v.depth.copy(), v.normal.copy()))
    return out


def _overlap(n):
    return [[j for j in range(n) if j != i] for i in range(n)]


def _median(vals):
    v = np.sort(np.asarray(vals, np.float64))
    idx = 0.5 * (len(v) - 1)
    lo, hi = int(np.floor(idx)), int(np.ceil(idx))
    return v[hi] if lo == hi else (hi - idx) * v[lo] + (idx - lo) * v[hi]


def _fuse_reference(opt, images, overlap):
    """fusion.cc:188-524 in float32 scalar arithmetic, one thread."""
    n = len(images)
    P, iP, iR, masks = [], [], [], []
    for im in images:
        K = np.asarray(im.K, f32).reshape(3, 3).copy()
        R = np.asarray(im.R, f32).reshape(3, 3)
        T = np.asarray(im.T, f32).reshape(3)
        dh, dw = im.depth_map.shape
        sx, sy = f32(dw) / f32(im.width), f32(dh) / f32(im.height)
        K[0, 0] *= sx; K[0, 2] *= sx; K[1, 1] *= sy; K[1, 2] *= sy
        RT = np.concatenate([R, T[:, None]], 1)
        Pm = np.zeros((3, 4), f32)
        for r in range(3):
            for c in range(4):
                Pm[r, c] = f32(f32(K[r, 0] * RT[0, c]) + f32(K[r, 1] * RT[1, c])) + f32(K[r, 2] * RT[2, c])
        a, b, c_, d, e, f_, g, h, i = [f32(x) for x in Pm[:, :3].ravel()]
        A = f32(e * i) - f32(f_ * h); B = -(f32(d * i) - f32(f_ * g)); Cc = f32(d * h) - f32(e * g)
        det = f32(f32(a * A) + f32(b * B)) + f32(c_ * Cc)
        inv_det = f32(1.0) / det
        Mi = np.array([[A * inv_det, -(f32(b * i) - f32(c_ * h)) * inv_det, (f32(b * f_) - f32(c_ * e)) * inv_det],
                       [B * inv_det, (f32(a * i) - f32(c_ * g)) * inv_det, -(f32(a * f_) - f32(c_ * d)) * inv_det],
                       [Cc * inv_det, -(f32(a * h) - f32(b * g)) * inv_det, (f32(a * e) - f32(b * d)) * inv_det]], f32)
        inv = np.zeros((3, 4), f32)
        inv[:, :3] = Mi
        for r in range(3):
            inv[r, 3] = -(f32(f32(Mi[r, 0] * Pm[0, 3]) + f32(Mi[r, 1] * Pm[1, 3])) + f32(Mi[r, 2] * Pm[2, 3]))
        P.append(Pm); iP.append(inv); iR.append(R.T.copy())
        masks.append(np.zeros((dh, dw), bool) if im.mask is None else (np.asarray(im.mask) != 0))
    max_sq = f32(opt.max_reproj_error * opt.max_reproj_error)
    min_cos = f32(np.cos(opt.max_normal_error * 0.017453292519943295769))
    used = [im.used for im in images]
    fused = [False] * n
    pts, nrm, col, vis = [], [], [], []

    def dot4(row, x):
        return f32(f32(f32(row[0] * x[0]) + f32(row[1] * x[1])) + f32(row[2] * x[2])) + f32(row[3] * x[3])

    def fuse_pixel(i0, r0, c0):
        queue = [(i0, r0, c0, 0)]
        ref_pt = np.zeros(4, f32); ref_n = np.zeros(3, f32)
        acc = [[] for _ in range(9)]
        seen = set()
        while queue:
            ii, row, cc, lvl = queue.pop()
            im = images[ii]
            if masks[ii][row, cc]:
                continue
            depth = f32(im.depth_map[row, cc])
            if depth <= 0:
                continue
            if lvl > 0:
                proj = [dot4(P[ii][r], ref_pt) for r in range(3)]
                if abs(f32(f32(proj[2] - depth) / depth)) > opt.max_depth_error:
                    continue
                cd = f32(f32(proj[0] / proj[2]) - f32(cc)); rd = f32(f32(proj[1] / proj[2]) - f32(row))
                if f32(f32(cd * cd) + f32(rd * rd)) > max_sq:
                    continue
            nl = [f32(im.normal_map[k, row, cc]) for k in range(3)]
            normal = np.array([f32(f32(f32(iR[ii][r, 0] * nl[0]) + f32(iR[ii][r, 1] * nl[1])) + f32(iR[ii][r, 2] * nl[2]))
                               for r in range(3)], f32)
            if lvl > 0:
                cosn = f32(f32(f32(ref_n[0] * normal[0]) + f32(ref_n[1] * normal[1])) + f32(ref_n[2] * normal[2]))
                if cosn < min_cos:
                    continue
            hx, hy = f32(f32(cc) * depth), f32(f32(row) * depth)
            xyz = np.array([f32(f32(f32(f32(iP[ii][r, 0] * hx) + f32(iP[ii][r, 1] * hy)) + f32(iP[ii][r, 2] * depth)) +
                                f32(iP[ii][r, 3] * f32(1))) for r in range(3)], f32)
            color = (0, 0, 0)
            if im.rgb is not None:
                dh, dw = im.depth_map.shape
                sx, sy = f32(dw) / f32(im.width), f32(dh) / f32(im.height)
                xx = int(np.floor(float(f32(cc) / sx) + 0.5)); yy = int(np.floor(float(f32(row) / sy) + 0.5))
                if 0 <= xx < im.rgb.shape[1] and 0 <= yy < im.rgb.shape[0]:
                    color = tuple(int(v) for v in im.rgb[yy, xx])
            masks[ii][row, cc] = True
            lo, hi = opt.bounding_box
            if any(xyz[k] < f32(lo[k]) or xyz[k] > f32(hi[k]) for k in range(3)):
                continue
            for k in range(3):
                acc[k].append(xyz[k]); acc[3 + k].append(normal[k]); acc[6 + k].append(color[k])
            seen.add(ii)
            if lvl == 0:
                ref_pt = np.array([xyz[0], xyz[1], xyz[2], 1], f32); ref_n = normal
            if len(acc[0]) >= opt.max_num_pixels:
                break
            if lvl >= opt.max_traversal_depth - 1:
                continue
            for nxt in overlap[ii]:
                if not used[nxt] or fused[nxt]:
                    continue
                x4 = np.array([xyz[0], xyz[1], xyz[2], 1], f32)
                npj = [f32(f32(f32(f32(P[nxt][r, 0] * x4[0]) + f32(P[nxt][r, 1] * x4[1])) + f32(P[nxt][r, 2] * x4[2])) + P[nxt][r, 3])
                       for r in range(3)]
                with np.errstate(all="ignore"):
                    qc, qr = f32(npj[0] / npj[2]), f32(npj[1] / npj[2])
                if not (np.isfinite(qc) and np.isfinite(qr)):
                    continue
                ncol = int(np.floor(float(qc) + 0.5)) if qc >= 0 else -int(np.floor(-float(qc) + 0.5))
                nrow = int(np.floor(float(qr) + 0.5)) if qr >= 0 else -int(np.floor(-float(qr) + 0.5))
                dh2, dw2 = images[nxt].depth_map.shape
                if ncol < 0 or nrow < 0 or ncol >= dw2 or nrow >= dh2:
                    continue
                queue.append((nxt, nrow, ncol, lvl + 1))
        if len(acc[0]) < opt.min_num_pixels:
            return
        fn = np.array([f32(_median(acc[3 + k])) for k in range(3)], f32)
        norm = f32(np.sqrt(f32(f32(f32(fn[0] * fn[0]) + f32(fn[1] * fn[1])) + f32(fn[2] * fn[2]))))
        if norm < np.finfo(f32).eps:
            return
        pts.append([f32(_median(acc[k])) for k in range(3)])
        nrm.append([f32(fn[k] / norm) for k in range(3)])
        col.append([int(min(255, max(0, np.floor(float(f32(_median(acc[6 + k]))) + 0.5)))) for k in range(3)])
        vis.append(sorted(seen))

    idx = 0
    while idx >= 0:
        if used[idx]:
            dh, dw = images[idx].depth_map.shape
            for r in range(dh):
                for c in range(dw):
                    fuse_pixel(idx, r, c)
        fused[idx] = True
        nxt = -1
        for j in overlap[idx]:
            if used[j] and not fused[j]:
                nxt = j
                break
        if nxt < 0:
            nxt = next((j for j in range(n) if used[j] and not fused[j]), -1)
        idx = nxt
    return np.array(pts, f32).reshape(-1, 3), np.array(nrm, f32).reshape(-1, 3), np.array(col, np.uint8).reshape(-1, 3), vis


def _same(a, b):
    return (len(a.xyz) == len(b.xyz) and np.array_equal(a.xyz, b.xyz) and np.array_equal(a.normal, b.normal)
            and np.array_equal(a.rgb, b.rgb) and all(np.array_equal(u, v) for u, v in zip(a.visibility, b.visibility)))


def _on_surface(views, got, w, h):
    """every fused point reprojects onto the ground-truth depth of the images that saw it"""
    checked = 0
    for p, vis in zip(got.xyz, got.visibility):
        assert len(vis) >= 1 and len(set(vis)) == len(vis) and list(vis) == sorted(vis)
        for i in vis:
            v = views[i]
            pc = np.asarray(v.R, np.float64) @ p + np.asarray(v.T, np.float64)
            px = np.asarray(v.K, np.float64) @ (pc / pc[2])
            c, r = int(round(px[0])), int(round(px[1]))
            if 1 <= c < w - 1 and 1 <= r < h - 1:
                d = v.depth[r - 1:r + 2, c - 1:c + 2]
                assert np.min(np.abs(d - pc[2]) / pc[2]) < 0.03
                checked += 1
    return checked


# ------------------------------------------------------------------------------------------------
# the checker (CPU)
# ------------------------------------------------------------------------------------------------

def test_sequential_oracle_equals_python_restatement():
    views = scene(4, 32, 24)
    opt = fusion.StereoFusionOptions(min_num_pixels=3, max_reproj_error=1.5, max_depth_error=0.02)
    got = fusion_oracle.fuse(opt, _images(views), _overlap(4), mode=0)
    wp, wn, wc, wv = _fuse_reference(opt, _images(views), _overlap(4))
    assert len(got.xyz) == len(wp) > 50
    assert np.array_equal(got.xyz, wp) and np.array_equal(got.normal, wn) and np.array_equal(got.rgb, wc)
    assert all(list(a) == b for a, b in zip(got.visibility, wv))
    assert _same(fusion_oracle.fuse(opt, _images(views), _overlap(4), mode=0), got)  # deterministic


def test_seed_order_against_ground_truth_and_row_major():
    """mode 1 (the reference's Fuse, turns in the order of its pool schedule, one thread per ten-row stripe):
    points on the rendered surface, every pixel consumed at most once, and the same cloud as the row-major
    order up to which pixel of a neighbourhood gets its turn first."""
    views = scene(5, 64, 48)
    par = fusion_oracle.fuse(fusion.StereoFusionOptions(), _images(views), _overlap(5), mode=1)
    seq = fusion_oracle.fuse(fusion.StereoFusionOptions(), _images(views), _overlap(5), mode=0)
    assert len(par.xyz) > 300
    np.testing.assert_allclose(np.linalg.norm(par.normal, axis=1), 1.0, atol=1e-5)
    assert _on_surface(views, par, 64, 48) > 500
    assert sum(len(v) for v in par.visibility) <= 5 * len(par.xyz)
    # another legal order of the same turns: the clouds differ in a few per cent of the points
    assert abs(len(par.xyz) - len(seq.xyz)) <= 0.05 * len(seq.xyz)
    # nearest row-major point of every point: within about two pixel footprints (one pixel covers
    # ~0.3 scene units at the scene depth of ~16)
    d = np.sqrt(((par.xyz[:, None, :] - seq.xyz[None, :, :]) ** 2).sum(-1)).min(1)
    assert np.median(d) < 0.15 and np.percentile(d, 95) < 0.6, (np.median(d), np.percentile(d, 95))


def test_round_schedule_equals_sequential_turns():
    """mode 2 simulates fusion.hip's schedule (one sequential wave per pool thread, speculative passes over a
    window of ticks, tentative marks, rank cut, committed prefix) on the CPU with the waves interleaved at
    random: bit-identical to the sequential turns of mode 1 on every case the GPU test runs, in a handful of
    passes per image."""
    import ctypes as C
    for name in sorted(_CASES):
        opt, images, overlap = _case(name)
        want = fusion_oracle.fuse(opt, images, overlap, mode=1)
        got = fusion_oracle.fuse(opt, images, overlap, mode=2)
        assert _same(got, want), name
        rounds, walks = C.c_longlong(), C.c_longlong()
        fusion_oracle.lib().fuo_last_rounds(C.byref(rounds), C.byref(walks))
        n_img = sum(1 for im in images if im.used)
        seeds = sum(im.depth_map.size for im in images if im.used)
        assert rounds.value <= 8 * n_img and walks.value <= 1.5 * seeds, (name, rounds.value, walks.value, seeds)


def test_oracle_options_masks_and_bounding_box():
    _options_masks_and_bounding_box(lambda o, im, ov: fusion_oracle.fuse(o, im, ov, mode=1))
    _options_masks_and_bounding_box(lambda o, im, ov: fusion_oracle.fuse(o, im, ov, mode=0))


def _options_masks_and_bounding_box(fuse):
    views = scene(4, 48, 36)
    base = fuse(fusion.StereoFusionOptions(min_num_pixels=2), _images(views), _overlap(4))
    # a higher minimum support yields a subset-sized result
    strict = fuse(fusion.StereoFusionOptions(min_num_pixels=4), _images(views), _overlap(4))
    assert 0 < len(strict.xyz) < len(base.xyz)
    assert all(len(v) >= 1 for v in strict.visibility)
    # bounding box: no point outside
    lo, hi = (-0.3, -0.3, -10.0), (0.3, 0.3, 10.0)
    boxed = fuse(fusion.StereoFusionOptions(min_num_pixels=2, bounding_box=(lo, hi)), _images(views), _overlap(4))
    assert 0 < len(boxed.xyz) < len(base.xyz)
    assert (boxed.xyz >= np.array(lo, np.float32) - 1e-3).all() and (boxed.xyz <= np.array(hi, np.float32) + 1e-3).all()
    # masking all of image 0 removes it from every visibility list
    imgs = _images(views)
    imgs[0].mask = np.ones(views[0].gray.shape, np.uint8)
    masked = fuse(fusion.StereoFusionOptions(min_num_pixels=2), imgs, _overlap(4))
    assert all(0 not in v for v in masked.visibility) and len(masked.xyz) > 0
    # an unused image (missing inputs) is skipped the same way
    imgs = _images(views)
    imgs[0].used = False
    skipped = fuse(fusion.StereoFusionOptions(min_num_pixels=2), imgs, _overlap(4))
    assert all(0 not in v for v in skipped.visibility)
    # non-positive depths are never fused
    imgs = _images(views)
    for im in imgs:
        im.depth_map[:] = 0
    assert len(fuse(fusion.StereoFusionOptions(), imgs, _overlap(4)).xyz) == 0
    # option checks (fusion.cc:96-106)
    assert not fusion.StereoFusionOptions(min_num_pixels=10, max_num_pixels=5).Check()
    assert not fusion.StereoFusionOptions(max_traversal_depth=0).Check()
    with pytest.raises(ValueError):
        fuse(fusion.StereoFusionOptions(check_num_images=0), _images(views), _overlap(4))
    # max_num_pixels caps the support of a point
    capped = fuse(fusion.StereoFusionOptions(min_num_pixels=2, max_num_pixels=2), _images(views), _overlap(4))
    assert len(capped.xyz) > 0 and max(len(v) for v in capped.visibility) <= 2


# ------------------------------------------------------------------------------------------------
# the HIP path against the checker (bit for bit)
# ------------------------------------------------------------------------------------------------

_CASES = {
    "defaults_5x64x48": (5, 64, 48, dict(), {}),
    "tight_4x32x24": (4, 32, 24, dict(min_num_pixels=3, max_reproj_error=1.5, max_depth_error=0.02), {}),
    "bbox": (4, 48, 36, dict(min_num_pixels=2, bounding_box=((-0.3, -0.3, -10.0), (0.3, 0.3, 10.0))), {}),
    "cap2": (4, 48, 36, dict(min_num_pixels=2, max_num_pixels=2), {}),
    "depth1": (4, 48, 36, dict(min_num_pixels=1, max_traversal_depth=1), {}),
    "depth2": (5, 48, 36, dict(min_num_pixels=2, max_traversal_depth=2), {}),
    "mask0": (4, 48, 36, dict(min_num_pixels=2), {"mask": 0}),
    "unused1": (4, 48, 36, dict(min_num_pixels=2), {"unused": 1}),
    "no_rgb": (4, 48, 36, dict(min_num_pixels=2), {"no_rgb": True}),
    "loose_7x80x60": (7, 80, 60, dict(min_num_pixels=2, max_reproj_error=4.0, max_depth_error=0.05,
                                      max_normal_error=30.0), {}),
    "sparse_overlap": (6, 64, 48, dict(min_num_pixels=2), {"ring": True}),
    "half_size_maps": (4, 64, 48, dict(min_num_pixels=2), {"half": True}),
}


def _case(name):
    n, w, h, okw, extra = _CASES[name]
    views = scene(n, w, h)
    images = _images(views, with_rgb=not extra.get("no_rgb"))
    if "mask" in extra:
        m = np.zeros((h, w), np.uint8)
        m[::2, 1::3] = 1
        images[extra["mask"]].mask = m
    if "unused" in extra:
        images[extra["unused"]].used = False
    if extra.get("half"):  # depth maps at half the model image size (max_image_size in the workspace)
        for im in images:
            im.depth_map = np.ascontiguousarray(im.depth_map[::2, ::2])
            im.normal_map = np.ascontiguousarray(im.normal_map[:, ::2, ::2])
    overlap = [[(i + 1) % n, (i + 2) % n] for i in range(n)] if extra.get("ring") else _overlap(n)
    return fusion.StereoFusionOptions(**okw), images, overlap


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_CASES))
def test_hip_fusion_equals_parallel_oracle(name):
    opt, images, overlap = _case(name)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    got = fusion.fuse(opt, images, overlap)
    assert len(want.xyz) > 20
    assert _same(got, want), (len(got.xyz), len(want.xyz))
    assert _same(fusion.fuse(opt, images, overlap), got)  # marks and ranks decide, not timing


@pytest.mark.gpu
def test_hip_fusion_depth_first_walks_and_the_fallback_to_them():
    """The walks of the default build are breadth-first where that is provably the depth-first result
    (colmap_amd/csrc/fusion.hip: walk_turn_wide, seed_group). COLMAP_AMD_FUSION_WIDE=0 runs the depth-first walk alone:
    same cloud; and with max_traversal_depth = 20 on noisy maps many breadth-first walks meet the bound, take their
    marks back and are repeated depth-first: same cloud again (fusion_last_redone_walks says how many)."""
    import ctypes as C
    from colmap_amd._lib import lib
    from switches import switches
    rng = np.random.default_rng(1)
    images = _images(scene(4, 24, 160))
    for im in images:
        im.depth_map = (im.depth_map * (1 + 0.01 * rng.standard_normal(im.depth_map.shape))).astype(np.float32)
    loose = dict(min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05, max_normal_error=30.0)
    lib().fusion_last_redone_walks.restype = C.c_int64
    for kw in (dict(), dict(max_traversal_depth=20)):
        opt = fusion.StereoFusionOptions(**loose, **kw)
        want = fusion_oracle.fuse(opt, images, _overlap(4), mode=1)
        assert len(want.xyz) > 1500
        assert _same(fusion.fuse(opt, images, _overlap(4)), want)
        redone = lib().fusion_last_redone_walks()
        assert (redone > 20) == bool(kw), redone
        with switches(lib(), COLMAP_AMD_FUSION_WIDE=0):
            assert _same(fusion.fuse(opt, images, _overlap(4)), want)
            assert lib().fusion_last_redone_walks() == 0


@pytest.mark.gpu
def test_hip_fusion_many_seeds_per_lane():
    """24 pool threads (waves) over 230 k pixels, 55 k points."""
    views = scene(3, 320, 240)
    opt = fusion.StereoFusionOptions(min_num_pixels=2)
    want = fusion_oracle.fuse(opt, _images(views), _overlap(3), mode=1)
    got = fusion.fuse(opt, _images(views), _overlap(3))
    assert len(want.xyz) > 10000 and _same(got, want)
    assert _on_surface(views, got, 320, 240) > 10000


@pytest.mark.gpu
@pytest.mark.parametrize("num_threads", [1, 3, -1])
def test_hip_fusion_pool_sizes_and_cut_passes(num_threads):
    """Noisy maps of a narrow image: the stripes of one window meet each other's marks and passes are cut
    (tests/test_fusion_emul.py shows the simulation's conflict counts for this input). num_threads = 1 is the
    reference's sequential run = the checker's row-major mode 0."""
    rng = np.random.default_rng(1)
    images = _images(scene(4, 24, 160))
    for im in images:
        im.depth_map = (im.depth_map * (1 + 0.01 * rng.standard_normal(im.depth_map.shape))).astype(np.float32)
    opt = fusion.StereoFusionOptions(num_threads=num_threads, min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05,
                                     max_normal_error=30.0)
    want = fusion_oracle.fuse(opt, images, _overlap(4), mode=1)
    got = fusion.fuse(opt, images, _overlap(4))
    assert len(want.xyz) > 2000 and _same(got, want)
    if num_threads == 1:
        assert _same(got, fusion_oracle.fuse(opt, images, _overlap(4), mode=0))


@pytest.mark.gpu
def test_hip_fusion_options_masks_and_bounding_box():
    _options_masks_and_bounding_box(fusion.fuse)


def test_fusion_needs_a_gpu():
    """No CPU path: without a device fusion_run reports it."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    views = scene(2, 16, 12)
    with pytest.raises(RuntimeError, match="no HIP device"):
        fusion.fuse(fusion.StereoFusionOptions(), _images(views), _overlap(2))


def test_ply_and_visibility_files(tmp_path):
    pts = fusion.FusedPoints(np.arange(12, dtype=np.float32).reshape(4, 3), np.eye(4, 3, dtype=np.float32),
                             np.arange(12, dtype=np.uint8).reshape(4, 3), [np.array([0, 2]), np.array([1]), np.array([], int), np.array([3, 4, 5])])
    p = str(tmp_path / "fused.ply")
    fusion.write_binary_ply_points(p, pts)
    raw = open(p, "rb").read()
    assert raw.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\n")
    assert len(raw.split(b"end_header\n", 1)[1]) == 4 * (6 * 4 + 3)          # util/ply.cc:412-428
    back = fusion.read_binary_ply_points(p)
    assert np.array_equal(back.xyz, pts.xyz) and np.array_equal(back.normal, pts.normal) and np.array_equal(back.rgb, pts.rgb)
    fusion.write_points_visibility(p + ".vis", pts.visibility)
    assert os.path.getsize(p + ".vis") == 8 + 4 * 4 + 4 * 6                  # fusion.cc:526-541
    vis = fusion.read_points_visibility(p + ".vis", 4)
    assert all(np.array_equal(a, b) for a, b in zip(vis, pts.visibility))
    with pytest.raises(ValueError):
        fusion.read_points_visibility(p + ".vis", 5)


@pytest.mark.gpu
def test_stereo_fusion_command_on_a_workspace(tmp_path):
    """exe/mvs.cc:299-386: workspace with geometric maps + fusion.cfg -> fused.ply + fused.ply.vis."""
    views = scene(5, 64, 48)
    ws = str(tmp_path / "dense")
    names = write_dense_workspace(ws, views)
    w = W.Workspace(ws)
    for i, v in enumerate(views):
        for path, arr in ((w.GetDepthMapPath(i, "geometric"), v.depth), (w.GetNormalMapPath(i, "geometric"), v.normal)):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            mvs.write_mat(path, arr)
    with open(os.path.join(ws, "stereo", "fusion.cfg"), "w") as f:
        f.write("\n".join(names[:4]) + "\nmissing.png\n" if False else "\n".join(names[:4]) + "\n")
    out = str(tmp_path / "fused.ply")
    assert fusion.main(["--workspace_path", ws, "--output_path", out, "--StereoFusion.min_num_pixels", "3"]) == 0
    pts = fusion.read_binary_ply_points(out)
    vis = fusion.read_points_visibility(out + ".vis", len(pts.xyz))
    assert len(pts.xyz) > 200 and all(set(v) <= {0, 1, 2, 3} for v in vis)      # image 4 is not in fusion.cfg
    # grey images: r = g = b = the bitmap value at the pixel
    assert (pts.rgb[:, 0] == pts.rgb[:, 1]).all() and (pts.rgb[:, 1] == pts.rgb[:, 2]).all()
    # the same through the class, and as a model (output_type BIN): sparse points replaced by the fused ones
    fuser = fusion.StereoFusion(fusion.StereoFusionOptions(min_num_pixels=3), ws)
    fuser.Run()
    assert np.array_equal(fuser.GetFusedPoints().xyz, pts.xyz)
    out_model = str(tmp_path / "fused_model")
    assert fusion.main(["--workspace_path", ws, "--output_path", out_model, "--output_type", "BIN",
                        "--StereoFusion.min_num_pixels", "3"]) == 0
    sm = W.read_sparse_model(out_model)
    assert len(sm.points3D) == len(pts.xyz) and len(sm.images) == 5
    np.testing.assert_array_equal(sm.points3D[1].xyz.astype(np.float32), pts.xyz[0])
    with pytest.raises(SystemExit):
        fusion.main(["--workspace_path", ws, "--output_path", out, "--input_type", "depth"])
    # pycolmap.stereo_fusion(output_path, workspace_path, ..., options, output_type) (pycolmap/pipeline/mvs.cc:182-193)
    from colmap_amd import pipeline
    out2 = str(tmp_path / "fused2.ply")
    again = pipeline.stereo_fusion(out2, ws, options=fusion.StereoFusionOptions(min_num_pixels=3), output_type="ply")
    assert np.array_equal(again.xyz, pts.xyz) and open(out2, "rb").read() == open(out, "rb").read()


# ------------------------------------------------------------------------------------------------
# the reference's own test of this path (mvs/fusion_test.cc:45-140), restated
# ------------------------------------------------------------------------------------------------

def _reference_integration_inputs():
    """StereoFusion.Integration: SynthesizeDataset with one rig, one camera, two frames, 30 x 20 images (the
    default SIMPLE_RADIAL parameters stay {1280, 512, 384, 0.05}), constant depth 5, normals (0, 0, 1), bitmap
    colour (0, 64, 128); min_num_pixels 1, max_num_pixels 100, max_traversal_depth 10, check_num_images 10."""
    from colmap_amd import scene as S
    rec = S.SynthesizeDataset(S.SyntheticDatasetOptions(num_rigs=1, num_cameras_per_rig=1, num_frames_per_rig=2,
                                                        camera_width=30, camera_height=20), seed=0)
    images = []
    for iid in rec.RegImageIds():
        im = rec.images[iid]
        cam = rec.cameras[im.camera_id]
        K = np.array([[cam.params[0], 0, cam.params[1]], [0, cam.params[0], cam.params[2]], [0, 0, 1]], np.float32)
        R = S.quat_to_rot(im.cam_from_world[:4]).astype(np.float32)
        T = im.cam_from_world[4:].astype(np.float32)
        rgb = np.zeros((20, 30, 3), np.uint8)
        rgb[:] = (0, 64, 128)
        normal = np.zeros((3, 20, 30), np.float32)
        normal[2] = 1.0
        images.append(fusion.FusionImage(30, 20, K, R, T, rgb, np.full((20, 30), 5.0, np.float32), normal))
    opt = fusion.StereoFusionOptions(min_num_pixels=1, max_num_pixels=100, max_traversal_depth=10, check_num_images=10)
    return opt, images, [[1], [0]]


def _check_reference_integration(pts):
    assert len(pts.xyz) > 0 and len(pts.xyz) == len(pts.visibility)       # EXPECT_GT(size, 0), sizes equal
    assert (pts.xyz > -10.0).all() and (pts.xyz < 10.0).all()              # every coordinate in (-10, 10)
    assert (pts.rgb == np.array([0, 64, 128], np.uint8)).all()            # colour of the bitmap
    np.testing.assert_allclose((pts.normal.astype(np.float64) ** 2).sum(1), 1.0, rtol=4 * np.finfo(np.float32).eps)
    assert all(len(v) > 0 for v in pts.visibility)


def test_reference_integration_case_oracle():
    opt, images, overlap = _reference_integration_inputs()
    for mode in (0, 1, 2):
        _check_reference_integration(fusion_oracle.fuse(opt, images, overlap, mode=mode))


@pytest.mark.gpu
def test_reference_integration_case_hip():
    opt, images, overlap = _reference_integration_inputs()
    got = fusion.fuse(opt, images, overlap)
    _check_reference_integration(got)
    assert _same(got, fusion_oracle.fuse(opt, images, overlap, mode=1))


def test_reference_visibility_file_cases(tmp_path):
    """ReadPointsVisibility.RoundTrip / SizeMismatch (mvs/fusion_test.cc:142-175), same data."""
    expected = [[0, 1, 2], [1, 3], [], [0, 2, 3, 4]]
    p = str(tmp_path / "test.vis")
    fusion.write_points_visibility(p, expected)
    actual = fusion.read_points_visibility(p, len(expected))
    assert [list(a) for a in actual] == expected
    fusion.write_points_visibility(p, [[0, 1], [2]])
    with pytest.raises(ValueError):
        fusion.read_points_visibility(p, 5)
