"""GPU parity tests: the HIP PatchMatch path (through the C ABI) against the CPU oracle.

The arithmetic of the path is specified operation by operation (oracle/pm_oracle.c
header), so the bar is BIT-EXACT equality of every output map, not a tolerance:
a stochastic argmin algorithm amplifies one-ulp differences into different depth
maps, so anything weaker than exact equality would not be a meaningful check.
The oracle runs in `order=1` (the kernel's evaluation order of the NCC sums);
tests/test_pm_oracle.py bounds the difference between that order and the
reference's sequential order (`order=0`).
"""
import ctypes as C
import os

import numpy as np
import pytest

from pm_common import scene, oracle_inputs, hip_problem, paired_options
from colmap_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _run_both(pm_oracle, views, ref, src, maps=None, **kw):
    from colmap_amd import mvs
    dmin, dmax = syn.depth_range(views, ref)
    o, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, **kw)
    imgs = oracle_inputs(views, maps is not None, maps)
    want = pm_oracle.run(o, imgs, ref, src, want_cost=True)
    pm = mvs.PatchMatch(h, hip_problem(views, ref, src, maps))
    pm.Run()
    got = dict(depth=pm.GetDepthMap(), normal=pm.GetNormalMap(), sel_prob=pm.GetSelProbMap(),
               cost=pm.GetCostMap(), mask=pm.GetConsistencyMask())
    return want, got, pm


def _assert_equal(want, got, keys=("depth", "normal", "cost", "sel_prob", "mask")):
    for k in keys:
        a, b = want[k], got[k]
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError(f"{k}: {len(bad)} of {a.size} values differ, first at {bad[0]}: "
                                 f"oracle {a[tuple(bad[0])]!r} hip {b[tuple(bad[0])]!r}")


def test_pose_tables_and_ref_filter(pm_oracle):
    from colmap_amd import mvs
    views = scene()
    _, h = paired_options(pm_oracle, depth_min=1.0, depth_max=10.0, geom_consistency=0, filter=0)
    pm = mvs.PatchMatch(h, hip_problem(views, 2, [0, 1, 3, 4]))
    pm.Create()
    poses, K, iK = pm.GetPoseTables()
    o_poses, o_K, o_iK = pm_oracle.pose_tables(oracle_inputs(views), 2, [0, 1, 3, 4])
    assert np.array_equal(K, o_K) and np.array_equal(iK, o_iK)
    assert np.array_equal(poses, o_poses)
    img, s, ss = pm.GetRefFilter()
    o_img, o_s, o_ss = pm_oracle.filter_ref_image(views[2].gray, 5, 1, 5.0, float(np.float32(0.2)))
    assert np.array_equal(img, o_img) and np.array_equal(s, o_s) and np.array_equal(ss, o_ss)


def test_initial_state_and_cost(pm_oracle):
    """PRNG seeding, random depth/normal initialisation, ComputeInitialCost."""
    views = scene()
    want, got, _ = _run_both(pm_oracle, views, 2, [0, 1, 3, 4], geom_consistency=0, filter=0,
                             max_sweeps=0)
    _assert_equal(want, got, ("depth", "normal", "cost"))


@pytest.mark.parametrize("nsweeps", [1, 2, 3, 4])
def test_each_sweep_direction(pm_oracle, nsweeps):
    """Every sweep direction of the virtual rotation against the oracle's physical rotation."""
    views = scene()
    want, got, _ = _run_both(pm_oracle, views, 2, [0, 1, 3, 4], geom_consistency=0, filter=0,
                             max_sweeps=nsweeps)
    _assert_equal(want, got, ("depth", "normal", "cost", "sel_prob"))


def test_full_photometric_with_filter(pm_oracle):
    views = scene()
    want, got, pm = _run_both(pm_oracle, views, 2, [0, 1, 3, 4], geom_consistency=0, filter=1)
    _assert_equal(want, got)
    # consistency graph list (GetConsistentImageIdxs, reference patch_match_cuda.cu:1367-1391)
    flat = pm.GetConsistentImageIdxs()
    mask = want["mask"]
    src = [0, 1, 3, 4]
    expect = []
    for r in range(mask.shape[1]):
        for c in range(mask.shape[2]):
            ids = [src[d] for d in range(mask.shape[0]) if mask[d, r, c]]
            if ids:
                expect += [c, r, len(ids)] + ids
    assert np.array_equal(flat, np.array(expect, np.int32))
    ms, n = pm.GetSweepTiming()
    assert n == 20 and ms > 0


def test_geometric_consistency_and_filter(pm_oracle):
    views = scene(3, 80, 60)
    maps = []
    for ref in range(3):
        dmin, dmax = syn.depth_range(views, ref)
        o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=0,
                                      num_iterations=1, order=1)
        r = pm_oracle.run(o, oracle_inputs(views), ref, [i for i in range(3) if i != ref])
        maps.append((r["depth"], r["normal"]))
    want, got, _ = _run_both(pm_oracle, views, 1, [0, 2], maps=maps, geom_consistency=1, filter=1,
                             num_iterations=2)
    _assert_equal(want, got)
    want, got, _ = _run_both(pm_oracle, views, 1, [0, 2], maps=maps, geom_consistency=1, filter=0,
                             num_iterations=1)
    _assert_equal(want, got)


@pytest.mark.parametrize("radius,step", [(3, 1), (5, 2), (8, 1), (4, 2)])
def test_window_shapes(pm_oracle, radius, step):
    views = scene(4, 64, 48)
    want, got, _ = _run_both(pm_oracle, views, 1, [0, 2, 3], geom_consistency=0, filter=1,
                             window_radius=radius, window_step=step, num_iterations=1)
    _assert_equal(want, got)


@pytest.mark.parametrize("cols,threads", [(1, 64), (3, 64), (4, 128), (8, 256), (16, 64)])
def test_group_shapes_do_not_change_results(pm_oracle, cols, threads):
    """Ragged image width (67 is not a multiple of any group width) and every
    workgroup geometry give the same bits."""
    views = scene(4, 67, 45)
    want, got, _ = _run_both(pm_oracle, views, 1, [0, 2, 3], geom_consistency=0, filter=1,
                             num_iterations=1, columns_per_group=cols, threads_per_group=threads)
    _assert_equal(want, got)


@pytest.mark.parametrize("fp_global", ["0", "1"])
@pytest.mark.parametrize("geom", [0, 1])
def test_four_wave_workgroups_ragged_width_both_addressing_modes(pm_oracle, request, geom, fp_global):
    """pm_sweep_quad_kernel (four waves per workgroup sharing the read-only LDS tables, 64 task slots per batch)
    against the oracle: ragged width (67 columns = 33 groups of two + one column; 34 groups = 8 workgroups + 2 waves),
    S = 6; packed images addressed through the buffer resource (the default) or by explicit indices (what problems
    whose images lie more than 4 GB apart and cannot be re-homed get)."""
    from colmap_amd import mvs
    from switches import set_switch
    set_switch(mvs.lib(), "COLMAP_AMD_PM_FP_GLOBAL", fp_global)
    request.addfinalizer(lambda: set_switch(mvs.lib(), "COLMAP_AMD_PM_FP_GLOBAL", None))
    views = scene(7, 67, 45)
    maps = None
    if geom:
        maps = [(v.depth.copy(), v.normal.copy()) for v in views]
    want, got, pm = _run_both(pm_oracle, views, 3, [0, 1, 2, 4, 5, 6], maps=maps, geom_consistency=geom, filter=1,
                              num_iterations=1)
    _assert_equal(want, got)
    assert pm.GetSweepKernelName() == "pm_sweep_quad_kernel" + (" (explicit indices)" if fp_global == "1" else "")


@pytest.mark.parametrize("geom", [0, 1])
def test_two_waves_per_column_pair_kernel(pm_oracle, request, geom):
    """pm_sweep_pair_kernel (a helper wave shares pass B of the NCC phases: what a lone large problem runs, chosen by
    occupancy in RunBatchAsync, forced here) against the oracle: ragged width, S = 6."""
    from colmap_amd import mvs
    from switches import set_switch
    set_switch(mvs.lib(), "COLMAP_AMD_PM_HELP", "2")
    request.addfinalizer(lambda: set_switch(mvs.lib(), "COLMAP_AMD_PM_HELP", None))
    views = scene(7, 67, 45)
    maps = [(v.depth.copy(), v.normal.copy()) for v in views] if geom else None
    want, got, pm = _run_both(pm_oracle, views, 3, [0, 1, 2, 4, 5, 6], maps=maps, geom_consistency=geom, filter=1,
                              num_iterations=1)
    _assert_equal(want, got)
    assert pm.GetSweepKernelName() == "pm_sweep_pair_kernel"


@pytest.mark.parametrize("wave", ["0", "1"])
def test_generic_kernel_equals_wave_kernels(pm_oracle, wave):
    """The generic sweep kernel (any window size; explicit strip indices through global loads) and the 11 x 11 wave
    kernels (indices formed by the address unit from a swizzled buffer resource) read the same packed images: the
    5 x 5 window (window_radius 2) only runs the former, the default window the latter -- both against the oracle,
    and the handle reports which kernel ran."""
    views = scene(5, 67, 45)
    radius = 2 if wave == "0" else 5
    want, got, pm = _run_both(pm_oracle, views, 2, [0, 1, 3, 4], geom_consistency=0, filter=1, num_iterations=1,
                              window_radius=radius)
    _assert_equal(want, got)
    assert pm.GetSweepKernelName() == ("pm_sweep_kernel" if wave == "0" else "pm_sweep_quad_kernel")


def test_single_source_and_many_samples(pm_oracle):
    views = scene(4, 64, 48)
    want, got, _ = _run_both(pm_oracle, views, 1, [2], geom_consistency=0, filter=1,
                             filter_min_num_consistent=1, num_iterations=1, num_samples=25)
    _assert_equal(want, got)


def test_sources_larger_than_reference_slot(pm_oracle):
    """Source images of unequal size share a max-size slot (reference :1596-1622)."""
    big = scene(3, 96, 72)
    small = scene(3, 64, 48)
    views = [small[0], big[1], big[2]]
    want, got, _ = _run_both(pm_oracle, views, 1, [0, 2], geom_consistency=0, filter=0,
                             num_iterations=1)
    _assert_equal(want, got)


def test_batched_run_equals_single_runs(pm_oracle):
    """pm_run_batch: three different reference images solved by shared launches give the
    same bits as the oracle run on each problem separately."""
    from colmap_amd import mvs
    views = scene(5, 67, 45)
    probs = [(1, [0, 2, 3]), (2, [1, 3, 4]), (3, [1, 2, 4])]
    pms, wants = [], []
    for ref, src in probs:
        dmin, dmax = syn.depth_range(views, ref)
        o, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1,
                              num_iterations=1, columns_per_group=2, threads_per_group=128)
        wants.append(pm_oracle.run(o, oracle_inputs(views), ref, src, want_cost=True))
        pms.append(mvs.PatchMatch(h, hip_problem(views, ref, src)))
    mvs.run_batch(pms)
    for pm, want in zip(pms, wants):
        got = dict(depth=pm.GetDepthMap(), normal=pm.GetNormalMap(), sel_prob=pm.GetSelProbMap(),
                   cost=pm.GetCostMap(), mask=pm.GetConsistencyMask())
        _assert_equal(want, got)
    # mismatched shapes are rejected
    other = scene(4, 64, 48)
    dmin, dmax = syn.depth_range(other, 1)
    _, h2 = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1,
                           num_iterations=1)
    with pytest.raises(mvs.PatchMatchError):
        mvs.run_batch([pms[0], mvs.PatchMatch(h2, hip_problem(other, 1, [0, 2, 3]))])


def test_concurrent_runs_from_host_threads_share_launches(pm_oracle):
    """pm_run called from several host threads at once -- how the reference's controller drives the seam, one
    problem per worker thread (mvs/patch_match.cc:190-204) -- is coalesced into batched launches: the results are the
    oracle's bits for every problem, and at least two of the calls shared their sweep launches. A problem of another
    shape that calls at the same time is solved on its own."""
    import threading
    from colmap_amd import mvs
    views = scene(6, 67, 45)
    probs = [(1, [0, 2, 3]), (2, [1, 3, 4]), (3, [1, 2, 4]), (4, [2, 3, 5])]
    pms, wants = [], []
    for ref, src in probs:
        dmin, dmax = syn.depth_range(views, ref)
        o, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1, num_iterations=1)
        wants.append(pm_oracle.run(o, oracle_inputs(views), ref, src, want_cost=True))
        pms.append(mvs.PatchMatch(h, hip_problem(views, ref, src)))
    other = scene(4, 40, 30)
    dmin, dmax = syn.depth_range(other, 1)
    o2, h2 = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1, num_iterations=1)
    wants.append(pm_oracle.run(o2, oracle_inputs(other), 1, [0, 2, 3], want_cost=True))
    pms.append(mvs.PatchMatch(h2, hip_problem(other, 1, [0, 2, 3])))
    for pm in pms:
        pm.Create()
    gate, errs = threading.Barrier(len(pms)), []

    def work(pm):
        try:
            gate.wait()
            pm.Run()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(pm,)) for pm in pms]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for pm, want in zip(pms, wants):
        got = dict(depth=pm.GetDepthMap(), normal=pm.GetNormalMap(), sel_prob=pm.GetSelProbMap(),
                   cost=pm.GetCostMap(), mask=pm.GetConsistencyMask())
        _assert_equal(want, got)
    shapes = [pm.GetLaunchShape()[0] for pm in pms]
    assert max(shapes[:4]) >= 2, shapes          # calls that arrived together shared launches
    assert shapes[4] == 1, shapes                # the odd shape ran alone


def test_image_cache_shares_sources_without_changing_results(pm_oracle):
    """pm_create_cached: problems that name the same bitmaps gather from one packed copy; the
    outputs are the bits of the uncached run, and eviction never drops an image in use."""
    from colmap_amd import mvs
    views = scene(5, 67, 45)
    images = hip_problem(views, 1, [0, 2]).images
    probs = [(1, [0, 2, 3]), (2, [1, 3, 4]), (3, [1, 2, 4])]
    cache = mvs.ImageCache(0)
    cached, plain = [], []
    for ref, src in probs:
        dmin, dmax = syn.depth_range(views, ref)
        _, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1,
                              num_iterations=1)
        cached.append(mvs.PatchMatch(h, mvs.PatchMatch.Problem(ref, src, images), cache))
        plain.append(mvs.PatchMatch(h, mvs.PatchMatch.Problem(ref, src, images)))
    mvs.run_batch(cached)
    mvs.run_batch(plain)
    st = cache.stats()
    assert st["entries"] == 5 and st["misses"] == 5 and st["hits"] == 4
    for a, b in zip(cached, plain):
        np.testing.assert_array_equal(a.GetDepthMap(), b.GetDepthMap())
        np.testing.assert_array_equal(a.GetNormalMap(), b.GetNormalMap())
        np.testing.assert_array_equal(a.GetSelProbMap(), b.GetSelProbMap())
    # capacity 0: only entries no live problem uses may go
    cached[1].close(); cached[2].close()
    cache.set_capacity(0)
    assert cache.stats()["entries"] == 3           # sources 0, 2, 3 of the live problem stay
    again = mvs.PatchMatch(cached[0].options_, cached[0].problem_, cache)   # three hits
    again.Run()
    np.testing.assert_array_equal(again.GetDepthMap(), plain[0].GetDepthMap())
    assert cache.stats()["hits"] == 7
    again.close()
    cached[0].close()
    cache.set_capacity(0)
    assert cache.stats()["entries"] == 0
    cache.close()


def test_cached_images_in_distant_slabs_are_rehomed(pm_oracle):
    """Packed source images come out of slabs, and the images of one problem must lie within one buffer resource's
    span (4 GB on the hardware). In the allocator's test mode -- four images per slab, the span = one slab -- a walk
    along seven views makes problems whose cached sources sit in two slabs: those are copied into one slab
    (pm_api.cpp RehomeSourceImages), every problem keeps the buffer-resource kernels, and the results are the bits
    of the uncached runs."""
    from colmap_amd import mvs
    L = mvs.lib()
    L.pm_debug_set_image_slab_slots.restype = C.c_ulonglong
    mvs.release_cached_memory()
    before = L.pm_debug_set_image_slab_slots(C.c_size_t(4))
    try:
        views = scene(7, 35, 27)
        images = hip_problem(views, 1, [0, 2]).images
        probs = [(1, [0, 2, 3]), (4, [3, 5, 6]), (2, [0, 3, 6]), (5, [1, 4, 6])]
        cache = mvs.ImageCache(0)
        for ref, src in probs:
            dmin, dmax = syn.depth_range(views, ref)
            _, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1,
                                  num_iterations=1)
            a = mvs.PatchMatch(h, mvs.PatchMatch.Problem(ref, src, images), cache)
            b = mvs.PatchMatch(h, mvs.PatchMatch.Problem(ref, src, images))
            a.Run()
            b.Run()
            assert "explicit" not in a.GetSweepKernelName() and "explicit" not in b.GetSweepKernelName()
            np.testing.assert_array_equal(a.GetDepthMap(), b.GetDepthMap())
            np.testing.assert_array_equal(a.GetNormalMap(), b.GetNormalMap())
            np.testing.assert_array_equal(a.GetSelProbMap(), b.GetSelProbMap())
            a.close()
            b.close()
        cache.close()
        assert L.pm_debug_set_image_slab_slots(C.c_size_t(4)) > before   # and the path was taken
    finally:
        mvs.release_cached_memory()
        L.pm_debug_set_image_slab_slots(C.c_size_t(0))


def test_error_behaviour():
    from colmap_amd import mvs
    views = scene(3, 64, 48)
    opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=1.0, depth_max=5.0, sigma_spatial=5.0,
                                geom_consistency=False)
    with pytest.raises(mvs.PatchMatchError):
        mvs.PatchMatch(opt, hip_problem(views, 1, [1, 2])).Run()      # ref as source
    with pytest.raises(mvs.PatchMatchError):
        mvs.PatchMatch(opt, hip_problem(views, 1, [])).Run()          # no sources
    bad = mvs.PatchMatchOptions(gpu_index="0,1", depth_min=1.0, depth_max=5.0, sigma_spatial=5.0)
    with pytest.raises(mvs.PatchMatchError):
        mvs.PatchMatch(bad, hip_problem(views, 1, [0, 2])).Run()      # exactly one GPU index
    geo = mvs.PatchMatchOptions(gpu_index="0", depth_min=1.0, depth_max=5.0, sigma_spatial=5.0,
                                geom_consistency=True)
    with pytest.raises(mvs.PatchMatchError):
        mvs.PatchMatch(geo, hip_problem(views, 1, [0, 2])).Run()      # missing depth/normal maps
    win = mvs.PatchMatchOptions(gpu_index="0", depth_min=1.0, depth_max=5.0, sigma_spatial=5.0,
                                geom_consistency=False, window_step=3)
    with pytest.raises(mvs.PatchMatchError):
        mvs.PatchMatch(win, hip_problem(views, 1, [0, 2])).Run()      # window_step <= 2


def test_device_xorwow_equals_rocrand_device():
    """rng_init / rng_uniform of the product kernels against rocrand_init(seed, 0, 0) /
    rocrand_uniform executed on the same GPU (tests/hip/rocrand_pin.hip), 1006 seeds x 40 draws."""
    import ctypes as C
    from colmap_amd._lib import lib
    from test_pm_oracle import _pin_seeds, rocrand_pin_lib
    seeds, nd = _pin_seeds(), 40
    want = np.zeros((len(seeds), nd), np.float32)
    got = np.zeros_like(want)
    assert rocrand_pin_lib().rocrand_pin_device(seeds.ctypes.data_as(C.c_void_p), len(seeds), nd,
                                                want.ctypes.data_as(C.c_void_p)) == 0
    assert lib().pm_debug_rng_streams(0, seeds.ctypes.data_as(C.c_void_p), len(seeds), nd,
                                      got.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- BASELINE.json shapes against the oracle (S = 20 / M = 15 was only property-tested) ----

def test_baseline_source_count_s20_m15_photometric(pm_oracle):
    """config[1]'s view count and sample count (S = 20 sources, 15 Monte-Carlo samples, 5 x 4 sweeps,
    photometric + filter: task lists of up to 4 x 15 + 20 entries per column, 20 pose records,
    20-wide CDF) on 96 x 72 images, bit-compared with the oracle."""
    views = scene(22, 96, 72, 3.6 * 21)
    ref = 10
    src = [i for i in range(21) if i != ref]
    assert len(src) == 20
    want, got, _ = _run_both(pm_oracle, views, ref, src, geom_consistency=0, filter=1)
    _assert_equal(want, got)


def test_baseline_source_count_s20_m15_geometric(pm_oracle):
    """The same shape with the geometric consistency term and both filters (config[2]'s pass 2);
    source depth / normal maps = the renderer's ground truth."""
    views = scene(22, 96, 72, 3.6 * 21)
    ref = 11
    src = [i for i in range(1, 22) if i != ref]
    maps = [(v.depth, v.normal) for v in views]
    want, got, _ = _run_both(pm_oracle, views, ref, src, maps=maps, geom_consistency=1, filter=1,
                             num_iterations=2)
    _assert_equal(want, got)


def test_baseline_config0_640x480_two_pass(pm_oracle):
    """BASELINE.json config[0]: 3 pinhole images 640 x 480 (f = 600), S = 2, default options --
    photometric pass for every image, then the geometric pass on the middle one, all bit-compared."""
    from colmap_amd import mvs
    views = syn.make_scene(3, 640, 480, focal=600.0, arc_deg=8.0)
    maps = []
    for ref in range(3):
        src = [i for i in range(3) if i != ref]
        want, got, _ = _run_both(pm_oracle, views, ref, src, geom_consistency=0, filter=0)
        _assert_equal(want, got, ("depth", "normal", "cost", "sel_prob"))
        maps.append((got["depth"], got["normal"]))
    want, got, _ = _run_both(pm_oracle, views, 1, [0, 2], maps=maps, geom_consistency=1, filter=1)
    _assert_equal(want, got)
    kept = got["depth"] > 0
    rel = np.abs(got["depth"][kept] - views[1].depth[kept]) / views[1].depth[kept]
    # with S = 2 the filter needs BOTH sources consistent (filter_min_num_consistent = 2): about half
    # of the pixels survive on this scene (0.49 for the oracle)
    assert kept.mean() > 0.3 and np.median(rel) < 5e-3


def test_bench_cpu_baseline_crop_problem(pm_oracle):
    """The exact problem bench.py hands to the oracle for `cpu_baseline` (a 512 x 384 centre crop of a
    2560 x 1920 reference image against its 20 full-resolution sources, 5 x 4 sweeps, photometric +
    filter): the HIP path on the same inputs gives the same bits -- full-resolution source images
    (packed 2563 x 1923 footprints, fp32 entry index), S = 20, M = 15."""
    from colmap_amd import mvs
    from pm_common import bench_crop_problem
    mixed, ref, src, crop, (dmin, dmax) = bench_crop_problem(device="cuda")
    o, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1)
    want = pm_oracle.run(o, oracle_inputs(mixed), ref, src, want_cost=True)
    pm = mvs.PatchMatch(h, hip_problem(mixed, ref, src))
    pm.Run()
    got = dict(depth=pm.GetDepthMap(), normal=pm.GetNormalMap(), sel_prob=pm.GetSelProbMap(),
               cost=pm.GetCostMap(), mask=pm.GetConsistencyMask())
    _assert_equal(want, got)
    kept = got["depth"] > 0
    rel = np.abs(got["depth"][kept] - crop.depth[kept]) / crop.depth[kept]
    assert kept.mean() > 0.5 and np.median(rel) < 5e-3


def test_against_committed_golden_fixture():
    """tests/golden/pm_48x36.npz (device-order oracle output, committed) reproduced by the HIP path."""
    import os
    from colmap_amd import mvs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pm_48x36.npz"))
    images = [mvs.Image(g["K"][i], g["R"][i], g["T"][i], g["gray"][i]) for i in range(len(g["gray"]))]
    dmin, dmax = [float(x) for x in g["depth_range"]]
    opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=dmin, depth_max=dmax, sigma_spatial=5.0,
                                geom_consistency=False, filter=True, num_iterations=1)
    pm = mvs.PatchMatch(opt, mvs.PatchMatch.Problem(1, [0, 2, 3], images))
    pm.Run()
    got = dict(depth=pm.GetDepthMap(), normal=pm.GetNormalMap(), sel_prob=pm.GetSelProbMap(),
               cost=pm.GetCostMap(), mask=pm.GetConsistencyMask())
    _assert_equal({k: g[f"order1_{k}"] for k in got}, got)


def test_full_size_properties():
    """BASELINE.json config[1] shape (2560x1920, S=20, photometric + filter) through
    size-independent properties: the oracle would need hours at this size."""
    import torch
    from colmap_amd import mvs
    W, H, S = 2560, 1920, 20
    views = syn.make_scene(S + 2, W, H, arc_deg=3.6 * (S + 1), device="cuda")
    images = [mvs.Image(v.K, v.R, v.T, torch.from_numpy(v.gray).cuda()) for v in views]
    outs = []
    for ref in (S // 2, S // 2 + 1):
        src = [i for i in range(ref - S // 2, ref + S // 2 + 1) if i != ref][:S]
        dmin, dmax = syn.depth_range(views, ref)
        opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=dmin, depth_max=dmax, sigma_spatial=5.0,
                                    geom_consistency=False, filter=True)
        outs.append((ref, src, dmin, dmax, mvs.PatchMatch(opt, mvs.PatchMatch.Problem(ref, src, images))))
    mvs.run_batch([o[-1] for o in outs])
    for ref, src, dmin, dmax, pm in outs:
        depth, normal, mask, sel = pm.GetDepthMap(), pm.GetNormalMap(), pm.GetConsistencyMask(), pm.GetSelProbMap()
        filtered = depth == 0
        kept = depth > 0
        # the algorithm (reference included) does not forbid a non-positive depth hypothesis from
        # winning (PropagateDepth can return one, patch_match_cuda.cu:210-236); it must be very rare
        odd = ~filtered & ~kept
        assert odd.mean() < 1e-4, odd.mean()
        assert 0.5 < kept.mean() <= 1.0
        nrm = np.linalg.norm(normal, axis=0)
        np.testing.assert_allclose(nrm[kept], 1.0, atol=1e-4)             # unit normals where kept
        assert np.all(normal[:, filtered] == 0) and np.all(mask[:, filtered] == 0)  # filtered pixels fully zeroed
        assert np.all(mask[:, kept].sum(0) >= 2)                          # filter_min_num_consistent
        assert np.all((sel >= 0) & (sel <= 1)) and np.isfinite(sel).all()
        # normals face the camera (GenerateRandomNormal / PerturbNormal keep n . ray < 0)
        ys, xs = np.nonzero(kept)
        K = views[ref].K
        ray = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, float)], 0)
        assert ((normal[:, ys, xs] * ray).sum(0) < 0).mean() > 0.999
        gt = views[ref].depth
        rel = np.abs(depth[kept] - gt[kept]) / gt[kept]
        assert np.median(rel) < 2e-3 and (rel < 0.01).mean() > 0.9        # accuracy vs ground truth
        # the consistency-graph list is consistent with the mask
        flat = pm.GetConsistentImageIdxs()
        assert len(flat) == 3 * int((~filtered).sum()) + int(mask.sum())
    # idempotence of the deterministic pipeline: solving the first problem alone gives the same bits
    ref, src, dmin, dmax, pm = outs[0]
    opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=dmin, depth_max=dmax, sigma_spatial=5.0,
                                geom_consistency=False, filter=True)
    solo = mvs.PatchMatch(opt, mvs.PatchMatch.Problem(ref, src, images))
    solo.Run()
    assert np.array_equal(solo.GetDepthMap(), pm.GetDepthMap())
    assert np.array_equal(solo.GetNormalMap(), pm.GetNormalMap())


def test_controller_two_pass_files_and_resume(pm_oracle, tmp_path):
    """PatchMatchController: photometric pass for all images, in-memory exchange, geometric pass
    with filtering (reference patch_match.cc:183-204); outputs in Mat format; resume skips work.
    The whole two-pass pipeline is bit-identical to the oracle run the same way."""
    import os
    from colmap_amd import mvs
    views = scene(3, 64, 48)
    ws = [mvs.WorkspaceImage(f"img{i}.png", v.K, v.R, v.T, v.gray, syn.depth_range(views, i)) for i, v in enumerate(views)]
    opt = mvs.PatchMatchOptions(gpu_index="0", geom_consistency=True, filter=True, num_iterations=1)
    ctl = mvs.PatchMatchController(opt, ws, str(tmp_path), batch_size=3)
    out = ctl.Run()
    # oracle, same schedule
    maps = []
    for ref in range(3):
        dmin, dmax = syn.depth_range(views, ref)
        o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=0, num_iterations=1, order=1)
        r = pm_oracle.run(o, oracle_inputs(views), ref, [i for i in range(3) if i != ref])
        maps.append((r["depth"], r["normal"]))
    for ref in range(3):
        dmin, dmax = syn.depth_range(views, ref)
        o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=1, filter=1, num_iterations=1, order=1)
        r = pm_oracle.run(o, oracle_inputs(views, True, maps), ref, [i for i in range(3) if i != ref])
        assert np.array_equal(out[ref][0], r["depth"]) and np.array_equal(out[ref][1], r["normal"])
        for kind in ("photometric", "geometric"):
            assert os.path.exists(tmp_path / "stereo" / "depth_maps" / f"img{ref}.png.{kind}.bin")
        assert np.array_equal(mvs.read_mat(str(tmp_path / "stereo" / "normal_maps" / f"img{ref}.png.geometric.bin")), r["normal"])
    # resume: everything exists -> no GPU work, same results
    ctl2 = mvs.PatchMatchController(opt, ws, str(tmp_path), batch_size=3)
    out2 = ctl2.Run()
    assert all(np.array_equal(out2[k][0], out[k][0]) for k in out)


def _two_rank_controller_worker(rank, world, port, root, q):
    import os
    import torch.distributed as dist
    from colmap_amd import mvs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views = scene(4, 64, 48)
        ws = [mvs.WorkspaceImage(f"img{i}.png", v.K, v.R, v.T, v.gray, syn.depth_range(views, i)) for i, v in enumerate(views)]
        opt = mvs.PatchMatchOptions(gpu_index="0", geom_consistency=True, filter=True, num_iterations=1)
        ctl = mvs.PatchMatchController(opt, ws, os.path.join(root, f"rank{rank}"), batch_size=2, rank=rank, world_size=world)
        out = ctl.Run()
        q.put((rank, {k: (v[0].copy(), v[1].copy()) for k, v in out.items()}, ctl.timings["map_exchange"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_geometric_pass_exchanges_maps_on_device(tmp_path):
    """Two ranks (sharing GPU 0, gloo process group): each runs the photometric pass of its own
    reference images, the maps are exchanged as packed device buffers, each runs the geometric pass
    of its problems against maps that came from the other rank. Bit-identical to the one-rank run
    (which keeps everything in HBM)."""
    import socket
    import torch.multiprocessing as mp
    from colmap_amd import mvs
    views = scene(4, 64, 48)
    ws = [mvs.WorkspaceImage(f"img{i}.png", v.K, v.R, v.T, v.gray, syn.depth_range(views, i)) for i, v in enumerate(views)]
    opt = mvs.PatchMatchOptions(gpu_index="0", geom_consistency=True, filter=True, num_iterations=1)
    solo_ctl = mvs.PatchMatchController(opt, ws, str(tmp_path / "solo"), batch_size=2)
    solo = solo_ctl.Run()
    assert solo_ctl.timings["map_exchange"]["transport"] == "local"
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_controller_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, out0, info0), (_, out1, info1) = res
    assert sorted(out0) == [0, 2] and sorted(out1) == [1, 3]
    assert info0["images"] == info1["images"] == 4 and info0["bytes"] == 4 * 2 * 4 * 64 * 48
    for k, (d, n) in {**out0, **out1}.items():
        assert np.array_equal(d, solo[k][0]) and np.array_equal(n, solo[k][1])


def test_patch_match_stereo_cli_on_a_workspace(tmp_path):
    """`python -m colmap_amd.patch_match_stereo` on an undistorted workspace on disk (exe/mvs.cc:
    228-279): sparse model -> depth ranges and `__auto__` sources, photometric + geometric pass,
    Mat / consistency-graph files, accuracy against the ground-truth depth of the renderer."""
    import os
    from colmap_amd import mvs, patch_match_stereo as cli, workspace as W
    from pm_common import write_dense_workspace
    views = scene(5, 96, 72)
    ws = str(tmp_path / "dense")
    names = write_dense_workspace(ws, views, cfg_spec="__auto__, 3")
    rc = cli.main(["--workspace_path", ws, "--PatchMatchStereo.gpu_index", "0",
                   "--PatchMatchStereo.num_iterations", "3", "--PatchMatchStereo.write_consistency_graph", "1"])
    assert rc == 0
    w = W.Workspace(ws)
    ranges = w.GetModel().ComputeDepthRanges()
    for i, name in enumerate(names):
        for kind in ("photometric", "geometric"):
            assert os.path.exists(w.GetDepthMapPath(i, kind)) and os.path.exists(w.GetNormalMapPath(i, kind))
        depth = mvs.read_mat(w.GetDepthMapPath(i, "geometric"))
        normal = mvs.read_mat(w.GetNormalMapPath(i, "geometric"))
        assert depth.shape == (72, 96) and normal.shape == (3, 72, 96)
        kept = depth > 0
        assert kept.mean() > 0.3
        assert depth[kept].min() >= ranges[i][0] * 0.999 and depth[kept].max() <= ranges[i][1] * 1.001
        rel = np.abs(depth[kept] - views[i].depth[kept]) / views[i].depth[kept]
        assert np.median(rel) < 0.01, (i, np.median(rel))
        n = np.linalg.norm(normal[:, kept], axis=0)
        assert np.allclose(n, 1.0, atol=1e-4)
        gw, gh, graph = W.read_consistency_graph(w.GetConsistencyGraphPath(i, "geometric"))
        assert (gw, gh) == (96, 72) and len(graph) == int(kept.sum())
        # every filtered-in pixel lists >= filter_min_num_consistent source images of this problem
        assert min(len(v) for v in graph.values()) >= 2
    # the consumer of the maps: stereo_fusion on the same workspace (exe/mvs.cc:299-386)
    from colmap_amd import fusion
    from colmap_amd.__main__ import main as colmap_amd_main
    with open(os.path.join(ws, "stereo", "fusion.cfg"), "w") as f:
        f.write("\n".join(names) + "\n")
    ply = os.path.join(ws, "fused.ply")
    assert colmap_amd_main(["stereo_fusion", "--workspace_path", ws, "--output_path", ply,
                            "--StereoFusion.min_num_pixels", "3"]) == 0
    pts = fusion.read_binary_ply_points(ply)
    vis = fusion.read_points_visibility(ply + ".vis", len(pts.xyz))
    assert len(pts.xyz) > 100
    errs = []
    for p, v in zip(pts.xyz[::5], vis[::5]):        # fused points sit on the rendered surfaces
        i = int(v[0])
        pc = np.asarray(views[i].R, np.float64) @ p + np.asarray(views[i].T, np.float64)
        px = np.asarray(views[i].K, np.float64) @ (pc / pc[2])
        c, r = int(round(px[0])), int(round(px[1]))
        if 0 <= c < 96 and 0 <= r < 72:
            errs.append(abs(views[i].depth[r, c] - pc[2]) / pc[2])
    assert len(errs) > 10 and np.median(errs) < 0.02
    # second invocation: everything exists -> nothing recomputed, files untouched
    before = {p: os.path.getmtime(p) for p in [w.GetDepthMapPath(i, "geometric") for i in range(len(names))]}
    assert cli.main(["--workspace_path", ws, "--PatchMatchStereo.gpu_index", "0",
                     "--PatchMatchStereo.num_iterations", "3", "--PatchMatchStereo.write_consistency_graph", "1"]) == 0
    assert all(os.path.getmtime(p) == t for p, t in before.items())
