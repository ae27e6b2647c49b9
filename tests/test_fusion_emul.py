"""colmap_amd/csrc/fusion.hip -- the UNMODIFIED product source, host loop and kernels -- executed on the CPU and
compared bit for bit with the checker (oracle/fusion_oracle.cpp mode 1 = the reference's Fuse() taken sequentially
in the order of its own pool schedule, mvs/fusion.cc:253-269, 293-337).

tests/hip_emul/ is a HIP stand-in for exactly this purpose (fibers for the lanes of a workgroup, the cross-lane
primitives as barriers, atomics on one OS thread): test infrastructure, never loaded by the product, whose
library is built by hipcc and has no CPU path. What these tests pin without a GPU: the wave-cooperative walk (one
neighbour per lane, stack in LDS + spill), the tentative marks / rank cut / committed prefix of the passes, the
wave medians (counting and radix select), the (thread, tick) compaction and the per-thread concatenation. The GPU
tests of test_fusion.py run the same inputs through the hipcc build."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import fusion_oracle
from colmap_amd import fusion
from pm_common import scene
from test_fusion import _CASES, _case, _images, _overlap, _same

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_emul")
_SRC = os.path.join(os.path.dirname(_HERE), "..", "colmap_amd", "csrc", "fusion.hip")


class _EmulEntryPoints:
    """The fusion_* entry points of a CPU build of fusion.hip (same struct marshalling as the product binding)."""
    _built = False

    def __init__(self, name):
        path = os.path.join(_HERE, name)
        deps = [_SRC, os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_HERE, "hipcub", "hipcub.hpp"),
                os.path.join(_HERE, "build.sh")]
        if not _EmulEntryPoints._built and (not os.path.exists(path) or
                                            any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps)):
            subprocess.check_call(["sh", os.path.join(_HERE, "build.sh")])
        _EmulEntryPoints._built = True
        L = C.CDLL(path)
        L.fusion_last_error.restype = C.c_char_p
        L.fusion_num_points.restype = C.c_size_t
        self.run, self.num_points, self.get_points = L.fusion_run, L.fusion_num_points, L.fusion_get_points
        self.get_visibility, self.free, self.last_error = L.fusion_get_visibility, L.fusion_free, L.fusion_last_error
        self.lib = L

    def passes(self):
        st = [C.c_int64() for _ in range(4)]
        self.lib.fusion_last_stats(*[C.byref(x) for x in st])
        return st[2].value


@pytest.fixture(scope="module")
def emul():
    return _EmulEntryPoints("libfusion_emul.so")


@pytest.fixture(scope="module")
def emul_small():
    return _EmulEntryPoints("libfusion_emul_small.so")


def _noisy(n, w, h, sigma, seed=1):
    """Rendered maps with multiplicative depth noise: walks then leave the stripe of their start pixel, and on a
    narrow image the stripes above and below take their turns in the same pass -- marks meet, passes are cut."""
    images = _images(scene(n, w, h))
    rng = np.random.default_rng(seed)
    for im in images:
        im.depth_map = (im.depth_map * (1 + sigma * rng.standard_normal(im.depth_map.shape))).astype(np.float32)
    return images


_LOOSE = dict(min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05, max_normal_error=30.0)


# (the three slowest of the twelve -- depth1, loose_7x80x60, sparse_overlap: 14 s together -- run with COLMAP_AMD_TEST_SLOW=1;
#  sparse_overlap also runs in the table-tier test below)
_FAST_CASES = [n for n in sorted(_CASES) if os.environ.get("COLMAP_AMD_TEST_SLOW", "0") != "0" or
               n not in ("depth1", "loose_7x80x60", "sparse_overlap")]


@pytest.mark.parametrize("name", _FAST_CASES)
def test_emulated_kernel_equals_oracle_on_the_gpu_test_inputs(emul, name):
    opt, images, overlap = _case(name)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    got = fusion.fuse(opt, images, overlap, entry_points=emul)
    assert len(want.xyz) > 20 and _same(got, want), (len(got.xyz), len(want.xyz))


def _schedule():
    L = fusion_oracle.lib()
    d, c, cu, mu = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_double()
    L.fuo_last_schedule(C.byref(d), C.byref(c), C.byref(cu), C.byref(mu))
    return d.value, c.value, cu.value


def test_emulated_kernel_cut_passes(emul):
    """Marks of different pool threads meet (16 stripes of a 24-pixel-wide image in one window): the pass is cut at
    the later turn, the prefix commits, the rest is walked again -- and the result is still the sequential one. The
    simulation of the schedule in the checker (mode 2) sees the same situation."""
    images, overlap = _noisy(4, 24, 160, 0.01), _overlap(4)
    opt = fusion.StereoFusionOptions(**_LOOSE)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    assert _same(fusion_oracle.fuse(opt, images, overlap, mode=2), want)
    discarded, conflicts, cuts = _schedule()
    assert conflicts > 100 and cuts >= 2 and discarded > 10, (discarded, conflicts, cuts)
    got = fusion.fuse(opt, images, overlap, entry_points=emul)
    assert len(want.xyz) > 2000 and _same(got, want)
    seq = fusion_oracle.fuse(opt, images, overlap, mode=0)  # row-major: another order, about the same cloud
    assert abs(len(seq.xyz) - len(want.xyz)) < 0.03 * len(seq.xyz)


def test_emulated_kernel_blocks_in_reverse_order():
    """The waves of a pass run concurrently on the GPU; the stand-in runs them one after the other. Forward and
    reverse are the two extremes of the order in which they can reach a shared word: same result (a separate
    process, the order is read once per library load)."""
    code = ("import sys; sys.path[:0] = [%r, %r, %r]\n"
            "import test_fusion_emul as T, fusion_oracle\n"
            "from colmap_amd import fusion\n"
            "E = T._EmulEntryPoints('libfusion_emul.so')\n"
            "im, ov = T._noisy(4, 24, 160, 0.01), T._overlap(4)\n"
            "opt = fusion.StereoFusionOptions(**T._LOOSE)\n"
            "assert T._same(fusion.fuse(opt, im, ov, entry_points=E), fusion_oracle.fuse(opt, im, ov, mode=1))\n"
            "print('reverse ok', E.passes())\n") % (os.path.dirname(_HERE), os.path.join(os.path.dirname(_HERE), "..", "oracle"),
                                                     os.path.join(os.path.dirname(_HERE), ".."))
    env = dict(os.environ, HIP_EMUL_BLOCK_ORDER="reverse")
    out = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "reverse ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("shape", [(5, 64, 48, 0.003), (4, 24, 160, 0.01)] if os.environ.get("COLMAP_AMD_TEST_SLOW", "0") != "0"
                         else [(4, 24, 160, 0.01)])
@pytest.mark.parametrize("wide", [0, 1])
def test_emulated_kernel_overflow_paths(emul, emul_small, shape, wide):
    """The same source with a record buffer of 1 024 pixels per wave, 8 stack entries in LDS, a first stack spill of 8
    entries and medians staged up to 4 values: a wave whose buffer is full cuts the pass at its own turn, a walk
    whose stack overflows the spill cuts the pass and the host grows the spill, larger supports take the
    radix-select median. More passes than the product's capacities need; bit for bit the same points."""
    from switches import switches
    n, w, h, sigma = shape
    images, overlap = _noisy(n, w, h, sigma), _overlap(n)
    opt = fusion.StereoFusionOptions(max_num_pixels=1000, **_LOOSE)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    # wide = 0: depth-first walks only (the stack of a walk is deepest); 1: the default, breadth-first where that is exact
    with switches(emul_small.lib, COLMAP_AMD_FUSION_WIDE=wide), switches(emul.lib, COLMAP_AMD_FUSION_WIDE=wide):
        got = fusion.fuse(opt, images, overlap, entry_points=emul_small)
        small_passes = emul_small.passes()
        assert len(want.xyz) > 1500 and _same(got, want)
        assert max(len(v) for v in got.visibility) >= 4
        assert _same(fusion.fuse(opt, images, overlap, entry_points=emul), want)
        if wide == 0:
            assert small_passes > emul.passes(), (small_passes, emul.passes())


def test_wide_walks_fall_back_to_depth_first_at_the_traversal_limits(emul):
    """Breadth-first walks (walk_turn_wide) are exact only while no limit of the traversal can bind; a walk that would
    absorb more than min(max_traversal_depth - 1, max_num_pixels - 1, record capacity) pixels takes its marks back and
    is repeated depth-first. With max_traversal_depth = 20 on noisy maps with loose thresholds many walks pass 19
    pixels: same cloud as the sequential definition, with and without breadth-first walks."""
    from switches import switches
    images, overlap = _noisy(4, 24, 160, 0.01), _overlap(4)
    opt = fusion.StereoFusionOptions(max_traversal_depth=20, **_LOOSE)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    emul.lib.fusion_last_redone_walks.restype = C.c_int64
    with switches(emul.lib, COLMAP_AMD_FUSION_WIDE=0):
        assert _same(fusion.fuse(opt, images, overlap, entry_points=emul), want)
        assert emul.lib.fusion_last_redone_walks() == 0
    assert _same(fusion.fuse(opt, images, overlap, entry_points=emul), want)
    assert emul.lib.fusion_last_redone_walks() > 20, emul.lib.fusion_last_redone_walks()
    assert len(want.xyz) > 1500


def test_small_build_refuses_a_record_capacity_it_cannot_hold(emul_small):
    views = scene(4, 32, 24)
    with pytest.raises(RuntimeError, match="record buffer"):
        fusion.fuse(fusion.StereoFusionOptions(), _images(views), _overlap(4), entry_points=emul_small)


@pytest.mark.parametrize("num_threads", [1, 2, 3])
def test_emulated_kernel_pool_sizes(emul, num_threads):
    """StereoFusionOptions::num_threads = the size of the reference's pool (T waves take stripes t, t + T, ...).
    One thread is the reference's own sequential run: the checker's row-major mode 0, bit for bit. With the
    non-default caps (mode 0 honours max_num_pixels beyond the record capacity, modes 1 / 2 document the clamp)
    out of play the three agree."""
    images, overlap = _noisy(4, 24, 160, 0.01), _overlap(4)
    opt = fusion.StereoFusionOptions(num_threads=num_threads, **_LOOSE)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    got = fusion.fuse(opt, images, overlap, entry_points=emul)
    assert len(want.xyz) > 2000 and _same(got, want)
    if num_threads == 1:
        assert _same(got, fusion_oracle.fuse(opt, images, overlap, mode=0))


@pytest.mark.parametrize("tables", ["0", "1"])
def test_emulated_kernel_table_tiers(tables):
    """The walk kernel reads image descriptors and overlap lists from an LDS copy when they fit (tier 2, what every
    other test here runs), descriptors only (1), or from HBM (0): the development switch COLMAP_AMD_FUSION_LDS_TABLES caps the tier. Same
    points. (A separate process per tier.)"""
    code = ("import sys; sys.path[:0] = [%r, %r, %r]\n"
            "import test_fusion_emul as T, fusion_oracle\n"
            "from colmap_amd import fusion\n"
            "E = T._EmulEntryPoints('libfusion_emul.so')\n"
            "E.lib.colmap_amd_set_switch(b'COLMAP_AMD_FUSION_LDS_TABLES', b'%s')\n"
            "for name in ('defaults_5x64x48', 'sparse_overlap', 'mask0'):\n"
            "    opt, im, ov = T._case(name)\n"
            "    assert T._same(fusion.fuse(opt, im, ov, entry_points=E), fusion_oracle.fuse(opt, im, ov, mode=1)), name\n"
            "print('tiers ok')\n") % (os.path.dirname(_HERE), os.path.join(os.path.dirname(_HERE), "..", "oracle"),
                                       os.path.join(os.path.dirname(_HERE), ".."), tables)
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "tiers ok" in out.stdout, out.stderr[-2000:]


def test_emulated_kernel_more_images_than_the_lds_table_holds(emul):
    """130 images: the descriptors alone exceed the table (tier 0 chosen by the host), every image overlapping its
    four ring neighbours."""
    n = 130
    views = scene(n, 12, 10)
    images = _images(views)
    overlap = [[(i + d) % n for d in (-2, -1, 1, 2)] for i in range(n)]
    opt = fusion.StereoFusionOptions(min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05, max_normal_error=30.0)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    got = fusion.fuse(opt, images, overlap, entry_points=emul)
    assert len(want.xyz) > 100 and _same(got, want)


def test_emulated_kernel_image_listed_as_its_own_neighbour(emul):
    """Degenerate overlap lists (an image among its own neighbours, a neighbour listed twice): the pixel being expanded
    is masked by its own mark -- which the kernel issues without waiting for it -- and is not walked into again."""
    views = scene(4, 32, 24)
    images = _images(views)
    overlap = [[i] + [j for j in range(4) if j != i] + [(i + 1) % 4] for i in range(4)]
    opt = fusion.StereoFusionOptions(min_num_pixels=2)
    want = fusion_oracle.fuse(opt, images, overlap, mode=1)
    got = fusion.fuse(opt, images, overlap, entry_points=emul)
    assert len(want.xyz) > 100 and _same(got, want)
    assert _same(want, fusion_oracle.fuse(opt, images, [[j for j in range(4) if j != i] for i in range(4)], mode=1))


def _random_configuration(rng):
    n = int(rng.integers(2, 8))
    w, h = [(24, 18), (32, 24), (20, 50), (48, 36), (16, 64)][int(rng.integers(5))]
    sigma = float(rng.choice([0.0, 0.003, 0.01, 0.03]))
    images = _images(scene(n, w, h), with_rgb=bool(rng.integers(2)))
    for im in images:
        if sigma > 0:
            im.depth_map = (im.depth_map * (1 + sigma * rng.standard_normal(im.depth_map.shape))).astype(np.float32)
        if rng.random() < 0.3:
            im.depth_map = im.depth_map.copy()
            im.depth_map[rng.random(im.depth_map.shape) < 0.1] = 0
        if rng.random() < 0.2:
            im.mask = (rng.random(im.depth_map.shape) < 0.15).astype(np.uint8)
    style = int(rng.integers(4))
    if style == 0:
        overlap = [[j for j in range(n) if j != i] for i in range(n)]
    elif style == 1:
        overlap = [[(i + 1) % n] for i in range(n)]
    elif style == 2:   # short random lists: duplicates, the image itself
        overlap = [[int(x) for x in rng.integers(0, n, size=int(rng.integers(1, 9)))] for _ in range(n)]
    else:              # more than 64 neighbours: an expansion takes two 64-lane chunks
        overlap = [[int(x) for x in rng.integers(0, n, size=70)] for _ in range(n)]
    okw = dict(min_num_pixels=int(rng.integers(1, 6)), max_num_pixels=int(rng.choice([2, 5, 50, 1000])),
               max_traversal_depth=int(rng.choice([1, 2, 3, 100])), max_reproj_error=float(rng.choice([1.0, 2.0, 4.0])),
               max_depth_error=float(rng.choice([0.01, 0.05])), max_normal_error=float(rng.choice([10.0, 30.0])),
               num_threads=int(rng.choice([-1, 1, 2, 3])))
    okw["min_num_pixels"] = min(okw["min_num_pixels"], okw["max_num_pixels"])
    if rng.random() < 0.2:
        okw["bounding_box"] = ((-0.5, -0.5, -10.0), (0.5, 0.5, 10.0))
    return fusion.StereoFusionOptions(**okw), images, overlap, (n, w, h, sigma, style, okw)


@pytest.mark.parametrize("which,seed,count", [("emul", 11, 10), ("emul_small", 12, 6)])
def test_emulated_kernel_random_configurations(request, which, seed, count):
    """Seeded random scenes, options, pool sizes, masks, holes and overlap lists (all-to-all, chains, short random lists with
    duplicates and the image itself, 70-entry lists = two chunks per expansion) through both capacity builds, bit for bit
    against the sequential definition. (50 further configurations were run when the walk kernel was rewritten.)"""
    E = request.getfixturevalue(which)
    rng = np.random.default_rng(seed)
    for _ in range(count):
        opt, images, overlap, what = _random_configuration(rng)
        want = fusion_oracle.fuse(opt, images, overlap, mode=1)
        assert _same(fusion.fuse(opt, images, overlap, entry_points=E), want), what


def test_visibility_lists_behave_like_the_lists_they_stand_for(tmp_path):
    """fusion.VisibilityLists (what fuse() returns instead of millions of small arrays): len / indexing / slicing /
    iteration, and the .vis file written from it byte for byte the file written from plain lists
    (WritePointsVisibility, fusion.cc:526-541), read back to the same lists; a truncated file is refused."""
    import numpy as np
    from colmap_amd import fusion as F
    rng = np.random.default_rng(3)
    lists = [sorted(rng.choice(9, size=int(rng.integers(0, 6)), replace=False).tolist()) for _ in range(200)]
    ptr = np.concatenate([[0], np.cumsum([len(v) for v in lists])])
    idx = np.array([i for v in lists for i in v], np.int32)
    vis = F.VisibilityLists(ptr, idx)
    assert len(vis) == 200 and [list(v) for v in vis] == lists
    assert list(vis[7]) == lists[7] and list(vis[-1]) == lists[-1] and [list(v) for v in vis[10:13]] == lists[10:13]
    with pytest.raises(IndexError):
        vis[200]
    a, b = str(tmp_path / "a.vis"), str(tmp_path / "b.vis")
    F.write_points_visibility(a, vis)
    F.write_points_visibility(b, lists)
    assert open(a, "rb").read() == open(b, "rb").read()
    back = F.read_points_visibility(a, 200)
    assert [list(v) for v in back] == lists
    with pytest.raises(ValueError):
        F.read_points_visibility(a, 199)
    raw = open(a, "rb").read()
    open(a, "wb").write(raw[:-4])
    with pytest.raises(ValueError):
        F.read_points_visibility(a, 200)
