"""CPU-only: the C-ABI library is built, loads, and exports every symbol the public headers
declare (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:pm|ba|fusion)_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from colmap_amd import build
    build.build()
    return ctypes.CDLL(build.LIB_PATH)


@pytest.mark.parametrize("header", [h for h in ("colmap_amd_pm.h", "colmap_amd_ba.h", "colmap_amd_fusion.h")
                                    if os.path.exists(os.path.join(ROOT, "include", h))])
def test_exports_every_declared_symbol(lib, header):
    names = _declared(header)
    assert len(names) >= 4
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_options_defaults_match_reference(lib):
    """pm_options_init == PatchMatchOptions member initialisers (patch_match_options.h:37-126)."""
    from colmap_amd import mvs
    o = mvs.pm_options()
    lib.pm_options_init(ctypes.byref(o))
    d = mvs.PatchMatchOptions()
    for name in ("sigma_color", "ncc_sigma", "min_triangulation_angle", "incident_angle_sigma",
                 "geom_consistency_regularizer", "geom_consistency_max_cost", "filter_min_ncc",
                 "filter_min_triangulation_angle", "filter_geom_consistency_max_cost", "window_radius",
                 "window_step", "num_samples", "num_iterations", "filter_min_num_consistent"):
        assert getattr(o, name) == getattr(d, name), name
    assert o.geom_consistency == 1 and o.filter == 1 and o.depth_min == -1 and o.sigma_spatial == -1


def test_no_gpu_fails_loudly(lib):
    """Without a HIP device pm_create must fail with an error, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from colmap_amd import mvs
    from pm_common import scene, hip_problem
    views = scene(3, 64, 48)
    opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=1.0, depth_max=5.0, sigma_spatial=5.0,
                                geom_consistency=False)
    with pytest.raises(mvs.PatchMatchError, match="no HIP device|HIP error"):
        mvs.PatchMatch(opt, hip_problem(views, 1, [0, 2])).Run()


def test_mat_file_roundtrip(tmp_path):
    """Mat<float> file format (reference mvs/mat.cc:41-65): ASCII W&H&D& + little-endian f32."""
    import numpy as np
    from colmap_amd import mvs
    a = np.arange(3 * 4 * 5, dtype=np.float32).reshape(3, 4, 5)
    p = tmp_path / "n.bin"
    mvs.write_mat(str(p), a)
    raw = p.read_bytes()
    assert raw.startswith(b"5&4&3&") and len(raw) == 6 + a.nbytes
    assert np.array_equal(mvs.read_mat(str(p)), a)
    mvs.write_mat(str(p), a[0])
    assert np.array_equal(mvs.read_mat(str(p)), a[0])


def test_sweep_kernels_compile_to_the_intended_structure(tmp_path):
    """Compile the kernels to ISA (CPU only) and check what the design relies on: two sweep-kernel families only
    (generic, four-wave + its profiling build); the 11 x 11 kernels gather through the swizzled buffer resource
    (buffer_load_dword ... idxen offen: the packed-image index is formed by the address unit, pm_kernels.hip:
    ncc_front) with no global gather left in them, hold 4 waves per SIMD (<= 128 VGPRs) and never spill."""
    import os, re, subprocess
    from colmap_amd import build as B
    src = os.path.join(B.CSRC, "pm_kernels.hip")
    out = str(tmp_path / "pm.s")
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags +
                          ["-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    text = open(out).read()
    sweep = set(re.findall(r"^_ZN10colmap_amd\d+(pm_sweep_(?:[a-z0-9]+_)*kernel)I\w+:", text, re.M))
    assert sweep == {"pm_sweep_kernel", "pm_sweep_quad_kernel", "pm_sweep_quad_prof_kernel", "pm_sweep_pair_kernel"}, sweep
    kernels = re.findall(r"^(_ZN10colmap_amd\d+pm_sweep_quad_kernel\w+):.*?\.end_amdhsa_kernel", text, re.S | re.M)
    assert len(kernels) == 8   # 4 (geom, filter) variants x 2 addressing modes
    # two waves per column group (a lone large problem): buffer resource only, the helper wave's loop is a third task pass
    pairs = re.findall(r"^(_ZN10colmap_amd\d+pm_sweep_pair_kernel\w+):.*?\.end_amdhsa_kernel", text, re.S | re.M)
    assert len(pairs) == 4
    for name in kernels + pairs:
        body = text[text.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        lines = [l.split(";")[0].strip() for l in body.splitlines()]
        lines = [l for l in lines if l and not l.startswith(".")]
        gathers = [l for l in lines if l.startswith("buffer_load_dword")]
        if "pm_sweep_pair_kernel" in name:
            assert len(gathers) == 3 * 8 * 2 and all("idxen offen" in l for l in gathers), len(gathers)
        elif name.endswith("Lb1EEEvPKNS_8PmParamsE"):   # MUBUF = true
            # P4 and P6 loops x 8 gathers x (clamping + unclamped addressing)
            assert len(gathers) == 2 * 8 * 2 and all("idxen offen" in l for l in gathers), len(gathers)
        else:
            assert not gathers
        meta = text[text.index(".name:           " + name):]
        meta = meta[:meta.index(".wavefront_size")] if ".wavefront_size" in meta else meta[:2000]
        # photometric builds: 5 workgroups = 20 waves per CU (<= 96 VGPRs; a few loop-invariant registers may live in
        # scratch -- re-read once per NCC batch, never inside the tap rounds); geometric builds: 4 (<= 128, no scratch)
        geom = "pm_sweep_quad_kernelILb1E" in name or "pm_sweep_pair_kernelILb1E" in name
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= (128 if geom else 96)
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) <= (0 if geom else 64)
        scratch = [i for i, l in enumerate(lines) if l.startswith("scratch_")]
        loop_gathers = [i for i, l in enumerate(lines) if l.startswith("buffer_load_dword") and "idxen" in l]
        if scratch and loop_gathers:
            # no scratch access inside the tap rounds: between the first and last gather of either task pass
            first, last = loop_gathers[0], loop_gathers[15]
            assert not [i for i in scratch if first < i < last]
