"""The C++ host classes (include/colmap_amd/*.hpp: PatchMatch, StereoFusion, Mi355xBundleAdjuster, the pose-prior
adjuster) driven THROUGH the C ABI into the product's kernels on the CPU stand-in: tests/cpp/test_mvs_host.cc and
test_ba_host.cc linked against tests/hip_emul/lib{pm,fusion,ba}_emul.so instead of libcolmap_amd.so, and the GPU tests of
tests/test_cpp_host.py run with those binaries and with the Python mirror's library swapped the same way (both sides sit
on the same C ABI: bit-identical outputs). TEST INFRASTRUCTURE ONLY -- see tests/hip_emul/README.md."""
import os
import subprocess

import numpy as np
import pytest

import test_ba_emul
import test_cpp_host as H
import test_fusion_emul
import test_pm_emul
from colmap_amd import estimators as est, mvs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "hip_emul")


def _compile(name, libs, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp_emul") / name)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", name + ".cc"), "-L", EMUL] + ["-l" + l for l in libs] + \
          ["-Wl,-rpath," + EMUL, "-o", out]
    subprocess.check_call(cmd)
    return out


@pytest.fixture(scope="module")
def mvs_host_emul(tmp_path_factory):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and "HIP_EMUL_CXX" not in os.environ:
        pytest.skip("the stand-in is built with ROCm's clang++ as host compiler")
    test_pm_emul._emul_lib()                                  # (builds the libraries when a source is newer)
    test_fusion_emul._EmulEntryPoints("libfusion_emul.so")
    return _compile("test_mvs_host", ["pm_emul", "fusion_emul"], tmp_path_factory)


@pytest.fixture(scope="module")
def ba_host_emul(tmp_path_factory):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and "HIP_EMUL_CXX" not in os.environ:
        pytest.skip("the stand-in is built with ROCm's clang++ as host compiler")
    test_ba_emul._emul_lib()
    return _compile("test_ba_host", ["ba_emul"], tmp_path_factory)


def test_cpp_stereo_fusion(mvs_host_emul):
    """colmap_amd::mvs::StereoFusion over fusion_run: two fronto-parallel views of a plane; num_threads = 1 gives row-major
    turns (points by ascending row), num_threads = 2 the pool's stripe order."""
    H.test_cpp_stereo_fusion(mvs_host_emul)


def test_cpp_patch_match_equals_python_mirror(mvs_host_emul, tmp_path, monkeypatch):
    """colmap_amd::mvs::PatchMatch (C++) against colmap_amd.mvs.PatchMatch (Python), photometric + filter, one iteration:
    depth / normal / selection-probability maps and the consistency graph file, bit for bit."""
    from colmap_amd import synthetic as syn, workspace as W
    from pm_common import scene, hip_problem
    monkeypatch.setattr(mvs, "lib", test_pm_emul._emul_lib)
    views = scene(3, 40, 30)
    ref, src = 1, [0, 2]
    dmin, dmax = syn.depth_range(views, ref)
    d = str(tmp_path / "photo")
    os.makedirs(d)
    H._write_problem(d, views, ref, src, False, True, 1, dmin, dmax, None)
    r = subprocess.run([mvs_host_emul, "run", d], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = mvs.PatchMatchOptions(gpu_index="0", depth_min=dmin, depth_max=dmax, sigma_spatial=5.0, geom_consistency=False,
                              filter=True, num_iterations=1)
    pm = mvs.PatchMatch(o, hip_problem(views, ref, src, None))
    pm.Run()
    assert np.array_equal(mvs.read_mat(os.path.join(d, "depth.bin")), pm.GetDepthMap())
    assert np.array_equal(mvs.read_mat(os.path.join(d, "normal.bin")), pm.GetNormalMap())
    assert np.array_equal(mvs.read_mat(os.path.join(d, "sel_prob.bin")), pm.GetSelProbMap())
    gw, gh, graph = W.read_consistency_graph(os.path.join(d, "graph.bin"))
    assert (gw, gh) == (40, 30) and (pm.GetDepthMap() > 0).sum() == len(graph)


def test_cpp_bundle_adjusters_equal_python_mirror(ba_host_emul, tmp_path, monkeypatch):
    """Mi355xBundleAdjuster and the pose-prior adjuster in C++ against the Python mirror: the GPU tests of
    tests/test_cpp_host.py, both sides on the stand-in."""
    monkeypatch.setattr(est, "lib", test_ba_emul._emul_lib)
    H.test_cpp_bundle_adjuster_equals_python_mirror(ba_host_emul, tmp_path)
    H.test_cpp_pose_prior_adjuster_equals_python_mirror(ba_host_emul, tmp_path)
