"""colmap_amd/csrc/pm_kernels.hip + pm_api.cpp -- the UNMODIFIED product sources: the sweep kernels that carry the
headline metric, ComputeInitialCost, the filters, the host schedule -- executed on the CPU and compared BIT FOR BIT with
the checker (oracle/pm_oracle.c in device order), by the comparison functions of tests/test_pm_gpu.py with the library
swapped for a CPU build of the same files.

tests/hip_emul/ is a HIP stand-in for exactly this purpose (lanes as fibers; cross-lane primitives, DPP rows and
__syncthreads as barriers over the lanes still running; LDS as thread storage; buffer_load through a swizzled resource
by the address formula the kernel states). The five inline-assembly helpers of the kernels live in one header of the
product, colmap_amd/csrc/gfx950/pm_gfx950_asm.h, which the stand-in shadows with C++ restatements
(tests/hip_emul/pm/gfx950/pm_gfx950_asm.h) -- everything else is the product's source as hipcc compiles it. Test
infrastructure, never loaded by the product (its library is built by hipcc and has no CPU path). Sizes are small: a
lane is a fiber here, one cross-lane operation costs a microsecond per lane. The GPU tests run the same comparisons
through the hipcc build at the BASELINE shapes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import test_pm_gpu as G
from colmap_amd import mvs, synthetic as syn
from pm_common import hip_problem, oracle_inputs, paired_options, scene

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_emul")
_CSRC = os.path.join(os.path.dirname(_HERE), "..", "colmap_amd", "csrc")
_LIB = None


def _emul_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpm_emul.so")
        deps = [os.path.join(_CSRC, f) for f in ("pm_kernels.hip", "pm_api.cpp", "pm_internal.h")]
        deps += [os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_HERE, "pm", "gfx950", "pm_gfx950_asm.h"),
                 os.path.join(_HERE, "pm", "pm_stubs.cpp"), os.path.join(_HERE, "build_pm.sh"),
                 os.path.join(os.path.dirname(_HERE), "..", "include", "colmap_amd_pm.h")]
        if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            subprocess.check_call(["sh", os.path.join(_HERE, "build_pm.sh")])
        _LIB = C.CDLL(path)
        _LIB.pm_last_error.restype = C.c_char_p
        _LIB.pm_device_count.restype = C.c_int
    return _LIB


@pytest.fixture(autouse=True)
def emulated_library(monkeypatch):
    """mvs.PatchMatch -- what the GPU tests drive -- reaches the pm_* entry points of the CPU build."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and "HIP_EMUL_CXX" not in os.environ:
        pytest.skip("the stand-in is built with ROCm's clang++ as host compiler")
    monkeypatch.setattr(mvs, "lib", _emul_lib)


def test_initial_state_cost_pose_tables_and_reference_filter(pm_oracle):
    views = scene(4, 48, 36)
    want, got, pm = G._run_both(pm_oracle, views, 1, [0, 2, 3], geom_consistency=0, filter=0, max_sweeps=0)
    G._assert_equal(want, got, ("depth", "normal", "cost"))
    poses, K, iK = pm.GetPoseTables()
    o_poses, o_K, o_iK = pm_oracle.pose_tables(oracle_inputs(views), 1, [0, 2, 3])
    assert np.array_equal(K, o_K) and np.array_equal(iK, o_iK) and np.array_equal(poses, o_poses)
    img, s, ss = pm.GetRefFilter()
    o_img, o_s, o_ss = pm_oracle.filter_ref_image(views[1].gray, 5, 1, 5.0, float(np.float32(0.2)))
    assert np.array_equal(img, o_img) and np.array_equal(s, o_s) and np.array_equal(ss, o_ss)


@pytest.mark.parametrize("nsweeps", [1, 2, 3, 4])
def test_each_sweep_direction(pm_oracle, nsweeps):
    """Every direction of the virtual rotation (odd directions deal the taps column-major) against the oracle's
    physical rotation; the kernel is the shipped default, gathers through the swizzled buffer resource."""
    views = scene(4, 40, 30)
    want, got, pm = G._run_both(pm_oracle, views, 1, [0, 2, 3], geom_consistency=0, filter=0, max_sweeps=nsweeps)
    G._assert_equal(want, got, ("depth", "normal", "cost", "sel_prob"))
    assert pm.GetSweepKernelName() == "pm_sweep_quad_kernel"


def test_full_photometric_solve_with_filter(pm_oracle):
    """The 5 x 4 sweep schedule + photometric filter + the consistency-graph list, every output map."""
    views = scene(4, 48, 36)
    want, got, pm = G._run_both(pm_oracle, views, 1, [0, 2, 3], geom_consistency=0, filter=1)
    G._assert_equal(want, got)
    ms, n = pm.GetSweepTiming()
    assert n == 20
    assert (got["depth"] > 0).mean() > 0.3


def test_geometric_consistency_pass_and_both_filters(pm_oracle):
    views = scene(3, 48, 36)
    maps = []
    for ref in range(3):
        dmin, dmax = syn.depth_range(views, ref)
        o = pm_oracle.default_options(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=0, num_iterations=1,
                                      order=1)
        r = pm_oracle.run(o, oracle_inputs(views), ref, [i for i in range(3) if i != ref])
        maps.append((r["depth"], r["normal"]))
    want, got, _ = G._run_both(pm_oracle, views, 1, [0, 2], maps=maps, geom_consistency=1, filter=1, num_iterations=1)
    G._assert_equal(want, got)


@pytest.mark.parametrize("fp_global", ["0", "1"])
def test_buffer_resource_and_explicit_indices(pm_oracle, request, fp_global):
    """The four-wave kernel through the buffer resource, and with explicit strip indices (what problems whose images lie
    more than 4 GB apart get); ragged width, S = 6."""
    from switches import set_switch
    set_switch(mvs.lib(), "COLMAP_AMD_PM_FP_GLOBAL", fp_global)
    request.addfinalizer(lambda: set_switch(_emul_lib(), "COLMAP_AMD_PM_FP_GLOBAL", None))
    views = scene(7, 35, 27)
    want, got, pm = G._run_both(pm_oracle, views, 3, [0, 1, 2, 4, 5, 6], geom_consistency=0, filter=1, num_iterations=1)
    G._assert_equal(want, got)
    assert pm.GetSweepKernelName() == "pm_sweep_quad_kernel" + (" (explicit indices)" if fp_global == "1" else "")


@pytest.mark.parametrize("geom", [0, 1])
def test_two_waves_per_column_pair_kernel(pm_oracle, request, geom):
    """pm_sweep_pair_kernel (a helper wave shares pass B of the NCC phases: what a lone large problem runs) against the
    oracle: ragged width, S = 6 (batches of 24 and 6 tasks: both waves get rounds, odd and even counts), photometric
    with filter and the geometric pass with both filters."""
    from switches import set_switch
    set_switch(mvs.lib(), "COLMAP_AMD_PM_HELP", "2")
    request.addfinalizer(lambda: set_switch(_emul_lib(), "COLMAP_AMD_PM_HELP", None))
    views = scene(7, 35, 27)
    maps = [(v.depth.copy(), v.normal.copy()) for v in views] if geom else None
    want, got, pm = G._run_both(pm_oracle, views, 3, [0, 1, 2, 4, 5, 6], maps=maps, geom_consistency=geom, filter=1,
                                num_iterations=1)
    G._assert_equal(want, got)
    assert pm.GetSweepKernelName() == "pm_sweep_pair_kernel"


@pytest.mark.parametrize("radius,step", [(2, 1), (5, 2)])
def test_generic_kernel_other_windows(pm_oracle, radius, step):
    views = scene(4, 40, 30)
    want, got, pm = G._run_both(pm_oracle, views, 1, [0, 2, 3], geom_consistency=0, filter=1, window_radius=radius,
                                window_step=step, num_iterations=1)
    G._assert_equal(want, got)
    assert pm.GetSweepKernelName() == "pm_sweep_kernel"


def test_baseline_source_count_s20_m15(pm_oracle):
    """S = 20 sources, M = 15 samples (BASELINE config[1]'s LDS layout and task lists) on a tiny image, two sweeps."""
    views = scene(21, 24, 18, 3.6 * 20)
    src = [i for i in range(21) if i != 10]
    want, got, _ = G._run_both(pm_oracle, views, 10, src, geom_consistency=0, filter=0, max_sweeps=2)
    G._assert_equal(want, got, ("depth", "normal", "cost", "sel_prob"))


def test_concurrent_runs_from_host_threads_share_launches(pm_oracle):
    """pm_run from several host threads at once is coalesced into batched launches (pm_api.cpp: RunCoalesced)."""
    G.test_concurrent_runs_from_host_threads_share_launches(pm_oracle)


def test_batched_run_equals_single_runs(pm_oracle):
    views = scene(5, 35, 27)
    pms, wants = [], []
    for ref, src in [(1, [0, 2, 3]), (2, [1, 3, 4])]:
        dmin, dmax = syn.depth_range(views, ref)
        o, h = paired_options(pm_oracle, depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1, num_iterations=1)
        wants.append(pm_oracle.run(o, oracle_inputs(views), ref, src, want_cost=True))
        pms.append(mvs.PatchMatch(h, hip_problem(views, ref, src)))
    mvs.run_batch(pms)
    for pm, want in zip(pms, wants):
        got = dict(depth=pm.GetDepthMap(), normal=pm.GetNormalMap(), sel_prob=pm.GetSelProbMap(), cost=pm.GetCostMap(),
                   mask=pm.GetConsistencyMask())
        G._assert_equal(want, got)


def test_host_side_cases_of_the_gpu_suite(pm_oracle):
    """Error behaviour of the C ABI, source images larger than the reference's slot, one source with 25 samples: the GPU
    tests themselves, library swapped."""
    G.test_error_behaviour()
    G.test_cached_images_in_distant_slabs_are_rehomed(pm_oracle)
    G.test_sources_larger_than_reference_slot(pm_oracle)
    G.test_single_source_and_many_samples(pm_oracle)


def test_committed_golden_fixture():
    """tests/golden/pm_48x36.npz (committed answer of the device-order oracle: one iteration + filter, every output map)
    reproduced by the product's kernels on the stand-in -- the GPU test of the same name, library swapped."""
    G.test_against_committed_golden_fixture()
