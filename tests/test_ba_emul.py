"""colmap_amd/csrc/ba_kernels.hip + ba_schur_explicit.hip -- the UNMODIFIED product sources, host loop and kernels --
executed on the CPU against the fp64 checker (oracle/ba_oracle.c), by running parity tests of tests/test_ba_gpu.py
with the library swapped for a CPU build of the same files.

tests/hip_emul/ is a HIP stand-in for exactly this purpose (the lanes of a workgroup as fibers, cross-lane
primitives and __syncthreads as barriers, v_mfma_f64_16x16x4_f64 with the hardware's operand / result layout,
streams synchronous): test infrastructure, never loaded by the product, whose library is built by hipcc and has
no CPU path. What these tests pin without a GPU: the linearisation of every camera-model family, the c-order /
p-order layouts and the tiled point passes, the wave-per-chunk reductions, the Schur-Jacobi blocks on the (emulated)
matrix cores, the pipelined PCG with its device-side stopping test, the LM loop, both exact tiers (explicit reduced
camera system with fixed-point accumulation, blocked Cholesky), priors, rigs, robust losses, the adapters. Sizes are
the small ones of the GPU suite (a lane is a fiber here: ~1 us per cross-lane primitive and lane). The GPU tests
run the same functions through the hipcc build."""
import ctypes as C
import os
import subprocess

import pytest

import test_ba_gpu as G
from colmap_amd import estimators as est

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_emul")
_CSRC = os.path.join(os.path.dirname(_HERE), "..", "colmap_amd", "csrc")
_LIB = None


def _emul_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libba_emul.so")
        deps = [os.path.join(_CSRC, f) for f in ("ba_kernels.hip", "ba_schur_explicit.hip", "ba_schur_explicit.h")]
        deps += [os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_HERE, "rccl", "rccl.h"),
                 os.path.join(_HERE, "build_ba.sh"), os.path.join(os.path.dirname(_HERE), "..", "include", "colmap_amd_ba.h")]
        if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            subprocess.check_call(["sh", os.path.join(_HERE, "build_ba.sh")])
        _LIB = C.CDLL(path)
        _LIB.ba_last_error.restype = C.c_char_p
    return _LIB


@pytest.fixture(autouse=True)
def emulated_library(monkeypatch):
    """est.solve_flat(..., gpu_index=0) -- what the GPU tests call -- reaches ba_solve of the CPU build."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and "HIP_EMUL_CXX" not in os.environ:
        pytest.skip("the stand-in is built with ROCm's clang++ as host compiler")
    monkeypatch.setattr(est, "lib", _emul_lib)


def test_emulated_library_is_the_one_under_test():
    L = est.lib()
    assert L is _emul_lib() and os.path.basename(L._name) == "libba_emul.so"
    est._check_abi(L)


@pytest.mark.parametrize("frames,points,track,mixed", [(6, 40, 4, True), (12, 300, 5, False)])
def test_solution_matches_oracle(frames, points, track, mixed):
    G.test_solution_matches_oracle(frames, points, track, mixed)


def test_constant_blocks_gauges_and_partial_problems():
    G.test_constant_blocks_are_untouched_bitwise()
    G.test_shared_intrinsics_and_three_point_gauge()
    G.test_only_points_variable_and_only_cameras_variable()
    G.test_error_behaviour()


def test_reference_backend_cases_and_golden_fixture():
    G.test_backend_interface_reference_cases()
    G.test_against_committed_golden_fixture()
    G.test_reference_pose_prior_backend_case()


def test_exact_tiers():
    """DENSE_SCHUR / AUTO against the checker's exact tier; explicit formation against operator products."""
    G.test_dense_schur_tier_matches_oracle()
    G.test_exact_tier_explicit_formation_equals_operator_products()


def test_blocked_cholesky_beyond_one_panel():
    """n_c = 700: eleven 64-wide panels, three outer panels of 256, the two-stream lookahead order; the panel and
    trailing-update kernels on the emulated matrix cores (four waves per workgroup)."""
    fp = G._flat(120, 1500, 6, seed=120)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = G._both(fp, max_num_iterations=3, linear_solver_type=est.SOLVER_AUTO)
    assert got.linear_solver_used == want.linear_solver_used == est.SOLVER_SPARSE_SCHUR
    assert (got.log_linear_iters[:got.num_iterations] == 1).all()
    import numpy as np
    np.testing.assert_allclose(got.log_cost, want.log_cost, rtol=1e-7)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)


@pytest.mark.parametrize("model", [9, 14, 17])
def test_camera_model_families(model):
    """RADIAL_FISHEYE, SIMPLE_FISHEYE (no distortion parameter), EQUIRECTANGULAR; SIMPLE_RADIAL / PINHOLE / OPENCV
    variants run in the mixed problems above."""
    params = {9: (900.0, 512.0, 384.0, 0.03, -0.004), 14: (900.0, 512.0, 384.0), 17: (1024.0, 768.0)}[model]
    G._model_matches_oracle(model, params)


def test_rigs_priors_and_robust_loss():
    G.test_rig_frames_match_oracle()
    G.test_constant_rig_from_world_rotation_matches_oracle()
    G.test_pose_prior_adjuster_on_rigs_matches_oracle()
    G.test_robust_losses_match_oracle(est.LossFunctionType.SOFT_L1, 1.0)
