import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # experimental builds of the library (scripts/profile_pm_gather_diag.sh: e.g. the 16-bit packed-image format) are
    # put through the same parity tests by pointing the loader at them
    if os.environ.get("COLMAP_AMD_TEST_EMUL", "0") != "0":
        _use_stand_in()
    alt = os.environ.get("COLMAP_AMD_TEST_LIB")
    if alt:
        from colmap_amd import build as _b
        _b.LIB_PATH = os.path.abspath(alt)


# ---- order of the GPU suite ---------------------------------------------------------------------------------------
# `pytest -x` stops at the first failure, so the order decides what a red run still proves. The contract runs first:
# BASELINE-shape parity of both paths, the comparison with the reference's own build (oracle/_ref), the committed
# golden fixtures, fusion at the reference's defaults; then the rest of the hot paths; the long tail of camera models,
# losses and priors last -- one ill-conditioned long-tail case can then no longer hide the PatchMatch suite.
_CONTRACT = ("test_baseline_", "test_pm_ref.py::", "test_full_photometric_with_filter", "test_geometric_consistency_and_filter",
             "golden_fixture", "test_reference_integration_case_hip", "test_hip_fusion_equals_parallel_oracle",
             "test_solution_matches_oracle", "test_sparse_schur_tier_matches_oracle", "test_dense_schur_tier_matches_oracle",
             "test_backend_interface_reference_cases", "test_initial_state_and_cost", "test_pose_tables_and_ref_filter")
_LONG_TAIL = ("camera_models", "fisheye_models", "test_opencv_model", "test_radial_model", "robust_losses", "priors",
              "pose_prior", "_cli", "test_cpp_", "full_size_properties", "stereo_fusion_command")


# The two long solves through the reference's own build (tests/test_pm_ref.py: minutes each of a kernel that occupies a
# few dozen CUs) run in background processes from the moment the collection shows they are selected, and their tests
# come LAST: the reference solves overlap with the whole rest of the suite instead of adding to it (ref_pm_cases.py).
_REF_BACKGROUND = {"test_reference_full_solve_config0_photometric": ("config0_photometric", True),
                   "test_reference_full_solve_bench_crop": ("bench_crop_384x288", False)}


def _gpu_tier(nodeid: str) -> int:
    if any(k in nodeid for k in _REF_BACKGROUND):
        return 3
    if any(k in nodeid for k in _LONG_TAIL):
        return 2
    return 0 if any(k in nodeid for k in _CONTRACT) else 1


def _start_reference_workers(items):
    selected = [v for k, v in _REF_BACKGROUND.items() if any(k in it.nodeid for it in items)]
    if not selected:
        return
    try:
        import ref_pm
        import ref_pm_cases
        import torch
        if not (ref_pm.available() and torch.cuda.is_available()):
            return
        slow = int(os.environ.get("COLMAP_AMD_TEST_SLOW", "0") or 0)
        for name, oracle0 in selected:
            ref_pm_cases.start(name, fast=False, oracle0=oracle0 or slow >= 2)
            if slow and ref_pm.fast_available():
                ref_pm_cases.start(name, fast=True)
    except Exception as e:   # the tests then solve in-process
        sys.stderr.write(f"conftest: reference workers not started ({e!r})\n")


def pytest_collection_modifyitems(config, items):
    gpu = [i for i, it in enumerate(items) if it.get_closest_marker("gpu") is not None]
    ordered = sorted((items[i] for i in gpu), key=lambda it: _gpu_tier(it.nodeid))   # stable: file order inside a tier
    for slot, it in zip(gpu, ordered):
        items[slot] = it


def pytest_collection_finish(session):
    # after deselection (-m / -k): only what will really run starts a worker
    if not session.config.option.collectonly:
        _start_reference_workers(session.items)


class _StandInLibrary:
    """The three CPU builds of tests/hip_emul behind one object with the entry points of libcolmap_amd.so."""

    def __init__(self, libs):
        self._libs = libs

    def colmap_amd_set_switch(self, name, value):   # every stand-in library keeps its own table
        for L in self._libs:
            L.colmap_amd_set_switch(name, value)

    def __getattr__(self, name):
        for L in self._libs:
            try:
                return getattr(L, name)
            except AttributeError:
                pass
        raise AttributeError(name)


def _use_stand_in():
    """COLMAP_AMD_TEST_EMUL=1: every loader of the C-ABI library hands out the CPU stand-in builds instead, so that ANY
    test marked gpu can be tried without a GPU (`COLMAP_AMD_TEST_EMUL=1 pytest -m gpu tests/test_pm_gpu.py -k window`):
    a developer's tool next to the curated tests/test_*_emul.py; tests that need torch.cuda itself still need a GPU.
    Never set by the product or by the default test runs."""
    import ctypes as C
    import subprocess
    emul = os.path.join(ROOT, "tests", "hip_emul")
    for script in ("build.sh", "build_ba.sh", "build_pm.sh"):
        subprocess.check_call(["sh", os.path.join(emul, script)])
    lib = _StandInLibrary([C.CDLL(os.path.join(emul, n)) for n in ("libpm_emul.so", "libba_emul.so", "libfusion_emul.so")])
    lib.pm_last_error.restype = C.c_char_p
    lib.pm_device_count.restype = C.c_int
    lib.ba_last_error.restype = C.c_char_p
    from colmap_amd import _lib, estimators, fusion, mvs
    for mod in (_lib, estimators, fusion, mvs):
        mod.lib = lambda: lib


@pytest.fixture(scope="session")
def pm_oracle():
    import pm_oracle as m
    m.build()
    return m
