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
    alt = os.environ.get("COLMAP_AMD_TEST_LIB")
    if alt:
        from colmap_amd import build as _b
        _b.LIB_PATH = os.path.abspath(alt)


@pytest.fixture(scope="session")
def pm_oracle():
    import pm_oracle as m
    m.build()
    return m
