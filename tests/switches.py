"""Development switches of the library for the tests (colmap_amd/csrc/switches.h): the shipped library reads no
environment; kernel variants kept for A/B comparisons are selected through `colmap_amd_set_switch`."""
import contextlib


def set_switch(lib, name, value):
    """value None restores the built-in default."""
    lib.colmap_amd_set_switch(name.encode(), None if value is None else str(value).encode())


@contextlib.contextmanager
def switches(lib, **values):
    """with switches(lib, COLMAP_AMD_BA_FORM_PAIRS=0): ... -- set for the block, defaults restored afterwards."""
    for k, v in values.items():
        set_switch(lib, k, v)
    try:
        yield
    finally:
        for k in values:
            set_switch(lib, k, None)
