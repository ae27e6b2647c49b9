"""ctypes loader of the C-ABI library. Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_lib = None


class LibraryMissingError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise LibraryMissingError(
                f"{path} is missing: build it with `python -m colmap_amd.build` "
                "(__graft_entry__.build()); the MI355X paths have no CPU fallback")
        _lib = C.CDLL(path)
        _lib.pm_last_error.restype = C.c_char_p
        _lib.pm_device_count.restype = C.c_int
        if hasattr(_lib, "ba_last_error"):
            _lib.ba_last_error.restype = C.c_char_p
    return _lib
