"""ctypes binding of oracle/_ref/libref_pm.so: the REFERENCE's own PatchMatchCuda (its .cu files compiled
where they lie under /root/reference by `make -C oracle ref`, see oracle/ref_shim/README.md).

TEST INFRASTRUCTURE ONLY (GPU tests). The library is prebuilt in this container and travels to the GPU
box with the snapshot; nothing here reads /root/reference at run time."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import pm_oracle  # struct layouts are shared (oracle/ref_shim/ref_pm.cpp)

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
_PATH = os.path.join(_DIR, "libref_pm.so")
# the same sources built with -ffp-contract=fast: the reference perturbed by <= 2 ulp per operation (noise floor)
_PATH_FAST = os.path.join(_DIR, "libref_pm_fast.so")
_libs = {}


def available() -> bool:
    return os.path.exists(_PATH)


def fast_available() -> bool:
    return os.path.exists(_PATH_FAST)


def lib(fast: bool = False):
    path = _PATH_FAST if fast else _PATH
    if path not in _libs:
        L = C.CDLL(path)
        L.ref_pm_create.restype = C.c_void_p
        L.ref_pm_last_error.restype = C.c_char_p
        L.ref_pm_destroy.argtypes = [C.c_void_p]
        L.ref_pm_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.ref_pm_run.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        _libs[path] = L
    return _libs[path]


class RefPatchMatch:
    """PatchMatchCuda(options, problem) of the reference; `images` as for pm_oracle.run."""

    def __init__(self, options: pm_oracle.Options, images, ref_idx, src_idxs, fast: bool = False):
        L = self._L = lib(fast)
        self._arr, self._keep = pm_oracle.make_images(images)
        self.H, self.W = images[ref_idx]["gray"].shape
        self.S = len(src_idxs)
        src = (C.c_int * self.S)(*src_idxs)
        self._h = L.ref_pm_create(C.byref(options), len(images), self._arr, int(ref_idx), self.S, src)
        if not self._h:
            raise RuntimeError("ref_pm_create: " + L.ref_pm_last_error().decode())

    def state(self):
        """What the constructor left on the device (before Run)."""
        H, W = self.H, self.W
        out = dict(rng=np.zeros((H, W, 6), np.uint32), ref_image=np.zeros((H, W), np.uint8),
                   sum=np.zeros((H, W), np.float32), sqsum=np.zeros((H, W), np.float32),
                   depth=np.zeros((H, W), np.float32), normal=np.zeros((3, H, W), np.float32))
        rc = self._L.ref_pm_get_state(self._h, *[out[k].ctypes.data for k in ("rng", "ref_image", "sum", "sqsum", "depth", "normal")])
        if rc:
            raise RuntimeError("ref_pm_get_state: " + self._L.ref_pm_last_error().decode())
        return out

    def run(self):
        H, W, S = self.H, self.W, self.S
        out = dict(depth=np.zeros((H, W), np.float32), normal=np.zeros((3, H, W), np.float32),
                   sel_prob=np.zeros((S, H, W), np.float32), mask=np.zeros((S, H, W), np.uint8),
                   cost=np.zeros((S, H, W), np.float32))
        rc = self._L.ref_pm_run(self._h, *[out[k].ctypes.data for k in ("depth", "normal", "sel_prob", "mask", "cost")])
        if rc:
            raise RuntimeError("ref_pm_run: " + self._L.ref_pm_last_error().decode())
        return out

    def close(self):
        if self._h:
            self._L.ref_pm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
