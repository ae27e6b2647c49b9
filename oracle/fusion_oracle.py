"""ctypes binding of oracle/libfusion_oracle.so (fusion_oracle.cpp) -- TEST INFRASTRUCTURE ONLY.

    fuse(options, images, overlapping_images, mode)   mode 0: the reference's sequential walk, pixels row-major
                                                      mode 1: the same walk, turns in the order of the reference's
                                                              pool schedule (what fusion.hip computes)
                                                      mode 2: simulation of fusion.hip's passes (== mode 1)

Takes the same arguments as colmap_amd.fusion.fuse and reuses its marshalling, so both sides see the
identical structs.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfusion_oracle.so")
        src = os.path.join(_HERE, "fusion_oracle.cpp")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libfusion_oracle.so"])
        _LIB = C.CDLL(path)
        _LIB.fuo_last_error.restype = C.c_char_p
        _LIB.fuo_num_points.restype = C.c_size_t
    return _LIB


class _EntryPoints:
    def __init__(self, mode):
        L = lib()
        self.run = lambda *a: L.fuo_run(C.c_int32(mode), *a)
        self.num_points, self.get_points = L.fuo_num_points, L.fuo_get_points
        self.get_visibility, self.free, self.last_error = L.fuo_get_visibility, L.fuo_free, L.fuo_last_error


def fuse(options, images, overlapping_images, mode):
    from colmap_amd import fusion
    return fusion.fuse(options, images, overlapping_images, entry_points=_EntryPoints(mode))
