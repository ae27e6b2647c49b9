"""ctypes binding of oracle/libpm_oracle.so (CPU restatement of COLMAP PatchMatch).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package (colmap_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpm_oracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libpm_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Options(C.Structure):
    """Field order mirrors `pmo_options` in pm_oracle.c."""

    _fields_ = [
        ("depth_min", C.c_double), ("depth_max", C.c_double),
        ("sigma_spatial", C.c_double), ("sigma_color", C.c_double),
        ("ncc_sigma", C.c_double),
        ("min_triangulation_angle", C.c_double),
        ("incident_angle_sigma", C.c_double),
        ("geom_consistency_regularizer", C.c_double),
        ("geom_consistency_max_cost", C.c_double),
        ("filter_min_ncc", C.c_double),
        ("filter_min_triangulation_angle", C.c_double),
        ("filter_geom_consistency_max_cost", C.c_double),
        ("window_radius", C.c_int), ("window_step", C.c_int),
        ("num_samples", C.c_int), ("num_iterations", C.c_int),
        ("filter_min_num_consistent", C.c_int),
        ("geom_consistency", C.c_int), ("filter", C.c_int),
        ("max_sweeps", C.c_int), ("memoize", C.c_int), ("num_threads", C.c_int),
        ("order", C.c_int),
    ]


class Image(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("K", C.c_float * 9), ("R", C.c_float * 9), ("T", C.c_float * 3),
        ("gray", C.c_void_p), ("depth", C.c_void_p), ("normal", C.c_void_p),
    ]


class RNG(C.Structure):
    _fields_ = [("x", C.c_uint32 * 5), ("d", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.pmo_exp.restype = C.c_float
        _lib.pmo_exp.argtypes = [C.c_float]
        _lib.pmo_rng_uniform.restype = C.c_float
        _lib.pmo_rng_next.restype = C.c_uint32
        _lib.pmo_rng_init.argtypes = [C.POINTER(RNG), C.c_uint64]
        _lib.pmo_run.restype = C.c_int
        _lib.pmo_num_threads.restype = C.c_int
    return _lib


def default_options(**kw) -> Options:
    """PatchMatchOptions defaults (patch_match_options.h:37-126); sigma_spatial
    resolved to window_radius like PatchMatchController::ProcessProblem
    (patch_match.cc:436-438)."""
    o = Options()
    o.depth_min, o.depth_max = -1.0, -1.0
    o.sigma_spatial, o.sigma_color = -1.0, float(np.float32(0.2))
    o.ncc_sigma = float(np.float32(0.6))
    o.min_triangulation_angle = 1.0
    o.incident_angle_sigma = float(np.float32(0.9))
    o.geom_consistency_regularizer = float(np.float32(0.3))
    o.geom_consistency_max_cost = 3.0
    o.filter_min_ncc = float(np.float32(0.1))
    o.filter_min_triangulation_angle = 3.0
    o.filter_geom_consistency_max_cost = 1.0
    o.window_radius, o.window_step = 5, 1
    o.num_samples, o.num_iterations = 15, 5
    o.filter_min_num_consistent = 2
    o.geom_consistency, o.filter = 1, 1
    o.max_sweeps, o.memoize, o.num_threads = -1, 1, 0
    o.order = 0  # 0: reference evaluation order, 1: device (HIP kernel) order
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    if o.sigma_spatial <= 0:
        o.sigma_spatial = float(o.window_radius)
    return o


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_images(images):
    """images: list of dicts {K(3,3), R(3,3), T(3,), gray(H,W) u8, depth?, normal?}.
    Returns (ctypes array, keepalive list)."""
    arr = (Image * len(images))()
    keep = []
    for i, im in enumerate(images):
        g = np.ascontiguousarray(im["gray"], dtype=np.uint8)
        keep.append(g)
        arr[i].height, arr[i].width = g.shape
        arr[i].K[:] = _f32(im["K"]).ravel().tolist()
        arr[i].R[:] = _f32(im["R"]).ravel().tolist()
        arr[i].T[:] = _f32(im["T"]).ravel().tolist()
        arr[i].gray = g.ctypes.data
        d = im.get("depth")
        if d is not None:
            d = _f32(d); keep.append(d); arr[i].depth = d.ctypes.data
        n = im.get("normal")
        if n is not None:
            n = _f32(n); keep.append(n); arr[i].normal = n.ctypes.data
    return arr, keep


def run(options: Options, images, ref_idx: int, src_idxs, want_cost=False):
    """Full PatchMatch solve on the CPU. Returns dict(depth, normal, sel_prob, mask[, cost])."""
    L = lib()
    arr, keep = make_images(images)
    H, W = images[ref_idx]["gray"].shape
    S = len(src_idxs)
    src = (C.c_int * S)(*src_idxs)
    depth = np.zeros((H, W), np.float32)
    normal = np.zeros((3, H, W), np.float32)
    sel = np.zeros((S, H, W), np.float32)
    mask = np.zeros((S, H, W), np.uint8)
    cost = np.zeros((S, H, W), np.float32) if want_cost else None
    rc = L.pmo_run(C.byref(options), len(images), arr, int(ref_idx), S, src,
                   depth.ctypes.data_as(C.c_void_p), normal.ctypes.data_as(C.c_void_p),
                   sel.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                   cost.ctypes.data_as(C.c_void_p) if want_cost else None)
    if rc != 0:
        raise RuntimeError(f"pmo_run failed with code {rc}")
    out = dict(depth=depth, normal=normal, sel_prob=sel, mask=mask)
    if want_cost:
        out["cost"] = cost
    return out


def pose_tables(images, ref_idx, src_idxs):
    L = lib()
    arr, keep = make_images(images)
    S = len(src_idxs)
    src = (C.c_int * S)(*src_idxs)
    poses = np.zeros((4, S, 43), np.float32)
    K = np.zeros((4, 4), np.float32)
    iK = np.zeros((4, 4), np.float32)
    L.pmo_pose_tables(len(images), arr, int(ref_idx), S, src, poses.ctypes.data_as(C.c_void_p),
                      K.ctypes.data_as(C.c_void_p), iK.ctypes.data_as(C.c_void_p))
    return poses, K, iK


def filter_ref_image(gray, radius, step, sigma_spatial, sigma_color):
    L = lib()
    g = np.ascontiguousarray(gray, np.uint8)
    H, W = g.shape
    img = np.zeros((H, W), np.uint8)
    s = np.zeros((H, W), np.float32)
    ss = np.zeros((H, W), np.float32)
    L.pmo_filter_ref_image(g.ctypes.data_as(C.c_void_p), W, H, int(radius), int(step),
                           C.c_float(sigma_spatial), C.c_float(sigma_color),
                           img.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                           ss.ctypes.data_as(C.c_void_p))
    return img, s, ss


def exp_f32(x: np.ndarray) -> np.ndarray:
    L = lib()
    return np.array([L.pmo_exp(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)


def sincos_f32(a: np.ndarray):
    L = lib()
    s, c = C.c_float(), C.c_float()
    S, Cc = [], []
    for v in np.asarray(a, np.float32).ravel():
        L.pmo_sincos(C.c_float(float(v)), C.byref(s), C.byref(c))
        S.append(s.value); Cc.append(c.value)
    return np.array(S, np.float32), np.array(Cc, np.float32)


def rng_stream(seed: int, n: int):
    """First n raw uint32 and uniform floats of the per-pixel XORWOW stream."""
    L = lib()
    st = RNG()
    L.pmo_rng_init(C.byref(st), seed)
    raw = [L.pmo_rng_next(C.byref(st)) for _ in range(n)]
    L.pmo_rng_init(C.byref(st), seed)
    uni = [L.pmo_rng_uniform(C.byref(st)) for _ in range(n)]
    return np.array(raw, np.uint32), np.array(uni, np.float32)
