"""Depth-map fusion: colmap::mvs::StereoFusion (reference mvs/fusion.{h,cc}) and the
`colmap stereo_fusion` command (exe/mvs.cc:299-386) on top of the fusion C ABI
(include/colmap_amd_fusion.h -> colmap_amd/csrc/fusion.hip: the traversal, the medians and the
compaction run on the GPU; there is no CPU path).

    python -m colmap_amd.fusion --workspace_path DENSE --output_path DENSE/fused.ply \\
        [--input_type geometric] [--StereoFusion.min_num_pixels 5] ...
"""
from __future__ import annotations

import argparse
import collections.abc
import ctypes as C
import os
import struct
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import mvs
from . import workspace as W
from ._lib import lib

FLT_MAX = float(np.finfo(np.float32).max)


class fusion_options(C.Structure):
    _fields_ = [("min_num_pixels", C.c_int32), ("max_num_pixels", C.c_int32), ("max_traversal_depth", C.c_int32),
                ("check_num_images", C.c_int32), ("max_reproj_error", C.c_double), ("max_depth_error", C.c_double),
                ("max_normal_error", C.c_double), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("num_threads", C.c_int32)]


class fusion_image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("K", C.c_float * 9), ("R", C.c_float * 9),
                ("T", C.c_float * 3), ("rgb", C.c_void_p), ("bitmap_width", C.c_int32), ("bitmap_height", C.c_int32),
                ("depth_map", C.c_void_p), ("normal_map", C.c_void_p), ("depth_width", C.c_int32),
                ("depth_height", C.c_int32), ("mask", C.c_void_p), ("used", C.c_int32)]


@dataclass
class StereoFusionOptions:
    """colmap::mvs::StereoFusionOptions (mvs/fusion.h:46-94)."""
    mask_path: str = ""
    num_threads: int = -1
    max_image_size: int = -1
    min_num_pixels: int = 5
    max_num_pixels: int = 10000
    max_traversal_depth: int = 100
    max_reproj_error: float = float(np.float32(2.0))
    max_depth_error: float = float(np.float32(0.01))
    max_normal_error: float = float(np.float32(10.0))
    check_num_images: int = 50
    use_cache: bool = False
    cache_size: float = 32.0
    bounding_box: Tuple[Tuple[float, float, float], Tuple[float, float, float]] = (
        (-FLT_MAX, -FLT_MAX, -FLT_MAX), (FLT_MAX, FLT_MAX, FLT_MAX))

    def Check(self) -> bool:  # fusion.cc:96-106
        return (0 <= self.min_num_pixels <= self.max_num_pixels and self.max_traversal_depth > 0 and
                self.max_reproj_error >= 0 and self.max_depth_error >= 0 and self.max_normal_error >= 0 and
                self.check_num_images > 0 and self.cache_size > 0)

    def to_c(self) -> fusion_options:
        o = fusion_options()
        lib().fusion_options_init(C.byref(o))
        for name in ("min_num_pixels", "max_num_pixels", "max_traversal_depth", "check_num_images",
                     "max_reproj_error", "max_depth_error", "max_normal_error", "num_threads"):
            setattr(o, name, getattr(self, name))
        o.bbox_min[:] = [float(v) for v in self.bounding_box[0]]
        o.bbox_max[:] = [float(v) for v in self.bounding_box[1]]
        return o


@dataclass
class FusionImage:
    """One workspace image handed to the fusion: pose at the model size + bitmap + maps."""
    width: int
    height: int
    K: np.ndarray
    R: np.ndarray
    T: np.ndarray
    rgb: Optional[np.ndarray]        # (bh, bw, 3) uint8 or None
    depth_map: Optional[np.ndarray]  # (dh, dw) float32
    normal_map: Optional[np.ndarray]  # (3, dh, dw) float32
    mask: Optional[np.ndarray] = None  # (dh, dw), non-zero = pre-masked
    used: bool = True


class VisibilityLists(collections.abc.Sequence):
    """The per-point image lists as the library hands them over: one index array + row pointers (fusion_get_visibility).
    Behaves like the list of arrays it stands for (len, indexing, slicing, iteration, zip) without materialising
    millions of small arrays: building that Python list took longer than the whole fusion on the device for a
    few 2560 x 1920 images (~0.35 us per point)."""

    def __init__(self, ptr: np.ndarray, idx: np.ndarray):
        self.ptr = np.asarray(ptr, np.int64)
        self.idx = np.asarray(idx, np.int32)

    def __len__(self) -> int:
        return len(self.ptr) - 1

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return self.idx[self.ptr[k]:self.ptr[k + 1]]

    def __iter__(self):
        idx, ptr = self.idx, self.ptr.tolist()
        for k in range(len(ptr) - 1):
            yield idx[ptr[k]:ptr[k + 1]]

    def counts(self) -> np.ndarray:
        return np.diff(self.ptr)


@dataclass
class FusedPoints:
    xyz: np.ndarray     # (n, 3) float32
    normal: np.ndarray  # (n, 3) float32
    rgb: np.ndarray     # (n, 3) uint8
    visibility: Sequence[np.ndarray] = field(default_factory=list)  # image indices per point (VisibilityLists from fuse())


class _HipEntryPoints:
    """The fusion_* entry points of libcolmap_amd.so (include/colmap_amd_fusion.h)."""

    def __init__(self):
        L = lib()
        L.fusion_last_error.restype = C.c_char_p
        L.fusion_num_points.restype = C.c_size_t
        self.run, self.num_points, self.get_points = L.fusion_run, L.fusion_num_points, L.fusion_get_points
        self.get_visibility, self.free, self.last_error = L.fusion_get_visibility, L.fusion_free, L.fusion_last_error


def fuse(options: StereoFusionOptions, images: Sequence[FusionImage], overlapping_images: Sequence[Sequence[int]],
         entry_points=None) -> FusedPoints:
    """StereoFusion::Run on in-memory inputs (fusion.cc:188-343) through fusion_run (HIP kernels).
    `entry_points` lets the tests hand the identical marshalled structs to the checker in oracle/."""
    if not options.Check():
        raise ValueError("Check failed: options_.Check()")
    L = entry_points or _HipEntryPoints()
    n = len(images)
    arr = (fusion_image * n)()
    keep = []
    for i, im in enumerate(images):
        a = arr[i]
        a.width, a.height = int(im.width), int(im.height)
        a.K[:] = np.asarray(im.K, np.float32).ravel().tolist()
        a.R[:] = np.asarray(im.R, np.float32).ravel().tolist()
        a.T[:] = np.asarray(im.T, np.float32).ravel().tolist()
        a.used = 1 if im.used and im.depth_map is not None and im.normal_map is not None else 0
        if not a.used:
            continue
        d = np.ascontiguousarray(im.depth_map, np.float32)
        nm = np.ascontiguousarray(im.normal_map, np.float32)
        if nm.shape != (3,) + d.shape:
            raise ValueError(f"image {i}: normal map shape {nm.shape} != (3, {d.shape[0]}, {d.shape[1]})")
        keep += [d, nm]
        a.depth_map, a.normal_map = d.ctypes.data, nm.ctypes.data
        a.depth_height, a.depth_width = d.shape
        if im.rgb is not None:
            rgb = np.ascontiguousarray(im.rgb, np.uint8)
            keep.append(rgb)
            a.rgb = rgb.ctypes.data
            a.bitmap_height, a.bitmap_width = rgb.shape[:2]
        if im.mask is not None:
            m = np.ascontiguousarray(np.asarray(im.mask) != 0, np.uint8)
            if m.shape != d.shape:
                raise ValueError(f"image {i}: mask shape {m.shape} != depth map shape {d.shape}")
            keep.append(m)
            a.mask = m.ctypes.data
    ptr = np.zeros(n + 1, np.int32)
    for i in range(n):
        ptr[i + 1] = ptr[i] + len(overlapping_images[i])
    idx = np.array([j for lst in overlapping_images for j in lst], np.int32)
    copt = options.to_c()
    res = C.c_void_p()
    rc = L.run(C.byref(copt), C.c_int32(n), arr, ptr.ctypes.data_as(C.c_void_p),
               idx.ctypes.data_as(C.c_void_p) if len(idx) else None, C.byref(res))
    if rc != 0:
        raise RuntimeError(L.last_error().decode())
    try:
        m = L.num_points(res)
        pts = np.zeros((m, 6), np.float32)
        rgb = np.zeros((m, 3), np.uint8)
        L.get_points(res, pts.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p))
        total = C.c_size_t(0)
        L.get_visibility(res, None, None, C.byref(total))
        vptr = np.zeros(m + 1, np.int64)
        vidx = np.zeros(max(total.value, 1), np.int32)
        L.get_visibility(res, vptr.ctypes.data_as(C.c_void_p), vidx.ctypes.data_as(C.c_void_p), C.byref(total))
    finally:
        L.free(res)
    return FusedPoints(pts[:, :3].copy(), pts[:, 3:].copy(), rgb, VisibilityLists(vptr, vidx[:total.value]))


# ------------------------------------------------------------------------------------------------
# files
# ------------------------------------------------------------------------------------------------

def write_binary_ply_points(path: str, pts: FusedPoints, write_normal: bool = True, write_rgb: bool = True):
    """WriteBinaryPlyPoints (util/ply.cc:378-431)."""
    with open(path, "wb") as f:
        hdr = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {len(pts.xyz)}\n"
        hdr += "property float x\nproperty float y\nproperty float z\n"
        if write_normal:
            hdr += "property float nx\nproperty float ny\nproperty float nz\n"
        if write_rgb:
            hdr += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
        hdr += "end_header\n"
        f.write(hdr.encode())
        fields = [("xyz", "<f4", (3,))]
        if write_normal:
            fields.append(("n", "<f4", (3,)))
        if write_rgb:
            fields.append(("rgb", "u1", (3,)))
        rec = np.zeros(len(pts.xyz), np.dtype(fields))
        rec["xyz"] = pts.xyz
        if write_normal:
            rec["n"] = pts.normal
        if write_rgb:
            rec["rgb"] = pts.rgb
        f.write(rec.tobytes())


def read_binary_ply_points(path: str) -> FusedPoints:
    with open(path, "rb") as f:
        raw = f.read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.decode().splitlines() if l.startswith("element vertex")][0].split()[-1])
    rec = np.frombuffer(body, np.dtype([("xyz", "<f4", (3,)), ("n", "<f4", (3,)), ("rgb", "u1", (3,))]), count=n)
    return FusedPoints(rec["xyz"].copy(), rec["n"].copy(), rec["rgb"].copy())


def write_points_visibility(path: str, visibility: Sequence[Sequence[int]]):
    """WritePointsVisibility (fusion.cc:526-541)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(visibility)))
        if isinstance(visibility, VisibilityLists):  # one interleaved array (count, indices ...) per point, one write
            cnt = visibility.counts()
            out = np.empty(len(cnt) + len(visibility.idx), "<u4")
            head = visibility.ptr[:-1] + np.arange(len(cnt), dtype=np.int64)  # position of every count word
            out[head] = cnt
            body = np.ones(len(out), bool)
            body[head] = False
            out[body] = visibility.idx
            f.write(out.tobytes())
            return
        for v in visibility:
            f.write(struct.pack("<I", len(v)))
            f.write(np.asarray(v, "<u4").tobytes())


def read_points_visibility(path: str, num_points: int) -> List[np.ndarray]:
    """ReadPointsVisibility (fusion.cc:543-562)."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        if n != num_points:
            raise ValueError(f"Check failed: file_num_points == num_points ({n} vs. {num_points})")
        words = np.frombuffer(f.read(), "<u4")
    # (count, indices ...) records: walk the count words (the only sequential part), slice the rest
    ptr = np.zeros(n + 1, np.int64)
    pos = 0
    counts = np.empty(n, np.int64)
    for k in range(n):
        if pos >= len(words):
            raise ValueError("visibility file truncated")
        counts[k] = words[pos]
        pos += 1 + int(words[pos])
    if pos > len(words):
        raise ValueError("visibility file truncated")
    np.cumsum(counts, out=ptr[1:])
    head = ptr[:-1] + np.arange(n, dtype=np.int64)
    body = np.ones(pos, bool)
    body[head] = False
    return VisibilityLists(ptr, words[:pos][body].astype(np.int32))


# ------------------------------------------------------------------------------------------------
# StereoFusion on a workspace
# ------------------------------------------------------------------------------------------------

class StereoFusion:
    """colmap::mvs::StereoFusion (mvs/fusion.h:96-139)."""

    def __init__(self, options: StereoFusionOptions, workspace_path: str, workspace_format: str = "COLMAP",
                 pmvs_option_name: str = "option-all", input_type: str = "geometric"):
        if not options.Check():
            raise ValueError("Check failed: options_.Check()")
        self.options_ = options
        self.workspace_path_ = workspace_path
        self.workspace_format_ = workspace_format
        self.pmvs_option_name_ = pmvs_option_name
        self.input_type_ = input_type
        self.fused_: Optional[FusedPoints] = None
        self.warnings_: List[str] = []

    def GetFusedPoints(self) -> FusedPoints:
        return self.fused_

    def GetFusedPointsVisibility(self):
        return self.fused_.visibility

    def Run(self):
        from PIL import Image as PILImage
        pmvs = self.workspace_format_.lower() == "pmvs"
        ws = W.Workspace(self.workspace_path_, self.workspace_format_, input_type=self.input_type_,
                         max_image_size=self.options_.max_image_size,
                         stereo_folder=f"stereo-{self.pmvs_option_name_}" if pmvs else "stereo")  # (:151-156)
        model = ws.GetModel()
        cfg = os.path.join(self.workspace_path_, ws.stereo_folder, "fusion.cfg")
        with open(cfg) as f:
            names = [l.strip() for l in f if l.strip() and not l.startswith("#")]
        overlapping = model.GetMaxOverlappingImagesFromPMVS() or \
            model.GetMaxOverlappingImages(self.options_.check_num_images, 0.0)  # (:180-186)
        images = [FusionImage(im.width, im.height, im.K, im.R, im.T, None, None, None, used=False) for im in model.images]
        for name in names:
            idx = model.GetImageIdx(name)
            bpath, dpath, npath = ws.GetBitmapPath(idx), ws.GetDepthMapPath(idx), ws.GetNormalMapPath(idx)
            if not (os.path.exists(bpath) and os.path.exists(dpath) and os.path.exists(npath)):
                self.warnings_.append(f"Ignoring image {name}, because input does not exist.")  # (:204-213)
                continue
            im = images[idx]
            with PILImage.open(bpath) as b:
                rgb = np.asarray(b.convert("RGB"), np.uint8)
            if self.options_.max_image_size > 0 and rgb.shape[:2] != (im.height, im.width):
                rgb = np.asarray(PILImage.fromarray(rgb).resize((im.width, im.height), PILImage.BILINEAR), np.uint8)
            im.rgb = rgb
            im.depth_map = mvs.read_mat(dpath)
            im.normal_map = mvs.read_mat(npath)
            im.used = True
            im.mask = self._mask(name, im.depth_map.shape)
        self.fused_ = fuse(self.options_, images, overlapping)
        if len(self.fused_.xyz) == 0:
            self.warnings_.append("Could not fuse any points. This is likely caused by incorrect settings - filtering "
                                  "must be enabled for the last call to patch match stereo.")

    def _mask(self, image_name: str, shape):
        """InitFusedPixelMask (fusion.cc:359-399): <mask_path>/<name>.png (or <name> when it already ends
        in .png), rescaled with a box filter to the depth-map size, 0 = masked."""
        if not self.options_.mask_path:
            return None
        from PIL import Image as PILImage
        path = os.path.join(self.options_.mask_path, image_name + ".png")
        if not os.path.exists(path) and image_name.lower().endswith(".png"):
            path = os.path.join(self.options_.mask_path, image_name)
        if not os.path.exists(path):
            return None
        with PILImage.open(path) as m:
            g = m.convert("L").resize((shape[1], shape[0]), PILImage.BOX)
        return (np.asarray(g) == 0).astype(np.uint8)


def _parse_bool(v: str) -> bool:
    if v.lower() in ("1", "true", "yes", "on"):
        return True
    if v.lower() in ("0", "false", "no", "off"):
        return False
    raise argparse.ArgumentTypeError(f"not a boolean: {v}")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="stereo_fusion", description=__doc__.split("\n\n")[0])
    ap.add_argument("--workspace_path", required=True)
    ap.add_argument("--workspace_format", default="COLMAP", help="{COLMAP, PMVS}")
    ap.add_argument("--pmvs_option_name", default="option-all")
    ap.add_argument("--input_type", default="geometric", help="{photometric, geometric}")
    ap.add_argument("--output_type", default="PLY", help="{BIN, TXT, PLY}")
    ap.add_argument("--output_path", required=True)
    d = StereoFusionOptions()
    for name in ("mask_path", "num_threads", "max_image_size", "min_num_pixels", "max_num_pixels", "max_traversal_depth",
                 "max_reproj_error", "max_depth_error", "max_normal_error", "check_num_images", "use_cache", "cache_size"):
        v = getattr(d, name)
        ap.add_argument(f"--StereoFusion.{name}", dest=f"sf_{name}", type=_parse_bool if isinstance(v, bool) else type(v),
                        default=v)
    return ap


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    if a.workspace_format.lower() not in ("colmap", "pmvs"):
        raise SystemExit(f"Invalid `workspace_format` {a.workspace_format} - supported values are 'COLMAP' or 'PMVS'.")
    if a.input_type.lower() not in ("photometric", "geometric"):
        raise SystemExit(f"Invalid `input_type` {a.input_type} - supported values are 'photometric' and 'geometric'.")
    if a.output_type.lower() not in ("bin", "ply", "txt"):
        raise SystemExit(f"Invalid `output_type` {a.output_type} - supported values are 'bin', 'ply' and 'txt'.")
    opts = StereoFusionOptions(**{k[3:]: v for k, v in vars(a).items() if k.startswith("sf_")})
    fuser = StereoFusion(opts, a.workspace_path, a.workspace_format.lower(), a.pmvs_option_name, a.input_type.lower())
    fuser.Run()
    for w in fuser.warnings_:
        print("W", w, file=sys.stderr)
    pts = fuser.GetFusedPoints()
    print(f"Number of fused points: {len(pts.xyz)}")
    if a.output_type.lower() == "ply":
        write_binary_ply_points(a.output_path, pts)
        write_points_visibility(a.output_path + ".vis", pts.visibility)
    else:
        # Reconstruction::ImportPLY (exe/mvs.cc:360-377): the sparse model with its points replaced
        sm = W.read_sparse_model(os.path.join(a.workspace_path, "sparse"))
        for im in sm.images.values():
            im.point3D_ids = np.full(len(im.point3D_ids), -1, np.int64)
        sm.points3D = {k + 1: W.SparsePoint3D(k + 1, pts.xyz[k].astype(np.float64), tuple(int(c) for c in pts.rgb[k]), 0.0, [])
                       for k in range(len(pts.xyz))}
        os.makedirs(a.output_path, exist_ok=True)
        (W.write_model_binary if a.output_type.lower() == "bin" else W.write_model_text)(sm, a.output_path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
