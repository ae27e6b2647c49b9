"""Host-side mirror of the reference's PatchMatch interface over the C ABI.

Same names and argument meaning as colmap::mvs (reference src/colmap/mvs/
patch_match.h:55-96, patch_match_options.h:37-126) / pycolmap.PatchMatchOptions
(src/pycolmap/pipeline/mvs.cc:27-117): `PatchMatchOptions`, `Image`,
`PatchMatch.Problem`, `PatchMatch(options, problem).Run()`, `GetDepthMap()`,
`GetNormalMap()`, `GetSelProbMap()`, `GetConsistencyGraph()`. All compute happens
in colmap_amd/lib/libcolmap_amd.so (HIP, gfx950); this module is ctypes plumbing.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from ._lib import lib


class pm_options(C.Structure):
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
        ("window_radius", C.c_int32), ("window_step", C.c_int32),
        ("num_samples", C.c_int32), ("num_iterations", C.c_int32),
        ("filter_min_num_consistent", C.c_int32),
        ("geom_consistency", C.c_int32), ("filter", C.c_int32),
        ("gpu_index", C.c_int32),
        ("max_sweeps", C.c_int32), ("inputs_on_device", C.c_int32),
        ("columns_per_group", C.c_int32), ("threads_per_group", C.c_int32),
    ]


class pm_image(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("K", C.c_float * 9), ("R", C.c_float * 9), ("T", C.c_float * 3),
        ("gray", C.c_void_p), ("depth_map", C.c_void_p), ("normal_map", C.c_void_p),
    ]


class pm_problem(C.Structure):
    _fields_ = [
        ("ref_image_idx", C.c_int32), ("num_src_images", C.c_int32),
        ("src_image_idxs", C.POINTER(C.c_int32)),
        ("num_images", C.c_int32), ("images", C.POINTER(pm_image)),
    ]


class PatchMatchError(RuntimeError):
    pass


def _check(rc: int):
    if rc != 0:
        raise PatchMatchError(lib().pm_last_error().decode())


@dataclass
class PatchMatchOptions:
    """colmap::mvs::PatchMatchOptions (reference patch_match_options.h:37-126)."""
    depth_min: float = -1.0
    depth_max: float = -1.0
    sigma_spatial: float = -1.0
    sigma_color: float = float(np.float32(0.2))
    ncc_sigma: float = float(np.float32(0.6))
    min_triangulation_angle: float = 1.0
    incident_angle_sigma: float = float(np.float32(0.9))
    geom_consistency_regularizer: float = float(np.float32(0.3))
    geom_consistency_max_cost: float = 3.0
    filter_min_ncc: float = float(np.float32(0.1))
    filter_min_triangulation_angle: float = 3.0
    filter_geom_consistency_max_cost: float = 1.0
    cache_size: float = 32.0
    gpu_index: str = "-1"
    max_image_size: int = -1
    window_radius: int = 5
    window_step: int = 1
    num_samples: int = 15
    num_iterations: int = 5
    filter_min_num_consistent: int = 2
    num_threads: int = -1
    geom_consistency: bool = True
    filter: bool = True
    allow_missing_files: bool = False
    write_consistency_graph: bool = False
    # extensions (not in the reference)
    max_sweeps: int = 0
    columns_per_group: int = 0
    threads_per_group: int = 0

    def to_c(self, inputs_on_device: bool = False) -> pm_options:
        o = pm_options()
        for name, _ in pm_options._fields_:
            if name in ("gpu_index", "inputs_on_device"):
                continue
            setattr(o, name, getattr(self, name))
        gpu = [int(x) for x in str(self.gpu_index).split(",") if x.strip() != ""]
        if len(gpu) != 1:
            # PatchMatch::Check, reference patch_match.cc:70-73
            raise PatchMatchError("Check failed: gpu_indices.size() == 1")
        o.gpu_index = gpu[0]
        o.inputs_on_device = 1 if inputs_on_device else 0
        return o


def _torch_device(gpu_index):
    """torch device of a PatchMatchOptions::gpu_index ("-1" = the current device)."""
    import torch
    g = int(str(gpu_index).split(",")[0])
    return torch.device("cuda", g if g >= 0 else torch.cuda.current_device())


@dataclass
class Image:
    """colmap::mvs::Image (reference mvs/image.h:40-98) with its grey bitmap."""
    K: np.ndarray
    R: np.ndarray
    T: np.ndarray
    bitmap: object  # (H, W) uint8 numpy array, or a torch CUDA tensor of that shape

    def GetWidth(self) -> int:
        return int(self.bitmap.shape[1])

    def GetHeight(self) -> int:
        return int(self.bitmap.shape[0])


def _ptr_of(a, dtype, keep: list):
    """Pointer to a host numpy array or a device torch tensor."""
    if a is None:
        return None, False
    if hasattr(a, "data_ptr"):  # torch tensor
        t = a.contiguous()
        keep.append(t)
        return t.data_ptr(), bool(t.is_cuda)
    arr = np.ascontiguousarray(a, dtype=dtype)
    keep.append(arr)
    return arr.ctypes.data, False


def release_cached_memory():
    """pm_release_cached_memory: device buffers of destroyed handles go back to the driver."""
    lib().pm_release_cached_memory()


def set_cached_memory_limit(gigabytes: float):
    """pm_set_cached_memory_limit: bound of the free lists that keep the device buffers of destroyed handles."""
    _check(lib().pm_set_cached_memory_limit(C.c_double(gigabytes)))


class ImageCache:
    """Device-side cache of packed source images shared between problems (pm_image_cache)."""

    def __init__(self, gpu_index: int = -1):
        self._c = C.c_void_p()
        self._pinned: dict = {}  # id -> bitmap: a cached address must not be recycled
        _check(lib().pm_image_cache_create(C.c_int32(gpu_index), C.byref(self._c)))

    def set_capacity(self, max_bytes: int):
        _check(lib().pm_image_cache_set_capacity(self._c, C.c_size_t(max_bytes)))

    def stats(self):
        e, h, m = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(lib().pm_image_cache_stats(self._c, C.byref(e), C.byref(h), C.byref(m)))
        return dict(entries=e.value, hits=h.value, misses=m.value)

    def close(self):
        if self._c:
            lib().pm_image_cache_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PatchMatch:
    """colmap::mvs::PatchMatch (reference mvs/patch_match.h:55-96)."""

    @dataclass
    class Problem:
        ref_image_idx: int = -1
        src_image_idxs: List[int] = field(default_factory=list)
        images: Optional[Sequence[Image]] = None
        depth_maps: Optional[Sequence[Optional[np.ndarray]]] = None   # (H, W) float32 each
        normal_maps: Optional[Sequence[Optional[np.ndarray]]] = None  # (3, H, W) float32 each

    def __init__(self, options: PatchMatchOptions, problem: "PatchMatch.Problem",
                 cache: Optional[ImageCache] = None):
        self.options_ = options
        self.problem_ = problem
        self.cache_ = cache
        self._h = None
        self._dims = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def close(self):
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            lib().pm_destroy(h)

    # -- marshalling -----------------------------------------------------------------
    def _marshal(self):
        prob = self.problem_
        if prob.images is None:
            raise PatchMatchError("Check failed: problem_.images != nullptr")
        keep: list = []
        n = len(prob.images)
        arr = (pm_image * n)()
        on_device = None
        used = set(prob.src_image_idxs) | {prob.ref_image_idx}
        for i, im in enumerate(prob.images):
            if i not in used:
                continue
            arr[i].width, arr[i].height = im.GetWidth(), im.GetHeight()
            arr[i].K[:] = np.asarray(im.K, np.float32).ravel().tolist()
            arr[i].R[:] = np.asarray(im.R, np.float32).ravel().tolist()
            arr[i].T[:] = np.asarray(im.T, np.float32).ravel().tolist()
            ptr, dev = _ptr_of(im.bitmap, np.uint8, keep)
            arr[i].gray = ptr
            if self.cache_ is not None:
                self.cache_._pinned[ptr] = keep[-1]
            flags = [dev]
            if prob.depth_maps is not None and i < len(prob.depth_maps) and prob.depth_maps[i] is not None:
                d = prob.depth_maps[i]
                if tuple(d.shape) != (im.GetHeight(), im.GetWidth()):
                    raise PatchMatchError("Check failed: depth map size == image size")
                ptr, dev = _ptr_of(d, np.float32, keep)
                arr[i].depth_map = ptr
                flags.append(dev)
            if prob.normal_maps is not None and i < len(prob.normal_maps) and prob.normal_maps[i] is not None:
                nm = prob.normal_maps[i]
                if tuple(nm.shape) != (3, im.GetHeight(), im.GetWidth()):
                    raise PatchMatchError("Check failed: normal map size == image size")
                ptr, dev = _ptr_of(nm, np.float32, keep)
                arr[i].normal_map = ptr
                flags.append(dev)
            for f in flags:
                if on_device is None:
                    on_device = f
                elif on_device != f:
                    raise PatchMatchError("inputs must be all host arrays or all device tensors")
        src = (C.c_int32 * len(prob.src_image_idxs))(*prob.src_image_idxs)
        cprob = pm_problem(int(prob.ref_image_idx), len(prob.src_image_idxs), src, n, arr)
        keep += [arr, src]
        return cprob, keep, bool(on_device)

    # -- reference surface --------------------------------------------------------------
    def Check(self):
        cprob, keep, on_device = self._marshal()
        copt = self.options_.to_c(on_device)
        _check(lib().pm_check(C.byref(copt), C.byref(cprob)))

    def Create(self):
        """The reference constructs PatchMatchCuda inside Run(); split out so that callers
        (bench) can time upload and solve separately."""
        cprob, keep, on_device = self._marshal()
        copt = self.options_.to_c(on_device)
        self.close()
        h = C.c_void_p()
        if self.cache_ is not None:
            _check(lib().pm_create_cached(C.byref(copt), C.byref(cprob), self.cache_._c, C.byref(h)))
        else:
            _check(lib().pm_create(C.byref(copt), C.byref(cprob), C.byref(h)))
        self._h = h
        ref = self.problem_.images[self.problem_.ref_image_idx]
        self._dims = (ref.GetHeight(), ref.GetWidth(), len(self.problem_.src_image_idxs))

    def Run(self):
        if self._h is None:
            self.Create()
        _check(lib().pm_run(self._h))

    def RunAsync(self):
        if self._h is None:
            self.Create()
        _check(lib().pm_run_async(self._h))

    def Synchronize(self):
        _check(lib().pm_synchronize(self._h))

    def _get(self, fn, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        _check(fn(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def GetDepthMap(self) -> np.ndarray:
        H, W, S = self._dims
        return self._get(lib().pm_get_depth_map, (H, W))

    def GetNormalMap(self) -> np.ndarray:
        H, W, S = self._dims
        return self._get(lib().pm_get_normal_map, (3, H, W))

    def GetDeviceMaps(self):
        """Depth and normal map as ONE device tensor [4, H, W] (depth, then the three normal slices),
        copied device-to-device (pm_copy_maps_to_device): input of the geometric pass without a host
        round trip."""
        import torch
        H, W, S = self._dims
        dev = _torch_device(self.options_.gpu_index)
        t = torch.empty((4, H, W), dtype=torch.float32, device=dev)
        _check(lib().pm_copy_maps_to_device(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(t[1:].data_ptr())))
        return t

    def GetSelProbMap(self) -> np.ndarray:
        H, W, S = self._dims
        return self._get(lib().pm_get_sel_prob_map, (S, H, W))

    def GetConsistentImageIdxs(self) -> np.ndarray:
        n = C.c_size_t(0)
        _check(lib().pm_get_consistent_image_idxs(self._h, None, C.c_size_t(0), C.byref(n)))
        buf = np.empty(n.value, np.int32)
        _check(lib().pm_get_consistent_image_idxs(self._h, buf.ctypes.data_as(C.c_void_p),
                                                  C.c_size_t(n.value), C.byref(n)))
        return buf

    def GetConsistencyGraph(self):
        """(width, height, flat idx list) = the ctor arguments of ConsistencyGraph
        (reference patch_match.cc:149-154, consistency_graph.cc:121-139)."""
        H, W, S = self._dims
        return W, H, self.GetConsistentImageIdxs()

    # -- extras --------------------------------------------------------------------------
    def GetCostMap(self) -> np.ndarray:
        H, W, S = self._dims
        return self._get(lib().pm_get_cost_map, (S, H, W))

    def GetConsistencyMask(self) -> np.ndarray:
        H, W, S = self._dims
        return self._get(lib().pm_get_consistency_mask, (S, H, W), np.uint8)

    def GetRefFilter(self):
        H, W, S = self._dims
        img = np.empty((H, W), np.uint8)
        s = np.empty((H, W), np.float32)
        ss = np.empty((H, W), np.float32)
        _check(lib().pm_get_ref_filter(self._h, img.ctypes.data_as(C.c_void_p),
                                       s.ctypes.data_as(C.c_void_p), ss.ctypes.data_as(C.c_void_p)))
        return img, s, ss

    def GetPoseTables(self):
        H, W, S = self._dims
        poses = np.empty((4, S, 43), np.float32)
        K = np.empty((4, 4), np.float32)
        iK = np.empty((4, 4), np.float32)
        _check(lib().pm_get_pose_tables(self._h, poses.ctypes.data_as(C.c_void_p),
                                        K.ctypes.data_as(C.c_void_p), iK.ctypes.data_as(C.c_void_p)))
        return poses, K, iK

    def GetEvaluationCount(self):
        """(NCC evaluations executed by the sweep launches, by ComputeInitialCost) of the last run."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().pm_get_evaluation_count(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def EnablePhaseProfile(self, enable=True):
        _check(lib().pm_enable_phase_profile(self._h, 1 if enable else 0))

    def EnableProgressTrace(self, enable=True):
        """Debug: record when every wave of the 11 x 11 sweep kernel reaches rows 0, 128, 256, ... (last launch)."""
        _check(lib().pm_enable_progress_trace(self._h, 1 if enable else 0))

    def GetProgressTrace(self):
        """(groups, samples) uint64 array of 100 MHz device-clock stamps; 0 = row not reached."""
        g, s = C.c_int32(), C.c_int32()
        _check(lib().pm_get_progress_trace(self._h, None, C.c_size_t(0), C.byref(g), C.byref(s)))
        out = np.zeros((g.value, s.value), np.uint64)
        _check(lib().pm_get_progress_trace(self._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(g), C.byref(s)))
        return out

    def GetPhaseProfile(self):
        out = (C.c_ulonglong * 10)()
        _check(lib().pm_get_phase_profile(self._h, out))
        return list(out)

    PHASE_NAMES = ("setup", "P0_tile", "P1_hypotheses", "P1w_patch_weights", "P2_priors", "P3a_cdf", "P3b_draws",
                   "P3c_tasks", "P4_homographies", "P4_tap_rounds", "P4_normalise", "P5a_sums", "P5b_argmin",
                   "P5c_winner_tasks", "P6_homographies", "P6_tap_rounds", "P6_normalise", "P7_messages", "P8_rowend")

    def GetPhaseProfileSlots(self):
        """{phase: shader-clock cycles summed over the waves} of the profiled four-wave kernel + 'waves'."""
        out = (C.c_ulonglong * 24)()
        _check(lib().pm_get_phase_profile_slots(self._h, out, 24))
        d = {n: int(out[i]) for i, n in enumerate(self.PHASE_NAMES)}
        d["waves"] = int(out[23])
        return d

    def GetLaunchShape(self):
        """(reference images per sweep launch, sweep launches in flight together) of the last run."""
        a, b = C.c_int32(1), C.c_int32(1)
        _check(lib().pm_get_launch_shape(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def GetSweepTiming(self):
        ms = C.c_double(0)
        n = C.c_int32(0)
        _check(lib().pm_get_sweep_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


    def GetSweepKernelName(self):
        name = C.c_char_p()
        _check(lib().pm_get_sweep_kernel_name(self._h, C.byref(name)))
        return (name.value or b"").decode()

    def GetSweepTimes(self):
        """ms of every sweep launch of the last run (pm_get_sweep_times)."""
        n = C.c_int32(0)
        _check(lib().pm_get_sweep_times(self._h, None, 0, C.byref(n)))
        out = (C.c_float * max(n.value, 1))()
        _check(lib().pm_get_sweep_times(self._h, out, n.value, C.byref(n)))
        return [out[i] for i in range(n.value)]


def run_batch(pms: Sequence[PatchMatch], wait: bool = True):
    """Solve several same-shaped problems in one batched run (pm_run_batch): every
    kernel launch covers all of them. Bit-identical to running them one by one."""
    for pm in pms:
        if pm._h is None:
            pm.Create()
    arr = (C.c_void_p * len(pms))(*[pm._h for pm in pms])
    if wait:
        _check(lib().pm_run_batch(arr, len(pms)))
    else:
        _check(lib().pm_run_batch_async(arr, len(pms)))


def write_mat(path: str, a: np.ndarray):
    """Mat<float>::Write (reference mvs/mat.cc:58-65): ASCII `W&H&D&` + little-endian
    float32, slice-major."""
    a = np.asarray(a, np.float32)
    if a.ndim == 2:
        a = a[None]
    d, h, w = a.shape
    with open(path, "wb") as f:
        f.write(f"{w}&{h}&{d}&".encode())
        f.write(a.astype("<f4").tobytes())


def read_mat(path: str) -> np.ndarray:
    """Mat<float>::Read (reference mvs/mat.cc:41-56)."""
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"&", 3)
    w, h, d = int(parts[0]), int(parts[1]), int(parts[2])
    arr = np.frombuffer(parts[3], dtype="<f4", count=w * h * d).reshape(d, h, w)
    return arr[0] if d == 1 else arr


# ---------------------------------------------------------------------------------------------
# PatchMatchController: problem list, two-pass schedule, output files, resume
# ---------------------------------------------------------------------------------------------

@dataclass
class WorkspaceImage:
    """What the controller needs from the workspace per image (reference mvs/workspace.h,
    model.h): a name relative to `images/`, calibration, pose and the grey bitmap."""
    name: str
    K: np.ndarray
    R: np.ndarray
    T: np.ndarray
    bitmap: object
    depth_range: Optional[tuple] = None  # (min, max); reference: Model::ComputeDepthRanges (model.cc:178-218)


class PatchMatchController:
    """colmap::mvs::PatchMatchController (reference mvs/patch_match.cc:156-535) for an in-memory
    workspace:
      * problems from a `patch-match.cfg`-style list of (reference name, source spec) pairs where
        the source spec is "__all__" or an explicit list of names (:239-359; "__auto__" needs the
        sparse model's shared-point statistics and is left to the caller);
      * with geom_consistency: pass 1 = photometric, no filtering, ALL problems; barrier; pass 2 =
        the requested options (:183-204);
      * outputs `<workspace>/stereo/{depth_maps,normal_maps}/<name>.<photometric|geometric>.bin`
        in Mat<float> format (:530-534, mat.cc:58-65); a problem whose outputs exist is skipped
        (:410-414) -- the reference's resume mechanism;
      * problems are sharded over ranks like the thread-per-GPU pool (:177,190-204,394) and, unlike
        the reference, up to `batch_size` same-shaped problems of a rank share every kernel launch;
        between the passes the photometric maps stay in HBM (device-to-device copy out of every
        handle, colmap_amd.distributed.exchange_maps_device = RCCL all-gather between the ranks) and
        the geometric pass reads them as device inputs, instead of going through the file system.
    """

    def __init__(self, options: PatchMatchOptions, images: Sequence[WorkspaceImage], workspace_path: str,
                 problems: Optional[Sequence[tuple]] = None, batch_size: int = 8, rank: int = 0,
                 world_size: int = 1):
        self.options_ = options
        self.images_ = list(images)
        self.workspace_path_ = workspace_path
        self.batch_size_ = max(1, batch_size)
        self.rank_, self.world_ = rank, world_size
        self.image_cache_: Optional[ImageCache] = None  # packed sources shared by all problems
        names = [im.name for im in self.images_]
        self.index_ = {n: i for i, n in enumerate(names)}
        if problems is None:
            problems = [(n, "__all__") for n in names]
        self.problems_ = []
        for ref_name, spec in problems:
            ref = self.index_[ref_name]
            if spec == "__all__":
                src = [i for i in range(len(names)) if i != ref]
            elif isinstance(spec, str) and spec.startswith("__auto__"):
                raise PatchMatchError("__auto__ source selection needs the sparse model (not available here)")
            else:
                src = [self.index_[n] for n in spec]
            if not src:
                raise PatchMatchError(f"No source images for reference image {ref_name}")
            self.problems_.append((ref, src))
        self.timings = {}
        self.device_bitmaps_: dict = {}

    @classmethod
    def FromWorkspace(cls, options: PatchMatchOptions, workspace_path: str, workspace_format: str = "COLMAP",
                      pmvs_option_name: str = "option-all", config_path: str = "", **kw) -> "PatchMatchController":
        """PatchMatchController(options, workspace_path, workspace_format, pmvs_option_name,
        config_path) + ReadWorkspace + ReadProblems (patch_match.cc:159-237,239-359): the undistorted
        COLMAP workspace on disk -- `sparse/` model, `images/`, `stereo/patch-match.cfg`."""
        import os
        from . import workspace as W
        pmvs = workspace_format.lower() == "pmvs"
        ws = W.Workspace(workspace_path, workspace_format, max_image_size=options.max_image_size,
                         input_type="photometric" if options.geom_consistency else "",
                         stereo_folder=f"stereo-{pmvs_option_name}" if pmvs else "stereo")  # (:212-216)
        if pmvs:
            W.import_pmvs_workspace(ws, pmvs_option_name)  # (:229-233)
        model = ws.GetModel()
        cfg = config_path or os.path.join(workspace_path, ws.stereo_folder, "patch-match.cfg")
        with open(cfg) as f:
            lines = f.read().splitlines()
        warnings: List[str] = []
        problems = W.read_patch_match_config(lines, model, options.min_triangulation_angle, warnings.append)
        ranges = model.ComputeDepthRanges()
        used = sorted({i for ref, src in problems for i in [ref] + src})
        images = []
        for idx, im in enumerate(model.images):
            bitmap = None
            if idx in used:
                if os.path.exists(ws.GetBitmapPath(idx)):
                    bitmap = ws.GetBitmap(idx)
                    if bitmap.shape != (im.height, im.width):
                        raise PatchMatchError(f"Check failed: bitmap size of {model.GetImageName(idx)} "
                                              f"{bitmap.shape[::-1]} != model image size {(im.width, im.height)}")
                elif not options.allow_missing_files:
                    raise PatchMatchError(f"Missing image or map dependency for image {idx}: "
                                          f"{model.GetImageName(idx)}")  # (:494-500)
            rng = ranges[idx] if ranges[idx][0] > 0 and ranges[idx][1] > 0 else None
            images.append(WorkspaceImage(model.GetImageName(idx), im.K, im.R, im.T, bitmap, rng))
        # sources whose bitmap is missing are skipped (allow_missing_files, :486-493)
        kept = []
        for ref, src in problems:
            if images[ref].bitmap is None:
                continue
            src = [s_ for s_ in src if images[s_].bitmap is not None]
            if src:
                kept.append((model.GetImageName(ref), [model.GetImageName(s_) for s_ in src]))
        ctl = cls(options, images, workspace_path, problems=kept, **kw)
        ctl.workspace_ = ws
        ctl.warnings_ = warnings
        return ctl

    # -- paths ------------------------------------------------------------------------------
    def _paths(self, image_idx: int, output_type: str):
        import os
        name = f"{self.images_[image_idx].name}.{output_type}.bin"
        base = os.path.join(self.workspace_path_, getattr(getattr(self, "workspace_", None), "stereo_folder", "stereo"))
        return os.path.join(base, "depth_maps", name), os.path.join(base, "normal_maps", name)

    def _device(self, options: PatchMatchOptions):
        return _torch_device(options.gpu_index)

    def _device_bitmap(self, idx: int, device):
        """The grey bitmap of image `idx` in HBM (uploaded once, shared by both passes so that the packed
        source cache -- keyed by the bitmap address -- keeps hitting)."""
        import torch
        t = self.device_bitmaps_.get(idx)
        if t is None:
            t = torch.from_numpy(np.array(self.images_[idx].bitmap, np.uint8, order="C", copy=True)).to(device)  # (a read-only array cannot be wrapped)
            self.device_bitmaps_[idx] = t
        return t

    def _run_pass(self, options: PatchMatchOptions, maps: Optional[dict], device_maps: Optional[dict] = None):
        """One pass over this rank's problems. `maps`: image index -> (depth, normal) for the geometric
        pass, host arrays or device tensors. `device_maps`: when given (the photometric pass that
        precedes a geometric one), filled with image index -> device tensor [4, H, W] and all inputs are
        handed over as device tensors."""
        import os
        from . import distributed as D
        output_type = "geometric" if options.geom_consistency else "photometric"
        mine = [self.problems_[i] for i in D.shard_problems(len(self.problems_), self.rank_, self.world_)]
        on_device = device_maps is not None or (maps is not None and any(hasattr(v[0], "data_ptr") for v in maps.values()))
        if on_device:
            dev = self._device(options)
            used = sorted({i for ref, src in mine for i in [ref] + list(src)})
            images = [Image(im.K, im.R, im.T, self._device_bitmap(i, dev) if i in used else im.bitmap)
                      for i, im in enumerate(self.images_)]
        else:
            images = [Image(im.K, im.R, im.T, im.bitmap) for im in self.images_]
        if self.image_cache_ is None:
            self.image_cache_ = ImageCache(int(options.gpu_index))
            # PatchMatchOptions::cache_size (GB) bounds the reference's CachedWorkspace
            # (patch_match_options.h:118-125); here it bounds the packed sources kept in HBM
            self.image_cache_.set_capacity(int(options.cache_size * (1 << 30)))
        results = {}
        todo = []
        for ref, src in mine:
            dpath, npath = self._paths(ref, output_type)
            if os.path.exists(dpath) and os.path.exists(npath):  # resume (:410-414)
                results[ref] = (read_mat(dpath), read_mat(npath))
                if device_maps is not None:
                    import torch
                    device_maps[ref] = torch.from_numpy(np.concatenate([results[ref][0][None], results[ref][1]], 0)).to(dev)
                continue
            todo.append((ref, src))
        # batches of same-shaped problems
        def shape_key(prob):
            ref, src = prob
            return (self.images_[ref].bitmap.shape, len(src), tuple(sorted({self.images_[s].bitmap.shape for s in src})))
        todo.sort(key=shape_key)
        i = 0
        while i < len(todo):
            j = i
            while j < len(todo) and j - i < self.batch_size_ and shape_key(todo[j]) == shape_key(todo[i]):
                j += 1
            pms = []
            for ref, src in todo[i:j]:
                o = PatchMatchOptions(**{**options.__dict__})
                o.gpu_index = options.gpu_index
                rng = self.images_[ref].depth_range
                if (o.depth_min < 0 or o.depth_max < 0):
                    if rng is None:
                        raise PatchMatchError("You must manually set the minimum and maximum depth, since no "
                                              "sparse model is provided in the workspace.")  # (:425-434)
                    o.depth_min, o.depth_max = rng
                if o.sigma_spatial <= 0:
                    o.sigma_spatial = float(o.window_radius)  # (:436-438)
                o.filter_min_num_consistent = min(len(src), o.filter_min_num_consistent)  # (:458-460)
                prob = PatchMatch.Problem(ref, list(src), images)
                if options.geom_consistency:
                    prob.depth_maps = [maps[k][0] if k in maps else None for k in range(len(images))]
                    prob.normal_maps = [maps[k][1] if k in maps else None for k in range(len(images))]
                pms.append(PatchMatch(o, prob, self.image_cache_))
            run_batch(pms)
            for (ref, src), pm in zip(todo[i:j], pms):
                depth, normal = pm.GetDepthMap(), pm.GetNormalMap()
                dpath, npath = self._paths(ref, output_type)
                os.makedirs(os.path.dirname(dpath), exist_ok=True)
                os.makedirs(os.path.dirname(npath), exist_ok=True)
                write_mat(dpath, depth)
                write_mat(npath, normal)
                if options.write_consistency_graph:  # (:532-534)
                    from . import workspace as W
                    gpath = os.path.join(os.path.dirname(os.path.dirname(dpath)), "consistency_graphs",
                                         os.path.basename(dpath))
                    os.makedirs(os.path.dirname(gpath), exist_ok=True)
                    W.write_consistency_graph(gpath, depth.shape[1], depth.shape[0], pm.GetConsistentImageIdxs())
                results[ref] = (depth, normal)
                if device_maps is not None:
                    device_maps[ref] = pm.GetDeviceMaps()
                pm.close()
            i = j
        return results

    def Run(self):
        import time
        from . import distributed as D
        import torch
        opt = self.options_
        maps = None
        if opt.geom_consistency:
            photo = PatchMatchOptions(**{**opt.__dict__})
            photo.geom_consistency = False
            photo.filter = False
            t = time.time()
            # the photometric maps stay in HBM: device-to-device copy out of every handle, all-gather
            # over RCCL between the ranks (every rank needs the maps of its problems' sources), and the
            # geometric pass reads them as device inputs (the files written on the way are the
            # workspace output the reference also produces, nothing reads them back)
            dev_maps: dict = {}
            self._run_pass(photo, None, device_maps=dev_maps)
            merged = D.exchange_maps_device(dev_maps, self._device(opt))
            maps = {k: (v[0], v[1:]) for k, v in merged.items()}
            self.timings["photometric_s"] = time.time() - t
            self.timings["map_exchange"] = D.last_exchange_info()
        t = time.time()
        out = self._run_pass(opt, maps)
        self.timings["geometric_s" if opt.geom_consistency else "photometric_s"] = time.time() - t
        # The pooled device buffers (and the packed-image slabs) served both passes; the controller is done with
        # PatchMatch now, so they go back to the driver -- torch allocations of the same process (this controller's own
        # device bitmaps, whatever runs next) and other processes on the GPU would otherwise meet a plain OOM, since only
        # the library's own allocators retry after releasing the pool. Slabs with a live slot stay.
        release_cached_memory()
        return out
