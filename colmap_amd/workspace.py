"""The data formats either side of the PatchMatch path: COLMAP sparse models, the dense workspace
layout, `patch-match.cfg`, and the MVS-side model statistics the controller needs.

Mirrors (reference file:line in each docstring):
  scene/reconstruction_io_binary.cc / reconstruction_io_text.cc   cameras / images / points3D files
  mvs/model.cc:57-330                                             mvs::Model (depth ranges, shared points,
                                                                  triangulation angles, overlapping images)
  mvs/patch_match.cc:239-359                                      patch-match.cfg -> problems
  mvs/workspace.cc:38-141                                         workspace paths and bitmap loading
  mvs/consistency_graph.cc:69-139                                 consistency graph files
Host-side I/O only: nothing here touches the GPU.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# colmap::CameraModelId -> (name, number of parameters, fx index, fy index, cx index, cy index)
# (sensor/models.h:90-111 and the FocalLengthIdxs / PrincipalPointIdxs of each model)
CAMERA_MODELS = {
    0: ("SIMPLE_PINHOLE", 3, 0, 0, 1, 2),
    1: ("PINHOLE", 4, 0, 1, 2, 3),
    2: ("SIMPLE_RADIAL", 4, 0, 0, 1, 2),
    3: ("RADIAL", 5, 0, 0, 1, 2),
    4: ("OPENCV", 8, 0, 1, 2, 3),
    5: ("OPENCV_FISHEYE", 8, 0, 1, 2, 3),
    6: ("FULL_OPENCV", 12, 0, 1, 2, 3),
    7: ("FOV", 5, 0, 1, 2, 3),
    8: ("SIMPLE_RADIAL_FISHEYE", 4, 0, 0, 1, 2),
    9: ("RADIAL_FISHEYE", 5, 0, 0, 1, 2),
    10: ("THIN_PRISM_FISHEYE", 12, 0, 1, 2, 3),
    11: ("RAD_TAN_THIN_PRISM_FISHEYE", 16, 0, 1, 2, 3),
    12: ("SIMPLE_DIVISION", 4, 0, 0, 1, 2),
    13: ("DIVISION", 5, 0, 1, 2, 3),
    14: ("SIMPLE_FISHEYE", 3, 0, 0, 1, 2),
    15: ("FISHEYE", 4, 0, 1, 2, 3),
    16: ("EUCM", 6, 0, 1, 2, 3),
    17: ("EQUIRECTANGULAR", 2, None, None, None, None),  # width, height: no focal length / principal point
}
CAMERA_MODEL_IDS = {v[0]: k for k, v in CAMERA_MODELS.items()}
INVALID_POINT3D = 0xFFFFFFFFFFFFFFFF  # kInvalidPoint3DId (util/types.h)


@dataclass
class SparseCamera:
    camera_id: int
    model_id: int
    width: int
    height: int
    params: np.ndarray

    def CalibrationMatrix(self) -> np.ndarray:
        """Camera::CalibrationMatrix (scene/camera.cc:63-72)."""
        _, _, ifx, ify, icx, icy = CAMERA_MODELS[self.model_id]
        if ifx is None:
            raise ValueError(f"{CAMERA_MODELS[self.model_id][0]} cameras have no calibration matrix")
        K = np.eye(3)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = self.params[ifx], self.params[ify], self.params[icx], self.params[icy]
        return K


@dataclass
class SparseImage:
    image_id: int
    qvec: np.ndarray  # w x y z (file order)
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    point3D_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))  # -1 = none

    def RotationMatrix(self) -> np.ndarray:
        w, x, y, z = self.qvec
        return np.array([
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@dataclass
class SparsePoint3D:
    point3D_id: int
    xyz: np.ndarray
    rgb: Tuple[int, int, int] = (0, 0, 0)
    error: float = 0.0
    track: List[Tuple[int, int]] = field(default_factory=list)  # (image_id, point2D_idx)


@dataclass
class SparseRig:
    """scene/rig.h: reference sensor + (sensor id -> optional sensor_from_rig as w x y z tx ty tz);
    a sensor id is (type, id) with type 0 = CAMERA, 1 = IMU (util/types.h:144-148)."""
    rig_id: int
    ref_sensor: Optional[Tuple[int, int]] = None
    sensors: Dict[Tuple[int, int], Optional[np.ndarray]] = field(default_factory=dict)


@dataclass
class SparseFrame:
    """scene/frame.h: rig_from_world (w x y z tx ty tz) and the data ids (sensor type, sensor id, data id)."""
    frame_id: int
    rig_id: int
    rig_from_world: np.ndarray
    data_ids: List[Tuple[int, int, int]] = field(default_factory=list)


@dataclass
class SparseModel:
    cameras: Dict[int, SparseCamera] = field(default_factory=dict)
    images: Dict[int, SparseImage] = field(default_factory=dict)
    points3D: Dict[int, SparsePoint3D] = field(default_factory=dict)
    rigs: Dict[int, SparseRig] = field(default_factory=dict)      # empty: legacy model, one rig per camera
    frames: Dict[int, SparseFrame] = field(default_factory=dict)  # empty: legacy model, one frame per image


# ------------------------------------------------------------------------------------------------
# binary files (scene/reconstruction_io_binary.cc:107-131,173-291)
# ------------------------------------------------------------------------------------------------

def _rd(f, fmt):
    data = f.read(struct.calcsize(fmt))
    if len(data) != struct.calcsize(fmt):
        raise IOError("unexpected end of file")
    return struct.unpack("<" + fmt, data)


def read_cameras_binary(path: str) -> Dict[int, SparseCamera]:
    cams = {}
    with open(path, "rb") as f:
        (n,) = _rd(f, "Q")
        for _ in range(n):
            cid, model, w, h = _rd(f, "IiQQ")
            if model not in CAMERA_MODELS:
                raise ValueError(f"unknown camera model id {model}")
            params = np.array(_rd(f, "d" * CAMERA_MODELS[model][1]))
            cams[cid] = SparseCamera(cid, model, w, h, params)
    return cams


def read_images_binary(path: str) -> Dict[int, SparseImage]:
    imgs = {}
    with open(path, "rb") as f:
        (n,) = _rd(f, "Q")
        for _ in range(n):
            (iid,) = _rd(f, "I")
            q = np.array(_rd(f, "dddd"))
            t = np.array(_rd(f, "ddd"))
            (cid,) = _rd(f, "I")
            name = bytearray()
            while True:
                c = f.read(1)
                if c == b"\0" or c == b"":
                    break
                name += c
            (m,) = _rd(f, "Q")
            raw = np.frombuffer(f.read(24 * m), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<u8")]))
            xys = np.stack([raw["x"], raw["y"]], 1) if m else np.zeros((0, 2))
            ids = raw["id"].astype(np.int64) if m else np.zeros(0, np.int64)  # 2^64-1 -> -1
            imgs[iid] = SparseImage(iid, q, t, cid, name.decode(), xys, ids)
    return imgs


def read_points3D_binary(path: str) -> Dict[int, SparsePoint3D]:
    pts = {}
    with open(path, "rb") as f:
        (n,) = _rd(f, "Q")
        for _ in range(n):
            (pid,) = _rd(f, "Q")
            xyz = np.array(_rd(f, "ddd"))
            rgb = _rd(f, "BBB")
            (err,) = _rd(f, "d")
            (tl,) = _rd(f, "Q")
            tr = np.frombuffer(f.read(8 * tl), dtype="<u4").reshape(-1, 2)
            pts[pid] = SparsePoint3D(pid, xyz, rgb, err, [(int(a), int(b)) for a, b in tr])
    return pts


def read_rigs_binary(path: str) -> Dict[int, SparseRig]:
    """ReadRigsBinary (reconstruction_io_binary.cc:47-98)."""
    rigs = {}
    with open(path, "rb") as f:
        (n,) = _rd(f, "Q")
        for _ in range(n):
            rid, ns = _rd(f, "II")
            rig = SparseRig(rid)
            if ns > 0:
                rig.ref_sensor = tuple(_rd(f, "iI"))
            for _ in range(max(0, ns - 1)):
                sid = tuple(_rd(f, "iI"))
                (has_pose,) = _rd(f, "B")
                rig.sensors[sid] = np.array(_rd(f, "ddddddd")) if has_pose else None
            rigs[rid] = rig
    return rigs


def read_frames_binary(path: str) -> Dict[int, SparseFrame]:
    """ReadFramesBinary (reconstruction_io_binary.cc:132-164)."""
    frames = {}
    with open(path, "rb") as f:
        (n,) = _rd(f, "Q")
        for _ in range(n):
            fid, rid = _rd(f, "II")
            pose = np.array(_rd(f, "ddddddd"))
            (nd,) = _rd(f, "I")
            frames[fid] = SparseFrame(fid, rid, pose, [tuple(_rd(f, "iIQ")) for _ in range(nd)])
    return frames


def write_rigs_frames_binary(model: SparseModel, path: str):
    """WriteRigsBinary / WriteFramesBinary (reconstruction_io_binary.cc:293-399)."""
    with open(os.path.join(path, "rigs.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(model.rigs)))
        for rid in sorted(model.rigs):
            rig = model.rigs[rid]
            ns = (1 if rig.ref_sensor is not None else 0) + len(rig.sensors)
            f.write(struct.pack("<II", rid, ns))
            if rig.ref_sensor is not None:
                f.write(struct.pack("<iI", *rig.ref_sensor))
            for sid in sorted(rig.sensors):
                pose = rig.sensors[sid]
                f.write(struct.pack("<iIB", sid[0], sid[1], 0 if pose is None else 1))
                if pose is not None:
                    f.write(np.asarray(pose, "<f8").tobytes())
    with open(os.path.join(path, "frames.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(model.frames)))
        for fid in sorted(model.frames):
            fr = model.frames[fid]
            f.write(struct.pack("<II", fid, fr.rig_id) + np.asarray(fr.rig_from_world, "<f8").tobytes())
            f.write(struct.pack("<I", len(fr.data_ids)))
            for d in sorted(fr.data_ids):
                f.write(struct.pack("<iIQ", *d))


def write_model_binary(model: SparseModel, path: str):
    """WriteCamerasBinary / WriteImagesBinary / WritePoints3DBinary (sorted ids like the reference);
    rigs.bin / frames.bin too when the model carries rigs."""
    os.makedirs(path, exist_ok=True)
    if model.rigs or model.frames:
        write_rigs_frames_binary(model, path)
    with open(os.path.join(path, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(model.cameras)))
        for cid in sorted(model.cameras):
            c = model.cameras[cid]
            f.write(struct.pack("<IiQQ", c.camera_id, c.model_id, c.width, c.height))
            f.write(np.asarray(c.params, "<f8").tobytes())
    with open(os.path.join(path, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(model.images)))
        for iid in sorted(model.images):
            im = model.images[iid]
            f.write(struct.pack("<I", im.image_id))
            f.write(np.asarray(im.qvec, "<f8").tobytes() + np.asarray(im.tvec, "<f8").tobytes())
            f.write(struct.pack("<I", im.camera_id))
            f.write(im.name.encode() + b"\0")
            f.write(struct.pack("<Q", len(im.xys)))
            rec = np.zeros(len(im.xys), np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<u8")]))
            if len(im.xys):
                rec["x"], rec["y"] = im.xys[:, 0], im.xys[:, 1]
                rec["id"] = np.asarray(im.point3D_ids, np.int64).astype(np.uint64)
            f.write(rec.tobytes())
    with open(os.path.join(path, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(model.points3D)))
        for pid in sorted(model.points3D):
            p = model.points3D[pid]
            f.write(struct.pack("<Q", pid) + np.asarray(p.xyz, "<f8").tobytes())
            f.write(struct.pack("<BBBd", *p.rgb, p.error))
            f.write(struct.pack("<Q", len(p.track)))
            f.write(np.asarray(p.track, "<u4").reshape(-1, 2).tobytes())


# ------------------------------------------------------------------------------------------------
# text files (scene/reconstruction_io_text.cc)
# ------------------------------------------------------------------------------------------------

def _text_lines(path):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_text(path: str) -> Dict[int, SparseCamera]:
    cams = {}
    for line in _text_lines(path):
        t = line.split()
        model = CAMERA_MODEL_IDS[t[1]]
        cams[int(t[0])] = SparseCamera(int(t[0]), model, int(t[2]), int(t[3]), np.array(t[4:], float))
    return cams


def read_images_text(path: str) -> Dict[int, SparseImage]:
    imgs = {}
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f if not l.startswith("#")]
    # two lines per image; the second may be empty (no observations)
    while lines and not lines[-1].strip():
        lines.pop()
    i = 0
    while i < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        t = lines[i].split()
        iid, q, tv, cid, name = int(t[0]), np.array(t[1:5], float), np.array(t[5:8], float), int(t[8]), " ".join(t[9:])
        p = lines[i + 1].split() if i + 1 < len(lines) else []
        arr = np.array(p, float).reshape(-1, 3) if p else np.zeros((0, 3))
        imgs[iid] = SparseImage(iid, q, tv, cid, name, arr[:, :2].copy(), arr[:, 2].astype(np.int64))
        i += 2
    return imgs


def read_points3D_text(path: str) -> Dict[int, SparsePoint3D]:
    pts = {}
    for line in _text_lines(path):
        t = line.split()
        tr = np.array(t[8:], np.int64).reshape(-1, 2)
        pts[int(t[0])] = SparsePoint3D(int(t[0]), np.array(t[1:4], float), tuple(int(v) for v in t[4:7]),
                                       float(t[7]), [(int(a), int(b)) for a, b in tr])
    return pts


_SENSOR_TYPE_NAMES = {-1: "INVALID", 0: "CAMERA", 1: "IMU"}  # SensorType (util/types.h:144-148), text spelling
_SENSOR_TYPE_IDS = {v: k for k, v in _SENSOR_TYPE_NAMES.items()}


def read_rigs_text(path: str) -> Dict[int, SparseRig]:
    """ReadRigsText (reconstruction_io_text.cc:47-108): RIG_ID NUM_SENSORS [REF_TYPE REF_ID] then per other
    sensor TYPE ID HAS_POSE [QW QX QY QZ TX TY TZ]."""
    rigs = {}
    for line in _text_lines(path):
        t = line.split()
        rig = SparseRig(int(t[0]))
        n, i = int(t[1]), 2
        if n > 0:
            rig.ref_sensor = (_SENSOR_TYPE_IDS[t[i]], int(t[i + 1]))
            i += 2
        for _ in range(max(0, n - 1)):
            sid = (_SENSOR_TYPE_IDS[t[i]], int(t[i + 1]))
            has_pose = int(t[i + 2]) == 1
            i += 3
            if has_pose:
                rig.sensors[sid] = np.array(t[i:i + 7], float)
                i += 7
            else:
                rig.sensors[sid] = None
        rigs[rig.rig_id] = rig
    return rigs


def read_frames_text(path: str) -> Dict[int, SparseFrame]:
    """ReadFramesText (reconstruction_io_text.cc:160-205): FRAME_ID RIG_ID QW QX QY QZ TX TY TZ NUM_DATA_IDS then
    (SENSOR_TYPE SENSOR_ID DATA_ID) per data id."""
    frames = {}
    for line in _text_lines(path):
        t = line.split()
        fid, rid, nd = int(t[0]), int(t[1]), int(t[9])
        data = [(_SENSOR_TYPE_IDS[t[10 + 3 * k]], int(t[11 + 3 * k]), int(t[12 + 3 * k])) for k in range(nd)]
        frames[fid] = SparseFrame(fid, rid, np.array(t[2:9], float), data)
    return frames


def write_rigs_frames_text(model: SparseModel, path: str):
    """WriteRigsText / WriteFramesText (reconstruction_io_text.cc:370-517): 17 significant digits."""
    r = lambda v: repr(float(v))
    with open(os.path.join(path, "rigs.txt"), "w") as f:
        f.write("# Rig calib list with one line of data per calib:\n"
                "#   RIG_ID, NUM_SENSORS, REF_SENSOR_TYPE, REF_SENSOR_ID, SENSORS[] as (SENSOR_TYPE, SENSOR_ID, HAS_POSE, "
                "[QW, QX, QY, QZ, TX, TY, TZ])\n")
        f.write(f"# Number of rigs: {len(model.rigs)}\n")
        for rid in sorted(model.rigs):
            rig = model.rigs[rid]
            ns = (1 if rig.ref_sensor is not None else 0) + len(rig.sensors)
            parts = [str(rid), str(ns)]
            if rig.ref_sensor is not None:
                parts += [_SENSOR_TYPE_NAMES[rig.ref_sensor[0]], str(rig.ref_sensor[1])]
            for sid in sorted(rig.sensors):
                pose = rig.sensors[sid]
                parts += [_SENSOR_TYPE_NAMES[sid[0]], str(sid[1]), "0" if pose is None else "1"]
                if pose is not None:
                    parts += [r(v) for v in pose]
            f.write(" ".join(parts) + "\n")
    with open(os.path.join(path, "frames.txt"), "w") as f:
        f.write("# Frame list with one line of data per frame:\n"
                "#   FRAME_ID, RIG_ID, RIG_FROM_WORLD[QW, QX, QY, QZ, TX, TY, TZ], NUM_DATA_IDS, DATA_IDS[] as "
                "(SENSOR_TYPE, SENSOR_ID, DATA_ID)\n")
        f.write(f"# Number of frames: {len(model.frames)}\n")
        for fid in sorted(model.frames):
            fr = model.frames[fid]
            parts = [str(fid), str(fr.rig_id)] + [r(v) for v in fr.rig_from_world] + [str(len(fr.data_ids))]
            for d in sorted(fr.data_ids):
                parts += [_SENSOR_TYPE_NAMES[d[0]], str(d[1]), str(d[2])]
            f.write(" ".join(parts) + "\n")


def write_model_text(model: SparseModel, path: str):
    os.makedirs(path, exist_ok=True)
    if model.rigs or model.frames:  # Reconstruction::WriteText carries rigs.txt / frames.txt too
        write_rigs_frames_text(model, path)
    with open(os.path.join(path, "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        for cid in sorted(model.cameras):
            c = model.cameras[cid]
            f.write(f"{cid} {CAMERA_MODELS[c.model_id][0]} {c.width} {c.height} " +
                    " ".join(repr(float(v)) for v in c.params) + "\n")
    with open(os.path.join(path, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n")
        for iid in sorted(model.images):
            im = model.images[iid]
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in list(im.qvec) + list(im.tvec)) +
                    f" {im.camera_id} {im.name}\n")
            f.write(" ".join(f"{repr(float(x))} {repr(float(y))} {int(p)}" for (x, y), p in zip(im.xys, im.point3D_ids)) + "\n")
    with open(os.path.join(path, "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n")
        for pid in sorted(model.points3D):
            p = model.points3D[pid]
            f.write(f"{pid} " + " ".join(repr(float(v)) for v in p.xyz) + f" {p.rgb[0]} {p.rgb[1]} {p.rgb[2]} "
                    f"{repr(float(p.error))} " + " ".join(f"{a} {b}" for a, b in p.track) + "\n")


def read_sparse_model(path: str) -> SparseModel:
    """Reconstruction::Read: binary files if present, else text (scene/reconstruction.cc)."""
    if os.path.exists(os.path.join(path, "cameras.bin")):
        m = SparseModel(read_cameras_binary(os.path.join(path, "cameras.bin")),
                        read_images_binary(os.path.join(path, "images.bin")),
                        read_points3D_binary(os.path.join(path, "points3D.bin")))
        # newer models also carry rigs.bin / frames.bin (Reconstruction::ReadBinary); legacy ones do not
        if os.path.exists(os.path.join(path, "rigs.bin")) and os.path.exists(os.path.join(path, "frames.bin")):
            m.rigs = read_rigs_binary(os.path.join(path, "rigs.bin"))
            m.frames = read_frames_binary(os.path.join(path, "frames.bin"))
        return m
    if os.path.exists(os.path.join(path, "cameras.txt")):
        m = SparseModel(read_cameras_text(os.path.join(path, "cameras.txt")),
                        read_images_text(os.path.join(path, "images.txt")),
                        read_points3D_text(os.path.join(path, "points3D.txt")))
        # Reconstruction::ReadText: rigs.txt / frames.txt when present (legacy text models have neither)
        if os.path.exists(os.path.join(path, "rigs.txt")) and os.path.exists(os.path.join(path, "frames.txt")):
            m.rigs = read_rigs_text(os.path.join(path, "rigs.txt"))
            m.frames = read_frames_text(os.path.join(path, "frames.txt"))
        return m
    raise FileNotFoundError(f"cameras, images, points3D files do not exist at {path}")


# ------------------------------------------------------------------------------------------------
# mvs::Model (mvs/model.cc)
# ------------------------------------------------------------------------------------------------

@dataclass
class ModelImage:
    path: str
    width: int
    height: int
    K: np.ndarray  # (3,3) float32
    R: np.ndarray  # (3,3) float32
    T: np.ndarray  # (3,) float32


@dataclass
class ModelPoint:
    x: float
    y: float
    z: float
    track: List[int]  # image indices


def percentile(values: Sequence[float], p: float) -> float:
    """colmap::Percentile (math/math.h:205-224): linear interpolation between order statistics."""
    v = np.sort(np.asarray(values, np.float64))
    idx = p / 100.0 * (len(v) - 1)
    lo, hi = int(np.floor(idx)), int(np.ceil(idx))
    if lo == hi:
        return float(v[hi])
    return float((hi - idx) * v[lo] + (idx - lo) * v[hi])


class Model:
    """mvs::Model (mvs/model.h): images in RegImageIds order, points with tracks of image indices."""

    def __init__(self):
        self.images: List[ModelImage] = []
        self.points: List[ModelPoint] = []
        self.image_names_: List[str] = []
        self.image_name_to_idx_: Dict[str, int] = {}

    @staticmethod
    def ReadFromCOLMAP(path: str, sparse_path: str = "sparse", images_path: str = "images") -> "Model":
        """Model::ReadFromCOLMAP (model.cc:57-98)."""
        sm = read_sparse_model(os.path.join(path, sparse_path))
        return Model.FromSparseModel(sm, os.path.join(path, images_path))

    @staticmethod
    def FromSparseModel(sm: SparseModel, images_dir: str) -> "Model":
        m = Model()
        id_to_idx = {}
        for idx, iid in enumerate(sorted(sm.images)):
            im = sm.images[iid]
            cam = sm.cameras[im.camera_id]
            m.images.append(ModelImage(os.path.join(images_dir, im.name), cam.width, cam.height,
                                       cam.CalibrationMatrix().astype(np.float32),
                                       im.RotationMatrix().astype(np.float32), im.tvec.astype(np.float32)))
            id_to_idx[iid] = idx
            m.image_names_.append(im.name)
            m.image_name_to_idx_[im.name] = idx
        for pid in sorted(sm.points3D):
            p = sm.points3D[pid]
            m.points.append(ModelPoint(float(np.float32(p.xyz[0])), float(np.float32(p.xyz[1])),
                                       float(np.float32(p.xyz[2])), [id_to_idx[i] for i, _ in p.track]))
        return m

    # -- PMVS workspaces (mvs/model.cc:100-104,282-454) ---------------------------------------------
    @staticmethod
    def ReadFromPMVS(path: str) -> "Model":
        """Model::ReadFromPMVS: a Bundler export (`bundle.rd.out`) or a raw PMVS folder (`vis.dat`,
        `txt/%08d.txt` projection matrices); images are `visualize/%08d.jpg` in both."""
        m = Model()
        m.pmvs_vis_dat_ = []
        if m._read_from_bundler_pmvs(path) or m._read_from_raw_pmvs(path):
            return m
        raise ValueError("Invalid PMVS format")

    def GetMaxOverlappingImagesFromPMVS(self) -> List[List[int]]:
        return getattr(self, "pmvs_vis_dat_", [])

    def _add_pmvs_image(self, path, idx, K, R, T):
        from PIL import Image as PILImage
        name = f"{idx:08d}.jpg"
        image_path = os.path.join(path, "visualize", name)
        with PILImage.open(image_path) as b:
            w, h = b.size
        self.images.append(ModelImage(image_path, w, h, np.asarray(K, np.float32).reshape(3, 3),
                                      np.asarray(R, np.float32).reshape(3, 3), np.asarray(T, np.float32).reshape(3)))
        self.image_names_.append(name)
        self.image_name_to_idx_[name] = idx
        return w, h

    def _read_from_bundler_pmvs(self, path: str) -> bool:
        """Model::ReadFromBundlerPMVS (model.cc:282-356)."""
        bundle = os.path.join(path, "bundle.rd.out")
        if not os.path.exists(bundle):
            return False
        with open(bundle) as f:
            f.readline()  # header
            tok = f.read().split()
        it = iter(tok)
        num_images, num_points = int(next(it)), int(next(it))
        from PIL import Image as PILImage
        for idx in range(num_images):
            fl = np.float32(next(it))
            k1, k2 = np.float32(next(it)), np.float32(next(it))
            if k1 != 0 or k2 != 0:
                raise ValueError("Check failed: k1 == 0 && k2 == 0 (undistorted images expected)")
            R = np.array([np.float32(next(it)) for _ in range(9)], np.float32)
            R[3:] = -R[3:]
            T = np.array([np.float32(next(it)) for _ in range(3)], np.float32)
            T[1:] = -T[1:]
            with PILImage.open(os.path.join(path, "visualize", f"{idx:08d}.jpg")) as b:
                w, h = b.size
            K = np.array([fl, 0, w / 2.0, 0, fl, h / 2.0, 0, 0, 1], np.float32)
            self._add_pmvs_image(path, idx, K, R, T)
        for _ in range(num_points):
            x, y, z = float(np.float32(next(it))), float(np.float32(next(it))), float(np.float32(next(it)))
            for _c in range(3):
                next(it)
            n = int(next(it))
            track = []
            for _k in range(n):
                track.append(int(next(it)))
                next(it); next(it); next(it)
                if not track[-1] < len(self.images):
                    raise ValueError("Check failed: point.track[i] < images.size()")
            self.points.append(ModelPoint(x, y, z, track))
        return True

    def _read_from_raw_pmvs(self, path: str) -> bool:
        """Model::ReadFromRawPMVS (model.cc:358-454): P = K [R | T] decomposed by RQ
        (DecomposeProjectionMatrix, geometry/pose.cc:98-130), skew dropped."""
        vis = os.path.join(path, "vis.dat")
        if not os.path.exists(vis):
            return False
        import scipy.linalg
        idx = 0
        while os.path.exists(os.path.join(path, "visualize", f"{idx:08d}.jpg")):
            with open(os.path.join(path, "txt", f"{idx:08d}.txt")) as f:
                tok = f.read().split()
            if tok[0] != "CONTOUR":
                raise ValueError("Check failed: contour == CONTOUR")
            P = np.array(tok[1:13], np.float64).reshape(3, 4)
            RR, QQ = scipy.linalg.rq(P[:, :3])
            if np.linalg.det(QQ) < 0:  # DecomposeMatrixRQ makes the factorisation unique (math/matrix.h:69-73)
                QQ[1, :] *= -1.0
                RR[:, 1] *= -1.0
            U, _, Vt = np.linalg.svd(QQ)  # ComputeClosestRotationMatrix (geometry/pose.cc:88-96)
            R = U @ Vt
            if np.linalg.det(R) < 0:
                R = -R
            det_k = np.linalg.det(RR)
            if det_k == 0:
                raise ValueError("degenerate projection matrix")
            K = RR if det_k > 0 else -RR
            for i in range(3):
                if K[i, i] < 0:
                    K[:, i] = -K[:, i]
                    R[i, :] = -R[i, :]
            T = np.linalg.solve(np.triu(K), P[:, 3])
            if det_k < 0:
                T = -T
            K[0, 1] = K[1, 0] = K[2, 0] = K[2, 1] = 0.0
            K[2, 2] = 1.0
            self._add_pmvs_image(path, idx, K, R, T)
            idx += 1
        with open(vis) as f:
            tok = f.read().split()
        if tok[0] != "VISDATA":
            raise ValueError("Check failed: visdata == VISDATA")
        n = int(tok[1])
        if n != len(self.images):
            raise ValueError("Check failed: num_images == images.size()")
        self.pmvs_vis_dat_ = [[] for _ in range(n)]
        k = 2
        for _ in range(n):
            image_idx, m = int(tok[k]), int(tok[k + 1])
            k += 2
            for j in range(m):
                v = int(tok[k + j])
                if not (0 <= v < n):
                    raise ValueError("Check failed: visible_image_idx < num_images")
                if v != image_idx:
                    self.pmvs_vis_dat_[image_idx].append(v)
            k += m
        return True

    def GetImageIdx(self, name: str) -> int:
        if name not in self.image_name_to_idx_:
            raise KeyError(f"Image with name `{name}` does not exist")  # (:106-110)
        return self.image_name_to_idx_[name]

    def GetImageName(self, image_idx: int) -> str:
        return self.image_names_[image_idx]

    def ComputeDepthRanges(self) -> List[Tuple[float, float]]:
        """Model::ComputeDepthRanges (model.cc:178-218): 1st / 99th percentile element of the
        positive depths of the image's sparse points, stretched by 25 %; (-1, -1) without points."""
        depths: List[List[np.float32]] = [[] for _ in self.images]
        for pt in self.points:
            X = np.array([pt.x, pt.y, pt.z], np.float32)
            for idx in pt.track:
                im = self.images[idx]
                d = np.float32(np.dot(im.R[2], X)) + im.T[2]
                if d > 0:
                    depths[idx].append(np.float32(d))
        out = []
        for ds in depths:
            if not ds:
                out.append((-1.0, -1.0))
                continue
            v = np.sort(np.array(ds, np.float32))
            lo = v[int(len(v) * np.float32(0.01))]
            hi = v[int(len(v) * np.float32(0.99))]
            out.append((float(np.float32(lo) * np.float32(0.75)), float(np.float32(hi) * np.float32(1.25))))
        return out

    def ComputeSharedPoints(self) -> List[Dict[int, int]]:
        """Model::ComputeSharedPoints (model.cc:220-235)."""
        shared: List[Dict[int, int]] = [dict() for _ in self.images]
        for pt in self.points:
            tr = pt.track
            for i in range(len(tr)):
                for j in range(i):
                    a, b = tr[i], tr[j]
                    if a != b:
                        shared[a][b] = shared[a].get(b, 0) + 1
                        shared[b][a] = shared[b].get(a, 0) + 1
        return shared

    def ComputeTriangulationAngles(self, pct: float = 75.0) -> List[Dict[int, float]]:
        """Model::ComputeTriangulationAngles (model.cc:237-280): per image pair the given percentile
        of min(angle, pi - angle) between the viewing rays of the shared points."""
        centers = [(-im.R.T @ im.T).astype(np.float64) for im in self.images]  # ComputeProjectionCenter, float
        allang: List[Dict[int, List[float]]] = [dict() for _ in self.images]
        for pt in self.points:
            X = np.array([pt.x, pt.y, pt.z], np.float64)
            tr = pt.track
            for i in range(len(tr)):
                for j in range(i):
                    a, b = tr[i], tr[j]
                    if a == b:
                        continue
                    v1, v2 = X - centers[a], X - centers[b]
                    n1, n2 = v1 @ v1, v2 @ v2
                    ang = 0.0 if n1 == 0 or n2 == 0 else float(np.arccos(np.clip(v1 @ v2 / np.sqrt(n1 * n2), -1, 1)))
                    ang = np.float32(min(ang, np.pi - ang))
                    allang[a].setdefault(b, []).append(ang)
                    allang[b].setdefault(a, []).append(ang)
        return [{o: np.float32(percentile(v, pct)) for o, v in d.items()} for d in allang]

    def GetMaxOverlappingImages(self, num_images: int, min_triangulation_angle: float) -> List[List[int]]:
        """Model::GetMaxOverlappingImages (model.cc:118-170)."""
        min_rad = np.float32(np.deg2rad(min_triangulation_angle))
        shared = self.ComputeSharedPoints()
        angles = self.ComputeTriangulationAngles(75.0)
        out = []
        for idx in range(len(self.images)):
            cand = [(o, c) for o, c in sorted(shared[idx].items()) if angles[idx][o] >= min_rad]
            cand.sort(key=lambda t: -t[1])  # stable: ties keep ascending image index
            out.append([o for o, _ in cand[:num_images]])
        return out


# ------------------------------------------------------------------------------------------------
# patch-match.cfg (mvs/patch_match.cc:239-359)
# ------------------------------------------------------------------------------------------------

def read_patch_match_config(lines: Sequence[str], model: Model, min_triangulation_angle: float = 1.0,
                            warn=None) -> List[Tuple[int, List[int]]]:
    """PatchMatchController::ReadProblems: pairs of lines (reference image name, source spec);
    the spec is `__all__`, `__auto__, N` (N most-overlapping images whose 75th-percentile
    triangulation angle reaches PatchMatchOptions::min_triangulation_angle) or a comma-separated
    list of image names. Reference images without sources are dropped with a warning."""
    configs = []
    ref_name = ""
    for raw in lines:
        line = raw.strip()
        if not line or line[0] == "#":
            continue
        if not ref_name:
            ref_name = line
            continue
        configs.append((ref_name, [t.strip() for t in line.split(",") if t.strip()]))
        ref_name = ""
    shared = angles = None
    min_rad = np.float32(np.deg2rad(min_triangulation_angle))
    problems = []
    for ref_name, srcs in configs:
        ref = model.GetImageIdx(ref_name)
        if len(srcs) == 1 and srcs[0] == "__all__":
            src = [i for i in range(len(model.images)) if i != ref]
        elif len(srcs) == 2 and srcs[0] == "__auto__":
            if shared is None:
                shared = model.ComputeSharedPoints()
                angles = model.ComputeTriangulationAngles(75.0)
            max_num = int(srcs[1])
            cand = [(o, c) for o, c in sorted(shared[ref].items()) if angles[ref][o] >= min_rad]
            cand.sort(key=lambda t: -t[1])
            src = [o for o, _ in cand[:max_num]]
        else:
            src = [model.GetImageIdx(n) for n in srcs]
        if not src:
            if warn:
                warn(f"Ignoring reference image {ref_name}, because it has no source images.")
            continue
        problems.append((ref, src))
    return problems


def write_patch_match_config(path: str, names: Sequence[str], spec: str = "__auto__, 20"):
    """What `image_undistorter` writes (mvs/workspace.cc:296-324 / image/undistortion.cc)."""
    with open(path, "w") as f:
        for n in names:
            f.write(f"{n}\n{spec}\n")


# ------------------------------------------------------------------------------------------------
# bitmaps and the workspace layout (mvs/workspace.cc, sensor/bitmap.cc)
# ------------------------------------------------------------------------------------------------

def read_bitmap_grey(path: str) -> np.ndarray:
    """Bitmap::Read(path, as_rgb=false): 8-bit grey; colour images are converted with
    round(.2126 R + .7152 G + .0722 B) in float like Bitmap::CloneAsGrey (sensor/bitmap.cc:596-618)."""
    from PIL import Image as PILImage
    with PILImage.open(path) as im:
        if im.mode in ("L", "1", "LA"):  # already grey (+ alpha dropped, bitmap.cc:422)
            if im.mode == "LA":
                im = im.split()[0]
            return np.ascontiguousarray(np.asarray(im.convert("L"), np.uint8))
        rgb = np.asarray(im.convert("RGB"), np.float32)
    grey = np.float32(.2126) * rgb[..., 0] + np.float32(.7152) * rgb[..., 1] + np.float32(.0722) * rgb[..., 2]
    return np.ascontiguousarray(np.floor(grey + np.float32(0.5)).astype(np.uint8))  # std::round, values >= 0


class Workspace:
    """mvs::Workspace (mvs/workspace.h:46-104): the undistorted dense workspace
    `<path>/{images,sparse,stereo/{depth_maps,normal_maps,consistency_graphs,patch-match.cfg}}`."""

    def __init__(self, workspace_path: str, workspace_format: str = "COLMAP", stereo_folder: str = "stereo",
                 input_type: str = "", max_image_size: int = -1):
        fmt = workspace_format.lower()
        if fmt not in ("colmap", "pmvs"):
            raise ValueError("Invalid input format")  # Model::Read (model.cc:45-55)
        self.workspace_path = workspace_path
        self.stereo_folder = stereo_folder
        self.input_type = input_type
        self.max_image_size = max_image_size
        self.model = Model.ReadFromCOLMAP(workspace_path) if fmt == "colmap" else Model.ReadFromPMVS(workspace_path)
        if max_image_size > 0:
            for im in self.model.images:  # mvs::Image::Downsize (image.cc:75-95) via Workspace ctor (:44-48)
                _downsize(im, max_image_size, max_image_size)
        self._bitmaps: Dict[int, np.ndarray] = {}

    def GetModel(self) -> Model:
        return self.model

    def GetFileName(self, image_idx: int, output_type: Optional[str] = None) -> str:
        t = self.input_type if output_type is None else output_type
        return f"{self.model.GetImageName(image_idx)}.{t}.bin"  # (:56-60)

    def GetBitmapPath(self, image_idx: int) -> str:
        return self.model.images[image_idx].path

    def GetDepthMapPath(self, image_idx: int, output_type: Optional[str] = None) -> str:
        return os.path.join(self.workspace_path, self.stereo_folder, "depth_maps", self.GetFileName(image_idx, output_type))

    def GetNormalMapPath(self, image_idx: int, output_type: Optional[str] = None) -> str:
        return os.path.join(self.workspace_path, self.stereo_folder, "normal_maps", self.GetFileName(image_idx, output_type))

    def GetConsistencyGraphPath(self, image_idx: int, output_type: str) -> str:
        return os.path.join(self.workspace_path, self.stereo_folder, "consistency_graphs",
                            self.GetFileName(image_idx, output_type))

    def GetBitmap(self, image_idx: int) -> np.ndarray:
        """CachedWorkspace::GetBitmap (:180-196): grey bitmap, rescaled to the (downsized) model
        image size when max_image_size is set. Cached: the array's address is what the device-side
        image cache keys on."""
        if image_idx not in self._bitmaps:
            bmp = read_bitmap_grey(self.GetBitmapPath(image_idx))
            im = self.model.images[image_idx]
            if self.max_image_size > 0 and bmp.shape != (im.height, im.width):
                from PIL import Image as PILImage
                bmp = np.ascontiguousarray(np.asarray(
                    PILImage.fromarray(bmp).resize((im.width, im.height), PILImage.BILINEAR), np.uint8))
            self._bitmaps[image_idx] = bmp
        return self._bitmaps[image_idx]


def import_pmvs_workspace(workspace: "Workspace", option_name: str):
    """ImportPMVSWorkspace (mvs/workspace.cc:250-322): output folders + patch-match.cfg / fusion.cfg from
    the `timages` line of the PMVS option file and the visibility lists of vis.dat."""
    base = os.path.join(workspace.workspace_path, workspace.stereo_folder)
    for sub in ("", "depth_maps", "normal_maps", "consistency_graphs"):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
    model = workspace.GetModel()
    with open(os.path.join(workspace.workspace_path, option_name)) as f:
        lines = [l.strip() for l in f]
    for line in lines:
        if not line.startswith("timages"):
            continue
        elems = line.split()
        num = int(elems[1])
        if num == -1:
            if len(elems) != 4:
                raise ValueError("Check failed: elems.size() == 4")
            lo, hi = int(elems[2]), int(elems[3])
            if not lo < hi:
                raise ValueError("Check failed: range_lower < range_upper")
            idxs = list(range(lo, hi))
        else:
            if num + 2 != len(elems):
                raise ValueError("Check failed: num_images + 2 == elems.size()")
            idxs = [int(e) for e in elems[2:]]
        names = [model.GetImageName(i) for i in idxs]
        overlapping = model.GetMaxOverlappingImagesFromPMVS()
        with open(os.path.join(base, "patch-match.cfg"), "w") as pm, open(os.path.join(base, "fusion.cfg"), "w") as fu:
            for i, name in enumerate(names):
                pm.write(name + "\n")
                if not overlapping:
                    pm.write("__auto__, 20\n")
                else:
                    pm.write("".join(model.GetImageName(j) + ", " for j in overlapping[i]) + "\n")
                fu.write(name + "\n")


def _downsize(im: ModelImage, max_width: int, max_height: int):
    """mvs::Image::Downsize + Rescale (mvs/image.cc:66-95): one float factor so that the image fits
    max_width x max_height, new size = round(size * factor), K scaled by the realised per-axis ratios."""
    if im.width <= max_width and im.height <= max_height:
        return
    f = min(np.float32(max_width) / np.float32(im.width), np.float32(max_height) / np.float32(im.height))
    nw = int(np.floor(np.float32(im.width) * f + np.float32(0.5)))   # std::round of a positive float
    nh = int(np.floor(np.float32(im.height) * f + np.float32(0.5)))
    sx, sy = np.float32(nw) / np.float32(im.width), np.float32(nh) / np.float32(im.height)
    im.K = im.K.copy()
    im.K[0, 0] *= sx; im.K[0, 2] *= sx
    im.K[1, 1] *= sy; im.K[1, 2] *= sy
    im.width, im.height = nw, nh


# ------------------------------------------------------------------------------------------------
# consistency graphs (mvs/consistency_graph.cc:69-139, patch_match_cuda.cu GetConsistentImageIdxs)
# ------------------------------------------------------------------------------------------------

def write_consistency_graph(path: str, width: int, height: int, data: np.ndarray):
    """ConsistencyGraph::Write: ASCII `W&H&1&` then int32 records (col, row, n, idx_0..idx_{n-1})."""
    with open(path, "wb") as f:
        f.write(f"{width}&{height}&1&".encode())
        f.write(np.asarray(data, "<i4").tobytes())


def read_consistency_graph(path: str):
    """ConsistencyGraph::Read: returns (width, height, {(row, col): [image idxs]})."""
    with open(path, "rb") as f:
        raw = f.read()
    parts = raw.split(b"&", 3)
    w, h = int(parts[0]), int(parts[1])
    data = np.frombuffer(parts[3], "<i4")
    out = {}
    i = 0
    while i < len(data):
        if i + 2 >= len(data):
            raise ValueError(f"Corrupt consistency graph: insufficient data at offset {i}")
        col, row, n = int(data[i]), int(data[i + 1]), int(data[i + 2])
        if n < 0 or not (0 <= col < w) or not (0 <= row < h):
            raise ValueError(f"Corrupt consistency graph at offset {i}")
        if n > 0:
            out[(row, col)] = data[i + 3: i + 3 + n].tolist()
        i += 3 + n
    return w, h, out
