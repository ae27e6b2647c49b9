"""Minimal scene containers + the synthetic dataset generator the reference's BA tests and
benchmarks are built on (no COLMAP database / descriptors). Rigs: a frame holds rig_from_world,
a rig holds one constant-or-variable sensor_from_rig per non-reference camera (scene/rig.h,
scene/frame.h); an image without an explicit frame is its own trivial frame.

Restates the semantics of scene/synthetic.cc (SynthesizeDataset :339-672, SynthesizeNoise
:674-770): 3-D points = normalised uniform [-1,1]^3 vectors on the unit sphere (:370-377);
frames on a radius-5 sphere looking at the origin (:458-464), `num_cameras_per_rig` cameras per
frame (the first is the reference sensor, the others get a random sensor_from_rig: rotation about
z, Gaussian translation, :417-438); default camera SIMPLE_RADIAL {1280, 512, 384, 0.05}, 1024x768
(synthetic.h:54-57); dense visibility or tracks pruned to `track_length` (:648-668); noise on
2-D points, 3-D points, rig translation and rotation about the rig z axis (:686-728).
Random streams are numpy's, not the reference's PRNG: the tests that consume this use
properties (counts, accuracy), not the raw numbers.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# colmap::CameraModelId (sensor/models.h:90-111)
SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV = 0, 1, 2, 3, 4
OPENCV_FISHEYE, FULL_OPENCV, FOV, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE, THIN_PRISM_FISHEYE = 5, 6, 7, 8, 9, 10
SIMPLE_DIVISION, DIVISION, SIMPLE_FISHEYE, FISHEYE, EUCM = 12, 13, 14, 15, 16
RAD_TAN_THIN_PRISM_FISHEYE, EQUIRECTANGULAR = 11, 17
MODEL_NAMES = {SIMPLE_PINHOLE: "SIMPLE_PINHOLE", PINHOLE: "PINHOLE", SIMPLE_RADIAL: "SIMPLE_RADIAL", RADIAL: "RADIAL",
               OPENCV: "OPENCV", OPENCV_FISHEYE: "OPENCV_FISHEYE", FOV: "FOV",
               SIMPLE_RADIAL_FISHEYE: "SIMPLE_RADIAL_FISHEYE", RADIAL_FISHEYE: "RADIAL_FISHEYE",
               SIMPLE_DIVISION: "SIMPLE_DIVISION", DIVISION: "DIVISION", SIMPLE_FISHEYE: "SIMPLE_FISHEYE",
               FISHEYE: "FISHEYE", EUCM: "EUCM", FULL_OPENCV: "FULL_OPENCV",
               THIN_PRISM_FISHEYE: "THIN_PRISM_FISHEYE", RAD_TAN_THIN_PRISM_FISHEYE: "RAD_TAN_THIN_PRISM_FISHEYE",
               EQUIRECTANGULAR: "EQUIRECTANGULAR"}
MODEL_NUM_PARAMS = {SIMPLE_PINHOLE: 3, PINHOLE: 4, SIMPLE_RADIAL: 4, RADIAL: 5, OPENCV: 8,
                    OPENCV_FISHEYE: 8, FOV: 5, SIMPLE_RADIAL_FISHEYE: 4, RADIAL_FISHEYE: 5,
                    SIMPLE_DIVISION: 4, DIVISION: 5, SIMPLE_FISHEYE: 3, FISHEYE: 4, EUCM: 6,
                    FULL_OPENCV: 12, THIN_PRISM_FISHEYE: 12, RAD_TAN_THIN_PRISM_FISHEYE: 16, EQUIRECTANGULAR: 2}
# FocalLengthIdxs / PrincipalPointIdxs / ExtraParamsIdxs (sensor/models.h)
_ONE_F = (SIMPLE_PINHOLE, SIMPLE_RADIAL, RADIAL, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE, SIMPLE_DIVISION, SIMPLE_FISHEYE)
MODEL_FOCAL_IDXS = {m: ([0] if m in _ONE_F else [0, 1]) for m in MODEL_NUM_PARAMS}
MODEL_PP_IDXS = {m: ([1, 2] if m in _ONE_F else [2, 3]) for m in MODEL_NUM_PARAMS}
MODEL_EXTRA_IDXS = {m: list(range(3 if m in _ONE_F else 4, n)) for m, n in MODEL_NUM_PARAMS.items()}
# EQUIRECTANGULAR: (width, height) are metadata parameters, never refined (MetaDataParamsIdxs, models.h:2839-2842)
MODEL_FOCAL_IDXS[EQUIRECTANGULAR] = MODEL_PP_IDXS[EQUIRECTANGULAR] = MODEL_EXTRA_IDXS[EQUIRECTANGULAR] = []


@dataclass
class Camera:
    camera_id: int
    model_id: int
    width: int
    height: int
    params: np.ndarray  # float64


@dataclass
class Point2D:
    xy: np.ndarray
    point3D_id: int = -1

    def HasPoint3D(self) -> bool:
        return self.point3D_id >= 0


@dataclass
class Image:
    """scene/image.h. Without `frame_id_` the image is its own frame with a trivial rig (its
    camera is the reference sensor) and `cam_from_world` is the pose block; with it, the pose
    block is the frame's rig_from_world and `cam_from_world` is the derived composition."""
    image_id: int
    camera_id: int
    cam_from_world: np.ndarray  # 7 doubles: qx qy qz qw tx ty tz (Rigid3d::params, geometry/rigid3.h:46-70)
    points2D: List[Point2D] = field(default_factory=list)
    frame_id_: Optional[int] = None

    @property
    def frame_id(self) -> int:
        return self.image_id if self.frame_id_ is None else self.frame_id_


@dataclass
class Rig:
    """scene/rig.h: one reference sensor, sensor_from_rig (7 doubles) for every other camera."""
    rig_id: int
    ref_camera_id: int
    sensors: Dict[int, np.ndarray] = field(default_factory=dict)  # camera_id -> sensor_from_rig

    def IsRefSensor(self, camera_id: int) -> bool:
        return camera_id == self.ref_camera_id


@dataclass
class Frame:
    """scene/frame.h: rig_from_world shared by the images taken by the rig at one instant."""
    frame_id: int
    rig_id: int
    rig_from_world: np.ndarray
    image_ids: List[int] = field(default_factory=list)


def rigid_compose(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Rigid3d a * b (apply b, then a), params = quaternion xyzw + translation."""
    q = quat_mul(a[:4], b[:4])
    t = quat_to_rot(a[:4]) @ b[4:] + a[4:]
    return np.concatenate([q, t])


@dataclass
class Point3D:
    xyz: np.ndarray
    track: List[Tuple[int, int]] = field(default_factory=list)  # (image_id, point2D_idx)


class Reconstruction:
    def __init__(self):
        self.cameras: Dict[int, Camera] = {}
        self.images: Dict[int, Image] = {}
        self.points3D: Dict[int, Point3D] = {}
        self.rigs: Dict[int, Rig] = {}      # only non-trivial rigs
        self.frames: Dict[int, Frame] = {}  # only frames of non-trivial rigs

    def HasNonTrivialFrame(self, image_id: int) -> bool:
        return self.images[image_id].frame_id_ is not None

    def IsRefInFrame(self, image_id: int) -> bool:
        img = self.images[image_id]
        if img.frame_id_ is None:
            return True
        return self.rigs[self.frames[img.frame_id_].rig_id].IsRefSensor(img.camera_id)

    def SensorFromRig(self, image_id: int) -> np.ndarray:
        img = self.images[image_id]
        return self.rigs[self.frames[img.frame_id_].rig_id].sensors[img.camera_id]

    def UpdateCamFromWorld(self):
        """cam_from_world of the images of non-trivial frames from rig_from_world (and
        sensor_from_rig): Image::CamFromWorld, scene/image.h."""
        for fr in self.frames.values():
            rig = self.rigs[fr.rig_id]
            for im in fr.image_ids:
                img = self.images[im]
                if rig.IsRefSensor(img.camera_id):
                    img.cam_from_world = fr.rig_from_world.copy()
                else:
                    img.cam_from_world = rigid_compose(rig.sensors[img.camera_id], fr.rig_from_world)

    def RegImageIds(self) -> List[int]:
        return sorted(self.images)

    def ProjectionCenter(self, image_id: int) -> np.ndarray:
        """Image::ProjectionCenter: -R^T t of cam_from_world."""
        p = self.images[image_id].cam_from_world
        return -quat_to_rot(p[:4]).T @ p[4:]

    def Transform(self, scale: float, rot: np.ndarray, trans: np.ndarray):
        """Reconstruction::Transform(new_from_old_world = Sim3d(scale, rot, trans)) (scene/reconstruction.cc:
        788-805): sensor_from_rig translations scale, every pose block becomes TransformCameraWorld of
        itself (R' = R_c R^T, t' = s t_c - R' t), points X' = s R X + t."""
        R = np.asarray(rot, np.float64).reshape(3, 3)
        t = np.asarray(trans, np.float64)

        def cam(pose):
            Rn = quat_to_rot(pose[:4]) @ R.T
            return np.concatenate([rot_to_quat(Rn), scale * pose[4:] - Rn @ t])
        for rig in self.rigs.values():
            for cid, sfr in rig.sensors.items():
                if not rig.IsRefSensor(cid):
                    sfr[4:] = scale * sfr[4:]
        for fr in self.frames.values():
            fr.rig_from_world = cam(fr.rig_from_world)
        for img in self.images.values():
            if img.frame_id_ is None:
                img.cam_from_world = cam(img.cam_from_world)
        self.UpdateCamFromWorld()
        for pt in self.points3D.values():
            pt.xyz = scale * (R @ pt.xyz) + t

    def ComputeCentroid(self, min_percentile: float = 0.1, max_percentile: float = 0.9) -> np.ndarray:
        """ComputeBoundingBoxAndCentroid over the projection centres (geometry/normalization.cc:39-92)."""
        c = np.array([self.ProjectionCenter(i) for i in sorted(self.images)])
        end = len(c) - 1
        lo = min(end, int(np.floor(min_percentile * end)))
        hi = min(end, int(np.ceil(max_percentile * end)))
        return np.array([np.sort(c[:, k])[lo:hi + 1].mean() for k in range(3)])

    def Normalize(self, fixed_scale: bool = True) -> np.ndarray:
        """Reconstruction::Normalize(fixed_scale = true) (:698-727): translation by minus the centroid of the
        projection centres. Returns the translation of normalized_from_metric."""
        assert fixed_scale, "only the fixed-scale normalisation of the pose-prior adjuster is built"
        if len(self.images) < 2:
            return np.zeros(3)
        t = -self.ComputeCentroid()
        self.Transform(1.0, np.eye(3), t)
        return t

    def NumPoints3D(self) -> int:
        return len(self.points3D)

    def DeleteObservation(self, image_id: int, point2D_idx: int):
        """Reconstruction::DeleteObservation: removes the track element; a track that drops to
        length < 2 deletes the 3-D point (scene/reconstruction.cc)."""
        p2 = self.images[image_id].points2D[point2D_idx]
        pid = p2.point3D_id
        pt = self.points3D[pid]
        if len(pt.track) <= 2:
            for (im, idx) in pt.track:
                self.images[im].points2D[idx].point3D_id = -1
            del self.points3D[pid]
            return
        pt.track.remove((image_id, point2D_idx))
        p2.point3D_id = -1

    def copy(self) -> "Reconstruction":
        return copy.deepcopy(self)


def quat_from_two_vectors(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Eigen::Quaterniond::FromTwoVectors (xyzw)."""
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    c = float(a @ b)
    if c < -1 + 1e-12:
        axis = np.cross(a, [1.0, 0, 0])
        if np.linalg.norm(axis) < 1e-6:
            axis = np.cross(a, [0, 1.0, 0])
        axis /= np.linalg.norm(axis)
        return np.array([axis[0], axis[1], axis[2], 0.0])
    axis = np.cross(a, b)
    s = np.sqrt((1 + c) * 2)
    return np.array([axis[0] / s, axis[1] / s, axis[2] / s, s / 2])


def quat_to_rot(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (xyzw), w >= 0 branch-stable (Shepperd)."""
    m = np.asarray(R, np.float64)
    tr = np.trace(m)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (m[j, i] + m[i, j]) / s
        q[k] = (m[k, i] + m[i, k]) / s
        q[3] = (m[k, j] - m[j, k]) / s
    return q / np.linalg.norm(q)


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def img_from_cam(model_id: int, params: np.ndarray, uvw: np.ndarray) -> np.ndarray:
    """CameraModel::ImgFromCam for the supported models (sensor/models.h:1231-1404); uvw (N,3)."""
    if model_id in (SIMPLE_DIVISION, DIVISION):  # closed-form division model (models.h DivisionCameraModel)
        f1 = params[0]
        f2 = params[1] if model_id == DIVISION else params[0]
        ic = 2 if model_id == DIVISION else 1
        u, v, w = uvw[:, 0], uvw[:, 1], uvw[:, 2]
        r = 2.0 / (w + np.sqrt(w * w - 4.0 * (u * u + v * v) * params[ic + 2]))
        return np.stack([f1 * r * u + params[ic], f2 * r * v + params[ic + 1]], 1)
    if model_id == EQUIRECTANGULAR:  # azimuth over the width, elevation over the height
        u, v, w = uvw[:, 0], uvw[:, 1], uvw[:, 2]
        theta, phi = np.arctan2(u, w), np.arctan2(-v, np.sqrt(u * u + w * w))
        return np.stack([(theta / (2 * np.pi) + 0.5) * params[0], (0.5 - phi / np.pi) * params[1]], 1)
    if model_id == EUCM:
        f1, f2, c1, c2, alpha, beta = params
        u, v, w = uvw[:, 0], uvw[:, 1], uvw[:, 2]
        den = alpha * np.sqrt(beta * (u * u + v * v) + w * w) + (1.0 - alpha) * w
        return np.stack([f1 * u / den + c1, f2 * v / den + c2], 1)
    uu = uvw[:, 0] / uvw[:, 2]
    vv = uvw[:, 1] / uvw[:, 2]
    if model_id == FOV:  # FOVCameraModel::Distortion, general branch (omega, radius away from 0)
        f1, f2, c1, c2, omega = params
        radius = np.sqrt(uu * uu + vv * vv)
        with np.errstate(invalid="ignore", divide="ignore"):
            factor = np.where(radius * radius < 1e-4,
                              -2.0 * (np.tan(omega / 2) * (4 * np.tan(omega / 2) ** 2 * radius ** 2 - 3)) / (3 * omega),
                              np.arctan(2.0 * radius * np.tan(omega / 2.0)) / (radius * omega))
        return np.stack([f1 * uu * factor + c1, f2 * vv * factor + c2], 1)
    if model_id in (SIMPLE_FISHEYE, FISHEYE):
        r = np.sqrt(uu * uu + vv * vv)
        with np.errstate(invalid="ignore", divide="ignore"):
            sc = np.where(r < np.finfo(np.float64).eps, 1.0, np.arctan(r) / r)
        if model_id == FISHEYE:
            return np.stack([params[0] * sc * uu + params[2], params[1] * sc * vv + params[3]], 1)
        return np.stack([params[0] * sc * uu + params[1], params[0] * sc * vv + params[2]], 1)
    if model_id == SIMPLE_PINHOLE:
        f, c1, c2 = params
        return np.stack([f * uu + c1, f * vv + c2], 1)
    if model_id == PINHOLE:
        f1, f2, c1, c2 = params
        return np.stack([f1 * uu + c1, f2 * vv + c2], 1)
    if model_id in (OPENCV_FISHEYE, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE):
        # BasePerspectiveFisheyeCameraModel: equidistant projection, then a radial polynomial in theta^2
        r = np.sqrt(uu * uu + vv * vv)
        with np.errstate(invalid="ignore", divide="ignore"):
            sc = np.where(r < np.finfo(np.float64).eps, 1.0, np.arctan(r) / r)
        fu, fv = sc * uu, sc * vv
        t2 = fu * fu + fv * fv
        if model_id == OPENCV_FISHEYE:
            f1, f2, c1, c2 = params[:4]
            ks = params[4:8]
        else:
            f1 = f2 = params[0]
            c1, c2 = params[1:3]
            ks = params[3:]
        radial = sum(k * t2 ** (i + 1) for i, k in enumerate(ks))
        return np.stack([f1 * (fu + fu * radial) + c1, f2 * (fv + fv * radial) + c2], 1)
    if model_id == RAD_TAN_THIN_PRISM_FISHEYE:  # RadTanThinPrismFisheyeModel: radial on theta, then tangential + thin prism
        f1, f2, c1, c2 = params[:4]
        ks, (p0, p1), (s0, s1, s2, s3) = params[4:10], params[10:12], params[12:16]
        r = np.sqrt(uu * uu + vv * vv)
        with np.errstate(invalid="ignore", divide="ignore"):
            sc = np.where(r < np.finfo(np.float64).eps, 1.0, np.arctan(r) / r)
        fu, fv = sc * uu, sc * vv
        t2 = fu * fu + fv * fv
        th = 1.0 + sum(k * t2 ** (i + 1) for i, k in enumerate(ks))
        xr, yr = th * fu, th * fv
        r2 = xr * xr + yr * yr
        X = xr + 2 * p1 * xr * yr + p0 * (r2 + 2 * xr * xr) + s0 * r2 + s1 * r2 * r2
        Y = yr + 2 * p0 * xr * yr + p1 * (r2 + 2 * yr * yr) + s2 * r2 + s3 * r2 * r2
        return np.stack([f1 * X + c1, f2 * Y + c2], 1)
    if model_id == FULL_OPENCV:  # FullOpenCVCameraModel::Distortion: rational radial term + tangential
        f1, f2, c1, c2, k1, k2, p1, p2, k3, k4, k5, k6 = params
        r2 = uu * uu + vv * vv
        radial = (1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3) / (1 + k4 * r2 + k5 * r2 ** 2 + k6 * r2 ** 3)
        xd = uu * radial + 2 * p1 * uu * vv + p2 * (r2 + 2 * uu * uu)
        yd = vv * radial + 2 * p2 * uu * vv + p1 * (r2 + 2 * vv * vv)
        return np.stack([f1 * xd + c1, f2 * yd + c2], 1)
    if model_id == THIN_PRISM_FISHEYE:  # ThinPrismFisheyeCameraModel: equidistant, then radial + tangential + thin prism
        f1, f2, c1, c2, k1, k2, p1, p2, k3, k4, sx1, sy1 = params
        r = np.sqrt(uu * uu + vv * vv)
        with np.errstate(invalid="ignore", divide="ignore"):
            sc = np.where(r < np.finfo(np.float64).eps, 1.0, np.arctan(r) / r)
        fu, fv = sc * uu, sc * vv
        t2 = fu * fu + fv * fv
        radial = k1 * t2 + k2 * t2 ** 2 + k3 * t2 ** 3 + k4 * t2 ** 4
        du = fu * radial + 2 * p1 * fu * fv + p2 * (t2 + 2 * fu * fu) + sx1 * t2
        dv = fv * radial + 2 * p2 * fu * fv + p1 * (t2 + 2 * fv * fv) + sy1 * t2
        return np.stack([f1 * (fu + du) + c1, f2 * (fv + dv) + c2], 1)
    if model_id == OPENCV:  # sensor/models.h OpenCVCameraModel::ImgFromCam / Distortion
        f1, f2, c1, c2, k1, k2, p1, p2 = params
        r2 = uu * uu + vv * vv
        radial = k1 * r2 + k2 * r2 * r2
        du = uu * radial + 2 * p1 * uu * vv + p2 * (r2 + 2 * uu * uu)
        dv = vv * radial + 2 * p2 * uu * vv + p1 * (r2 + 2 * vv * vv)
        return np.stack([f1 * (uu + du) + c1, f2 * (vv + dv) + c2], 1)
    if model_id == RADIAL:
        f, c1, c2, k1, k2 = params
        r2 = uu * uu + vv * vv
        a = 1 + k1 * r2 + k2 * r2 * r2
        return np.stack([f * a * uu + c1, f * a * vv + c2], 1)
    f, c1, c2, k = params
    a = 1 + k * (uu * uu + vv * vv)
    return np.stack([f * a * uu + c1, f * a * vv + c2], 1)


@dataclass
class SyntheticDatasetOptions:
    num_rigs: int = 2
    num_cameras_per_rig: int = 1
    num_frames_per_rig: int = 5
    num_points3D: int = 100
    track_length: int = -1
    camera_width: int = 1024
    camera_height: int = 768
    camera_model_id: int = SIMPLE_RADIAL
    camera_params: Tuple[float, ...] = (1280.0, 512.0, 384.0, 0.05)
    num_points2D_without_point3D: int = 10
    sensor_from_rig_translation_stddev: float = 0.05
    sensor_from_rig_rotation_stddev: float = 5.0  # degrees, about the z axis
    # extension: alternate camera models per rig (BASELINE config 5 "mixed camera models")
    mixed_models: bool = False


@dataclass
class SyntheticNoiseOptions:
    rig_from_world_translation_stddev: float = 0.0
    rig_from_world_rotation_stddev: float = 0.0
    point3D_stddev: float = 0.0
    point2D_stddev: float = 0.0


def SynthesizeDataset(options: SyntheticDatasetOptions, seed: int = 0) -> Reconstruction:
    assert options.track_length == -1 or options.track_length >= 2
    rng = np.random.default_rng(seed)
    rec = Reconstruction()
    pts = rng.uniform(-1, 1, (options.num_points3D, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    for i in range(options.num_points3D):
        rec.points3D[i + 1] = Point3D(pts[i].copy())
    image_id = 0
    frame_id = 0
    ncam = options.num_cameras_per_rig
    for rig_idx in range(options.num_rigs):
        rig = None
        cam_ids = []
        for camera_idx in range(ncam):
            cam_id = rig_idx * ncam + camera_idx + 1
            model, params = options.camera_model_id, np.array(options.camera_params, np.float64)
            if options.mixed_models and rig_idx % 2 == 1:
                model = PINHOLE
                params = np.array([options.camera_params[0], options.camera_params[0],
                                   options.camera_params[1], options.camera_params[2]], np.float64)
            rec.cameras[cam_id] = Camera(cam_id, model, options.camera_width, options.camera_height, params)
            cam_ids.append(cam_id)
            if ncam > 1:
                if rig is None:
                    rig = Rig(rig_idx + 1, cam_id)
                else:
                    sfr = np.array([0.0, 0, 0, 1, 0, 0, 0])
                    if options.sensor_from_rig_rotation_stddev > 0:
                        ang = np.deg2rad(np.clip(rng.normal(0, options.sensor_from_rig_rotation_stddev), -180, 180))
                        sfr[:4] = [0, 0, np.sin(ang / 2), np.cos(ang / 2)]
                    if options.sensor_from_rig_translation_stddev > 0:
                        sfr[4:] = rng.normal(0, options.sensor_from_rig_translation_stddev, 3)
                    rig.sensors[cam_id] = sfr
        if rig is not None:
            rec.rigs[rig.rig_id] = rig
        for _ in range(options.num_frames_per_rig):
            v = rng.uniform(-1, 1, 3)
            view_dir = -v / np.linalg.norm(v)
            proj_center = -5.0 * view_dir
            q = quat_from_two_vectors(view_dir, np.array([0.0, 0.0, 1.0]))
            t = quat_to_rot(q) @ (-proj_center)
            rig_from_world = np.concatenate([q, t])
            frame = None
            if rig is not None:
                frame_id += 1
                frame = Frame(frame_id, rig.rig_id, rig_from_world.copy())
                rec.frames[frame_id] = frame
            for cam_id in cam_ids:
                image_id += 1
                cam = rec.cameras[cam_id]
                cfw = rig_from_world
                if rig is not None and not rig.IsRefSensor(cam_id):
                    cfw = rigid_compose(rig.sensors[cam_id], rig_from_world)
                img = Image(image_id, cam_id, cfw.copy(), frame_id_=frame.frame_id if frame else None)
                if frame is not None:
                    frame.image_ids.append(image_id)
                uvw = pts @ quat_to_rot(cfw[:4]).T + cfw[4:]
                xy = img_from_cam(cam.model_id, cam.params, uvw)
                vis = (uvw[:, 2] > 0) & (xy[:, 0] >= 0) & (xy[:, 1] >= 0) & (xy[:, 0] <= options.camera_width) & \
                      (xy[:, 1] <= options.camera_height)
                p2 = [Point2D(xy[i].copy(), i + 1) for i in np.nonzero(vis)[0]]
                for _ in range(options.num_points2D_without_point3D):
                    p2.append(Point2D(np.array([rng.uniform(0, options.camera_width),
                                                rng.uniform(0, options.camera_height)])))
                order = rng.permutation(len(p2))
                img.points2D = [p2[i] for i in order]
                for idx, p in enumerate(img.points2D):
                    if p.HasPoint3D():
                        rec.points3D[p.point3D_id].track.append((image_id, idx))
                rec.images[image_id] = img
    if options.track_length > 0:
        for pid in list(rec.points3D):
            tr = rec.points3D[pid].track
            if len(tr) <= options.track_length:
                continue
            order = rng.permutation(len(tr))
            for k in order[: len(tr) - options.track_length]:
                im, idx = tr[k]
                rec.images[im].points2D[idx].point3D_id = -1
            keep = sorted(order[len(tr) - options.track_length:])
            rec.points3D[pid].track = [tr[k] for k in keep]
    return rec


def SynthesizeNoise(options: SyntheticNoiseOptions, rec: Reconstruction, seed: int = 1):
    rng = np.random.default_rng(seed)
    for fid in sorted(rec.frames):
        fr = rec.frames[fid]
        if options.rig_from_world_rotation_stddev > 0:
            ang = np.deg2rad(np.clip(rng.normal(0, options.rig_from_world_rotation_stddev), -180, 180))
            dq = np.array([0, 0, np.sin(ang / 2), np.cos(ang / 2)])
            fr.rig_from_world[:4] = quat_mul(fr.rig_from_world[:4], dq)
        if options.rig_from_world_translation_stddev > 0:
            fr.rig_from_world[4:] += rng.normal(0, options.rig_from_world_translation_stddev, 3)
    rec.UpdateCamFromWorld()
    for image_id in rec.RegImageIds():
        img = rec.images[image_id]
        if img.frame_id_ is not None:
            continue
        if options.rig_from_world_rotation_stddev > 0:
            ang = np.deg2rad(np.clip(rng.normal(0, options.rig_from_world_rotation_stddev), -180, 180))
            dq = np.array([0, 0, np.sin(ang / 2), np.cos(ang / 2)])
            img.cam_from_world[:4] = quat_mul(img.cam_from_world[:4], dq)
        if options.rig_from_world_translation_stddev > 0:
            img.cam_from_world[4:] += rng.normal(0, options.rig_from_world_translation_stddev, 3)
    if options.point2D_stddev > 0:
        for image_id in sorted(rec.images):
            for p in rec.images[image_id].points2D:
                p.xy = p.xy + rng.normal(0, options.point2D_stddev, 2)
    if options.point3D_stddev > 0:
        for pid in sorted(rec.points3D):
            rec.points3D[pid].xyz = rec.points3D[pid].xyz + rng.normal(0, options.point3D_stddev, 3)


def synthesize_flat(num_frames: int, num_points: int, track_length: int, seed: int = 42,
                    mixed_models: bool = False, noise: Optional[SyntheticNoiseOptions] = None):
    """Vectorised generator for the large bench configs (1000 frames x 200k points would take
    minutes through the object model): returns flat arrays directly, same distributions as
    SynthesizeDataset/SynthesizeNoise with one camera per frame (num_rigs = num_frames)."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1, 1, (num_points, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    v = rng.uniform(-1, 1, (num_frames, 3))
    view = -v / np.linalg.norm(v, axis=1, keepdims=True)
    poses = np.zeros((num_frames, 7))
    Rs = np.zeros((num_frames, 3, 3))
    for i in range(num_frames):
        q = quat_from_two_vectors(view[i], np.array([0.0, 0.0, 1.0]))
        Rs[i] = quat_to_rot(q)
        poses[i, :4] = q
        poses[i, 4:] = Rs[i] @ (5.0 * view[i])
    cams = np.zeros((num_frames, 16))  # BA_CAM_STRIDE
    model = np.full(num_frames, SIMPLE_RADIAL, np.int32)
    cams[:, :4] = [1280.0, 512.0, 384.0, 0.05]
    if mixed_models == "three":  # BASELINE config[4] "mixed camera models": thirds of SIMPLE_RADIAL / PINHOLE / OPENCV
        model[1::3] = PINHOLE
        cams[1::3, :4] = [1280.0, 1280.0, 512.0, 384.0]
        model[2::3] = OPENCV
        cams[2::3, :8] = [1280.0, 1280.0, 512.0, 384.0, 0.05, 0.01, 1e-3, -1e-3]
    elif mixed_models:
        model[1::2] = PINHOLE
        cams[1::2, :4] = [1280.0, 1280.0, 512.0, 384.0]
    # every frame sees every point (all points project inside 1024x768 from radius 5); each
    # track = `track_length` distinct random frames (the reference shuffles and deletes, :662-668)
    obs_pose = rng.integers(0, num_frames, (num_points, track_length))
    obs_pose.sort(axis=1)
    for _ in range(64):
        dup = np.zeros_like(obs_pose, bool)
        dup[:, 1:] = obs_pose[:, 1:] == obs_pose[:, :-1]
        if not dup.any():
            break
        obs_pose[dup] = rng.integers(0, num_frames, int(dup.sum()))
        obs_pose.sort(axis=1)
    obs_point = np.repeat(np.arange(num_points), track_length)
    obs_pose = obs_pose.reshape(-1)
    # (column-wise: np.einsum over the gathered 20 M x 3 x 3 rotations of config[4] takes a minute)
    R9, P = Rs.reshape(num_frames, 9), pts[obs_point]
    uvw = poses[obs_pose, 4:]
    for i in range(3):
        uvw[:, i] += R9[obs_pose, 3 * i] * P[:, 0] + R9[obs_pose, 3 * i + 1] * P[:, 1] + R9[obs_pose, 3 * i + 2] * P[:, 2]
    xy = np.zeros((len(obs_pose), 2))
    for m in (SIMPLE_RADIAL, PINHOLE, OPENCV):
        sel = model[obs_pose] == m
        if sel.any():
            # per-observation params
            pr = cams[obs_pose[sel]]
            uu = uvw[sel, 0] / uvw[sel, 2]
            vv = uvw[sel, 1] / uvw[sel, 2]
            if m == SIMPLE_RADIAL:
                a = 1 + pr[:, 3] * (uu * uu + vv * vv)
                xy[sel] = np.stack([pr[:, 0] * a * uu + pr[:, 1], pr[:, 0] * a * vv + pr[:, 2]], 1)
            elif m == PINHOLE:
                xy[sel] = np.stack([pr[:, 0] * uu + pr[:, 2], pr[:, 1] * vv + pr[:, 3]], 1)
            else:  # OPENCV (reference sensor/models.h OpenCVCameraModel::Distortion)
                k1, k2, p1, p2 = pr[:, 4], pr[:, 5], pr[:, 6], pr[:, 7]
                r2 = uu * uu + vv * vv
                rad = k1 * r2 + k2 * r2 * r2
                du = uu * rad + 2 * p1 * uu * vv + p2 * (r2 + 2 * uu * uu)
                dv = vv * rad + 2 * p2 * uu * vv + p1 * (r2 + 2 * vv * vv)
                xy[sel] = np.stack([pr[:, 0] * (uu + du) + pr[:, 2], pr[:, 1] * (vv + dv) + pr[:, 3]], 1)
    if noise is not None:
        if noise.rig_from_world_rotation_stddev > 0:
            ang = np.deg2rad(np.clip(rng.normal(0, noise.rig_from_world_rotation_stddev, num_frames), -180, 180))
            for i in range(num_frames):
                poses[i, :4] = quat_mul(poses[i, :4], np.array([0, 0, np.sin(ang[i] / 2), np.cos(ang[i] / 2)]))
        if noise.rig_from_world_translation_stddev > 0:
            poses[:, 4:] += rng.normal(0, noise.rig_from_world_translation_stddev, (num_frames, 3))
        if noise.point2D_stddev > 0:
            xy += rng.normal(0, noise.point2D_stddev, xy.shape)
        if noise.point3D_stddev > 0:
            pts = pts + rng.normal(0, noise.point3D_stddev, pts.shape)
    return dict(poses=poses, cams=cams, cam_model=model, points=pts,
                obs_pose=obs_pose.astype(np.int32), obs_cam=obs_pose.astype(np.int32),
                obs_point=obs_point.astype(np.int32), obs_xy=xy)
