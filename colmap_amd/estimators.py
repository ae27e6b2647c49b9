"""Host-side mirror of the reference's bundle-adjustment interface over the C ABI.

Same names / meaning as colmap (reference src/colmap/estimators/bundle_adjustment.h:50-234):
`BundleAdjustmentGauge`, `BundleAdjustmentTerminationType`, `BundleAdjustmentBackend` (with
the new value `MI355X` after CERES=0, CASPAR=1 -- pycolmap pins those two,
pycolmap/estimators/bundle_adjustment_test.py:27-39), `BundleAdjustmentConfig`,
`BundleAdjustmentOptions`, `BundleAdjustmentSummary`, `BundleAdjuster`,
`CreateDefaultBundleAdjuster(options, config, reconstruction)`.

`flatten()` is the adapter: it applies the problem-construction rules of
`DefaultBundleAdjuster` (bundle_adjustment_ceres.cc:606-889) -- which observations become
residuals, which blocks are constant, gauge fixing -- and produces the SoA `ba_problem` of
include/colmap_amd_ba.h, the way CasparBundleAdjuster does for its solver
(bundle_adjustment_caspar.cc:61-377). `Solve()` writes variable blocks back in place
(:767-801). All numerics happen behind `ba_solve` in libcolmap_amd.so (HIP, gfx950).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Set

import numpy as np

from . import scene
from ._lib import lib

CAM_STRIDE = 16


class BundleAdjustmentGauge(enum.IntEnum):
    UNSPECIFIED = -1
    TWO_CAMS_FROM_WORLD = 0
    THREE_POINTS = 1


class LossFunctionType(enum.IntEnum):
    """CeresBundleAdjustmentOptions::LossFunctionType (bundle_adjustment_ceres.h:42)."""
    TRIVIAL = 0
    SOFT_L1 = 1
    CAUCHY = 2
    HUBER = 3


class BundleAdjustmentTerminationType(enum.IntEnum):
    CONVERGENCE = 0
    NO_CONVERGENCE = 1
    FAILURE = 2
    USER_SUCCESS = 3
    USER_FAILURE = 4


class BundleAdjustmentBackend(enum.IntEnum):
    CERES = 0
    CASPAR = 1
    MI355X = 2


class BundleAdjustmentConfig:
    """colmap::BundleAdjustmentConfig (bundle_adjustment.h:77-150)."""

    def __init__(self):
        self.fixed_gauge_ = BundleAdjustmentGauge.UNSPECIFIED
        self.image_ids_: Set[int] = set()
        self.variable_point3D_ids_: Set[int] = set()
        self.constant_point3D_ids_: Set[int] = set()
        self.ignored_point3D_ids_: Set[int] = set()
        self.constant_cam_intrinsics_: Set[int] = set()
        self.constant_rig_from_world_poses_: Set[int] = set()
        self.constant_sensor_from_rig_: Set[int] = set()

    def FixGauge(self, gauge):
        self.fixed_gauge_ = BundleAdjustmentGauge(gauge)

    def FixedGauge(self):
        return self.fixed_gauge_

    def NumImages(self):
        return len(self.image_ids_)

    def AddImage(self, image_id):
        self.image_ids_.add(image_id)

    def HasImage(self, image_id):
        return image_id in self.image_ids_

    def RemoveImage(self, image_id):
        self.image_ids_.discard(image_id)

    def Images(self):
        return sorted(self.image_ids_)

    def SetConstantCamIntrinsics(self, camera_id):
        self.constant_cam_intrinsics_.add(camera_id)

    def SetVariableCamIntrinsics(self, camera_id):
        self.constant_cam_intrinsics_.discard(camera_id)

    def HasConstantCamIntrinsics(self, camera_id):
        return camera_id in self.constant_cam_intrinsics_

    def SetConstantRigFromWorldPose(self, frame_id):
        self.constant_rig_from_world_poses_.add(frame_id)

    def SetVariableRigFromWorldPose(self, frame_id):
        self.constant_rig_from_world_poses_.discard(frame_id)

    def HasConstantRigFromWorldPose(self, frame_id):
        return frame_id in self.constant_rig_from_world_poses_

    def SetConstantSensorFromRigPose(self, sensor_id):
        self.constant_sensor_from_rig_.add(sensor_id)

    def SetVariableSensorFromRigPose(self, sensor_id):
        self.constant_sensor_from_rig_.discard(sensor_id)

    def HasConstantSensorFromRigPose(self, sensor_id):
        return sensor_id in self.constant_sensor_from_rig_

    def AddVariablePoint(self, point3D_id):
        assert point3D_id not in self.constant_point3D_ids_ and point3D_id not in self.ignored_point3D_ids_
        self.variable_point3D_ids_.add(point3D_id)

    def AddConstantPoint(self, point3D_id):
        assert point3D_id not in self.variable_point3D_ids_ and point3D_id not in self.ignored_point3D_ids_
        self.constant_point3D_ids_.add(point3D_id)

    def IgnorePoint(self, point3D_id):
        self.ignored_point3D_ids_.add(point3D_id)

    def IsIgnoredPoint(self, point3D_id):
        return point3D_id in self.ignored_point3D_ids_

    def VariablePoints(self):
        return sorted(self.variable_point3D_ids_)

    def ConstantPoints(self):
        return sorted(self.constant_point3D_ids_)


@dataclass
class SolverOptions:
    """The ceres::Solver::Options fields COLMAP sets (bundle_adjustment_ceres.cc:102-115) plus
    the Ceres defaults the solve depends on (trust-region schedule)."""
    max_num_iterations: int = 100
    max_linear_solver_iterations: int = 200
    function_tolerance: float = 0.0
    gradient_tolerance: float = 1e-4
    parameter_tolerance: float = 0.0
    initial_trust_region_radius: float = 1e4
    max_trust_region_radius: float = 1e16
    min_trust_region_radius: float = 1e-32
    min_relative_decrease: float = 1e-3
    min_lm_diagonal: float = 1e-6
    max_lm_diagonal: float = 1e32
    eta: float = 1e-1
    max_num_consecutive_invalid_steps: int = 10
    jacobi_scaling: bool = True
    # CeresBundleAdjustmentOptions::loss_function_type / loss_function_scale
    # (bundle_adjustment_ceres.h:42-51)
    loss_type: int = 0   # LossFunctionType.TRIVIAL
    loss_scale: float = 1.0
    # ba_options.linear_solver_type: ITERATIVE_SCHUR (implicit Schur PCG + Schur-Jacobi, the benchmarked
    # path), DENSE_SCHUR (reduced camera system formed and Cholesky-solved) or the reference's choice by
    # problem size (CeresBundleAdjustmentOptions::CreateSolverOptions, bundle_adjustment_ceres.cc:203-213)
    linear_solver_type: int = 0
    # ba_options.operator_precision (MI355X option): OPERATOR_F32 lets the inexact inner CG solve stream fp32
    # copies of the Jacobian columns (fp64 accumulation; everything else stays fp64) -- include/colmap_amd_ba.h
    operator_precision: int = 0
    # ba_options.iteration_callback: callable(summary) -> CALLBACK_CONTINUE / CALLBACK_TERMINATE / CALLBACK_ABORT, asked
    # after the initial evaluation and after every LM iteration (ceres::Solver::Options::callbacks as the reference's
    # controller uses them, controllers/bundle_adjustment.cc:40-57); `summary` has the fields of ba_iteration_summary
    iteration_callback: Optional[object] = None


@dataclass
class BundleAdjustmentOptions:
    """colmap::BundleAdjustmentOptions (bundle_adjustment.h:173-209)."""
    refine_focal_length: bool = True
    refine_principal_point: bool = False
    refine_extra_params: bool = True
    refine_sensor_from_rig: bool = True
    refine_rig_from_world: bool = True
    refine_points3D: bool = True
    min_track_length: int = 0
    constant_rig_from_world_rotation: bool = False
    print_summary: bool = True
    backend: BundleAdjustmentBackend = BundleAdjustmentBackend.MI355X
    gpu_index: str = "-1"
    # adapter level: the reference's solver choice by problem size (CreateSolverOptions,
    # bundle_adjustment_ceres.cc:203-213); solve_flat's own default stays ITERATIVE_SCHUR
    solver_options: SolverOptions = field(default_factory=lambda: SolverOptions(linear_solver_type=2))
    # The AUTO rule's thresholds. The reference keeps one pair per device class (bundle_adjustment_ceres.h:68-71:
    # 50 / 1000 images for its CPU solvers, 200 / 4000 for Ceres-CUDA). Measured on the MI355X over three seeds per size
    # (scripts/ba_tier_crossover.py, profiles/r06_ba_tier_crossover.json; criterion: time to the cost the
    # exact tier has after three LM steps): the exact tiers get there first at every size from 50 to 4000 images (at 350
    # and 1000 Schur-PCG is level with them, 14.4 vs 14.5 ms and 44.9 vs 47.4 ms; from 1500 on it does not reach that
    # cost within 30 LM iterations on any seed), so the rule is the reference's own GPU pair -- monotone in the image
    # count, exact wherever the reduced camera system fits the dense formation (n_c <= 32 768, about 4000 images).
    max_num_images_direct_dense_gpu_solver: int = 200
    max_num_images_direct_sparse_gpu_solver: int = 4000

    def Check(self) -> bool:
        return self.min_track_length >= 0


@dataclass
class BundleAdjustmentSummary:
    """colmap::BundleAdjustmentSummary (bundle_adjustment.h:63-74) + solver statistics."""
    termination_type: BundleAdjustmentTerminationType = BundleAdjustmentTerminationType.FAILURE
    num_residuals: int = 0
    num_iterations: int = 0
    num_successful_steps: int = 0
    num_effective_parameters: int = 0
    total_linear_iterations: int = 0
    initial_cost: float = 0.0
    final_cost: float = 0.0
    lm_seconds: float = 0.0
    log_cost: Optional[np.ndarray] = None
    log_linear_iters: Optional[np.ndarray] = None
    linear_solver_used: int = 0     # ba_result.linear_solver_used: the tier that ran (SOLVER_*)
    factor_seconds: float = 0.0     # exact tiers: time inside the blocked Cholesky
    setup_seconds: float = 0.0      # flattening into the device layout, index lists, upload: everything before the LM loop

    def IsSolutionUsable(self) -> bool:
        return self.termination_type in (BundleAdjustmentTerminationType.CONVERGENCE,
                                         BundleAdjustmentTerminationType.NO_CONVERGENCE,
                                         BundleAdjustmentTerminationType.USER_SUCCESS)

    def BriefReport(self) -> str:
        return (f"{self.termination_type.name}: {self.num_residuals} residuals, "
                f"{self.num_iterations} iterations, cost {self.initial_cost:.6e} -> {self.final_cost:.6e}")


# ---------------------------------------------------------------------------------------------
# Flat problem (SoA) shared by the C ABI binding and the oracle binding
# ---------------------------------------------------------------------------------------------

@dataclass
class FlatProblem:
    poses: np.ndarray          # (N_c, 7) f64
    cams: np.ndarray           # (N_k, 12) f64
    cam_model: np.ndarray      # (N_k,) i32
    points: np.ndarray         # (N_p, 3) f64
    obs_pose: np.ndarray       # (N_o,) i32
    obs_cam: np.ndarray
    obs_point: np.ndarray
    obs_xy: np.ndarray         # (N_o, 2) f64
    pose_const: np.ndarray     # (N_c,) u8
    pose_fixed_t: np.ndarray   # (N_c,) i8
    cam_const: np.ndarray      # (N_k, 12) u8
    point_const: np.ndarray    # (N_p,) u8
    # rigs: constant sensor_from_rig per observation (None: every frame is trivial)
    sensors: Optional[np.ndarray] = None      # (N_s, 7) f64
    obs_sensor: Optional[np.ndarray] = None   # (N_o,) i32, -1 = trivial
    sensor_const: Optional[np.ndarray] = None  # (N_s,) u8, 1 = constant sensor_from_rig (None: all constant)
    sensor_ids: Optional[list] = None          # camera id of every sensor slot (adapter write-back)
    # position priors (PosePriorBundleAdjuster): residual = sqrt_info (position + R^-1 t) of the pose block,
    # or of sensor_from_rig * rig_from_world when prior_sensor >= 0
    prior_pose: Optional[np.ndarray] = None       # (N_q,) i32
    prior_sensor: Optional[np.ndarray] = None     # (N_q,) i32, -1 = the pose block is the sensor pose
    prior_position: Optional[np.ndarray] = None   # (N_q, 3) f64
    prior_sqrt_info: Optional[np.ndarray] = None  # (N_q, 3, 3) f64: left square root of the information matrix
    prior_loss_type: int = 0
    prior_loss_scale: float = 1.0
    image_slots: Optional[dict] = None            # flatten(): parameterized image id -> (pose slot, sensor slot)
    # id maps for write-back
    pose_ids: List[int] = field(default_factory=list)
    cam_ids: List[int] = field(default_factory=list)
    point_ids: List[int] = field(default_factory=list)

    @staticmethod
    def from_arrays(d: dict, refine_focal=True, refine_pp=False, refine_extra=True) -> "FlatProblem":
        """Flat arrays of scene.synthesize_flat -> problem with COLMAP's default constant-ness
        (principal point fixed) and no gauge fixing yet."""
        n_c, n_k, n_p = len(d["poses"]), len(d["cams"]), len(d["points"])
        cam_const = np.ones((n_k, CAM_STRIDE), np.uint8)
        for k in range(n_k):
            m = int(d["cam_model"][k])
            if refine_focal:
                cam_const[k, scene.MODEL_FOCAL_IDXS[m]] = 0
            if refine_pp:
                cam_const[k, scene.MODEL_PP_IDXS[m]] = 0
            if refine_extra and scene.MODEL_EXTRA_IDXS[m]:
                cam_const[k, scene.MODEL_EXTRA_IDXS[m]] = 0
        return FlatProblem(
            poses=np.ascontiguousarray(d["poses"], np.float64), cams=np.ascontiguousarray(d["cams"], np.float64),
            cam_model=np.ascontiguousarray(d["cam_model"], np.int32),
            points=np.ascontiguousarray(d["points"], np.float64),
            obs_pose=np.ascontiguousarray(d["obs_pose"], np.int32), obs_cam=np.ascontiguousarray(d["obs_cam"], np.int32),
            obs_point=np.ascontiguousarray(d["obs_point"], np.int32), obs_xy=np.ascontiguousarray(d["obs_xy"], np.float64),
            pose_const=np.zeros(n_c, np.uint8), pose_fixed_t=np.full(n_c, -1, np.int8), cam_const=cam_const,
            point_const=np.zeros(n_p, np.uint8))

    def copy(self) -> "FlatProblem":
        import copy
        return copy.deepcopy(self)


def fix_gauge_two_cams(fp: FlatProblem, order: Optional[List[int]] = None,
                       frames: Optional[List[int]] = None):
    """FixGaugeWithTwoCamsFromWorld (bundle_adjustment_ceres.cc:308-416) on a flat problem: frame 1
    fully constant, the largest-baseline translation coordinate of frame 2 constant. `order` lists
    the pose slots of the candidate images in image-id order, `frames` their frame ids (two images
    of one frame are not two cameras). Returns True when the gauge was fixed with two cameras."""
    idx = order if order is not None else list(range(len(fp.poses)))
    frm = frames if frames is not None else idx
    used = np.zeros(len(fp.poses), bool)
    used[fp.obs_pose] = True
    cand = [(i, f) for i, f in zip(idx, frm) if used[i]]
    # first, the already fixed cameras (:347-357)
    image1 = None
    for i, f in cand:
        if fp.pose_const[i]:
            if image1 is None:
                image1 = (i, f)
            elif image1[1] != f:
                return True  # two frames already fixed
    image2, fixed_dim = None, 0
    for i, f in cand:
        if image1 is None:
            image1 = (i, f)
            continue
        if f == image1[1] or fp.pose_const[i]:
            continue
        # baseline = (frame1_from_world * inverse(frame2_from_world)).translation (:374-377)
        q1, t1 = fp.poses[image1[0], :4], fp.poses[image1[0], 4:]
        q2, t2 = fp.poses[i, :4], fp.poses[i, 4:]
        R1, R2 = scene.quat_to_rot(q1), scene.quat_to_rot(q2)
        baseline = t1 - R1 @ R2.T @ t2
        k = int(np.argmax(np.abs(baseline)))
        if abs(baseline[k]) > 1e-9:
            image2, fixed_dim = i, k
            break
    if image1 is None or image2 is None:
        return False
    fp.pose_const[image1[0]] = 1
    fp.pose_fixed_t[image2] = fixed_dim
    return True


def fix_gauge_three_points(fp: FlatProblem):
    """FixGaugeWithThreePoints (bundle_adjustment_ceres.cc:270-301): hold three points whose
    coordinates span rank 3 (already-constant points count first)."""
    used = np.zeros(len(fp.points), bool)
    used[fp.obs_point] = True
    chosen: List[np.ndarray] = []

    def maybe(pt):
        M = np.stack(chosen + [pt], 1)
        if np.linalg.matrix_rank(M) > len(chosen):
            chosen.append(pt)
            return True
        return False

    for j in np.nonzero(used)[0]:
        if fp.point_const[j] and maybe(fp.points[j]) and len(chosen) >= 3:
            return True
    for j in np.nonzero(used)[0]:
        if not fp.point_const[j] and maybe(fp.points[j]):
            fp.point_const[j] = 1
            if len(chosen) >= 3:
                return True
    return False


def flatten(options: BundleAdjustmentOptions, config: BundleAdjustmentConfig,
            rec: scene.Reconstruction) -> FlatProblem:
    """DefaultBundleAdjuster ctor (bundle_adjustment_ceres.cc:606-664)."""
    config_const_cams = set(config.constant_cam_intrinsics_)
    pose_ids: List[tuple] = []  # ("frame", frame_id) | ("image", image_id): where the block lives
    pose_index: Dict[tuple, int] = {}
    pose_params: List[np.ndarray] = []
    cam_ids: List[int] = []
    cam_index: Dict[int, int] = {}
    point_ids: List[int] = []
    point_index: Dict[int, int] = {}
    pose_is_const: List[int] = []
    sensor_index: Dict[int, int] = {}
    sensor_params: List[np.ndarray] = []
    sensor_is_const: List[int] = []
    sensor_rig: List[int] = []
    obs = []  # (pose, cam, point, x, y, sensor)
    num_obs_of_point: Dict[int, int] = {}

    def frame_pose(img):
        """(where, params) of the pose block an image's frame owns."""
        if img.frame_id_ is None:
            return ("image", img.image_id), img.cam_from_world
        return ("frame", img.frame_id_), rec.frames[img.frame_id_].rig_from_world

    def pose_slot(where, params, const):
        key = (where, const)
        if key not in pose_index:
            pose_index[key] = len(pose_ids)
            pose_ids.append(where)
            pose_params.append(np.asarray(params, np.float64))
            pose_is_const.append(1 if const else 0)
        return pose_index[key]

    def sensor_slot(camera_id, params, const, rig_id):
        if camera_id not in sensor_index:
            sensor_index[camera_id] = len(sensor_params)
            sensor_params.append(np.asarray(params, np.float64))
            sensor_is_const.append(1 if const else 0)
            sensor_rig.append(rig_id)
        return sensor_index[camera_id]

    def cam_slot(camera_id):
        if camera_id not in cam_index:
            cam_index[camera_id] = len(cam_ids)
            cam_ids.append(camera_id)
        return cam_index[camera_id]

    def point_slot(pid):
        if pid not in point_index:
            point_index[pid] = len(point_ids)
            point_ids.append(pid)
        return point_index[pid]

    def image_blocks(img, const_frame):
        """Pose slot and sensor slot of the residuals of `img`: AddImageWithTrivialFrame (:699-750)
        / AddImageWithNonTrivialFrame (:752-822)."""
        if rec.IsRefInFrame(img.image_id):
            where, params = frame_pose(img)
            return pose_slot(where, params, const_frame), -1
        sensor_from_rig = rec.SensorFromRig(img.image_id)
        const_sensor = (not options.refine_sensor_from_rig) or config.HasConstantSensorFromRigPose(img.camera_id)
        where, params = frame_pose(img)
        rig_id = rec.frames[img.frame_id_].rig_id
        if const_frame and const_sensor:
            # both constant: ReprojErrorConstantPoseCostFunctor on the composed pose (:769-772,797-803)
            return pose_slot(("image", img.image_id), scene.rigid_compose(sensor_from_rig, params), True), -1
        # constant sensor: RigReprojErrorConstantRigCostFunctor (:804-810); variable sensor: the general
        # RigReprojErrorCostFunctor with sensor_from_rig as a parameter block of its own (:811-820), also
        # when the frame is constant ("rare enough that we do not have a specialized cost function")
        return pose_slot(where, params, const_frame), sensor_slot(img.camera_id, sensor_from_rig, const_sensor, rig_id)

    parameterized_cams: Set[int] = set()
    image_slots: Dict[int, tuple] = {}  # parameterized image -> (pose slot, sensor slot or -1)
    gauge_order: List[int] = []  # pose slots of the config's images in image-id order
    gauge_frames: List[int] = []  # and their frame identities
    # AddImageToProblem (:688-697)
    for image_id in config.Images():
        img = rec.images[image_id]
        const_pose = (not options.refine_rig_from_world) or config.HasConstantRigFromWorldPose(img.frame_id)
        n = 0
        slot = None
        for p2 in img.points2D:
            if not p2.HasPoint3D() or config.IsIgnoredPoint(p2.point3D_id):
                continue
            pt = rec.points3D[p2.point3D_id]
            assert len(pt.track) > 1
            if options.min_track_length > 0 and len(pt.track) < options.min_track_length:
                continue
            n += 1
            num_obs_of_point[p2.point3D_id] = num_obs_of_point.get(p2.point3D_id, 0) + 1
            if slot is None:
                slot = image_blocks(img, const_pose)
            obs.append((slot[0], cam_slot(img.camera_id), point_slot(p2.point3D_id), p2.xy[0], p2.xy[1], slot[1]))
        if n > 0:
            image_slots[image_id] = slot
            parameterized_cams.add(img.camera_id)
            # gauge candidates: reference sensors and constant sensor_from_rig only
            # (IsParameterizedConstSensor, bundle_adjustment_ceres.cc:347-385)
            if slot[1] < 0 or sensor_is_const[slot[1]]:
                gauge_order.append(slot[0])
                gauge_frames.append(img.frame_id if img.frame_id_ is not None else -img.image_id - 1)
    # AddPointToProblem (:826-887): observations from images outside the config, constant pose
    for pid in config.VariablePoints() + config.ConstantPoints():
        pt = rec.points3D[pid]
        if options.min_track_length > 0 and len(pt.track) < options.min_track_length:
            continue
        if num_obs_of_point.get(pid, 0) == len(pt.track):
            num_obs_of_point.setdefault(pid, 0)
            continue
        num_obs_of_point.setdefault(pid, 0)
        for (im, idx) in pt.track:
            if config.HasImage(im):
                continue
            num_obs_of_point[pid] += 1
            img = rec.images[im]
            p2 = img.points2D[idx]
            # constant pose of the observing camera (:851-881), whatever its rig
            obs.append((pose_slot(("image", im), img.cam_from_world, True), cam_slot(img.camera_id),
                        point_slot(pid), p2.xy[0], p2.xy[1], -1))
            if img.camera_id not in parameterized_cams:
                parameterized_cams.add(img.camera_id)
                config_const_cams.add(img.camera_id)  # (:883-886)

    n_c, n_k, n_p = len(pose_ids), len(cam_ids), len(point_ids)
    poses = np.array(pose_params, np.float64).reshape(n_c, 7)
    cams = np.zeros((n_k, CAM_STRIDE))
    cam_model = np.zeros(n_k, np.int32)
    cam_const = np.ones((n_k, CAM_STRIDE), np.uint8)
    # ParameterizeCameras (:419-469)
    constant_camera = not (options.refine_focal_length or options.refine_principal_point or
                           options.refine_extra_params)
    for k, cid in enumerate(cam_ids):
        cam = rec.cameras[cid]
        m = cam.model_id
        if m not in scene.MODEL_NUM_PARAMS:
            raise ValueError(f"camera model {m} is not supported by the MI355X backend yet")
        cams[k, : len(cam.params)] = cam.params
        cam_model[k] = m
        if constant_camera or cid in config_const_cams:
            continue
        if options.refine_focal_length:
            cam_const[k, scene.MODEL_FOCAL_IDXS[m]] = 0
        if options.refine_principal_point:
            cam_const[k, scene.MODEL_PP_IDXS[m]] = 0
        if options.refine_extra_params and scene.MODEL_EXTRA_IDXS[m]:
            cam_const[k, scene.MODEL_EXTRA_IDXS[m]] = 0
    points = np.array([rec.points3D[i].xyz for i in point_ids], np.float64).reshape(n_p, 3)
    # ParameterizePoints (:548-563)
    point_const = np.zeros(n_p, np.uint8)
    for j, pid in enumerate(point_ids):
        if (not options.refine_points3D) or len(rec.points3D[pid].track) > num_obs_of_point.get(pid, 0):
            point_const[j] = 1
    for pid in config.ConstantPoints():
        if pid in point_index:
            point_const[point_index[pid]] = 1
    o = np.array(obs, np.float64).reshape(-1, 6)
    fp = FlatProblem(
        poses=poses, cams=cams, cam_model=cam_model, points=points,
        obs_pose=o[:, 0].astype(np.int32), obs_cam=o[:, 1].astype(np.int32),
        obs_point=o[:, 2].astype(np.int32), obs_xy=np.ascontiguousarray(o[:, 3:5]),
        pose_const=np.array(pose_is_const, np.uint8), pose_fixed_t=np.full(n_c, -1, np.int8),
        cam_const=cam_const, point_const=point_const, pose_ids=pose_ids, cam_ids=cam_ids,
        point_ids=point_ids)
    fp.image_slots = image_slots
    if sensor_params:
        fp.sensors = np.array(sensor_params, np.float64).reshape(-1, 7)
        fp.obs_sensor = np.ascontiguousarray(o[:, 5].astype(np.int32))
        fp.sensor_const = np.array(sensor_is_const, np.uint8)
        fp.sensor_ids = sorted(sensor_index, key=sensor_index.get)
        # a rig whose reference sensor is not part of the problem keeps its sensor_from_rig constant:
        # the relative poses would not be constrained (ParameterizeImages, :526-543)
        for k, rig_id in enumerate(sensor_rig):
            if rec.rigs[rig_id].ref_camera_id not in parameterized_cams:
                fp.sensor_const[k] = 1
    # gauge (:646-663)
    if config.FixedGauge() == BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD:
        if options.refine_rig_from_world:
            # candidates: the config's images in std::set<image_t> order (:347-385); every sensor is a
            # reference sensor or has a constant sensor_from_rig here (IsParameterizedConstSensor)
            if not fix_gauge_two_cams(fp, gauge_order, gauge_frames):
                fix_gauge_three_points(fp)
    elif config.FixedGauge() == BundleAdjustmentGauge.THREE_POINTS:
        fix_gauge_three_points(fp)
    if options.constant_rig_from_world_rotation:
        # SubsetManifold(7, {0, 1, 2, 3[, 4 + fixed_dim]}) on every variable rig_from_world
        # (bundle_adjustment_ceres.cc:404-408, 513-516): the rotation is held, only the translation
        # (minus the gauge coordinate of the second gauge frame) is refined
        var = fp.pose_const == 0
        k = fp.pose_fixed_t[var]
        fp.pose_fixed_t[var] = POSE_ROT_CONST + np.where(k >= 0, k, 3).astype(np.int8)
    return fp


# ---------------------------------------------------------------------------------------------
# C ABI binding
# ---------------------------------------------------------------------------------------------

class ba_problem(C.Structure):
    _fields_ = [
        ("num_poses", C.c_int32), ("num_cams", C.c_int32), ("num_points", C.c_int32),
        ("num_obs", C.c_int64),
        ("poses", C.c_void_p), ("cams", C.c_void_p), ("cam_model", C.c_void_p), ("points", C.c_void_p),
        ("obs_pose", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_point", C.c_void_p), ("obs_xy", C.c_void_p),
        ("pose_const", C.c_void_p), ("pose_fixed_t", C.c_void_p), ("cam_const", C.c_void_p),
        ("point_const", C.c_void_p),
        ("num_sensors", C.c_int32), ("sensors", C.c_void_p), ("obs_sensor", C.c_void_p),
        ("sensor_const", C.c_void_p),
        ("num_priors", C.c_int32), ("prior_pose", C.c_void_p), ("prior_sensor", C.c_void_p),
        ("prior_position", C.c_void_p), ("prior_sqrt_info", C.c_void_p),
        ("prior_loss_type", C.c_int32), ("prior_loss_scale", C.c_double),
    ]


class ba_options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("max_linear_solver_iterations", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double), ("eta", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("num_threads", C.c_int32), ("max_log", C.c_int32),
        ("loss_type", C.c_int32), ("loss_scale", C.c_double),
        ("linear_solver_type", C.c_int32), ("operator_precision", C.c_int32),
        ("iteration_callback", C.c_void_p), ("iteration_callback_user", C.c_void_p),
    ]


class ba_iteration_summary(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_successful", C.c_int32), ("linear_solver_iterations", C.c_int32),
                ("cost", C.c_double), ("cost_change", C.c_double), ("trust_region_radius", C.c_double),
                ("cumulative_time_in_seconds", C.c_double)]


ITERATION_CALLBACK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ba_iteration_summary))
CALLBACK_CONTINUE, CALLBACK_TERMINATE, CALLBACK_ABORT = 0, 1, 2


class ba_result(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int32), ("num_residuals", C.c_int32),
        ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
        ("num_effective_parameters", C.c_int32), ("total_linear_iterations", C.c_int64),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("lm_seconds", C.c_double),
        ("num_logged", C.c_int32),
        ("log_cost", C.c_void_p), ("log_radius", C.c_void_p), ("log_linear_iters", C.c_void_p),
        ("linear_solver_used", C.c_int32), ("factor_seconds", C.c_double), ("setup_seconds", C.c_double),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


class ba_comm(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world_size", C.c_int32), ("allreduce", ALLREDUCE_FN),
                ("user", C.c_void_p), ("rccl_comm", C.c_void_p), ("sharding", C.c_int32)]


SHARD_BY_IMAGE, SHARD_BY_POINT = 0, 1
SOLVER_ITERATIVE_SCHUR, SOLVER_DENSE_SCHUR, SOLVER_AUTO, SOLVER_SPARSE_SCHUR = 0, 1, 2, 3
OPERATOR_F64, OPERATOR_F32 = 0, 1
POSE_ROT_CONST = 4  # ba_problem.pose_fixed_t: + 4 = the rotation of the pose block is held (colmap_amd_ba.h)


class Communicator:
    """Sum-over-ranks transport for ba_solve_sharded: `backend="callback"` routes the all-reduce
    through torch.distributed on host memory (gloo: CPU-testable, also works with several ranks on
    one GPU); `backend="rccl"` creates an RCCL communicator inside the library (one GPU per rank,
    all-reduce on the solver's stream over xGMI)."""

    def __init__(self, backend: str = "callback", gpu_index: int = -1, sharding: int = SHARD_BY_IMAGE):
        import torch.distributed as dist
        self.sharding = int(sharding)  # SHARD_BY_IMAGE (BASELINE.json) or SHARD_BY_POINT (less traffic)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.backend = backend
        self._rccl = None
        self.calls = 0

        def _cb(user, buf, n):
            try:
                import numpy as np
                import torch
                arr = np.ctypeslib.as_array(buf, shape=(n,))
                t = torch.from_numpy(arr)
                if self.world_size > 1:
                    dist.all_reduce(t)  # in place on the caller's buffer
                self.calls += 1
                return 0
            except Exception:  # never raise through the C frame
                import traceback
                traceback.print_exc()
                return 1

        self._cb = ALLREDUCE_FN(_cb)
        if backend == "rccl":
            L = lib()
            ident = C.create_string_buffer(128)
            if self.rank == 0 and L.ba_rccl_unique_id(ident) != 0:
                raise RuntimeError(L.ba_last_error().decode())
            if self.world_size > 1:
                obj = [ident.raw]
                dist.broadcast_object_list(obj, src=0)
                ident = C.create_string_buffer(obj[0], 128)
            h = C.c_void_p()
            if L.ba_rccl_comm_create(ident, self.rank, self.world_size, gpu_index, C.byref(h)) != 0:
                raise RuntimeError(L.ba_last_error().decode())
            self._rccl = h
        elif backend != "callback":
            raise ValueError(backend)

    def to_c(self) -> ba_comm:
        return ba_comm(self.rank, self.world_size, self._cb, None, self._rccl, self.sharding)

    def close(self):
        if self._rccl:
            lib().ba_rccl_comm_destroy(self._rccl)
            self._rccl = None


def marshal_problem(fp: FlatProblem) -> ba_problem:
    p = ba_problem()
    p.num_poses, p.num_cams, p.num_points = len(fp.poses), len(fp.cams), len(fp.points)
    p.num_obs = len(fp.obs_pose)
    for name in ("poses", "cams", "cam_model", "points", "obs_pose", "obs_cam", "obs_point", "obs_xy",
                 "pose_const", "pose_fixed_t", "cam_const", "point_const"):
        a = getattr(fp, name)
        assert a.flags["C_CONTIGUOUS"], name
        setattr(p, name, a.ctypes.data)
    if fp.sensors is not None and len(fp.sensors):
        assert fp.sensors.flags["C_CONTIGUOUS"] and fp.obs_sensor.flags["C_CONTIGUOUS"]
        assert fp.sensors.dtype == np.float64 and fp.obs_sensor.dtype == np.int32
        p.num_sensors = len(fp.sensors)
        p.sensors = fp.sensors.ctypes.data
        p.obs_sensor = fp.obs_sensor.ctypes.data
        if fp.sensor_const is not None and not fp.sensor_const.all():
            assert fp.sensor_const.dtype == np.uint8 and fp.sensor_const.flags["C_CONTIGUOUS"]
            p.sensor_const = fp.sensor_const.ctypes.data
    if fp.prior_pose is not None and len(fp.prior_pose):
        n = len(fp.prior_pose)
        if fp.prior_sensor is None:
            fp.prior_sensor = np.full(n, -1, np.int32)
        for name, dt, shape in (("prior_pose", np.int32, (n,)), ("prior_sensor", np.int32, (n,)),
                                ("prior_position", np.float64, (n, 3)), ("prior_sqrt_info", np.float64, (n, 3, 3))):
            a = getattr(fp, name)
            assert a.dtype == dt and a.shape == shape and a.flags["C_CONTIGUOUS"], name
            setattr(p, name, a.ctypes.data)
        p.num_priors = n
        p.prior_loss_type = int(fp.prior_loss_type)
        p.prior_loss_scale = float(fp.prior_loss_scale)
    return p


def marshal_options(so: SolverOptions, max_log: int = 0, num_threads: int = 0) -> ba_options:
    o = ba_options()
    for name, _ in ba_options._fields_:
        if name in ("num_threads", "max_log", "iteration_callback", "iteration_callback_user"):
            continue
        setattr(o, name, getattr(so, name))
    if so.iteration_callback is not None:
        fn = so.iteration_callback

        def _trampoline(_user, summary):
            try:
                return int(fn(summary.contents))
            except Exception:  # an exception cannot cross the C frame: abort the solve instead
                return CALLBACK_ABORT
        cfn = ITERATION_CALLBACK_FN(_trampoline)
        o._iteration_callback_keepalive = cfn   # the struct must outlive the solve together with the thunk
        o.iteration_callback = C.cast(cfn, C.c_void_p).value
    o.loss_type = int(so.loss_type)
    o.jacobi_scaling = 1 if so.jacobi_scaling else 0
    o.num_threads = num_threads
    o.max_log = max_log
    return o


BA_ABI_VERSION = 4  # include/colmap_amd_ba.h: COLMAP_AMD_BA_ABI_VERSION (the ctypes mirrors below follow that layout)


def _check_abi(L):
    """The struct mirrors in this module must match the library's header: a library built from another
    header would read / write past the end of ba_options / ba_result."""
    L.ba_abi_version.restype = C.c_int32
    v = int(L.ba_abi_version())
    if v != BA_ABI_VERSION:
        raise RuntimeError(f"libcolmap_amd.so speaks BA ABI version {v}, this module {BA_ABI_VERSION}: rebuild the library")


def solve_flat(fp: FlatProblem, so: Optional[SolverOptions] = None, gpu_index: int = -1,
               max_log: int = 256, solve_fn=None, comm: Optional[Communicator] = None,
               num_threads: int = 0) -> BundleAdjustmentSummary:
    """ba_solve on a flat problem (in place). `solve_fn` lets the tests route the identical
    marshalled structs to the oracle library instead."""
    so = so or SolverOptions()
    p = marshal_problem(fp)
    o = marshal_options(so, max_log=max_log, num_threads=num_threads)   # (threads: the CPU checker's; unused on the GPU)
    r = ba_result()
    log_cost = np.zeros(max(max_log, 1))
    log_radius = np.zeros(max(max_log, 1))
    log_lin = np.zeros(max(max_log, 1), np.int32)
    r.log_cost, r.log_radius, r.log_linear_iters = log_cost.ctypes.data, log_radius.ctypes.data, log_lin.ctypes.data
    if solve_fn is None:
        L = lib()
        _check_abi(L)   # before either entry point: both take the same ctypes mirrors
        if comm is not None:
            cc = comm.to_c()
            rc = L.ba_solve_sharded(C.byref(p), C.byref(o), C.c_int32(gpu_index), C.byref(cc), C.byref(r))
        else:
            rc = L.ba_solve(C.byref(p), C.byref(o), C.c_int32(gpu_index), C.byref(r))
        if rc != 0:
            raise RuntimeError(L.ba_last_error().decode())
    else:
        rc = solve_fn(C.byref(p), C.byref(o), C.byref(r))
        if rc != 0:
            raise RuntimeError(f"oracle failed: {rc}")
    return BundleAdjustmentSummary(
        termination_type=BundleAdjustmentTerminationType(r.termination_type), num_residuals=r.num_residuals,
        num_iterations=r.num_iterations, num_successful_steps=r.num_successful_steps,
        num_effective_parameters=r.num_effective_parameters,
        total_linear_iterations=r.total_linear_iterations, initial_cost=r.initial_cost,
        final_cost=r.final_cost, lm_seconds=r.lm_seconds, log_cost=log_cost[: r.num_logged].copy(),
        log_linear_iters=log_lin[: r.num_logged].copy(), linear_solver_used=int(r.linear_solver_used),
        factor_seconds=float(r.factor_seconds), setup_seconds=float(r.setup_seconds))


def num_camera_parameters(fp: FlatProblem) -> int:
    """Size n_c of the reduced camera system (the matrix the exact tiers factor), by the library's own rule
    (ba_kernels.hip: tangent layout): blocks that at least one ACTIVE observation uses (an observation is active when any
    of its blocks is variable); a pose block has 3 rotation dimensions unless its rotation is held (pose_fixed_t >=
    POSE_ROT_CONST) plus 3 translation dimensions, 2 when one translation coordinate is held; an intrinsics block its
    variable entries; a variable sensor_from_rig block 6."""
    pose_const = np.asarray(fp.pose_const) != 0
    cam_nvar = (np.asarray(fp.cam_const) == 0).sum(1)
    pt_const = np.asarray(fp.point_const) != 0
    op, oc, ox = np.asarray(fp.obs_pose), np.asarray(fp.obs_cam), np.asarray(fp.obs_point)
    sens_var_obs = np.zeros(len(op), bool)
    sens_var = None
    if fp.sensors is not None and fp.obs_sensor is not None and fp.sensor_const is not None:
        sens_var = np.asarray(fp.sensor_const) == 0
        osn = np.asarray(fp.obs_sensor)
        sens_var_obs = (osn >= 0) & sens_var[np.maximum(osn, 0)]
    active = ~(pose_const[op] & (cam_nvar[oc] == 0) & pt_const[ox] & ~sens_var_obs)
    pose_used = np.zeros(len(pose_const), bool)
    pose_used[op[active]] = True
    cam_used = np.zeros(len(cam_nvar), bool)
    cam_used[oc[active]] = True
    pf = np.asarray(fp.pose_fixed_t).astype(np.int64)
    dim = np.where(pf >= POSE_ROT_CONST, 0, 3) + np.where((pf >= 0) & ((pf & 3) != 3), 2, 3)
    n = int(dim[pose_used & ~pose_const].sum()) + int(cam_nvar[cam_used].sum())
    if sens_var is not None:
        used = np.zeros(len(sens_var), bool)
        used[osn[active & (osn >= 0)]] = True
        n += 6 * int(np.count_nonzero(sens_var & used))
    return n


def shard_num_observations(fp: FlatProblem, rank: int, world_size: int, sharding: int = SHARD_BY_IMAGE) -> int:
    """Observations rank `rank` works on under image / point sharding (host-only, no GPU needed)."""
    L = lib()
    fn = L.ba_shard_num_observations_by_point if sharding == SHARD_BY_POINT else L.ba_shard_num_observations
    fn.restype = C.c_int64
    p = marshal_problem(fp)
    return int(fn(C.byref(p), C.c_int32(rank), C.c_int32(world_size)))


def resolve_linear_solver(num_images: int, max_dense: int = 200, max_sparse: int = 4000) -> int:
    """The AUTO rule (CreateSolverOptions, bundle_adjustment_ceres.cc:203-213): DENSE_SCHUR up to `max_dense` images,
    SPARSE_SCHUR up to `max_sparse`, ITERATIVE_SCHUR beyond."""
    if num_images <= max_dense:
        return SOLVER_DENSE_SCHUR
    return SOLVER_SPARSE_SCHUR if num_images <= max_sparse else SOLVER_ITERATIVE_SCHUR


class BundleAdjuster:
    """colmap::BundleAdjuster (bundle_adjustment.h:212-228) for backend MI355X."""

    def __init__(self, options: BundleAdjustmentOptions, config: BundleAdjustmentConfig,
                 reconstruction: scene.Reconstruction, solve_fn=None):
        assert options.Check()
        self.options_ = options
        self.config_ = config
        self.reconstruction_ = reconstruction
        self.problem_ = flatten(options, config, reconstruction)
        self._solve_fn = solve_fn

    def Options(self):
        return self.options_

    def Config(self):
        return self.config_

    def Solve(self) -> BundleAdjustmentSummary:
        fp = self.problem_
        if len(fp.obs_pose) == 0:
            return BundleAdjustmentSummary()  # zero residuals -> default summary (:667-669)
        gpu = [int(x) for x in str(self.options_.gpu_index).split(",") if x.strip()]
        so = self.options_.solver_options
        if so.linear_solver_type == SOLVER_AUTO:
            # CreateSolverOptions' rule on config.NumImages() (bundle_adjustment_ceres.cc:131,203-213) with this
            # backend's measured GPU thresholds (the reference's own GPU pair: bundle_adjustment_ceres.h:70-71) --
            # resolved here, where the image count is known: the flat C interface only sees pose blocks (a rig frame
            # with several sensors is one block)
            n_img = self.config_.NumImages()
            tier = resolve_linear_solver(n_img, self.options_.max_num_images_direct_dense_gpu_solver,
                                         self.options_.max_num_images_direct_sparse_gpu_solver)
            so = dataclasses.replace(so, linear_solver_type=tier)
        summary = solve_flat(fp, so, gpu[0] if gpu else -1, solve_fn=self._solve_fn)
        self.linear_solver_requested_, self.linear_solver_used_ = so.linear_solver_type, summary.linear_solver_used
        if summary.num_residuals == 0:
            return BundleAdjustmentSummary()
        # WriteResultsToReconstruction: only variable blocks (bundle_adjustment_caspar.cc:767-801)
        rec = self.reconstruction_
        for i, (kind, ident) in enumerate(fp.pose_ids):
            if fp.pose_const[i]:
                continue
            if kind == "frame":
                rec.frames[ident].rig_from_world = fp.poses[i].copy()
            else:
                rec.images[ident].cam_from_world = fp.poses[i].copy()
        if fp.sensors is not None and fp.sensor_const is not None:
            for k, cid in enumerate(fp.sensor_ids):
                if not fp.sensor_const[k]:
                    for rig in rec.rigs.values():
                        if cid in rig.sensors:
                            rig.sensors[cid] = fp.sensors[k].copy()
        rec.UpdateCamFromWorld()
        for k, cid in enumerate(fp.cam_ids):
            n = len(rec.cameras[cid].params)
            if not fp.cam_const[k, :n].all():
                rec.cameras[cid].params = fp.cams[k, :n].copy()
        for j, pid in enumerate(fp.point_ids):
            if not fp.point_const[j]:
                rec.points3D[pid].xyz = fp.points[j].copy()
        return summary


@dataclass
class PosePrior:
    """colmap::PosePrior (geometry/pose_prior.h:43-77), the fields the adjuster reads: the image the prior
    belongs to (corr_data_id of a camera sensor), its position in the world and the position covariance."""
    image_id: int
    position: np.ndarray = field(default_factory=lambda: np.full(3, np.nan))
    position_covariance: Optional[np.ndarray] = None

    def HasPosition(self) -> bool:
        return bool(np.isfinite(self.position).all())

    def HasPositionCov(self) -> bool:
        return self.position_covariance is not None and bool(np.isfinite(self.position_covariance).all())


@dataclass
class PosePriorBundleAdjustmentOptions:
    """PosePriorBundleAdjustmentOptions + CeresPosePriorBundleAdjustmentOptions (bundle_adjustment.h:252-262,
    bundle_adjustment_ceres.h:102-112)."""
    prior_position_fallback_stddev: float = 1.0
    prior_position_loss_function_type: LossFunctionType = LossFunctionType.TRIVIAL
    prior_position_loss_scale: float = float(np.sqrt(7.815))  # sqrt(kChiSquare95ThreeDof)
    alignment_ransac_options: Optional["RANSACOptions"] = None  # bundle_adjustment.h:258 (None: defaults)

    def Check(self) -> bool:
        return self.prior_position_fallback_stddev > 0 and self.prior_position_loss_scale > 0


def align_to_positions(src: np.ndarray, dst: np.ndarray):
    """Least-squares similarity (scale, R, t) with dst ~ scale R src + t (Umeyama). The reference estimates
    the same transform inside RANSAC (AlignReconstructionToPosePriors); here all correspondences are used.
    Returns None when the points are degenerate (fewer than 3, or collinear)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    if len(src) < 3:
        return None
    ms, md = src.mean(0), dst.mean(0)
    a, b = src - ms, dst - md
    var = (a * a).sum() / len(src)
    if var < 1e-24:
        return None
    U, S, Vt = np.linalg.svd(b.T @ a / len(src))
    if S[1] < 1e-12 * max(S[0], 1e-300):
        return None  # collinear: the rotation about the line is not determined
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    scale = float(np.trace(np.diag(S) @ D) / var)
    return scale, R, md - scale * R @ ms


K_CHI_SQUARE_95_THREE_DOF = 7.814727903251179  # math/math.h:47


@dataclass
class RANSACOptions:
    """colmap::RANSACOptions (optim/ransac.h:50-83), the fields the alignment uses."""
    max_error: float = 0.0          # <= 0: from the priors' covariances (alignment.cc:284-294)
    min_inlier_ratio: float = 0.1
    confidence: float = 0.99
    dyn_num_trials_multiplier: float = 3.0
    min_num_trials: int = 0
    max_num_trials: int = 10000
    random_seed: int = 0            # (the reference's -1 = nondeterministic; a fixed stream here)


def _lcg(state: int) -> int:
    return (state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF


def align_to_positions_robust(src: np.ndarray, dst: np.ndarray, max_error: float, opt: Optional[RANSACOptions] = None):
    """AlignReconstructionToPosePriors' estimator (estimators/alignment.cc:240-299 -> EstimateSim3dRobust):
    RANSAC over 3-point similarity hypotheses with the position error |dst - (s R src + t)| <= max_error as the
    inlier test, a local refit on the inliers of every improving hypothesis, and the final least-squares
    similarity over the best inlier set. One outlier prior no longer skews the frame every prior residual is
    expressed in. Deterministic sample stream (64-bit LCG), mirrored by AlignToPositionsRobust in
    include/colmap_amd/bundle_adjustment.hpp. Returns (scale, R, t) or None."""
    opt = opt or RANSACOptions()
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    n = len(src)
    if n < 3 or not max_error > 0:
        return None

    def inliers_of(tf):
        s, R, t = tf
        return np.linalg.norm(dst - (s * src @ R.T + t), axis=1) <= max_error

    best_inl, best_count = None, 0
    state = _lcg(0x9E3779B97F4A7C15 ^ (opt.random_seed & 0xFFFFFFFF))
    needed = float(opt.max_num_trials)
    trial = 0
    while trial < opt.max_num_trials and (trial < needed or trial < opt.min_num_trials):
        trial += 1
        idx = []
        while len(idx) < 3:
            state = _lcg(state)
            k = int((state >> 33) % n)
            if k not in idx:
                idx.append(k)
        tf = align_to_positions(src[idx], dst[idx])
        if tf is None:
            continue
        inl = inliers_of(tf)
        count = int(inl.sum())
        if count <= best_count:
            continue
        for _ in range(4):  # local optimisation: refit on the support while it grows
            tf2 = align_to_positions(src[inl], dst[inl]) if count >= 3 else None
            if tf2 is None:
                break
            inl2 = inliers_of(tf2)
            if int(inl2.sum()) <= count:
                break
            inl, count = inl2, int(inl2.sum())
        best_inl, best_count = inl, count
        w = min(max(best_count / n, opt.min_inlier_ratio), 1.0 - 1e-12)
        needed = opt.dyn_num_trials_multiplier * np.log(1.0 - opt.confidence) / np.log(1.0 - w ** 3)
    if best_count < 3:
        return None
    return align_to_positions(src[best_inl], dst[best_inl])


class PosePriorBundleAdjuster(BundleAdjuster):
    """PosePriorBundleAdjuster (bundle_adjustment_ceres.cc:900-1085): priors without a position or for
    images outside the config are dropped; with >= 3 usable priors the reconstruction is aligned to them
    (similarity), normalised (fixed scale) and every parameterized image with a variable pose or
    sensor_from_rig block gets a covariance-weighted position residual -- the priors own the gauge; otherwise
    the two-camera gauge is fixed and the priors are not used. Solve() undoes the normalisation."""

    def __init__(self, options: BundleAdjustmentOptions, prior_options: PosePriorBundleAdjustmentOptions,
                 config: BundleAdjustmentConfig, pose_priors: Sequence[PosePrior],
                 reconstruction: scene.Reconstruction, solve_fn=None):
        assert options.Check() and prior_options.Check()
        import copy
        config = copy.deepcopy(config)
        self.prior_options_ = prior_options
        self.pose_priors_ = [p for p in pose_priors if p.HasPosition() and config.HasImage(p.image_id)]
        self.normalized_from_metric_ = np.zeros(3)
        self.use_prior_position_ = len(self.pose_priors_) >= 3 and self._align(reconstruction)
        if self.use_prior_position_:
            self.normalized_from_metric_ = reconstruction.Normalize(fixed_scale=True)
        else:
            config.FixGauge(BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD)
        super().__init__(options, config, reconstruction, solve_fn=solve_fn)
        if self.use_prior_position_:
            self._add_priors()

    def _align(self, rec: scene.Reconstruction) -> bool:
        src = np.array([rec.ProjectionCenter(p.image_id) for p in self.pose_priors_])
        dst = np.array([p.position for p in self.pose_priors_])
        ropt = getattr(self.prior_options_, "alignment_ransac_options", None) or RANSACOptions()
        max_error = ropt.max_error
        if max_error <= 0:  # alignment.cc:284-294: 95 % chi-square quantile of the median prior variance
            rms = [float(np.trace(p.position_covariance)) / 3.0 for p in self.pose_priors_
                   if p.HasPositionCov() and np.trace(p.position_covariance) > 0]
            if not rms:
                rms = [self.prior_options_.prior_position_fallback_stddev ** 2]
            max_error = float(np.sqrt(K_CHI_SQUARE_95_THREE_DOF * np.median(rms)))
        tf = align_to_positions_robust(src, dst, max_error, ropt)
        if tf is None:
            return False
        rec.Transform(*tf)
        return True

    def _add_priors(self):
        fp = self.problem_
        rows = []
        for pr in self.pose_priors_:
            slot = (fp.image_slots or {}).get(pr.image_id)
            if slot is None:  # not parameterized: no reprojection constraints (:955-961)
                continue
            pose_slot, sens_slot = slot
            const_sensor = sens_slot < 0 or bool(fp.sensor_const[sens_slot])
            if fp.pose_const[pose_slot] and const_sensor:  # (:997-1000)
                continue
            cov = (np.asarray(pr.position_covariance, np.float64) if pr.HasPositionCov()
                   else self.prior_options_.prior_position_fallback_stddev ** 2 * np.eye(3))
            L = np.linalg.cholesky(np.linalg.inv(cov))  # LeftSqrtInformation: cov^-1 = L L^T, weight = L^T
            rows.append((pose_slot, sens_slot, np.asarray(pr.position, np.float64) + self.normalized_from_metric_, L.T))
        if not rows:
            return
        fp.prior_pose = np.array([r[0] for r in rows], np.int32)
        fp.prior_sensor = np.array([r[1] for r in rows], np.int32)
        fp.prior_position = np.ascontiguousarray(np.stack([r[2] for r in rows]))
        fp.prior_sqrt_info = np.ascontiguousarray(np.stack([r[3] for r in rows]))
        fp.prior_loss_type = int(self.prior_options_.prior_position_loss_function_type)
        fp.prior_loss_scale = float(self.prior_options_.prior_position_loss_scale)

    def Solve(self) -> BundleAdjustmentSummary:
        summary = super().Solve()
        self.reconstruction_.Transform(1.0, np.eye(3), -self.normalized_from_metric_)  # Inverse(normalized_from_metric)
        return summary


def CreatePosePriorBundleAdjuster(options: BundleAdjustmentOptions, prior_options: PosePriorBundleAdjustmentOptions,
                                  config: BundleAdjustmentConfig, pose_priors: Sequence[PosePrior],
                                  reconstruction: scene.Reconstruction, solve_fn=None) -> PosePriorBundleAdjuster:
    """colmap::CreatePosePriorBundleAdjuster (bundle_adjustment.cc:373-395)."""
    if options.backend != BundleAdjustmentBackend.MI355X:
        raise ValueError(f"backend {options.backend!r} is not built here: only MI355X")
    return PosePriorBundleAdjuster(options, prior_options, config, pose_priors, reconstruction, solve_fn=solve_fn)


def CreateDefaultBundleAdjuster(options: BundleAdjustmentOptions, config: BundleAdjustmentConfig,
                                reconstruction: scene.Reconstruction) -> BundleAdjuster:
    """colmap::CreateDefaultBundleAdjuster (bundle_adjustment.cc:314-334): backend switch."""
    if options.backend != BundleAdjustmentBackend.MI355X:
        raise ValueError(f"backend {options.backend!r} is not built here: only MI355X "
                         "(the reference's CERES/CASPAR need Ceres/CUDA)")
    return BundleAdjuster(options, config, reconstruction)
