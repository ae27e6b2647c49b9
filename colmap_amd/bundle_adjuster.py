"""`colmap bundle_adjuster` with the MI355X backend (reference exe/sfm.cc:175-206,
controllers/bundle_adjustment.cc:61-106, option names controllers/option_manager.cc:510-570):

    python -m colmap_amd.bundle_adjuster --input_path SPARSE --output_path OUT \\
        [--BundleAdjustment.refine_principal_point 0] [--BundleAdjustmentCeres.max_num_iterations 100] ...

reads a sparse model (binary or text; rigs.bin / frames.bin when present), removes observations
with negative depth, adjusts all registered images with the two-camera gauge, updates the point
errors and writes the model back in the binary format.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Optional

import numpy as np

from . import estimators as est
from . import scene
from . import workspace as W

_CAMERA = 0  # SensorType::CAMERA


def _xyzw_t(wxyz_t: np.ndarray) -> np.ndarray:
    return np.array([wxyz_t[1], wxyz_t[2], wxyz_t[3], wxyz_t[0], wxyz_t[4], wxyz_t[5], wxyz_t[6]], np.float64)


def _wxyz_t(xyzw_t: np.ndarray) -> np.ndarray:
    return np.array([xyzw_t[3], xyzw_t[0], xyzw_t[1], xyzw_t[2], xyzw_t[4], xyzw_t[5], xyzw_t[6]], np.float64)


def _rigid_inverse(p: np.ndarray) -> np.ndarray:
    """Inverse(Rigid3d) on xyzw + t params (geometry/rigid3.h:97-103)."""
    q = np.array([-p[0], -p[1], -p[2], p[3]], np.float64)
    return np.concatenate([q, -(scene.quat_to_rot(q) @ p[4:])])


def _sensor_from_rig_of(sm: W.SparseModel, rig_id: int, camera_id: int) -> Optional[np.ndarray]:
    """xyzw + t sensor_from_rig of a camera in a rig of the file model; None for the reference sensor
    (identity) or when the rig does not list the camera."""
    rig = sm.rigs.get(rig_id)
    if rig is None or rig.ref_sensor == (_CAMERA, camera_id):
        return None
    pose = rig.sensors.get((_CAMERA, camera_id))
    return None if pose is None else _xyzw_t(pose)


def reconstruction_from_sparse_model(sm: W.SparseModel) -> scene.Reconstruction:
    """The slice of colmap::Reconstruction bundle adjustment touches. A legacy model (no rigs /
    frames files) gets one trivial frame per image (CreateOneRigPerCamera / CreateFrameForImage,
    reconstruction_io_binary.cc:176-214); rigs with a single sensor stay trivial frames too."""
    rec = scene.Reconstruction()
    for cid, c in sm.cameras.items():
        rec.cameras[cid] = scene.Camera(cid, c.model_id, c.width, c.height, np.array(c.params, np.float64))
    frame_of_image = {}
    for rid, rig in sm.rigs.items():
        cam_sensors = {sid: pose for sid, pose in rig.sensors.items() if sid[0] == _CAMERA}
        if rig.ref_sensor is None or rig.ref_sensor[0] != _CAMERA or not cam_sensors:
            continue  # single-camera rig (or an IMU reference, not adjusted here): trivial frames
        r = scene.Rig(rid, rig.ref_sensor[1])
        for sid, pose in cam_sensors.items():
            if pose is None:
                raise ValueError(f"rig {rid}: sensor {sid} has no sensor_from_rig")
            r.sensors[sid[1]] = _xyzw_t(pose)
        rec.rigs[rid] = r
    for fid, fr in sm.frames.items():
        if fr.rig_id not in rec.rigs:
            continue
        f = scene.Frame(fid, fr.rig_id, _xyzw_t(fr.rig_from_world))
        for (stype, _sid, data_id) in sorted(fr.data_ids):
            if stype == _CAMERA and data_id in sm.images:
                f.image_ids.append(int(data_id))
                frame_of_image[int(data_id)] = fid
        rec.frames[fid] = f
    # With rigs / frames files the frame poses are authoritative: the reference ignores the pose
    # stored in images.bin (reconstruction_io_binary.cc:176-214). Images of trivial (single-camera)
    # rigs therefore take cam_from_world = sensor_from_rig * rig_from_world from their frame.
    file_frame_pose = {}
    for fid, fr in sm.frames.items():
        if fr.rig_id in rec.rigs:
            continue
        for (stype, sid, data_id) in fr.data_ids:
            if stype == _CAMERA and data_id in sm.images:
                rfw = _xyzw_t(fr.rig_from_world)
                sfr = _sensor_from_rig_of(sm, fr.rig_id, int(sid))
                file_frame_pose[int(data_id)] = rfw if sfr is None else scene.rigid_compose(sfr, rfw)
    for iid, im in sm.images.items():
        pose = np.concatenate([[im.qvec[1], im.qvec[2], im.qvec[3], im.qvec[0]], im.tvec]).astype(np.float64)
        pose = file_frame_pose.get(iid, pose)
        img = scene.Image(iid, im.camera_id, pose, frame_id_=frame_of_image.get(iid))
        img.points2D = [scene.Point2D(np.array(xy, np.float64), int(pid) if pid >= 0 and int(pid) in sm.points3D else -1)
                        for xy, pid in zip(im.xys, im.point3D_ids)]
        rec.images[iid] = img
    for pid, p in sm.points3D.items():
        rec.points3D[pid] = scene.Point3D(np.array(p.xyz, np.float64), [(int(a), int(b)) for a, b in p.track])
    return rec


def sparse_model_from_reconstruction(rec: scene.Reconstruction, names: Optional[dict] = None) -> W.SparseModel:
    """The inverse of reconstruction_from_sparse_model (used to put synthetic datasets on disk)."""
    sm = W.SparseModel()
    for cid, c in rec.cameras.items():
        sm.cameras[cid] = W.SparseCamera(cid, c.model_id, c.width, c.height, np.array(c.params, np.float64))
    for iid, img in rec.images.items():
        p = _wxyz_t(img.cam_from_world)
        xys = np.array([q.xy for q in img.points2D], np.float64).reshape(-1, 2)
        ids = np.array([q.point3D_id for q in img.points2D], np.int64)
        sm.images[iid] = W.SparseImage(iid, p[:4], p[4:], img.camera_id, (names or {}).get(iid, f"image{iid:05d}.png"),
                                       xys, ids)
    for pid, pt in rec.points3D.items():
        sm.points3D[pid] = W.SparsePoint3D(pid, np.array(pt.xyz, np.float64), (0, 0, 0), 0.0, list(pt.track))
    for rid, rig in rec.rigs.items():
        sm.rigs[rid] = W.SparseRig(rid, (_CAMERA, rig.ref_camera_id),
                                   {(_CAMERA, cid): _wxyz_t(pose) for cid, pose in rig.sensors.items()})
    for fid, fr in rec.frames.items():
        sm.frames[fid] = W.SparseFrame(fid, fr.rig_id, _wxyz_t(fr.rig_from_world),
                                       [(_CAMERA, rec.images[i].camera_id, i) for i in fr.image_ids])
    return sm


def filter_observations_with_negative_depth(rec: scene.Reconstruction) -> int:
    """ObservationManager::FilterObservationsWithNegativeDepth (sfm/observation_manager.cc:409-433):
    z = cam_from_world.row(2) . [X; 1] >= epsilon (HasPointPositiveDepth, scene/projection.cc:137-141)."""
    n = 0
    eps = np.finfo(np.float64).eps
    for iid in rec.RegImageIds():
        img = rec.images[iid]
        R2 = scene.quat_to_rot(img.cam_from_world[:4])[2]
        tz = img.cam_from_world[6]
        for idx, p2 in enumerate(img.points2D):
            if p2.HasPoint3D() and p2.point3D_id in rec.points3D:
                if not (R2 @ rec.points3D[p2.point3D_id].xyz + tz >= eps):
                    rec.DeleteObservation(iid, idx)
                    n += 1
    return n


def point3D_errors(rec: scene.Reconstruction) -> dict:
    """Reconstruction::UpdatePoint3DErrors (scene/reconstruction.cc:959-975): mean reprojection error
    (pixels) over the track; a point behind a camera contributes sqrt(DBL_MAX)."""
    out = {}
    for pid, pt in rec.points3D.items():
        if not pt.track:
            out[pid] = 0.0
            continue
        e = 0.0
        for im, idx in pt.track:
            img = rec.images[im]
            cam = rec.cameras[img.camera_id]
            pc = scene.quat_to_rot(img.cam_from_world[:4]) @ pt.xyz + img.cam_from_world[4:]
            if pc[2] < np.finfo(np.float64).eps:
                e += np.sqrt(np.finfo(np.float64).max)
                continue
            xy = scene.img_from_cam(cam.model_id, cam.params, pc[None])[0]
            e += float(np.linalg.norm(xy - img.points2D[idx].xy))
        out[pid] = e / len(pt.track)
    return out


def update_sparse_model(sm: W.SparseModel, rec: scene.Reconstruction):
    """Reconstruction -> files: poses, intrinsics, points, tracks (observations may have been
    deleted), point errors."""
    for cid, c in rec.cameras.items():
        sm.cameras[cid].params = np.array(c.params, np.float64)
    for iid, img in rec.images.items():
        im = sm.images[iid]
        p = _wxyz_t(img.cam_from_world)
        im.qvec, im.tvec = p[:4], p[4:]
        im.point3D_ids = np.array([q.point3D_id for q in img.points2D], np.int64)
    for fid, fr in rec.frames.items():
        sm.frames[fid].rig_from_world = _wxyz_t(fr.rig_from_world)
    # frames of trivial (single-camera) rigs are not in rec.frames: their rig_from_world follows the
    # adjusted image pose -- rig_from_world = inverse(sensor_from_rig) * cam_from_world -- because a
    # reader of the output takes poses from frames.bin, not from images.bin
    for fid, fr in sm.frames.items():
        if fid in rec.frames:
            continue
        for (stype, sid, data_id) in fr.data_ids:
            if stype == _CAMERA and int(data_id) in rec.images:
                cfw = rec.images[int(data_id)].cam_from_world
                sfr = _sensor_from_rig_of(sm, fr.rig_id, int(sid))
                rfw = cfw if sfr is None else scene.rigid_compose(_rigid_inverse(sfr), cfw)
                fr.rig_from_world = _wxyz_t(rfw)
                break
    errs = point3D_errors(rec)
    for pid in list(sm.points3D):
        if pid not in rec.points3D:
            del sm.points3D[pid]
            continue
        sm.points3D[pid].xyz = np.array(rec.points3D[pid].xyz, np.float64)
        sm.points3D[pid].track = list(rec.points3D[pid].track)
        sm.points3D[pid].error = errs[pid]


class BundleAdjustmentController:
    """colmap::BundleAdjustmentController (controllers/bundle_adjustment.cc:61-106)."""

    def __init__(self, options: est.BundleAdjustmentOptions, reconstruction: scene.Reconstruction, solve_fn=None):
        self.options_ = options
        self.reconstruction_ = reconstruction
        self._solve_fn = solve_fn  # tests route the identical problem to the oracle library
        self.summary: Optional[est.BundleAdjustmentSummary] = None
        self.num_filtered_observations = 0

    def Run(self):
        rec = self.reconstruction_
        if len(rec.images) == 0:
            print("E Need at least one registered frame.", file=sys.stderr)
            return
        self.num_filtered_observations = filter_observations_with_negative_depth(rec)
        config = est.BundleAdjustmentConfig()
        for image_id in rec.RegImageIds():
            config.AddImage(image_id)
        config.FixGauge(est.BundleAdjustmentGauge.TWO_CAMS_FROM_WORLD)
        if self._solve_fn is None:
            ba = est.CreateDefaultBundleAdjuster(self.options_, config, rec)
        else:
            ba = est.BundleAdjuster(self.options_, config, rec, solve_fn=self._solve_fn)
        self.summary = ba.Solve()


def _parse_bool(v: str) -> bool:
    if v.lower() in ("1", "true", "yes", "on"):
        return True
    if v.lower() in ("0", "false", "no", "off"):
        return False
    raise argparse.ArgumentTypeError(f"not a boolean: {v}")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="bundle_adjuster", description=__doc__.split("\n\n")[0])
    ap.add_argument("--input_path", required=True)
    ap.add_argument("--output_path", required=True)
    d = est.BundleAdjustmentOptions()
    for name in ("refine_focal_length", "refine_principal_point", "refine_extra_params", "refine_rig_from_world",
                 "refine_sensor_from_rig", "refine_points3D", "constant_rig_from_world_rotation"):
        ap.add_argument(f"--BundleAdjustment.{name}", dest=f"ba_{name}", type=_parse_bool, default=getattr(d, name))
    ap.add_argument("--BundleAdjustment.min_track_length", dest="ba_min_track_length", type=int, default=0)
    ap.add_argument("--BundleAdjustment.backend", dest="backend", default="MI355X", help="{MI355X}")
    so = est.SolverOptions()
    # the solver options COLMAP exposes for Ceres keep their names (option_manager.cc:532-570)
    for name, typ in (("max_num_iterations", int), ("max_linear_solver_iterations", int), ("function_tolerance", float),
                      ("gradient_tolerance", float), ("parameter_tolerance", float)):
        ap.add_argument(f"--BundleAdjustmentCeres.{name}", dest=f"so_{name}", type=typ, default=getattr(so, name))
    ap.add_argument("--BundleAdjustmentCeres.gpu_index", dest="gpu_index", default="-1")
    ap.add_argument("--BundleAdjustmentCeres.loss_function_type", dest="loss_type", default="TRIVIAL",
                    help="{TRIVIAL, SOFT_L1, CAUCHY, HUBER}")
    ap.add_argument("--BundleAdjustmentCeres.loss_function_scale", dest="loss_scale", type=float, default=1.0)
    return ap


def options_from_args(a) -> est.BundleAdjustmentOptions:
    if a.backend.upper() != "MI355X":
        raise SystemExit(f"BundleAdjustment.backend {a.backend}: only MI355X is built here")
    so = est.SolverOptions()
    for k, v in vars(a).items():
        if k.startswith("so_"):
            setattr(so, k[3:], v)
    so.loss_type = int(est.LossFunctionType[a.loss_type.upper()])
    so.loss_scale = a.loss_scale
    kw = {k[3:]: v for k, v in vars(a).items() if k.startswith("ba_")}
    return est.BundleAdjustmentOptions(gpu_index=a.gpu_index, solver_options=so, **kw)


def main(argv=None, solve_fn=None) -> int:
    a = build_parser().parse_args(argv)
    if not os.path.isdir(a.input_path):
        print("E `input_path` is not a directory", file=sys.stderr)
        return 1
    if not os.path.isdir(a.output_path):
        print("E `output_path` is not a directory", file=sys.stderr)
        return 1
    sm = W.read_sparse_model(a.input_path)
    rec = reconstruction_from_sparse_model(sm)
    ctl = BundleAdjustmentController(options_from_args(a), rec, solve_fn=solve_fn)
    ctl.Run()
    if ctl.summary is not None:
        print(ctl.summary.BriefReport())
    update_sparse_model(sm, rec)
    W.write_model_binary(sm, a.output_path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
