"""ctypes binding of oracle/libba_oracle.so (fp64 CPU restatement of the BA solve).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline leg).
The struct layouts are those of include/colmap_amd_ba.h, so the marshalled problem of
colmap_amd.estimators can be handed to either library unchanged.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libba_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ba_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "libba_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.bao_solve.restype = C.c_int
        _lib.bao_num_threads.restype = C.c_int
    return _lib


def solve_fn(p, o, r):
    """Drop-in for colmap_amd.estimators.solve_flat(solve_fn=...)."""
    return lib().bao_solve(p, o, r)


_lib_fast = None


def lib_fast():
    """The same source compiled with -O3 and the compiler's default contraction (oracle/Makefile): a few-ulp perturbation
    of the oracle's own arithmetic. Only tests/ba_compare.py uses it, to measure the noise floor of a comparison."""
    global _lib_fast
    if _lib_fast is None:
        path = os.path.join(_HERE, "libba_oracle_fast.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "ba_oracle.c")):
            subprocess.check_call(["make", "-C", _HERE, "libba_oracle_fast.so"], stdout=subprocess.DEVNULL)
        _lib_fast = C.CDLL(path)
        _lib_fast.bao_solve.restype = C.c_int
    return _lib_fast


def solve_fn_fast(p, o, r):
    return lib_fast().bao_solve(p, o, r)


def reproj_error(model, point, pose, params, xy, want_jac=True):
    point = np.ascontiguousarray(point, np.float64)
    pose = np.ascontiguousarray(pose, np.float64)
    prm = np.zeros(16)
    prm[: len(params)] = params
    xy = np.ascontiguousarray(xy, np.float64)
    P = NUM_PARAMS[model]
    r = np.zeros(2)
    Jpt, Jpose, Jpar = np.zeros((2, 3)), np.zeros((2, 7)), np.zeros((2, P))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().bao_reproj_error(C.c_int(model), vp(point), vp(pose), vp(prm), vp(xy), vp(r),
                           vp(Jpt) if want_jac else None, vp(Jpose) if want_jac else None,
                           vp(Jpar) if want_jac else None)
    return r, Jpt, Jpose, Jpar


NUM_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 6: 12, 10: 12, 11: 16, 17: 2, 7: 5, 8: 4, 9: 5, 12: 4, 13: 5, 14: 3, 15: 4, 16: 6}


def rig_reproj_error(model, point, rig_from_world, sensor_from_rig, params, xy, want_jac=True):
    """RigReprojErrorConstantRigCostFunctor with analytic Jacobians (w.r.t. point, rig_from_world
    and the intrinsics)."""
    point = np.ascontiguousarray(point, np.float64)
    pose = np.ascontiguousarray(rig_from_world, np.float64)
    sens = np.ascontiguousarray(sensor_from_rig, np.float64)
    prm = np.zeros(16)
    prm[: len(params)] = params
    xy = np.ascontiguousarray(xy, np.float64)
    P = NUM_PARAMS[model]
    r = np.zeros(2)
    Jpt, Jpose, Jpar = np.zeros((2, 3)), np.zeros((2, 7)), np.zeros((2, P))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().bao_rig_reproj_error(C.c_int(model), vp(point), vp(pose), vp(sens), vp(prm), vp(xy), vp(r),
                               vp(Jpt) if want_jac else None, vp(Jpose) if want_jac else None,
                               vp(Jpar) if want_jac else None)
    return r, Jpt, Jpose, Jpar


def rig_reproj_error_sensor(model, point, rig_from_world, sensor_from_rig, params, xy):
    """RigReprojErrorCostFunctor (variable sensor_from_rig): also the 2 x 7 Jacobian w.r.t. it."""
    point = np.ascontiguousarray(point, np.float64)
    pose = np.ascontiguousarray(rig_from_world, np.float64)
    sens = np.ascontiguousarray(sensor_from_rig, np.float64)
    prm = np.zeros(16)
    prm[: len(params)] = params
    xy = np.ascontiguousarray(xy, np.float64)
    P = NUM_PARAMS[model]
    r = np.zeros(2)
    Jpt, Jpose, Jpar, Jsens = np.zeros((2, 3)), np.zeros((2, 7)), np.zeros((2, P)), np.zeros((2, 7))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().bao_rig_reproj_error_sensor(C.c_int(model), vp(point), vp(pose), vp(sens), vp(prm), vp(xy), vp(r),
                                      vp(Jpt), vp(Jpose), vp(Jpar), vp(Jsens))
    return r, Jpt, Jpose, Jpar, Jsens


def loss(loss_type, scale, s):
    """ceres::LossFunction::Evaluate: (rho, rho', rho'') at s = |r|^2."""
    rho = np.zeros(3)
    lib().bao_loss(C.c_int(int(loss_type)), C.c_double(scale), C.c_double(s), rho.ctypes.data_as(C.c_void_p))
    return rho


def quat_plus(q, d):
    q = np.ascontiguousarray(q, np.float64)
    d = np.ascontiguousarray(d, np.float64)
    out = np.zeros(4)
    lib().bao_quat_plus(q.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def position_prior(position, pose, sensor=None, want_jac=True):
    """AbsolutePosePositionPriorCostFunctor / AbsoluteRigPosePositionPriorCostFunctor (unweighted):
    residual (3,), d/d pose (3, 7), d/d sensor_from_rig (3, 7) or None."""
    pos = np.ascontiguousarray(position, np.float64)
    pose = np.ascontiguousarray(pose, np.float64)
    sens = None if sensor is None else np.ascontiguousarray(sensor, np.float64)
    r = np.zeros(3)
    Jp = np.zeros((3, 7))
    Js = np.zeros((3, 7))
    dp = C.POINTER(C.c_double)
    lib().bao_position_prior(pos.ctypes.data_as(dp), pose.ctypes.data_as(dp),
                             sens.ctypes.data_as(dp) if sens is not None else None, r.ctypes.data_as(dp),
                             Jp.ctypes.data_as(dp) if want_jac else None,
                             Js.ctypes.data_as(dp) if (want_jac and sens is not None) else None)
    return r, (Jp if want_jac else None), (Js if (want_jac and sens is not None) else None)
