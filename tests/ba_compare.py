"""How two bundle-adjustment solutions of ONE problem are compared (test infrastructure).

What a solve determines is compared directly: the cost (relative), the first LM iterations, the poses and the points
(absolute). The intrinsics are compared through what they DO -- the pixel every observed point lands on, projected
through either camera from the same pose and point -- not coefficient by coefficient: the reference's own test does the
same (`ReconstructionNear`, bundle_adjustment_test.cc:344-349, compares poses and centres, never raw distortion
coefficients), and high-order coefficients that the scene does not observe (k3 / k4 of OPENCV_FISHEYE at a narrow field
of view, the rational terms of FULL_OPENCV) drift to 1e2 .. 1e8 in BOTH solvers while the projection inside the observed
region stays put.

Every bar is the larger of a base value (what well-conditioned problems reach with room to spare) and a multiple of the
measured NOISE FLOOR of that very problem: the oracle against a build of itself with the compiler's default
floating-point contraction (`oracle/libba_oracle_fast.so`, a few-ulp perturbation per operation). A problem whose
solution moves by 1e-5 under such a perturbation cannot be asked to agree to 1e-7 with anything.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

import ba_oracle

FLOOR_MARGIN = 10.0      # a bar is never tighter than this many times the measured floor
MAX_PROJECTIONS = 20000  # every observation up to this many, beyond it a fixed-stride sample of this size
# the models whose intrinsics a scene always observes (focal lengths, principal point, one radial term): compared
# coefficient by coefficient as well as through their projections
WELL_CONDITIONED_MODELS = (0, 1, 2)   # SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL


def projection_diff_px(a, b) -> float:
    """Largest distance, in pixels, between the projections of the observed points through the cameras of solution `a`
    and of solution `b` -- both from a's poses and points, so that only the intrinsics differ."""
    n = len(a.obs_pose)
    idx = np.arange(0, n, max(1, n // MAX_PROJECTIONS))
    worst = 0.0
    zero = np.zeros(2)
    for o in idx:
        k = int(a.obs_cam[o])
        model = int(a.cam_model[k])
        P = ba_oracle.NUM_PARAMS[model]
        pt, pose = a.points[a.obs_point[o]], a.poses[a.obs_pose[o]]
        s = -1 if a.obs_sensor is None else int(a.obs_sensor[o])
        if s >= 0:
            ra = ba_oracle.rig_reproj_error(model, pt, pose, a.sensors[s], a.cams[k, :P], zero, want_jac=False)[0]
            rb = ba_oracle.rig_reproj_error(model, pt, pose, a.sensors[s], b.cams[k, :P], zero, want_jac=False)[0]
        else:
            ra = ba_oracle.reproj_error(model, pt, pose, a.cams[k, :P], zero, want_jac=False)[0]
            rb = ba_oracle.reproj_error(model, pt, pose, b.cams[k, :P], zero, want_jac=False)[0]
        worst = max(worst, float(np.abs(ra - rb).max()))
    return worst


def cams_rel_diff(a, b) -> float:
    """Largest relative difference between the variable intrinsics of the well-conditioned camera models."""
    worst = 0.0
    for k in range(len(a.cam_model)):
        model = int(a.cam_model[k])
        if model not in WELL_CONDITIONED_MODELS:
            continue
        P = ba_oracle.NUM_PARAMS[model]
        ca, cb = np.asarray(a.cams[k, :P], float), np.asarray(b.cams[k, :P], float)
        scale = np.maximum(np.abs(ca), 1.0 if model != 2 else np.array([1.0, 1.0, 1.0, 1e-2]))
        worst = max(worst, float((np.abs(ca - cb) / scale).max()))
    return worst


@dataclass
class Diff:
    cost_rel: float
    traj_rel: float
    points: float
    poses: float
    sensors: float
    proj_px: float
    cams_rel: float = 0.0


    def scaled(self, f):
        return Diff(*(f * getattr(self, k) for k in self.__dataclass_fields__))


# No floor, however large it measures, loosens a bar beyond this: a problem whose own perturbed solve moves further
# than that is reported (the comparison fails) instead of passing with an arbitrary disagreement.
CEILING = Diff(cost_rel=1e-4, traj_rel=1e-4, points=1e-2, poses=1e-2, sensors=1e-2, proj_px=1e-2, cams_rel=1e-3)


def diff(a, ra, b, rb, n_traj=4) -> Diff:
    n = min(n_traj, len(ra.log_cost), len(rb.log_cost))
    la, lb = np.asarray(ra.log_cost[:n], float), np.asarray(rb.log_cost[:n], float)
    sens = 0.0
    if a.sensors is not None and len(a.sensors):
        sens = float(np.abs(a.sensors - b.sensors).max())
    return Diff(cost_rel=abs(ra.final_cost - rb.final_cost) / max(abs(ra.final_cost), 1e-300),
                traj_rel=float((np.abs(la - lb) / np.maximum(np.abs(la), 1e-300)).max()) if n else 0.0,
                points=float(np.abs(a.points - b.points).max()), poses=float(np.abs(a.poses - b.poses).max()),
                sensors=sens, proj_px=projection_diff_px(a, b), cams_rel=cams_rel_diff(a, b))


def assert_solutions_close(a, want, b, got, floor=None, cost_rtol=1e-8, param_atol=1e-6, traj_rtol=1e-7,
                           proj_atol=1e-5, cams_rtol=1e-6):
    """`(a, want)` = the oracle's solution and summary, `(b, got)` = the HIP solve's; `floor` = a callable returning
    diff(oracle, perturbed oracle) of the same problem, or None. The base bars are tried first; only a comparison that
    misses one of them pays for the perturbed oracle solve, and is then held to max(base, FLOOR_MARGIN x floor). Counts
    are exact; so is the termination type, except for two solves stalled on the same plateau (below)."""
    assert got.num_residuals == want.num_residuals
    assert got.num_effective_parameters == want.num_effective_parameters
    assert abs(got.initial_cost - want.initial_cost) <= 1e-12 * want.initial_cost
    d = diff(a, want, b, got)
    base = Diff(cost_rel=cost_rtol, traj_rel=traj_rtol, points=param_atol, poses=param_atol, sensors=param_atol,
                proj_px=proj_atol, cams_rel=cams_rtol)

    def misses(bars):
        return [f"{k}: {getattr(d, k):.3e} > {getattr(bars, k):.3e}" for k in d.__dataclass_fields__
                if not getattr(d, k) <= getattr(bars, k)]

    bars, f = base, None
    if misses(base) and floor is not None:
        f = (floor() if callable(floor) else floor).scaled(FLOOR_MARGIN)
        bars = Diff(*(max(getattr(base, k), min(getattr(f, k), getattr(CEILING, k))) for k in d.__dataclass_fields__))
    bad = misses(bars)
    assert not bad, "HIP vs oracle beyond the bar (floor x %g = %s): %s" % (FLOOR_MARGIN, f, "; ".join(bad))
    # the termination type is compared last: two solvers at the noise floor of an ill-conditioned problem may stop a
    # few LM iterations apart, but not with a different verdict -- with ONE exception, which is a property of the
    # problem and not of either solver: both have STALLED (the last accepted costs agree to 1e-12 and no longer move,
    # every further step is accepted or rejected by rounding noise) and the trust region of one collapses below
    # min_trust_region_radius (CONVERGENCE) a few coin flips before max_num_iterations cuts the other off
    # (NO_CONVERGENCE). Measured on the RAD_TAN_THIN_PRISM_FISHEYE case at gradient_tolerance 1e-10: oracle 43
    # iterations, the same kernels on the CPU stand-in 54, on the GPU > 60 = the limit.
    if got.termination_type != want.termination_type:
        assert _stalled_pair(want, got), (want.termination_type, got.termination_type)
    return d, bars


def _stalled(s, tail=6, rtol=1e-12):
    """The solve ended on a plateau: its last `tail` logged costs (accepted or not, the log holds the current cost)
    agree to `rtol`."""
    n = int(s.num_iterations)
    c = np.asarray(s.log_cost[:n + 1], float)[-tail:]
    return len(c) == tail and float(c.max() - c.min()) <= rtol * abs(float(c[-1]))


def _stalled_pair(want, got):
    types = {int(want.termination_type), int(got.termination_type)}
    return (types == {0, 1}  # CONVERGENCE / NO_CONVERGENCE
            and _stalled(want) and _stalled(got)
            and abs(want.final_cost - got.final_cost) <= 1e-12 * abs(want.final_cost))
