/*
 * ba_oracle.c -- CPU restatement (fp64) of COLMAP's bundle-adjustment solve.
 *
 * THIS FILE IS TEST INFRASTRUCTURE: the checker the HIP path is compared against
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg). The product
 * (colmap_amd/) never includes, links or calls it.
 *
 * Two layers, two pinning levels:
 *  (1) COLMAP-side arithmetic -- reprojection residual + analytic Jacobians -- follows the
 *      in-tree sources line by line and IS pinned by the reference's own tests
 *      (tests/test_ba_oracle.py restates reprojection_error_test.cc:41-72 and checks the
 *      Jacobians against finite differences like :211-325):
 *        estimators/cost_functions/reprojection_error.h:61-138   (residual, J layout)
 *        estimators/cost_functions/quaternion_utils.h:105-153    (R(q) p and d/dq)
 *        sensor/models_jacobian.h:139-321                        (SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL)
 *        sensor/models.h:281-285                                 (HasProjectableDepth)
 *  (2) The Levenberg-Marquardt / Schur / PCG arithmetic lives in ceres-solver, a
 *      third-party dependency that is NOT vendored in /root/reference (find_package(Ceres),
 *      cmake/FindDependencies.cmake:108-116; unpinned version, vcpkg.json:15-22). It is
 *      restated here from Ceres' published algorithm (trust_region_minimizer.cc,
 *      levenberg_marquardt_strategy.cc, implicit_schur_complement.cc,
 *      conjugate_gradients_solver.cc, schur_jacobi_preconditioner.cc) with the options
 *      COLMAP sets at its call sites (bundle_adjustment_ceres.cc:102-115,122-232):
 *      **iteration-trajectory parity with Ceres is unpinned**; the final solution is
 *      pinned against scipy.optimize.least_squares and the reference tests' expectations
 *      (residual counts 594 / 80, constant blocks untouched, 0.1 deg / 0.1 accuracy).
 *
 * Build: oracle/Makefile.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BAO_API __attribute__((visibility("default")))
#define BAO_CAM_STRIDE 16

/* COLMAP CameraModelId values (sensor/models.h:90-111) */
enum {
  BAO_SIMPLE_PINHOLE = 0, BAO_PINHOLE = 1, BAO_SIMPLE_RADIAL = 2, BAO_RADIAL = 3, BAO_OPENCV = 4,
  BAO_OPENCV_FISHEYE = 5, BAO_FULL_OPENCV = 6, BAO_FOV = 7, BAO_SIMPLE_RADIAL_FISHEYE = 8, BAO_RADIAL_FISHEYE = 9,
  BAO_THIN_PRISM_FISHEYE = 10, BAO_RAD_TAN_THIN_PRISM_FISHEYE = 11, BAO_SIMPLE_DIVISION = 12, BAO_DIVISION = 13, BAO_SIMPLE_FISHEYE = 14, BAO_FISHEYE = 15, BAO_EUCM = 16,
  BAO_EQUIRECTANGULAR = 17
};

typedef struct {
  int32_t num_poses, num_cams, num_points;
  int64_t num_obs;
  double* poses;       /* [num_poses][7] qx qy qz qw tx ty tz (Rigid3d::params) */
  double* cams;        /* [num_cams][12] */
  int32_t* cam_model;  /* [num_cams] */
  double* points;      /* [num_points][3] */
  int32_t* obs_pose;   /* [num_obs] */
  int32_t* obs_cam;
  int32_t* obs_point;
  double* obs_xy;      /* [num_obs][2] */
  uint8_t* pose_const;    /* [num_poses] 1: SetParameterBlockConstant */
  int8_t* pose_fixed_t;   /* [num_poses] -1 or translation coordinate held by the gauge */
  uint8_t* cam_const;     /* [num_cams][12] per-parameter constant mask (SubsetManifold) */
  uint8_t* point_const;   /* [num_points] */
  /* rigs with constant sensor_from_rig (RigReprojErrorConstantRigCostFunctor,
   * reprojection_error.h:386-417): the pose block of an observation is the frame's
   * rig_from_world, the camera sees sensor_from_rig * rig_from_world * X */
  int32_t num_sensors;
  double* sensors;        /* [num_sensors][7] or NULL */
  int32_t* obs_sensor;    /* [num_obs] index into sensors, -1: trivial (NULL: all trivial) */
  uint8_t* sensor_const;  /* [num_sensors] 1: constant; NULL: all constant. A variable sensor_from_rig is
                             a parameter block of its own (RigReprojErrorCostFunctor,
                             reprojection_error.h:344-384) and is updated in place */
  /* position priors (PosePriorBundleAdjuster::AddImagePosePriorToProblem, bundle_adjustment_ceres.cc:
   * 986-1038): residual = sqrt_info * (position + R(q)^-1 t) of the sensor_from_world pose
   * (AbsolutePosePositionPriorCostFunctor, cost_functions/pose_prior.h:76-96), or of
   * sensor_from_rig * rig_from_world (AbsoluteRigPosePositionPriorCostFunctor, :98-129) */
  int32_t num_priors;
  int32_t* prior_pose;      /* [num_priors] pose block (cam_from_world or rig_from_world) */
  int32_t* prior_sensor;    /* [num_priors] sensor_from_rig index or -1; NULL: all -1 */
  double* prior_position;   /* [num_priors][3] */
  double* prior_sqrt_info;  /* [num_priors][9] row-major left factor (cov^-1 = L L^T, stored L^T) */
  int32_t prior_loss_type;
  double prior_loss_scale;
} bao_problem;

typedef struct {
  int32_t max_num_iterations;
  int32_t max_linear_solver_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal, eta;
  int32_t max_num_consecutive_invalid_steps;
  int32_t jacobi_scaling;
  int32_t num_threads;
  int32_t max_log; /* capacity of the per-iteration log arrays in bao_result */
  /* CeresBundleAdjustmentOptions::LossFunctionType + scale (bundle_adjustment_ceres.h:42-51,
   * CreateLossFunction bundle_adjustment_ceres.cc:66-80) */
  int32_t loss_type;
  double loss_scale;
  /* 0: ITERATIVE_SCHUR + SCHUR_JACOBI (implicit Schur complement, PCG); 1: DENSE_SCHUR and 3: SPARSE_SCHUR (the
   * reduced camera system is formed explicitly and solved exactly by Cholesky); 2: the reference's rule by
   * problem size (bundle_adjustment_ceres.cc:203-213 with the CPU thresholds of bundle_adjustment_ceres.h:68-69:
   * <= 50 images dense, <= 1000 sparse, else iterative) */
  int32_t linear_solver_type;
  int32_t operator_precision; /* colmap_amd_ba.h BA_OPERATOR_*: storage option of the HIP path; the oracle is fp64 throughout */
} bao_options;

enum { BAO_LOSS_TRIVIAL = 0, BAO_LOSS_SOFT_L1 = 1, BAO_LOSS_CAUCHY = 2, BAO_LOSS_HUBER = 3 };

/* BundleAdjustmentTerminationType (estimators/bundle_adjustment.h:50-57) */
enum { BAO_CONVERGENCE = 0, BAO_NO_CONVERGENCE = 1, BAO_FAILURE = 2 };

typedef struct {
  int32_t termination_type;
  int32_t num_residuals; /* residuals touching >= 1 variable block */
  int32_t num_iterations, num_successful_steps;
  int32_t num_effective_parameters;
  int64_t total_linear_iterations;
  double initial_cost, final_cost;
  double lm_seconds; /* time inside the LM loop (linearise + solve + evaluate) */
  int32_t num_logged;
  double* log_cost;      /* [max_log] cost after each iteration */
  double* log_radius;
  int32_t* log_linear_iters;
  int32_t linear_solver_used; /* 0 iterative, 1 / 3 exact (explicit reduced camera system + Cholesky) */
  double factor_seconds;      /* unused by the oracle */
} bao_result;

/* ------------------------------------------------------------------------- */
/* Per-residual math                                                          */
/* ------------------------------------------------------------------------- */

/* QuaternionRotatePointWithJac, quaternion_utils.h:105-153 */
static void quat_rotate_jac(const double* q, const double* pt, double out[3], double* J) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  const double px = pt[0], py = pt[1], pz = pt[2];
  const double qx_py = qx * py, qx_pz = qx * pz;
  const double qy_px = qy * px, qy_pz = qy * pz;
  const double qz_px = qz * px, qz_py = qz * py;
  const double v_x_p0 = qy_pz - qz_py;
  const double v_x_p1 = qz_px - qx_pz;
  const double v_x_p2 = qx_py - qy_px;
  const double vv0 = qy * v_x_p2 - qz * v_x_p1;
  const double vv1 = qz * v_x_p0 - qx * v_x_p2;
  const double vv2 = qx * v_x_p1 - qy * v_x_p0;
  out[0] = px + 2.0 * (qw * v_x_p0 + vv0);
  out[1] = py + 2.0 * (qw * v_x_p1 + vv1);
  out[2] = pz + 2.0 * (qw * v_x_p2 + vv2);
  if (J) {
    const double qx_px = qx * px, qy_py = qy * py, qz_pz = qz * pz;
    const double qw_px = qw * px, qw_py = qw * py, qw_pz = qw * pz;
    J[0] = 2.0 * (qy_py + qz_pz);
    J[1] = 2.0 * (-2.0 * qy_px + qx_py + qw_pz);
    J[2] = 2.0 * (-2.0 * qz_px - qw_py + qx_pz);
    J[3] = 2.0 * (-qz_py + qy_pz);
    J[4] = 2.0 * (qy_px - 2.0 * qx_py - qw_pz);
    J[5] = 2.0 * (qx_px + qz_pz);
    J[6] = 2.0 * (qw_px - 2.0 * qz_py + qy_pz);
    J[7] = 2.0 * (qz_px - qx_pz);
    J[8] = 2.0 * (qz_px + qw_py - 2.0 * qx_pz);
    J[9] = 2.0 * (-qw_px + qz_py - 2.0 * qy_pz);
    J[10] = 2.0 * (qx_px + qy_py);
    J[11] = 2.0 * (-qy_px + qx_py);
  }
}

/* Eigen::Quaterniond::toRotationMatrix for xyzw storage */
static void quat_to_rot(const double* q, double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* pose_fixed_t encoding: -1 = nothing held; 0..2 = that translation coordinate held (two-camera gauge,
 * bundle_adjustment_ceres.cc:402-415); +4 = additionally the rotation is held
 * (constant_rig_from_world_rotation, :404-408,513-516): 4..6 rotation and coordinate, 7 rotation only */
static int pose_fixed_coord(int v) { return v < 0 ? -1 : ((v & 3) == 3 ? -1 : (v & 3)); }
static int pose_rot_const(int v) { return v >= 4; }
static int pose_tangent_dim(int v) { return (pose_rot_const(v) ? 0 : 3) + (pose_fixed_coord(v) >= 0 ? 2 : 3); }

static int num_params_of(int model) {
  switch (model) {
    case BAO_SIMPLE_PINHOLE: return 3;
    case BAO_PINHOLE: return 4;
    case BAO_SIMPLE_RADIAL: return 4;
    case BAO_RADIAL: return 5;
    case BAO_OPENCV: return 8;
    case BAO_OPENCV_FISHEYE: return 8;
    case BAO_SIMPLE_RADIAL_FISHEYE: return 4;
    case BAO_RADIAL_FISHEYE: return 5;
    case BAO_FOV: return 5;
    case BAO_SIMPLE_DIVISION: return 4;
    case BAO_DIVISION: return 5;
    case BAO_SIMPLE_FISHEYE: return 3;
    case BAO_FISHEYE: return 4;
    case BAO_EUCM: return 6;
    case BAO_FULL_OPENCV: return 12;
    case BAO_THIN_PRISM_FISHEYE: return 12;
    case BAO_RAD_TAN_THIN_PRISM_FISHEYE: return 16;
    case BAO_EQUIRECTANGULAR: return 2;
    default: return -1;
  }
}

/* internal::FisheyeProjectionWithJac (sensor/models_jacobian.h:51-80): equidistant projection of the
 * normalised coordinates (a, b), J = d(uu, vv) / d(a, b) row-major 2 x 2 */
static void fisheye_projection_with_jac(double a, double b, double* uu, double* vv, double* J) {
  const double r2 = a * a + b * b;
  const double r = sqrt(r2);
  if (r < 2.220446049250313e-16) {
    *uu = a;
    *vv = b;
    if (J) { J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0; }
    return;
  }
  const double theta = atan(r);
  const double s = theta / r;
  *uu = s * a;
  *vv = s * b;
  if (J) {
    const double g = (r / (1.0 + r2) - theta) / (r2 * r);
    J[0] = s + a * a * g;
    J[1] = a * b * g;
    J[2] = a * b * g;
    J[3] = s + b * b * g;
  }
}

/* the three radial fisheye models (models_jacobian.h:726-942): radial polynomial of the squared
 * fisheye radius t2 with nk coefficients k[], focal lengths (f1, f2), applied after the equidistant
 * projection; J_params columns: [focal..., cx, cy, k...] in the model's parameter order */
static void radial_fisheye_with_jac(double f1, double f2, double c1, double c2, const double* k, int nk,
                                    int two_focals, double u, double v, double w, double* x, double* y,
                                    double* J_params, double* J_uvw) {
  const double inv_w = 1.0 / w;
  const double a = u * inv_w, b = v * inv_w;
  double uu, vv, Jf[4] = {0, 0, 0, 0};
  fisheye_projection_with_jac(a, b, &uu, &vv, J_uvw ? Jf : NULL);
  const double uu2 = uu * uu, vv2 = vv * vv;
  const double t2 = uu2 + vv2;
  double tp[4]; /* t2, t4, t6, t8 */
  tp[0] = t2; tp[1] = t2 * t2; tp[2] = tp[1] * t2; tp[3] = tp[1] * tp[1];
  double radial = 0.0;
  for (int i = 0; i < nk; ++i) radial += k[i] * tp[i];
  const double uu_d = uu + uu * radial, vv_d = vv + vv * radial;
  *x = f1 * uu_d + c1;
  *y = f2 * vv_d + c2;
  if (J_uvw) {
    double d_radial = 0.0;
    for (int i = 0; i < nk; ++i) d_radial += (double)(i + 1) * k[i] * (i == 0 ? 1.0 : tp[i - 1]);
    const double cross = 2.0 * uu * vv * d_radial;
    const double ipjd[4] = {1.0 + radial + 2.0 * uu2 * d_radial, cross, cross, 1.0 + radial + 2.0 * vv2 * d_radial};
    const double m[4] = {ipjd[0] * Jf[0] + ipjd[1] * Jf[2], ipjd[0] * Jf[1] + ipjd[1] * Jf[3],
                         ipjd[2] * Jf[0] + ipjd[3] * Jf[2], ipjd[2] * Jf[1] + ipjd[3] * Jf[3]};
    const double Jab[4] = {f1 * m[0], f1 * m[1], f2 * m[2], f2 * m[3]};
    J_uvw[0] = Jab[0] * inv_w; J_uvw[1] = Jab[1] * inv_w; J_uvw[2] = -(Jab[0] * a + Jab[1] * b) * inv_w;
    J_uvw[3] = Jab[2] * inv_w; J_uvw[4] = Jab[3] * inv_w; J_uvw[5] = -(Jab[2] * a + Jab[3] * b) * inv_w;
  }
  if (J_params) {
    const int P = (two_focals ? 4 : 3) + nk;
    double* r0 = J_params;
    double* r1 = J_params + P;
    int c = 0;
    if (two_focals) { r0[0] = uu_d; r0[1] = 0.0; r1[0] = 0.0; r1[1] = vv_d; c = 2; }
    else { r0[0] = uu_d; r1[0] = vv_d; c = 1; }
    r0[c] = 1.0; r0[c + 1] = 0.0; r1[c] = 0.0; r1[c + 1] = 1.0;
    for (int i = 0; i < nk; ++i) { r0[c + 2 + i] = f1 * uu * tp[i]; r1[c + 2 + i] = f2 * vv * tp[i]; }
  }
}

/* ImgFromCamWithJac, sensor/models_jacobian.h:139-321. J_params row-major 2 x P. */
static int img_from_cam_jac(int model, const double* params, double u, double v, double w,
                            double* x, double* y, double* J_params, double* J_uvw) {
  if (model == BAO_EQUIRECTANGULAR) { /* models_jacobian.h:1502-1565: no cheirality test, (w, h) are metadata */
    const double width = params[0], height = params[1];
    const double horizontal = sqrt(u * u + w * w);
    if (horizontal + fabs(v) < 2.220446049250313e-16) return 0;
    const double theta = atan2(u, w);
    const double phi = atan2(-v, horizontal);
    const double kInv2Pi = 1.0 / (2.0 * M_PI), kInvPi = 1.0 / M_PI;
    *x = (theta * kInv2Pi + 0.5) * width;
    *y = (0.5 - phi * kInvPi) * height;
    if (J_uvw) {
      const double R2 = horizontal * horizontal;
      const double N2 = R2 + v * v;
      const double dtheta_du = w / R2, dtheta_dw = -u / R2;
      const double dphi_du = u * v / (N2 * horizontal), dphi_dv = -horizontal / N2, dphi_dw = v * w / (N2 * horizontal);
      J_uvw[0] = width * kInv2Pi * dtheta_du; J_uvw[1] = 0.0; J_uvw[2] = width * kInv2Pi * dtheta_dw;
      J_uvw[3] = -height * kInvPi * dphi_du; J_uvw[4] = -height * kInvPi * dphi_dv; J_uvw[5] = -height * kInvPi * dphi_dw;
    }
    if (J_params) {
      J_params[0] = theta * kInv2Pi + 0.5; J_params[1] = 0.0;
      J_params[2] = 0.0; J_params[3] = 0.5 - phi * kInvPi;
    }
    return 1;
  }
  if (model == BAO_SIMPLE_DIVISION || model == BAO_DIVISION) {
    /* models_jacobian.h:1291-1411 + internal::DivisionScaleWithJac :88-113 (no cheirality test) */
    const int two = model == BAO_DIVISION;
    const double f1 = params[0], f2 = two ? params[1] : params[0];
    const int ic = two ? 2 : 1;
    const double c1 = params[ic], c2 = params[ic + 1], k = params[ic + 2];
    const double rho2 = u * u + v * v;
    const double disc_sq = w * w - 4.0 * rho2 * k;
    if (disc_sq < 0.0) return 0;
    const double disc = sqrt(disc_sq);
    const double r = 2.0 / (w + disc);
    const double inv_disc = 1.0 / disc, r_sq = r * r;
    const double dr_du = 2.0 * r_sq * k * u * inv_disc, dr_dv = 2.0 * r_sq * k * v * inv_disc;
    const double dr_dw = -0.5 * r_sq * (1.0 + w * inv_disc), dr_dk = r_sq * rho2 * inv_disc;
    *x = f1 * r * u + c1;
    *y = f2 * r * v + c2;
    if (J_uvw) {
      J_uvw[0] = f1 * (r + u * dr_du); J_uvw[1] = f1 * u * dr_dv; J_uvw[2] = f1 * u * dr_dw;
      J_uvw[3] = f2 * v * dr_du; J_uvw[4] = f2 * (r + v * dr_dv); J_uvw[5] = f2 * v * dr_dw;
    }
    if (J_params) {
      const int P = two ? 5 : 4;
      for (int i = 0; i < 2 * P; ++i) J_params[i] = 0.0;
      J_params[0] = r * u;
      J_params[P + (two ? 1 : 0)] = r * v;
      J_params[ic] = 1.0;
      J_params[P + ic + 1] = 1.0;
      J_params[ic + 2] = f1 * u * dr_dk;
      J_params[P + ic + 2] = f2 * v * dr_dk;
    }
    return 1;
  }
  /* HasProjectableDepth (models.h:281-285), check_cheirality = true */
  if (!(w >= 2.220446049250313e-16)) return 0;
  if (model == BAO_EUCM) { /* models_jacobian.h:1413-1500 */
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3];
    const double alpha = params[4], beta = params[5];
    const double q = u * u + v * v;
    const double rho2 = beta * q + w * w;
    if (rho2 < 0.0) return 0;
    const double rho = sqrt(rho2);
    const double den = alpha * rho + (1.0 - alpha) * w;
    if (!(den >= 2.220446049250313e-16)) return 0;
    const double xn = u / den, yn = v / den;
    *x = f1 * xn + c1;
    *y = f2 * yn + c2;
    const double inv_rho = 1.0 / rho, inv_den = 1.0 / den, inv_den2 = inv_den * inv_den;
    const double dden_du = alpha * beta * u * inv_rho, dden_dv = alpha * beta * v * inv_rho;
    const double dden_dw = alpha * w * inv_rho + (1.0 - alpha);
    const double dden_dalpha = rho - w, dden_dbeta = alpha * q * 0.5 * inv_rho;
    if (J_uvw) {
      J_uvw[0] = f1 * (inv_den - u * dden_du * inv_den2);
      J_uvw[1] = f1 * (-u * dden_dv * inv_den2);
      J_uvw[2] = f1 * (-u * dden_dw * inv_den2);
      J_uvw[3] = f2 * (-v * dden_du * inv_den2);
      J_uvw[4] = f2 * (inv_den - v * dden_dv * inv_den2);
      J_uvw[5] = f2 * (-v * dden_dw * inv_den2);
    }
    if (J_params) {
      J_params[0] = xn; J_params[1] = 0.0; J_params[2] = 1.0; J_params[3] = 0.0;
      J_params[4] = f1 * (-u * dden_dalpha * inv_den2); J_params[5] = f1 * (-u * dden_dbeta * inv_den2);
      J_params[6] = 0.0; J_params[7] = yn; J_params[8] = 0.0; J_params[9] = 1.0;
      J_params[10] = f2 * (-v * dden_dalpha * inv_den2); J_params[11] = f2 * (-v * dden_dbeta * inv_den2);
    }
    return 1;
  }
  const double inv_w = 1.0 / w;
  const double uu = u * inv_w, vv = v * inv_w;
  if (model == BAO_FOV) { /* models_jacobian.h:627-724 */
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3], omega = params[4];
    const double a = uu, b = vv;
    const double radius2 = a * a + b * b, omega2 = omega * omega;
    const double kEpsilon = 1e-4;
    double factor, factor_r, factor_omega;
    if (omega2 < kEpsilon) {
      factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
      factor_r = omega2 / 3.0;
      factor_omega = 2.0 * omega * radius2 / 3.0 - omega / 6.0;
    } else if (radius2 < kEpsilon) {
      const double t = tan(omega / 2.0), t2 = t * t;
      const double Q = t * (4.0 * t2 * radius2 - 3.0);
      factor = -2.0 * Q / (3.0 * omega);
      factor_r = -8.0 * t * t2 / (3.0 * omega);
      const double dt_domega = 0.5 * (1.0 + t2);
      const double Q_omega = dt_domega * (12.0 * t2 * radius2 - 3.0);
      factor_omega = -2.0 / (3.0 * omega2) * (Q_omega * omega - Q);
    } else {
      const double radius = sqrt(radius2), t = tan(omega / 2.0);
      const double arg = 2.0 * radius * t, atan_arg = atan(arg);
      const double inv_denom_arg = 1.0 / (1.0 + arg * arg);
      factor = atan_arg / (radius * omega);
      factor_r = (2.0 * t * radius * inv_denom_arg - atan_arg) / (2.0 * radius2 * radius * omega);
      factor_omega = (radius * omega * (1.0 + t * t) * inv_denom_arg - atan_arg) / (radius * omega2);
    }
    const double du = a * factor, dv = b * factor;
    *x = f1 * du + c1;
    *y = f2 * dv + c2;
    if (J_uvw) {
      const double cross = 2.0 * a * b * factor_r;
      const double Jab[4] = {f1 * (factor + 2.0 * a * a * factor_r), f1 * cross, f2 * cross,
                             f2 * (factor + 2.0 * b * b * factor_r)};
      J_uvw[0] = Jab[0] * inv_w; J_uvw[1] = Jab[1] * inv_w; J_uvw[2] = -(Jab[0] * a + Jab[1] * b) * inv_w;
      J_uvw[3] = Jab[2] * inv_w; J_uvw[4] = Jab[3] * inv_w; J_uvw[5] = -(Jab[2] * a + Jab[3] * b) * inv_w;
    }
    if (J_params) {
      J_params[0] = du; J_params[1] = 0.0; J_params[2] = 1.0; J_params[3] = 0.0; J_params[4] = f1 * a * factor_omega;
      J_params[5] = 0.0; J_params[6] = dv; J_params[7] = 0.0; J_params[8] = 1.0; J_params[9] = f2 * b * factor_omega;
    }
    return 1;
  }
  if (model == BAO_SIMPLE_FISHEYE) { /* :1190-1236: equidistant projection, no distortion */
    radial_fisheye_with_jac(params[0], params[0], params[1], params[2], params + 3, 0, 0, u, v, w, x, y,
                            J_params, J_uvw);
    return 1;
  }
  if (model == BAO_FISHEYE) { /* :1238-1288 */
    radial_fisheye_with_jac(params[0], params[1], params[2], params[3], params + 4, 0, 1, u, v, w, x, y,
                            J_params, J_uvw);
    return 1;
  }
  if (model == BAO_SIMPLE_PINHOLE) {
    const double f = params[0], c1 = params[1], c2 = params[2];
    *x = f * uu + c1;
    *y = f * vv + c2;
    if (J_uvw) {
      const double f_inv_w = f * inv_w;
      J_uvw[0] = f_inv_w; J_uvw[1] = 0.0; J_uvw[2] = -f_inv_w * uu;
      J_uvw[3] = 0.0; J_uvw[4] = f_inv_w; J_uvw[5] = -f_inv_w * vv;
    }
    if (J_params) {
      J_params[0] = uu; J_params[1] = 1.0; J_params[2] = 0.0;
      J_params[3] = vv; J_params[4] = 0.0; J_params[5] = 1.0;
    }
    return 1;
  }
  if (model == BAO_PINHOLE) {
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3];
    *x = f1 * uu + c1;
    *y = f2 * vv + c2;
    if (J_uvw) {
      J_uvw[0] = f1 * inv_w; J_uvw[1] = 0.0; J_uvw[2] = -f1 * inv_w * uu;
      J_uvw[3] = 0.0; J_uvw[4] = f2 * inv_w; J_uvw[5] = -f2 * inv_w * vv;
    }
    if (J_params) {
      J_params[0] = uu; J_params[1] = 0.0; J_params[2] = 1.0; J_params[3] = 0.0;
      J_params[4] = 0.0; J_params[5] = vv; J_params[6] = 0.0; J_params[7] = 1.0;
    }
    return 1;
  }
  if (model == BAO_SIMPLE_RADIAL_FISHEYE || model == BAO_RADIAL_FISHEYE) { /* :726-859 */
    radial_fisheye_with_jac(params[0], params[0], params[1], params[2], params + 3,
                            model == BAO_SIMPLE_RADIAL_FISHEYE ? 1 : 2, 0, u, v, w, x, y, J_params, J_uvw);
    return 1;
  }
  if (model == BAO_OPENCV_FISHEYE) { /* :862-942 */
    radial_fisheye_with_jac(params[0], params[1], params[2], params[3], params + 4, 4, 1, u, v, w, x, y,
                            J_params, J_uvw);
    return 1;
  }
  if (model == BAO_FULL_OPENCV) { /* models_jacobian.h:498-625 */
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3];
    const double k1 = params[4], k2 = params[5], p1 = params[6], p2 = params[7];
    const double k3 = params[8], k4 = params[9], k5 = params[10], k6 = params[11];
    const double uu2 = uu * uu, vv2 = vv * vv, uv = uu * vv;
    const double r2 = uu2 + vv2, r4 = r2 * r2, r6 = r4 * r2;
    /* rational radial term num / den */
    const double num = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double den = 1.0 + k4 * r2 + k5 * r4 + k6 * r6;
    const double inv_den = 1.0 / den;
    const double radial = num * inv_den;
    const double xd = uu * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * uu2);
    const double yd = vv * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * vv2);
    *x = f1 * xd + c1;
    *y = f2 * yd + c2;
    if (J_uvw) {
      const double num_prime = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
      const double den_prime = k4 + 2.0 * k5 * r2 + 3.0 * k6 * r4;
      const double d_radial_d_r2 = (num_prime * den - num * den_prime) * inv_den * inv_den;
      const double cross = 2.0 * uv * d_radial_d_r2;
      const double xd_duu = radial + 2.0 * uu2 * d_radial_d_r2 + 2.0 * p1 * vv + 6.0 * p2 * uu;
      const double xd_dvv = cross + 2.0 * p1 * uu + 2.0 * p2 * vv;
      const double yd_duu = cross + 2.0 * p2 * vv + 2.0 * p1 * uu;
      const double yd_dvv = radial + 2.0 * vv2 * d_radial_d_r2 + 2.0 * p2 * uu + 6.0 * p1 * vv;
      const double a00 = f1 * xd_duu, a01 = f1 * xd_dvv, a10 = f2 * yd_duu, a11 = f2 * yd_dvv;
      J_uvw[0] = a00 * inv_w; J_uvw[1] = a01 * inv_w; J_uvw[2] = -(a00 * uu + a01 * vv) * inv_w;
      J_uvw[3] = a10 * inv_w; J_uvw[4] = a11 * inv_w; J_uvw[5] = -(a10 * uu + a11 * vv) * inv_w;
    }
    if (J_params) {
      const double rp[3] = {r2, r4, r6};
      const double neg_num_inv_den2 = -num * inv_den * inv_den;
      double* r0 = J_params;
      double* r1 = J_params + 12;
      r0[0] = xd; r0[1] = 0.0; r0[2] = 1.0; r0[3] = 0.0;
      r1[0] = 0.0; r1[1] = yd; r1[2] = 0.0; r1[3] = 1.0;
      r0[4] = f1 * uu * rp[0] * inv_den; r0[5] = f1 * uu * rp[1] * inv_den;
      r1[4] = f2 * vv * rp[0] * inv_den; r1[5] = f2 * vv * rp[1] * inv_den;
      r0[6] = f1 * 2.0 * uv; r0[7] = f1 * (r2 + 2.0 * uu2);
      r1[6] = f2 * (r2 + 2.0 * vv2); r1[7] = f2 * 2.0 * uv;
      r0[8] = f1 * uu * rp[2] * inv_den; r1[8] = f2 * vv * rp[2] * inv_den;
      for (int i = 0; i < 3; ++i) {
        r0[9 + i] = f1 * uu * (neg_num_inv_den2 * rp[i]);
        r1[9 + i] = f2 * vv * (neg_num_inv_den2 * rp[i]);
      }
    }
    return 1;
  }
  if (model == BAO_RAD_TAN_THIN_PRISM_FISHEYE) { /* models_jacobian.h:1049-1188 */
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3];
    const double* k = params + 4; /* k0..k5 */
    const double p0 = params[10], p1 = params[11], s0 = params[12], s1 = params[13], s2 = params[14], s3 = params[15];
    const double a = uu, b = vv;
    double fu, fv, Jf[4] = {0, 0, 0, 0};
    fisheye_projection_with_jac(a, b, &fu, &fv, J_uvw ? Jf : NULL);
    const double theta2 = fu * fu + fv * fv;
    double th_radial = 1.0, d_th_radial = 0.0, theta_pow[6], power = 1.0;
    for (int i = 0; i < 6; ++i) {
      const double prev_power = power;
      power *= theta2;
      theta_pow[i] = power;
      th_radial += k[i] * power;
      d_th_radial += (double)(i + 1) * k[i] * prev_power;
    }
    const double xr = th_radial * fu, yr = th_radial * fv;
    const double xr2 = xr * xr, yr2 = yr * yr, xyr = xr * yr;
    const double r2 = xr2 + yr2, r4 = r2 * r2;
    const double dx_tang = 2.0 * p1 * xyr + p0 * (r2 + 2.0 * xr2);
    const double dy_tang = 2.0 * p0 * xyr + p1 * (r2 + 2.0 * yr2);
    const double X = xr + dx_tang + (s0 * r2 + s1 * r4);
    const double Y = yr + dy_tang + (s2 * r2 + s3 * r4);
    *x = f1 * X + c1;
    *y = f2 * Y + c2;
    const double B[4] = {1.0 + 2.0 * p1 * yr + 6.0 * p0 * xr + 2.0 * s0 * xr + 4.0 * s1 * xr * r2,
                         2.0 * p1 * xr + 2.0 * p0 * yr + 2.0 * s0 * yr + 4.0 * s1 * yr * r2,
                         2.0 * p0 * yr + 2.0 * p1 * xr + 2.0 * s2 * xr + 4.0 * s3 * xr * r2,
                         1.0 + 2.0 * p0 * xr + 6.0 * p1 * yr + 2.0 * s2 * yr + 4.0 * s3 * yr * r2};
    if (J_uvw) {
      const double cross = 2.0 * fu * fv * d_th_radial;
      const double A[4] = {th_radial + 2.0 * fu * fu * d_th_radial, cross, cross, th_radial + 2.0 * fv * fv * d_th_radial};
      const double m2[4] = {B[0] * A[0] + B[1] * A[2], B[0] * A[1] + B[1] * A[3], B[2] * A[0] + B[3] * A[2], B[2] * A[1] + B[3] * A[3]};
      const double m[4] = {m2[0] * Jf[0] + m2[1] * Jf[2], m2[0] * Jf[1] + m2[1] * Jf[3],
                           m2[2] * Jf[0] + m2[3] * Jf[2], m2[2] * Jf[1] + m2[3] * Jf[3]};
      const double Jab[4] = {f1 * m[0], f1 * m[1], f2 * m[2], f2 * m[3]};
      J_uvw[0] = Jab[0] * inv_w; J_uvw[1] = Jab[1] * inv_w; J_uvw[2] = -(Jab[0] * a + Jab[1] * b) * inv_w;
      J_uvw[3] = Jab[2] * inv_w; J_uvw[4] = Jab[3] * inv_w; J_uvw[5] = -(Jab[2] * a + Jab[3] * b) * inv_w;
    }
    if (J_params) {
      double* r0 = J_params;
      double* r1 = J_params + 16;
      for (int c = 0; c < 32; ++c) J_params[c] = 0.0;
      r0[0] = X; r0[2] = 1.0;
      r1[1] = Y; r1[3] = 1.0;
      for (int i = 0; i < 6; ++i) {
        const double dxr = fu * theta_pow[i], dyr = fv * theta_pow[i];
        r0[4 + i] = f1 * (B[0] * dxr + B[1] * dyr);
        r1[4 + i] = f2 * (B[2] * dxr + B[3] * dyr);
      }
      r0[10] = f1 * (r2 + 2.0 * xr2); r0[11] = f1 * 2.0 * xyr;
      r1[10] = f2 * 2.0 * xyr; r1[11] = f2 * (r2 + 2.0 * yr2);
      r0[12] = f1 * r2; r0[13] = f1 * r4;
      r1[14] = f2 * r2; r1[15] = f2 * r4;
    }
    return 1;
  }
  if (model == BAO_THIN_PRISM_FISHEYE) { /* models_jacobian.h:944-1047 */
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3];
    const double k1 = params[4], k2 = params[5], p1 = params[6], p2 = params[7];
    const double k3 = params[8], k4 = params[9], sx1 = params[10], sy1 = params[11];
    const double a = uu, b = vv; /* normalised coordinates */
    double fu, fv, Jf[4] = {0, 0, 0, 0};
    fisheye_projection_with_jac(a, b, &fu, &fv, J_uvw ? Jf : NULL);
    const double fu2 = fu * fu, fv2 = fv * fv, fuv = fu * fv;
    const double r2 = fu2 + fv2, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
    const double radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
    const double du = fu * radial + 2.0 * p1 * fuv + p2 * (r2 + 2.0 * fu2) + sx1 * r2;
    const double dv = fv * radial + 2.0 * p2 * fuv + p1 * (r2 + 2.0 * fv2) + sy1 * r2;
    const double fu_d = fu + du, fv_d = fv + dv;
    *x = f1 * fu_d + c1;
    *y = f2 * fv_d + c2;
    if (J_uvw) {
      const double d_radial = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4 + 4.0 * k4 * r6;
      const double cross = 2.0 * fuv * d_radial;
      const double ipjd[4] = {
          1.0 + radial + 2.0 * fu2 * d_radial + 2.0 * p1 * fv + 6.0 * p2 * fu + 2.0 * sx1 * fu,
          cross + 2.0 * p1 * fu + 2.0 * p2 * fv + 2.0 * sx1 * fv,
          cross + 2.0 * p2 * fv + 2.0 * p1 * fu + 2.0 * sy1 * fu,
          1.0 + radial + 2.0 * fv2 * d_radial + 2.0 * p2 * fu + 6.0 * p1 * fv + 2.0 * sy1 * fv};
      const double m[4] = {ipjd[0] * Jf[0] + ipjd[1] * Jf[2], ipjd[0] * Jf[1] + ipjd[1] * Jf[3],
                           ipjd[2] * Jf[0] + ipjd[3] * Jf[2], ipjd[2] * Jf[1] + ipjd[3] * Jf[3]};
      const double Jab[4] = {f1 * m[0], f1 * m[1], f2 * m[2], f2 * m[3]};
      J_uvw[0] = Jab[0] * inv_w; J_uvw[1] = Jab[1] * inv_w; J_uvw[2] = -(Jab[0] * a + Jab[1] * b) * inv_w;
      J_uvw[3] = Jab[2] * inv_w; J_uvw[4] = Jab[3] * inv_w; J_uvw[5] = -(Jab[2] * a + Jab[3] * b) * inv_w;
    }
    if (J_params) {
      double* r0 = J_params;
      double* r1 = J_params + 12;
      r0[0] = fu_d; r0[1] = 0.0; r0[2] = 1.0; r0[3] = 0.0;
      r1[0] = 0.0; r1[1] = fv_d; r1[2] = 0.0; r1[3] = 1.0;
      r0[4] = f1 * fu * r2; r0[5] = f1 * fu * r4; r0[6] = f1 * 2.0 * fuv; r0[7] = f1 * (r2 + 2.0 * fu2);
      r1[4] = f2 * fv * r2; r1[5] = f2 * fv * r4; r1[6] = f2 * (r2 + 2.0 * fv2); r1[7] = f2 * 2.0 * fuv;
      r0[8] = f1 * fu * r6; r0[9] = f1 * fu * r8; r0[10] = f1 * r2; r0[11] = 0.0;
      r1[8] = f2 * fv * r6; r1[9] = f2 * fv * r8; r1[10] = 0.0; r1[11] = f2 * r2;
    }
    return 1;
  }
  if (model == BAO_OPENCV) { /* models_jacobian.h:401-496 */
    const double f1 = params[0], f2 = params[1], c1 = params[2], c2 = params[3];
    const double k1 = params[4], k2 = params[5], p1 = params[6], p2 = params[7];
    const double uu2 = uu * uu, vv2 = vv * vv, uv = uu * vv;
    const double r2 = uu2 + vv2;
    const double r4 = r2 * r2;
    const double radial = k1 * r2 + k2 * r4;
    const double du = uu * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * uu2);
    const double dv = vv * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * vv2);
    const double xd = uu + du, yd = vv + dv;
    *x = f1 * xd + c1;
    *y = f2 * yd + c2;
    if (J_uvw) {
      const double d_radial_d_r2 = k1 + 2.0 * k2 * r2;
      const double cross = 2.0 * uv * d_radial_d_r2;
      const double du_duu = radial + 2.0 * uu2 * d_radial_d_r2 + 2.0 * p1 * vv + 6.0 * p2 * uu;
      const double du_dvv = cross + 2.0 * p1 * uu + 2.0 * p2 * vv;
      const double dv_duu = cross + 2.0 * p2 * vv + 2.0 * p1 * uu;
      const double dv_dvv = radial + 2.0 * vv2 * d_radial_d_r2 + 2.0 * p2 * uu + 6.0 * p1 * vv;
      const double a00 = f1 * (1.0 + du_duu);
      const double a01 = f1 * du_dvv;
      const double a10 = f2 * dv_duu;
      const double a11 = f2 * (1.0 + dv_dvv);
      J_uvw[0] = a00 * inv_w; J_uvw[1] = a01 * inv_w; J_uvw[2] = -(a00 * uu + a01 * vv) * inv_w;
      J_uvw[3] = a10 * inv_w; J_uvw[4] = a11 * inv_w; J_uvw[5] = -(a10 * uu + a11 * vv) * inv_w;
    }
    if (J_params) {
      J_params[0] = xd; J_params[1] = 0.0; J_params[2] = 1.0; J_params[3] = 0.0;
      J_params[4] = f1 * uu * r2; J_params[5] = f1 * uu * r4; J_params[6] = f1 * 2.0 * uv;
      J_params[7] = f1 * (r2 + 2.0 * uu2);
      J_params[8] = 0.0; J_params[9] = yd; J_params[10] = 0.0; J_params[11] = 1.0;
      J_params[12] = f2 * vv * r2; J_params[13] = f2 * vv * r4; J_params[14] = f2 * (r2 + 2.0 * vv2);
      J_params[15] = f2 * 2.0 * uv;
    }
    return 1;
  }
  if (model == BAO_RADIAL) { /* models_jacobian.h:323-398 */
    const double f = params[0], c1 = params[1], c2 = params[2], k1 = params[3], k2 = params[4];
    const double uu2 = uu * uu, vv2 = vv * vv;
    const double r2 = uu2 + vv2;
    const double r4 = r2 * r2;
    const double radial = k1 * r2 + k2 * r4;
    const double xd = uu * (1.0 + radial), yd = vv * (1.0 + radial);
    *x = f * xd + c1;
    *y = f * yd + c2;
    if (J_uvw) {
      const double d_radial_d_r2 = k1 + 2.0 * k2 * r2;
      const double cross = 2.0 * uu * vv * d_radial_d_r2;
      const double a00 = f * (1.0 + radial + 2.0 * uu2 * d_radial_d_r2);
      const double a01 = f * cross;
      const double a10 = f * cross;
      const double a11 = f * (1.0 + radial + 2.0 * vv2 * d_radial_d_r2);
      J_uvw[0] = a00 * inv_w; J_uvw[1] = a01 * inv_w; J_uvw[2] = -(a00 * uu + a01 * vv) * inv_w;
      J_uvw[3] = a10 * inv_w; J_uvw[4] = a11 * inv_w; J_uvw[5] = -(a10 * uu + a11 * vv) * inv_w;
    }
    if (J_params) {
      J_params[0] = xd; J_params[1] = 1.0; J_params[2] = 0.0; J_params[3] = f * uu * r2; J_params[4] = f * uu * r4;
      J_params[5] = yd; J_params[6] = 0.0; J_params[7] = 1.0; J_params[8] = f * vv * r2; J_params[9] = f * vv * r4;
    }
    return 1;
  }
  /* SIMPLE_RADIAL */
  {
    const double f = params[0], c1 = params[1], c2 = params[2], k = params[3];
    const double uu2 = uu * uu, vv2 = vv * vv;
    const double r2 = uu2 + vv2;
    const double k_r2 = k * r2;
    const double alpha = 1.0 + k_r2;
    const double xd = alpha * uu, yd = alpha * vv;
    *x = f * xd + c1;
    *y = f * yd + c2;
    if (J_uvw) {
      const double two_k = 2.0 * k;
      const double f_inv_w = f * inv_w;
      const double beta = 1.0 + 3.0 * k_r2;
      const double two_k_uu_vv = two_k * uu * vv;
      J_uvw[0] = f_inv_w * (alpha + two_k * uu2);
      J_uvw[1] = f_inv_w * two_k_uu_vv;
      J_uvw[2] = -f_inv_w * uu * beta;
      J_uvw[3] = f_inv_w * two_k_uu_vv;
      J_uvw[4] = f_inv_w * (alpha + two_k * vv2);
      J_uvw[5] = -f_inv_w * vv * beta;
    }
    if (J_params) {
      J_params[0] = xd; J_params[1] = 1.0; J_params[2] = 0.0; J_params[3] = f * uu * r2;
      J_params[4] = yd; J_params[5] = 0.0; J_params[6] = 1.0; J_params[7] = f * vv * r2;
    }
    return 1;
  }
}

/* AnalyticalReprojErrorCostFunction::Evaluate, reprojection_error.h:68-134.
 * J_point 2x3, J_pose 2x7 (quaternion xyzw then translation), J_params 2xP, row-major. */
BAO_API int bao_reproj_error(int model, const double* point, const double* pose,
                             const double* params, const double* xy, double* residuals,
                             double* J_point, double* J_pose, double* J_params) {
  double J_Rp_quat[12], J_uvw[6], pc[3];
  const int P = num_params_of(model);
  quat_rotate_jac(pose, point, pc, J_pose ? J_Rp_quat : NULL);
  pc[0] += pose[4]; pc[1] += pose[5]; pc[2] += pose[6];
  if (!img_from_cam_jac(model, params, pc[0], pc[1], pc[2], &residuals[0], &residuals[1],
                        J_params, (J_point || J_pose) ? J_uvw : NULL)) {
    residuals[0] = residuals[1] = 0.0;
    if (J_pose) memset(J_pose, 0, sizeof(double) * 14);
    if (J_point) memset(J_point, 0, sizeof(double) * 6);
    if (J_params) memset(J_params, 0, sizeof(double) * 2 * P);
    return 1;
  }
  residuals[0] -= xy[0];
  residuals[1] -= xy[1];
  if (J_point) {
    double R[9];
    quat_to_rot(pose, R);
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 3; ++c)
        J_point[3 * r + c] = J_uvw[3 * r] * R[c] + J_uvw[3 * r + 1] * R[3 + c] + J_uvw[3 * r + 2] * R[6 + c];
  }
  if (J_pose) {
    for (int r = 0; r < 2; ++r) {
      for (int c = 0; c < 4; ++c)
        J_pose[7 * r + c] = J_uvw[3 * r] * J_Rp_quat[c] + J_uvw[3 * r + 1] * J_Rp_quat[4 + c] +
                            J_uvw[3 * r + 2] * J_Rp_quat[8 + c];
      for (int c = 0; c < 3; ++c) J_pose[7 * r + 4 + c] = J_uvw[3 * r + c];
    }
  }
  return 1;
}

/* RigReprojErrorConstantRigCostFunctor (reprojection_error.h:344-417): the same residual seen
 * through a constant sensor_from_rig: p_cam = R_s (R_r X + t_r) + t_s. The reference differentiates
 * it automatically; the chain rule gives J_point = J_uvw R_s R_r, J_pose = J_uvw R_s [dR_rX/dq | I]. */
static int rig_reproj_error_full(int model, const double* point, const double* rig_from_world,
                                 const double* sensor_from_rig, const double* params, const double* xy,
                                 double* residuals, double* J_point, double* J_pose, double* J_params,
                                 double* J_sensor);

BAO_API int bao_rig_reproj_error(int model, const double* point, const double* rig_from_world,
                                 const double* sensor_from_rig, const double* params, const double* xy,
                                 double* residuals, double* J_point, double* J_pose, double* J_params) {
  return rig_reproj_error_full(model, point, rig_from_world, sensor_from_rig, params, xy, residuals, J_point,
                               J_pose, J_params, NULL);
}

/* The same with the Jacobian w.r.t. sensor_from_rig (2 x 7: quaternion xyzw, translation): the point in
 * the camera frame is R_s p_rig + t_s, so d/dq_s = J_uvw d(R_s p_rig)/dq_s and d/dt_s = J_uvw. */
BAO_API int bao_rig_reproj_error_sensor(int model, const double* point, const double* rig_from_world,
                                        const double* sensor_from_rig, const double* params, const double* xy,
                                        double* residuals, double* J_point, double* J_pose, double* J_params,
                                        double* J_sensor) {
  return rig_reproj_error_full(model, point, rig_from_world, sensor_from_rig, params, xy, residuals, J_point,
                               J_pose, J_params, J_sensor);
}

static int rig_reproj_error_full(int model, const double* point, const double* rig_from_world,
                                 const double* sensor_from_rig, const double* params, const double* xy,
                                 double* residuals, double* J_point, double* J_pose, double* J_params,
                                 double* J_sensor) {
  double J_Rp_quat[12], J_uvw[6], pr[3], pc[3], Rs[9];
  const int P = num_params_of(model);
  quat_rotate_jac(rig_from_world, point, pr, J_pose ? J_Rp_quat : NULL);
  pr[0] += rig_from_world[4]; pr[1] += rig_from_world[5]; pr[2] += rig_from_world[6];
  quat_to_rot(sensor_from_rig, Rs);
  for (int r = 0; r < 3; ++r)
    pc[r] = Rs[3 * r] * pr[0] + Rs[3 * r + 1] * pr[1] + Rs[3 * r + 2] * pr[2] + sensor_from_rig[4 + r];
  if (!img_from_cam_jac(model, params, pc[0], pc[1], pc[2], &residuals[0], &residuals[1],
                        J_params, (J_point || J_pose || J_sensor) ? J_uvw : NULL)) {
    residuals[0] = residuals[1] = 0.0;
    if (J_sensor) memset(J_sensor, 0, sizeof(double) * 14);
    if (J_pose) memset(J_pose, 0, sizeof(double) * 14);
    if (J_point) memset(J_point, 0, sizeof(double) * 6);
    if (J_params) memset(J_params, 0, sizeof(double) * 2 * P);
    return 1;
  }
  residuals[0] -= xy[0];
  residuals[1] -= xy[1];
  if (J_sensor) {
    double J_Rs_quat[12], tmp[3];
    quat_rotate_jac(sensor_from_rig, pr, tmp, J_Rs_quat);
    for (int r = 0; r < 2; ++r) {
      for (int c = 0; c < 4; ++c)
        J_sensor[7 * r + c] = J_uvw[3 * r] * J_Rs_quat[c] + J_uvw[3 * r + 1] * J_Rs_quat[4 + c] +
                              J_uvw[3 * r + 2] * J_Rs_quat[8 + c];
      for (int c = 0; c < 3; ++c) J_sensor[7 * r + 4 + c] = J_uvw[3 * r + c];
    }
  }
  if (J_point || J_pose) {
    double Jr[6]; /* J_uvw * R_s: derivative w.r.t. the point in the rig frame */
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 3; ++c)
        Jr[3 * r + c] = J_uvw[3 * r] * Rs[c] + J_uvw[3 * r + 1] * Rs[3 + c] + J_uvw[3 * r + 2] * Rs[6 + c];
    if (J_point) {
      double R[9];
      quat_to_rot(rig_from_world, R);
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
          J_point[3 * r + c] = Jr[3 * r] * R[c] + Jr[3 * r + 1] * R[3 + c] + Jr[3 * r + 2] * R[6 + c];
    }
    if (J_pose) {
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 4; ++c)
          J_pose[7 * r + c] = Jr[3 * r] * J_Rp_quat[c] + Jr[3 * r + 1] * J_Rp_quat[4 + c] +
                              Jr[3 * r + 2] * J_Rp_quat[8 + c];
        for (int c = 0; c < 3; ++c) J_pose[7 * r + 4 + c] = Jr[3 * r + c];
      }
    }
  }
  return 1;
}

/* ceres::LossFunction::Evaluate for the four losses COLMAP can select
 * (bundle_adjustment_ceres.cc:66-80). Ceres is not vendored in the reference tree; restated from
 * its published loss_function.cc: rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s), s = |r|^2. */
BAO_API void bao_loss(int type, double a, double s, double rho[3]) {
  const double kMin = 2.2250738585072014e-308; /* std::numeric_limits<double>::min() */
  if (type == BAO_LOSS_HUBER) {
    const double b = a * a;
    if (s > b) {
      const double r = sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = fmax(kMin, a / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  } else if (type == BAO_LOSS_SOFT_L1) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double tmp = sqrt(sum);
    rho[0] = 2.0 * b * (tmp - 1.0);
    rho[1] = fmax(kMin, 1.0 / tmp);
    rho[2] = -(c * rho[1]) / (2.0 * sum);
  } else if (type == BAO_LOSS_CAUCHY) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = b * log(sum);
    rho[1] = fmax(kMin, inv);
    rho[2] = -c * (inv * inv);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

/* ceres::internal::Corrector (corrector.cc): rescales the residual and its Jacobian so that the
 * Gauss-Newton model of 1/2 |r'|^2 matches the second-order model of 1/2 rho(|r|^2) (Triggs).
 * Returns the factors: r' = residual_scaling * r,  J' = sqrt_rho1 * (J - alpha_sq_norm * r r^T J). */
static void corrector(double sq_norm, const double rho[3], double* sqrt_rho1, double* residual_scaling,
                      double* alpha_sq_norm) {
  *sqrt_rho1 = sqrt(rho[1]);
  if (sq_norm == 0.0 || rho[2] <= 0.0) {
    *residual_scaling = *sqrt_rho1;
    *alpha_sq_norm = 0.0;
    return;
  }
  const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
  const double alpha = 1.0 - sqrt(D);
  *residual_scaling = *sqrt_rho1 / (1.0 - alpha);
  *alpha_sq_norm = alpha / sq_norm;
}

/* ceres::EigenQuaternionManifold (xyzw storage): x_plus = [sin|d|/|d| d, cos|d|] (*) x ;
 * PlusJacobian at d = 0 (4x3 row-major) = [[w, z,-y],[-z, w, x],[y,-x, w],[-x,-y,-z]]. */
BAO_API void bao_quat_plus(const double* q, const double* d, double* out) {
  const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0.0) { memcpy(out, q, 32); return; }
  const double s = sin(n) / n;
  const double dx = s * d[0], dy = s * d[1], dz = s * d[2], dw = cos(n);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  /* Hamilton product dq * q */
  out[0] = dw * x + dx * w + dy * z - dz * y;
  out[1] = dw * y - dx * z + dy * w + dz * x;
  out[2] = dw * z + dx * y - dy * x + dz * w;
  out[3] = dw * w - dx * x - dy * y - dz * z;
}

static void quat_plus_jac(const double* q, double J[12]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  J[0] = w;  J[1] = z;  J[2] = -y;
  J[3] = -z; J[4] = w;  J[5] = x;
  J[6] = y;  J[7] = -x; J[8] = w;
  J[9] = -x; J[10] = -y; J[11] = -z;
}

/* Position-prior residual and its ambient Jacobians (pose_prior.h:76-129, autodiff in the
 * reference; the quaternion normalisation term of Eigen's inverse() lies along q and vanishes under
 * the manifold's PlusJacobian, so the conjugate is differentiated).
 *   no sensor:  r0 = pos + R(q)^T t
 *   sensor:     r0 = pos + R(q_r)^T (t_r + R(q_s)^T t_s)
 * J_pose / J_sens: 3 x 7 row-major w.r.t. (qx qy qz qw tx ty tz) of rig_from_world / sensor_from_rig. */
static void position_prior(const double* pos, const double* pose, const double* sens, double r0[3],
                           double* J_pose, double* J_sens) {
  double w_[3] = {pose[4], pose[5], pose[6]};
  double Rs[9];
  if (sens) {
    const double qsc[4] = {-sens[0], -sens[1], -sens[2], sens[3]};
    double v[3];
    quat_rotate_jac(qsc, sens + 4, v, NULL);
    for (int c = 0; c < 3; ++c) w_[c] += v[c];
    quat_to_rot(sens, Rs);
  }
  const double qc[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
  double v[3], Jq[12];
  quat_rotate_jac(qc, w_, v, J_pose ? Jq : NULL);
  for (int c = 0; c < 3; ++c) r0[c] = pos[c] + v[c];
  if (!J_pose) return;
  double Rr[9];
  quat_to_rot(pose, Rr);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) J_pose[7 * r + c] = -Jq[4 * r + c];  /* d conj / d q */
    J_pose[7 * r + 3] = Jq[4 * r + 3];
    for (int c = 0; c < 3; ++c) J_pose[7 * r + 4 + c] = Rr[3 * c + r];  /* R_r^T */
  }
  if (sens && J_sens) {
    const double qsc[4] = {-sens[0], -sens[1], -sens[2], sens[3]};
    double vs[3], Jqs[12];
    quat_rotate_jac(qsc, sens + 4, vs, Jqs);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 4; ++c) {  /* R_r^T * d(R(conj q_s) t_s)/dq_s */
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += Rr[3 * k + r] * Jqs[4 * k + c];
        J_sens[7 * r + c] = c < 3 ? -a : a;
      }
      for (int c = 0; c < 3; ++c) {  /* R_r^T R_s^T */
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += Rr[3 * k + r] * Rs[3 * c + k];
        J_sens[7 * r + 4 + c] = a;
      }
    }
  }
}

BAO_API void bao_position_prior(const double* pos, const double* pose, const double* sens, double* r0,
                                double* J_pose, double* J_sens) {
  position_prior(pos, pose, sens, r0, J_pose, J_sens);
}

/* ------------------------------------------------------------------------- */
/* Program: tangent-space layout of the variable blocks                        */
/* ------------------------------------------------------------------------- */

#define MAX_CB 28 /* max tangent width of the camera-side blocks seen by one residual: 6 + P_t (P_t <= 16) + 6 */

typedef struct {
  const bao_problem* p;
  int loss_type;       /* robust loss applied to every residual block */
  double loss_scale;
  int64_t n_obs;       /* active observations */
  int64_t* obs;        /* indices into the problem's observation arrays */
  int* pose_off;       /* [num_poses] offset into the camera-side vector, -1 const */
  int* pose_dim;       /* 0, 5 or 6 */
  int* cam_off;        /* [num_cams] */
  int* cam_dim;
  int* cam_var;        /* [num_cams][12] indices of variable params */
  int* sens_off;       /* [num_sensors] offset of a variable sensor_from_rig (6 wide), -1 const */
  const double* sensors; /* sensor_from_rig values the linearisation reads (current or candidate) */
  int* point_off;      /* [num_points] offset into point-side vector (3 each), -1 const */
  int n_c, n_p;        /* sizes of camera-side / point-side tangent vectors */
  /* CSR by point and by camera-side block of active observation slots */
  int64_t *pt_ptr, *pt_idx;
  int n_blk;
  int *blk_off, *blk_dim, *blk_kind; /* kind 0: pose block, 1: intrinsics block, 2: sensor_from_rig block */
  int64_t *blk_ptr, *blk_idx;
} program;

static void program_free(program* g) {
  free(g->obs); free(g->pose_off); free(g->pose_dim); free(g->cam_off); free(g->cam_dim);
  free(g->cam_var); free(g->point_off); free(g->pt_ptr); free(g->pt_idx); free(g->sens_off);
  free(g->blk_off); free(g->blk_dim); free(g->blk_kind); free(g->blk_ptr); free(g->blk_idx);
}

static void program_build(program* g, const bao_problem* p) {
  memset(g, 0, sizeof(*g));
  g->p = p;
  g->pose_off = (int*)malloc(sizeof(int) * (p->num_poses + 1));
  g->pose_dim = (int*)calloc(p->num_poses + 1, sizeof(int));
  g->cam_off = (int*)malloc(sizeof(int) * (p->num_cams + 1));
  g->cam_dim = (int*)calloc(p->num_cams + 1, sizeof(int));
  g->cam_var = (int*)calloc((size_t)(p->num_cams + 1) * BAO_CAM_STRIDE, sizeof(int));
  g->point_off = (int*)malloc(sizeof(int) * (p->num_points + 1));
  /* which blocks are referenced by an observation that has >= 1 variable block */
  uint8_t* pose_used = (uint8_t*)calloc(p->num_poses + 1, 1);
  uint8_t* cam_used = (uint8_t*)calloc(p->num_cams + 1, 1);
  uint8_t* point_used = (uint8_t*)calloc(p->num_points + 1, 1);
  int* cam_nvar = (int*)calloc(p->num_cams + 1, sizeof(int));
  for (int k = 0; k < p->num_cams; ++k) {
    const int P = num_params_of(p->cam_model[k]);
    for (int j = 0; j < P; ++j)
      if (!p->cam_const[(size_t)k * BAO_CAM_STRIDE + j]) cam_nvar[k]++;
  }
  g->obs = (int64_t*)malloc(sizeof(int64_t) * (p->num_obs + 1));
  const int ns = p->num_sensors > 0 ? p->num_sensors : 0;
  uint8_t* sens_used = (uint8_t*)calloc((size_t)ns + 1, 1);
  g->sens_off = (int*)malloc(sizeof(int) * ((size_t)ns + 1));
  g->sensors = p->sensors;
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int pi = p->obs_pose[o], ci = p->obs_cam[o], xi = p->obs_point[o];
    const int si = p->obs_sensor ? p->obs_sensor[o] : -1;
    const int sens_var = si >= 0 && p->sensor_const && !p->sensor_const[si];
    const int var = (!p->pose_const[pi]) || cam_nvar[ci] > 0 || (!p->point_const[xi]) || sens_var;
    if (!var) continue; /* all blocks constant: not part of the reduced program */
    g->obs[g->n_obs++] = o;
    pose_used[pi] = cam_used[ci] = point_used[xi] = 1;
    if (sens_var) sens_used[si] = 1;
  }
  int off = 0;
  for (int i = 0; i < p->num_poses; ++i) {
    if (p->pose_const[i] || !pose_used[i]) { g->pose_off[i] = -1; g->pose_dim[i] = 0; continue; }
    g->pose_dim[i] = pose_tangent_dim(p->pose_fixed_t[i]);
    g->pose_off[i] = off;
    off += g->pose_dim[i];
  }
  for (int k = 0; k < p->num_cams; ++k) {
    if (cam_nvar[k] == 0 || !cam_used[k]) { g->cam_off[k] = -1; g->cam_dim[k] = 0; continue; }
    const int P = num_params_of(p->cam_model[k]);
    int d = 0;
    for (int j = 0; j < P; ++j)
      if (!p->cam_const[(size_t)k * BAO_CAM_STRIDE + j]) g->cam_var[(size_t)k * BAO_CAM_STRIDE + d++] = j;
    g->cam_dim[k] = d;
    g->cam_off[k] = off;
    off += d;
  }
  for (int k = 0; k < ns; ++k) {
    if (!sens_used[k]) { g->sens_off[k] = -1; continue; }
    g->sens_off[k] = off;
    off += 6;
  }
  g->n_c = off;
  int poff = 0;
  for (int j = 0; j < p->num_points; ++j) {
    if (p->point_const[j] || !point_used[j]) { g->point_off[j] = -1; continue; }
    g->point_off[j] = poff;
    poff += 3;
  }
  g->n_p = poff;
  /* CSR of active observations by point */
  g->pt_ptr = (int64_t*)calloc((size_t)p->num_points + 2, sizeof(int64_t));
  g->pt_idx = (int64_t*)malloc(sizeof(int64_t) * (g->n_obs + 1));
  for (int64_t a = 0; a < g->n_obs; ++a) g->pt_ptr[p->obs_point[g->obs[a]] + 1]++;
  for (int j = 0; j < p->num_points; ++j) g->pt_ptr[j + 1] += g->pt_ptr[j];
  int64_t* fill = (int64_t*)calloc((size_t)p->num_points + 1, sizeof(int64_t));
  for (int64_t a = 0; a < g->n_obs; ++a) {
    const int j = p->obs_point[g->obs[a]];
    g->pt_idx[g->pt_ptr[j] + fill[j]++] = a;
  }
  free(fill);
  /* camera-side blocks and their observation lists */
  g->blk_off = (int*)malloc(sizeof(int) * ((size_t)p->num_poses + p->num_cams + ns + 1));
  g->blk_dim = (int*)malloc(sizeof(int) * ((size_t)p->num_poses + p->num_cams + ns + 1));
  g->blk_kind = (int*)malloc(sizeof(int) * ((size_t)p->num_poses + p->num_cams + ns + 1));
  int* blk_of_sens = (int*)malloc(sizeof(int) * ((size_t)ns + 1));
  int* blk_of_pose = (int*)malloc(sizeof(int) * (p->num_poses + 1));
  int* blk_of_cam = (int*)malloc(sizeof(int) * (p->num_cams + 1));
  for (int i = 0; i < p->num_poses; ++i) {
    blk_of_pose[i] = -1;
    if (g->pose_off[i] < 0) continue;
    blk_of_pose[i] = g->n_blk;
    g->blk_off[g->n_blk] = g->pose_off[i]; g->blk_dim[g->n_blk] = g->pose_dim[i]; g->blk_kind[g->n_blk++] = 0;
  }
  for (int k = 0; k < p->num_cams; ++k) {
    blk_of_cam[k] = -1;
    if (g->cam_off[k] < 0) continue;
    blk_of_cam[k] = g->n_blk;
    g->blk_off[g->n_blk] = g->cam_off[k]; g->blk_dim[g->n_blk] = g->cam_dim[k]; g->blk_kind[g->n_blk++] = 1;
  }
  for (int k = 0; k < ns; ++k) {
    blk_of_sens[k] = -1;
    if (g->sens_off[k] < 0) continue;
    blk_of_sens[k] = g->n_blk;
    g->blk_off[g->n_blk] = g->sens_off[k]; g->blk_dim[g->n_blk] = 6; g->blk_kind[g->n_blk++] = 2;
  }
  g->blk_ptr = (int64_t*)calloc((size_t)g->n_blk + 2, sizeof(int64_t));
  for (int64_t a = 0; a < g->n_obs; ++a) {
    const int bp = blk_of_pose[p->obs_pose[g->obs[a]]], bc = blk_of_cam[p->obs_cam[g->obs[a]]];
    const int si = p->obs_sensor ? p->obs_sensor[g->obs[a]] : -1;
    const int bs = si >= 0 ? blk_of_sens[si] : -1;
    if (bp >= 0) g->blk_ptr[bp + 1]++;
    if (bc >= 0) g->blk_ptr[bc + 1]++;
    if (bs >= 0) g->blk_ptr[bs + 1]++;
  }
  for (int b = 0; b < g->n_blk; ++b) g->blk_ptr[b + 1] += g->blk_ptr[b];
  g->blk_idx = (int64_t*)malloc(sizeof(int64_t) * (g->blk_ptr[g->n_blk] + 1));
  int64_t* bfill = (int64_t*)calloc((size_t)g->n_blk + 1, sizeof(int64_t));
  for (int64_t a = 0; a < g->n_obs; ++a) {
    const int bp = blk_of_pose[p->obs_pose[g->obs[a]]], bc = blk_of_cam[p->obs_cam[g->obs[a]]];
    const int si = p->obs_sensor ? p->obs_sensor[g->obs[a]] : -1;
    const int bs = si >= 0 ? blk_of_sens[si] : -1;
    if (bp >= 0) g->blk_idx[g->blk_ptr[bp] + bfill[bp]++] = a;
    if (bc >= 0) g->blk_idx[g->blk_ptr[bc] + bfill[bc]++] = a;
    if (bs >= 0) g->blk_idx[g->blk_ptr[bs] + bfill[bs]++] = a;
  }
  free(bfill); free(blk_of_pose); free(blk_of_cam); free(blk_of_sens);
  free(pose_used); free(cam_used); free(point_used); free(cam_nvar); free(sens_used);
}

/* OpenMP only pays off on large problems (and a 128-thread team spinning on a 40-point loop
 * is pathologically slow) */
#define BAO_PAR(n) if ((n) > 20000)

/* per-active-observation linearisation in the tangent space */
typedef struct {
  double r[2];          /* residual (loss-corrected when the Jacobian was requested) */
  double cost;          /* 1/2 rho(|r|^2) of the uncorrected residual */
  double Jc[2][MAX_CB]; /* pose tangent columns, intrinsics tangent columns, sensor_from_rig tangent columns */
  double Jp[2][3];
  int pose_dim, cam_dim, sens_dim; /* widths inside Jc */
  int so;                          /* tangent offset of the sensor block, -1 */
} lin_obs;

/* first column of block kind `kind` inside lin_obs::Jc */
static inline int lin_base(const lin_obs* L, int kind) {
  return kind == 0 ? 0 : (kind == 1 ? L->pose_dim : L->pose_dim + L->cam_dim);
}

static void linearize_obs(const program* g, const double* poses, const double* cams,
                          const double* points, int64_t a, lin_obs* L, int want_jac) {
  const bao_problem* p = g->p;
  const int64_t o = g->obs[a];
  const int pi = p->obs_pose[o], ci = p->obs_cam[o], xi = p->obs_point[o];
  const int model = p->cam_model[ci];
  double Jpt[6], Jpose[14], Jpar[32], Jsens[14];
  const int si = p->obs_sensor ? p->obs_sensor[o] : -1;
  L->so = si >= 0 ? g->sens_off[si] : -1;
  L->sens_dim = L->so >= 0 ? 6 : 0;
  if (si >= 0)
    rig_reproj_error_full(model, points + 3 * (size_t)xi, poses + 7 * (size_t)pi, g->sensors + 7 * (size_t)si,
                          cams + (size_t)ci * BAO_CAM_STRIDE, p->obs_xy + 2 * o, L->r,
                          want_jac ? Jpt : NULL, want_jac ? Jpose : NULL, want_jac ? Jpar : NULL,
                          (want_jac && L->so >= 0) ? Jsens : NULL);
  else
    bao_reproj_error(model, points + 3 * (size_t)xi, poses + 7 * (size_t)pi,
                     cams + (size_t)ci * BAO_CAM_STRIDE, p->obs_xy + 2 * o, L->r,
                     want_jac ? Jpt : NULL, want_jac ? Jpose : NULL, want_jac ? Jpar : NULL);
  L->pose_dim = g->pose_dim[pi];
  L->cam_dim = g->cam_dim[ci];
  /* robust loss: cost 1/2 rho(s); Corrector factors for the residual block */
  const double sq_norm = L->r[0] * L->r[0] + L->r[1] * L->r[1];
  double rho[3], sqrt_rho1 = 1.0, residual_scaling = 1.0, alpha_sq_norm = 0.0;
  bao_loss(g->loss_type, g->loss_scale, sq_norm, rho);
  L->cost = 0.5 * rho[0];
  if (!want_jac) return;
  if (g->loss_type != BAO_LOSS_TRIVIAL) {
    corrector(sq_norm, rho, &sqrt_rho1, &residual_scaling, &alpha_sq_norm);
    /* CorrectJacobian on the ambient Jacobians of every block, then CorrectResiduals */
    const double r0 = L->r[0], r1 = L->r[1];
#define BAO_CORRECT(J, stride, n)                                               \
    for (int c_ = 0; c_ < (n); ++c_) {                                          \
      const double j0 = (J)[c_], j1 = (J)[(stride) + c_];                        \
      const double rtj = r0 * j0 + r1 * j1;                                      \
      (J)[c_] = sqrt_rho1 * (j0 - alpha_sq_norm * r0 * rtj);                     \
      (J)[(stride) + c_] = sqrt_rho1 * (j1 - alpha_sq_norm * r1 * rtj);          \
    }
    BAO_CORRECT(Jpt, 3, 3)
    BAO_CORRECT(Jpose, 7, 7)
    if (L->so >= 0) { BAO_CORRECT(Jsens, 7, 7) }
    { const int P_ = num_params_of(model); BAO_CORRECT(Jpar, P_, P_) }
#undef BAO_CORRECT
    L->r[0] *= residual_scaling;
    L->r[1] *= residual_scaling;
  }
  memset(L->Jc, 0, sizeof(L->Jc));
  memset(L->Jp, 0, sizeof(L->Jp));
  if (L->pose_dim > 0) {
    double PJ[12];
    quat_plus_jac(poses + 7 * (size_t)pi, PJ);
    const int fixed = pose_fixed_coord(p->pose_fixed_t[pi]);
    const int rotc = pose_rot_const(p->pose_fixed_t[pi]);
    for (int r = 0; r < 2; ++r) {
      int d = 0;
      if (!rotc) {
        for (int c = 0; c < 3; ++c)
          L->Jc[r][c] = Jpose[7 * r + 0] * PJ[c] + Jpose[7 * r + 1] * PJ[3 + c] +
                        Jpose[7 * r + 2] * PJ[6 + c] + Jpose[7 * r + 3] * PJ[9 + c];
        d = 3;
      }
      for (int c = 0; c < 3; ++c) {
        if (c == fixed) continue;
        L->Jc[r][d++] = Jpose[7 * r + 4 + c];
      }
    }
  }
  if (L->cam_dim > 0) {
    const int P = num_params_of(model);
    for (int r = 0; r < 2; ++r)
      for (int d = 0; d < L->cam_dim; ++d)
        L->Jc[r][L->pose_dim + d] = Jpar[P * r + g->cam_var[(size_t)ci * BAO_CAM_STRIDE + d]];
  }
  if (L->so >= 0) {
    double PJ[12];
    quat_plus_jac(g->sensors + 7 * (size_t)si, PJ);
    const int base = L->pose_dim + L->cam_dim;
    for (int r = 0; r < 2; ++r) {
      for (int c = 0; c < 3; ++c) {
        L->Jc[r][base + c] = Jsens[7 * r + 0] * PJ[c] + Jsens[7 * r + 1] * PJ[3 + c] +
                             Jsens[7 * r + 2] * PJ[6 + c] + Jsens[7 * r + 3] * PJ[9 + c];
        L->Jc[r][base + 3 + c] = Jsens[7 * r + 4 + c];
      }
    }
  }
  if (g->point_off[xi] >= 0)
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 3; ++c) L->Jp[r][c] = Jpt[3 * r + c];
}

/* Linearised position prior: residual (sqrt-information weighted, loss-corrected) and the tangent
 * columns of its pose block and, when variable, its sensor_from_rig block. */
typedef struct {
  double r[3];
  double J[3][12];     /* [pose tangent (pose_dim) | sensor tangent (sens_dim)] */
  int pose_dim, sens_dim;
  int po, so;          /* tangent offsets, -1: constant */
  double cost;
} lin_prior;

static int prior_active(const program* g, int k) {
  const bao_problem* p = g->p;
  const int si = p->prior_sensor ? p->prior_sensor[k] : -1;
  return g->pose_off[p->prior_pose[k]] >= 0 || (si >= 0 && g->sens_off[si] >= 0);
}

static void linearize_prior(const program* g, const double* poses, int k, lin_prior* L, int want_jac) {
  const bao_problem* p = g->p;
  const int pi = p->prior_pose[k];
  const int si = p->prior_sensor ? p->prior_sensor[k] : -1;
  const double* pose = poses + 7 * (size_t)pi;
  const double* sens = si >= 0 ? g->sensors + 7 * (size_t)si : NULL;
  double r0[3], Jpose[21], Jsens[21];
  L->po = g->pose_off[pi];
  L->pose_dim = L->po >= 0 ? g->pose_dim[pi] : 0;
  L->so = si >= 0 ? g->sens_off[si] : -1;
  L->sens_dim = L->so >= 0 ? 6 : 0;
  position_prior(p->prior_position + 3 * (size_t)k, pose, sens, r0, want_jac ? Jpose : NULL,
                 (want_jac && L->so >= 0) ? Jsens : NULL);
  const double* A = p->prior_sqrt_info + 9 * (size_t)k;
  for (int r = 0; r < 3; ++r) L->r[r] = A[3 * r] * r0[0] + A[3 * r + 1] * r0[1] + A[3 * r + 2] * r0[2];
  const double sq_norm = L->r[0] * L->r[0] + L->r[1] * L->r[1] + L->r[2] * L->r[2];
  double rho[3];
  bao_loss(p->prior_loss_type, p->prior_loss_scale, sq_norm, rho);
  L->cost = 0.5 * rho[0];
  if (!want_jac) return;
  memset(L->J, 0, sizeof(L->J));
  double Jt[3][12]; /* unweighted tangent columns */
  memset(Jt, 0, sizeof(Jt));
  if (L->pose_dim > 0) {
    double PJ[12];
    quat_plus_jac(pose, PJ);
    const int fixed = pose_fixed_coord(p->pose_fixed_t[pi]);
    const int rotc = pose_rot_const(p->pose_fixed_t[pi]);
    for (int r = 0; r < 3; ++r) {
      int d = 0;
      if (!rotc) {
        for (int c = 0; c < 3; ++c)
          Jt[r][c] = Jpose[7 * r] * PJ[c] + Jpose[7 * r + 1] * PJ[3 + c] + Jpose[7 * r + 2] * PJ[6 + c] +
                     Jpose[7 * r + 3] * PJ[9 + c];
        d = 3;
      }
      for (int c = 0; c < 3; ++c) {
        if (c == fixed) continue;
        Jt[r][d++] = Jpose[7 * r + 4 + c];
      }
    }
  }
  if (L->so >= 0) {
    double PJ[12];
    quat_plus_jac(sens, PJ);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        Jt[r][L->pose_dim + c] = Jsens[7 * r] * PJ[c] + Jsens[7 * r + 1] * PJ[3 + c] + Jsens[7 * r + 2] * PJ[6 + c] +
                                 Jsens[7 * r + 3] * PJ[9 + c];
        Jt[r][L->pose_dim + 3 + c] = Jsens[7 * r + 4 + c];
      }
  }
  const int w = L->pose_dim + L->sens_dim;
  for (int r = 0; r < 3; ++r)
    for (int d = 0; d < w; ++d) L->J[r][d] = A[3 * r] * Jt[0][d] + A[3 * r + 1] * Jt[1][d] + A[3 * r + 2] * Jt[2][d];
  if (p->prior_loss_type != BAO_LOSS_TRIVIAL) {
    double sqrt_rho1, residual_scaling, alpha_sq_norm;
    corrector(sq_norm, rho, &sqrt_rho1, &residual_scaling, &alpha_sq_norm);
    for (int d = 0; d < w; ++d) {
      const double rtj = L->r[0] * L->J[0][d] + L->r[1] * L->J[1][d] + L->r[2] * L->J[2][d];
      for (int r = 0; r < 3; ++r) L->J[r][d] = sqrt_rho1 * (L->J[r][d] - alpha_sq_norm * L->r[r] * rtj);
    }
    for (int r = 0; r < 3; ++r) L->r[r] *= residual_scaling;
  }
}

/* J x of a prior for a camera-side vector x */
static void prior_jx(const lin_prior* L, const double* x, double out[3]) {
  for (int r = 0; r < 3; ++r) {
    double v = 0.0;
    for (int d = 0; d < L->pose_dim; ++d) v += L->J[r][d] * x[L->po + d];
    for (int d = 0; d < L->sens_dim; ++d) v += L->J[r][L->pose_dim + d] * x[L->so + d];
    out[r] = v;
  }
}

static double evaluate_cost(const program* g, const double* poses, const double* cams,
                            const double* points) {
  double cost = 0.0;
  for (int k = 0; k < g->p->num_priors; ++k) {
    if (!prior_active(g, k)) continue;
    lin_prior L;
    linearize_prior(g, poses, k, &L, 0);
    cost += L.cost;
  }
#pragma omp parallel for reduction(+ : cost) schedule(static) BAO_PAR(g->n_obs)
  for (int64_t a = 0; a < g->n_obs; ++a) {
    lin_obs L;
    linearize_obs(g, poses, cams, points, a, &L, 0);
    cost += L.cost;
  }
  return cost;
}

/* x_plus = Plus(x, delta) for every variable block */
/* x_plus of the variable sensor_from_rig blocks: Plus() of a full pose block */
static void apply_step_sensors(const program* g, const double* dc, const double* sensors, double* nsensors) {
  const bao_problem* p = g->p;
  if (p->num_sensors <= 0) return;
  memcpy(nsensors, sensors, sizeof(double) * 7 * (size_t)p->num_sensors);
  for (int k = 0; k < p->num_sensors; ++k) {
    if (g->sens_off[k] < 0) continue;
    const double* d = dc + g->sens_off[k];
    bao_quat_plus(sensors + 7 * (size_t)k, d, nsensors + 7 * (size_t)k);
    for (int c = 0; c < 3; ++c) nsensors[7 * (size_t)k + 4 + c] += d[3 + c];
  }
}

static void apply_step(const program* g, const double* dc, const double* dp, const double* poses,
                       const double* cams, const double* points, double* nposes, double* ncams,
                       double* npoints) {
  const bao_problem* p = g->p;
  memcpy(nposes, poses, sizeof(double) * 7 * (size_t)p->num_poses);
  memcpy(ncams, cams, sizeof(double) * BAO_CAM_STRIDE * (size_t)p->num_cams);
  memcpy(npoints, points, sizeof(double) * 3 * (size_t)p->num_points);
  for (int i = 0; i < p->num_poses; ++i) {
    if (g->pose_off[i] < 0) continue;
    const double* d = dc + g->pose_off[i];
    int k = 0;
    if (!pose_rot_const(p->pose_fixed_t[i])) {
      bao_quat_plus(poses + 7 * (size_t)i, d, nposes + 7 * (size_t)i);
      k = 3;
    }
    for (int c = 0; c < 3; ++c) {
      if (c == pose_fixed_coord(p->pose_fixed_t[i])) continue;
      nposes[7 * (size_t)i + 4 + c] += d[k++];
    }
  }
  for (int k = 0; k < p->num_cams; ++k) {
    if (g->cam_off[k] < 0) continue;
    for (int d = 0; d < g->cam_dim[k]; ++d)
      ncams[(size_t)k * BAO_CAM_STRIDE + g->cam_var[(size_t)k * BAO_CAM_STRIDE + d]] += dc[g->cam_off[k] + d];
  }
  for (int j = 0; j < p->num_points; ++j) {
    if (g->point_off[j] < 0) continue;
    for (int c = 0; c < 3; ++c) npoints[3 * (size_t)j + c] += dp[g->point_off[j] + c];
  }
}

/* ------------------------------------------------------------------------- */
/* Linear algebra on the block-sparse Jacobian                                 */
/* ------------------------------------------------------------------------- */

typedef struct {
  const program* g;
  lin_obs* L;         /* [n_obs] scaled Jacobian blocks and residuals */
  double* Dc;         /* [n_c] LM diagonal (D, not D^2), camera side */
  double* Dp;         /* [n_p] point side */
  double* Cinv;       /* [num_points][9] inverse of (E^T E + Dp^2) blocks */
  double* Minv;       /* block-Jacobi preconditioner: per camera-side block, dim x dim inverse, packed */
  int* blk_off;       /* offset in the camera-side vector per block */
  int* blk_dim;
  int64_t* blk_moff;  /* offset into Minv */
  int n_blk;
  lin_prior* P;       /* [n_prior] linearised position priors (scaled) */
  int n_prior;
} linsys;

static int cam_offsets(const program* g, int64_t a, int* po, int* co) {
  const bao_problem* p = g->p;
  const int64_t o = g->obs[a];
  *po = g->pose_off[p->obs_pose[o]];
  *co = g->cam_off[p->obs_cam[o]];
  return 0;
}

/* gather the camera-side entries a residual sees */
static inline void gather_c(const program* g, const lin_obs* L, int po, int co, const double* x,
                            double* xc) {
  for (int d = 0; d < L->pose_dim; ++d) xc[d] = x[po + d];
  for (int d = 0; d < L->cam_dim; ++d) xc[L->pose_dim + d] = x[co + d];
  for (int d = 0; d < L->sens_dim; ++d) xc[L->pose_dim + L->cam_dim + d] = x[L->so + d];
}

static void invert_sym(const double* A, int n, double* Ainv) {
  /* Gauss-Jordan with partial pivoting on a small dense block */
  double M[MAX_CB][2 * MAX_CB];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { M[i][j] = A[i * n + j]; M[i][n + j] = (i == j); }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 2 * n; ++j) { double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
    const double inv = 1.0 / M[c][c];
    for (int j = 0; j < 2 * n; ++j) M[c][j] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[r][c];
      if (f != 0.0) for (int j = 0; j < 2 * n; ++j) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ainv[i * n + j] = M[i][n + j];
}

/* y = S x = (B + Dc^2) x - E C^-1 E^T x   (implicit Schur complement) */
static void schur_multiply(const linsys* s, const double* x, double* y, double* tmp_p) {
  const program* g = s->g;
  const bao_problem* p = g->p;
  /* point pass: u_j = C_j^-1 E_j^T x */
#pragma omp parallel for schedule(static) BAO_PAR(g->n_obs)
  for (int j = 0; j < p->num_points; ++j) {
    if (g->point_off[j] < 0) continue;
    double t[3] = {0, 0, 0};
    for (int64_t k = g->pt_ptr[j]; k < g->pt_ptr[j + 1]; ++k) {
      const int64_t a = g->pt_idx[k];
      const lin_obs* L = &s->L[a];
      int po, co;
      cam_offsets(g, a, &po, &co);
      double xc[MAX_CB];
      gather_c(g, L, po, co, x, xc);
      const int w = L->pose_dim + L->cam_dim + L->sens_dim;
      for (int r = 0; r < 2; ++r) {
        double jx = 0.0;
        for (int d = 0; d < w; ++d) jx += L->Jc[r][d] * xc[d];
        for (int c = 0; c < 3; ++c) t[c] += L->Jp[r][c] * jx;
      }
    }
    const double* Ci = s->Cinv + 9 * (size_t)j;
    double* u = tmp_p + g->point_off[j];
    for (int r = 0; r < 3; ++r) u[r] = Ci[3 * r] * t[0] + Ci[3 * r + 1] * t[1] + Ci[3 * r + 2] * t[2];
  }
  /* camera pass: y = Dc^2 x + sum_obs Jc^T (Jc x - Jp u), accumulated block by block */
#pragma omp parallel for schedule(dynamic, 8) BAO_PAR(g->n_obs)
  for (int b = 0; b < g->n_blk; ++b) {
    const int off = g->blk_off[b], dim = g->blk_dim[b], kind = g->blk_kind[b];
    double acc[MAX_CB];
    for (int d = 0; d < dim; ++d) acc[d] = s->Dc[off + d] * s->Dc[off + d] * x[off + d];
    for (int64_t k = g->blk_ptr[b]; k < g->blk_ptr[b + 1]; ++k) {
      const int64_t a = g->blk_idx[k];
      const lin_obs* L = &s->L[a];
      int po, co;
      cam_offsets(g, a, &po, &co);
      const int w = L->pose_dim + L->cam_dim + L->sens_dim;
      double xc[MAX_CB];
      gather_c(g, L, po, co, x, xc);
      const int xi = p->obs_point[g->obs[a]];
      const double* u = g->point_off[xi] >= 0 ? tmp_p + g->point_off[xi] : NULL;
      const int base = lin_base(L, kind);
      for (int r = 0; r < 2; ++r) {
        double v = 0.0;
        for (int d = 0; d < w; ++d) v += L->Jc[r][d] * xc[d];
        if (u) v -= L->Jp[r][0] * u[0] + L->Jp[r][1] * u[1] + L->Jp[r][2] * u[2];
        for (int d = 0; d < dim; ++d) acc[d] += L->Jc[r][base + d] * v;
      }
    }
    for (int d = 0; d < dim; ++d) y[off + d] = acc[d];
  }
  /* position priors: no point block, they add J^T J x to the camera side directly */
  for (int k = 0; k < s->n_prior; ++k) {
    const lin_prior* L = &s->P[k];
    double jx[3];
    prior_jx(L, x, jx);
    for (int r = 0; r < 3; ++r) {
      for (int d = 0; d < L->pose_dim; ++d) y[L->po + d] += L->J[r][d] * jx[r];
      for (int d = 0; d < L->sens_dim; ++d) y[L->so + d] += L->J[r][L->pose_dim + d] * jx[r];
    }
  }
}

static void precond_apply(const linsys* s, const double* r, double* z) {
  for (int b = 0; b < s->n_blk; ++b) {
    const int n = s->blk_dim[b], off = s->blk_off[b];
    const double* Mi = s->Minv + s->blk_moff[b];
    for (int i = 0; i < n; ++i) {
      double v = 0.0;
      for (int j = 0; j < n; ++j) v += Mi[i * n + j] * r[off + j];
      z[off + i] = v;
    }
  }
}

static double dot(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* Ceres ConjugateGradientsSolver (conjugate_gradients_solver.cc): preconditioned CG with the
 * Q-decrease termination  zeta = k (Q_k - Q_{k-1}) / Q_k < q_tolerance  (r_tolerance unused). */
static int pcg(const linsys* s, const double* b, double* x, int max_iter, double q_tol,
               double* ws /* 5*n_c + n_p */) {
  const int n = s->g->n_c;
  double *r = ws, *z = ws + n, *pdir = ws + 2 * n, *q = ws + 3 * n, *tmp = ws + 4 * n;
  double* tmp_p = ws + 5 * n;
  memset(x, 0, sizeof(double) * n);
  memcpy(r, b, sizeof(double) * n);
  const double norm_b = sqrt(dot(b, b, n));
  if (norm_b == 0.0) return 0;
  double rho = 1.0, Q0 = -0.5 * dot(x, b, n); /* x = 0 */
  int it = 0;
  for (it = 1; it <= max_iter; ++it) {
    precond_apply(s, r, z);
    const double last_rho = rho;
    rho = dot(r, z, n);
    if (!(rho > 0.0) || !isfinite(rho)) break;
    if (it == 1) memcpy(pdir, z, sizeof(double) * n);
    else {
      const double beta = rho / last_rho;
      for (int i = 0; i < n; ++i) pdir[i] = z[i] + beta * pdir[i];
    }
    schur_multiply(s, pdir, q, tmp_p);
    const double pq = dot(pdir, q, n);
    if (!(pq > 0.0) || !isfinite(pq)) break;
    const double alpha = rho / pq;
    for (int i = 0; i < n; ++i) x[i] += alpha * pdir[i];
    for (int i = 0; i < n; ++i) r[i] -= alpha * q[i];
    /* Q = -0.5 x^T (b + r) */
    for (int i = 0; i < n; ++i) tmp[i] = b[i] + r[i];
    const double Q1 = -0.5 * dot(x, tmp, n);
    const double zeta = it * (Q1 - Q0) / Q1;
    if (zeta < q_tol) break;
    Q0 = Q1;
    if (sqrt(dot(r, r, n)) <= 1e-30 * norm_b) break;
  }
  return it > max_iter ? max_iter : it;
}

/* ------------------------------------------------------------------------- */
/* Levenberg-Marquardt (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy) */
/* ------------------------------------------------------------------------- */

/* DENSE_SCHUR: S = B + Dc^2 - E C^-1 E^T column by column through the implicit operator, Cholesky, solve.
 * Returns 0 when S is not positive definite (the LM loop treats the step as invalid). */
static int dense_schur_solve(const linsys* s, const double* b, double* x, double* tmp_p) {
  const int n = s->g->n_c;
  double* S = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* e = (double*)calloc(n, sizeof(double));
  double* col = (double*)malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) {
    e[i] = 1.0;
    schur_multiply(s, e, col, tmp_p);
    e[i] = 0.0;
    for (int r = 0; r < n; ++r) S[(size_t)r * n + i] = col[r];
  }
  int ok = 1;
  /* in-place lower Cholesky of the lower triangle */
  for (int k = 0; k < n && ok; ++k) {
    double d = S[(size_t)k * n + k];
    if (!(d > 0.0)) { ok = 0; break; }
    d = sqrt(d);
    S[(size_t)k * n + k] = d;
    for (int i = k + 1; i < n; ++i) S[(size_t)i * n + k] /= d;
    for (int j = k + 1; j < n; ++j) {
      const double ljk = S[(size_t)j * n + k];
      for (int i = j; i < n; ++i) S[(size_t)i * n + j] -= S[(size_t)i * n + k] * ljk;
    }
  }
  if (ok) {
    for (int i = 0; i < n; ++i) {
      double v = b[i];
      for (int k = 0; k < i; ++k) v -= S[(size_t)i * n + k] * x[k];
      x[i] = v / S[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = x[i];
      for (int k = i + 1; k < n; ++k) v -= S[(size_t)k * n + i] * x[k];
      x[i] = v / S[(size_t)i * n + i];
    }
  } else {
    for (int i = 0; i < n; ++i) x[i] = NAN;
  }
  free(S); free(e); free(col);
  return ok;
}

/* DENSE_SCHUR / SPARSE_SCHUR at any size: the reduced camera system formed EXPLICITLY,
 *   S = B + Dc^2 - E C^-1 E^T,   B = sum_obs Jc^T Jc,   E C^-1 E^T = sum_points (sum_a W_a)^ C_j^-1 (sum_a' W_a')^T
 * with W_a = Jc_a^T Jp_a (w x 3) per observation, then an exact Cholesky solve (what Ceres' Schur eliminator +
 * dense / sparse Cholesky compute; only the storage of S differs: dense here). Row blocks are owned by one
 * thread each (a block's rows receive contributions only through that block's observations), so the
 * sums are deterministic without atomics. Returns 0 when S is not positive definite. */
static void obs_col_index(const program* g, const lin_obs* L, int64_t a, int* idx) {
  int po, co;
  cam_offsets(g, a, &po, &co);
  int k = 0;
  for (int d = 0; d < L->pose_dim; ++d) idx[k++] = po + d;
  for (int d = 0; d < L->cam_dim; ++d) idx[k++] = co + d;
  for (int d = 0; d < L->sens_dim; ++d) idx[k++] = L->so + d;
}

__attribute__((optimize("O3"))) static int blocked_cholesky(double* S, int n) {
  /* lower triangle, row-major, right-looking with NB-wide panels */
  enum { NB = 64 };
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = (k0 + NB < n) ? NB : n - k0;
    for (int k = k0; k < k0 + kb; ++k) {  /* diagonal block */
      double d = S[(size_t)k * n + k];
      for (int m = k0; m < k; ++m) d -= S[(size_t)k * n + m] * S[(size_t)k * n + m];
      if (!(d > 0.0)) return 0;
      d = sqrt(d);
      S[(size_t)k * n + k] = d;
      for (int i = k + 1; i < k0 + kb; ++i) {
        double v = S[(size_t)i * n + k];
        for (int m = k0; m < k; ++m) v -= S[(size_t)i * n + m] * S[(size_t)k * n + m];
        S[(size_t)i * n + k] = v / d;
      }
    }
    const int r0 = k0 + kb;
#pragma omp parallel for schedule(static)
    for (int i = r0; i < n; ++i) {  /* panel: L_ik = A_ik L_kk^-T */
      double* Si = S + (size_t)i * n;
      for (int k = k0; k < k0 + kb; ++k) {
        double v = Si[k];
        const double* Sk = S + (size_t)k * n;
        for (int m = k0; m < k; ++m) v -= Si[m] * Sk[m];
        Si[k] = v / Sk[k];
      }
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = r0; i < n; ++i) {  /* trailing update: A_ij -= L_i,panel . L_j,panel */
      double* Si = S + (size_t)i * n;
      for (int j = r0; j <= i; ++j) {
        const double* Sj = S + (size_t)j * n;
        double acc = 0.0;
        for (int m = k0; m < k0 + kb; ++m) acc += Si[m] * Sj[m];
        Si[j] -= acc;
      }
    }
  }
  return 1;
}

static int explicit_schur_solve(const linsys* s, const double* b, double* x) {
  const program* g = s->g;
  const bao_problem* p = g->p;
  const int n = g->n_c;
  double* S = (double*)calloc((size_t)n * n, sizeof(double));
  if (!S) return 0;
#pragma omp parallel for schedule(dynamic, 4)
  for (int blk = 0; blk < g->n_blk; ++blk) {
    const int off = g->blk_off[blk], dim = g->blk_dim[blk], kind = g->blk_kind[blk];
    for (int d = 0; d < dim; ++d) S[(size_t)(off + d) * n + off + d] += s->Dc[off + d] * s->Dc[off + d];
    for (int64_t k = g->blk_ptr[blk]; k < g->blk_ptr[blk + 1]; ++k) {
      const int64_t a = g->blk_idx[k];
      const lin_obs* L = &s->L[a];
      const int w = L->pose_dim + L->cam_dim + L->sens_dim;
      const int base = lin_base(L, kind);
      int idx[MAX_CB];
      obs_col_index(g, L, a, idx);
      /* B: rows of this block x every camera-side column the observation sees */
      for (int d = 0; d < dim; ++d)
        for (int c = 0; c < w; ++c)
          S[(size_t)(off + d) * n + idx[c]] += L->Jc[0][base + d] * L->Jc[0][c] + L->Jc[1][base + d] * L->Jc[1][c];
      const int xi = p->obs_point[g->obs[a]];
      if (g->point_off[xi] < 0) continue;
      /* T = W_{a,blk} C^-1 (dim x 3) */
      const double* Ci = s->Cinv + 9 * (size_t)xi;
      double W[MAX_CB][3], T[MAX_CB][3];
      for (int d = 0; d < dim; ++d)
        for (int c = 0; c < 3; ++c)
          W[d][c] = L->Jc[0][base + d] * L->Jp[0][c] + L->Jc[1][base + d] * L->Jp[1][c];
      for (int d = 0; d < dim; ++d)
        for (int c = 0; c < 3; ++c) T[d][c] = W[d][0] * Ci[c] + W[d][1] * Ci[3 + c] + W[d][2] * Ci[6 + c];
      for (int64_t k2 = g->pt_ptr[xi]; k2 < g->pt_ptr[xi + 1]; ++k2) {
        const int64_t a2 = g->pt_idx[k2];
        const lin_obs* L2 = &s->L[a2];
        const int w2 = L2->pose_dim + L2->cam_dim + L2->sens_dim;
        int idx2[MAX_CB];
        obs_col_index(g, L2, a2, idx2);
        for (int c2 = 0; c2 < w2; ++c2) {
          const double w0 = L2->Jc[0][c2] * L2->Jp[0][0] + L2->Jc[1][c2] * L2->Jp[1][0];
          const double w1 = L2->Jc[0][c2] * L2->Jp[0][1] + L2->Jc[1][c2] * L2->Jp[1][1];
          const double w2v = L2->Jc[0][c2] * L2->Jp[0][2] + L2->Jc[1][c2] * L2->Jp[1][2];
          for (int d = 0; d < dim; ++d)
            S[(size_t)(off + d) * n + idx2[c2]] -= T[d][0] * w0 + T[d][1] * w1 + T[d][2] * w2v;
        }
      }
    }
  }
  for (int k = 0; k < s->n_prior; ++k) {  /* position priors: J^T J on their pose / sensor columns */
    const lin_prior* L = &s->P[k];
    const int w = L->pose_dim + L->sens_dim;
    int idx[12];
    for (int d = 0; d < L->pose_dim; ++d) idx[d] = L->po + d;
    for (int d = 0; d < L->sens_dim; ++d) idx[L->pose_dim + d] = L->so + d;
    for (int i = 0; i < w; ++i)
      for (int j = 0; j < w; ++j)
        S[(size_t)idx[i] * n + idx[j]] += L->J[0][i] * L->J[0][j] + L->J[1][i] * L->J[1][j] + L->J[2][i] * L->J[2][j];
  }
  const int ok = blocked_cholesky(S, n);
  if (ok) {
    for (int i = 0; i < n; ++i) {
      double v = b[i];
      const double* Si = S + (size_t)i * n;
      for (int k = 0; k < i; ++k) v -= Si[k] * x[k];
      x[i] = v / Si[i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = x[i];
      for (int k = i + 1; k < n; ++k) v -= S[(size_t)k * n + i] * x[k];
      x[i] = v / S[(size_t)i * n + i];
    }
  } else {
    for (int i = 0; i < n; ++i) x[i] = NAN;
  }
  free(S);
  return ok;
}

/* bundle_adjustment_ceres.h:68-69 (CPU thresholds): DENSE_SCHUR up to 50 images, SPARSE_SCHUR up to 1000 */
#define BAO_DENSE_MAX_IMAGES 50
#define BAO_SPARSE_MAX_IMAGES 1000
#define BAO_DENSE_MAX_DIM 1024
#define BAO_EXPLICIT_MAX_DIM 32768

BAO_API void bao_options_init(bao_options* o) {
  /* COLMAP's CeresBundleAdjustmentOptions ctor (bundle_adjustment_ceres.cc:102-115) over
   * Ceres Solver::Options defaults */
  o->max_num_iterations = 100;
  o->max_linear_solver_iterations = 200;
  o->function_tolerance = 0.0;
  o->gradient_tolerance = 1e-4;
  o->parameter_tolerance = 0.0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->eta = 1e-1;
  o->max_num_consecutive_invalid_steps = 10;
  o->jacobi_scaling = 1;
  o->num_threads = 0;
  o->max_log = 0;
  o->loss_type = BAO_LOSS_TRIVIAL; /* bundle_adjustment_ceres.h:42-51 */
  o->loss_scale = 1.0;
  o->linear_solver_type = 0;
}

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

BAO_API int bao_solve(bao_problem* p, const bao_options* opt, bao_result* res) {
#ifdef _OPENMP
  if (opt->num_threads > 0) omp_set_num_threads(opt->num_threads);
#endif
  program g;
  program_build(&g, p);
  g.loss_type = opt->loss_type;
  g.loss_scale = opt->loss_scale;
  memset(res, 0, offsetof(bao_result, log_cost));
  int n_prior = 0;
  for (int k = 0; k < p->num_priors; ++k) n_prior += prior_active(&g, k);
  res->num_residuals = (int32_t)(2 * g.n_obs) + 3 * n_prior;
  res->num_effective_parameters = g.n_c + g.n_p;
  res->termination_type = BAO_FAILURE;
  if (g.n_obs == 0) { program_free(&g); return 0; }

  const int nc = g.n_c, np = g.n_p;
  linsys s;
  memset(&s, 0, sizeof(s));
  s.g = &g;
  s.L = (lin_obs*)malloc(sizeof(lin_obs) * g.n_obs);
  s.Dc = (double*)calloc(nc + 1, sizeof(double));
  s.Dp = (double*)calloc(np + 1, sizeof(double));
  s.Cinv = (double*)calloc((size_t)p->num_points * 9 + 1, sizeof(double));
  s.P = (lin_prior*)malloc(sizeof(lin_prior) * (n_prior + 1));
  s.n_prior = n_prior;
  /* camera-side blocks (same order as program.blk_*) */
  s.blk_off = (int*)malloc(sizeof(int) * (g.n_blk + 1));
  s.blk_dim = (int*)malloc(sizeof(int) * (g.n_blk + 1));
  s.blk_moff = (int64_t*)malloc(sizeof(int64_t) * (g.n_blk + 1));
  int64_t moff = 0;
  for (int b = 0; b < g.n_blk; ++b) {
    s.blk_off[b] = g.blk_off[b]; s.blk_dim[b] = g.blk_dim[b]; s.blk_moff[b] = moff;
    moff += g.blk_dim[b] * g.blk_dim[b];
  }
  s.n_blk = g.n_blk;
  s.Minv = (double*)calloc(moff + 1, sizeof(double));
  int* blk_of = (int*)malloc(sizeof(int) * (nc + 1)); /* camera-side index -> block */
  for (int b = 0; b < s.n_blk; ++b) for (int d = 0; d < s.blk_dim[b]; ++d) blk_of[s.blk_off[b] + d] = b;

  double* scale_c = (double*)malloc(sizeof(double) * (nc + 1));
  double* scale_p = (double*)malloc(sizeof(double) * (np + 1));
  double* gc = (double*)malloc(sizeof(double) * (nc + 1));
  double* gp = (double*)malloc(sizeof(double) * (np + 1));
  double* diag_c = (double*)malloc(sizeof(double) * (nc + 1));
  double* diag_p = (double*)malloc(sizeof(double) * (np + 1));
  double* rhs = (double*)malloc(sizeof(double) * (nc + 1));
  double* dc = (double*)malloc(sizeof(double) * (nc + 1));
  double* dp = (double*)malloc(sizeof(double) * (np + 1));
  double* ws = (double*)malloc(sizeof(double) * ((size_t)5 * nc + np + 8));
  double* Mblk = (double*)calloc(moff + 1, sizeof(double));
  double* nposes = (double*)malloc(sizeof(double) * 7 * (size_t)p->num_poses + 8);
  /* candidate values of the variable sensor_from_rig blocks (NULL: none is variable) */
  double* nsens = NULL;
  for (int k = 0; k < p->num_sensors; ++k)
    if (g.sens_off[k] >= 0 && !nsens) nsens = (double*)malloc(sizeof(double) * 7 * (size_t)p->num_sensors + 8);
  double* ncams = (double*)malloc(sizeof(double) * BAO_CAM_STRIDE * (size_t)p->num_cams + 8);
  double* npoints = (double*)malloc(sizeof(double) * 3 * (size_t)p->num_points + 8);

  double radius = opt->initial_trust_region_radius;
  double decrease_factor = 2.0;
  int invalid_steps = 0;
  int need_linearize = 1;
  int have_scale = 0;
  double cost = 0.0;
  const double t_start = now_s();

  for (int iter = 0;; ++iter) {
    if (need_linearize) {
      /* residuals + Jacobians (Evaluate) */
      double c = 0.0;
#pragma omp parallel for reduction(+ : c) schedule(static) BAO_PAR(g.n_obs)
      for (int64_t a = 0; a < g.n_obs; ++a) {
        linearize_obs(&g, p->poses, p->cams, p->points, a, &s.L[a], 1);
        c += s.L[a].cost;
      }
      for (int k = 0, n = 0; k < p->num_priors; ++k) {
        if (!prior_active(&g, k)) continue;
        linearize_prior(&g, p->poses, k, &s.P[n], 1);
        c += s.P[n++].cost;
      }
      cost = c;
      if (iter == 0) res->initial_cost = cost;
      /* gradient (unscaled) g = J^T r, and column norms: camera side per block, point side per point */
#pragma omp parallel for schedule(dynamic, 8) BAO_PAR(g.n_obs)
      for (int b = 0; b < g.n_blk; ++b) {
        const int off = g.blk_off[b], dim = g.blk_dim[b], kind = g.blk_kind[b];
        double ga[MAX_CB] = {0}, da[MAX_CB] = {0};
        for (int64_t k = g.blk_ptr[b]; k < g.blk_ptr[b + 1]; ++k) {
          const lin_obs* L = &s.L[g.blk_idx[k]];
          const int base = lin_base(L, kind);
          for (int r = 0; r < 2; ++r)
            for (int d = 0; d < dim; ++d) { const double v = L->Jc[r][base + d]; ga[d] += v * L->r[r]; da[d] += v * v; }
        }
        for (int d = 0; d < dim; ++d) { gc[off + d] = ga[d]; diag_c[off + d] = da[d]; }
      }
      for (int k = 0; k < s.n_prior; ++k) {
        const lin_prior* L = &s.P[k];
        for (int r = 0; r < 3; ++r) {
          for (int d = 0; d < L->pose_dim; ++d) { const double v = L->J[r][d]; gc[L->po + d] += v * L->r[r]; diag_c[L->po + d] += v * v; }
          for (int d = 0; d < L->sens_dim; ++d) { const double v = L->J[r][L->pose_dim + d]; gc[L->so + d] += v * L->r[r]; diag_c[L->so + d] += v * v; }
        }
      }
#pragma omp parallel for schedule(static) BAO_PAR(g.n_obs)
      for (int j = 0; j < p->num_points; ++j) {
        const int pto = g.point_off[j];
        if (pto < 0) continue;
        double ga[3] = {0, 0, 0}, da[3] = {0, 0, 0};
        for (int64_t k = g.pt_ptr[j]; k < g.pt_ptr[j + 1]; ++k) {
          const lin_obs* L = &s.L[g.pt_idx[k]];
          for (int r = 0; r < 2; ++r)
            for (int c2 = 0; c2 < 3; ++c2) { ga[c2] += L->Jp[r][c2] * L->r[r]; da[c2] += L->Jp[r][c2] * L->Jp[r][c2]; }
        }
        for (int c2 = 0; c2 < 3; ++c2) { gp[pto + c2] = ga[c2]; diag_p[pto + c2] = da[c2]; }
      }
      /* convergence test on the projected gradient: ||x - Plus(x, -g)||_inf */
      {
        for (int i = 0; i < nc; ++i) dc[i] = -gc[i];
        for (int i = 0; i < np; ++i) dp[i] = -gp[i];
        apply_step(&g, dc, dp, p->poses, p->cams, p->points, nposes, ncams, npoints);
        double gmax = 0.0;
        if (nsens) {
          apply_step_sensors(&g, dc, p->sensors, nsens);
          for (size_t i = 0; i < 7 * (size_t)p->num_sensors; ++i) gmax = fmax(gmax, fabs(nsens[i] - p->sensors[i]));
        }
        for (size_t i = 0; i < 7 * (size_t)p->num_poses; ++i) gmax = fmax(gmax, fabs(nposes[i] - p->poses[i]));
        for (size_t i = 0; i < BAO_CAM_STRIDE * (size_t)p->num_cams; ++i) gmax = fmax(gmax, fabs(ncams[i] - p->cams[i]));
        for (size_t i = 0; i < 3 * (size_t)p->num_points; ++i) gmax = fmax(gmax, fabs(npoints[i] - p->points[i]));
        if (gmax <= opt->gradient_tolerance) { res->termination_type = BAO_CONVERGENCE; res->num_iterations = iter; break; }
      }
      /* Jacobi scaling, computed once from the initial Jacobian: 1 / (1 + ||col||) */
      if (!have_scale) {
        for (int i = 0; i < nc; ++i) scale_c[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(diag_c[i])) : 1.0;
        for (int i = 0; i < np; ++i) scale_p[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(diag_p[i])) : 1.0;
        have_scale = 1;
      }
      /* scale the Jacobian columns in place */
#pragma omp parallel for schedule(static) BAO_PAR(g.n_obs)
      for (int64_t a = 0; a < g.n_obs; ++a) {
        lin_obs* L = &s.L[a];
        int po, co; cam_offsets(&g, a, &po, &co);
        const int pto = g.point_off[p->obs_point[g.obs[a]]];
        for (int r = 0; r < 2; ++r) {
          for (int d = 0; d < L->pose_dim; ++d) L->Jc[r][d] *= scale_c[po + d];
          for (int d = 0; d < L->cam_dim; ++d) L->Jc[r][L->pose_dim + d] *= scale_c[co + d];
          for (int d = 0; d < L->sens_dim; ++d) L->Jc[r][L->pose_dim + L->cam_dim + d] *= scale_c[L->so + d];
          if (pto >= 0) for (int c2 = 0; c2 < 3; ++c2) L->Jp[r][c2] *= scale_p[pto + c2];
        }
      }
      for (int k = 0; k < s.n_prior; ++k) {
        lin_prior* L = &s.P[k];
        for (int r = 0; r < 3; ++r) {
          for (int d = 0; d < L->pose_dim; ++d) L->J[r][d] *= scale_c[L->po + d];
          for (int d = 0; d < L->sens_dim; ++d) L->J[r][L->pose_dim + d] *= scale_c[L->so + d];
        }
      }
      for (int i = 0; i < nc; ++i) { diag_c[i] *= scale_c[i] * scale_c[i]; gc[i] *= scale_c[i]; }
      for (int i = 0; i < np; ++i) { diag_p[i] *= scale_p[i] * scale_p[i]; gp[i] *= scale_p[i]; }
      need_linearize = 0;
    }
    if (iter >= opt->max_num_iterations) { res->termination_type = BAO_NO_CONVERGENCE; res->num_iterations = iter; break; }

    /* LM diagonal D = sqrt(clamp(diag(J^T J)) / radius) */
    for (int i = 0; i < nc; ++i) s.Dc[i] = sqrt(fmin(fmax(diag_c[i], opt->min_lm_diagonal), opt->max_lm_diagonal) / radius);
    for (int i = 0; i < np; ++i) s.Dp[i] = sqrt(fmin(fmax(diag_p[i], opt->min_lm_diagonal), opt->max_lm_diagonal) / radius);

    /* point blocks C_j = E_j^T E_j + Dp^2 and their inverses; camera blocks of B + Dc^2 */
    memset(Mblk, 0, sizeof(double) * moff);
#pragma omp parallel for schedule(static) BAO_PAR(g.n_obs)
    for (int j = 0; j < p->num_points; ++j) {
      if (g.point_off[j] < 0) continue;
      double C[9] = {0};
      for (int64_t k = g.pt_ptr[j]; k < g.pt_ptr[j + 1]; ++k) {
        const lin_obs* L = &s.L[g.pt_idx[k]];
        for (int r = 0; r < 2; ++r)
          for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y) C[3 * x + y] += L->Jp[r][x] * L->Jp[r][y];
      }
      for (int x = 0; x < 3; ++x) C[4 * x] += s.Dp[g.point_off[j] + x] * s.Dp[g.point_off[j] + x];
      invert_sym(C, 3, s.Cinv + 9 * (size_t)j);
    }
    /* position priors add J_b^T J_b to their blocks of B */
    for (int k = 0; k < s.n_prior; ++k) {
      const lin_prior* L = &s.P[k];
      for (int part = 0; part < 2; ++part) {
        const int off = part == 0 ? L->po : L->so, dim = part == 0 ? L->pose_dim : L->sens_dim;
        const int base = part == 0 ? 0 : L->pose_dim;
        if (off < 0 || dim == 0) continue;
        double* M = Mblk + s.blk_moff[blk_of[off]];
        for (int r = 0; r < 3; ++r)
          for (int x = 0; x < dim; ++x)
            for (int y = 0; y < dim; ++y) M[x * dim + y] += L->J[r][base + x] * L->J[r][base + y];
      }
    }
    /* SCHUR_JACOBI: block diagonal of S = B + Dc^2 - E C^-1 E^T, per camera-side block:
     * B_bb = sum_o J_b,o^T J_b,o ; correction = sum over pairs (o, o') of observations of the
     * same point that share the block of W_o C^-1 W_o'^T with W = J_b^T J_p */
#pragma omp parallel for schedule(dynamic, 8) BAO_PAR(g.n_obs)
    for (int b = 0; b < s.n_blk; ++b) {
      const int off = g.blk_off[b], dim = g.blk_dim[b], kind = g.blk_kind[b];
      double* M = Mblk + s.blk_moff[b];
      for (int64_t k = g.blk_ptr[b]; k < g.blk_ptr[b + 1]; ++k) {
        const int64_t a1 = g.blk_idx[k];
        const lin_obs* L1 = &s.L[a1];
        const int b1 = lin_base(L1, kind);
        for (int r = 0; r < 2; ++r)
          for (int x = 0; x < dim; ++x)
            for (int y = 0; y < dim; ++y) M[x * dim + y] += L1->Jc[r][b1 + x] * L1->Jc[r][b1 + y];
        const int j = p->obs_point[g.obs[a1]];
        if (g.point_off[j] < 0) continue;
        const double* Ci = s.Cinv + 9 * (size_t)j;
        double W1[MAX_CB][3], T[MAX_CB][3];
        for (int x = 0; x < dim; ++x)
          for (int c = 0; c < 3; ++c) W1[x][c] = L1->Jc[0][b1 + x] * L1->Jp[0][c] + L1->Jc[1][b1 + x] * L1->Jp[1][c];
        for (int x = 0; x < dim; ++x)
          for (int c = 0; c < 3; ++c) T[x][c] = W1[x][0] * Ci[c] + W1[x][1] * Ci[3 + c] + W1[x][2] * Ci[6 + c];
        for (int64_t k2 = g.pt_ptr[j]; k2 < g.pt_ptr[j + 1]; ++k2) {
          const int64_t a2 = g.pt_idx[k2];
          int po2, co2; cam_offsets(&g, a2, &po2, &co2);
          const lin_obs* L2 = &s.L[a2];
          if ((kind == 0 ? po2 : (kind == 1 ? co2 : L2->so)) != off) continue;
          const int b2 = lin_base(L2, kind);
          for (int y = 0; y < dim; ++y) {
            double W2[3];
            for (int c = 0; c < 3; ++c) W2[c] = L2->Jc[0][b2 + y] * L2->Jp[0][c] + L2->Jc[1][b2 + y] * L2->Jp[1][c];
            for (int x = 0; x < dim; ++x) M[x * dim + y] -= T[x][0] * W2[0] + T[x][1] * W2[1] + T[x][2] * W2[2];
          }
        }
      }
      for (int d = 0; d < dim; ++d) M[d * dim + d] += s.Dc[off + d] * s.Dc[off + d];
      invert_sym(M, dim, s.Minv + s.blk_moff[b]);
    }

    /* reduced right-hand side: solve (J^T J + D^2) y = J^T r ; step = -y.
     * rhs = g_c - E C^-1 g_p */
    {
      double* u = ws; /* C^-1 g_p per point */
      for (int j = 0; j < p->num_points; ++j) {
        if (g.point_off[j] < 0) continue;
        const double* Ci = s.Cinv + 9 * (size_t)j;
        const double* gj = gp + g.point_off[j];
        for (int r = 0; r < 3; ++r) u[g.point_off[j] + r] = Ci[3 * r] * gj[0] + Ci[3 * r + 1] * gj[1] + Ci[3 * r + 2] * gj[2];
      }
#pragma omp parallel for schedule(dynamic, 8) BAO_PAR(g.n_obs)
      for (int b = 0; b < g.n_blk; ++b) {
        const int off = g.blk_off[b], dim = g.blk_dim[b], kind = g.blk_kind[b];
        double acc[MAX_CB];
        for (int d = 0; d < dim; ++d) acc[d] = gc[off + d];
        for (int64_t k = g.blk_ptr[b]; k < g.blk_ptr[b + 1]; ++k) {
          const int64_t a = g.blk_idx[k];
          const lin_obs* L = &s.L[a];
          const int pto = g.point_off[p->obs_point[g.obs[a]]];
          if (pto < 0) continue;
          const int base = lin_base(L, kind);
          for (int r = 0; r < 2; ++r) {
            const double v = L->Jp[r][0] * u[pto] + L->Jp[r][1] * u[pto + 1] + L->Jp[r][2] * u[pto + 2];
            for (int d = 0; d < dim; ++d) acc[d] -= L->Jc[r][base + d] * v;
          }
        }
        for (int d = 0; d < dim; ++d) rhs[off + d] = acc[d];
      }
    }
    int lin_iters = 0;
    /* 1 DENSE_SCHUR, 3 SPARSE_SCHUR: exact solve of the explicitly formed S (the tiers differ only in how
     * Ceres stores S); 2 AUTO: the reference's rule on the image count (pose blocks stand in for images) */
    const int lst = opt->linear_solver_type;
    const int exact = (lst == 1 || lst == 3 || (lst == 2 && p->num_poses <= BAO_SPARSE_MAX_IMAGES)) &&
                      nc <= BAO_EXPLICIT_MAX_DIM;
    /* BAO_DENSE_BY_PRODUCTS=1: the round-2 formation (n_c operator products), kept as a cross-check */
    const char* e_prod = getenv("BAO_DENSE_BY_PRODUCTS");
    const int operator_products = e_prod && atoi(e_prod) != 0;
    res->linear_solver_used = exact ? (lst == 3 || (lst == 2 && p->num_poses > BAO_DENSE_MAX_IMAGES) ? 3 : 1) : 0;
    if (nc > 0 && exact && operator_products && nc <= BAO_DENSE_MAX_DIM) { dense_schur_solve(&s, rhs, dc, ws); lin_iters = 1; }
    else if (nc > 0 && exact) { explicit_schur_solve(&s, rhs, dc); lin_iters = 1; }
    else if (nc > 0) lin_iters = pcg(&s, rhs, dc, opt->max_linear_solver_iterations, opt->eta, ws);
    res->total_linear_iterations += lin_iters;
    /* back-substitution: y_p = C^-1 (g_p - E^T y_c) */
    for (int j = 0; j < p->num_points; ++j) {
      if (g.point_off[j] < 0) continue;
      double t[3] = {gp[g.point_off[j]], gp[g.point_off[j] + 1], gp[g.point_off[j] + 2]};
      for (int64_t k = g.pt_ptr[j]; k < g.pt_ptr[j + 1]; ++k) {
        const int64_t a = g.pt_idx[k];
        const lin_obs* L = &s.L[a];
        int po, co; cam_offsets(&g, a, &po, &co);
        double xc[MAX_CB];
        gather_c(&g, L, po, co, dc, xc);
        const int w = L->pose_dim + L->cam_dim + L->sens_dim;
        for (int r = 0; r < 2; ++r) {
          double jx = 0.0;
          for (int d = 0; d < w; ++d) jx += L->Jc[r][d] * xc[d];
          for (int c = 0; c < 3; ++c) t[c] -= L->Jp[r][c] * jx;
        }
      }
      const double* Ci = s.Cinv + 9 * (size_t)j;
      for (int r = 0; r < 3; ++r) dp[g.point_off[j] + r] = Ci[3 * r] * t[0] + Ci[3 * r + 1] * t[1] + Ci[3 * r + 2] * t[2];
    }
    /* step = -y ; model cost change = -(J step) . (r + J step / 2) */
    for (int i = 0; i < nc; ++i) dc[i] = -dc[i];
    for (int i = 0; i < np; ++i) dp[i] = -dp[i];
    double model_change = 0.0;
#pragma omp parallel for reduction(+ : model_change) schedule(static) BAO_PAR(g.n_obs)
    for (int64_t a = 0; a < g.n_obs; ++a) {
      const lin_obs* L = &s.L[a];
      int po, co; cam_offsets(&g, a, &po, &co);
      const int pto = g.point_off[p->obs_point[g.obs[a]]];
      double xc[MAX_CB];
      gather_c(&g, L, po, co, dc, xc);
      const int w = L->pose_dim + L->cam_dim + L->sens_dim;
      for (int r = 0; r < 2; ++r) {
        double m = 0.0;
        for (int d = 0; d < w; ++d) m += L->Jc[r][d] * xc[d];
        if (pto >= 0) m += L->Jp[r][0] * dp[pto] + L->Jp[r][1] * dp[pto + 1] + L->Jp[r][2] * dp[pto + 2];
        model_change -= m * (L->r[r] + 0.5 * m);
      }
    }
    for (int k = 0; k < s.n_prior; ++k) {
      double m[3];
      prior_jx(&s.P[k], dc, m);
      for (int r = 0; r < 3; ++r) model_change -= m[r] * (s.P[k].r[r] + 0.5 * m[r]);
    }
    int accepted = 0;
    double new_cost = cost;
    if (!(model_change > 0.0) || !isfinite(model_change)) {
      if (++invalid_steps >= opt->max_num_consecutive_invalid_steps) { res->termination_type = BAO_FAILURE; res->num_iterations = iter + 1; break; }
      radius /= decrease_factor; decrease_factor *= 2.0;
    } else {
      invalid_steps = 0;
      /* undo the Jacobi scaling of the step, x_plus = Plus(x, step) */
      for (int i = 0; i < nc; ++i) ws[i] = dc[i] * scale_c[i];
      double* dps = ws + nc;
      for (int i = 0; i < np; ++i) dps[i] = dp[i] * scale_p[i];
      apply_step(&g, ws, dps, p->poses, p->cams, p->points, nposes, ncams, npoints);
      if (nsens) {
        apply_step_sensors(&g, ws, p->sensors, nsens);
        g.sensors = nsens;  /* the candidate reads the candidate sensor_from_rig values */
      }
      new_cost = evaluate_cost(&g, nposes, ncams, npoints);
      g.sensors = p->sensors;
      const double rho = (cost - new_cost) / model_change;
      if (rho > opt->min_relative_decrease) {
        accepted = 1;
        memcpy(p->poses, nposes, sizeof(double) * 7 * (size_t)p->num_poses);
        memcpy(p->cams, ncams, sizeof(double) * BAO_CAM_STRIDE * (size_t)p->num_cams);
        memcpy(p->points, npoints, sizeof(double) * 3 * (size_t)p->num_points);
        if (nsens) memcpy(p->sensors, nsens, sizeof(double) * 7 * (size_t)p->num_sensors);
        const double t = 2.0 * rho - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        radius = fmin(opt->max_trust_region_radius, radius);
        decrease_factor = 2.0;
        res->num_successful_steps++;
        need_linearize = 1;
        const double change = fabs(cost - new_cost);
        if (change <= opt->function_tolerance * cost && opt->function_tolerance > 0) {
          cost = new_cost; res->termination_type = BAO_CONVERGENCE; res->num_iterations = iter + 1;
          if (res->num_logged < opt->max_log) { res->log_cost[res->num_logged] = cost; res->log_radius[res->num_logged] = radius; res->log_linear_iters[res->num_logged++] = lin_iters; }
          break;
        }
      } else {
        radius /= decrease_factor; decrease_factor *= 2.0;
      }
    }
    if (res->num_logged < opt->max_log) {
      res->log_cost[res->num_logged] = accepted ? new_cost : cost;
      res->log_radius[res->num_logged] = radius;
      res->log_linear_iters[res->num_logged++] = lin_iters;
    }
    if (radius < opt->min_trust_region_radius) { res->termination_type = BAO_CONVERGENCE; res->num_iterations = iter + 1; break; }
  }
  res->final_cost = evaluate_cost(&g, p->poses, p->cams, p->points);
  res->lm_seconds = now_s() - t_start;
  for (int k = 0; k < p->num_sensors && nsens; ++k) {
    if (g.sens_off[k] < 0) continue;
    double* q = p->sensors + 7 * (size_t)k;
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] /= n;
  }
  free(nsens);
  /* quaternions are re-normalised when written back (bundle_adjustment_ceres.cc:491,508) */
  for (int i = 0; i < p->num_poses; ++i) {
    if (g.pose_off[i] < 0) continue;
    double* q = p->poses + 7 * (size_t)i;
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] /= n;
  }

  free(s.P);
  free(s.L); free(s.Dc); free(s.Dp); free(s.Cinv); free(s.Minv); free(s.blk_off); free(s.blk_dim);
  free(s.blk_moff); free(blk_of); free(scale_c); free(scale_p); free(gc); free(gp); free(diag_c);
  free(diag_p); free(rhs); free(dc); free(dp); free(ws); free(Mblk); free(nposes); free(ncams);
  free(npoints);
  program_free(&g);
  return 0;
}

BAO_API int bao_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
