/*
 * colmap_amd_ba.h -- C ABI of the MI355X-native bundle-adjustment solve.
 *
 * Drop-in boundary: colmap::BundleAdjuster::Solve() behind
 * CreateDefaultBundleAdjuster (reference src/colmap/estimators/bundle_adjustment.h:212-234,
 * bundle_adjustment.cc:314-334), as a third BundleAdjustmentBackend value next to CERES and
 * CASPAR (bundle_adjustment.h:60). The adapter (a BundleAdjuster subclass, see
 * INTEGRATION.md) flattens the Reconstruction into `ba_problem` exactly like
 * CasparBundleAdjuster does for its solver (bundle_adjustment_caspar.cc:61-377), calls
 * ba_solve(), and writes the variable blocks back (:767-801). ba_solve() replaces
 * ceres::Solve (bundle_adjustment_ceres.cc:582) with its COLMAP-side cost functions
 * (cost_functions/reprojection_error.h:61-212, sensor/models_jacobian.h:139-321): a
 * Levenberg-Marquardt loop whose linear step is an implicit-Schur preconditioned CG
 * (Ceres ITERATIVE_SCHUR + SCHUR_JACOBI, bundle_adjustment_ceres.cc:203-213) on the GPU.
 *
 * Plain C types only; arrays are host memory, row-major, updated in place for variable blocks.
 * Returns 0 on success; ba_last_error() holds the message otherwise. No CPU fallback.
 */
#ifndef COLMAP_AMD_BA_H_
#define COLMAP_AMD_BA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BA_CAM_STRIDE 16 /* doubles reserved per camera parameter block (RAD_TAN_THIN_PRISM_FISHEYE has 16) */

/* colmap::CameraModelId values of the supported models (sensor/models.h:90-111) */
enum {
  BA_SIMPLE_PINHOLE = 0, BA_PINHOLE = 1, BA_SIMPLE_RADIAL = 2, BA_RADIAL = 3, BA_OPENCV = 4,
  BA_OPENCV_FISHEYE = 5, BA_FULL_OPENCV = 6, BA_FOV = 7, BA_SIMPLE_RADIAL_FISHEYE = 8, BA_RADIAL_FISHEYE = 9,
  BA_THIN_PRISM_FISHEYE = 10, BA_RAD_TAN_THIN_PRISM_FISHEYE = 11, BA_SIMPLE_DIVISION = 12, BA_DIVISION = 13, BA_SIMPLE_FISHEYE = 14, BA_FISHEYE = 15, BA_EUCM = 16,
  BA_EQUIRECTANGULAR = 17
};

#define BA_POSE_ROT_CONST 4
typedef struct ba_problem {
  int32_t num_poses, num_cams, num_points;
  int64_t num_obs;
  double* poses;      /* [num_poses][7]  Rigid3d::params: qx qy qz qw tx ty tz (geometry/rigid3.h:46-70) */
  double* cams;       /* [num_cams][BA_CAM_STRIDE]  Camera::params (scene/camera.h:61) */
  int32_t* cam_model; /* [num_cams] */
  double* points;     /* [num_points][3]  Point3D::xyz */
  int32_t* obs_pose;  /* [num_obs] index of the cam_from_world pose block */
  int32_t* obs_cam;   /* [num_obs] index of the camera (intrinsics) block */
  int32_t* obs_point; /* [num_obs] index of the 3-D point block */
  double* obs_xy;     /* [num_obs][2] Point2D::xy */
  /* constant-ness, as ceres::Problem would hold it after DefaultBundleAdjuster's ctor */
  uint8_t* pose_const;   /* [num_poses] SetParameterBlockConstant */
  int8_t* pose_fixed_t;  /* [num_poses] -1, or the translation coordinate (0..2) the gauge holds
                            (SubsetManifold, bundle_adjustment_ceres.cc:402-415); + BA_POSE_ROT_CONST (4)
                            when the rotation is held too (options.constant_rig_from_world_rotation,
                            :404-408,513-516): 4..6 = rotation and that coordinate, 7 = rotation only */
  uint8_t* cam_const;    /* [num_cams][BA_CAM_STRIDE] per-parameter mask (SubsetManifold, :419-469) */
  uint8_t* point_const;  /* [num_points] */
  /* Rigs (AddImageWithNonTrivialFrame, bundle_adjustment_ceres.cc:752-822): the pose block of such an
   * observation is the frame's rig_from_world and the camera sees sensor_from_rig * rig_from_world * X.
   * A sensor_from_rig is constant (RigReprojErrorConstantRigCostFunctor, cost_functions/
   * reprojection_error.h:386-417) or, with options.refine_sensor_from_rig, a 7-parameter block of its
   * own (RigReprojErrorCostFunctor, :344-384) that is updated in place. */
  int32_t num_sensors;
  double* sensors;       /* [num_sensors][7] Rigid3d::params (in/out for variable sensors), or NULL */
  int32_t* obs_sensor;   /* [num_obs] index into sensors, -1 = trivial frame; NULL = all trivial */
  uint8_t* sensor_const; /* [num_sensors] 1 = constant block; NULL = every sensor_from_rig is constant */
  /* Position priors (PosePriorBundleAdjuster::AddImagePosePriorToProblem, bundle_adjustment_ceres.cc:
   * 986-1038): per prior a 3-residual block sqrt_info * (position + R(q)^-1 t) on the pose block
   * (AbsolutePosePositionPriorCostFunctor, cost_functions/pose_prior.h:76-96: the image is the
   * reference sensor of its frame) or on sensor_from_rig * rig_from_world
   * (AbsoluteRigPosePositionPriorCostFunctor, :98-129) with CovarianceWeightedCostFunctor's left square
   * root of the information matrix (cost_functions/utils.h:124-165) and its own loss function
   * (prior_position_loss_function_type / _scale). Priors whose blocks are all constant are ignored. */
  int32_t num_priors;
  int32_t* prior_pose;     /* [num_priors] pose block (cam_from_world, or the frame's rig_from_world) */
  int32_t* prior_sensor;   /* [num_priors] index into sensors, -1 = no sensor_from_rig; NULL = all -1 */
  double* prior_position;  /* [num_priors][3] position in the (normalised) world frame */
  double* prior_sqrt_info; /* [num_priors][9] row-major: cov^-1 = L L^T, this is L^T */
  int32_t prior_loss_type; /* BA_LOSS_* */
  double prior_loss_scale;
} ba_problem;

/* ceres::Solver::Options fields that reach the solve (COLMAP's values:
 * bundle_adjustment_ceres.cc:102-115; the rest are Ceres defaults). */
struct ba_iteration_summary;
typedef struct ba_options {
  int32_t max_num_iterations;           /* 100 */
  int32_t max_linear_solver_iterations; /* 200 */
  double function_tolerance;            /* 0 */
  double gradient_tolerance;            /* 1e-4 */
  double parameter_tolerance;           /* 0 */
  double initial_trust_region_radius;   /* 1e4 */
  double max_trust_region_radius;       /* 1e16 */
  double min_trust_region_radius;       /* 1e-32 */
  double min_relative_decrease;         /* 1e-3 */
  double min_lm_diagonal;               /* 1e-6 */
  double max_lm_diagonal;               /* 1e32 */
  double eta;                           /* 1e-1: inexact-Newton forcing, CG Q-tolerance */
  int32_t max_num_consecutive_invalid_steps; /* 10 */
  int32_t jacobi_scaling;               /* 1 */
  int32_t num_threads;                  /* unused on the GPU */
  int32_t max_log;                      /* capacity of the log arrays in ba_result */
  /* CeresBundleAdjustmentOptions::loss_function_type / _scale (bundle_adjustment_ceres.h:42-51),
   * applied to every reprojection residual like CreateLossFunction (bundle_adjustment_ceres.cc:66-80) */
  int32_t loss_type;                    /* BA_LOSS_TRIVIAL */
  double loss_scale;                    /* 1.0 */
  /* Linear solver of the LM step (ceres::LinearSolverType as CreateSolverOptions picks it,
   * bundle_adjustment_ceres.cc:203-213): BA_SOLVER_ITERATIVE_SCHUR = the implicit Schur complement with
   * PCG + Schur-Jacobi (what BASELINE.json benchmarks; default); BA_SOLVER_DENSE_SCHUR and
   * BA_SOLVER_SPARSE_SCHUR = the exact tiers: the reduced camera system S = B + D^2 - E C^-1 E^T formed
   * explicitly on the device (one wave per 3-D point scatters J_a^T (delta_ab - G_ab) J_b into the camera
   * pairs the point connects) and solved by a blocked Cholesky factorisation on the f64 matrix cores
   * (colmap_amd/csrc/ba_schur_explicit.hip). Both names run the same code: Ceres' two tiers differ only in
   * how S is stored, here it is dense in HBM up to a camera-side dimension of 32 768 (8.6 GB).
   * BA_SOLVER_AUTO = the reference's rule with its CPU thresholds (bundle_adjustment_ceres.h:68-69):
   * DENSE_SCHUR up to 50 images, SPARSE_SCHUR up to 1000, else iterative -- on the number of pose blocks,
   * which is what this flat interface knows; the BundleAdjuster adapters resolve AUTO on the image count
   * themselves. ba_result.linear_solver_used reports the tier that ran (an exact tier requested beyond its
   * size limit, or in an image-sharded solve beyond dimension 1024, runs the iterative one). */
  int32_t linear_solver_type;           /* BA_SOLVER_ITERATIVE_SCHUR */
  /* Storage precision of the Jacobian columns the PCG operator S x streams (MI355X option, no reference
   * counterpart). BA_OPERATOR_F64 (default): everything reads the fp64 columns; the LM / PCG trajectory
   * follows the fp64 oracle to 1e-7. BA_OPERATOR_F32: the inner, inexact (eta = 0.1) CG solve reads fp32
   * copies of the scaled columns with fp64 accumulation -- half the bytes per CG iteration; cost, gradient,
   * Schur-Jacobi blocks, reduced right-hand side, back-substitution and step evaluation stay fp64, so the
   * converged solution is unchanged (final cost to 1e-8 relative under a tight gradient tolerance) while
   * intermediate costs follow the fp64 trajectory only to ~1e-6 relative. Ignored (fp64) with variable
   * sensor_from_rig blocks, tracks longer than a point tile, and in sharded solves. */
  int32_t operator_precision;           /* BA_OPERATOR_F64 */
  /* Iteration callback = ceres::Solver::Options::callbacks as COLMAP uses them: the controller's
   * BundleAdjustmentIterationCallback asks CheckIfStopped() after every iteration and ends the solve with
   * SOLVER_TERMINATE_SUCCESSFULLY (controllers/bundle_adjustment.cc:40-57,84-86). Called on the calling thread after
   * the initial evaluation (iteration 0) and after every LM iteration, from the host loop between two iterations'
   * launches; the return value is BA_CALLBACK_CONTINUE, BA_CALLBACK_TERMINATE (-> termination_type BA_USER_SUCCESS) or
   * BA_CALLBACK_ABORT (-> BA_USER_FAILURE). Either way the parameter blocks hold the last ACCEPTED step
   * (estimators/bundle_adjustment.h:63-69: USER_SUCCESS is a usable solution). In a sharded solve every rank calls it
   * and all ranks must return the same value. NULL = none. */
  int (*iteration_callback)(void* user, const struct ba_iteration_summary* summary);
  void* iteration_callback_user;
} ba_options;
/* what a callback sees (the fields of ceres::IterationSummary COLMAP's callers read) */
typedef struct ba_iteration_summary {
  int32_t iteration;            /* 0 = initial evaluation */
  int32_t step_is_successful;   /* the step of this iteration was accepted */
  int32_t linear_solver_iterations;
  double cost;                  /* of the current (last accepted) state */
  double cost_change;           /* old - new for an accepted step, else 0 */
  double trust_region_radius;
  double cumulative_time_in_seconds; /* since the LM loop started */
} ba_iteration_summary;
enum { BA_CALLBACK_CONTINUE = 0, BA_CALLBACK_TERMINATE = 1, BA_CALLBACK_ABORT = 2 };
enum { BA_SOLVER_ITERATIVE_SCHUR = 0, BA_SOLVER_DENSE_SCHUR = 1, BA_SOLVER_AUTO = 2, BA_SOLVER_SPARSE_SCHUR = 3 };
enum { BA_OPERATOR_F64 = 0, BA_OPERATOR_F32 = 1 };

/* CeresBundleAdjustmentOptions::LossFunctionType */
enum { BA_LOSS_TRIVIAL = 0, BA_LOSS_SOFT_L1 = 1, BA_LOSS_CAUCHY = 2, BA_LOSS_HUBER = 3 };

/* colmap::BundleAdjustmentTerminationType (bundle_adjustment.h:50-57) */
enum { BA_CONVERGENCE = 0, BA_NO_CONVERGENCE = 1, BA_FAILURE = 2, BA_USER_SUCCESS = 3, BA_USER_FAILURE = 4 };

/* colmap::BundleAdjustmentSummary (bundle_adjustment.h:63-74) + solver statistics */
typedef struct ba_result {
  int32_t termination_type;
  int32_t num_residuals;             /* residuals touching >= 1 variable block */
  int32_t num_iterations, num_successful_steps;
  int32_t num_effective_parameters;
  int64_t total_linear_iterations;
  double initial_cost, final_cost;
  double lm_seconds;                 /* wall time of the LM loop, inputs resident in HBM */
  int32_t num_logged;
  double* log_cost;                  /* [max_log] or NULL */
  double* log_radius;
  int32_t* log_linear_iters;
  int32_t linear_solver_used;        /* BA_SOLVER_* tier that ran (never BA_SOLVER_AUTO) */
  double factor_seconds;             /* exact tiers: time inside the blocked Cholesky (matrix-core kernels), summed */
  double setup_seconds;              /* everything before the LM loop: flattening into the device layout, sorting, index
                                        lists, upload (the mapper calls BA hundreds of times: sfm/incremental_mapper.cc:
                                        939,1086,1201 -- set-up is product time; lm_seconds + setup_seconds + the final
                                        write-back is the whole-solve time the reference's harness reports,
                                        benchmark/runtime/bundle_adjustment.cc:146-157) */
} ba_result;

/* Multi-GPU: every rank holds the full parameter set and calls ba_solve_sharded with the SAME
 * problem; the observations are sharded and partial sums are combined by an in-place sum over ranks.
 *  BA_SHARD_BY_IMAGE (obs_pose % world_size == rank; what BASELINE.json names):
 *   per LM iteration: cost scalars, J^T r + column norms, E^T E (6 / point), Schur-Jacobi blocks;
 *   per PCG iteration: E^T x (3 doubles / point) and the camera-space vector J_c^T v.
 *  BA_SHARD_BY_POINT (obs_point % world_size == rank; SURVEY.md section 8e's alternative): all
 *   observations of a point are on one rank, so E^T E, C^-1, E^T x stay local and only camera-space
 *   vectors travel: per LM iteration cost scalars, J_c^T r + column norms, Schur-Jacobi blocks, per
 *   PCG iteration J_c^T v (6 N_c + sum P_t doubles); the points are gathered once at the end.
 * Transport: either a host callback (any communicator: gloo, MPI, ...) or an RCCL communicator
 * created with ba_rccl_comm_create (all-reduce on the solver's stream, xGMI). No counterpart in the
 * reference: Ceres and Caspar are single-device (bundle_adjustment_ceres.cc:189-191). */
typedef int (*ba_allreduce_fn)(void* user, double* buffer, int64_t count); /* in-place sum, host memory; 0 = ok */
typedef struct ba_comm {
  int32_t rank, world_size;
  ba_allreduce_fn allreduce; /* used when rccl_comm is NULL */
  void* user;
  void* rccl_comm;           /* from ba_rccl_comm_create, or NULL */
  int32_t sharding;          /* BA_SHARD_BY_IMAGE (0) or BA_SHARD_BY_POINT (1) */
} ba_comm;
enum { BA_SHARD_BY_IMAGE = 0, BA_SHARD_BY_POINT = 1 };

void ba_options_init(ba_options* options);

int ba_solve_sharded(ba_problem* problem, const ba_options* options, int32_t gpu_index, const ba_comm* comm,
                     ba_result* result);
/* Number of observations rank `rank` of `world_size` works on (host-only helper, no GPU needed):
 * active observations (>= 1 variable block) whose pose index satisfies pose % world_size == rank. */
int64_t ba_shard_num_observations(const ba_problem* problem, int32_t rank, int32_t world_size);
/* The same for BA_SHARD_BY_POINT (point % world_size == rank). */
int64_t ba_shard_num_observations_by_point(const ba_problem* problem, int32_t rank, int32_t world_size);
/* RCCL transport: rank 0 obtains a 128-byte id, every rank creates its communicator from it. */
int ba_rccl_unique_id(char id[128]);
int ba_rccl_comm_create(const char id[128], int32_t rank, int32_t world_size, int32_t gpu_index, void** comm);
void ba_rccl_comm_destroy(void* comm);

/* BundleAdjuster::Solve for a flattened problem. gpu_index: device ordinal, -1 = current. */
int ba_solve(ba_problem* problem, const ba_options* options, int32_t gpu_index, ba_result* result);

/* Per-kernel HIP-event timing of the last ba_solve on this thread (roofline accounting):
 * milliseconds and launch count of the dominant kernel family (implicit Schur product). */
int ba_last_spmv_timing(double* total_ms, int64_t* launches, int64_t* bytes_per_launch);
/* The same for the f64 MFMA kernel (Schur-Jacobi blocks, one launch per LM iteration): the fraction of
 * the LM time spent inside the dense contraction is what the bench line reports as mfma_time_frac. */
int ba_last_mfma_timing(double* total_ms, int64_t* launches);
/* Which PCG loop the linear solves of the last ba_solve / ba_solve_sharded on this thread took: the pipelined one
 * (iteration k + 1 enqueued before iteration k is looked at, stopping test on the device, no blocking host read per
 * iteration: single-GPU solves and point-sharded ones -- there the one collective per iteration is an all-reduce of the
 * camera-space vector on the solver's stream) or the step-by-step one (image-sharded solves, solves with priors). */
int ba_last_pcg_loops(int64_t* pipelined_solves, int64_t* stepwise_solves);

const char* ba_last_error(void);
/* Layout version of ba_options / ba_result / ba_problem as this header declares them. The structs have grown by
 * appended fields (operator_precision, linear_solver_used, factor_seconds, iteration_callback, setup_seconds); a caller built against another header
 * would pass shorter structs. Callers compare ba_abi_version() with COLMAP_AMD_BA_ABI_VERSION once, before the first
 * solve (the C++ and Python adapters of this repository do). */
#define COLMAP_AMD_BA_ABI_VERSION 4
int32_t ba_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* COLMAP_AMD_BA_H_ */
