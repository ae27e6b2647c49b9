// colmap_amd/bundle_adjustment.hpp -- C++ host side of the MI355X bundle-adjustment path.
//
// Restates, on the standard library only (Eigen / Ceres / glog are not available in this build
// environment), the reference interface of this path:
//   estimators/bundle_adjustment.h:50-234        BundleAdjustmentConfig / Options / Summary /
//                                                BundleAdjuster / CreateDefaultBundleAdjuster
//   estimators/bundle_adjustment_ceres.cc:270-889 problem construction rules of DefaultBundleAdjuster
//   scene/{camera,image,point2d,point3d,rig,frame,reconstruction}.h   the slice BA touches
// and implements the third backend (MI355X) the way CasparBundleAdjuster implements the second
// (estimators/bundle_adjustment_caspar.cc:61-377 flatten, :767-801 write back): flatten the
// Reconstruction into the SoA ba_problem of the C ABI (colmap_amd_ba.h), ba_solve(), write the
// variable blocks back in place.
//
// Header-only; link with libcolmap_amd.so. Namespace colmap_amd instead of colmap.
#ifndef COLMAP_AMD_BUNDLE_ADJUSTMENT_HPP_
#define COLMAP_AMD_BUNDLE_ADJUSTMENT_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../colmap_amd_ba.h"

namespace colmap_amd {

using camera_t = uint32_t;
using image_t = uint32_t;
using frame_t = uint32_t;
using rig_t = uint32_t;
using point2D_t = uint32_t;
using point3D_t = uint64_t;
constexpr point3D_t kInvalidPoint3DId = static_cast<point3D_t>(-1);

#define COLMAP_AMD_BA_CHECK(cond)                                                        \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      std::ostringstream os_;                                                            \
      os_ << "[" << __FILE__ << ":" << __LINE__ << "] Check failed: " #cond;              \
      throw std::invalid_argument(os_.str());                                            \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Scene containers: the members of the reference classes that bundle adjustment reads or writes.
// ---------------------------------------------------------------------------------------------

// Rigid3d (geometry/rigid3.h:46-70): params = quaternion xyzw + translation.
struct Rigid3d {
  std::array<double, 7> params{0, 0, 0, 1, 0, 0, 0};
};

inline void QuatToRot(const double* q, double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

// a * b: apply b, then a (Rigid3d operator*, geometry/rigid3.h)
// rotation matrix -> unit quaternion (xyzw)
inline void RotToQuat(const double R[9], double* q) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; q[3] = 0.25 * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(1.0 + R[4 * i] - R[4 * j] - R[4 * k]) * 2;
    q[i] = 0.25 * s;
    q[j] = (R[3 * j + i] + R[3 * i + j]) / s;
    q[k] = (R[3 * k + i] + R[3 * i + k]) / s;
    q[3] = (R[3 * k + j] - R[3 * j + k]) / s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int c = 0; c < 4; ++c) q[c] /= n;
}

inline Rigid3d Compose(const Rigid3d& a, const Rigid3d& b) {
  const double *qa = a.params.data(), *qb = b.params.data();
  Rigid3d out;
  out.params[0] = qa[3] * qb[0] + qa[0] * qb[3] + qa[1] * qb[2] - qa[2] * qb[1];
  out.params[1] = qa[3] * qb[1] - qa[0] * qb[2] + qa[1] * qb[3] + qa[2] * qb[0];
  out.params[2] = qa[3] * qb[2] + qa[0] * qb[1] - qa[1] * qb[0] + qa[2] * qb[3];
  out.params[3] = qa[3] * qb[3] - qa[0] * qb[0] - qa[1] * qb[1] - qa[2] * qb[2];
  double R[9];
  QuatToRot(qa, R);
  for (int r = 0; r < 3; ++r)
    out.params[4 + r] = R[3 * r] * qb[4] + R[3 * r + 1] * qb[5] + R[3 * r + 2] * qb[6] + qa[4 + r];
  return out;
}

// CameraModelId (sensor/models.h:90-111): the models the MI355X backend supports.
enum class CameraModelId : int {
  SIMPLE_PINHOLE = 0, PINHOLE = 1, SIMPLE_RADIAL = 2, RADIAL = 3, OPENCV = 4,
  OPENCV_FISHEYE = 5, FULL_OPENCV = 6, FOV = 7, SIMPLE_RADIAL_FISHEYE = 8, RADIAL_FISHEYE = 9,
  THIN_PRISM_FISHEYE = 10, RAD_TAN_THIN_PRISM_FISHEYE = 11, SIMPLE_DIVISION = 12, DIVISION = 13, SIMPLE_FISHEYE = 14, FISHEYE = 15, EUCM = 16,
  EQUIRECTANGULAR = 17
};

struct CameraModelInfo {
  int num_params;
  std::vector<size_t> focal_length_idxs, principal_point_idxs, extra_params_idxs;
};

inline const CameraModelInfo* GetCameraModelInfo(int model_id) {
  static const CameraModelInfo kSimplePinhole{3, {0}, {1, 2}, {}}, kPinhole{4, {0, 1}, {2, 3}, {}},
      kSimpleRadial{4, {0}, {1, 2}, {3}}, kRadial{5, {0}, {1, 2}, {3, 4}}, kOpenCV{8, {0, 1}, {2, 3}, {4, 5, 6, 7}},
      kTwoFocalOneExtra{5, {0, 1}, {2, 3}, {4}}, kEucm{6, {0, 1}, {2, 3}, {4, 5}},
      kTwelve{12, {0, 1}, {2, 3}, {4, 5, 6, 7, 8, 9, 10, 11}},
      kSixteen{16, {0, 1}, {2, 3}, {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}}, kEquirectangular{2, {}, {}, {}};
  switch (model_id) {
    case 0: case 14: return &kSimplePinhole;  // SIMPLE_PINHOLE, SIMPLE_FISHEYE: f cx cy
    case 1: case 15: return &kPinhole;        // PINHOLE, FISHEYE: fx fy cx cy
    case 2: case 8: case 12: return &kSimpleRadial;  // SIMPLE_RADIAL, SIMPLE_RADIAL_FISHEYE, SIMPLE_DIVISION: f cx cy k
    case 3: case 9: return &kRadial;         // RADIAL, RADIAL_FISHEYE: f cx cy k1 k2
    case 4: case 5: return &kOpenCV;         // OPENCV, OPENCV_FISHEYE: fx fy cx cy + four extra
    case 7: case 13: return &kTwoFocalOneExtra;  // FOV (omega), DIVISION (k): fx fy cx cy + one extra
    case 16: return &kEucm;                  // EUCM: fx fy cx cy alpha beta
    case 11: return &kSixteen;               // RAD_TAN_THIN_PRISM_FISHEYE: k0..k5 p0 p1 s0..s3
    case 17: return &kEquirectangular;       // EQUIRECTANGULAR: width height, metadata only (never refined)
    case 6: case 10: return &kTwelve;        // FULL_OPENCV (k1 k2 p1 p2 k3 k4 k5 k6), THIN_PRISM_FISHEYE (k1 k2 p1 p2 k3 k4 sx1 sy1)
    default: return nullptr;
  }
}

struct Camera {  // scene/camera.h
  camera_t camera_id = 0;
  int model_id = static_cast<int>(CameraModelId::SIMPLE_RADIAL);
  size_t width = 0, height = 0;
  std::vector<double> params;
};

struct Point2D {  // scene/point2d.h
  std::array<double, 2> xy{0, 0};
  point3D_t point3D_id = kInvalidPoint3DId;
  bool HasPoint3D() const { return point3D_id != kInvalidPoint3DId; }
};

struct TrackElement {
  image_t image_id;
  point2D_t point2D_idx;
};

struct Point3D {  // scene/point3d.h
  std::array<double, 3> xyz{0, 0, 0};
  std::vector<TrackElement> track;
};

struct Rig {  // scene/rig.h: one reference camera, sensor_from_rig for every other camera
  rig_t rig_id = 0;
  camera_t ref_camera_id = 0;
  std::map<camera_t, Rigid3d> sensors_from_rig;
  bool IsRefSensor(camera_t camera_id) const { return camera_id == ref_camera_id; }
};

struct Frame {  // scene/frame.h
  frame_t frame_id = 0;
  rig_t rig_id = 0;
  Rigid3d rig_from_world;
  std::vector<image_t> image_ids;
};

struct Image {  // scene/image.h
  image_t image_id = 0;
  camera_t camera_id = 0;
  // Pose block of a trivial frame (the image is its own frame); derived composition otherwise.
  Rigid3d cam_from_world;
  std::optional<frame_t> frame_id;  // set: member of a non-trivial rig's frame
  std::vector<Point2D> points2D;
  frame_t FrameId() const { return frame_id ? *frame_id : image_id; }
};

class Reconstruction {  // scene/reconstruction.h
 public:
  std::map<camera_t, ::colmap_amd::Camera> cameras;
  std::map<image_t, ::colmap_amd::Image> images;
  std::map<point3D_t, ::colmap_amd::Point3D> points3D;
  std::map<rig_t, Rig> rigs;        // non-trivial rigs only
  std::map<frame_t, Frame> frames;  // their frames

  ::colmap_amd::Camera& Camera(camera_t id) { return cameras.at(id); }
  ::colmap_amd::Image& Image(image_t id) { return images.at(id); }
  ::colmap_amd::Point3D& Point3D(point3D_t id) { return points3D.at(id); }
  const ::colmap_amd::Camera& Camera(camera_t id) const { return cameras.at(id); }
  const ::colmap_amd::Image& Image(image_t id) const { return images.at(id); }
  const ::colmap_amd::Point3D& Point3D(point3D_t id) const { return points3D.at(id); }
  std::vector<image_t> RegImageIds() const {
    std::vector<image_t> ids;
    for (const auto& kv : images) ids.push_back(kv.first);
    return ids;
  }
  size_t NumPoints3D() const { return points3D.size(); }

  bool IsRefInFrame(const ::colmap_amd::Image& image) const {
    if (!image.frame_id) return true;
    return rigs.at(frames.at(*image.frame_id).rig_id).IsRefSensor(image.camera_id);
  }
  const Rigid3d& SensorFromRig(const ::colmap_amd::Image& image) const {
    return rigs.at(frames.at(*image.frame_id).rig_id).sensors_from_rig.at(image.camera_id);
  }
  // Image::ProjectionCenter: -R^T t
  std::array<double, 3> ProjectionCenter(image_t id) const {
    const Rigid3d& p = images.at(id).cam_from_world;
    double R[9];
    QuatToRot(p.params.data(), R);
    std::array<double, 3> c{};
    for (int i = 0; i < 3; ++i) c[i] = -(R[i] * p.params[4] + R[3 + i] * p.params[5] + R[6 + i] * p.params[6]);
    return c;
  }
  // Reconstruction::Transform(new_from_old_world = Sim3d(scale, R, t)) (scene/reconstruction.cc:788-805)
  void Transform(double scale, const double R[9], const double t[3]) {
    auto cam = [&](Rigid3d& pose) {  // TransformCameraWorld: R' = R_c R^T, t' = s t_c - R' t
      double Rc[9], Rn[9];
      QuatToRot(pose.params.data(), Rc);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[3 * i + j] = Rc[3 * i] * R[3 * j] + Rc[3 * i + 1] * R[3 * j + 1] + Rc[3 * i + 2] * R[3 * j + 2];
      RotToQuat(Rn, pose.params.data());
      for (int i = 0; i < 3; ++i)
        pose.params[4 + i] = scale * pose.params[4 + i] - (Rn[3 * i] * t[0] + Rn[3 * i + 1] * t[1] + Rn[3 * i + 2] * t[2]);
    };
    for (auto& [rid, rig] : rigs)
      for (auto& [cid, sfr] : rig.sensors_from_rig)
        if (!rig.IsRefSensor(cid))
          for (int i = 0; i < 3; ++i) sfr.params[4 + i] *= scale;
    for (auto& [fid, frame] : frames) cam(frame.rig_from_world);
    for (auto& [iid, image] : images)
      if (!image.frame_id) cam(image.cam_from_world);
    UpdateCamFromWorld();
    for (auto& [pid, pt] : points3D) {
      const std::array<double, 3> x = pt.xyz;
      for (int i = 0; i < 3; ++i) pt.xyz[i] = scale * (R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2]) + t[i];
    }
  }
  // Reconstruction::Normalize(fixed_scale = true) (:698-727): translation by minus the centroid of the
  // projection centres inside the [0.1, 0.9] percentile range (geometry/normalization.cc:39-92).
  // Returns the translation of normalized_from_metric.
  std::array<double, 3> NormalizeFixedScale() {
    std::array<double, 3> t{0, 0, 0};
    if (images.size() < 2) return t;
    std::vector<double> coords[3];
    for (const auto& kv : images) {
      const auto c = ProjectionCenter(kv.first);
      for (int k = 0; k < 3; ++k) coords[k].push_back(c[k]);
    }
    const size_t end = coords[0].size() - 1;
    const size_t lo = std::min<size_t>(end, static_cast<size_t>(std::floor(0.1 * end)));
    const size_t hi = std::min<size_t>(end, static_cast<size_t>(std::ceil(0.9 * end)));
    for (int k = 0; k < 3; ++k) {
      std::sort(coords[k].begin(), coords[k].end());
      double sum = 0;
      for (size_t i = lo; i <= hi; ++i) sum += coords[k][i];
      t[k] = -sum / static_cast<double>(hi - lo + 1);
    }
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Transform(1.0, I, t.data());
    return t;
  }
  // Image::CamFromWorld of the images of non-trivial frames
  void UpdateCamFromWorld() {
    for (auto& [fid, frame] : frames) {
      const Rig& rig = rigs.at(frame.rig_id);
      for (const image_t id : frame.image_ids) {
        auto& image = images.at(id);
        image.cam_from_world = rig.IsRefSensor(image.camera_id)
                                   ? frame.rig_from_world
                                   : Compose(rig.sensors_from_rig.at(image.camera_id), frame.rig_from_world);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// estimators/bundle_adjustment.h
// ---------------------------------------------------------------------------------------------

enum class BundleAdjustmentGauge { UNSPECIFIED = -1, TWO_CAMS_FROM_WORLD = 0, THREE_POINTS = 1 };  // :44-48
enum class BundleAdjustmentTerminationType {  // :50-57
  CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2, USER_SUCCESS = 3, USER_FAILURE = 4
};
enum class BundleAdjustmentBackend { CERES = 0, CASPAR = 1, MI355X = 2 };  // :60 + the new value

struct BundleAdjustmentSummary {  // :63-74
  virtual ~BundleAdjustmentSummary() = default;
  BundleAdjustmentTerminationType termination_type = BundleAdjustmentTerminationType::FAILURE;
  int num_residuals = 0;
  // solver statistics of the MI355X backend (CeresBundleAdjustmentSummary carries ceres's)
  int num_iterations = 0, num_successful_steps = 0, num_effective_parameters = 0;
  int64_t total_linear_iterations = 0;
  double initial_cost = 0, final_cost = 0, lm_seconds = 0, setup_seconds = 0;

  bool IsSolutionUsable() const {
    return termination_type == BundleAdjustmentTerminationType::CONVERGENCE ||
           termination_type == BundleAdjustmentTerminationType::NO_CONVERGENCE ||
           termination_type == BundleAdjustmentTerminationType::USER_SUCCESS;
  }
  virtual std::string BriefReport() const {
    std::ostringstream os;
    os << "MI355X BA: " << num_residuals << " residuals, " << num_iterations << " iterations, cost " << initial_cost
       << " -> " << final_cost;
    return os.str();
  }
};

class BundleAdjustmentConfig {  // :77-150, .cc:55-256
 public:
  void FixGauge(BundleAdjustmentGauge gauge) { fixed_gauge_ = gauge; }
  BundleAdjustmentGauge FixedGauge() const { return fixed_gauge_; }

  size_t NumImages() const { return image_ids_.size(); }
  void AddImage(image_t image_id) { image_ids_.insert(image_id); }
  bool HasImage(image_t image_id) const { return image_ids_.count(image_id) > 0; }
  void RemoveImage(image_t image_id) { image_ids_.erase(image_id); }
  const std::set<image_t>& Images() const { return image_ids_; }

  void SetConstantCamIntrinsics(camera_t id) { constant_cam_intrinsics_.insert(id); }
  void SetVariableCamIntrinsics(camera_t id) { constant_cam_intrinsics_.erase(id); }
  bool HasConstantCamIntrinsics(camera_t id) const { return constant_cam_intrinsics_.count(id) > 0; }

  void SetConstantSensorFromRigPose(camera_t sensor_id) { constant_sensor_from_rig_poses_.insert(sensor_id); }
  void SetVariableSensorFromRigPose(camera_t sensor_id) { constant_sensor_from_rig_poses_.erase(sensor_id); }
  bool HasConstantSensorFromRigPose(camera_t sensor_id) const {
    return constant_sensor_from_rig_poses_.count(sensor_id) > 0;
  }

  void SetConstantRigFromWorldPose(frame_t id) { constant_rig_from_world_poses_.insert(id); }
  void SetVariableRigFromWorldPose(frame_t id) { constant_rig_from_world_poses_.erase(id); }
  bool HasConstantRigFromWorldPose(frame_t id) const { return constant_rig_from_world_poses_.count(id) > 0; }

  void AddVariablePoint(point3D_t id) {
    COLMAP_AMD_BA_CHECK(!HasConstantPoint(id));
    COLMAP_AMD_BA_CHECK(!IsIgnoredPoint(id));
    variable_point3D_ids_.insert(id);
  }
  void AddConstantPoint(point3D_t id) {
    COLMAP_AMD_BA_CHECK(!HasVariablePoint(id));
    COLMAP_AMD_BA_CHECK(!IsIgnoredPoint(id));
    constant_point3D_ids_.insert(id);
  }
  void IgnorePoint(point3D_t id) {
    COLMAP_AMD_BA_CHECK(!HasVariablePoint(id));
    COLMAP_AMD_BA_CHECK(!HasConstantPoint(id));
    ignored_point3D_ids_.insert(id);
  }
  bool HasPoint(point3D_t id) const { return HasVariablePoint(id) || HasConstantPoint(id); }
  bool HasVariablePoint(point3D_t id) const { return variable_point3D_ids_.count(id) > 0; }
  bool HasConstantPoint(point3D_t id) const { return constant_point3D_ids_.count(id) > 0; }
  bool IsIgnoredPoint(point3D_t id) const { return ignored_point3D_ids_.count(id) > 0; }
  const std::set<point3D_t>& VariablePoints() const { return variable_point3D_ids_; }
  const std::set<point3D_t>& ConstantPoints() const { return constant_point3D_ids_; }

  // NumResiduals (.cc:80-130): 2 x observations of the config's images (not ignored), plus the
  // observations of added points from images outside the config
  size_t NumResiduals(const Reconstruction& reconstruction) const {
    size_t num_observations = 0;
    for (const image_t image_id : image_ids_)
      for (const Point2D& p : reconstruction.Image(image_id).points2D)
        if (p.HasPoint3D() && !IsIgnoredPoint(p.point3D_id)) ++num_observations;
    auto outside = [&](const std::set<point3D_t>& ids) {
      for (const point3D_t id : ids)
        for (const TrackElement& el : reconstruction.Point3D(id).track)
          if (!HasImage(el.image_id)) ++num_observations;
    };
    outside(variable_point3D_ids_);
    outside(constant_point3D_ids_);
    return 2 * num_observations;
  }

 private:
  BundleAdjustmentGauge fixed_gauge_ = BundleAdjustmentGauge::UNSPECIFIED;
  std::set<image_t> image_ids_;
  std::set<point3D_t> variable_point3D_ids_, constant_point3D_ids_, ignored_point3D_ids_;
  std::set<camera_t> constant_cam_intrinsics_, constant_sensor_from_rig_poses_;
  std::set<frame_t> constant_rig_from_world_poses_;
};

// Options of the MI355X backend: what ceres::Solver::Options carries in CeresBundleAdjustmentOptions
// (bundle_adjustment_ceres.h:40-89, defaults .cc:102-115).
struct Mi355xBundleAdjustmentOptions {
  enum class LossFunctionType { TRIVIAL = 0, SOFT_L1 = 1, CAUCHY = 2, HUBER = 3 };
  LossFunctionType loss_function_type = LossFunctionType::TRIVIAL;
  double loss_function_scale = 1.0;
  // Passed to ba_solve as it is. The reference's controller stops a running adjustment through
  // solver_options.callbacks (controllers/bundle_adjustment.cc:40-57,84-86); here that is
  // solver_options.iteration_callback / iteration_callback_user (colmap_amd_ba.h): return BA_CALLBACK_TERMINATE from
  // it when BaseController::CheckIfStopped() says so and Solve() returns USER_SUCCESS with the last accepted step.
  ba_options solver_options;
  // The AUTO rule's thresholds. The reference keeps one pair per device class (bundle_adjustment_ceres.h:68-71:
  // 50 / 1000 images for its CPU solvers, 200 / 4000 for Ceres-CUDA). Measured on the MI355X over three seeds per size
  // (scripts/ba_tier_crossover.py, profiles/r06_ba_tier_crossover.json; criterion: time to the cost the
  // exact tier has after three LM steps): the exact tiers get there first from 50 to 4000 images (Schur-PCG is level with
  // them at 350 and 1000 and does not reach that cost within 30 LM iterations from 1500 on), so the rule is the
  // reference's own GPU pair -- monotone, exact wherever the reduced camera system fits the dense formation.
  int max_num_images_direct_dense_gpu_solver = 200;
  int max_num_images_direct_sparse_gpu_solver = 4000;
  Mi355xBundleAdjustmentOptions() {
    ba_options_init(&solver_options);
    // the reference's solver choice by problem size (CreateSolverOptions, bundle_adjustment_ceres.cc:203-213)
    solver_options.linear_solver_type = BA_SOLVER_AUTO;
  }
};

struct BundleAdjustmentOptions {  // :173-209
  bool refine_focal_length = true;
  bool refine_principal_point = false;
  bool refine_extra_params = true;
  bool refine_sensor_from_rig = true;
  bool refine_rig_from_world = true;
  bool refine_points3D = true;
  bool constant_rig_from_world_rotation = false;
  int min_track_length = 0;
  bool print_summary = true;
  std::string gpu_index = "-1";
  BundleAdjustmentBackend backend = BundleAdjustmentBackend::MI355X;
  std::shared_ptr<Mi355xBundleAdjustmentOptions> mi355x = std::make_shared<Mi355xBundleAdjustmentOptions>();

  bool Check() const { return min_track_length >= 0; }
};

class BundleAdjuster {  // :212-228
 public:
  BundleAdjuster(BundleAdjustmentOptions options, BundleAdjustmentConfig config)
      : options_(std::move(options)), config_(std::move(config)) {
    COLMAP_AMD_BA_CHECK(options_.Check());
  }
  virtual ~BundleAdjuster() = default;
  virtual std::shared_ptr<BundleAdjustmentSummary> Solve() = 0;
  const BundleAdjustmentOptions& Options() const { return options_; }
  const BundleAdjustmentConfig& Config() const { return config_; }
  // BA_SOLVER_* tier the last Solve() asked for (AUTO resolved on the image count) and the one that ran
  int LinearSolverRequested() const { return linear_solver_requested_; }
  int LinearSolverUsed() const { return linear_solver_used_; }

 protected:
  BundleAdjustmentOptions options_;
  BundleAdjustmentConfig config_;
  int linear_solver_requested_ = BA_SOLVER_ITERATIVE_SCHUR, linear_solver_used_ = BA_SOLVER_ITERATIVE_SCHUR;
};

// ---------------------------------------------------------------------------------------------
// The MI355X backend
// ---------------------------------------------------------------------------------------------
class Mi355xBundleAdjuster : public BundleAdjuster {
 public:
  Mi355xBundleAdjuster(BundleAdjustmentOptions options, BundleAdjustmentConfig config, Reconstruction& reconstruction)
      : BundleAdjuster(std::move(options), std::move(config)), reconstruction_(reconstruction) {
    Flatten();
  }

  std::shared_ptr<BundleAdjustmentSummary> Solve() override {
    auto summary = std::make_shared<BundleAdjustmentSummary>();
    if (obs_pose_.empty()) return summary;  // bundle_adjustment_ceres.cc:667-669
    ba_problem p = Problem();
    ba_options so = options_.mi355x->solver_options;
    so.loss_type = static_cast<int32_t>(options_.mi355x->loss_function_type);
    so.loss_scale = options_.mi355x->loss_function_scale;
    if (so.linear_solver_type == BA_SOLVER_AUTO) {
      // CreateSolverOptions' rule on config.NumImages() (bundle_adjustment_ceres.cc:131,203-213) with this backend's
      // measured GPU thresholds (the reference's own GPU pair: bundle_adjustment_ceres.h:70-71), resolved here where
      // the image count is known: the flat C interface only sees pose blocks (a rig frame with several sensors is
      // one block)
      const size_t n_img = config_.NumImages();
      const size_t nd = (size_t)std::max(options_.mi355x->max_num_images_direct_dense_gpu_solver, 0);
      const size_t ns = (size_t)std::max(options_.mi355x->max_num_images_direct_sparse_gpu_solver, 0);
      so.linear_solver_type = n_img <= nd ? BA_SOLVER_DENSE_SCHUR : (n_img <= ns ? BA_SOLVER_SPARSE_SCHUR : BA_SOLVER_ITERATIVE_SCHUR);
    }
    linear_solver_requested_ = so.linear_solver_type;
    ba_result res{};
    int gpu = -1;
    if (!options_.gpu_index.empty()) gpu = std::stoi(options_.gpu_index);  // single GPU (:189-191)
    if (ba_abi_version() != COLMAP_AMD_BA_ABI_VERSION)
      throw std::runtime_error("libcolmap_amd.so and colmap_amd_ba.h disagree on the layout of ba_options / ba_result");
    if (ba_solve(&p, &so, gpu, &res) != 0) throw std::runtime_error(ba_last_error());
    if (res.num_residuals == 0) return summary;
    WriteBack();
    summary->termination_type = static_cast<BundleAdjustmentTerminationType>(res.termination_type);
    summary->num_residuals = res.num_residuals;
    summary->num_iterations = res.num_iterations;
    summary->num_successful_steps = res.num_successful_steps;
    summary->num_effective_parameters = res.num_effective_parameters;
    summary->total_linear_iterations = res.total_linear_iterations;
    summary->initial_cost = res.initial_cost;
    summary->final_cost = res.final_cost;
    summary->lm_seconds = res.lm_seconds;
    summary->setup_seconds = res.setup_seconds;
    linear_solver_used_ = res.linear_solver_used;  // differs from the requested tier when that one did not apply
    return summary;
  }

  // The flattened problem (what the C ABI sees) -- exposed for tests and for sharded solves.
  ba_problem Problem() {
    ba_problem p{};
    p.num_poses = static_cast<int32_t>(pose_const_.size());
    p.num_cams = static_cast<int32_t>(cam_model_.size());
    p.num_points = static_cast<int32_t>(point_const_.size());
    p.num_obs = static_cast<int64_t>(obs_pose_.size());
    p.poses = poses_.data();
    p.cams = cams_.data();
    p.cam_model = cam_model_.data();
    p.points = points_.data();
    p.obs_pose = obs_pose_.data();
    p.obs_cam = obs_cam_.data();
    p.obs_point = obs_point_.data();
    p.obs_xy = obs_xy_.data();
    p.pose_const = pose_const_.data();
    p.pose_fixed_t = pose_fixed_t_.data();
    p.cam_const = cam_const_.data();
    p.point_const = point_const_.data();
    p.num_sensors = static_cast<int32_t>(sensors_.size() / 7);
    p.sensors = sensors_.empty() ? nullptr : sensors_.data();
    p.obs_sensor = sensors_.empty() ? nullptr : obs_sensor_.data();
    const bool any_variable_sensor = std::count(sensor_const_.begin(), sensor_const_.end(), 0) > 0;
    p.sensor_const = any_variable_sensor ? sensor_const_.data() : nullptr;
    p.num_priors = static_cast<int32_t>(prior_pose_.size());
    if (p.num_priors > 0) {
      p.prior_pose = prior_pose_.data();
      p.prior_sensor = prior_sensor_.data();
      p.prior_position = prior_position_.data();
      p.prior_sqrt_info = prior_sqrt_info_.data();
      p.prior_loss_type = prior_loss_type_;
      p.prior_loss_scale = prior_loss_scale_;
    }
    return p;
  }
  // Residuals touching >= 1 variable block / variable tangent dimensions, computed on the host
  // (what ceres reports as num_residuals_reduced / num_effective_parameters_reduced).
  size_t NumResidualsReduced() {
    ba_problem p = Problem();
    return 2 * static_cast<size_t>(ba_shard_num_observations(&p, 0, 1));
  }
  size_t NumEffectiveParametersReduced() const {
    std::vector<char> pose_used(pose_const_.size(), 0), cam_used(cam_model_.size(), 0), pt_used(point_const_.size(), 0);
    std::vector<int> cam_nvar(cam_model_.size(), 0);
    for (size_t k = 0; k < cam_model_.size(); ++k)
      for (int j = 0; j < GetCameraModelInfo(cam_model_[k])->num_params; ++j)
        cam_nvar[k] += cam_const_[k * BA_CAM_STRIDE + j] ? 0 : 1;
    std::vector<char> sens_used(sensor_const_.size(), 0);
    for (size_t o = 0; o < obs_pose_.size(); ++o) {
      const int si = obs_sensor_[o];
      const bool sens_var = si >= 0 && !sensor_const_[si];
      if (pose_const_[obs_pose_[o]] && cam_nvar[obs_cam_[o]] == 0 && point_const_[obs_point_[o]] && !sens_var) continue;
      pose_used[obs_pose_[o]] = cam_used[obs_cam_[o]] = pt_used[obs_point_[o]] = 1;
      if (sens_var) sens_used[si] = 1;
    }
    size_t n = 6 * static_cast<size_t>(std::count(sens_used.begin(), sens_used.end(), 1));
    for (size_t i = 0; i < pose_const_.size(); ++i)
      if (pose_used[i] && !pose_const_[i]) {
        const int pf = pose_fixed_t_[i];
        n += (pf >= BA_POSE_ROT_CONST ? 0 : 3) + ((pf >= 0 && (pf & 3) != 3) ? 2 : 3);
      }
    for (size_t k = 0; k < cam_model_.size(); ++k)
      if (cam_used[k]) n += cam_nvar[k];
    for (size_t j = 0; j < point_const_.size(); ++j)
      if (pt_used[j] && !point_const_[j]) n += 3;
    return n;
  }
  size_t NumPoseBlocks() const { return pose_const_.size(); }
  size_t NumConstantPoseBlocks() const { return std::count(pose_const_.begin(), pose_const_.end(), 1); }

 private:
  struct PoseRef {
    bool is_frame;  // the block lives in a Frame (rig_from_world) or in an Image (cam_from_world)
    uint32_t id;
  };

  int PoseSlot(bool is_frame, uint32_t id, bool constant, const Rigid3d& pose) {
    const auto key = std::make_tuple(is_frame, id, constant);
    auto it = pose_index_.find(key);
    if (it != pose_index_.end()) return it->second;
    const int slot = static_cast<int>(pose_refs_.size());
    pose_index_.emplace(key, slot);
    pose_refs_.push_back({is_frame, id});
    poses_.insert(poses_.end(), pose.params.begin(), pose.params.end());
    pose_const_.push_back(constant ? 1 : 0);
    pose_fixed_t_.push_back(-1);
    return slot;
  }
  int CamSlot(camera_t id) {
    auto it = cam_index_.find(id);
    if (it != cam_index_.end()) return it->second;
    const int slot = static_cast<int>(cam_ids_.size());
    cam_index_.emplace(id, slot);
    cam_ids_.push_back(id);
    return slot;
  }
  int PointSlot(point3D_t id) {
    auto it = point_index_.find(id);
    if (it != point_index_.end()) return it->second;
    const int slot = static_cast<int>(point_ids_.size());
    point_index_.emplace(id, slot);
    point_ids_.push_back(id);
    return slot;
  }
  int SensorSlot(camera_t id, const Rigid3d& sensor_from_rig, bool constant, rig_t rig_id) {
    auto it = sensor_index_.find(id);
    if (it != sensor_index_.end()) return it->second;
    const int slot = static_cast<int>(sensors_.size() / 7);
    sensor_index_.emplace(id, slot);
    sensors_.insert(sensors_.end(), sensor_from_rig.params.begin(), sensor_from_rig.params.end());
    sensor_const_.push_back(constant ? 1 : 0);
    sensor_ids_.push_back(id);
    sensor_rig_.push_back(rig_id);
    return slot;
  }
  void AddObservation(int pose, int cam, int point, const Point2D& p2, int sensor) {
    obs_pose_.push_back(pose);
    obs_cam_.push_back(cam);
    obs_point_.push_back(point);
    obs_xy_.push_back(p2.xy[0]);
    obs_xy_.push_back(p2.xy[1]);
    obs_sensor_.push_back(sensor);
  }

  // Pose slot and sensor slot of an image's residuals: AddImageWithTrivialFrame (:699-750) /
  // AddImageWithNonTrivialFrame (:752-822).
  std::pair<int, int> ImageBlocks(const Image& image, bool constant_frame) {
    const Reconstruction& rec = reconstruction_;
    const bool in_frame = image.frame_id.has_value();
    const Rigid3d& frame_pose = in_frame ? rec.frames.at(*image.frame_id).rig_from_world : image.cam_from_world;
    if (rec.IsRefInFrame(image))
      return {PoseSlot(in_frame, in_frame ? *image.frame_id : image.image_id, constant_frame, frame_pose), -1};
    const bool constant_sensor =
        !options_.refine_sensor_from_rig || config_.HasConstantSensorFromRigPose(image.camera_id);
    const Rigid3d& sensor_from_rig = rec.SensorFromRig(image);
    if (constant_frame && constant_sensor)  // ReprojErrorConstantPoseCostFunctor on the composition (:769-772,797-803)
      return {PoseSlot(false, image.image_id, true, Compose(sensor_from_rig, frame_pose)), -1};
    // constant sensor: RigReprojErrorConstantRigCostFunctor (:804-810); variable sensor: the general
    // RigReprojErrorCostFunctor with sensor_from_rig as a parameter block of its own (:811-820), also
    // under a constant frame
    return {PoseSlot(true, *image.frame_id, constant_frame, frame_pose),
            SensorSlot(image.camera_id, sensor_from_rig, constant_sensor, rec.frames.at(*image.frame_id).rig_id)};
  }

  void Flatten() {  // DefaultBundleAdjuster ctor (bundle_adjustment_ceres.cc:606-664)
    Reconstruction& rec = reconstruction_;
    std::set<camera_t> config_const_cams;
    for (const auto& kv : rec.cameras)
      if (config_.HasConstantCamIntrinsics(kv.first)) config_const_cams.insert(kv.first);
    std::unordered_map<point3D_t, size_t> num_obs_of_point;
    std::set<camera_t> parameterized_cams;
    std::vector<int> gauge_slots;
    std::vector<int64_t> gauge_frames;

    for (const image_t image_id : config_.Images()) {  // AddImageToProblem (:688-697)
      const Image& image = rec.Image(image_id);
      const bool constant_pose =
          !options_.refine_rig_from_world || config_.HasConstantRigFromWorldPose(image.FrameId());
      size_t num_observations = 0;
      std::pair<int, int> blocks{-1, -1};
      for (const Point2D& p2 : image.points2D) {
        if (!p2.HasPoint3D() || config_.IsIgnoredPoint(p2.point3D_id)) continue;
        const Point3D& point3D = rec.Point3D(p2.point3D_id);
        COLMAP_AMD_BA_CHECK(point3D.track.size() > 1);
        if (options_.min_track_length > 0 && static_cast<int>(point3D.track.size()) < options_.min_track_length)
          continue;
        if (num_observations == 0) blocks = ImageBlocks(image, constant_pose);
        ++num_observations;
        ++num_obs_of_point[p2.point3D_id];
        AddObservation(blocks.first, CamSlot(image.camera_id), PointSlot(p2.point3D_id), p2, blocks.second);
      }
      if (num_observations > 0) {
        image_slots_[image_id] = blocks;
        parameterized_cams.insert(image.camera_id);
        // gauge candidates: reference sensors and constant sensor_from_rig only (IsParameterizedConstSensor, :347-385)
        if (blocks.second < 0 || sensor_const_[blocks.second]) {
          gauge_slots.push_back(blocks.first);
          gauge_frames.push_back(image.frame_id ? static_cast<int64_t>(*image.frame_id)
                                                : -static_cast<int64_t>(image.image_id) - 1);
        }
      }
    }
    auto add_point = [&](point3D_t point3D_id) {  // AddPointToProblem (:826-887)
      const Point3D& point3D = rec.Point3D(point3D_id);
      if (options_.min_track_length > 0 && static_cast<int>(point3D.track.size()) < options_.min_track_length) return;
      size_t& n = num_obs_of_point[point3D_id];
      if (n == point3D.track.size()) return;
      for (const TrackElement& el : point3D.track) {
        if (config_.HasImage(el.image_id)) continue;
        ++n;
        const Image& image = rec.Image(el.image_id);
        AddObservation(PoseSlot(false, image.image_id, true, image.cam_from_world), CamSlot(image.camera_id),
                       PointSlot(point3D_id), image.points2D.at(el.point2D_idx), -1);
        if (parameterized_cams.insert(image.camera_id).second) config_const_cams.insert(image.camera_id);  // :883-886
      }
    };
    for (const point3D_t id : config_.VariablePoints()) add_point(id);  // (:633-638)
    for (const point3D_t id : config_.ConstantPoints()) add_point(id);

    // a rig whose reference sensor is not part of the problem keeps its sensor_from_rig constant (:526-543)
    for (size_t k = 0; k < sensor_const_.size(); ++k)
      if (!parameterized_cams.count(rec.rigs.at(sensor_rig_[k]).ref_camera_id)) sensor_const_[k] = 1;
    // ParameterizeCameras (:419-469)
    const bool constant_camera =
        !options_.refine_focal_length && !options_.refine_principal_point && !options_.refine_extra_params;
    cams_.assign(cam_ids_.size() * BA_CAM_STRIDE, 0.0);
    cam_model_.assign(cam_ids_.size(), 0);
    cam_const_.assign(cam_ids_.size() * BA_CAM_STRIDE, 1);
    for (size_t k = 0; k < cam_ids_.size(); ++k) {
      const Camera& camera = rec.Camera(cam_ids_[k]);
      const CameraModelInfo* info = GetCameraModelInfo(camera.model_id);
      if (!info || static_cast<int>(camera.params.size()) != info->num_params)
        throw std::invalid_argument("camera model " + std::to_string(camera.model_id) +
                                    " is not supported by the MI355X backend yet");
      std::copy(camera.params.begin(), camera.params.end(), cams_.begin() + k * BA_CAM_STRIDE);
      cam_model_[k] = camera.model_id;
      if (constant_camera || config_const_cams.count(camera.camera_id)) continue;
      auto free_idxs = [&](const std::vector<size_t>& idxs) {
        for (const size_t j : idxs) cam_const_[k * BA_CAM_STRIDE + j] = 0;
      };
      if (options_.refine_focal_length) free_idxs(info->focal_length_idxs);
      if (options_.refine_principal_point) free_idxs(info->principal_point_idxs);
      if (options_.refine_extra_params) free_idxs(info->extra_params_idxs);
    }
    // ParameterizePoints (:548-563)
    points_.resize(point_ids_.size() * 3);
    point_const_.assign(point_ids_.size(), 0);
    for (size_t j = 0; j < point_ids_.size(); ++j) {
      const Point3D& point3D = rec.Point3D(point_ids_[j]);
      std::copy(point3D.xyz.begin(), point3D.xyz.end(), points_.begin() + 3 * j);
      if (!options_.refine_points3D || point3D.track.size() > num_obs_of_point[point_ids_[j]]) point_const_[j] = 1;
    }
    for (const point3D_t id : config_.ConstantPoints()) {
      auto it = point_index_.find(id);
      if (it != point_index_.end()) point_const_[it->second] = 1;
    }
    // gauge (:646-663)
    if (config_.FixedGauge() == BundleAdjustmentGauge::TWO_CAMS_FROM_WORLD) {
      if (options_.refine_rig_from_world && !FixGaugeWithTwoCamsFromWorld(gauge_slots, gauge_frames))
        FixGaugeWithThreePoints();
    } else if (config_.FixedGauge() == BundleAdjustmentGauge::THREE_POINTS) {
      FixGaugeWithThreePoints();
    }
    if (options_.constant_rig_from_world_rotation) {
      // SubsetManifold(7, {0, 1, 2, 3[, 4 + fixed_dim]}) on every variable rig_from_world (:404-408, 513-516)
      for (size_t i = 0; i < pose_const_.size(); ++i)
        if (!pose_const_[i])
          pose_fixed_t_[i] = static_cast<int8_t>(BA_POSE_ROT_CONST + (pose_fixed_t_[i] >= 0 ? pose_fixed_t_[i] : 3));
    }
  }

  // FixGaugeWithTwoCamsFromWorld (:308-416) on the flattened blocks.
  bool FixGaugeWithTwoCamsFromWorld(const std::vector<int>& slots, const std::vector<int64_t>& frames) {
    std::vector<char> used(pose_const_.size(), 0);
    for (const int s : obs_pose_) used[s] = 1;
    int image1 = -1, image2 = -1, fixed_dim = 0;
    int64_t frame1 = 0;
    for (size_t i = 0; i < slots.size(); ++i) {
      if (!used[slots[i]] || !pose_const_[slots[i]]) continue;
      if (image1 < 0) {
        image1 = slots[i];
        frame1 = frames[i];
      } else if (frame1 != frames[i]) {
        return true;  // two frames already fixed
      }
    }
    for (size_t i = 0; i < slots.size(); ++i) {
      const int s = slots[i];
      if (!used[s]) continue;
      if (image1 < 0) {
        image1 = s;
        frame1 = frames[i];
        continue;
      }
      if (frames[i] == frame1 || pose_const_[s]) continue;
      // baseline = (frame1_from_world * Inverse(frame2_from_world)).translation (:374-377)
      double R1[9], R2[9];
      QuatToRot(&poses_[7 * image1], R1);
      QuatToRot(&poses_[7 * s], R2);
      const double* t1 = &poses_[7 * image1 + 4];
      const double* t2 = &poses_[7 * s + 4];
      double r2t[3], baseline[3];
      for (int r = 0; r < 3; ++r) r2t[r] = R2[r] * t2[0] + R2[3 + r] * t2[1] + R2[6 + r] * t2[2];  // R2^T t2
      for (int r = 0; r < 3; ++r)
        baseline[r] = t1[r] - (R1[3 * r] * r2t[0] + R1[3 * r + 1] * r2t[1] + R1[3 * r + 2] * r2t[2]);
      int k = 0;
      for (int r = 1; r < 3; ++r)
        if (std::abs(baseline[r]) > std::abs(baseline[k])) k = r;
      if (std::abs(baseline[k]) > 1e-9) {
        image2 = s;
        fixed_dim = k;
        break;
      }
    }
    if (image1 < 0 || image2 < 0) return false;
    pose_const_[image1] = 1;
    pose_fixed_t_[image2] = static_cast<int8_t>(fixed_dim);
    return true;
  }

  // FixGaugeWithThreePoints (:270-301): three points whose coordinates span rank 3.
  bool FixGaugeWithThreePoints() {
    std::vector<char> used(point_const_.size(), 0);
    for (const int s : obs_point_) used[s] = 1;
    std::vector<std::array<double, 3>> chosen;
    auto maybe = [&](size_t j) {
      std::array<double, 3> p{points_[3 * j], points_[3 * j + 1], points_[3 * j + 2]};
      // rank increase test by Gram-Schmidt against the chosen points
      std::array<double, 3> v = p;
      std::vector<std::array<double, 3>> basis;
      for (const auto& c : chosen) {
        std::array<double, 3> b = c;
        for (const auto& e : basis) {
          const double d = b[0] * e[0] + b[1] * e[1] + b[2] * e[2];
          for (int i = 0; i < 3; ++i) b[i] -= d * e[i];
        }
        const double nb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        if (nb > 0) {
          for (int i = 0; i < 3; ++i) b[i] /= nb;
          basis.push_back(b);
        }
      }
      const double scale = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (const auto& e : basis) {
        const double d = v[0] * e[0] + v[1] * e[1] + v[2] * e[2];
        for (int i = 0; i < 3; ++i) v[i] -= d * e[i];
      }
      const double nv = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (nv > 1e-12 * std::max(1.0, scale)) {
        chosen.push_back(p);
        return true;
      }
      return false;
    };
    for (size_t j = 0; j < point_const_.size(); ++j)
      if (used[j] && point_const_[j] && maybe(j) && chosen.size() >= 3) return true;
    for (size_t j = 0; j < point_const_.size(); ++j)
      if (used[j] && !point_const_[j] && maybe(j)) {
        point_const_[j] = 1;
        if (chosen.size() >= 3) return true;
      }
    return false;
  }

  void WriteBack() {  // variable blocks only (bundle_adjustment_caspar.cc:767-801)
    Reconstruction& rec = reconstruction_;
    for (size_t i = 0; i < pose_refs_.size(); ++i) {
      if (pose_const_[i]) continue;
      Rigid3d& dst = pose_refs_[i].is_frame ? rec.frames.at(pose_refs_[i].id).rig_from_world
                                            : rec.images.at(pose_refs_[i].id).cam_from_world;
      std::copy(poses_.begin() + 7 * i, poses_.begin() + 7 * i + 7, dst.params.begin());
    }
    for (size_t k = 0; k < sensor_const_.size(); ++k) {
      if (sensor_const_[k]) continue;
      Rigid3d& dst = rec.rigs.at(sensor_rig_[k]).sensors_from_rig.at(sensor_ids_[k]);
      std::copy(sensors_.begin() + 7 * k, sensors_.begin() + 7 * k + 7, dst.params.begin());
    }
    rec.UpdateCamFromWorld();
    for (size_t k = 0; k < cam_ids_.size(); ++k) {
      Camera& camera = rec.Camera(cam_ids_[k]);
      bool variable = false;
      for (size_t j = 0; j < camera.params.size(); ++j) variable |= cam_const_[k * BA_CAM_STRIDE + j] == 0;
      if (variable) std::copy_n(cams_.begin() + k * BA_CAM_STRIDE, camera.params.size(), camera.params.begin());
    }
    for (size_t j = 0; j < point_ids_.size(); ++j)
      if (!point_const_[j]) std::copy_n(points_.begin() + 3 * j, 3, rec.Point3D(point_ids_[j]).xyz.begin());
  }

  Reconstruction& reconstruction_;
  std::map<std::tuple<bool, uint32_t, bool>, int> pose_index_;
  std::vector<PoseRef> pose_refs_;
  std::unordered_map<camera_t, int> cam_index_, sensor_index_;
  std::unordered_map<point3D_t, int> point_index_;
  std::vector<camera_t> cam_ids_;
  std::vector<point3D_t> point_ids_;
  std::vector<double> poses_, cams_, points_, obs_xy_, sensors_;
  std::vector<int32_t> cam_model_, obs_pose_, obs_cam_, obs_point_, obs_sensor_;
  std::vector<uint8_t> pose_const_, cam_const_, point_const_, sensor_const_;
  std::vector<camera_t> sensor_ids_;
  std::vector<rig_t> sensor_rig_;
  std::vector<int8_t> pose_fixed_t_;

 protected:
  std::map<image_t, std::pair<int, int>> image_slots_;  // parameterized image -> (pose slot, sensor slot or -1)
  // position priors of the flattened problem (filled by PosePriorBundleAdjuster)
  std::vector<int32_t> prior_pose_, prior_sensor_;
  std::vector<double> prior_position_, prior_sqrt_info_;
  int32_t prior_loss_type_ = BA_LOSS_TRIVIAL;
  double prior_loss_scale_ = 1.0;
  uint8_t PoseConst(int slot) const { return pose_const_[slot]; }
  uint8_t SensorConst(int slot) const { return sensor_const_[slot]; }
  Reconstruction& Rec() { return reconstruction_; }
};

// ---------------------------------------------------------------------------------------------
// Pose-prior bundle adjustment (bundle_adjustment.h:236-270, bundle_adjustment_ceres.cc:900-1085)
// ---------------------------------------------------------------------------------------------
struct PosePrior {  // geometry/pose_prior.h:43-77, the fields the adjuster reads
  image_t image_id = 0;  // corr_data_id of a camera sensor
  std::array<double, 3> position{{std::nan(""), std::nan(""), std::nan("")}};
  std::array<double, 9> position_covariance{{std::nan(""), std::nan(""), std::nan(""), std::nan(""), std::nan(""),
                                             std::nan(""), std::nan(""), std::nan(""), std::nan("")}};
  bool HasPosition() const { return std::isfinite(position[0]) && std::isfinite(position[1]) && std::isfinite(position[2]); }
  bool HasPositionCov() const {
    for (const double v : position_covariance)
      if (!std::isfinite(v)) return false;
    return true;
  }
};

struct PosePriorBundleAdjustmentOptions {  // bundle_adjustment.h:252-262 + bundle_adjustment_ceres.h:102-112
  double prior_position_fallback_stddev = 1.0;
  int prior_position_loss_function_type = BA_LOSS_TRIVIAL;
  double prior_position_loss_scale = 2.7955321496988725;  // sqrt(kChiSquare95ThreeDof = 7.815)
  struct RANSACOptions* alignment_ransac_options = nullptr;  // bundle_adjustment.h:258 (nullptr: defaults)
  bool Check() const { return prior_position_fallback_stddev > 0 && prior_position_loss_scale > 0; }
};

// Least-squares similarity dst ~ scale R src + t over all correspondences (Horn's closed form: the
// rotation is the dominant eigenvector of a symmetric 4 x 4 matrix, found by Jacobi sweeps). The
// reference estimates the same transform inside RANSAC (AlignReconstructionToPosePriors). false:
// fewer than three points or a degenerate (collinear) configuration.
inline bool AlignToPositions(const std::vector<std::array<double, 3>>& src, const std::vector<std::array<double, 3>>& dst,
                             double* scale, double R[9], double t[3]) {
  const size_t n = src.size();
  if (n < 3 || dst.size() != n) return false;
  double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { ms[k] += src[i][k] / n; md[k] += dst[i][k] / n; }
  double S[9] = {0}, var = 0;  // S = sum a b^T
  for (size_t i = 0; i < n; ++i) {
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) { a[k] = src[i][k] - ms[k]; b[k] = dst[i][k] - md[k]; var += a[k] * a[k]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) S[3 * r + c] += a[r] * b[c];
  }
  if (var < 1e-24) return false;
  // collinearity: second largest eigenvalue of S^T S negligible
  {
    double M[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[3 * r + c] = S[r] * S[c] + S[3 + r] * S[3 + c] + S[6 + r] * S[6 + c];
    // eigenvalues of the symmetric 3 x 3 by Jacobi
    double A[9];
    std::copy(M, M + 9, A);
    for (int sweep = 0; sweep < 50; ++sweep)
      for (int p = 0; p < 3; ++p)
        for (int q = p + 1; q < 3; ++q) {
          if (std::abs(A[3 * p + q]) < 1e-300) continue;
          const double th = 0.5 * std::atan2(2 * A[3 * p + q], A[3 * q + q] - A[3 * p + p]);
          const double c = std::cos(th), s_ = std::sin(th);
          for (int k = 0; k < 3; ++k) {
            const double akp = A[3 * k + p], akq = A[3 * k + q];
            A[3 * k + p] = c * akp - s_ * akq; A[3 * k + q] = s_ * akp + c * akq;
          }
          for (int k = 0; k < 3; ++k) {
            const double apk = A[3 * p + k], aqk = A[3 * q + k];
            A[3 * p + k] = c * apk - s_ * aqk; A[3 * q + k] = s_ * apk + c * aqk;
          }
        }
    double ev[3] = {A[0], A[4], A[8]};
    std::sort(ev, ev + 3);
    if (std::sqrt(std::max(ev[1], 0.0)) < 1e-12 * std::sqrt(std::max(ev[2], 1e-300))) return false;
  }
  const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
  double N[16] = {Sxx + Syy + Szz, Syz - Szy,       Szx - Sxz,        Sxy - Syx,
                  Syz - Szy,       Sxx - Syy - Szz, Sxy + Syx,        Szx + Sxz,
                  Szx - Sxz,       Sxy + Syx,       -Sxx + Syy - Szz, Syz + Szy,
                  Sxy - Syx,       Szx + Sxz,       Syz + Szy,        -Sxx - Syy + Szz};
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 100; ++sweep)
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (std::abs(N[4 * p + q]) < 1e-300) continue;
        const double th = 0.5 * std::atan2(2 * N[4 * p + q], N[4 * q + q] - N[4 * p + p]);
        const double c = std::cos(th), s_ = std::sin(th);
        for (int k = 0; k < 4; ++k) {
          const double a = N[4 * k + p], b = N[4 * k + q];
          N[4 * k + p] = c * a - s_ * b; N[4 * k + q] = s_ * a + c * b;
        }
        for (int k = 0; k < 4; ++k) {
          const double a = N[4 * p + k], b = N[4 * q + k];
          N[4 * p + k] = c * a - s_ * b; N[4 * q + k] = s_ * a + c * b;
        }
        for (int k = 0; k < 4; ++k) {
          const double a = V[4 * k + p], b = V[4 * k + q];
          V[4 * k + p] = c * a - s_ * b; V[4 * k + q] = s_ * a + c * b;
        }
      }
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (N[5 * k] > N[5 * best]) best = k;
  const double qw = V[best], qx = V[4 + best], qy = V[8 + best], qz = V[12 + best];  // Horn: (w, x, y, z)
  const double q[4] = {qx, qy, qz, qw};
  QuatToRot(q, R);
  double num = 0;
  for (size_t i = 0; i < n; ++i) {
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) { a[k] = src[i][k] - ms[k]; b[k] = dst[i][k] - md[k]; }
    for (int r = 0; r < 3; ++r) num += b[r] * (R[3 * r] * a[0] + R[3 * r + 1] * a[1] + R[3 * r + 2] * a[2]);
  }
  *scale = num / var;
  for (int r = 0; r < 3; ++r) t[r] = md[r] - *scale * (R[3 * r] * ms[0] + R[3 * r + 1] * ms[1] + R[3 * r + 2] * ms[2]);
  return true;
}

// colmap::RANSACOptions (optim/ransac.h:50-83), the fields the alignment uses
struct RANSACOptions {
  double max_error = 0.0;  // <= 0: from the priors' covariances (alignment.cc:284-294)
  double min_inlier_ratio = 0.1;
  double confidence = 0.99;
  double dyn_num_trials_multiplier = 3.0;
  int min_num_trials = 0;
  int max_num_trials = 10000;
  int random_seed = 0;  // (the reference's -1 = nondeterministic; a fixed stream here)
};

// AlignReconstructionToPosePriors' estimator (estimators/alignment.cc:240-299 -> EstimateSim3dRobust): RANSAC
// over 3-point similarity hypotheses, inlier test |dst - (s R src + t)| <= max_error, local refit on the support
// of every improving hypothesis, final least-squares similarity over the best inlier set. Same deterministic
// sample stream as colmap_amd/estimators.py::align_to_positions_robust.
inline bool AlignToPositionsRobust(const std::vector<std::array<double, 3>>& src, const std::vector<std::array<double, 3>>& dst,
                                   double max_error, const RANSACOptions& opt, double* scale, double R[9], double t[3]) {
  const size_t n = src.size();
  if (n < 3 || dst.size() != n || !(max_error > 0)) return false;
  auto lcg = [](uint64_t s) { return s * 6364136223846793005ull + 1442695040888963407ull; };
  auto subset = [&](const std::vector<char>& inl, std::vector<std::array<double, 3>>* a, std::vector<std::array<double, 3>>* b) {
    a->clear(); b->clear();
    for (size_t i = 0; i < n; ++i) if (inl[i]) { a->push_back(src[i]); b->push_back(dst[i]); }
  };
  auto support = [&](double s_, const double* R_, const double* t_, std::vector<char>* inl) {
    size_t c = 0;
    inl->assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
      double e2 = 0;
      for (int r = 0; r < 3; ++r) {
        const double d = dst[i][r] - (s_ * (R_[3 * r] * src[i][0] + R_[3 * r + 1] * src[i][1] + R_[3 * r + 2] * src[i][2]) + t_[r]);
        e2 += d * d;
      }
      if (std::sqrt(e2) <= max_error) { (*inl)[i] = 1; ++c; }
    }
    return c;
  };
  std::vector<char> best_inl, inl, inl2;
  size_t best = 0;
  uint64_t state = lcg(0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)opt.random_seed);
  double needed = opt.max_num_trials;
  std::vector<std::array<double, 3>> a, b;
  for (int trial = 0; trial < opt.max_num_trials && (trial < needed || trial < opt.min_num_trials);) {
    ++trial;
    size_t idx[3];
    for (int k = 0; k < 3;) {
      state = lcg(state);
      const size_t c = (size_t)((state >> 33) % n);
      bool dup = false;
      for (int j = 0; j < k; ++j) dup = dup || idx[j] == c;
      if (!dup) idx[k++] = c;
    }
    a = {src[idx[0]], src[idx[1]], src[idx[2]]};
    b = {dst[idx[0]], dst[idx[1]], dst[idx[2]]};
    double s_, R_[9], t_[3];
    if (!AlignToPositions(a, b, &s_, R_, t_)) continue;
    size_t count = support(s_, R_, t_, &inl);
    if (count <= best) continue;
    for (int it = 0; it < 4 && count >= 3; ++it) {  // local optimisation
      subset(inl, &a, &b);
      if (!AlignToPositions(a, b, &s_, R_, t_)) break;
      const size_t c2 = support(s_, R_, t_, &inl2);
      if (c2 <= count) break;
      inl.swap(inl2);
      count = c2;
    }
    best_inl = inl;
    best = count;
    const double w = std::min(std::max((double)best / n, opt.min_inlier_ratio), 1.0 - 1e-12);
    needed = opt.dyn_num_trials_multiplier * std::log(1.0 - opt.confidence) / std::log(1.0 - w * w * w);
  }
  if (best < 3) return false;
  subset(best_inl, &a, &b);
  return AlignToPositions(a, b, scale, R, t);
}

class PosePriorBundleAdjuster : public Mi355xBundleAdjuster {
 public:
  struct Prepared {  // what has to happen to the reconstruction BEFORE it is flattened
    BundleAdjustmentConfig config;
    std::vector<PosePrior> pose_priors;
    bool use_prior_position = false;
    std::array<double, 3> normalized_from_metric{0, 0, 0};
  };
  // drops unusable priors, aligns + normalises the reconstruction or falls back to the two-camera gauge
  // (bundle_adjustment_ceres.cc:913-936)
  static Prepared Prepare(const BundleAdjustmentConfig& config, const std::vector<PosePrior>& pose_priors,
                          Reconstruction& rec, const PosePriorBundleAdjustmentOptions& prior_options = PosePriorBundleAdjustmentOptions()) {
    Prepared out;
    out.config = config;
    for (const PosePrior& p : pose_priors)
      if (p.HasPosition() && config.HasImage(p.image_id)) out.pose_priors.push_back(p);
    if (out.pose_priors.size() >= 3) {
      std::vector<std::array<double, 3>> src, dst;
      for (const PosePrior& p : out.pose_priors) { src.push_back(rec.ProjectionCenter(p.image_id)); dst.push_back(p.position); }
      double scale, R[9], t[3];
      const RANSACOptions ropt = prior_options.alignment_ransac_options ? *prior_options.alignment_ransac_options : RANSACOptions();
      double max_error = ropt.max_error;
      if (!(max_error > 0)) {  // alignment.cc:284-294: 95 % chi-square quantile (3 dof) of the median prior variance
        std::vector<double> rms;
        for (const PosePrior& p : out.pose_priors) {
          const double tr = p.position_covariance[0] + p.position_covariance[4] + p.position_covariance[8];
          if (p.HasPositionCov() && tr > 0) rms.push_back(tr / 3.0);
        }
        if (rms.empty()) rms.push_back(prior_options.prior_position_fallback_stddev * prior_options.prior_position_fallback_stddev);
        std::sort(rms.begin(), rms.end());
        const double med = rms.size() % 2 ? rms[rms.size() / 2] : 0.5 * (rms[rms.size() / 2 - 1] + rms[rms.size() / 2]);
        max_error = std::sqrt(7.814727903251179 * med);
      }
      if (AlignToPositionsRobust(src, dst, max_error, ropt, &scale, R, t)) {
        rec.Transform(scale, R, t);
        out.use_prior_position = true;
      }
    }
    if (out.use_prior_position) out.normalized_from_metric = rec.NormalizeFixedScale();
    else out.config.FixGauge(BundleAdjustmentGauge::TWO_CAMS_FROM_WORLD);
    return out;
  }

  PosePriorBundleAdjuster(BundleAdjustmentOptions options, PosePriorBundleAdjustmentOptions prior_options,
                          Prepared prepared, Reconstruction& reconstruction)
      : Mi355xBundleAdjuster(std::move(options), prepared.config, reconstruction),
        prior_options_(prior_options), prepared_(std::move(prepared)) {
    COLMAP_AMD_BA_CHECK(prior_options_.Check());
    if (prepared_.use_prior_position) AddPriors();
  }

  bool UsesPriorPositions() const { return prepared_.use_prior_position; }
  size_t NumPriors() const { return prior_pose_.size(); }

  std::shared_ptr<BundleAdjustmentSummary> Solve() override {
    auto summary = Mi355xBundleAdjuster::Solve();
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double back[3] = {-prepared_.normalized_from_metric[0], -prepared_.normalized_from_metric[1],
                            -prepared_.normalized_from_metric[2]};
    Rec().Transform(1.0, I, back);  // Inverse(normalized_from_metric)
    return summary;
  }

 private:
  void AddPriors() {  // AddImagePosePriorToProblem (:986-1038) for every parameterized image
    for (const PosePrior& pr : prepared_.pose_priors) {
      const auto it = image_slots_.find(pr.image_id);
      if (it == image_slots_.end()) continue;
      const int pose_slot = it->second.first, sens_slot = it->second.second;
      const bool const_sensor = sens_slot < 0 || SensorConst(sens_slot);
      if (PoseConst(pose_slot) && const_sensor) continue;
      double cov[9];
      if (pr.HasPositionCov()) std::copy(pr.position_covariance.begin(), pr.position_covariance.end(), cov);
      else {
        const double v = prior_options_.prior_position_fallback_stddev * prior_options_.prior_position_fallback_stddev;
        const double d[9] = {v, 0, 0, 0, v, 0, 0, 0, v};
        std::copy(d, d + 9, cov);
      }
      // LeftSqrtInformation (cost_functions/utils.h:159-161): cov^-1 = L L^T, weight = L^T
      const double a = cov[0], b = cov[1], c = cov[2], d = cov[4], e = cov[5], f = cov[8];
      const double A = d * f - e * e, B = c * e - b * f, Cc = b * e - c * d;
      const double det = a * A + b * B + c * Cc;
      const double inf[9] = {A / det, B / det, Cc / det, B / det, (a * f - c * c) / det, (b * c - a * e) / det,
                             Cc / det, (b * c - a * e) / det, (a * d - b * b) / det};
      double L[9] = {0};
      L[0] = std::sqrt(inf[0]);
      L[3] = inf[3] / L[0]; L[6] = inf[6] / L[0];
      L[4] = std::sqrt(inf[4] - L[3] * L[3]);
      L[7] = (inf[7] - L[6] * L[3]) / L[4];
      L[8] = std::sqrt(inf[8] - L[6] * L[6] - L[7] * L[7]);
      prior_pose_.push_back(pose_slot);
      prior_sensor_.push_back(sens_slot);
      for (int k = 0; k < 3; ++k) prior_position_.push_back(pr.position[k] + prepared_.normalized_from_metric[k]);
      for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) prior_sqrt_info_.push_back(L[3 * col + r]);  // L^T, row-major
    }
    prior_loss_type_ = prior_options_.prior_position_loss_function_type;
    prior_loss_scale_ = prior_options_.prior_position_loss_scale;
  }

  PosePriorBundleAdjustmentOptions prior_options_;
  Prepared prepared_;
};

// CreatePosePriorBundleAdjuster (bundle_adjustment.cc:373-395)
inline std::unique_ptr<BundleAdjuster> CreatePosePriorBundleAdjuster(const BundleAdjustmentOptions& options,
                                                                     const PosePriorBundleAdjustmentOptions& prior_options,
                                                                     const BundleAdjustmentConfig& config,
                                                                     std::vector<PosePrior> pose_priors,
                                                                     Reconstruction& reconstruction) {
  if (options.backend != BundleAdjustmentBackend::MI355X)
    throw std::invalid_argument("BundleAdjustmentBackend CERES / CASPAR are not built here (they need Ceres / CUDA)");
  return std::make_unique<PosePriorBundleAdjuster>(options, prior_options,
                                                   PosePriorBundleAdjuster::Prepare(config, pose_priors, reconstruction, prior_options),
                                                   reconstruction);
}

// CreateDefaultBundleAdjuster (bundle_adjustment.cc:314-334): backend switch.
inline std::unique_ptr<BundleAdjuster> CreateDefaultBundleAdjuster(const BundleAdjustmentOptions& options,
                                                                   const BundleAdjustmentConfig& config,
                                                                   Reconstruction& reconstruction) {
  switch (options.backend) {
    case BundleAdjustmentBackend::MI355X:
      return std::make_unique<Mi355xBundleAdjuster>(options, config, reconstruction);
    case BundleAdjustmentBackend::CERES:
    case BundleAdjustmentBackend::CASPAR:
      break;
  }
  throw std::invalid_argument("BundleAdjustmentBackend CERES / CASPAR are not built here (they need Ceres / CUDA)");
}

}  // namespace colmap_amd

#endif  // COLMAP_AMD_BUNDLE_ADJUSTMENT_HPP_
