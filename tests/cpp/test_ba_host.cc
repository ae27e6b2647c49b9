// Exercises include/colmap_amd/bundle_adjustment.hpp (the C++ host side of the BA path).
//   test_ba_host counts FILE        host-only: flatten FILE, print the problem statistics
//   test_ba_host solve FILE OUT     flatten, solve on the GPU, write the adjusted reconstruction
//   test_ba_host api                host-only: config / options / factory behaviour
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "colmap_amd/bundle_adjustment.hpp"

using namespace colmap_amd;

#define EXPECT(cond)                                                                 \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "%s:%d: EXPECT failed: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static Rigid3d ReadPose(std::istream& f) {
  Rigid3d p;
  for (auto& v : p.params) f >> v;
  return p;
}

struct Spec {
  Reconstruction rec;
  BundleAdjustmentConfig config;
  BundleAdjustmentOptions options;
};

static void ReadSpec(const std::string& path, Spec* s) {
  std::ifstream f(path);
  if (!f.is_open()) throw std::runtime_error("cannot open " + path);
  std::string tag;
  size_t n;
  f >> tag >> n;  // cameras
  for (size_t i = 0; i < n; ++i) {
    Camera c;
    size_t np;
    f >> c.camera_id >> c.model_id >> c.width >> c.height >> np;
    c.params.resize(np);
    for (auto& v : c.params) f >> v;
    s->rec.cameras[c.camera_id] = c;
  }
  f >> tag >> n;  // rigs
  for (size_t i = 0; i < n; ++i) {
    Rig r;
    size_t ns;
    f >> r.rig_id >> r.ref_camera_id >> ns;
    for (size_t j = 0; j < ns; ++j) {
      camera_t cid;
      f >> cid;
      r.sensors_from_rig[cid] = ReadPose(f);
    }
    s->rec.rigs[r.rig_id] = r;
  }
  f >> tag >> n;  // frames
  for (size_t i = 0; i < n; ++i) {
    Frame fr;
    size_t ni;
    f >> fr.frame_id >> fr.rig_id;
    fr.rig_from_world = ReadPose(f);
    f >> ni;
    fr.image_ids.resize(ni);
    for (auto& v : fr.image_ids) f >> v;
    s->rec.frames[fr.frame_id] = fr;
  }
  f >> tag >> n;  // images
  for (size_t i = 0; i < n; ++i) {
    Image im;
    long long frame;
    size_t np;
    f >> im.image_id >> im.camera_id >> frame;
    if (frame >= 0) im.frame_id = static_cast<frame_t>(frame);
    im.cam_from_world = ReadPose(f);
    f >> np;
    im.points2D.resize(np);
    for (auto& p : im.points2D) {
      long long pid;
      f >> p.xy[0] >> p.xy[1] >> pid;
      p.point3D_id = pid < 0 ? kInvalidPoint3DId : static_cast<point3D_t>(pid);
    }
    s->rec.images[im.image_id] = im;
  }
  f >> tag >> n;  // points
  for (size_t i = 0; i < n; ++i) {
    point3D_t pid;
    Point3D p;
    size_t nt;
    f >> pid >> p.xyz[0] >> p.xyz[1] >> p.xyz[2] >> nt;
    p.track.resize(nt);
    for (auto& el : p.track) f >> el.image_id >> el.point2D_idx;
    s->rec.points3D[pid] = p;
  }
  int gauge;
  f >> tag >> gauge;  // gauge
  s->config.FixGauge(static_cast<BundleAdjustmentGauge>(gauge));
  auto read_ids = [&](auto&& fn) {
    f >> tag >> n;
    for (size_t i = 0; i < n; ++i) {
      unsigned long long id;
      f >> id;
      fn(id);
    }
  };
  read_ids([&](auto id) { s->config.AddImage(static_cast<image_t>(id)); });
  read_ids([&](auto id) { s->config.SetConstantCamIntrinsics(static_cast<camera_t>(id)); });
  read_ids([&](auto id) { s->config.SetConstantRigFromWorldPose(static_cast<frame_t>(id)); });
  read_ids([&](auto id) { s->config.SetConstantSensorFromRigPose(static_cast<camera_t>(id)); });
  read_ids([&](auto id) { s->config.AddVariablePoint(id); });
  read_ids([&](auto id) { s->config.AddConstantPoint(id); });
  read_ids([&](auto id) { s->config.IgnorePoint(id); });
  int b[7], loss, max_iter;
  double loss_scale, grad_tol;
  f >> tag;
  for (int& v : b) f >> v;
  f >> s->options.min_track_length >> loss >> loss_scale >> max_iter >> grad_tol;
  if (!f.good()) throw std::runtime_error("malformed spec " + path);
  s->options.refine_focal_length = b[0];
  s->options.refine_principal_point = b[1];
  s->options.refine_extra_params = b[2];
  s->options.refine_sensor_from_rig = b[3];
  s->options.refine_rig_from_world = b[4];
  s->options.refine_points3D = b[5];
  s->options.constant_rig_from_world_rotation = b[6];
  s->options.gpu_index = "0";
  s->options.mi355x->loss_function_type = static_cast<Mi355xBundleAdjustmentOptions::LossFunctionType>(loss);
  s->options.mi355x->loss_function_scale = loss_scale;
  s->options.mi355x->solver_options.max_num_iterations = max_iter;
  s->options.mi355x->solver_options.gradient_tolerance = grad_tol;
}

static int Counts(const std::string& path) {
  Spec s;
  ReadSpec(path, &s);
  Mi355xBundleAdjuster ba(s.options, s.config, s.rec);
  std::printf("%zu %zu %zu %zu %zu\n", ba.NumResidualsReduced(), ba.NumEffectiveParametersReduced(), ba.NumPoseBlocks(),
              ba.NumConstantPoseBlocks(), s.config.NumResiduals(s.rec));
  return 0;
}

// priors file: one line per prior "image_id x y z c00 c01 .. c22" (nan = not given)
static std::vector<PosePrior> ReadPriors(const std::string& path) {
  std::vector<PosePrior> priors;
  std::ifstream f(path);
  if (!f.is_open()) throw std::runtime_error("cannot open " + path);
  unsigned long long id;
  while (f >> id) {
    PosePrior p;
    p.image_id = static_cast<image_t>(id);
    std::string tok;
    auto num = [&]() { f >> tok; return tok == "nan" ? std::nan("") : std::stod(tok); };
    for (double& v : p.position) v = num();
    for (double& v : p.position_covariance) v = num();
    priors.push_back(p);
  }
  return priors;
}

static int Solve(const std::string& path, const std::string& out, const std::string& priors_path = "") {
  Spec s;
  ReadSpec(path, &s);
  std::unique_ptr<BundleAdjuster> ba;
  if (priors_path.empty()) {
    ba = CreateDefaultBundleAdjuster(s.options, s.config, s.rec);
  } else {
    s.options.mi355x->solver_options.function_tolerance = 1e-12;
    ba = CreatePosePriorBundleAdjuster(s.options, PosePriorBundleAdjustmentOptions(), s.config, ReadPriors(priors_path), s.rec);
    auto* pp = dynamic_cast<PosePriorBundleAdjuster*>(ba.get());
    std::printf("priors used %d count %zu\n", pp->UsesPriorPositions() ? 1 : 0, pp->NumPriors());
  }
  auto summary = ba->Solve();
  std::ofstream f(out);
  f << std::setprecision(17);
  f << static_cast<int>(summary->termination_type) << " " << summary->num_residuals << " "
    << summary->num_effective_parameters << " " << summary->num_iterations << " " << summary->initial_cost << " "
    << summary->final_cost << "\n";
  for (const auto& [id, im] : s.rec.images) {
    f << "image " << id;
    for (double v : im.cam_from_world.params) f << " " << v;
    f << "\n";
  }
  for (const auto& [id, fr] : s.rec.frames) {
    f << "frame " << id;
    for (double v : fr.rig_from_world.params) f << " " << v;
    f << "\n";
  }
  for (const auto& [id, c] : s.rec.cameras) {
    f << "camera " << id;
    for (double v : c.params) f << " " << v;
    f << "\n";
  }
  for (const auto& [id, p] : s.rec.points3D) f << "point " << id << " " << p.xyz[0] << " " << p.xyz[1] << " " << p.xyz[2] << "\n";
  std::printf("%s\n", summary->BriefReport().c_str());
  return summary->IsSolutionUsable() ? 0 : 4;
}

static int Api() {
  BundleAdjustmentConfig config;
  EXPECT(config.FixedGauge() == BundleAdjustmentGauge::UNSPECIFIED);
  config.AddImage(3);
  config.AddImage(1);
  config.AddImage(3);
  EXPECT(config.NumImages() == 2 && config.HasImage(1) && !config.HasImage(2));
  EXPECT(*config.Images().begin() == 1);
  config.RemoveImage(1);
  EXPECT(config.NumImages() == 1);
  config.SetConstantCamIntrinsics(7);
  EXPECT(config.HasConstantCamIntrinsics(7));
  config.SetVariableCamIntrinsics(7);
  EXPECT(!config.HasConstantCamIntrinsics(7));
  config.SetConstantRigFromWorldPose(2);
  EXPECT(config.HasConstantRigFromWorldPose(2) && !config.HasConstantRigFromWorldPose(3));
  config.SetConstantSensorFromRigPose(5);
  EXPECT(config.HasConstantSensorFromRigPose(5));
  config.AddVariablePoint(10);
  config.AddConstantPoint(11);
  config.IgnorePoint(12);
  EXPECT(config.HasPoint(10) && config.HasPoint(11) && !config.HasPoint(12) && config.IsIgnoredPoint(12));
  bool threw = false;
  try {
    config.AddConstantPoint(10);  // already variable (bundle_adjustment.cc:206-211)
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  EXPECT(threw);
  BundleAdjustmentOptions options;
  EXPECT(options.Check() && options.refine_focal_length && !options.refine_principal_point);
  EXPECT(options.mi355x->solver_options.max_num_iterations == 100);             // bundle_adjustment_ceres.cc:102-115
  EXPECT(options.mi355x->solver_options.max_linear_solver_iterations == 200);
  EXPECT(options.mi355x->solver_options.gradient_tolerance == 1e-4);
  options.min_track_length = -1;
  EXPECT(!options.Check());
  options = BundleAdjustmentOptions();
  Reconstruction rec;
  options.backend = BundleAdjustmentBackend::CERES;
  threw = false;
  try {
    CreateDefaultBundleAdjuster(options, BundleAdjustmentConfig(), rec);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  EXPECT(threw);
  options.backend = BundleAdjustmentBackend::MI355X;
  auto ba = CreateDefaultBundleAdjuster(options, BundleAdjustmentConfig(), rec);
  auto summary = ba->Solve();  // empty problem: default summary without touching the GPU (:667-669)
  EXPECT(summary->num_residuals == 0 && summary->termination_type == BundleAdjustmentTerminationType::FAILURE);
  EXPECT(!summary->IsSolutionUsable());
  Rigid3d a, b;
  a.params = {0, 0, std::sin(0.25), std::cos(0.25), 1, 2, 3};
  b.params = {0, 0, 0, 1, 0.5, 0, 0};
  const Rigid3d c = Compose(a, b);
  EXPECT(std::abs(c.params[4] - (1 + 0.5 * std::cos(0.5))) < 1e-15 && std::abs(c.params[5] - (2 + 0.5 * std::sin(0.5))) < 1e-15);
  // AlignToPositions recovers a known similarity; Reconstruction::Transform / NormalizeFixedScale move poses
  // and points consistently (projection centres transform like points)
  {
    const double qs[4] = {0.1, -0.2, 0.3, 0.9};
    double qn[4], Rt[9];
    const double nq = std::sqrt(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
    for (int i = 0; i < 4; ++i) qn[i] = qs[i] / nq;
    QuatToRot(qn, Rt);
    const double sc = 1.7, tt[3] = {3.0, -2.0, 1.0};
    std::vector<std::array<double, 3>> src, dst;
    for (int i = 0; i < 7; ++i) {
      std::array<double, 3> a_{{std::sin(1.3 * i), std::cos(0.7 * i) * 2, 0.3 * i - 1}}, b_{};
      for (int r = 0; r < 3; ++r) b_[r] = sc * (Rt[3 * r] * a_[0] + Rt[3 * r + 1] * a_[1] + Rt[3 * r + 2] * a_[2]) + tt[r];
      src.push_back(a_);
      dst.push_back(b_);
    }
    double s2, R2[9], t2[3];
    EXPECT(AlignToPositions(src, dst, &s2, R2, t2));
    EXPECT(std::abs(s2 - sc) < 1e-12);
    for (int i = 0; i < 9; ++i) EXPECT(std::abs(R2[i] - Rt[i]) < 1e-12);
    for (int i = 0; i < 3; ++i) EXPECT(std::abs(t2[i] - tt[i]) < 1e-11);
    std::vector<std::array<double, 3>> line = {{{0, 0, 0}}, {{1, 1, 1}}, {{2, 2, 2}}};
    EXPECT(!AlignToPositions(line, line, &s2, R2, t2));  // collinear
    double qr[4];
    RotToQuat(Rt, qr);
    for (int i = 0; i < 4; ++i) EXPECT(std::abs(qr[i] - qn[i]) < 1e-14);
    Reconstruction rec;
    for (image_t id = 1; id <= 3; ++id) {
      colmap_amd::Image im;
      im.image_id = id;
      im.cam_from_world.params = {0, std::sin(0.1 * id), 0, std::cos(0.1 * id), 0.5 * id, -1.0, 2.0 + id};
      rec.images[id] = im;
    }
    rec.points3D[1].xyz = {1, 2, 3};
    const auto c_before = rec.ProjectionCenter(2);
    rec.Transform(sc, Rt, tt);
    const auto c_after = rec.ProjectionCenter(2);
    for (int r = 0; r < 3; ++r)
      EXPECT(std::abs(c_after[r] - (sc * (Rt[3 * r] * c_before[0] + Rt[3 * r + 1] * c_before[1] + Rt[3 * r + 2] * c_before[2]) + tt[r])) < 1e-12);
    EXPECT(std::abs(rec.points3D[1].xyz[0] - (sc * (Rt[0] * 1 + Rt[1] * 2 + Rt[2] * 3) + tt[0])) < 1e-12);
    const auto shift = rec.NormalizeFixedScale();
    double mean[3] = {0, 0, 0};
    for (image_t id = 1; id <= 3; ++id) {
      const auto c = rec.ProjectionCenter(id);
      for (int r = 0; r < 3; ++r) mean[r] += c[r] / 3;
    }
    for (int r = 0; r < 3; ++r) EXPECT(std::abs(mean[r]) < 1e-12 && std::isfinite(shift[r]));
    PosePrior pr;
    EXPECT(!pr.HasPosition() && !pr.HasPositionCov());
    EXPECT(PosePriorBundleAdjustmentOptions().Check());
  }
  std::printf("api OK\n");
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc >= 2 && std::string(argv[1]) == "api") return Api();
    if (argc >= 3 && std::string(argv[1]) == "counts") return Counts(argv[2]);
    if (argc >= 4 && std::string(argv[1]) == "solve") return Solve(argv[2], argv[3]);
    if (argc >= 5 && std::string(argv[1]) == "prior") return Solve(argv[2], argv[3], argv[4]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  std::fprintf(stderr, "usage: test_ba_host api | counts FILE | solve FILE OUT | prior FILE OUT PRIORS\n");
  return 2;
}
