// colmap_amd/mvs.hpp -- C++ host side of the MI355X PatchMatch path.
//
// The reference's host code for this path is C++ (src/colmap/mvs/{mat,depth_map,normal_map,image,
// consistency_graph,patch_match,patch_match_options}.{h,cc}); its third-party dependencies (Eigen,
// glog, OpenImageIO, Boost) are not available in this build environment, so the same interface is
// restated here on the standard library only, on top of the C ABI (colmap_amd_pm.h). Names, argument
// meaning, defaults and error behaviour follow the reference classes cited at each declaration; the
// namespace is colmap_amd::mvs instead of colmap::mvs. A failed check throws std::invalid_argument
// (THROW_CHECK / LOG(FATAL_THROW) throw in the reference, util/logging.h), a device failure throws
// std::runtime_error (CUDA_SAFE_CALL, util/cudacc.cc:56-65).
//
// Header-only; link with libcolmap_amd.so.
#ifndef COLMAP_AMD_MVS_HPP_
#define COLMAP_AMD_MVS_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../colmap_amd_fusion.h"
#include "../colmap_amd_pm.h"

namespace colmap_amd {
namespace mvs {

#define COLMAP_AMD_CHECK(cond)                                                                  \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      std::ostringstream os_;                                                                   \
      os_ << "[" << __FILE__ << ":" << __LINE__ << "] Check failed: " #cond;                     \
      throw std::invalid_argument(os_.str());                                                   \
    }                                                                                           \
  } while (0)

#define COLMAP_AMD_CHECK_MSG(cond, msg)                                                         \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      std::ostringstream os_;                                                                   \
      os_ << "[" << __FILE__ << ":" << __LINE__ << "] Check failed: " #cond " " << msg;          \
      throw std::invalid_argument(os_.str());                                                   \
    }                                                                                           \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Mat<T> (reference mvs/mat.h:40-212): slice-major W x H x D array with the `W&H&D&` + raw
// little-endian file format.
// ---------------------------------------------------------------------------------------------
template <typename T>
class Mat {
 public:
  Mat() : Mat(0, 0, 0) {}
  Mat(size_t width, size_t height, size_t depth) : width_(width), height_(height), depth_(depth) {
    data_.resize(width_ * height_ * depth_, 0);
  }

  size_t GetWidth() const { return width_; }
  size_t GetHeight() const { return height_; }
  size_t GetDepth() const { return depth_; }
  size_t GetNumBytes() const { return data_.size() * sizeof(T); }

  T Get(size_t row, size_t col, size_t slice = 0) const {
    return data_.at(slice * width_ * height_ + row * width_ + col);
  }
  void GetSlice(size_t row, size_t col, T* values) const {
    for (size_t slice = 0; slice < depth_; ++slice) values[slice] = Get(row, col, slice);
  }
  T* GetPtr() { return data_.data(); }
  const T* GetPtr() const { return data_.data(); }
  const std::vector<T>& GetData() const { return data_; }

  void Set(size_t row, size_t col, T value) { Set(row, col, 0, value); }
  void Set(size_t row, size_t col, size_t slice, T value) {
    data_.at(slice * width_ * height_ + row * width_ + col) = value;
  }
  void Fill(T value) { std::fill(data_.begin(), data_.end(), value); }

  // mat.h:150-186
  void Read(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    COLMAP_AMD_CHECK_MSG(file.is_open(), path);
    char unused;
    file >> width_ >> unused >> height_ >> unused >> depth_ >> unused;
    COLMAP_AMD_CHECK_MSG(file.good() && width_ > 0 && height_ > 0 && depth_ > 0, path);
    data_.resize(width_ * height_ * depth_);
    file.read(reinterpret_cast<char*>(data_.data()), static_cast<std::streamsize>(GetNumBytes()));
    COLMAP_AMD_CHECK_MSG(static_cast<size_t>(file.gcount()) == GetNumBytes(), path);
  }
  // mat.h:188-204
  void Write(const std::string& path) const {
    std::ofstream file(path, std::ios::binary);
    COLMAP_AMD_CHECK_MSG(file.is_open(), path);
    file << width_ << "&" << height_ << "&" << depth_ << "&";
    file.write(reinterpret_cast<const char*>(data_.data()), static_cast<std::streamsize>(GetNumBytes()));
  }

 protected:
  size_t width_ = 0;
  size_t height_ = 0;
  size_t depth_ = 0;
  std::vector<T> data_;
};

// DepthMap (reference mvs/depth_map.h:44-68): Mat<float> with the depth range it was computed in.
class DepthMap : public Mat<float> {
 public:
  DepthMap() : DepthMap(0, 0, -1.0f, -1.0f) {}
  DepthMap(size_t width, size_t height, float depth_min, float depth_max)
      : Mat<float>(width, height, 1), depth_min_(depth_min), depth_max_(depth_max) {}
  DepthMap(const Mat<float>& mat, float depth_min, float depth_max)
      : Mat<float>(mat.GetWidth(), mat.GetHeight(), mat.GetDepth()), depth_min_(depth_min), depth_max_(depth_max) {
    COLMAP_AMD_CHECK(mat.GetDepth() == 1);  // depth_map.cc:49
    data_ = mat.GetData();
  }
  float GetDepthMin() const { return depth_min_; }
  float GetDepthMax() const { return depth_max_; }
  float Get(size_t row, size_t col) const { return data_.at(row * width_ + col); }

 private:
  float depth_min_ = -1.0f;
  float depth_max_ = -1.0f;
};

// NormalMap (reference mvs/normal_map.h:44-60): three slices nx, ny, nz.
class NormalMap : public Mat<float> {
 public:
  NormalMap() : Mat<float>(0, 0, 3) {}
  NormalMap(size_t width, size_t height) : Mat<float>(width, height, 3) {}
  explicit NormalMap(const Mat<float>& mat) : Mat<float>(mat.GetWidth(), mat.GetHeight(), mat.GetDepth()) {
    COLMAP_AMD_CHECK(mat.GetDepth() == 3);  // normal_map.cc:45
    data_ = mat.GetData();
  }
};

// Grey bitmap held by an Image: the slice of colmap::Bitmap (sensor/bitmap.h) this path uses.
class Bitmap {
 public:
  Bitmap() = default;
  Bitmap(int width, int height, std::vector<uint8_t> grey)
      : width_(width), height_(height), data_(std::move(grey)) {
    COLMAP_AMD_CHECK(data_.size() == static_cast<size_t>(width_) * height_);
  }
  int Width() const { return width_; }
  int Height() const { return height_; }
  bool IsGrey() const { return true; }
  bool IsEmpty() const { return data_.empty(); }
  const std::vector<uint8_t>& RowMajorData() const { return data_; }

 private:
  int width_ = 0, height_ = 0;
  std::vector<uint8_t> data_;
};

// Pose helpers (reference mvs/image.cc:97-150), float like the reference.
inline void ComposeProjectionMatrix(const float K[9], const float R[9], const float T[3], float P[12]) {
  float RT[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) RT[4 * r + c] = R[3 * r + c];
    RT[4 * r + 3] = T[r];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c)
      P[4 * r + c] = K[3 * r] * RT[c] + K[3 * r + 1] * RT[4 + c] + K[3 * r + 2] * RT[8 + c];
}

inline void ComputeProjectionCenter(const float R[9], const float T[3], float C[3]) {
  for (int i = 0; i < 3; ++i) C[i] = -(R[i] * T[0] + R[3 + i] * T[1] + R[6 + i] * T[2]);
}

// Image (reference mvs/image.h:40-98).
class Image {
 public:
  Image() = default;
  Image(const std::string& path, size_t width, size_t height, const float K[9], const float R[9], const float T[3])
      : path_(path), width_(width), height_(height) {
    std::memcpy(K_, K, sizeof(K_));
    std::memcpy(R_, R, sizeof(R_));
    std::memcpy(T_, T, sizeof(T_));
    ComposeProjectionMatrix(K_, R_, T_, P_);
  }
  void SetBitmap(Bitmap bitmap) {
    COLMAP_AMD_CHECK(width_ == static_cast<size_t>(bitmap.Width()));    // image.cc:60-61
    COLMAP_AMD_CHECK(height_ == static_cast<size_t>(bitmap.Height()));
    bitmap_ = std::move(bitmap);
  }
  const Bitmap& GetBitmap() const { return bitmap_; }
  const std::string& GetPath() const { return path_; }
  size_t GetWidth() const { return width_; }
  size_t GetHeight() const { return height_; }
  const float* GetK() const { return K_; }
  const float* GetR() const { return R_; }
  const float* GetT() const { return T_; }
  const float* GetP() const { return P_; }

 private:
  std::string path_;
  size_t width_ = 0, height_ = 0;
  float K_[9] = {0}, R_[9] = {0}, T_[3] = {0}, P_[12] = {0};
  Bitmap bitmap_;
};

// ConsistencyGraph (reference mvs/consistency_graph.h:44-78, .cc:42-139).
class ConsistencyGraph {
 public:
  ConsistencyGraph() = default;
  ConsistencyGraph(size_t width, size_t height, std::vector<int> data)
      : width_(width), height_(height), data_(std::move(data)) {
    InitializeMap();
  }
  size_t GetNumBytes() const { return (data_.size() + map_.size()) * sizeof(int); }
  void GetImageIdxs(int row, int col, int* num_images, const int** image_idxs) const {
    const int index = map_.at(static_cast<size_t>(row) * width_ + col);
    if (index == kNoConsistentImageIds) {
      *num_images = 0;
      *image_idxs = nullptr;
    } else {
      *num_images = data_.at(index);
      *image_idxs = &data_.at(index + 1);
    }
  }
  void Read(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    COLMAP_AMD_CHECK_MSG(file.is_open(), path);
    size_t depth = 0;
    char unused;
    file >> width_ >> unused >> height_ >> unused >> depth >> unused;
    COLMAP_AMD_CHECK(width_ > 0 && height_ > 0 && depth > 0);
    const std::streampos pos = file.tellg();
    file.seekg(0, std::ios::end);
    const size_t num_bytes = static_cast<size_t>(file.tellg() - pos);
    data_.resize(num_bytes / sizeof(int));
    file.seekg(pos);
    file.read(reinterpret_cast<char*>(data_.data()), static_cast<std::streamsize>(data_.size() * sizeof(int)));
    InitializeMap();
  }
  void Write(const std::string& path) const {
    std::ofstream file(path, std::ios::binary);
    COLMAP_AMD_CHECK_MSG(file.is_open(), path);
    file << width_ << "&" << height_ << "&" << 1 << "&";
    file.write(reinterpret_cast<const char*>(data_.data()), static_cast<std::streamsize>(data_.size() * sizeof(int)));
  }

 private:
  static constexpr int kNoConsistentImageIds = -1;
  void InitializeMap() {
    map_.assign(width_ * height_, kNoConsistentImageIds);
    for (size_t i = 0; i < data_.size();) {
      COLMAP_AMD_CHECK_MSG(i + 2 < data_.size(), "Corrupt consistency graph: insufficient data at offset " << i);
      const int col = data_.at(i), row = data_.at(i + 1), num_images = data_.at(i + 2);
      COLMAP_AMD_CHECK_MSG(num_images >= 0, "Corrupt consistency graph: negative num_images at offset " << i);
      COLMAP_AMD_CHECK(col >= 0 && col < static_cast<int>(width_));
      COLMAP_AMD_CHECK(row >= 0 && row < static_cast<int>(height_));
      if (num_images > 0) map_[static_cast<size_t>(row) * width_ + col] = static_cast<int>(i + 2);
      i += 3 + num_images;
    }
  }
  size_t width_ = 0, height_ = 0;
  std::vector<int> data_;
  std::vector<int> map_;
};

// PatchMatchOptions (reference mvs/patch_match_options.h:37-126, Check patch_match_options.cc:73-100).
struct PatchMatchOptions {
  int max_image_size = -1;
  std::string gpu_index = "-1";
  double depth_min = -1.0f;
  double depth_max = -1.0f;
  int window_radius = 5;
  int window_step = 1;
  double sigma_spatial = -1;
  double sigma_color = 0.2f;
  int num_samples = 15;
  double ncc_sigma = 0.6f;
  double min_triangulation_angle = 1.0f;
  double incident_angle_sigma = 0.9f;
  int num_iterations = 5;
  bool geom_consistency = true;
  double geom_consistency_regularizer = 0.3f;
  double geom_consistency_max_cost = 3.0f;
  bool filter = true;
  double filter_min_ncc = 0.1f;
  double filter_min_triangulation_angle = 3.0f;
  int filter_min_num_consistent = 2;
  double filter_geom_consistency_max_cost = 1.0f;
  double cache_size = 32.0;
  bool allow_missing_files = false;
  bool write_consistency_graph = false;
  int num_threads = -1;

  static constexpr int kMaxPatchMatchWindowRadius = 32;

  bool Check() const {
#define COLMAP_AMD_OPTION(cond) \
  if (!(cond)) {                \
    return false;               \
  }
    if (depth_min != -1.0f || depth_max != -1.0f) {
      COLMAP_AMD_OPTION(depth_min <= depth_max);
      COLMAP_AMD_OPTION(depth_min >= 0.0f);
    }
    COLMAP_AMD_OPTION(window_radius <= kMaxPatchMatchWindowRadius);
    COLMAP_AMD_OPTION(sigma_color > 0.0f);
    COLMAP_AMD_OPTION(window_radius > 0);
    COLMAP_AMD_OPTION(window_step > 0);
    COLMAP_AMD_OPTION(window_step <= 2);
    COLMAP_AMD_OPTION(num_samples > 0);
    COLMAP_AMD_OPTION(ncc_sigma > 0.0f);
    COLMAP_AMD_OPTION(min_triangulation_angle >= 0.0f);
    COLMAP_AMD_OPTION(min_triangulation_angle < 180.0f);
    COLMAP_AMD_OPTION(incident_angle_sigma > 0.0f);
    COLMAP_AMD_OPTION(num_iterations > 0);
    COLMAP_AMD_OPTION(geom_consistency_regularizer >= 0.0f);
    COLMAP_AMD_OPTION(geom_consistency_max_cost >= 0.0f);
    COLMAP_AMD_OPTION(filter_min_ncc >= -1.0f);
    COLMAP_AMD_OPTION(filter_min_ncc <= 1.0f);
    COLMAP_AMD_OPTION(filter_min_triangulation_angle >= 0.0f);
    COLMAP_AMD_OPTION(filter_min_triangulation_angle <= 180.0f);
    COLMAP_AMD_OPTION(filter_min_num_consistent >= 0);
    COLMAP_AMD_OPTION(filter_geom_consistency_max_cost >= 0.0f);
    COLMAP_AMD_OPTION(cache_size > 0);
    COLMAP_AMD_OPTION(num_threads >= -1);
#undef COLMAP_AMD_OPTION
    return true;
  }
};

inline std::vector<int> CSVToIntVector(const std::string& csv) {  // util/string.h CSVToVector<int>
  std::vector<int> out;
  std::stringstream ss(csv);
  std::string item;
  while (std::getline(ss, item, ',')) {
    const auto b = item.find_first_not_of(" \t"), e = item.find_last_not_of(" \t");
    if (b == std::string::npos) continue;
    out.push_back(std::stoi(item.substr(b, e - b + 1)));
  }
  return out;
}

// PatchMatch (reference mvs/patch_match.h:55-96, patch_match.cc:47-154). The pimpl that is a
// PatchMatchCuda in the reference is a pm_handle of the C ABI here.
class PatchMatch {
 public:
  struct Problem {
    int ref_image_idx = -1;                          // index of the reference image
    std::vector<int> src_image_idxs;                 // indices of the source images
    std::vector<Image>* images = nullptr;            // all images, borrowed
    std::vector<DepthMap>* depth_maps = nullptr;     // photometric maps of all images (geom_consistency)
    std::vector<NormalMap>* normal_maps = nullptr;

    void Print(std::ostream& os = std::cout) const {  // patch_match.cc:47-65
      os << "PatchMatch::Problem\nref_image_idx: " << ref_image_idx << "\nsrc_image_idxs: ";
      for (size_t i = 0; i < src_image_idxs.size(); ++i) os << (i ? " " : "") << src_image_idxs[i];
      os << std::endl;
    }
  };

  PatchMatch(const PatchMatchOptions& options, const Problem& problem) : options_(options), problem_(problem) {}
  ~PatchMatch() {
    if (handle_) pm_destroy(handle_);
  }
  PatchMatch(const PatchMatch&) = delete;
  PatchMatch& operator=(const PatchMatch&) = delete;

  // patch_match.cc:67-126
  void Check() const {
    COLMAP_AMD_CHECK(options_.Check());
    COLMAP_AMD_CHECK(!options_.gpu_index.empty());
    const std::vector<int> gpu_indices = CSVToIntVector(options_.gpu_index);
    COLMAP_AMD_CHECK(gpu_indices.size() == 1);
    COLMAP_AMD_CHECK(gpu_indices[0] >= -1);

    COLMAP_AMD_CHECK(problem_.images != nullptr);
    if (options_.geom_consistency) {
      COLMAP_AMD_CHECK(problem_.depth_maps != nullptr);
      COLMAP_AMD_CHECK(problem_.normal_maps != nullptr);
      COLMAP_AMD_CHECK(problem_.depth_maps->size() == problem_.images->size());
      COLMAP_AMD_CHECK(problem_.normal_maps->size() == problem_.images->size());
    }
    COLMAP_AMD_CHECK(problem_.src_image_idxs.size() > 0);

    std::set<int> unique_image_idxs(problem_.src_image_idxs.begin(), problem_.src_image_idxs.end());
    unique_image_idxs.insert(problem_.ref_image_idx);
    COLMAP_AMD_CHECK(problem_.src_image_idxs.size() + 1 == unique_image_idxs.size());

    for (const int image_idx : unique_image_idxs) {
      COLMAP_AMD_CHECK_MSG(image_idx >= 0, image_idx);
      COLMAP_AMD_CHECK_MSG(image_idx < static_cast<int>(problem_.images->size()), image_idx);
      const Image& image = problem_.images->at(image_idx);
      COLMAP_AMD_CHECK_MSG(image.GetBitmap().Width() > 0, image_idx);
      COLMAP_AMD_CHECK_MSG(image.GetBitmap().Height() > 0, image_idx);
      COLMAP_AMD_CHECK_MSG(image.GetBitmap().IsGrey(), image_idx);
      COLMAP_AMD_CHECK_MSG(image.GetWidth() == static_cast<size_t>(image.GetBitmap().Width()), image_idx);
      COLMAP_AMD_CHECK_MSG(image.GetHeight() == static_cast<size_t>(image.GetBitmap().Height()), image_idx);
      // Make sure, the calibration matrix only contains fx, fy, cx, cy.
      COLMAP_AMD_CHECK_MSG(std::abs(image.GetK()[1] - 0.0f) < 1e-6f, image_idx);
      COLMAP_AMD_CHECK_MSG(std::abs(image.GetK()[3] - 0.0f) < 1e-6f, image_idx);
      COLMAP_AMD_CHECK_MSG(std::abs(image.GetK()[6] - 0.0f) < 1e-6f, image_idx);
      COLMAP_AMD_CHECK_MSG(std::abs(image.GetK()[7] - 0.0f) < 1e-6f, image_idx);
      COLMAP_AMD_CHECK_MSG(std::abs(image.GetK()[8] - 1.0f) < 1e-6f, image_idx);
      if (options_.geom_consistency) {
        COLMAP_AMD_CHECK_MSG(image_idx < static_cast<int>(problem_.depth_maps->size()), image_idx);
        const DepthMap& depth_map = problem_.depth_maps->at(image_idx);
        COLMAP_AMD_CHECK_MSG(image.GetWidth() == depth_map.GetWidth(), image_idx);
        COLMAP_AMD_CHECK_MSG(image.GetHeight() == depth_map.GetHeight(), image_idx);
      }
    }
    if (options_.geom_consistency) {
      const Image& ref_image = problem_.images->at(problem_.ref_image_idx);
      const NormalMap& ref_normal_map = problem_.normal_maps->at(problem_.ref_image_idx);
      COLMAP_AMD_CHECK(ref_image.GetWidth() == ref_normal_map.GetWidth());
      COLMAP_AMD_CHECK(ref_image.GetHeight() == ref_normal_map.GetHeight());
    }
  }

  // patch_match.cc:128-135: Check(), construct the device object (upload), run all sweeps.
  void Run() {
    Check();
    pm_options c;
    pm_options_init(&c);
    c.depth_min = options_.depth_min;
    c.depth_max = options_.depth_max;
    c.sigma_spatial = options_.sigma_spatial;
    c.sigma_color = options_.sigma_color;
    c.ncc_sigma = options_.ncc_sigma;
    c.min_triangulation_angle = options_.min_triangulation_angle;
    c.incident_angle_sigma = options_.incident_angle_sigma;
    c.geom_consistency_regularizer = options_.geom_consistency_regularizer;
    c.geom_consistency_max_cost = options_.geom_consistency_max_cost;
    c.filter_min_ncc = options_.filter_min_ncc;
    c.filter_min_triangulation_angle = options_.filter_min_triangulation_angle;
    c.filter_geom_consistency_max_cost = options_.filter_geom_consistency_max_cost;
    c.window_radius = options_.window_radius;
    c.window_step = options_.window_step;
    c.num_samples = options_.num_samples;
    c.num_iterations = options_.num_iterations;
    c.filter_min_num_consistent = options_.filter_min_num_consistent;
    c.geom_consistency = options_.geom_consistency ? 1 : 0;
    c.filter = options_.filter ? 1 : 0;
    c.gpu_index = CSVToIntVector(options_.gpu_index)[0];

    std::vector<pm_image> images(problem_.images->size());
    std::memset(images.data(), 0, images.size() * sizeof(pm_image));
    std::set<int> used(problem_.src_image_idxs.begin(), problem_.src_image_idxs.end());
    used.insert(problem_.ref_image_idx);
    for (const int i : used) {
      const Image& im = problem_.images->at(i);
      pm_image& ci = images[i];
      ci.width = static_cast<int32_t>(im.GetWidth());
      ci.height = static_cast<int32_t>(im.GetHeight());
      std::memcpy(ci.K, im.GetK(), sizeof(ci.K));
      std::memcpy(ci.R, im.GetR(), sizeof(ci.R));
      std::memcpy(ci.T, im.GetT(), sizeof(ci.T));
      ci.gray = im.GetBitmap().RowMajorData().data();
      if (options_.geom_consistency) {
        ci.depth_map = problem_.depth_maps->at(i).GetPtr();
        // only the reference image must carry a normal map of the right size (:120-125)
        const NormalMap& nm = problem_.normal_maps->at(i);
        ci.normal_map = nm.GetWidth() == im.GetWidth() && nm.GetHeight() == im.GetHeight() ? nm.GetPtr() : nullptr;
      }
    }
    pm_problem cp;
    cp.ref_image_idx = problem_.ref_image_idx;
    cp.num_src_images = static_cast<int32_t>(problem_.src_image_idxs.size());
    cp.src_image_idxs = problem_.src_image_idxs.data();
    cp.num_images = static_cast<int32_t>(images.size());
    cp.images = images.data();
    if (handle_) {
      pm_destroy(handle_);
      handle_ = nullptr;
    }
    Device(pm_create(&c, &cp, &handle_));
    Device(pm_run(handle_));
  }

  // patch_match.cc:137-154
  DepthMap GetDepthMap() const {
    Mat<float> m(RefWidth(), RefHeight(), 1);
    Device(pm_get_depth_map(Handle(), m.GetPtr()));
    return DepthMap(m, static_cast<float>(options_.depth_min), static_cast<float>(options_.depth_max));
  }
  NormalMap GetNormalMap() const {
    Mat<float> m(RefWidth(), RefHeight(), 3);
    Device(pm_get_normal_map(Handle(), m.GetPtr()));
    return NormalMap(m);
  }
  Mat<float> GetSelProbMap() const {
    Mat<float> m(RefWidth(), RefHeight(), problem_.src_image_idxs.size());
    Device(pm_get_sel_prob_map(Handle(), m.GetPtr()));
    return m;
  }
  ConsistencyGraph GetConsistencyGraph() const {
    size_t n = 0;
    Device(pm_get_consistent_image_idxs(Handle(), nullptr, 0, &n));
    std::vector<int> idxs(n);
    Device(pm_get_consistent_image_idxs(Handle(), idxs.data(), n, &n));
    return ConsistencyGraph(RefWidth(), RefHeight(), std::move(idxs));
  }

 private:
  static void Device(int rc) {
    if (rc != 0) throw std::runtime_error(pm_last_error());
  }
  pm_handle* Handle() const {
    if (!handle_) throw std::logic_error("PatchMatch::Run() has not been called");
    return handle_;
  }
  size_t RefWidth() const { return problem_.images->at(problem_.ref_image_idx).GetWidth(); }
  size_t RefHeight() const { return problem_.images->at(problem_.ref_image_idx).GetHeight(); }

  const PatchMatchOptions options_;
  const Problem problem_;
  pm_handle* handle_ = nullptr;
};

// ---------------------------------------------------------------------------------------------
// Depth-map fusion (reference mvs/fusion.h:46-139, fusion.cc) on in-memory inputs. The workspace
// reading of StereoFusion::Run stays with the caller; the traversal runs on the GPU (fusion.hip) and equals the
// reference's algorithm run sequentially in a fixed pixel order (colmap_amd_fusion.h).
// ---------------------------------------------------------------------------------------------
struct StereoFusionOptions {  // fusion.h:46-94 (the fields that reach the traversal)
  int min_num_pixels = 5;
  int max_num_pixels = 10000;
  int max_traversal_depth = 100;
  double max_reproj_error = 2.0f;
  double max_depth_error = 0.01f;
  double max_normal_error = 10.0f;
  int check_num_images = 50;
  // fusion.h:53: the size of the reference's thread pool -- here the (deterministic) turn order: its threads take the
  // ten-row stripes t, t + T, ... in step; 1 = row-major = the reference with one thread; -1 = one thread per stripe
  int num_threads = -1;
  float bounding_box_min[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  float bounding_box_max[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};

  bool Check() const {  // fusion.cc:96-106
    return min_num_pixels >= 0 && min_num_pixels <= max_num_pixels && max_traversal_depth > 0 &&
           max_reproj_error >= 0 && max_depth_error >= 0 && max_normal_error >= 0 && check_num_images > 0;
  }
};

struct PlyPoint {  // util/ply.h:39-49
  float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0;
  uint8_t r = 0, g = 0, b = 0;
};

struct FusionInput {  // one workspace image: pose at the model size, colour bitmap, depth / normal maps
  const Image* image = nullptr;
  const uint8_t* rgb = nullptr;  // bitmap_height x bitmap_width x 3, or nullptr
  int bitmap_width = 0, bitmap_height = 0;
  const DepthMap* depth_map = nullptr;  // nullptr: the image is not used (fusion.cc:204-213)
  const NormalMap* normal_map = nullptr;
  const Mat<char>* mask = nullptr;      // depth-map sized, > 0 = pre-masked (fusion.cc:359-399)
};

class StereoFusion {
 public:
  explicit StereoFusion(const StereoFusionOptions& options) : options_(options) { COLMAP_AMD_CHECK(options_.Check()); }

  // overlapping_images[i]: Model::GetMaxOverlappingImages(check_num_images, 0) of image i
  void Run(const std::vector<FusionInput>& inputs, const std::vector<std::vector<int>>& overlapping_images) {
    COLMAP_AMD_CHECK(inputs.size() == overlapping_images.size());
    fusion_options o;
    fusion_options_init(&o);
    o.min_num_pixels = options_.min_num_pixels;
    o.max_num_pixels = options_.max_num_pixels;
    o.max_traversal_depth = options_.max_traversal_depth;
    o.check_num_images = options_.check_num_images;
    o.max_reproj_error = options_.max_reproj_error;
    o.max_depth_error = options_.max_depth_error;
    o.max_normal_error = options_.max_normal_error;
    o.num_threads = options_.num_threads;  // the size of the reference's pool = the turn order (colmap_amd_fusion.h)
    std::memcpy(o.bbox_min, options_.bounding_box_min, sizeof(o.bbox_min));
    std::memcpy(o.bbox_max, options_.bounding_box_max, sizeof(o.bbox_max));
    std::vector<fusion_image> images(inputs.size());
    std::memset(images.data(), 0, images.size() * sizeof(fusion_image));
    std::vector<std::vector<uint8_t>> masks(inputs.size());
    for (size_t i = 0; i < inputs.size(); ++i) {
      const FusionInput& in = inputs[i];
      COLMAP_AMD_CHECK(in.image != nullptr);
      fusion_image& f = images[i];
      f.width = static_cast<int32_t>(in.image->GetWidth());
      f.height = static_cast<int32_t>(in.image->GetHeight());
      std::memcpy(f.K, in.image->GetK(), sizeof(f.K));
      std::memcpy(f.R, in.image->GetR(), sizeof(f.R));
      std::memcpy(f.T, in.image->GetT(), sizeof(f.T));
      f.used = in.depth_map != nullptr && in.normal_map != nullptr;
      if (!f.used) continue;
      COLMAP_AMD_CHECK(in.depth_map->GetWidth() == in.normal_map->GetWidth());
      COLMAP_AMD_CHECK(in.depth_map->GetHeight() == in.normal_map->GetHeight());
      f.depth_map = in.depth_map->GetPtr();
      f.normal_map = in.normal_map->GetPtr();
      f.depth_width = static_cast<int32_t>(in.depth_map->GetWidth());
      f.depth_height = static_cast<int32_t>(in.depth_map->GetHeight());
      f.rgb = in.rgb;
      f.bitmap_width = in.bitmap_width;
      f.bitmap_height = in.bitmap_height;
      if (in.mask) {
        COLMAP_AMD_CHECK(in.mask->GetWidth() == in.depth_map->GetWidth());
        COLMAP_AMD_CHECK(in.mask->GetHeight() == in.depth_map->GetHeight());
        masks[i].resize(in.mask->GetData().size());
        for (size_t k = 0; k < masks[i].size(); ++k) masks[i][k] = in.mask->GetData()[k] > 0 ? 1 : 0;
        f.mask = masks[i].data();
      }
    }
    std::vector<int32_t> ptr(inputs.size() + 1, 0), idx;
    for (size_t i = 0; i < inputs.size(); ++i) {
      idx.insert(idx.end(), overlapping_images[i].begin(), overlapping_images[i].end());
      ptr[i + 1] = static_cast<int32_t>(idx.size());
    }
    fusion_result* res = nullptr;
    if (fusion_run(&o, static_cast<int32_t>(images.size()), images.data(), ptr.data(), idx.empty() ? nullptr : idx.data(),
                   &res) != 0)
      throw std::runtime_error(fusion_last_error());
    const size_t n = fusion_num_points(res);
    std::vector<float> xn(6 * n);
    std::vector<uint8_t> rgb(3 * n);
    fusion_get_points(res, xn.data(), rgb.data());
    size_t total = 0;
    fusion_get_visibility(res, nullptr, nullptr, &total);
    std::vector<int64_t> vptr(n + 1, 0);
    std::vector<int32_t> vidx(total + 1, 0);
    fusion_get_visibility(res, vptr.data(), vidx.data(), &total);
    fusion_free(res);
    fused_points_.assign(n, PlyPoint());
    fused_points_visibility_.assign(n, {});
    for (size_t k = 0; k < n; ++k) {
      PlyPoint& p = fused_points_[k];
      p.x = xn[6 * k]; p.y = xn[6 * k + 1]; p.z = xn[6 * k + 2];
      p.nx = xn[6 * k + 3]; p.ny = xn[6 * k + 4]; p.nz = xn[6 * k + 5];
      p.r = rgb[3 * k]; p.g = rgb[3 * k + 1]; p.b = rgb[3 * k + 2];
      fused_points_visibility_[k].assign(vidx.begin() + vptr[k], vidx.begin() + vptr[k + 1]);
    }
  }

  const std::vector<PlyPoint>& GetFusedPoints() const { return fused_points_; }
  const std::vector<std::vector<int>>& GetFusedPointsVisibility() const { return fused_points_visibility_; }

 private:
  const StereoFusionOptions options_;
  std::vector<PlyPoint> fused_points_;
  std::vector<std::vector<int>> fused_points_visibility_;
};

}  // namespace mvs
}  // namespace colmap_amd

#endif  // COLMAP_AMD_MVS_HPP_
