// TEST INFRASTRUCTURE (oracle/ref_shim/README.md) -- host glue around the reference's own
// PatchMatchCuda (compiled from /root/reference/src/colmap/mvs/patch_match_cuda.cu as it lies).
// Holds (1) the software texture's host side, (2) the host members the reference implements in
// files that need Eigen / OpenImageIO (mvs/image.cc, depth_map.cc, normal_map.cc, util/cudacc.cc,
// util/cuda.cc) restated without them, (3) the extern "C" entry points tests/ref_pm.py binds.
// No __global__ / __device__ code lives here: everything that runs on the GPU is the reference's.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

// The checker reads PatchMatchCuda's intermediate state (PRNG states, filtered reference image,
// initial costs), which the class keeps private.
#define private public
#include "colmap/mvs/patch_match_cuda.h"
#undef private
#include "colmap/util/cuda.h"
#include "colmap/util/cudacc.h"
#include "colmap/util/logging.h"

// ---------------------------------------------------------------------------------------------
// (1) software texture, host side
// ---------------------------------------------------------------------------------------------
namespace ref_shim {

hipError_t Malloc3DArray(hipArray_t* array, const hipChannelFormatDesc* desc, hipExtent extent, unsigned int) {
  auto* a = new SoftArray();
  a->width = extent.width;
  a->height = extent.height;
  a->depth = extent.depth == 0 ? 1 : extent.depth;
  a->elem_bytes = desc->x / 8;
  const hipError_t e = hipMalloc(&a->data, a->width * a->height * a->depth * a->elem_bytes);
  if (e != hipSuccess) {
    delete a;
    return e;
  }
  *array = reinterpret_cast<hipArray_t>(a);
  return hipSuccess;
}

hipError_t FreeArray(hipArray_t array) {
  auto* a = reinterpret_cast<SoftArray*>(array);
  const hipError_t e = hipFree(a->data);
  delete a;
  return e;
}

// The path only copies whole pitched blocks (host or device) into an array (cuda_texture.h:88-123).
hipError_t Memcpy3D(const hipMemcpy3DParms* p) {
  auto* a = reinterpret_cast<SoftArray*>(p->dstArray);
  if (!a || !p->srcPtr.ptr) return hipErrorInvalidValue;
  const size_t row_bytes = p->extent.width * a->elem_bytes;
  const size_t rows = p->extent.height * (p->extent.depth == 0 ? 1 : p->extent.depth);
  return hipMemcpy2D(a->data, a->width * a->elem_bytes, p->srcPtr.ptr, p->srcPtr.pitch, row_bytes, rows, p->kind);
}

hipError_t CreateTextureObject(hipTextureObject_t* tex, const hipResourceDesc* res, const hipTextureDesc* desc,
                               const void*) {
  // What a CDNA device answers for a linear-filtered layered texture according to the reference's
  // own notes (patch_match_cuda.cu:416-425, :1631-1645); the reference asks for point filtering on gfx9.
  if (desc->filterMode != hipFilterModePoint) return hipErrorNotSupported;
  if (desc->addressMode[0] != hipAddressModeBorder || desc->addressMode[1] != hipAddressModeBorder ||
      desc->normalizedCoords)
    return hipErrorNotSupported;
  auto* a = reinterpret_cast<SoftArray*>(res->res.array.array);
  SoftTexture t;
  t.data = a->data;
  t.width = static_cast<int>(a->width);
  t.height = static_cast<int>(a->height);
  t.depth = static_cast<int>(a->depth);
  t.elem_bytes = a->elem_bytes;
  t.normalized_float = desc->readMode == hipReadModeNormalizedFloat ? 1 : 0;
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, sizeof(SoftTexture));
  if (e != hipSuccess) return e;
  e = hipMemcpy(d, &t, sizeof(SoftTexture), hipMemcpyHostToDevice);
  *tex = reinterpret_cast<hipTextureObject_t>(d);
  return e;
}

hipError_t DestroyTextureObject(hipTextureObject_t tex) { return hipFree(reinterpret_cast<void*>(tex)); }

}  // namespace ref_shim

// ---------------------------------------------------------------------------------------------
// (2) host members of reference classes whose own translation units cannot be built here
// ---------------------------------------------------------------------------------------------
namespace colmap {

// util/cudacc.cc, util/cuda.cc
void CudaSafeCall(const cudaError_t error, const std::string& file, const int line) {
  if (error != cudaSuccess) {
    std::ostringstream s;
    s << "HIP error at " << file << ":" << line << " - " << cudaGetErrorString(error);
    throw std::runtime_error(s.str());
  }
}
void CudaCheck(const char* file, const int line) { CudaSafeCall(cudaGetLastError(), file, line); }
void CudaSyncAndCheck(const char* file, const int line) {
  CudaSafeCall(cudaDeviceSynchronize(), file, line);
  CudaSafeCall(cudaGetLastError(), file, line);
}
CudaTimer::CudaTimer() : elapsed_time_(0.0f) {
  (void)cudaEventCreate(&start_);
  (void)cudaEventCreate(&stop_);
  (void)cudaEventRecord(start_, 0);
}
CudaTimer::~CudaTimer() {
  (void)cudaEventDestroy(start_);
  (void)cudaEventDestroy(stop_);
}
void CudaTimer::Print(const std::string&) {}
int GetNumCudaDevices() {
  int n = 0;
  (void)cudaGetDeviceCount(&n);
  return n;
}
int FindBestCudaDevice() { return 0; }
void SetBestCudaDevice(const int gpu_index) { CudaSafeCall(cudaSetDevice(gpu_index < 0 ? 0 : gpu_index), __FILE__, __LINE__); }

namespace mvs {

// mvs/image.cc:97-150, in plain float (the same restatement as oracle/pm_oracle.c:206-270, which
// tests/test_pm_oracle.py pins against the reference's image_test.cc answers).
namespace {
void Mul33(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
void RefShimInverse4x4(const float m[16], float inv[16]) {
  // identical expression order to oracle/pm_oracle.c:230-256 so that both checkers feed the device
  // code bit-identical pose tables (the tables themselves are pinned against image_test.cc there)
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float inv_det = 1.0f / det;
  for (int i = 0; i < 12; ++i) inv[i] = inv[i] * inv_det;
}
}  // namespace

void ComputeRelativePose(const float R1[9], const float T1[3], const float R2[9], const float T2[3], float R[9],
                         float T[3]) {
  float R1t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R1t[3 * i + j] = R1[3 * j + i];
  Mul33(R2, R1t, R);
  for (int i = 0; i < 3; ++i) T[i] = T2[i] - (R[3 * i] * T1[0] + R[3 * i + 1] * T1[1] + R[3 * i + 2] * T1[2]);
}

void ComposeProjectionMatrix(const float K[9], const float R[9], const float T[3], float P[12]) {
  float RT[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) RT[4 * i + j] = R[3 * i + j];
    RT[4 * i + 3] = T[i];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) P[4 * i + j] = K[3 * i] * RT[j] + K[3 * i + 1] * RT[4 + j] + K[3 * i + 2] * RT[8 + j];
}

void ComposeInverseProjectionMatrix(const float K[9], const float R[9], const float T[3], float inv_P[12]) {
  // general 4x4 inverse of [P; 0 0 0 1] by cofactors, written as a loop over 3x3 minors
  float m[16];
  ComposeProjectionMatrix(K, R, T, m);
  m[12] = m[13] = m[14] = 0.0f;
  m[15] = 1.0f;
  float inv[16];
  RefShimInverse4x4(m, inv);
  std::memcpy(inv_P, inv, 12 * sizeof(float));
}

void ComputeProjectionCenter(const float R[9], const float T[3], float C[3]) {
  for (int i = 0; i < 3; ++i) C[i] = -(R[i] * T[0] + R[3 + i] * T[1] + R[6 + i] * T[2]);
}

void RotatePose(const float RR[9], float R[9], float T[3]) {
  float Rn[9], Tn[3];
  Mul33(RR, R, Rn);
  for (int i = 0; i < 3; ++i) Tn[i] = RR[3 * i] * T[0] + RR[3 * i + 1] * T[1] + RR[3 * i + 2] * T[2];
  std::memcpy(R, Rn, sizeof(Rn));
  std::memcpy(T, Tn, sizeof(Tn));
}

Image::Image() {}
Image::Image(const std::filesystem::path& path, const size_t width, const size_t height, const float* K,
             const float* R, const float* T)
    : path_(path), width_(width), height_(height) {
  std::memcpy(K_, K, 9 * sizeof(float));
  std::memcpy(R_, R, 9 * sizeof(float));
  std::memcpy(T_, T, 3 * sizeof(float));
  ComposeProjectionMatrix(K_, R_, T_, P_);
  ComposeInverseProjectionMatrix(K_, R_, T_, inv_P_);
}
void Image::SetBitmap(Bitmap bitmap) {
  THROW_CHECK_EQ(width_, static_cast<size_t>(bitmap.Width()));
  THROW_CHECK_EQ(height_, static_cast<size_t>(bitmap.Height()));
  bitmap_ = std::move(bitmap);
}

// mvs/depth_map.cc, normal_map.cc: the constructors only (the rest needs image/warp.h).
DepthMap::DepthMap() : DepthMap(0, 0, -1.0f, -1.0f) {}
DepthMap::DepthMap(const size_t width, const size_t height, const float depth_min, const float depth_max)
    : Mat<float>(width, height, 1), depth_min_(depth_min), depth_max_(depth_max) {}
DepthMap::DepthMap(const Mat<float>& mat, const float depth_min, const float depth_max)
    : Mat<float>(mat.GetWidth(), mat.GetHeight(), mat.GetDepth()), depth_min_(depth_min), depth_max_(depth_max) {
  THROW_CHECK_EQ(mat.GetDepth(), 1u);
  data_ = mat.GetData();
}
NormalMap::NormalMap() : Mat<float>(0, 0, 3) {}
NormalMap::NormalMap(const size_t width, const size_t height) : Mat<float>(width, height, 3) {}
NormalMap::NormalMap(const Mat<float>& mat) : Mat<float>(mat.GetWidth(), mat.GetHeight(), mat.GetDepth()) {
  THROW_CHECK_EQ(mat.GetDepth(), 3u);
  data_ = mat.GetData();
}

}  // namespace mvs
}  // namespace colmap

// ---------------------------------------------------------------------------------------------
// (3) C entry points (layouts = pmo_options / pmo_image of oracle/pm_oracle.c, so that
//     tests/ref_pm.py reuses oracle/pm_oracle.py's ctypes structs)
// ---------------------------------------------------------------------------------------------
extern "C" {

typedef struct {
  double depth_min, depth_max, sigma_spatial, sigma_color, ncc_sigma, min_triangulation_angle,
      incident_angle_sigma, geom_consistency_regularizer, geom_consistency_max_cost, filter_min_ncc,
      filter_min_triangulation_angle, filter_geom_consistency_max_cost;
  int window_radius, window_step, num_samples, num_iterations, filter_min_num_consistent, geom_consistency,
      filter;
  int max_sweeps, memoize, num_threads, order;  // oracle-only, ignored
} ref_pm_options;

typedef struct {
  int width, height;
  float K[9], R[9], T[3];
  const uint8_t* gray;
  const float* depth;   // height * width or NULL
  const float* normal;  // 3 * height * width slice-major or NULL
} ref_pm_image;

static std::string g_error;
const char* ref_pm_last_error(void) { return g_error.c_str(); }

struct ref_pm_handle {
  std::vector<colmap::mvs::Image> images;
  std::vector<colmap::mvs::DepthMap> depth_maps;
  std::vector<colmap::mvs::NormalMap> normal_maps;
  colmap::mvs::PatchMatchOptions options;
  colmap::mvs::PatchMatch::Problem problem;
  std::unique_ptr<colmap::mvs::PatchMatchCuda> pm;
};

// Builds the reference's PatchMatchCuda (constructor: reference-image filter, textures, pose
// tables, PRNG, random / given initial depth and normals). Returns NULL on error.
ref_pm_handle* ref_pm_create(const ref_pm_options* o, int n_images, const ref_pm_image* images, int ref_idx,
                             int n_src, const int* src_idxs) {
  try {
    auto h = std::make_unique<ref_pm_handle>();
    colmap::mvs::PatchMatchOptions& p = h->options;
    p.depth_min = o->depth_min;
    p.depth_max = o->depth_max;
    p.sigma_spatial = o->sigma_spatial;
    p.sigma_color = o->sigma_color;
    p.ncc_sigma = o->ncc_sigma;
    p.min_triangulation_angle = o->min_triangulation_angle;
    p.incident_angle_sigma = o->incident_angle_sigma;
    p.geom_consistency_regularizer = o->geom_consistency_regularizer;
    p.geom_consistency_max_cost = o->geom_consistency_max_cost;
    p.filter_min_ncc = o->filter_min_ncc;
    p.filter_min_triangulation_angle = o->filter_min_triangulation_angle;
    p.filter_geom_consistency_max_cost = o->filter_geom_consistency_max_cost;
    p.window_radius = o->window_radius;
    p.window_step = o->window_step;
    p.num_samples = o->num_samples;
    p.num_iterations = o->num_iterations;
    p.filter_min_num_consistent = o->filter_min_num_consistent;
    p.geom_consistency = o->geom_consistency != 0;
    p.filter = o->filter != 0;
    p.gpu_index = "0";
    for (int i = 0; i < n_images; ++i) {
      const ref_pm_image& im = images[i];
      h->images.emplace_back("", im.width, im.height, im.K, im.R, im.T);
      h->images.back().SetBitmap(colmap::Bitmap(im.width, im.height, im.gray));
      colmap::mvs::DepthMap d(im.width, im.height, static_cast<float>(o->depth_min), static_cast<float>(o->depth_max));
      colmap::mvs::NormalMap n(im.width, im.height);
      if (im.depth) std::memcpy(d.GetPtr(), im.depth, sizeof(float) * im.width * im.height);
      if (im.normal) std::memcpy(n.GetPtr(), im.normal, sizeof(float) * 3 * im.width * im.height);
      h->depth_maps.push_back(std::move(d));
      h->normal_maps.push_back(std::move(n));
    }
    h->problem.ref_image_idx = ref_idx;
    h->problem.src_image_idxs.assign(src_idxs, src_idxs + n_src);
    h->problem.images = &h->images;
    h->problem.depth_maps = &h->depth_maps;
    h->problem.normal_maps = &h->normal_maps;
    h->pm = std::make_unique<colmap::mvs::PatchMatchCuda>(h->options, h->problem);
    return h.release();
  } catch (const std::exception& e) {
    g_error = e.what();
    return nullptr;
  }
}

void ref_pm_destroy(ref_pm_handle* h) { delete h; }

// State the constructor left on the device. rng: H*W*6 uint32 per pixel {x[0..4], d};
// ref_image: H*W bytes as re-quantised by FilterKernel; sum / sqsum: H*W floats; depth H*W;
// normal 3*H*W slice-major. Any pointer may be NULL.
int ref_pm_get_state(ref_pm_handle* h, uint32_t* rng, uint8_t* ref_image, float* sum, float* sqsum, float* depth,
                     float* normal) {
  try {
    colmap::mvs::PatchMatchCuda& pm = *h->pm;
    const size_t W = pm.depth_map_->GetWidth(), H = pm.depth_map_->GetHeight();
    if (rng) {
      static_assert(sizeof(curandState) == 40, "rocRAND XORWOW state: double, float, d, x[5]");
      std::vector<curandState> st(W * H);
      pm.rand_state_map_->CopyToHost(st.data(), W * sizeof(curandState));
      for (size_t i = 0; i < W * H; ++i) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&st[i]);
        for (int k = 0; k < 5; ++k) rng[6 * i + k] = w[4 + k];  // x[0..4] at byte 16
        rng[6 * i + 5] = w[3];                                  // d at byte 12
      }
    }
    if (ref_image) pm.ref_image_->image->CopyToHost(ref_image, W);
    if (sum) pm.ref_image_->sum_image->CopyToHost(sum, W * sizeof(float));
    if (sqsum) pm.ref_image_->squared_sum_image->CopyToHost(sqsum, W * sizeof(float));
    if (depth) pm.depth_map_->CopyToHost(depth, W * sizeof(float));
    if (normal) pm.normal_map_->CopyToHost(normal, W * sizeof(float));
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
}

// PatchMatchCuda::Run() and its getters. cost: S*H*W (cost_map_ after the last sweep; with
// num_iterations == 0 these are ComputeInitialCost's values). mask: S*H*W consistency mask.
int ref_pm_run(ref_pm_handle* h, float* depth, float* normal, float* sel_prob, uint8_t* mask, float* cost) {
  try {
    colmap::mvs::PatchMatchCuda& pm = *h->pm;
    pm.Run();
    const auto d = pm.GetDepthMap();
    const auto n = pm.GetNormalMap();
    if (depth) std::memcpy(depth, d.GetPtr(), d.GetNumBytes());
    if (normal) std::memcpy(normal, n.GetPtr(), n.GetNumBytes());
    if (sel_prob) {
      const auto s = pm.GetSelProbMap();
      std::memcpy(sel_prob, s.GetPtr(), s.GetNumBytes());
    }
    if (mask && pm.consistency_mask_->GetWidth() > 0) {
      const auto m = pm.consistency_mask_->CopyToMat();
      std::memcpy(mask, m.GetPtr(), m.GetNumBytes());
    }
    if (cost) {
      const auto c = pm.cost_map_->CopyToMat();
      std::memcpy(cost, c.GetPtr(), c.GetNumBytes());
    }
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
}

}  // extern "C"
