// pm_api.cpp -- host side of the PatchMatch C ABI (include/colmap_amd_pm.h).
//
// Mirrors the host half of the reference's PatchMatchCuda
// (src/colmap/mvs/patch_match_cuda.cu:1290-1939): option/problem validation,
// uploads, pose tables for the four sweep directions, the sweep schedule. The
// device half lives in pm_kernels.hip. Compiled with hipcc; no CPU fallback:
// every entry point fails with an error if there is no usable HIP device.
#include "../../include/colmap_amd_pm.h"
#include "pm_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <tuple>
#include <stdexcept>
#include <string>
#include <vector>

using namespace colmap_amd;

namespace {

thread_local std::string g_last_error;

struct PmError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define PM_CHECK(cond, msg)                                              \
  do {                                                                   \
    if (!(cond)) throw PmError(std::string("Check failed: ") + #cond + " " + (msg)); \
  } while (0)

#define HIP_CALL(expr)                                                                  \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      throw PmError(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)

// Device buffers of destroyed handles are kept for the next handle of the same shape (a controller or a
// benchmark creates and destroys ~10 buffers of up to 1.3 GB per reference image: hipMalloc / hipFree
// would map, clear and unmap those pages every time and hipFree synchronises the device). Exact-size
// free lists per device; bounded by 64 GB unless pm_set_cached_memory_limit says otherwise; pm_release_cached_memory()
// returns everything to the driver. A buffer only comes back here after pm_destroy synchronised the
// handle's own stream AND the stream its last (batched) run was enqueued on, so the next user cannot
// race with the previous one. The cap is additionally bounded by a quarter of the device's memory,
// and a failed hipMalloc releases the pool and retries once (cached memory must never cause an OOM).
class DevPool {
 public:
  static DevPool& Get() {
    static DevPool* pool = new DevPool();  // never destroyed: no hipFree after the runtime is gone
    return *pool;
  }
  void* Take(int dev, size_t bytes) {
    std::lock_guard<std::mutex> lock(mu_);
    auto it = free_.find({dev, bytes});
    if (it == free_.end() || it->second.empty()) return nullptr;
    void* p = it->second.back();
    it->second.pop_back();
    pooled_ -= bytes;
    return p;
  }
  void Give(int dev, size_t bytes, void* p) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (pooled_ + bytes <= cap_) {
        free_[{dev, bytes}].push_back(p);
        pooled_ += bytes;
        return;
      }
    }
    (void)hipFree(p);
  }
  // pm_set_cached_memory_limit: bytes the free lists may hold from now on (what is pooled beyond it goes back at once)
  void SetCap(size_t bytes) {
    bool shrink;
    {
      std::lock_guard<std::mutex> lock(mu_);
      cap_ = bytes;
      shrink = pooled_ > cap_;
    }
    if (shrink) Release();
  }
  void Release() {
    std::lock_guard<std::mutex> lock(mu_);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : free_) {
      (void)hipSetDevice(kv.first.first);
      for (void* p : kv.second) (void)hipFree(p);
    }
    (void)hipSetDevice(cur);
    free_.clear();
    pooled_ = 0;
  }

 private:
  DevPool() {
    cap_ = 64ull << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) cap_ = std::min(cap_, total_b / 4);
  }
  std::mutex mu_;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_;
  size_t pooled_ = 0, cap_ = 0;
};

template <typename T>
struct DevBuf {
  T* ptr = nullptr;
  size_t count = 0;
  int dev = -1;
  void alloc(size_t n) {
    free();
    if (n == 0) return;
    HIP_CALL(hipGetDevice(&dev));
    void* p = DevPool::Get().Take(dev, n * sizeof(T));
    if (!p) {
      hipError_t e = hipMalloc(&p, n * sizeof(T));
      if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        DevPool::Get().Release();
        e = hipMalloc(&p, n * sizeof(T));
      }
      HIP_CALL(e);
    }
    ptr = static_cast<T*>(p);
    count = n;
  }
  void free() {
    if (ptr) DevPool::Get().Give(dev, count * sizeof(T), ptr);
    ptr = nullptr;
    count = 0;
  }
  ~DevBuf() { free(); }
};

// ---- host pose helpers (reference mvs/image.cc:97-150), float like the reference ----
void Mat33Mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

void ComputeRelativePose(const float R1[9], const float T1[3], const float R2[9], const float T2[3],
                         float R[9], float T[3]) {
  float R1t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R1t[3 * i + j] = R1[3 * j + i];
  Mat33Mul(R2, R1t, R);
  for (int i = 0; i < 3; ++i)
    T[i] = T2[i] - (R[3 * i] * T1[0] + R[3 * i + 1] * T1[1] + R[3 * i + 2] * T1[2]);
}

void ComposeProjectionMatrix(const float K[9], const float R[9], const float T[3], float P[12]) {
  float RT[12];
  for (int i = 0; i < 3; ++i) {
    RT[4 * i] = R[3 * i];
    RT[4 * i + 1] = R[3 * i + 1];
    RT[4 * i + 2] = R[3 * i + 2];
    RT[4 * i + 3] = T[i];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j)
      P[4 * i + j] = K[3 * i] * RT[j] + K[3 * i + 1] * RT[4 + j] + K[3 * i + 2] * RT[8 + j];
}

void ComposeInverseProjectionMatrix(const float K[9], const float R[9], const float T[3],
                                    float inv_P[12]) {
  float m[16];
  ComposeProjectionMatrix(K, R, T, m);
  m[12] = m[13] = m[14] = 0.0f;
  m[15] = 1.0f;
  // explicit cofactor table (general 4x4 inverse), term order fixed so that the
  // result is a deterministic function of m
  float inv[16];
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
  for (int i = 0; i < 12; ++i) inv_P[i] = inv[i] * inv_det;
}

void ComputeProjectionCenter(const float R[9], const float T[3], float C[3]) {
  for (int i = 0; i < 3; ++i) C[i] = -(R[i] * T[0] + R[3 + i] * T[1] + R[6 + i] * T[2]);
}

void RotatePose(const float RR[9], float R[9], float T[3]) {
  float Rn[9], Tn[3];
  Mat33Mul(RR, R, Rn);
  for (int i = 0; i < 3; ++i) Tn[i] = RR[3 * i] * T[0] + RR[3 * i + 1] * T[1] + RR[3 * i + 2] * T[2];
  std::memcpy(R, Rn, sizeof(Rn));
  std::memcpy(T, Tn, sizeof(Tn));
}

void CheckOptions(const pm_options& o) {
  // PatchMatchOptions::Check, reference mvs/patch_match_options.cc:73-100
  if (o.depth_min != -1.0f || o.depth_max != -1.0f) {
    PM_CHECK(o.depth_min <= o.depth_max, "depth_min <= depth_max");
    PM_CHECK(o.depth_min >= 0.0, "depth_min >= 0");
  }
  PM_CHECK(o.window_radius <= 32, "window_radius <= kMaxPatchMatchWindowRadius");
  PM_CHECK(o.sigma_color > 0.0, "");
  PM_CHECK(o.window_radius > 0, "");
  PM_CHECK(o.window_step > 0, "");
  PM_CHECK(o.window_step <= 2, "");
  PM_CHECK(o.num_samples > 0, "");
  PM_CHECK(o.ncc_sigma > 0.0, "");
  PM_CHECK(o.min_triangulation_angle >= 0.0, "");
  PM_CHECK(o.min_triangulation_angle < 180.0, "");
  PM_CHECK(o.incident_angle_sigma > 0.0, "");
  PM_CHECK(o.num_iterations > 0, "");
  PM_CHECK(o.geom_consistency_regularizer >= 0.0, "");
  PM_CHECK(o.geom_consistency_max_cost >= 0.0, "");
  PM_CHECK(o.filter_min_ncc >= -1.0, "");
  PM_CHECK(o.filter_min_ncc <= 1.0, "");
  PM_CHECK(o.filter_min_triangulation_angle >= 0.0, "");
  PM_CHECK(o.filter_min_triangulation_angle <= 180.0, "");
  PM_CHECK(o.filter_min_num_consistent >= 0, "");
  PM_CHECK(o.filter_geom_consistency_max_cost >= 0.0, "");
  // the reference's kernel dispatch only instantiates radius 1..20 (patch_match_cuda.cu:1313-1337)
  PM_CHECK(o.window_radius <= 20, "window size not supported (reference instantiates radius 1..20)");
  PM_CHECK(o.sigma_spatial > 0.0,
           "sigma_spatial must be resolved by the caller (PatchMatchController sets it to "
           "window_radius, patch_match.cc:436-438)");
  PM_CHECK(o.depth_min > 0.0 && o.depth_max > 0.0,
           "depth range must be set (PatchMatchController::ProcessProblem, patch_match.cc:425-434)");
}

void CheckProblem(const pm_options& o, const pm_problem& p) {
  // PatchMatch::Check, reference mvs/patch_match.cc:67-126
  PM_CHECK(o.gpu_index >= -1, "gpu_index >= -1");
  PM_CHECK(p.images != nullptr, "problem.images");
  PM_CHECK(p.num_src_images > 0, "src_image_idxs.size() > 0");
  PM_CHECK(p.src_image_idxs != nullptr, "src_image_idxs");
  std::set<int> unique(p.src_image_idxs, p.src_image_idxs + p.num_src_images);
  unique.insert(p.ref_image_idx);
  PM_CHECK((int)unique.size() == p.num_src_images + 1,
           "duplicate source images or reference image used as source");
  for (int idx : unique) {
    PM_CHECK(idx >= 0, "image_idx >= 0");
    PM_CHECK(idx < p.num_images, "image_idx < images.size()");
    const pm_image& im = p.images[idx];
    PM_CHECK(im.width > 0 && im.height > 0, "bitmap size");
    PM_CHECK(im.gray != nullptr, "grey bitmap");
    PM_CHECK(std::abs(im.K[1] - 0.0f) < 1e-6f, "K[1]");
    PM_CHECK(std::abs(im.K[3] - 0.0f) < 1e-6f, "K[3]");
    PM_CHECK(std::abs(im.K[6] - 0.0f) < 1e-6f, "K[6]");
    PM_CHECK(std::abs(im.K[7] - 0.0f) < 1e-6f, "K[7]");
    PM_CHECK(std::abs(im.K[8] - 1.0f) < 1e-6f, "K[8]");
    if (o.geom_consistency) PM_CHECK(im.depth_map != nullptr, "depth map for geom_consistency");
  }
  if (o.geom_consistency) {
    PM_CHECK(p.images[p.ref_image_idx].normal_map != nullptr, "reference normal map");
    PM_CHECK(p.images[p.ref_image_idx].depth_map != nullptr, "reference depth map");
  }
}

}  // namespace

// Packed source images come out of slabs of equally sized slots (one hipMalloc per slab, at most 3.5 GB):
// the images of a problem are then neighbours in the address space, which is what lets the 11 x 11 sweep kernels
// address all of them through ONE buffer resource (pm_kernels.hip: fp_resource -- every image must end within 4 GB
// of the lowest one). A problem whose sources straddle two slabs that lie further apart takes the explicit-index
// build of the kernels instead; nothing else depends on the slabs. Slots return to their slab when the last
// problem (or the image cache) drops the image; pm_release_cached_memory() frees the slabs that are empty.
// Test hook (pm_debug_set_image_slab_limits): slab size and the span one buffer resource may cover. The defaults are the
// hardware's (3.5 GB slabs, 32-bit buffer offsets); the tests shrink both to exercise the re-homing of shared images
// with a few small images instead of > 4 GB of them.
static size_t g_fp_slab_slots = 0;                 // 0: by image size (below)
static std::atomic<unsigned long long> g_fp_rehomed{0};
static inline uint64_t FpSpanLimit(size_t image_bytes) {
  return g_fp_slab_slots ? (uint64_t)g_fp_slab_slots * image_bytes + 4097 : (1ull << 32);
}

class FpSlabPool {
 public:
  static FpSlabPool& Get() {
    static FpSlabPool* pool = new FpSlabPool();  // leaked on purpose, like DevPool
    return *pool;
  }
  uint32_t* Take(size_t bytes) {
    int dev = 0;
    HIP_CALL(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu_);
    for (auto it = slabs_.rbegin(); it != slabs_.rend(); ++it)
      if (it->dev == dev && it->slot_bytes == bytes && !it->free_slots.empty()) {
        const int k = it->free_slots.back();
        it->free_slots.pop_back();
        return reinterpret_cast<uint32_t*>(it->base + (size_t)k * bytes);
      }
    Slab sl;
    sl.dev = dev;
    sl.slot_bytes = bytes;
    // slab size: 3.5 GB for full-size images (178 slots at 2560 x 1920), 256 MB for small ones
    const size_t target = bytes >= (4u << 20) ? (size_t)(3.5 * (1ull << 30)) : (256u << 20);
    int n = (int)std::max<size_t>(1, std::min<size_t>(4096, target / bytes));
    // test mode: n slots per slab and as much unused memory behind them, so that two slabs never fit one span
    const size_t pad = g_fp_slab_slots ? 2 : 1;
    if (g_fp_slab_slots) n = (int)g_fp_slab_slots;
    void* p = nullptr;
    hipError_t e = hipErrorOutOfMemory;
    for (; n >= 1; n /= 2) {  // a slab that does not fit any more shrinks down to a single slot
      e = hipMalloc(&p, pad * n * bytes);
      if (e == hipSuccess) break;
      (void)hipGetLastError();
      if (n == 1) break;
    }
    if (e == hipErrorOutOfMemory) {
      DevPool::Get().Release();
      n = 1;
      e = hipMalloc(&p, bytes);
    }
    HIP_CALL(e);
    sl.base = static_cast<char*>(p);
    sl.nslots = n;
    for (int k = n - 1; k >= 1; --k) sl.free_slots.push_back(k);
    slabs_.push_back(sl);
    return reinterpret_cast<uint32_t*>(sl.base);
  }
  // The slab (identified by its base address) that holds `ptr`; null for a pointer the pool does not own.
  const char* SlabOf(const uint32_t* ptr) {
    std::lock_guard<std::mutex> lock(mu_);
    const char* c = reinterpret_cast<const char*>(ptr);
    for (auto& sl : slabs_)
      if (c >= sl.base && c < sl.base + (size_t)sl.nslots * sl.slot_bytes) return sl.base;
    return nullptr;
  }
  // A free slot of the given slab, or null when it has none (or its slots have another size).
  uint32_t* TakeFrom(const char* slab, size_t bytes) {
    std::lock_guard<std::mutex> lock(mu_);
    for (auto& sl : slabs_)
      if (sl.base == slab && sl.slot_bytes == bytes && !sl.free_slots.empty()) {
        const int k = sl.free_slots.back();
        sl.free_slots.pop_back();
        return reinterpret_cast<uint32_t*>(sl.base + (size_t)k * bytes);
      }
    return nullptr;
  }
  // Free slots of the newest slab of this size on the current device (0: none).
  int FreeInNewest(size_t bytes, const char** slab) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu_);
    for (auto it = slabs_.rbegin(); it != slabs_.rend(); ++it)
      if (it->dev == dev && it->slot_bytes == bytes) {
        *slab = it->base;
        return (int)it->free_slots.size();
      }
    return 0;
  }
  void Give(uint32_t* ptr) {
    std::lock_guard<std::mutex> lock(mu_);
    char* c = reinterpret_cast<char*>(ptr);
    for (auto& sl : slabs_)
      if (c >= sl.base && c < sl.base + (size_t)sl.nslots * sl.slot_bytes) {
        sl.free_slots.push_back((int)((c - sl.base) / sl.slot_bytes));
        return;
      }
  }
  void Release() {  // frees the slabs none of whose slots is in use
    std::lock_guard<std::mutex> lock(mu_);
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t keep = 0;
    for (size_t i = 0; i < slabs_.size(); ++i) {
      if ((int)slabs_[i].free_slots.size() == slabs_[i].nslots) {
        (void)hipSetDevice(slabs_[i].dev);
        (void)hipFree(slabs_[i].base);
      } else {
        if (keep != i) slabs_[keep] = std::move(slabs_[i]);
        ++keep;
      }
    }
    slabs_.resize(keep);
    (void)hipSetDevice(cur);
  }

 private:
  struct Slab {
    char* base = nullptr;
    size_t slot_bytes = 0;
    int nslots = 0, dev = 0;
    std::vector<int> free_slots;
  };
  std::mutex mu_;
  std::vector<Slab> slabs_;
};

struct FpBuf {
  uint32_t* ptr = nullptr;
  size_t count = 0;
  void alloc(size_t n) {
    ptr = FpSlabPool::Get().Take(n * sizeof(uint32_t));
    count = n;
  }
  void adopt(uint32_t* p, size_t n) {  // a slot taken from the pool by the caller
    ptr = p;
    count = n;
  }
  ~FpBuf() {
    if (ptr) FpSlabPool::Get().Give(ptr);
  }
};

// A packed source image (2x2 footprints + zero ring) in HBM. Shared between problems through
// pm_image_cache: neighbouring reference images use mostly the same sources (28 distinct images
// for 8 consecutive references with S = 20), so a batch gathers from one copy instead of eight.
struct FpEntry {
  FpBuf data;
};

struct pm_image_cache {
  int device = 0;
  std::mutex mu;
  // key: (caller's bitmap pointer, image width, image height, slot width, slot height)
  using Key = std::tuple<const void*, int, int, int, int>;
  std::map<Key, std::shared_ptr<FpEntry>> entries;
  std::vector<Key> order;  // insertion order, for eviction
  size_t hits = 0, misses = 0;
  size_t bytes = 0, capacity = ~(size_t)0;

  // Drops the oldest entries no live problem references until the cache fits its capacity.
  void Trim() {
    size_t keep = 0;
    for (size_t i = 0; i < order.size(); ++i) {
      auto it = entries.find(order[i]);
      if (bytes > capacity && it != entries.end() && it->second.use_count() == 1) {
        bytes -= it->second->data.count * sizeof(uint32_t);
        entries.erase(it);
      } else {
        order[keep++] = order[i];
      }
    }
    order.resize(keep);
  }
};

// Problems alive per device (created, not yet destroyed): what a launch can expect to share the GPU with
// (RunBatchAsync: columns per wave by occupancy).
static std::atomic<int> g_live_handles[16];

struct pm_handle {
  pm_options opt;
  int device = 0;
  hipStream_t stream = nullptr;
  int W = 0, H = 0, S = 0, src_w = 0, src_h = 0;
  std::vector<int> src_idxs;
  // host pose tables
  std::vector<float> poses_host;  // [4][S][43]
  float ref_K[4][4], ref_inv_K[4][4];
  // device buffers
  DevBuf<float> rec;
  std::vector<std::shared_ptr<FpEntry>> src_fp;  // per source image (possibly shared)
  DevBuf<const uint32_t*> src_fp_tab;
  DevBuf<float> src_depth;
  DevBuf<uint8_t> ref_img;
  DevBuf<float> ref_sum, ref_sqsum;
  DevBuf<uint32_t> rng;
  DevBuf<float> draws;  // the random numbers of one sweep per pixel (pm_draw_kernel), shapes served by the 11 x 11 kernel
  DevBuf<uint8_t> mask;
  DevBuf<float> poses;  // [4][S][43]
  DevBuf<float> out_depth, out_normal, out_sel, out_cost;
  DevBuf<unsigned long long> prof;
  DevBuf<unsigned long long> trace;  // progress trace of the last sweep launch (debug)
  DevBuf<uint32_t> src_fp_off;       // [S] the packed source images as slots of the problem's buffer resource
  const uint32_t* fp_base = nullptr; // its base (lowest address among them); null: the images lie too far apart
  const char* sweep_kernel = "";     // kernel of the last sweep launch (pm_get_sweep_kernel_name)
  DevBuf<unsigned long long> evals;  // NCC evaluations of the sweep launches of the last run
  DevBuf<PmParams> plan;  // per-launch parameter blocks of the last (batched) run
  hipStream_t run_stream = nullptr;  // stream the last run was enqueued on
  PmParams base;
  int threads = 64;
  int sweeps_done = 0;
  int final_sel_off = 0;
  bool ran = false;
  std::vector<hipEvent_t> ev;
  double sweep_ms = 0.0;
  int sweep_launches = 0;
  int launch_images = 1;       // reference images per sweep launch of the last run (its sub-batch)
  int launch_concurrency = 1;  // sub-batches of that run in flight together (RunBatchSplitAsync)

  bool counted = false;  // in g_live_handles
  ~pm_handle() {
    if (counted) --g_live_handles[device & 15];
    for (auto e : ev) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

void BuildPoseTables(pm_handle* h, const pm_problem& prob) {
  // InitTransforms, reference patch_match_cuda.cu:1694-1808
  const pm_image& ref = prob.images[prob.ref_image_idx];
  for (int i = 0; i < 4; ++i) {
    h->ref_K[i][0] = ref.K[0];
    h->ref_K[i][1] = ref.K[2];
    h->ref_K[i][2] = ref.K[4];
    h->ref_K[i][3] = ref.K[5];
  }
  std::swap(h->ref_K[1][0], h->ref_K[1][2]);
  std::swap(h->ref_K[1][1], h->ref_K[1][3]);
  h->ref_K[1][3] = h->W - 1 - h->ref_K[1][3];
  h->ref_K[2][1] = h->W - 1 - h->ref_K[2][1];
  h->ref_K[2][3] = h->H - 1 - h->ref_K[2][3];
  std::swap(h->ref_K[3][0], h->ref_K[3][2]);
  std::swap(h->ref_K[3][1], h->ref_K[3][3]);
  h->ref_K[3][1] = h->H - 1 - h->ref_K[3][1];
  for (int i = 0; i < 4; ++i) {
    h->ref_inv_K[i][0] = 1.0f / h->ref_K[i][0];
    h->ref_inv_K[i][1] = -h->ref_K[i][1] / h->ref_K[i][0];
    h->ref_inv_K[i][2] = 1.0f / h->ref_K[i][2];
    h->ref_inv_K[i][3] = -h->ref_K[i][3] / h->ref_K[i][2];
  }
  float rotated_R[9], rotated_T[3];
  std::memcpy(rotated_R, ref.R, sizeof(rotated_R));
  std::memcpy(rotated_T, ref.T, sizeof(rotated_T));
  const float R_z90[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  h->poses_host.assign((size_t)4 * h->S * kPoseStride, 0.0f);
  for (int i = 0; i < 4; ++i) {
    for (int s = 0; s < h->S; ++s) {
      const pm_image& im = prob.images[h->src_idxs[s]];
      float* p = h->poses_host.data() + ((size_t)i * h->S + s) * kPoseStride;
      p[0] = im.K[0]; p[1] = im.K[2]; p[2] = im.K[4]; p[3] = im.K[5];
      float rel_R[9], rel_T[3];
      ComputeRelativePose(rotated_R, rotated_T, im.R, im.T, rel_R, rel_T);
      std::memcpy(p + 4, rel_R, sizeof(rel_R));
      std::memcpy(p + 13, rel_T, sizeof(rel_T));
      ComputeProjectionCenter(rel_R, rel_T, p + 16);
      ComposeProjectionMatrix(im.K, rel_R, rel_T, p + 19);
      ComposeInverseProjectionMatrix(im.K, rel_R, rel_T, p + 31);
    }
    RotatePose(R_z90, rotated_R, rotated_T);
  }
}

PmParams ParamsForSweep(const pm_handle* h, int rot) {
  PmParams p = h->base;
  p.rot = rot;
  for (int k = 0; k < 4; ++k) {
    p.refK[k] = h->ref_K[rot][k];
    p.refInvK[k] = h->ref_inv_K[rot][k];
  }
  p.poses = h->poses.ptr + (size_t)rot * h->S * kPoseStride;
  return p;
}

// True when the packed images of `tab` can be read through one buffer resource (see Create).
bool FpSpanFits(const std::vector<const uint32_t*>& tab, size_t fp_count) {
  const uint32_t* lo = tab[0];
  for (auto p : tab) lo = std::min(lo, p);
  bool ok = ((uintptr_t)lo % 256) == 0;
  for (auto p : tab) {
    const uint64_t d = (uint64_t)((const char*)p - (const char*)lo);
    ok = ok && d % 256 == 0 && d + fp_count * sizeof(uint32_t) + 4096 < FpSpanLimit(fp_count * sizeof(uint32_t));
  }
  return ok;
}

void RehomeSourceImages(pm_handle* h, pm_image_cache* cache, const pm_problem& prob, size_t fp_count,
                        std::vector<const uint32_t*>* tab) {
  FpSlabPool& pool = FpSlabPool::Get();
  const size_t bytes = fp_count * sizeof(uint32_t);
  const int S = h->S;
  std::vector<const char*> slab(S);
  for (int s = 0; s < S; ++s) slab[s] = pool.SlabOf((*tab)[s]);
  // candidates, best first: the slab with most of the problem's images, then the newest slab
  std::map<const char*, int> votes;
  for (int s = 0; s < S; ++s)
    if (slab[s]) ++votes[slab[s]];
  std::vector<std::pair<int, const char*>> cand;
  for (auto& v : votes) cand.push_back({v.second, v.first});
  std::sort(cand.begin(), cand.end(), [](auto& a, auto& b) { return a.first > b.first; });
  const char* newest = nullptr;
  if (pool.FreeInNewest(bytes, &newest) > 0 && !votes.count(newest)) cand.push_back({0, newest});
  auto try_slab = [&](const char* target) -> bool {
    std::vector<std::pair<int, uint32_t*>> moved;
    for (int s = 0; s < S; ++s) {
      if (slab[s] == target) continue;
      uint32_t* p = pool.TakeFrom(target, bytes);
      if (!p) {  // not enough room in this slab: hand the slots back
        for (auto& m : moved) pool.Give(m.second);
        return false;
      }
      moved.push_back({s, p});
    }
    // all copies first, one synchronisation, THEN the new entries become visible (to this handle and, through the
    // cache, to other threads' Create calls, which launch on their own streams): nobody can pick up a half-written
    // image. If a copy fails the taken slots go back to the pool and nothing has changed.
    try {
      for (auto& m : moved)
        HIP_CALL(hipMemcpyAsync(m.second, (*tab)[m.first], bytes, hipMemcpyDeviceToDevice, h->stream));
      HIP_CALL(hipStreamSynchronize(h->stream));
    } catch (...) {
      for (auto& m : moved) pool.Give(m.second);
      throw;
    }
    for (auto& m : moved) {
      const int s = m.first;
      auto e = std::make_shared<FpEntry>();
      e->data.adopt(m.second, fp_count);
      if (cache) {
        const pm_image& im = prob.images[h->src_idxs[s]];
        const auto key = std::make_tuple((const void*)im.gray, im.width, im.height, h->src_w, h->src_h);
        std::lock_guard<std::mutex> lock(cache->mu);
        auto it = cache->entries.find(key);
        if (it != cache->entries.end() && it->second == h->src_fp[s]) it->second = e;
      }
      h->src_fp[s] = e;   // (the old copy is released here at the earliest: the copies above are complete)
      (*tab)[s] = m.second;
      ++g_fp_rehomed;
    }
    return true;
  };
  for (auto& c : cand)
    if (try_slab(c.second)) return;
  // every slab the problem touches is full: a fresh one (Take opens it when no slab has a free slot)
  uint32_t* probe = pool.Take(bytes);
  const char* fresh = pool.SlabOf(probe);
  pool.Give(probe);
  if (fresh && try_slab(fresh)) return;
  // no slab has room for all of them: the problem takes the explicit-index build of the kernels
}

void Create(const pm_options& opt_in, const pm_problem& prob, pm_image_cache* cache, pm_handle* h) {
  CheckOptions(opt_in);
  CheckProblem(opt_in, prob);
  h->opt = opt_in;
  const pm_options& opt = h->opt;

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    throw PmError("no HIP device available: the MI355X PatchMatch path has no CPU fallback");
  if (opt.gpu_index >= 0) {
    PM_CHECK(opt.gpu_index < ndev, "gpu_index < device count");
    h->device = opt.gpu_index;
  } else {
    HIP_CALL(hipGetDevice(&h->device));
  }
  HIP_CALL(hipSetDevice(h->device));
  if (cache) {
    // a cache holds device pointers of one GPU: bind it on first use, refuse another device later
    std::lock_guard<std::mutex> lock(cache->mu);
    if (cache->device < 0) cache->device = h->device;
    PM_CHECK(cache->device == h->device, "image cache and problem are on the same GPU");
  }
  HIP_CALL(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  ++g_live_handles[h->device & 15];
  h->counted = true;

  const pm_image& ref = prob.images[prob.ref_image_idx];
  h->W = ref.width;
  h->H = ref.height;
  h->S = prob.num_src_images;
  h->src_idxs.assign(prob.src_image_idxs, prob.src_image_idxs + prob.num_src_images);
  const int W = h->W, H = h->H, S = h->S;
  const hipMemcpyKind in_kind = opt.inputs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;

  // ---- InitSourceImages (reference patch_match_cuda.cu:1595-1692) ------------------
  for (int s = 0; s < S; ++s) {
    const pm_image& im = prob.images[h->src_idxs[s]];
    h->src_w = std::max(h->src_w, im.width);
    h->src_h = std::max(h->src_h, im.height);
  }
  const size_t slot = (size_t)h->src_w * h->src_h;
  {
    DevBuf<uint8_t> staging;
    const size_t fp_count = pm_fp_entries(h->src_w, h->src_h);
    std::vector<const uint32_t*> tab(S);
    h->src_fp.resize(S);
    for (int s = 0; s < S; ++s) {
      const pm_image& im = prob.images[h->src_idxs[s]];
      std::shared_ptr<FpEntry> e;
      const auto key = std::make_tuple((const void*)im.gray, im.width, im.height, h->src_w, h->src_h);
      if (cache) {
        std::lock_guard<std::mutex> lock(cache->mu);
        auto it = cache->entries.find(key);
        if (it != cache->entries.end()) {
          e = it->second;
          ++cache->hits;
        }
      }
      if (!e) {
        e = std::make_shared<FpEntry>();
        e->data.alloc(fp_count);
        if (!staging.ptr) staging.alloc(slot);
        HIP_CALL(hipMemsetAsync(staging.ptr, 0, slot, h->stream));
        // contiguous copy into the max-size slot without re-pitching, exactly as the
        // reference's memcpy (patch_match_cuda.cu:1617-1622)
        HIP_CALL(hipMemcpyAsync(staging.ptr, im.gray, (size_t)im.width * im.height, in_kind, h->stream));
        pm_launch_build_footprint(staging.ptr, e->data.ptr, 1, h->src_w, h->src_h, h->stream);
        HIP_CALL(hipStreamSynchronize(h->stream));
        if (cache) {
          std::lock_guard<std::mutex> lock(cache->mu);
          // two threads may miss the same key at once: only the first insertion is accounted for, the
          // second adopts the entry that is already there
          const auto ins = cache->entries.emplace(key, e);
          ++cache->misses;
          if (ins.second) {
            cache->order.push_back(key);
            cache->bytes += fp_count * sizeof(uint32_t);
            cache->Trim();
          } else {
            e = ins.first->second;
          }
        }
      }
      h->src_fp[s] = e;
      tab[s] = e->data.ptr;
    }
    // One buffer resource per problem (below) needs the S images within 4 GB of each other. Images shared through
    // the cache may sit in slabs that lie further apart (a long run packs more images than one slab holds): those of
    // them outside the slab that holds most of the problem's images are then re-homed -- copied device to device into
    // a free slot of that slab, the cache handed the new copy; problems that still use the old one keep it alive.
    if (!FpSpanFits(tab, fp_count)) RehomeSourceImages(h, cache, prob, fp_count, &tab);
    h->src_fp_tab.alloc(S);
    HIP_CALL(hipMemcpyAsync(h->src_fp_tab.ptr, tab.data(), S * sizeof(const uint32_t*), hipMemcpyHostToDevice,
                            h->stream));
    // The 11 x 11 sweep kernels read all S images through ONE buffer resource when they can (pm_kernels.hip:
    // fp_resource): base = the lowest image address, an image = the slot (address - base) / kFpStrip in the offset
    // register. The address unit forms the buffer offset in 32 bits, so every image must END within 4 GB of the
    // base; problems whose images lie further apart (separate allocations, shared through the image cache) take
    // the explicit-index build of the same kernels (fp_base = null).
    {
      const uint32_t* lo = tab[0];
      for (int s = 0; s < S; ++s) lo = std::min(lo, tab[s]);
      std::vector<uint32_t> offs(S);
      bool ok = ((uintptr_t)lo % 256) == 0;
      for (int s = 0; s < S; ++s) {
        const uint64_t d = (uint64_t)((const char*)tab[s] - (const char*)lo);
        ok = ok && d % 256 == 0 && d + fp_count * sizeof(uint32_t) + 4096 < FpSpanLimit(fp_count * sizeof(uint32_t));
        offs[s] = (uint32_t)((d / kFpStrip) & 0xffffffffull);
      }
      h->src_fp_off.alloc(S);
      HIP_CALL(hipMemcpyAsync(h->src_fp_off.ptr, offs.data(), S * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
      HIP_CALL(hipStreamSynchronize(h->stream));  // `offs` dies with this scope
      h->fp_base = ok ? lo : nullptr;
    }
    HIP_CALL(hipStreamSynchronize(h->stream));
  }
  if (opt.geom_consistency) {
    h->src_depth.alloc(slot * S);
    HIP_CALL(hipMemsetAsync(h->src_depth.ptr, 0, slot * S * sizeof(float), h->stream));
    for (int s = 0; s < S; ++s) {
      const pm_image& im = prob.images[h->src_idxs[s]];
      // row copy into the padded slot (patch_match_cuda.cu:1668-1673)
      HIP_CALL(hipMemcpy2DAsync(h->src_depth.ptr + slot * s, (size_t)h->src_w * sizeof(float),
                                im.depth_map, (size_t)im.width * sizeof(float),
                                (size_t)im.width * sizeof(float), im.height, in_kind, h->stream));
    }
  }

  // ---- InitRefImage (reference :1578-1593) ---------------------------------------
  h->ref_img.alloc((size_t)W * H);
  h->ref_sum.alloc((size_t)W * H);
  h->ref_sqsum.alloc((size_t)W * H);
  {
    DevBuf<uint8_t> gray;
    gray.alloc((size_t)W * H);
    HIP_CALL(hipMemcpyAsync(gray.ptr, ref.gray, (size_t)W * H, in_kind, h->stream));
    pm_launch_filter_ref(gray.ptr, W, H, opt.window_radius, opt.window_step, (float)opt.sigma_spatial,
                         (float)opt.sigma_color, h->ref_img.ptr, h->ref_sum.ptr, h->ref_sqsum.ptr,
                         h->stream);
    HIP_CALL(hipStreamSynchronize(h->stream));
  }

  // ---- InitTransforms -------------------------------------------------------------
  BuildPoseTables(h, prob);
  h->poses.alloc(h->poses_host.size());
  HIP_CALL(hipMemcpyAsync(h->poses.ptr, h->poses_host.data(), h->poses_host.size() * sizeof(float),
                          hipMemcpyHostToDevice, h->stream));

  // ---- InitWorkspaceMemory (reference :1810-1857) ---------------------------------
  PmParams& b = h->base;
  std::memset(&b, 0, sizeof(b));
  b.W = W; b.H = H; b.S = S; b.src_w = h->src_w; b.src_h = h->src_h;
  b.fp_xmax = (float)(h->src_w + kFpRingX); b.fp_ymax = (float)(h->src_h + kFpRingY);
  b.fp_rows1 = pm_fp_height(h->src_h) - 1;
  b.radius = opt.window_radius;
  b.step = opt.window_step;
  b.ntap1d = (2 * b.radius) / b.step + 1;
  b.ntaps = b.ntap1d * b.ntap1d;
  b.num_samples = opt.num_samples;
  b.rec_stride = 4 + 3 * S;
  b.sel_out_off = 4 + S;       // sweep 0 writes half A ...
  b.sel_in_off = 4 + 2 * S;    // ... and reads half B (= 0.5)
  b.C = pm_pick_columns(S, b.ntaps, b.num_samples, opt.geom_consistency != 0, b.radius,
                        opt.columns_per_group);
  h->threads = opt.threads_per_group > 0 ? ((opt.threads_per_group + 63) / 64) * 64 : 128;
  h->threads = std::min(h->threads, 256);  // pm_sweep_kernel __launch_bounds__
  // SweepOptions (reference :1420-1438); doubles narrowed to float where the reference does
  const float sigma_spatial = (float)opt.sigma_spatial;
  const float sigma_color = (float)opt.sigma_color;
  b.spatial_norm = 1.0f / (2.0f * sigma_spatial * sigma_spatial);
  b.color_norm = 1.0f / (2.0f * sigma_color * sigma_color);
  const float ncc_sigma = (float)opt.ncc_sigma;
  const float min_tri = (float)(opt.min_triangulation_angle * 0.0174532925199432);
  const float inc_sigma = (float)opt.incident_angle_sigma;
  // LikelihoodComputer ctor (reference :700-707, 796-802)
  b.cos_min_tri = std::cos(min_tri);
  b.inv_inc_sigma_sq = -0.5f / (inc_sigma * inc_sigma);
  b.inv_ncc_sigma_sq = -0.5f / (ncc_sigma * ncc_sigma);
  b.ncc_norm = (float)(2.0f / (std::sqrt(2.0f * M_PI) * ncc_sigma *
                               erff(2.0f / (ncc_sigma * 1.414213562f))));
  b.geom_reg = (float)opt.geom_consistency_regularizer;
  b.geom_max_cost = (float)opt.geom_consistency_max_cost;
  b.filter_min_ncc = (float)opt.filter_min_ncc;
  b.filter_cos_min_tri =
      std::cos((float)(opt.filter_min_triangulation_angle * 0.0174532925199432));
  b.filter_geom_max_cost = (float)opt.filter_geom_consistency_max_cost;
  b.filter_min_num_consistent = opt.filter_min_num_consistent;

  h->rec.alloc((size_t)W * H * b.rec_stride);
  h->rng.alloc((size_t)W * H * kRngWords);
  b.rec = h->rec.ptr;
  b.src_fp_tab = h->src_fp_tab.ptr;
  b.fp_base = h->fp_base;
  b.src_fp_off = h->src_fp_off.ptr;
  b.src_depth = h->src_depth.ptr;
  b.ref_img = h->ref_img.ptr;
  b.ref_sum = h->ref_sum.ptr;
  b.ref_sqsum = h->ref_sqsum.ptr;
  b.rng = h->rng.ptr;
  b.mask = nullptr;
  b.draws = nullptr;
  {
    // the 11 x 11 kernel reads the sweep's random numbers from a per-pixel table (one column per wave is the
    // shape RunBatchAsync may lower to; if that fits the workgroup's LDS budget the table is needed)
    PmParams probe = b;
    probe.C = 1;
    if (pm_sweep_uses_draws(probe, opt.geom_consistency != 0)) {
      h->draws.alloc((size_t)W * H * pm_draw_stride(b.num_samples));
      b.draws = h->draws.ptr;
    }
  }

  PmParams p0 = ParamsForSweep(h, 0);
  if (opt.geom_consistency) {
    DevBuf<float> d, n;
    d.alloc((size_t)W * H);
    n.alloc((size_t)3 * W * H);
    HIP_CALL(hipMemcpyAsync(d.ptr, ref.depth_map, (size_t)W * H * sizeof(float), in_kind, h->stream));
    HIP_CALL(hipMemcpyAsync(n.ptr, ref.normal_map, (size_t)3 * W * H * sizeof(float), in_kind,
                            h->stream));
    pm_launch_init_state(p0, false, 0.0f, 0.0f, d.ptr, n.ptr, h->stream);
    HIP_CALL(hipStreamSynchronize(h->stream));
  } else {
    pm_launch_init_state(p0, true, (float)opt.depth_min, (float)opt.depth_max, nullptr, nullptr,
                         h->stream);
  }
  if (opt.filter) {
    h->mask.alloc((size_t)S * W * H);
    b.mask = h->mask.ptr;
  }
  h->evals.alloc(1);
  b.evals = h->evals.ptr;
  h->out_depth.alloc((size_t)W * H);
  h->out_normal.alloc((size_t)3 * W * H);
  h->out_sel.alloc((size_t)S * W * H);
  h->out_cost.alloc((size_t)S * W * H);
  const int total_sweeps = opt.num_iterations * 4;
  h->ev.resize((size_t)2 * total_sweeps);
  for (auto& e2 : h->ev) HIP_CALL(hipEventCreate(&e2));
  HIP_CALL(hipGetLastError());
  HIP_CALL(hipStreamSynchronize(h->stream));
}

// RunWithWindowSizeAndStep, reference patch_match_cuda.cu:1393-1546, for a batch of
// `n` problems of identical shape solved together: every kernel launch covers all n
// reference images (grid.y / grid.z = problem), so one sweep launch fills the GPU
// even when a single image has too few columns. All work is enqueued on hs[0]'s
// stream; n == 1 is the reference's one-problem-at-a-time Run().
void RunBatchAsync(pm_handle** hs, int n, hipStream_t run_st = nullptr) {
  PM_CHECK(n >= 1, "empty batch");
  pm_handle* h0 = hs[0];
  HIP_CALL(hipSetDevice(h0->device));
  // the stream the run is enqueued on: the first handle's own, or the caller's (sub-batches: RunBatchSplitAsync)
  const hipStream_t st = run_st ? run_st : h0->stream;
  if (run_st) HIP_CALL(hipStreamSynchronize(h0->stream));  // create-time work of the leader (the others: below)
  const pm_options& opt = h0->opt;
  for (int b = 1; b < n; ++b) {
    const pm_handle* h = hs[b];
    PM_CHECK(h->device == h0->device, "batch on one device");
    PM_CHECK(h->W == h0->W && h->H == h0->H && h->S == h0->S && h->src_w == h0->src_w &&
                 h->src_h == h0->src_h,
             "batched problems must have identical image sizes and source counts");
    const pm_options& o = h->opt;
    PM_CHECK(o.window_radius == opt.window_radius && o.window_step == opt.window_step &&
                 o.num_samples == opt.num_samples && o.num_iterations == opt.num_iterations &&
                 o.geom_consistency == opt.geom_consistency && o.filter == opt.filter &&
                 o.max_sweeps == opt.max_sweeps,
             "batched problems must share window / sample / iteration / filter options");
    // work enqueued on other streams at create time must be complete
    HIP_CALL(hipStreamSynchronize(h->stream));
  }
  // Columns per wave by occupancy. A wave sweeps C columns top to bottom, so a launch has (problems x columns / C)
  // waves for 16 wave slots per CU. C = 2 is the fastest shape when the GPU is full (pm_pick_columns), but ONE
  // 2560 x 1920 problem -- how the reference's controller drives the seam, one problem per GPU thread
  // (mvs/patch_match.cc:190-204) -- then has 960 .. 1 280 waves for 4 096 slots: with fewer than ~3/4 of the slots
  // covered by everything alive on the device, one column per wave doubles the waves. The results do not depend
  // on C (tests: group shapes); an explicit columns_per_group is respected.
  // One launch geometry for the batch: the columns per wave of THIS run are a property of the run (every parameter
  // block of the run carries it), never written back to a handle -- a handle re-run in another batch, or traced, sees
  // its own shape again.
  int run_C = h0->base.C, run_help = 1;
  for (int b = 1; b < n; ++b) run_C = std::min(run_C, hs[b]->base.C);
  {
    bool automatic = h0->base.ntaps == 121;
    for (int b = 0; b < n; ++b) automatic = automatic && hs[b]->opt.columns_per_group <= 0;
    if (automatic && dev_switch_int("COLMAP_AMD_PM_COLS", 0) <= 0) {
      static std::atomic<int> cus[16];
      int ncu = cus[h0->device & 15].load();
      if (ncu == 0) {
        HIP_CALL(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h0->device));
        ncu = std::max(ncu, 1);
        cus[h0->device & 15] = ncu;
      }
      const long long slots = 16ll * ncu;
      const int alive = std::max(n, g_live_handles[h0->device & 15].load());
      const long long waves2 = (long long)alive * ((std::min(h0->W, h0->H) + 1) / 2);
      // (images too small to fill the GPU either way keep the common shape: their time is launch latency)
      const int C = (std::min(h0->W, h0->H) >= 512 && waves2 * 4 < slots * 3) ? 1 : 2;
      run_C = std::min(run_C, C);
      // ... and when even one wave per column leaves the GPU half empty (20 wave slots per CU for the photometric
      // kernel; ONE 2560 x 1920 problem = 1 920 .. 2 560 waves for 5 120), a second wave per column shares the NCC
      // rounds (pm_sweep_pair_kernel): bit-identical, test_group_shapes_do_not_change_results.
      if (C == 1 && run_C == 1 && 2 * waves2 * 10 <= 20ll * ncu * 6) run_help = 2;
    }
    // COLMAP_AMD_PM_HELP (tests, A/B runs): 1 = never, 2 = always (one column per wave, any image size)
    const int help_switch = dev_switch_int("COLMAP_AMD_PM_HELP", 0);
    if (help_switch == 1) run_help = 1;
    if (help_switch == 2 && h0->base.ntaps == 121) {
      run_C = 1;
      run_help = 2;
    }
  }
  const int total_sweeps = opt.num_iterations * 4;
  const int limit = opt.max_sweeps > 0 ? std::min(opt.max_sweeps, total_sweeps)
                                       : (opt.max_sweeps < 0 ? 0 : total_sweeps);
  // parameter blocks: [initial cost | sweep 0 | ... | sweep limit-1] x n problems
  std::vector<PmParams> host((size_t)(limit + 1) * n);
  for (int b = 0; b < n; ++b) {
    host[b] = ParamsForSweep(hs[b], 0);
    host[b].C = run_C;
    host[b].help = run_help;
  }
  const float total_num_steps = (float)total_sweeps;
  // workgroup -> (problem, column group) mapping of a batched sweep launch (pm_sweep_kernel)
  const int xcd_map_env = dev_switch_int("COLMAP_AMD_PM_XCD_MAP", 0);
  const int xcd_map = (xcd_map_env == 1 && n % 8 == 0) ? 1 : (xcd_map_env == 2 ? 2 : 0);
  int sel_out = h0->base.sel_out_off, sel_in = h0->base.sel_in_off;
  // one kernel serves the whole batch: buffer-resource addressing only if every problem's images allow it
  bool fp_resource_all = true;
  for (int b = 0; b < n; ++b) fp_resource_all = fp_resource_all && hs[b]->fp_base != nullptr;
  for (int k = 0; k < limit; ++k) {
    const int iter = k / 4, sweep = k % 4;
    for (int b = 0; b < n; ++b) {
      PmParams p = ParamsForSweep(hs[b], k % 4);
      // exponentially reduce the perturbation, linearly increase the influence of the
      // previous selection probabilities (reference :1446-1451)
      p.perturbation = 1.0f / std::pow(2.0f, iter + sweep / 4.0f);
#ifdef COLMAP_AMD_DIAG_BUILD
      {  // diagnostic only (results are garbage): fixed perturbation, to time a launch without far-flung random hypotheses
        const double pert = dev_switch_double("COLMAP_AMD_PM_DIAG_PERT", -1.0);
        if (pert >= 0.0) p.perturbation = (float)pert;
      }
#endif
      p.perturbation_pi = (float)(p.perturbation * M_PI);
      p.prev_sel_prob_weight = (float)(iter * 4 + sweep) / total_num_steps;
      p.sel_out_off = sel_out;
      p.sel_in_off = sel_in;
      p.xcd_map = xcd_map;
      p.C = run_C;
      p.help = run_help;
#ifdef COLMAP_AMD_DIAG_BUILD
      p.ablate = dev_switch_int("COLMAP_AMD_PM_ABLATE", 0);  // profiling builds only: results are garbage
#else
      p.ablate = 0;
#endif
      if (!fp_resource_all) p.fp_base = nullptr;
      host[(size_t)(k + 1) * n + b] = p;
    }
    std::swap(sel_out, sel_in);  // Rotate(): prev_sel_prob <- sel_prob (reference :1911-1915)
  }
  h0->plan.alloc(host.size());
  HIP_CALL(hipMemcpyAsync(h0->plan.ptr, host.data(), host.size() * sizeof(PmParams),
                          hipMemcpyHostToDevice, st));
  // the stream-ordered copy reads `host` asynchronously only for pageable memory in
  // theory; make the lifetime explicit
  HIP_CALL(hipStreamSynchronize(st));

  for (int b = 0; b < n; ++b) {
    pm_handle* h = hs[b];
    if (h->mask.ptr) HIP_CALL(hipMemsetAsync(h->mask.ptr, 0, h->mask.count, st));
    if (h->prof.ptr)
      HIP_CALL(hipMemsetAsync(h->prof.ptr, 0, kPmProfSlots * sizeof(unsigned long long), st));
    HIP_CALL(hipMemsetAsync(h->evals.ptr, 0, sizeof(unsigned long long), st));
  }
  pm_launch_initial_cost(host[0], h0->plan.ptr, n, st);
  const bool geom = opt.geom_consistency != 0;
  for (int k = 0; k < limit; ++k) {
    const bool last_sweep = k == total_sweeps - 1;
    const bool fphoto = last_sweep && opt.filter;
    const bool fgeom = last_sweep && opt.filter && geom;
    for (int b = 0; b < n; ++b)  // debug progress trace: every launch starts from an empty buffer
      if (hs[b]->trace.ptr)
        HIP_CALL(hipMemsetAsync(hs[b]->trace.ptr, 0, hs[b]->trace.count * sizeof(unsigned long long), st));
    pm_launch_draws(host[(size_t)(k + 1) * n], h0->plan.ptr + (size_t)(k + 1) * n, n, geom, st);
    HIP_CALL(hipEventRecord(h0->ev[2 * k], st));   // the events bracket the sweep kernel alone
    h0->sweep_kernel = pm_launch_sweep(host[(size_t)(k + 1) * n], h0->plan.ptr + (size_t)(k + 1) * n, n,
                                       h0->threads, geom, fphoto, fgeom, st);
    HIP_CALL(hipEventRecord(h0->ev[2 * k + 1], st));
  }
  for (int b = 0; b < n; ++b) {
    pm_handle* h = hs[b];
    h->sweeps_done = b == 0 ? limit : 0;
    h->final_sel_off = sel_in;  // the half written by the last sweep
    PmParams pe = ParamsForSweep(h, 0);
    pm_launch_extract(pe, h->final_sel_off, h->out_depth.ptr, h->out_normal.ptr, h->out_sel.ptr,
                      h->out_cost.ptr, st);
    h->ran = true;
    h->run_stream = st;
  }
  HIP_CALL(hipGetLastError());
}

void RunAsync(pm_handle* h) {
  RunBatchAsync(&h, 1);
  h->launch_images = 1;
  h->launch_concurrency = 1;
}

// pm_run_batch: a batch of 16 or more problems runs as TWO sub-batches (whole multiples of eight images where the
// count allows: a launch maps problem = workgroup id % batch, so eight problems sit on one XCD's L2 each) on the
// streams of their first handles, enqueued back to back. A sweep launch ends with a drain -- the last of its column
// groups finish one by one while the rest of the GPU idles, and the next sweep of the same images cannot start before
// that -- but the problems of the other sub-batch do not depend on it: its launches fill the drain. Measured at
// 16 x 2560 x 1920: 7.74 -> 8.07 Mpix/s; three or four sub-batches lose again (7.6-7.7: problems spread over XCDs,
// more distinct source images per L2). Results cannot change: problems are independent.
// Streams of the two sub-batches, two per device, created once and never destroyed. They carry a CU mask with every
// CU enabled: a masked stream owns its hardware queue, while ordinary streams are dealt to a pool of four queues by
// use count -- the streams of two handles may share one, and two launches in one queue run one after the other
// (measured: the same 16-image run at 8.3 Mpix/s on one process's streams and 7.3 on another's).
hipStream_t SubBatchStream(int device, int k) {
  static std::mutex mu;
  static hipStream_t streams[16][2];
  std::lock_guard<std::mutex> lock(mu);
  hipStream_t& st = streams[device & 15][k & 1];
  if (!st) {
    int ncu = 0;
    HIP_CALL(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    std::vector<uint32_t> mask((size_t)(std::max(ncu, 1) + 31) / 32, 0u);
    for (int i = 0; i < ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    HIP_CALL(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  }
  return st;
}

void RunBatchSplitAsync(pm_handle** hs, int n) {
  const bool split = n >= 16 && dev_switch_int("COLMAP_AMD_PM_BATCH_SPLIT", 1) != 0;
  if (!split) {
    RunBatchAsync(hs, n);
    for (int b = 0; b < n; ++b) {
      hs[b]->launch_images = n;
      hs[b]->launch_concurrency = 1;
    }
    return;
  }
  const int first = std::min(n - 8, ((n / 2 + 7) / 8) * 8);
  RunBatchAsync(hs, first, SubBatchStream(hs[0]->device, 0));
  RunBatchAsync(hs + first, n - first, SubBatchStream(hs[0]->device, 1));
  for (int b = 0; b < n; ++b) {
    hs[b]->launch_images = b < first ? first : n - first;
    hs[b]->launch_concurrency = 2;
  }
}

void Synchronize(pm_handle* h) {
  HIP_CALL(hipSetDevice(h->device));
  HIP_CALL(hipStreamSynchronize(h->run_stream ? h->run_stream : h->stream));
  // the batch's work is complete: later copies use the handle's own stream (the stream of the
  // batch leader may be destroyed before this handle)
  h->run_stream = nullptr;
  h->sweep_ms = 0.0;
  h->sweep_launches = h->sweeps_done;
  for (int i = 0; i < h->sweeps_done; ++i) {
    float ms = 0.0f;
    HIP_CALL(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
    h->sweep_ms += ms;
  }
}

template <typename T>
void CopyOut(pm_handle* h, const T* dev, T* out, size_t n) {
  PM_CHECK(h->ran, "pm_run must be called first");
  HIP_CALL(hipSetDevice(h->device));
  hipStream_t st = h->run_stream ? h->run_stream : h->stream;
  HIP_CALL(hipMemcpyAsync(out, dev, n * sizeof(T), hipMemcpyDeviceToHost, st));
  HIP_CALL(hipStreamSynchronize(st));
}

template <typename F>
int Guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  } catch (...) {
    g_last_error = "unknown error";
    return 2;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// pm_run from several host threads at once -- how the reference's controller drives the seam, one problem per worker
// thread (mvs/patch_match.cc:190-204, its `gpu_index = 0,0,...` idiom for several problems per GPU) -- is coalesced:
// the calls that arrive together run as ONE batch (RunBatchSplitAsync: every launch covers all of them), each caller
// returns when its own problem is done. The first arrival leads: it waits while further calls keep arriving (until
// 300 us pass without one, at most 3 ms -- 250 ms while other threads are still inside pm_create --, or until every
// live handle of the device has called), runs the calls that can
// share launches with the oldest pending one (same image size / source count / options: what pm_run_batch requires),
// and hands leadership on. Results do not depend on it (a batch equals its single runs bit for bit); a lone caller pays
// at most the 300 us. COLMAP_AMD_PM_COALESCE=0 (development switch) runs every call on its own.
// ---------------------------------------------------------------------------------------------------------------
struct RunCall {
  pm_handle* h;
  bool done = false;
  std::string err;
};
struct RunCoalescer {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<RunCall*> pending;
  bool leader_active = false;
};
static RunCoalescer g_coalescer[16];
static std::atomic<int> g_creating{0};   // pm_create calls in progress (any device)

static bool BatchCompatible(const pm_handle* a, const pm_handle* b) {
  const pm_options& x = a->opt;
  const pm_options& y = b->opt;
  return a->device == b->device && a->W == b->W && a->H == b->H && a->S == b->S && a->src_w == b->src_w &&
         a->src_h == b->src_h && x.window_radius == y.window_radius && x.window_step == y.window_step &&
         x.num_samples == y.num_samples && x.num_iterations == y.num_iterations &&
         x.geom_consistency == y.geom_consistency && x.filter == y.filter && x.max_sweeps == y.max_sweeps &&
         a->base.prof == nullptr && b->base.prof == nullptr && a->base.trace == nullptr && b->base.trace == nullptr;
}

void RunCoalesced(pm_handle* h) {
  if (dev_switch_int("COLMAP_AMD_PM_COALESCE", 1) == 0 || h->base.prof || h->base.trace) {
    RunAsync(h);
    Synchronize(h);
    return;
  }
  RunCoalescer& Q = g_coalescer[h->device & 15];
  RunCall me{h};
  std::unique_lock<std::mutex> lock(Q.mu);
  Q.pending.push_back(&me);
  Q.cv.notify_all();  // a gathering leader counts arrivals
  while (!me.done) {
    if (Q.leader_active) {
      Q.cv.wait(lock);
      continue;
    }
    Q.leader_active = true;
    {
      // gather: while other threads are inside pm_create (they are about to call: the controller's workers create and
      // run in a loop) up to 250 ms, otherwise until 300 us pass without an arrival (at most 3 ms); never beyond the
      // number of live handles. A caller that is alone -- nobody creating, nobody arriving -- leaves after 300 us.
      const auto t0 = std::chrono::steady_clock::now();
      size_t seen = Q.pending.size();
      while ((int)seen < g_live_handles[h->device & 15].load() + g_creating.load()) {
        const auto waited = std::chrono::steady_clock::now() - t0;
        const bool others_creating = g_creating.load() > 0;
        if (waited > (others_creating ? std::chrono::milliseconds(250) : std::chrono::milliseconds(3))) break;
        Q.cv.wait_for(lock, std::chrono::microseconds(300));
        if (Q.pending.size() == seen && g_creating.load() == 0) break;  // quiet: nobody else is about to call
        seen = Q.pending.size();
      }
    }
    std::vector<RunCall*> batch, rest;
    for (RunCall* c : Q.pending) (BatchCompatible(Q.pending[0]->h, c->h) ? batch : rest).push_back(c);
    Q.pending.swap(rest);
    lock.unlock();
    std::string err;
    try {
      std::vector<pm_handle*> hs;
      for (RunCall* c : batch) hs.push_back(c->h);
      if (hs.size() == 1) {
        RunAsync(hs[0]);
      } else {
        RunBatchSplitAsync(hs.data(), (int)hs.size());
      }
      for (pm_handle* x : hs) Synchronize(x);
    } catch (const std::exception& e) {
      err = e.what();
      if (err.empty()) err = "pm_run failed";
    }
    lock.lock();
    for (RunCall* c : batch) {
      c->done = true;
      c->err = err;
    }
    Q.leader_active = false;
    Q.cv.notify_all();
  }
  lock.unlock();
  if (!me.err.empty()) throw std::runtime_error(me.err);
}

extern "C" {

void pm_options_init(pm_options* o) {
  // reference mvs/patch_match_options.h:37-126 (float literals narrowed like the source)
  std::memset(o, 0, sizeof(*o));
  o->depth_min = -1.0f;
  o->depth_max = -1.0f;
  o->sigma_spatial = -1;
  o->sigma_color = 0.2f;
  o->ncc_sigma = 0.6f;
  o->min_triangulation_angle = 1.0f;
  o->incident_angle_sigma = 0.9f;
  o->geom_consistency_regularizer = 0.3f;
  o->geom_consistency_max_cost = 3.0f;
  o->filter_min_ncc = 0.1f;
  o->filter_min_triangulation_angle = 3.0f;
  o->filter_geom_consistency_max_cost = 1.0f;
  o->window_radius = 5;
  o->window_step = 1;
  o->num_samples = 15;
  o->num_iterations = 5;
  o->filter_min_num_consistent = 2;
  o->geom_consistency = 1;
  o->filter = 1;
  o->gpu_index = -1;
}

int pm_check(const pm_options* options, const pm_problem* problem) {
  return Guard([&] {
    PM_CHECK(options && problem, "null argument");
    CheckOptions(*options);
    CheckProblem(*options, *problem);
  });
}

int pm_image_cache_create(int32_t gpu_index, pm_image_cache** out) {
  return Guard([&] {
    PM_CHECK(out, "null argument");
    PM_CHECK(gpu_index >= -1, "gpu_index >= -1");
    auto* c = new pm_image_cache();
    c->device = gpu_index;  // -1: the device of the first problem that uses the cache
    *out = c;
  });
}

void pm_image_cache_destroy(pm_image_cache* cache) { delete cache; }

int pm_image_cache_set_capacity(pm_image_cache* cache, size_t max_bytes) {
  return Guard([&] {
    PM_CHECK(cache, "null argument");
    std::lock_guard<std::mutex> lock(cache->mu);
    cache->capacity = max_bytes;
    cache->Trim();
  });
}

int pm_image_cache_stats(pm_image_cache* cache, size_t* entries, size_t* hits, size_t* misses) {
  return Guard([&] {
    PM_CHECK(cache, "null argument");
    std::lock_guard<std::mutex> lock(cache->mu);
    if (entries) *entries = cache->entries.size();
    if (hits) *hits = cache->hits;
    if (misses) *misses = cache->misses;
  });
}

static int CreateImpl(const pm_options* options, const pm_problem* problem, pm_image_cache* cache, pm_handle** out) {
  if (out) *out = nullptr;
  pm_handle* h = nullptr;
  struct Creating {   // a pm_run that gathers concurrent calls waits for creations in flight (RunCoalesced)
    Creating() { ++g_creating; }
    ~Creating() { --g_creating; }
  } creating;
  const int rc = Guard([&] {
    PM_CHECK(options && problem && out, "null argument");
    h = new pm_handle();
    Create(*options, *problem, cache, h);
  });
  if (rc != 0) {
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}

int pm_create(const pm_options* options, const pm_problem* problem, pm_handle** out) {
  return CreateImpl(options, problem, nullptr, out);
}

int pm_create_cached(const pm_options* options, const pm_problem* problem, pm_image_cache* cache,
                     pm_handle** out) {
  return CreateImpl(options, problem, cache, out);
}

int pm_run_async(pm_handle* h) {
  return Guard([&] {
    PM_CHECK(h, "null handle");
    RunAsync(h);
  });
}

int pm_synchronize(pm_handle* h) {
  return Guard([&] {
    PM_CHECK(h, "null handle");
    Synchronize(h);
  });
}

int pm_run(pm_handle* h) {
  return Guard([&] {
    PM_CHECK(h, "null handle");
    RunCoalesced(h);
  });
}

int pm_run_batch_async(pm_handle** handles, int32_t n) {
  return Guard([&] {
    PM_CHECK(handles && n >= 1, "empty batch");
    for (int i = 0; i < n; ++i) PM_CHECK(handles[i], "null handle");
    RunBatchSplitAsync(handles, n);
  });
}

int pm_run_batch(pm_handle** handles, int32_t n) {
  return Guard([&] {
    PM_CHECK(handles && n >= 1, "empty batch");
    for (int i = 0; i < n; ++i) PM_CHECK(handles[i], "null handle");
    RunBatchSplitAsync(handles, n);
    for (int i = 0; i < n; ++i) Synchronize(handles[i]);
  });
}

int pm_get_depth_map(pm_handle* h, float* out) {
  return Guard([&] { PM_CHECK(h && out, "null"); CopyOut(h, h->out_depth.ptr, out, (size_t)h->W * h->H); });
}
int pm_get_normal_map(pm_handle* h, float* out) {
  return Guard([&] { PM_CHECK(h && out, "null"); CopyOut(h, h->out_normal.ptr, out, (size_t)3 * h->W * h->H); });
}
int pm_get_sel_prob_map(pm_handle* h, float* out) {
  return Guard([&] { PM_CHECK(h && out, "null"); CopyOut(h, h->out_sel.ptr, out, (size_t)h->S * h->W * h->H); });
}
int pm_get_cost_map(pm_handle* h, float* out) {
  return Guard([&] { PM_CHECK(h && out, "null"); CopyOut(h, h->out_cost.ptr, out, (size_t)h->S * h->W * h->H); });
}

int pm_get_consistency_mask(pm_handle* h, uint8_t* out) {
  return Guard([&] {
    PM_CHECK(h && out, "null");
    const size_t n = (size_t)h->S * h->W * h->H;
    if (h->mask.ptr) CopyOut(h, h->mask.ptr, out, n);
    else std::memset(out, 0, n);
  });
}

int pm_get_consistent_image_idxs(pm_handle* h, int32_t* buf, size_t capacity, size_t* count) {
  // GetConsistentImageIdxs, reference patch_match_cuda.cu:1367-1391
  return Guard([&] {
    PM_CHECK(h && count, "null");
    const size_t n = (size_t)h->S * h->W * h->H;
    std::vector<uint8_t> mask(n, 0);
    if (h->mask.ptr) CopyOut(h, h->mask.ptr, mask.data(), n);
    std::vector<int32_t> out;
    std::vector<int32_t> pix;
    for (int r = 0; r < h->H; ++r) {
      for (int c = 0; c < h->W; ++c) {
        pix.clear();
        for (int d = 0; d < h->S; ++d)
          if (mask[((size_t)d * h->H + r) * h->W + c]) pix.push_back(h->src_idxs[d]);
        if (!pix.empty()) {
          out.push_back(c);
          out.push_back(r);
          out.push_back((int32_t)pix.size());
          out.insert(out.end(), pix.begin(), pix.end());
        }
      }
    }
    *count = out.size();
    if (buf) {
      PM_CHECK(capacity >= out.size(), "buffer too small");
      std::memcpy(buf, out.data(), out.size() * sizeof(int32_t));
    }
  });
}

int pm_get_ref_filter(pm_handle* h, uint8_t* image, float* sum, float* sqsum) {
  return Guard([&] {
    PM_CHECK(h, "null");
    HIP_CALL(hipSetDevice(h->device));
    const size_t n = (size_t)h->W * h->H;
    if (image) HIP_CALL(hipMemcpy(image, h->ref_img.ptr, n, hipMemcpyDeviceToHost));
    if (sum) HIP_CALL(hipMemcpy(sum, h->ref_sum.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    if (sqsum) HIP_CALL(hipMemcpy(sqsum, h->ref_sqsum.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int pm_get_pose_tables(pm_handle* h, float* poses, float* ref_K, float* ref_inv_K) {
  return Guard([&] {
    PM_CHECK(h, "null");
    if (poses) std::memcpy(poses, h->poses_host.data(), h->poses_host.size() * sizeof(float));
    if (ref_K) std::memcpy(ref_K, h->ref_K, sizeof(h->ref_K));
    if (ref_inv_K) std::memcpy(ref_inv_K, h->ref_inv_K, sizeof(h->ref_inv_K));
  });
}

int pm_debug_rng_streams(int32_t gpu_index, const uint64_t* seeds, int32_t nseeds, int32_t ndraws,
                         float* out) {
  return Guard([&] {
    PM_CHECK(seeds && out && nseeds > 0 && ndraws > 0, "bad arguments");
    HIP_CALL(hipSetDevice(gpu_index));
    DevBuf<unsigned long long> d_seeds;
    DevBuf<float> d_out;
    d_seeds.alloc((size_t)nseeds);
    d_out.alloc((size_t)nseeds * ndraws);
    HIP_CALL(hipMemcpy(d_seeds.ptr, seeds, sizeof(uint64_t) * nseeds, hipMemcpyHostToDevice));
    pm_launch_rng_streams(d_seeds.ptr, nseeds, ndraws, d_out.ptr, nullptr);
    HIP_CALL(hipDeviceSynchronize());
    HIP_CALL(hipMemcpy(out, d_out.ptr, sizeof(float) * (size_t)nseeds * ndraws, hipMemcpyDeviceToHost));
  });
}

int pm_get_sweep_timing(pm_handle* h, double* total_ms, int32_t* num_launches) {
  return Guard([&] {
    PM_CHECK(h, "null");
    if (total_ms) *total_ms = h->sweep_ms;
    if (num_launches) *num_launches = h->sweep_launches;
  });
}

int pm_get_launch_shape(pm_handle* h, int32_t* images_per_launch, int32_t* concurrent_launches) {
  return Guard([&] {
    PM_CHECK(h, "null");
    if (images_per_launch) *images_per_launch = h->launch_images;
    if (concurrent_launches) *concurrent_launches = h->launch_concurrency;
  });
}

int pm_get_sweep_kernel_name(pm_handle* h, const char** name) {
  return Guard([&] {
    PM_CHECK(h && name, "null");
    *name = h->sweep_kernel;
  });
}

int pm_get_sweep_times(pm_handle* h, float* ms, int32_t capacity, int32_t* num_launches) {
  return Guard([&] {
    PM_CHECK(h && num_launches, "null");
    *num_launches = h->sweep_launches;
    for (int i = 0; ms && i < h->sweep_launches && i < capacity; ++i)
      HIP_CALL(hipEventElapsedTime(&ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
  });
}

int pm_get_device_maps(pm_handle* h, const float** depth, const float** normal) {
  return Guard([&] {
    PM_CHECK(h && h->ran, "run first");
    if (depth) *depth = h->out_depth.ptr;
    if (normal) *normal = h->out_normal.ptr;
  });
}

int pm_copy_maps_to_device(pm_handle* h, float* depth_dev, float* normal_dev) {
  return Guard([&] {
    PM_CHECK(h && h->ran, "run first");
    HIP_CALL(hipSetDevice(h->device));
    const size_t n = (size_t)h->W * h->H;
    if (depth_dev)
      HIP_CALL(hipMemcpyAsync(depth_dev, h->out_depth.ptr, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (normal_dev)
      HIP_CALL(hipMemcpyAsync(normal_dev, h->out_normal.ptr, 3 * n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIP_CALL(hipStreamSynchronize(h->stream));
  });
}

int pm_get_evaluation_count(pm_handle* h, unsigned long long* sweep_evals, unsigned long long* initial_evals) {
  return Guard([&] {
    PM_CHECK(h && h->ran, "pm_run must be called first");
    HIP_CALL(hipSetDevice(h->device));
    if (sweep_evals)
      HIP_CALL(hipMemcpy(sweep_evals, h->evals.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (initial_evals) *initial_evals = (unsigned long long)h->W * h->H * h->S;  // ComputeInitialCost
  });
}

int pm_enable_phase_profile(pm_handle* h, int enable) {
  return Guard([&] {
    PM_CHECK(h, "null");
    HIP_CALL(hipSetDevice(h->device));
    if (enable) {
      h->prof.alloc(kPmProfSlots);
      h->base.prof = h->prof.ptr;
    } else {
      h->base.prof = nullptr;
    }
  });
}

int pm_enable_progress_trace(pm_handle* h, int enable) {
  return Guard([&] {
    PM_CHECK(h, "null");
    HIP_CALL(hipSetDevice(h->device));
    if (enable) {
      const int longest = std::max(h->W, h->H);
      h->base.trace_stride = longest / 128 + 2;
      const size_t groups = (size_t)longest;  // one column per wave is the finest launch shape (RunBatchAsync may choose it)
      h->trace.alloc(groups * h->base.trace_stride);
      HIP_CALL(hipMemset(h->trace.ptr, 0, groups * h->base.trace_stride * sizeof(unsigned long long)));
      h->base.trace = h->trace.ptr;
    } else {
      h->base.trace = nullptr;
    }
  });
}

int pm_get_progress_trace(pm_handle* h, unsigned long long* out, size_t capacity, int32_t* groups, int32_t* samples) {
  return Guard([&] {
    PM_CHECK(h && h->trace.ptr && groups && samples, "trace not enabled");
    HIP_CALL(hipSetDevice(h->device));
    *samples = h->base.trace_stride;
    *groups = (int32_t)(h->trace.count / h->base.trace_stride);
    if (out) {
      PM_CHECK(capacity >= h->trace.count, "buffer too small");
      HIP_CALL(hipMemcpy(out, h->trace.ptr, h->trace.count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
  });
}

int pm_get_phase_profile(pm_handle* h, unsigned long long* out10) {
  return Guard([&] {
    PM_CHECK(h && out10 && h->prof.ptr, "profile not enabled");
    HIP_CALL(hipSetDevice(h->device));
    HIP_CALL(hipMemcpy(out10, h->prof.ptr, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  });
}

int pm_get_phase_profile_slots(pm_handle* h, unsigned long long* out, int32_t capacity) {
  return Guard([&] {
    PM_CHECK(h && out && h->prof.ptr, "profile not enabled");
    PM_CHECK(capacity >= kPmProfSlots, "buffer too small");
    HIP_CALL(hipSetDevice(h->device));
    HIP_CALL(hipMemcpy(out, h->prof.ptr, kPmProfSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  });
}

void pm_destroy(pm_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  // a non-leader handle of a batched run has its kernels on the leader's stream; that stream may
  // already be destroyed (the leader went first), in which case only a device-wide wait is safe
  // (pm_synchronize resets run_stream, so this is the exception / GC path only)
  if (h->run_stream && h->run_stream != h->stream) (void)hipDeviceSynchronize();
  delete h;
}

int pm_set_cached_memory_limit(double gigabytes) {
  return Guard([&] {
    PM_CHECK(gigabytes >= 0.0 && gigabytes <= 1e6, "limit in GB, >= 0");
    DevPool::Get().SetCap(static_cast<size_t>(gigabytes * (double)(1ull << 30)));
  });
}

void pm_release_cached_memory(void) {
  DevPool::Get().Release();
  FpSlabPool::Get().Release();
}

unsigned long long pm_debug_set_image_slab_slots(size_t slots) {
  g_fp_slab_slots = slots;
  return g_fp_rehomed.load();
}

const char* pm_last_error(void) { return g_last_error.c_str(); }

int pm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
