// fusion_oracle.cpp -- TEST INFRASTRUCTURE ONLY (tests/, never linked or imported by the product).
//
// CPU checker of the depth-map fusion step (reference src/colmap/mvs/fusion.cc): the reference's
// algorithm restated statement by statement (StereoFusion::Run / Fuse, fusion.cc:253-320, 401-524),
// one pixel's turn after the other, every turn masking what it absorbs at once. Three modes:
//   mode 0  the pixels of an image take their turns in row-major order (the reference with
//           num_threads = 1). Pinned against a float32 Python restatement in tests/test_fusion.py.
//   mode 1  the turns follow the schedule of the reference's own thread pool (fusion.cc:253-269, 293-300: the tasks
//           are stripes of kRowStride = 10 rows, each walked row-major by one thread) with its T = num_threads
//           threads advancing in step: thread t takes stripes t, t + T, ..., and in every tick each thread takes the
//           next pixel of its stripe (struct Pool). T = 1 is mode 0's order; num_threads <= 0 means one thread per
//           stripe. The points come out as the reference collects them: per thread, threads concatenated
//           (fusion.cc:322-337). Two documented differences from mode 0, both of colmap_amd/csrc/fusion.hip: at most
//           record_capacity pixels per walk, and a neighbour projection is range-tested as a float before the
//           conversion to int. The HIP path must reproduce this sequential run bit for bit.
//   mode 2  a plain C++ simulation of HOW fusion.hip executes that order (RunPasses below): one sequential wave per
//           pool thread, all threads speculating through a window of ticks at once, tentative marks in a per-pixel
//           claim word, the lowest rank that lost a claim cuts the committed prefix. The threads are interleaved
//           pseudo-randomly, so the test "mode 2 == mode 1" is the executable form of the argument that the
//           parallel schedule equals the sequential one whatever the timing.
//   mode 3  the reference's pool AS IT RUNS: T = num_threads real threads (<= 0: all cores) pulling the stripes of an
//           image from a shared counter and racing on the pixel masks exactly as mvs/fusion.cc does (its result is
//           timing-dependent for T > 1, and so is this mode's). Not a checker of anything: it exists to TIME the
//           reference's own multi-threaded path on the host's cores (bench.py cpu_baseline of the fusion leg).
// All arithmetic float like the reference (Eigen::Vector3f / Matrix<float,3,4>), medians through
// colmap::Percentile (math/math.h:205-224). Build: oracle/Makefile (-ffp-contract=off).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <climits>
#include <cstdint>

#include "../include/colmap_amd_fusion.h"

namespace {

thread_local std::string g_error;

struct Fail : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define FU_CHECK(cond, msg)                                             \
  do {                                                                  \
    if (!(cond)) throw Fail(std::string("Check failed: ") + (msg));     \
  } while (0)

// mvs/image.cc:106-135
void ComposeProjectionMatrix(const float K[9], const float R[9], const float T[3], float P[12]) {
  float RT[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) RT[4 * r + c] = R[3 * r + c];
    RT[4 * r + 3] = T[r];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) P[4 * r + c] = K[3 * r] * RT[c] + K[3 * r + 1] * RT[4 + c] + K[3 * r + 2] * RT[8 + c];
}

// inverse of [P; 0 0 0 1], top three rows: [M^-1 | -M^-1 p] with M = P(:, 0:3), by the adjugate
void ComposeInverseProjectionMatrix(const float K[9], const float R[9], const float T[3], float inv_P[12]) {
  float P[12];
  ComposeProjectionMatrix(K, R, T, P);
  const float a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[6], g = P[8], h = P[9], i = P[10];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  const float inv_det = 1.0f / det;
  const float Mi[9] = {A * inv_det,           -(b * i - c * h) * inv_det, (b * f - c * e) * inv_det,
                       B * inv_det,           (a * i - c * g) * inv_det,  -(a * f - c * d) * inv_det,
                       C * inv_det,           -(a * h - b * g) * inv_det, (a * e - b * d) * inv_det};
  for (int r = 0; r < 3; ++r) {
    for (int col = 0; col < 3; ++col) inv_P[4 * r + col] = Mi[3 * r + col];
    inv_P[4 * r + 3] = -(Mi[3 * r] * P[3] + Mi[3 * r + 1] * P[7] + Mi[3 * r + 2] * P[11]);
  }
}

// colmap::Percentile(elems, 50) (math/math.h:205-234), returns double like the reference
template <typename T>
double Median(std::vector<T>& elems) {
  const double idx_double = 50.0 / 100. * (elems.size() - 1);
  const double left_idx_double = std::floor(idx_double);
  const size_t left_idx = static_cast<size_t>(left_idx_double);
  const double right_idx_double = std::ceil(idx_double);
  const size_t right_idx = static_cast<size_t>(right_idx_double);
  std::nth_element(elems.begin(), elems.begin() + right_idx, elems.end());
  const double right = elems[right_idx];
  if (left_idx == right_idx) return right;
  const double left = *std::max_element(elems.begin(), elems.begin() + right_idx);
  return (right_idx_double - idx_double) * left + (idx_double - left_idx_double) * right;
}

uint8_t TruncateCastU8(float v) {  // TruncateCast<float, uint8_t> (math/math.h)
  return static_cast<uint8_t>(std::min(255.0f, std::max(0.0f, v)));
}

struct FusionData {
  int image_idx, row, col, traversal_depth;
};

}  // namespace

struct fusion_result {
  std::vector<float> xyz_normal;
  std::vector<uint8_t> rgb;
  std::vector<int64_t> vis_ptr{0};
  std::vector<int32_t> vis_idx;
};

namespace {

// modes 1 / 2: pixels one walk can record = the lane state of fusion.hip (record_capacity there): max_num_pixels
// itself between 1 024 and 16 384, so the reference's default 10 000 is not clamped
inline int RecordCapacity(int max_num_pixels) { return std::min(std::max(max_num_pixels, 1024), 16384); }

struct Fuser {
  int mode = 0;
  const fusion_options& opt;
  const int n;
  const fusion_image* images;
  const int32_t* optr;
  const int32_t* oidx;
  const float max_squared_reproj_error, min_cos_normal_error;
  std::vector<char> used, fused;
  std::vector<std::vector<char>> masks;
  std::vector<float> P, inv_P, inv_R, scale;  // per image 12 / 12 / 9 / 2 floats
  fusion_result* out;

  Fuser(const fusion_options& o, int n_, const fusion_image* im, const int32_t* op, const int32_t* oi, fusion_result* r)
      : opt(o), n(n_), images(im), optr(op), oidx(oi),
        max_squared_reproj_error(static_cast<float>(o.max_reproj_error * o.max_reproj_error)),
        min_cos_normal_error(static_cast<float>(std::cos(o.max_normal_error * 0.017453292519943295769))),
        used(n_, 0), fused(n_, 0), masks(n_), P(12 * (size_t)n_), inv_P(12 * (size_t)n_), inv_R(9 * (size_t)n_),
        scale(2 * (size_t)n_), out(r) {}

  void Init() {  // fusion.cc:201-251
    for (int i = 0; i < n; ++i) {
      const fusion_image& im = images[i];
      if (!im.used) continue;
      FU_CHECK(im.depth_map && im.normal_map && im.depth_width > 0 && im.depth_height > 0, "depth / normal map");
      FU_CHECK(im.width > 0 && im.height > 0, "image size");
      used[i] = 1;
      masks[i].assign((size_t)im.depth_width * im.depth_height, 0);
      if (im.mask)
        for (size_t k = 0; k < masks[i].size(); ++k) masks[i][k] = im.mask[k] ? 1 : 0;
      scale[2 * i] = static_cast<float>(im.depth_width) / im.width;
      scale[2 * i + 1] = static_cast<float>(im.depth_height) / im.height;
      float K[9];
      std::memcpy(K, im.K, sizeof(K));
      K[0] *= scale[2 * i]; K[2] *= scale[2 * i];
      K[4] *= scale[2 * i + 1]; K[5] *= scale[2 * i + 1];
      ComposeProjectionMatrix(K, im.R, im.T, &P[12 * (size_t)i]);
      ComposeInverseProjectionMatrix(K, im.R, im.T, &inv_P[12 * (size_t)i]);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) inv_R[9 * (size_t)i + 3 * r + c] = im.R[3 * c + r];
    }
  }

  int FindNextImage(int prev) const {  // fusion.cc:51-73
    for (int k = optr[prev]; k < optr[prev + 1]; ++k)
      if (used[oidx[k]] && !fused[oidx[k]]) return oidx[k];
    for (int i = 0; i < n; ++i)
      if (used[i] && !fused[i]) return i;
    return -1;
  }

  // The schedule of the reference's thread pool (fusion.cc:253-269, 293-300) with T threads advancing in step:
  // stripe k = rows [10 k, 10 k + 10) is the k-th task; thread t runs tasks t, t + T, t + 2T, ...; a task takes
  // L = 10 W ticks (the last stripe of an image may be shorter: its thread idles). The turn of thread t in tick
  // tau has rank tau * T + t in the sequential order that defines the result.
  struct Pool {
    int W, H, ns, T, G;
    long long L;
    Pool(int w, int h, int num_threads) : W(w), H(h) {
      ns = (h + 9) / 10;
      T = num_threads <= 0 ? ns : std::min(num_threads, ns);
      G = (ns + T - 1) / T;
      L = 10ll * w;
    }
    long long ticks() const { return (long long)G * L; }
    int seed(long long tau, int t) const {  // pixel whose turn thread t takes in tick tau, or -1
      const long long k = (tau / L) * T + t;
      if (k >= ns) return -1;
      const long long pos = tau % L;
      const int row = (int)(10 * k + pos / W);
      if (row >= H) return -1;
      return row * W + (int)(pos % W);
    }
  };

  std::vector<fusion_result> per_thread;  // task_fused_points_[thread_id] (fusion.cc:198-199)
  void Concatenate() {                    // fusion.cc:322-337
    for (const fusion_result& r : per_thread) {
      const int64_t base = (int64_t)out->vis_idx.size();
      out->xyz_normal.insert(out->xyz_normal.end(), r.xyz_normal.begin(), r.xyz_normal.end());
      out->rgb.insert(out->rgb.end(), r.rgb.begin(), r.rgb.end());
      out->vis_idx.insert(out->vis_idx.end(), r.vis_idx.begin(), r.vis_idx.end());
      for (size_t k = 1; k < r.vis_ptr.size(); ++k) out->vis_ptr.push_back(base + r.vis_ptr[k]);
    }
  }

  void Run() {  // fusion.cc:253-320
    for (int image_idx = 0; image_idx >= 0; image_idx = FindNextImage(image_idx)) {
      if (used[image_idx]) {
        const int width = images[image_idx].depth_width, height = images[image_idx].depth_height;
        if (mode == 0) {
          for (int row = 0; row < height; ++row)
            for (int col = 0; col < width; ++col) Fuse(image_idx, row, col, out);
        } else if (mode == 3) {  // fusion.cc:253-269, 293-300: the pool, for real
          const int ns = (height + 9) / 10;
          int T = opt.num_threads <= 0 ? (int)std::max(1u, std::thread::hardware_concurrency()) : opt.num_threads;
          T = std::min(T, ns);
          if ((int)per_thread.size() < T) per_thread.resize(T);
          std::atomic<int> next{0};
          std::vector<std::thread> pool;
          for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t] {
              for (int k = next++; k < ns; k = next++)
                for (int row = 10 * k; row < std::min(10 * k + 10, height); ++row)
                  for (int col = 0; col < width; ++col) Fuse(image_idx, row, col, &per_thread[t]);
            });
          for (auto& th : pool) th.join();
        } else {
          const Pool pl(width, height, opt.num_threads);
          if ((int)per_thread.size() < pl.T) per_thread.resize(pl.T);
          for (long long tau = 0; tau < pl.ticks(); ++tau)
            for (int t = 0; t < pl.T; ++t) {
              const int s_ = pl.seed(tau, t);
              if (s_ >= 0) Fuse(image_idx, s_ / width, s_ % width, &per_thread[t]);
            }
        }
      }
      fused[image_idx] = 1;
    }
    if (mode != 0) Concatenate();
  }

  void Fuse(int image_idx0, int row0, int col0, fusion_result* dst) {  // fusion.cc:401-524
    std::vector<FusionData> queue;
    queue.push_back({image_idx0, row0, col0, 0});
    float ref_point[4] = {0, 0, 0, 0};
    float ref_normal[3] = {0, 0, 0};
    std::vector<float> px, py, pz, nx, ny, nz;
    std::vector<uint8_t> cr, cg, cb;
    std::vector<int> vis_order;  // insertion order of distinct images
    std::unordered_set<int> vis;
    int recorded = 0;
    const int kRecordCap = RecordCapacity(opt.max_num_pixels);
    const size_t max_pixels = (mode == 0 || mode == 3) ? (size_t)opt.max_num_pixels : (size_t)std::min(opt.max_num_pixels, kRecordCap);

    while (!queue.empty()) {
      const FusionData data = queue.back();
      queue.pop_back();
      const int image_idx = data.image_idx, row = data.row, col = data.col, depth_level = data.traversal_depth;
      const fusion_image& im = images[image_idx];
      std::vector<char>& mask = masks[image_idx];
      const size_t pix = (size_t)row * im.depth_width + col;
      if (mask[pix] > 0) continue;
      const float depth = im.depth_map[pix];
      if (depth <= 0.0f) continue;
      const float* Pi = &P[12 * (size_t)image_idx];
      if (depth_level > 0) {
        float proj[3];
        for (int r = 0; r < 3; ++r)
          proj[r] = Pi[4 * r] * ref_point[0] + Pi[4 * r + 1] * ref_point[1] + Pi[4 * r + 2] * ref_point[2] +
                    Pi[4 * r + 3] * ref_point[3];
        const float depth_error = std::abs((proj[2] - depth) / depth);
        if (depth_error > opt.max_depth_error) continue;
        const float col_diff = proj[0] / proj[2] - col;
        const float row_diff = proj[1] / proj[2] - row;
        const float squared_reproj_error = col_diff * col_diff + row_diff * row_diff;
        if (squared_reproj_error > max_squared_reproj_error) continue;
      }
      const size_t slice = (size_t)im.depth_width * im.depth_height;
      const float nl[3] = {im.normal_map[pix], im.normal_map[slice + pix], im.normal_map[2 * slice + pix]};
      const float* iR = &inv_R[9 * (size_t)image_idx];
      float normal[3];
      for (int r = 0; r < 3; ++r) normal[r] = iR[3 * r] * nl[0] + iR[3 * r + 1] * nl[1] + iR[3 * r + 2] * nl[2];
      if (depth_level > 0) {
        const float cos_normal_error = ref_normal[0] * normal[0] + ref_normal[1] * normal[1] + ref_normal[2] * normal[2];
        if (cos_normal_error < min_cos_normal_error) continue;
      }
      const float* iP = &inv_P[12 * (size_t)image_idx];
      const float hx = col * depth, hy = row * depth;
      float xyz[3];
      for (int r = 0; r < 3; ++r) xyz[r] = iP[4 * r] * hx + iP[4 * r + 1] * hy + iP[4 * r + 2] * depth + iP[4 * r + 3] * 1.0f;
      // colour: nearest neighbour at the bitmap scale (InterpolateNearestNeighbor, bitmap.cc:329-334);
      // outside the bitmap -> BitmapColor(0)
      uint8_t color[3] = {0, 0, 0};
      if (im.rgb) {
        const int xx = static_cast<int>(std::round(static_cast<double>(col / scale[2 * image_idx])));
        const int yy = static_cast<int>(std::round(static_cast<double>(row / scale[2 * image_idx + 1])));
        if (xx >= 0 && yy >= 0 && xx < im.bitmap_width && yy < im.bitmap_height)
          std::memcpy(color, im.rgb + 3 * ((size_t)yy * im.bitmap_width + xx), 3);
      }
      if (mode != 0 && mode != 3 && recorded >= kRecordCap) break;
      ++recorded;
      mask[pix] = 1;
      if (xyz[0] < opt.bbox_min[0] || xyz[1] < opt.bbox_min[1] || xyz[2] < opt.bbox_min[2] ||
          xyz[0] > opt.bbox_max[0] || xyz[1] > opt.bbox_max[1] || xyz[2] > opt.bbox_max[2])
        continue;
      px.push_back(xyz[0]); py.push_back(xyz[1]); pz.push_back(xyz[2]);
      nx.push_back(normal[0]); ny.push_back(normal[1]); nz.push_back(normal[2]);
      cr.push_back(color[0]); cg.push_back(color[1]); cb.push_back(color[2]);
      if (vis.insert(image_idx).second) vis_order.push_back(image_idx);
      if (depth_level == 0) {
        ref_point[0] = xyz[0]; ref_point[1] = xyz[1]; ref_point[2] = xyz[2]; ref_point[3] = 1.0f;
        std::memcpy(ref_normal, normal, sizeof(normal));
      }
      if (px.size() >= max_pixels) break;
      if (depth_level >= opt.max_traversal_depth - 1) continue;
      for (int k = optr[image_idx]; k < optr[image_idx + 1]; ++k) {
        const int next = oidx[k];
        if (!used[next] || fused[next]) continue;
        const float* Pn = &P[12 * (size_t)next];
        float np[3];
        for (int r = 0; r < 3; ++r) np[r] = Pn[4 * r] * xyz[0] + Pn[4 * r + 1] * xyz[1] + Pn[4 * r + 2] * xyz[2] + Pn[4 * r + 3];
        int next_col, next_row;
        if (mode == 0 || mode == 3) {
          next_col = static_cast<int>(std::round(np[0] / np[2]));
          next_row = static_cast<int>(std::round(np[1] / np[2]));
          if (next_col < 0 || next_row < 0 || next_col >= images[next].depth_width || next_row >= images[next].depth_height)
            continue;
        } else {
          // range test on the rounded float: a NaN / out-of-int-range quotient is rejected here (the
          // reference converts first, which is undefined for such values)
          const float fcol = std::round(np[0] / np[2]), frow = std::round(np[1] / np[2]);
          if (!(fcol >= 0.0f && frow >= 0.0f && fcol < static_cast<float>(images[next].depth_width) &&
                frow < static_cast<float>(images[next].depth_height)))
            continue;
          next_col = static_cast<int>(fcol);
          next_row = static_cast<int>(frow);
        }
        queue.push_back({next, next_row, next_col, depth_level + 1});
      }
    }

    Emit(px, py, pz, nx, ny, nz, cr, cg, cb, vis_order, dst);
  }

  // fusion.cc:491-523
  void Emit(std::vector<float>& px, std::vector<float>& py, std::vector<float>& pz, std::vector<float>& nx,
            std::vector<float>& ny, std::vector<float>& nz, std::vector<uint8_t>& cr, std::vector<uint8_t>& cg,
            std::vector<uint8_t>& cb, std::vector<int>& vis_order, fusion_result* dst) const {
    if (px.size() < static_cast<size_t>(opt.min_num_pixels) || px.empty()) return;
    float fn[3] = {static_cast<float>(Median(nx)), static_cast<float>(Median(ny)), static_cast<float>(Median(nz))};
    const float norm = std::sqrt(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
    if (norm < FLT_EPSILON) return;
    const float pt[6] = {static_cast<float>(Median(px)), static_cast<float>(Median(py)), static_cast<float>(Median(pz)),
                         fn[0] / norm, fn[1] / norm, fn[2] / norm};
    dst->xyz_normal.insert(dst->xyz_normal.end(), pt, pt + 6);
    dst->rgb.push_back(TruncateCastU8(std::round(static_cast<float>(Median(cr)))));
    dst->rgb.push_back(TruncateCastU8(std::round(static_cast<float>(Median(cg)))));
    dst->rgb.push_back(TruncateCastU8(std::round(static_cast<float>(Median(cb)))));
    // the reference copies a FlatHashSet (unspecified order); here: sorted image indices
    std::sort(vis_order.begin(), vis_order.end());
    dst->vis_idx.insert(dst->vis_idx.end(), vis_order.begin(), vis_order.end());
    dst->vis_ptr.push_back(static_cast<int64_t>(dst->vis_idx.size()));
  }

  // -------------------------------------------------------------------------------------------
  // mode 2: how fusion.hip executes the pool schedule, simulated
  //
  // One wave per pool thread walks that thread's turns one after the other, exactly like the reference's thread
  // does -- but the T waves run concurrently, and a turn of rank r = tau * T + t must see the marks of ALL turns of
  // lower rank, also those of other threads that may not have happened yet. So the turns are speculative:
  //   * every pixel has one 64-bit word: 0 free, kCommitted (masked for good), or epoch << 32 | ~rank = tentative
  //     mark of the turn `rank` of this pass (atomicMax: the lowest rank of the pass keeps the word);
  //   * a wave treats as masked: committed pixels and the tentative marks of its OWN thread (its earlier turns
  //     of this pass, and the pixels of the walk in progress); marks of other threads read as free;
  //   * whenever a mark meets a mark of another turn of the same pass, the later of the two turns (the higher
  //     rank) has seen -- or will have seen -- a mask state the sequential order would not have given it:
  //     rstar = min(rstar, that rank);
  //   * at the end of the pass every turn of rank < rstar is final (by induction over the ranks: it met no mark
  //     of a lower rank of another thread, the lower ranks of its own thread are final, so the masks it saw are
  //     the sequential ones on every pixel it looked at) and commits: its pixels become kCommitted, it is fused;
  //     everything else is forgotten by bumping the epoch, and the next pass starts at rstar. The lowest rank of
  //     a pass is never the later of two turns, so every pass commits at least one turn.
  // A pass covers a window of P ticks (doubling after a pass without a cut, halving after one with); a wave whose
  // record buffer is full, or whose next turn is already beyond rstar, stops early (rstar = its rank: cutting the
  // prefix is always safe). Inside a walk the neighbours of an absorbed pixel are tested when they are PUSHED
  // (depth, reprojection and normal tests depend only on the walk's first pixel; a pixel masked at push time
  // stays masked) so that the stack only holds candidates whose masks have to be looked at again when popped.
  // -------------------------------------------------------------------------------------------
  static constexpr unsigned long long kCommitted = ~0ull;
  struct Node { int image, pix, level; };
  struct Rec { int image, pix; bool in_box; };
  struct WalkRec { long long tau; int first, count; };
  struct Lane {  // one pool thread = one wave of fusion.hip
    std::vector<Rec> rec;
    std::vector<WalkRec> walks;
    long long tau = 0, nodes = 0, scans = 0;
    bool stopped = false;
  };
  std::vector<std::vector<unsigned long long>> word;
  unsigned epoch = 1;
  int simT = 1;
  unsigned long long rstar = 0;
  long long passes_run = 0, walks_run = 0, walks_discarded = 0, conflicts = 0, cuts = 0;
  double model_us = 0.0;  // critical path under a latency model: per pass max over waves of (3 us per absorbed pixel + 1 us per 64 scanned turns) + 40 us

  bool MaskedFor(unsigned long long w, int t) const {
    if (w == kCommitted) return true;
    return (unsigned)(w >> 32) == epoch && (int)((0xFFFFFFFFu - (unsigned)w) % (unsigned)simT) == t;
  }

  // world-frame normal, 3-D point and bounding-box test of a pixel (fusion.cc:437-466)
  void PixelGeometry(int img, int pix, float normal[3], float xyz[3], bool* in_box) const {
    const fusion_image& im = images[img];
    const int row = pix / im.depth_width, col = pix % im.depth_width;
    const float depth = im.depth_map[pix];
    const size_t slice = (size_t)im.depth_width * im.depth_height;
    const float nl[3] = {im.normal_map[pix], im.normal_map[slice + pix], im.normal_map[2 * slice + pix]};
    const float* iR = &inv_R[9 * (size_t)img];
    for (int r = 0; r < 3; ++r) normal[r] = iR[3 * r] * nl[0] + iR[3 * r + 1] * nl[1] + iR[3 * r + 2] * nl[2];
    const float* iP = &inv_P[12 * (size_t)img];
    const float hx = col * depth, hy = row * depth;
    for (int r = 0; r < 3; ++r) xyz[r] = iP[4 * r] * hx + iP[4 * r + 1] * hy + iP[4 * r + 2] * depth + iP[4 * r + 3] * 1.0f;
    *in_box = !(xyz[0] < opt.bbox_min[0] || xyz[1] < opt.bbox_min[1] || xyz[2] < opt.bbox_min[2] ||
                xyz[0] > opt.bbox_max[0] || xyz[1] > opt.bbox_max[1] || xyz[2] > opt.bbox_max[2]);
  }

  // the tests of a pixel reached at traversal depth > 0 that do not depend on the masks (fusion.cc:407-447)
  bool PassesStatic(int img, int pix, const float ref_point[4], const float ref_normal[3]) const {
    const fusion_image& im = images[img];
    const int row = pix / im.depth_width, col = pix % im.depth_width;
    const float depth = im.depth_map[pix];
    if (depth <= 0.0f) return false;
    const float* Pi = &P[12 * (size_t)img];
    float proj[3];
    for (int r = 0; r < 3; ++r)
      proj[r] = Pi[4 * r] * ref_point[0] + Pi[4 * r + 1] * ref_point[1] + Pi[4 * r + 2] * ref_point[2] + Pi[4 * r + 3] * ref_point[3];
    const float depth_error = std::abs((proj[2] - depth) / depth);
    if (depth_error > opt.max_depth_error) return false;
    const float col_diff = proj[0] / proj[2] - col;
    const float row_diff = proj[1] / proj[2] - row;
    if (col_diff * col_diff + row_diff * row_diff > max_squared_reproj_error) return false;
    const size_t slice = (size_t)im.depth_width * im.depth_height;
    const float nl[3] = {im.normal_map[pix], im.normal_map[slice + pix], im.normal_map[2 * slice + pix]};
    const float* iR = &inv_R[9 * (size_t)img];
    float normal[3];
    for (int r = 0; r < 3; ++r) normal[r] = iR[3 * r] * nl[0] + iR[3 * r + 1] * nl[1] + iR[3 * r + 2] * nl[2];
    const float c = ref_normal[0] * normal[0] + ref_normal[1] * normal[1] + ref_normal[2] * normal[2];
    return !(c < min_cos_normal_error);
  }

  // One turn of thread t (rank `rank`, start pixel `seed` of image I: free for this thread, positive depth).
  // false: the wave's record buffer is full -- the turn is abandoned and cuts the pass.
  bool WalkSim(int t, unsigned rank, int I, int seed, Lane& ln, size_t cap, long long tau) {
    const unsigned long long key = ((unsigned long long)epoch << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
    const int kRecordCap = RecordCapacity(opt.max_num_pixels);
    const size_t max_pixels = (size_t)std::min(opt.max_num_pixels, kRecordCap);
    std::vector<Node> stack;
    const size_t first = ln.rec.size();
    float ref_point[4] = {0, 0, 0, 1.0f}, ref_normal[3] = {0, 0, 0};
    Node cand{I, seed, 0};
    size_t n_in = 0;
    int recorded = 0;
    for (;;) {
      float normal[3], xyz[3];
      bool in_box;
      PixelGeometry(cand.image, cand.pix, normal, xyz, &in_box);
      if (recorded >= kRecordCap) break;
      if (ln.rec.size() >= cap) {
        rstar = std::min<unsigned long long>(rstar, rank);
        ln.rec.resize(first);
        return false;
      }
      ln.rec.push_back({cand.image, cand.pix, in_box});
      ++recorded;
      ++ln.nodes;
      unsigned long long& w = word[cand.image][cand.pix];
      const unsigned long long old = w;  // atomicMax returns the old word
      w = std::max(w, key);
      if ((unsigned)(old >> 32) == epoch) {
        const unsigned other = 0xFFFFFFFFu - (unsigned)old;
        if (other != rank) {
          rstar = std::min<unsigned long long>(rstar, std::max(other, rank));
          ++conflicts;
        }
      }
      bool expand = false;
      if (in_box) {
        ++n_in;
        if (cand.level == 0) {
          ref_point[0] = xyz[0]; ref_point[1] = xyz[1]; ref_point[2] = xyz[2];
          std::memcpy(ref_normal, normal, sizeof(ref_normal));
        }
        if (n_in >= max_pixels) break;
        expand = cand.level < opt.max_traversal_depth - 1;
      }
      if (expand) {
        for (int k = optr[cand.image]; k < optr[cand.image + 1]; ++k) {
          const int next = oidx[k];
          if (!used[next] || fused[next]) continue;
          const float* Pn = &P[12 * (size_t)next];
          float np[3];
          for (int r = 0; r < 3; ++r) np[r] = Pn[4 * r] * xyz[0] + Pn[4 * r + 1] * xyz[1] + Pn[4 * r + 2] * xyz[2] + Pn[4 * r + 3];
          const float fcol = std::round(np[0] / np[2]), frow = std::round(np[1] / np[2]);
          if (!(fcol >= 0.0f && frow >= 0.0f && fcol < static_cast<float>(images[next].depth_width) &&
                frow < static_cast<float>(images[next].depth_height)))
            continue;
          const int q = static_cast<int>(frow) * images[next].depth_width + static_cast<int>(fcol);
          if (MaskedFor(word[next][q], t)) continue;
          if (!PassesStatic(next, q, ref_point, ref_normal)) continue;
          stack.push_back({next, q, cand.level + 1});
        }
      }
      bool found = false;
      while (!stack.empty() && !found) {
        const Node e = stack.back();
        stack.pop_back();
        if (MaskedFor(word[e.image][e.pix], t)) continue;
        cand = e;
        found = true;
      }
      if (!found) break;
    }
    ln.walks.push_back({tau, (int)first, (int)(ln.rec.size() - first)});
    return true;
  }

  // a committed turn: mask its pixels for good and fuse them (fusion.cc:449-466, 491-523)
  void CommitWalk(const Lane& ln, const WalkRec& wr, fusion_result* dst) {
    std::vector<float> px, py, pz, nx, ny, nz;
    std::vector<uint8_t> cr, cg, cb;
    std::vector<int> vis;
    for (int e = wr.first; e < wr.first + wr.count; ++e) {
      const Rec& q = ln.rec[e];
      word[q.image][q.pix] = kCommitted;
      if (!q.in_box) continue;
      float normal[3], xyz[3];
      bool in_box;
      PixelGeometry(q.image, q.pix, normal, xyz, &in_box);
      const fusion_image& im = images[q.image];
      const int row = q.pix / im.depth_width, col = q.pix % im.depth_width;
      uint8_t color[3] = {0, 0, 0};
      if (im.rgb) {
        const int xx = static_cast<int>(std::round(static_cast<double>(col / scale[2 * q.image])));
        const int yy = static_cast<int>(std::round(static_cast<double>(row / scale[2 * q.image + 1])));
        if (xx >= 0 && yy >= 0 && xx < im.bitmap_width && yy < im.bitmap_height)
          std::memcpy(color, im.rgb + 3 * ((size_t)yy * im.bitmap_width + xx), 3);
      }
      px.push_back(xyz[0]); py.push_back(xyz[1]); pz.push_back(xyz[2]);
      nx.push_back(normal[0]); ny.push_back(normal[1]); nz.push_back(normal[2]);
      cr.push_back(color[0]); cg.push_back(color[1]); cb.push_back(color[2]);
      if (std::find(vis.begin(), vis.end(), q.image) == vis.end()) vis.push_back(q.image);
    }
    Emit(px, py, pz, nx, ny, nz, cr, cg, cb, vis, dst);
  }

  static long long EnvLL(const char* name, long long dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atoll(e) : dflt;
  }

  void RunPasses() {
    // the schedule constants of fusion.hip (kWindowMin / kWindowMax / the record buffer of a wave)
    const long long p_min = EnvLL("FUO_WINDOW_MIN", 16), p_max = EnvLL("FUO_WINDOW_MAX", 4096);
    const long long p_first = EnvLL("FUO_WINDOW_FIRST", 256);
    const size_t cap = (size_t)std::max<long long>(EnvLL("FUO_RECORD_CAP", 1 << 16), RecordCapacity(opt.max_num_pixels));
    const int sched = (int)EnvLL("FUO_INTERLEAVE", 0);  // 0 pseudo-random, 1 thread T-1 first, 2 thread 0 first
    word.assign(n, {});
    for (int i = 0; i < n; ++i) {
      if (!used[i]) continue;
      word[i].assign(masks[i].size(), 0ull);
      for (size_t k = 0; k < masks[i].size(); ++k)
        if (masks[i][k]) word[i][k] = kCommitted;
    }
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int I = 0; I >= 0; I = FindNextImage(I)) {
      if (used[I]) {
        const Pool pl(images[I].depth_width, images[I].depth_height, opt.num_threads);
        const int T = simT = pl.T;
        if ((int)per_thread.size() < T) per_thread.resize(T);
        std::vector<Lane> lanes(T);
        const unsigned long long r_end = (unsigned long long)pl.ticks() * T;
        FU_CHECK(r_end < 0xFFFFFFF0ull, "turns of one image < 2^32");
        unsigned long long r_next = 0;
        long long P = p_first;
        while (r_next < r_end) {
          FU_CHECK(epoch < 0xFFFFFFFEu, "epoch counter");
          const long long tau0 = (long long)(r_next / T), tau_end = std::min(tau0 + P, pl.ticks());
          rstar = (unsigned long long)tau_end * T;
          std::vector<int> live;
          for (int t = 0; t < T; ++t) {
            Lane& ln = lanes[t];
            ln.rec.clear(); ln.walks.clear();
            ln.tau = tau0 + (t < (int)(r_next % T) ? 1 : 0);
            ln.nodes = ln.scans = 0;
            ln.stopped = false;
            live.push_back(t);
          }
          while (!live.empty()) {  // the waves run concurrently: any interleaving of their turns
            size_t pick;
            if (sched == 1) pick = live.size() - 1;
            else if (sched == 2) pick = 0;
            else { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; pick = (size_t)((lcg >> 33) % live.size()); }
            const int t = live[pick];
            Lane& ln = lanes[t];
            bool done = false;
            if (ln.tau >= tau_end) done = true;
            else {
              const unsigned long long rank = (unsigned long long)ln.tau * T + t;
              if (rank >= rstar) done = true;  // this turn cannot commit in this pass any more
              else {
                const int s_ = pl.seed(ln.tau, t);
                ++ln.scans;
                if (s_ < 0 || MaskedFor(word[I][s_], t) || images[I].depth_map[s_] <= 0.0f) ++ln.tau;
                else {
                  ++walks_run;
                  if (WalkSim(t, (unsigned)rank, I, s_, ln, cap, ln.tau)) ++ln.tau;
                  else done = true;
                }
              }
            }
            if (done) { live[pick] = live.back(); live.pop_back(); }
          }
          FU_CHECK(rstar > r_next, "pass made no progress");
          double worst = 0.0;
          for (int t = 0; t < T; ++t) {
            const Lane& ln = lanes[t];
            worst = std::max(worst, 3.0 * ln.nodes + ln.scans / 64.0);
            for (const WalkRec& wr : ln.walks) {
              if ((unsigned long long)wr.tau * T + t < rstar) CommitWalk(ln, wr, &per_thread[t]);
              else ++walks_discarded;
            }
          }
          model_us += worst + 40.0;
          const bool cut = rstar < (unsigned long long)tau_end * T;
          if (cut) ++cuts;
          P = cut ? std::max(p_min, P / 2) : std::min(p_max, 2 * P);
          r_next = rstar;
          ++epoch;
          ++passes_run;
        }
      }
      fused[I] = 1;
    }
    Concatenate();
  }
};

long long g_rounds = 0, g_walks = 0, g_discarded = 0, g_conflicts = 0, g_cuts = 0;
double g_model_us = 0.0;

template <typename F>
int Guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
}

}  // namespace

#define FUO_API __attribute__((visibility("default")))
extern "C" {

FUO_API void fuo_options_init(fusion_options* o) {
  o->min_num_pixels = 5;
  o->max_num_pixels = 10000;
  o->max_traversal_depth = 100;
  o->check_num_images = 50;
  o->max_reproj_error = 2.0f;
  o->max_depth_error = 0.01f;
  o->max_normal_error = 10.0f;
  for (int i = 0; i < 3; ++i) {
    o->bbox_min[i] = -FLT_MAX;
    o->bbox_max[i] = FLT_MAX;
  }
  o->num_threads = -1;
}

FUO_API int fuo_options_check(const fusion_options* o) {
  if (!o) return 1;
  return (o->min_num_pixels >= 0 && o->min_num_pixels <= o->max_num_pixels && o->max_traversal_depth > 0 &&
          o->max_reproj_error >= 0 && o->max_depth_error >= 0 && o->max_normal_error >= 0 && o->check_num_images > 0)
             ? 0
             : 1;
}

FUO_API int fuo_run(int32_t mode, const fusion_options* options, int32_t num_images, const fusion_image* images,
                    const int32_t* overlap_ptr, const int32_t* overlap_idx, fusion_result** out) {
  if (out) *out = nullptr;
  fusion_result* r = nullptr;
  const int rc = Guard([&] {
    FU_CHECK(options && images && overlap_ptr && out, "null argument");
    FU_CHECK(fuo_options_check(options) == 0, "options.Check()");
    FU_CHECK(mode >= 0 && mode <= 3, "mode");
    FU_CHECK(num_images > 0, "num_images > 0");
    for (int i = 0; i < num_images; ++i) {
      FU_CHECK(overlap_ptr[i] <= overlap_ptr[i + 1], "overlap_ptr is monotone");
      for (int k = overlap_ptr[i]; k < overlap_ptr[i + 1]; ++k)
        FU_CHECK(overlap_idx && overlap_idx[k] >= 0 && overlap_idx[k] < num_images, "overlap index in range");
    }
    r = new fusion_result();
    Fuser fuser(*options, num_images, images, overlap_ptr, overlap_idx, r);
    fuser.mode = mode;
    fuser.Init();
    if (mode == 2) fuser.RunPasses();
    else fuser.Run();
    g_rounds = fuser.passes_run;
    g_walks = fuser.walks_run;
    g_discarded = fuser.walks_discarded;
    g_conflicts = fuser.conflicts;
    g_cuts = fuser.cuts;
    g_model_us = fuser.model_us;
  });
  if (rc != 0) {
    delete r;
    return rc;
  }
  *out = r;
  return 0;
}

FUO_API size_t fuo_num_points(const fusion_result* r) { return r ? r->rgb.size() / 3 : 0; }

FUO_API int fuo_get_points(const fusion_result* r, float* xyz_normal, uint8_t* rgb) {
  return Guard([&] {
    FU_CHECK(r, "null argument");
    if (xyz_normal && !r->xyz_normal.empty()) std::memcpy(xyz_normal, r->xyz_normal.data(), r->xyz_normal.size() * sizeof(float));
    if (rgb && !r->rgb.empty()) std::memcpy(rgb, r->rgb.data(), r->rgb.size());
  });
}

FUO_API int fuo_get_visibility(const fusion_result* r, int64_t* vis_ptr, int32_t* vis_idx, size_t* total) {
  return Guard([&] {
    FU_CHECK(r, "null argument");
    if (total) *total = r->vis_idx.size();
    if (vis_ptr) std::memcpy(vis_ptr, r->vis_ptr.data(), r->vis_ptr.size() * sizeof(int64_t));
    if (vis_idx && !r->vis_idx.empty()) std::memcpy(vis_idx, r->vis_idx.data(), r->vis_idx.size() * sizeof(int32_t));
  });
}

FUO_API void fuo_free(fusion_result* r) { delete r; }

// mode 2: passes, turns walked (committed + discarded), and what the simulation says about the schedule
FUO_API void fuo_last_rounds(long long* rounds, long long* walks) {
  *rounds = g_rounds;
  *walks = g_walks;
}
FUO_API void fuo_last_schedule(long long* discarded, long long* conflicts, long long* cuts, double* model_us) {
  *discarded = g_discarded;
  *conflicts = g_conflicts;
  *cuts = g_cuts;
  *model_us = g_model_us;
}

FUO_API const char* fuo_last_error(void) { return g_error.c_str(); }

}  // extern "C"
