// fusion_oracle.cpp -- TEST INFRASTRUCTURE ONLY (tests/, never linked or imported by the product).
//
// CPU checker of the depth-map fusion step (reference src/colmap/mvs/fusion.cc): the reference's
// algorithm restated statement by statement (StereoFusion::Run / Fuse, fusion.cc:253-320, 401-524),
// one pixel's turn after the other, every turn masking what it absorbs at once. Two modes:
//   mode 0  the pixels of an image take their turns in row-major order (the reference with
//           num_threads = 1). Pinned against a float32 Python restatement in tests/test_fusion.py.
//   mode 1  the turns follow the fixed pseudo-random seed order of colmap_amd/csrc/fusion.hip, with its
//           two documented differences: at most 1024 pixels per walk, and a neighbour projection
//           is range-tested as a float before the conversion to int. The HIP path (speculative
//           walks + claim words + commit rounds) must reproduce this sequential run bit for bit.
//   mode 2  a plain C++ simulation of fusion.hip's rounds (speculate / claim / commit, RunRounds below):
//           executable statement of why the parallel schedule equals mode 1; tests compare 2 == 1.
// All arithmetic float like the reference (Eigen::Vector3f / Matrix<float,3,4>), medians through
// colmap::Percentile (math/math.h:205-224). Build: oracle/Makefile (-ffp-contract=off).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
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

  // the seed order of fusion.hip: pixels by ascending MurmurHash3 finaliser of their index
  static std::vector<int> SeedOrder(int n_px) {
    auto hash = [](uint32_t v) {
      v ^= v >> 16; v *= 0x85EBCA6Bu; v ^= v >> 13; v *= 0xC2B2AE35u; v ^= v >> 16;
      return v;
    };
    std::vector<int> order(n_px);
    for (int i = 0; i < n_px; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hash((uint32_t)a) < hash((uint32_t)b); });
    return order;
  }

  void Run() {  // fusion.cc:253-320, one thread
    for (int image_idx = 0; image_idx >= 0; image_idx = FindNextImage(image_idx)) {
      if (used[image_idx]) {
        const int width = images[image_idx].depth_width, height = images[image_idx].depth_height;
        if (mode == 0) {
          for (int row = 0; row < height; ++row)
            for (int col = 0; col < width; ++col) Fuse(image_idx, row, col);
        } else {
          for (const int s_ : SeedOrder(width * height)) Fuse(image_idx, s_ / width, s_ % width);
        }
      }
      fused[image_idx] = 1;
    }
  }

  void Fuse(int image_idx0, int row0, int col0) {  // fusion.cc:401-524
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
    const size_t max_pixels = mode == 0 ? (size_t)opt.max_num_pixels : (size_t)std::min(opt.max_num_pixels, kRecordCap);

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
      if (mode != 0 && recorded >= kRecordCap) break;
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
        if (mode == 0) {
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

    Emit(px, py, pz, nx, ny, nz, cr, cg, cb, vis_order, out);
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
  // mode 2: the round schedule of fusion.hip, simulated
  // -------------------------------------------------------------------------------------------
  struct Rec {
    int image, pix;
    float xyz[3], normal[3];
    uint8_t color[3];
    bool in_box;
  };
  struct WalkOut {
    std::vector<Rec> recs;
    bool capped = false, overflow = false;
  };

  // One walk against stamps (0 free, else the round that masked the pixel; masked = stamp < round).
  WalkOut Walk(int I, int seed, const std::vector<std::vector<unsigned>>& stamp, unsigned round, bool closure) const {
    WalkOut w;
    std::vector<FusionData> queue;
    queue.push_back({I, seed / images[I].depth_width, seed % images[I].depth_width, 0});
    float ref_point[4] = {0, 0, 0, 0}, ref_normal[3] = {0, 0, 0};
    const int kRecordCap = RecordCapacity(opt.max_num_pixels);
    const size_t max_pixels = (size_t)std::min(opt.max_num_pixels, kRecordCap);
    size_t n_in = 0;
    while (!queue.empty()) {
      const FusionData d = queue.back();
      queue.pop_back();
      const fusion_image& im = images[d.image_idx];
      const int pix = d.row * im.depth_width + d.col;
      const unsigned st = stamp[d.image_idx][pix];
      if (st != 0 && st < round) continue;
      bool seen = false;
      for (const Rec& q : w.recs) seen |= q.image == d.image_idx && q.pix == pix;
      if (seen) continue;
      const float depth = im.depth_map[pix];
      if (depth <= 0.0f) continue;
      const float* Pi = &P[12 * (size_t)d.image_idx];
      if (d.traversal_depth > 0) {
        float proj[3];
        for (int r = 0; r < 3; ++r)
          proj[r] = Pi[4 * r] * ref_point[0] + Pi[4 * r + 1] * ref_point[1] + Pi[4 * r + 2] * ref_point[2] +
                    Pi[4 * r + 3] * ref_point[3];
        const float depth_error = std::abs((proj[2] - depth) / depth);
        if (depth_error > opt.max_depth_error) continue;
        const float col_diff = proj[0] / proj[2] - d.col;
        const float row_diff = proj[1] / proj[2] - d.row;
        if (col_diff * col_diff + row_diff * row_diff > max_squared_reproj_error) continue;
      }
      const size_t slice = (size_t)im.depth_width * im.depth_height;
      const float nl[3] = {im.normal_map[pix], im.normal_map[slice + pix], im.normal_map[2 * slice + pix]};
      const float* iR = &inv_R[9 * (size_t)d.image_idx];
      Rec q;
      q.image = d.image_idx;
      q.pix = pix;
      for (int r = 0; r < 3; ++r) q.normal[r] = iR[3 * r] * nl[0] + iR[3 * r + 1] * nl[1] + iR[3 * r + 2] * nl[2];
      if (d.traversal_depth > 0) {
        const float c = ref_normal[0] * q.normal[0] + ref_normal[1] * q.normal[1] + ref_normal[2] * q.normal[2];
        if (c < min_cos_normal_error) continue;
      }
      const float* iP = &inv_P[12 * (size_t)d.image_idx];
      const float hx = d.col * depth, hy = d.row * depth;
      for (int r = 0; r < 3; ++r) q.xyz[r] = iP[4 * r] * hx + iP[4 * r + 1] * hy + iP[4 * r + 2] * depth + iP[4 * r + 3] * 1.0f;
      q.color[0] = q.color[1] = q.color[2] = 0;
      if (im.rgb) {
        const int xx = static_cast<int>(std::round(static_cast<double>(d.col / scale[2 * d.image_idx])));
        const int yy = static_cast<int>(std::round(static_cast<double>(d.row / scale[2 * d.image_idx + 1])));
        if (xx >= 0 && yy >= 0 && xx < im.bitmap_width && yy < im.bitmap_height)
          std::memcpy(q.color, im.rgb + 3 * ((size_t)yy * im.bitmap_width + xx), 3);
      }
      q.in_box = !(q.xyz[0] < opt.bbox_min[0] || q.xyz[1] < opt.bbox_min[1] || q.xyz[2] < opt.bbox_min[2] ||
                   q.xyz[0] > opt.bbox_max[0] || q.xyz[1] > opt.bbox_max[1] || q.xyz[2] > opt.bbox_max[2]);
      if ((int)w.recs.size() >= kRecordCap) { w.capped = w.overflow = true; break; }
      w.recs.push_back(q);
      if (!q.in_box) continue;
      ++n_in;
      if (d.traversal_depth == 0) {
        ref_point[0] = q.xyz[0]; ref_point[1] = q.xyz[1]; ref_point[2] = q.xyz[2]; ref_point[3] = 1.0f;
        std::memcpy(ref_normal, q.normal, sizeof(ref_normal));
      }
      if (!closure && n_in >= max_pixels) { w.capped = true; break; }
      if (!closure && d.traversal_depth >= opt.max_traversal_depth - 1) { w.capped = true; continue; }
      for (int k = optr[d.image_idx]; k < optr[d.image_idx + 1]; ++k) {
        const int next = oidx[k];
        if (!used[next] || fused[next]) continue;
        const float* Pn = &P[12 * (size_t)next];
        float np[3];
        for (int r = 0; r < 3; ++r) np[r] = Pn[4 * r] * q.xyz[0] + Pn[4 * r + 1] * q.xyz[1] + Pn[4 * r + 2] * q.xyz[2] + Pn[4 * r + 3];
        const float fcol = std::round(np[0] / np[2]), frow = std::round(np[1] / np[2]);
        if (!(fcol >= 0.0f && frow >= 0.0f && fcol < static_cast<float>(images[next].depth_width) &&
              frow < static_cast<float>(images[next].depth_height)))
          continue;
        queue.push_back({next, static_cast<int>(frow), static_cast<int>(fcol), d.traversal_depth + 1});
      }
    }
    return w;
  }

  long long rounds_run = 0, walks_run = 0;

  void RunRounds() {
    std::vector<std::vector<unsigned>> stamp(n);
    std::vector<std::vector<unsigned long long>> claim(n);
    for (int i = 0; i < n; ++i) {
      if (!used[i]) continue;
      stamp[i].assign(masks[i].size(), 0u);
      claim[i].assign(masks[i].size(), 0ull);
      for (size_t k = 0; k < masks[i].size(); ++k) stamp[i][k] = masks[i][k] ? 1u : 0u;
    }
    unsigned round = 2;
    for (int I = 0; I >= 0; I = FindNextImage(I)) {
      if (used[I]) {
        const int n_px = images[I].depth_width * images[I].depth_height;
        const std::vector<int> order = SeedOrder(n_px);
        std::vector<int> rank_of(n_px);
        for (int k = 0; k < n_px; ++k) rank_of[order[k]] = k;
        std::vector<fusion_result> per_seed(n_px);
        std::vector<int> active;
        int offered = 0;
        const int head = std::min(1 << 15, std::max(256, n_px / 64));  // the schedule of fusion.hip (its default first chunk)
        for (; !active.empty() || offered < n_px; ++round) {
          auto key_of = [&](int seed) {
            return ((unsigned long long)round << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)rank_of[seed]);
          };
          unsigned barrier = 0xFFFFFFFFu;
          for (int seed : active) {  // speculate
            const unsigned long long key = key_of(seed);
            WalkOut w = Walk(I, seed, stamp, round, false);
            for (const Rec& q : w.recs) claim[q.image][q.pix] = std::max(claim[q.image][q.pix], key);
            if (w.capped) {
              WalkOut c = Walk(I, seed, stamp, round, true);
              for (const Rec& q : c.recs) claim[q.image][q.pix] = std::max(claim[q.image][q.pix], key);
              if (c.overflow) barrier = std::min(barrier, (unsigned)rank_of[seed]);
            }
          }
          std::vector<int> next;
          for (int seed : active) {  // commit
            const unsigned long long key = key_of(seed);
            WalkOut w = Walk(I, seed, stamp, round, false);
            bool mine = (unsigned)rank_of[seed] <= barrier;
            for (const Rec& q : w.recs) mine = mine && claim[q.image][q.pix] == key;
            if (!mine) { next.push_back(seed); continue; }
            std::vector<float> px, py, pz, nx, ny, nz;
            std::vector<uint8_t> cr, cg, cb;
            std::vector<int> vis;
            for (const Rec& q : w.recs) {
              stamp[q.image][q.pix] = round;  // reads of this round see stamp == round: still free
              if (!q.in_box) continue;
              px.push_back(q.xyz[0]); py.push_back(q.xyz[1]); pz.push_back(q.xyz[2]);
              nx.push_back(q.normal[0]); ny.push_back(q.normal[1]); nz.push_back(q.normal[2]);
              cr.push_back(q.color[0]); cg.push_back(q.color[1]); cb.push_back(q.color[2]);
              if (std::find(vis.begin(), vis.end(), q.image) == vis.end()) vis.push_back(q.image);
            }
            Emit(px, py, pz, nx, ny, nz, cr, cg, cb, vis, &per_seed[seed]);
          }
          FU_CHECK(active.empty() || next.size() < active.size(), "round made no progress");
          if (!active.empty()) {
            ++rounds_run;
            walks_run += (long long)active.size();
          }
          if (offered < n_px) {  // first turns in rank order: a small head, then doubling
            const int upto = std::min(n_px, std::max(offered + head, 2 * offered));
            for (int k = offered; k < upto; ++k) {
              const int seed = order[k];
              if (stamp[I][seed] != 0 || images[I].depth_map[seed] <= 0.0f) continue;
              next.push_back(seed);
            }
            offered = upto;
          }
          active.swap(next);
        }
        for (int k = 0; k < n_px; ++k) {  // output in rank order
          const fusion_result& r = per_seed[order[k]];
          if (r.rgb.empty()) continue;
          out->xyz_normal.insert(out->xyz_normal.end(), r.xyz_normal.begin(), r.xyz_normal.end());
          out->rgb.insert(out->rgb.end(), r.rgb.begin(), r.rgb.end());
          out->vis_idx.insert(out->vis_idx.end(), r.vis_idx.begin(), r.vis_idx.end());
          out->vis_ptr.push_back(static_cast<int64_t>(out->vis_idx.size()));
        }
      }
      fused[I] = 1;
    }
  }
};

long long g_rounds = 0, g_walks = 0;

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
    FU_CHECK(mode >= 0 && mode <= 2, "mode");
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
    if (mode == 2) fuser.RunRounds();
    else fuser.Run();
    g_rounds = fuser.rounds_run;
    g_walks = fuser.walks_run;
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

FUO_API void fuo_last_rounds(long long* rounds, long long* walks) {
  *rounds = g_rounds;
  *walks = g_walks;
}

FUO_API const char* fuo_last_error(void) { return g_error.c_str(); }

}  // extern "C"
