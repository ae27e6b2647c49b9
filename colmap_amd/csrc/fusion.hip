// fusion.hip -- depth-map fusion on the GPU behind include/colmap_amd_fusion.h.
//
// What it computes: StereoFusion::Run / Fuse of the reference (src/colmap/mvs/fusion.cc:253-524) --
// for every image I in FindNextImage order (:51-73), every pixel of I in turn walks the consistency
// graph (depth, reprojection and normal tests against the pixel it started from), masks what it
// absorbs and fuses it into one point. In the reference the turns within an image are taken by a
// thread pool, so the order is row-major only for num_threads = 1. Here the order is a fixed
// pseudo-random permutation of the pixels (ascending seed_hash) and the result is exactly the
// reference's algorithm run in that order -- but the turns are not executed one after the other:
//
//   speculate  every undecided pixel ("seed") walks against the masks committed so far and claims the
//              pixels it would absorb: 64-bit atomicMax on a per-pixel word  round << 32 | ~rank, so
//              the earliest seed of the order holds the claim. A walk that hit a cap (traversal depth,
//              max_num_pixels) also claims its closure, because under more masks it may take another
//              path but cannot leave the closure.
//   commit     the same walk again; a seed that holds the claim on every pixel it absorbs cannot be
//              affected by any seed before it (their walks only shrink when more pixels get masked), so
//              its turn is final: it stamps its pixels into the mask and fuses them -- Percentile-50
//              medians of position, normal and colour (math/math.h:205-234), minimum size, normal
//              length, sorted distinct images. Everything else goes into the next round; the first
//              undecided seed of the order always commits, typically almost all do.
//   compact    per image, a device scan in rank order writes the points in the order the sequential
//              algorithm would have produced them.
//
// The checker is oracle/fusion_oracle.cpp: mode 1 runs the reference's Fuse() sequentially in the same
// seed order (tests/test_fusion.py compares bit for bit), mode 0 in row-major order. All arithmetic is
// float / double as the reference writes it; the library is built with -ffp-contract=off.
//
// Data layout in HBM: every image's depth map, slice-major normal map and bitmap stay resident for
// the whole run (4 + 12 + 3 bytes per pixel), plus a 4-byte mask stamp and an 8-byte claim word per
// depth-map pixel of every image. The state of a walk (<= record_capacity(max_num_pixels) absorbed pixels,
// 1 024 .. 16 384 slots -- the reference's default of 10 000 is held in full: pixel, image, level,
// point, normal, colour; one expansion frame per expanded pixel) lives in global arrays laid out
// [slot][lane] so that the lanes of a wave touch consecutive addresses; a launch uses a fixed number
// of resident lanes that stride over the undecided seeds.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/colmap_amd_fusion.h"

#define FUSION_API __attribute__((visibility("default")))

extern "C" void pm_release_cached_memory(void);  // pm_api.cpp (same library)

namespace {

thread_local std::string g_error;

struct Fail : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define FU_CHECK(cond, msg)                                         \
  do {                                                              \
    if (!(cond)) throw Fail(std::string("Check failed: ") + (msg)); \
  } while (0)

#define FU_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) throw Fail(std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

// Pixels one walk can record = the lane state of a seed: max_num_pixels itself between 1 024 and 16 384 (the
// reference's default 10 000 is NOT clamped: 40 B x 10 000 x 32 768 lanes = 13 GB of the 288 GB), smaller
// options keep the 1 024-slot state, larger ones are clamped to 16 384 (21 GB). oracle/fusion_oracle.cpp mirrors it.
constexpr int kElemCapMin = 1024, kElemCapMax = 16384;
inline int record_capacity(int max_num_pixels) { return std::min(std::max(max_num_pixels, kElemCapMin), kElemCapMax); }
constexpr int kLanes = 1 << 15;  // resident lanes per launch (256 CUs x 2 waves)
constexpr int kBlock = 64;
constexpr int kRankCount = 128;  // medians: rank counting (staged in LDS) up to this many values, radix select above

struct DevImage {
  float P[12], inv_P[12], inv_R[9];
  float sx, sy;          // depth map size / model image size
  const float* depth;
  const float* normal;   // [3][dh][dw]
  const uint8_t* rgb;    // [bh][bw][3] or nullptr
  int dw, dh, bw, bh;
  long long pix_off;     // first mask / claim word of this image
  int pos;               // step at which the image is fused; -1: not used
};

struct Params {
  const DevImage* images;
  const int* optr;
  const int* oidx;
  unsigned* mask;               // per pixel: 0 free, 1 masked on input, else the round that absorbed it
  unsigned long long* claim;    // per pixel: round << 32 | ~priority of the best claimer of that round
  unsigned round;               // current round stamp (>= 2, grows over the whole run)
  int step;                     // position of `image` in the fusion order
  int image;
  int num_seeds;
  const int* rank_of;           // priority of a seed = its rank in the seed order
  const int* active;            // undecided seeds that have had their first turn offered
  int num_active;
  int* next_active;
  int* next_count;
  int* lane_walk;               // per lane: recorded pixels of the speculate walk | capped << 31 (state reuse)
  int lanes;                    // lanes of a launch = columns of the lane state: min(kLanes, seeds of the largest image)
  int reuse;                    // num_active <= lanes: a lane keeps its walk from speculate to commit
  unsigned* barrier;            // lowest priority value among seeds whose closure overflowed the record
  int elem_cap;                 // min(max_num_pixels, rec_cap)
  int rec_cap;                  // record_capacity(max_num_pixels): slots of the lane state
  int max_level;                // max_traversal_depth - 1
  int min_num_pixels;
  double max_depth_error;
  float max_sq_reproj, min_cos_normal;
  float bmin[3], bmax[3];
  // lane state [slot][lanes]
  unsigned *e_pix, *e_meta, *e_rgb, *frame;
  float *e_x, *e_y, *e_z, *e_nx, *e_ny, *e_nz;
  // per-seed outputs
  int *valid, *nvis, *vis_off;
  float* pt;            // [num_seeds][6]
  unsigned char* col;   // [num_seeds][3]
  int* pool;            // visibility lists, allocated with an atomic cursor
  long long pool_cap;   // entries; the host checks the cursor against it after every image
  unsigned long long* pool_cursor;
};

// meta word of a recorded pixel: image (16 bits) | level (15 bits) << 16 | in-box << 31
__device__ inline unsigned pack_meta(int image, int level, bool in_box) {
  return (unsigned)image | ((unsigned)level << 16) | (in_box ? 0x80000000u : 0u);
}

__device__ inline unsigned long long claim_key(unsigned round, unsigned prio) {
  return ((unsigned long long)round << 32) | (unsigned long long)(0xFFFFFFFFu - prio);
}

#define SLOT(buf, e) buf[(size_t)(e) * p.lanes + lane]

// k-th smallest (0-based) of m floats: MSB-first radix select on the order-preserving integer key
template <typename Get>
__device__ float radix_select(int m, int k, Get get) {
  unsigned prefix = 0u, care = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    care |= 1u << bit;
    int zeros = 0;
    for (int a = 0; a < m; ++a) {
      const unsigned u = __float_as_uint(get(a));
      const unsigned key = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
      zeros += ((key & care) == prefix);
    }
    if (k >= zeros) {
      k -= zeros;
      prefix |= 1u << bit;
    }
  }
  const unsigned u = (prefix >> 31) ? (prefix ^ 0x80000000u) : ~prefix;
  return __uint_as_float(u);
}

// colmap::Percentile(values, 50) of m values (math/math.h:205-234): the two middle order statistics,
// interpolated in double like the reference.
// `col` is the lane's private column of a [kRankCount][kBlock] LDS array: up to kRankCount values are staged there
// once and ranked out of LDS (m^2 reads of ~64 cycles instead of m^2 global loads).
template <typename Get>
__device__ double median_of(int m, Get get, float* col) {
  const double idx = 0.5 * (double)(m - 1);
  const double lf = floor(idx), rc = ceil(idx);
  const int li = (int)lf, ri = (int)rc;
  double left = 0.0, right = 0.0;
  if (m <= kRankCount) {
    for (int a = 0; a < m; ++a) col[a * kBlock] = get(a);
    for (int a = 0; a < m; ++a) {
      const float v = col[a * kBlock];
      int lt = 0, le = 0;
      for (int b = 0; b < m; ++b) {
        const float w = col[b * kBlock];
        lt += w < v;
        le += w <= v;
      }
      if (lt <= li && li < le) left = (double)v;
      if (lt <= ri && ri < le) right = (double)v;
    }
  } else {
    right = (double)radix_select(m, ri, get);
    left = li == ri ? right : (double)radix_select(m, li, get);
  }
  if (li == ri) return right;
  return (rc - idx) * left + (idx - lf) * right;
}

struct Walk {
  int ne;         // recorded (absorbed) pixels, bounding-box rejects included
  bool capped;    // a cap of the reference's walk was hit (traversal depth, max_num_pixels, record size)
  bool overflow;  // CLOSURE: the record is full, the closure is not known
};

// StereoFusion::Fuse's traversal (fusion.cc:401-489) of `seed` against the masks committed before
// round p.round. The reference's stack of expanded neighbours is kept as one frame per expanded pixel
// (pixel slot, next overlap entry to try, counted down): the same depth-first order without
// materialising the neighbours. CLOSURE: ignore max_traversal_depth and max_num_pixels -- everything
// the seed could absorb under any superset of the current masks. CLAIM: atomicMax the claim word of
// every recorded pixel. ATTR: also record normal and colour (for the fuse step).
// FAST (the first claiming walk of a seed in a round): "did this walk absorb the pixel already?" is read off
// the claim word -- equal to the walk's key: yes; below it (older round / later seed): no, since the walk's own
// claim would have raised it; above it (an earlier seed of the order holds the pixel): undecided, look
// through the record. The word is read past the L1 (the claims are L2 atomics).
template <bool CLOSURE, bool CLAIM, bool ATTR, bool FAST = false>
__device__ Walk walk(const Params& p, int lane, int seed, unsigned long long key) {
  Walk w{0, false, false};
  const DevImage& I0 = p.images[p.image];
  int ne = 0, nin = 0, nf = 0;
  float ref[3] = {0.f, 0.f, 0.f}, refn[3] = {0.f, 0.f, 0.f};
  int img = p.image, row = seed / I0.dw, col = seed % I0.dw, level = 0;
  bool have = true;
  while (have) {
    do {  // ---- visit (img, row, col, level): fusion.cc:416-472 ----
      const DevImage& im = p.images[img];
      const int pix = row * im.dw + col;
      const unsigned mk = p.mask[im.pix_off + pix];
      if (mk != 0u && mk < p.round) break;  // masked before this round
      bool seen = false, scan = true;
      if (FAST) {
        const unsigned long long w0 = __hip_atomic_load(p.claim + im.pix_off + pix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = w0 == key;
        scan = w0 > key;
      }
      if (scan) {
        int e = 0;
        for (; e + 4 <= ne; e += 4) {  // four record entries per round trip
          const unsigned p0 = SLOT(p.e_pix, e), p1 = SLOT(p.e_pix, e + 1), p2 = SLOT(p.e_pix, e + 2), p3 = SLOT(p.e_pix, e + 3);
          const unsigned m0 = SLOT(p.e_meta, e), m1 = SLOT(p.e_meta, e + 1), m2 = SLOT(p.e_meta, e + 2), m3 = SLOT(p.e_meta, e + 3);
          seen |= (p0 == (unsigned)pix && (int)(m0 & 0xFFFFu) == img) | (p1 == (unsigned)pix && (int)(m1 & 0xFFFFu) == img) |
                  (p2 == (unsigned)pix && (int)(m2 & 0xFFFFu) == img) | (p3 == (unsigned)pix && (int)(m3 & 0xFFFFu) == img);
        }
        for (; e < ne; ++e) seen |= SLOT(p.e_pix, e) == (unsigned)pix && (int)(SLOT(p.e_meta, e) & 0xFFFFu) == img;
      }
      if (seen) break;  // masked by this walk
      const float depth = im.depth[pix];
      if (depth <= 0.0f) break;
      if (level > 0) {
        float proj[3];
        for (int r = 0; r < 3; ++r)
          proj[r] = im.P[4 * r] * ref[0] + im.P[4 * r + 1] * ref[1] + im.P[4 * r + 2] * ref[2] + im.P[4 * r + 3] * 1.0f;
        const float depth_error = fabsf((proj[2] - depth) / depth);
        if ((double)depth_error > p.max_depth_error) break;
        const float col_diff = proj[0] / proj[2] - (float)col;
        const float row_diff = proj[1] / proj[2] - (float)row;
        if (col_diff * col_diff + row_diff * row_diff > p.max_sq_reproj) break;
      }
      const size_t slice = (size_t)im.dw * im.dh;
      const float nl0 = im.normal[pix], nl1 = im.normal[slice + pix], nl2 = im.normal[2 * slice + pix];
      float nrm[3];
      for (int r = 0; r < 3; ++r) nrm[r] = im.inv_R[3 * r] * nl0 + im.inv_R[3 * r + 1] * nl1 + im.inv_R[3 * r + 2] * nl2;
      if (level > 0) {
        const float c = refn[0] * nrm[0] + refn[1] * nrm[1] + refn[2] * nrm[2];
        if (c < p.min_cos_normal) break;
      }
      const float hx = (float)col * depth, hy = (float)row * depth;
      float xyz[3];
      for (int r = 0; r < 3; ++r)
        xyz[r] = im.inv_P[4 * r] * hx + im.inv_P[4 * r + 1] * hy + im.inv_P[4 * r + 2] * depth + im.inv_P[4 * r + 3] * 1.0f;
      const bool in_box = !(xyz[0] < p.bmin[0] || xyz[1] < p.bmin[1] || xyz[2] < p.bmin[2] || xyz[0] > p.bmax[0] ||
                            xyz[1] > p.bmax[1] || xyz[2] > p.bmax[2]);
      if (ne >= p.rec_cap) {  // record capacity: the walk ends
        w.capped = true; w.overflow = true; nf = 0;
        break;
      }
      SLOT(p.e_pix, ne) = (unsigned)pix;
      SLOT(p.e_meta, ne) = pack_meta(img, level, in_box);
      SLOT(p.e_x, ne) = xyz[0]; SLOT(p.e_y, ne) = xyz[1]; SLOT(p.e_z, ne) = xyz[2];
      if (ATTR) {
        unsigned rgb = 0u;
        if (im.rgb) {  // nearest neighbour at the bitmap scale (bitmap.cc:329-334), colour 0 outside
          const int xx = (int)round((double)((float)col / im.sx));
          const int yy = (int)round((double)((float)row / im.sy));
          if (xx >= 0 && yy >= 0 && xx < im.bw && yy < im.bh) {
            const uint8_t* c3 = im.rgb + 3 * ((size_t)yy * im.bw + xx);
            rgb = (unsigned)c3[0] | ((unsigned)c3[1] << 8) | ((unsigned)c3[2] << 16);
          }
        }
        SLOT(p.e_nx, ne) = nrm[0]; SLOT(p.e_ny, ne) = nrm[1]; SLOT(p.e_nz, ne) = nrm[2];
        SLOT(p.e_rgb, ne) = rgb;
      }
      if (CLAIM) atomicMax(p.claim + im.pix_off + pix, key);
      ++ne;
      if (!in_box) break;
      ++nin;
      if (level == 0) {
        ref[0] = xyz[0]; ref[1] = xyz[1]; ref[2] = xyz[2];
        refn[0] = nrm[0]; refn[1] = nrm[1]; refn[2] = nrm[2];
      }
      if (!CLOSURE && nin >= p.elem_cap) {  // max_num_pixels reached (fusion.cc:470-472)
        w.capped = true; nf = 0;
        break;
      }
      if (!CLOSURE && level >= p.max_level) {
        w.capped = true;
        break;
      }
      if (level >= 32766) { w.capped = true; w.overflow = true; nf = 0; break; }
      SLOT(p.frame, nf) = (unsigned)(ne - 1) | ((unsigned)(p.optr[img + 1] - p.optr[img]) << 12);
      ++nf;
    } while (false);
    // ---- next node: top frame, neighbours in reverse list order (fusion.cc:474-488) ----
    have = false;
    while (nf > 0 && !have) {
      const unsigned f = SLOT(p.frame, nf - 1);
      const int e = (int)(f & 0xFFFu);
      int k = (int)(f >> 12);
      if (k == 0) { --nf; continue; }
      --k;
      SLOT(p.frame, nf - 1) = (unsigned)e | ((unsigned)k << 12);
      const unsigned meta = SLOT(p.e_meta, e);
      const int pimg = (int)(meta & 0xFFFFu);
      const int next = p.oidx[p.optr[pimg] + k];
      const DevImage& nx = p.images[next];
      if (nx.pos < p.step) continue;  // not used (-1) or fused in an earlier step
      const float x = SLOT(p.e_x, e), y = SLOT(p.e_y, e), z = SLOT(p.e_z, e);
      float np[3];
      for (int r = 0; r < 3; ++r) np[r] = nx.P[4 * r] * x + nx.P[4 * r + 1] * y + nx.P[4 * r + 2] * z + nx.P[4 * r + 3];
      const float fcol = roundf(np[0] / np[2]), frow = roundf(np[1] / np[2]);
      if (!(fcol >= 0.0f && frow >= 0.0f && fcol < (float)nx.dw && frow < (float)nx.dh)) continue;
      img = next; row = (int)frow; col = (int)fcol; level = (int)((meta >> 16) & 0x7FFFu) + 1;
      have = true;
    }
  }
  w.ne = ne;
  return w;
}

__device__ inline int seed_of(const Params& p, int idx) { return p.active[idx]; }
__device__ inline unsigned prio_of(const Params& p, int seed) { return (unsigned)p.rank_of[seed]; }

// Round, first half: every undecided seed walks against the committed masks and claims what it would
// absorb. A seed whose walk hit a cap also claims its closure: under more masks a capped walk can take
// another path, but never leaves the closure. A closure that does not fit the record holds back every
// seed after it in the order (barrier).
__global__ void __launch_bounds__(kBlock) fusion_speculate_kernel(Params p) {
  const int lane = blockIdx.x * kBlock + threadIdx.x;
  for (int idx = lane; idx < p.num_active; idx += p.lanes) {
    const int seed = seed_of(p, idx);
    const unsigned prio = prio_of(p, seed);
    const unsigned long long key = claim_key(p.round, prio);
    const Walk w = p.reuse ? walk<false, true, true, true>(p, lane, seed, key) : walk<false, true, false, true>(p, lane, seed, key);
    if (w.capped) {
      const Walk c = walk<true, true, false>(p, lane, seed, key);
      if (c.overflow) atomicMin(p.barrier, prio);
    }
    if (p.reuse) p.lane_walk[lane] = w.ne | (w.capped ? (int)0x80000000 : 0);  // capped: the closure walk overwrote the state
  }
}

// Round, second half: the same walk again (deterministic: masks written in this round carry this
// round's stamp and read as free). A seed that holds the claim of every pixel it absorbs -- no seed
// before it in the order can take any of them, now or after its own re-walk -- is final: it masks
// its pixels and fuses them (fusion.cc:491-523). The others wait for the next round.
__global__ void __launch_bounds__(kBlock) fusion_commit_kernel(Params p) {
  __shared__ float stage[kRankCount * kBlock];
  float* col = stage + threadIdx.x;
  const int lane = blockIdx.x * kBlock + threadIdx.x;
  const unsigned barrier = *p.barrier;
  for (int idx = lane; idx < p.num_active; idx += p.lanes) {
    const int seed = seed_of(p, idx);
    const unsigned prio = prio_of(p, seed);
    const unsigned long long key = claim_key(p.round, prio);
    int ne;
    if (p.reuse && p.lane_walk[lane] >= 0) ne = p.lane_walk[lane];
    else ne = walk<false, false, true>(p, lane, seed, key).ne;
    bool mine = prio <= barrier;
    for (int e = 0; e < ne && mine; ++e) {
      const DevImage& im = p.images[SLOT(p.e_meta, e) & 0xFFFFu];
      mine = p.claim[im.pix_off + SLOT(p.e_pix, e)] == key;
    }
    if (!mine) {
      p.next_active[atomicAdd(p.next_count, 1)] = seed;
      continue;
    }
    int m = 0;  // in-box pixels, their slots compacted into `frame`
    for (int e = 0; e < ne; ++e) {
      const unsigned meta = SLOT(p.e_meta, e);
      p.mask[p.images[meta & 0xFFFFu].pix_off + SLOT(p.e_pix, e)] = p.round;
      if (!(meta >> 31)) continue;
      SLOT(p.frame, m) = (unsigned)e;
      ++m;
    }
    if (m < p.min_num_pixels || m == 0) continue;
    const float fnx = (float)median_of(m, [&](int a) { return SLOT(p.e_nx, SLOT(p.frame, a)); }, col);
    const float fny = (float)median_of(m, [&](int a) { return SLOT(p.e_ny, SLOT(p.frame, a)); }, col);
    const float fnz = (float)median_of(m, [&](int a) { return SLOT(p.e_nz, SLOT(p.frame, a)); }, col);
    const float norm = sqrtf(fnx * fnx + fny * fny + fnz * fnz);
    if (norm < FLT_EPSILON) continue;
    float* out = p.pt + 6 * (size_t)seed;
    out[0] = (float)median_of(m, [&](int a) { return SLOT(p.e_x, SLOT(p.frame, a)); }, col);
    out[1] = (float)median_of(m, [&](int a) { return SLOT(p.e_y, SLOT(p.frame, a)); }, col);
    out[2] = (float)median_of(m, [&](int a) { return SLOT(p.e_z, SLOT(p.frame, a)); }, col);
    out[3] = fnx / norm; out[4] = fny / norm; out[5] = fnz / norm;
    for (int ch = 0; ch < 3; ++ch) {
      const float v = roundf((float)median_of(m, [&](int a) {
        return (float)((SLOT(p.e_rgb, SLOT(p.frame, a)) >> (8 * ch)) & 0xFFu);
      }, col));
      p.col[3 * (size_t)seed + ch] = (unsigned char)fminf(255.0f, fmaxf(0.0f, v));
    }
    // distinct images, ascending (the reference copies an unordered set)
    int nvis = 0;
    for (int last = -1;;) {
      int best = 0x7FFFFFFF;
      for (int a = 0; a < m; ++a) {
        const int ia = (int)(SLOT(p.e_meta, SLOT(p.frame, a)) & 0xFFFFu);
        if (ia > last && ia < best) best = ia;
      }
      if (best == 0x7FFFFFFF) break;
      last = best;
      ++nvis;
    }
    const unsigned long long off64 = atomicAdd(p.pool_cursor, (unsigned long long)nvis);
    const bool fits = off64 + (unsigned long long)nvis <= (unsigned long long)p.pool_cap;  // else: the host fails the run
    const int off = fits ? (int)off64 : 0;
    int last = -1;
    for (int v = 0; v < nvis; ++v) {
      int best = 0x7FFFFFFF;
      for (int a = 0; a < m; ++a) {
        const int ia = (int)(SLOT(p.e_meta, SLOT(p.frame, a)) & 0xFFFFu);
        if (ia > last && ia < best) best = ia;
      }
      if (fits) p.pool[off + v] = best;
      last = best;
    }
    p.vis_off[seed] = off;
    p.nvis[seed] = nvis;
    p.valid[seed] = 1;
  }
}
#undef SLOT

// The order in which the pixels of an image take their turn: ascending hash of the pixel index
// (fmix32 of MurmurHash3, a bijection of the 32-bit integers, so the keys are distinct). The
// reference's row-major order is only one of the orders its thread pool can produce; a pseudo-random
// one keeps walks that compete for the same pixels from forming long chains of decreasing rank.
__host__ __device__ inline unsigned seed_hash(unsigned s) {
  s ^= s >> 16; s *= 0x85EBCA6Bu; s ^= s >> 13; s *= 0xC2B2AE35u; s ^= s >> 16;
  return s;
}

__global__ void fusion_keys_kernel(int num_seeds, unsigned* __restrict__ keys, int* __restrict__ seeds) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= num_seeds) return;
  keys[s] = seed_hash((unsigned)s);
  seeds[s] = s;
}

__global__ void fusion_invert_kernel(int num_seeds, const int* __restrict__ order, int* __restrict__ rank_of) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < num_seeds) rank_of[order[k]] = k;
}

// The seeds of ranks [k0, k1) get their first turn: appended to the undecided list unless their own
// pixel is masked already (then their turn is empty) or has no depth.
__global__ void fusion_offer_kernel(Params p, const int* __restrict__ order, int k0, int k1) {
  const int k = k0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= k1) return;
  const int seed = order[k];
  const DevImage& im = p.images[p.image];
  if (p.mask[im.pix_off + seed] != 0u || im.depth[seed] <= 0.0f) return;
  p.next_active[atomicAdd(p.next_count, 1)] = seed;
}

// seed order -> output order: entry k is the seed of rank k
__global__ void fusion_rank_kernel(int num_seeds, const int* __restrict__ order, const int* __restrict__ valid,
                                   const int* __restrict__ nvis, int* __restrict__ valid_r, int* __restrict__ nvis_r) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > num_seeds) return;
  if (k == num_seeds) { valid_r[k] = 0; nvis_r[k] = 0; return; }  // the scans run over num_seeds + 1 entries
  const int s = order[k];
  valid_r[k] = valid[s];
  nvis_r[k] = valid[s] ? nvis[s] : 0;
}

__global__ void fusion_compact_kernel(int num_seeds, const int* __restrict__ order, const int* __restrict__ valid_r,
                                      const int* __restrict__ scan_valid, const int* __restrict__ nvis_r,
                                      const int* __restrict__ scan_vis, const int* __restrict__ vis_off,
                                      const int* __restrict__ pool, const float* __restrict__ pt,
                                      const unsigned char* __restrict__ col, float* __restrict__ out_pt,
                                      unsigned char* __restrict__ out_col, int* __restrict__ out_nvis,
                                      int* __restrict__ out_vis) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_seeds || !valid_r[k]) return;
  const int s = order[k];
  const int o = scan_valid[k];
  for (int c = 0; c < 6; ++c) out_pt[6 * (size_t)o + c] = pt[6 * (size_t)s + c];
  for (int c = 0; c < 3; ++c) out_col[3 * (size_t)o + c] = col[3 * (size_t)s + c];
  out_nvis[o] = nvis_r[k];
  for (int w = 0; w < nvis_r[k]; ++w) out_vis[scan_vis[k] + w] = pool[vis_off[s] + w];
}

__global__ void fusion_premask_kernel(size_t n, const unsigned char* __restrict__ in, unsigned* __restrict__ mask) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && in[i]) mask[i] = 1u;  // masked before the first round
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    release();
    n = count;
    hipError_t e_alloc = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    if (e_alloc == hipErrorOutOfMemory) {  // memory cached by the PatchMatch buffer pool is not "in use"
      (void)hipGetLastError();
      pm_release_cached_memory();
      e_alloc = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    }
    FU_HIP(e_alloc);
  }
  void upload(const T* h, size_t count) {
    alloc(count);
    if (count) FU_HIP(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  ~DevBuf() { release(); }
};

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

// top three rows of [P; 0 0 0 1]^-1 = [M^-1 | -M^-1 p], M^-1 by the adjugate
void ComposeInverseProjectionMatrix(const float P[12], float inv_P[12]) {
  const float a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[6], g = P[8], h = P[9], i = P[10];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  const float inv_det = 1.0f / det;
  const float Mi[9] = {A * inv_det, -(b * i - c * h) * inv_det, (b * f - c * e) * inv_det,
                       B * inv_det, (a * i - c * g) * inv_det,  -(a * f - c * d) * inv_det,
                       C * inv_det, -(a * h - b * g) * inv_det, (a * e - b * d) * inv_det};
  for (int r = 0; r < 3; ++r) {
    for (int col = 0; col < 3; ++col) inv_P[4 * r + col] = Mi[3 * r + col];
    inv_P[4 * r + 3] = -(Mi[3 * r] * P[3] + Mi[3 * r + 1] * P[7] + Mi[3 * r + 2] * P[11]);
  }
}

struct Stats {
  long long images = 0, seeds = 0, rounds = 0, walks = 0;
  double upload_seconds = 0.0, device_seconds = 0.0;  // host maps -> HBM + workspace setup | rounds, medians, compaction, read-back
};
Stats g_stats;

template <typename F>
int Guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const Fail& e) {
    g_error = e.what();
    return 1;
  } catch (const std::exception& e) {
    g_error = std::string("internal error: ") + e.what();
    return 2;
  }
}

}  // namespace

struct fusion_result {
  std::vector<float> xyz_normal;
  std::vector<uint8_t> rgb;
  std::vector<int64_t> vis_ptr{0};
  std::vector<int32_t> vis_idx;
};

namespace {

void Run(const fusion_options& opt, int n, const fusion_image* images, const int32_t* optr, const int32_t* oidx,
         fusion_result* out) {
  const auto t_begin = std::chrono::steady_clock::now();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw Fail("no HIP device: the fusion kernels need a GPU (there is no CPU path)");
  FU_CHECK(n < 65536, "at most 65535 images");
  FU_CHECK(opt.max_traversal_depth <= 32767, "max_traversal_depth <= 32767");
  // fusion order (FindNextImage, fusion.cc:51-73): depends on the overlap lists only
  std::vector<char> used(n, 0), fused(n, 0);
  for (int i = 0; i < n; ++i) {
    const fusion_image& im = images[i];
    if (!im.used) continue;
    FU_CHECK(im.depth_map && im.normal_map && im.depth_width > 0 && im.depth_height > 0, "depth / normal map");
    FU_CHECK(im.width > 0 && im.height > 0, "image size");
    FU_CHECK((int64_t)im.depth_width * im.depth_height < (1ll << 31), "depth map size");
    used[i] = 1;
  }
  std::vector<int> order_of_images, pos(n, -1);
  if (n > 0) {
    for (int cur = 0; cur >= 0;) {
      if (used[cur]) {
        pos[cur] = (int)order_of_images.size();
        order_of_images.push_back(cur);
      }
      fused[cur] = 1;
      int nxt = -1;
      for (int k = optr[cur]; k < optr[cur + 1] && nxt < 0; ++k)
        if (used[oidx[k]] && !fused[oidx[k]]) nxt = oidx[k];
      for (int i = 0; i < n && nxt < 0; ++i)
        if (used[i] && !fused[i]) nxt = i;
      cur = nxt;
    }
  }
  if (order_of_images.empty()) return;

  // resident maps + descriptors
  std::vector<DevImage> h_img(n);
  std::vector<DevBuf<float>> d_depth(n), d_normal(n);
  std::vector<DevBuf<uint8_t>> d_rgb(n);
  long long total_pix = 0;
  int max_seeds = 0;
  for (int i = 0; i < n; ++i) {
    DevImage& d = h_img[i];
    std::memset(&d, 0, sizeof(d));
    d.pos = pos[i];
    if (!used[i]) continue;
    const fusion_image& im = images[i];
    const size_t npix = (size_t)im.depth_width * im.depth_height;
    d.sx = static_cast<float>(im.depth_width) / im.width;
    d.sy = static_cast<float>(im.depth_height) / im.height;
    float K[9];
    std::memcpy(K, im.K, sizeof(K));
    K[0] *= d.sx; K[2] *= d.sx;
    K[4] *= d.sy; K[5] *= d.sy;
    ComposeProjectionMatrix(K, im.R, im.T, d.P);
    ComposeInverseProjectionMatrix(d.P, d.inv_P);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) d.inv_R[3 * r + c] = im.R[3 * c + r];
    d_depth[i].upload(im.depth_map, npix);
    d_normal[i].upload(im.normal_map, 3 * npix);
    d.depth = d_depth[i].p;
    d.normal = d_normal[i].p;
    if (im.rgb) {
      FU_CHECK(im.bitmap_width > 0 && im.bitmap_height > 0, "bitmap size");
      d_rgb[i].upload(im.rgb, 3 * (size_t)im.bitmap_width * im.bitmap_height);
      d.rgb = d_rgb[i].p;
    }
    d.dw = im.depth_width; d.dh = im.depth_height; d.bw = im.bitmap_width; d.bh = im.bitmap_height;
    d.pix_off = total_pix;
    total_pix += (long long)npix;
    max_seeds = std::max(max_seeds, (int)npix);
  }
  // mask / claim words are indexed with 64-bit offsets: no limit on the workspace size. The visibility pool is
  // refilled per reference image (cursor reset every step) and its int offsets only have to cover what ONE
  // image's walks absorb: capacity min(total pixels, 2^31 - 1), an overflow fails the run instead of wrapping.
  const long long pool_cap = std::min<long long>(total_pix, 0x7FFFFFFFll);
  DevBuf<DevImage> d_img;
  d_img.upload(h_img.data(), h_img.size());
  DevBuf<int> d_optr, d_oidx;
  d_optr.upload(optr, (size_t)n + 1);
  d_oidx.upload(oidx, (size_t)optr[n]);
  for (int i = 0; i < n; ++i)
    for (int k = optr[i]; k < optr[i + 1]; ++k) FU_CHECK(oidx[k] >= 0 && oidx[k] < n, "overlap index");
  for (int i = 0; i < n; ++i) FU_CHECK(optr[i + 1] - optr[i] < (1 << 20), "overlap list length");

  DevBuf<unsigned> d_mask, d_barrier;
  DevBuf<unsigned long long> d_claim, d_cursor;
  DevBuf<int> d_next_count;
  d_mask.alloc((size_t)total_pix);
  d_claim.alloc((size_t)total_pix);
  FU_HIP(hipMemset(d_mask.p, 0, sizeof(unsigned) * (size_t)total_pix));
  FU_HIP(hipMemset(d_claim.p, 0, sizeof(unsigned long long) * (size_t)total_pix));
  d_cursor.alloc(1); d_barrier.alloc(1); d_next_count.alloc(1);
  for (int i = 0; i < n; ++i) {
    if (!used[i] || !images[i].mask) continue;
    const size_t npix = (size_t)images[i].depth_width * images[i].depth_height;
    DevBuf<uint8_t> m;
    m.upload(images[i].mask, npix);
    hipLaunchKernelGGL(fusion_premask_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, npix, m.p,
                       d_mask.p + h_img[i].pix_off);
    FU_HIP(hipDeviceSynchronize());
  }

  Params p;
  std::memset(&p, 0, sizeof(p));
  p.images = d_img.p; p.optr = d_optr.p; p.oidx = d_oidx.p; p.mask = d_mask.p; p.claim = d_claim.p;
  // Lane state: a walk cannot record more pixels than the workspace has, and no more lanes than the largest
  // image has seeds are ever active -- the state is sized by both (a 4 x 48 x 36 workspace takes 2 MB, not
  // the 13 GB of 10 000 slots x 32 768 lanes).
  p.rec_cap = (int)std::min<long long>(record_capacity(opt.max_num_pixels), std::max<long long>(total_pix, 1));
  p.elem_cap = std::min(opt.max_num_pixels, p.rec_cap);
  p.lanes = std::min(kLanes, (max_seeds + kBlock - 1) / kBlock * kBlock);
  p.max_level = opt.max_traversal_depth - 1;
  p.min_num_pixels = opt.min_num_pixels;
  p.max_depth_error = opt.max_depth_error;
  p.max_sq_reproj = static_cast<float>(opt.max_reproj_error * opt.max_reproj_error);
  p.min_cos_normal = static_cast<float>(std::cos(opt.max_normal_error * 0.017453292519943295769));
  for (int c = 0; c < 3; ++c) { p.bmin[c] = opt.bbox_min[c]; p.bmax[c] = opt.bbox_max[c]; }
  DevBuf<unsigned> e_pix, e_meta, e_rgb, frame;
  DevBuf<float> e_f[6];
  const size_t state = (size_t)p.rec_cap * p.lanes;
  e_pix.alloc(state); e_meta.alloc(state); e_rgb.alloc(state); frame.alloc(state);
  for (auto& b : e_f) b.alloc(state);
  p.e_pix = e_pix.p; p.e_meta = e_meta.p; p.e_rgb = e_rgb.p; p.frame = frame.p;
  p.e_x = e_f[0].p; p.e_y = e_f[1].p; p.e_z = e_f[2].p; p.e_nx = e_f[3].p; p.e_ny = e_f[4].p; p.e_nz = e_f[5].p;
  DevBuf<int> valid, nvis, vis_off, valid_r, nvis_r, scan_valid, scan_vis, pool, out_nvis, out_vis, list_a, list_b;
  DevBuf<float> pt, out_pt;
  DevBuf<unsigned char> col, out_col;
  const size_t ms = (size_t)max_seeds;
  valid.alloc(ms); nvis.alloc(ms); vis_off.alloc(ms); valid_r.alloc(ms + 1); nvis_r.alloc(ms + 1);
  scan_valid.alloc(ms + 1); scan_vis.alloc(ms + 1); list_a.alloc(ms); list_b.alloc(ms);
  pool.alloc((size_t)pool_cap); out_nvis.alloc(ms); out_vis.alloc((size_t)pool_cap);
  p.pool_cap = pool_cap;
  pt.alloc(6 * ms); out_pt.alloc(6 * ms); col.alloc(3 * ms); out_col.alloc(3 * ms);
  p.valid = valid.p; p.nvis = nvis.p; p.vis_off = vis_off.p; p.pt = pt.p; p.col = col.p; p.pool = pool.p;
  p.pool_cursor = d_cursor.p; p.barrier = d_barrier.p; p.next_count = d_next_count.p;
  size_t tmp_bytes = 0;
  FU_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, valid_r.p, scan_valid.p, (int)ms + 1));
  DevBuf<unsigned char> tmp;
  tmp.alloc(tmp_bytes + 16);

  DevBuf<unsigned> keys_in, keys_out;
  DevBuf<int> seeds_in, order, rank_of, lane_walk;
  keys_in.alloc(ms); keys_out.alloc(ms); seeds_in.alloc(ms); order.alloc(ms); rank_of.alloc(ms); lane_walk.alloc(p.lanes);
  p.rank_of = rank_of.p; p.lane_walk = lane_walk.p;
  size_t sort_bytes = 0;
  FU_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_in.p, keys_out.p, seeds_in.p, order.p, (int)ms));
  DevBuf<unsigned char> sort_tmp;
  sort_tmp.alloc(sort_bytes + 16);

  std::vector<float> h_pt;
  std::vector<unsigned char> h_col;
  std::vector<int> h_nvis, h_vis;
  unsigned round = 2;  // 0 = free, 1 = masked on input
  g_stats = Stats();
  FU_HIP(hipDeviceSynchronize());
  const auto t_setup = std::chrono::steady_clock::now();
  g_stats.upload_seconds = std::chrono::duration<double>(t_setup - t_begin).count();
  for (int step = 0; step < (int)order_of_images.size(); ++step) {
    const int I = order_of_images[step];
    const int ns = h_img[I].dw * h_img[I].dh;
    p.step = step; p.image = I; p.num_seeds = ns;
    // seed order of this image: pixels sorted by their hash
    hipLaunchKernelGGL(fusion_keys_kernel, dim3((ns + 255) / 256), dim3(256), 0, 0, ns, keys_in.p, seeds_in.p);
    size_t sb = sort_bytes;
    FU_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, sb, keys_in.p, keys_out.p, seeds_in.p, order.p, ns));
    hipLaunchKernelGGL(fusion_invert_kernel, dim3((ns + 255) / 256), dim3(256), 0, 0, ns, order.p, rank_of.p);
    FU_HIP(hipMemsetAsync(d_cursor.p, 0, sizeof(unsigned long long), 0));
    FU_HIP(hipMemsetAsync(valid.p, 0, sizeof(int) * (size_t)ns, 0));
    // Rounds. The undecided list must hold EVERY undecided seed up to some rank (a seed may only commit
    // when all seeds before it have claimed), so first turns are offered in rank order: a small head
    // of the order first, then doubling -- by the time the bulk of the seeds is offered most of their
    // pixels are masked and their turns are empty.
    int* lists[2] = {list_a.p, list_b.p};
    p.active = lists[0];
    p.num_active = 0;
    int offered = 0;
    // First chunk = min(ns / 64, kLanes), then doubling. Measured (profiles/r03_fusion_schedule.log): on 8 x 1280 x 960
    // first chunk ns / 1024 -> 16.5 rounds per image, 9.8 Mpix/s; ns / 256 -> 14.5, 10.5; ns / 64 -> 12.75, 10.9; on
    // 8 x 2560 x 1920 ns / 64 exceeds the resident lanes, the first rounds lose the walk reuse between speculate and
    // commit, and on PatchMatch's own (noisier) maps that costs more than the saved rounds (bench leg 10.6 against
    // 14.8 Mpix/s) -- hence the cap. Growing by 4 instead of 2 saves rounds but doubles the walks that lose their
    // claims (6.6 Mpix/s). The result does not depend on the schedule (bit-exact against the sequential algorithm
    // for any of them). Knobs for experiments: COLMAP_AMD_FUSION_HEAD_DIV, COLMAP_AMD_FUSION_GROWTH.
    static const int head_div = [] { const char* e = getenv("COLMAP_AMD_FUSION_HEAD_DIV"); return e && atoi(e) > 0 ? atoi(e) : 64; }();
    static const int growth = [] { const char* e = getenv("COLMAP_AMD_FUSION_GROWTH"); return e && atoi(e) > 1 ? atoi(e) : 2; }();
    const int head = std::min(p.lanes, std::max(256, ns / head_div));
    for (int it = 0; p.num_active > 0 || offered < ns; ++it, ++round) {
      FU_CHECK(round != 0xFFFFFFFFu, "round counter");
      p.round = round;
      p.next_active = lists[(it + 1) & 1];
      FU_HIP(hipMemsetAsync(d_barrier.p, 0xFF, sizeof(unsigned), 0));
      FU_HIP(hipMemsetAsync(d_next_count.p, 0, sizeof(int), 0));
      if (p.num_active > 0) {
        p.reuse = p.num_active <= p.lanes ? 1 : 0;
        const int grid = std::min(p.lanes, (p.num_active + kBlock - 1) / kBlock * kBlock) / kBlock;
        hipLaunchKernelGGL(fusion_speculate_kernel, dim3(grid), dim3(kBlock), 0, 0, p);
        hipLaunchKernelGGL(fusion_commit_kernel, dim3(grid), dim3(kBlock), 0, 0, p);
        g_stats.rounds += 1;
        g_stats.walks += p.num_active;
      }
      if (offered < ns) {
        const int upto = (int)std::min<long long>(ns, std::max<long long>((long long)offered + head, (long long)growth * offered));
        hipLaunchKernelGGL(fusion_offer_kernel, dim3((upto - offered + 255) / 256), dim3(256), 0, 0, p, order.p, offered, upto);
        offered = upto;
      }
      int left = 0;
      FU_HIP(hipMemcpy(&left, d_next_count.p, sizeof(int), hipMemcpyDeviceToHost));
      FU_HIP(hipGetLastError());
      p.active = p.next_active;
      p.num_active = left;
    }
    hipLaunchKernelGGL(fusion_rank_kernel, dim3((ns + 256) / 256), dim3(256), 0, 0, ns, order.p, valid.p, nvis.p,
                       valid_r.p, nvis_r.p);
    size_t tb = tmp_bytes;
    FU_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, valid_r.p, scan_valid.p, ns + 1));
    tb = tmp_bytes;
    FU_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, nvis_r.p, scan_vis.p, ns + 1));
    hipLaunchKernelGGL(fusion_compact_kernel, dim3((ns + 255) / 256), dim3(256), 0, 0, ns, order.p, valid_r.p,
                       scan_valid.p, nvis_r.p, scan_vis.p, vis_off.p, pool.p, pt.p, col.p, out_pt.p, out_col.p,
                       out_nvis.p, out_vis.p);
    unsigned long long used = 0;
    FU_HIP(hipMemcpy(&used, d_cursor.p, sizeof(used), hipMemcpyDeviceToHost));
    FU_CHECK(used <= (unsigned long long)pool_cap, "visibility pool overflow (more than 2^31 - 1 visibility entries for one reference image)");
    int totals[2] = {0, 0};
    FU_HIP(hipMemcpy(&totals[0], scan_valid.p + ns, sizeof(int), hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(&totals[1], scan_vis.p + ns, sizeof(int), hipMemcpyDeviceToHost));
    FU_HIP(hipGetLastError());
    g_stats.images += 1;
    g_stats.seeds += ns;
    const size_t np = (size_t)totals[0], nv = (size_t)totals[1];
    if (np == 0) continue;
    h_pt.resize(6 * np); h_col.resize(3 * np); h_nvis.resize(np); h_vis.resize(nv);
    FU_HIP(hipMemcpy(h_pt.data(), out_pt.p, sizeof(float) * 6 * np, hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(h_col.data(), out_col.p, 3 * np, hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(h_nvis.data(), out_nvis.p, sizeof(int) * np, hipMemcpyDeviceToHost));
    if (nv) FU_HIP(hipMemcpy(h_vis.data(), out_vis.p, sizeof(int) * nv, hipMemcpyDeviceToHost));
    out->xyz_normal.insert(out->xyz_normal.end(), h_pt.begin(), h_pt.end());
    out->rgb.insert(out->rgb.end(), h_col.begin(), h_col.end());
    out->vis_idx.insert(out->vis_idx.end(), h_vis.begin(), h_vis.end());
    int64_t base = out->vis_ptr.back();
    for (size_t k = 0; k < np; ++k) {
      base += h_nvis[k];
      out->vis_ptr.push_back(base);
    }
  }
  g_stats.device_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup).count();
}

}  // namespace

extern "C" {

FUSION_API void fusion_options_init(fusion_options* o) {
  if (!o) return;
  o->min_num_pixels = 5;
  o->max_num_pixels = 10000;
  o->max_traversal_depth = 100;
  o->check_num_images = 50;
  o->max_reproj_error = 2.0;
  o->max_depth_error = 0.01;
  o->max_normal_error = 10.0;
  for (int c = 0; c < 3; ++c) {
    o->bbox_min[c] = -FLT_MAX;
    o->bbox_max[c] = FLT_MAX;
  }
}

// StereoFusionOptions::Check (fusion.cc:96-106)
FUSION_API int fusion_options_check(const fusion_options* o) {
  if (!o) return 1;
  if (o->min_num_pixels < 0) return 1;
  if (o->min_num_pixels > o->max_num_pixels) return 1;
  if (o->max_traversal_depth <= 0) return 1;
  if (o->max_reproj_error < 0 || o->max_depth_error < 0 || o->max_normal_error < 0) return 1;
  if (o->check_num_images <= 0) return 1;
  return 0;
}

FUSION_API int fusion_run(const fusion_options* options, int32_t num_images, const fusion_image* images,
                          const int32_t* overlap_ptr, const int32_t* overlap_idx, fusion_result** out) {
  if (out) *out = nullptr;
  fusion_result* r = nullptr;
  const int rc = Guard([&] {
    FU_CHECK(options && images && overlap_ptr && out, "null argument");
    FU_CHECK(overlap_idx || overlap_ptr[num_images] == 0, "null argument");
    FU_CHECK(fusion_options_check(options) == 0, "options.Check()");
    FU_CHECK(num_images > 0, "num_images");
    r = new fusion_result();
    Run(*options, num_images, images, overlap_ptr, overlap_idx, r);
  });
  if (rc != 0) {
    delete r;
    return rc;
  }
  *out = r;
  return 0;
}

FUSION_API size_t fusion_num_points(const fusion_result* r) { return r ? r->rgb.size() / 3 : 0; }

FUSION_API int fusion_get_points(const fusion_result* r, float* xyz_normal, uint8_t* rgb) {
  return Guard([&] {
    FU_CHECK(r, "null argument");
    if (xyz_normal && !r->xyz_normal.empty())
      std::memcpy(xyz_normal, r->xyz_normal.data(), r->xyz_normal.size() * sizeof(float));
    if (rgb && !r->rgb.empty()) std::memcpy(rgb, r->rgb.data(), r->rgb.size());
  });
}

FUSION_API int fusion_get_visibility(const fusion_result* r, int64_t* vis_ptr, int32_t* vis_idx, size_t* total) {
  return Guard([&] {
    FU_CHECK(r, "null argument");
    if (total) *total = r->vis_idx.size();
    if (vis_ptr) std::memcpy(vis_ptr, r->vis_ptr.data(), r->vis_ptr.size() * sizeof(int64_t));
    if (vis_idx && !r->vis_idx.empty()) std::memcpy(vis_idx, r->vis_idx.data(), r->vis_idx.size() * sizeof(int32_t));
  });
}

FUSION_API void fusion_free(fusion_result* r) { delete r; }

FUSION_API void fusion_last_timing(double* upload_seconds, double* device_seconds) {
  if (upload_seconds) *upload_seconds = g_stats.upload_seconds;
  if (device_seconds) *device_seconds = g_stats.device_seconds;
}

FUSION_API void fusion_last_stats(int64_t* images, int64_t* seeds, int64_t* rounds, int64_t* walks) {
  if (images) *images = g_stats.images;
  if (seeds) *seeds = g_stats.seeds;
  if (rounds) *rounds = g_stats.rounds;
  if (walks) *walks = g_stats.walks;
}

FUSION_API const char* fusion_last_error(void) { return g_error.c_str(); }

}  // extern "C"
