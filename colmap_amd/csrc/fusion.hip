// fusion.hip -- depth-map fusion on the GPU behind include/colmap_amd_fusion.h.
//
// What it computes: StereoFusion::Run / Fuse of the reference (src/colmap/mvs/fusion.cc:253-524) --
// for every image I in FindNextImage order (:51-73), every pixel of I in turn walks the consistency
// graph (depth, reprojection and normal tests against the pixel it started from), masks what it
// absorbs and fuses it into one point. In the reference the turns of an image are taken by a thread
// pool whose tasks are stripes of ten rows, each walked row-major by one thread (:253-269, 293-300).
// The turn order here IS that pool's schedule with its T = num_threads threads advancing in step
// (thread t takes stripes t, t + T, ...; every tick each thread takes the next pixel of its stripe):
// T = 1 is the reference with one thread, bit for bit; the default is one thread per stripe. The result
// is defined as the reference's Fuse() executed sequentially in that order (rank = tick * T + thread),
// the points collected per thread and concatenated like fusion.cc:322-337.
//
// How it runs: one WAVE per pool thread walks that thread's turns one after the other, the T waves
// concurrently, in passes over a window of ticks:
//   word      every pixel has one 64-bit word: 0 free, ~0 committed (masked for good), or
//             epoch << 32 | ~rank = tentative mark of the turn `rank` of this pass (atomicMax: the lowest
//             rank keeps the word). A wave treats committed pixels and the marks of its OWN thread as
//             masked; marks of other threads read as free.
//   cut       whenever a mark meets a mark of another turn of the same pass, the later of the two turns has
//             seen (or will have seen) masks the sequential order would not have given it:
//             rstar = min(rstar, its rank). A wave whose record buffer is full cuts at its own rank.
//   commit    every turn of rank < rstar is final -- it met no mark of a lower rank of another thread and
//             the lower ranks of its own thread are final, so on every pixel it looked at it saw the
//             sequential masks: its pixels become committed and are fused (Percentile-50 medians of
//             position, normal and colour, math/math.h:205-234; minimum size; normal length; sorted
//             distinct images). Everything else is forgotten by bumping the epoch; the next pass starts at
//             rstar. The lowest rank of a pass is never the later of two turns: every pass makes progress.
//   walk      the 64 lanes of a wave share ONE walk: the neighbours of an absorbed pixel are projected, and
//             their mask word, depth and normal fetched and tested, one neighbour per lane (those tests
//             depend only on the walk's first pixel, and a pixel masked when pushed stays masked), the
//             survivors go on a stack in LDS; popping looks at the words of the top 64 entries at once.
//             A walk is a chain of dependent round trips to memory, so the chain is kept short (walk_turn):
//             descriptors and overlap lists from an LDS copy, the mark's return value looked at after the
//             neighbour loads are in flight, the entry pushed last taken without a second look.
//   compact   per image, a device scan in (thread, tick) order; the host concatenates per thread.
//
// The checker is oracle/fusion_oracle.cpp: mode 1 runs the reference's Fuse() sequentially in the same
// order (tests/test_fusion.py compares bit for bit), mode 2 simulates the passes above with the waves
// interleaved at random, mode 0 is row-major (= num_threads 1). All arithmetic is float / double as the
// reference writes it; the library is built with -ffp-contract=off.
//
// Data layout in HBM: all depth maps in one array, all normal maps in one array of xyz triples (one
// cache line per pixel instead of three slices), the bitmaps, one 8-byte word per depth-map pixel, all
// indexed by a 64-bit global pixel offset; per wave a record buffer (pixel, image | in-box) of the turns of
// the pass, a list of its walks, nine float columns for the medians, and the overflow of its stack.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/colmap_amd_fusion.h"
#include "switches.h"

using colmap_amd::dev_switch_int;

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

// Pixels one walk can record: max_num_pixels itself between 1 024 and 16 384 (the reference's default 10 000 is NOT
// clamped), smaller options keep 1 024, larger ones are clamped to 16 384. oracle/fusion_oracle.cpp mirrors it.
constexpr int kElemCapMin = 1024, kElemCapMax = 16384;
inline int record_capacity(int max_num_pixels) { return std::min(std::max(max_num_pixels, kElemCapMin), kElemCapMax); }
constexpr int kRowStride = 10;        // rows of a pool task (fusion.cc:250-254)
constexpr int kWave = 64;
// The four capacities below only decide WHERE the data of a walk lives and when a pass is cut, never the result.
// tests/hip_emul builds this file a second time with tiny values (-DFUSION_RECORD_BUF=... etc.) so that the overflow
// paths (record buffer full, stack spill, spill growth, radix-select medians) run on inputs of a few thousand pixels.
#ifndef FUSION_RECORD_BUF
#define FUSION_RECORD_BUF (1 << 15)
#endif
#ifndef FUSION_STACK_LDS
#define FUSION_STACK_LDS 2048
#endif
#ifndef FUSION_STACK_SPILL
#define FUSION_STACK_SPILL (1 << 14)
#endif
#ifndef FUSION_MEDIAN_STAGE
#define FUSION_MEDIAN_STAGE 2048
#endif
constexpr int kRecordBuf = FUSION_RECORD_BUF;   // recorded pixels of one wave in one pass (>= the record capacity of a walk: a pass's first turn always fits)
constexpr int kWindowFirst = 256, kWindowMin = 16, kWindowMax = 32768;  // ticks of a pass: doubled after a pass without a cut, halved after a cut (8192 -> 32768: 0.695 -> 0.667 s at 8 x 2560 x 1920)
constexpr int kStackLds = FUSION_STACK_LDS;     // stack entries of a walk held in LDS (16 B each); the rest spills to HBM
constexpr int kStackSpill = FUSION_STACK_SPILL; // ... first size of that spill per wave (grown by the host when a walk overflows it)
constexpr int kCommitWaves = 16;      // waves per pool thread in the commit kernel (a border stripe has ten times the walks of an inner one: 4 -> 16 waves, 0.78 -> 0.70 s at 8 x 2560 x 1920)
constexpr int kTableBytes = 20 * 1024;  // LDS copy of the image descriptors + overlap lists of the walk kernel, when they fit
constexpr int kStage = FUSION_MEDIAN_STAGE;     // medians: values staged in LDS and ranked by counting; radix select above
constexpr unsigned long long kCommitted = ~0ull;

struct DevImage {
  float P[12], inv_P[12], inv_R[9];
  float sx, sy;          // depth map size / model image size
  const uint8_t* rgb;    // [bh][bw][3] or nullptr
  int dw, dh, bw, bh;
  long long pix_off;     // global offset of the image's first pixel (word / depth / normal arrays)
  int pos;               // step at which the image is fused; -1: not used
};

static_assert(sizeof(DevImage) % 8 == 0, "descriptors are copied to LDS word by word and hold 8-byte members");

struct PassCtl {         // device-resident control words of the pass loop (read back once per pass)
  unsigned rstar[2];     // lowest rank that must not commit, slot = pass parity (the other slot is reset by the commit kernel)
  unsigned flags;        // bit 0: a walk overflowed the stack spill
  unsigned redone;       // breadth-first walks that met the traversal limits and were repeated depth-first (statistics)
  unsigned long long walks, nodes, cursor;  // turns walked, pixels recorded (statistics); visibility pool cursor
};

struct Params {
  const DevImage* images;
  const int* optr;
  const int* oidx;
  unsigned long long* word;     // per pixel: 0 free | kCommitted | epoch << 32 | ~rank
  const float* depth;           // all depth maps, by global pixel offset
  const float* normal;          // all normal maps as xyz triples, by global pixel offset
  PassCtl* ctl;
  unsigned epoch;
  int slot;                     // pass parity
  int step, image;
  // the pool schedule of this image: T threads, stripes of L = 10 W ticks
  int T, W, H, ns;
  unsigned L;
  // this pass: ticks [tau0 (+1 for threads below rmod), tau_end), ranks below `limit`
  unsigned tau0, rmod, tau_end, limit;
  int elem_cap;                 // min(max_num_pixels, rec_cap)
  int rec_cap;                  // record_capacity(max_num_pixels), at most the pixels of the workspace
  int max_level;                // max_traversal_depth - 1
  int min_num_pixels;
  double max_depth_error;
  float max_sq_reproj, min_cos_normal;
  float bmin[3], bmax[3];
  // per wave (= pool thread): records [T][kRecordBuf], walks [T][window], values [T][9][kRecordBuf], stack spill [T][spill]
  unsigned *rec_pix, *rec_meta;
  unsigned* rec_box;            // [T][kRecordBuf]: at first + a, the record index of the a-th in-box pixel of the walk starting at first
  unsigned *w_tau, *w_first, *w_count;
  int* n_walks;
  int window_cap;
  float* vals;
  unsigned long long* spill_goff;
  uint2* spill_pm;
  float* spill_d;
  int spill_cap;
  // what the walk kernel copies into LDS at entry (a node's chain of dependent loads then starts in LDS instead of
  // HBM): 1 = image descriptors + overlap offsets, 2 = also the overlap lists; 0 = neither fits kTableBytes
  int lds_tables;
  int n_images, n_overlap;
  // breadth-first walks (walk_turn_wide): lanes per popped entry = the longest overlap list, entries popped together =
  // 64 / that; 0 = depth-first walks only. wide_bound = the walk size up to which the absorbed SET cannot depend on
  // the order of the traversal; a walk that would pass it takes its marks back and is repeated depth-first.
  int wide_group, wide_bound;
  // per-seed outputs
  int *valid, *nvis, *vis_off;
  float* pt;            // [num_seeds][6]
  unsigned char* col;   // [num_seeds][3]
  int* pool;            // visibility lists, allocated with an atomic cursor
  long long pool_cap;   // entries; the host checks the cursor against it after every image
};

__device__ inline unsigned long long ld_word(const unsigned long long* p) {  // the words are L2 atomics: read past the L1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// masked for pool thread t: committed, or a tentative mark of this pass made by t itself
__device__ inline bool masked_for(unsigned long long w, unsigned epoch, unsigned T, unsigned t) {
  if (w == kCommitted) return true;
  return (unsigned)(w >> 32) == epoch && (0xFFFFFFFFu - (unsigned)w) % T == t;
}

__device__ inline int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline unsigned uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ inline unsigned long long uniform(unsigned long long v) {
  const unsigned lo = uniform((unsigned)v), hi = uniform((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ inline unsigned long long shfl64(unsigned long long v, int src) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

// pixel of image (W x H) whose turn pool thread t takes in tick tau, or -1 (struct Pool of the oracle)
__device__ inline int seed_of(const Params& p, unsigned tau, unsigned t) {
  const unsigned k = (tau / p.L) * (unsigned)p.T + t;
  if (k >= (unsigned)p.ns) return -1;
  const unsigned pos = tau % p.L;
  const unsigned row = kRowStride * k + pos / (unsigned)p.W;
  if (row >= (unsigned)p.H) return -1;
  return (int)(row * (unsigned)p.W + pos % (unsigned)p.W);
}

// LDS pointers carry address space 3 explicitly: the accesses are ds_* instructions, and the compiler cannot merge the
// two halves of put / get into one access through a selected generic pointer (it did: a pointer table in scratch).
#define FUSION_LDS __attribute__((address_space(3)))
struct WaveStack {  // the walk's stack: entries [0, kStackLds) in LDS, the rest in the wave's spill
  FUSION_LDS unsigned long long* cand;  // [kWave] walk_turn_wide: the pixel every lane wants to push ...
  FUSION_LDS int* slot;                 // [2 * kWave] ... and which lane holds a hash slot: duplicates of one batch are dropped
  FUSION_LDS unsigned long long* win;   // [1] bit d: the start pixel of tick win_tau0 + d of this pool thread was absorbed by the current walk
  FUSION_LDS unsigned long long* lds_goff;
  FUSION_LDS uint2* lds_pm;     // x = pixel, y = image | level << 16
  FUSION_LDS float* lds_d;      // depth of the pixel (loaded when the entry was tested: a pop needs no second load for it)
  unsigned long long* spill_goff;
  uint2* spill_pm;
  float* spill_d;
  __device__ void put(int i, unsigned long long goff, uint2 pm, float d) const {
    if (i < kStackLds) { lds_goff[i] = goff; lds_pm[i].x = pm.x; lds_pm[i].y = pm.y; lds_d[i] = d; }
    else { spill_goff[i - kStackLds] = goff; spill_pm[i - kStackLds] = pm; spill_d[i - kStackLds] = d; }
  }
  __device__ void get(int i, unsigned long long* goff, uint2* pm, float* d) const {
    if (i < kStackLds) { *goff = lds_goff[i]; pm->x = lds_pm[i].x; pm->y = lds_pm[i].y; *d = lds_d[i]; }
    else { *goff = spill_goff[i - kStackLds]; *pm = spill_pm[i - kStackLds]; *d = spill_d[i - kStackLds]; }
  }
};

// Image descriptors and overlap lists as the walk reads them: the arrays in HBM, or the wave's copy in LDS
// (generic pointers: the same loads serve both).
struct WalkTables {
  const DevImage* images;
  const int* optr;
  const int* oidx;
};

// The start pixels of the next 64 ticks of a pool thread are looked at ONCE (fusion_walk_kernel): between two of its turns
// only its own walks can change which of them are free for it -- other threads' marks read as free, commits happen
// between passes. A walk therefore notes which pixels of that window it absorbed (pixel of the image being fused ->
// stripe -> thread and tick, the inverse of seed_of).
__device__ inline void note_window_pixel(const Params& p, const WaveStack& st, unsigned t, unsigned win_tau0, int img, int pix) {
  if (img != p.image) return;
  const unsigned row = (unsigned)pix / (unsigned)p.W, col = (unsigned)pix - row * (unsigned)p.W;
  const unsigned k = row / (unsigned)kRowStride;
  if (k % (unsigned)p.T != t) return;
  const unsigned tick = (k / (unsigned)p.T) * p.L + (row - k * (unsigned)kRowStride) * (unsigned)p.W + col;
  const unsigned d = tick - win_tau0;
  if (d < (unsigned)kWave) __hip_atomic_fetch_or(st.win, 1ull << d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One turn of pool thread t: StereoFusion::Fuse's traversal (fusion.cc:401-489) from `seed` (free for t, positive
// depth), executed by the 64 lanes of the wave together (every variable that steers the control flow is wave-uniform).
// rec_n: pixels the wave has recorded in this pass; n_walks: its walks. false: record buffer or stack spill full --
// the turn is abandoned and cuts the pass at its own rank.
//
// The time of a walk is a chain of dependent memory round trips per absorbed pixel, so the chain is kept short:
//  * the mark (atomicMax) is issued first and its return value looked at LAST, after the neighbour loads of the
//    same pixel are in flight (what it decides -- the rank cut -- does not steer the walk);
//  * descriptors and overlap lists come from LDS when they fit (WalkTables): the chain to a neighbour's mask word,
//    depth and normal is then one trip to HBM instead of three;
//  * the topmost entry pushed by an expansion is taken without looking at its mask word again: it was free when it
//    was tested a moment ago and the only mark made since is that of the pixel being expanded, which is another pixel;
//    its depth travels in the stack entry. Older entries are looked at again when they surface (the walk may have
//    absorbed them by another path).
__device__ __forceinline__ bool walk_turn(const Params& p, const WalkTables& tb, const WaveStack& st, int lane, unsigned t, unsigned tau,
                          unsigned win_tau0, unsigned rank, int seed, float seed_depth, int* rec_n, int* n_walks, unsigned long long* stat_nodes) {
  const unsigned long long key = ((unsigned long long)p.epoch << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
  unsigned* const rec_pix = p.rec_pix + (size_t)t * kRecordBuf;
  unsigned* const rec_meta = p.rec_meta + (size_t)t * kRecordBuf;
  const int first = *rec_n;
  int n = first, recorded = 0, nin = 0, sp = 0;
  float ref[3] = {0.f, 0.f, 0.f}, refn[3] = {0.f, 0.f, 0.f};
  int img = p.image, pix = seed, level = 0;
  unsigned long long goff = (unsigned long long)tb.images[p.image].pix_off + (unsigned long long)seed;
  float depth = seed_depth;
  bool ok = true;
  for (;;) {
    // ---- absorb (img, pix, level): fusion.cc:437-472 ----
    const DevImage& im = tb.images[img];
    const int row = pix / im.dw, col = pix - row * im.dw;
    const float hx = (float)col * depth, hy = (float)row * depth;
    float xyz[3];
    for (int r = 0; r < 3; ++r)
      xyz[r] = im.inv_P[4 * r] * hx + im.inv_P[4 * r + 1] * hy + im.inv_P[4 * r + 2] * depth + im.inv_P[4 * r + 3] * 1.0f;
    const bool in_box = !(xyz[0] < p.bmin[0] || xyz[1] < p.bmin[1] || xyz[2] < p.bmin[2] || xyz[0] > p.bmax[0] ||
                          xyz[1] > p.bmax[1] || xyz[2] > p.bmax[2]);
    if (recorded >= p.rec_cap) break;  // record capacity of one walk: the walk ends
    if (n >= kRecordBuf) { ok = false; break; }
    unsigned long long old = 0ull;
    if (lane == 0) {
      rec_pix[n] = (unsigned)pix;
      rec_meta[n] = (unsigned)img | (in_box ? 0x80000000u : 0u);
      old = atomicMax(p.word + goff, key);  // (looked at below, after the neighbour loads have been issued)
      note_window_pixel(p, st, t, win_tau0, img, pix);
    }
    ++n; ++recorded;
    bool expand = false, capped = false;
    if (in_box) {
      ++nin;
      if (level == 0) {
        const float* nl = p.normal + 3 * goff;
        const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
        for (int r = 0; r < 3; ++r) refn[r] = im.inv_R[3 * r] * nl0 + im.inv_R[3 * r + 1] * nl1 + im.inv_R[3 * r + 2] * nl2;
        ref[0] = xyz[0]; ref[1] = xyz[1]; ref[2] = xyz[2];
      }
      capped = nin >= p.elem_cap;  // max_num_pixels reached (fusion.cc:470-472): the walk ends after the mark is settled
      expand = !capped && level < p.max_level;
    }
    int pushed = 0;  // entries this expansion put on the stack
    if (expand) {
      // ---- neighbours (fusion.cc:474-488), one per lane; pushed in list order if they pass the tests of
      // fusion.cc:407-447 that do not depend on the masks, and are not masked now ----
      const int o0 = tb.optr[img], nov = tb.optr[img + 1] - o0;
      for (int base = 0; base < nov; base += kWave) {
        const int k = base + lane;
        bool pass = false;
        unsigned long long qoff = 0ull;
        int q = 0, next = 0;
        float d = 0.0f;
        if (k < nov) {
          next = tb.oidx[o0 + k];
          const DevImage& nx = tb.images[next];
          if (nx.pos >= p.step) {  // used, and not fused in an earlier step
            float np[3];
            for (int r = 0; r < 3; ++r) np[r] = nx.P[4 * r] * xyz[0] + nx.P[4 * r + 1] * xyz[1] + nx.P[4 * r + 2] * xyz[2] + nx.P[4 * r + 3];
            const float fcol = roundf(np[0] / np[2]), frow = roundf(np[1] / np[2]);
            if (fcol >= 0.0f && frow >= 0.0f && fcol < (float)nx.dw && frow < (float)nx.dh) {
              const int qcol = (int)fcol, qrow = (int)frow;
              q = qrow * nx.dw + qcol;
              qoff = (unsigned long long)nx.pix_off + (unsigned long long)q;
              const unsigned long long w = ld_word(p.word + qoff);
              d = p.depth[qoff];
              const float* nl = p.normal + 3 * qoff;
              const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
              // (qoff != goff: the pixel being expanded -- an image listed as its own neighbour -- is masked by the
              // mark issued a moment ago, which this load is not ordered behind)
              if (qoff != goff && !masked_for(w, p.epoch, (unsigned)p.T, t) && d > 0.0f) {
                float proj[3];
                for (int r = 0; r < 3; ++r)
                  proj[r] = nx.P[4 * r] * ref[0] + nx.P[4 * r + 1] * ref[1] + nx.P[4 * r + 2] * ref[2] + nx.P[4 * r + 3] * 1.0f;
                const float depth_error = fabsf((proj[2] - d) / d);
                const float col_diff = proj[0] / proj[2] - (float)qcol;
                const float row_diff = proj[1] / proj[2] - (float)qrow;
                float nrm[3];
                for (int r = 0; r < 3; ++r) nrm[r] = nx.inv_R[3 * r] * nl0 + nx.inv_R[3 * r + 1] * nl1 + nx.inv_R[3 * r + 2] * nl2;
                const float c = refn[0] * nrm[0] + refn[1] * nrm[1] + refn[2] * nrm[2];
                pass = !((double)depth_error > p.max_depth_error) && !(col_diff * col_diff + row_diff * row_diff > p.max_sq_reproj) &&
                       !(c < p.min_cos_normal);
              }
            }
          }
        }
        const unsigned long long m = __ballot(pass);
        const int cnt = __popcll(m);
        if (sp + cnt > kStackLds + p.spill_cap) { ok = false; break; }
        if (pass) st.put(sp + __popcll(m & ((1ull << lane) - 1ull)), qoff, make_uint2((unsigned)q, (unsigned)next | ((unsigned)(level + 1) << 16)), d);
        sp += cnt;
        pushed += cnt;
      }
    }
    // ---- the mark's old value: a mark of another turn of this pass means the later of the two turns must not commit ----
    old = shfl64(old, 0);
    if ((unsigned)(old >> 32) == p.epoch) {
      const unsigned other = 0xFFFFFFFFu - (unsigned)old;
      if (other != rank && lane == 0) atomicMin(&p.ctl->rstar[p.slot], other > rank ? other : rank);
    }
    if (!ok) {
      if (lane == 0) atomicOr(&p.ctl->flags, 1u);
      break;
    }
    if (capped) break;
    if (expand) __syncthreads();  // the pushes (LDS / spill) before the pops of other lanes
    // ---- next pixel: the topmost stack entry that is not masked (fusion.cc:408-414) ----
    bool found = false;
    if (pushed > 0) {  // the entry this expansion pushed last: known to be free, its depth is in the entry
      uint2 epm;
      st.get(sp - 1, &goff, &epm, &depth);
      sp -= 1;
      pix = (int)epm.x;
      img = (int)(epm.y & 0xFFFFu);
      level = (int)(epm.y >> 16);
      found = true;
    }
    while (!found && sp > 0) {  // older entries: 64 per look, their mask words read again
      const int idx = sp - 1 - lane;
      unsigned long long eoff = 0ull;
      uint2 epm = make_uint2(0u, 0u);
      bool free_ = false;
      float ed = 0.0f;
      if (idx >= 0) {
        st.get(idx, &eoff, &epm, &ed);
        const unsigned long long w = ld_word(p.word + eoff);
        free_ = !masked_for(w, p.epoch, (unsigned)p.T, t);
      }
      const unsigned long long m = __ballot(free_);
      if (m == 0ull) {
        sp = sp > kWave ? sp - kWave : 0;
        continue;
      }
      const int j = __ffsll((long long)m) - 1;
      sp = sp - 1 - j;
      goff = shfl64(eoff, j);
      pix = __shfl((int)epm.x, j);
      const unsigned meta = (unsigned)__shfl((int)epm.y, j);
      depth = __shfl(ed, j);
      img = (int)(meta & 0xFFFFu);
      level = (int)(meta >> 16);
      found = true;
    }
    __syncthreads();  // the reads of this look before the pushes of the next expansion
    if (!found) break;
    goff = uniform(goff); pix = uniform(pix); img = uniform(img); level = uniform(level); depth = uniform(depth);
  }
  if (!ok) {  // abandoned: nothing of it is recorded, the pass is cut at this turn
    if (lane == 0) atomicMin(&p.ctl->rstar[p.slot], rank);
    return false;
  }
  if (lane == 0) {
    const int wi = *n_walks;
    p.w_tau[(size_t)t * p.window_cap + wi] = tau;
    p.w_first[(size_t)t * p.window_cap + wi] = (unsigned)first;
    p.w_count[(size_t)t * p.window_cap + wi] = (unsigned)(n - first);
  }
  *stat_nodes += (unsigned long long)(n - first);
  *rec_n = n;
  *n_walks += 1;
  return true;
}

// The same turn with the frontier expanded SEVERAL ENTRIES AT A TIME (round 6). walk_turn is a chain of one memory
// round trip per absorbed pixel -- pop an entry, mark it, fetch its neighbours' words / depths / normals, push the
// survivors -- with up to check_num_images lanes busy; here the top 64 / wide_group stack entries are popped together,
// lane (u, k) = (popped entry, neighbour): their marks (atomicMax) and the neighbour loads of ALL of them go out in one
// round trip, so a walk takes about as many trips as it has levels instead of as many as it has pixels.
// Why the result is the same: which pixels a walk absorbs is a closure -- a pixel is absorbed iff it is reachable from the
// seed over pixels that pass the tests of fusion.cc:407-447, and those tests compare with the SEED's point and normal,
// not with the path -- so the absorbed set does not depend on the order of the traversal, and neither do the medians, the
// sorted visibility list or the marks, AS LONG AS none of the three limits of the traversal can bind: the level limit
// (fusion.cc:473: levels are at most the number of absorbed pixels), max_num_pixels and the record capacity. All three
// are implied by "the walk absorbs at most wide_bound = min(max_traversal_depth - 1, max_num_pixels - 1, record capacity)
// pixels"; a walk that would exceed it takes its marks back (compare-and-swap of its own key to "free": a mark another
// turn has replaced since stays, and whoever looked at the word in between has at worst recorded a conflict that cuts
// the pass early -- never a wrong mask) and the turn is repeated by walk_turn: return value 2. An entry that was absorbed
// by another path since it was pushed -- or twice in one batch -- is recognised by the old value of its own mark.
__device__ __forceinline__ int walk_turn_wide(const Params& p, const WalkTables& tb, const WaveStack& st, int lane, unsigned t, unsigned tau,
                               unsigned win_tau0, unsigned rank, int seed, float seed_depth, int* rec_n, int* n_walks, unsigned long long* stat_nodes,
                               int resume_sp = -1, const float* resume_ref = nullptr, const float* resume_refn = nullptr) {
  const unsigned long long key = ((unsigned long long)p.epoch << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
  unsigned* const rec_pix = p.rec_pix + (size_t)t * kRecordBuf;
  unsigned* const rec_meta = p.rec_meta + (size_t)t * kRecordBuf;
  const int first = *rec_n;
  const int G = p.wide_group, NBF = kWave / G;
  const int u_wide = lane / G, k_wide = lane - u_wide * G;
  int n = first, sp = 1;
  float ref[3] = {0.f, 0.f, 0.f}, refn[3] = {0.f, 0.f, 0.f};
  if (resume_sp >= 0) {
    // the seed's batch has been taken already (seed_group): the seed is marked and recorded at `first`, its surviving
    // neighbours are on the stack, the reference point and normal come with the call
    n = first + 1;
    sp = resume_sp;
    for (int r = 0; r < 3; ++r) { ref[r] = resume_ref[r]; refn[r] = resume_refn[r]; }
  } else if (lane == 0) {
    st.put(0, (unsigned long long)tb.images[p.image].pix_off + (unsigned long long)seed, make_uint2((unsigned)seed, (unsigned)p.image), seed_depth);
  }
  __syncthreads();
  bool ok = true, narrow = false;
  while (sp > 0) {
    const int npop = sp < NBF ? sp : NBF;
    if (n - first + npop > p.wide_bound) { ok = false; narrow = true; break; }
    if (n + npop > kRecordBuf) { ok = false; break; }
    // ---- the popped entries: lane (u, k) reads entry u; the seed (alone in the first batch) is read by every lane, so
    // that the walk's reference point and normal below are computed identically everywhere ----
    const bool first_batch = n == first;
    const int u = first_batch ? 0 : u_wide, k = first_batch ? lane : k_wide;
    const bool have = u < npop;
    unsigned long long goff = 0ull;
    uint2 epm = make_uint2(0u, 0u);
    float depth = 0.0f;
    if (have) st.get(sp - 1 - u, &goff, &epm, &depth);
    __syncthreads();  // every lane has its entry before the pushes below reuse the slots
    const int pix = (int)epm.x, img = (int)(epm.y & 0xFFFFu), level = (int)(epm.y >> 16);
    unsigned long long old = 0ull;
    if (have && k == 0) old = atomicMax(p.word + goff, key);  // (looked at below, after the neighbour loads have been issued)
    const DevImage& im = tb.images[img];
    const int row = pix / im.dw, col = pix - row * im.dw;
    const float hx = (float)col * depth, hy = (float)row * depth;
    float xyz[3];
    for (int r = 0; r < 3; ++r)
      xyz[r] = im.inv_P[4 * r] * hx + im.inv_P[4 * r + 1] * hy + im.inv_P[4 * r + 2] * depth + im.inv_P[4 * r + 3] * 1.0f;
    const bool in_box = !(xyz[0] < p.bmin[0] || xyz[1] < p.bmin[1] || xyz[2] < p.bmin[2] || xyz[0] > p.bmax[0] ||
                          xyz[1] > p.bmax[1] || xyz[2] > p.bmax[2]);
    if (first_batch) {  // the seed: the walk's reference point and normal, for every lane
      const float* nl = p.normal + 3 * goff;
      const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
      for (int r = 0; r < 3; ++r) refn[r] = im.inv_R[3 * r] * nl0 + im.inv_R[3 * r + 1] * nl1 + im.inv_R[3 * r + 2] * nl2;
      for (int r = 0; r < 3; ++r) ref[r] = xyz[r];
    }
    // ---- neighbour k of entry u (fusion.cc:474-488 and the mask-independent tests of :407-447) ----
    bool pass = false;
    unsigned long long qoff = 0ull;
    int q = 0, next = 0;
    float d = 0.0f;
    if (have && in_box && level < p.max_level) {
      const int o0 = tb.optr[img], nov = tb.optr[img + 1] - o0;
      if (k < nov) {
        next = tb.oidx[o0 + k];
        const DevImage& nx = tb.images[next];
        if (nx.pos >= p.step) {  // used, and not fused in an earlier step
          float np[3];
          for (int r = 0; r < 3; ++r) np[r] = nx.P[4 * r] * xyz[0] + nx.P[4 * r + 1] * xyz[1] + nx.P[4 * r + 2] * xyz[2] + nx.P[4 * r + 3];
          const float fcol = roundf(np[0] / np[2]), frow = roundf(np[1] / np[2]);
          if (fcol >= 0.0f && frow >= 0.0f && fcol < (float)nx.dw && frow < (float)nx.dh) {
            const int qcol = (int)fcol, qrow = (int)frow;
            q = qrow * nx.dw + qcol;
            qoff = (unsigned long long)nx.pix_off + (unsigned long long)q;
            const unsigned long long w = ld_word(p.word + qoff);
            d = p.depth[qoff];
            const float* nl = p.normal + 3 * qoff;
            const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
            if (qoff != goff && !masked_for(w, p.epoch, (unsigned)p.T, t) && d > 0.0f) {
              float proj[3];
              for (int r = 0; r < 3; ++r)
                proj[r] = nx.P[4 * r] * ref[0] + nx.P[4 * r + 1] * ref[1] + nx.P[4 * r + 2] * ref[2] + nx.P[4 * r + 3] * 1.0f;
              const float depth_error = fabsf((proj[2] - d) / d);
              const float col_diff = proj[0] / proj[2] - (float)qcol;
              const float row_diff = proj[1] / proj[2] - (float)qrow;
              float nrm[3];
              for (int r = 0; r < 3; ++r) nrm[r] = nx.inv_R[3 * r] * nl0 + nx.inv_R[3 * r + 1] * nl1 + nx.inv_R[3 * r + 2] * nl2;
              const float c = refn[0] * nrm[0] + refn[1] * nrm[1] + refn[2] * nrm[2];
              pass = !((double)depth_error > p.max_depth_error) && !(col_diff * col_diff + row_diff * row_diff > p.max_sq_reproj) &&
                     !(c < p.min_cos_normal);
            }
          }
        }
      }
    }
    // ---- the marks' old values: is the entry absorbed (free for this thread until now)? ----
    old = shfl64(old, (!first_batch && u * G < kWave) ? u * G : 0);
    const bool absorbed = have && !masked_for(old, p.epoch, (unsigned)p.T, t);
    if (absorbed && k == 0 && (unsigned)(old >> 32) == p.epoch) {  // a mark of another turn of this pass: the later turn must not commit
      const unsigned other = 0xFFFFFFFFu - (unsigned)old;
      if (other != rank) atomicMin(&p.ctl->rstar[p.slot], other > rank ? other : rank);
    }
    {
      const unsigned long long ma = __ballot(absorbed && k == 0);
      if (absorbed && k == 0) {
        note_window_pixel(p, st, t, win_tau0, img, pix);
        const int at = n + __popcll(ma & ((1ull << lane) - 1ull));
        rec_pix[at] = (unsigned)pix;
        rec_meta[at] = (unsigned)img | (in_box ? 0x80000000u : 0u);
      }
      n += __popcll(ma);
    }
    // ---- push the survivors of the absorbed entries ----
    pass = pass && absorbed;
    {
      // The same pixel is usually a neighbour of several entries of the batch (every absorbed pixel of image A projects
      // near the same pixel of image B): pushed once per parent it would come back as that many stack entries, all but
      // one of them dead. One survivor per pixel: the lane that holds the pixel's hash slot (lanes whose slot is held by
      // another pixel all stay: harmless).
      const int hs = (int)((qoff * 0x9E3779B97F4A7C15ull) >> 57);  // 7 bits
      st.cand[lane] = pass ? qoff : ~0ull;
      if (pass) st.slot[hs] = lane;
      __syncthreads();
      if (pass) {
        const int holder = st.slot[hs];
        if (holder != lane && st.cand[holder] == qoff) pass = false;
      }
    }
    const unsigned long long m = __ballot(pass);
    const int cnt = __popcll(m);
    const int base = sp - npop;
    if (base + cnt > kStackLds + p.spill_cap) {
      if (lane == 0) atomicOr(&p.ctl->flags, 1u);
      ok = false;
      break;
    }
    if (pass) st.put(base + __popcll(m & ((1ull << lane) - 1ull)), qoff, make_uint2((unsigned)q, (unsigned)next | ((unsigned)(level + 1) << 16)), d);
    sp = base + cnt;
    __syncthreads();  // the pushes (LDS / spill) before the pops of the next batch
  }
  if (narrow) {  // the limits of the traversal could bind: undo the marks, the caller repeats the turn depth-first
    __syncthreads();  // (the records below were written by other lanes)
    for (int e = first + lane; e < n; e += kWave) {
      const unsigned long long goff = (unsigned long long)tb.images[rec_meta[e] & 0xFFFFu].pix_off + (unsigned long long)rec_pix[e];
      unsigned long long expect = key;
      __hip_atomic_compare_exchange_strong(p.word + goff, &expect, 0ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) atomicAdd(&p.ctl->redone, 1u);
    __syncthreads();
    return 2;
  }
  if (!ok) {  // abandoned: nothing of it is recorded, the pass is cut at this turn
    if (lane == 0) atomicMin(&p.ctl->rstar[p.slot], rank);
    return 0;
  }
  if (lane == 0) {
    const int wi = *n_walks;
    p.w_tau[(size_t)t * p.window_cap + wi] = tau;
    p.w_first[(size_t)t * p.window_cap + wi] = (unsigned)first;
    p.w_count[(size_t)t * p.window_cap + wi] = (unsigned)(n - first);
  }
  *stat_nodes += (unsigned long long)(n - first);
  *rec_n = n;
  *n_walks += 1;
  return 1;
}

// A pass, first half: wave t takes the turns of pool thread t in the window one after the other.
__global__ void __launch_bounds__(kWave) fusion_walk_kernel(Params p) {
  __shared__ unsigned long long s_goff[kStackLds];
  __shared__ uint2 s_pm[kStackLds];
  __shared__ float s_d[kStackLds];
  __shared__ __attribute__((aligned(16))) int s_tab[kTableBytes / 4];
  const unsigned t = blockIdx.x;
  const int lane = threadIdx.x;
  WalkTables tb{p.images, p.optr, p.oidx};
  if (p.lds_tables >= 1) {  // descriptors + overlap offsets (+ the lists) into LDS: word-wise copies
    const int wi = (int)(sizeof(DevImage) / 4) * p.n_images, wo = p.n_images + 1;
    const int* src = reinterpret_cast<const int*>(p.images);
    for (int i = lane; i < wi; i += kWave) s_tab[i] = src[i];
    for (int i = lane; i < wo; i += kWave) s_tab[wi + i] = p.optr[i];
    tb.images = reinterpret_cast<const DevImage*>(s_tab);
    tb.optr = s_tab + wi;
    if (p.lds_tables >= 2) {
      for (int i = lane; i < p.n_overlap; i += kWave) s_tab[wi + wo + i] = p.oidx[i];
      tb.oidx = s_tab + wi + wo;
    }
    __syncthreads();
  }
  __shared__ unsigned long long s_cand[kWave];
  __shared__ int s_slot[2 * kWave];
  __shared__ unsigned long long s_win[1];
  WaveStack st{(FUSION_LDS unsigned long long*)s_cand, (FUSION_LDS int*)s_slot, (FUSION_LDS unsigned long long*)s_win,
               (FUSION_LDS unsigned long long*)s_goff, (FUSION_LDS uint2*)s_pm, (FUSION_LDS float*)s_d,
               p.spill_goff + (size_t)t * p.spill_cap, p.spill_pm + (size_t)t * p.spill_cap, p.spill_d + (size_t)t * p.spill_cap};
  const unsigned long long img_off = (unsigned long long)tb.images[p.image].pix_off;
  unsigned tau = p.tau0 + (t < p.rmod ? 1u : 0u);
  int rec_n = 0, n_walks = 0;
  unsigned long long walks = 0ull, nodes = 0ull;
  bool stop = false;
  while (!stop && tau < p.tau_end) {
    // the next 64 turns of this thread: which start pixels are free (for this thread) and have a depth? Looked at once
    // per window: until the window is used up only this wave's own walks can change the answer, and they say so
    // (note_window_pixel) -- one round trip per 64 ticks instead of one per walk.
    const unsigned my_tau = tau + (unsigned)lane;
    int s = -1;
    float d = 0.0f;
    bool cand = false;
    if (my_tau < p.tau_end) {
      s = seed_of(p, my_tau, t);
      if (s >= 0) {
        const unsigned long long w = ld_word(p.word + img_off + (unsigned long long)s);
        d = p.depth[img_off + (unsigned long long)s];
        cand = d > 0.0f && !masked_for(w, p.epoch, (unsigned)p.T, t);
      }
    }
    unsigned rs = ld_u32(&p.ctl->rstar[p.slot]);  // other waves lower it while this one runs: lane 0's view counts
    rs = uniform((unsigned)__shfl((int)(rs < p.limit ? rs : p.limit), 0));
    unsigned long long m = __ballot(cand);
    const unsigned win_tau0 = tau;
    const unsigned next_tau = p.tau_end - tau > (unsigned)kWave ? tau + (unsigned)kWave : p.tau_end;
    while (m != 0ull) {
      // ---- several start pixels at once (seed_group below): worthwhile when the thread has a run of turns whose walks
      // absorb nothing but their start pixel (stripes the other images do not see: ten times the turns of an ordinary
      // stripe, and a pass lasts as long as its slowest wave) ----
      if (p.wide_group > 0 && (m & (m - 1ull)) != 0ull && rec_n + kWave <= kRecordBuf && n_walks + kWave <= p.window_cap) {
        const int G = p.wide_group, NBF = kWave / G;
        const int u = lane / G, k = lane - u * G;
        // the group: the lowest NBF candidates of the window whose ranks can still commit
        int ng = 0, my_j = 0;
        {
          unsigned long long mm = m;
          for (int g = 0; g < NBF && mm != 0ull; ++g) {
            const int jj = __ffsll((long long)mm) - 1;
            mm &= mm - 1ull;
            if ((win_tau0 + (unsigned)jj) * (unsigned)p.T + t >= rs) break;
            if (u == g) my_j = jj;
            ++ng;
          }
        }
        if (ng >= 2) {
          const bool have = u < ng;
          const int seed_u = __shfl(s, my_j);
          const float depth_u = __shfl(d, my_j);
          const unsigned tau_u = win_tau0 + (unsigned)my_j;
          const unsigned rank_u = tau_u * (unsigned)p.T + t;
          const unsigned long long key_u = ((unsigned long long)p.epoch << 32) | (unsigned long long)(0xFFFFFFFFu - rank_u);
          const unsigned long long goff = img_off + (unsigned long long)seed_u;
          unsigned long long old = 0ull;
          if (have && k == 0) old = atomicMax(p.word + goff, key_u);
          const DevImage& im = tb.images[p.image];
          const int row = seed_u / im.dw, col = seed_u - row * im.dw;
          const float hx = (float)col * depth_u, hy = (float)row * depth_u;
          float xyz[3], refn[3];
          for (int r = 0; r < 3; ++r)
            xyz[r] = im.inv_P[4 * r] * hx + im.inv_P[4 * r + 1] * hy + im.inv_P[4 * r + 2] * depth_u + im.inv_P[4 * r + 3] * 1.0f;
          const bool in_box = !(xyz[0] < p.bmin[0] || xyz[1] < p.bmin[1] || xyz[2] < p.bmin[2] || xyz[0] > p.bmax[0] ||
                                xyz[1] > p.bmax[1] || xyz[2] > p.bmax[2]);
          {
            const float* nl = p.normal + 3 * goff;
            const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
            for (int r = 0; r < 3; ++r) refn[r] = im.inv_R[3 * r] * nl0 + im.inv_R[3 * r + 1] * nl1 + im.inv_R[3 * r + 2] * nl2;
          }
          // neighbour k of start pixel u: the tests of walk_turn_wide with this pixel as the walk's reference
          bool pass = false;
          unsigned long long qoff = 0ull;
          int q = 0, next = 0;
          float qd = 0.0f;
          if (have && in_box && 0 < p.max_level) {
            const int o0 = tb.optr[p.image], nov = tb.optr[p.image + 1] - o0;
            if (k < nov) {
              next = tb.oidx[o0 + k];
              const DevImage& nx = tb.images[next];
              if (nx.pos >= p.step) {
                float np[3];
                for (int r = 0; r < 3; ++r) np[r] = nx.P[4 * r] * xyz[0] + nx.P[4 * r + 1] * xyz[1] + nx.P[4 * r + 2] * xyz[2] + nx.P[4 * r + 3];
                const float fcol = roundf(np[0] / np[2]), frow = roundf(np[1] / np[2]);
                if (fcol >= 0.0f && frow >= 0.0f && fcol < (float)nx.dw && frow < (float)nx.dh) {
                  const int qcol = (int)fcol, qrow = (int)frow;
                  q = qrow * nx.dw + qcol;
                  qoff = (unsigned long long)nx.pix_off + (unsigned long long)q;
                  const unsigned long long w = ld_word(p.word + qoff);
                  qd = p.depth[qoff];
                  const float* nl = p.normal + 3 * qoff;
                  const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
                  if (qoff != goff && !masked_for(w, p.epoch, (unsigned)p.T, t) && qd > 0.0f) {
                    float proj[3];
                    for (int r = 0; r < 3; ++r)
                      proj[r] = nx.P[4 * r] * xyz[0] + nx.P[4 * r + 1] * xyz[1] + nx.P[4 * r + 2] * xyz[2] + nx.P[4 * r + 3] * 1.0f;
                    const float depth_error = fabsf((proj[2] - qd) / qd);
                    const float col_diff = proj[0] / proj[2] - (float)qcol;
                    const float row_diff = proj[1] / proj[2] - (float)qrow;
                    float nrm[3];
                    for (int r = 0; r < 3; ++r) nrm[r] = nx.inv_R[3 * r] * nl0 + nx.inv_R[3 * r + 1] * nl1 + nx.inv_R[3 * r + 2] * nl2;
                    const float c = refn[0] * nrm[0] + refn[1] * nrm[1] + refn[2] * nrm[2];
                    pass = !((double)depth_error > p.max_depth_error) && !(col_diff * col_diff + row_diff * row_diff > p.max_sq_reproj) &&
                           !(c < p.min_cos_normal);
                  }
                }
              }
            }
          }
          // marks of other turns of this pass under the start pixels: the later turn must not commit (as in the walks)
          if (have && k == 0 && (unsigned)(old >> 32) == p.epoch) {
            const unsigned other = 0xFFFFFFFFu - (unsigned)old;
            if (other != rank_u) atomicMin(&p.ctl->rstar[p.slot], other > rank_u ? other : rank_u);
          }
          // the first start pixel with a surviving neighbour: the turns before it are complete (they absorbed their
          // start pixel and nothing else, which no other turn of the group can tell from the sequential run); it goes on
          // as an ordinary walk; the turns behind it take their marks back and are looked at again after that walk
          const unsigned long long pm = __ballot(pass);
          int ustar = ng;
          for (int g = ng - 1; g >= 0; --g)
            if ((pm >> (g * G)) & ((1ull << G) - 1ull)) ustar = g;
          if (have && k == 0 && u > ustar) {
            unsigned long long expect = key_u;
            __hip_atomic_compare_exchange_strong(p.word + goff, &expect, 0ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          const int nrec = ustar < ng ? ustar + 1 : ng;
          if (have && k == 0 && u < nrec) {
            unsigned* const rp = p.rec_pix + (size_t)t * kRecordBuf;
            unsigned* const rm = p.rec_meta + (size_t)t * kRecordBuf;
            rp[rec_n + u] = (unsigned)seed_u;
            rm[rec_n + u] = (unsigned)p.image | (in_box ? 0x80000000u : 0u);
            if (u < ustar) {
              p.w_tau[(size_t)t * p.window_cap + n_walks + u] = tau_u;
              p.w_first[(size_t)t * p.window_cap + n_walks + u] = (unsigned)(rec_n + u);
              p.w_count[(size_t)t * p.window_cap + n_walks + u] = 1u;
            }
          }
          // the candidates that are settled leave the window: start pixels 0 .. min(ustar, ng - 1) of the group
          {
            unsigned long long mm = m;
            for (int g = 0; g < nrec; ++g) mm &= mm - 1ull;
            m = mm;
          }
          walks += (unsigned long long)nrec;
          nodes += (unsigned long long)ustar;
          rec_n += ustar;
          n_walks += ustar;
          if (ustar == ng) continue;
          // ---- start pixel ustar goes on: its survivors onto the stack, its point and normal to every lane ----
          const int src = ustar * G;
          const bool mine = pass && u == ustar;
          const unsigned long long sm = __ballot(mine);
          if (mine) st.put(__popcll(sm & ((1ull << lane) - 1ull)), qoff, make_uint2((unsigned)q, (unsigned)next | (1u << 16)), qd);
          float wref[3], wrefn[3];
          for (int r = 0; r < 3; ++r) { wref[r] = __shfl(xyz[r], src); wrefn[r] = __shfl(refn[r], src); }
          const unsigned tj = uniform((unsigned)__shfl((int)tau_u, src));
          const unsigned rank = tj * (unsigned)p.T + t;
          const int seed = uniform(__shfl(seed_u, src));
          const float sd = uniform(__shfl(depth_u, src));
          if (lane == 0) *st.win = 0ull;
          __syncthreads();
          int done = walk_turn_wide(p, tb, st, lane, t, tj, win_tau0, rank, seed, sd, &rec_n, &n_walks, &nodes, __popcll(sm), wref, wrefn);
          if (done == 2) {
            if (lane == 0) *st.win = 0ull;
            __syncthreads();
            done = walk_turn(p, tb, st, lane, t, tj, win_tau0, rank, seed, sd, &rec_n, &n_walks, &nodes) ? 1 : 0;
          }
          if (!done) { stop = true; break; }
          __syncthreads();
          m &= ~*st.win;
          __syncthreads();
          continue;
        }
      }
      const int j = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      const unsigned tj = win_tau0 + (unsigned)j;
      const unsigned rank = tj * (unsigned)p.T + t;
      if (rank >= rs) { stop = true; break; }
      const int seed = uniform(__shfl(s, j));
      const float sd = uniform(__shfl(d, j));
      ++walks;
      if (lane == 0) *st.win = 0ull;
      __syncthreads();
      int done = p.wide_group > 0 ? walk_turn_wide(p, tb, st, lane, t, tj, win_tau0, rank, seed, sd, &rec_n, &n_walks, &nodes) : 2;
      if (done == 2) {
        if (lane == 0) *st.win = 0ull;  // (the repeated walk notes its own pixels)
        __syncthreads();
        done = walk_turn(p, tb, st, lane, t, tj, win_tau0, rank, seed, sd, &rec_n, &n_walks, &nodes) ? 1 : 0;
      }
      if (!done) { stop = true; break; }
      __syncthreads();
      m &= ~*st.win;  // start pixels of this window the walk absorbed are no longer free for this thread
      __syncthreads();
    }
    if (!stop && (unsigned long long)next_tau * (unsigned)p.T + t >= (unsigned long long)rs) break;  // nothing from here on can commit in this pass
    tau = next_tau;
  }
  if (lane == 0) {
    p.n_walks[t] = n_walks;
    atomicAdd(&p.ctl->walks, walks);
    atomicAdd(&p.ctl->nodes, nodes);
  }
}

// lanes of ONE wave hand data to each other through LDS / global memory: release + acquire at workgroup scope (the wave
// executes in lock step, so no barrier instruction is needed; the fences keep compiler and memory pipeline in order)
__device__ inline void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <typename T>
__device__ inline T wave_min(T v) {
  for (int o = 32; o > 0; o >>= 1) {
    const T w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}
__device__ inline int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// colmap::Percentile(values, 50) (math/math.h:205-234) of the m values get(0 .. m-1), by the 64 lanes of a wave:
// the two middle order statistics, interpolated in double like the reference. Up to kStage values are staged in
// the wave's LDS column `s` and ranked by counting (value v is the k-th smallest iff #less <= k < #less-or-equal);
// above that an MSB-first radix select on the order-preserving integer key reads them where they are.
template <typename Get>
__device__ double wave_median(int m, Get get, float* s, int lane) {
  const double idx = 0.5 * (double)(m - 1);
  const double lf = floor(idx), rc = ceil(idx);
  const int li = (int)lf, ri = (int)rc;
  float left = 0.0f, right = 0.0f;
  if (m <= kStage) {
    for (int a = lane; a < m; a += kWave) s[a] = get(a);
    wave_fence();
    for (int a0 = 0; a0 < m; a0 += kWave) {
      const int a = a0 + lane;
      const float v = a < m ? s[a] : 0.0f;
      int lt = 0, le = 0;
      for (int b = 0; b < m; ++b) {
        const float w = s[b];
        lt += w < v;
        le += w <= v;
      }
      const unsigned long long ml = __ballot(a < m && lt <= li && li < le);
      const unsigned long long mr = __ballot(a < m && lt <= ri && ri < le);
      if (ml) left = __shfl(v, __ffsll((long long)ml) - 1);
      if (mr) right = __shfl(v, __ffsll((long long)mr) - 1);
    }
    wave_fence();  // the column is refilled by the next call
  } else {
    for (int which = 0; which < 2; ++which) {
      int k = which == 0 ? ri : li;
      if (which == 1 && li == ri) { left = right; break; }
      unsigned prefix = 0u, care = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        care |= 1u << bit;
        int zeros = 0;
        for (int a = lane; a < m; a += kWave) {
          const unsigned u = __float_as_uint(get(a));
          const unsigned key = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
          zeros += ((key & care) == prefix);
        }
        zeros = wave_sum(zeros);
        if (k >= zeros) {
          k -= zeros;
          prefix |= 1u << bit;
        }
      }
      const unsigned u = (prefix >> 31) ? (prefix ^ 0x80000000u) : ~prefix;
      (which == 0 ? right : left) = __uint_as_float(u);
    }
  }
  if (li == ri) return (double)right;
  return (rc - idx) * (double)left + (idx - lf) * (double)right;
}

// A pass, second half: the turns of rank < rstar are final -- their pixels become committed and are fused
// (fusion.cc:449-466, 491-523). Block t = pool thread t; first every recorded pixel gets its point, normal and
// colour (nine float columns), then wave w fuses walks w, w + 4, ...
__global__ void __launch_bounds__(kWave * kCommitWaves) fusion_commit_kernel(Params p) {
  __shared__ float stage[kCommitWaves][kStage];
  const unsigned t = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const unsigned* rec_pix = p.rec_pix + (size_t)t * kRecordBuf;
  const unsigned* rec_meta = p.rec_meta + (size_t)t * kRecordBuf;
  float* const vals = p.vals + (size_t)t * 9 * kRecordBuf;
  const int nw = p.n_walks[t];
  unsigned rs = p.ctl->rstar[p.slot];
  rs = rs < p.limit ? rs : p.limit;
  if (t == 0 && tid == 0) p.ctl->rstar[p.slot ^ 1] = 0xFFFFFFFFu;  // the next pass's slot
  int total = 0;
  if (nw > 0) total = (int)(p.w_first[(size_t)t * p.window_cap + nw - 1] + p.w_count[(size_t)t * p.window_cap + nw - 1]);
  for (int e = tid; e < total; e += kWave * kCommitWaves) {
    const unsigned meta = rec_meta[e];
    if (!(meta >> 31)) continue;
    const int img = (int)(meta & 0xFFFFu), pix = (int)rec_pix[e];
    const DevImage& im = p.images[img];
    const unsigned long long goff = (unsigned long long)im.pix_off + (unsigned long long)pix;
    const int row = pix / im.dw, col = pix - row * im.dw;
    const float depth = p.depth[goff];
    const float* nl = p.normal + 3 * goff;
    const float nl0 = nl[0], nl1 = nl[1], nl2 = nl[2];
    const float hx = (float)col * depth, hy = (float)row * depth;
    for (int r = 0; r < 3; ++r) {
      vals[(size_t)r * kRecordBuf + e] = im.inv_P[4 * r] * hx + im.inv_P[4 * r + 1] * hy + im.inv_P[4 * r + 2] * depth + im.inv_P[4 * r + 3] * 1.0f;
      vals[(size_t)(3 + r) * kRecordBuf + e] = im.inv_R[3 * r] * nl0 + im.inv_R[3 * r + 1] * nl1 + im.inv_R[3 * r + 2] * nl2;
    }
    unsigned rgb = 0u;
    if (im.rgb) {  // nearest neighbour at the bitmap scale (bitmap.cc:329-334), colour 0 outside
      const int xx = (int)round((double)((float)col / im.sx));
      const int yy = (int)round((double)((float)row / im.sy));
      if (xx >= 0 && yy >= 0 && xx < im.bw && yy < im.bh) {
        const uint8_t* c3 = im.rgb + 3 * ((size_t)yy * im.bw + xx);
        rgb = (unsigned)c3[0] | ((unsigned)c3[1] << 8) | ((unsigned)c3[2] << 16);
      }
    }
    for (int ch = 0; ch < 3; ++ch) vals[(size_t)(6 + ch) * kRecordBuf + e] = (float)((rgb >> (8 * ch)) & 0xFFu);
  }
  __syncthreads();
  float* const col = stage[wave];
  for (int wi = wave; wi < nw; wi += kCommitWaves) {
    const unsigned tau = p.w_tau[(size_t)t * p.window_cap + wi];
    if (tau * (unsigned)p.T + t >= rs) continue;  // not final: forgotten with the epoch
    const int first = (int)p.w_first[(size_t)t * p.window_cap + wi], count = (int)p.w_count[(size_t)t * p.window_cap + wi];
    unsigned* const box = p.rec_box + (size_t)t * kRecordBuf + first;
    int m = 0;  // in-box pixels; their record indices compacted into `box`
    for (int e0 = 0; e0 < count; e0 += kWave) {
      const int e = e0 + lane;
      bool inb = false;
      if (e < count) {
        const unsigned meta = rec_meta[first + e];
        const unsigned long long goff = (unsigned long long)p.images[meta & 0xFFFFu].pix_off + (unsigned long long)rec_pix[first + e];
        __hip_atomic_store(p.word + goff, kCommitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        inb = (meta >> 31) != 0u;
      }
      const unsigned long long mk = __ballot(inb);
      const int at = m + __popcll(mk & ((1ull << lane) - 1ull));
      if (inb) box[at] = (unsigned)(first + e);
      m += __popcll(mk);
    }
    if (m < p.min_num_pixels || m == 0) continue;
    wave_fence();  // `box` is written and read by different lanes of this wave
    auto rec_of = [&](int a) -> int { return (int)box[a]; };
    double med[9];
    for (int ch = 0; ch < 9; ++ch) {
      const float* column = vals + (size_t)ch * kRecordBuf;
      med[ch] = wave_median(m, [&](int a) { return column[rec_of(a)]; }, col, lane);
    }
    const float fnx = (float)med[3], fny = (float)med[4], fnz = (float)med[5];
    const float norm = sqrtf(fnx * fnx + fny * fny + fnz * fnz);
    if (norm < FLT_EPSILON) continue;
    // distinct images, ascending (the reference copies an unordered set)
    int nvis = 0;
    for (int last = -1;;) {
      int best = 0x7FFFFFFF;
      for (int a = lane; a < m; a += kWave) {
        const int ia = (int)(rec_meta[rec_of(a)] & 0xFFFFu);
        if (ia > last && ia < best) best = ia;
      }
      best = wave_min(best);
      if (best == 0x7FFFFFFF) break;
      last = best;
      ++nvis;
    }
    unsigned long long off64 = 0ull;
    if (lane == 0) off64 = atomicAdd(&p.ctl->cursor, (unsigned long long)nvis);
    off64 = shfl64(off64, 0);
    const bool fits = off64 + (unsigned long long)nvis <= (unsigned long long)p.pool_cap;  // else: the host fails the run
    const int off = fits ? (int)off64 : 0;
    int last = -1;
    for (int v = 0; v < nvis; ++v) {
      int best = 0x7FFFFFFF;
      for (int a = lane; a < m; a += kWave) {
        const int ia = (int)(rec_meta[rec_of(a)] & 0xFFFFu);
        if (ia > last && ia < best) best = ia;
      }
      best = wave_min(best);
      if (fits && lane == 0) p.pool[off + v] = best;
      last = best;
    }
    if (lane == 0) {
      const int seed = (int)rec_pix[first];  // a walk's first record is its start pixel
      float* out = p.pt + 6 * (size_t)seed;
      out[0] = (float)med[0]; out[1] = (float)med[1]; out[2] = (float)med[2];
      out[3] = fnx / norm; out[4] = fny / norm; out[5] = fnz / norm;
      for (int ch = 0; ch < 3; ++ch) {
        const float v = roundf((float)med[6 + ch]);
        p.col[3 * (size_t)seed + ch] = (unsigned char)fminf(255.0f, fmaxf(0.0f, v));
      }
      p.vis_off[seed] = off;
      p.nvis[seed] = nvis;
      p.valid[seed] = 1;
    }
  }
}

// Output order of an image's points: as the reference collects them -- per pool thread, in the order of its
// turns (fusion.cc:322-337): key = (thread, tick).
__global__ void fusion_keys_kernel(int num_seeds, int W, int T, unsigned L, int G, unsigned* __restrict__ keys,
                                   int* __restrict__ seeds) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= num_seeds) return;
  const int row = s / W, col = s - row * W;
  const int k = row / kRowStride;
  const unsigned tau = (unsigned)(k / T) * L + (unsigned)((row - k * kRowStride) * W + col);
  keys[s] = (unsigned)(k % T) * ((unsigned)G * L) + tau;
  seeds[s] = s;
}

// output order -> compacted output: entry k is the seed at position k of the order
__global__ void fusion_rank_kernel(int num_seeds, const int* __restrict__ order, const int* __restrict__ valid,
                                   const int* __restrict__ nvis, int* __restrict__ valid_r, int* __restrict__ nvis_r) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > num_seeds) return;
  if (k == num_seeds) { valid_r[k] = 0; nvis_r[k] = 0; return; }  // the scans run over num_seeds + 1 entries
  const int s = order[k];
  valid_r[k] = valid[s];
  nvis_r[k] = valid[s] ? nvis[s] : 0;
}

__global__ void fusion_compact_kernel(int num_seeds, int W, int T, const int* __restrict__ order, const int* __restrict__ valid_r,
                                      const int* __restrict__ scan_valid, const int* __restrict__ nvis_r,
                                      const int* __restrict__ scan_vis, const int* __restrict__ vis_off,
                                      const int* __restrict__ pool, const float* __restrict__ pt,
                                      const unsigned char* __restrict__ col, float* __restrict__ out_pt,
                                      unsigned char* __restrict__ out_col, int* __restrict__ out_nvis,
                                      int* __restrict__ out_vis, int* __restrict__ out_thread) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_seeds || !valid_r[k]) return;
  const int s = order[k];
  const int o = scan_valid[k];
  for (int c = 0; c < 6; ++c) out_pt[6 * (size_t)o + c] = pt[6 * (size_t)s + c];
  for (int c = 0; c < 3; ++c) out_col[3 * (size_t)o + c] = col[3 * (size_t)s + c];
  out_nvis[o] = nvis_r[k];
  out_thread[o] = ((s / W) / kRowStride) % T;
  for (int w = 0; w < nvis_r[k]; ++w) out_vis[scan_vis[k] + w] = pool[vis_off[s] + w];
}

__global__ void fusion_premask_kernel(size_t n, const unsigned char* __restrict__ in, unsigned long long* __restrict__ word) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && in[i]) word[i] = kCommitted;  // masked before the first turn
}

// slice-major normal map [3][n] -> xyz triples [n][3]
__global__ void fusion_normal_kernel(size_t n, const float* __restrict__ in, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[3 * i] = in[i];
  out[3 * i + 1] = in[n + i];
  out[3 * i + 2] = in[2 * n + i];
}

__global__ void fusion_ctl_reset_kernel(PassCtl* ctl) {
  ctl->rstar[0] = ctl->rstar[1] = 0xFFFFFFFFu;
  ctl->flags = 0u;
  ctl->redone = 0u;
  ctl->walks = ctl->nodes = ctl->cursor = 0ull;
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
  long long images = 0, seeds = 0, rounds = 0, walks = 0;  // rounds = passes; walks = turns walked (committed or not)
  long long nodes = 0, cuts = 0;                            // pixels recorded by those walks; passes that ended in a cut
  long long redone = 0;                                     // breadth-first walks repeated depth-first
  double upload_seconds = 0.0, device_seconds = 0.0;  // host maps -> HBM + workspace setup | passes, medians, compaction, read-back
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

int EnvInt(const char* name, int dflt) {
  return dev_switch_int(name, dflt);
}

void Run(const fusion_options& opt, int n, const fusion_image* images, const int32_t* optr, const int32_t* oidx,
         fusion_result* out) {
  const auto t_begin = std::chrono::steady_clock::now();
  // development switch COLMAP_AMD_FUSION_TIMING=1: where the set-up time goes (stderr)
  const bool timing = dev_switch_int("COLMAP_AMD_FUSION_TIMING", 0) != 0;
  auto mark = [&, last = t_begin](const char* what) mutable {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "fusion set-up: %-28s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - last).count());
    last = now;
  };
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
  for (int i = 0; i < n; ++i)
    for (int k = optr[i]; k < optr[i + 1]; ++k) FU_CHECK(oidx[k] >= 0 && oidx[k] < n, "overlap index");
  int max_overlap = 1;
  for (int i = 0; i < n; ++i) {
    FU_CHECK(optr[i + 1] - optr[i] < (1 << 20), "overlap list length");
    max_overlap = std::max(max_overlap, optr[i + 1] - optr[i]);
  }

  // resident maps + descriptors
  std::vector<DevImage> h_img(n);
  std::vector<DevBuf<uint8_t>> d_rgb(n);
  long long total_pix = 0;
  int max_seeds = 0, max_threads = 1, max_height = 1;
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
    if (im.rgb) {
      FU_CHECK(im.bitmap_width > 0 && im.bitmap_height > 0, "bitmap size");
      d_rgb[i].upload(im.rgb, 3 * (size_t)im.bitmap_width * im.bitmap_height);
      d.rgb = d_rgb[i].p;
    }
    d.dw = im.depth_width; d.dh = im.depth_height; d.bw = im.bitmap_width; d.bh = im.bitmap_height;
    d.pix_off = total_pix;
    total_pix += (long long)npix;
    max_seeds = std::max(max_seeds, (int)npix);
    max_height = std::max(max_height, im.depth_height);
  }
  // pool threads: one wave each. num_threads <= 0: one thread per stripe (the reference's default pool, all cores, is at
  // least that large for ordinary images and then behaves the same in step).
  auto threads_of = [&](int height) {
    const int ns = (height + kRowStride - 1) / kRowStride;
    return opt.num_threads <= 0 ? ns : std::min(opt.num_threads, ns);
  };
  max_threads = threads_of(max_height);
  mark("descriptors + colour upload");
  // all depth maps / normal maps (as xyz triples) / words in one array each, indexed by the global pixel offset
  DevBuf<float> d_depth, d_normal, d_stage;
  DevBuf<unsigned long long> d_word;
  d_depth.alloc((size_t)total_pix);
  d_normal.alloc(3 * (size_t)total_pix);
  d_stage.alloc(3 * (size_t)max_seeds);
  d_word.alloc((size_t)total_pix);
  FU_HIP(hipMemset(d_word.p, 0, sizeof(unsigned long long) * (size_t)total_pix));
  mark("map allocations + memset");
  for (int i = 0; i < n; ++i) {
    if (!used[i]) continue;
    const size_t npix = (size_t)images[i].depth_width * images[i].depth_height;
    FU_HIP(hipMemcpy(d_depth.p + h_img[i].pix_off, images[i].depth_map, npix * sizeof(float), hipMemcpyHostToDevice));
    FU_HIP(hipMemcpy(d_stage.p, images[i].normal_map, 3 * npix * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fusion_normal_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, npix, d_stage.p,
                       d_normal.p + 3 * (size_t)h_img[i].pix_off);
    if (images[i].mask) {
      DevBuf<uint8_t> m;
      m.upload(images[i].mask, npix);
      hipLaunchKernelGGL(fusion_premask_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, npix, m.p,
                         d_word.p + h_img[i].pix_off);
      FU_HIP(hipDeviceSynchronize());
    }
    FU_HIP(hipDeviceSynchronize());  // the staging buffer is reused by the next image
  }
  d_stage.release();
  mark("depth / normal upload");
  // The visibility pool is refilled per reference image (cursor reset every step) and its int offsets only have to cover
  // what ONE image's walks absorb: capacity min(total pixels, 2^31 - 1), an overflow fails the run instead of wrapping.
  const long long pool_cap = std::min<long long>(total_pix, 0x7FFFFFFFll);
  DevBuf<DevImage> d_img;
  d_img.upload(h_img.data(), h_img.size());
  DevBuf<int> d_optr, d_oidx;
  d_optr.upload(optr, (size_t)n + 1);
  d_oidx.upload(oidx, (size_t)optr[n]);

  Params p;
  std::memset(&p, 0, sizeof(p));
  p.images = d_img.p; p.optr = d_optr.p; p.oidx = d_oidx.p; p.word = d_word.p; p.depth = d_depth.p; p.normal = d_normal.p;
  p.rec_cap = (int)std::min<long long>(record_capacity(opt.max_num_pixels), std::max<long long>(total_pix, 1));
  p.elem_cap = std::min(opt.max_num_pixels, p.rec_cap);
  FU_CHECK(p.rec_cap <= kRecordBuf, "record buffer of a wave smaller than the record capacity of one walk");
  p.max_level = opt.max_traversal_depth - 1;
  p.min_num_pixels = opt.min_num_pixels;
  p.max_depth_error = opt.max_depth_error;
  p.max_sq_reproj = static_cast<float>(opt.max_reproj_error * opt.max_reproj_error);
  p.min_cos_normal = static_cast<float>(std::cos(opt.max_normal_error * 0.017453292519943295769));
  for (int c = 0; c < 3; ++c) { p.bmin[c] = opt.bbox_min[c]; p.bmax[c] = opt.bbox_max[c]; }
  // per-wave state. Schedule knobs for experiments (the result does not depend on them: bit-exact against the
  // sequential algorithm for any window): COLMAP_AMD_FUSION_WINDOW_FIRST / _MAX.
  const int window_first = std::max(1, EnvInt("COLMAP_AMD_FUSION_WINDOW_FIRST", kWindowFirst));
  const int window_max = std::max(window_first, EnvInt("COLMAP_AMD_FUSION_WINDOW_MAX", kWindowMax));
  const size_t TT = (size_t)max_threads;
  DevBuf<unsigned> rec_pix, rec_meta, rec_box, w_tau, w_first, w_count;
  DevBuf<int> n_walks;
  DevBuf<float> vals;
  DevBuf<unsigned long long> spill_goff;
  DevBuf<uint2> spill_pm;
  DevBuf<float> spill_d;
  DevBuf<PassCtl> d_ctl;
  rec_pix.alloc(TT * kRecordBuf); rec_meta.alloc(TT * kRecordBuf); rec_box.alloc(TT * kRecordBuf);
  w_tau.alloc(TT * window_max); w_first.alloc(TT * window_max); w_count.alloc(TT * window_max);
  n_walks.alloc(TT); vals.alloc(TT * 9 * kRecordBuf);
  int spill_cap = kStackSpill;
  spill_goff.alloc(TT * spill_cap); spill_pm.alloc(TT * spill_cap); spill_d.alloc(TT * spill_cap);
  d_ctl.alloc(1);
  p.rec_pix = rec_pix.p; p.rec_meta = rec_meta.p; p.rec_box = rec_box.p;
  p.w_tau = w_tau.p; p.w_first = w_first.p; p.w_count = w_count.p; p.n_walks = n_walks.p; p.window_cap = window_max;
  p.vals = vals.p; p.spill_goff = spill_goff.p; p.spill_pm = spill_pm.p; p.spill_d = spill_d.p; p.spill_cap = spill_cap; p.ctl = d_ctl.p;
  // LDS copies of the walk kernel: descriptors + overlap offsets, and the overlap lists, when they fit
  p.n_images = n;
  p.n_overlap = optr[n];
  {
    const size_t desc = (size_t)n * sizeof(DevImage) + ((size_t)n + 1) * sizeof(int);
    p.lds_tables = desc > (size_t)kTableBytes ? 0 : (desc + (size_t)optr[n] * sizeof(int) > (size_t)kTableBytes ? 1 : 2);
    p.lds_tables = std::min(p.lds_tables, std::max(0, dev_switch_int("COLMAP_AMD_FUSION_LDS_TABLES", 2)));
  }
  // breadth-first walks (walk_turn_wide) where their result provably equals the depth-first one (COLMAP_AMD_FUSION_WIDE=0:
  // depth-first only; tests compare both)
  {
    const int bound = std::min(std::min(p.max_level, p.elem_cap - 1), p.rec_cap);
    const bool wide = dev_switch_int("COLMAP_AMD_FUSION_WIDE", 1) != 0 && max_overlap <= kWave / 2 && bound >= 16;
    p.wide_group = wide ? std::max(max_overlap, 1) : 0;
    p.wide_bound = bound;
  }
  mark("per-wave state allocations");
  // a stack can never hold more than (pixels a walk records) x (longest overlap list) entries
  const long long spill_bound = (long long)p.rec_cap * max_overlap + kWave;

  DevBuf<int> valid, nvis, vis_off, valid_r, nvis_r, scan_valid, scan_vis, pool, out_nvis, out_vis, out_thread;
  DevBuf<float> pt, out_pt;
  DevBuf<unsigned char> col, out_col;
  const size_t ms = (size_t)max_seeds;
  valid.alloc(ms); nvis.alloc(ms); vis_off.alloc(ms); valid_r.alloc(ms + 1); nvis_r.alloc(ms + 1);
  scan_valid.alloc(ms + 1); scan_vis.alloc(ms + 1);
  pool.alloc((size_t)pool_cap); out_nvis.alloc(ms); out_vis.alloc((size_t)pool_cap); out_thread.alloc(ms);
  p.pool_cap = pool_cap;
  pt.alloc(6 * ms); out_pt.alloc(6 * ms); col.alloc(3 * ms); out_col.alloc(3 * ms);
  p.valid = valid.p; p.nvis = nvis.p; p.vis_off = vis_off.p; p.pt = pt.p; p.col = col.p; p.pool = pool.p;
  size_t tmp_bytes = 0;
  FU_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, valid_r.p, scan_valid.p, (int)ms + 1));
  DevBuf<unsigned char> tmp;
  tmp.alloc(tmp_bytes + 16);
  DevBuf<unsigned> keys_in, keys_out;
  DevBuf<int> seeds_in, order;
  keys_in.alloc(ms); keys_out.alloc(ms); seeds_in.alloc(ms); order.alloc(ms);
  size_t sort_bytes = 0;
  FU_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_in.p, keys_out.p, seeds_in.p, order.p, (int)ms));
  DevBuf<unsigned char> sort_tmp;
  sort_tmp.alloc(sort_bytes + 16);

  // the points of every image, in (thread, tick) order, with their thread; concatenated per thread at the end
  struct Chunk {
    std::vector<float> pt;
    std::vector<unsigned char> col;
    std::vector<int> nvis, vis, thread;
  };
  std::vector<Chunk> chunks;
  unsigned epoch = 1;  // 0 would make the free word look like a mark
  mark("scratch allocations");
  g_stats = Stats();
  FU_HIP(hipDeviceSynchronize());
  const auto t_setup = std::chrono::steady_clock::now();
  g_stats.upload_seconds = std::chrono::duration<double>(t_setup - t_begin).count();
  for (int step = 0; step < (int)order_of_images.size(); ++step) {
    const int I = order_of_images[step];
    const int W = h_img[I].dw, H = h_img[I].dh, ns_px = W * H;
    const int ns = (H + kRowStride - 1) / kRowStride, T = threads_of(H), G = (ns + T - 1) / T;
    const unsigned long long L = (unsigned long long)kRowStride * W, ticks = (unsigned long long)G * L;
    const unsigned long long r_end = ticks * (unsigned long long)T;
    FU_CHECK(r_end < 0xFFFFFFF0ull, "turns of one image < 2^32");
    p.step = step; p.image = I; p.T = T; p.W = W; p.H = H; p.ns = ns; p.L = (unsigned)L;
    hipLaunchKernelGGL(fusion_ctl_reset_kernel, dim3(1), dim3(1), 0, 0, d_ctl.p);
    FU_HIP(hipMemsetAsync(valid.p, 0, sizeof(int) * (size_t)ns_px, 0));
    // passes: ranks [r_next, limit) are walked speculatively, [r_next, rstar) commit
    unsigned long long r_next = 0;
    long long window = window_first;
    PassCtl h_ctl;
    std::memset(&h_ctl, 0, sizeof(h_ctl));
    for (int pass = 0; r_next < r_end; ++pass, ++epoch) {
      FU_CHECK(epoch < 0xFFFFFFFEu, "epoch counter");
      const unsigned long long tau0 = r_next / (unsigned long long)T;
      const unsigned long long tau_end = std::min<unsigned long long>(tau0 + (unsigned long long)window, ticks);
      p.epoch = epoch; p.slot = pass & 1;
      p.tau0 = (unsigned)tau0; p.rmod = (unsigned)(r_next % (unsigned long long)T);
      p.tau_end = (unsigned)tau_end; p.limit = (unsigned)(tau_end * (unsigned long long)T);
      hipLaunchKernelGGL(fusion_walk_kernel, dim3(T), dim3(kWave), 0, 0, p);
      hipLaunchKernelGGL(fusion_commit_kernel, dim3(T), dim3(kWave * kCommitWaves), 0, 0, p);
      FU_HIP(hipMemcpy(&h_ctl, d_ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost));
      FU_HIP(hipGetLastError());
      const unsigned long long rstar = std::min<unsigned long long>(h_ctl.rstar[pass & 1], p.limit);
      g_stats.rounds += 1;
      if (h_ctl.flags & 1u) {  // a walk overflowed the stack spill: it cut the pass at its own rank; give it room
        FU_CHECK((long long)spill_cap < spill_bound, "stack overflow beyond its bound");
        spill_cap = (int)std::min<long long>(4ll * spill_cap, spill_bound);
        spill_goff.alloc(TT * spill_cap); spill_pm.alloc(TT * spill_cap); spill_d.alloc(TT * spill_cap);
        p.spill_goff = spill_goff.p; p.spill_pm = spill_pm.p; p.spill_d = spill_d.p; p.spill_cap = spill_cap;
        FU_HIP(hipMemsetAsync(&d_ctl.p->flags, 0, sizeof(unsigned), 0));
      } else {
        FU_CHECK(rstar > r_next, "pass made no progress");
      }
      const bool cut = rstar < (unsigned long long)p.limit;
      if (cut) g_stats.cuts += 1;
      window = cut ? std::max<long long>(kWindowMin, window / 2) : std::min<long long>(window_max, 2 * window);
      r_next = rstar;
    }
    g_stats.walks += (long long)h_ctl.walks;
    g_stats.nodes += (long long)h_ctl.nodes;
    g_stats.redone += (long long)h_ctl.redone;
    FU_CHECK(h_ctl.cursor <= (unsigned long long)pool_cap, "visibility pool overflow (more than 2^31 - 1 visibility entries for one reference image)");
    // output order of this image: (thread, tick)
    hipLaunchKernelGGL(fusion_keys_kernel, dim3((ns_px + 255) / 256), dim3(256), 0, 0, ns_px, W, T, (unsigned)L, G, keys_in.p, seeds_in.p);
    size_t sb = sort_bytes;
    FU_HIP(hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, sb, keys_in.p, keys_out.p, seeds_in.p, order.p, ns_px));
    hipLaunchKernelGGL(fusion_rank_kernel, dim3((ns_px + 256) / 256), dim3(256), 0, 0, ns_px, order.p, valid.p, nvis.p,
                       valid_r.p, nvis_r.p);
    size_t tb = tmp_bytes;
    FU_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, valid_r.p, scan_valid.p, ns_px + 1));
    tb = tmp_bytes;
    FU_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, nvis_r.p, scan_vis.p, ns_px + 1));
    hipLaunchKernelGGL(fusion_compact_kernel, dim3((ns_px + 255) / 256), dim3(256), 0, 0, ns_px, W, T, order.p, valid_r.p,
                       scan_valid.p, nvis_r.p, scan_vis.p, vis_off.p, pool.p, pt.p, col.p, out_pt.p, out_col.p,
                       out_nvis.p, out_vis.p, out_thread.p);
    int totals[2] = {0, 0};
    FU_HIP(hipMemcpy(&totals[0], scan_valid.p + ns_px, sizeof(int), hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(&totals[1], scan_vis.p + ns_px, sizeof(int), hipMemcpyDeviceToHost));
    FU_HIP(hipGetLastError());
    g_stats.images += 1;
    g_stats.seeds += ns_px;
    const size_t np = (size_t)totals[0], nv = (size_t)totals[1];
    if (np == 0) continue;
    chunks.emplace_back();
    Chunk& c = chunks.back();
    c.pt.resize(6 * np); c.col.resize(3 * np); c.nvis.resize(np); c.vis.resize(nv); c.thread.resize(np);
    FU_HIP(hipMemcpy(c.pt.data(), out_pt.p, sizeof(float) * 6 * np, hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(c.col.data(), out_col.p, 3 * np, hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(c.nvis.data(), out_nvis.p, sizeof(int) * np, hipMemcpyDeviceToHost));
    FU_HIP(hipMemcpy(c.thread.data(), out_thread.p, sizeof(int) * np, hipMemcpyDeviceToHost));
    if (nv) FU_HIP(hipMemcpy(c.vis.data(), out_vis.p, sizeof(int) * nv, hipMemcpyDeviceToHost));
  }
  // task_fused_points_[thread] concatenated over the threads (fusion.cc:322-337): every chunk is sorted by thread
  std::vector<size_t> at(chunks.size(), 0), vat(chunks.size(), 0);
  for (int t = 0; t < max_threads; ++t) {
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      const Chunk& c = chunks[ci];
      const size_t b = at[ci];
      size_t e = b, nv = 0;
      while (e < c.thread.size() && c.thread[e] == t) nv += (size_t)c.nvis[e++];
      if (e == b) continue;
      out->xyz_normal.insert(out->xyz_normal.end(), c.pt.begin() + 6 * b, c.pt.begin() + 6 * e);
      out->rgb.insert(out->rgb.end(), c.col.begin() + 3 * b, c.col.begin() + 3 * e);
      out->vis_idx.insert(out->vis_idx.end(), c.vis.begin() + vat[ci], c.vis.begin() + vat[ci] + nv);
      int64_t base = out->vis_ptr.back();
      for (size_t k = b; k < e; ++k) {
        base += c.nvis[k];
        out->vis_ptr.push_back(base);
      }
      at[ci] = e;
      vat[ci] += nv;
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
  o->num_threads = -1;
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

FUSION_API int64_t fusion_last_redone_walks(void) { return g_stats.redone; }

FUSION_API const char* fusion_last_error(void) { return g_error.c_str(); }

}  // extern "C"
