// ba_schur_explicit.hip -- exact linear-solver tiers (DENSE_SCHUR / SPARSE_SCHUR) of the bundle-adjustment
// backend for gfx950: the reduced camera system S = B + Dc^2 - E C^-1 E^T formed EXPLICITLY on the device and
// solved by a blocked Cholesky factorisation whose panel and trailing-update contractions run on the f64
// matrix cores (v_mfma_f64_16x16x4_f64).
//
// What it replaces: Ceres' SchurEliminator + dense / sparse Cholesky of the reduced camera matrix as COLMAP
// selects them by problem size (reference estimators/bundle_adjustment_ceres.cc:203-213, thresholds
// bundle_adjustment_ceres.h:68-71). Restated for the CPU in oracle/ba_oracle.c: explicit_schur_solve.
//
// MI355X-first choices:
//  * S is stored DENSE in HBM whatever its block sparsity (n_c = 8 000 at 1 000 images is 512 MB, n_c = 32 768
//    is 8.6 GB of 288 GB); sparsity is exploited where it costs -- in the formation, which touches only the
//    camera pairs a point connects -- and the factorisation is a dense GEMM-shaped job for the matrix cores
//    instead of a sparse supernodal one.
//  * Formation, pair-major (the default; FormArgs::pairs): the contribution of an ordered pair (a, b) of
//    observations of one point is  J_a^T (delta_ab I - E_a C^-1 E_b^T) J_b = delta_ab J_a^T J_a - F_a G_b^T  with
//    F = J^T (E C^-1), G = J^T E (w x 3 each). One kernel writes a record {F, G, J, tangent indices} per observation;
//    the incidences -- the unordered observation pairs of every variable point plus a self pair per observation --
//    are listed ONCE per solve on the device, sorted by the pair of pose blocks (hipCUB radix sort, stable: the
//    order is deterministic); then one wave per 64 consecutive incidences, lane (r, c) accumulating
//    -F_a[r] . G_b[c] in a register while the two images stay the same and adding the sum to S once per run.
//    At 1 000 images x 200 000 points x track 10: 1.1e7 incidences, 4 x 16-byte loads per incidence and lane (the F
//    block of one 704-byte record, the G block of the other), ~4e7 atomics -- against 6.4e8 atomics (one per term) for
//  * the point-major formation (COLMAP_AMD_BA_FORM_PAIRS=0, and the fallback when the lists cannot be built): one
//    wave per 3-D point stages the observations of the point in LDS and its lanes walk the (a, i, b, k) element
//    space with the column index fastest. Only the lower triangle is written.
//    In both, the atomics are INTEGER adds of 2^-60 fixed-point values (FIXED): order-independent, so this tier is
//    bit-reproducible like the iterative one (the pair-major sums inside a run are in list order).
//  * Factorisation: right-looking, outer panels of 256 columns. Inside a panel 64-wide steps on the panel's own rows only
//    -- diagonal block: two waves with the matrices in registers (the factorisation in one, the inverse of the triangle
//    in the other), so that the panel solve below it becomes a GEMM  X = A_panel L_kk^-T --, then the strip below the
//    panel in one pass (chol_strip_kernel: a workgroup's 64 x 256 tile stays in the accumulators through the block-column
//    recursion), then the trailing update C_IJ -= X_I X_J^T in 128 x 128 (4 x 4 MFMA tiles per wave, next K chunk
//    prefetched into registers) or 64 x 64 tiles per workgroup, accumulators initialised from the tile.
//  * Triangular solves with the stored block inverses: the forward one rides along with the factorisation (the
//    right-hand side is row n of the matrix), the backward one takes one launch per outer panel of 256 columns.
#include "ba_schur_explicit.h"
#include "switches.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <climits>
#include <cmath>
#include <stdexcept>
#include <string>

namespace ba_explicit {

namespace {

#define BAX_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)

constexpr int NB = 64;  // panel width = tile size
typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------
// Formation
// ---------------------------------------------------------------------------------------------------------
template <int WMAX>
struct ObsSlot {
  double jc[2][WMAX];  // camera-side columns (pose | intrinsics | sensor), scaled as the solver stores them
  double jp[2][3];     // point block E_a
  double pj[2][3];     // E_a C^-1
  int idx[WMAX];       // tangent index of every column
  int w;
};

template <int WMAX>
__device__ __forceinline__ void load_slot(const FormArgs& A, ObsSlot<WMAX>& s, int a, const double* Ci, bool var) {
  const size_t N = (size_t)A.n_obs;
  const int c = A.a2c[a];
  const int pi = A.a_pose[a], ci = A.a_cam[a];
  const int si = A.a_sensor ? A.a_sensor[a] : -1;
  const int po = A.pose_off[pi], co = A.cam_off[ci];
  const int so = (si >= 0 && A.sens_off) ? A.sens_off[si] : -1;
  int w = 0;
  if (po >= 0) {
    const int pdim = A.pose_dim[pi];
    for (int d = 0; d < pdim; ++d) {
      s.jc[0][w] = A.Jpose[(size_t)d * N + c];
      s.jc[1][w] = A.Jpose[(size_t)(kPoseDim + d) * N + c];
      s.idx[w++] = po + d;
    }
  }
  if (co >= 0) {
    const int cdim = A.cam_dim[ci];
    for (int d = 0; d < cdim; ++d) {
      s.jc[0][w] = A.Jcam[(size_t)d * N + c];
      s.jc[1][w] = A.Jcam[(size_t)(A.kd + d) * N + c];
      s.idx[w++] = co + d;
    }
  }
  if (so >= 0) {
    for (int d = 0; d < 6; ++d) {
      s.jc[0][w] = A.Jsens[(size_t)d * N + c];
      s.jc[1][w] = A.Jsens[(size_t)(6 + d) * N + c];
      s.idx[w++] = so + d;
    }
  }
  s.w = w;
  for (int r = 0; r < 2; ++r) {
    double e[3];
    for (int m = 0; m < 3; ++m) e[m] = A.Jpt[(size_t)(r * 3 + m) * N + a];
    for (int m = 0; m < 3; ++m) {
      s.jp[r][m] = e[m];
      s.pj[r][m] = var ? e[0] * Ci[m] + e[1] * Ci[3 + m] + e[2] * Ci[6 + m] : 0.0;
    }
  }
}

// FIXED: contributions are accumulated as 64-bit fixed-point numbers (2^-60 units) with INTEGER atomics --
// integer addition is associative, so the sum does not depend on the order the hardware serves the atomics
// in and the whole exact tier is bit-reproducible run to run. With Jacobi scaling every column of the
// Jacobian has norm < 1, so every entry of B - E C^-1 E^T and every partial sum of it is bounded by 1
// (Cauchy-Schwarz): +-8 of range is ample and 2^-60 = 8.7e-19 is finer than the fp64 rounding of the terms.
// Without Jacobi scaling (ba_options.jacobi_scaling = 0) there is no such bound: hardware fp64 atomics then
// (reproducible to rounding only).
constexpr double kFixedScale = 1152921504606846976.0;  // 2^60
constexpr double kFixedTermBound = 2.0;  // |term| < 1 under Jacobi scaling; NaN, Inf and anything beyond 2 raise FormArgs::bad

template <int WMAX, bool FIXED>
__global__ void __launch_bounds__(64) form_kernel(FormArgs A, double* __restrict__ S) {
  constexpr int CH = 16;  // observations of a point staged per chunk
  __shared__ ObsSlot<WMAX> sa[CH], sb[CH];
  __shared__ int s_wmax[2];
  const int j = blockIdx.x;
  const int beg = A.pt_ptr[j], t = A.pt_ptr[j + 1] - beg;
  if (t == 0) return;
  const bool var = A.pt_off[j] >= 0;
  double Ci[9];
  for (int m = 0; m < 9; ++m) Ci[m] = var ? A.Cinv[9 * (size_t)j + m] : 0.0;
  const int lane = threadIdx.x;
  const size_t n = (size_t)A.n_c;
  for (int a0 = 0; a0 < t; a0 += CH) {
    const int na = min(CH, t - a0);
    __syncthreads();
    if (lane < na) load_slot<WMAX>(A, sa[lane], beg + a0 + lane, Ci, var);
    __syncthreads();
    if (lane == 0) {
      int wm = 0;
      for (int q = 0; q < na; ++q) wm = max(wm, sa[q].w);
      s_wmax[0] = wm;
    }
    for (int b0 = 0; b0 < t; b0 += CH) {
      if (!var && b0 != a0) continue;  // a constant point couples nothing: only J_a^T J_a
      const int nb = min(CH, t - b0);
      __syncthreads();
      if (lane < nb) load_slot<WMAX>(A, sb[lane], beg + b0 + lane, Ci, var);
      __syncthreads();
      if (lane == 0) {
        int wm = 0;
        for (int q = 0; q < nb; ++q) wm = max(wm, sb[q].w);
        s_wmax[1] = wm;
      }
      __syncthreads();
      const int wa = s_wmax[0], wb = s_wmax[1];
      const int total = na * wa * nb * wb;
      for (int e = lane; e < total; e += 64) {
        const int k = e % wb;
        int q = e / wb;
        const int b = q % nb;
        q /= nb;
        const int i = q % wa;
        const int a = q / wa;
        const ObsSlot<WMAX>& oa = sa[a];
        const ObsSlot<WMAX>& ob = sb[b];
        if (i >= oa.w || k >= ob.w) continue;
        const int row = oa.idx[i], col = ob.idx[k];
        if (row < col) continue;  // lower triangle only
        const bool self = (a0 + a) == (b0 + b);
        if (!var && !self) continue;
        double m00 = self ? 1.0 : 0.0, m01 = 0.0, m10 = 0.0, m11 = self ? 1.0 : 0.0;
        if (var) {
          m00 -= oa.pj[0][0] * ob.jp[0][0] + oa.pj[0][1] * ob.jp[0][1] + oa.pj[0][2] * ob.jp[0][2];
          m01 -= oa.pj[0][0] * ob.jp[1][0] + oa.pj[0][1] * ob.jp[1][1] + oa.pj[0][2] * ob.jp[1][2];
          m10 -= oa.pj[1][0] * ob.jp[0][0] + oa.pj[1][1] * ob.jp[0][1] + oa.pj[1][2] * ob.jp[0][2];
          m11 -= oa.pj[1][0] * ob.jp[1][0] + oa.pj[1][1] * ob.jp[1][1] + oa.pj[1][2] * ob.jp[1][2];
        }
        const double b0v = ob.jc[0][k], b1v = ob.jc[1][k];
        const double val = oa.jc[0][i] * (m00 * b0v + m01 * b1v) + oa.jc[1][i] * (m10 * b0v + m11 * b1v);
        if (FIXED) {
          if (!(fabs(val) < kFixedTermBound) && A.bad) *A.bad = 1;  // (every writer stores the same value)
          atomicAdd(reinterpret_cast<unsigned long long*>(S) + (size_t)row * n + col,
                    (unsigned long long)(long long)__double2ll_rn(val * kFixedScale));
        } else {
          unsafeAtomicAdd(S + (size_t)row * n + col, val);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Pair-major formation (FormArgs::pairs): records, incidence lists, one wave per 64 incidences
// ---------------------------------------------------------------------------------------------------------
// Record of an observation: two blocks of W rows of 4 doubles, row r = column r of its camera-side Jacobian J (2 x w):
//   F block: { F[r][0..2], J[0][r] }      G block: { G[r][0..2], J[1][r] }      F = J^T (E C^-1), G = J^T E (w x 3 each)
// (rows at FIXED slots: pose columns 0..5, intrinsics 6..6+kd-1, sensor_from_rig columns after them; a column the
// observation does not have is a zero row), then the tangent index of every row as ints (-1: no such column; padded to
// a multiple of four) and four ints {pose, camera, sensor_from_rig, number of columns}. With it the contribution of an ordered pair (a, b) of observations
// of one point is  J_a^T (delta_ab I - E_a C^-1 E_b^T) J_b = delta_ab J_a^T J_a - F_a G_b^T: a pair reads the F block
// of one record and the G block of the other -- each a contiguous 32 W bytes (the two halves of a row side by side
// would leave half of every fetched line unused).
constexpr int kIncBatch = 4;  // incidences staged per batch by form_pairs_kernel
constexpr int rec_rows4(int W) { return (W + 3) / 4 * 4; }
constexpr int rec_stride(int W) { return 8 * W + (rec_rows4(W) + 4) / 2; }  // doubles; even: records stay 16-byte aligned
inline int pick_width(const FormArgs& a) {  // the instantiated widths of both formations
  const int wmax = kPoseDim + a.kd + (a.Jsens ? 6 : 0);
  return wmax <= 10 ? 10 : wmax <= 14 ? 14 : wmax <= 20 ? 20 : 28;
}

// A lane builds the record of one observation in LDS; the workgroup then writes its records -- one contiguous piece of
// kRecThreads x 704 bytes at W = 10 -- with coalesced 16-byte stores. (A lane storing its own record straight to HBM
// wrote 44 16-byte pieces at a 704-byte stride from its neighbours': 1.17 ms for 1.4 GB at BA-1, 1.2 TB/s.)
template <int W>
constexpr int rec_threads() { return W <= 10 ? 64 : 32; }  // records per workgroup: kRecThreads x rec_stride x 8 B of LDS
template <int W>
__global__ void __launch_bounds__(rec_threads<W>()) form_records_kernel(FormArgs A, double* __restrict__ rec) {
  constexpr int ST = rec_stride(W), W4 = rec_rows4(W), TB = rec_threads<W>();
  __shared__ __attribute__((aligned(16))) double stage[TB * ST];
  const int a0 = blockIdx.x * TB;
  const bool live = a0 + (int)threadIdx.x < A.n_obs;
  const int a = live ? a0 + (int)threadIdx.x : A.n_obs - 1;  // (an idle lane rebuilds the last record and drops it)
  const size_t N = (size_t)A.n_obs;
  const int j = A.a_pt[a];
  const bool var = A.pt_off[j] >= 0;
  double e[2][3], pj[2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int m = 0; m < 3; ++m) e[r][m] = A.Jpt[(size_t)(r * 3 + m) * N + a];
  double Ci[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) Ci[m] = var ? A.Cinv[9 * (size_t)j + m] : 0.0;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int m = 0; m < 3; ++m) pj[r][m] = var ? e[r][0] * Ci[m] + e[r][1] * Ci[3 + m] + e[r][2] * Ci[6 + m] : 0.0;
  double* R = stage + (size_t)threadIdx.x * ST;
  int* RI = reinterpret_cast<int*>(R + 8 * W);
  const int c = A.a2c[a];
  const int pi = A.a_pose[a], ci = A.a_cam[a];
  const int si = A.a_sensor ? A.a_sensor[a] : -1;
  const int po = A.pose_off[pi], co = A.cam_off[ci];
  const int so = (si >= 0 && A.sens_off) ? A.sens_off[si] : -1;
  // Fixed slots: row d = pose column d, row 6 + d = intrinsics column d, row 6 + kd + d = sensor column d; a column the
  // observation does not have (constant block, 5-wide pose, fewer refined intrinsics) is a zero row with index -1. Every
  // loop has a compile-time trip count: all loads of a lane are in flight together (rolled over the block's dimension,
  // each trip waited for its own pair of loads -- 8 to 20 serial round trips per lane).
  int wv = 0;
  auto put = [&](int slot, bool ok, double j0, double j1, int idx) {
    double2* fr = reinterpret_cast<double2*>(R + 4 * slot);
    double2* gr = reinterpret_cast<double2*>(R + 4 * W + 4 * slot);
    fr[0] = make_double2(j0 * pj[0][0] + j1 * pj[1][0], j0 * pj[0][1] + j1 * pj[1][1]);
    fr[1] = make_double2(j0 * pj[0][2] + j1 * pj[1][2], j0);
    gr[0] = make_double2(j0 * e[0][0] + j1 * e[1][0], j0 * e[0][1] + j1 * e[1][1]);
    gr[1] = make_double2(j0 * e[0][2] + j1 * e[1][2], j1);
    RI[slot] = ok ? idx : -1;
    wv += ok ? 1 : 0;
  };
  const int pdim = po >= 0 ? A.pose_dim[pi] : 0;
  const int cdim = co >= 0 ? A.cam_dim[ci] : 0;
  {
    double j0[kPoseDim], j1[kPoseDim];
#pragma unroll
    for (int d = 0; d < kPoseDim; ++d) {
      j0[d] = d < pdim ? A.Jpose[(size_t)d * N + c] : 0.0;
      j1[d] = d < pdim ? A.Jpose[(size_t)(kPoseDim + d) * N + c] : 0.0;
    }
#pragma unroll
    for (int d = 0; d < kPoseDim; ++d) put(d, d < pdim, j0[d], j1[d], po + d);
  }
  {
    constexpr int KDMAX = W - kPoseDim;  // (>= kd for every instance: pick_width)
    double j0[KDMAX], j1[KDMAX];
#pragma unroll
    for (int d = 0; d < KDMAX; ++d) {
      j0[d] = d < cdim ? A.Jcam[(size_t)d * N + c] : 0.0;
      j1[d] = d < cdim ? A.Jcam[(size_t)(A.kd + d) * N + c] : 0.0;
    }
#pragma unroll
    for (int d = 0; d < KDMAX; ++d)
      if (d < A.kd) put(kPoseDim + d, d < cdim, j0[d], j1[d], co + d);
  }
  const int s0 = kPoseDim + A.kd;  // first sensor slot
  if (A.Jsens) {
    double j0[6], j1[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      j0[d] = so >= 0 ? A.Jsens[(size_t)d * N + c] : 0.0;
      j1[d] = so >= 0 ? A.Jsens[(size_t)(6 + d) * N + c] : 0.0;
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) put(s0 + d, so >= 0, j0[d], j1[d], so + d);
  }
  for (int r = s0 + (A.Jsens ? 6 : 0); r < W; ++r) put(r, false, 0.0, 0.0, -1);  // (slots the problem never uses)
  for (int r = W; r < W4; ++r) RI[r] = -1;
  RI[W4] = pi;
  RI[W4 + 1] = ci;
  RI[W4 + 2] = si;
  RI[W4 + 3] = wv;
  __syncthreads();
  const int nrec = min(TB, A.n_obs - a0);
  const double2* src = reinterpret_cast<const double2*>(stage);
  double2* dst = reinterpret_cast<double2*>(rec + (size_t)a0 * ST);  // (ST is even: records stay 16-byte aligned)
  for (int e = threadIdx.x; e < nrec * (ST / 2); e += TB) dst[e] = src[e];
}

// Number of incidences of every point (self pairs of all its observations; the unordered pairs too when the point
// is variable -- a constant point couples nothing), n_points + 1 entries for the exclusive scan.
__global__ void inc_count_kernel(FormArgs A, unsigned long long* __restrict__ cnt) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > A.n_points) return;
  unsigned long long v = 0ull;
  if (j < A.n_points) {
    const unsigned long long t = (unsigned long long)(A.pt_ptr[j + 1] - A.pt_ptr[j]);
    v = A.pt_off[j] >= 0 ? t * (t + 1) / 2 : t;
  }
  cnt[j] = v;
}

// The incidences of point j at off[j] ...: key = the unordered pair of pose blocks as a triangular index, value =
// the two p-order slots, the observation of the higher pose block first.
__global__ void inc_emit_kernel(FormArgs A, const unsigned long long* __restrict__ off, unsigned* __restrict__ keys,
                                unsigned long long* __restrict__ vals) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.n_points) return;
  const int beg = A.pt_ptr[j], t = A.pt_ptr[j + 1] - beg;
  const bool var = A.pt_off[j] >= 0;
  unsigned long long o = off[j];
  auto tri = [](unsigned long long hi, unsigned long long lo) { return (unsigned)(hi * (hi + 1) / 2 + lo); };
  for (int a = 0; a < t; ++a) {
    const unsigned long long sa = (unsigned long long)(beg + a);
    const unsigned long long pa = (unsigned long long)A.a_pose[beg + a];
    keys[o] = tri(pa, pa);
    vals[o] = (sa << 32) | sa;
    ++o;
    if (!var) continue;
    for (int b = a + 1; b < t; ++b) {
      const unsigned long long sb = (unsigned long long)(beg + b);
      const unsigned long long pb = (unsigned long long)A.a_pose[beg + b];
      const bool a_first = pa >= pb;
      keys[o] = a_first ? tri(pa, pb) : tri(pb, pa);
      vals[o] = a_first ? ((sa << 32) | sb) : ((sb << 32) | sa);
      ++o;
    }
  }
}

// One wave per 64 consecutive incidences. Lane -> the entries e = lane + 64 j of the W x W block, e = (r, c): row r of
// the first observation's record against row c of the second's. A run = consecutive incidences whose two observations
// have the same (pose, camera, sensor) triples (normally: the same two images): the targets in S are the same, the sum
// stays in a register and goes to S once, with one (integer, for FIXED) atomic add per entry -- runs of one pair of
// blocks can straddle waves, and blocks shared between images (intrinsics, rig poses) are reached from many pairs.
// Lower triangle: an unordered pair {a, b} stands for both ordered pairs, (a, b) at (idx_a[r], idx_b[c]) and its
// transpose; exactly one of the two lies in the lower triangle unless the indices are equal (a shared block's
// diagonal), where both do: weight 2. A self pair is one ordered pair: entries with idx[r] >= idx[c] only.
// The order of the additions inside a run is the order of the sorted list (deterministic); across runs the
// fixed-point atomics are order-independent: the tier stays bit-reproducible.
template <int W, bool FIXED>
__global__ void __launch_bounds__(64, (W * W + 63) / 64 <= 2 ? 5 : 1) form_pairs_kernel(FormArgs A, const unsigned long long* __restrict__ inc,
                                                        long long n_inc, const double* __restrict__ rec,
                                                        double* __restrict__ S) {
  constexpr int ST = rec_stride(W), W4 = rec_rows4(W), NE = (W * W + 63) / 64;
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * 64;
  const int cnt = (int)(n_inc - base < 64 ? n_inc - base : 64);
  const unsigned long long mine = lane < cnt ? inc[base + lane] : 0ull;
  const int mine_hi = (int)(unsigned)(mine >> 32), mine_lo = (int)(unsigned)mine;
  int er[NE], ec[NE];
  double acc[NE];
  float wgt[NE];     // 0: no such entry / upper triangle; 1; 2: the diagonal of a block both observations share
  unsigned tgt[NE];  // element index in S (n_c < 65 536: form() checks)
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = lane + 64 * j;
    er[j] = e / W;  // (>= W: no such entry)
    ec[j] = e % W;
    acc[j] = 0.0;
    wgt[j] = 0.f;
    tgt[j] = 0u;
  }
  const unsigned n = (unsigned)A.n_c;
  int s0 = -2, s1 = -2, s2 = -2, s3 = -2, s4 = -2, s5 = -2;
  bool cself = false;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      if (wgt[j] != 0.f) {
        const double val = acc[j] * (double)wgt[j];
        if (FIXED) {
          if (!(fabs(val) < kFixedTermBound) && A.bad) *A.bad = 1;  // (every writer stores the same value)
          atomicAdd(reinterpret_cast<unsigned long long*>(S) + tgt[j],
                    (unsigned long long)(long long)__double2ll_rn(val * kFixedScale));
        } else {
          unsafeAtomicAdd(S + tgt[j], val);
        }
      }
      acc[j] = 0.0;
    }
  };
  // Incidences are taken in batches of kIncBatch through LDS (round 6). A lane that fetches its own rows -- F row r of one
  // record, G row c of the other, for each of its NE entries, plus the headers: 8 NE + 6 loads per incidence, all
  // dependent on the incidence word -- keeps ONE incidence in flight per wave, and the kernel was the latency of 64
  // such round trips per wave (3.8 ms at BA-1 for 8.4 GB; a second register set was built in round 4 and dropped: as many
  // registers as it hides latency). Staged, the wave fetches the two blocks and the two headers of an incidence as
  // 4 W + (W4 + 4) / 2 contiguous 16-byte pieces, dealt over the lanes: 3 loads per lane cover FOUR incidences at W = 10,
  // the next batch's pieces are in flight (in registers) while this batch is consumed out of LDS, and the arithmetic
  // reads rows out of LDS. Same sums in the same order: the staging is invisible to the result.
  constexpr int HC = (W4 + 4) / 4;            // 16-byte pieces of a header
  constexpr int NCH = 4 * W + 2 * HC;         // pieces per incidence: F block, G block, header of each record
  constexpr int STI = 2 * NCH;                // doubles per staged incidence
  constexpr int NLD = (kIncBatch * NCH + 63) / 64;
  __shared__ __attribute__((aligned(16))) double stage[kIncBatch * STI];
  double2 pre[NLD];
  auto fetch = [&](int t0) {  // the pieces of incidences t0 .. t0 + kIncBatch - 1 (those below cnt) into `pre`
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int e = lane + 64 * k;
      const int q = e / NCH, d = e - q * NCH;
      const int t = t0 + q;
      pre[k] = double2{0.0, 0.0};
      // (the incidence words sit in lane t's registers: a shuffle, with every lane taking part)
      const unsigned hi = (unsigned)__shfl(mine_hi, t & 63), lo = (unsigned)__shfl(mine_lo, t & 63);
      if (q < kIncBatch && t < cnt) {
        const double* rh = rec + (size_t)hi * ST;
        const double* rl = rec + (size_t)lo * ST;
        const double* src = d < 2 * W ? rh + 2 * d
                          : d < 4 * W ? rl + 4 * W + 2 * (d - 2 * W)
                          : d < 4 * W + HC ? rh + 8 * W + 2 * (d - 4 * W) : rl + 8 * W + 2 * (d - 4 * W - HC);
        pre[k] = *reinterpret_cast<const double2*>(src);
      }
    }
  };
  auto publish = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int e = lane + 64 * k;
      if (e < kIncBatch * NCH) *reinterpret_cast<double2*>(stage + 2 * e) = pre[k];
    }
  };
  auto process = [&](int t, const double* I) {  // I = the staged incidence: F block, G block, header hi, header lo
    const bool self = __builtin_amdgcn_readlane(mine_hi, t) == __builtin_amdgcn_readlane(mine_lo, t);
    const int* hh = reinterpret_cast<const int*>(I + 8 * W);
    const int* hl = hh + (W4 + 4);
    const int h0 = hh[W4], h1 = hh[W4 + 1], h2 = hh[W4 + 2], l0 = hl[W4], l1 = hl[W4 + 1], l2 = hl[W4 + 2];
    if (h0 != s0 || h1 != s1 || h2 != s2 || l0 != s3 || l1 != s4 || l2 != s5 || self != cself) {
      flush();
      s0 = h0; s1 = h1; s2 = h2; s3 = l0; s4 = l1; s5 = l2;
      cself = self;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int ih = er[j] < W ? hh[er[j]] : -1;
        const int il = hl[ec[j]];
        const bool valid = ih >= 0 && il >= 0;
        if (self) {
          wgt[j] = (valid && ih >= il) ? 1.f : 0.f;
          tgt[j] = valid ? (unsigned)ih * n + (unsigned)il : 0u;
        } else {
          wgt[j] = valid ? (ih == il ? 2.f : 1.f) : 0.f;
          tgt[j] = valid ? (unsigned)(ih > il ? ih : il) * n + (unsigned)(ih > il ? il : ih) : 0u;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int r = er[j] < W ? er[j] : 0;
      const double2* fr = reinterpret_cast<const double2*>(I + 4 * r);                // F row r of the first record
      const double2* gr = reinterpret_cast<const double2*>(I + 4 * W + 4 * ec[j]);    // G row c of the second
      const double2 f0 = fr[0], f1 = fr[1], g0 = gr[0], g1 = gr[1];
      double v = fma(f1.x, g1.x, fma(f0.y, g0.y, f0.x * g0.x));
      if (self) {
        // a self pair adds J[:, r] . J[:, c]: J[0][c] rides in F row c, J[1][r] in G row r (one record: both staged)
        const double j1r = I[4 * W + 4 * r + 3], j0c = I[4 * ec[j] + 3];
        v = fma(f1.y, j0c, j1r * g1.y) - v;
      } else {
        v = -v;
      }
      acc[j] += v;
    }
  };
  fetch(0);
  for (int t0 = 0; t0 < cnt; t0 += kIncBatch) {
    publish();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t0 + kIncBatch < cnt) fetch(t0 + kIncBatch);
#pragma unroll
    for (int q = 0; q < kIncBatch; ++q)
      if (t0 + q < cnt) process(t0 + q, stage + q * STI);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane has read the batch before the next one overwrites it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  flush();
}

// in place: 64-bit fixed point -> double (lower triangle; the upper one is never read)
__global__ void fixed_to_double_kernel(double* __restrict__ S, size_t n, const int* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const long long q = reinterpret_cast<const long long*>(S)[i];
  S[i] = (double)q * (1.0 / kFixedScale);
  if (i == 0 && bad && *bad != 0) S[0] = NAN;  // a term the fixed point could not hold: the factorisation must fail
}

__global__ void diag_kernel(int n, const double* __restrict__ Dc, double* __restrict__ S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) S[(size_t)i * n + i] += Dc[i] * Dc[i];
}

// J: [3][12][count] tangent columns of the position priors (pose columns, then sensor columns)
template <bool FIXED>
__global__ void prior_rows_kernel(double* __restrict__ S, int n, const double* __restrict__ J,
                                  const int* __restrict__ po, const int* __restrict__ so,
                                  const int* __restrict__ pdim, int count, int* __restrict__ bad) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count * 144) return;
  const int kprior = e / 144, i = (e % 144) / 12, k = e % 12;
  const int pd = pdim[kprior];
  const int w = pd + (so[kprior] >= 0 ? 6 : 0);
  if (i >= w || k >= w) return;
  const int row = i < pd ? po[kprior] + i : so[kprior] + (i - pd);
  const int col = k < pd ? po[kprior] + k : so[kprior] + (k - pd);
  if (row < col) return;
  double v = 0.0;
  for (int r = 0; r < 3; ++r)
    v += J[((size_t)r * 12 + i) * count + kprior] * J[((size_t)r * 12 + k) * count + kprior];
  if (FIXED) {  // (several images of a rig frame put several priors on one pose block: same integer accumulation)
    if (!(fabs(v) < kFixedTermBound) && bad) *bad = 1;
    atomicAdd(reinterpret_cast<unsigned long long*>(S) + (size_t)row * n + col,
              (unsigned long long)(long long)__double2ll_rn(v * kFixedScale));
  } else {
    unsafeAtomicAdd(S + (size_t)row * n + col, v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Blocked Cholesky
// ---------------------------------------------------------------------------------------------------------

// v_readlane_b32 x 2 (HIP's __shfl is a ds_bpermute: an LDS round trip on the critical chain of the kernel below)
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// 1 / d to ~1 ulp without the IEEE division sequence (v_div_scale x 2, v_div_fmas, v_div_fixup around the same
// v_rcp_f64 + two Newton steps): the pivots are positive normal numbers, and the reciprocal sits on the serial chain
// of every column step below.
__device__ __forceinline__ double rcp_f64(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  return y;
}

// Diagonal block [k0, k0 + kb): L_kk in place (lower), its inverse (64 x 64, zero-padded) to Linv -- 125 of these
// kernels in a row sit on the critical path of the factorisation at BA-1. A BLOCKED factorisation over 16 x 16 tiles
// with the off-diagonal work on the matrix cores (round 6). It replaces the register kernel of rounds 4-5 (two waves,
// a 64 x 64 matrix in 128 registers each, a chain of 64 column steps with up to 63 dependent-issue fp64 FMAs per step in
// ~85 KB of straight-line code: 22-28 us alone, 35 us on average and up to 160 us beside the lookahead stream's bulk
// update -- instruction fetch through a busy fabric, vector fp64 FMAs sharing the matrix cores' pipeline, ROUND_NOTES
// round 5). Blocked, the serial part is four 16-step eliminations of a 16 x 16 tile (at most 15 FMAs per step),
// everything else is 16 x 16 x 16 products, in 10 KB of code: 20 us alone, 28.7 us on average in the BA-1 solve
// (profiles/r06_ba_chol_diag_blocked.txt), and the translation unit compiles in seconds instead of four minutes.
//   for kk = 0 .. 3:  (a) wave 0: tile (kk, kk) -> L16, L16^-1         lanes 0..15 = rows of A~ (elimination without
//                         scaling: the loop works on A~ with L = A~ D^-1/2, D = diag(pivots), one sqrt per lane at the
//                         end), lanes 16..31 = columns of Y = A~^-1, the same column broadcasts (16 doubles through LDS
//                         per step) and the same instructions for both
//                     (b) L[i][kk] = A[i][kk] L16^-T (i > kk);   Y[kk][j] = L16^-1 R[kk][j] (j < kk)
//                     (c) A[i][j] -= L[i][kk] L[j][kk]^T (kk < j <= i);   R[i][j] -= L[i][kk] Y[kk][j] (j <= kk < i)
// R starts as the identity (its diagonal tiles live in the registers of (a)) and ends as L^-1: the inverse rides along
// with the factorisation, block forward substitution on the identity, without a barrier of its own. The matrix and R
// live in LDS (55 KB); four waves share the tiles of (b) and (c); three workgroup barriers per kk. Rows / columns beyond
// kb carry a unit diagonal and are not stored.
// Measured and dropped (round 6, profiles/r06_ba_chol_chain_variants.txt): ONE launch per outer panel of 256 columns
// doing the four diagonal blocks, the panel solves and the inner updates in one four-wave workgroup with the X blocks
// in 122 KB of LDS (chol_outer_kernel) -- 150-160 us per panel on an idle GPU against 217 us for the ten launches it
// replaced, but 250-330 us beside the bulk update where the separate launches spread over several CUs: 63.3 against
// 65.0 LM-it/s at BA-1.
template <int C>
__device__ __forceinline__ void diag16_steps(double (&a)[16], double& dmine, double* col, int lane) {
  if constexpr (C < 16) {
    const double ac = a[C];
    const double d = readlane_f64(ac, C);  // the pivot: entry C of row C (factor lane C)
    dmine = lane == C ? ac : dmine;
    double* cb = col + 32 * (C & 1);
    cb[lane < 16 ? lane : 16 + (lane & 15)] = ac;  // column C of A~ for every lane (the other lanes' stores land in a
                                                    // dummy half of the buffer: a guarded store is a branch per step)
    const double m = ac * rcp_f64(d);      // factor lanes: the multiplier of row `lane`; inverse lanes: y_C
    a[C] = lane >= 16 ? m : ac;
    // (one wave: LDS serves a wave's instructions in order, the fences only pin the compiler's order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int cc = C + 1; cc < 16; ++cc) a[cc] = fma(-m, cb[cc], a[cc]);
    __builtin_amdgcn_sched_barrier(0);  // (keeps the broadcast values of a step from staying live across later steps)
    diag16_steps<C + 1>(a, dmine, col, lane);
  }
}

constexpr int TB = 16, NT = NB / TB, RS = TB + 1, RBLK = TB * RS, RP_DOUBLES = NT * (NT + 1) / 2 * RBLK;
__device__ __forceinline__ double* r_tile(double* Rp, int i, int j) { return Rp + (i * (i + 1) / 2 + j) * RBLK; }

// The kk loop above on a block that lies in LDS: Am = the block (lower triangle, unit diagonal beyond kb), Rp = zeros,
// both published by a barrier. Afterwards Am holds L (lower), Rp the tiles (i, j <= i) of L^-1 (rows of RS doubles; the
// upper parts of its diagonal tiles are exact zeros), published by the last barrier. Returns (wave 0, lanes < 16) whether
// a pivot below kb was not positive. NWAVES = waves of the workgroup.
template <int NWAVES>
__device__ __forceinline__ bool diag64_blocked(double (*Am)[NB + 1], double* Rp, double* col, int kb, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  bool bad = false;
#pragma unroll 1
  for (int kk = 0; kk < NT; ++kk) {
    const int d0 = TB * kk;
    if (wave == 0) {  // (a)
      double a[TB];
#pragma unroll
      for (int c = 0; c < TB; ++c) {  // (unconditional reads + selects: guarded reads are a branch and a wait each)
        const double v = Am[d0 + li][d0 + c];
        a[c] = lane < TB ? v : (c == li ? 1.0 : 0.0);
      }
      double dmine = 1.0;
      diag16_steps<0>(a, dmine, col, lane);
      const bool okp = dmine > 0.0;
      bad = bad || (lane < TB && d0 + lane < kb && !okp);
      const double rs = okp ? 1.0 / sqrt(dmine) : NAN;
      const double dsq = dmine * rs;  // sqrt(pivot)
      double* Rd = r_tile(Rp, kk, kk);
#pragma unroll
      for (int c = 0; c < TB; ++c) {
        const double rs_c = readlane_f64(rs, c), dsq_c = readlane_f64(dsq, c);
        // factor lanes: L[lane][c];   inverse lanes: (L16^-1)[c][li] = sqrt(d_c) Y[c][li]
        double* dst = lane < TB ? &Am[d0 + li][d0 + c] : &Rd[c * RS + li];
        const double val = lane < TB ? (c < lane ? a[c] * rs_c : (c == lane ? dsq : 0.0)) : a[c] * dsq_c;
        if (lane < 2 * TB) *dst = val;
      }
    }
    __syncthreads();
    // (b) three tiles: L[i][kk] for i > kk, Y[kk][j] for j < kk
    for (int item = wave; item < NT - 1; item += NWAVES) {
      const double* Li = r_tile(Rp, kk, kk);
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
      if (item < NT - 1 - kk) {
        const int i = kk + 1 + item;
#pragma unroll
        for (int ks = 0; ks < TB / 4; ++ks) {
          const int m = 4 * ks + lk;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Am[TB * i + li][d0 + m], Li[li * RS + m], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Am[TB * i + lk + 4 * reg][d0 + li] = acc[reg];
      } else {
        const int j = item - (NT - 1 - kk);
        double* Rj = r_tile(Rp, kk, j);
#pragma unroll
        for (int ks = 0; ks < TB / 4; ++ks) {
          const int m = 4 * ks + lk;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[li * RS + m], Rj[m * RS + li], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Rj[(lk + 4 * reg) * RS + li] = acc[reg];
      }
    }
    __syncthreads();
    {  // (c) tiles (i, j), i > kk: j <= kk updates R, j > kk the matrix
      int t = 0;
#pragma unroll 1
      for (int i = kk + 1; i < NT; ++i)
#pragma unroll 1
        for (int j = 0; j <= i; ++j, ++t) {
          if (t % NWAVES != wave) continue;
          const bool upd_r = j <= kk;
          double* Ct = upd_r ? r_tile(Rp, i, j) : &Am[TB * i][TB * j];
          const int cs = upd_r ? RS : NB + 1;
          const double* Bt = upd_r ? r_tile(Rp, kk, j) : &Am[TB * j][d0];
          v4f64 acc;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[reg] = Ct[(lk + 4 * reg) * cs + li];
#pragma unroll
          for (int ks = 0; ks < TB / 4; ++ks) {
            const int m = 4 * ks + lk;
            const double b = upd_r ? Bt[m * RS + li] : Bt[li * (NB + 1) + m];  // Y[kk][j][m][c]  |  L[j][kk][c][m]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Am[TB * i + li][d0 + m], b, acc, 0, 0, 0);
          }
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) Ct[(lk + 4 * reg) * cs + li] = acc[reg];
        }
    }
    __syncthreads();
  }
  return bad;
}

// Block [k0, k0 + kb) of S into Am (lane = column, coalesced; the NB / NWAVES loads of a thread in flight together; an
// unconditional load of a valid address -- a guarded one is a branch and a wait), Rp = 0; no barrier.
template <int NWAVES>
__device__ __forceinline__ void diag64_load(const double* __restrict__ S, int n, int k0, int kb, double (*Am)[NB + 1],
                                            double* Rp, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  double stage[NB / NWAVES];
#pragma unroll
  for (int i = 0; i < NB / NWAVES; ++i) {
    const int r = wave + NWAVES * i;
    const bool ok = r < kb && lane <= r;
    const double v = S[ok ? (size_t)(k0 + r) * n + k0 + lane : (size_t)k0 * n + k0];
    stage[i] = ok ? v : (r == lane ? 1.0 : 0.0);
  }
#pragma unroll
  for (int i = 0; i < NB / NWAVES; ++i) Am[wave + NWAVES * i][lane] = stage[i];
  for (int e = tid; e < RP_DOUBLES; e += 64 * NWAVES) Rp[e] = 0.0;
}

// L (lower) back to S, the zero-padded 64 x 64 inverse to Linv.
template <int NWAVES>
__device__ __forceinline__ void diag64_store(double* __restrict__ S, int n, int k0, int kb, double (*Am)[NB + 1], double* Rp,
                                             double* __restrict__ Linv, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
  for (int i = 0; i < NB / NWAVES; ++i) {
    const int r = wave + NWAVES * i;
    if (r < kb && lane <= r) S[(size_t)(k0 + r) * n + k0 + lane] = Am[r][lane];
    const int ti = r / TB, tj = lane / TB;
    Linv[r * NB + lane] = (r < kb && lane < kb && tj <= ti) ? r_tile(Rp, ti, tj)[(r % TB) * RS + lane % TB] : 0.0;
  }
}

// (two workgroups' worth of registers at most: the kernel has to fit beside the bulk update's 256-register waves)
__global__ void __launch_bounds__(256, 2) chol_diag_kernel(double* __restrict__ S, int n, int k0, int kb,
                                                             double* __restrict__ Linv, int* __restrict__ info) {
  __shared__ double Am[NB][NB + 1];   // the block; tiles (i, j <= i) are final L once kk passed j
  __shared__ double Rp[RP_DOUBLES];   // R / Y / L^-1: tile (i, j <= i) at (i (i + 1) / 2 + j) * RBLK, rows of 17
  __shared__ __attribute__((aligned(16))) double col[4 * TB];
  const int tid = threadIdx.x;
  diag64_load<4>(S, n, k0, kb, Am, Rp, tid);
  __syncthreads();
  if (diag64_blocked<4>(Am, Rp, col, kb, tid)) *info = 1;
  diag64_store<4>(S, n, k0, kb, Am, Rp, Linv, tid);
}

// Panel below the diagonal block: X = A_panel L_kk^-T, in place. One workgroup per 64 rows; wave w owns rows
// 16 w .. 16 w + 15 and all four 16-column tiles. MFMA operand layout (as in ba_block_gram_kernel): lane l
// supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; the result registers hold
// D[row = (l >> 4) + 4 reg][col = l & 15].
__global__ void __launch_bounds__(256) chol_panel_kernel(double* __restrict__ S, int n, int nrows, int k0, int kb,
                                                         const double* __restrict__ Linv) {
  __shared__ double sA[NB][NB + 1];
  __shared__ double sLi[NB][NB + 1];
  const int tid = threadIdx.x;
  const int r0 = k0 + kb + NB * blockIdx.x;
  const int nr = min(NB, nrows - r0);
  {  // all 32 loads of a thread in flight together (rolled, the loop waited for every pair before its LDS writes)
    const int c = tid & 63, rq = tid >> 6;
    double va[16], vl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = rq + 4 * i;
      va[i] = (r < nr && c < kb) ? S[(size_t)(r0 + r) * n + k0 + c] : 0.0;
      vl[i] = Linv[r * NB + c];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sA[rq + 4 * i][c] = va[i];
      sLi[rq + 4 * i][c] = vl[i];
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int ks = 0; ks < NB / 4; ++ks) {
    const int m = 4 * ks + lk;
    const double a = sA[16 * wave + li][m];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const double b = sLi[16 * ct + li][m];  // B[k][j] = (L_kk^-1)[j][k]
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ct], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = 16 * wave + lk + 4 * reg, c = 16 * ct + li;
      if (r < nr && c < kb) S[(size_t)(r0 + r) * n + k0 + c] = acc[ct][reg];
    }
}

// The strip below an outer panel [o0, o0 + ow), ow <= 256: X = A_strip L_PP^-T for the rows [row0, nrows), in place, once
// the panel's diagonal block L_PP is final (its 64-blocks L_jk in S, the inverses of its diagonal blocks in Linv). One
// workgroup per 64 rows, the 64 x 256 tile in the matrix-core accumulators (wave w: rows 16 w .. 16 w + 15, sixteen 16 x 16
// tiles = 64 doubles per lane) for the whole block-column recursion
//     X_k = A_k L_kk^-T ;   A_j -= X_k L_jk^T   (j > k),      k = 0 .. 3,
// so the strip is read once and written once. (Done 64 columns at a time over all rows -- panel kernel, then the update of
// the panel's remaining columns -- the same arithmetic passed ten times over the strip: 4.9 GB per factorisation at
// n = 8 000, and every pass was a launch on the critical chain.) X_k goes from accumulator layout to operand layout
// through a per-wave LDS tile; L_kk^-1 / L_jk are staged in LDS for all four waves.
__global__ void __launch_bounds__(256, 2) chol_strip_kernel(double* __restrict__ S, int n, int nrows, int o0, int ow, int row0,
                                                         const double* __restrict__ Linv) {
  __shared__ double xs[4][16][NB + 1];
  __shared__ double sB[NB][NB + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int r0 = row0 + NB * blockIdx.x + 16 * wave;  // first row of the wave
  v4f64 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = r0 + lk + 4 * reg, c = 16 * t + li;
      acc[t][reg] = (r < nrows && c < ow) ? S[(size_t)r * n + o0 + c] : 0.0;
    }
  auto stage = [&](int k) {  // block k of the tile, accumulator layout -> xs[wave][row][column]
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xs[wave][lk + 4 * reg][16 * tt + li] = acc[4 * k + tt][reg];
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (NB * k >= ow) break;  // (a short last outer panel)
    stage(k);
    __syncthreads();  // xs written; the previous readers of sB are done
    {
      const double* Li = Linv + (size_t)k * NB * NB;
      double v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = Li[tid + 256 * i];
#pragma unroll
      for (int i = 0; i < 16; ++i) sB[(tid + 256 * i) >> 6][(tid + 256 * i) & 63] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {  // X_k = A_k L_kk^-T
      v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < NB / 4; ++ks) {
        const int m = 4 * ks + lk;
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[wave][li][m], sB[16 * ct + li][m], x, 0, 0, 0);
      }
      acc[4 * k + ct] = x;
    }
    __syncthreads();  // every wave has read xs (its own) and sB
    stage(k);         // X_k in operand position for the updates
#pragma unroll
    for (int j = k + 1; j < 4; ++j) {
      if (NB * j >= ow) break;
      {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = tid + 256 * i, jc = e >> 6, m = e & 63;
          v[i] = (NB * j + jc < ow && NB * k + m < ow) ? S[(size_t)(o0 + NB * j + jc) * n + o0 + NB * k + m] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sB[(tid + 256 * i) >> 6][(tid + 256 * i) & 63] = v[i];
      }
      __syncthreads();
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {
          const int m = 4 * ks + lk;
          acc[4 * j + ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xs[wave][li][m], sB[16 * ct + li][m], acc[4 * j + ct], 0, 0, 0);
        }
      __syncthreads();  // before sB is overwritten
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = r0 + lk + 4 * reg, c = 16 * t + li;
      if (r < nrows && c < ow) S[(size_t)r * n + o0 + c] = acc[t][reg];
    }
}

// Trailing update C_IJ -= X_I X_J^T for the 64 x 64 tiles I >= J of the trailing matrix (rows / columns from
// t0 = k0 + kb). Wave w owns the 32 x 32 quadrant (w >> 1, w & 1): 2 x 2 MFMA tiles; K = kb in halves of 32.
// (general form: C[r][c] -= sum_{m in [k0, k0 + kb)} S[r][m] S[c][m] for rows r >= row0 and columns c in
//  [col0, cend), lower triangle; a column window restricts the update to part of the trailing matrix)
__global__ void __launch_bounds__(256) chol_update_kernel(double* __restrict__ S, int n, int nrows, int k0, int kb,
                                                          int row0, int col0, int cend) {
  const int I = blockIdx.y, J = blockIdx.x;
  __shared__ double sI[NB][33];
  __shared__ double sJ[NB][33];
  const int tid = threadIdx.x;
  const int ri = row0 + NB * I, rj = col0 + NB * J;
  if (rj > ri + NB - 1 || rj >= cend) return;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int qi = 32 * (wave >> 1), qj = 32 * (wave & 1);
  // The accumulators start from the tile itself and the A operand enters negated, D = C - X_I X_J^T: the 16 loads of
  // a lane are in flight together with the first operand loads and the epilogue only stores. (The read-modify-write
  // epilogue this replaces compiled to one branch + load + wait + store per element: 16 serial round trips per lane.)
  v4f64 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = ri + qi + 16 * a + lk + 4 * reg, c = rj + qj + 16 * b + li;
        acc[a][b][reg] = S[(r < nrows && c < n) ? (size_t)r * n + c : (size_t)ri * n + rj];  // (out of range: any valid address)
      }
  // thread -> (row r = tid / 32 + 8 i, column m = tid % 32) of a half: all 16 loads of a thread are issued together,
  // and those of the next half right after the barrier that publishes this one (under its 32 matrix-core instructions)
  const int pr = tid >> 5, pm = tid & 31;
  const double* gI = S + (size_t)(ri + pr) * n + k0 + pm;
  const double* gJ = S + (size_t)(rj + pr) * n + k0 + pm;
  double pI[8], pJ[8];
  auto fetch = [&](int kh) {
    const bool kok = kh + pm < kb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = pr + 8 * i;
      pI[i] = (kok && ri + r < nrows) ? gI[(size_t)8 * i * n + kh] : 0.0;
      pJ[i] = (kok && rj + r < n) ? gJ[(size_t)8 * i * n + kh] : 0.0;
    }
  };
  fetch(0);
  for (int kh = 0; kh < kb; kh += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sI[pr + 8 * i][pm] = -pI[i];
      sJ[pr + 8 * i][pm] = pJ[i];
    }
    __syncthreads();
    if (kh + 32 < kb) fetch(kh + 32);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int m = 4 * ks + lk;
      const double a0 = sI[qi + li][m], a1 = sI[qi + 16 + li][m];
      const double b0 = sJ[qj + li][m], b1 = sJ[qj + 16 + li][m];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = ri + qi + 16 * a + lk + 4 * reg, c = rj + qj + 16 * b + li;
        if (r < nrows && c < cend && c <= r) S[(size_t)r * n + c] = acc[a][b][reg];
      }
}

// The same update with 128 x 128 tiles per workgroup: wave w owns the 64 x 64 quadrant (w >> 1, w & 1) as 4 x 4
// MFMA tiles (64 accumulator doubles per lane), K = kb streamed through LDS in chunks of 16. 16 flop per byte
// loaded instead of 8: used while the trailing matrix has enough 128-tiles to fill the chip.
// The next chunk's 16 doubles per thread are fetched into registers right after the barrier that publishes the
// current chunk, so the global-load latency runs under the 64 matrix-core instructions of the chunk instead of in
// front of them (the version without the prefetch left the matrix cores idle for a load round trip per chunk:
// 14 TFLOP/s over the trailing updates at BA-1). A diagonal tile (I == J) reads its rows once. As in the 64 x 64
// kernel the accumulators start from the tile and the epilogue only stores (it was 64 serial load-wait-store round
// trips per lane).
__global__ void __launch_bounds__(256, 2) chol_update128_kernel(double* __restrict__ S, int n, int nrows, int k0, int kb,
                                                                int row0, int col0, int cend) {
  const int I = blockIdx.y, J = blockIdx.x;
  constexpr int T = 128, KC = 16, PF = T * KC / 256;
  __shared__ double sI[T][KC + 1];
  __shared__ double sJ[T][KC + 1];
  const int tid = threadIdx.x;
  const int ri = row0 + T * I, rj = col0 + T * J;
  if (rj > ri + T - 1 || rj >= cend) return;
  const bool same = ri == rj;  // diagonal tile: X_J = X_I
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int qi = 64 * (wave >> 1), qj = 64 * (wave & 1);
  v4f64 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = ri + qi + 16 * a + lk + 4 * reg, c = rj + qj + 16 * b + li;
        acc[a][b][reg] = S[(r < nrows && c < n) ? (size_t)r * n + c : (size_t)ri * n + rj];  // (out of range: any valid address)
      }
  // thread -> (row r = tid / 16 + 16 i, column m = tid % 16) of a chunk: 16 lanes read 128 contiguous bytes of a row
  const int pr = tid >> 4, pm = tid & 15;
  const double* gI = S + (size_t)(ri + pr) * n + k0 + pm;
  const double* gJ = S + (size_t)(rj + pr) * n + k0 + pm;
  double pI[PF], pJ[PF];
  auto fetch = [&](int kc) {
    const bool kok = kc + pm < kb;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int r = pr + 16 * i;
      pI[i] = (kok && ri + r < nrows) ? gI[(size_t)16 * i * n + kc] : 0.0;
      pJ[i] = (!same && kok && rj + r < n) ? gJ[(size_t)16 * i * n + kc] : 0.0;
    }
  };
  fetch(0);
  for (int kc = 0; kc < kb; kc += KC) {
    __syncthreads();  // the previous chunk's readers are done
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      sI[pr + 16 * i][pm] = -pI[i];  // D = C - X_I X_J^T
      sJ[pr + 16 * i][pm] = same ? pI[i] : pJ[i];
    }
    __syncthreads();
    if (kc + KC < kb) fetch(kc + KC);
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int m = 4 * ks + lk;
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = sI[qi + 16 * a + li][m];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = sJ[qj + 16 * b + li][m];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = ri + qi + 16 * a + lk + 4 * reg, c = rj + qj + 16 * b + li;
        if (r < nrows && c < cend && c <= r) S[(size_t)r * n + c] = acc[a][b][reg];
      }
}

// Backward substitution, one launch per OUTER panel of up to 256 columns [p0, p0 + pw): x_P <- L_PP^-T w_P (final), then
// the columns left of the panel: w_j -= L_Pj^T x_P. Every workgroup recomputes x_P -- the panel's up to four 64-blocks
// from the last to the first, each  x_k = L_kk^-T w_k  with the stored inverse, then  w_j -= L_kj^T x_k  for the blocks
// j < k of the panel: ten small matrix-vector products out of L2 -- and workgroup 0 publishes it; workgroup 1 + t
// updates the 64 columns of tile t with all 256 terms. 32 launches at n = 8 000 instead of 125 (one per 64-block, each a
// dependent launch of ~10 us). Thread = (column, quarter): a quarter of the terms each, the four partial sums added in
// quarter order (a fixed tree: reproducible). Working vector w and output x must be different arrays: workgroup 0
// publishes the panel while the other workgroups of the same launch still read its raw values.
__global__ void __launch_bounds__(256) solve_backward_panel_kernel(const double* __restrict__ S, int n, int p0, int pw,
                                                                   const double* __restrict__ Linv, double* __restrict__ w,
                                                                   double* __restrict__ x) {
  constexpr int OBK = 256;
  __shared__ double xs[OBK];
  __shared__ double wv[OBK];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x, col = tid & 63, q = tid >> 6;
  wv[tid] = tid < pw ? w[p0 + tid] : 0.0;
  xs[tid] = 0.0;
  __syncthreads();
  const int nb = (pw + NB - 1) / NB;
  for (int kb = nb - 1; kb >= 0; --kb) {
    const int k0 = p0 + NB * kb, kw = min(NB, pw - NB * kb);
    const double* Li = Linv + (size_t)kb * NB * NB;
    {
      double v = 0.0;  // (L^-1)^T: column `col` of L^-1 (zeros above the diagonal, zero-padded beyond kw)
#pragma unroll
      for (int j = 0; j < 16; ++j) v += Li[(16 * q + j) * NB + col] * wv[NB * kb + 16 * q + j];
      part[q][col] = v;
    }
    __syncthreads();
    if (q == 0) xs[NB * kb + col] = col < kw ? ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col])) : 0.0;
    __syncthreads();
    for (int jb = 0; jb < kb; ++jb) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = 16 * q + j;
        acc += (m < kw ? S[(size_t)(k0 + m) * n + p0 + NB * jb + col] : 0.0) * xs[NB * kb + m];
      }
      part[q][col] = acc;
      __syncthreads();
      if (q == 0) wv[NB * jb + col] -= (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
    if (tid < pw) x[p0 + tid] = xs[tid];
    return;
  }
  const int c = NB * (blockIdx.x - 1) + col;
  double acc = 0.0;
  if (c < p0) {
#pragma unroll
    for (int mb = 0; mb < OBK / NB; ++mb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int m = NB * mb + 16 * q + j;
        acc += (m < pw ? S[(size_t)(p0 + m) * n + c] : 0.0) * xs[m];
      }
    }
  }
  part[q][col] = acc;
  __syncthreads();
  if (q == 0 && c < p0) w[c] -= (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
}

__global__ void nan_fill_kernel(int n, const int* __restrict__ info, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && *info != 0) x[i] = NAN;
}

}  // namespace

void form(const FormArgs& a, double* S, hipStream_t st) {
  const size_t n = (size_t)a.n_c;
  BAX_HIP(hipMemsetAsync(S, 0, n * n * sizeof(double), st));
  if (a.bad) BAX_HIP(hipMemsetAsync(a.bad, 0, sizeof(int), st));
  if (a.n_points <= 0 || a.n_obs <= 0) return;
  const int width = pick_width(a);
  if (a.pairs && a.pairs->inc && a.pairs->n_inc > 0 && a.a_pt && a.n_c < 65536) {
    const PairLists& pl = *a.pairs;
    const unsigned gp = (unsigned)((pl.n_inc + 63) / 64);
#define BAX_PAIRS(W)                                                                                                       \
  do {                                                                                                                     \
    if (pl.rec_doubles < (size_t)a.n_obs * rec_stride(W)) throw std::runtime_error("pair lists: record buffer");            \
    hipLaunchKernelGGL((form_records_kernel<W>), dim3((a.n_obs + rec_threads<W>() - 1) / rec_threads<W>()), dim3(rec_threads<W>()), 0, st, a, pl.rec);                                   \
    if (a.fixed_point)                                                                                                     \
      hipLaunchKernelGGL((form_pairs_kernel<W, true>), dim3(gp), dim3(64), 0, st, a, pl.inc, pl.n_inc, pl.rec, S);         \
    else                                                                                                                   \
      hipLaunchKernelGGL((form_pairs_kernel<W, false>), dim3(gp), dim3(64), 0, st, a, pl.inc, pl.n_inc, pl.rec, S);        \
  } while (0)
    if (width == 10) BAX_PAIRS(10);
    else if (width == 14) BAX_PAIRS(14);
    else if (width == 20) BAX_PAIRS(20);
    else BAX_PAIRS(28);
#undef BAX_PAIRS
    return;
  }
#define BAX_FORM(W)                                                                                              \
  do {                                                                                                           \
    if (a.fixed_point) hipLaunchKernelGGL((form_kernel<W, true>), dim3(a.n_points), dim3(64), 0, st, a, S);      \
    else hipLaunchKernelGGL((form_kernel<W, false>), dim3(a.n_points), dim3(64), 0, st, a, S);                   \
  } while (0)
  if (width == 10) BAX_FORM(10);
  else if (width == 14) BAX_FORM(14);
  else if (width == 20) BAX_FORM(20);
  else BAX_FORM(28);
#undef BAX_FORM
}

void free_pair_lists(PairLists& pl) {
  if (pl.inc) (void)hipFree(pl.inc);
  if (pl.rec) (void)hipFree(pl.rec);
  pl = PairLists{};
}

bool build_pair_lists(const FormArgs& a, PairLists& pl, hipStream_t st) {
  free_pair_lists(pl);
  if (!a.a_pt || a.n_points <= 0 || a.n_obs <= 0 || a.n_poses <= 0) return false;
  if ((unsigned long long)a.n_poses * ((unsigned long long)a.n_poses + 1) / 2 > 0xffffffffull) return false;
  const size_t rec_doubles = (size_t)a.n_obs * rec_stride(pick_width(a));
  if (rec_doubles * sizeof(double) > ((size_t)32 << 30)) return false;
  unsigned long long *cnt = nullptr, *off = nullptr, *vals_in = nullptr, *vals_out = nullptr;
  unsigned *keys_in = nullptr, *keys_out = nullptr;
  void* tmp = nullptr;
  double* rec = nullptr;
  auto release = [&](bool keep) {
    if (cnt) (void)hipFree(cnt);
    if (off) (void)hipFree(off);
    if (vals_in) (void)hipFree(vals_in);
    if (keys_in) (void)hipFree(keys_in);
    if (keys_out) (void)hipFree(keys_out);
    if (tmp) (void)hipFree(tmp);
    if (!keep) {
      if (vals_out) (void)hipFree(vals_out);
      if (rec) (void)hipFree(rec);
      (void)hipGetLastError();  // an allocation that failed is not an error of the solve
    }
  };
  auto alloc = [&](void** p_, size_t bytes) { return hipMalloc(p_, bytes ? bytes : 1) == hipSuccess; };
  const int np1 = a.n_points + 1;
  size_t scan_bytes = 0;
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, cnt, off, np1) != hipSuccess) return false;
  if (!alloc((void**)&cnt, sizeof(*cnt) * np1) || !alloc((void**)&off, sizeof(*off) * np1) || !alloc(&tmp, scan_bytes)) {
    release(false);
    return false;
  }
  // (from here on a failing call releases everything and reports "not applicable": the caller falls back to the
  //  point-major kernel; a sticky device error will surface at the caller's next checked call)
  auto failed = [&](hipError_t e) {
    if (e == hipSuccess) return false;
    release(false);
    return true;
  };
  hipLaunchKernelGGL(inc_count_kernel, dim3((unsigned)((np1 + 255) / 256)), dim3(256), 0, st, a, cnt);
  if (failed(hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, cnt, off, np1, st))) return false;
  unsigned long long total = 0;
  if (failed(hipMemcpyAsync(&total, off + a.n_points, sizeof(total), hipMemcpyDeviceToHost, st))) return false;
  if (failed(hipStreamSynchronize(st))) return false;
  (void)hipFree(tmp);
  tmp = nullptr;
  if (total == 0 || total > (unsigned long long)INT_MAX) {
    release(false);
    return false;
  }
  size_t sort_bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_in, keys_out, vals_in, vals_out, (int)total) != hipSuccess ||
      !alloc((void**)&keys_in, sizeof(unsigned) * total) || !alloc((void**)&keys_out, sizeof(unsigned) * total) ||
      !alloc((void**)&vals_in, sizeof(*vals_in) * total) || !alloc((void**)&vals_out, sizeof(*vals_out) * total) ||
      !alloc(&tmp, sort_bytes) || !alloc((void**)&rec, sizeof(double) * rec_doubles)) {
    release(false);
    return false;
  }
  hipLaunchKernelGGL(inc_emit_kernel, dim3((unsigned)((a.n_points + 127) / 128)), dim3(128), 0, st, a, off, keys_in, vals_in);
  if (failed(hipcub::DeviceRadixSort::SortPairs(tmp, sort_bytes, keys_in, keys_out, vals_in, vals_out, (int)total, 0, 32, st)))
    return false;
  if (failed(hipStreamSynchronize(st))) return false;
  release(true);
  pl.inc = vals_out;
  pl.n_inc = (long long)total;
  pl.rec = rec;
  pl.rec_doubles = rec_doubles;
  return true;
}

void finish(double* S, int n_c, bool fixed_point, const int* bad, hipStream_t st) {
  const size_t n = (size_t)n_c;
  if (fixed_point)
    hipLaunchKernelGGL(fixed_to_double_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st, S, n, bad);
}

void add_lm_diagonal(double* S, int n, const double* Dc, hipStream_t st) {
  hipLaunchKernelGGL(diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, Dc, S);
}

void add_prior_rows(double* S, int n, const double* J, const int* po, const int* so, const int* pdim, int count,
                    bool fixed_point, int* bad, hipStream_t st) {
  if (count <= 0) return;
  const dim3 grid((unsigned)((count * 144 + 255) / 256));
  if (fixed_point) hipLaunchKernelGGL(prior_rows_kernel<true>, grid, dim3(256), 0, st, S, n, J, po, so, pdim, count, bad);
  else hipLaunchKernelGGL(prior_rows_kernel<false>, grid, dim3(256), 0, st, S, n, J, po, so, pdim, count, bad);
}

void factor_solve(double* S, int n, const double* rhs, double* x, const Workspace& ws, hipStream_t st,
                  hipEvent_t ev_a, hipEvent_t ev_b, double* mfma_ms) {
  BAX_HIP(hipMemsetAsync(ws.info, 0, sizeof(int), st));
  if (mfma_ms) *mfma_ms = 0.0;
  // The forward substitution rides along with the factorisation: the right-hand side is row n of the matrix (the
  // caller's buffer has n + 1 rows). Factoring [[S, b], [b^T, .]] leaves y^T = (L^-1 b)^T in that row -- the panel
  // kernel turns its block k into y_k, the trailing updates subtract L_jk y_k from the blocks to its right -- so the
  // 125 launches of a forward sweep at n = 8 000 (one per block step, each a dependent chain) are gone; the row's own
  // diagonal entry does not exist and is never touched. Every row bound below is nrows, every column bound n.
  const int nrows = n + 1;
  BAX_HIP(hipMemcpyAsync(S + (size_t)n * n, rhs, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  // The whole factorisation is bracketed by the two events: the diagonal-block kernels in between are
  // < 2 % of its time at n >= 2 000, the rest are the two matrix-core kernels.
  if (ev_a) BAX_HIP(hipEventRecord(ev_a, st));
  // Two-level right-looking factorisation. Every update reads and writes the part of the matrix it updates
  // once, whatever its depth K: with 64-deep updates of the whole trailing matrix the factorisation moves
  // 2 x (n^2 / 2) x 8 B x n / 64 bytes = 8 flop per byte -- HBM-bound at a sixth of the f64 MFMA peak (measured
  // 12 TFLOP/s at n = 8 000). So the trailing matrix is updated once per OUTER panel of 256 columns (32 flop
  // per byte); inside an outer panel the 64-wide steps update only the panel's remaining columns.
  constexpr int OB = 256;
  // update(k0, kb, t0, cend): C[r][c] -= sum_m S[r][m] S[c][m], m in [k0, k0 + kb), rows r >= t0, columns
  // c in [t0, cend). update_cols(.., c0, cend): the same for columns [c0, cend) only (rows r >= c0: the lower
  // triangle has no entries above the diagonal of the first column).
  auto launch = [&](int k0, int kb, int row0, int col0, int cend, hipStream_t s_, int rows_end) {
    const int rows = rows_end - row0, cols = std::min(cend, n) - col0;
    if (rows <= 0 || cols <= 0) return;
    // 128 x 128 tiles where they are many (the bulk of the trailing matrix); the 256 columns of the next outer panel
    // (U1, on the critical path: rows / 64 workgroups of 128-tiles would leave more than half of the 256 CUs without
    // work) and small remainders take 64 x 64 tiles: four times the workgroups, a quarter of the work each
    if (rows >= ws.min_rows128 && cols > 256) {
      hipLaunchKernelGGL(chol_update128_kernel, dim3((cols + 127) / 128, (rows + 127) / 128), dim3(256), 0, s_, S, n, rows_end,
                         k0, kb, row0, col0, std::min(cend, n));
    } else {
      hipLaunchKernelGGL(chol_update_kernel, dim3((cols + NB - 1) / NB, (rows + NB - 1) / NB), dim3(256), 0, s_, S, n, rows_end,
                         k0, kb, row0, col0, std::min(cend, n));
    }
  };
  auto update = [&](int k0, int kb, int t0, int cend, hipStream_t s_) { launch(k0, kb, t0, t0, cend, s_, nrows); };
  auto update_cols = [&](int k0, int kb, int /*t0*/, int c0, int cend, hipStream_t s_) { launch(k0, kb, c0, c0, cend, s_, nrows); };
  // Lookahead over two streams (ws.st2 set): the outer update of panel o is split into U1 = the columns of the
  // NEXT outer panel (main stream, the next panel's steps need them) and U2 = everything right of that (second
  // stream), so U2(o) runs while the 64-wide steps of panel o + 1 -- serial, latency-bound kernels -- are under
  // way. U1(o + 1) writes a region U2(o) also writes: the main stream waits for U2(o) before it.
  const bool lookahead = ws.st2 != nullptr && ws.ev_panel != nullptr && ws.ev_u2 != nullptr;
  bool u2_pending = false;
  for (int o0 = 0; o0 < n; o0 += OB) {
    const int oend = std::min(o0 + OB, n);
    for (int k0 = o0; k0 < oend; k0 += NB) {
      const int kb = std::min(NB, n - k0);
      double* Li = ws.Linv + (size_t)(k0 / NB) * NB * NB;
      hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, S, n, k0, kb, Li, ws.info);
      // the 64-wide steps stay inside the outer panel's own rows [k0 + kb, oend): at most three workgroups each
      const int below = oend - k0 - kb;
      if (below > 0) {
        hipLaunchKernelGGL(chol_panel_kernel, dim3((below + NB - 1) / NB), dim3(256), 0, st, S, n, oend, k0, kb, Li);
        launch(k0, kb, k0 + kb, k0 + kb, oend, st, oend);
      }
    }
    // everything below the outer panel (incl. the right-hand side's row): the whole block-column recursion in one pass
    hipLaunchKernelGGL(chol_strip_kernel, dim3((nrows - oend + NB - 1) / NB), dim3(256), 0, st, S, n, nrows, o0, oend - o0, oend,
                       ws.Linv + (size_t)(o0 / NB) * NB * NB);
    if (oend >= n) break;
    if (!lookahead) {
      update(o0, oend - o0, oend, n, st);  // everything right of the outer panel, K = 256
      continue;
    }
    const int next_end = std::min(oend + OB, n);
    BAX_HIP(hipEventRecord(ws.ev_panel, st));                       // panel o is final
    if (u2_pending) BAX_HIP(hipStreamWaitEvent(st, ws.ev_u2, 0));   // U2(o - 1) wrote where U1(o) writes
    update(o0, oend - o0, oend, next_end, st);                      // U1: columns of the next outer panel
    if (next_end < n) {
      BAX_HIP(hipStreamWaitEvent(ws.st2, ws.ev_panel, 0));
      update_cols(o0, oend - o0, oend, next_end, n, ws.st2);        // U2: columns right of the next outer panel
      BAX_HIP(hipEventRecord(ws.ev_u2, ws.st2));
      u2_pending = true;
    }
  }
  if (u2_pending) BAX_HIP(hipStreamWaitEvent(st, ws.ev_u2, 0));
  if (ev_b) BAX_HIP(hipEventRecord(ev_b, st));
  // L^T x = y with y = row n of the factor, ws.tmp as the working vector (a step reads its own panel of it raw -- in
  // every workgroup -- and updates the columns to its left) and x as the output, one launch per outer panel.
  BAX_HIP(hipMemcpyAsync(ws.tmp, S + (size_t)n * n, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  for (int p0 = (n - 1) / OB * OB; p0 >= 0; p0 -= OB) {
    const int pw = std::min(OB, n - p0);
    hipLaunchKernelGGL(solve_backward_panel_kernel, dim3(1 + p0 / NB), dim3(256), 0, st, S, n, p0, pw,
                       ws.Linv + (size_t)(p0 / NB) * NB * NB, ws.tmp, x);
  }
  hipLaunchKernelGGL(nan_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, ws.info, x);
  if (mfma_ms && ev_a && ev_b) {
    BAX_HIP(hipEventSynchronize(ev_b));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev_a, ev_b) == hipSuccess) *mfma_ms = ms;
  }
}

}  // namespace ba_explicit
