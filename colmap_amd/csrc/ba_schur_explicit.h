// ba_schur_explicit.h -- the exact linear-solver tiers of the bundle-adjustment backend (DENSE_SCHUR /
// SPARSE_SCHUR of ceres::LinearSolverType as COLMAP selects them, reference
// estimators/bundle_adjustment_ceres.cc:203-213): the reduced camera system formed explicitly on the
// device and solved by a blocked Cholesky factorisation on the f64 matrix cores. Internal interface
// between ba_kernels.hip (the LM loop) and ba_schur_explicit.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace ba_explicit {

constexpr int kPoseDim = 6;  // row stride of the pose-tangent columns (PD in ba_kernels.hip)

// Pair-major incidence lists of the formation (topology only: built once per solve by build_pair_lists, owned by
// the caller, released by free_pair_lists). An incidence is an unordered pair {a, b} of observations of one variable
// 3-D point, or the self pair (a, a) of any observation; the list is sorted by the (unordered) pair of pose blocks the
// two observations belong to, in a deterministic order (stable radix sort of a deterministic emission).
struct PairLists {
  unsigned long long* inc = nullptr;  // [n_inc] (p-order slot of the first observation << 32) | slot of the second
  long long n_inc = 0;
  double* rec = nullptr;              // per-observation records of the current linearisation (rewritten by form())
  size_t rec_doubles = 0;             // capacity of rec
};

// Everything the formation reads; all pointers are device pointers of the solver's current linearisation.
struct FormArgs {
  int n_obs, n_points, n_c;
  int n_poses;                  // number of pose blocks (variable or not): bounds the sort keys of the pair lists
  int kd;                       // row stride of the intrinsics-tangent columns
  const double* Jpose;          // c-order [2 * kPoseDim][n_obs]
  const double* Jcam;           // c-order [2 * kd][n_obs]
  const double* Jsens;          // c-order [2 * 6][n_obs] or NULL
  const double* Jpt;            // p-order [2 * 3][n_obs]
  const double* Cinv;           // [n_points][9] inverse point blocks (E^T E + Dp^2)^-1
  const int* a2c;               // p-order slot -> c-order slot
  const int* pt_ptr;            // [n_points + 1] p-order segments
  const int* pt_off;            // [n_points] tangent offset of the point, -1 constant
  const int *a_pose, *a_cam;    // p-order topology
  const int* a_pt;              // p-order: point of the observation (pair-major formation only; may be NULL without it)
  const PairLists* pairs;       // pair-major formation (see form()); NULL: the point-major kernel with one atomic per term
  const int* a_sensor;          // p-order sensor_from_rig index or NULL
  const int *pose_off, *pose_dim, *cam_off, *cam_dim;
  const int* sens_off;          // [n_sensors] tangent offset of a variable sensor_from_rig, or NULL
  bool fixed_point;             // accumulate in 64-bit fixed point (needs Jacobi-scaled columns): bit-reproducible
  // Fixed-point accumulation only: device flag (or NULL) raised when a term is not representable -- NaN / Inf, or
  // outside the bound Jacobi scaling guarantees (|term| < 1; the 2^-60 fixed point wraps at +-8). fp64 atomics would
  // have propagated the NaN; the integer conversion would turn it into a finite but wrong matrix. form() clears it,
  // add_prior_rows() raises it too, finish() then poisons S[0][0] with NaN: the sum over the ranks of a sharded solve
  // carries it to every rank, the first pivot fails, and factor_solve() fills x with NaN like a failed LLT.
  int* bad;
};

// S (n_c x n_c, row-major, LOWER triangle valid) = B - E C^-1 E^T of this rank's observations. S is cleared
// inside. The LM diagonal is added separately (after the sum over ranks of a sharded solve).
// Two formations of the same matrix:
//  * a.pairs == NULL: one wave per 3-D point, one atomic add per term (6.4e8 of them at 1 000 images x 200 000 points);
//  * a.pairs != NULL: pair-major. A record per observation (F = J^T E C^-1, G = J^T E: w x 3 each, and the columns of
//    J), then one wave per 64 consecutive incidences of the sorted list: lane (r, c) accumulates
//    -F_a[r] . G_b[c] (+ J_a[:, r] . J_a[:, c] for a self pair) in a register while the pair of blocks stays the
//    same and adds the sum to S once per run -- 15-20 x fewer atomics, no atomic inside a run, and 4 loads of
//    16 bytes per incidence and lane from two contiguous records instead of a point's staged observations.
void form(const FormArgs& a, double* S, hipStream_t st);
// Builds the lists for the topology in `a` (synchronises `st`). false: not applicable (no p-order point indices, more
// pose blocks or incidences than the 32-bit sort keys / counts hold, records beyond 32 GB, or an allocation failed) --
// nothing is left allocated, the caller passes pairs = NULL.
bool build_pair_lists(const FormArgs& a, PairLists& pl, hipStream_t st);
void free_pair_lists(PairLists& pl);
void add_lm_diagonal(double* S, int n, const double* Dc /* D, not D^2 */, hipStream_t st);

// Adds J^T J of the position priors to the lower triangle of S. J: [3][12][count] tangent columns (pose_dim
// pose columns, then 6 sensor_from_rig columns when so >= 0); po / so: tangent offsets (-1 constant).
void add_prior_rows(double* S, int n, const double* J, const int* po, const int* so, const int* pdim, int count,
                    bool fixed_point, int* bad /* FormArgs::bad or NULL */, hipStream_t st);
// After form (+ add_prior_rows): turns the fixed-point accumulators into doubles (no-op for fp64 accumulation).
void finish(double* S, int n_c, bool fixed_point, const int* bad /* FormArgs::bad or NULL */, hipStream_t st);

struct Workspace {
  double* Linv = nullptr;  // [ceil(n / 64)][64][64] inverses of the diagonal blocks of L
  double* tmp = nullptr;   // [n] second vector of the triangular solves
  int* info = nullptr;     // device flag: != 0 when a pivot was not positive
  // optional lookahead: a second stream and two events (all three or none)
  hipStream_t st2 = nullptr;
  hipEvent_t ev_panel = nullptr, ev_u2 = nullptr;
  // trailing updates with at least this many rows (and more than 256 columns) run 128 x 128 tiles, smaller ones 64 x 64
  // (a knob for the tests: the stand-in cannot afford the sizes at which the large tiles pay)
  int min_rows128 = 12 * 128;
  size_t linv_doubles(int n) const { return (size_t)((n + 63) / 64) * 64 * 64; }
};

// In-place blocked Cholesky S = L L^T (lower, row-major), then x = S^-1 rhs. THE BUFFER OF S HOLDS n + 1 ROWS of n
// doubles: the right-hand side is factored along as row n (forward substitution without a sweep of its own). A non-positive (or NaN) pivot fills x
// with NaN (the LM loop then rejects the step like a failed LLT). `mfma_ms` (optional, host) receives the
// time spent inside the matrix-core kernels (panel + trailing update), measured with the two events given.
void factor_solve(double* S, int n, const double* rhs, double* x, const Workspace& ws, hipStream_t st,
                  hipEvent_t ev_a, hipEvent_t ev_b, double* mfma_ms);

}  // namespace ba_explicit
