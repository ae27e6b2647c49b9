// ba_kernels.hip -- gfx950 bundle-adjustment solve: Jacobian builder (one reprojection
// residual per lane) + Levenberg-Marquardt with an implicit-Schur preconditioned CG.
//
// What it replaces: ceres::Solve as COLMAP configures it for large problems
// (ITERATIVE_SCHUR + SCHUR_JACOBI, reference estimators/bundle_adjustment_ceres.cc:203-213)
// together with the in-tree cost functions it evaluates
// (cost_functions/reprojection_error.h:61-212, quaternion_utils.h:105-153,
// sensor/models_jacobian.h:139-321). Semantics of the trust-region loop follow Ceres
// (restated in oracle/ba_oracle.c, which this file is tested against); the layout is new:
//
//  * SoA Jacobian in HBM, fp64: per observation 2x6 pose-tangent, 2x4 intrinsics-tangent,
//    2x3 point columns and the residual, each column contiguous over observations so that
//    the observation-parallel kernels (one lane per residual) read/write coalesced.
//  * Two observation orders, fixed once on the host: "c-order" (sorted by camera, then pose)
//    holds the camera-side Jacobian columns, so every pose / intrinsics block is a contiguous
//    range that the camera-side reductions stream; "p-order" (sorted by 3-D point) holds the
//    point columns, so the point-side passes (C_j = E_j^T E_j + D, C^-1 E^T x,
//    back-substitution) are a lane per point walking a contiguous segment. Only the 16-byte
//    per-observation vectors (J_c x, v) cross between the orders through the permutation.
//  * Camera-side reductions (gradient, J^T v, block-Jacobi Gram blocks) run a wave per chunk
//    of a parameter block's observation list with a wave-level tree reduction into a
//    per-chunk partial; a second tiny kernel adds a block's chunks in order. No atomics
//    anywhere: results are bit-reproducible run to run.
//  * The Schur-Jacobi blocks B_ii = sum J_i^T J_i (pose 6x6 / intrinsics up to 4x4 per
//    block, K = 2 x #observations) are dense Gram contractions and run on the f64 matrix
//    cores (v_mfma_f64_16x16x4_f64: A = B = a 4-row slab of J, D accumulates the 16x16
//    Gram tile).
//  * S x = (B + D^2) x - E C^-1 E^T x is never formed: three kernels per product.
#include "../../include/colmap_amd_ba.h"
#include "ba_schur_explicit.h"
#include "switches.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <string>
#include <vector>

using colmap_amd::dev_switch_int;

extern "C" void pm_release_cached_memory(void);  // pm_api.cpp (same library)

namespace {

thread_local std::string g_ba_error;
thread_local double g_spmv_ms = 0.0;
thread_local long long g_spmv_launches = 0;
thread_local long long g_spmv_bytes = 0;
thread_local double g_mfma_ms = 0.0;       // time inside the f64 MFMA Gram kernel (Schur-Jacobi blocks)
thread_local long long g_mfma_launches = 0;
thread_local long long g_pcg_pipelined_solves = 0, g_pcg_stepwise_solves = 0;  // linear solves per PCG loop, last ba_solve

#define BA_HIP(expr)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)

constexpr int PD = 6;        // pose tangent width (5 when the gauge holds a translation coordinate)
// The intrinsics tangent width KD (row stride of Jcam / cam_var) and the widest camera-side block BD
// (stride of the per-chunk partials) are chosen per problem: <KD, BD> = <4, 6> when no camera has
// more than 4 variable intrinsics (every model with the principal point fixed except OPENCV), else
// <8, 8>. The kernels whose register footprint depends on them are templates; the others read
// V.kd / V.bd.
constexpr int KD_MAX = 8;
constexpr int KD_WIDE = 16;  // FULL_OPENCV / THIN_PRISM_FISHEYE (12), RAD_TAN_THIN_PRISM_FISHEYE (16): third <KD, BD> tier
constexpr int NPAR_WIDE = 16;
constexpr int NPAR = 8;      // max number of parameters of a supported camera model (J_params is 2 x NPAR)
static int chunk_size() {     // observations per camera-side reduction chunk (one wave each)
  const int v = dev_switch_int("COLMAP_AMD_BA_CHUNK", 512);
  return v >= 64 ? v : 512;
}
constexpr int NSCALAR = 16;
constexpr int TILE_OBS = 512;  // observations staged in LDS per point-pass workgroup
constexpr int TILE_PTS = 256;

// Sum over ranks of a device vector (in place). world == 1: nothing. Two transports: a caller
// supplied host callback (any communicator: gloo, MPI, ...) or RCCL on the solver's stream.
struct Comm {
  int rank = 0, world = 1;
  bool by_point = false;  // observations sharded by 3-D point instead of by image (ba_comm::sharding)
  ba_allreduce_fn fn = nullptr;
  void* user = nullptr;
  ncclComm_t nccl = nullptr;
  std::vector<double> host;
  long long calls = 0, doubles = 0;

  void allreduce(double* dev, size_t n, hipStream_t st) {
    if (world <= 1 || n == 0) return;
    ++calls;
    doubles += (long long)n;
    if (nccl) {
      const ncclResult_t r = ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, nccl, st);
      if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL all-reduce failed: ") + ncclGetErrorString(r));
      return;
    }
    host.resize(n);
    BA_HIP(hipMemcpyAsync(host.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    if (fn(user, host.data(), (int64_t)n) != 0) throw std::runtime_error("all-reduce callback failed");
    BA_HIP(hipMemcpyAsync(dev, host.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    BA_HIP(hipStreamSynchronize(st));
  }
};

enum Scalar { S_COST = 0, S_GMAX, S_RHO, S_RHO_LAST, S_PQ, S_Q, S_MODEL, S_NEWCOST, S_ITER };

// ------------------------------------------------------------------------------------------
// Device-side view of the problem
// ------------------------------------------------------------------------------------------
struct View {
  int n_obs, n_poses, n_cams, n_points, n_c, n_p, n_blk, n_chunks;
  // parameters (current / candidate)
  double *poses, *cams, *points;
  // topology
  const int *o_pose, *o_cam, *o_pt;  // per observation, c-order (sorted by camera, then pose)
  const double* o_xy;                // c-order
  const int* o_sensor;               // c-order: sensor_from_rig of the observation, -1 none; or NULL
  double* sensors;                   // [n][7] (current / candidate)
  const int* sens_off;               // [n] tangent offset of a variable sensor_from_rig (6-wide), -1 constant; or NULL
  double* Jsens;                     // c-order [2 x 6][n_obs] sensor-tangent columns (only with variable sensors)
  int n_sensors;
  int loss_type;                     // BA_LOSS_*
  const int* stop;                   // pipelined PCG: kernels of an iteration enqueued past convergence return at once; else NULL
  double loss_scale;
  const int *c2a, *a2c;              // c-order position <-> p-order position (sorted by point)
  // p-order copies of the per-observation topology (the point-side linearisation pass reads them
  // coalesced); NULL: the c-order pass scatters the point columns through c2a instead
  const int *a_pose, *a_cam, *a_pt, *a_sensor;
  const double* a_xy;
  const unsigned char* solo;         // c-order: bit k set = no other observation of this point shares block kind k
  const int *pose_off, *pose_dim, *pose_fix;  // pose_fix: held translation coordinate or -1, + 4 when
                                              // the rotation is held (ba_problem::pose_fixed_t)
  const int *cam_off, *cam_dim, *cam_var, *cam_model;
  const int *pt_off, *pt_ptr;
  const int* tile_pt;  // point tiles: points [tile_pt[t], tile_pt[t+1]) have <= TILE_OBS observations
  const int4* tile_info;  // {first point, end point, first p-order observation, observations} of tile t: one load
  int n_tiles;         // 0: some track is longer than a tile, use the untiled kernel
  const int *blk_off, *blk_dim, *blk_kind, *blk_moff;
  const int *chunk_blk, *chunk_beg, *chunk_end;  // chunks are c-order ranges
  const int* blk_chunk_ptr;  // chunks of block b: [blk_chunk_ptr[b], blk_chunk_ptr[b+1])
  const int* blk_fin_end;    // the finalize kernels add the rows [blk_chunk_ptr[b], blk_fin_end[b]): all of the block's
                             // chunks, or only the first one for a HEAVY block (more than kHeavyChunks chunks: a camera
                             // shared by many images), whose rows ba_cpart_heavy_reduce_kernel has summed into it
  const int* heavy_blk;      // [n_heavy] the heavy blocks
  int n_heavy;
  // Observation pairs of one point inside one block (shared intrinsics, rig frames), per block kind k with such pairs:
  // a_boff[k][a] = tangent offset of p-order observation a's block of kind k (-1: none / constant), Wp[k][a] =
  // J_k,a^T E_a (wdim[k] x 3), rebuilt with every linearisation by ba_obs_w_kernel. Null for a kind without pairs.
  const int* a_boff[3];
  double* Wp[3];
  int wdim[3];
  double* cpart;             // [n_chunks][bd*bd] per-chunk partial results (no atomics)
  int kd, bd;                // intrinsics tangent width (Jcam / cam_var row stride), widest block
  // linearisation
  double *Jpose, *Jcam, *res;  // c-order
  double *Jpt, *res_p;         // p-order
  // fp32 copies of the same (scaled) columns for the PCG operator only (ba_options / COLMAP_AMD_BA_OPERATOR_F32):
  // the inexact inner solve streams half the bytes, accumulation stays fp64; cost, gradient, Schur-Jacobi
  // blocks, reduced right-hand side, back-substitution and step evaluation read the fp64 columns. NULL: off.
  float *Jpose32, *Jcam32, *Jpt32;
  double *scale_c, *scale_p;
  double* scalars;
};

// The two 16-byte-per-observation vectors that cross between the observation orders inside an implicit
// product, jx = J_c x and v = J_c x - E u, are both stored in C-ORDER as interleaved (row 0, row 1) pairs:
// the camera-side kernels (ba_obs_jx, ba_block_jtv) access them coalesced, and the point pass -- the one
// kernel that works in p-order -- GATHERS jx (one 16-byte load per observation through a2c) and SCATTERS v
// (one 16-byte store). Measured at BA-1 per product: two planes with 8-byte scattered stores on both
// crossings 106 + 81 + 56 us (obs_jx, point pass, jtv); gathers on both crossings 54 + 59 + 107 us (the
// gather inside the wave-per-chunk reduction of jtv is the expensive one); this layout: see DESIGN.md 2.4.
__device__ __forceinline__ double2 pair_load(const double* __restrict__ p, int i) {
  return reinterpret_cast<const double2*>(p)[i];
}
__device__ __forceinline__ void pair_store(double* __restrict__ p, int i, double a, double b) {
  reinterpret_cast<double2*>(p)[i] = make_double2(a, b);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Deterministic workgroup sum (fixed tree: lanes by xor-shuffle, then waves in index order).
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double wave_part[16];
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wave_part[wave] = v;
  __syncthreads();
  double total = 0.0;
  for (int w = 0; w < nwaves; ++w) total += wave_part[w];
  return total;
}

// out[0] = sum(partials[0..n)) in a fixed order (single workgroup)
__global__ void __launch_bounds__(1024) ba_final_sum_kernel(const double* __restrict__ partials, int n,
                                                            double* __restrict__ out) {
  double acc = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {  // eight loads in flight, added in the rolled loop's order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n; i += 1024) acc += partials[i];
  const double total = block_sum(acc);
  if (threadIdx.x == 0) *out = total;
}

__device__ __forceinline__ void atomic_max_pos(double* addr, double v) {
  // non-negative doubles order like their bit patterns
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// ------------------------------------------------------------------------------------------
// Per-residual math (reference reprojection_error.h:68-134)
// ------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ int num_params_of(int model) {
  switch (model) {
    case BA_SIMPLE_PINHOLE: case BA_SIMPLE_FISHEYE: return 3;
    case BA_RADIAL: case BA_RADIAL_FISHEYE: case BA_FOV: case BA_DIVISION: return 5;
    case BA_EUCM: return 6;
    case BA_OPENCV: case BA_OPENCV_FISHEYE: return 8;
    case BA_FULL_OPENCV: case BA_THIN_PRISM_FISHEYE: return 12;
    case BA_RAD_TAN_THIN_PRISM_FISHEYE: return 16;
    case BA_EQUIRECTANGULAR: return 2;
    default: return 4;  // PINHOLE, SIMPLE_RADIAL, SIMPLE_RADIAL_FISHEYE, SIMPLE_DIVISION, FISHEYE
  }
}
__host__ inline bool model_supported(int model) {
  return model == BA_SIMPLE_PINHOLE || model == BA_PINHOLE || model == BA_SIMPLE_RADIAL || model == BA_RADIAL ||
         model == BA_OPENCV || model == BA_OPENCV_FISHEYE || model == BA_SIMPLE_RADIAL_FISHEYE ||
         model == BA_RADIAL_FISHEYE || model == BA_FOV || model == BA_SIMPLE_DIVISION || model == BA_DIVISION ||
         model == BA_SIMPLE_FISHEYE || model == BA_FISHEYE || model == BA_EUCM || model == BA_FULL_OPENCV ||
         model == BA_THIN_PRISM_FISHEYE || model == BA_RAD_TAN_THIN_PRISM_FISHEYE || model == BA_EQUIRECTANGULAR;
}

// QuaternionRotatePointWithJac, quaternion_utils.h:105-153
__device__ __forceinline__ void quat_rotate(const double* q, const double* p, double out[3], double* J) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  const double px = p[0], py = p[1], pz = p[2];
  const double qx_py = qx * py, qx_pz = qx * pz, qy_px = qy * px, qy_pz = qy * pz, qz_px = qz * px,
               qz_py = qz * py;
  const double c0 = qy_pz - qz_py, c1 = qz_px - qx_pz, c2 = qx_py - qy_px;
  const double d0 = qy * c2 - qz * c1, d1 = qz * c0 - qx * c2, d2 = qx * c1 - qy * c0;
  out[0] = px + 2.0 * (qw * c0 + d0);
  out[1] = py + 2.0 * (qw * c1 + d1);
  out[2] = pz + 2.0 * (qw * c2 + d2);
  if (J) {
    const double qx_px = qx * px, qy_py = qy * py, qz_pz = qz * pz, qw_px = qw * px, qw_py = qw * py,
                 qw_pz = qw * pz;
    J[0] = 2.0 * (qy_py + qz_pz);
    J[1] = 2.0 * (-2.0 * qy_px + qx_py + qw_pz);
    J[2] = 2.0 * (-2.0 * qz_px - qw_py + qx_pz);
    J[3] = 2.0 * (-qz_py + qy_pz);
    J[4] = 2.0 * (qy_px - 2.0 * qx_py - qw_pz);
    J[5] = 2.0 * (qx_px + qz_pz);
    J[6] = 2.0 * (qw_px - 2.0 * qz_py + qy_pz);
    J[7] = 2.0 * (qz_px - qx_pz);
    J[8] = 2.0 * (qz_px + qw_py - 2.0 * qx_pz);
    J[9] = 2.0 * (-qw_px + qz_py - 2.0 * qy_pz);
    J[10] = 2.0 * (qx_px + qy_py);
    J[11] = 2.0 * (-qy_px + qx_py);
  }
}

// ImgFromCamWithJac, sensor/models_jacobian.h:139-321 (+ HasProjectableDepth, models.h:281-285)
template <bool JAC, int NP>
__device__ __forceinline__ bool img_from_cam(int model, const double* prm, double u, double v, double w,
                                             double& x, double& y, double* Jpar, double* Juvw) {
  if (model == BA_SIMPLE_DIVISION || model == BA_DIVISION) {
    // Fitzgibbon's one-parameter division model, closed form (internal::DivisionScaleWithJac,
    // models_jacobian.h:88-113; :1291-1411): no cheirality test, the discriminant decides
    const bool two_f = model == BA_DIVISION;
    const double f1 = prm[0], f2 = two_f ? prm[1] : prm[0];
    const int ic = two_f ? 2 : 1;
    const double k = prm[ic + 2];
    const double rho2 = u * u + v * v;
    const double disc_sq = w * w - 4.0 * rho2 * k;
    if (disc_sq < 0.0) return false;
    const double disc = sqrt(disc_sq);
    const double r = 2.0 / (w + disc);
    x = f1 * r * u + prm[ic];
    y = f2 * r * v + prm[ic + 1];
    if (JAC) {
      const double inv_disc = 1.0 / disc, r_sq = r * r;
      const double dr_du = 2.0 * r_sq * k * u * inv_disc, dr_dv = 2.0 * r_sq * k * v * inv_disc;
      const double dr_dw = -0.5 * r_sq * (1.0 + w * inv_disc), dr_dk = r_sq * rho2 * inv_disc;
      Juvw[0] = f1 * (r + u * dr_du); Juvw[1] = f1 * u * dr_dv; Juvw[2] = f1 * u * dr_dw;
      Juvw[3] = f2 * v * dr_du; Juvw[4] = f2 * (r + v * dr_dv); Juvw[5] = f2 * v * dr_dw;
#pragma unroll
      for (int c = 0; c < NP; ++c) Jpar[c] = Jpar[NP + c] = 0.0;
      Jpar[0] = r * u;
      Jpar[NP + (two_f ? 1 : 0)] = r * v;
      Jpar[ic] = 1.0;
      Jpar[NP + ic + 1] = 1.0;
      Jpar[ic + 2] = f1 * u * dr_dk;
      Jpar[NP + ic + 2] = f2 * v * dr_dk;
    }
    return true;
  }
  if (model == BA_EQUIRECTANGULAR) {
    // spherical panorama (models_jacobian.h:1502-1565): every non-zero direction projects, no cheirality
    // test; the two parameters (width, height) are metadata and always constant
    const double width = prm[0], height = prm[1];
    const double horizontal = sqrt(u * u + w * w);
    if (horizontal + fabs(v) < 2.220446049250313e-16) return false;
    const double theta = atan2(u, w), phi = atan2(-v, horizontal);
    const double kInv2Pi = 1.0 / (2.0 * 3.14159265358979323846), kInvPi = 1.0 / 3.14159265358979323846;
    x = (theta * kInv2Pi + 0.5) * width;
    y = (0.5 - phi * kInvPi) * height;
    if (JAC) {
      const double R2 = horizontal * horizontal, N2 = R2 + v * v;
      const double inv_R2 = 1.0 / R2, inv_N2 = 1.0 / N2, inv_N2_h = inv_N2 / horizontal;
      Juvw[0] = width * kInv2Pi * (w * inv_R2); Juvw[1] = 0.0; Juvw[2] = width * kInv2Pi * (-u * inv_R2);
      Juvw[3] = -height * kInvPi * (u * v * inv_N2_h); Juvw[4] = -height * kInvPi * (-horizontal * inv_N2);
      Juvw[5] = -height * kInvPi * (v * w * inv_N2_h);
#pragma unroll
      for (int c = 0; c < NP; ++c) Jpar[c] = Jpar[NP + c] = 0.0;
      Jpar[0] = theta * kInv2Pi + 0.5;
      Jpar[NP + 1] = 0.5 - phi * kInvPi;
    }
    return true;
  }
  if (!(w >= 2.220446049250313e-16)) return false;
  if (model == BA_EUCM) {  // models_jacobian.h:1413-1500
    const double f1 = prm[0], f2 = prm[1], alpha = prm[4], beta = prm[5];
    const double q = u * u + v * v;
    const double rho2 = beta * q + w * w;
    if (rho2 < 0.0) return false;
    const double rho = sqrt(rho2);
    const double den = alpha * rho + (1.0 - alpha) * w;
    if (!(den >= 2.220446049250313e-16)) return false;
    const double xn = u / den, yn = v / den;
    x = f1 * xn + prm[2];
    y = f2 * yn + prm[3];
    if (JAC) {
      const double inv_rho = 1.0 / rho, inv_den = 1.0 / den, inv_den2 = inv_den * inv_den;
      const double dden_du = alpha * beta * u * inv_rho, dden_dv = alpha * beta * v * inv_rho;
      const double dden_dw = alpha * w * inv_rho + (1.0 - alpha);
      const double dden_dalpha = rho - w, dden_dbeta = alpha * q * 0.5 * inv_rho;
      Juvw[0] = f1 * (inv_den - u * dden_du * inv_den2); Juvw[1] = f1 * (-u * dden_dv * inv_den2);
      Juvw[2] = f1 * (-u * dden_dw * inv_den2);
      Juvw[3] = f2 * (-v * dden_du * inv_den2); Juvw[4] = f2 * (inv_den - v * dden_dv * inv_den2);
      Juvw[5] = f2 * (-v * dden_dw * inv_den2);
      Jpar[0] = xn; Jpar[1] = 0.0; Jpar[2] = 1.0; Jpar[3] = 0.0;
      Jpar[4] = f1 * (-u * dden_dalpha * inv_den2); Jpar[5] = f1 * (-u * dden_dbeta * inv_den2);
      Jpar[NP + 0] = 0.0; Jpar[NP + 1] = yn; Jpar[NP + 2] = 0.0; Jpar[NP + 3] = 1.0;
      Jpar[NP + 4] = f2 * (-v * dden_dalpha * inv_den2); Jpar[NP + 5] = f2 * (-v * dden_dbeta * inv_den2);
    }
    return true;
  }
  const double inv_w = 1.0 / w;
  const double uu = u * inv_w, vv = v * inv_w;
  if (model == BA_FOV) {  // models_jacobian.h:627-724 (the three branches of FOVCameraModel::Distortion)
    const double f1 = prm[0], f2 = prm[1], omega = prm[4];
    const double a = uu, b = vv;
    const double radius2 = a * a + b * b, omega2 = omega * omega;
    const double kEpsilon = 1e-4;
    double factor, factor_r, factor_omega;
    if (omega2 < kEpsilon) {
      factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
      factor_r = omega2 / 3.0;
      factor_omega = 2.0 * omega * radius2 / 3.0 - omega / 6.0;
    } else if (radius2 < kEpsilon) {
      const double t = tan(omega / 2.0), t2 = t * t;
      const double Q = t * (4.0 * t2 * radius2 - 3.0);
      factor = -2.0 * Q / (3.0 * omega);
      factor_r = -8.0 * t * t2 / (3.0 * omega);
      const double dt_domega = 0.5 * (1.0 + t2);
      const double Q_omega = dt_domega * (12.0 * t2 * radius2 - 3.0);
      factor_omega = -2.0 / (3.0 * omega2) * (Q_omega * omega - Q);
    } else {
      const double radius = sqrt(radius2), t = tan(omega / 2.0);
      const double arg = 2.0 * radius * t, atan_arg = atan(arg);
      const double inv_denom_arg = 1.0 / (1.0 + arg * arg);
      factor = atan_arg / (radius * omega);
      factor_r = (2.0 * t * radius * inv_denom_arg - atan_arg) / (2.0 * radius2 * radius * omega);
      factor_omega = (radius * omega * (1.0 + t * t) * inv_denom_arg - atan_arg) / (radius * omega2);
    }
    const double du = a * factor, dv = b * factor;
    x = f1 * du + prm[2];
    y = f2 * dv + prm[3];
    if (JAC) {
      const double cross = 2.0 * a * b * factor_r;
      const double A0 = f1 * (factor + 2.0 * a * a * factor_r), A1 = f1 * cross, A2 = f2 * cross,
                   A3 = f2 * (factor + 2.0 * b * b * factor_r);
      Juvw[0] = A0 * inv_w; Juvw[1] = A1 * inv_w; Juvw[2] = -(A0 * a + A1 * b) * inv_w;
      Juvw[3] = A2 * inv_w; Juvw[4] = A3 * inv_w; Juvw[5] = -(A2 * a + A3 * b) * inv_w;
      Jpar[0] = du; Jpar[1] = 0.0; Jpar[2] = 1.0; Jpar[3] = 0.0; Jpar[4] = f1 * a * factor_omega;
      Jpar[NP + 0] = 0.0; Jpar[NP + 1] = dv; Jpar[NP + 2] = 0.0; Jpar[NP + 3] = 1.0;
      Jpar[NP + 4] = f2 * b * factor_omega;
    }
    return true;
  }
  if (model == BA_SIMPLE_PINHOLE) {
    const double f = prm[0];
    x = f * uu + prm[1];
    y = f * vv + prm[2];
    if (JAC) {
      const double fw = f * inv_w;
      Juvw[0] = fw; Juvw[1] = 0.0; Juvw[2] = -fw * uu; Juvw[3] = 0.0; Juvw[4] = fw; Juvw[5] = -fw * vv;
      Jpar[0] = uu; Jpar[1] = 1.0; Jpar[2] = 0.0;
      Jpar[NP + 0] = vv; Jpar[NP + 1] = 0.0; Jpar[NP + 2] = 1.0;
    }
    return true;
  }
  if (model == BA_PINHOLE) {
    const double f1 = prm[0], f2 = prm[1];
    x = f1 * uu + prm[2];
    y = f2 * vv + prm[3];
    if (JAC) {
      Juvw[0] = f1 * inv_w; Juvw[1] = 0.0; Juvw[2] = -f1 * inv_w * uu;
      Juvw[3] = 0.0; Juvw[4] = f2 * inv_w; Juvw[5] = -f2 * inv_w * vv;
      Jpar[0] = uu; Jpar[1] = 0.0; Jpar[2] = 1.0; Jpar[3] = 0.0;
      Jpar[NP + 0] = 0.0; Jpar[NP + 1] = vv; Jpar[NP + 2] = 0.0; Jpar[NP + 3] = 1.0;
    }
    return true;
  }
  if (model == BA_OPENCV_FISHEYE || model == BA_SIMPLE_RADIAL_FISHEYE || model == BA_RADIAL_FISHEYE ||
      model == BA_SIMPLE_FISHEYE || model == BA_FISHEYE) {
    // equidistant projection (internal::FisheyeProjectionWithJac, models_jacobian.h:51-80) followed by a
    // radial polynomial in the squared fisheye radius (:726-942); SIMPLE_FISHEYE / FISHEYE (:1190-1288)
    // are the same projection without distortion coefficients (nk = 0: identical values, the
    // polynomial terms vanish exactly)
    const bool two_f = model == BA_OPENCV_FISHEYE || model == BA_FISHEYE;
    const double f1 = prm[0], f2 = two_f ? prm[1] : prm[0];
    const int ic = two_f ? 2 : 1;                                  // index of cx
    const int nk = model == BA_OPENCV_FISHEYE ? 4 : (model == BA_RADIAL_FISHEYE ? 2 : (model == BA_SIMPLE_RADIAL_FISHEYE ? 1 : 0));
    const double* k = prm + ic + 2;
    const double a = uu, b = vv;                                   // normalised coordinates
    const double r2 = a * a + b * b;
    const double r = sqrt(r2);
    double fu, fv, Jf0 = 1.0, Jf1 = 0.0, Jf2 = 0.0, Jf3 = 1.0;
    if (r < 2.220446049250313e-16) {
      fu = a;
      fv = b;
    } else {
      const double theta = atan(r);
      const double sc = theta / r;
      fu = sc * a;
      fv = sc * b;
      if (JAC) {
        const double g = (r / (1.0 + r2) - theta) / (r2 * r);
        Jf0 = sc + a * a * g; Jf1 = a * b * g; Jf2 = a * b * g; Jf3 = sc + b * b * g;
      }
    }
    const double fu2 = fu * fu, fv2 = fv * fv, t2 = fu2 + fv2;
    double tp[4];
    tp[0] = t2; tp[1] = t2 * t2; tp[2] = tp[1] * t2; tp[3] = tp[1] * tp[1];
    double radial = 0.0;
    for (int i = 0; i < nk; ++i) radial += k[i] * tp[i];
    const double fu_d = fu + fu * radial, fv_d = fv + fv * radial;
    x = f1 * fu_d + prm[ic];
    y = f2 * fv_d + prm[ic + 1];
    if (JAC) {
      double d_radial = 0.0;
      for (int i = 0; i < nk; ++i) d_radial += (double)(i + 1) * k[i] * (i == 0 ? 1.0 : tp[i - 1]);
      const double cross = 2.0 * fu * fv * d_radial;
      const double i0 = 1.0 + radial + 2.0 * fu2 * d_radial, i3 = 1.0 + radial + 2.0 * fv2 * d_radial;
      const double m0 = i0 * Jf0 + cross * Jf2, m1 = i0 * Jf1 + cross * Jf3;
      const double m2 = cross * Jf0 + i3 * Jf2, m3 = cross * Jf1 + i3 * Jf3;
      const double A0 = f1 * m0, A1 = f1 * m1, A2 = f2 * m2, A3 = f2 * m3;
      Juvw[0] = A0 * inv_w; Juvw[1] = A1 * inv_w; Juvw[2] = -(A0 * a + A1 * b) * inv_w;
      Juvw[3] = A2 * inv_w; Juvw[4] = A3 * inv_w; Juvw[5] = -(A2 * a + A3 * b) * inv_w;
#pragma unroll
      for (int c = 0; c < NP; ++c) Jpar[c] = Jpar[NP + c] = 0.0;
      Jpar[0] = fu_d;
      Jpar[NP + (two_f ? 1 : 0)] = fv_d;
      Jpar[ic] = 1.0;
      Jpar[NP + ic + 1] = 1.0;
      for (int i = 0; i < nk; ++i) {
        Jpar[ic + 2 + i] = f1 * fu * tp[i];
        Jpar[NP + ic + 2 + i] = f2 * fv * tp[i];
      }
    }
    return true;
  }
  if (NP >= 12 && model == BA_FULL_OPENCV) {  // models_jacobian.h:498-625: rational radial term num / den + tangential
    const double f1 = prm[0], f2 = prm[1], k1 = prm[4], k2 = prm[5], p1 = prm[6], p2 = prm[7];
    const double k3 = prm[8], k4 = prm[9], k5 = prm[10], k6 = prm[11];
    const double uu2 = uu * uu, vv2 = vv * vv, uv = uu * vv, r2 = uu2 + vv2, r4 = r2 * r2, r6 = r4 * r2;
    const double num = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double den = 1.0 + k4 * r2 + k5 * r4 + k6 * r6;
    const double inv_den = 1.0 / den;
    const double radial = num * inv_den;
    const double xd = uu * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * uu2);
    const double yd = vv * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * vv2);
    x = f1 * xd + prm[2];
    y = f2 * yd + prm[3];
    if (JAC) {
      const double num_prime = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
      const double den_prime = k4 + 2.0 * k5 * r2 + 3.0 * k6 * r4;
      const double d_radial = (num_prime * den - num * den_prime) * inv_den * inv_den;
      const double cross = 2.0 * uv * d_radial;
      const double a00 = f1 * (radial + 2.0 * uu2 * d_radial + 2.0 * p1 * vv + 6.0 * p2 * uu);
      const double a01 = f1 * (cross + 2.0 * p1 * uu + 2.0 * p2 * vv);
      const double a10 = f2 * (cross + 2.0 * p2 * vv + 2.0 * p1 * uu);
      const double a11 = f2 * (radial + 2.0 * vv2 * d_radial + 2.0 * p2 * uu + 6.0 * p1 * vv);
      Juvw[0] = a00 * inv_w; Juvw[1] = a01 * inv_w; Juvw[2] = -(a00 * uu + a01 * vv) * inv_w;
      Juvw[3] = a10 * inv_w; Juvw[4] = a11 * inv_w; Juvw[5] = -(a10 * uu + a11 * vv) * inv_w;
      const double n1 = r2 * inv_den, n2 = r4 * inv_den, n3 = r6 * inv_den;
      const double nn = -num * inv_den * inv_den;
      const double d4 = nn * r2, d5 = nn * r4, d6 = nn * r6;
      Jpar[0] = xd; Jpar[1] = 0.0; Jpar[2] = 1.0; Jpar[3] = 0.0;
      Jpar[4] = f1 * uu * n1; Jpar[5] = f1 * uu * n2; Jpar[6] = f1 * 2.0 * uv; Jpar[7] = f1 * (r2 + 2.0 * uu2);
      Jpar[8] = f1 * uu * n3; Jpar[9] = f1 * uu * d4; Jpar[10] = f1 * uu * d5; Jpar[11] = f1 * uu * d6;
      Jpar[NP + 0] = 0.0; Jpar[NP + 1] = yd; Jpar[NP + 2] = 0.0; Jpar[NP + 3] = 1.0;
      Jpar[NP + 4] = f2 * vv * n1; Jpar[NP + 5] = f2 * vv * n2; Jpar[NP + 6] = f2 * (r2 + 2.0 * vv2);
      Jpar[NP + 7] = f2 * 2.0 * uv;
      Jpar[NP + 8] = f2 * vv * n3; Jpar[NP + 9] = f2 * vv * d4; Jpar[NP + 10] = f2 * vv * d5; Jpar[NP + 11] = f2 * vv * d6;
    }
    return true;
  }
  if (NP >= 16 && model == BA_RAD_TAN_THIN_PRISM_FISHEYE) {
    // models_jacobian.h:1049-1188: equidistant projection, radial polynomial th_radial(theta^2) with six
    // coefficients, then tangential (p0, p1) + thin-prism (s0..s3) distortion of the radially distorted point
    const double f1 = prm[0], f2 = prm[1];
    const double* k = prm + 4;
    const double p0 = prm[10], p1 = prm[11], s0 = prm[12], s1 = prm[13], s2 = prm[14], s3 = prm[15];
    const double a = uu, b = vv;
    const double rr2 = a * a + b * b;
    const double rr = sqrt(rr2);
    double fu, fv, Jf0 = 1.0, Jf1 = 0.0, Jf2 = 0.0, Jf3 = 1.0;
    if (rr < 2.220446049250313e-16) {
      fu = a;
      fv = b;
    } else {
      const double theta = atan(rr);
      const double sc = theta / rr;
      fu = sc * a;
      fv = sc * b;
      if (JAC) {
        const double g = (rr / (1.0 + rr2) - theta) / (rr2 * rr);
        Jf0 = sc + a * a * g; Jf1 = a * b * g; Jf2 = a * b * g; Jf3 = sc + b * b * g;
      }
    }
    const double theta2 = fu * fu + fv * fv;
    double th_radial = 1.0, d_th_radial = 0.0, theta_pow[6], power = 1.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double prev = power;
      power *= theta2;
      theta_pow[i] = power;
      th_radial += k[i] * power;
      d_th_radial += (double)(i + 1) * k[i] * prev;
    }
    const double xr = th_radial * fu, yr = th_radial * fv;
    const double xr2 = xr * xr, yr2 = yr * yr, xyr = xr * yr, r2 = xr2 + yr2, r4 = r2 * r2;
    const double X = xr + 2.0 * p1 * xyr + p0 * (r2 + 2.0 * xr2) + s0 * r2 + s1 * r4;
    const double Y = yr + 2.0 * p0 * xyr + p1 * (r2 + 2.0 * yr2) + s2 * r2 + s3 * r4;
    x = f1 * X + prm[2];
    y = f2 * Y + prm[3];
    if (JAC) {
      const double B00 = 1.0 + 2.0 * p1 * yr + 6.0 * p0 * xr + 2.0 * s0 * xr + 4.0 * s1 * xr * r2;
      const double B01 = 2.0 * p1 * xr + 2.0 * p0 * yr + 2.0 * s0 * yr + 4.0 * s1 * yr * r2;
      const double B10 = 2.0 * p0 * yr + 2.0 * p1 * xr + 2.0 * s2 * xr + 4.0 * s3 * xr * r2;
      const double B11 = 1.0 + 2.0 * p0 * xr + 6.0 * p1 * yr + 2.0 * s2 * yr + 4.0 * s3 * yr * r2;
      const double cross = 2.0 * fu * fv * d_th_radial;
      const double A0 = th_radial + 2.0 * fu * fu * d_th_radial, A3 = th_radial + 2.0 * fv * fv * d_th_radial;
      const double n0 = B00 * A0 + B01 * cross, n1 = B00 * cross + B01 * A3;
      const double n2 = B10 * A0 + B11 * cross, n3 = B10 * cross + B11 * A3;
      const double m0 = n0 * Jf0 + n1 * Jf2, m1 = n0 * Jf1 + n1 * Jf3;
      const double m2 = n2 * Jf0 + n3 * Jf2, m3 = n2 * Jf1 + n3 * Jf3;
      const double J0 = f1 * m0, J1 = f1 * m1, J2 = f2 * m2, J3 = f2 * m3;
      Juvw[0] = J0 * inv_w; Juvw[1] = J1 * inv_w; Juvw[2] = -(J0 * a + J1 * b) * inv_w;
      Juvw[3] = J2 * inv_w; Juvw[4] = J3 * inv_w; Juvw[5] = -(J2 * a + J3 * b) * inv_w;
#pragma unroll
      for (int c = 0; c < NP; ++c) Jpar[c] = Jpar[NP + c] = 0.0;
      Jpar[0] = X; Jpar[2] = 1.0;
      Jpar[NP + 1] = Y; Jpar[NP + 3] = 1.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double dxr = fu * theta_pow[i], dyr = fv * theta_pow[i];
        Jpar[4 + i] = f1 * (B00 * dxr + B01 * dyr);
        Jpar[NP + 4 + i] = f2 * (B10 * dxr + B11 * dyr);
      }
      Jpar[10] = f1 * (r2 + 2.0 * xr2); Jpar[11] = f1 * 2.0 * xyr;
      Jpar[NP + 10] = f2 * 2.0 * xyr; Jpar[NP + 11] = f2 * (r2 + 2.0 * yr2);
      Jpar[12] = f1 * r2; Jpar[13] = f1 * r4;
      Jpar[NP + 14] = f2 * r2; Jpar[NP + 15] = f2 * r4;
    }
    return true;
  }
  if (NP >= 12 && model == BA_THIN_PRISM_FISHEYE) {
    // models_jacobian.h:944-1047: equidistant projection, then radial (k1..k4) + tangential (p1, p2) +
    // thin-prism (sx1, sy1) distortion in fisheye coordinates
    const double f1 = prm[0], f2 = prm[1], k1 = prm[4], k2 = prm[5], p1 = prm[6], p2 = prm[7];
    const double k3 = prm[8], k4 = prm[9], sx1 = prm[10], sy1 = prm[11];
    const double a = uu, b = vv;
    const double rr2 = a * a + b * b;
    const double rr = sqrt(rr2);
    double fu, fv, Jf0 = 1.0, Jf1 = 0.0, Jf2 = 0.0, Jf3 = 1.0;
    if (rr < 2.220446049250313e-16) {
      fu = a;
      fv = b;
    } else {
      const double theta = atan(rr);
      const double sc = theta / rr;
      fu = sc * a;
      fv = sc * b;
      if (JAC) {
        const double g = (rr / (1.0 + rr2) - theta) / (rr2 * rr);
        Jf0 = sc + a * a * g; Jf1 = a * b * g; Jf2 = a * b * g; Jf3 = sc + b * b * g;
      }
    }
    const double fu2 = fu * fu, fv2 = fv * fv, fuv = fu * fv, r2 = fu2 + fv2, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
    const double radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
    const double du = fu * radial + 2.0 * p1 * fuv + p2 * (r2 + 2.0 * fu2) + sx1 * r2;
    const double dv = fv * radial + 2.0 * p2 * fuv + p1 * (r2 + 2.0 * fv2) + sy1 * r2;
    const double fu_d = fu + du, fv_d = fv + dv;
    x = f1 * fu_d + prm[2];
    y = f2 * fv_d + prm[3];
    if (JAC) {
      const double d_radial = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4 + 4.0 * k4 * r6;
      const double cross = 2.0 * fuv * d_radial;
      const double i0 = 1.0 + radial + 2.0 * fu2 * d_radial + 2.0 * p1 * fv + 6.0 * p2 * fu + 2.0 * sx1 * fu;
      const double i1 = cross + 2.0 * p1 * fu + 2.0 * p2 * fv + 2.0 * sx1 * fv;
      const double i2 = cross + 2.0 * p2 * fv + 2.0 * p1 * fu + 2.0 * sy1 * fu;
      const double i3 = 1.0 + radial + 2.0 * fv2 * d_radial + 2.0 * p2 * fu + 6.0 * p1 * fv + 2.0 * sy1 * fv;
      const double m0 = i0 * Jf0 + i1 * Jf2, m1 = i0 * Jf1 + i1 * Jf3;
      const double m2 = i2 * Jf0 + i3 * Jf2, m3 = i2 * Jf1 + i3 * Jf3;
      const double A0 = f1 * m0, A1 = f1 * m1, A2 = f2 * m2, A3 = f2 * m3;
      Juvw[0] = A0 * inv_w; Juvw[1] = A1 * inv_w; Juvw[2] = -(A0 * a + A1 * b) * inv_w;
      Juvw[3] = A2 * inv_w; Juvw[4] = A3 * inv_w; Juvw[5] = -(A2 * a + A3 * b) * inv_w;
      Jpar[0] = fu_d; Jpar[1] = 0.0; Jpar[2] = 1.0; Jpar[3] = 0.0;
      Jpar[4] = f1 * fu * r2; Jpar[5] = f1 * fu * r4; Jpar[6] = f1 * 2.0 * fuv; Jpar[7] = f1 * (r2 + 2.0 * fu2);
      Jpar[8] = f1 * fu * r6; Jpar[9] = f1 * fu * r8; Jpar[10] = f1 * r2; Jpar[11] = 0.0;
      Jpar[NP + 0] = 0.0; Jpar[NP + 1] = fv_d; Jpar[NP + 2] = 0.0; Jpar[NP + 3] = 1.0;
      Jpar[NP + 4] = f2 * fv * r2; Jpar[NP + 5] = f2 * fv * r4; Jpar[NP + 6] = f2 * (r2 + 2.0 * fv2);
      Jpar[NP + 7] = f2 * 2.0 * fuv;
      Jpar[NP + 8] = f2 * fv * r6; Jpar[NP + 9] = f2 * fv * r8; Jpar[NP + 10] = 0.0; Jpar[NP + 11] = f2 * r2;
    }
    return true;
  }
  if (model == BA_OPENCV) {  // models_jacobian.h:401-496
    const double f1 = prm[0], f2 = prm[1], k1 = prm[4], k2 = prm[5], p1 = prm[6], p2 = prm[7];
    const double uu2 = uu * uu, vv2 = vv * vv, uv = uu * vv, r2 = uu2 + vv2, r4 = r2 * r2;
    const double radial = k1 * r2 + k2 * r4;
    const double du = uu * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * uu2);
    const double dv = vv * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * vv2);
    const double xd = uu + du, yd = vv + dv;
    x = f1 * xd + prm[2];
    y = f2 * yd + prm[3];
    if (JAC) {
      const double d_radial_d_r2 = k1 + 2.0 * k2 * r2;
      const double cross = 2.0 * uv * d_radial_d_r2;
      const double du_duu = radial + 2.0 * uu2 * d_radial_d_r2 + 2.0 * p1 * vv + 6.0 * p2 * uu;
      const double du_dvv = cross + 2.0 * p1 * uu + 2.0 * p2 * vv;
      const double dv_duu = cross + 2.0 * p2 * vv + 2.0 * p1 * uu;
      const double dv_dvv = radial + 2.0 * vv2 * d_radial_d_r2 + 2.0 * p2 * uu + 6.0 * p1 * vv;
      const double a00 = f1 * (1.0 + du_duu);
      const double a01 = f1 * du_dvv;
      const double a10 = f2 * dv_duu;
      const double a11 = f2 * (1.0 + dv_dvv);
      Juvw[0] = a00 * inv_w; Juvw[1] = a01 * inv_w; Juvw[2] = -(a00 * uu + a01 * vv) * inv_w;
      Juvw[3] = a10 * inv_w; Juvw[4] = a11 * inv_w; Juvw[5] = -(a10 * uu + a11 * vv) * inv_w;
      Jpar[0] = xd; Jpar[1] = 0.0; Jpar[2] = 1.0; Jpar[3] = 0.0;
      Jpar[4] = f1 * uu * r2; Jpar[5] = f1 * uu * r4; Jpar[6] = f1 * 2.0 * uv; Jpar[7] = f1 * (r2 + 2.0 * uu2);
      Jpar[NP + 0] = 0.0; Jpar[NP + 1] = yd; Jpar[NP + 2] = 0.0; Jpar[NP + 3] = 1.0;
      Jpar[NP + 4] = f2 * vv * r2; Jpar[NP + 5] = f2 * vv * r4; Jpar[NP + 6] = f2 * (r2 + 2.0 * vv2);
      Jpar[NP + 7] = f2 * 2.0 * uv;
    }
    return true;
  }
  if (model == BA_RADIAL) {  // models_jacobian.h:323-398
    const double f = prm[0], k1 = prm[3], k2 = prm[4];
    const double uu2 = uu * uu, vv2 = vv * vv, r2 = uu2 + vv2, r4 = r2 * r2;
    const double radial = k1 * r2 + k2 * r4;
    const double xd = uu * (1.0 + radial), yd = vv * (1.0 + radial);
    x = f * xd + prm[1];
    y = f * yd + prm[2];
    if (JAC) {
      const double d_radial_d_r2 = k1 + 2.0 * k2 * r2;
      const double cross = 2.0 * uu * vv * d_radial_d_r2;
      const double a00 = f * (1.0 + radial + 2.0 * uu2 * d_radial_d_r2);
      const double a01 = f * cross;
      const double a10 = f * cross;
      const double a11 = f * (1.0 + radial + 2.0 * vv2 * d_radial_d_r2);
      Juvw[0] = a00 * inv_w; Juvw[1] = a01 * inv_w; Juvw[2] = -(a00 * uu + a01 * vv) * inv_w;
      Juvw[3] = a10 * inv_w; Juvw[4] = a11 * inv_w; Juvw[5] = -(a10 * uu + a11 * vv) * inv_w;
      Jpar[0] = xd; Jpar[1] = 1.0; Jpar[2] = 0.0; Jpar[3] = f * uu * r2; Jpar[4] = f * uu * r4;
      Jpar[NP + 0] = yd; Jpar[NP + 1] = 0.0; Jpar[NP + 2] = 1.0; Jpar[NP + 3] = f * vv * r2;
      Jpar[NP + 4] = f * vv * r4;
    }
    return true;
  }
  const double f = prm[0], k = prm[3];
  const double uu2 = uu * uu, vv2 = vv * vv, r2 = uu2 + vv2, k_r2 = k * r2, alpha = 1.0 + k_r2;
  const double xd = alpha * uu, yd = alpha * vv;
  x = f * xd + prm[1];
  y = f * yd + prm[2];
  if (JAC) {
    const double two_k = 2.0 * k, fw = f * inv_w, beta = 1.0 + 3.0 * k_r2, tkuv = two_k * uu * vv;
    Juvw[0] = fw * (alpha + two_k * uu2); Juvw[1] = fw * tkuv; Juvw[2] = -fw * uu * beta;
    Juvw[3] = fw * tkuv; Juvw[4] = fw * (alpha + two_k * vv2); Juvw[5] = -fw * vv * beta;
    Jpar[0] = xd; Jpar[1] = 1.0; Jpar[2] = 0.0; Jpar[3] = f * uu * r2;
    Jpar[NP + 0] = yd; Jpar[NP + 1] = 0.0; Jpar[NP + 2] = 1.0; Jpar[NP + 3] = f * vv * r2;
  }
  return true;
}

// ceres::LossFunction::Evaluate for the losses COLMAP can select (CreateLossFunction,
// bundle_adjustment_ceres.cc:66-80; Ceres loss_function.cc restated, see oracle/ba_oracle.c:bao_loss):
// rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s), s = |r|^2.
__device__ __forceinline__ void loss_eval(int type, double a, double s, double rho[3]) {
  const double kMin = 2.2250738585072014e-308;
  if (type == BA_LOSS_HUBER) {
    const double b = a * a;
    if (s > b) {
      const double r = sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = fmax(kMin, a / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  } else if (type == BA_LOSS_SOFT_L1) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double tmp = sqrt(sum);
    rho[0] = 2.0 * b * (tmp - 1.0);
    rho[1] = fmax(kMin, 1.0 / tmp);
    rho[2] = -(c * rho[1]) / (2.0 * sum);
  } else if (type == BA_LOSS_CAUCHY) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = b * log(sum);
    rho[1] = fmax(kMin, inv);
    rho[2] = -c * (inv * inv);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

__device__ __forceinline__ void quat_to_rot(const double* q, double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
               tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

template <typename JT> struct JSel;
template <> struct JSel<double> {
  static __device__ __forceinline__ const double* pose(const View& V) { return V.Jpose; }
  static __device__ __forceinline__ const double* cam(const View& V) { return V.Jcam; }
  static __device__ __forceinline__ const double* pt(const View& V) { return V.Jpt; }
};
template <> struct JSel<float> {
  static __device__ __forceinline__ const float* pose(const View& V) { return V.Jpose32; }
  static __device__ __forceinline__ const float* cam(const View& V) { return V.Jcam32; }
  static __device__ __forceinline__ const float* pt(const View& V) { return V.Jpt32; }
};

// Evaluate one observation; JAC: also the tangent-space, column-scaled Jacobian blocks.
// Jpar is laid out 2 x NPAR whatever the model (columns beyond the model's parameters are not read).
// PSIDE: also write the point-side columns and the p-order residual (scattered through c2a); false when
// ba_linearize_point_kernel produces them in its own p-order pass.
// ONLY >= 0: the PLAIN instantiation for the commonest problem shape -- every camera of model ONLY, no sensor_from_rig
// observations, trivial loss, no fp32 operator copies (the host checks; launch_linearize). The same expressions with the
// model switch, the rig branch and the corrector folded away at compile time: bit-identical columns, two thirds of the
// registers.
template <bool JAC, int KD, bool PSIDE = true, int ONLY = -1>
__global__ void __launch_bounds__(256) ba_linearize_kernel(View V, const double* __restrict__ poses,
                                                          const double* __restrict__ cams,
                                                          const double* __restrict__ points,
                                                          const double* __restrict__ sensors,
                                                          double* __restrict__ partials) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (o < V.n_obs) {
    const int pi = V.o_pose[o], ci = V.o_cam[o], xi = V.o_pt[o];
    const double* q = poses + 7 * (size_t)pi;
    const double* prm = cams + BA_CAM_STRIDE * (size_t)ci;
    const double* X = points + 3 * (size_t)xi;
    constexpr bool PLAIN = ONLY >= 0;
    const int model = PLAIN ? ONLY : V.cam_model[ci];
    constexpr int NP = KD > KD_MAX ? NPAR_WIDE : NPAR;  // J_params columns (12-parameter models only in the wide tier)
    double JR[12], Juvw[6], Jpar[2 * NP], pc[3];
    quat_rotate(q, X, pc, JAC ? JR : nullptr);
    pc[0] += q[4]; pc[1] += q[5]; pc[2] += q[6];
    // sensor_from_rig (RigReprojErrorCostFunctor / ...ConstantRigCostFunctor, reprojection_error.h:
    // 344-417): p_cam = R_s p_rig + t_s
    const int si = (!PLAIN && V.o_sensor) ? V.o_sensor[o] : -1;
    const int soff = (si >= 0 && V.sens_off) ? V.sens_off[si] : -1;
    double Rs[9], JRs[12], prig[3] = {pc[0], pc[1], pc[2]};
    if (si >= 0) {
      const double* sfr = sensors + 7 * (size_t)si;
      quat_to_rot(sfr, Rs);
      if (JAC && soff >= 0) {
        double tmp[3];
        quat_rotate(sfr, prig, tmp, JRs);  // d(R_s p_rig)/dq_s
      }
      const double p0 = pc[0], p1 = pc[1], p2 = pc[2];
      pc[0] = Rs[0] * p0 + Rs[1] * p1 + Rs[2] * p2 + sfr[4];
      pc[1] = Rs[3] * p0 + Rs[4] * p1 + Rs[5] * p2 + sfr[5];
      pc[2] = Rs[6] * p0 + Rs[7] * p1 + Rs[8] * p2 + sfr[6];
    }
    double rx = 0.0, ry = 0.0;
    const bool ok = img_from_cam<JAC, NP>(model, prm, pc[0], pc[1], pc[2], rx, ry, Jpar, Juvw);
    if (ok) {
      rx -= V.o_xy[2 * (size_t)o];
      ry -= V.o_xy[2 * (size_t)o + 1];
    } else {
      rx = ry = 0.0;  // behind the camera: zero residual and Jacobian (:96-116)
    }
    // robust loss: cost = 1/2 rho(|r|^2)
    const double sq_norm = rx * rx + ry * ry;
    double rho[3];
    if (PLAIN) { rho[0] = sq_norm; rho[1] = 1.0; rho[2] = 0.0; }
    else loss_eval(V.loss_type, V.loss_scale, sq_norm, rho);
    cost = 0.5 * rho[0];
    if (JAC) {
      double Js[2][6];  // sensor_from_rig tangent columns: J_uvw [dR_s p/dq_s PlusJacobian | I]
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Js[r][c] = 0.0;
      if (ok && soff >= 0) {
        const double* sfr = sensors + 7 * (size_t)si;
        const double x = sfr[0], y = sfr[1], z = sfr[2], w = sfr[3];
        const double PJs[12] = {w, z, -y, -z, w, x, y, -x, w, -x, -y, -z};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double Jq[4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            Jq[c] = Juvw[3 * r] * JRs[c] + Juvw[3 * r + 1] * JRs[4 + c] + Juvw[3 * r + 2] * JRs[8 + c];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            Js[r][c] = Jq[0] * PJs[c] + Jq[1] * PJs[3 + c] + Jq[2] * PJs[6 + c] + Jq[3] * PJs[9 + c];
            Js[r][3 + c] = Juvw[3 * r + c];
          }
        }
      }
      if (ok && si >= 0) {  // derivative w.r.t. the point in the rig frame: J_uvw R_s
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const double j0 = Juvw[3 * r], j1 = Juvw[3 * r + 1], j2 = Juvw[3 * r + 2];
          Juvw[3 * r] = j0 * Rs[0] + j1 * Rs[3] + j2 * Rs[6];
          Juvw[3 * r + 1] = j0 * Rs[1] + j1 * Rs[4] + j2 * Rs[7];
          Juvw[3 * r + 2] = j0 * Rs[2] + j1 * Rs[5] + j2 * Rs[8];
        }
      }
      const size_t N = (size_t)V.n_obs;
      const int a = V.c2a[o];  // p-order slot of this observation
      const int pdim = V.pose_dim[pi], poff = V.pose_off[pi];
      const int cdim = V.cam_dim[ci], coff = V.cam_off[ci];
      const int ptoff = V.pt_off[xi];
      // pose block: J_uvw * dRp/dq * PlusJacobian (EigenQuaternionManifold, xyzw) | J_uvw
      double Jp[2][PD], Jk[2][KD], Jx[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < PD; ++c) Jp[r][c] = 0.0;
#pragma unroll
        for (int c = 0; c < KD; ++c) Jk[r][c] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) Jx[r][c] = 0.0;
      }
      if (ok && pdim > 0) {
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        const double PJ[12] = {w, z, -y, -z, w, x, y, -x, w, -x, -y, -z};
        const int pf = V.pose_fix[pi];
        const bool rotc = pf >= 4;  // constant_rig_from_world_rotation: translation columns only
        const int fix = pf < 0 ? -1 : ((pf & 3) == 3 ? -1 : (pf & 3));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          // translation columns with the held coordinate removed; selects, not Jp[r][d++] -- a dynamically
          // indexed local array lives in scratch
          const double t0 = fix == 0 ? Juvw[3 * r + 1] : Juvw[3 * r];
          const double t1 = (fix == 0 || fix == 1) ? Juvw[3 * r + 2] : Juvw[3 * r + 1];
          const double t2 = fix < 0 ? Juvw[3 * r + 2] : 0.0;
          if (!rotc) {
            double Jq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
              Jq[c] = Juvw[3 * r] * JR[c] + Juvw[3 * r + 1] * JR[4 + c] + Juvw[3 * r + 2] * JR[8 + c];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              Jp[r][c] = Jq[0] * PJ[c] + Jq[1] * PJ[3 + c] + Jq[2] * PJ[6 + c] + Jq[3] * PJ[9 + c];
            Jp[r][3] = t0; Jp[r][4] = t1; Jp[r][5] = t2;
          } else {
            Jp[r][0] = t0; Jp[r][1] = t1; Jp[r][2] = t2;
          }
        }
      }
      // intrinsics block: variable subset of the model's parameters
      if (ok) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < KD; ++c)
            if (c < cdim) {
              // select chain instead of Jpar[dynamic index]: a dynamically indexed local array lives in scratch
              const int idx = V.cam_var[KD * ci + c];
              double sel = 0.0;
#pragma unroll
              for (int j = 0; j < NP; ++j) sel = (idx == j) ? Jpar[NP * r + j] : sel;
              Jk[r][c] = sel;
            }
      }
      // point block: J_uvw * R(q)
      if (PSIDE && ok && ptoff >= 0) {
        double R[9];
        quat_to_rot(q, R);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c)
            Jx[r][c] = Juvw[3 * r] * R[c] + Juvw[3 * r + 1] * R[3 + c] + Juvw[3 * r + 2] * R[6 + c];
      }
      // ceres::internal::Corrector: r' = residual_scaling r, J' = sqrt(rho') (J - alpha/|r|^2 r r^T J)
      if (!PLAIN && V.loss_type != BA_LOSS_TRIVIAL) {
        const double sqrt_rho1 = sqrt(rho[1]);
        double residual_scaling = sqrt_rho1, alpha_sq_norm = 0.0;
        if (!(sq_norm == 0.0 || rho[2] <= 0.0)) {
          const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(D);
          residual_scaling = sqrt_rho1 / (1.0 - alpha);
          alpha_sq_norm = alpha / sq_norm;
        }
        auto correct = [&](double& j0, double& j1) {
          const double rtj = rx * j0 + ry * j1;
          j0 = sqrt_rho1 * (j0 - alpha_sq_norm * rx * rtj);
          j1 = sqrt_rho1 * (j1 - alpha_sq_norm * ry * rtj);
        };
#pragma unroll
        for (int c = 0; c < PD; ++c) correct(Jp[0][c], Jp[1][c]);
#pragma unroll
        for (int c = 0; c < 6; ++c) correct(Js[0][c], Js[1][c]);
#pragma unroll
        for (int c = 0; c < KD; ++c) correct(Jk[0][c], Jk[1][c]);
        if (PSIDE) {
#pragma unroll
          for (int c = 0; c < 3; ++c) correct(Jx[0][c], Jx[1][c]);
        }
        rx *= residual_scaling;
        ry *= residual_scaling;
      }
      V.res[o] = rx;
      V.res[N + o] = ry;
      if (PSIDE) {
        V.res_p[a] = rx;
        V.res_p[N + a] = ry;
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < PD; ++c) {
          const double s = (c < pdim) ? V.scale_c[poff + c] : 0.0;
          V.Jpose[(size_t)(r * PD + c) * N + o] = Jp[r][c] * s;
          if (!PLAIN && V.Jpose32) V.Jpose32[(size_t)(r * PD + c) * N + o] = (float)(Jp[r][c] * s);
        }
#pragma unroll
        for (int c = 0; c < KD; ++c) {
          const double s = (c < cdim) ? V.scale_c[coff + c] : 0.0;
          V.Jcam[(size_t)(r * KD + c) * N + o] = Jk[r][c] * s;
          if (!PLAIN && V.Jcam32) V.Jcam32[(size_t)(r * KD + c) * N + o] = (float)(Jk[r][c] * s);
        }
        if (!PLAIN && V.sens_off) {
#pragma unroll
          for (int c = 0; c < 6; ++c)
            V.Jsens[(size_t)(r * 6 + c) * N + o] = soff >= 0 ? Js[r][c] * V.scale_c[soff + c] : 0.0;
        }
        if (PSIDE) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double s = (ptoff >= 0) ? V.scale_p[ptoff + c] : 0.0;
            V.Jpt[(size_t)(r * 3 + c) * N + a] = Jx[r][c] * s;
            if (!PLAIN && V.Jpt32) V.Jpt32[(size_t)(r * 3 + c) * N + a] = (float)(Jx[r][c] * s);
          }
        }
      }
    }
  }
  cost = block_sum(cost);
  if (threadIdx.x == 0) partials[blockIdx.x] = cost;
}

// Point side of the linearisation in its own p-order pass: one lane per p-order slot re-evaluates the
// observation (R X + t, the projection and J_uvw: ~200 fp64 flops, free next to the memory traffic) and
// writes the 2 x 3 point columns and the p-order residual COALESCED. The c-order pass used to scatter these
// eight doubles per lane through c2a: eight 64-byte write transactions for 64 bytes of payload
// (WRITE_SIZE 1.14 GB per launch at BA-1 for 0.45 GB of columns). Same expressions in the same order as
// ba_linearize_kernel, so the columns are bit-identical to the scattered ones.
template <int KD, int ONLY = -1>
__global__ void __launch_bounds__(256) ba_linearize_point_kernel(View V, const double* __restrict__ poses,
                                                                const double* __restrict__ cams,
                                                                const double* __restrict__ points,
                                                                const double* __restrict__ sensors) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= V.n_obs) return;
  const int pi = V.a_pose[a], ci = V.a_cam[a], xi = V.a_pt[a];
  const double* q = poses + 7 * (size_t)pi;
  const double* prm = cams + BA_CAM_STRIDE * (size_t)ci;
  const double* X = points + 3 * (size_t)xi;
  constexpr bool PLAIN = ONLY >= 0;
  const int model = PLAIN ? ONLY : V.cam_model[ci];
  constexpr int NP = KD > KD_MAX ? NPAR_WIDE : NPAR;
  double Juvw[6], Jpar[2 * NP], pc[3];
  quat_rotate(q, X, pc, nullptr);
  pc[0] += q[4]; pc[1] += q[5]; pc[2] += q[6];
  const int si = (!PLAIN && V.a_sensor) ? V.a_sensor[a] : -1;
  double Rs[9];
  if (si >= 0) {
    const double* sfr = sensors + 7 * (size_t)si;
    quat_to_rot(sfr, Rs);
    const double p0 = pc[0], p1 = pc[1], p2 = pc[2];
    pc[0] = Rs[0] * p0 + Rs[1] * p1 + Rs[2] * p2 + sfr[4];
    pc[1] = Rs[3] * p0 + Rs[4] * p1 + Rs[5] * p2 + sfr[5];
    pc[2] = Rs[6] * p0 + Rs[7] * p1 + Rs[8] * p2 + sfr[6];
  }
  double rx = 0.0, ry = 0.0;
  const bool ok = img_from_cam<true, NP>(model, prm, pc[0], pc[1], pc[2], rx, ry, Jpar, Juvw);
  if (ok) {
    rx -= V.a_xy[2 * (size_t)a];
    ry -= V.a_xy[2 * (size_t)a + 1];
  } else {
    rx = ry = 0.0;
  }
  const double sq_norm = rx * rx + ry * ry;
  if (ok && si >= 0) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double j0 = Juvw[3 * r], j1 = Juvw[3 * r + 1], j2 = Juvw[3 * r + 2];
      Juvw[3 * r] = j0 * Rs[0] + j1 * Rs[3] + j2 * Rs[6];
      Juvw[3 * r + 1] = j0 * Rs[1] + j1 * Rs[4] + j2 * Rs[7];
      Juvw[3 * r + 2] = j0 * Rs[2] + j1 * Rs[5] + j2 * Rs[8];
    }
  }
  const int ptoff = V.pt_off[xi];
  double Jx[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (ok && ptoff >= 0) {
    double R[9];
    quat_to_rot(q, R);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        Jx[r][c] = Juvw[3 * r] * R[c] + Juvw[3 * r + 1] * R[3 + c] + Juvw[3 * r + 2] * R[6 + c];
  }
  if (!PLAIN && V.loss_type != BA_LOSS_TRIVIAL) {
    double rho[3];
    loss_eval(V.loss_type, V.loss_scale, sq_norm, rho);
    const double sqrt_rho1 = sqrt(rho[1]);
    double residual_scaling = sqrt_rho1, alpha_sq_norm = 0.0;
    if (!(sq_norm == 0.0 || rho[2] <= 0.0)) {
      const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
      const double alpha = 1.0 - sqrt(D);
      residual_scaling = sqrt_rho1 / (1.0 - alpha);
      alpha_sq_norm = alpha / sq_norm;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double rtj = rx * Jx[0][c] + ry * Jx[1][c];
      const double j0 = Jx[0][c], j1 = Jx[1][c];
      Jx[0][c] = sqrt_rho1 * (j0 - alpha_sq_norm * rx * rtj);
      Jx[1][c] = sqrt_rho1 * (j1 - alpha_sq_norm * ry * rtj);
    }
    rx *= residual_scaling;
    ry *= residual_scaling;
  }
  const size_t N = (size_t)V.n_obs;
  V.res_p[a] = rx;
  V.res_p[N + a] = ry;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double s = (ptoff >= 0) ? V.scale_p[ptoff + c] : 0.0;
      V.Jpt[(size_t)(r * 3 + c) * N + a] = Jx[r][c] * s;
      if (!PLAIN && V.Jpt32) V.Jpt32[(size_t)(r * 3 + c) * N + a] = (float)(Jx[r][c] * s);
    }
}

// ------------------------------------------------------------------------------------------
// Point-side passes: one lane per 3-D point, contiguous observation segment
// ------------------------------------------------------------------------------------------

// g_p = E^T r and diag(E^T E)
__global__ void ba_point_grad_kernel(View V, double* __restrict__ gp, double* __restrict__ diag_p) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points) return;
  const int off = V.pt_off[j];
  if (off < 0) return;
  const size_t N = (size_t)V.n_obs;
  double g[3] = {0, 0, 0}, d[3] = {0, 0, 0};
  for (int o = V.pt_ptr[j]; o < V.pt_ptr[j + 1]; ++o)
    for (int r = 0; r < 2; ++r) {
      const double rr = V.res_p[r * N + o];
      for (int c = 0; c < 3; ++c) {
        const double J = V.Jpt[(size_t)(r * 3 + c) * N + o];
        g[c] += J * rr;
        d[c] += J * J;
      }
    }
  for (int c = 0; c < 3; ++c) {
    gp[off + c] = g[c];
    diag_p[off + c] = d[c];
  }
}

// E_j^T E_j of this rank's observations (upper triangle xx xy xz yy yz zz), [6][n_points]
__global__ void ba_point_gram_kernel(View V, double* __restrict__ Craw) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points) return;
  const size_t N = (size_t)V.n_obs;
  double C[6] = {0, 0, 0, 0, 0, 0};
  if (V.pt_off[j] >= 0)
    for (int o = V.pt_ptr[j]; o < V.pt_ptr[j + 1]; ++o)
      for (int r = 0; r < 2; ++r) {
        const double a = V.Jpt[(size_t)(r * 3 + 0) * N + o], b = V.Jpt[(size_t)(r * 3 + 1) * N + o],
                     c = V.Jpt[(size_t)(r * 3 + 2) * N + o];
        C[0] += a * a; C[1] += a * b; C[2] += a * c; C[3] += b * b; C[4] += b * c; C[5] += c * c;
      }
  for (int e = 0; e < 6; ++e) Craw[(size_t)e * V.n_points + j] = C[e];
}

// C_j = (sum over ranks of E_j^T E_j) + Dp^2, inverted (3x3 cofactor inverse)
__global__ void ba_point_blocks_kernel(View V, const double* __restrict__ Craw, const double* __restrict__ Dp,
                                       double* __restrict__ Cinv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points) return;
  const int off = V.pt_off[j];
  if (off < 0) return;
  double C[6];
  for (int e = 0; e < 6; ++e) C[e] = Craw[(size_t)e * V.n_points + j];
  C[0] += Dp[off] * Dp[off];
  C[3] += Dp[off + 1] * Dp[off + 1];
  C[5] += Dp[off + 2] * Dp[off + 2];
  const double a = C[0], b = C[1], c = C[2], d = C[3], e = C[4], f = C[5];
  const double A = d * f - e * e, B = c * e - b * f, Cc = b * e - c * d;
  const double det = a * A + b * B + c * Cc;
  const double id = 1.0 / det;
  double* out = Cinv + 9 * (size_t)j;
  out[0] = A * id; out[1] = B * id; out[2] = Cc * id;
  out[3] = B * id; out[4] = (a * f - c * c) * id; out[5] = (b * c - a * e) * id;
  out[6] = Cc * id; out[7] = (b * c - a * e) * id; out[8] = (a * d - b * b) * id;
}

// MODE 0 (Schur product): u = C^-1 E^T jx ; v_o = jx_o - E_o u
// MODE 1 (reduced rhs):   u = C^-1 g_p     ; v_o = -E_o u
// MODE 2 (back-subst.):   dp = C^-1 (g_p - E^T jx)
// MODE 3 (tiled kernel only): MODE 2 + the model cost change of the step, one partial sum per tile
// MODE 4 (tiled kernel only): MODE 1 + G_o = E_o C^-1 E_o^T of every observation for the Schur-Jacobi blocks
template <int MODE>
__global__ void ba_point_pass_kernel(View V, const double* __restrict__ Cinv, const double* __restrict__ jx,
                                     const double* __restrict__ gp, double* __restrict__ v,
                                     double* __restrict__ dp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points) return;
  const int off = V.pt_off[j];
  const size_t N = (size_t)V.n_obs;
  const int beg = V.pt_ptr[j], end = V.pt_ptr[j + 1];
  if (off < 0) {  // constant point: no point block
    if (MODE == 0) for (int o = beg; o < end; ++o) { const int c = V.a2c[o]; const double2 j2 = pair_load(jx, c); pair_store(v, c, j2.x, j2.y); }
    if (MODE == 1) for (int o = beg; o < end; ++o) pair_store(v, V.a2c[o], 0.0, 0.0);
    return;
  }
  double t[3] = {0, 0, 0};
  if (MODE != 1) {
    for (int o = beg; o < end; ++o) {
      const double2 j2 = pair_load(jx, V.a2c[o]);
      for (int r = 0; r < 2; ++r) {
        const double x = r ? j2.y : j2.x;
        for (int c = 0; c < 3; ++c) t[c] += V.Jpt[(size_t)(r * 3 + c) * N + o] * x;
      }
    }
  }
  if (MODE == 1) for (int c = 0; c < 3; ++c) t[c] = gp[off + c];
  if (MODE == 2) for (int c = 0; c < 3; ++c) t[c] = gp[off + c] - t[c];
  const double* Ci = Cinv + 9 * (size_t)j;
  double u[3];
  for (int r = 0; r < 3; ++r) u[r] = Ci[3 * r] * t[0] + Ci[3 * r + 1] * t[1] + Ci[3 * r + 2] * t[2];
  if (MODE == 2) {
    for (int c = 0; c < 3; ++c) dp[off + c] = u[c];
    return;
  }
  for (int o = beg; o < end; ++o) {
    const int c = V.a2c[o];
    double2 j2 = make_double2(0.0, 0.0);
    if (MODE == 0) j2 = pair_load(jx, c);
    double out[2];
    for (int r = 0; r < 2; ++r) {
      const double eu = V.Jpt[(size_t)(r * 3 + 0) * N + o] * u[0] + V.Jpt[(size_t)(r * 3 + 1) * N + o] * u[1] +
                        V.Jpt[(size_t)(r * 3 + 2) * N + o] * u[2];
      out[r] = (r ? j2.y : j2.x) - eu;
    }
    pair_store(v, c, out[0], out[1]);
  }
}

// Sharded variants (observations of a point live on several ranks): E^T jx partial -> all-reduce
// -> C^-1 and the per-observation update.
__global__ void ba_point_t_kernel(View V, const double* __restrict__ jx, double* __restrict__ t) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points) return;
  const int off = V.pt_off[j];
  if (off < 0) return;
  const size_t N = (size_t)V.n_obs;
  double acc[3] = {0, 0, 0};
  for (int o = V.pt_ptr[j]; o < V.pt_ptr[j + 1]; ++o) {
    const double2 j2 = pair_load(jx, V.a2c[o]);
    for (int r = 0; r < 2; ++r) {
      const double x = r ? j2.y : j2.x;
      for (int c = 0; c < 3; ++c) acc[c] += V.Jpt[(size_t)(r * 3 + c) * N + o] * x;
    }
  }
  for (int c = 0; c < 3; ++c) t[off + c] = acc[c];
}
// MODE 0: v_o = jx_o - E_o C^-1 t ; MODE 2: dp = C^-1 (g_p - t)
template <int MODE>
__global__ void ba_point_apply_kernel(View V, const double* __restrict__ Cinv, const double* __restrict__ jx,
                                      const double* __restrict__ gp, const double* __restrict__ t,
                                      double* __restrict__ v, double* __restrict__ dp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points) return;
  const int off = V.pt_off[j];
  const size_t N = (size_t)V.n_obs;
  const int beg = V.pt_ptr[j], end = V.pt_ptr[j + 1];
  if (off < 0) {
    if (MODE == 0) for (int o = beg; o < end; ++o) { const int c = V.a2c[o]; const double2 j2 = pair_load(jx, c); pair_store(v, c, j2.x, j2.y); }
    return;
  }
  double tt[3];
  for (int c = 0; c < 3; ++c) tt[c] = MODE == 2 ? gp[off + c] - t[off + c] : t[off + c];
  const double* Ci = Cinv + 9 * (size_t)j;
  double u[3];
  for (int r = 0; r < 3; ++r) u[r] = Ci[3 * r] * tt[0] + Ci[3 * r + 1] * tt[1] + Ci[3 * r + 2] * tt[2];
  if (MODE == 2) {
    for (int c = 0; c < 3; ++c) dp[off + c] = u[c];
    return;
  }
  for (int o = beg; o < end; ++o) {
    const int c = V.a2c[o];
    const double2 j2 = pair_load(jx, c);
    double out[2];
    for (int r = 0; r < 2; ++r) {
      const double eu = V.Jpt[(size_t)(r * 3 + 0) * N + o] * u[0] + V.Jpt[(size_t)(r * 3 + 1) * N + o] * u[1] +
                        V.Jpt[(size_t)(r * 3 + 2) * N + o] * u[2];
      out[r] = (r ? j2.y : j2.x) - eu;
    }
    pair_store(v, c, out[0], out[1]);
  }
}

// Same passes over tiles of consecutive points (<= TILE_PTS points, <= TILE_OBS observations per workgroup), a LANE
// PER OBSERVATION: the lane keeps its observation's six point columns, its (gathered) jx pair and its c-order
// position in registers from the first load to the final store; only the six products E_o[r][c] jx_o[r] of every
// observation and the 3-vector u = C^-1 t of every point cross the lanes, through 30 KB of LDS. A point's lane adds
// its observations' products in the order of the untiled kernel (bit-identical sums) -- 60 LDS reads per point where
// the columns used to be walked twice (160 reads and two dependent passes per point on 51 of 256 lanes at track length 10). Everything
// a lane needs later (C^-1, g_p, the tangent offset, the observation range) is requested before the first barrier.
template <int MODE, typename JT = double>
__global__ void __launch_bounds__(TILE_PTS) ba_point_pass_tiled_kernel(View V, const double* __restrict__ Cinv,
                                                                       const double* __restrict__ jx,
                                                                       const double* __restrict__ gp,
                                                                       double* __restrict__ v,
                                                                       double* __restrict__ dp,
                                                                       double* __restrict__ model_part = nullptr,
                                                                       double* __restrict__ Gout = nullptr) {
  constexpr bool kRhs = MODE == 1 || MODE == 4, kG = MODE == 4;
  constexpr bool kDp = MODE == 2 || MODE == 3, kModel = MODE == 3, kStoreV = MODE <= 1 || MODE == 4;
  constexpr int KPT = TILE_OBS / TILE_PTS;  // observations per lane
  __shared__ double st[6][kRhs ? 1 : TILE_OBS];  // the six products E_o[r][c] * jx_o[r] of every observation
  __shared__ double su[3][MODE == 2 ? 1 : TILE_PTS];  // u of the tile's points (0 for a constant point; MODE 3: y_p)
  __shared__ unsigned char sfix[MODE == 2 ? 1 : TILE_PTS];
  __shared__ double sci[kG ? 9 : 1][kG ? TILE_PTS : 1];  // MODE 4: C^-1 of the tile's points for the observations' lanes
  [[maybe_unused]] double res2[KPT][2];
  const int t = blockIdx.x, tid = threadIdx.x;
  // (the stop flag of the pipelined PCG is looked at AFTER the loads below are on their way: a workgroup's life is a
  //  chain of dependent round trips -- flag, tile, positions, gather -- and this takes the first one off it; a
  //  stopped launch only writes scratch nobody reads)
  const int stop = V.stop ? *V.stop : 0;
  const int4 ti = V.tile_info[t];
  const int p0 = ti.x, p1 = ti.y, a0 = ti.z, na = ti.w;
  const size_t N = (size_t)V.n_obs;
  const JT* __restrict__ Jp = JSel<JT>::pt(V);
  double J[KPT][6];
  double2 x2[KPT];
  int cpos[KPT], lpt[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int i = tid + k * TILE_PTS;
    x2[k] = make_double2(0.0, 0.0);
    cpos[k] = 0; lpt[k] = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) J[k][c] = 0.0;
    if (i < na) {
      cpos[k] = V.a2c[a0 + i];
      if (MODE != 2) lpt[k] = V.a_pt[a0 + i] - p0;
#pragma unroll
      for (int c = 0; c < 6; ++c) J[k][c] = (double)Jp[(size_t)c * N + a0 + i];
      if (!kRhs) x2[k] = pair_load(jx, cpos[k]);  // gather: jx lives in c-order
      if (kModel) { res2[k][0] = V.res_p[a0 + i]; res2[k][1] = V.res_p[N + a0 + i]; }
    }
  }
  // the point's lane: what it needs after the barrier is on its way now
  const int j = p0 + tid;
  int off = -1, beg = 0, end = 0;
  double Ci[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  if (j < p1) {
    off = V.pt_off[j];
    beg = V.pt_ptr[j] - a0;
    end = V.pt_ptr[j + 1] - a0;
    if (off >= 0) {
#pragma unroll
      for (int e = 0; e < 9; ++e) Ci[e] = Cinv[9 * (size_t)j + e];
      if (MODE != 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = gp[off + c];
      }
    }
  }
  if (stop) return;
  if (!kRhs) {
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = tid + k * TILE_PTS;
      if (i < na) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          st[c][i] = J[k][c] * x2[k].x;
          st[3 + c][i] = J[k][3 + c] * x2[k].y;
        }
      }
    }
    __syncthreads();
  }
  if (j < p1) {
    double u[3] = {0.0, 0.0, 0.0};
    if (off >= 0) {
      double tt[3] = {0.0, 0.0, 0.0};
      if (!kRhs) {
        // (the order of the untiled kernel and of the oracle: observation, residual row, component -- an inner
        //  solve of fifty CG iterations on an ill-conditioned system amplifies any other association to 1e-6)
        for (int o = beg; o < end; ++o) {
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) tt[c] += st[r * 3 + c][o];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) tt[c] = kRhs ? g[c] : kDp ? g[c] - tt[c] : tt[c];
#pragma unroll
      for (int r = 0; r < 3; ++r) u[r] = Ci[3 * r] * tt[0] + Ci[3 * r + 1] * tt[1] + Ci[3 * r + 2] * tt[2];
      if (kDp) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dp[off + c] = u[c];
      }
    }
    if (MODE != 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) su[c][tid] = u[c];
      sfix[tid] = off < 0 ? 1 : 0;
      if (kG) {
#pragma unroll
        for (int e = 0; e < 9; ++e) sci[e][tid] = Ci[e];
      }
    }
  }
  if (MODE == 2) return;
  __syncthreads();
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int i = tid + k * TILE_PTS;
    if (i < na) {
      const int lp = lpt[k];
      const double u0 = su[0][lp], u1 = su[1][lp], u2 = su[2][lp];
      if (kModel) {  // model cost change -(J step).(r + J step / 2), J step = -(jx + E y_p) (ba_model_from_jx_kernel)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double m = r ? x2[k].y : x2[k].x;
          if (!sfix[lp]) m += J[k][r * 3] * u0 + J[k][r * 3 + 1] * u1 + J[k][r * 3 + 2] * u2;
          m = -m;
          acc -= m * (res2[k][r] + 0.5 * m);
        }
        continue;
      }
      if (kG) {  // G_o = E_o C^-1 E_o^T (ba_obs_schur_g_kernel: the same expressions), one 32-byte record at the c-order position
        double g00 = 0.0, g01 = 0.0, g11 = 0.0;
        if (!sfix[lp]) {
          double ci[9];
#pragma unroll
          for (int e = 0; e < 9; ++e) ci[e] = sci[e][lp];
          const double e00 = J[k][0], e01 = J[k][1], e02 = J[k][2], e10 = J[k][3], e11 = J[k][4], e12 = J[k][5];
          const double t00 = e00 * ci[0] + e01 * ci[3] + e02 * ci[6], t01 = e00 * ci[1] + e01 * ci[4] + e02 * ci[7],
                       t02 = e00 * ci[2] + e01 * ci[5] + e02 * ci[8];
          const double t10 = e10 * ci[0] + e11 * ci[3] + e12 * ci[6], t11 = e10 * ci[1] + e11 * ci[4] + e12 * ci[7],
                       t12 = e10 * ci[2] + e11 * ci[5] + e12 * ci[8];
          g00 = t00 * e00 + t01 * e01 + t02 * e02;
          g01 = t00 * e10 + t01 * e11 + t02 * e12;
          g11 = t10 * e10 + t11 * e11 + t12 * e12;
        }
        double2* rec = reinterpret_cast<double2*>(Gout + 4 * (size_t)cpos[k]);
        rec[0] = make_double2(g00, g01);
        rec[1] = make_double2(g11, 0.0);
      }
      double out[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double eu = J[k][r * 3 + 0] * u0 + J[k][r * 3 + 1] * u1 + J[k][r * 3 + 2] * u2;
        const double xr = MODE == 0 ? (r ? x2[k].y : x2[k].x) : 0.0;
        out[r] = sfix[lp] ? xr : xr - eu;  // constant point: no point block, v = jx (MODE 0) / 0 (MODE 1)
      }
      if (kStoreV) pair_store(v, cpos[k], out[0], out[1]);  // one 16-byte scattered store
    }
  }
  if (kModel) {
    acc = block_sum(acc);
    if (tid == 0) model_part[t] = acc;
  }
}

// Point-major reductions over the same tiles, columns staged through LDS (coalesced loads over the tile's
// observation range, the per-point segment walk reads LDS).
// MODE 0: g_p = E^T r, the column norms of E and E^T E (upper triangle, [6][n_points]) of the points
//         (replaces ba_point_grad_kernel + ba_point_gram_kernel when tiles exist)
// MODE 1: model cost change -(J step).(r + J step / 2) with J step = -(jx + E y_p), partial sum per tile
template <int MODE>
__global__ void __launch_bounds__(TILE_PTS) ba_point_reduce_tiled_kernel(View V, const double* __restrict__ jx,
                                                                         const double* __restrict__ dpv,
                                                                         double* __restrict__ out0,
                                                                         double* __restrict__ out1,
                                                                         double* __restrict__ Craw) {
  __shared__ double sJ[6][TILE_OBS];
  __shared__ double sr[2][TILE_OBS];
  __shared__ double sx[MODE == 1 ? 2 : 1][MODE == 1 ? TILE_OBS : 1];
  const int t = blockIdx.x;
  const int p0 = V.tile_pt[t], p1 = V.tile_pt[t + 1];
  const int a0 = V.pt_ptr[p0], na = V.pt_ptr[p1] - a0;
  const size_t N = (size_t)V.n_obs;
  for (int i = threadIdx.x; i < na; i += TILE_PTS) {
#pragma unroll
    for (int c = 0; c < 6; ++c) sJ[c][i] = V.Jpt[(size_t)c * N + a0 + i];
    sr[0][i] = V.res_p[a0 + i];
    sr[1][i] = V.res_p[N + a0 + i];
    if (MODE == 1) {
      const double2 j2 = pair_load(jx, V.a2c[a0 + i]);
      sx[0][i] = j2.x;
      sx[1][i] = j2.y;
    }
  }
  __syncthreads();
  const int j = p0 + threadIdx.x;
  double acc = 0.0;
  if (j < p1) {
    const int off = V.pt_off[j];
    const int beg = V.pt_ptr[j] - a0, end = V.pt_ptr[j + 1] - a0;
    if (MODE == 0) {
      double C[6] = {0, 0, 0, 0, 0, 0};
      if (off >= 0) {
        double g[3] = {0, 0, 0};
        for (int o = beg; o < end; ++o)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const double rr = sr[r][o];
            const double a = sJ[r * 3][o], b = sJ[r * 3 + 1][o], c = sJ[r * 3 + 2][o];
            g[0] += a * rr; g[1] += b * rr; g[2] += c * rr;
            C[0] += a * a; C[1] += a * b; C[2] += a * c; C[3] += b * b; C[4] += b * c; C[5] += c * c;
          }
        for (int c = 0; c < 3; ++c) out0[off + c] = g[c];
        out1[off] = C[0]; out1[off + 1] = C[3]; out1[off + 2] = C[5];  // column norms = diagonal of E^T E
      }
      for (int e = 0; e < 6; ++e) Craw[(size_t)e * V.n_points + j] = C[e];
    } else {
      double y[3] = {0.0, 0.0, 0.0};
      if (off >= 0) { y[0] = dpv[off]; y[1] = dpv[off + 1]; y[2] = dpv[off + 2]; }
      for (int o = beg; o < end; ++o)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double m = sx[r][o];
          if (off >= 0) m += sJ[r * 3][o] * y[0] + sJ[r * 3 + 1][o] * y[1] + sJ[r * 3 + 2][o] * y[2];
          m = -m;
          acc -= m * (sr[r][o] + 0.5 * m);
        }
    }
  }
  if (MODE == 1) {
    acc = block_sum(acc);
    if (threadIdx.x == 0) out0[blockIdx.x] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Observation-parallel passes
// ------------------------------------------------------------------------------------------

// jx_o = Jc_o x  (both residual rows), x a camera-side vector
template <int KD, typename JT = double>
__global__ void ba_obs_jx_kernel(View V, const double* __restrict__ x, double* __restrict__ jx) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= V.n_obs) return;
  if (V.stop && *V.stop) return;
  const size_t N = (size_t)V.n_obs;
  const JT* __restrict__ Jpo = JSel<JT>::pose(V);
  const JT* __restrict__ Jca = JSel<JT>::cam(V);
  const int pi = V.o_pose[o], ci = V.o_cam[o];
  const int pdim = V.pose_dim[pi], poff = V.pose_off[pi], cdim = V.cam_dim[ci], coff = V.cam_off[ci];
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int c = 0; c < PD; ++c)
    if (c < pdim) {
      const double xv = x[poff + c];
      a0 += (double)Jpo[(size_t)c * N + o] * xv;
      a1 += (double)Jpo[(size_t)(PD + c) * N + o] * xv;
    }
#pragma unroll
  for (int c = 0; c < KD; ++c)
    if (c < cdim) {
      const double xv = x[coff + c];
      a0 += (double)Jca[(size_t)c * N + o] * xv;
      a1 += (double)Jca[(size_t)(KD + c) * N + o] * xv;
    }
  if (V.sens_off) {
    const int si = V.o_sensor[o];
    const int soff = si >= 0 ? V.sens_off[si] : -1;
    if (soff >= 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double xv = x[soff + c];
        a0 += V.Jsens[(size_t)c * N + o] * xv;
        a1 += V.Jsens[(size_t)(6 + c) * N + o] * xv;
      }
    }
  }
  pair_store(jx, o, a0, a1);  // c-order, coalesced
}

// model cost change: -(J step) . (r + J step / 2), step = (dc, dp) already negated
__global__ void __launch_bounds__(256) ba_model_kernel(View V, const double* __restrict__ dc,
                                                      const double* __restrict__ dp,
                                                      double* __restrict__ partials) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (o < V.n_obs) {
    const size_t N = (size_t)V.n_obs;
    const int pi = V.o_pose[o], ci = V.o_cam[o], xi = V.o_pt[o];
    const int pdim = V.pose_dim[pi], poff = V.pose_off[pi], cdim = V.cam_dim[ci], coff = V.cam_off[ci];
    const int ptoff = V.pt_off[xi];
    for (int r = 0; r < 2; ++r) {
      double m = 0.0;
      for (int c = 0; c < pdim; ++c) m += V.Jpose[(size_t)(r * PD + c) * N + o] * dc[poff + c];
      for (int c = 0; c < cdim; ++c) m += V.Jcam[(size_t)(r * V.kd + c) * N + o] * dc[coff + c];
      if (V.sens_off) {
        const int si = V.o_sensor[o];
        const int soff = si >= 0 ? V.sens_off[si] : -1;
        if (soff >= 0)
          for (int c = 0; c < 6; ++c) m += V.Jsens[(size_t)(r * 6 + c) * N + o] * dc[soff + c];
      }
      if (ptoff >= 0)
        for (int c = 0; c < 3; ++c) m += V.Jpt[(size_t)(r * 3 + c) * N + V.c2a[o]] * dp[ptoff + c];
      acc -= m * (V.res[r * N + o] + 0.5 * m);
    }
  }
  acc = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// The same quantity from what the back-substitution left behind: jx = J_c y_c (p-order) is still in
// place, so J step = -(jx + J_p y_p) needs only the point columns, the residual and jx (10 doubles per
// observation instead of 28). Lane per point over its contiguous p-order segment; y_p = dpv (not negated).
__global__ void __launch_bounds__(256) ba_model_from_jx_kernel(View V, const double* __restrict__ jx,
                                                              const double* __restrict__ dpv,
                                                              double* __restrict__ partials) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (j < V.n_points) {
    const size_t N = (size_t)V.n_obs;
    const int off = V.pt_off[j];
    double y[3] = {0.0, 0.0, 0.0};
    if (off >= 0) { y[0] = dpv[off]; y[1] = dpv[off + 1]; y[2] = dpv[off + 2]; }
    for (int o = V.pt_ptr[j]; o < V.pt_ptr[j + 1]; ++o) {
      const double2 j2 = pair_load(jx, V.a2c[o]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double m = r ? j2.y : j2.x;
        if (off >= 0)
          m += V.Jpt[(size_t)(r * 3) * N + o] * y[0] + V.Jpt[(size_t)(r * 3 + 1) * N + o] * y[1] +
               V.Jpt[(size_t)(r * 3 + 2) * N + o] * y[2];
        m = -m;
        acc -= m * (V.res_p[r * N + o] + 0.5 * m);
      }
    }
  }
  acc = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------
// Camera-side reductions: one wave per chunk of a parameter block's observation list
// ------------------------------------------------------------------------------------------

// fp32 operator copy of the same column (never for variable sensor_from_rig blocks: the fp32 operator is
// not enabled for problems that have them)
__device__ __forceinline__ const float* blk_col32(const View& V, int kind, int r, int c) {
  const size_t N = (size_t)V.n_obs;
  return kind == 0 ? V.Jpose32 + (size_t)(r * PD + c) * N : V.Jcam32 + (size_t)(r * V.kd + c) * N;
}
__device__ __forceinline__ const double* blk_col(const View& V, int kind, int r, int c) {
  const size_t N = (size_t)V.n_obs;
  if (kind == 2) return V.Jsens + (size_t)(r * 6 + c) * N;  // variable sensor_from_rig block
  return kind == 0 ? V.Jpose + (size_t)(r * PD + c) * N : V.Jcam + (size_t)(r * V.kd + c) * N;
}

// y_b += J_b^T v  (v: 2 rows per observation). With DIAG: also diag_b += colsq(J_b).
// DIAG (gradient pass): v is the c-order residual in two planes. Otherwise v is the c-order pair vector of an
// implicit product (one 16-byte load per observation).
template <bool DIAG, int BD, typename JT = double>
__global__ void __launch_bounds__(64) ba_block_jtv_kernel(View V, const double* __restrict__ v,
                                                         double* __restrict__ y, double* __restrict__ diag) {
  if (V.stop && *V.stop) return;
  const int ch = blockIdx.x;
  const int b = V.chunk_blk[ch];
  const int kind = V.blk_kind[b], dim = V.blk_dim[b], off = V.blk_off[b];
  const size_t N = (size_t)V.n_obs;
  double acc[BD], dacc[BD];
#pragma unroll
  for (int c = 0; c < BD; ++c) acc[c] = dacc[c] = 0.0;
  // two observations per lane and trip (o, o + 64): twice the loads in flight; the accumulation order
  // of a lane stays o, o + 64, o + 128, ... as in the rolled loop
  const int end = V.chunk_end[ch];
  for (int o = V.chunk_beg[ch] + threadIdx.x; o < end; o += 128) {
    const int o2 = o + 64;
    const bool two = o2 < end;
    const int oz = two ? o2 : o;
    double v0, v1, w0, w1;
    if (DIAG) {
      v0 = v[o]; v1 = v[N + o];
      w0 = v[oz]; w1 = v[N + oz];
    } else {
      const double2 p0 = pair_load(v, o), p1 = pair_load(v, oz);
      v0 = p0.x; v1 = p0.y;
      w0 = p1.x; w1 = p1.y;
    }
    double j0[BD], j1[BD], k0[BD], k1[BD];
#pragma unroll
    for (int c = 0; c < BD; ++c)
      if (c < dim) {
        if constexpr (sizeof(JT) == 4) {
          const float* c0 = blk_col32(V, kind, 0, c);
          const float* c1 = blk_col32(V, kind, 1, c);
          j0[c] = (double)c0[o]; j1[c] = (double)c1[o];
          k0[c] = (double)c0[oz]; k1[c] = (double)c1[oz];
        } else {
          const double* c0 = blk_col(V, kind, 0, c);
          const double* c1 = blk_col(V, kind, 1, c);
          j0[c] = c0[o]; j1[c] = c1[o];
          k0[c] = c0[oz]; k1[c] = c1[oz];
        }
      }
#pragma unroll
    for (int c = 0; c < BD; ++c)
      if (c < dim) {
        acc[c] += j0[c] * v0 + j1[c] * v1;
        if (DIAG) dacc[c] += j0[c] * j0[c] + j1[c] * j1[c];
        if (two) {
          acc[c] += k0[c] * w0 + k1[c] * w1;
          if (DIAG) dacc[c] += k0[c] * k0[c] + k1[c] * k1[c];
        }
      }
  }
  (void)off;
#pragma unroll
  for (int c = 0; c < BD; ++c)
    if (c < dim) {
      const double s = wave_sum(acc[c]);
      if (threadIdx.x == 0) V.cpart[(size_t)ch * BD * BD + c] = s;
      if (DIAG) {
        const double d = wave_sum(dacc[c]);
        if (threadIdx.x == 0) V.cpart[(size_t)ch * BD * BD + BD + c] = d;
      }
    }
}

// A block with thousands of chunks -- ONE camera shared by every image of a 1 000-image problem has 4 000 -- makes every
// finalize kernel below a single thread walking thousands of dependent loads (1.3 ms each at BA-1 with shared intrinsics:
// 60 % of that solve, profiles/r05_ba_shared_intrinsics_kernel_stats_before.csv). For such HEAVY blocks this kernel first
// sums the first `width` entries of all the block's rows into its first row: one workgroup per (heavy block, 64 or 16
// entries), the threads of an entry stride over the rows, their partial sums are added in thread order -- a fixed tree,
// the same bits on every run. The finalize kernels then read that one row (View::blk_fin_end).
constexpr int kHeavyChunks = 64;  // (development switch COLMAP_AMD_BA_HEAVY_CHUNKS: the tests lower it to reach the path)
__global__ void __launch_bounds__(1024) ba_cpart_heavy_reduce_kernel(View V, int width) {
  __shared__ double part[1024];
  const int b = V.heavy_blk[blockIdx.x];
  const int epw = width <= 16 ? 16 : 64;            // entries per workgroup
  const int groups = 1024 / epw;
  const int el = threadIdx.x % epw, g = threadIdx.x / epw;
  const int e = el + epw * blockIdx.y;
  const int bb = V.bd * V.bd;
  const int ch0 = V.blk_chunk_ptr[b], ch1 = V.blk_chunk_ptr[b + 1];
  double s = 0.0;
  if (e < width) {
    // eight rows per round trip (a thread's 60-odd rows used to be one dependent load each), added in row order
    constexpr int U = 8;
    for (int ch = ch0 + g; ch < ch1; ch += U * groups) {
      double row[U];
#pragma unroll
      for (int k = 0; k < U; ++k) row[k] = ch + k * groups < ch1 ? V.cpart[(size_t)(ch + k * groups) * bb + e] : 0.0;
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (ch + k * groups < ch1) s += row[k];
    }
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && e < width) {
    double t = 0.0;
    for (int k = 0; k < groups; ++k) t += part[k * epw + el];
    V.cpart[(size_t)ch0 * bb + e] = t;
  }
}

// y_b = sum over the block's chunks, in chunk order (deterministic); lane per block
template <bool DIAG>
__global__ void ba_block_vec_finalize_kernel(View V, double* __restrict__ y, double* __restrict__ diag) {
  // thread per (block, component) -- a lane per block walked its components' loads one after the other
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = t / V.bd, c = t - b * V.bd;
  if (b >= V.n_blk) return;
  const int dim = V.blk_dim[b], off = V.blk_off[b];
  if (c >= dim) return;
  double s = 0.0, d = 0.0;
  for (int ch = V.blk_chunk_ptr[b]; ch < V.blk_fin_end[b]; ++ch) {
    s += V.cpart[(size_t)ch * V.bd * V.bd + c];
    if (DIAG) d += V.cpart[(size_t)ch * V.bd * V.bd + V.bd + c];
  }
  y[off + c] = s;  // every camera-side entry belongs to exactly one block: no memset before, no accumulate
  if (DIAG) diag[off + c] = d;
}

// q_b = Dc_b^2 x_b + sum over the block's chunks (the tail of an implicit product on a single GPU without
// priors: ba_block_vec_finalize_kernel<false> + ba_dsq_x_kernel + ba_add_kernel in one launch, same operations)
__global__ void ba_block_vec_finalize_q_kernel(View V, const double* __restrict__ Dc, const double* __restrict__ x,
                                               double* __restrict__ q) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = t / V.bd, c = t - b * V.bd;
  if (b >= V.n_blk) return;
  const int dim = V.blk_dim[b], off = V.blk_off[b];
  if (c >= dim) return;
  double s = 0.0;
  for (int ch = V.blk_chunk_ptr[b]; ch < V.blk_fin_end[b]; ++ch) s += V.cpart[(size_t)ch * V.bd * V.bd + c];
  q[off + c] = Dc[off + c] * Dc[off + c] * x[off + c] + s;
}

// M_b (+)= sum over the block's chunks of the dim x dim partials; lane per block
template <bool ACCUMULATE>
__global__ void ba_block_mat_finalize_kernel(View V, double* __restrict__ M) {
  // thread per (block, entry): a lane per block walked dim^2 x chunks dependent loads one after the other (37 us for
  // the 2 000 blocks of 1 000 images); the chunk order of the sum is unchanged
  const int bb = V.bd * V.bd;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = t / bb, e = t - b * bb;
  if (b >= V.n_blk) return;
  const int dim = V.blk_dim[b];
  if (e >= dim * dim) return;
  double* Mb = M + V.blk_moff[b];
  double s = 0.0;
  for (int ch = V.blk_chunk_ptr[b]; ch < V.blk_fin_end[b]; ++ch) s += V.cpart[(size_t)ch * bb + e];
  Mb[e] = ACCUMULATE ? Mb[e] + s : s;
}

// Schur-Jacobi diagonal blocks on the f64 matrix cores.
//
// M_b = B_bb - sum_j W_bj C_j^-1 W_bj^T with W_bj = sum_{o in (b,j)} J_b,o^T E_o. For an observation
// that is the only one of its point in block b (every pose block of a COLMAP track, every
// intrinsics block of a per-image camera) the point's term is J_b,o^T G_o J_b,o with the 2 x 2
// G_o = E_o C_j^-1 E_o^T, so the block is ONE Gram-like contraction
//     M_b = sum_o J_b,o^T (I - G_o) J_b,o
// -- A = a 4-row slab of J_b (two observations x two residual rows), B = (I - G) applied to the
// same slab, v_mfma_f64_16x16x4_f64 accumulates the 16 x 16 tile (columns >= dim are zero). G_o is
// computed where C^-1 is local (ba_obs_schur_g_kernel, p-order) and stored in c-order. Observation
// pairs of one point inside one block (shared intrinsics, rig frames) add their cross terms in
// ba_block_schur_cross_kernel; the self term is already in (I - G).
// One wave per chunk. Lane l holds column i = l & 15 and slab row k = l >> 4 (k = 2 * observation
// in slab + residual row). C/D layout: col = l & 15, row = (l >> 4) + 4 * reg.
typedef double v4f64 __attribute__((ext_vector_type(4)));

// G_o = E_o C_j^-1 E_o^T (g00, g01, g11), lane per observation in p-order, stored at the c-order
// position as a record {g00, g01, g11, 0}; constant points (no point block) have G = 0.
__global__ void __launch_bounds__(256) ba_obs_schur_g_kernel(View V, const double* __restrict__ Cinv,
                                                             double* __restrict__ G) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= V.n_obs) return;
  const size_t N = (size_t)V.n_obs;
  const int c = V.a2c[a];
  const int j = V.o_pt[c];
  double g00 = 0.0, g01 = 0.0, g11 = 0.0;
  if (V.pt_off[j] >= 0) {
    const double* Ci = Cinv + 9 * (size_t)j;
    const double e00 = V.Jpt[a], e01 = V.Jpt[N + a], e02 = V.Jpt[2 * N + a];
    const double e10 = V.Jpt[3 * N + a], e11 = V.Jpt[4 * N + a], e12 = V.Jpt[5 * N + a];
    // T = E C^-1 (2 x 3)
    const double t00 = e00 * Ci[0] + e01 * Ci[3] + e02 * Ci[6], t01 = e00 * Ci[1] + e01 * Ci[4] + e02 * Ci[7],
                 t02 = e00 * Ci[2] + e01 * Ci[5] + e02 * Ci[8];
    const double t10 = e10 * Ci[0] + e11 * Ci[3] + e12 * Ci[6], t11 = e10 * Ci[1] + e11 * Ci[4] + e12 * Ci[7],
                 t12 = e10 * Ci[2] + e11 * Ci[5] + e12 * Ci[8];
    g00 = t00 * e00 + t01 * e01 + t02 * e02;
    g01 = t00 * e10 + t01 * e11 + t02 * e12;
    g11 = t10 * e10 + t11 * e11 + t12 * e12;
  }
  // one 32-byte record per observation at its c-order position (a whole sector per lane; three planes meant three
  // scattered 8-byte stores, each a partial write of its own sector)
  double2* rec = reinterpret_cast<double2*>(G + 4 * (size_t)c);
  rec[0] = make_double2(g00, g01);
  rec[1] = make_double2(g11, 0.0);
}

__global__ void __launch_bounds__(64) ba_block_gram_kernel(View V, const double* __restrict__ G) {
  const int ch = blockIdx.x;
  const int b = V.chunk_blk[ch];
  const int kind = V.blk_kind[b], dim = V.blk_dim[b];
  const int lane = threadIdx.x;
  const int i = lane & 15, k = lane >> 4;  // column, row-in-slab
  const int r = k & 1, oo = k >> 1;        // residual row, observation within the slab
  const int beg = V.chunk_beg[ch], end = V.chunk_end[ch];
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  const bool col_ok = i < dim;
  const double* col0 = col_ok ? blk_col(V, kind, 0, i) : nullptr;
  const double* col1 = col_ok ? blk_col(V, kind, 1, i) : nullptr;
  constexpr int U = 4;  // slabs per trip: the loads of four MFMAs are in flight together
  for (int s = beg; s < end; s += 2 * U) {
    double a[U], bb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = s + 2 * u + oo;
      a[u] = bb[u] = 0.0;
      if (col_ok && idx < end) {
        const double j0 = col0[idx], j1 = col1[idx];
        const double2 ga = reinterpret_cast<const double2*>(G)[2 * (size_t)idx];
        const double g00 = ga.x, g01 = ga.y, g11 = G[4 * (size_t)idx + 2];
        a[u] = r ? j1 : j0;
        bb[u] = r ? j1 - (g01 * j0 + g11 * j1) : j0 - (g00 * j0 + g01 * j1);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bb[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = k + 4 * reg;
    if (row < dim && i < dim) V.cpart[(size_t)ch * V.bd * V.bd + row * dim + i] = acc[reg];
  }
}

// The same contraction with the operands staged through LDS: per trip the wave loads 32 consecutive
// observations of every Jacobian column of the block (and of G) as contiguous 256-byte segments -- half a
// wave per (row, column) pair -- and the 16 slabs of the trip feed the matrix core from LDS. In the kernel
// above every lane loads its own operand: 16 different columns per load instruction. Same slabs, same zero
// padding; the trips of a chunk are dealt to four waves whose tiles are added at the end, so the result differs
// from the kernel above by the order of the additions (rounding).
constexpr int GRAM_TRIP = 32;
constexpr int GRAM_WAVES = 4;
template <int BD>  // widest block of the problem: bounds the prefetch registers (6, 8 or 16 column pairs per lane)
__global__ void __launch_bounds__(64 * GRAM_WAVES, BD <= 8 ? 4 : 2) ba_block_gram_lds_kernel(View V, const double* __restrict__ G) {
  // One workgroup per chunk, FOUR waves, wave w taking the trips w, w + 4, ... of the chunk with its own LDS tile and
  // accumulator; the four partial tiles are added in wave order at the end (a fixed tree). One wave per chunk left
  // ~2 waves per SIMD on the chip (a chunk is up to 2 048 observations, a camera one or two chunks), each a serial
  // chain of 64 trips of (LDS round trip + 16 dependent-issue matrix-core instructions): latency, not bandwidth.
  __shared__ double sJ[GRAM_WAVES][2][16][GRAM_TRIP + 1];  // [wave][row][column][observation], padded against bank conflicts
  __shared__ double sG[GRAM_WAVES][3][GRAM_TRIP + 1];
  const int ch = blockIdx.x;
  const int b = V.chunk_blk[ch];
  const int kind = V.blk_kind[b], dim = V.blk_dim[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, k = lane >> 4;  // column, row-in-slab
  const int r = k & 1, oo = k >> 1;        // residual row, observation within the slab
  const int half = lane >> 5, lo = lane & 31;
  const int beg = V.chunk_beg[ch], end = V.chunk_end[ch];
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  const bool col_ok = i < dim;
  // The operands of the wave's next trip are loaded into registers while the matrix core works on this one (a lane's
  // share: one 32-observation segment of <= 16 (row, column) pairs -- pairs cr = half, half + 2, ... -- and
  // up to two entries of G).
  double pj[BD], pg[2];
  auto prefetch = [&](int s) {
    const int n = min(GRAM_TRIP, end - s);
#pragma unroll
    for (int q = 0; q < BD; ++q) {
      const int cr = half + 2 * q;
      pj[q] = (cr < 2 * dim && lo < n) ? blk_col(V, kind, cr & 1, cr >> 1)[s + lo] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // the trip's 32 records of G: 1 KB, contiguous
      const int e = lane + 64 * q;
      pg[q] = (e >> 2) < n ? G[4 * (size_t)s + e] : 0.0;
    }
  };
  auto wave_sync = [] {  // the tile is handed from lane to lane of ONE wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  constexpr int STEP = GRAM_WAVES * GRAM_TRIP;
  const int first = beg + wave * GRAM_TRIP;
  if (first < end) prefetch(first);
  for (int s = first; s < end; s += STEP) {
#pragma unroll
    for (int q = 0; q < BD; ++q) {
      const int cr = half + 2 * q;
      if (cr < 2 * dim) sJ[wave][cr & 1][cr >> 1][lo] = pj[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = lane + 64 * q;
      if ((e & 3) < 3) sG[wave][e & 3][e >> 2] = pg[q];
    }
    wave_sync();
    if (s + STEP < end) prefetch(s + STEP);
#pragma unroll 4
    for (int u = 0; u < GRAM_TRIP / 2; ++u) {
      const int o = 2 * u + oo;
      double a = 0.0, bb = 0.0;
      if (col_ok) {
        const double j0 = sJ[wave][0][i][o], j1 = sJ[wave][1][i][o];
        const double g00 = sG[wave][0][o], g01 = sG[wave][1][o], g11 = sG[wave][2][o];
        a = r ? j1 : j0;
        bb = r ? j1 - (g01 * j0 + g11 * j1) : j0 - (g00 * j0 + g01 * j1);
      }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
    }
    wave_sync();
  }
  double* part = &sJ[wave][0][0][0];  // the wave's own tile memory (its last trip is behind a wave_sync): [4][64]
  if (wave > 0) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[reg * 64 + lane] = acc[reg];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = k + 4 * reg;
    double t = acc[reg];
#pragma unroll
    for (int w = 1; w < GRAM_WAVES; ++w) t += (&sJ[w][0][0][0])[reg * 64 + lane];
    if (row < dim && i < dim) V.cpart[(size_t)ch * V.bd * V.bd + row * dim + i] = t;
  }
}

// Cross terms - sum_{o != o2 in (b,j)} (J_b,o^T E_o) C_j^-1 (J_b,o2^T E_o2)^T of observation pairs of one
// point inside one block (only launched when such pairs exist). One lane per observation.
// W_o = J_b,o^T E_o (dim x 3) of every observation that has a partner of its point in one of its blocks, stored in
// p-order: the partners of an observation are then NEIGHBOURS in memory (ba_block_schur_cross_kernel walks them), and
// a W is computed once per linearisation instead of once per partner visit.
template <int BD>
__global__ void ba_obs_w_kernel(View V) {
  // a lane per P-ORDER slot: the point columns and the stores of W (wd x 3 doubles per slot, neighbours in memory) are
  // coalesced, only the block's 2 x wd camera-side entries are gathered through a2c (a lane per c-order slot gathered
  // the six point columns AND scattered the 12 .. 24 stores: 370 us at BA-1 with one shared camera)
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= V.n_obs) return;
  const int o = V.a2c[a];
  const unsigned so = V.solo[o];
  const size_t N = (size_t)V.n_obs;
  double e[2][3];
  bool have_e = false;
#pragma unroll
  for (int kind = 0; kind < 3; ++kind) {
    if (V.Wp[kind] == nullptr || ((so >> kind) & 1) || V.a_boff[kind][a] < 0) continue;
    if (!have_e) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) e[r][c] = V.Jpt[(size_t)(r * 3 + c) * N + a];
      have_e = true;
    }
    const int wd = V.wdim[kind];
    double* w = V.Wp[kind] + (size_t)a * wd * 3;
#pragma unroll
    for (int x = 0; x < BD; ++x) {
      if (x >= wd) continue;
      const double j0 = blk_col(V, kind, 0, x)[o], j1 = blk_col(V, kind, 1, x)[o];
#pragma unroll
      for (int c = 0; c < 3; ++c) w[x * 3 + c] = j0 * e[0][c] + j1 * e[1][c];
    }
  }
}

template <int BD>
__global__ void __launch_bounds__(64) ba_block_schur_cross_kernel(View V, const double* __restrict__ Cinv) {
  const int ch = blockIdx.x;
  const int b = V.chunk_blk[ch];
  const int kind = V.blk_kind[b], dim = V.blk_dim[b], boff = V.blk_off[b];
  const size_t N = (size_t)V.n_obs;
  const int* ab = V.a_boff[kind];
  const double* Wk = V.Wp[kind];
  const int wd = V.wdim[kind];
  double acc[BD * BD];
#pragma unroll
  for (int e = 0; e < BD * BD; ++e) acc[e] = 0.0;
  if (ab != nullptr)  // (a kind without pairs: every observation is solo, the partials below are zeros)
  for (int o = V.chunk_beg[ch] + threadIdx.x; o < V.chunk_end[ch]; o += 64) {
    if ((V.solo[o] >> kind) & 1) continue;  // no partner in this block
    const int xi = V.o_pt[o];
    if (V.pt_off[xi] < 0) continue;
    const int a = V.c2a[o];
    const double* Ci = Cinv + 9 * (size_t)xi;
    double W1[BD][3];
#pragma unroll
    for (int x = 0; x < BD; ++x)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        W1[x][c] = (x < dim) ? blk_col(V, kind, 0, x)[o] * V.Jpt[(size_t)c * N + a] +
                                   blk_col(V, kind, 1, x)[o] * V.Jpt[(size_t)(3 + c) * N + a]
                             : 0.0;
    double T[BD][3];  // W1 * Cinv
#pragma unroll
    for (int x = 0; x < BD; ++x)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[x][c] = W1[x][0] * Ci[c] + W1[x][1] * Ci[3 + c] + W1[x][2] * Ci[6 + c];
    for (int a2 = V.pt_ptr[xi]; a2 < V.pt_ptr[xi + 1]; ++a2) {
      if (a2 == a) continue;  // the self term is part of (I - G) in the Gram kernel
      if (ab[a2] != boff) continue;
      const double* w2 = Wk + (size_t)a2 * wd * 3;  // the partner's W (ba_obs_w_kernel: the same expression, the same bits)
#pragma unroll
      for (int y = 0; y < BD; ++y) {
        if (y >= dim) continue;
        const double W2[3] = {w2[y * 3], w2[y * 3 + 1], w2[y * 3 + 2]};
#pragma unroll
        for (int x = 0; x < BD; ++x)
          if (x < dim) acc[x * BD + y] -= T[x][0] * W2[0] + T[x][1] * W2[1] + T[x][2] * W2[2];
      }
    }
  }
#pragma unroll
  for (int x = 0; x < BD; ++x)
#pragma unroll
    for (int y = 0; y < BD; ++y)
      if (x < dim && y < dim) {
        const double sx = wave_sum(acc[x * BD + y]);
        if (threadIdx.x == 0) V.cpart[(size_t)ch * BD * BD + x * dim + y] = sx;
      }
}

// ------------------------------------------------------------------------------------------
// Image-sharded solves: the cross-rank part of the Schur-Jacobi blocks of SHARED intrinsics.
// For an intrinsics block b and a point p let W_{b,p} = sum over the observations o of p that use b of
// J_b,o^T J_p,o (dim_b x 3). The block is M_b = B_b - sum_p W_{b,p} C_p^-1 W_{b,p}^T. A rank sees only its own
// observations: its G term plus its pair term is B_b^r - sum_p W^r C^-1 W^r^T, and the sum of that over the
// ranks misses every pair of observations of one point that sit on DIFFERENT ranks. Correction, for the
// (point, block) incidences whose observations span more than one rank (the host lists them; every rank sees the
// whole problem): all-reduce W, then
//   M_b += sum_p W^r C^-1 W^r^T                      (every rank, its own part: undoes what it subtracted)
//   M_b -= sum_{p : p mod world = rank} W C^-1 W^T    (the exact term, each incidence on one rank)
// before the all-reduce of M. The image-sharded solve then has the single-GPU preconditioner, CG trajectory and
// iteration counts (tests/test_ba_gpu.py: test_three_rank_sharded_solve_with_shared_intrinsics).
// ------------------------------------------------------------------------------------------
struct IncView {
  int n;                 // incidences that span ranks
  const int* pt;         // [n] point
  const int* blk;        // [n] block
  const int* ptr;        // [n + 1] this rank's observations of the incidence (c-order indices) ...
  const int* obs;        // ... as a CSR list
  // The incidences are sorted by block; a block's run is cut into chunks of <= kIncChunk of them (fixed-order sums:
  // ba_inc_correct_kernel writes one partial per chunk, ba_inc_finalize_kernel adds a block's partials in chunk order)
  int n_chunks;
  const int* chunk_blk;  // [n_chunks]
  const int* chunk_beg;  // [n_chunks + 1] first incidence of the chunk (chunk_beg[c + 1] = its end)
  const int* blk_chunk;  // [n_blk + 1] chunks of every block
  double* part;          // [n_chunks][KDT * KDT] partial sums
};
constexpr int kIncChunk = 128;
template <int KDT>
__global__ void ba_inc_w_kernel(View V, IncView I, double* __restrict__ W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I.n) return;
  const int dim = V.blk_dim[I.blk[i]];
  const size_t N = (size_t)V.n_obs;
  double w[KDT][3];
#pragma unroll
  for (int x = 0; x < KDT; ++x) w[x][0] = w[x][1] = w[x][2] = 0.0;
  for (int k = I.ptr[i]; k < I.ptr[i + 1]; ++k) {
    const int o = I.obs[k], a = V.c2a[o];
#pragma unroll
    for (int x = 0; x < KDT; ++x)
      if (x < dim) {
        const double j0 = blk_col(V, 1, 0, x)[o], j1 = blk_col(V, 1, 1, x)[o];
#pragma unroll
        for (int c = 0; c < 3; ++c) w[x][c] += j0 * V.Jpt[(size_t)c * N + a] + j1 * V.Jpt[(size_t)(3 + c) * N + a];
      }
  }
#pragma unroll
  for (int x = 0; x < KDT; ++x)
#pragma unroll
    for (int c = 0; c < 3; ++c) W[((size_t)i * KDT + x) * 3 + c] = w[x][c];
}
// One workgroup per chunk of incidences of ONE block, thread per entry (x, y) of the block: the thread walks the
// chunk's incidences in list order and writes its partial sum -- no atomics, the order of every sum is fixed by the
// host-built lists (a solve is reproducible run to run whatever the hardware's timing).
template <int KDT>
__global__ void __launch_bounds__(KDT * KDT) ba_inc_correct_kernel(View V, IncView I, const double* __restrict__ Cinv,
                                                                 const double* __restrict__ Wloc, const double* __restrict__ Wtot,
                                                                 int rank, int world) {
  const int ch = blockIdx.x;
  const int b = I.chunk_blk[ch], dim = V.blk_dim[b];
  const int x = threadIdx.x / KDT, y = threadIdx.x % KDT;
  double acc = 0.0;
  if (x < dim && y < dim) {
    for (int i = I.chunk_beg[ch]; i < I.chunk_beg[ch + 1]; ++i) {
      const int xi = I.pt[i];
      if (V.pt_off[xi] < 0) continue;  // a constant point has no C^-1: its observations do not couple
      const double* Ci = Cinv + 9 * (size_t)xi;
      const bool mine = xi % world == rank;
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !mine) break;
        const double* Wp = (pass == 0 ? Wloc : Wtot) + (size_t)i * KDT * 3;
        const double t0 = Wp[x * 3 + 0] * Ci[0] + Wp[x * 3 + 1] * Ci[3] + Wp[x * 3 + 2] * Ci[6];
        const double t1 = Wp[x * 3 + 0] * Ci[1] + Wp[x * 3 + 1] * Ci[4] + Wp[x * 3 + 2] * Ci[7];
        const double t2 = Wp[x * 3 + 0] * Ci[2] + Wp[x * 3 + 1] * Ci[5] + Wp[x * 3 + 2] * Ci[8];
        const double term = t0 * Wp[y * 3 + 0] + t1 * Wp[y * 3 + 1] + t2 * Wp[y * 3 + 2];
        acc += pass == 0 ? term : -term;
      }
    }
  }
  I.part[(size_t)ch * KDT * KDT + threadIdx.x] = acc;
}
// M_b += the block's chunk partials, in chunk order; thread per (block, entry)
template <int KDT>
__global__ void ba_inc_finalize_kernel(View V, IncView I, double* __restrict__ M) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = t / (KDT * KDT), e = t % (KDT * KDT);
  if (b >= V.n_blk) return;
  const int c0 = I.blk_chunk[b], c1 = I.blk_chunk[b + 1];
  if (c0 == c1) return;
  const int dim = V.blk_dim[b], x = e / KDT, y = e % KDT;
  if (x >= dim || y >= dim) return;
  double s = 0.0;
  for (int c = c0; c < c1; ++c) s += I.part[(size_t)c * KDT * KDT + e];
  M[V.blk_moff[b] + x * dim + y] += s;
}

// ------------------------------------------------------------------------------------------
// Single GPU: the pair terms of the Schur-Jacobi blocks per INCIDENCE instead of per pair.
// For a point p and a block b that holds t >= 2 of p's observations the cross terms are
//   - sum_{o != o2} W_o C^-1 W_o2^T  =  - ( Ws C^-1 Ws^T - sum_o W_o C^-1 W_o^T ),   Ws = sum_o W_o,
// t + 1 products instead of t (t - 1): ba_block_schur_cross_kernel walks every partner of every observation (O(t^2) per
// incidence, 1.26 ms at BA-1 with one shared camera). The host lists the incidences sorted by block (PairView = the
// IncView layout, `obs` holding p-order observation indices), W_o comes from the p-order buffer of ba_obs_w_kernel (an
// incidence's members are neighbours there). One workgroup per chunk of <= kPairChunk incidences of ONE block, thread per
// entry (x, y), partial per chunk; ba_pair_finalize_kernel adds a block's partials with a fixed tree. No atomics.
// (Small chunks: a workgroup walks its incidences one after the other with dim^2 of its KDT^2 lanes busy -- what makes it
// fast is many workgroups in flight, 6 250 at BA-1 with one shared camera.)
// ------------------------------------------------------------------------------------------
constexpr int kPairChunk = 32;
template <int KDT>
__global__ void __launch_bounds__(KDT * KDT) ba_pair_cross_kernel(View V, IncView I, const double* __restrict__ Cinv) {
  const int ch = blockIdx.x;
  const int b = I.chunk_blk[ch], dim = V.blk_dim[b], kind = V.blk_kind[b];
  const int x = threadIdx.x / KDT, y = threadIdx.x % KDT;
  const double* Wk = V.Wp[kind];
  const int ws = V.wdim[kind] * 3;
  double acc = 0.0;
  if (x < dim && y < dim) {
    for (int i = I.chunk_beg[ch]; i < I.chunk_beg[ch + 1]; ++i) {
      const double* Ci = Cinv + 9 * (size_t)I.pt[i];
      double sx[3] = {0.0, 0.0, 0.0}, sy[3] = {0.0, 0.0, 0.0}, self = 0.0;
      for (int k = I.ptr[i]; k < I.ptr[i + 1]; ++k) {
        const double* w = Wk + (size_t)I.obs[k] * ws;
        const double wx0 = w[x * 3], wx1 = w[x * 3 + 1], wx2 = w[x * 3 + 2];
        const double wy0 = w[y * 3], wy1 = w[y * 3 + 1], wy2 = w[y * 3 + 2];
        sx[0] += wx0; sx[1] += wx1; sx[2] += wx2;
        sy[0] += wy0; sy[1] += wy1; sy[2] += wy2;
        const double t0 = wx0 * Ci[0] + wx1 * Ci[3] + wx2 * Ci[6];
        const double t1 = wx0 * Ci[1] + wx1 * Ci[4] + wx2 * Ci[7];
        const double t2 = wx0 * Ci[2] + wx1 * Ci[5] + wx2 * Ci[8];
        self += t0 * wy0 + t1 * wy1 + t2 * wy2;
      }
      const double t0 = sx[0] * Ci[0] + sx[1] * Ci[3] + sx[2] * Ci[6];
      const double t1 = sx[0] * Ci[1] + sx[1] * Ci[4] + sx[2] * Ci[7];
      const double t2 = sx[0] * Ci[2] + sx[1] * Ci[5] + sx[2] * Ci[8];
      acc -= (t0 * sy[0] + t1 * sy[1] + t2 * sy[2]) - self;
    }
  }
  I.part[(size_t)ch * KDT * KDT + threadIdx.x] = acc;
}
// M_b += the partials of block b's chunks: one workgroup per block that has pair incidences (I.pair_blk), 64 entries per
// grid.y slice, 16 threads per entry striding over the chunks, their sums added in thread order (a fixed tree).
template <int KDT>
__global__ void __launch_bounds__(1024) ba_pair_finalize_kernel(View V, IncView I, const int* __restrict__ pair_blk,
                                                                double* __restrict__ M) {
  __shared__ double part[1024];
  const int b = pair_blk[blockIdx.x];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = el + 64 * blockIdx.y;
  const int c0 = I.blk_chunk[b], c1 = I.blk_chunk[b + 1];
  double s = 0.0;
  if (e < KDT * KDT)
    for (int c = c0 + g; c < c1; c += 16) s += I.part[(size_t)c * KDT * KDT + e];
  part[threadIdx.x] = s;
  __syncthreads();
  const int dim = V.blk_dim[b], x = e / KDT, y = e % KDT;
  if (g == 0 && e < KDT * KDT && x < dim && y < dim) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += part[k * 64 + el];
    M[V.blk_moff[b] + x * dim + y] += t;
  }
}

// M_b += Dc^2 on the diagonal, then invert (Gauss-Jordan with partial pivoting); lane per block
template <int BD>
__global__ void __launch_bounds__(64) ba_block_invert_kernel(View V, const double* __restrict__ Dc, const double* __restrict__ M,
                                       double* __restrict__ Minv) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= V.n_blk) return;
  const int n = V.blk_dim[b], off = V.blk_off[b];
  double A[BD][2 * BD];
  if constexpr (BD <= 8) {
    // The same Gauss-Jordan elimination with partial pivoting, every loop unrolled to the template width and
    // predicated on n, the row exchange written as selects: all indices are compile-time constants, so the
    // augmented matrix lives in registers (the rolled version below keeps it in scratch: 80 us for 2 000 blocks).
#pragma unroll
    for (int i = 0; i < BD; ++i) {
#pragma unroll
      for (int j = 0; j < BD; ++j) A[i][j] = (i < n && j < n) ? M[V.blk_moff[b] + i * n + j] : (i == j ? 1.0 : 0.0);
#pragma unroll
      for (int j = 0; j < BD; ++j) A[i][BD + j] = (i == j) ? 1.0 : 0.0;
      if (i < n) A[i][i] += Dc[off + i] * Dc[off + i];
    }
#pragma unroll
    for (int c = 0; c < BD; ++c) {
      if (c < n) {
        int piv = c;
        double best = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < BD; ++r)
          if (r < n && fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
#pragma unroll
        for (int r = c + 1; r < BD; ++r) {
          const bool sw = piv == r;
#pragma unroll
          for (int j = 0; j < 2 * BD; ++j) {
            const double t = A[c][j], u = A[r][j];
            A[c][j] = sw ? u : t;
            A[r][j] = sw ? t : u;
          }
        }
        const double inv = 1.0 / A[c][c];
#pragma unroll
        for (int j = 0; j < 2 * BD; ++j) A[c][j] *= inv;
#pragma unroll
        for (int r = 0; r < BD; ++r) {
          if (r == c || r >= n) continue;
          const double f = A[r][c];
#pragma unroll
          for (int j = 0; j < 2 * BD; ++j) A[r][j] -= f * A[c][j];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BD; ++i)
#pragma unroll
      for (int j = 0; j < BD; ++j)
        if (i < n && j < n) Minv[V.blk_moff[b] + i * n + j] = A[i][BD + j];
    return;
  }
  for (int i = 0; i < BD; ++i)
    for (int j = 0; j < 2 * BD; ++j) A[i][j] = 0.0;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) A[i][j] = M[V.blk_moff[b] + i * n + j];
    A[i][i] += Dc[off + i] * Dc[off + i];
    A[i][BD + i] = 1.0;
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 2 * BD; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
    const double inv = 1.0 / A[c][c];
    for (int j = 0; j < 2 * BD; ++j) A[c][j] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = A[r][c];
      for (int j = 0; j < 2 * BD; ++j) A[r][j] -= f * A[c][j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Minv[V.blk_moff[b] + i * n + j] = A[i][BD + j];
}

// ------------------------------------------------------------------------------------------
// Small vector kernels (camera-side vectors are a few thousand entries)
// ------------------------------------------------------------------------------------------
__global__ void ba_lm_diag_kernel(int n, const double* __restrict__ diag, double radius, double lo, double hi,
                                  double* __restrict__ D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) D[i] = sqrt(fmin(fmax(diag[i], lo), hi) / radius);
}
__global__ void ba_scale_kernel(int n, const double* __restrict__ diag, int enable, double* __restrict__ s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) s[i] = enable ? 1.0 / (1.0 + sqrt(diag[i])) : 1.0;
}
__global__ void ba_add_kernel(int n, const double* __restrict__ x, double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}
__global__ void ba_dsq_x_kernel(int n, const double* __restrict__ D, const double* __restrict__ x,
                                double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = D[i] * D[i] * x[i];
}
// z_b = Minv_b r_b, lane per block over many workgroups; rho = r.z as one partial per workgroup
// (fixed tree inside a workgroup, partials added in index order by pcg_rho: deterministic).
__global__ void __launch_bounds__(256) ba_pcg_precond_kernel(View V, const double* __restrict__ Minv,
                                                             const double* __restrict__ r,
                                                             double* __restrict__ z, double* __restrict__ part) {
  double rho = 0.0;
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < V.n_blk) {
    const int n = V.blk_dim[b], off = V.blk_off[b];
    const double* Mi = Minv + V.blk_moff[b];
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += Mi[i * n + j] * r[off + j];
      z[off + i] = s;
      rho += s * r[off + i];
    }
  }
  rho = block_sum(rho);
  if (threadIdx.x == 0) part[blockIdx.x] = rho;
}
__device__ __forceinline__ double pcg_rho(const double* __restrict__ part, int nparts) {
  double rho = 0.0;
  for (int w = 0; w < nparts; ++w) rho += part[w];
  return rho;
}
// p = z + (rho / rho_last) p   (first iteration: p = z)
__global__ void ba_pcg_dir_kernel(int n, const double* __restrict__ scalars, const double* __restrict__ part,
                                  int nparts, int first, const double* __restrict__ z, double* __restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  p[i] = first ? z[i] : z[i] + (pcg_rho(part, nparts) / scalars[S_RHO_LAST]) * p[i];
}
// One single-workgroup kernel per PCG iteration for everything around the three streaming kernels of the
// implicit product (single GPU, no priors; the sharded / prior path keeps the separate kernels because
// all-reduces sit between the steps). A lane owns whole blocks, so every step below touches only entries
// the same lane produced -- the only synchronisations are the three workgroup sums:
//   tail of iteration k : q_b = Dc_b^2 p_b + sum of the block's J_b^T v chunk partials; pq = p.q;
//                         alpha = rho / pq; x += alpha p; r -= alpha q; Q = -x.(b + r) / 2
//   head of iteration k+1: z_b = Minv_b r_b; rho' = r.z; p = z + (rho' / rho) p
// HEAD_ONLY starts a solve: x = 0, r = b, p = z = Minv b. scalars: S_RHO = the rho iteration k used (what
// the host tests), S_RHO_LAST = rho' for the next call, S_PQ, S_Q.
template <bool HEAD_ONLY>
__global__ void __launch_bounds__(1024) ba_pcg_fused_kernel(View V, const double* __restrict__ Dc,
                                                            const double* __restrict__ Minv,
                                                            const double* __restrict__ rhs, double* __restrict__ scalars,
                                                            double* __restrict__ x, double* __restrict__ r,
                                                            double* __restrict__ z, double* __restrict__ p,
                                                            double* __restrict__ q) {
  const int bd2 = V.bd * V.bd;
  double rho = HEAD_ONLY ? 0.0 : scalars[S_RHO_LAST];
  if (!HEAD_ONLY) {
    double pq = 0.0;
    for (int b = threadIdx.x; b < V.n_blk; b += 1024) {
      const int dim = V.blk_dim[b], off = V.blk_off[b];
      for (int c = 0; c < dim; ++c) {
        double sacc = 0.0;
        for (int ch = V.blk_chunk_ptr[b]; ch < V.blk_fin_end[b]; ++ch) sacc += V.cpart[(size_t)ch * bd2 + c];
        const double d = Dc[off + c], pv = p[off + c];
        const double qv = d * d * pv + sacc;
        q[off + c] = qv;
        pq += pv * qv;
      }
    }
    pq = block_sum(pq);
    const double alpha = rho / pq;
    double Q = 0.0;
    for (int b = threadIdx.x; b < V.n_blk; b += 1024) {
      const int dim = V.blk_dim[b], off = V.blk_off[b];
      for (int c = 0; c < dim; ++c) {
        const double xn = x[off + c] + alpha * p[off + c];
        const double rn = r[off + c] - alpha * q[off + c];
        x[off + c] = xn;
        r[off + c] = rn;
        Q += -0.5 * xn * (rhs[off + c] + rn);
      }
    }
    Q = block_sum(Q);
    if (threadIdx.x == 0) {
      scalars[S_Q] = Q;
      scalars[S_PQ] = pq;
      scalars[S_RHO] = rho;
    }
  }
  double rho_new = 0.0;
  for (int b = threadIdx.x; b < V.n_blk; b += 1024) {
    const int n = V.blk_dim[b], off = V.blk_off[b];
    const double* Mi = Minv + V.blk_moff[b];
    for (int i = 0; i < n; ++i) {
      if (HEAD_ONLY) { x[off + i] = 0.0; r[off + i] = rhs[off + i]; }
    }
    for (int i = 0; i < n; ++i) {
      double sacc = 0.0;
      for (int j = 0; j < n; ++j) sacc += Mi[i * n + j] * r[off + j];
      z[off + i] = sacc;
      rho_new += sacc * r[off + i];
    }
  }
  rho_new = block_sum(rho_new);
  const double beta = HEAD_ONLY ? 0.0 : rho_new / rho;
  for (int b = threadIdx.x; b < V.n_blk; b += 1024) {
    const int n = V.blk_dim[b], off = V.blk_off[b];
    for (int i = 0; i < n; ++i) p[off + i] = HEAD_ONLY ? z[off + i] : z[off + i] + beta * p[off + i];
  }
  if (threadIdx.x == 0) {
    scalars[S_RHO_LAST] = rho_new;
    if (HEAD_ONLY) scalars[S_RHO] = rho_new;
  }
}

// ---------------------------------------------------------------------------
// Pipelined PCG (single GPU, no priors): three small multi-workgroup kernels per iteration around the three
// streaming kernels of the implicit product, the stopping test on the device, the host one iteration behind.
//   dir_k   (thread per entry)  closes iteration k-1 -- Q = sum of its partials, the stopping test of Ceres'
//           CG (zeta = k (Q_k - Q_{k-1}) / Q_k < eta), the iteration's scalars into a pinned host slot -- and, unless
//           it stopped, p = z + (rho_k / rho_{k-1}) p;
//   tail_k  (lane per block)    q_b = Dc_b^2 p_b + sum of the block's J_b^T v chunk partials, partials of p.q;
//   step_k  (lane per block)    alpha = rho / pq, x += alpha p, r -= alpha q, partials of Q = -x.(b + r) / 2,
//           z_b = Minv_b r_b, partials of rho' = r.z.
// Sums over workgroups are added by every wave that needs them in one fixed order (pcg_sum: strided lanes, then the
// xor tree): deterministic, the same bits everywhere.
// Partials alternate between two banks by iteration parity, so a kernel never reads what its own grid is writing.
// The host enqueues iteration k+1 before it looks at iteration k; once the device has set `stop` every later kernel
// (the streaming kernels through View::stop) returns at entry.
// ---------------------------------------------------------------------------
struct PcgHostSlot {
  double rho, pq, Q;
  int stop, iter;
};
struct PcgDev {
  double* part;        // [2 banks][3: pq, rho, Q][nparts]
  int nparts;
  int* stop;           // device flag
  PcgHostSlot* host;   // [2] pinned, device-visible
  double* qhist;       // [2] Q of the last two iterations
};
// (every wave of every workgroup adds the partials alike -- lane l takes the entries l, l + 64, ..., then the fixed
//  xor tree of wave_sum: one load round trip whatever nparts is, and the same bits in every wave. All 64 lanes of
//  the calling wave must be here.)
constexpr int PCGP_T = 64;  // lanes (= blocks) per workgroup of the init / tail / step kernels: 4 x the CUs of 256
__device__ __forceinline__ double pcg_sum(const double* __restrict__ part, int nparts) {
  double s = 0.0;
  for (int w = (int)(threadIdx.x & 63); w < nparts; w += 64) s += part[w];
  return wave_sum(s);
}
// x = 0, r = b, z = Minv r, partials of rho_1 into bank 1 (component loops unrolled to the block width like the step
// kernel's: the sums in the rolled order)
template <int BD>
__global__ void __launch_bounds__(PCGP_T) ba_pcgp_init_kernel(View V, PcgDev D, const double* __restrict__ Minv,
                                                           const double* __restrict__ rhs, double* __restrict__ x,
                                                           double* __restrict__ r, double* __restrict__ z) {
  double rho = 0.0;
  const int b = blockIdx.x * PCGP_T + threadIdx.x;
  if (b < V.n_blk) {
    const int n = V.blk_dim[b], off = V.blk_off[b];
    const double* Mi = Minv + V.blk_moff[b];
    double bv[BD];
#pragma unroll
    for (int i = 0; i < BD; ++i) bv[i] = i < n ? rhs[off + i] : 0.0;
    double mi[BD][BD];
    if constexpr (BD <= 8) {
#pragma unroll
      for (int i = 0; i < BD; ++i)
#pragma unroll
        for (int j = 0; j < BD; ++j) mi[i][j] = (i < n && j < n) ? Mi[i * n + j] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < BD; ++i)
      if (i < n) { x[off + i] = 0.0; r[off + i] = bv[i]; }
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      if (i < n) {
        double sacc = 0.0;
        if constexpr (BD <= 8) {
#pragma unroll
          for (int j = 0; j < BD; ++j)
            if (j < n) sacc += mi[i][j] * bv[j];
        } else {
#pragma unroll
          for (int j = 0; j < BD; ++j)
            if (j < n) sacc += Mi[i * n + j] * bv[j];
        }
        z[off + i] = sacc;
        rho += sacc * bv[i];
      }
    }
  }
  rho = block_sum(rho);
  if (threadIdx.x == 0) D.part[(1 * 3 + 1) * D.nparts + blockIdx.x] = rho;
  if (blockIdx.x == 0 && threadIdx.x == 0) { *D.stop = 0; D.qhist[0] = 0.0; D.qhist[1] = 0.0; }
}
__global__ void ba_pcgp_dir_kernel(int n, PcgDev D, int k, int max_iter, double q_tol, const double* __restrict__ z,
                                   double* __restrict__ p) {
  if (*D.stop) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double* bank_prev = D.part + (size_t)((k - 1) & 1) * 3 * D.nparts;   // what iteration k-1 consumed / produced
  const double* bank = D.part + (size_t)(k & 1) * 3 * D.nparts;
  const double rho_new = pcg_sum(bank + D.nparts, D.nparts);                // rho_k
  bool stop = false;
  double rho_prev = 0.0, pq = 0.0, Q1 = 0.0;
  if (k == 1) {
    stop = rho_new == 0.0;                                                    // zero right-hand side: nothing to solve
  } else {
    rho_prev = pcg_sum(bank_prev + D.nparts, D.nparts);                       // rho_{k-1}
    pq = pcg_sum(bank_prev, D.nparts);
    Q1 = pcg_sum(bank_prev + 2 * D.nparts, D.nparts);
    const double Q0 = D.qhist[k & 1];                                         // Q_{k-2}
    const int it = k - 1;
    if (!(rho_prev > 0.0) || !isfinite(rho_prev) || !(pq > 0.0) || !isfinite(pq)) stop = true;
    else if ((double)it * (Q1 - Q0) / Q1 < q_tol) stop = true;
    else if (it >= max_iter) stop = true;
  }
  if (i == 0) {
    PcgHostSlot h;
    h.rho = rho_prev; h.pq = pq; h.Q = Q1; h.stop = stop ? 1 : 0; h.iter = k - 1;
    D.host[(k - 1) & 1] = h;
    if (k > 1) D.qhist[(k - 1) & 1] = Q1;
    __threadfence_system();
  }
  if (stop) {
    // every workgroup decides alike from the same partials; the flag is for the kernels behind this one. No
    // workgroup may return before all have read it as 0, which the kernel boundary cannot give: workgroups that
    // start late would see 1 and skip -- harmless, they would have stopped as well.
    if (i == 0) *D.stop = 1;
    return;
  }
  if (i >= n) return;
  p[i] = k == 1 ? z[i] : z[i] + (rho_new / rho_prev) * p[i];
}
// (tail and step kernels: a lane per block, every loop over the block's components unrolled to the template width and
//  predicated on its dimension -- rolled, each trip waited for its own loads: 6 - 12 dependent round trips per lane and
//  launch; the order of every sum is the rolled loops', so the results are bit-identical)
// `reduced` (point-sharded solves): J_c^T v of every block, already summed over the block's chunks AND over the ranks
// (ba_block_vec_finalize_kernel + all-reduce) -- the tail then only adds D^2 p and takes p.q; null: this GPU's chunk
// partials are the whole sum.
template <int BD>
__global__ void __launch_bounds__(PCGP_T) ba_pcgp_tail_kernel(View V, PcgDev D, int k, const double* __restrict__ Dc,
                                                           const double* __restrict__ p, double* __restrict__ q,
                                                           const double* __restrict__ reduced) {
  if (*D.stop) return;
  const int bd2 = V.bd * V.bd;
  double pq = 0.0;
  const int b = blockIdx.x * PCGP_T + threadIdx.x;
  if (b < V.n_blk) {
    const int dim = V.blk_dim[b], off = V.blk_off[b];
    const int ch0 = V.blk_chunk_ptr[b], ch1 = V.blk_fin_end[b];
    double sacc[BD], d[BD], pv[BD];
#pragma unroll
    for (int c = 0; c < BD; ++c) {
      sacc[c] = 0.0;
      d[c] = c < dim ? Dc[off + c] : 0.0;
      pv[c] = c < dim ? p[off + c] : 0.0;
    }
    if (reduced) {
#pragma unroll
      for (int c = 0; c < BD; ++c) sacc[c] = c < dim ? reduced[off + c] : 0.0;
    } else {
      for (int ch = ch0; ch < ch1; ch += 4) {  // four chunks' rows per round trip, added in chunk order
        double row[4][BD];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int c = 0; c < BD; ++c) row[k][c] = (ch + k < ch1 && c < dim) ? V.cpart[(size_t)(ch + k) * bd2 + c] : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ch + k < ch1) {
#pragma unroll
            for (int c = 0; c < BD; ++c) sacc[c] += row[k][c];
          }
      }
    }
#pragma unroll
    for (int c = 0; c < BD; ++c) {
      if (c < dim) {
        const double qv = d[c] * d[c] * pv[c] + sacc[c];
        q[off + c] = qv;
        pq += pv[c] * qv;
      }
    }
  }
  pq = block_sum(pq);
  if (threadIdx.x == 0) D.part[(size_t)(k & 1) * 3 * D.nparts + blockIdx.x] = pq;
}
template <int BD>
__global__ void __launch_bounds__(PCGP_T) ba_pcgp_step_kernel(View V, PcgDev D, int k, const double* __restrict__ Minv,
                                                           const double* __restrict__ rhs, const double* __restrict__ p,
                                                           const double* __restrict__ q, double* __restrict__ x,
                                                           double* __restrict__ r, double* __restrict__ z) {
  if (*D.stop) return;
  double* bank = D.part + (size_t)(k & 1) * 3 * D.nparts;
  double* bank_next = D.part + (size_t)((k + 1) & 1) * 3 * D.nparts;
  const double alpha = pcg_sum(bank + D.nparts, D.nparts) / pcg_sum(bank, D.nparts);
  double Q = 0.0, rho = 0.0;
  const int b = blockIdx.x * PCGP_T + threadIdx.x;
  if (b < V.n_blk) {
    const int n = V.blk_dim[b], off = V.blk_off[b];
    const double* Mi = Minv + V.blk_moff[b];
    double xv[BD], pv[BD], rv[BD], qv[BD], bv[BD];
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      const bool ok = i < n;
      xv[i] = ok ? x[off + i] : 0.0;
      pv[i] = ok ? p[off + i] : 0.0;
      rv[i] = ok ? r[off + i] : 0.0;
      qv[i] = ok ? q[off + i] : 0.0;
      bv[i] = ok ? rhs[off + i] : 0.0;
    }
    double mi[BD][BD];
    if constexpr (BD <= 8) {
#pragma unroll
      for (int i = 0; i < BD; ++i)
#pragma unroll
        for (int j = 0; j < BD; ++j) mi[i][j] = (i < n && j < n) ? Mi[i * n + j] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      if (i < n) {
        const double xn = xv[i] + alpha * pv[i];
        const double rn = rv[i] - alpha * qv[i];
        x[off + i] = xn;
        r[off + i] = rn;
        rv[i] = rn;
        Q += -0.5 * xn * (bv[i] + rn);
      }
    }
#pragma unroll
    for (int i = 0; i < BD; ++i) {
      if (i < n) {
        double sacc = 0.0;
        if constexpr (BD <= 8) {
#pragma unroll
          for (int j = 0; j < BD; ++j)
            if (j < n) sacc += mi[i][j] * rv[j];
        } else {  // (the 16-wide tier: 256 matrix entries per lane do not fit in registers)
#pragma unroll
          for (int j = 0; j < BD; ++j)
            if (j < n) sacc += Mi[i * n + j] * rv[j];
        }
        z[off + i] = sacc;
        rho += sacc * rv[i];
      }
    }
  }
  Q = block_sum(Q);
  rho = block_sum(rho);
  if (threadIdx.x == 0) {
    bank[2 * D.nparts + blockIdx.x] = Q;
    bank_next[D.nparts + blockIdx.x] = rho;
  }
}

__global__ void __launch_bounds__(1024) ba_dot_kernel(int n, const double* __restrict__ a,
                                                      const double* __restrict__ b, double* __restrict__ out) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) v += a[i] * b[i];
  v = block_sum(v);
  if (threadIdx.x == 0) *out = v;
}
// pq = p.q ; alpha = rho / pq ; x += alpha p ; r -= alpha q ; Q = -0.5 x (b + r); publishes rho
// (for the host and as the next iteration's rho_last)
__global__ void __launch_bounds__(1024) ba_pcg_update_kernel(int n, double* __restrict__ scalars,
                                                             const double* __restrict__ part, int nparts,
                                                             const double* __restrict__ p,
                                                             const double* __restrict__ q,
                                                             const double* __restrict__ b, double* __restrict__ x,
                                                             double* __restrict__ r) {
  double pq = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) pq += p[i] * q[i];
  pq = block_sum(pq);
  const double rho = pcg_rho(part, nparts);
  double Q = 0.0;
  const double alpha = rho / pq;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double xn = x[i] + alpha * p[i];
    const double rn = r[i] - alpha * q[i];
    x[i] = xn;
    r[i] = rn;
    Q += -0.5 * xn * (b[i] + rn);
  }
  Q = block_sum(Q);
  if (threadIdx.x == 0) {
    scalars[S_Q] = Q;
    scalars[S_PQ] = pq;
    scalars[S_RHO] = rho;
    scalars[S_RHO_LAST] = rho;
  }
}
// y = a * x / s  (s may be null)
__global__ void ba_axpby_kernel(int n, double a, const double* __restrict__ x, const double* __restrict__ s,
                                double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] * (s ? s[i] : 1.0);
}

// x_plus = Plus(x, step): quaternion (x) translation, subset of intrinsics, points -- ONE launch for all parameter
// blocks (workgroups are dealt to points, poses, cameras, sensors in that order), and the step is read through a
// StepSrc, so that the vector kernels that used to prepare it are gone:
//   tf 0: step = a            tf 1: step = (-1 * a) / s   (projected-gradient test: a = scaled gradient, s = column scale)
//   tf 3: step = (-a) * s     (candidate: a = the solver's y, s = column scale)          -- the same roundings as
// two vector kernels (y = a x / s, y = a x s) applied one after the other, which this replaces.
// MAXDIFF: also max |x_plus - x| over everything -> scalars[S_GMAX] (replaces three ba_maxdiff launches).
struct StepSrc {
  const double* a;
  const double* s;
  int tf;
};
__device__ __forceinline__ double step_val(const StepSrc& S, int k) {
  const double x = S.a[k];
  return S.tf == 0 ? x : S.tf == 1 ? (-1.0 * x) / S.s[k] : (-1.0 * x) * S.s[k];
}
__device__ __forceinline__ double plus_quat_trans(const double* q, double* o, const double d[6], bool rotc, int fix) {
  for (int c = 0; c < 7; ++c) o[c] = q[c];
  const double n = rotc ? 0.0 : sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n != 0.0) {
    const double s = sin(n) / n;
    const double dx = s * d[0], dy = s * d[1], dz = s * d[2], dw = cos(n);
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    o[0] = dw * x + dx * w + dy * z - dz * y;
    o[1] = dw * y - dx * z + dy * w + dz * x;
    o[2] = dw * z + dx * y - dy * x + dz * w;
    o[3] = dw * w - dx * x - dy * y - dz * z;
  }
  int k = rotc ? 0 : 3;
  for (int c = 0; c < 3; ++c) {
    if (c == fix) continue;
    o[4 + c] += d[k++];
  }
  double m = 0.0;
  for (int c = 0; c < 7; ++c) m = fmax(m, fabs(o[c] - q[c]));
  return m;
}
template <bool MAXDIFF>
__global__ void __launch_bounds__(128) ba_apply_all_kernel(View V, StepSrc Sc, StepSrc Sp, const double* __restrict__ Pin,
                                                           double* __restrict__ P2, const double* __restrict__ Cin,
                                                           double* __restrict__ C2, const double* __restrict__ Xin,
                                                           double* __restrict__ X2, const double* __restrict__ Sin,
                                                           double* __restrict__ S2) {
  const int nb_pt = (V.n_points + 127) / 128, nb_pose = (V.n_poses + 127) / 128, nb_cam = (V.n_cams + 127) / 128;
  int blk = blockIdx.x;
  double m = 0.0;
  if (blk < nb_pt) {
    const int j = blk * 128 + threadIdx.x;
    if (j < V.n_points) {
      const int off = V.pt_off[j];
      for (int c = 0; c < 3; ++c) {
        const double xin = Xin[3 * (size_t)j + c];
        const double xo = xin + (off >= 0 ? step_val(Sp, off + c) : 0.0);
        X2[3 * (size_t)j + c] = xo;
        m = fmax(m, fabs(xo - xin));
      }
    }
  } else if ((blk -= nb_pt) < nb_pose) {
    const int i = blk * 128 + threadIdx.x;
    if (i < V.n_poses) {
      const double* q = Pin + 7 * (size_t)i;
      double* o = P2 + 7 * (size_t)i;
      const int off = V.pose_off[i];
      if (off < 0) {
        for (int c = 0; c < 7; ++c) o[c] = q[c];
      } else {
        const int pf = V.pose_fix[i];
        const bool rotc = pf >= 4;
        const int fix = pf < 0 ? -1 : ((pf & 3) == 3 ? -1 : (pf & 3));
        const int dim = V.pose_dim[i];
        double d[6];
        for (int c = 0; c < 6; ++c) d[c] = c < dim ? step_val(Sc, off + c) : 0.0;
        m = plus_quat_trans(q, o, d, rotc, fix);
      }
    }
  } else if ((blk -= nb_pose) < nb_cam) {
    const int k = blk * 128 + threadIdx.x;
    if (k < V.n_cams) {
      for (int c = 0; c < BA_CAM_STRIDE; ++c) C2[BA_CAM_STRIDE * (size_t)k + c] = Cin[BA_CAM_STRIDE * (size_t)k + c];
      const int off = V.cam_off[k];
      if (off >= 0)
        for (int d = 0; d < V.cam_dim[k]; ++d) {
          const int idx = V.cam_var[V.kd * k + d];
          const double cin = Cin[BA_CAM_STRIDE * (size_t)k + idx];
          const double co = cin + step_val(Sc, off + d);
          C2[BA_CAM_STRIDE * (size_t)k + idx] = co;
          m = fmax(m, fabs(co - cin));
        }
    }
  } else {
    blk -= nb_cam;
    const int i = blk * 128 + threadIdx.x;
    if (i < V.n_sensors) {
      const double* q = Sin + 7 * (size_t)i;
      double* o = S2 + 7 * (size_t)i;
      const int off = V.sens_off ? V.sens_off[i] : -1;
      if (off < 0) {
        for (int c = 0; c < 7; ++c) o[c] = q[c];
      } else {
        double d[6];
        for (int c = 0; c < 6; ++c) d[c] = step_val(Sc, off + c);
        m = plus_quat_trans(q, o, d, false, -1);
      }
    }
  }
  if (MAXDIFF) {
    // one atomic per workgroup, and only from a workgroup that can still raise the maximum (3 000 waves hammering one
    // address cost 37 us at BA-1; the max is order-independent, so skipping a value that is not above it changes nothing)
    __shared__ double wmax[2];
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmax(wmax[0], wmax[1]);
      const double cur = __longlong_as_double((long long)__atomic_load_n(
          reinterpret_cast<const unsigned long long*>(V.scalars + S_GMAX), __ATOMIC_RELAXED));
      if (m > cur) atomic_max_pos(V.scalars + S_GMAX, m);
    }
  }
}
__global__ void ba_renorm_sensor_quat_kernel(View V, double* __restrict__ sensors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V.n_sensors || !V.sens_off || V.sens_off[i] < 0) return;
  double* q = sensors + 7 * (size_t)i;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int c = 0; c < 4; ++c) q[c] /= n;
}
// point sharding: zero the variable points another rank owns (point index % world != rank)
__global__ void ba_keep_own_points_kernel(View V, int rank, int world, double* __restrict__ points) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V.n_points || V.pt_off[j] < 0 || j % world == rank) return;
  points[3 * (size_t)j] = points[3 * (size_t)j + 1] = points[3 * (size_t)j + 2] = 0.0;
}

__global__ void ba_renorm_quat_kernel(View V, double* __restrict__ poses) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V.n_poses || V.pose_off[i] < 0) return;
  double* q = poses + 7 * (size_t)i;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int c = 0; c < 4; ++c) q[c] /= n;
}

// ------------------------------------------------------------------------------------------
// Host-side solver
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Position priors (PosePriorBundleAdjuster, bundle_adjustment_ceres.cc:900-1038;
// AbsolutePosePositionPriorCostFunctor / AbsoluteRigPosePositionPriorCostFunctor,
// cost_functions/pose_prior.h:76-129). A prior is a 3-residual block on a pose block and, for an image
// that is not the reference sensor of its frame, on its sensor_from_rig block. It has no point block:
// it adds to the camera side of the normal equations directly. There are at most as many priors as
// images, so the kernels are small; sums run in a fixed order (one lane per target block walks its
// priors, single-workgroup tree reductions for the scalars) to keep the solve bit-reproducible.
// ------------------------------------------------------------------------------------------
struct PriorView {
  int n;                     // active priors
  const int *pose, *sens;    // [n] pose block index; sensor_from_rig index or -1
  const double *pos, *A;     // [n][3] position, [n][9] left sqrt information (row-major)
  const int *po, *so;        // [n] tangent offsets (-1: constant block)
  const int *pdim;           // [n] pose tangent width (0 when constant)
  double *r;                 // [3][n] weighted, loss-corrected residual
  double *J;                 // [3][12][n] tangent columns: pose (pdim), then sensor (6), column scaled
  double *jx;                // [3][n] J x of the current product
  int loss_type;
  double loss_scale;
  // targets: (block, prior, first column) sorted by block; tb_ptr over the distinct blocks
  int n_tblk;
  const int *tb_blk, *tb_ptr, *tg_prior, *tg_base;
};

// r0 = position + R(q_r)^T (t_r + R(q_s)^T t_s) and its ambient Jacobians (3 x 7 each, row-major)
__device__ __forceinline__ void prior_residual(const double* pos, const double* pose, const double* sens, double r0[3],
                                               double* Jpose, double* Jsens) {
  double w[3] = {pose[4], pose[5], pose[6]};
  double Rs[9], Jqs[12];
  if (sens) {
    const double qsc[4] = {-sens[0], -sens[1], -sens[2], sens[3]};
    double v[3];
    quat_rotate(qsc, sens + 4, v, Jsens ? Jqs : nullptr);
    w[0] += v[0]; w[1] += v[1]; w[2] += v[2];
    quat_to_rot(sens, Rs);
  }
  const double qc[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
  double v[3], Jq[12];
  quat_rotate(qc, w, v, Jpose ? Jq : nullptr);
  r0[0] = pos[0] + v[0]; r0[1] = pos[1] + v[1]; r0[2] = pos[2] + v[2];
  if (!Jpose) return;
  double Rr[9];
  quat_to_rot(pose, Rr);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Jpose[7 * r + c] = -Jq[4 * r + c];  // d conj(q) / d q
    Jpose[7 * r + 3] = Jq[4 * r + 3];
    for (int c = 0; c < 3; ++c) Jpose[7 * r + 4 + c] = Rr[3 * c + r];  // R_r^T
  }
  if (sens && Jsens) {
    double tmp[21];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 4; ++c) {
        const double a = Rr[r] * Jqs[c] + Rr[3 + r] * Jqs[4 + c] + Rr[6 + r] * Jqs[8 + c];
        tmp[7 * r + c] = c < 3 ? -a : a;
      }
      for (int c = 0; c < 3; ++c) tmp[7 * r + 4 + c] = Rr[r] * Rs[3 * c] + Rr[3 + r] * Rs[3 * c + 1] + Rr[6 + r] * Rs[3 * c + 2];
    }
    for (int e = 0; e < 21; ++e) Jsens[e] = tmp[e];
  }
}

// One workgroup: linearise every prior (JAC: residual + Jacobians), add the summed cost to *cost_slot
template <bool JAC>
__global__ void __launch_bounds__(256) ba_prior_linearize_kernel(View V, PriorView Q, const double* __restrict__ poses,
                                                                 const double* __restrict__ sensors,
                                                                 double* __restrict__ cost_slot) {
  __shared__ double red[256];
  double cost = 0.0;
  for (int k = threadIdx.x; k < Q.n; k += 256) {
    const int pi = Q.pose[k], si = Q.sens[k];
    const double* pose = poses + 7 * (size_t)pi;
    const double* sens = si >= 0 ? sensors + 7 * (size_t)si : nullptr;
    const int pdim = Q.pdim[k], so = Q.so[k], po = Q.po[k];
    double r0[3], Jpose[21], Jsens[21];
    prior_residual(Q.pos + 3 * (size_t)k, pose, sens, r0, JAC ? Jpose : nullptr, (JAC && so >= 0) ? Jsens : nullptr);
    const double* A = Q.A + 9 * (size_t)k;
    double r[3];
    for (int i = 0; i < 3; ++i) r[i] = A[3 * i] * r0[0] + A[3 * i + 1] * r0[1] + A[3 * i + 2] * r0[2];
    const double sq_norm = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    double rho[3];
    loss_eval(Q.loss_type, Q.loss_scale, sq_norm, rho);
    cost += 0.5 * rho[0];
    if (JAC) {
      double Jt[3][12];
      for (int i = 0; i < 3; ++i)
        for (int d = 0; d < 12; ++d) Jt[i][d] = 0.0;
      if (pdim > 0) {
        const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
        const double PJ[12] = {w, z, -y, -z, w, x, y, -x, w, -x, -y, -z};
        const int pf = V.pose_fix[pi];
        const bool rotc = pf >= 4;
        const int fix = pf < 0 ? -1 : ((pf & 3) == 3 ? -1 : (pf & 3));
        for (int i = 0; i < 3; ++i) {
          int d = 0;
          if (!rotc) {
            for (int c = 0; c < 3; ++c)
              Jt[i][c] = Jpose[7 * i] * PJ[c] + Jpose[7 * i + 1] * PJ[3 + c] + Jpose[7 * i + 2] * PJ[6 + c] +
                         Jpose[7 * i + 3] * PJ[9 + c];
            d = 3;
          }
          for (int c = 0; c < 3; ++c) {
            if (c == fix) continue;
            Jt[i][d++] = Jpose[7 * i + 4 + c];
          }
        }
      }
      if (so >= 0) {
        const double x = sens[0], y = sens[1], z = sens[2], w = sens[3];
        const double PJ[12] = {w, z, -y, -z, w, x, y, -x, w, -x, -y, -z};
        for (int i = 0; i < 3; ++i)
          for (int c = 0; c < 3; ++c) {
            Jt[i][pdim + c] = Jsens[7 * i] * PJ[c] + Jsens[7 * i + 1] * PJ[3 + c] + Jsens[7 * i + 2] * PJ[6 + c] +
                              Jsens[7 * i + 3] * PJ[9 + c];
            Jt[i][pdim + 3 + c] = Jsens[7 * i + 4 + c];
          }
      }
      const int wdt = pdim + (so >= 0 ? 6 : 0);
      double J[3][12];
      for (int i = 0; i < 3; ++i)
        for (int d = 0; d < 12; ++d)
          J[i][d] = d < wdt ? A[3 * i] * Jt[0][d] + A[3 * i + 1] * Jt[1][d] + A[3 * i + 2] * Jt[2][d] : 0.0;
      if (Q.loss_type != BA_LOSS_TRIVIAL) {  // ceres::internal::Corrector on the 3-residual block
        const double sqrt_rho1 = sqrt(rho[1]);
        double residual_scaling = sqrt_rho1, alpha_sq_norm = 0.0;
        if (!(sq_norm == 0.0 || rho[2] <= 0.0)) {
          const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(D);
          residual_scaling = sqrt_rho1 / (1.0 - alpha);
          alpha_sq_norm = alpha / sq_norm;
        }
        for (int d = 0; d < wdt; ++d) {
          const double rtj = r[0] * J[0][d] + r[1] * J[1][d] + r[2] * J[2][d];
          for (int i = 0; i < 3; ++i) J[i][d] = sqrt_rho1 * (J[i][d] - alpha_sq_norm * r[i] * rtj);
        }
        for (int i = 0; i < 3; ++i) r[i] *= residual_scaling;
      }
      for (int i = 0; i < 3; ++i) {
        Q.r[(size_t)i * Q.n + k] = r[i];
        for (int d = 0; d < 12; ++d) {
          double sc = 1.0;
          if (d < pdim) sc = V.scale_c[po + d];
          else if (d < wdt) sc = V.scale_c[so + d - pdim];
          Q.J[((size_t)i * 12 + d) * Q.n + k] = J[i][d] * sc;
        }
      }
    }
  }
  red[threadIdx.x] = cost;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *cost_slot += red[0];
}

// jx_k = J_k x (3 per prior)
__global__ void ba_prior_jx_kernel(PriorView Q, const double* __restrict__ x) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Q.n) return;
  const int po = Q.po[k], so = Q.so[k], pdim = Q.pdim[k];
  for (int i = 0; i < 3; ++i) {
    double v = 0.0;
    for (int d = 0; d < pdim; ++d) v += Q.J[((size_t)i * 12 + d) * Q.n + k] * x[po + d];
    if (so >= 0)
      for (int d = 0; d < 6; ++d) v += Q.J[((size_t)i * 12 + pdim + d) * Q.n + k] * x[so + d];
    Q.jx[(size_t)i * Q.n + k] = v;
  }
}

// Lane per target block, its priors in list order.
// MODE 0: g += J^T r, diag += column norms; MODE 1: M_b += J^T J; MODE 2: y += J^T jx
template <int MODE>
__global__ void ba_prior_accumulate_kernel(View V, PriorView Q, double* __restrict__ out, double* __restrict__ diag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Q.n_tblk) return;
  const int b = Q.tb_blk[t];
  const int off = V.blk_off[b], dim = V.blk_dim[b];
  for (int e = Q.tb_ptr[t]; e < Q.tb_ptr[t + 1]; ++e) {
    const int k = Q.tg_prior[e], base = Q.tg_base[e];
    for (int i = 0; i < 3; ++i) {
      const double ri = MODE == 0 ? Q.r[(size_t)i * Q.n + k] : (MODE == 2 ? Q.jx[(size_t)i * Q.n + k] : 0.0);
      for (int x = 0; x < dim; ++x) {
        const double jx = Q.J[((size_t)i * 12 + base + x) * Q.n + k];
        if (MODE == 0) { out[off + x] += jx * ri; diag[off + x] += jx * jx; }
        if (MODE == 2) out[off + x] += jx * ri;
        if (MODE == 1)
          for (int y = 0; y < dim; ++y) out[V.blk_moff[b] + x * dim + y] += jx * Q.J[((size_t)i * 12 + base + y) * Q.n + k];
      }
    }
  }
}

// model cost change of the priors: - (J step) . (r + J step / 2), added to *slot (one workgroup)
__global__ void __launch_bounds__(256) ba_prior_model_kernel(PriorView Q, const double* __restrict__ step,
                                                             double* __restrict__ slot) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int k = threadIdx.x; k < Q.n; k += 256) {
    const int po = Q.po[k], so = Q.so[k], pdim = Q.pdim[k];
    for (int i = 0; i < 3; ++i) {
      double m = 0.0;
      for (int d = 0; d < pdim; ++d) m += Q.J[((size_t)i * 12 + d) * Q.n + k] * step[po + d];
      if (so >= 0)
        for (int d = 0; d < 6; ++d) m += Q.J[((size_t)i * 12 + pdim + d) * Q.n + k] * step[so + d];
      acc -= m * (Q.r[(size_t)i * Q.n + k] + 0.5 * m);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *slot += red[0];
}

// ------------------------------------------------------------------------------------------
// DENSE_SCHUR tier (ceres::DENSE_SCHUR as CreateSolverOptions selects it for <= 50 images,
// bundle_adjustment_ceres.cc:203-213): S is formed column by column through the implicit operator
// (S e_i), then factored and solved by one workgroup. n <= 1024.
// ------------------------------------------------------------------------------------------
__global__ void ba_unit_vector_kernel(int n, int i, double* __restrict__ e) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) e[k] = k == i ? 1.0 : 0.0;
}

// in-place lower Cholesky of the column-major (= row-major, S is symmetric) n x n matrix S, then
// x = S^-1 b. A non-positive pivot yields NaNs: the LM loop rejects the step like a failed LLT.
__global__ void __launch_bounds__(1024) ba_dense_cholesky_solve_kernel(int n, double* __restrict__ S,
                                                                       const double* __restrict__ b,
                                                                       double* __restrict__ x) {
  const int tid = threadIdx.x, T = blockDim.x;
  // S(i, j) = S[j * n + i]: column j is what the j-th operator product wrote
  for (int k = 0; k < n; ++k) {
    if (tid == 0) S[(size_t)k * n + k] = sqrt(S[(size_t)k * n + k]);
    __syncthreads();
    const double d = S[(size_t)k * n + k];
    for (int i = k + 1 + tid; i < n; i += T) S[(size_t)k * n + i] /= d;
    __syncthreads();
    // trailing update of the lower triangle: column j, rows i >= j
    const int m = n - k - 1;
    for (int e = tid; e < m * m; e += T) {
      const int j = k + 1 + e / m, i = k + 1 + e % m;
      if (i >= j) S[(size_t)j * n + i] -= S[(size_t)k * n + i] * S[(size_t)k * n + j];
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += T) x[i] = b[i];
  __syncthreads();
  for (int k = 0; k < n; ++k) {  // L y = b
    if (tid == 0) x[k] /= S[(size_t)k * n + k];
    __syncthreads();
    const double yk = x[k];
    for (int i = k + 1 + tid; i < n; i += T) x[i] -= S[(size_t)k * n + i] * yk;
    __syncthreads();
  }
  for (int k = n - 1; k >= 0; --k) {  // L^T x = y
    if (tid == 0) x[k] /= S[(size_t)k * n + k];
    __syncthreads();
    const double xk = x[k];
    for (int i = tid; i < k; i += T) x[i] -= S[(size_t)i * n + k] * xk;
    __syncthreads();
  }
}

template <typename T>
struct Buf {
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
    BA_HIP(e_alloc);
    BA_HIP(hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)));
  }
  void upload(const std::vector<T>& h) {
    alloc(h.size());
    if (!h.empty()) BA_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  // non-owning window into another buffer (sub-vectors of one all-reduce payload)
  void alias(T* ptr, size_t count) {
    release();
    p = ptr;
    n = count;
    owns = false;
  }
  void release() {
    if (p && owns) (void)hipFree(p);
    p = nullptr;
    n = 0;
    owns = true;
  }
  bool owns = true;
  ~Buf() { release(); }
};

inline int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

// Host-side set-up loops over the observations (gathers into the device orders: random reads that one core serves at
// a cache miss a time): contiguous index ranges on up to 16 threads. fn(begin, end, thread).
template <typename F>
inline void host_parallel_for(int64_t n, F&& fn) {
  const int64_t grain = 1 << 16;
  int T = (int)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u), (n + grain - 1) / grain);
  if (T <= 1) {
    fn((int64_t)0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(T);
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T, t); });
  for (auto& x : th) x.join();
}

// sync + check after every launch; read per launch (an atomic load unless some switch is set), so that
// colmap_amd_set_switch("COLMAP_AMD_BA_DEBUG", "1") takes effect whenever it is called
static inline bool ba_debug() { return dev_switch_int("COLMAP_AMD_BA_DEBUG", 0) != 0; }
#define BA_LAUNCH(kernel, grid, block, stream, ...)                                    \
  do {                                                                                 \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                   \
    if (ba_debug()) {                                                                  \
      hipError_t e_ = hipStreamSynchronize(stream);                                    \
      if (e_ == hipSuccess) e_ = hipGetLastError();                                    \
      std::fprintf(stderr, "[ba] %s grid=%d -> %s\n", #kernel, (int)(grid).x, hipGetErrorString(e_)); \
      if (e_ != hipSuccess) throw std::runtime_error(std::string("kernel failed: ") + #kernel); \
    }                                                                                  \
  } while (0)

struct Solver {
  const ba_options& opt;
  ba_problem& prob;
  Comm& comm;
  View V{};
  hipStream_t st = nullptr;
  // topology
  Buf<int> o_sensor, sens_off;
  Buf<double> sensors, sensors2, Jsens;
  std::vector<int> h_sens_off;
  Buf<int> o_pose, o_cam, o_pt, pose_off, pose_dim, pose_fix, cam_off, cam_dim, cam_var, cam_model, pt_off,
      pt_ptr, blk_off, blk_dim, blk_kind, blk_moff, chunk_blk, chunk_beg, chunk_end, blk_chunk_ptr, blk_fin_end, heavy_blk, c2a, a2c, tile_pt;
  Buf<int4> tile_info;
  int n_heavy = 0;
  Buf<int> a_pose, a_cam, a_pt, a_sensor;  // p-order topology for the point-side linearisation pass
  Buf<double> a_xy;
  bool split_linearize = true;
  Buf<float> Jpose32, Jcam32, Jpt32;
  bool op32 = false;  // PCG operator streams the fp32 copies
  int plain_model = -1;  // >= 0: every camera has this model (SIMPLE_PINHOLE / PINHOLE / SIMPLE_RADIAL), no rig observations, trivial loss, no fp32 copies
  bool pcg_all_ranks_have_work = false;  // sharded solves: every rank has observations and chunk lists (agreed in run())
  Buf<unsigned char> solo;
  Buf<double> o_xy, poses, cams, points, poses2, cams2, points2, Jpose, Jcam, Jpt, res, res_p, scale_c, scale_p,
      scalars, gc, gp, diag_c, diag_p, Dc, Dp, Cinv, M, Minv, rhs, x, r, z, pdir, q, dp, jx, v, stepc, stepp, partials, cpart, Craw, tbuf, tmpc, Gobs, pcg_part, maxbuf;
  Buf<double> lin_sums;  // [g_c | diag_c | g_p | diag_p | E^T E]: one all-reduce per linearisation
  // image-sharded solves: (point, shared intrinsics block) incidences whose observations span ranks (ba_inc_* kernels)
  Buf<int> inc_pt, inc_blk, inc_ptr, inc_obs, inc_chunk_blk, inc_chunk_beg, inc_blk_chunk;
  Buf<double> inc_part;
  Buf<double> inc_wloc, inc_wtot;
  IncView IV{};
  IncView PV{};  // single GPU: (point, block) incidences with >= 2 observations in the block (ba_pair_cross_kernel)
  Buf<int> pv_pt, pv_blk, pv_ptr, pv_mem, pv_chunk_blk, pv_chunk_beg, pv_blk_chunk, pv_pair_blk;
  Buf<double> pv_part;
  int n_pair_blk = 0;
  // pipelined PCG (pcg_pipelined): partial sums, stop flag, Q history on the device; per-iteration scalars in pinned
  // host memory the device writes directly; events of two iterations in flight
  Buf<double> pcgp_part, pcgp_qhist;
  Buf<int> pcgp_stop;
  PcgHostSlot* pcgp_host = nullptr;      // [2], hipHostMalloc
  PcgHostSlot* pcgp_host_dev = nullptr;  // the same, as the device sees it
  hipEvent_t pcgp_ev_dir[2] = {nullptr, nullptr}, pcgp_ev_s0[2] = {nullptr, nullptr}, pcgp_ev_s1[2] = {nullptr, nullptr};
  Buf<double> Sdense;  // exact tiers: the reduced camera system, n_c x n_c
  Buf<double> chol_linv, chol_tmp;  // blocked Cholesky workspace (ba_schur_explicit.h)
  Buf<int> chol_info;
  bool dense_by_products = false;   // legacy formation (n_c operator products): image-sharded solves only
  double factor_ms = 0.0;
  // position priors
  Buf<int> pr_pose, pr_sens, pr_po, pr_so, pr_pdim, pr_tb_blk, pr_tb_ptr, pr_tg_prior, pr_tg_base;
  Buf<double> pr_pos, pr_A, pr_r, pr_J, pr_jx;
  PriorView Q{};
  bool use_dense = false;
  bool use_priors() const { return Q.n > 0 && comm.rank == 0; }  // sums are all-reduced: one rank contributes them
  int moff_total = 0;
  long long n_paired = 0;  // (observation, block kind) slots that have a partner of the same point in the block
  long long n_paired_kind[3] = {0, 0, 0};
  Buf<int> a_boff[3];
  Buf<double> Wp[3];
  int kd = 4, bd = PD;  // intrinsics tangent width / widest camera-side block of this problem
  std::vector<int> h_pose_off, h_cam_off, h_pt_off;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  hipStream_t st_chol = nullptr;  // second stream + events of the Cholesky lookahead (exact tiers)
  ba_explicit::PairLists pair_lists;  // pair-major formation of the exact tiers (built with the solve's other structures)
  hipEvent_t ev_chol_panel = nullptr, ev_chol_u2 = nullptr;

  Solver(ba_problem& p_, const ba_options& o_, Comm& c_) : opt(o_), prob(p_), comm(c_) {}
  ~Solver() {
    for (int k = 0; k < 2; ++k) {
      if (pcgp_ev_dir[k]) (void)hipEventDestroy(pcgp_ev_dir[k]);
      if (pcgp_ev_s0[k]) (void)hipEventDestroy(pcgp_ev_s0[k]);
      if (pcgp_ev_s1[k]) (void)hipEventDestroy(pcgp_ev_s1[k]);
    }
    if (pcgp_host) (void)hipHostFree(pcgp_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev2) (void)hipEventDestroy(ev2);
    if (ev3) (void)hipEventDestroy(ev3);
    if (ev_chol_panel) (void)hipEventDestroy(ev_chol_panel);
    if (ev_chol_u2) (void)hipEventDestroy(ev_chol_u2);
    if (st_chol) (void)hipStreamDestroy(st_chol);
    ba_explicit::free_pair_lists(pair_lists);
    if (st) (void)hipStreamDestroy(st);
  }

  // Before a finalize kernel: the heavy blocks' rows summed into their first row (ba_cpart_heavy_reduce_kernel).
  // `width` = the entries of a row the finalize kernel is going to read.
  void heavy_reduce(int width) {
    if (n_heavy <= 0) return;
    const int epw = width <= 16 ? 16 : 64;
    BA_LAUNCH(ba_cpart_heavy_reduce_kernel, dim3(n_heavy, (width + epw - 1) / epw), dim3(1024), st, V, width);
  }

  double scalar(int slot) {
    double v = 0.0;
    BA_HIP(hipMemcpyAsync(&v, scalars.p + slot, sizeof(double), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    return v;
  }
  // all scalar slots with ONE copy and ONE synchronisation (single-GPU solves read two results per sync point)
  void scalars_to_host(double* h) {
    BA_HIP(hipMemcpyAsync(h, scalars.p, sizeof(double) * NSCALAR, hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
  }
  double scalar_sum(int slot) {  // value summed over ranks
    comm.allreduce(scalars.p + slot, 1, st);
    return scalar(slot);
  }
  // max over ranks of a non-negative scalar through the sum transport: every rank contributes its
  // value in its own slot of a zeroed world-sized vector. Only point sharding needs it (there a
  // rank sees the gradient of its own points only; with image sharding all vectors are replicated).
  double scalar_max(int slot) {
    if (comm.world == 1 || !comm.by_point) return scalar(slot);
    BA_HIP(hipMemsetAsync(maxbuf.p, 0, sizeof(double) * comm.world, st));
    BA_HIP(hipMemcpyAsync(maxbuf.p + comm.rank, scalars.p + slot, sizeof(double), hipMemcpyDeviceToDevice, st));
    comm.allreduce(maxbuf.p, comm.world, st);
    std::vector<double> h(comm.world);
    BA_HIP(hipMemcpyAsync(h.data(), maxbuf.p, sizeof(double) * comm.world, hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    return *std::max_element(h.begin(), h.end());
  }
  void zero_scalar(int slot) { BA_HIP(hipMemsetAsync(scalars.p + slot, 0, sizeof(double), st)); }

  // Reduced program: active observations (>= 1 variable block), tangent offsets, CSR structures.
  int build(ba_result* res_out) {
    const ba_problem& p = prob;
    // COLMAP_AMD_BA_TIMING=1 (development switch): where the set-up time goes, to stderr
    const bool timing = dev_switch_int("COLMAP_AMD_BA_TIMING", 0) != 0;
    auto t_stage = std::chrono::steady_clock::now();
    auto stage = [&](const char* name) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[ba set-up] %-14s %8.2f ms\n", name, 1e3 * std::chrono::duration<double>(now - t_stage).count());
      t_stage = now;
    };
    std::vector<int> cam_nvar(p.num_cams, 0);
    std::vector<int> wide_cam_var((size_t)p.num_cams * KD_WIDE, 0), h_cam_dim(p.num_cams, 0);
    int max_nvar = 0, max_npar = 0;
    for (int k = 0; k < p.num_cams; ++k) {
      const int model = p.cam_model[k];
      if (!model_supported(model))
        throw std::runtime_error("unsupported camera model id " + std::to_string(model) +
                                 " (supported: SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV, "
                                 "OPENCV_FISHEYE, FOV, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE, SIMPLE_DIVISION, "
                                 "DIVISION, SIMPLE_FISHEYE, FISHEYE, EUCM, FULL_OPENCV, THIN_PRISM_FISHEYE, "
                                 "RAD_TAN_THIN_PRISM_FISHEYE, EQUIRECTANGULAR)");
      const int P = num_params_of(model);
      for (int j = 0; j < P; ++j)
        if (!p.cam_const[(size_t)k * BA_CAM_STRIDE + j]) wide_cam_var[(size_t)k * KD_WIDE + cam_nvar[k]++] = j;
      max_nvar = std::max(max_nvar, cam_nvar[k]);
      max_npar = std::max(max_npar, P);
    }
    // <KD, BD> of this problem (see the constants at the top of the file); a 12-parameter model takes
    // the wide tier whatever its number of variable intrinsics (only that tier evaluates 12 J_params columns)
    if (max_npar > NPAR || max_nvar > KD_MAX) kd = bd = KD_WIDE;
    else if (max_nvar <= 4) { kd = 4; bd = PD; }
    else kd = bd = KD_MAX;
    std::vector<int> h_cam_var((size_t)p.num_cams * kd, 0);
    for (int k = 0; k < p.num_cams; ++k)
      for (int d = 0; d < cam_nvar[k]; ++d) h_cam_var[(size_t)k * kd + d] = wide_cam_var[(size_t)k * KD_WIDE + d];
    std::vector<int64_t> active;
    active.reserve(p.num_obs / comm.world + 1);
    int64_t n_active_global = 0;
    std::vector<char> pose_used(p.num_poses, 0), cam_used(p.num_cams, 0), pt_used(p.num_points, 0),
        sens_used(std::max(p.num_sensors, 0) + 1, 0);
    for (int64_t o = 0; o < p.num_obs; ++o) {
      const int pi = p.obs_pose[o], ci = p.obs_cam[o], xi = p.obs_point[o];
      if (pi < 0 || pi >= p.num_poses || ci < 0 || ci >= p.num_cams || xi < 0 || xi >= p.num_points)
        throw std::runtime_error("observation index out of range");
      if (p.obs_sensor && (p.obs_sensor[o] < -1 || p.obs_sensor[o] >= p.num_sensors))
        throw std::runtime_error("observation sensor index out of range");
      const int sv = p.obs_sensor ? p.obs_sensor[o] : -1;
      const bool sens_var = sv >= 0 && p.sensor_const != nullptr && !p.sensor_const[sv];
      if (p.pose_const[pi] && cam_nvar[ci] == 0 && p.point_const[xi] && !sens_var) continue;
      ++n_active_global;
      pose_used[pi] = cam_used[ci] = pt_used[xi] = 1;  // layout = all ranks' observations
      if (sens_var) sens_used[sv] = 1;
      // image sharding (BASELINE.json: "images shard across the GPUs") or point sharding (every
      // observation of a point on one rank: the point-side quantities stay local)
      if ((comm.by_point ? xi : pi) % comm.world == comm.rank) active.push_back(o);
    }
    const int n = (int)active.size();
    stage("scan");
    // The two orders of the observations, by stable COUNTING sorts (the keys are block indices; comparison sorts of
    // 2 M ... 20 M observations were most of the 0.3 s ... 5.5 s a solve spent before its first kernel):
    // p-order: sorted by point (stable: keeps the caller's order inside a track)
    std::vector<int> h_pt_ptr(p.num_points + 1, 0);
    for (int a = 0; a < n; ++a) h_pt_ptr[p.obs_point[active[a]] + 1]++;
    for (int j = 0; j < p.num_points; ++j) h_pt_ptr[j + 1] += h_pt_ptr[j];
    {
      std::vector<int> cursor(h_pt_ptr.begin(), h_pt_ptr.end() - 1);
      std::vector<int64_t> sorted((size_t)n);
      for (int a = 0; a < n; ++a) sorted[(size_t)cursor[p.obs_point[active[a]]]++] = active[a];
      active.swap(sorted);
    }
    // c-order: p-order positions sorted by (camera, pose) -> every camera-side block is a range. Least significant
    // key first: stable by pose, then stable by camera; ties keep the p-order (what std::stable_sort on the pair gave).
    std::vector<int> h_c2a(n), h_a2c(n);
    {
      std::vector<int> by_pose((size_t)n), cnt((size_t)std::max(p.num_poses, p.num_cams) + 1, 0);
      for (int a = 0; a < n; ++a) cnt[(size_t)p.obs_pose[active[a]] + 1]++;
      for (int i = 0; i < p.num_poses; ++i) cnt[(size_t)i + 1] += cnt[i];
      for (int a = 0; a < n; ++a) by_pose[(size_t)cnt[p.obs_pose[active[a]]]++] = a;
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int a = 0; a < n; ++a) cnt[(size_t)p.obs_cam[active[a]] + 1]++;
      for (int k = 0; k < p.num_cams; ++k) cnt[(size_t)k + 1] += cnt[k];
      for (int i = 0; i < n; ++i) {
        const int a = by_pose[i];
        h_c2a[(size_t)cnt[p.obs_cam[active[a]]]++] = a;
      }
    }
    for (int c = 0; c < n; ++c) h_a2c[h_c2a[c]] = c;
    stage("orders");
    // point tiles for the LDS-staged point passes
    std::vector<int> h_tile_pt;
    {
      bool ok = true;
      int j = 0;
      h_tile_pt.push_back(0);
      while (j < p.num_points) {
        int j1 = j, obs = 0;
        while (j1 < p.num_points && j1 - j < TILE_PTS && obs + (h_pt_ptr[j1 + 1] - h_pt_ptr[j1]) <= TILE_OBS) {
          obs += h_pt_ptr[j1 + 1] - h_pt_ptr[j1];
          ++j1;
        }
        if (j1 == j) { ok = false; break; }  // a single track longer than a tile
        h_tile_pt.push_back(j1);
        j = j1;
      }
      if (!ok) h_tile_pt.assign(1, 0);
    }
    // the same topology in p-order (position a <-> c-order position h_a2c[a])
    {
      split_linearize = dev_switch_int("COLMAP_AMD_BA_SPLIT_LINEARIZE", 1) != 0;  // read per solve: tests toggle it
    }
    const bool has_sensors = p.obs_sensor != nullptr && p.num_sensors > 0 && p.sensors != nullptr;
    std::vector<int> h_a_pose, h_a_cam, h_a_pt, h_a_sensor;
    std::vector<double> h_a_xy;
    {  // (the explicit Schur formation reads it too, whatever the linearisation does)
      h_a_pose.resize(n); h_a_cam.resize(n); h_a_pt.resize(n); h_a_xy.resize((size_t)2 * n);
      if (has_sensors) h_a_sensor.resize(n);
      host_parallel_for(n, [&](int64_t a0, int64_t a1, int) {
        for (int64_t a = a0; a < a1; ++a) {
          const int64_t o = active[a];
          h_a_pose[a] = p.obs_pose[o];
          h_a_cam[a] = p.obs_cam[o];
          h_a_pt[a] = p.obs_point[o];
          h_a_xy[2 * (size_t)a] = p.obs_xy[2 * o];
          h_a_xy[2 * (size_t)a + 1] = p.obs_xy[2 * o + 1];
          if (has_sensors) h_a_sensor[a] = p.obs_sensor[o];
        }
      });
    }
    // solo flags: does another observation of the same point use the same pose / camera? (a track's entries of the
    // p-order arrays are neighbours in memory: the quadratic loop over a track stays in cache)
    std::vector<unsigned char> h_solo(n, 0);
    {
      long long paired_t[16][4] = {};   // per thread: n_paired, n_paired_kind[0..2]
      host_parallel_for(p.num_points, [&](int64_t j0, int64_t j1, int t) {
        long long* acc = paired_t[t];
        for (int64_t j = j0; j < j1; ++j)
          for (int a = h_pt_ptr[j]; a < h_pt_ptr[j + 1]; ++a) {
            int same_pose = 0, same_cam = 0, same_sens = 0;
            const int sa = has_sensors ? h_a_sensor[a] : (p.obs_sensor ? p.obs_sensor[active[a]] : -1);
            const int pose_a = h_a_pose[a], cam_a = h_a_cam[a];
            for (int a2 = h_pt_ptr[j]; a2 < h_pt_ptr[j + 1]; ++a2) {
              same_pose += h_a_pose[a2] == pose_a;
              same_cam += h_a_cam[a2] == cam_a;
              same_sens += sa >= 0 && (has_sensors ? h_a_sensor[a2] : p.obs_sensor[active[a2]]) == sa;
            }
            h_solo[h_a2c[a]] = (unsigned char)((same_pose == 1 ? 1 : 0) | (same_cam == 1 ? 2 : 0) | (same_sens <= 1 ? 4 : 0));
            const int k0 = same_pose != 1 && !p.pose_const[pose_a];
            const int k1 = same_cam != 1 && cam_nvar[cam_a] > 0;
            const int k2 = same_sens > 1 && sens_used[sa];
            acc[0] += k0 + k1 + k2;
            acc[1] += k0;
            acc[2] += k1;
            acc[3] += k2;
          }
      });
      for (int t = 0; t < 16; ++t) {
        n_paired += paired_t[t][0];
        for (int k = 0; k < 3; ++k) n_paired_kind[k] += paired_t[t][1 + k];
      }
    }
    std::vector<int> h_o_pose(n), h_o_cam(n), h_o_pt(n), h_o_sensor;
    if (has_sensors) h_o_sensor.resize(n);
    std::vector<double> h_xy((size_t)2 * n);
    host_parallel_for(n, [&](int64_t c0, int64_t c1, int) {
      for (int64_t c = c0; c < c1; ++c) {
        const int a = h_c2a[c];   // c-order from the p-order copies: one indirection, 4-byte indices
        if (has_sensors) h_o_sensor[c] = h_a_sensor[a];
        h_o_pose[c] = h_a_pose[a];
        h_o_cam[c] = h_a_cam[a];
        h_o_pt[c] = h_a_pt[a];
        h_xy[2 * (size_t)c] = h_a_xy[2 * (size_t)a];
        h_xy[2 * (size_t)c + 1] = h_a_xy[2 * (size_t)a + 1];
      }
    });
    stage("topology");
    // tangent layout: pose blocks, then intrinsics blocks (camera side); points
    h_pose_off.assign(p.num_poses, -1);
    h_cam_off.assign(p.num_cams, -1);
    h_pt_off.assign(p.num_points, -1);
    std::vector<int> h_pose_dim(p.num_poses, 0), h_pose_fix(p.num_poses, -1);
    std::vector<int> h_blk_off, h_blk_dim, h_blk_kind, h_blk_moff;
    std::vector<int> blk_of_pose(p.num_poses, -1), blk_of_cam(p.num_cams, -1);
    int off = 0, moff = 0;
    for (int i = 0; i < p.num_poses; ++i) {
      if (p.pose_const[i] || !pose_used[i]) continue;
      const int pf = p.pose_fixed_t[i];
      if (pf < -1 || pf > 7) throw std::runtime_error("pose_fixed_t out of range");
      h_pose_fix[i] = pf;
      h_pose_dim[i] = (pf >= 4 ? 0 : 3) + ((pf >= 0 && (pf & 3) != 3) ? 2 : 3);
      h_pose_off[i] = off;
      blk_of_pose[i] = (int)h_blk_off.size();
      h_blk_off.push_back(off); h_blk_dim.push_back(h_pose_dim[i]); h_blk_kind.push_back(0);
      h_blk_moff.push_back(moff);
      off += h_pose_dim[i];
      moff += h_pose_dim[i] * h_pose_dim[i];
    }
    for (int k = 0; k < p.num_cams; ++k) {
      if (cam_nvar[k] == 0 || !cam_used[k]) continue;
      h_cam_dim[k] = cam_nvar[k];
      h_cam_off[k] = off;
      blk_of_cam[k] = (int)h_blk_off.size();
      h_blk_off.push_back(off); h_blk_dim.push_back(cam_nvar[k]); h_blk_kind.push_back(1);
      h_blk_moff.push_back(moff);
      off += cam_nvar[k];
      moff += cam_nvar[k] * cam_nvar[k];
    }
    // variable sensor_from_rig blocks (RigReprojErrorCostFunctor's cam_from_rig parameter block,
    // bundle_adjustment_ceres.cc:804-812): full 6-dimensional pose tangent, block kind 2
    h_sens_off.assign(std::max(p.num_sensors, 0), -1);
    std::vector<int> blk_of_sens(std::max(p.num_sensors, 0), -1);
    int n_var_sensors = 0;
    for (int sidx = 0; sidx < p.num_sensors; ++sidx) {
      if (!sens_used[sidx]) continue;
      h_sens_off[sidx] = off;
      blk_of_sens[sidx] = (int)h_blk_off.size();
      h_blk_off.push_back(off); h_blk_dim.push_back(6); h_blk_kind.push_back(2);
      h_blk_moff.push_back(moff);
      off += 6;
      moff += 36;
      ++n_var_sensors;
    }
    const int n_c = off;
    moff_total = moff;
    int poff = 0;
    for (int j = 0; j < p.num_points; ++j) {
      if (p.point_const[j] || !pt_used[j]) continue;
      h_pt_off[j] = poff;
      poff += 3;
    }
    // per-block c-order runs, split into chunks. A camera's observations are one run (c-order is
    // sorted by camera first); a pose seen through several cameras (a rig frame) owns one run per
    // camera. Every chunk is a contiguous range of one block.
    const int n_blk = (int)h_blk_off.size();
    std::vector<std::vector<std::pair<int, int>>> runs(n_blk);
    auto add_run = [&](int b, int c) {
      if (b < 0) return;
      auto& r = runs[b];
      if (!r.empty() && r.back().second == c) r.back().second = c + 1;
      else r.emplace_back(c, c + 1);
    };
    for (int c = 0; c < n; ++c) {
      add_run(blk_of_pose[h_o_pose[c]], c);
      add_run(blk_of_cam[h_o_cam[c]], c);
      if (has_sensors && h_o_sensor[c] >= 0) add_run(blk_of_sens[h_o_sensor[c]], c);
    }
    const int CHUNK = chunk_size() & ~1;  // even: the MFMA Gram kernel consumes observation pairs
    std::vector<int> h_chunk_blk, h_chunk_beg, h_chunk_end, h_blk_chunk_ptr(n_blk + 1, 0);
    for (int b = 0; b < n_blk; ++b) {
      h_blk_chunk_ptr[b] = (int)h_chunk_blk.size();
      for (const auto& run : runs[b])
        for (int s = run.first; s < run.second; s += CHUNK) {
          h_chunk_blk.push_back(b);
          h_chunk_beg.push_back(s);
          h_chunk_end.push_back(std::min(s + CHUNK, run.second));
        }
    }
    h_blk_chunk_ptr[n_blk] = (int)h_chunk_blk.size();
    std::vector<int> h_blk_fin_end(std::max(n_blk, 1), 0), h_heavy;
    const int heavy_chunks = std::max(dev_switch_int("COLMAP_AMD_BA_HEAVY_CHUNKS", kHeavyChunks), 1);
    for (int b = 0; b < n_blk; ++b) {
      const bool heavy = h_blk_chunk_ptr[b + 1] - h_blk_chunk_ptr[b] > heavy_chunks;
      h_blk_fin_end[b] = heavy ? h_blk_chunk_ptr[b] + 1 : h_blk_chunk_ptr[b + 1];
      if (heavy) h_heavy.push_back(b);
    }

    // position priors whose pose or sensor_from_rig block is variable
    {
      std::vector<int> q_pose, q_sens, q_po, q_so, q_pdim;
      std::vector<double> q_pos, q_A;
      std::vector<std::array<int, 3>> targets;  // block, prior, first column
      if (p.num_priors < 0) throw std::runtime_error("num_priors < 0");
      for (int k = 0; k < p.num_priors; ++k) {
        const int pi = p.prior_pose[k];
        const int si = p.prior_sensor ? p.prior_sensor[k] : -1;
        if (pi < 0 || pi >= p.num_poses || si < -1 || si >= p.num_sensors) throw std::runtime_error("prior index out of range");
        const int po = h_pose_off[pi], so = si >= 0 ? h_sens_off[si] : -1;
        if (po < 0 && so < 0) continue;
        const int kk = (int)q_pose.size();
        const int pdim = po >= 0 ? h_pose_dim[pi] : 0;
        q_pose.push_back(pi); q_sens.push_back(si); q_po.push_back(po); q_so.push_back(so); q_pdim.push_back(pdim);
        q_pos.insert(q_pos.end(), p.prior_position + 3 * (size_t)k, p.prior_position + 3 * (size_t)k + 3);
        q_A.insert(q_A.end(), p.prior_sqrt_info + 9 * (size_t)k, p.prior_sqrt_info + 9 * (size_t)k + 9);
        if (po >= 0) targets.push_back({blk_of_pose[pi], kk, 0});
        if (so >= 0) targets.push_back({blk_of_sens[si], kk, pdim});
      }
      Q = PriorView{};
      Q.n = (int)q_pose.size();
      if (Q.n > 0) {
        if (p.prior_loss_type < BA_LOSS_TRIVIAL || p.prior_loss_type > BA_LOSS_HUBER || !(p.prior_loss_scale > 0.0))
          throw std::runtime_error("prior loss type / scale");
        std::stable_sort(targets.begin(), targets.end(), [](const std::array<int, 3>& a, const std::array<int, 3>& b) { return a[0] < b[0]; });
        std::vector<int> tb_blk, tb_ptr, tg_prior, tg_base;
        for (size_t e = 0; e < targets.size(); ++e) {
          if (e == 0 || targets[e][0] != targets[e - 1][0]) { tb_blk.push_back(targets[e][0]); tb_ptr.push_back((int)e); }
          tg_prior.push_back(targets[e][1]);
          tg_base.push_back(targets[e][2]);
        }
        tb_ptr.push_back((int)targets.size());
        pr_pose.upload(q_pose); pr_sens.upload(q_sens); pr_po.upload(q_po); pr_so.upload(q_so); pr_pdim.upload(q_pdim);
        pr_pos.upload(q_pos); pr_A.upload(q_A);
        pr_tb_blk.upload(tb_blk); pr_tb_ptr.upload(tb_ptr); pr_tg_prior.upload(tg_prior); pr_tg_base.upload(tg_base);
        pr_r.alloc(3 * (size_t)Q.n); pr_J.alloc(36 * (size_t)Q.n); pr_jx.alloc(3 * (size_t)Q.n);
        Q.pose = pr_pose.p; Q.sens = pr_sens.p; Q.pos = pr_pos.p; Q.A = pr_A.p; Q.po = pr_po.p; Q.so = pr_so.p;
        Q.pdim = pr_pdim.p; Q.r = pr_r.p; Q.J = pr_J.p; Q.jx = pr_jx.p;
        Q.loss_type = p.prior_loss_type; Q.loss_scale = p.prior_loss_scale;
        Q.n_tblk = (int)tb_blk.size();
        Q.tb_blk = pr_tb_blk.p; Q.tb_ptr = pr_tb_ptr.p; Q.tg_prior = pr_tg_prior.p; Q.tg_base = pr_tg_base.p;
      }
    }
    res_out->num_residuals = (int32_t)(2 * n_active_global) + 3 * Q.n;
    res_out->num_effective_parameters = n_c + poff;
    if (n_var_sensors > 0 && comm.world > 1)
      throw std::runtime_error("refine_sensor_from_rig is not supported by the sharded solve");
    if (n_active_global == 0) return 0;
    if (n == 0)
      throw std::runtime_error("rank " + std::to_string(comm.rank) + " holds no observation: use fewer ranks "
                               "than images");

    stage("blocks+chunks");
    // upload
    o_pose.upload(h_o_pose); o_cam.upload(h_o_cam); o_pt.upload(h_o_pt); o_xy.upload(h_xy);
    if (has_sensors) {
      o_sensor.upload(h_o_sensor);
      sensors.upload(std::vector<double>(p.sensors, p.sensors + (size_t)7 * p.num_sensors));
      sensors2.alloc(sensors.n);
      BA_HIP(hipMemcpy(sensors2.p, sensors.p, sizeof(double) * sensors.n, hipMemcpyDeviceToDevice));
      if (n_var_sensors > 0) sens_off.upload(h_sens_off);
    }
    pose_off.upload(h_pose_off); pose_dim.upload(h_pose_dim); pose_fix.upload(h_pose_fix);
    cam_off.upload(h_cam_off); cam_dim.upload(h_cam_dim); cam_var.upload(h_cam_var);
    cam_model.upload(std::vector<int>(p.cam_model, p.cam_model + p.num_cams));
    pt_off.upload(h_pt_off); pt_ptr.upload(h_pt_ptr);
    blk_off.upload(h_blk_off); blk_dim.upload(h_blk_dim); blk_kind.upload(h_blk_kind); blk_moff.upload(h_blk_moff);
    // Image sharding: (point, intrinsics block) incidences whose observations sit on more than one rank -- the
    // pairs the local Schur-Jacobi terms cannot see (ba_inc_* kernels). Every rank walks the whole problem, so all
    // ranks build the same list in the same order; a rank's own observations of an incidence go into a CSR list.
    IV = IncView{};
    if (comm.world > 1 && !comm.by_point) {
      // global pass: per (point, camera) the set of ranks that hold an observation of it
      std::vector<std::pair<long long, int>> keys;  // (point * num_cams + cam, rank)
      for (int64_t o = 0; o < p.num_obs; ++o) {
        const int pi = p.obs_pose[o], ci = p.obs_cam[o], xi = p.obs_point[o];
        const int sv = p.obs_sensor ? p.obs_sensor[o] : -1;
        const bool sens_var = sv >= 0 && p.sensor_const != nullptr && !p.sensor_const[sv];
        if (p.pose_const[pi] && cam_nvar[ci] == 0 && p.point_const[xi] && !sens_var) continue;  // not active
        if (cam_nvar[ci] == 0 || p.point_const[xi]) continue;                                     // no coupling through this block
        keys.emplace_back((long long)xi * p.num_cams + ci, pi % comm.world);
      }
      std::sort(keys.begin(), keys.end());
      keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
      std::vector<long long> spanning;
      for (size_t k = 0; k + 1 < keys.size(); ++k)
        if (keys[k].first == keys[k + 1].first && (spanning.empty() || spanning.back() != keys[k].first))
          spanning.push_back(keys[k].first);
      if (!spanning.empty()) {
        const int ni = (int)spanning.size();
        std::vector<int> h_pt(ni), h_blk(ni), h_ptr(ni + 1, 0), h_obs;
        // device order of the incidences: by block, then by (point, camera) key -- a block's run is contiguous, so
        // its terms are summed in one fixed order (ba_inc_correct_kernel); `pos` = key-order index -> device index
        std::vector<int> by_blk(ni), pos(ni);
        std::iota(by_blk.begin(), by_blk.end(), 0);
        std::stable_sort(by_blk.begin(), by_blk.end(), [&](int a, int b) {
          return blk_of_cam[(int)(spanning[a] % p.num_cams)] < blk_of_cam[(int)(spanning[b] % p.num_cams)];
        });
        for (int i = 0; i < ni; ++i) {
          pos[by_blk[i]] = i;
          h_pt[i] = (int)(spanning[by_blk[i]] / p.num_cams);
          h_blk[i] = blk_of_cam[(int)(spanning[by_blk[i]] % p.num_cams)];
        }
        // this rank's observations, c-order index c, by incidence
        std::vector<std::pair<int, int>> mine;  // (incidence, c)
        for (int c = 0; c < n; ++c) {
          const long long key = (long long)h_o_pt[c] * p.num_cams + h_o_cam[c];
          const auto it = std::lower_bound(spanning.begin(), spanning.end(), key);
          if (it != spanning.end() && *it == key) mine.emplace_back(pos[(int)(it - spanning.begin())], c);
        }
        std::sort(mine.begin(), mine.end());
        for (const auto& m : mine) h_ptr[m.first + 1]++;
        for (int i = 0; i < ni; ++i) h_ptr[i + 1] += h_ptr[i];
        h_obs.reserve(mine.size());
        for (const auto& m : mine) h_obs.push_back(m.second);
        if (h_obs.empty()) h_obs.push_back(0);
        // chunks of <= kIncChunk incidences of one block, and every block's range of chunks
        const int nblk = (int)h_blk_off.size();
        std::vector<int> h_cblk, h_cbeg, h_bchunk(nblk + 1, 0);
        for (int i = 0; i < ni;) {
          int e = i;
          while (e < ni && h_blk[e] == h_blk[i] && e - i < kIncChunk) ++e;
          h_cblk.push_back(h_blk[i]);
          h_cbeg.push_back(i);
          h_bchunk[h_blk[i] + 1]++;
          i = e;
        }
        h_cbeg.push_back(ni);
        for (int b = 0; b < nblk; ++b) h_bchunk[b + 1] += h_bchunk[b];
        inc_chunk_blk.upload(h_cblk); inc_chunk_beg.upload(h_cbeg); inc_blk_chunk.upload(h_bchunk);
        inc_part.alloc(h_cblk.size() * (size_t)KD_WIDE * KD_WIDE);
        IV.n_chunks = (int)h_cblk.size(); IV.chunk_blk = inc_chunk_blk.p; IV.chunk_beg = inc_chunk_beg.p;
        IV.blk_chunk = inc_blk_chunk.p; IV.part = inc_part.p;
        inc_pt.upload(h_pt); inc_blk.upload(h_blk); inc_ptr.upload(h_ptr); inc_obs.upload(h_obs);
        inc_wloc.alloc((size_t)ni * KD_WIDE * 3); inc_wtot.alloc((size_t)ni * KD_WIDE * 3);
        IV.n = ni; IV.pt = inc_pt.p; IV.blk = inc_blk.p; IV.ptr = inc_ptr.p; IV.obs = inc_obs.p;
      }
    }
    // pairs of observations of one point inside one block: p-order block offsets per kind and room for the W's
    // (ba_obs_w_kernel / ba_block_schur_cross_kernel), only for the kinds that have such pairs
    for (int kind = 0; kind < 3; ++kind) {
      V.a_boff[kind] = nullptr; V.Wp[kind] = nullptr;
      V.wdim[kind] = kind == 1 ? kd : 6;
      if (n_paired_kind[kind] <= 0) continue;
      std::vector<int> h_ab(n, -1);
      for (int a = 0; a < n; ++a) {
        if (kind == 0) h_ab[a] = h_pose_off[h_a_pose[a]];
        else if (kind == 1) h_ab[a] = h_cam_off[h_a_cam[a]];
        else h_ab[a] = (has_sensors && h_a_sensor[a] >= 0) ? h_sens_off[h_a_sensor[a]] : -1;
      }
      a_boff[kind].upload(h_ab);
      Wp[kind].alloc((size_t)n * V.wdim[kind] * 3);
      V.a_boff[kind] = a_boff[kind].p; V.Wp[kind] = Wp[kind].p;
    }
    // single GPU: the pair incidences themselves, sorted by block (ba_pair_cross_kernel); a sharded solve keeps the
    // per-observation kernel for its local pairs (the cross-rank ones are IV's)
    PV = IncView{};
    n_pair_blk = 0;
    if (n_paired > 0 && comm.world == 1 && dev_switch_int("COLMAP_AMD_BA_PAIR_INCIDENCES", 1) != 0) {
      std::vector<int> blk_of_off(std::max(off, 0) + 1, -1);  // tangent offset -> block
      for (int b = 0; b < n_blk; ++b) blk_of_off[h_blk_off[b]] = b;
      struct Inc { int blk, pt, first, count; };
      std::vector<Inc> incs;
      std::vector<int> members;  // p-order observation indices, grouped per incidence
      std::vector<std::pair<int, int>> grp;  // (block offset, a) of one point and kind
      for (int kind = 0; kind < 3; ++kind) {
        if (n_paired_kind[kind] <= 0) continue;
        std::vector<int> h_ab(n);
        for (int a = 0; a < n; ++a) {
          if (kind == 0) h_ab[a] = h_pose_off[h_a_pose[a]];
          else if (kind == 1) h_ab[a] = h_cam_off[h_a_cam[a]];
          else h_ab[a] = (has_sensors && h_a_sensor[a] >= 0) ? h_sens_off[h_a_sensor[a]] : -1;
        }
        for (int j = 0; j < p.num_points; ++j) {
          if (h_pt_off[j] < 0) continue;  // a constant point has no C^-1: its observations do not couple
          grp.clear();
          for (int a = h_pt_ptr[j]; a < h_pt_ptr[j + 1]; ++a)
            if (h_ab[a] >= 0) grp.emplace_back(h_ab[a], a);
          std::sort(grp.begin(), grp.end());
          for (size_t i = 0; i < grp.size();) {
            size_t e = i;
            while (e < grp.size() && grp[e].first == grp[i].first) ++e;
            if (e - i >= 2) {
              incs.push_back({blk_of_off[grp[i].first], j, (int)members.size(), (int)(e - i)});
              for (size_t k = i; k < e; ++k) members.push_back(grp[k].second);
            }
            i = e;
          }
        }
      }
      if (!incs.empty()) {
        std::stable_sort(incs.begin(), incs.end(), [](const Inc& a, const Inc& b) { return a.blk < b.blk; });
        const int ni = (int)incs.size();
        std::vector<int> h_pt(ni), h_blk(ni), h_ptr(ni + 1, 0), h_mem;
        h_mem.reserve(members.size());
        for (int i = 0; i < ni; ++i) {
          h_pt[i] = incs[i].pt; h_blk[i] = incs[i].blk;
          h_mem.insert(h_mem.end(), members.begin() + incs[i].first, members.begin() + incs[i].first + incs[i].count);
          h_ptr[i + 1] = (int)h_mem.size();
        }
        std::vector<int> h_cblk, h_cbeg, h_bchunk(n_blk + 1, 0), h_pblk;
        for (int i = 0; i < ni;) {
          int e = i;
          while (e < ni && h_blk[e] == h_blk[i] && e - i < kPairChunk) ++e;
          if (h_pblk.empty() || h_pblk.back() != h_blk[i]) h_pblk.push_back(h_blk[i]);
          h_cblk.push_back(h_blk[i]); h_cbeg.push_back(i);
          h_bchunk[h_blk[i] + 1]++;
          i = e;
        }
        h_cbeg.push_back(ni);
        for (int b = 0; b < n_blk; ++b) h_bchunk[b + 1] += h_bchunk[b];
        pv_pt.upload(h_pt); pv_blk.upload(h_blk); pv_ptr.upload(h_ptr); pv_mem.upload(h_mem);
        pv_chunk_blk.upload(h_cblk); pv_chunk_beg.upload(h_cbeg); pv_blk_chunk.upload(h_bchunk); pv_pair_blk.upload(h_pblk);
        pv_part.alloc(h_cblk.size() * (size_t)KD_WIDE * KD_WIDE);
        PV.n = ni; PV.pt = pv_pt.p; PV.blk = pv_blk.p; PV.ptr = pv_ptr.p; PV.obs = pv_mem.p;
        PV.n_chunks = (int)h_cblk.size(); PV.chunk_blk = pv_chunk_blk.p; PV.chunk_beg = pv_chunk_beg.p;
        PV.blk_chunk = pv_blk_chunk.p; PV.part = pv_part.p;
        n_pair_blk = (int)h_pblk.size();
      }
    }
    chunk_blk.upload(h_chunk_blk); chunk_beg.upload(h_chunk_beg); chunk_end.upload(h_chunk_end);
    c2a.upload(h_c2a); a2c.upload(h_a2c); solo.upload(h_solo); tile_pt.upload(h_tile_pt);
    {
      std::vector<int4> h_tile_info(std::max<size_t>(h_tile_pt.size(), 2) - 1, make_int4(0, 0, 0, 0));
      for (size_t t = 0; t + 1 < h_tile_pt.size(); ++t) {
        const int q0 = h_tile_pt[t], q1 = h_tile_pt[t + 1];
        h_tile_info[t] = make_int4(q0, q1, h_pt_ptr[q0], h_pt_ptr[q1] - h_pt_ptr[q0]);
      }
      tile_info.upload(h_tile_info);
    }
    a_pose.upload(h_a_pose); a_cam.upload(h_a_cam); a_pt.upload(h_a_pt); a_xy.upload(h_a_xy);
    if (has_sensors) a_sensor.upload(h_a_sensor);
    V.a_pose = a_pose.p; V.a_cam = a_cam.p; V.a_pt = a_pt.p; V.a_xy = a_xy.p;
    V.a_sensor = has_sensors ? a_sensor.p : nullptr;
    V.n_tiles = (int)h_tile_pt.size() - 1;
    blk_chunk_ptr.upload(h_blk_chunk_ptr);
    blk_fin_end.upload(h_blk_fin_end);
    n_heavy = (int)h_heavy.size();
    if (h_heavy.empty()) h_heavy.push_back(0);
    heavy_blk.upload(h_heavy);
    cpart.alloc((size_t)h_chunk_blk.size() * bd * bd);
    poses.upload(std::vector<double>(p.poses, p.poses + 7 * (size_t)p.num_poses));
    cams.upload(std::vector<double>(p.cams, p.cams + BA_CAM_STRIDE * (size_t)p.num_cams));
    points.upload(std::vector<double>(p.points, p.points + 3 * (size_t)p.num_points));
    poses2.alloc(poses.n); cams2.alloc(cams.n); points2.alloc(points.n);
    const size_t N = (size_t)n;
    Jpose.alloc(2 * PD * N); Jcam.alloc(2 * (size_t)kd * N); Jpt.alloc(6 * N); res.alloc(2 * N); res_p.alloc(2 * N);
    jx.alloc(2 * N); v.alloc(2 * N); Gobs.alloc(4 * N);
    if (n_var_sensors > 0) Jsens.alloc(12 * N);
    {
      const bool op32_env = dev_switch_int("COLMAP_AMD_BA_OPERATOR_F32", 0) != 0;
      op32 = (op32_env || opt.operator_precision == BA_OPERATOR_F32) && n_var_sensors == 0 && V.n_tiles > 0 && comm.world == 1;
      if (op32) { Jpose32.alloc(2 * PD * N); Jcam32.alloc(2 * (size_t)kd * N); Jpt32.alloc(6 * N); }
      V.Jpose32 = op32 ? Jpose32.p : nullptr;
      V.Jcam32 = op32 ? Jcam32.p : nullptr;
      V.Jpt32 = op32 ? Jpt32.p : nullptr;
    }
    {
      plain_model = -1;
      const bool plain_ok = dev_switch_int("COLMAP_AMD_BA_PLAIN_LINEARIZE", 1) != 0 && !op32 && !has_sensors &&
                            opt.loss_type == BA_LOSS_TRIVIAL && p.num_cams > 0;
      if (plain_ok) {
        const int m0 = p.cam_model[0];
        bool same = m0 == BA_SIMPLE_PINHOLE || m0 == BA_PINHOLE || m0 == BA_SIMPLE_RADIAL;
        for (int k = 1; same && k < p.num_cams; ++k) same = p.cam_model[k] == m0;
        if (same) plain_model = m0;
      }
    }
    // gradient, column norms and E^T E of a linearisation are summed over the ranks of a sharded solve in ONE
    // all-reduce: they live back to back in `lin_sums` [g_c | diag_c | g_p | diag_p | E^T E] (point sharding
    // reduces the camera-side prefix only)
    lin_sums.alloc(2 * (size_t)n_c + 2 * (size_t)poff + 6 * (size_t)p.num_points);
    gc.alias(lin_sums.p, n_c); diag_c.alias(lin_sums.p + n_c, n_c);
    gp.alias(lin_sums.p + 2 * (size_t)n_c, poff); diag_p.alias(lin_sums.p + 2 * (size_t)n_c + poff, poff);
    Craw.alias(lin_sums.p + 2 * (size_t)n_c + 2 * (size_t)poff, 6 * (size_t)p.num_points);
    scale_c.alloc(n_c); scale_p.alloc(poff);
    Dc.alloc(n_c); Dp.alloc(poff); rhs.alloc(n_c); x.alloc(n_c); r.alloc(n_c); z.alloc(n_c); pdir.alloc(n_c);
    q.alloc(n_c); dp.alloc(poff); stepc.alloc(n_c); stepp.alloc(poff);
    Cinv.alloc(9 * (size_t)p.num_points); M.alloc(moff); Minv.alloc(moff);
    tbuf.alloc(poff); tmpc.alloc(n_c);
    scalars.alloc(NSCALAR);
    pcg_part.alloc((size_t)grid_for(n_blk, 256) + 1);
    partials.alloc((size_t)std::max(grid_for(n, 256), std::max(p.num_points, 1)) + 1);  // one slot per linearise workgroup / point tile

    V.n_obs = n; V.n_poses = p.num_poses; V.n_cams = p.num_cams; V.n_points = p.num_points;
    V.n_c = n_c; V.n_p = poff; V.n_blk = n_blk; V.n_chunks = (int)h_chunk_blk.size();
    V.poses = poses.p; V.cams = cams.p; V.points = points.p;
    V.o_pose = o_pose.p; V.o_cam = o_cam.p; V.o_pt = o_pt.p; V.o_xy = o_xy.p;
    V.kd = kd;
    V.bd = bd;
    V.o_sensor = has_sensors ? o_sensor.p : nullptr;
    V.sensors = has_sensors ? sensors.p : nullptr;
    V.sens_off = n_var_sensors > 0 ? sens_off.p : nullptr;
    V.Jsens = n_var_sensors > 0 ? Jsens.p : nullptr;
    V.n_sensors = has_sensors ? p.num_sensors : 0;
    if (opt.loss_type < BA_LOSS_TRIVIAL || opt.loss_type > BA_LOSS_HUBER)
      throw std::runtime_error("unknown loss_type " + std::to_string(opt.loss_type));
    if (opt.loss_type != BA_LOSS_TRIVIAL && !(opt.loss_scale > 0.0))
      throw std::runtime_error("loss_scale must be positive");
    V.loss_type = opt.loss_type;
    V.loss_scale = opt.loss_scale;
    V.pose_off = pose_off.p; V.pose_dim = pose_dim.p; V.pose_fix = pose_fix.p;
    V.cam_off = cam_off.p; V.cam_dim = cam_dim.p; V.cam_var = cam_var.p; V.cam_model = cam_model.p;
    V.pt_off = pt_off.p; V.pt_ptr = pt_ptr.p;
    V.blk_off = blk_off.p; V.blk_dim = blk_dim.p; V.blk_kind = blk_kind.p; V.blk_moff = blk_moff.p;
    V.chunk_blk = chunk_blk.p; V.chunk_beg = chunk_beg.p; V.chunk_end = chunk_end.p;
    V.c2a = c2a.p; V.a2c = a2c.p; V.solo = solo.p; V.tile_pt = tile_pt.p; V.tile_info = tile_info.p;
    V.blk_chunk_ptr = blk_chunk_ptr.p; V.cpart = cpart.p;
    V.blk_fin_end = blk_fin_end.p; V.heavy_blk = heavy_blk.p; V.n_heavy = n_heavy;
    V.Jpose = Jpose.p; V.Jcam = Jcam.p; V.Jpt = Jpt.p; V.res = res.p; V.res_p = res_p.p;
    V.scale_c = scale_c.p; V.scale_p = scale_p.p; V.scalars = scalars.p;
    maxbuf.alloc(std::max(comm.world, 1));
    stage("upload+alloc");
    // uploads / memsets above ran on the NULL stream, the solve runs on a non-blocking stream
    BA_HIP(hipDeviceSynchronize());
    return (int)std::min<int64_t>(n_active_global, 1 << 30);
  }

  void launch_linearize(bool jac, const double* P, const double* Cm, const double* X, const double* Sn, int slot) {
    const int g = grid_for(V.n_obs, 256);
    if (split_linearize && kd == 4 && plain_model >= 0) {
      // one camera model, no rig observations, trivial loss, no fp32 copies: the PLAIN instantiations (same bits)
      auto launch = [&](auto tag) {
        constexpr int M = decltype(tag)::value;
        if (jac) {
          BA_LAUNCH((ba_linearize_kernel<true, 4, false, M>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
          // (the point side stays generic: its PLAIN instantiation -- 48 registers, eight waves per SIMD -- measured
          //  88 us against 79 at BA-1; same bits either way)
          BA_LAUNCH((ba_linearize_point_kernel<4>), dim3(g), dim3(256), st, V, P, Cm, X, Sn);
        } else {
          BA_LAUNCH((ba_linearize_kernel<false, 4, true, M>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
        }
      };
      if (plain_model == BA_SIMPLE_PINHOLE) launch(std::integral_constant<int, BA_SIMPLE_PINHOLE>{});
      else if (plain_model == BA_PINHOLE) launch(std::integral_constant<int, BA_PINHOLE>{});
      else launch(std::integral_constant<int, BA_SIMPLE_RADIAL>{});
    } else if (jac && split_linearize) {  // camera side in c-order, point side in p-order: every store coalesced
      if (kd == 4) {
        BA_LAUNCH((ba_linearize_kernel<true, 4, false>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
        BA_LAUNCH((ba_linearize_point_kernel<4>), dim3(g), dim3(256), st, V, P, Cm, X, Sn);
      } else if (kd == KD_WIDE) {
        BA_LAUNCH((ba_linearize_kernel<true, KD_WIDE, false>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
        BA_LAUNCH((ba_linearize_point_kernel<KD_WIDE>), dim3(g), dim3(256), st, V, P, Cm, X, Sn);
      } else {
        BA_LAUNCH((ba_linearize_kernel<true, KD_MAX, false>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
        BA_LAUNCH((ba_linearize_point_kernel<KD_MAX>), dim3(g), dim3(256), st, V, P, Cm, X, Sn);
      }
    } else if (kd == 4) {
      if (jac) BA_LAUNCH((ba_linearize_kernel<true, 4>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
      else BA_LAUNCH((ba_linearize_kernel<false, 4>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
    } else if (kd == KD_WIDE) {
      if (jac) BA_LAUNCH((ba_linearize_kernel<true, KD_WIDE>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
      else BA_LAUNCH((ba_linearize_kernel<false, KD_WIDE>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
    } else {
      if (jac) BA_LAUNCH((ba_linearize_kernel<true, KD_MAX>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
      else BA_LAUNCH((ba_linearize_kernel<false, KD_MAX>), dim3(g), dim3(256), st, V, P, Cm, X, Sn, partials.p);
    }
    BA_LAUNCH(ba_final_sum_kernel, dim3(1), dim3(1024), st, partials.p, g, scalars.p + slot);
    if (use_priors()) {
      if (jac) BA_LAUNCH(ba_prior_linearize_kernel<true>, dim3(1), dim3(256), st, V, Q, P, Sn, scalars.p + slot);
      else BA_LAUNCH(ba_prior_linearize_kernel<false>, dim3(1), dim3(256), st, V, Q, P, Sn, scalars.p + slot);
    }
  }

  // gradient of the (scaled) Jacobian and its squared column norms
  void gradient_and_diag() {
    if (V.n_chunks == 0) {
      BA_HIP(hipMemsetAsync(gc.p, 0, sizeof(double) * std::max(V.n_c, 1), st));
      BA_HIP(hipMemsetAsync(diag_c.p, 0, sizeof(double) * std::max(V.n_c, 1), st));
    } else {
      if (bd == PD) BA_LAUNCH((ba_block_jtv_kernel<true, PD>), dim3(V.n_chunks), dim3(64), st, V, res.p, gc.p, diag_c.p);
      else if (bd == KD_WIDE) BA_LAUNCH((ba_block_jtv_kernel<true, KD_WIDE>), dim3(V.n_chunks), dim3(64), st, V, res.p, gc.p, diag_c.p);
      else BA_LAUNCH((ba_block_jtv_kernel<true, KD_MAX>), dim3(V.n_chunks), dim3(64), st, V, res.p, gc.p, diag_c.p);
      heavy_reduce(2 * bd);
      BA_LAUNCH(ba_block_vec_finalize_kernel<true>, dim3(grid_for(V.n_blk * bd, 128)), dim3(128), st, V, gc.p, diag_c.p);
    }
    // (g_p and the point column norms are written for every variable point by either kernel)
    if (V.n_tiles > 0)
      BA_LAUNCH(ba_point_reduce_tiled_kernel<0>, dim3(V.n_tiles), dim3(TILE_PTS), st, V, nullptr, nullptr, gp.p, diag_p.p, Craw.p);
    else {
      BA_LAUNCH(ba_point_grad_kernel, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, gp.p, diag_p.p);
      BA_LAUNCH(ba_point_gram_kernel, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, Craw.p);
    }
    if (use_priors())
      BA_LAUNCH(ba_prior_accumulate_kernel<0>, dim3(grid_for(Q.n_tblk, 64)), dim3(64), st, V, Q, gc.p, diag_c.p);
    // one all-reduce for everything a linearisation sums over ranks (E^T E of the points changes only here).
    // Point sharding: a point's gradient, column norms and E^T E are complete locally -> camera-side prefix only.
    if (comm.by_point) comm.allreduce(lin_sums.p, 2 * (size_t)V.n_c, st);
    else comm.allreduce(lin_sums.p, lin_sums.n, st);
  }

  // y = (sum over ranks of J_c^T v) for this rank's observations, into tmpc
  void block_jtv_reduced(const double* vin, const double* x_for_priors = nullptr) {
    if (V.n_chunks == 0) BA_HIP(hipMemsetAsync(tmpc.p, 0, sizeof(double) * std::max(V.n_c, 1), st));
    if (V.n_chunks > 0) {
      if (bd == PD) BA_LAUNCH((ba_block_jtv_kernel<false, PD>), dim3(V.n_chunks), dim3(64), st, V, vin, tmpc.p, nullptr);
      else if (bd == KD_WIDE) BA_LAUNCH((ba_block_jtv_kernel<false, KD_WIDE>), dim3(V.n_chunks), dim3(64), st, V, vin, tmpc.p, nullptr);
      else BA_LAUNCH((ba_block_jtv_kernel<false, KD_MAX>), dim3(V.n_chunks), dim3(64), st, V, vin, tmpc.p, nullptr);
      heavy_reduce(bd);
      BA_LAUNCH(ba_block_vec_finalize_kernel<false>, dim3(grid_for(V.n_blk * bd, 128)), dim3(128), st, V, tmpc.p, nullptr);
    }
    if (x_for_priors && use_priors()) {  // + sum over priors J^T (J x)
      BA_LAUNCH(ba_prior_jx_kernel, dim3(grid_for(Q.n, 64)), dim3(64), st, Q, x_for_priors);
      BA_LAUNCH(ba_prior_accumulate_kernel<2>, dim3(grid_for(Q.n_tblk, 64)), dim3(64), st, V, Q, tmpc.p, nullptr);
    }
    comm.allreduce(tmpc.p, V.n_c, st);
  }

  template <int MODE>
  void point_pass() {
    if (V.n_tiles > 0)
      BA_LAUNCH(ba_point_pass_tiled_kernel<MODE>, dim3(V.n_tiles), dim3(TILE_PTS), st, V, Cinv.p, jx.p, gp.p, v.p, dp.p);
    else
      BA_LAUNCH(ba_point_pass_kernel<MODE>, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, Cinv.p, jx.p, gp.p,
                v.p, dp.p);
  }

  // q = S x = (B + Dc^2) x - E C^-1 E^T x
  // `inexact`: called by the CG iteration (may stream the fp32 operator copies); the exact formation by
  // operator products passes false.
  void schur_multiply(const double* xin, double* qout, bool inexact = false) {
    if (comm.world == 1 && !use_priors() && V.n_chunks > 0 && V.n_obs > 0) {
      // single GPU, no priors: nothing sits between J_c^T v and the block sums -> one tail kernel
      schur_streams(xin, inexact && op32);
      heavy_reduce(bd);
      BA_LAUNCH(ba_block_vec_finalize_q_kernel, dim3(grid_for(V.n_blk * bd, 128)), dim3(128), st, V, Dc.p, xin, qout);
      return;
    }
    const int go = grid_for(V.n_obs, 256);
    if (V.n_obs > 0) {
      if (kd == 4) BA_LAUNCH(ba_obs_jx_kernel<4>, dim3(go), dim3(256), st, V, xin, jx.p);
      else if (kd == KD_WIDE) BA_LAUNCH(ba_obs_jx_kernel<KD_WIDE>, dim3(go), dim3(256), st, V, xin, jx.p);
      else BA_LAUNCH(ba_obs_jx_kernel<KD_MAX>, dim3(go), dim3(256), st, V, xin, jx.p);
    }
    if (comm.world == 1 || comm.by_point) {
      point_pass<0>();  // E^T x, C^-1 and E u of a point are local
    } else {
      BA_HIP(hipMemsetAsync(tbuf.p, 0, sizeof(double) * std::max(V.n_p, 1), st));
      BA_LAUNCH(ba_point_t_kernel, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, jx.p, tbuf.p);
      comm.allreduce(tbuf.p, V.n_p, st);
      BA_LAUNCH(ba_point_apply_kernel<0>, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, Cinv.p, jx.p, gp.p,
                tbuf.p, v.p, dp.p);
    }
    block_jtv_reduced(v.p, xin);
    BA_LAUNCH(ba_dsq_x_kernel, dim3(grid_for(V.n_c, 256)), dim3(256), st, V.n_c, Dc.p, xin, qout);
    BA_LAUNCH(ba_add_kernel, dim3(grid_for(V.n_c, 256)), dim3(256), st, V.n_c, tmpc.p, qout);
  }

  // what the explicit formation of the exact tiers reads (ba_schur_explicit.h)
  ba_explicit::FormArgs form_args() {
    ba_explicit::FormArgs fa{};
    fa.n_obs = V.n_obs; fa.n_points = V.n_points; fa.n_c = V.n_c; fa.kd = kd;
    fa.Jpose = V.Jpose; fa.Jcam = V.Jcam; fa.Jsens = V.sens_off ? V.Jsens : nullptr; fa.Jpt = V.Jpt;
    fa.Cinv = Cinv.p; fa.a2c = V.a2c; fa.pt_ptr = V.pt_ptr; fa.pt_off = V.pt_off;
    fa.a_pose = V.a_pose; fa.a_cam = V.a_cam; fa.a_sensor = V.sens_off ? V.a_sensor : nullptr;
    fa.a_pt = V.a_pt; fa.n_poses = V.n_poses;
    fa.pose_off = V.pose_off; fa.pose_dim = V.pose_dim; fa.cam_off = V.cam_off; fa.cam_dim = V.cam_dim;
    fa.sens_off = V.sens_off;
    fa.fixed_point = opt.jacobi_scaling != 0;  // columns of norm < 1: integer accumulation, bit-reproducible
    fa.bad = chol_info.p + 1;                  // raised by a term the fixed point cannot hold (NaN, out of bound)
    fa.pairs = nullptr;
    return fa;
  }

  // DENSE_SCHUR: x = S^-1 rhs with S built from n_c operator products
  int dense_schur() {
    const int n = V.n_c;
    if (Sdense.n < (size_t)n * n + n) throw std::runtime_error("dense Schur buffer");
    if (!dense_by_products) {
      // explicit formation (pair-major, or one wave per point) + blocked Cholesky on the f64 matrix cores
      ba_explicit::FormArgs fa = form_args();
      fa.pairs = pair_lists.inc ? &pair_lists : nullptr;
      ba_explicit::form(fa, Sdense.p, st);
      if (use_priors()) ba_explicit::add_prior_rows(Sdense.p, n, Q.J, Q.po, Q.so, Q.pdim, Q.n, fa.fixed_point, fa.bad, st);
      ba_explicit::finish(Sdense.p, n, fa.fixed_point, fa.bad, st);
      if (comm.world > 1) comm.allreduce(Sdense.p, (size_t)n * n, st);  // point sharding: partial sums per rank
      ba_explicit::add_lm_diagonal(Sdense.p, n, Dc.p, st);
      ba_explicit::Workspace ws;
      ws.Linv = chol_linv.p; ws.tmp = chol_tmp.p; ws.info = chol_info.p;
      ws.st2 = st_chol; ws.ev_panel = ev_chol_panel; ws.ev_u2 = ev_chol_u2;
      double ms = 0.0;
      ba_explicit::factor_solve(Sdense.p, n, rhs.p, x.p, ws, st, ev0, ev1, &ms);
      factor_ms += ms;
      return 1;
    }
    for (int i = 0; i < n; ++i) {
      BA_LAUNCH(ba_unit_vector_kernel, dim3(grid_for(n, 256)), dim3(256), st, n, i, pdir.p);
      schur_multiply(pdir.p, Sdense.p + (size_t)i * n);
    }
    BA_LAUNCH(ba_dense_cholesky_solve_kernel, dim3(1), dim3(1024), st, n, Sdense.p, rhs.p, x.p);
    return 1;
  }

  // The three streaming kernels of one implicit product on a single GPU: J_c p (c-order -> p-order),
  // the point pass, J_c^T v into per-chunk partials. ba_pcg_fused_kernel finishes the product.
  void schur_streams(const double* xin, bool use32) {
    const int go = grid_for(V.n_obs, 256);
    if (use32) {  // the inexact inner solve streams the fp32 copies of the columns (fp64 accumulation)
      if (kd == 4) BA_LAUNCH((ba_obs_jx_kernel<4, float>), dim3(go), dim3(256), st, V, xin, jx.p);
      else if (kd == KD_WIDE) BA_LAUNCH((ba_obs_jx_kernel<KD_WIDE, float>), dim3(go), dim3(256), st, V, xin, jx.p);
      else BA_LAUNCH((ba_obs_jx_kernel<KD_MAX, float>), dim3(go), dim3(256), st, V, xin, jx.p);
      BA_LAUNCH((ba_point_pass_tiled_kernel<0, float>), dim3(V.n_tiles), dim3(TILE_PTS), st, V, Cinv.p, jx.p, gp.p, v.p, dp.p);
      if (bd == PD) BA_LAUNCH((ba_block_jtv_kernel<false, PD, float>), dim3(V.n_chunks), dim3(64), st, V, v.p, tmpc.p, nullptr);
      else if (bd == KD_WIDE) BA_LAUNCH((ba_block_jtv_kernel<false, KD_WIDE, float>), dim3(V.n_chunks), dim3(64), st, V, v.p, tmpc.p, nullptr);
      else BA_LAUNCH((ba_block_jtv_kernel<false, KD_MAX, float>), dim3(V.n_chunks), dim3(64), st, V, v.p, tmpc.p, nullptr);
      return;
    }
    if (kd == 4) BA_LAUNCH(ba_obs_jx_kernel<4>, dim3(go), dim3(256), st, V, xin, jx.p);
    else if (kd == KD_WIDE) BA_LAUNCH(ba_obs_jx_kernel<KD_WIDE>, dim3(go), dim3(256), st, V, xin, jx.p);
    else BA_LAUNCH(ba_obs_jx_kernel<KD_MAX>, dim3(go), dim3(256), st, V, xin, jx.p);
    point_pass<0>();
    if (bd == PD) BA_LAUNCH((ba_block_jtv_kernel<false, PD>), dim3(V.n_chunks), dim3(64), st, V, v.p, tmpc.p, nullptr);
    else if (bd == KD_WIDE) BA_LAUNCH((ba_block_jtv_kernel<false, KD_WIDE>), dim3(V.n_chunks), dim3(64), st, V, v.p, tmpc.p, nullptr);
    else BA_LAUNCH((ba_block_jtv_kernel<false, KD_MAX>), dim3(V.n_chunks), dim3(64), st, V, v.p, tmpc.p, nullptr);
  }

  int pcg_fused(int max_iter, double q_tol) {
    BA_LAUNCH(ba_pcg_fused_kernel<true>, dim3(1), dim3(1024), st, V, Dc.p, Minv.p, rhs.p, scalars.p, x.p, r.p, z.p, pdir.p, q.p);
    if (scalar(S_RHO) == 0.0) return 0;
    double Q0 = 0.0;
    int it;
    for (it = 1; it <= max_iter; ++it) {
      BA_HIP(hipEventRecord(ev0, st));
      schur_streams(pdir.p, op32);
      BA_HIP(hipEventRecord(ev1, st));
      heavy_reduce(bd);
      BA_LAUNCH(ba_pcg_fused_kernel<false>, dim3(1), dim3(1024), st, V, Dc.p, Minv.p, rhs.p, scalars.p, x.p, r.p, z.p, pdir.p, q.p);
      double h[NSCALAR];
      BA_HIP(hipMemcpyAsync(h, scalars.p, sizeof(h), hipMemcpyDeviceToHost, st));
      BA_HIP(hipStreamSynchronize(st));
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { g_spmv_ms += ms; g_spmv_launches += 1; }
      const double rho = h[S_RHO], pq = h[S_PQ], Q1 = h[S_Q];
      if (!(rho > 0.0) || !std::isfinite(rho) || !(pq > 0.0) || !std::isfinite(pq)) break;
      const double zeta = it * (Q1 - Q0) / Q1;
      if (zeta < q_tol) break;
      Q0 = Q1;
    }
    return std::min(it, max_iter);
  }

  int pcg_pipelined(int max_iter, double q_tol) {
    const int n = V.n_c, gv = std::max(grid_for(n, 256), 1), nparts = grid_for(V.n_blk, PCGP_T);
    if (!pcgp_host) {
      BA_HIP(hipHostMalloc((void**)&pcgp_host, 2 * sizeof(PcgHostSlot), hipHostMallocMapped));
      BA_HIP(hipHostGetDevicePointer((void**)&pcgp_host_dev, pcgp_host, 0));
      pcgp_part.alloc((size_t)2 * 3 * nparts);
      pcgp_qhist.alloc(2);
      pcgp_stop.alloc(1);
      for (int k = 0; k < 2; ++k) {
        BA_HIP(hipEventCreateWithFlags(&pcgp_ev_dir[k], hipEventDisableTiming));
        BA_HIP(hipEventCreate(&pcgp_ev_s0[k]));
        BA_HIP(hipEventCreate(&pcgp_ev_s1[k]));
      }
    }
    PcgDev D;
    D.part = pcgp_part.p; D.nparts = nparts; D.stop = pcgp_stop.p; D.host = pcgp_host_dev; D.qhist = pcgp_qhist.p;
    if (bd == PD) BA_LAUNCH(ba_pcgp_init_kernel<PD>, dim3(nparts), dim3(PCGP_T), st, V, D, Minv.p, rhs.p, x.p, r.p, z.p);
    else if (bd == KD_MAX) BA_LAUNCH(ba_pcgp_init_kernel<KD_MAX>, dim3(nparts), dim3(PCGP_T), st, V, D, Minv.p, rhs.p, x.p, r.p, z.p);
    else BA_LAUNCH(ba_pcgp_init_kernel<KD_WIDE>, dim3(nparts), dim3(PCGP_T), st, V, D, Minv.p, rhs.p, x.p, r.p, z.p);
    V.stop = pcgp_stop.p;  // the streaming kernels of an iteration enqueued past convergence return at entry
    auto enqueue = [&](int k) {
      BA_LAUNCH(ba_pcgp_dir_kernel, dim3(gv), dim3(256), st, n, D, k, max_iter, q_tol, z.p, pdir.p);
      BA_HIP(hipEventRecord(pcgp_ev_dir[k & 1], st));
      BA_HIP(hipEventRecord(pcgp_ev_s0[k & 1], st));
      schur_streams(pdir.p, op32);
      BA_HIP(hipEventRecord(pcgp_ev_s1[k & 1], st));
      heavy_reduce(bd);
      const double* reduced = nullptr;
      if (comm.world > 1) {
        // point sharding: every rank holds J_c^T v of its own observations -- block sums, then ONE all-reduce of the
        // camera-space vector (n_c doubles: 64 KB at BA-1) on the solver's stream. With the RCCL transport nothing
        // here touches the host: the iteration stays enqueued one ahead, the stop flag is computed identically on
        // every rank from the identical reduced vector (an iteration enqueued past convergence still runs its
        // all-reduce on every rank, on stale data nobody reads).
        BA_LAUNCH(ba_block_vec_finalize_kernel<false>, dim3(grid_for(V.n_blk * bd, 128)), dim3(128), st, V, tmpc.p, nullptr);
        comm.allreduce(tmpc.p, (size_t)n, st);
        reduced = tmpc.p;
      }
      if (bd == PD) {
        BA_LAUNCH(ba_pcgp_tail_kernel<PD>, dim3(nparts), dim3(PCGP_T), st, V, D, k, Dc.p, pdir.p, q.p, reduced);
        BA_LAUNCH(ba_pcgp_step_kernel<PD>, dim3(nparts), dim3(PCGP_T), st, V, D, k, Minv.p, rhs.p, pdir.p, q.p, x.p, r.p, z.p);
      } else if (bd == KD_MAX) {
        BA_LAUNCH(ba_pcgp_tail_kernel<KD_MAX>, dim3(nparts), dim3(PCGP_T), st, V, D, k, Dc.p, pdir.p, q.p, reduced);
        BA_LAUNCH(ba_pcgp_step_kernel<KD_MAX>, dim3(nparts), dim3(PCGP_T), st, V, D, k, Minv.p, rhs.p, pdir.p, q.p, x.p, r.p, z.p);
      } else {
        BA_LAUNCH(ba_pcgp_tail_kernel<KD_WIDE>, dim3(nparts), dim3(PCGP_T), st, V, D, k, Dc.p, pdir.p, q.p, reduced);
        BA_LAUNCH(ba_pcgp_step_kernel<KD_WIDE>, dim3(nparts), dim3(PCGP_T), st, V, D, k, Minv.p, rhs.p, pdir.p, q.p, x.p, r.p, z.p);
      }
    };
    enqueue(1);
    BA_HIP(hipEventSynchronize(pcgp_ev_dir[1]));
    int done = 0;
    if (!pcgp_host[0].stop) {  // (zero right-hand side otherwise)
      for (int k = 1;; ++k) {
        enqueue(k + 1);                                   // its dir kernel closes iteration k
        BA_HIP(hipEventSynchronize(pcgp_ev_dir[(k + 1) & 1]));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pcgp_ev_s0[k & 1], pcgp_ev_s1[k & 1]) == hipSuccess) { g_spmv_ms += ms; g_spmv_launches += 1; }
        const PcgHostSlot h = pcgp_host[k & 1];
        if (h.iter != k) throw std::runtime_error("pipelined PCG: host slot out of step");
        if (h.stop) { done = std::min(k, max_iter); break; }
      }
    }
    V.stop = nullptr;
    return done;
  }

  int pcg(int max_iter, double q_tol) {
    // measured at BA-1: the single-workgroup kernel takes 133 us against 54 us for the five small kernels it
    // replaces (a lane's blocks are chains of dependent global loads that one workgroup cannot hide): opt-in only
    const bool fused_env = dev_switch_int("COLMAP_AMD_BA_PCG_FUSED", 0) != 0;
    if (fused_env && comm.world == 1 && !use_priors() && V.n_chunks > 0 && V.n_obs > 0 && V.n_blk <= 65536)
      return pcg_fused(max_iter, q_tol);
    // single GPU, no priors: three small kernels per iteration, stopping test on the device, host one iteration
    // behind (COLMAP_AMD_BA_PCG_PIPELINED=0: the step-by-step loop below, which sharded / prior solves always take)
    const bool pipelined = dev_switch_int("COLMAP_AMD_BA_PCG_PIPELINED", 1) != 0;
    // ... and point-sharded solves whose ranks all hold observations (pcg_all_ranks_have_work: agreed once per solve,
    // the ranks must take the same path); image-sharded ones all-reduce inside the point pass as well and keep the loop below
    const bool local_ok = !use_priors() && V.n_chunks > 0 && V.n_obs > 0;
    if (pipelined && local_ok && (comm.world == 1 || (comm.by_point && pcg_all_ranks_have_work))) {
      ++g_pcg_pipelined_solves;
      return pcg_pipelined(max_iter, q_tol);
    }
    ++g_pcg_stepwise_solves;
    const int n = V.n_c;
    const int gv = grid_for(n, 256);
    BA_HIP(hipMemsetAsync(x.p, 0, sizeof(double) * n, st));
    BA_HIP(hipMemcpyAsync(r.p, rhs.p, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    BA_LAUNCH(ba_dot_kernel, dim3(1), dim3(1024), st, n, rhs.p, rhs.p, scalars.p + S_RHO);
    if (scalar(S_RHO) == 0.0) return 0;
    double Q0 = 0.0;
    int it;
    const int nparts = grid_for(V.n_blk, 256);
    for (it = 1; it <= max_iter; ++it) {
      BA_LAUNCH(ba_pcg_precond_kernel, dim3(nparts), dim3(256), st, V, Minv.p, r.p, z.p, pcg_part.p);
      BA_LAUNCH(ba_pcg_dir_kernel, dim3(gv), dim3(256), st, n, scalars.p, pcg_part.p, nparts, it == 1 ? 1 : 0, z.p, pdir.p);
      BA_HIP(hipEventRecord(ev0, st));
      schur_multiply(pdir.p, q.p, true);
      BA_HIP(hipEventRecord(ev1, st));
      BA_LAUNCH(ba_pcg_update_kernel, dim3(1), dim3(1024), st, n, scalars.p, pcg_part.p, nparts, pdir.p, q.p, rhs.p, x.p, r.p);
      double h[NSCALAR];
      BA_HIP(hipMemcpyAsync(h, scalars.p, sizeof(h), hipMemcpyDeviceToHost, st));
      BA_HIP(hipStreamSynchronize(st));
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { g_spmv_ms += ms; g_spmv_launches += 1; }
      const double rho = h[S_RHO], pq = h[S_PQ], Q1 = h[S_Q];
      if (!(rho > 0.0) || !std::isfinite(rho) || !(pq > 0.0) || !std::isfinite(pq)) break;
      const double zeta = it * (Q1 - Q0) / Q1;
      if (zeta < q_tol) break;
      Q0 = Q1;
    }
    return std::min(it, max_iter);
  }

  // Plus() of every parameter block in one launch; `tf` says how the step is read (StepSrc), `maxdiff` adds
  // max |x_plus - x| into scalars[S_GMAX]
  void apply_step(const double* sc, const double* sc_scale, const double* sp, const double* sp_scale, int tf, bool maxdiff,
                  double* P2, double* C2, double* X2) {
    const int nb = grid_for(V.n_points, 128) + grid_for(V.n_poses, 128) + grid_for(V.n_cams, 128) +
                   (V.sens_off ? grid_for(V.n_sensors, 128) : 0);
    const StepSrc Sc{sc, sc_scale, tf}, Sp{sp, sp_scale, tf};
    if (maxdiff)
      BA_LAUNCH(ba_apply_all_kernel<true>, dim3(std::max(nb, 1)), dim3(128), st, V, Sc, Sp, poses.p, P2, cams.p, C2, points.p, X2,
                sensors.p, sensors2.p);
    else
      BA_LAUNCH(ba_apply_all_kernel<false>, dim3(std::max(nb, 1)), dim3(128), st, V, Sc, Sp, poses.p, P2, cams.p, C2, points.p, X2,
                sensors.p, sensors2.p);
  }

  void run(ba_result* out) {
    const auto t_entry = std::chrono::steady_clock::now();
    BA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    BA_HIP(hipEventCreate(&ev0));
    BA_HIP(hipEventCreate(&ev1));
    BA_HIP(hipEventCreate(&ev2));
    BA_HIP(hipEventCreate(&ev3));
    out->termination_type = BA_FAILURE;
    const int n = build(out);
    if (n == 0) return;
    const int nc = V.n_c, np = V.n_p;
    const int gvc = grid_for(nc, 256), gvp = grid_for(np, 256);
    if (opt.linear_solver_type < BA_SOLVER_ITERATIVE_SCHUR || opt.linear_solver_type > BA_SOLVER_SPARSE_SCHUR)
      throw std::runtime_error("linear_solver_type");
    {
      // CreateSolverOptions' rule (bundle_adjustment_ceres.cc:203-213, CPU thresholds bundle_adjustment_ceres.h:
      // 68-69) on the number of pose blocks; both exact tiers run the explicit reduced camera system
      const int lst = opt.linear_solver_type;
      int tier = lst;
      if (lst == BA_SOLVER_AUTO)
        tier = prob.num_poses <= 50 ? BA_SOLVER_DENSE_SCHUR : (prob.num_poses <= 1000 ? BA_SOLVER_SPARSE_SCHUR : BA_SOLVER_ITERATIVE_SCHUR);
      const bool want_exact = tier == BA_SOLVER_DENSE_SCHUR || tier == BA_SOLVER_SPARSE_SCHUR;
      // an image-sharded solve splits a point's observations over the ranks: only the operator-product formation
      // (every product is all-reduced) is correct there, and only affordable for small systems
      dense_by_products = want_exact && comm.world > 1 && !comm.by_point;
      if (want_exact && dev_switch_int("COLMAP_AMD_BA_DENSE_BY_PRODUCTS", 0) != 0) dense_by_products = true;
      use_dense = want_exact && nc > 0 && nc <= (dense_by_products ? 1024 : 32768);
      out->linear_solver_used = use_dense ? tier : BA_SOLVER_ITERATIVE_SCHUR;
    }
    if (use_dense) {
      Sdense.alloc((size_t)nc * nc + nc);  // + the row of the right-hand side (ba_explicit::factor_solve)
      if (!dense_by_products) {
        ba_explicit::Workspace ws;
        chol_linv.alloc(ws.linv_doubles(nc)); chol_tmp.alloc(nc); chol_info.alloc(2);  // [pivot flag, formation flag]
        if (dev_switch_int("COLMAP_AMD_BA_CHOL_LOOKAHEAD", 1) != 0) {
          // the second stream carries bulk trailing updates that run beside the serial chain of small kernels on the
          // main stream: lowest priority, so that a freed workgroup slot goes to the chain first
          int prio_least = 0, prio_greatest = 0;
          BA_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
          BA_HIP(hipStreamCreateWithPriority(&st_chol, hipStreamNonBlocking, prio_least));
          BA_HIP(hipEventCreateWithFlags(&ev_chol_panel, hipEventDisableTiming));
          BA_HIP(hipEventCreateWithFlags(&ev_chol_u2, hipEventDisableTiming));
        }
        // pair-major formation: its incidence lists depend on the topology only -- built here, with the other
        // per-solve structures, not inside the LM loop (COLMAP_AMD_BA_FORM_PAIRS=0: the point-major kernel)
        if (dev_switch_int("COLMAP_AMD_BA_FORM_PAIRS", 1) != 0) (void)ba_explicit::build_pair_lists(form_args(), pair_lists, st);
      }
      BA_HIP(hipDeviceSynchronize());  // the allocation's memset runs on the NULL stream
    }
    if (comm.world > 1) {
      // which PCG loop runs must not depend on the rank: one sum over ranks of "this rank could not take the pipelined one"
      const double mine = (!use_priors() && V.n_chunks > 0 && V.n_obs > 0) ? 0.0 : 1.0;
      BA_HIP(hipMemcpyAsync(scalars.p + S_ITER, &mine, sizeof(double), hipMemcpyHostToDevice, st));
      BA_HIP(hipStreamSynchronize(st));
      comm.allreduce(scalars.p + S_ITER, 1, st);
      pcg_all_ranks_have_work = scalar(S_ITER) == 0.0;
    }
    factor_ms = 0.0;
    g_spmv_ms = 0.0; g_spmv_launches = 0;
    g_mfma_ms = 0.0; g_mfma_launches = 0;
    g_pcg_pipelined_solves = 0; g_pcg_stepwise_solves = 0;
    // bytes one implicit-Schur product streams: Jc (2x10) once for jx, Jp (2x3) twice, jx/v, Jc again
    g_spmv_bytes = (long long)V.n_obs * (2 * (PD + kd) * 8 * 2 + 6 * 8 * 2 + 4 * 8 * 3);

    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    int invalid_steps = 0;
    bool need_linearize = true, have_scale = false;
    double cost = 0.0;
    // scale = 1 until computed
    BA_LAUNCH(ba_scale_kernel, dim3(std::max(gvc, 1)), dim3(256), st, nc, diag_c.p, 0, scale_c.p);
    BA_LAUNCH(ba_scale_kernel, dim3(std::max(gvp, 1)), dim3(256), st, np, diag_p.p, 0, scale_p.p);
    BA_HIP(hipStreamSynchronize(st));
    const auto t_start = std::chrono::steady_clock::now();
    out->setup_seconds = std::chrono::duration<double>(t_start - t_entry).count();
    // ceres::IterationCallback as COLMAP uses it (controllers/bundle_adjustment.cc:40-57): asked between two
    // iterations; the parameter blocks hold the last accepted step when it ends the solve
    auto user_stop = [&](int iteration, double cost_now, double cost_change, bool step_ok, double rad, int lin) -> bool {
      if (!opt.iteration_callback) return false;
      ba_iteration_summary sm;
      sm.iteration = iteration;
      sm.step_is_successful = step_ok ? 1 : 0;
      sm.linear_solver_iterations = lin;
      sm.cost = cost_now;
      sm.cost_change = cost_change;
      sm.trust_region_radius = rad;
      sm.cumulative_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      const int rc = opt.iteration_callback(opt.iteration_callback_user, &sm);
      if (rc == BA_CALLBACK_CONTINUE) return false;
      out->termination_type = rc == BA_CALLBACK_TERMINATE ? BA_USER_SUCCESS : BA_USER_FAILURE;
      out->num_iterations = iteration;
      return true;
    };

    for (int iter = 0;; ++iter) {
      if (need_linearize) {
        launch_linearize(true, poses.p, cams.p, points.p, sensors.p, S_COST);
        gradient_and_diag();
        if (!have_scale) {
          // Jacobi scaling from the initial Jacobian, then re-linearise with it
          BA_LAUNCH(ba_scale_kernel, dim3(std::max(gvc, 1)), dim3(256), st, nc, diag_c.p,
                             opt.jacobi_scaling, scale_c.p);
          BA_LAUNCH(ba_scale_kernel, dim3(std::max(gvp, 1)), dim3(256), st, np, diag_p.p,
                             opt.jacobi_scaling, scale_p.p);
          launch_linearize(true, poses.p, cams.p, points.p, sensors.p, S_COST);
          gradient_and_diag();
          have_scale = true;
        }
        const bool one_sync = comm.world == 1;  // cost and gradient norm read together below
        if (!one_sync) {
          cost = scalar_sum(S_COST);
          if (iter == 0) out->initial_cost = cost;
        }
        // projected-gradient test: ||x - Plus(x, -g)||_inf with the unscaled gradient g = s * g_scaled
        // (the stored Jacobian is column-scaled: g_scaled = s * g, so g = g_scaled / s)
        // (one launch: the step -g / s is formed on the fly, Plus() of all blocks, the max of the differences)
        zero_scalar(S_GMAX);
        apply_step(gc.p, scale_c.p, gp.p, scale_p.p, 1, true, poses2.p, cams2.p, points2.p);
        double gmax;
        if (one_sync) {
          double h[NSCALAR];
          scalars_to_host(h);
          cost = h[S_COST];
          if (iter == 0) out->initial_cost = cost;
          gmax = h[S_GMAX];
        } else {
          gmax = scalar_max(S_GMAX);
        }
        if (gmax <= opt.gradient_tolerance) {
          out->termination_type = BA_CONVERGENCE;
          out->num_iterations = iter;
          break;
        }
        need_linearize = false;
        if (iter == 0 && user_stop(0, cost, 0.0, true, radius, 0)) break;
      }
      if (iter >= opt.max_num_iterations) {
        out->termination_type = BA_NO_CONVERGENCE;
        out->num_iterations = iter;
        break;
      }
      // LM diagonal, point blocks, Schur-Jacobi preconditioner
      BA_LAUNCH(ba_lm_diag_kernel, dim3(std::max(gvc, 1)), dim3(256), st, nc, diag_c.p, radius,
                         opt.min_lm_diagonal, opt.max_lm_diagonal, Dc.p);
      BA_LAUNCH(ba_lm_diag_kernel, dim3(std::max(gvp, 1)), dim3(256), st, np, diag_p.p, radius,
                         opt.min_lm_diagonal, opt.max_lm_diagonal, Dp.p);
      BA_LAUNCH(ba_point_blocks_kernel, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, Craw.p, Dp.p, Cinv.p);
      int lin_iters = 0;
      bool mfma_pending = false;
      if (nc > 0) {
        if (V.n_chunks == 0) BA_HIP(hipMemsetAsync(M.p, 0, sizeof(double) * std::max(moff_total, 1), st));
        const bool rhs_pass_fused = V.n_chunks > 0 && V.n_tiles > 0 && (comm.world == 1 || comm.by_point);
        if (V.n_chunks > 0) {  // (ba_block_mat_finalize_kernel<false> assigns every entry of every block)
          // (tiles, points local: the reduced right-hand side's point pass -- which holds E_o and needs C^-1 anyway --
          //  leaves G_o behind; v is not touched until block_jtv_reduced below)
          if (rhs_pass_fused)
            BA_LAUNCH((ba_point_pass_tiled_kernel<4>), dim3(V.n_tiles), dim3(TILE_PTS), st, V, Cinv.p, jx.p, gp.p, v.p, dp.p, (double*)nullptr, Gobs.p);
          else
            BA_LAUNCH(ba_obs_schur_g_kernel, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V, Cinv.p, Gobs.p);
          BA_HIP(hipEventRecord(ev2, st));
          const bool gram_lds = dev_switch_int("COLMAP_AMD_BA_GRAM_LDS", 1) != 0;
          if (gram_lds && bd == PD) BA_LAUNCH(ba_block_gram_lds_kernel<PD>, dim3(V.n_chunks), dim3(64 * GRAM_WAVES), st, V, Gobs.p);
          else if (gram_lds && bd == KD_MAX) BA_LAUNCH(ba_block_gram_lds_kernel<KD_MAX>, dim3(V.n_chunks), dim3(64 * GRAM_WAVES), st, V, Gobs.p);
          else if (gram_lds) BA_LAUNCH(ba_block_gram_lds_kernel<KD_WIDE>, dim3(V.n_chunks), dim3(64 * GRAM_WAVES), st, V, Gobs.p);
          else BA_LAUNCH(ba_block_gram_kernel, dim3(V.n_chunks), dim3(64), st, V, Gobs.p);
          BA_HIP(hipEventRecord(ev3, st));
          mfma_pending = true;
          heavy_reduce(bd * bd);
          BA_LAUNCH(ba_block_mat_finalize_kernel<false>, dim3(grid_for(V.n_blk * bd * bd, 128)), dim3(128), st, V, M.p);
          if (n_paired > 0) {  // observation pairs of a point inside one block: shared intrinsics, rig frames
            if (bd == PD) BA_LAUNCH(ba_obs_w_kernel<PD>, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V);
            else if (bd == KD_WIDE) BA_LAUNCH(ba_obs_w_kernel<KD_WIDE>, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V);
            else BA_LAUNCH(ba_obs_w_kernel<KD_MAX>, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V);
            if (PV.n > 0) {  // per incidence (single GPU)
              if (bd == KD_WIDE) {
                BA_LAUNCH(ba_pair_cross_kernel<KD_WIDE>, dim3(PV.n_chunks), dim3(KD_WIDE * KD_WIDE), st, V, PV, Cinv.p);
                BA_LAUNCH(ba_pair_finalize_kernel<KD_WIDE>, dim3(n_pair_blk, KD_WIDE * KD_WIDE / 64), dim3(1024), st, V, PV, pv_pair_blk.p, M.p);
              } else {
                BA_LAUNCH(ba_pair_cross_kernel<KD_MAX>, dim3(PV.n_chunks), dim3(KD_MAX * KD_MAX), st, V, PV, Cinv.p);
                BA_LAUNCH(ba_pair_finalize_kernel<KD_MAX>, dim3(n_pair_blk, 1), dim3(1024), st, V, PV, pv_pair_blk.p, M.p);
              }
            } else {  // per observation (sharded solves: the local pairs)
              if (bd == PD) BA_LAUNCH(ba_block_schur_cross_kernel<PD>, dim3(V.n_chunks), dim3(64), st, V, Cinv.p);
              else if (bd == KD_WIDE) BA_LAUNCH(ba_block_schur_cross_kernel<KD_WIDE>, dim3(V.n_chunks), dim3(64), st, V, Cinv.p);
              else BA_LAUNCH(ba_block_schur_cross_kernel<KD_MAX>, dim3(V.n_chunks), dim3(64), st, V, Cinv.p);
              heavy_reduce(bd * bd);
              BA_LAUNCH(ba_block_mat_finalize_kernel<true>, dim3(grid_for(V.n_blk * bd * bd, 128)), dim3(128), st, V, M.p);
            }
          }
        }
        if (use_priors())
          BA_LAUNCH(ba_prior_accumulate_kernel<1>, dim3(grid_for(Q.n_tblk, 64)), dim3(64), st, V, Q, M.p, nullptr);
        if (IV.n > 0) {  // image sharding: pairs of observations of a point in a shared intrinsics block on different ranks
          const int gi = grid_for(IV.n, 128);
          const size_t wn = (size_t)IV.n * (kd == KD_WIDE ? KD_WIDE : KD_MAX) * 3;
          if (kd == KD_WIDE) BA_LAUNCH(ba_inc_w_kernel<KD_WIDE>, dim3(gi), dim3(128), st, V, IV, inc_wloc.p);
          else BA_LAUNCH(ba_inc_w_kernel<KD_MAX>, dim3(gi), dim3(128), st, V, IV, inc_wloc.p);
          BA_HIP(hipMemcpyAsync(inc_wtot.p, inc_wloc.p, sizeof(double) * wn, hipMemcpyDeviceToDevice, st));
          comm.allreduce(inc_wtot.p, wn, st);
          if (kd == KD_WIDE) {
            BA_LAUNCH(ba_inc_correct_kernel<KD_WIDE>, dim3(IV.n_chunks), dim3(KD_WIDE * KD_WIDE), st, V, IV, Cinv.p, inc_wloc.p,
                      inc_wtot.p, comm.rank, comm.world);
            BA_LAUNCH(ba_inc_finalize_kernel<KD_WIDE>, dim3(grid_for((size_t)V.n_blk * KD_WIDE * KD_WIDE, 256)), dim3(256), st, V, IV, M.p);
          } else {
            BA_LAUNCH(ba_inc_correct_kernel<KD_MAX>, dim3(IV.n_chunks), dim3(KD_MAX * KD_MAX), st, V, IV, Cinv.p, inc_wloc.p,
                      inc_wtot.p, comm.rank, comm.world);
            BA_LAUNCH(ba_inc_finalize_kernel<KD_MAX>, dim3(grid_for((size_t)V.n_blk * KD_MAX * KD_MAX, 256)), dim3(256), st, V, IV, M.p);
          }
        }
        comm.allreduce(M.p, (size_t)moff_total, st);
        if (bd == PD) BA_LAUNCH(ba_block_invert_kernel<PD>, dim3(grid_for(V.n_blk, 64)), dim3(64), st, V, Dc.p, M.p, Minv.p);
        else if (bd == KD_WIDE) BA_LAUNCH(ba_block_invert_kernel<KD_WIDE>, dim3(grid_for(V.n_blk, 64)), dim3(64), st, V, Dc.p, M.p, Minv.p);
        else BA_LAUNCH(ba_block_invert_kernel<KD_MAX>, dim3(grid_for(V.n_blk, 64)), dim3(64), st, V, Dc.p, M.p, Minv.p);
        // reduced rhs = g_c - E C^-1 g_p  (g_p, C^-1 are global; the J_c^T part is summed over ranks)
        if (!rhs_pass_fused) point_pass<1>();
        block_jtv_reduced(v.p);
        BA_HIP(hipMemcpyAsync(rhs.p, gc.p, sizeof(double) * nc, hipMemcpyDeviceToDevice, st));
        BA_LAUNCH(ba_add_kernel, dim3(grid_for(nc, 256)), dim3(256), st, nc, tmpc.p, rhs.p);
        lin_iters = use_dense ? dense_schur() : pcg(opt.max_linear_solver_iterations, opt.eta);
        out->total_linear_iterations += lin_iters;
      }
      // back-substitution y_p = C^-1 (g_p - E^T y_c); step = -(y_c, y_p)
      if (V.n_obs > 0) {
        if (kd == 4) BA_LAUNCH(ba_obs_jx_kernel<4>, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V, x.p, jx.p);
        else if (kd == KD_WIDE) BA_LAUNCH(ba_obs_jx_kernel<KD_WIDE>, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V, x.p, jx.p);
        else BA_LAUNCH(ba_obs_jx_kernel<KD_MAX>, dim3(grid_for(V.n_obs, 256)), dim3(256), st, V, x.p, jx.p);
      }
      // (tiles: the same launch leaves the model cost change's partial sums behind -- columns, jx and y_p are in its registers)
      const bool model_fused = (comm.world == 1 || comm.by_point) && V.n_tiles > 0;
      if (model_fused) {
        BA_LAUNCH((ba_point_pass_tiled_kernel<3>), dim3(V.n_tiles), dim3(TILE_PTS), st, V, Cinv.p, jx.p, gp.p, v.p, dp.p, partials.p);
      } else if (comm.world == 1 || comm.by_point) {
        point_pass<2>();
      } else {
        BA_HIP(hipMemsetAsync(tbuf.p, 0, sizeof(double) * std::max(np, 1), st));
        BA_LAUNCH(ba_point_t_kernel, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, jx.p, tbuf.p);
        comm.allreduce(tbuf.p, np, st);
        BA_LAUNCH(ba_point_apply_kernel<2>, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, Cinv.p, jx.p, gp.p,
                  tbuf.p, v.p, dp.p);
      }
      // (the negated, unscaled camera-side step: only the priors' model term reads it; Plus() forms its own)
      if (use_priors()) BA_LAUNCH(ba_axpby_kernel, dim3(std::max(gvc, 1)), dim3(256), st, nc, -1.0, x.p, nullptr, stepc.p);
      // model cost change -(J step).(r + J step / 2): jx = J_c y_c of the back-substitution is still in place
      if (model_fused) {
        BA_LAUNCH(ba_final_sum_kernel, dim3(1), dim3(1024), st, partials.p, V.n_tiles, scalars.p + S_MODEL);
      } else if (V.n_tiles > 0) {
        BA_LAUNCH(ba_point_reduce_tiled_kernel<1>, dim3(V.n_tiles), dim3(TILE_PTS), st, V, jx.p, dp.p, partials.p, nullptr, nullptr);
        BA_LAUNCH(ba_final_sum_kernel, dim3(1), dim3(1024), st, partials.p, V.n_tiles, scalars.p + S_MODEL);
      } else {
        BA_LAUNCH(ba_model_from_jx_kernel, dim3(grid_for(V.n_points, 256)), dim3(256), st, V, jx.p, dp.p, partials.p);
        BA_LAUNCH(ba_final_sum_kernel, dim3(1), dim3(1024), st, partials.p, grid_for(V.n_points, 256), scalars.p + S_MODEL);
      }
      if (use_priors()) BA_LAUNCH(ba_prior_model_kernel, dim3(1), dim3(256), st, Q, stepc.p, scalars.p + S_MODEL);
      // Single GPU: the candidate is evaluated before the model change is known (it is almost always valid; an
      // invalid step wastes one cost evaluation) and both numbers come back with one synchronisation.
      const bool speculate = comm.world == 1;
      double spec_new_cost = 0.0;
      double model_change;
      if (speculate) {
        apply_step(x.p, scale_c.p, dp.p, scale_p.p, 3, false, poses2.p, cams2.p, points2.p);  // step = (-y) * column scale
        launch_linearize(false, poses2.p, cams2.p, points2.p, sensors2.p, S_NEWCOST);
        double h[NSCALAR];
        scalars_to_host(h);
        model_change = h[S_MODEL];
        spec_new_cost = h[S_NEWCOST];
      } else {
        model_change = scalar_sum(S_MODEL);  // (synchronises the stream)
      }
      if (mfma_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev2, ev3) == hipSuccess) { g_mfma_ms += ms; g_mfma_launches += 1; }
      }
      bool accepted = false;
      double new_cost = cost;
      if (!(model_change > 0.0) || !std::isfinite(model_change)) {
        if (++invalid_steps >= opt.max_num_consecutive_invalid_steps) {
          out->termination_type = BA_FAILURE;
          out->num_iterations = iter + 1;
          break;
        }
        radius /= decrease_factor;
        decrease_factor *= 2.0;
      } else {
        invalid_steps = 0;
        // undo the Jacobi scaling of the step and evaluate the candidate
        if (speculate) {
          new_cost = spec_new_cost;
        } else {
          apply_step(x.p, scale_c.p, dp.p, scale_p.p, 3, false, poses2.p, cams2.p, points2.p);
          launch_linearize(false, poses2.p, cams2.p, points2.p, sensors2.p, S_NEWCOST);
          new_cost = scalar_sum(S_NEWCOST);
        }
        const double rho = (cost - new_cost) / model_change;
        if (rho > opt.min_relative_decrease) {
          accepted = true;
          std::swap(poses.p, poses2.p);
          std::swap(cams.p, cams2.p);
          std::swap(points.p, points2.p);
          if (V.sens_off) std::swap(sensors.p, sensors2.p);
          V.poses = poses.p; V.cams = cams.p; V.points = points.p; V.sensors = sensors.p;
          const double t = 2.0 * rho - 1.0;
          radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
          radius = std::min(opt.max_trust_region_radius, radius);
          decrease_factor = 2.0;
          out->num_successful_steps++;
          need_linearize = true;
          if (opt.function_tolerance > 0 && std::fabs(cost - new_cost) <= opt.function_tolerance * cost) {
            log(out, new_cost, radius, lin_iters);
            out->termination_type = BA_CONVERGENCE;
            out->num_iterations = iter + 1;
            break;
          }
        } else {
          radius /= decrease_factor;
          decrease_factor *= 2.0;
        }
      }
      log(out, accepted ? new_cost : cost, radius, lin_iters);
      if (radius < opt.min_trust_region_radius) {
        out->termination_type = BA_CONVERGENCE;
        out->num_iterations = iter + 1;
        break;
      }
      if (user_stop(iter + 1, accepted ? new_cost : cost, accepted ? cost - new_cost : 0.0, accepted, radius, lin_iters)) break;
    }
    launch_linearize(false, poses.p, cams.p, points.p, sensors.p, S_NEWCOST);
    out->final_cost = scalar_sum(S_NEWCOST);
    out->lm_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    out->factor_seconds = factor_ms * 1e-3;
    BA_LAUNCH(ba_renorm_quat_kernel, dim3(grid_for(V.n_poses, 128)), dim3(128), st, V, poses.p);
    if (comm.world > 1 && comm.by_point) {
      // every rank moved its own points only: zero the others' variable points and sum over ranks
      BA_LAUNCH(ba_keep_own_points_kernel, dim3(grid_for(V.n_points, 128)), dim3(128), st, V, comm.rank, comm.world, points.p);
      comm.allreduce(points.p, 3 * (size_t)V.n_points, st);
    }
    if (V.sens_off) {
      BA_LAUNCH(ba_renorm_sensor_quat_kernel, dim3(grid_for(V.n_sensors, 128)), dim3(128), st, V, sensors.p);
      std::vector<double> hs(sensors.n);
      BA_HIP(hipMemcpyAsync(hs.data(), sensors.p, sizeof(double) * sensors.n, hipMemcpyDeviceToHost, st));
      BA_HIP(hipStreamSynchronize(st));
      for (int sidx = 0; sidx < prob.num_sensors; ++sidx)
        if (h_sens_off[sidx] >= 0)
          std::memcpy(prob.sensors + 7 * (size_t)sidx, hs.data() + 7 * (size_t)sidx, 7 * sizeof(double));
    }
    // write back variable blocks only (constant blocks stay bit-identical)
    std::vector<double> hp(poses.n), hc(cams.n), hx(points.n);
    BA_HIP(hipMemcpyAsync(hp.data(), poses.p, sizeof(double) * poses.n, hipMemcpyDeviceToHost, st));
    BA_HIP(hipMemcpyAsync(hc.data(), cams.p, sizeof(double) * cams.n, hipMemcpyDeviceToHost, st));
    BA_HIP(hipMemcpyAsync(hx.data(), points.p, sizeof(double) * points.n, hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < prob.num_poses; ++i)
      if (h_pose_off[i] >= 0) std::memcpy(prob.poses + 7 * (size_t)i, hp.data() + 7 * (size_t)i, 7 * sizeof(double));
    for (int k = 0; k < prob.num_cams; ++k)
      if (h_cam_off[k] >= 0)
        std::memcpy(prob.cams + BA_CAM_STRIDE * (size_t)k, hc.data() + BA_CAM_STRIDE * (size_t)k,
                    BA_CAM_STRIDE * sizeof(double));
    for (int j = 0; j < prob.num_points; ++j)
      if (h_pt_off[j] >= 0) std::memcpy(prob.points + 3 * (size_t)j, hx.data() + 3 * (size_t)j, 3 * sizeof(double));
  }

  void log(ba_result* out, double c, double rad, int lin) {
    if (out->log_cost && out->num_logged < opt.max_log) {
      out->log_cost[out->num_logged] = c;
      if (out->log_radius) out->log_radius[out->num_logged] = rad;
      if (out->log_linear_iters) out->log_linear_iters[out->num_logged] = lin;
      out->num_logged++;
    }
  }
};

}  // namespace

extern "C" {

int32_t ba_abi_version(void) { return COLMAP_AMD_BA_ABI_VERSION; }

void ba_options_init(ba_options* o) {
  // CeresBundleAdjustmentOptions ctor (bundle_adjustment_ceres.cc:102-115) over Ceres defaults
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 100;
  o->max_linear_solver_iterations = 200;
  o->function_tolerance = 0.0;
  o->gradient_tolerance = 1e-4;
  o->parameter_tolerance = 0.0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->eta = 1e-1;
  o->max_num_consecutive_invalid_steps = 10;
  o->jacobi_scaling = 1;
  o->loss_type = BA_LOSS_TRIVIAL;  // bundle_adjustment_ceres.h:42-51
  o->loss_scale = 1.0;
  o->linear_solver_type = BA_SOLVER_ITERATIVE_SCHUR;
}

static int SolveImpl(ba_problem* problem, const ba_options* options, int32_t gpu_index, const ba_comm* c,
                     ba_result* result) {
  try {
    if (!problem || !options || !result) throw std::runtime_error("null argument");
    double* lc = result->log_cost;
    double* lr = result->log_radius;
    int32_t* ll = result->log_linear_iters;
    std::memset(result, 0, sizeof(*result));
    result->log_cost = lc; result->log_radius = lr; result->log_linear_iters = ll;
    result->termination_type = BA_FAILURE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw std::runtime_error("no HIP device available: the MI355X bundle-adjustment backend has no CPU fallback");
    if (gpu_index >= ndev) throw std::runtime_error("gpu_index out of range");
    if (gpu_index >= 0) BA_HIP(hipSetDevice(gpu_index));
    Comm comm;
    if (c) {
      if (c->world_size < 1 || c->rank < 0 || c->rank >= c->world_size) throw std::runtime_error("bad rank / world_size");
      comm.rank = c->rank;
      comm.world = c->world_size;
      comm.fn = c->allreduce;
      comm.user = c->user;
      comm.nccl = reinterpret_cast<ncclComm_t>(c->rccl_comm);
      comm.by_point = c->sharding == BA_SHARD_BY_POINT;
      if (c->sharding != BA_SHARD_BY_IMAGE && c->sharding != BA_SHARD_BY_POINT)
        throw std::runtime_error("ba_comm.sharding must be BA_SHARD_BY_IMAGE or BA_SHARD_BY_POINT");
      if (comm.world > 1 && !comm.nccl && !comm.fn) throw std::runtime_error("ba_comm needs rccl_comm or an allreduce callback");
    }
    Solver s(*problem, *options, comm);
    s.run(result);
    return 0;
  } catch (const std::exception& e) {
    g_ba_error = e.what();
    return 1;
  }
}

int ba_solve(ba_problem* problem, const ba_options* options, int32_t gpu_index, ba_result* result) {
  return SolveImpl(problem, options, gpu_index, nullptr, result);
}

int ba_solve_sharded(ba_problem* problem, const ba_options* options, int32_t gpu_index, const ba_comm* comm,
                     ba_result* result) {
  return SolveImpl(problem, options, gpu_index, comm, result);
}

static int64_t ShardNumObservations(const ba_problem* p, int32_t rank, int32_t world_size, bool by_point) {
  if (!p || world_size < 1 || rank < 0 || rank >= world_size) return -1;
  std::vector<char> cam_var(p->num_cams, 0);
  for (int k = 0; k < p->num_cams; ++k) {
    const int P = num_params_of(p->cam_model[k]);
    for (int j = 0; j < P; ++j)
      if (!p->cam_const[(size_t)k * BA_CAM_STRIDE + j]) cam_var[k] = 1;
  }
  int64_t n = 0;
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int pi = p->obs_pose[o], xi = p->obs_point[o];
    const int si = p->obs_sensor ? p->obs_sensor[o] : -1;
    const bool sens_var = si >= 0 && p->sensor_const && !p->sensor_const[si];
    if (p->pose_const[pi] && !cam_var[p->obs_cam[o]] && p->point_const[xi] && !sens_var) continue;
    if ((by_point ? xi : pi) % world_size == rank) ++n;
  }
  return n;
}

int64_t ba_shard_num_observations(const ba_problem* p, int32_t rank, int32_t world_size) {
  return ShardNumObservations(p, rank, world_size, false);
}
int64_t ba_shard_num_observations_by_point(const ba_problem* p, int32_t rank, int32_t world_size) {
  return ShardNumObservations(p, rank, world_size, true);
}

int ba_rccl_unique_id(char id[128]) {
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the ABI buffer");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) { g_ba_error = "ncclGetUniqueId failed"; return 1; }
  std::memset(id, 0, 128);
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int ba_rccl_comm_create(const char id[128], int32_t rank, int32_t world_size, int32_t gpu_index, void** comm) {
  try {
    if (!comm) throw std::runtime_error("null argument");
    if (gpu_index >= 0) BA_HIP(hipSetDevice(gpu_index));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, world_size, u, rank);
    if (r != ncclSuccess) throw std::runtime_error(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
    *comm = c;
    return 0;
  } catch (const std::exception& e) {
    g_ba_error = e.what();
    return 1;
  }
}

void ba_rccl_comm_destroy(void* comm) {
  if (comm) (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(comm));
}

int ba_last_mfma_timing(double* total_ms, int64_t* launches) {
  if (total_ms) *total_ms = g_mfma_ms;
  if (launches) *launches = g_mfma_launches;
  return 0;
}

int ba_last_pcg_loops(int64_t* pipelined_solves, int64_t* stepwise_solves) {
  if (pipelined_solves) *pipelined_solves = g_pcg_pipelined_solves;
  if (stepwise_solves) *stepwise_solves = g_pcg_stepwise_solves;
  return 0;
}

int ba_last_spmv_timing(double* total_ms, int64_t* launches, int64_t* bytes_per_launch) {
  if (total_ms) *total_ms = g_spmv_ms;
  if (launches) *launches = g_spmv_launches;
  if (bytes_per_launch) *bytes_per_launch = g_spmv_bytes;
  return 0;
}

const char* ba_last_error(void) { return g_ba_error.c_str(); }

}  // extern "C"
