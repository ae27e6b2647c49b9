/*
 * pm_oracle.c -- CPU restatement of COLMAP's PatchMatch multi-view stereo.
 *
 * THIS FILE IS TEST INFRASTRUCTURE. It is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg). The
 * product (colmap_amd/) never includes, links, or calls it.
 *
 * PARITY STATUS: **pinned against the reference itself** (round 3). The reference tree holds no golden
 * vectors / known-answer tests for the PatchMatch algorithm, but its own
 * patch_match_cuda.cu / gpu_mat_prng.cu / gpu_mat_ref_image.cu compile for gfx950 where they lie
 * (`make -C oracle ref` -> oracle/_ref/libref_pm.so; oracle/ref_shim/README.md: a software texture
 * stands in for the image instructions gfx950 lacks, glog / Eigen / Bitmap stubs for the absent
 * dependencies). tests/test_pm_ref.py runs that library on the GPU box against this file in the
 * reference's order (order = 0): PRNG states, re-quantised reference image and first random depth
 * bit-exact, bilateral sums within 2e-7, random normals bit-equal, ComputeInitialCost within 8.8e-5
 * (mean 9e-7; the only arithmetic difference is device libm exp / sincos vs the polynomials below),
 * one iteration of sweeps: every pixel's depth identical; config[0] full solve: 99.2 % of the pixels
 * within 1e-2, same filter decisions on 99.8 %, same accuracy against ground truth
 * (profiles/r03_pm_ref_parity.json). Also pinned: the host pose helpers (mvs/image_test.cc known
 * answers, tests/test_pm_oracle.py), the CCW rotation index map (mvs/gpu_mat_test.cu), the XORWOW
 * generator against rocRAND on the device (tests/hip/rocrand_pin.hip).
 *
 * All citations are relative to /root/reference/src/colmap/mvs/ unless noted.
 *
 * Arithmetic specification (shared, independently implemented, by the HIP kernel
 * in colmap_amd/csrc/pm_kernels.hip; both must be bit-identical):
 *   - IEEE-754 binary32 everywhere the reference uses float, round-to-nearest-even,
 *     subnormals kept, NO implicit contraction (-ffp-contract=off); fused
 *     multiply-adds appear only where written as fmaf().
 *   - division and sqrt correctly rounded; rsqrt(x) := 1.0f / sqrtf(x).
 *   - exp / sin / cos are the polynomial kernels pm_exp / pm_sincos below (device
 *     libm and host libm differ in the last ulp, which a stochastic argmin
 *     algorithm amplifies; a fixed polynomial makes CPU == GPU exactly).
 *   - texture emulation: uint8 texel -> (float)b / 255.0f; point fetch = floor();
 *     border = 0; bilinear = the four-point formula of the reference's gfx9 path
 *     (patch_match_cuda.cu:426-442), full float weights.
 *   - PRNG: XORWOW, one state per pixel, seeded like hipRAND/rocRAND's
 *     rocrand_init(seed = linear thread id, subsequence 0, offset 0)
 *     (gpu_mat_prng.cu:36-48; generator recurrence /opt/rocm/include/rocrand/
 *     rocrand_xorwow.h -- the library the reference's HIP build links).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PMO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Public structs (flat, ctypes-friendly)                                    */
/* ------------------------------------------------------------------------- */

/* Mirrors PatchMatchOptions (patch_match_options.h:37-126); only the fields
 * that reach the kernel. Doubles like the reference, converted to float at the
 * same places the reference converts them (SweepOptions, patch_match_cuda.cu:
 * 1420-1438). */
typedef struct {
  double depth_min, depth_max;
  double sigma_spatial, sigma_color;
  double ncc_sigma;
  double min_triangulation_angle;       /* degrees */
  double incident_angle_sigma;
  double geom_consistency_regularizer;
  double geom_consistency_max_cost;
  double filter_min_ncc;
  double filter_min_triangulation_angle; /* degrees */
  double filter_geom_consistency_max_cost;
  int window_radius, window_step;
  int num_samples, num_iterations;
  int filter_min_num_consistent;
  int geom_consistency; /* bool */
  int filter;           /* bool */
  /* oracle-only controls */
  int max_sweeps;  /* <0: all 4*num_iterations; else stop after this many */
  int memoize;     /* 1: reuse bit-identical NCC values within a pixel step */
  int num_threads; /* <=0: OpenMP default */
  int order;       /* 0: reference order (sequential taps, incremental homography
                      stepping, patch_match_cuda.cu:503-569); 1: device order (taps
                      dealt to 16 lanes, per-tap direct homography, fixed 16-lane tree
                      sum) -- the order the HIP kernel evaluates the same sums in */
} pmo_options;

/* Mirrors mvs::Image (image.h:40-98): K,R,T row-major float + grey bitmap. */
typedef struct {
  int width, height;
  float K[9], R[9], T[3];
  const uint8_t* gray;  /* height*width, row-major */
  const float* depth;   /* height*width or NULL (geom_consistency input) */
  const float* normal;  /* 3*height*width slice-major or NULL */
} pmo_image;

/* ------------------------------------------------------------------------- */
/* Fixed-polynomial transcendental kernels (arithmetic spec)                 */
/* ------------------------------------------------------------------------- */

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* exp(x): Cody-Waite reduction + degree-5 minimax (Cephes expf constants),
 * every step an explicit float op. Returns 0 for x < -87. */
PMO_API float pmo_exp(float x) {
  if (!(x >= -87.0f)) {
    if (x != x) return x;
    return 0.0f;
  }
  if (x > 88.0f) x = 88.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = fmaf(p, r2, r) + 1.0f;
  const int ni = (int)n;
  return y * bits2f((uint32_t)(ni + 127) << 23);
}

/* sin/cos for |a| <= ~1e4: quadrant reduction with a 3-term pi/2 split, then
 * Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4]. */
PMO_API void pmo_sincos(float a, float* s_out, float* c_out) {
  const float q = rintf(a * 0.636619772367581343f); /* 2/pi */
  float r = fmaf(q, -1.5703125f, a);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188216e-8f, r);
  const float z = r * r;
  /* sin poly */
  float sp = -1.9515295891e-4f;
  sp = fmaf(sp, z, 8.3321608736e-3f);
  sp = fmaf(sp, z, -1.6666654611e-1f);
  const float sv = fmaf(sp * z, r, r);
  /* cos poly */
  float cp = 2.443315711809948e-5f;
  cp = fmaf(cp, z, -1.388731625493765e-3f);
  cp = fmaf(cp, z, 4.166664568298827e-2f);
  const float cv = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
  const int qi = ((int)q) & 3;
  float s, c;
  switch (qi) {
    case 0: s = sv; c = cv; break;
    case 1: s = cv; c = -sv; break;
    case 2: s = -sv; c = -cv; break;
    default: s = -cv; c = sv; break;
  }
  *s_out = s;
  *c_out = c;
}

static inline float pm_rsqrt(float x) { return 1.0f / sqrtf(x); }

/* float -> int with the saturating semantics of v_cvt_i32_f32 (NaN -> 0). */
static inline int sat_f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT32_MAX;
  if (f <= -2147483648.0f) return INT32_MIN;
  return (int)f;
}

/* ------------------------------------------------------------------------- */
/* XORWOW (gpu_mat_prng.cu:36-48; rocrand_xorwow.h)                          */
/* ------------------------------------------------------------------------- */

typedef struct { uint32_t x[5]; uint32_t d; } pmo_rng;

PMO_API void pmo_rng_init(pmo_rng* st, uint64_t seed) {
  st->x[0] = 123456789U; st->x[1] = 362436069U; st->x[2] = 521288629U;
  st->x[3] = 88675123U;  st->x[4] = 5783321U;   st->d = 6615241U;
  const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fU;
  const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbU;
  const uint32_t t0 = 1228688033U * s0;
  const uint32_t t1 = 2073658381U * s1;
  st->x[0] += t0; st->x[1] ^= t0; st->x[2] += t1; st->x[3] ^= t1; st->x[4] += t0;
  st->d += t1 + t0;
  /* subsequence 0, offset 0: no skip-ahead */
}

PMO_API uint32_t pmo_rng_next(pmo_rng* st) {
  const uint32_t t = st->x[0] ^ (st->x[0] >> 2);
  st->x[0] = st->x[1]; st->x[1] = st->x[2]; st->x[2] = st->x[3]; st->x[3] = st->x[4];
  st->x[4] = (st->x[4] ^ (st->x[4] << 4)) ^ (t ^ (t << 1));
  st->d += 362437U;
  return st->d + st->x[4];
}

/* curand_uniform / hiprand_uniform: (0, 1]  (rocrand_uniform.h:65-68) */
PMO_API float pmo_rng_uniform(pmo_rng* st) {
  const uint32_t v = pmo_rng_next(st);
  return 2.3283064e-10f + ((float)v * 2.3283064e-10f);
}

/* ------------------------------------------------------------------------- */
/* Host pose helpers (image.cc:97-150), float like the reference             */
/* ------------------------------------------------------------------------- */

static void mat33_mul(const float A[9], const float B[9], float C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

/* R = R2 * R1^T ; T = T2 - R * T1  (image.cc:97-113) */
PMO_API void pmo_compute_relative_pose(const float R1[9], const float T1[3], const float R2[9],
                                       const float T2[3], float R[9], float T[3]) {
  float R1t[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R1t[3 * i + j] = R1[3 * j + i];
  mat33_mul(R2, R1t, R);
  for (int i = 0; i < 3; ++i)
    T[i] = T2[i] - (R[3 * i + 0] * T1[0] + R[3 * i + 1] * T1[1] + R[3 * i + 2] * T1[2]);
}

/* P = K [R | T]  (image.cc:115-124) */
PMO_API void pmo_compose_projection_matrix(const float K[9], const float R[9], const float T[3],
                                           float P[12]) {
  float RT[12];
  for (int i = 0; i < 3; ++i) {
    RT[4 * i + 0] = R[3 * i + 0]; RT[4 * i + 1] = R[3 * i + 1];
    RT[4 * i + 2] = R[3 * i + 2]; RT[4 * i + 3] = T[i];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j)
      P[4 * i + j] = K[3 * i + 0] * RT[0 + j] + K[3 * i + 1] * RT[4 + j] + K[3 * i + 2] * RT[8 + j];
}

/* General 4x4 inverse by cofactors in float (Eigen's fixed-size 4x4 inverse is a
 * cofactor expansion as well); top three rows returned (image.cc:126-137). */
PMO_API void pmo_compose_inverse_projection_matrix(const float K[9], const float R[9],
                                                   const float T[3], float inv_P[12]) {
  float m[16];
  pmo_compose_projection_matrix(K, R, T, m);
  m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
  float inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float inv_det = 1.0f / det;
  for (int i = 0; i < 12; ++i) inv_P[i] = inv[i] * inv_det;
}

/* C = -R^T T  (image.cc:139-144) */
PMO_API void pmo_compute_projection_center(const float R[9], const float T[3], float C[3]) {
  for (int i = 0; i < 3; ++i)
    C[i] = -(R[0 + i] * T[0] + R[3 + i] * T[1] + R[6 + i] * T[2]);
}

/* R = RR * R ; T = RR * T  (image.cc:146-152) */
PMO_API void pmo_rotate_pose(const float RR[9], float R[9], float T[3]) {
  float Rn[9], Tn[3];
  mat33_mul(RR, R, Rn);
  for (int i = 0; i < 3; ++i) Tn[i] = RR[3 * i + 0] * T[0] + RR[3 * i + 1] * T[1] + RR[3 * i + 2] * T[2];
  memcpy(R, Rn, sizeof(Rn));
  memcpy(T, Tn, sizeof(Tn));
}

/* ------------------------------------------------------------------------- */
/* Problem state                                                             */
/* ------------------------------------------------------------------------- */

#define PMO_POSE_STRIDE 43 /* K4 R9 T3 C3 P12 invP12 (patch_match_cuda.cu:1762) */


typedef struct {
  int ref_w, ref_h, S, src_w, src_h;
  uint8_t* src_images;
  float* src_depths;
  float* poses[4];      /* [S][43] per rotation */
  float ref_K[4][4], ref_inv_K[4][4];
  /* rotating state */
  int rot;
  int W, H; /* current dims */
  uint8_t* ref;
  float *ref_sum, *ref_sqsum;
  float *depth, *normal, *cost, *sel, *prev_sel;
  pmo_rng* rand;
  uint8_t* mask; /* [S][H][W] last sweep only */
  /* texel LUT */
  float lut[256];
} pmo_state;

/* ------------------------------------------------------------------------- */
/* Texture emulation                                                         */
/* ------------------------------------------------------------------------- */

/* ref image: point filter, normalized float, border 0 (BindRefImageTexture,
 * patch_match_cuda.cu:1565-1576) */
static inline float tex_ref(const pmo_state* st, int col, int row) {
  if (col < 0 || row < 0 || col >= st->W || row >= st->H) return 0.0f;
  return st->lut[st->ref[(size_t)row * st->W + col]];
}

static inline float tex_src_point(const pmo_state* st, int s, int ix, int iy) {
  if (ix < 0 || iy < 0 || ix >= st->src_w || iy >= st->src_h) return 0.0f;
  return st->lut[st->src_images[((size_t)s * st->src_h + iy) * st->src_w + ix]];
}

/* SampleLayeredBilinear, patch_match_cuda.cu:426-442 */
static inline float tex_src_bilinear(const pmo_state* st, int s, float x, float y) {
  const float px = x - 0.5f;
  const float py = y - 0.5f;
  const float fx = floorf(px);
  const float fy = floorf(py);
  const float wx = px - fx;
  const float wy = py - fy;
  const int ix = sat_f2i(fx);
  const int iy = sat_f2i(fy);
  /* guard +1 overflow at INT32_MAX */
  const int ix1 = ix == INT32_MAX ? ix : ix + 1;
  const int iy1 = iy == INT32_MAX ? iy : iy + 1;
  const float c00 = tex_src_point(st, s, ix, iy);
  const float c10 = tex_src_point(st, s, ix1, iy);
  const float c01 = tex_src_point(st, s, ix, iy1);
  const float c11 = tex_src_point(st, s, ix1, iy1);
  const float top = fmaf(c10, wx, c00 * (1.0f - wx));
  const float bot = fmaf(c11, wx, c01 * (1.0f - wx));
  return fmaf(bot, wy, top * (1.0f - wy));
}

/* Device-order variant: blend the four raw texels (exact integers in float) and scale
 * the blend once by 1/255 (the float nearest to 1/255, 0x1.010102p-8), instead of
 * normalising each texel first. Same bilinear polynomial, one rounding order. */
static inline float tex_src_raw(const pmo_state* st, int s, int ix, int iy) {
  if (ix < 0 || iy < 0 || ix >= st->src_w || iy >= st->src_h) return 0.0f;
  return (float)st->src_images[((size_t)s * st->src_h + iy) * st->src_w + ix];
}
static inline float tex_src_bilinear_raw(const pmo_state* st, int s, float px, float py) {
  /* (px, py) are texel-space coordinates: the reference's +0.5 (texture centre, :527-528) and the
   * -0.5 of its own four-point emulation (:430-431) cancel and are not evaluated in device order;
   * the blend is written in lerp form */
  const float fx = floorf(px);
  const float fy = floorf(py);
  const float wx = px - fx;
  const float wy = py - fy;
  const int ix = sat_f2i(fx);
  const int iy = sat_f2i(fy);
  const int ix1 = ix == INT32_MAX ? ix : ix + 1;
  const int iy1 = iy == INT32_MAX ? iy : iy + 1;
  const float c00 = tex_src_raw(st, s, ix, iy);
  const float c10 = tex_src_raw(st, s, ix1, iy);
  const float c01 = tex_src_raw(st, s, ix, iy1);
  const float c11 = tex_src_raw(st, s, ix1, iy1);
  const float top = fmaf(wx, c10 - c00, c00);
  const float bot = fmaf(wx, c11 - c01, c01);
  return fmaf(wy, bot - top, top) * 0x1.010102p-8f;
}

/* source depth: point filter, element type, border 0, sampled at (+0.5,+0.5)
 * (patch_match_cuda.cu:635-636, 1677-1690) */
static inline float tex_src_depth(const pmo_state* st, int s, float x, float y) {
  const float fx = floorf(x), fy = floorf(y);
  /* float-domain range test so that NaN/inf coordinates hit the border */
  if (!(fx >= 0.0f && fy >= 0.0f && fx < (float)st->src_w && fy < (float)st->src_h)) return 0.0f;
  return st->src_depths[((size_t)s * st->src_h + (int)fy) * st->src_w + (int)fx];
}

/* ------------------------------------------------------------------------- */
/* Device-function restatements                                              */
/* ------------------------------------------------------------------------- */

static inline float dot3(const float a[3], const float b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* GenerateRandomNormal, patch_match_cuda.cu:94-123 */
static void generate_random_normal(const float inv_K[4], int row, int col, pmo_rng* rng,
                                   float normal[3]) {
  float v1 = 0.0f, v2 = 0.0f, s = 2.0f;
  while (s >= 1.0f) {
    v1 = 2.0f * pmo_rng_uniform(rng) - 1.0f;
    v2 = 2.0f * pmo_rng_uniform(rng) - 1.0f;
    s = v1 * v1 + v2 * v2;
  }
  const float s_norm = sqrtf(1.0f - s);
  normal[0] = 2.0f * v1 * s_norm;
  normal[1] = 2.0f * v2 * s_norm;
  normal[2] = 1.0f - 2.0f * s;
  const float view_ray[3] = {inv_K[0] * col + inv_K[1], inv_K[2] * row + inv_K[3], 1.0f};
  if (dot3(normal, view_ray) > 0) {
    normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2];
  }
}

/* PerturbDepth, patch_match_cuda.cu:125-131 (+GenerateRandomDepth :88-92) */
static float perturb_depth(float perturbation, float depth, pmo_rng* rng) {
  const float depth_min = (1.0f - perturbation) * depth;
  const float depth_max = (1.0f + perturbation) * depth;
  return pmo_rng_uniform(rng) * (depth_max - depth_min) + depth_min;
}

/* PerturbNormal, patch_match_cuda.cu:133-196 (recursion unrolled as a loop) */
static void perturb_normal(const float inv_K[4], int row, int col, float perturbation,
                           const float normal[3], pmo_rng* rng, float out[3]) {
  for (int num_trials = 0;; ++num_trials) {
    const float a1 = (pmo_rng_uniform(rng) - 0.5f) * perturbation;
    const float a2 = (pmo_rng_uniform(rng) - 0.5f) * perturbation;
    const float a3 = (pmo_rng_uniform(rng) - 0.5f) * perturbation;
    float sin_a1, sin_a2, sin_a3, cos_a1, cos_a2, cos_a3;
    pmo_sincos(a1, &sin_a1, &cos_a1);
    pmo_sincos(a2, &sin_a2, &cos_a2);
    pmo_sincos(a3, &sin_a3, &cos_a3);
    float R[9];
    R[0] = cos_a2 * cos_a3;
    R[1] = -cos_a2 * sin_a3;
    R[2] = sin_a2;
    R[3] = cos_a1 * sin_a3 + cos_a3 * sin_a1 * sin_a2;
    R[4] = cos_a1 * cos_a3 - sin_a1 * sin_a2 * sin_a3;
    R[5] = -cos_a2 * sin_a1;
    R[6] = sin_a1 * sin_a3 - cos_a1 * cos_a3 * sin_a2;
    R[7] = cos_a3 * sin_a1 + cos_a1 * sin_a2 * sin_a3;
    R[8] = cos_a1 * cos_a2;
    out[0] = R[0] * normal[0] + R[1] * normal[1] + R[2] * normal[2];
    out[1] = R[3] * normal[0] + R[4] * normal[1] + R[5] * normal[2];
    out[2] = R[6] * normal[0] + R[7] * normal[1] + R[8] * normal[2];
    const float view_ray[3] = {inv_K[0] * col + inv_K[1], inv_K[2] * row + inv_K[3], 1.0f};
    if (dot3(out, view_ray) >= 0.0f) {
      const int kMaxNumTrials = 3;
      if (num_trials < kMaxNumTrials) {
        perturbation = 0.5f * perturbation;
        continue;
      }
      out[0] = normal[0]; out[1] = normal[1]; out[2] = normal[2];
      return;
    }
    const float inv_norm = pm_rsqrt(dot3(out, out));
    out[0] *= inv_norm; out[1] *= inv_norm; out[2] *= inv_norm;
    return;
  }
}

/* ComputePointAtDepth, patch_match_cuda.cu:198-205 */
static inline void point_at_depth(const float inv_K[4], float row, float col, float depth,
                                  float p[3]) {
  p[0] = depth * (inv_K[0] * col + inv_K[1]);
  p[1] = depth * (inv_K[2] * row + inv_K[3]);
  p[2] = depth;
}

/* PropagateDepth, patch_match_cuda.cu:210-236 */
static float propagate_depth(const float inv_K[4], float depth1, const float normal1[3], float row1,
                             float row2) {
  const float x1 = depth1 * (inv_K[2] * row1 + inv_K[3]);
  const float y1 = depth1;
  const float x2 = x1 + normal1[2];
  const float y2 = y1 - normal1[1];
  const float x4 = inv_K[2] * row2 + inv_K[3];
  const float denom = x2 - x1 + x4 * (y1 - y2);
  const float kEps = 1e-5f;
  if (fabsf(denom) < kEps) return depth1;
  const float nom = y1 * x2 - x1 * y2;
  return nom / denom;
}

/* ComputeViewingAngles, patch_match_cuda.cu:241-269 */
static void viewing_angles(const float* pose, const float point[3], const float normal[3],
                           float* cos_tri, float* cos_inc) {
  const float* C = pose + 16;
  const float SX[3] = {C[0] - point[0], C[1] - point[1], C[2] - point[2]};
  const float RX_inv_norm = pm_rsqrt(dot3(point, point));
  const float SX_inv_norm = pm_rsqrt(dot3(SX, SX));
  *cos_inc = dot3(SX, normal) * SX_inv_norm;
  *cos_tri = -dot3(SX, point) * RX_inv_norm * SX_inv_norm;
}

/* ComposeHomography, patch_match_cuda.cu:271-332 */
static void compose_homography(const float inv_K[4], const float* pose, int row, int col,
                               float depth, const float normal[3], float H[9]) {
  const float* K = pose;
  const float* R = pose + 4;
  const float* T = pose + 13;
  const float dist = depth * (normal[0] * (inv_K[0] * col + inv_K[1]) +
                              normal[1] * (inv_K[2] * row + inv_K[3]) + normal[2]);
  const float inv_dist = 1.0f / dist;
  const float inv_dist_N0 = inv_dist * normal[0];
  const float inv_dist_N1 = inv_dist * normal[1];
  const float inv_dist_N2 = inv_dist * normal[2];
  H[0] = inv_K[0] * (K[0] * (R[0] + inv_dist_N0 * T[0]) + K[1] * (R[6] + inv_dist_N0 * T[2]));
  H[1] = inv_K[2] * (K[0] * (R[1] + inv_dist_N1 * T[0]) + K[1] * (R[7] + inv_dist_N1 * T[2]));
  H[2] = K[0] * (R[2] + inv_dist_N2 * T[0]) + K[1] * (R[8] + inv_dist_N2 * T[2]) +
         inv_K[1] * (K[0] * (R[0] + inv_dist_N0 * T[0]) + K[1] * (R[6] + inv_dist_N0 * T[2])) +
         inv_K[3] * (K[0] * (R[1] + inv_dist_N1 * T[0]) + K[1] * (R[7] + inv_dist_N1 * T[2]));
  H[3] = inv_K[0] * (K[2] * (R[3] + inv_dist_N0 * T[1]) + K[3] * (R[6] + inv_dist_N0 * T[2]));
  H[4] = inv_K[2] * (K[2] * (R[4] + inv_dist_N1 * T[1]) + K[3] * (R[7] + inv_dist_N1 * T[2]));
  H[5] = K[2] * (R[5] + inv_dist_N2 * T[1]) + K[3] * (R[8] + inv_dist_N2 * T[2]) +
         inv_K[1] * (K[2] * (R[3] + inv_dist_N0 * T[1]) + K[3] * (R[6] + inv_dist_N0 * T[2])) +
         inv_K[3] * (K[2] * (R[4] + inv_dist_N1 * T[1]) + K[3] * (R[7] + inv_dist_N1 * T[2]));
  H[6] = inv_K[0] * (R[6] + inv_dist_N0 * T[2]);
  H[7] = inv_K[2] * (R[7] + inv_dist_N1 * T[2]);
  H[8] = R[8] + inv_K[1] * (R[6] + inv_dist_N0 * T[2]) + inv_K[3] * (R[7] + inv_dist_N1 * T[2]) +
         inv_dist_N2 * T[2];
}

/* BilateralWeightComputer::Compute, gpu_mat_ref_image.h:70-90 */
static inline float bilateral_weight(float spatial_norm, float color_norm, float row_diff,
                                     float col_diff, float color1, float color2) {
  const float spatial_dist_squared = row_diff * row_diff + col_diff * col_diff;
  const float color_dist = color1 - color2;
  return pmo_exp(-spatial_dist_squared * spatial_norm - color_dist * color_dist * color_norm);
}

typedef struct {
  float spatial_norm, color_norm;
  int radius, step;
} ncc_params;

/* PhotoConsistencyCostComputer::Compute, patch_match_cuda.cu:489-593.
 * `weights`/`refc` (optional, both or neither): the (2r/step+1)^2 bilateral
 * weights and reference colours of the patch centred at (row, col), precomputed
 * once per pixel step -- a pure function of the same inputs, so bit-identical to
 * recomputing them per call as the reference does (:538-539). */
static float ncc_cost(const pmo_state* st, const ncc_params* np, const float inv_K[4],
                      const float* pose, int s, int row, int col, float depth,
                      const float normal[3], float ref_sum, float ref_sqsum,
                      const float* weights, const float* refc) {
  float tform[9];
  compose_homography(inv_K, pose, row, col, depth, normal, tform);
  const int kWindowStep = np->step;
  const int kWindowRadius = np->radius;
  float tform_step[8];
  for (int i = 0; i < 8; ++i) tform_step[i] = kWindowStep * tform[i];

  const int row_start = row - kWindowRadius;
  const int col_start = col - kWindowRadius;
  float col_src = tform[0] * col_start + tform[1] * row_start + tform[2];
  float row_src = tform[3] * col_start + tform[4] * row_start + tform[5];
  float z = tform[6] * col_start + tform[7] * row_start + tform[8];
  float base_col_src = col_src, base_row_src = row_src, base_z = z;

  const float ref_center_color = tex_ref(st, col, row);
  float src_color_sum = 0.0f, src_color_squared_sum = 0.0f, src_ref_color_sum = 0.0f,
        bilateral_weight_sum = 0.0f;
  int tap = 0;
  for (int wr = -kWindowRadius; wr <= kWindowRadius; wr += kWindowStep) {
    for (int wc = -kWindowRadius; wc <= kWindowRadius; wc += kWindowStep, ++tap) {
      const float inv_z = 1.0f / z;
      const float norm_col_src = fmaf(inv_z, col_src, 0.5f);
      const float norm_row_src = fmaf(inv_z, row_src, 0.5f);
      float ref_color, bw;
      if (weights) {
        ref_color = refc[tap];
        bw = weights[tap];
      } else {
        ref_color = tex_ref(st, col + wc, row + wr);
        bw = bilateral_weight(np->spatial_norm, np->color_norm, (float)wr, (float)wc,
                              ref_center_color, ref_color);
      }
      const float src_color = tex_src_bilinear(st, s, norm_col_src, norm_row_src);
      const float bws = bw * src_color;
      src_color_sum += bws;
      src_color_squared_sum = fmaf(bws, src_color, src_color_squared_sum);
      src_ref_color_sum = fmaf(bws, ref_color, src_ref_color_sum);
      bilateral_weight_sum += bw;
      col_src += tform_step[0];
      row_src += tform_step[3];
      z += tform_step[6];
    }
    base_col_src += tform_step[1];
    base_row_src += tform_step[4];
    base_z += tform_step[7];
    col_src = base_col_src;
    row_src = base_row_src;
    z = base_z;
  }
  const float inv_bws = 1.0f / bilateral_weight_sum;
  src_color_sum *= inv_bws;
  src_color_squared_sum *= inv_bws;
  src_ref_color_sum *= inv_bws;
  const float ref_color_var = ref_sqsum - ref_sum * ref_sum;
  const float src_color_var = src_color_squared_sum - src_color_sum * src_color_sum;
  const float kMinVar = 1e-5f;
  const float kMaxCost = 2.0f;
  if (ref_color_var < kMinVar || src_color_var < kMinVar) return kMaxCost;
  const float covar = src_ref_color_sum - ref_sum * src_color_sum;
  const float var = sqrtf(ref_color_var * src_color_var);
  return fmaxf(0.0f, fminf(kMaxCost, 1.0f - covar / var));
}

/* The same cost as ncc_cost() evaluated in the HIP kernel's order ("device order"):
 *  - tap t = wrow * n1d + wcol (t = wcol * n1d + wrow in the odd sweep directions, st->rot & 1)
 *    is dealt to lane j = t % 16 of a 16-lane group; a
 *    lane accumulates its taps t = j + 16 k in increasing k, even k and odd k separately
 *    (the halves of the packed fp32 registers), and adds the two partial sums;
 *  - the warped coordinate of a tap is evaluated directly from the homography whose constant
 *    column was moved to the window origin, fma(H0, dx, fma(H1, dy, C2)) with
 *    C2 = fma(H0, col - r, fma(H1, row - r, H2)) and dx, dy the tap's offset in the window,
 *    instead of the reference's running sums (:554-568); eight taps of a lane share one
 *    division (prefix/suffix products), and the +0.5/-0.5 texel-centre round trip is dropped;
 *  - the bilinear blend is taken over the raw texels and scaled once by 1/255
 *    (tex_src_bilinear_raw) instead of normalising the four texels first;
 *  - the 16 partial sums are combined by the fixed tree the DPP cross-lane adds
 *    implement (row_mirror, row_half_mirror, quad reverse, quad swap):
 *    total = ((q0+q7)+(q3+q4)) + ((q1+q6)+(q2+q5)),  q_l = p_l + p_(15-l).
 * Mathematically identical to :489-593; only the rounding order differs. */
static inline float tree16(const float p[16]) {
  float q[8];
  for (int l = 0; l < 8; ++l) q[l] = p[l] + p[15 - l];
  return ((q[0] + q[7]) + (q[3] + q[4])) + ((q[1] + q[6]) + (q[2] + q[5]));
}

/* Diagnostic only (scripts/pm_order_decomposition.py): PMO_DEVICE_MIX=<bits> reverts ONE ingredient of the device
 * order at a time to the reference's form, to attribute the difference between the two orders --
 *   1  sums accumulated tap by tap in window order (no lane dealing, no even/odd halves, no tree)
 *   2  warped coordinates by the reference's running sums (:554-568) instead of fma(H0, dx, fma(H1, dy, C2))
 *   4  one division per tap instead of the shared one of eight taps
 *   8  the reference's bilinear sample (texel-centre round trip, texels normalised first)
 * 15 = ncc_cost() with memoised weights, bit for bit. 16 = every intermediate in double (ncc_cost_double). 0 / unset = the
 * device order. */
static int g_device_mix = -1;
static int device_mix(void) {
  if (g_device_mix < 0) {
    const char* e = getenv("PMO_DEVICE_MIX");
    g_device_mix = e && *e ? atoi(e) & 31 : 0;
  }
  return g_device_mix;
}
PMO_API void pmo_set_device_mix(int bits) { g_device_mix = bits & 31; }

/* 16: the same cost from the same float inputs (homography entries, weights, reference colours, patch sums) with every
 * intermediate in double -- the yardstick both float orders are measured against */
static float ncc_cost_double(const pmo_state* st, const ncc_params* np, const float tf[9], int s, int row, int col,
                             float ref_sum, float ref_sqsum, const float* weights, const float* refc) {
  const int n1d = (2 * np->radius) / np->step + 1;
  double sum = 0.0, sq = 0.0, sref = 0.0, wsum = 0.0;
  for (int wr = 0, pos = 0; wr < n1d; ++wr)
    for (int wc = 0; wc < n1d; ++wc, ++pos) {
      const double x = (double)(col - np->radius + wc * np->step), y = (double)(row - np->radius + wr * np->step);
      const double z = (double)tf[6] * x + (double)tf[7] * y + (double)tf[8];
      const double px = ((double)tf[0] * x + (double)tf[1] * y + (double)tf[2]) / z;
      const double py = ((double)tf[3] * x + (double)tf[4] * y + (double)tf[5]) / z;
      double c = 0.0;
      if (px == px && py == py && fabs(px) < 1e9 && fabs(py) < 1e9) {
        const double fx = floor(px), fy = floor(py), wx = px - fx, wy = py - fy;
        const int ix = (int)fx, iy = (int)fy;
        const double c00 = tex_src_raw(st, s, ix, iy), c10 = tex_src_raw(st, s, ix + 1, iy);
        const double c01 = tex_src_raw(st, s, ix, iy + 1), c11 = tex_src_raw(st, s, ix + 1, iy + 1);
        const double top = c00 + wx * (c10 - c00), bot = c01 + wx * (c11 - c01);
        c = (top + wy * (bot - top)) / 255.0;
      }
      const double bw = weights[pos];
      sum += bw * c; sq += bw * c * c; sref += bw * c * (double)refc[pos]; wsum += bw;
    }
  sum /= wsum; sq /= wsum; sref /= wsum;
  const double rvar = (double)ref_sqsum - (double)ref_sum * (double)ref_sum, svar = sq - sum * sum;
  if (rvar < 1e-5 || svar < 1e-5) return 2.0f;
  const double cost = 1.0 - (sref - (double)ref_sum * sum) / sqrt(rvar * svar);
  return (float)fmax(0.0, fmin(2.0, cost));
}

static float ncc_cost_device_mixed(const pmo_state* st, const ncc_params* np, const float tf[9], int s, int row, int col,
                                   float ref_sum, float ref_sqsum, const float* weights, const float* refc, int mix) {
  if (mix & 16) return ncc_cost_double(st, np, tf, s, row, col, ref_sum, ref_sqsum, weights, refc);
  const int n1d = (2 * np->radius) / np->step + 1;
  const int ntaps = n1d * n1d;
  const int transpose = st->rot & 1;
  const float x0f = (float)(col - np->radius), y0f = (float)(row - np->radius);
  const float c2 = fmaf(tf[0], x0f, fmaf(tf[1], y0f, tf[2]));
  const float c5 = fmaf(tf[3], x0f, fmaf(tf[4], y0f, tf[5]));
  const float c8 = fmaf(tf[6], x0f, fmaf(tf[7], y0f, tf[8]));
  /* per tap (window position pos = wrow * n1d + wcol): warped coordinates and divisor */
  float csrc[1024], rsrc[1024], zz[1024], inv[1024];
  if (ntaps > 1024) return 2.0f;
  if (mix & 2) {  /* the reference's stepping */
    float tform_step[8];
    for (int i = 0; i < 8; ++i) tform_step[i] = np->step * tf[i];
    const int row_start = row - np->radius, col_start = col - np->radius;
    float col_src = tf[0] * col_start + tf[1] * row_start + tf[2];
    float row_src = tf[3] * col_start + tf[4] * row_start + tf[5];
    float z = tf[6] * col_start + tf[7] * row_start + tf[8];
    float bc = col_src, br = row_src, bz = z;
    for (int wr = 0, pos = 0; wr < n1d; ++wr) {
      for (int wc = 0; wc < n1d; ++wc, ++pos) {
        csrc[pos] = col_src; rsrc[pos] = row_src; zz[pos] = z;
        col_src += tform_step[0]; row_src += tform_step[3]; z += tform_step[6];
      }
      bc += tform_step[1]; br += tform_step[4]; bz += tform_step[7];
      col_src = bc; row_src = br; z = bz;
    }
  } else {
    for (int wr = 0, pos = 0; wr < n1d; ++wr)
      for (int wc = 0; wc < n1d; ++wc, ++pos) {
        const float dx = (float)(wc * np->step), dy = (float)(wr * np->step);
        csrc[pos] = fmaf(tf[0], dx, fmaf(tf[1], dy, c2));
        rsrc[pos] = fmaf(tf[3], dx, fmaf(tf[4], dy, c5));
        zz[pos] = fmaf(tf[6], dx, fmaf(tf[7], dy, c8));
      }
  }
  /* which window position lane j holds as its k-th tap (the device's dealing) */
  #define PMO_POS_OF(t) (transpose ? ((t) % n1d) * n1d + (t) / n1d : (t))
  if (mix & 4) {
    for (int pos = 0; pos < ntaps; ++pos) inv[pos] = 1.0f / zz[pos];
  } else {
    const int nchunk = (ntaps + 127) / 128;
    for (int j = 0; j < 16; ++j)
      for (int cb = 0; cb < nchunk; ++cb) {
        float z8[8], pre[8], run = 1.0f;
        int p8[8];
        for (int k = 0; k < 8; ++k) {
          const int t = j + 16 * (8 * cb + k);
          p8[k] = t < ntaps ? PMO_POS_OF(t) : -1;
          z8[k] = p8[k] >= 0 ? zz[p8[k]] : 1.0f;
          pre[k] = run;
          run = run * z8[k];
        }
        const float rinv = 1.0f / run;
        float suf = 1.0f;
        for (int k = 7; k >= 0; --k) {
          if (p8[k] >= 0) inv[p8[k]] = (pre[k] * suf) * rinv;
          suf = suf * z8[k];
        }
      }
  }
  float color[1024];
  for (int pos = 0; pos < ntaps; ++pos)
    color[pos] = (mix & 8) ? tex_src_bilinear(st, s, fmaf(inv[pos], csrc[pos], 0.5f), fmaf(inv[pos], rsrc[pos], 0.5f))
                           : tex_src_bilinear_raw(st, s, inv[pos] * csrc[pos], inv[pos] * rsrc[pos]);
  float src_color_sum, src_color_squared_sum, src_ref_color_sum, bilateral_weight_sum;
  if (mix & 1) {
    src_color_sum = src_color_squared_sum = src_ref_color_sum = bilateral_weight_sum = 0.0f;
    for (int pos = 0; pos < ntaps; ++pos) {
      const float bw = weights[pos], bws = bw * color[pos];
      src_color_sum += bws;
      src_color_squared_sum = fmaf(bws, color[pos], src_color_squared_sum);
      src_ref_color_sum = fmaf(bws, refc[pos], src_ref_color_sum);
      bilateral_weight_sum += bw;
    }
  } else {
    float a_sum[16], a_sq[16], a_ref[16], a_w[16];
    for (int j = 0; j < 16; ++j) {
      float e_sum[2] = {0.0f, 0.0f}, e_sq[2] = {0.0f, 0.0f}, e_ref[2] = {0.0f, 0.0f};
      a_w[j] = 0.0f;
      for (int kk = 0; j + 16 * kk < ntaps; ++kk) {
        const int pos = PMO_POS_OF(j + 16 * kk);
        const float bw = weights[pos], bws = bw * color[pos];
        e_sum[kk & 1] += bws;
        e_sq[kk & 1] = fmaf(bws, color[pos], e_sq[kk & 1]);
        e_ref[kk & 1] = fmaf(bws, refc[pos], e_ref[kk & 1]);
        a_w[j] += bw;
      }
      a_sum[j] = e_sum[0] + e_sum[1];
      a_sq[j] = e_sq[0] + e_sq[1];
      a_ref[j] = e_ref[0] + e_ref[1];
    }
    src_color_sum = tree16(a_sum);
    src_color_squared_sum = tree16(a_sq);
    src_ref_color_sum = tree16(a_ref);
    bilateral_weight_sum = tree16(a_w);
  }
  #undef PMO_POS_OF
  const float inv_bws = 1.0f / bilateral_weight_sum;
  src_color_sum *= inv_bws;
  src_color_squared_sum *= inv_bws;
  src_ref_color_sum *= inv_bws;
  const float ref_color_var = ref_sqsum - ref_sum * ref_sum;
  const float src_color_var = src_color_squared_sum - src_color_sum * src_color_sum;
  if (ref_color_var < 1e-5f || src_color_var < 1e-5f) return 2.0f;
  const float covar = src_ref_color_sum - ref_sum * src_color_sum;
  const float var = sqrtf(ref_color_var * src_color_var);
  return fmaxf(0.0f, fminf(2.0f, 1.0f - covar / var));
}

static float ncc_cost_device(const pmo_state* st, const ncc_params* np, const float inv_K[4],
                             const float* pose, int s, int row, int col, float depth,
                             const float normal[3], float ref_sum, float ref_sqsum,
                             const float* weights, const float* refc) {
  float tf[9];
  compose_homography(inv_K, pose, row, col, depth, normal, tf);
  if (device_mix() != 0) return ncc_cost_device_mixed(st, np, tf, s, row, col, ref_sum, ref_sqsum, weights, refc, device_mix());
  const int n1d = (2 * np->radius) / np->step + 1;
  const int ntaps = n1d * n1d;
  /* In the odd sweep directions (the buffers are rotated by 90 or 270 degrees) the device deals the
   * taps to the lanes column-major -- tap t sits at window (row, col) = (t % n1d, t / n1d) -- so that
   * consecutive lanes still walk along a row of the (never rotated) source image: the sums are the
   * same sums, grouped differently. */
  const int transpose = st->rot & 1;
  /* device order: the constant column of the homography is moved to the window origin once per
   * evaluation, taps address the patch by small non-negative offsets */
  const float x0f = (float)(col - np->radius), y0f = (float)(row - np->radius);
  const float c2 = fmaf(tf[0], x0f, fmaf(tf[1], y0f, tf[2]));
  const float c5 = fmaf(tf[3], x0f, fmaf(tf[4], y0f, tf[5]));
  const float c8 = fmaf(tf[6], x0f, fmaf(tf[7], y0f, tf[8]));
  float a_sum[16], a_sq[16], a_ref[16], a_w[16];
  /* lane j owns taps t = j + 16 k; they are processed in chunks of eight k: the eight projective
   * divisors share one correctly rounded division,
   *   inv_k = (prefix_k * suffix_k) * (1 / prod z),  prefix in increasing k, suffix in decreasing k,
   * taps beyond the window contribute z = 1 (they are not sampled). Even-k and odd-k taps (the
   * two halves of the device's packed registers) accumulate separately and are added per lane. */
  const int nchunk = (ntaps + 127) / 128;
  for (int j = 0; j < 16; ++j) {
    float e_sum[2] = {0.0f, 0.0f}, e_sq[2] = {0.0f, 0.0f}, e_ref[2] = {0.0f, 0.0f};
    a_w[j] = 0.0f;
    for (int cb = 0; cb < nchunk; ++cb) {
      float csrc[8], rsrc[8], zz[8], pre[8], inv[8];
      int pos[8];
      float run = 1.0f;
      for (int k = 0; k < 8; ++k) {
        const int t = j + 16 * (8 * cb + k);
        const int valid = t < ntaps;
        const int tt = valid ? t : 0;
        int wrow = tt / n1d, wcol = tt - wrow * n1d;
        if (transpose) { const int sw = wrow; wrow = wcol; wcol = sw; }
        pos[k] = wrow * n1d + wcol;
        const float dx = (float)(wcol * np->step);
        const float dy = (float)(wrow * np->step);
        csrc[k] = fmaf(tf[0], dx, fmaf(tf[1], dy, c2));
        rsrc[k] = fmaf(tf[3], dx, fmaf(tf[4], dy, c5));
        zz[k] = valid ? fmaf(tf[6], dx, fmaf(tf[7], dy, c8)) : 1.0f;
        pre[k] = run;
        run = run * zz[k];
      }
      const float rinv = 1.0f / run;
      float suf = 1.0f;
      for (int k = 7; k >= 0; --k) {
        inv[k] = (pre[k] * suf) * rinv;
        suf = suf * zz[k];
      }
      for (int k = 0; k < 8; ++k) {
        const int t = j + 16 * (8 * cb + k);
        if (t >= ntaps) continue;
        const float src_color = tex_src_bilinear_raw(st, s, inv[k] * csrc[k], inv[k] * rsrc[k]);
        const float bw = weights[pos[k]];
        const float bws = bw * src_color;
        e_sum[k & 1] += bws;
        e_sq[k & 1] = fmaf(bws, src_color, e_sq[k & 1]);
        e_ref[k & 1] = fmaf(bws, refc[pos[k]], e_ref[k & 1]);
        a_w[j] += bw;
      }
    }
    a_sum[j] = e_sum[0] + e_sum[1];
    a_sq[j] = e_sq[0] + e_sq[1];
    a_ref[j] = e_ref[0] + e_ref[1];
  }
  float src_color_sum = tree16(a_sum);
  float src_color_squared_sum = tree16(a_sq);
  float src_ref_color_sum = tree16(a_ref);
  const float bilateral_weight_sum = tree16(a_w);
  const float inv_bws = 1.0f / bilateral_weight_sum;
  src_color_sum *= inv_bws;
  src_color_squared_sum *= inv_bws;
  src_ref_color_sum *= inv_bws;
  const float ref_color_var = ref_sqsum - ref_sum * ref_sum;
  const float src_color_var = src_color_squared_sum - src_color_sum * src_color_sum;
  const float kMinVar = 1e-5f;
  const float kMaxCost = 2.0f;
  if (ref_color_var < kMinVar || src_color_var < kMinVar) return kMaxCost;
  const float covar = src_ref_color_sum - ref_sum * src_color_sum;
  const float var = sqrtf(ref_color_var * src_color_var);
  return fmaxf(0.0f, fminf(kMaxCost, 1.0f - covar / var));
}

/* order dispatch */
static float ncc_any(const pmo_options* opt, const pmo_state* st, const ncc_params* np,
                     const float inv_K[4], const float* pose, int s, int row, int col, float depth,
                     const float normal[3], float ref_sum, float ref_sqsum, const float* weights,
                     const float* refc) {
  if (opt->order == 1)
    return ncc_cost_device(st, np, inv_K, pose, s, row, col, depth, normal, ref_sum, ref_sqsum,
                           weights, refc);
  return ncc_cost(st, np, inv_K, pose, s, row, col, depth, normal, ref_sum, ref_sqsum,
                  opt->memoize ? weights : NULL, opt->memoize ? refc : NULL);
}

/* ComputeGeomConsistencyCost, patch_match_cuda.cu:601-667 */
static float geom_cost(const pmo_state* st, const float K4[4], const float inv_K[4],
                       const float* pose, int s, float row, float col, float depth,
                       float max_cost) {
  const float* P = pose + 19;
  const float* inv_P = pose + 31;
  float fp[3];
  point_at_depth(inv_K, row, col, depth, fp);
  const float inv_forward_z = 1.0f / (P[8] * fp[0] + P[9] * fp[1] + P[10] * fp[2] + P[11]);
  float src_col = inv_forward_z * (P[0] * fp[0] + P[1] * fp[1] + P[2] * fp[2] + P[3]);
  float src_row = inv_forward_z * (P[4] * fp[0] + P[5] * fp[1] + P[6] * fp[2] + P[7]);
  const float src_depth = tex_src_depth(st, s, src_col + 0.5f, src_row + 0.5f);
  if (src_depth == 0.0f) return max_cost;
  src_col *= src_depth;
  src_row *= src_depth;
  const float bx = inv_P[0] * src_col + inv_P[1] * src_row + inv_P[2] * src_depth + inv_P[3];
  const float by = inv_P[4] * src_col + inv_P[5] * src_row + inv_P[6] * src_depth + inv_P[7];
  const float bz = inv_P[8] * src_col + inv_P[9] * src_row + inv_P[10] * src_depth + inv_P[11];
  const float inv_bz = 1.0f / bz;
  const float backward_col = inv_bz * (K4[0] * bx + K4[1] * bz);
  const float backward_row = inv_bz * (K4[2] * by + K4[3] * bz);
  const float diff_col = col - backward_col;
  const float diff_row = row - backward_row;
  return fminf(max_cost, sqrtf(diff_col * diff_col + diff_row * diff_row));
}

/* LikelihoodComputer, patch_match_cuda.cu:698-832 */
typedef struct {
  float cos_min_tri, inv_inc_sigma_sq, inv_ncc_sigma_sq, ncc_norm;
} likelihood;

static void likelihood_init(likelihood* L, float ncc_sigma, float min_tri_angle,
                            float inc_sigma) {
  L->cos_min_tri = cosf(min_tri_angle);
  L->inv_inc_sigma_sq = -0.5f / (inc_sigma * inc_sigma);
  L->inv_ncc_sigma_sq = -0.5f / (ncc_sigma * ncc_sigma);
  /* :796-802, evaluated in double like the source expression */
  L->ncc_norm = (float)(2.0f / (sqrt(2.0f * M_PI) * ncc_sigma *
                                erff(2.0f / (ncc_sigma * 1.414213562f))));
}

static inline float ncc_prob(const likelihood* L, float cost) {
  return pmo_exp(cost * cost * L->inv_ncc_sigma_sq) * L->ncc_norm;
}

static inline float message(const likelihood* L, int forward, float cost, float prev) {
  const float kUniformProb = 0.5f;
  const float kNoChangeProb = 0.99999f;
  const float kChangeProb = 1.0f - kNoChangeProb;
  const float emission = ncc_prob(L, cost);
  float zn0, zn1;
  if (forward) {
    zn0 = (prev * kChangeProb + (1.0f - prev) * kNoChangeProb) * kUniformProb;
    zn1 = (prev * kNoChangeProb + (1.0f - prev) * kChangeProb) * emission;
  } else {
    zn0 = prev * emission * kChangeProb + (1.0f - prev) * kUniformProb * kNoChangeProb;
    zn1 = prev * emission * kNoChangeProb + (1.0f - prev) * kUniformProb * kChangeProb;
  }
  return zn1 / (zn0 + zn1);
}

static inline float sel_prob_fn(float alpha, float beta, float prev, float prev_weight) {
  const float zn0 = (1.0f - alpha) * (1.0f - beta);
  const float zn1 = alpha * beta;
  const float curr = zn1 / (zn0 + zn1);
  return prev_weight * prev + (1.0f - prev_weight) * curr;
}

static inline float tri_prob(const likelihood* L, float cos_tri) {
  if (cos_tri > L->cos_min_tri) {
    const float scaled = 1.0f - (1.0f - cos_tri) / (1.0f - L->cos_min_tri);
    const float lik = 1.0f - scaled * scaled;
    return fminf(1.0f, fmaxf(0.0f, lik));
  }
  return 1.0f;
}

static inline float inc_prob(const likelihood* L, float cos_inc) {
  const float x = 1.0f - fmaxf(0.0f, cos_inc);
  return pmo_exp(x * x * L->inv_inc_sigma_sq);
}

static inline void h_apply(const float H[9], const float v[2], float r[2]) {
  const float inv_z = 1.0f / (H[6] * v[0] + H[7] * v[1] + H[8]);
  r[0] = inv_z * (H[0] * v[0] + H[1] * v[1] + H[2]);
  r[1] = inv_z * (H[3] * v[0] + H[4] * v[1] + H[5]);
}

/* ComputeResolutionProb, patch_match_cuda.cu:759-791 */
static float res_prob(const float H[9], float row, float col, int window_size) {
  const int kWindowRadius = window_size / 2;
  float s1[2], s2[2], s3[2], s4[2];
  const float r1[2] = {col - kWindowRadius, row - kWindowRadius};
  const float r2[2] = {col - kWindowRadius, row + kWindowRadius};
  const float r3[2] = {col + kWindowRadius, row + kWindowRadius};
  const float r4[2] = {col + kWindowRadius, row - kWindowRadius};
  h_apply(H, r1, s1); h_apply(H, r2, s2); h_apply(H, r3, s3); h_apply(H, r4, s4);
  const float ref_area = (float)(window_size * window_size);
  const float src_area =
      fabsf(0.5f * (s1[0] * s2[1] - s2[0] * s1[1] - s1[0] * s4[1] + s2[0] * s3[1] -
                    s3[0] * s2[1] + s4[0] * s1[1] + s3[0] * s4[1] - s4[0] * s3[1]));
  if (ref_area > src_area) return src_area / ref_area;
  return ref_area / src_area;
}

/* ------------------------------------------------------------------------- */
/* Sweep options (SweepOptions, patch_match_cuda.cu:914-931)                 */
/* ------------------------------------------------------------------------- */
typedef struct {
  float perturbation, depth_min, depth_max;
  int num_samples;
  float sigma_spatial, sigma_color, ncc_sigma, min_triangulation_angle, incident_angle_sigma,
      prev_sel_prob_weight, geom_consistency_regularizer, geom_consistency_max_cost,
      filter_min_ncc, filter_min_triangulation_angle;
  int filter_min_num_consistent;
  float filter_geom_consistency_max_cost;
} sweep_options;

static inline size_t idx3(const pmo_state* st, int s, int row, int col) {
  return ((size_t)s * st->H + row) * st->W + col;
}

static int num_taps_1d(int radius, int step) { return (2 * radius) / step + 1; }

/* Gather the reference patch colours and bilateral weights around (row, col). */
static void patch_weights(const pmo_state* st, const ncc_params* np, int row, int col,
                          float* weights, float* refc) {
  const float center = tex_ref(st, col, row);
  int tap = 0;
  for (int wr = -np->radius; wr <= np->radius; wr += np->step)
    for (int wc = -np->radius; wc <= np->radius; wc += np->step, ++tap) {
      const float c = tex_ref(st, col + wc, row + wr);
      refc[tap] = c;
      weights[tap] = bilateral_weight(np->spatial_norm, np->color_norm, (float)wr, (float)wc,
                                      center, c);
    }
}

/* ComputeInitialCost, patch_match_cuda.cu:863-912 */
static void compute_initial_cost(pmo_state* st, const pmo_options* opt) {
  ncc_params np;
  const float ss = (float)opt->sigma_spatial, sc = (float)opt->sigma_color;
  np.spatial_norm = 1.0f / (2.0f * ss * ss);
  np.color_norm = 1.0f / (2.0f * sc * sc);
  np.radius = opt->window_radius;
  np.step = opt->window_step;
  const int nt = num_taps_1d(np.radius, np.step);
  const float* inv_K = st->ref_inv_K[st->rot];
#pragma omp parallel
  {
    float* w = (float*)malloc(sizeof(float) * nt * nt * 2);
    float* rc = w + nt * nt;
#pragma omp for schedule(dynamic, 4)
    for (int col = 0; col < st->W; ++col) {
      for (int row = 0; row < st->H; ++row) {
        const float depth = st->depth[(size_t)row * st->W + col];
        float normal[3];
        for (int k = 0; k < 3; ++k) normal[k] = st->normal[idx3(st, k, row, col)];
        const float rs = st->ref_sum[(size_t)row * st->W + col];
        const float rss = st->ref_sqsum[(size_t)row * st->W + col];
        if (opt->memoize || opt->order == 1) patch_weights(st, &np, row, col, w, rc);
        for (int s = 0; s < st->S; ++s) {
          st->cost[idx3(st, s, row, col)] =
              ncc_any(opt, st, &np, inv_K, st->poses[st->rot] + s * PMO_POSE_STRIDE, s, row, col,
                      depth, normal, rs, rss, w, rc);
        }
      }
    }
    free(w);
  }
}

/* SweepFromTopToBottom for one column, patch_match_cuda.cu:933-1288 */
static void sweep_column(pmo_state* st, const pmo_options* opt, const sweep_options* so,
                         int geom_term, int filter_photo, int filter_geom, int col,
                         float* scratch) {
  const int S = st->S, H = st->H;
  const float* inv_K = st->ref_inv_K[st->rot];
  const float* K4 = st->ref_K[st->rot];
  const float* poses = st->poses[st->rot];
  const float kUniformProb = 0.5f;
  likelihood L;
  likelihood_init(&L, so->ncc_sigma, so->min_triangulation_angle, so->incident_angle_sigma);

  ncc_params np;
  np.spatial_norm = 1.0f / (2.0f * so->sigma_spatial * so->sigma_spatial);
  np.color_norm = 1.0f / (2.0f * so->sigma_color * so->sigma_color);
  np.radius = opt->window_radius;
  np.step = opt->window_step;
  const int nt = num_taps_1d(np.radius, np.step);
  const int window_size = 2 * np.radius + 1;

  float* forward_message = scratch;            /* [S] */
  float* sampling_probs = scratch + S;         /* [S] */
  float* memo = scratch + 2 * S;               /* [5][S] */
  float* w = memo + 5 * S;                     /* [nt*nt] */
  float* rc = w + nt * nt;                     /* [nt*nt] */
  unsigned char* memo_valid = (unsigned char*)(rc + nt * nt); /* [5*S] */

  /* backward messages (:976-989) */
  for (int s = 0; s < S; ++s) {
    float beta = kUniformProb;
    for (int row = H - 1; row >= 0; --row) {
      const float cost = st->cost[idx3(st, s, row, col)];
      beta = message(&L, 0, cost, beta);
      st->sel[idx3(st, s, row, col)] = beta;
    }
    forward_message[s] = kUniformProb;
  }

  /* :1022-1028 */
  pmo_rng rng = st->rand[col]; /* row 0 */
  float prev_depth = st->depth[col];
  float prev_normal[3] = {st->normal[idx3(st, 0, 0, col)], st->normal[idx3(st, 1, 0, col)],
                          st->normal[idx3(st, 2, 0, col)]};

  for (int row = 0; row < H; ++row) {
    const float ref_sum = st->ref_sum[(size_t)row * st->W + col];
    const float ref_sqsum = st->ref_sqsum[(size_t)row * st->W + col];
    if (opt->memoize || opt->order == 1) patch_weights(st, &np, row, col, w, rc);
    if (opt->memoize) memset(memo_valid, 0, 5 * S);

    /* :1047-1048 */
    prev_depth = propagate_depth(inv_K, prev_depth, prev_normal, (float)(row - 1), (float)row);

    /* :1051-1052 */
    const float curr_depth = st->depth[(size_t)row * st->W + col];
    float curr_normal[3] = {st->normal[idx3(st, 0, row, col)], st->normal[idx3(st, 1, row, col)],
                            st->normal[idx3(st, 2, row, col)]};

    /* :1055-1062 */
    const float rand_depth = perturb_depth(so->perturbation, curr_depth, &rng);
    float rand_normal[3];
    perturb_normal(inv_K, row, col, (float)(so->perturbation * M_PI), curr_normal, &rng,
                   rand_normal);

    /* :1067-1104 */
    float point[3];
    point_at_depth(inv_K, (float)row, (float)col, curr_depth, point);
    for (int s = 0; s < S; ++s) {
      const float* pose = poses + s * PMO_POSE_STRIDE;
      const float cost = st->cost[idx3(st, s, row, col)];
      const float alpha = message(&L, 1, cost, forward_message[s]);
      const float beta = st->sel[idx3(st, s, row, col)];
      const float prev_prob = st->prev_sel[idx3(st, s, row, col)];
      const float sp = sel_prob_fn(alpha, beta, prev_prob, so->prev_sel_prob_weight);
      float cos_tri, cos_inc;
      viewing_angles(pose, point, curr_normal, &cos_tri, &cos_inc);
      const float tp = tri_prob(&L, cos_tri);
      const float ip = inc_prob(&L, cos_inc);
      float Hm[9];
      compose_homography(inv_K, pose, row, col, curr_depth, curr_normal, Hm);
      const float rp = res_prob(Hm, (float)row, (float)col, window_size);
      sampling_probs[s] = sp * tp * ip * rp;
    }

    /* TransformPDFToCDF :683-696 */
    {
      float prob_sum = 0.0f;
      for (int i = 0; i < S; ++i) prob_sum += sampling_probs[i];
      const float inv_prob_sum = 1.0f / prob_sum;
      float cum = 0.0f;
      for (int i = 0; i < S; ++i) {
        const float prob = sampling_probs[i] * inv_prob_sum;
        cum += prob;
        sampling_probs[i] = cum;
      }
    }

    /* :1115-1126 */
    float costs[5] = {0, 0, 0, 0, 0};
    const float depths[5] = {curr_depth, prev_depth, rand_depth, curr_depth, rand_depth};
    const float* normals[5] = {curr_normal, prev_normal, rand_normal, rand_normal, curr_normal};

    /* :1128-1173 */
    for (int sample = 0; sample < so->num_samples; ++sample) {
      const float rand_prob = pmo_rng_uniform(&rng) - FLT_EPSILON;
      int src = -1;
      for (int s = 0; s < S; ++s) {
        if (sampling_probs[s] > rand_prob) { src = s; break; }
      }
      if (src == -1) continue;
      const float* pose = poses + src * PMO_POSE_STRIDE;
      costs[0] += st->cost[idx3(st, src, row, col)];
      if (geom_term)
        costs[0] += so->geom_consistency_regularizer *
                    geom_cost(st, K4, inv_K, pose, src, (float)row, (float)col, depths[0],
                              so->geom_consistency_max_cost);
      for (int i = 1; i < 5; ++i) {
        float c;
        if (opt->memoize && memo_valid[i * S + src]) {
          c = memo[i * S + src];
        } else {
          c = ncc_any(opt, st, &np, inv_K, pose, src, row, col, depths[i], normals[i], ref_sum,
                      ref_sqsum, w, rc);
          if (opt->memoize) { memo[i * S + src] = c; memo_valid[i * S + src] = 1; }
        }
        costs[i] += c;
        if (geom_term)
          costs[i] += so->geom_consistency_regularizer *
                      geom_cost(st, K4, inv_K, pose, src, (float)row, (float)col, depths[i],
                                so->geom_consistency_max_cost);
      }
    }

    /* FindMinCost :670-681 (ties -> highest index) */
    int min_idx = 0;
    {
      float min_cost = costs[0];
      for (int i = 1; i < 5; ++i)
        if (costs[i] <= min_cost) { min_cost = costs[i]; min_idx = i; }
    }
    const float best_depth = depths[min_idx];
    const float best_normal[3] = {normals[min_idx][0], normals[min_idx][1], normals[min_idx][2]};

    /* :1181-1182 */
    st->depth[(size_t)row * st->W + col] = best_depth;
    for (int k = 0; k < 3; ++k) st->normal[idx3(st, k, row, col)] = best_normal[k];

    /* :1186-1207 */
    for (int s = 0; s < S; ++s) {
      float cost;
      if (min_idx == 0) {
        cost = st->cost[idx3(st, s, row, col)];
      } else {
        if (opt->memoize && memo_valid[min_idx * S + s]) {
          cost = memo[min_idx * S + s];
        } else {
          cost = ncc_any(opt, st, &np, inv_K, poses + s * PMO_POSE_STRIDE, s, row, col, best_depth,
                         best_normal, ref_sum, ref_sqsum, w, rc);
        }
        st->cost[idx3(st, s, row, col)] = cost;
      }
      const float alpha = message(&L, 1, cost, forward_message[s]);
      const float beta = st->sel[idx3(st, s, row, col)];
      const float prev_prob = st->prev_sel[idx3(st, s, row, col)];
      const float prob = sel_prob_fn(alpha, beta, prev_prob, so->prev_sel_prob_weight);
      forward_message[s] = alpha;
      st->sel[idx3(st, s, row, col)] = prob;
    }

    /* :1209-1276 */
    if (filter_photo || filter_geom) {
      int num_consistent = 0;
      float best_point[3];
      point_at_depth(inv_K, (float)row, (float)col, best_depth, best_point);
      const float min_ncc_prob = ncc_prob(&L, 1.0f - so->filter_min_ncc);
      const float cos_min_tri = cosf(so->filter_min_triangulation_angle);
      for (int s = 0; s < S; ++s) {
        const float* pose = poses + s * PMO_POSE_STRIDE;
        float cos_tri, cos_inc;
        viewing_angles(pose, best_point, best_normal, &cos_tri, &cos_inc);
        if (cos_tri > cos_min_tri || cos_inc <= 0.0f) continue;
        int ok;
        if (!filter_geom) {
          ok = st->sel[idx3(st, s, row, col)] >= min_ncc_prob;
        } else if (!filter_photo) {
          ok = geom_cost(st, K4, inv_K, pose, s, (float)row, (float)col, best_depth,
                         so->geom_consistency_max_cost) <= so->filter_geom_consistency_max_cost;
        } else {
          ok = st->sel[idx3(st, s, row, col)] >= min_ncc_prob &&
               geom_cost(st, K4, inv_K, pose, s, (float)row, (float)col, best_depth,
                         so->geom_consistency_max_cost) <= so->filter_geom_consistency_max_cost;
        }
        if (ok) {
          st->mask[idx3(st, s, row, col)] = 1;
          num_consistent += 1;
        }
      }
      if (num_consistent < so->filter_min_num_consistent) {
        st->depth[(size_t)row * st->W + col] = 0.0f;
        for (int k = 0; k < 3; ++k) st->normal[idx3(st, k, row, col)] = 0.0f;
        for (int s = 0; s < S; ++s) st->mask[idx3(st, s, row, col)] = 0;
      }
    }

    /* :1279-1282 */
    prev_depth = best_depth;
    prev_normal[0] = best_normal[0]; prev_normal[1] = best_normal[1]; prev_normal[2] = best_normal[2];
  }
  st->rand[col] = rng; /* :1285-1287 */
}

/* ------------------------------------------------------------------------- */
/* Rotation (cuda_rotate.h:57-75; Rotate patch_match_cuda.cu:1859-1939)      */
/* ------------------------------------------------------------------------- */

#define DEFINE_ROTATE(NAME, T)                                                        \
  static T* NAME(const T* in, int W, int H, int D) {                                  \
    T* out = (T*)malloc(sizeof(T) * (size_t)W * H * D);                               \
    for (int d = 0; d < D; ++d)                                                       \
      for (int y = 0; y < H; ++y)                                                     \
        for (int x = 0; x < W; ++x)                                                   \
          out[((size_t)d * W + (W - 1 - x)) * H + y] = in[((size_t)d * H + y) * W + x]; \
    return out;                                                                       \
  }
DEFINE_ROTATE(rotate_f32, float)
DEFINE_ROTATE(rotate_u8, uint8_t)
DEFINE_ROTATE(rotate_rng, pmo_rng)

/* exported for the gpu_mat_test.cu known-answer check */
PMO_API void pmo_rotate_f32(const float* in, int W, int H, int D, float* out) {
  float* r = rotate_f32(in, W, H, D);
  memcpy(out, r, sizeof(float) * (size_t)W * H * D);
  free(r);
}

static void rotate_state(pmo_state* st) {
  const int W = st->W, H = st->H, S = st->S;
#define ROT(field, fn, D) do { void* n = fn(st->field, W, H, D); free(st->field); st->field = n; } while (0)
  ROT(rand, rotate_rng, 1);
  ROT(depth, rotate_f32, 1);
  /* RotateNormalMap :849-861 then rotate */
  for (size_t i = 0; i < (size_t)W * H; ++i) {
    const float nx = st->normal[i], ny = st->normal[(size_t)W * H + i];
    st->normal[i] = ny;
    st->normal[(size_t)W * H + i] = -nx;
  }
  ROT(normal, rotate_f32, 3);
  ROT(ref, rotate_u8, 1);
  ROT(ref_sum, rotate_f32, 1);
  ROT(ref_sqsum, rotate_f32, 1);
  /* prev_sel <- rotated sel ; sel fresh (:1911-1915) */
  free(st->prev_sel);
  st->prev_sel = rotate_f32(st->sel, W, H, S);
  ROT(cost, rotate_f32, S);
  if (st->mask) ROT(mask, rotate_u8, S);
#undef ROT
  st->rot = (st->rot + 1) % 4;
  st->W = H;
  st->H = W;
}

/* ------------------------------------------------------------------------- */
/* Setup                                                                     */
/* ------------------------------------------------------------------------- */

/* FilterKernel, gpu_mat_ref_image.cu:39-81 */
PMO_API void pmo_filter_ref_image(const uint8_t* gray, int W, int H, int radius, int step,
                                  float sigma_spatial, float sigma_color, uint8_t* out_image,
                                  float* out_sum, float* out_sqsum) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;
  const float sn = 1.0f / (2.0f * sigma_spatial * sigma_spatial);
  const float cn = 1.0f / (2.0f * sigma_color * sigma_color);
#pragma omp parallel for schedule(static)
  for (int row = 0; row < H; ++row) {
    for (int col = 0; col < W; ++col) {
      const float center = lut[gray[(size_t)row * W + col]];
      float color_sum = 0.0f, color_squared_sum = 0.0f, bws = 0.0f;
      for (int wr = -radius; wr <= radius; wr += step) {
        for (int wc = -radius; wc <= radius; wc += step) {
          const int r = row + wr, c = col + wc;
          const float color = (r < 0 || c < 0 || r >= H || c >= W) ? 0.0f : lut[gray[(size_t)r * W + c]];
          const float bw = bilateral_weight(sn, cn, (float)wr, (float)wc, center, color);
          color_sum += bw * color;
          color_squared_sum += bw * color * color;
          bws += bw;
        }
      }
      color_sum /= bws;
      color_squared_sum /= bws;
      out_image[(size_t)row * W + col] = (uint8_t)(255.0f * center);
      out_sum[(size_t)row * W + col] = color_sum;
      out_sqsum[(size_t)row * W + col] = color_squared_sum;
    }
  }
}

/* InitTransforms, patch_match_cuda.cu:1694-1808 */
static void init_transforms(pmo_state* st, const pmo_image* images, int ref_idx, const int* src) {
  const pmo_image* ref = &images[ref_idx];
  float (*K)[4] = st->ref_K;
  for (int i = 0; i < 4; ++i) {
    K[i][0] = ref->K[0]; K[i][1] = ref->K[2]; K[i][2] = ref->K[4]; K[i][3] = ref->K[5];
  }
  float t;
  /* 90 */
  t = K[1][0]; K[1][0] = K[1][2]; K[1][2] = t;
  t = K[1][1]; K[1][1] = K[1][3]; K[1][3] = t;
  K[1][3] = st->ref_w - 1 - K[1][3];
  /* 180 */
  K[2][1] = st->ref_w - 1 - K[2][1];
  K[2][3] = st->ref_h - 1 - K[2][3];
  /* 270 */
  t = K[3][0]; K[3][0] = K[3][2]; K[3][2] = t;
  t = K[3][1]; K[3][1] = K[3][3]; K[3][3] = t;
  K[3][1] = st->ref_h - 1 - K[3][1];
  for (int i = 0; i < 4; ++i) {
    st->ref_inv_K[i][0] = 1.0f / K[i][0];
    st->ref_inv_K[i][1] = -K[i][1] / K[i][0];
    st->ref_inv_K[i][2] = 1.0f / K[i][2];
    st->ref_inv_K[i][3] = -K[i][3] / K[i][2];
  }
  float rotated_R[9], rotated_T[3];
  memcpy(rotated_R, ref->R, sizeof(rotated_R));
  memcpy(rotated_T, ref->T, sizeof(rotated_T));
  const float R_z90[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
  for (int i = 0; i < 4; ++i) {
    st->poses[i] = (float*)malloc(sizeof(float) * PMO_POSE_STRIDE * st->S);
    for (int s = 0; s < st->S; ++s) {
      const pmo_image* im = &images[src[s]];
      float* p = st->poses[i] + s * PMO_POSE_STRIDE;
      p[0] = im->K[0]; p[1] = im->K[2]; p[2] = im->K[4]; p[3] = im->K[5];
      float rel_R[9], rel_T[3];
      pmo_compute_relative_pose(rotated_R, rotated_T, im->R, im->T, rel_R, rel_T);
      memcpy(p + 4, rel_R, sizeof(rel_R));
      memcpy(p + 13, rel_T, sizeof(rel_T));
      pmo_compute_projection_center(rel_R, rel_T, p + 16);
      pmo_compose_projection_matrix(im->K, rel_R, rel_T, p + 19);
      pmo_compose_inverse_projection_matrix(im->K, rel_R, rel_T, p + 31);
    }
    pmo_rotate_pose(R_z90, rotated_R, rotated_T);
  }
}

/* exported so tests can compare the pose tables with the product's host code */
PMO_API int pmo_pose_tables(int n_images, const pmo_image* images, int ref_idx, int n_src,
                            const int* src_idxs, float* out_poses /*[4][S][43]*/,
                            float* out_ref_K /*[4][4]*/, float* out_ref_inv_K /*[4][4]*/) {
  (void)n_images;
  pmo_state st;
  memset(&st, 0, sizeof(st));
  st.ref_w = images[ref_idx].width;
  st.ref_h = images[ref_idx].height;
  st.S = n_src;
  init_transforms(&st, images, ref_idx, src_idxs);
  for (int i = 0; i < 4; ++i) {
    memcpy(out_poses + (size_t)i * n_src * PMO_POSE_STRIDE, st.poses[i],
           sizeof(float) * n_src * PMO_POSE_STRIDE);
    memcpy(out_ref_K + 4 * i, st.ref_K[i], 16);
    memcpy(out_ref_inv_K + 4 * i, st.ref_inv_K[i], 16);
    free(st.poses[i]);
  }
  return 0;
}

static void free_state(pmo_state* st) {
  free(st->src_images); free(st->src_depths);
  for (int i = 0; i < 4; ++i) free(st->poses[i]);
  free(st->ref); free(st->ref_sum); free(st->ref_sqsum);
  free(st->depth); free(st->normal); free(st->cost); free(st->sel); free(st->prev_sel);
  free(st->rand); free(st->mask);
}

/* PatchMatch::Check subset (patch_match.cc:67-126). Returns 0 if ok. */
static int check_problem(const pmo_options* opt, int n_images, const pmo_image* images,
                         int ref_idx, int n_src, const int* src) {
  if (n_src <= 0) return 1;
  if (ref_idx < 0 || ref_idx >= n_images) return 2;
  for (int i = 0; i < n_src; ++i) {
    if (src[i] < 0 || src[i] >= n_images) return 3;
    if (src[i] == ref_idx) return 4;
    for (int j = 0; j < i; ++j) if (src[j] == src[i]) return 4;
  }
  for (int i = -1; i < n_src; ++i) {
    const pmo_image* im = &images[i < 0 ? ref_idx : src[i]];
    if (im->width <= 0 || im->height <= 0 || !im->gray) return 5;
    if (fabsf(im->K[1]) >= 1e-6f || fabsf(im->K[3]) >= 1e-6f || fabsf(im->K[6]) >= 1e-6f ||
        fabsf(im->K[7]) >= 1e-6f || fabsf(im->K[8] - 1.0f) >= 1e-6f) return 6;
    if (opt->geom_consistency && !im->depth) return 7;
  }
  if (opt->geom_consistency && !images[ref_idx].normal) return 8;
  if (opt->window_radius <= 0 || opt->window_radius > 32 || opt->window_step <= 0 ||
      opt->window_step > 2 || opt->num_samples <= 0 || opt->num_iterations <= 0) return 9;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Driver: PatchMatchCuda ctor + Run (patch_match_cuda.cu:1290-1302, 1393-1546) */
/* ------------------------------------------------------------------------- */

PMO_API int pmo_run(const pmo_options* opt, int n_images, const pmo_image* images, int ref_idx,
                    int n_src, const int* src_idxs, float* out_depth /*H*W*/,
                    float* out_normal /*3*H*W*/, float* out_sel_prob /*S*H*W or NULL*/,
                    uint8_t* out_mask /*S*H*W or NULL*/, float* out_cost /*S*H*W or NULL*/) {
  const int err = check_problem(opt, n_images, images, ref_idx, n_src, src_idxs);
  if (err) return err;
#ifdef _OPENMP
  if (opt->num_threads > 0) omp_set_num_threads(opt->num_threads);
#endif
  pmo_state st;
  memset(&st, 0, sizeof(st));
  const pmo_image* ref = &images[ref_idx];
  const int W = ref->width, H = ref->height, S = n_src;
  st.ref_w = W; st.ref_h = H; st.S = S; st.W = W; st.H = H; st.rot = 0;
  for (int i = 0; i < 256; ++i) st.lut[i] = (float)i / 255.0f;

  /* InitRefImage :1578-1593 */
  st.ref = (uint8_t*)malloc((size_t)W * H);
  st.ref_sum = (float*)malloc(sizeof(float) * W * H);
  st.ref_sqsum = (float*)malloc(sizeof(float) * W * H);
  pmo_filter_ref_image(ref->gray, W, H, opt->window_radius, opt->window_step,
                       (float)opt->sigma_spatial, (float)opt->sigma_color, st.ref, st.ref_sum,
                       st.ref_sqsum);

  /* InitSourceImages :1595-1692 */
  for (int s = 0; s < S; ++s) {
    const pmo_image* im = &images[src_idxs[s]];
    if (im->width > st.src_w) st.src_w = im->width;
    if (im->height > st.src_h) st.src_h = im->height;
  }
  const size_t slot = (size_t)st.src_w * st.src_h;
  st.src_images = (uint8_t*)calloc(slot * S, 1);
  for (int s = 0; s < S; ++s) {
    const pmo_image* im = &images[src_idxs[s]];
    /* contiguous memcpy without re-pitching, as the reference (:1617-1622) */
    memcpy(st.src_images + slot * s, im->gray, (size_t)im->width * im->height);
  }
  if (opt->geom_consistency) {
    st.src_depths = (float*)calloc(slot * S, sizeof(float));
    for (int s = 0; s < S; ++s) {
      const pmo_image* im = &images[src_idxs[s]];
      for (int r = 0; r < im->height; ++r) /* row copy (:1668-1673) */
        memcpy(st.src_depths + slot * s + (size_t)r * st.src_w, im->depth + (size_t)r * im->width,
               sizeof(float) * im->width);
    }
  }

  init_transforms(&st, images, ref_idx, src_idxs);

  /* InitWorkspaceMemory :1810-1857 */
  st.rand = (pmo_rng*)malloc(sizeof(pmo_rng) * W * H);
  {
    const int gx = (W - 1) / 32 + 1; /* GpuMat block 32x16 (gpu_mat.h:159-160,332-340) */
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        const uint64_t block = (uint64_t)(row / 16) * gx + (col / 32);
        const uint64_t id = block * 16 * 32 + (uint64_t)(row % 16) * 32 + (col % 32);
        pmo_rng_init(&st.rand[(size_t)row * W + col], id);
      }
  }
  st.depth = (float*)malloc(sizeof(float) * W * H);
  st.normal = (float*)malloc(sizeof(float) * 3 * W * H);
  st.cost = (float*)malloc(sizeof(float) * (size_t)S * W * H);
  st.sel = (float*)malloc(sizeof(float) * (size_t)S * W * H);
  st.prev_sel = (float*)malloc(sizeof(float) * (size_t)S * W * H);
  for (size_t i = 0; i < (size_t)S * W * H; ++i) { st.prev_sel[i] = 0.5f; st.sel[i] = 0.0f; }
  if (opt->geom_consistency) {
    memcpy(st.depth, ref->depth, sizeof(float) * W * H);
    memcpy(st.normal, ref->normal, sizeof(float) * 3 * W * H);
  } else {
    /* FillWithRandomNumbers gpu_mat.h:370-387 */
    const float dmin = (float)opt->depth_min, dmax = (float)opt->depth_max;
    for (size_t i = 0; i < (size_t)W * H; ++i)
      st.depth[i] = pmo_rng_uniform(&st.rand[i]) * (dmax - dmin) + dmin;
    /* InitNormalMap :835-846 */
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        float n[3];
        generate_random_normal(st.ref_inv_K[0], row, col, &st.rand[(size_t)row * W + col], n);
        for (int k = 0; k < 3; ++k) st.normal[((size_t)k * H + row) * W + col] = n[k];
      }
  }

  /* RunWithWindowSizeAndStep :1393-1546 */
  compute_initial_cost(&st, opt);

  sweep_options so;
  so.depth_min = (float)opt->depth_min;
  so.depth_max = (float)opt->depth_max;
  so.sigma_spatial = (float)opt->sigma_spatial;
  so.sigma_color = (float)opt->sigma_color;
  so.num_samples = opt->num_samples;
  so.ncc_sigma = (float)opt->ncc_sigma;
  so.min_triangulation_angle = (float)(opt->min_triangulation_angle * 0.0174532925199432);
  so.incident_angle_sigma = (float)opt->incident_angle_sigma;
  so.geom_consistency_regularizer = (float)opt->geom_consistency_regularizer;
  so.geom_consistency_max_cost = (float)opt->geom_consistency_max_cost;
  so.filter_min_ncc = (float)opt->filter_min_ncc;
  so.filter_min_triangulation_angle =
      (float)(opt->filter_min_triangulation_angle * 0.0174532925199432);
  so.filter_min_num_consistent = opt->filter_min_num_consistent;
  so.filter_geom_consistency_max_cost = (float)opt->filter_geom_consistency_max_cost;

  const float total_num_steps = (float)(opt->num_iterations * 4);
  const int nt = num_taps_1d(opt->window_radius, opt->window_step);
  const size_t scratch_floats = (size_t)7 * S + 2 * nt * nt + (5 * S + 3) / 4 + 4;
  int sweeps_done = 0;
  for (int iter = 0; iter < opt->num_iterations; ++iter) {
    for (int sweep = 0; sweep < 4; ++sweep) {
      if (opt->max_sweeps >= 0 && sweeps_done >= opt->max_sweeps) goto done;
      so.perturbation = 1.0f / powf(2.0f, iter + sweep / 4.0f);
      so.prev_sel_prob_weight = (float)(iter * 4 + sweep) / total_num_steps;
      const int last_sweep = iter == opt->num_iterations - 1 && sweep == 3;
      int geom_term = opt->geom_consistency ? 1 : 0;
      int fphoto = 0, fgeom = 0;
      if (last_sweep && opt->filter) {
        free(st.mask);
        st.mask = (uint8_t*)calloc((size_t)S * st.W * st.H, 1);
        fphoto = 1;
        fgeom = opt->geom_consistency ? 1 : 0;
      }
#pragma omp parallel
      {
        float* scratch = (float*)malloc(sizeof(float) * scratch_floats);
#pragma omp for schedule(dynamic, 1)
        for (int col = 0; col < st.W; ++col)
          sweep_column(&st, opt, &so, geom_term, fphoto, fgeom, col, scratch);
        free(scratch);
      }
      rotate_state(&st);
      ++sweeps_done;
    }
  }
done:
  /* bring a truncated run back to the un-rotated frame (pure data movement) */
  while (st.rot != 0) {
    /* rotate_state() sets prev_sel = rotate(sel); swap so that the real
     * prev_sel is what gets re-oriented (sel is scratch between sweeps) */
    float* tmp = st.sel;
    st.sel = st.prev_sel;
    st.prev_sel = tmp;
    rotate_state(&st);
  }

  memcpy(out_depth, st.depth, sizeof(float) * W * H);
  memcpy(out_normal, st.normal, sizeof(float) * 3 * W * H);
  if (out_sel_prob) memcpy(out_sel_prob, st.prev_sel, sizeof(float) * (size_t)S * W * H);
  if (out_cost) memcpy(out_cost, st.cost, sizeof(float) * (size_t)S * W * H);
  if (out_mask) {
    if (st.mask) memcpy(out_mask, st.mask, (size_t)S * W * H);
    else memset(out_mask, 0, (size_t)S * W * H);
  }
  free_state(&st);
  return 0;
}

PMO_API int pmo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
