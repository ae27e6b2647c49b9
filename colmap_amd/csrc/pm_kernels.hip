// pm_kernels.hip -- gfx950 (MI355X / CDNA4) PatchMatch multi-view-stereo kernels.
//
// What the reference computes: src/colmap/mvs/patch_match_cuda.cu (one CUDA
// thread per image column, 32-thread blocks, every buffer physically rotated by
// 90 degrees between sweeps). How it is computed here is different by design:
//
//  * A workgroup owns C adjacent image columns of the (virtually rotated) sweep
//    frame and walks them row by row. Inside one row step the independent pieces
//    of work -- per-view priors (C*S items) and the bilaterally weighted NCC
//    evaluations (up to C*4*min(M,S) + C*S items) -- are spread over the workgroup
//    through an LDS task list. Each NCC evaluation is computed by a 16-lane group
//    (taps dealt to lanes, DPP tree sum): the lanes of one gather instruction then
//    touch a few cache lines of ONE warped patch instead of 64 unrelated patches
//    (measured: the lane-per-evaluation version was bound by the L1 tag rate of
//    64-line gathers, not by VALU), and the wavefronts stay densely packed although
//    the algorithm is sequential along a column.
//  * Identical NCC evaluations inside a row step (the same hypothesis/view pair
//    drawn by several Monte-Carlo samples, patch_match_cuda.cu:1128-1172, and the
//    re-evaluation of the winner, :1188-1197) are computed once. The 121
//    bilateral weights of the reference patch (recomputed per evaluation by the
//    reference, :538-539) are computed once per row step into LDS.
//  * No buffer is ever rotated (reference Rotate(), :1859-1939, ~470 B/pixel/sweep
//    of pure HBM traffic): the sweep frame is a coordinate transform. Per-pixel
//    state is one contiguous record {depth, normal, cost[S], selA[S], selB[S]} so
//    that every sweep direction touches whole cache lines.
//  * Source images are stored as packed 2x2 bilinear footprints with a zero
//    border ring: one aligned 4-byte gather per tap instead of four byte
//    gathers plus bounds checks (gfx9 has no linear-filtered layered textures,
//    patch_match_cuda.cu:416-425).
//
// Arithmetic is specified in oracle/pm_oracle.c (header) and implemented here
// independently; compile with -ffp-contract=off.
#include "pm_internal.h"

#include "gfx950/pm_gfx950_asm.h"  // ubyte0..3, reduce16x3, launder_vgpr, llvm_struct_buffer_load_u32 (see its header)

#include <float.h>
#include <math.h>
#include <stdlib.h>

namespace colmap_amd {

// ---------------------------------------------------------------------------
// Scalar helpers (arithmetic spec)
// ---------------------------------------------------------------------------

__device__ __forceinline__ float pm_exp(float x) {
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
  return y * __uint_as_float((unsigned)(ni + 127) << 23);
}

__device__ __forceinline__ void pm_sincos(float a, float* s_out, float* c_out) {
  const float q = rintf(a * 0.636619772367581343f);
  float r = fmaf(q, -1.5703125f, a);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188216e-8f, r);
  const float z = r * r;
  float sp = -1.9515295891e-4f;
  sp = fmaf(sp, z, 8.3321608736e-3f);
  sp = fmaf(sp, z, -1.6666654611e-1f);
  const float sv = fmaf(sp * z, r, r);
  float cp = 2.443315711809948e-5f;
  cp = fmaf(cp, z, -1.388731625493765e-3f);
  cp = fmaf(cp, z, 4.166664568298827e-2f);
  const float cv = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
  const int qi = ((int)q) & 3;
  const float s = (qi == 0) ? sv : (qi == 1) ? cv : (qi == 2) ? -sv : -cv;
  const float c = (qi == 0) ? cv : (qi == 1) ? -sv : (qi == 2) ? -cv : sv;
  *s_out = s;
  *c_out = c;
}

__device__ __forceinline__ float pm_rsqrt(float x) { return 1.0f / sqrtf(x); }

// (float)b / 255.0f for an integer-valued b in [0,255], exactly, in three VALU ops.
__device__ __forceinline__ float texel_norm(float b) {
  const float r = 0x1.010102p-8f;
  const float q = b * r;
  const float e = fmaf(-255.0f, q, b);
  return fmaf(e, r, q);
}

struct Rng {
  uint32_t x0, x1, x2, x3, x4, d;
};

__device__ __forceinline__ void rng_init(Rng& st, unsigned long long seed) {
  st.x0 = 123456789U; st.x1 = 362436069U; st.x2 = 521288629U; st.x3 = 88675123U;
  st.x4 = 5783321U; st.d = 6615241U;
  const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fU;
  const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbU;
  const uint32_t t0 = 1228688033U * s0;
  const uint32_t t1 = 2073658381U * s1;
  st.x0 += t0; st.x1 ^= t0; st.x2 += t1; st.x3 ^= t1; st.x4 += t0;
  st.d += t1 + t0;
}

__device__ __forceinline__ float rng_uniform(Rng& st) {
  const uint32_t t = st.x0 ^ (st.x0 >> 2);
  st.x0 = st.x1; st.x1 = st.x2; st.x2 = st.x3; st.x3 = st.x4;
  st.x4 = (st.x4 ^ (st.x4 << 4)) ^ (t ^ (t << 1));
  st.d += 362437U;
  const uint32_t v = st.d + st.x4;
  return 2.3283064e-10f + ((float)v * 2.3283064e-10f);
}

__device__ __forceinline__ Rng rng_load(const uint32_t* p) {
  Rng r; r.x0 = p[0]; r.x1 = p[1]; r.x2 = p[2]; r.x3 = p[3]; r.x4 = p[4]; r.d = p[5]; return r;
}
__device__ __forceinline__ void rng_store(uint32_t* p, const Rng& r) {
  p[0] = r.x0; p[1] = r.x1; p[2] = r.x2; p[3] = r.x3; p[4] = r.x4; p[5] = r.d;
}

// ---------------------------------------------------------------------------
// Virtual rotation: sweep-frame (row, col) -> un-rotated pixel index.
// Equivalent to applying CudaRotateKernel (reference cuda_rotate.h:57-75)
// `rot` times: rot 1: x = W-1-row, y = col; rot 2: y = H-1-row, x = W-1-col;
// rot 3: x = row, y = H-1-col.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rot_width(const PmParams& p) { return (p.rot & 1) ? p.H : p.W; }
__device__ __forceinline__ int rot_height(const PmParams& p) { return (p.rot & 1) ? p.W : p.H; }

__device__ __forceinline__ int pix_index(const PmParams& p, int row, int col) {
  int x, y;
  switch (p.rot) {
    case 0: x = col; y = row; break;
    case 1: x = p.W - 1 - row; y = col; break;
    case 2: x = p.W - 1 - col; y = p.H - 1 - row; break;
    default: x = row; y = p.H - 1 - col; break;
  }
  return y * p.W + x;
}

// normals are stored in the un-rotated frame; RotateNormalMap (reference
// patch_match_cuda.cu:849-861) applied `rot` times is a signed swap (exact).
__device__ __forceinline__ void normal_to_sweep(int rot, float nx, float ny, float& ox, float& oy) {
  switch (rot) {
    case 0: ox = nx; oy = ny; break;
    case 1: ox = ny; oy = -nx; break;
    case 2: ox = -nx; oy = -ny; break;
    default: ox = -ny; oy = nx; break;
  }
}
__device__ __forceinline__ void normal_from_sweep(int rot, float sx, float sy, float& nx, float& ny) {
  switch (rot) {
    case 0: nx = sx; ny = sy; break;
    case 1: nx = -sy; ny = sx; break;
    case 2: nx = -sx; ny = -sy; break;
    default: nx = sy; ny = -sx; break;
  }
}

// reference-image texel in the sweep frame: point fetch, border 0, /255
__device__ __forceinline__ float ref_texel(const PmParams& p, int row, int col) {
  if (row < 0 || col < 0 || row >= rot_height(p) || col >= rot_width(p)) return 0.0f;
  return texel_norm((float)p.ref_img[pix_index(p, row, col)]);
}

// ---------------------------------------------------------------------------
// Geometry (restating patch_match_cuda.cu device functions)
// ---------------------------------------------------------------------------

// LDS pointers carry address space 3 explicitly so that every access is a ds_*
// instruction (generic pointers kept in a struct degrade to flat_* loads).
#define LDS_AS __attribute__((address_space(3)))
// global (address space 1) view of a pointer loaded from the parameter block: without it
// the gathers compile to flat_load (pointer provenance is unknown to the compiler)
typedef __attribute__((address_space(1))) const uint32_t gbl_u32;
typedef LDS_AS float lds_f32;
typedef __attribute__((address_space(1))) const float gbl_f32;
typedef LDS_AS int lds_i32;
typedef LDS_AS uint32_t lds_u32;
typedef LDS_AS uint64_t lds_u64;
typedef LDS_AS char lds_char;


__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
  return a0 * b0 + a1 * b1 + a2 * b2;
}

// ComposeHomography, patch_match_cuda.cu:271-332
// (PoseP: pointer to one source image's pose record -- LDS in most kernels, global memory in the
// pose-global build of the single-wave sweep kernel)
template <typename PoseP>
__device__ __forceinline__ void compose_homography(const float* iK, PoseP pose, int row,
                                                   int col, float depth, float n0, float n1,
                                                   float n2, float H[9]) {
  const PoseP K = pose;
  const PoseP R = pose + 4;
  const PoseP T = pose + 13;
  const float dist = depth * (n0 * (iK[0] * col + iK[1]) + n1 * (iK[2] * row + iK[3]) + n2);
  const float inv_dist = 1.0f / dist;
  const float N0 = inv_dist * n0;
  const float N1 = inv_dist * n1;
  const float N2 = inv_dist * n2;
  H[0] = iK[0] * (K[0] * (R[0] + N0 * T[0]) + K[1] * (R[6] + N0 * T[2]));
  H[1] = iK[2] * (K[0] * (R[1] + N1 * T[0]) + K[1] * (R[7] + N1 * T[2]));
  H[2] = K[0] * (R[2] + N2 * T[0]) + K[1] * (R[8] + N2 * T[2]) +
         iK[1] * (K[0] * (R[0] + N0 * T[0]) + K[1] * (R[6] + N0 * T[2])) +
         iK[3] * (K[0] * (R[1] + N1 * T[0]) + K[1] * (R[7] + N1 * T[2]));
  H[3] = iK[0] * (K[2] * (R[3] + N0 * T[1]) + K[3] * (R[6] + N0 * T[2]));
  H[4] = iK[2] * (K[2] * (R[4] + N1 * T[1]) + K[3] * (R[7] + N1 * T[2]));
  H[5] = K[2] * (R[5] + N2 * T[1]) + K[3] * (R[8] + N2 * T[2]) +
         iK[1] * (K[2] * (R[3] + N0 * T[1]) + K[3] * (R[6] + N0 * T[2])) +
         iK[3] * (K[2] * (R[4] + N1 * T[1]) + K[3] * (R[7] + N1 * T[2]));
  H[6] = iK[0] * (R[6] + N0 * T[2]);
  H[7] = iK[2] * (R[7] + N1 * T[2]);
  H[8] = R[8] + iK[1] * (R[6] + N0 * T[2]) + iK[3] * (R[7] + N1 * T[2]) + N2 * T[2];
}

// BilateralWeightComputer::Compute, gpu_mat_ref_image.h:70-90
__device__ __forceinline__ float bilateral_weight(float spatial_norm, float color_norm,
                                                  float row_diff, float col_diff, float c1,
                                                  float c2) {
  const float sds = row_diff * row_diff + col_diff * col_diff;
  const float cd = c1 - c2;
  return pm_exp(-sds * spatial_norm - cd * cd * color_norm);
}

// Two-wide fp32 vectors: CDNA4 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on register
// pairs at the rate of the scalar forms, so the NCC arithmetic is written on pairs of taps.
// Every packed operation is the IEEE operation of its two halves (bit-identical to scalar code).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_bcast(float a) { return (v2f)(a); }

// Packed source images ("footprints"): entry (ex, ey) is one dword = the 2 x 2 bilinear neighbourhood of texel
// position (ex - kFpRingX, ey - kFpRingY) .. (+1, +1), with an all-zero border ring (pm_build_footprint_kernel).
// Layout (pm_internal.h): vertical STRIPS of kFpStrip entries; inside a strip the rows follow each other, so
//   entry index = (ex / strip) * strip * rows + strip * ey + (ex % strip) = ex + (ex & ~(strip - 1)) * (rows - 1) + strip * ey.
// The 11 x 11 sweep kernels never compute this index: a swizzled buffer resource (fp_resource) lets the address
// unit form it from the pair (ex, 4 * ey + slot); see ncc_front.
__device__ __forceinline__ unsigned fp_index(unsigned ix, unsigned iy, unsigned rows1) {
  return __umul24(ix & ~(unsigned)(kFpStrip - 1), rows1) + ix + iy * (unsigned)kFpStrip;
}

// Footprint gather of one tap (generic kernels): floor of the warped coordinate clamped to the zero ring
// [-2, w] x [-2, h] in the float domain (one v_med3 each), so that arbitrarily distant / non-finite taps read an
// all-zero footprint entry (SampleLayeredBilinear, patch_match_cuda.cu:426-442; the +0.5 / -0.5 texel centre
// round trip of the reference cancels and is not evaluated in device order).
// fxr = floor(x) + kFpRingX, fyr = floor(y) + kFpRingY (the ring offset is added in the float domain, packed).
__device__ __forceinline__ uint32_t tap_gather(const PmParams& p, gbl_u32* fp, float fxr, float fyr) {
  const float cx = __builtin_amdgcn_fmed3f(fxr, (float)(kFpRingX - 2), p.fp_xmax);
  const float cy = __builtin_amdgcn_fmed3f(fyr, (float)(kFpRingY - 2), p.fp_ymax);
  return fp[fp_index((unsigned)(int)cx, (unsigned)(int)cy, (unsigned)p.fp_rows1)];
}

// (ubyte0 .. ubyte3 -- byte k of a packed footprint entry as float, v_cvt_f32_ubyteK -- live in gfx950/pm_gfx950_asm.h)

// Cross-lane add inside a 16-lane DPP row. The four steps (row_mirror,
// row_half_mirror, quad reverse, quad swap) leave in every lane
//   ((q0+q7)+(q3+q4)) + ((q1+q6)+(q2+q5)),  q_l = p_l + p_(15-l)
// -- the tree oracle/pm_oracle.c:tree16 restates.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float reduce16(float v) {
  v = v + dpp_row<0x140>(v);  // row_mirror:      l <-> 15-l
  v = v + dpp_row<0x141>(v);  // row_half_mirror: l <-> 7-l within each half
  v = v + dpp_row<0x1B>(v);   // quad_perm [3,2,1,0]
  v = v + dpp_row<0xB1>(v);   // quad_perm [1,0,3,2]
  return v;
}

// Stride of one column's per-tap planes in LDS: taps padded to whole 128-tap chunks, the padding
// holds weight 0 / colour 0 so that the tail of the last chunk needs no predication.
__host__ __device__ inline int tap_stride(int ntaps) { return ((ntaps + 127) / 128) * 128; }

// Per-tap bilateral weights and reference colours of one lane (taps j + 16 k, k = 0..7) held in
// registers across consecutive evaluations of the same pixel column (fixed window <= 128 taps).
struct TapRegs {
  v2f w[4], r[4];
};
template <int N1D>
struct TapRegsUsed {
  static constexpr bool value = N1D > 0 && N1D * N1D <= 128;
};
__device__ __forceinline__ void tap_regs_load(TapRegs& R, const lds_f32* wgt, const lds_f32* refc, int j) {
  const lds_f32* wj = wgt + j;
  const lds_f32* rj = refc + j;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    R.w[q][0] = wj[32 * q];
    R.w[q][1] = wj[32 * q + 16];
    R.r[q][0] = rj[32 * q];
    R.r[q][1] = rj[32 * q + 16];
  }
}

// PhotoConsistencyCostComputer::Compute, patch_match_cuda.cu:489-593, evaluated by a
// 16-lane group: tap t = wrow*n1d + wcol belongs to lane t % 16, so one gather
// instruction of a group covers 16 consecutive taps (~1.5 window rows, a handful of
// cache lines) instead of 16 unrelated patches. `H` is the homography of the
// (hypothesis, view) pair with its constant column already moved to the window origin
// (centre_homography below; LDS, precomputed once per task), `wgt` / `refc` are the per-tap
// bilateral weights and reference colours of the pixel's column. All 16 lanes return the same cost.
//
// Device order (restated in oracle/pm_oracle.c: ncc_cost_device): a lane's taps t = j + 16 k are
// processed in chunks of 8; tap pairs (k, k+1) travel through the arithmetic as the two halves
// of packed registers; the eight projective divisors z_k of a chunk share ONE correctly rounded
// division -- inv_k = (prod_{i<k} z_i * prod_{i>k} z_i) / prod z_i, prefix products taken in
// increasing k and the suffix product in decreasing k; taps beyond the window use z = 1, weight 0;
// even-k and odd-k taps accumulate separately and are added before the cross-lane tree.
template <int N1D>
__device__ __forceinline__ void ncc_group(const PmParams& p, const lds_f32* H, gbl_u32* fp,
                                          const lds_f32* wgt, const lds_f32* refc, const TapRegs& R,
                                          int j, float& s_sum, float& s_sq, float& s_ref) {
  const float h0 = H[0], h1 = H[1], h2 = H[2], h3 = H[3], h4 = H[4], h5 = H[5], h6 = H[6],
              h7 = H[7], h8 = H[8];
  const int n1d = N1D > 0 ? N1D : p.ntap1d;
  const int ntaps = n1d * n1d;
  const lds_f32* wj = wgt + j;
  const lds_f32* rj = refc + j;
  v2f a_sum = pk_bcast(0.0f), a_sq = pk_bcast(0.0f), a_ref = pk_bcast(0.0f);
  auto chunk = [&](int kb) {
    v2f csrc[4], rsrc[4], pre[4], suf[4];
    float zz[8];
    float run = 1.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v2f dx, dy;
      bool valid[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int t = j + 16 * (kb + 2 * q + e);
        valid[e] = t < ntaps;
        const int tt = valid[e] ? t : 0;
        int wrow = tt / n1d;
        int wcol = tt - wrow * n1d;
        if (p.rot & 1) {  // odd sweep directions: taps dealt column-major (see patch_weights)
          const int sw = wrow; wrow = wcol; wcol = sw;
        }
        dx[e] = (float)(wcol * p.step);
        dy[e] = (float)(wrow * p.step);
      }
      csrc[q] = pk_fma(pk_bcast(h0), dx, pk_fma(pk_bcast(h1), dy, pk_bcast(h2)));
      rsrc[q] = pk_fma(pk_bcast(h3), dx, pk_fma(pk_bcast(h4), dy, pk_bcast(h5)));
      const v2f z = pk_fma(pk_bcast(h6), dx, pk_fma(pk_bcast(h7), dy, pk_bcast(h8)));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        zz[2 * q + e] = valid[e] ? z[e] : 1.0f;
        pre[q][e] = run;
        run = run * zz[2 * q + e];
      }
    }
    const float rinv = 1.0f / run;
    float sfx = 1.0f;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
      suf[k >> 1][k & 1] = sfx;
      sfx = sfx * zz[k];
    }
    v2f wx[4], wy[4];
    uint32_t tex[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v2f inv_z = (pre[q] * suf[q]) * pk_bcast(rinv);
      const v2f px = inv_z * csrc[q];
      const v2f py = inv_z * rsrc[q];
      v2f fx, fy;
      fx[0] = floorf(px[0]);
      fx[1] = floorf(px[1]);
      fy[0] = floorf(py[0]);
      fy[1] = floorf(py[1]);
      wx[q] = px - fx;
      wy[q] = py - fy;
      const v2f fx2 = fx + pk_bcast((float)kFpRingX);
      const v2f fy2 = fy + pk_bcast((float)kFpRingY);
      tex[2 * q] = tap_gather(p, fp, fx2[0], fy2[0]);
      tex[2 * q + 1] = tap_gather(p, fp, fx2[1], fy2[1]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // bilinear blend of the four raw texels (exact small integers in float) in lerp form, then
      // one scale by 1/255: the device-order reading of "bilinear fetch of a normalised uint8
      // texture" (oracle/pm_oracle.c: tex_src_bilinear_raw)
      v2f c00, c10, c01, c11;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int t = j + 16 * (kb + 2 * q + e);
        const uint32_t x = t < ntaps ? tex[2 * q + e] : 0u;
        c00[e] = ubyte0(x);
        c10[e] = ubyte1(x);
        c01[e] = ubyte2(x);
        c11[e] = ubyte3(x);
      }
      const v2f top = pk_fma(wx[q], c10 - c00, c00);
      const v2f bot = pk_fma(wx[q], c11 - c01, c01);
      const v2f src = pk_fma(wy[q], bot - top, top) * pk_bcast(0x1.010102p-8f);
      const int t0 = 16 * (kb + 2 * q);  // one base register + immediate offsets: ds_read2_b32
      v2f w2, r2;
      if (TapRegsUsed<N1D>::value) {
        w2 = R.w[q];
        r2 = R.r[q];
      } else {
        w2[0] = wj[t0];
        w2[1] = wj[t0 + 16];
        r2[0] = rj[t0];
        r2[1] = rj[t0 + 16];
      }
      const v2f bws = w2 * src;
      a_sum = a_sum + bws;
      a_sq = pk_fma(bws, src, a_sq);
      a_ref = pk_fma(bws, r2, a_ref);
    }
  };
  if (N1D > 0) {
    constexpr int NCHUNK = (N1D * N1D + 127) / 128 > 0 ? (N1D * N1D + 127) / 128 : 1;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) chunk(8 * c);
  } else {
    const int nchunk = (ntaps + 127) / 128;
    for (int c = 0; c < nchunk; ++c) chunk(8 * c);
  }
  // The three window sums (not yet normalised); ncc_finish() turns them into the cost. The callers
  // park them in LDS and finish all evaluations of a phase lane-per-evaluation: square root and
  // division are then paid once per evaluation instead of once per lane of its group.
  s_sum = reduce16(a_sum[0] + a_sum[1]);
  s_sq = reduce16(a_sq[0] + a_sq[1]);
  s_ref = reduce16(a_ref[0] + a_ref[1]);
}

// Tail of PhotoConsistencyCostComputer::Compute (patch_match_cuda.cu:575-592).
__device__ __forceinline__ float ncc_finish(float s_sum, float s_sq, float s_ref, float ref_sum,
                                            float ref_sqsum, float inv_w) {
  s_sum *= inv_w;
  s_sq *= inv_w;
  s_ref *= inv_w;
  const float ref_var = ref_sqsum - ref_sum * ref_sum;
  const float src_var = s_sq - s_sum * s_sum;
  const float kMinVar = 1e-5f;
  if (ref_var < kMinVar || src_var < kMinVar) return 2.0f;
  const float covar = s_ref - ref_sum * s_sum;
  const float var = sqrtf(ref_var * src_var);
  return fmaxf(0.0f, fminf(2.0f, 1.0f - covar / var));
}

// Moves the constant column of a homography to the window origin (col - r, row - r): taps then
// address the patch by small non-negative offsets (device order, oracle: ncc_cost_device).
__device__ __forceinline__ void centre_homography(float* Hm, int row, int col, int radius) {
  const float x0 = (float)(col - radius), y0 = (float)(row - radius);
  Hm[2] = fmaf(Hm[0], x0, fmaf(Hm[1], y0, Hm[2]));
  Hm[5] = fmaf(Hm[3], x0, fmaf(Hm[4], y0, Hm[5]));
  Hm[8] = fmaf(Hm[6], x0, fmaf(Hm[7], y0, Hm[8]));
}

// ComputeGeomConsistencyCost, patch_match_cuda.cu:601-667
template <typename PoseP>
__device__ __forceinline__ float geom_cost(const PmParams& p, PoseP pose, int s, float row,
                                           float col, float depth) {
  const PoseP P = pose + 19;
  const PoseP iP = pose + 31;
  const float* iK = p.refInvK;
  const float f0 = depth * (iK[0] * col + iK[1]);
  const float f1 = depth * (iK[2] * row + iK[3]);
  const float f2 = depth;
  const float inv_fz = 1.0f / (P[8] * f0 + P[9] * f1 + P[10] * f2 + P[11]);
  float src_col = inv_fz * (P[0] * f0 + P[1] * f1 + P[2] * f2 + P[3]);
  float src_row = inv_fz * (P[4] * f0 + P[5] * f1 + P[6] * f2 + P[7]);
  const float sx = floorf(src_col + 0.5f);
  const float sy = floorf(src_row + 0.5f);
  float src_depth = 0.0f;
  if (sx >= 0.0f && sy >= 0.0f && sx < (float)p.src_w && sy < (float)p.src_h) {
    src_depth = p.src_depth[((size_t)s * p.src_h + (int)sy) * p.src_w + (int)sx];
  }
  if (src_depth == 0.0f) return p.geom_max_cost;
  src_col *= src_depth;
  src_row *= src_depth;
  const float bx = iP[0] * src_col + iP[1] * src_row + iP[2] * src_depth + iP[3];
  const float by = iP[4] * src_col + iP[5] * src_row + iP[6] * src_depth + iP[7];
  const float bz = iP[8] * src_col + iP[9] * src_row + iP[10] * src_depth + iP[11];
  const float inv_bz = 1.0f / bz;
  const float back_col = inv_bz * (p.refK[0] * bx + p.refK[1] * bz);
  const float back_row = inv_bz * (p.refK[2] * by + p.refK[3] * bz);
  const float dc = col - back_col;
  const float dr = row - back_row;
  return fminf(p.geom_max_cost, sqrtf(dc * dc + dr * dr));
}

// LikelihoodComputer, patch_match_cuda.cu:698-832
__device__ __forceinline__ float ncc_prob(const PmParams& p, float cost) {
  return pm_exp(cost * cost * p.inv_ncc_sigma_sq) * p.ncc_norm;
}

template <bool kForward>
__device__ __forceinline__ float hmm_message(const PmParams& p, float cost, float prev) {
  const float kUniformProb = 0.5f;
  const float kNoChangeProb = 0.99999f;
  const float kChangeProb = 1.0f - kNoChangeProb;
  const float emission = ncc_prob(p, cost);
  float zn0, zn1;
  if (kForward) {
    zn0 = (prev * kChangeProb + (1.0f - prev) * kNoChangeProb) * kUniformProb;
    zn1 = (prev * kNoChangeProb + (1.0f - prev) * kChangeProb) * emission;
  } else {
    zn0 = prev * emission * kChangeProb + (1.0f - prev) * kUniformProb * kNoChangeProb;
    zn1 = prev * emission * kNoChangeProb + (1.0f - prev) * kUniformProb * kChangeProb;
  }
  return zn1 / (zn0 + zn1);
}

__device__ __forceinline__ float sel_prob_fn(float alpha, float beta, float prev, float w) {
  const float zn0 = (1.0f - alpha) * (1.0f - beta);
  const float zn1 = alpha * beta;
  const float curr = zn1 / (zn0 + zn1);
  return w * prev + (1.0f - w) * curr;
}

// ComputeViewingAngles, patch_match_cuda.cu:241-269
template <typename PoseP>
__device__ __forceinline__ void viewing_angles(PoseP pose, float p0, float p1, float p2,
                                               float n0, float n1, float n2, float& cos_tri,
                                               float& cos_inc) {
  const PoseP C = pose + 16;
  const float s0 = C[0] - p0, s1 = C[1] - p1, s2 = C[2] - p2;
  const float rx_inv = pm_rsqrt(dot3(p0, p1, p2, p0, p1, p2));
  const float sx_inv = pm_rsqrt(dot3(s0, s1, s2, s0, s1, s2));
  cos_inc = dot3(s0, s1, s2, n0, n1, n2) * sx_inv;
  cos_tri = -dot3(s0, s1, s2, p0, p1, p2) * rx_inv * sx_inv;
}

__device__ __forceinline__ float tri_prob(const PmParams& p, float cos_tri) {
  if (cos_tri > p.cos_min_tri) {
    const float scaled = 1.0f - (1.0f - cos_tri) / (1.0f - p.cos_min_tri);
    const float lik = 1.0f - scaled * scaled;
    return fminf(1.0f, fmaxf(0.0f, lik));
  }
  return 1.0f;
}

__device__ __forceinline__ float inc_prob(const PmParams& p, float cos_inc) {
  const float x = 1.0f - fmaxf(0.0f, cos_inc);
  return pm_exp(x * x * p.inv_inc_sigma_sq);
}

__device__ __forceinline__ void h_apply(const float H[9], float v0, float v1, float& r0, float& r1) {
  const float inv_z = 1.0f / (H[6] * v0 + H[7] * v1 + H[8]);
  r0 = inv_z * (H[0] * v0 + H[1] * v1 + H[2]);
  r1 = inv_z * (H[3] * v0 + H[4] * v1 + H[5]);
}

// ComputeResolutionProb, patch_match_cuda.cu:759-791
__device__ __forceinline__ float res_prob(const float H[9], float row, float col, int radius) {
  const int ws = 2 * radius + 1;
  float a0, a1, b0, b1, c0, c1, d0, d1;
  h_apply(H, col - radius, row - radius, a0, a1);
  h_apply(H, col - radius, row + radius, b0, b1);
  h_apply(H, col + radius, row + radius, c0, c1);
  h_apply(H, col + radius, row - radius, d0, d1);
  const float ref_area = (float)(ws * ws);
  const float src_area = fabsf(0.5f * (a0 * b1 - b0 * a1 - a0 * d1 + b0 * c1 - c0 * b1 + d0 * a1 +
                                       c0 * d1 - d0 * c1));
  if (ref_area > src_area) return src_area / ref_area;
  return ref_area / src_area;
}

// PropagateDepth, patch_match_cuda.cu:210-236
__device__ __forceinline__ float propagate_depth(const float* iK, float depth1, float n1y,
                                                 float n1z, float row1, float row2) {
  const float x1 = depth1 * (iK[2] * row1 + iK[3]);
  const float y1 = depth1;
  const float x2 = x1 + n1z;
  const float y2 = y1 - n1y;
  const float x4 = iK[2] * row2 + iK[3];
  const float denom = x2 - x1 + x4 * (y1 - y2);
  if (fabsf(denom) < 1e-5f) return depth1;
  const float nom = y1 * x2 - x1 * y2;
  return nom / denom;
}

// PerturbNormal, patch_match_cuda.cu:133-196 (recursion as a loop)
__device__ __forceinline__ void perturb_normal(const float* iK, int row, int col, float perturbation,
                                               float n0, float n1, float n2, Rng& rng, float& o0,
                                               float& o1, float& o2) {
  for (int trial = 0;; ++trial) {
    const float a1 = (rng_uniform(rng) - 0.5f) * perturbation;
    const float a2 = (rng_uniform(rng) - 0.5f) * perturbation;
    const float a3 = (rng_uniform(rng) - 0.5f) * perturbation;
    float s1, s2, s3, c1, c2, c3;
    pm_sincos(a1, &s1, &c1);
    pm_sincos(a2, &s2, &c2);
    pm_sincos(a3, &s3, &c3);
    const float R0 = c2 * c3;
    const float R1 = -c2 * s3;
    const float R2 = s2;
    const float R3 = c1 * s3 + c3 * s1 * s2;
    const float R4 = c1 * c3 - s1 * s2 * s3;
    const float R5 = -c2 * s1;
    const float R6 = s1 * s3 - c1 * c3 * s2;
    const float R7 = c3 * s1 + c1 * s2 * s3;
    const float R8 = c1 * c2;
    o0 = R0 * n0 + R1 * n1 + R2 * n2;
    o1 = R3 * n0 + R4 * n1 + R5 * n2;
    o2 = R6 * n0 + R7 * n1 + R8 * n2;
    const float v0 = iK[0] * col + iK[1];
    const float v1 = iK[2] * row + iK[3];
    if (dot3(o0, o1, o2, v0, v1, 1.0f) >= 0.0f) {
      if (trial < 3) {
        perturbation = 0.5f * perturbation;
        continue;
      }
      o0 = n0; o1 = n1; o2 = n2;
      return;
    }
    const float inv_norm = pm_rsqrt(dot3(o0, o1, o2, o0, o1, o2));
    o0 *= inv_norm; o1 *= inv_norm; o2 *= inv_norm;
    return;
  }
}

// ---------------------------------------------------------------------------
// Setup kernels
// ---------------------------------------------------------------------------

// 2x2 footprint packing with a zero ring, tiled (pm_internal.h): entry (ex, ey) covers texels
// (x, y) = (ex - kFpRingX, ey - kFpRingY) .. (x+1, y+1); out-of-image texels are 0 (border mode,
// patch_match_cuda.cu:1627-1629). Entries x = -2 and x = w (y likewise) are entirely zero, so clamping a tap's
// integer coordinate to [-2, w] reproduces the border for arbitrarily distant taps.
__global__ void pm_build_footprint_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ fp,
                                          int w, int h, int pw, int ph) {
  const int ex = blockIdx.x * blockDim.x + threadIdx.x;
  const int ey = blockIdx.y;
  const int s = blockIdx.z;
  if (ex >= pw) return;
  const int x = ex - kFpRingX, y = ey - kFpRingY;
  const uint8_t* img = src + (size_t)s * w * h;
  auto tex = [&](int xx, int yy) -> uint32_t {
    return (xx >= 0 && yy >= 0 && xx < w && yy < h) ? (uint32_t)img[(size_t)yy * w + xx] : 0u;
  };
  fp[(size_t)s * pw * ph + fp_index((unsigned)ex, (unsigned)ey, (unsigned)(ph - 1))] =
      tex(x, y) | (tex(x + 1, y) << 8) | (tex(x, y + 1) << 16) | (tex(x + 1, y + 1) << 24);
}

// FilterKernel, gpu_mat_ref_image.cu:39-81
__global__ void pm_filter_ref_kernel(const uint8_t* __restrict__ gray, int W, int H, int radius,
                                     int step, float spatial_norm, float color_norm,
                                     uint8_t* __restrict__ out_img, float* __restrict__ out_sum,
                                     float* __restrict__ out_sqsum) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y * blockDim.y + threadIdx.y;
  if (col >= W || row >= H) return;
  const float center = texel_norm((float)gray[(size_t)row * W + col]);
  float color_sum = 0.0f, color_sq = 0.0f, bws = 0.0f;
  for (int wr = -radius; wr <= radius; wr += step) {
    for (int wc = -radius; wc <= radius; wc += step) {
      const int r = row + wr, c = col + wc;
      const float color =
          (r < 0 || c < 0 || r >= H || c >= W) ? 0.0f : texel_norm((float)gray[(size_t)r * W + c]);
      const float bw = bilateral_weight(spatial_norm, color_norm, (float)wr, (float)wc, center, color);
      color_sum += bw * color;
      color_sq += bw * color * color;
      bws += bw;
    }
  }
  color_sum /= bws;
  color_sq /= bws;
  out_img[(size_t)row * W + col] = (uint8_t)(255.0f * center);
  out_sum[(size_t)row * W + col] = color_sum;
  out_sqsum[(size_t)row * W + col] = color_sq;
}

// InitRandomStateKernel (gpu_mat_prng.cu:36-48) + FillWithRandomNumbersKernel
// (gpu_mat.h:370-387) + InitNormalMap (patch_match_cuda.cu:835-846) + the
// prev_sel_prob fill (:1833-1835), fused: one pass over the pixel records.
__global__ void pm_init_state_kernel(const PmParams p, int random_init, float depth_min, float depth_max,
                                     const float* __restrict__ init_depth,
                                     const float* __restrict__ init_normal) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y * blockDim.y + threadIdx.y;
  if (col >= p.W || row >= p.H) return;
  const int pix = row * p.W + col;
  // seed = linear thread id of the reference's 32x16-block launch
  const unsigned long long gx = (unsigned long long)((p.W - 1) / 32 + 1);
  const unsigned long long block = (unsigned long long)(row / 16) * gx + (unsigned long long)(col / 32);
  const unsigned long long id = block * 512ull + (unsigned long long)(row % 16) * 32ull + (col % 32);
  Rng rng;
  rng_init(rng, id);
  float* rec = p.rec + (size_t)pix * p.rec_stride;
  float depth, n0, n1, n2;
  if (random_init) {
    depth = rng_uniform(rng) * (depth_max - depth_min) + depth_min;
    // GenerateRandomNormal, patch_match_cuda.cu:94-123 (rotation 0 calibration)
    float v1 = 0.0f, v2 = 0.0f, s = 2.0f;
    while (s >= 1.0f) {
      v1 = 2.0f * rng_uniform(rng) - 1.0f;
      v2 = 2.0f * rng_uniform(rng) - 1.0f;
      s = v1 * v1 + v2 * v2;
    }
    const float s_norm = sqrtf(1.0f - s);
    n0 = 2.0f * v1 * s_norm;
    n1 = 2.0f * v2 * s_norm;
    n2 = 1.0f - 2.0f * s;
    const float r0 = p.refInvK[0] * col + p.refInvK[1];
    const float r1 = p.refInvK[2] * row + p.refInvK[3];
    if (dot3(n0, n1, n2, r0, r1, 1.0f) > 0) {
      n0 = -n0; n1 = -n1; n2 = -n2;
    }
  } else {
    depth = init_depth[pix];
    n0 = init_normal[pix];
    n1 = init_normal[(size_t)p.W * p.H + pix];
    n2 = init_normal[(size_t)2 * p.W * p.H + pix];
  }
  rec[0] = depth; rec[1] = n0; rec[2] = n1; rec[3] = n2;
  for (int s = 0; s < p.S; ++s) {
    rec[p.sel_in_off + s] = 0.5f;
    rec[p.sel_out_off + s] = 0.0f;
  }
  rng_store(p.rng + (size_t)pix * kRngWords, rng);
}

// ---------------------------------------------------------------------------
// LDS carve-up shared by the initial-cost and sweep kernels
// ---------------------------------------------------------------------------
struct Lds {
  lds_f32* poses;   // [S][pstride]
  int pstride;
  lds_u64* fpb;     // [S] packed source images (global addresses; generic kernels)
  lds_u32* fpo;     // [S] packed source images as slots of the problem's buffer resource (11 x 11 sweep kernels)
  lds_f32* tile;    // [win][C + 2r] reference colours, ring-buffered rows
  lds_f32* wgt;     // [C][tap_stride] bilateral weights (0 beyond ntaps)
  lds_f32* refc;    // [C][tap_stride] reference colours of the taps
  lds_f32* fm;      // [C][S] forward messages
  lds_f32* q;       // [C][S] sampling pdf -> cdf
  lds_f32* costv;   // [C][S] cost_map values of this row
  lds_f32* betav;   // [C][S] backward messages of this row
  lds_f32* prevv;   // [C][S] previous sel probs of this row
  lds_f32* ncc;     // [C][5][S] NCC per hypothesis/view, < 0: not computed (generic kernel)
  lds_f32* cost5;   // wave kernel: [C][5][S + 1] cost-map row (hypothesis 0), NCC of hypotheses 1..4; slot S of a row = 0
  lds_u32* drawn;   // wave kernel: [C][(S + 31) / 32] bitmap of the views drawn in this row
  lds_u32* desc;    // wave kernel: [cap] pass-B descriptor per slot of the batch (run_tasks_wave)
  lds_f32* geo;     // [C][5][S] geometric cost (GEOM only)
  lds_f32* hyp;     // [C][5][4] depth, normal
  lds_f32* colf;    // [C][8] ref_sum, ref_sqsum, point[3], pad
  lds_f32* us;      // [C][M] uniform draws
  lds_i32* sv;      // [C][M] sampled view per draw (-1: none)
  lds_i32* best;    // [C]
  lds_f32* csum;    // [C][5] accumulated hypothesis costs
  LDS_AS uint8_t* flags;  // [C][S] filter flags
  lds_u32* tasks;
  lds_f32* th;      // [max_tasks][9] homography of each queued NCC task
  lds_i32* ntasks;
  lds_f32* tapg;    // wave kernel: [4][128] tap tables (tap_tables_init)
  lds_u32* ring;    // wave kernel: [2][8][64] landing zone of the footprint gathers
  LDS_AS uint8_t* tin;  // (unused: the wave kernel's Lds::desc lives at this offset)
};

struct LdsOffsets {
  uint32_t poses, fpb, tile, wgt, refc, fm, q, costv, betav, prevv, ncc, geo, hyp, colf, us, sv, best, csum,
      flags, tasks, th, ntasks, tapg, ring, tin, total;
  uint32_t drawn = 0;
  uint32_t priv_stride = 0;  // multi-wave sweep kernel: bytes between the private regions of consecutive waves
};

// Pose record kept in LDS: K4 R9 T3 C3 always; the projection matrices P12 invP12 only serve the
// geometric consistency term.
__host__ __device__ inline int lds_pose_stride(bool geom) { return geom ? kPoseStride : 19; }

__host__ __device__ inline LdsOffsets lds_offsets(int C, int S, int radius, int ntaps, int M,
                                                  bool geom) {
  LdsOffsets o;
  uint32_t off = 0;
  auto take = [&](uint32_t bytes) {
    const uint32_t at = off;
    off += (bytes + 15u) & ~15u;
    return at;
  };
  const int win = 2 * radius + 1;
  const int tw = C + 2 * radius;
  const int ms = M < S ? M : S;
  // per drawn view: hypotheses 1..4 (+ one geometric-cost-only task); winner pass: <= S per column
  const int per_view = geom ? 5 : 4;
  const int max_tasks = C * (per_view * ms > S ? per_view * ms : S);
  o.poses = take(4u * S * lds_pose_stride(geom));
  o.fpb = take(8u * S);
  o.tile = take(4u * win * tw);
  o.wgt = take(4u * C * tap_stride(ntaps));
  o.refc = take(4u * C * tap_stride(ntaps));
  o.fm = take(4u * C * S);
  o.q = take(4u * C * S);
  o.costv = take(4u * C * S);
  o.betav = take(4u * C * S);
  o.prevv = take(4u * C * S);
  o.ncc = take(4u * C * 5 * S);
  o.geo = take(geom ? 4u * C * 5 * S : 0u);
  o.hyp = take(4u * C * 20);
  o.colf = take(4u * C * 8);
  o.us = take(4u * C * M);
  o.sv = take(4u * C * M);
  o.best = take(4u * C);
  o.csum = take(4u * C * 5);
  o.flags = take(1u * C * S);
  o.tasks = take(4u * max_tasks);
  o.th = take(36u * max_tasks);
  o.ntasks = take(16u);
  o.tapg = 0;
  o.ring = 0;
  o.tin = 0;
  o.total = off;
  return o;
}

__device__ __forceinline__ void lds_load_poses(const PmParams& p, Lds& L, bool geom, int tid, int nt) {
  L.pstride = lds_pose_stride(geom);
  for (int i = tid; i < p.S * L.pstride; i += nt) {
    const int s = i / L.pstride;
    L.poses[i] = p.poses[s * kPoseStride + (i - s * L.pstride)];
  }
  for (int i = tid; i < p.S; i += nt) L.fpb[i] = (uint64_t)p.src_fp_tab[i];
}

__device__ __forceinline__ void lds_bind(Lds& L, lds_char* base, const LdsOffsets& o) {
  L.poses = (lds_f32*)(base + o.poses);
  L.fpb = (lds_u64*)(base + o.fpb);
  L.fpo = (lds_u32*)(base + o.fpb);
  L.tile = (lds_f32*)(base + o.tile);
  L.wgt = (lds_f32*)(base + o.wgt);
  L.refc = (lds_f32*)(base + o.refc);
  L.fm = (lds_f32*)(base + o.fm);
  L.q = (lds_f32*)(base + o.q);
  L.costv = (lds_f32*)(base + o.costv);
  L.betav = (lds_f32*)(base + o.betav);
  L.prevv = (lds_f32*)(base + o.prevv);
  L.ncc = (lds_f32*)(base + o.ncc);
  L.cost5 = (lds_f32*)(base + o.ncc);
  L.drawn = (lds_u32*)(base + o.drawn);
  L.desc = (lds_u32*)(base + o.tin);
  L.geo = (lds_f32*)(base + o.geo);
  L.hyp = (lds_f32*)(base + o.hyp);
  L.colf = (lds_f32*)(base + o.colf);
  L.us = (lds_f32*)(base + o.us);
  L.sv = (lds_i32*)(base + o.sv);
  L.best = (lds_i32*)(base + o.best);
  L.csum = (lds_f32*)(base + o.csum);
  L.flags = (LDS_AS uint8_t*)(base + o.flags);
  L.tasks = (lds_u32*)(base + o.tasks);
  L.th = (lds_f32*)(base + o.th);
  L.ntasks = (lds_i32*)(base + o.ntasks);
  L.tapg = (lds_f32*)(base + o.tapg);
  L.ring = (lds_u32*)(base + o.ring);
  L.tin = (LDS_AS uint8_t*)(base + o.tin);
}

__device__ __forceinline__ uint32_t task_pack(int c, int i, int s, int geom_only) {
  return ((uint32_t)c << 24) | ((uint32_t)geom_only << 23) | ((uint32_t)i << 20) | (uint32_t)s;
}

// Load one sweep-frame row of the reference image into the LDS ring tile.
__device__ __forceinline__ void tile_load_row(const PmParams& p, const Lds& L, int col0, int row,
                                              int tid, int nthreads) {
  const int win = 2 * p.radius + 1;
  const int tw = p.C + 2 * p.radius;
  int slot = row % win;
  if (slot < 0) slot += win;
  for (int lc = tid; lc < tw; lc += nthreads) {
    L.tile[slot * tw + lc] = ref_texel(p, row, col0 - p.radius + lc);
  }
}

// Bilateral weights + reference colours of the patches centred on (row, col0+c).
__device__ __forceinline__ void patch_weights(const PmParams& p, const Lds& L, int row, int tid,
                                              int nthreads) {
  const int win = 2 * p.radius + 1;
  const int tw = p.C + 2 * p.radius;
  const int ts = tap_stride(p.ntaps);
  const int total = p.C * ts;
  for (int item = tid; item < total; item += nthreads) {
    const int c = item / ts;
    const int tap = item - c * ts;
    if (tap >= p.ntaps) {
      L.wgt[item] = 0.0f;
      L.refc[item] = 0.0f;
      continue;
    }
    // Tap index -> window position: row-major, but column-major in the odd sweep directions, where a
    // row of the rotated window runs along a COLUMN of the (never rotated) source images: the 16
    // consecutive taps of one gather instruction then still fall into one or two source rows
    // (oracle/pm_oracle.c: ncc_cost_device, `transpose`).
    int trow = tap / p.ntap1d;
    int tcol = tap - trow * p.ntap1d;
    if (p.rot & 1) {
      const int sw = trow; trow = tcol; tcol = sw;
    }
    const int wr_ = -p.radius + trow * p.step;
    const int wc_ = -p.radius + tcol * p.step;
    int slot_c = row % win;
    int slot = (row + wr_) % win;
    if (slot < 0) slot += win;
    if (slot_c < 0) slot_c += win;
    const float center = L.tile[slot_c * tw + c + p.radius];
    const float color = L.tile[slot * tw + c + p.radius + wc_];
    const float bw = bilateral_weight(p.spatial_norm, p.color_norm, (float)wr_, (float)wc_, center, color);
    L.wgt[item] = bw;
    L.refc[item] = color;
  }
}

// 1 / (sum of the bilateral weights of column c's patch), summed in the same lane
// partition and tree as the other NCC sums: the reference accumulates this sum inside
// every evaluation (:546,571) although it only depends on the reference patch.
__device__ __forceinline__ void patch_weight_sums(const PmParams& p, const Lds& L, int ncols,
                                                  int tid, int nthreads) {
  const int g = tid >> 4, j = tid & 15, ng = nthreads >> 4;
  for (int c = g; c < ncols; c += ng) {
    float w_sum = 0.0f;
    for (int t = j; t < p.ntaps; t += 16) w_sum += L.wgt[c * tap_stride(p.ntaps) + t];
    w_sum = reduce16(w_sum);
    if (j == 0) L.colf[c * 8 + 5] = 1.0f / w_sum;
  }
}

// ---------------------------------------------------------------------------
// ComputeInitialCost (patch_match_cuda.cu:863-912): C adjacent pixels of one row
// per workgroup, C*S NCC evaluations spread over the lanes. Rotation 0.
// ---------------------------------------------------------------------------
template <int N1D>
__global__ void __launch_bounds__(64) pm_initial_cost_kernel(const PmParams* __restrict__ pp) {
  const PmParams& p = pp[blockIdx.z];  // batch of reference images: one launch, grid.z problems
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lds L;
  lds_bind(L, (lds_char*)smem, lds_offsets(p.C, p.S, p.radius, p.ntaps, p.num_samples, false));
  const int tid = threadIdx.x, nt = blockDim.x;
  const int row = blockIdx.y;
  const int col0 = blockIdx.x * p.C;
  lds_load_poses(p, L, false, tid, nt);
  for (int r = row - p.radius; r <= row + p.radius; ++r) tile_load_row(p, L, col0, r, tid, nt);
  __syncthreads();
  patch_weights(p, L, row, tid, nt);
  __syncthreads();
  patch_weight_sums(p, L, p.C, tid, nt);
  // per (pixel, view): homography, lane per task
  for (int item = tid; item < p.C * p.S; item += nt) {
    const int c = item / p.S;
    const int s = item - c * p.S;
    const int col = col0 + c;
    if (col >= p.W) continue;
    const float* rec = p.rec + (size_t)(row * p.W + col) * p.rec_stride;
    float Hm[9];
    compose_homography(p.refInvK, L.poses + s * L.pstride, row, col, rec[0], rec[1], rec[2], rec[3], Hm);
    centre_homography(Hm, row, col, p.radius);
    for (int k = 0; k < 9; ++k) L.th[item * 9 + k] = Hm[k];
  }
  __syncthreads();
  // NCC: 16-lane group per task
  const int g = tid >> 4, j = tid & 15, ng = nt >> 4;
  TapRegs R;
  int c_held = -1;
  for (int item = g; item < p.C * p.S; item += ng) {
    const int c = item / p.S;
    const int s = item - c * p.S;
    const int col = col0 + c;
    if (col >= p.W) continue;
    if (TapRegsUsed<N1D>::value && c != c_held) {
      tap_regs_load(R, L.wgt + c * tap_stride(p.ntaps), L.refc + c * tap_stride(p.ntaps), j);
      c_held = c;
    }
    float s_sum, s_sq, s_ref;
    ncc_group<N1D>(p, L.th + item * 9, (gbl_u32*)L.fpb[s], L.wgt + c * tap_stride(p.ntaps),
                   L.refc + c * tap_stride(p.ntaps), R, j, s_sum, s_sq, s_ref);
    if (j == 0) {
      L.th[item * 9 + 0] = s_sum;  // the homography of this evaluation is no longer needed
      L.th[item * 9 + 1] = s_sq;
      L.th[item * 9 + 2] = s_ref;
    }
  }
  __syncthreads();
  for (int item = tid; item < p.C * p.S; item += nt) {
    const int c = item / p.S;
    const int s = item - c * p.S;
    const int col = col0 + c;
    if (col >= p.W) continue;
    const int pix = row * p.W + col;
    p.rec[(size_t)pix * p.rec_stride + 4 + s] =
        ncc_finish(L.th[item * 9 + 0], L.th[item * 9 + 1], L.th[item * 9 + 2], p.ref_sum[pix],
                   p.ref_sqsum[pix], L.colf[c * 8 + 5]);
  }
}

// ---------------------------------------------------------------------------
// SweepFromTopToBottom (patch_match_cuda.cu:933-1288)
// ---------------------------------------------------------------------------

// Run every queued NCC / geometric-cost task. Pass A (lane per task): homography of
// the (hypothesis, view) pair and, with GEOM, the geometric consistency cost. Pass B
// (16-lane group per task): the bilaterally weighted NCC.
template <int N1D, bool GEOM>
__device__ __forceinline__ int run_tasks(const PmParams& p, const Lds& L, int row, int col0,
                                         int tid, int nt) {
  const int n = *L.ntasks;
  for (int t = tid; t < n; t += nt) {
    const uint32_t task = L.tasks[t];
    const int c = task >> 24;
    const int geom_only = (task >> 23) & 1;
    const int i = (task >> 20) & 7;
    const int s = task & 0xfffff;
    const lds_f32* h = L.hyp + (c * 5 + i) * 4;
    const lds_f32* pose = L.poses + s * L.pstride;
    const int col = col0 + c;
    if (!geom_only) {
      float Hm[9];
      compose_homography(p.refInvK, pose, row, col, h[0], h[1], h[2], h[3], Hm);
      centre_homography(Hm, row, col, p.radius);
      for (int k = 0; k < 9; ++k) L.th[t * 9 + k] = Hm[k];
    }
    if (GEOM) {
      L.geo[(c * 5 + i) * p.S + s] = geom_cost(p, pose, s, (float)row, (float)col, h[0]);
    }
  }
  __syncthreads();
  const int g = tid >> 4, j = tid & 15, ng = nt >> 4;
  // A wave's four groups work on the four hypotheses of one (column, view) block, and the blocks
  // of a column are mostly adjacent in the list: the column's tap weights stay in registers
  // until the column changes.
  TapRegs R;
  int c_held = -1;
  for (int t = g; t < n; t += ng) {
    const uint32_t task = L.tasks[t];
    if ((task >> 23) & 1) continue;  // geometric cost only
    const int c = task >> 24;
    const int s = task & 0xfffff;
    if (TapRegsUsed<N1D>::value && c != c_held) {
      tap_regs_load(R, L.wgt + c * tap_stride(p.ntaps), L.refc + c * tap_stride(p.ntaps), j);
      c_held = c;
    }
    float s_sum, s_sq, s_ref;
    ncc_group<N1D>(p, L.th + t * 9, (gbl_u32*)L.fpb[s], L.wgt + c * tap_stride(p.ntaps),
                   L.refc + c * tap_stride(p.ntaps), R, j, s_sum, s_sq, s_ref);
    if (j == 0) {
      L.th[t * 9 + 0] = s_sum;  // the homography of this task is no longer needed
      L.th[t * 9 + 1] = s_sq;
      L.th[t * 9 + 2] = s_ref;
    }
  }
  __syncthreads();
  // lane per task: normalisation, variances, square root, division
  for (int t = tid; t < n; t += nt) {
    const uint32_t task = L.tasks[t];
    if ((task >> 23) & 1) continue;
    const int c = task >> 24;
    const int i = (task >> 20) & 7;
    const int s = task & 0xfffff;
    L.ncc[(c * 5 + i) * p.S + s] = ncc_finish(L.th[t * 9 + 0], L.th[t * 9 + 1], L.th[t * 9 + 2],
                                              L.colf[c * 8 + 0], L.colf[c * 8 + 1], L.colf[c * 8 + 5]);
  }
  return n;  // queued tasks (with GEOM this includes the geometric-cost-only entries)
}

// ---------------------------------------------------------------------------
// 11 x 11 sweep kernels: the NCC evaluation split into the part before the footprint gathers (warp, shared
// division, address: ncc_front) and the part after them (byte conversion, lerp, window sums: ncc_back), so that
// all eight gathers of a lane are in flight before the first texel is consumed.
// ---------------------------------------------------------------------------
struct NccStage {  // what the back half needs from the front half besides the texels: the bilinear fractions
  v2f wx[4], wy[4];
};

// ---------------------------------------------------------------------------
// Packed-image gathers of the 11 x 11 sweep kernels: MUBUF loads through ONE swizzled buffer resource per problem.
// With swizzling enabled the address unit computes (gfx9 buffer addressing; verified on gfx950 by
// scripts/ubench/mubuf_addr.hip, profiles/r04_ubench_mubuf_addr.log)
//   address = base + ((index / IS) * stride + (offset / 4) * 4) * IS + (index % IS) * 4
// for index stride IS and element size 4. With IS = kFpStrip, stride = 4 * rows, index = ex and
// offset = 4 * ey + slot this is the strip layout's entry (ex, ey) of the packed image that starts `IS * slot`
// bytes behind the base: the tap address costs ONE VALU instruction (v_lshl_add_u32) instead of the seven of an
// explicit block index plus a 64-bit add. `slot` of a source image = (its address - lowest address of the
// problem's sources) / IS (pm_api.cpp: the images of a problem must lie within IS * 4 GB of each other, which
// the allocator's pool makes the normal case; otherwise the generic kernel runs).
// ---------------------------------------------------------------------------
// (v4i and llvm_struct_buffer_load_u32: gfx950/pm_gfx950_asm.h)

__device__ __forceinline__ v4i fp_resource(const PmParams& p) {
  const uint64_t b = (uint64_t)p.fp_base;
  const uint32_t stride = 4u * ((uint32_t)p.fp_rows1 + 1u);  // bytes; < 16384 (pm_fp_resource_ok)
  v4i r;
  r[0] = (int)(uint32_t)b;
  r[1] = (int)(((uint32_t)(b >> 32) & 0xffffu) | (stride << 16) | 0x80000000u);  // swizzle enable
  r[2] = 0x7fffffff;                                                             // records: no range limit in use
  // raw 32-bit data format; element size 4 bytes, index stride = strip width
  r[3] = 0x00020000 | (1 << 19) | ((kFpStrip == 8 ? 0 : kFpStrip == 16 ? 1 : kFpStrip == 32 ? 2 : 3) << 21);
  return r;
}

// Front half of an evaluation: warp, shared division, address, gathers. Arithmetic and order are those of
// ncc_group (oracle/pm_oracle.c: ncc_cost_device); fixed 11 x 11 window (121 taps = one 128-tap chunk).
// FAST: every tap of the four evaluations of this round is known to fall inside the packed image
// (patch_inside below), so the clamp of the tap coordinate to the zero ring is dropped: index and row are the
// truncated coordinates themselves (>= 1, truncation = floor), `slot` then already points at texel (0, 0), and
// the fractions come from v_fract_f32 (== x - floor(x) exactly for x >= 0). Same entry, same bits.
// MUBUF = false: the same entries through explicit strip indices and global loads (`gbase` = the image, or with
// FAST its entry of texel (0, 0)) -- three more VALU instructions per tap; for problems whose packed images do not
// fit one buffer resource (further than 4 GB apart: the address unit forms the buffer offset in 32 bits).
template <bool FAST, bool MUBUF>
__device__ __forceinline__ void ncc_front(const PmParams& p, const v4i srd, const lds_f32* H, uint32_t slot,
                                          gbl_u32* gbase, const lds_f32* tg, int j, NccStage& st, uint32_t* tex) {
  const float h0 = H[0], h1 = H[1], h2 = H[2], h3 = H[3], h4 = H[4], h5 = H[5], h6 = H[6],
              h7 = H[7], h8 = H[8];
  v2f csrc[4], rsrc[4], pre[4], suf[4];
  float zz[8];
  float run = 1.0f;
  const lds_f32* tj = tg + j;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v2f dx, dy;
    dx[0] = tj[32 * q];
    dx[1] = tj[32 * q + 16];
    dy[0] = tj[128 + 32 * q];
    dy[1] = tj[128 + 32 * q + 16];
    csrc[q] = pk_fma(pk_bcast(h0), dx, pk_fma(pk_bcast(h1), dy, pk_bcast(h2)));
    rsrc[q] = pk_fma(pk_bcast(h3), dx, pk_fma(pk_bcast(h4), dy, pk_bcast(h5)));
    const v2f z = pk_fma(pk_bcast(h6), dx, pk_fma(pk_bcast(h7), dy, pk_bcast(h8)));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool valid = j + 16 * (2 * q + e) < 121;
      zz[2 * q + e] = valid ? z[e] : 1.0f;
      pre[q][e] = run;
      run = run * zz[2 * q + e];
    }
  }
  const float rinv = 1.0f / run;
  float sfx = 1.0f;
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    suf[k >> 1][k & 1] = sfx;
    sfx = sfx * zz[k];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v2f inv_z = (pre[q] * suf[q]) * pk_bcast(rinv);
    const v2f px = inv_z * csrc[q];
    const v2f py = inv_z * rsrc[q];
    int ix[2], iy[2];
    if (FAST) {
      st.wx[q][0] = __builtin_amdgcn_fractf(px[0]);
      st.wx[q][1] = __builtin_amdgcn_fractf(px[1]);
      st.wy[q][0] = __builtin_amdgcn_fractf(py[0]);
      st.wy[q][1] = __builtin_amdgcn_fractf(py[1]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // Lanes whose tap lies beyond the window (t >= 121: weight 0, divisor forced to 1, so the coordinate is
        // the un-normalised numerator) have no inside guarantee: they read texel (0, 0) instead.
        const bool v = j + 16 * (2 * q + e) < 121;
        ix[e] = v ? (int)px[e] : 0;
        iy[e] = v ? (int)py[e] : 0;
      }
    } else {
      v2f fx, fy;
      fx[0] = floorf(px[0]);
      fx[1] = floorf(px[1]);
      fy[0] = floorf(py[0]);
      fy[1] = floorf(py[1]);
      st.wx[q] = px - fx;
      st.wy[q] = py - fy;
      const v2f fx2 = fx + pk_bcast((float)kFpRingX);
      const v2f fy2 = fy + pk_bcast((float)kFpRingY);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        ix[e] = (int)__builtin_amdgcn_fmed3f(fx2[e], (float)(kFpRingX - 2), p.fp_xmax);
        iy[e] = (int)__builtin_amdgcn_fmed3f(fy2[e], (float)(kFpRingY - 2), p.fp_ymax);
      }
    }
    if (MUBUF) {
      tex[2 * q] = llvm_struct_buffer_load_u32(srd, ix[0], (int)(((uint32_t)iy[0] << 2) + slot), 0, 0);
      tex[2 * q + 1] = llvm_struct_buffer_load_u32(srd, ix[1], (int)(((uint32_t)iy[1] << 2) + slot), 0, 0);
    } else {
      tex[2 * q] = gbase[fp_index((unsigned)ix[0], (unsigned)iy[0], (unsigned)p.fp_rows1)];
      tex[2 * q + 1] = gbase[fp_index((unsigned)ix[1], (unsigned)iy[1], (unsigned)p.fp_rows1)];
    }
  }
}
// (reduce16x3 -- the three 16-lane tree sums of an evaluation as one block of 12 v_add_f32_dpp -- lives in
// gfx950/pm_gfx950_asm.h; same tree as reduce16)

__device__ __forceinline__ void ncc_back(const NccStage& st, const uint32_t tex[8], const TapRegs& R, int j,
                                         float& s_sum, float& s_sq, float& s_ref) {
  v2f a_sum = pk_bcast(0.0f), a_sq = pk_bcast(0.0f), a_ref = pk_bcast(0.0f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v2f c00, c10, c01, c11;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t x = j + 16 * (2 * q + e) < 121 ? tex[2 * q + e] : 0u;
      c00[e] = ubyte0(x);
      c10[e] = ubyte1(x);
      c01[e] = ubyte2(x);
      c11[e] = ubyte3(x);
    }
    const v2f top = pk_fma(st.wx[q], c10 - c00, c00);
    const v2f bot = pk_fma(st.wx[q], c11 - c01, c01);
    const v2f src = pk_fma(st.wy[q], bot - top, top) * pk_bcast(0x1.010102p-8f);
    const v2f bws = R.w[q] * src;
    a_sum = a_sum + bws;
    a_sq = pk_fma(bws, src, a_sq);
    a_ref = pk_fma(bws, R.r[q], a_ref);
  }
  s_sum = a_sum[0] + a_sum[1];
  s_sq = a_sq[0] + a_sq[1];
  s_ref = a_ref[0] + a_ref[1];
  reduce16x3(s_sum, s_sq, s_ref);
}

// Optional phase profile: PROF instantiation only; wave 0 / lane 0 accumulates
// shader-clock deltas per phase and adds them to p.prof[] at the end.
#define PM_PROF_MARK(slot)                                   \
  if (PROF) {                                                \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_acc[slot] += now_ - prof_t;                         \
    prof_t = now_;                                           \
  }

// Static ISA census (scripts/isa_phase_census.py compiles this file with -DPM_ISA_MARKERS and counts the
// instructions between the comment markers); expands to nothing in every other build.
#ifdef PM_ISA_MARKERS
#define PM_MARK(name) asm volatile("; PMARK " name)
#else
#define PM_MARK(name)
#endif

template <int N1D, bool GEOM, bool FILTER_PHOTO, bool FILTER_GEOM, bool PROF>
__global__ void __launch_bounds__(256, 3) pm_sweep_kernel(const PmParams* __restrict__ pp) {
  // Batch of reference images: one launch, grid.y problems. Workgroups are dealt to the 8 XCDs
  // round-robin by linear id, so problem = id % batch keeps each problem's source-image band in
  // (at most 8 / batch) XCD L2s instead of spreading every problem over all eight. (Measured
  // alternative: XCD k sweeping the k-th eighth of the columns of every problem keeps the band in
  // L2 even better but loses 11 % to load imbalance between image regions.)
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned group = lin / gridDim.y;
  unsigned prob = lin - group * gridDim.y;
  if (pp[0].xcd_map == 1) {
    // neighbouring reference images on one XCD (batch a multiple of 8): XCD x = lin % 8 sweeps
    // problems x * nb/8 .. (x + 1) * nb/8 - 1, which share all but one of their source images
    prob = (prob & 7u) * (gridDim.y >> 3) + (prob >> 3);
  }
  const PmParams& p = pp[prob];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lds L;
  lds_bind(L, (lds_char*)smem, lds_offsets(p.C, p.S, p.radius, p.ntaps, p.num_samples, GEOM));
  const int tid = threadIdx.x, nt = blockDim.x;
  const int S = p.S, M = p.num_samples, C = p.C;
  const int RW = rot_width(p), RH = rot_height(p);
  const int col0 = group * C;
  const int ncols = min(C, RW - col0);  // valid columns of this group
  const float* iK = p.refInvK;

  unsigned long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_t = PROF ? __builtin_readcyclecounter() : 0ull;

  lds_load_poses(p, L, GEOM, tid, nt);

  // ---- backward messages for all rows (:976-989); stored in sel_out ----------
  for (int item = tid; item < ncols * S; item += nt) {
    const int c = item / S;
    const int s = item - c * S;
    float beta = 0.5f;
    for (int row = RH - 1; row >= 0; --row) {
      float* rec = p.rec + (size_t)pix_index(p, row, col0 + c) * p.rec_stride;
      beta = hmm_message<false>(p, rec[4 + s], beta);
      rec[p.sel_out_off + s] = beta;
    }
    L.fm[c * S + s] = 0.5f;
  }

  // ---- per-column state kept by the column's lane (:1022-1028) ---------------
  Rng rng;
  rng.x0 = rng.x1 = rng.x2 = rng.x3 = rng.x4 = rng.d = 0;
  const bool col_lane = tid < ncols;
  if (col_lane) {
    const int pix0 = pix_index(p, 0, col0 + tid);
    rng = rng_load(p.rng + (size_t)pix0 * kRngWords);
    const float* rec = p.rec + (size_t)pix0 * p.rec_stride;
    float sx, sy;
    normal_to_sweep(p.rot, rec[1], rec[2], sx, sy);
    lds_f32* h1 = L.hyp + (tid * 5 + 1) * 4;
    h1[0] = rec[0]; h1[1] = sx; h1[2] = sy; h1[3] = rec[3];
  }
  // reference tile rows [-r, r-1]; row r arrives in the first loop iteration
  for (int r = -p.radius; r < p.radius; ++r) tile_load_row(p, L, col0, r, tid, nt);
  __syncthreads();
  PM_PROF_MARK(0)

  unsigned evals2 = 0;
  for (int row = 0; row < RH; ++row) {
    // ---- P0: scroll the reference tile (LocalRefImage::Read, :357-410) -------
    tile_load_row(p, L, col0, row + p.radius, tid, nt);
    if (tid == 0) *L.ntasks = 0;
    __syncthreads();
    PM_PROF_MARK(1)

    // ---- P1: hypotheses (lane per column) + patch weights (all lanes) --------
    if (col_lane) {
      const int c = tid;
      const int col = col0 + c;
      const int pix = pix_index(p, row, col);
      const float* rec = p.rec + (size_t)pix * p.rec_stride;
      lds_f32* h = L.hyp + c * 20;
      // propagate the previous row's plane (:1047-1048)
      h[4] = propagate_depth(iK, h[4], h[6], h[7], (float)(row - 1), (float)row);
      // current parameters (:1051-1052)
      const float cd = rec[0];
      float cn0, cn1;
      normal_to_sweep(p.rot, rec[1], rec[2], cn0, cn1);
      const float cn2 = rec[3];
      // random parameters (:1055-1062)
      const float dmin = (1.0f - p.perturbation) * cd;
      const float dmax = (1.0f + p.perturbation) * cd;
      const float rd = rng_uniform(rng) * (dmax - dmin) + dmin;
      float rn0, rn1, rn2;
      perturb_normal(iK, row, col, p.perturbation_pi, cn0, cn1, cn2, rng, rn0, rn1, rn2);
      for (int m = 0; m < M; ++m) L.us[c * M + m] = rng_uniform(rng) - FLT_EPSILON;  // :1129
      h[0] = cd; h[1] = cn0; h[2] = cn1; h[3] = cn2;
      h[8] = rd; h[9] = rn0; h[10] = rn1; h[11] = rn2;
      h[12] = cd; h[13] = rn0; h[14] = rn1; h[15] = rn2;
      h[16] = rd; h[17] = cn0; h[18] = cn1; h[19] = cn2;
      lds_f32* cf = L.colf + c * 8;
      cf[0] = p.ref_sum[pix];
      cf[1] = p.ref_sqsum[pix];
      // ComputePointAtDepth (:1067-1068)
      cf[2] = cd * (iK[0] * col + iK[1]);
      cf[3] = cd * (iK[2] * row + iK[3]);
      cf[4] = cd;
    }
    patch_weights(p, L, row, tid, nt);
    for (int item = tid; item < ncols * 5 * S; item += nt) L.ncc[item] = -1.0f;
    __syncthreads();
    PM_PROF_MARK(2)

    // ---- P2: per-view selection priors (:1070-1104), lane per (column, view) --
    patch_weight_sums(p, L, ncols, tid, nt);
    for (int item = tid; item < ncols * S; item += nt) {
      const int c = item / S;
      const int s = item - c * S;
      const int col = col0 + c;
      const float* rec = p.rec + (size_t)pix_index(p, row, col) * p.rec_stride;
      const lds_f32* pose = L.poses + s * L.pstride;
      const lds_f32* h = L.hyp + c * 20;
      const lds_f32* cf = L.colf + c * 8;
      const float cost = rec[4 + s];
      const float beta = rec[p.sel_out_off + s];
      const float prev = rec[p.sel_in_off + s];
      L.costv[item] = cost;
      L.betav[item] = beta;
      L.prevv[item] = prev;
      const float alpha = hmm_message<true>(p, cost, L.fm[item]);
      const float sp = sel_prob_fn(alpha, beta, prev, p.prev_sel_prob_weight);
      float cos_tri, cos_inc;
      viewing_angles(pose, cf[2], cf[3], cf[4], h[1], h[2], h[3], cos_tri, cos_inc);
      const float tp = tri_prob(p, cos_tri);
      const float ip = inc_prob(p, cos_inc);
      float Hm[9];
      compose_homography(iK, pose, row, col, h[0], h[1], h[2], h[3], Hm);
      const float rp = res_prob(Hm, (float)row, (float)col, p.radius);
      L.q[item] = sp * tp * ip * rp;
    }
    __syncthreads();
    PM_PROF_MARK(3)

    // ---- P3a: TransformPDFToCDF (:683-696), sequential sum order, lane per column
    if (col_lane) {
      const int c = tid;
      lds_f32* q = L.q + c * S;
      float prob_sum = 0.0f;
#pragma unroll 4
      for (int i = 0; i < S; ++i) prob_sum += q[i];
      const float inv_prob_sum = 1.0f / prob_sum;
      float cum = 0.0f;
#pragma unroll 4
      for (int i = 0; i < S; ++i) {
        cum += q[i] * inv_prob_sum;
        q[i] = cum;
      }
    }
    __syncthreads();
    // ---- P3b: Monte-Carlo view draws (:1128-1138), lane per (column, draw) ----
    for (int item = tid; item < ncols * M; item += nt) {
      const int c = item / M;
      const float u = L.us[item];
      const lds_f32* q = L.q + c * S;
      int src = -1;
      for (int s = 0; s < S; ++s) {
        if (q[s] > u) { src = s; break; }
      }
      L.sv[item] = src;
    }
    __syncthreads();
    // ---- P3c: one task set per distinct drawn view, lane per (column, view) ---
    for (int item = tid; item < ncols * S; item += nt) {
      const int c = item / S;
      const int s = item - c * S;
      bool drawn = false;
      for (int m = 0; m < M; ++m) drawn |= (L.sv[c * M + m] == s);
      if (drawn) {
        const int n_new = GEOM ? 5 : 4;
        const int base = __hip_atomic_fetch_add(L.ntasks, n_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int i = 1; i < 5; ++i) L.tasks[base + i - 1] = task_pack(c, i, s, 0);
        if (GEOM) L.tasks[base + 4] = task_pack(c, 0, s, 1);
      }
    }
    __syncthreads();
    PM_PROF_MARK(4)

    // ---- P4: NCC of hypotheses 1..4 against the drawn views (:1157-1172) -----
    evals2 += (unsigned)run_tasks<N1D, GEOM>(p, L, row, col0, tid, nt);
    __syncthreads();
    if (tid == 0) *L.ntasks = 0;
    __syncthreads();
    PM_PROF_MARK(5)

    // ---- P5a: accumulate in draw order (:1144-1172), lane per (column, hypothesis)
    for (int item = tid; item < ncols * 5; item += nt) {
      const int c = item / 5;
      const int i = item - c * 5;
      float acc = 0.0f;
      for (int m = 0; m < M; ++m) {
        const int src = L.sv[c * M + m];
        if (src < 0) continue;
        acc += (i == 0) ? L.costv[c * S + src] : L.ncc[(c * 5 + i) * S + src];
        if (GEOM) acc += p.geom_reg * L.geo[(c * 5 + i) * S + src];
      }
      L.csum[item] = acc;
    }
    __syncthreads();
    // ---- P5b: argmin, store, next row's previous state (:1176-1182,1279-1282) --
    if (col_lane) {
      const int c = tid;
      int min_idx = 0;
      float min_cost = L.csum[c * 5];
#pragma unroll
      for (int i = 1; i < 5; ++i) {
        const float ci = L.csum[c * 5 + i];
        if (ci <= min_cost) { min_cost = ci; min_idx = i; }
      }
      L.best[c] = min_idx;
      const lds_f32* hb = L.hyp + (c * 5 + min_idx) * 4;
      const float bd = hb[0], b0 = hb[1], b1 = hb[2], b2 = hb[3];
      float* rec = p.rec + (size_t)pix_index(p, row, col0 + c) * p.rec_stride;
      float nx, ny;
      normal_from_sweep(p.rot, b0, b1, nx, ny);
      rec[0] = bd; rec[1] = nx; rec[2] = ny; rec[3] = b2;
      lds_f32* h1 = L.hyp + (c * 5 + 1) * 4;
      h1[0] = bd; h1[1] = b0; h1[2] = b1; h1[3] = b2;
    }
    __syncthreads();
    // ---- P5c: winner vs. the views not evaluated yet, lane per (column, view) --
    for (int item = tid; item < ncols * S; item += nt) {
      const int c = item / S;
      const int s = item - c * S;
      const int k = L.best[c];
      if (k != 0 && L.ncc[(c * 5 + k) * S + s] < 0.0f) {
        const int base = __hip_atomic_fetch_add(L.ntasks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        L.tasks[base] = task_pack(c, k, s, 0);
      }
    }
    __syncthreads();
    PM_PROF_MARK(6)

    // ---- P6: NCC of the winner against the remaining views (:1188-1197) ------
    evals2 += (unsigned)run_tasks<N1D, false>(p, L, row, col0, tid, nt);
    __syncthreads();
    PM_PROF_MARK(7)

    // ---- P7: cost map, forward message, selection probability (:1186-1207) ---
    for (int item = tid; item < ncols * S; item += nt) {
      const int c = item / S;
      const int s = item - c * S;
      const int col = col0 + c;
      const int k = L.best[c];
      float* rec = p.rec + (size_t)pix_index(p, row, col) * p.rec_stride;
      float cost;
      if (k == 0) {
        cost = L.costv[item];
      } else {
        cost = L.ncc[(c * 5 + k) * S + s];
        rec[4 + s] = cost;
      }
      const float alpha = hmm_message<true>(p, cost, L.fm[item]);
      const float prob = sel_prob_fn(alpha, L.betav[item], L.prevv[item], p.prev_sel_prob_weight);
      L.fm[item] = alpha;
      rec[p.sel_out_off + s] = prob;
      if (FILTER_PHOTO || FILTER_GEOM) {
        // :1209-1265
        const lds_f32* hb = L.hyp + (c * 5 + 1) * 4;  // == best (stored in P5)
        const lds_f32* pose = L.poses + s * L.pstride;
        const float bp0 = hb[0] * (iK[0] * col + iK[1]);
        const float bp1 = hb[0] * (iK[2] * row + iK[3]);
        const float bp2 = hb[0];
        float cos_tri, cos_inc;
        viewing_angles(pose, bp0, bp1, bp2, hb[1], hb[2], hb[3], cos_tri, cos_inc);
        int ok = 0;
        if (!(cos_tri > p.filter_cos_min_tri || cos_inc <= 0.0f)) {
          const float min_ncc_prob = ncc_prob(p, 1.0f - p.filter_min_ncc);
          bool photo_ok = true, geom_ok = true;
          if (FILTER_PHOTO) photo_ok = prob >= min_ncc_prob;
          if (FILTER_GEOM)
            geom_ok = geom_cost(p, pose, s, (float)row, (float)col, hb[0]) <= p.filter_geom_max_cost;
          ok = (photo_ok && geom_ok) ? 1 : 0;
        }
        L.flags[item] = ok;
      }
    }
    if (FILTER_PHOTO || FILTER_GEOM) {
      __syncthreads();
      // ---- P8: consistency count (:1267-1275) ---------------------------------
      if (col_lane) {
        const int c = tid;
        int num = 0;
        for (int s = 0; s < S; ++s) num += L.flags[c * S + s];
        const int pix = pix_index(p, row, col0 + c);
        if (num < p.filter_min_num_consistent) {
          float* rec = p.rec + (size_t)pix * p.rec_stride;
          rec[0] = 0.0f; rec[1] = 0.0f; rec[2] = 0.0f; rec[3] = 0.0f;
        } else {
          for (int s = 0; s < S; ++s)
            if (L.flags[c * S + s]) p.mask[(size_t)s * p.W * p.H + pix] = 1;
        }
      }
    }
    __syncthreads();
    PM_PROF_MARK(8)
  }

  if (col_lane) {
    rng_store(p.rng + (size_t)pix_index(p, 0, col0 + tid) * kRngWords, rng);  // :1285-1287
  }
  if (tid == 0 && p.evals) atomicAdd(p.evals, (unsigned long long)evals2);
  if (PROF && tid == 0 && p.prof) {
    for (int i = 0; i < 10; ++i) atomicAdd(p.prof + i, prof_acc[i]);
  }
}

// ---------------------------------------------------------------------------
// Wave-per-column-group sweep kernel for the 11 x 11 window (sweep_wave_body): every wave owns C adjacent columns
// of the sweep frame and walks them row by row without ever meeting another wave -- the phases of a row step are
// separated by LDS fences only. Four waves form a workgroup and share one LDS copy of the read-only per-problem
// tables (pm_sweep_quad_kernel). Other windows take the generic kernel above.
//
// What a row step does NOT do (round 6): everything of the reference's row step that depends only on the state the
// sweep starts from -- the column's random numbers, i.e. the perturbed depth and normal (:1055-1062) and the M
// uniforms of the view draws (:1129) -- is produced for all rows by pm_draw_kernel (lane per column, 64 columns per
// wave instead of 2 of 64 lanes) into PmParams::draws before the sweep launch. The lane-per-column steps that remain
// (CDF, argmin) are sequential sums the reference's order pins; draws, task lists and the Monte-Carlo sums are
// organised so that no lane loops over what another lane could hold (binary search in the CDF, a drawn-view bitmap,
// ballots instead of atomics, zero slots instead of "no view" branches).
// ---------------------------------------------------------------------------
constexpr int kQuadWaves = 4;   // waves per workgroup
// Five workgroups = 20 waves per CU for the photometric sweeps at S = 20: 96 VGPRs (two loop-invariant register pairs
// live in scratch and are re-read once per NCC batch, outside the tap rounds) and 31.7 KB of LDS with 56 task slots
// per batch (a row has ~42 + ~30 tasks at C = 2); measured 490 -> 480 ms per 16-image launch against four workgroups
// with 64 slots. The geometric pass (38 KB of LDS) stays at four.
#ifndef PM_QUAD_CAP
#define PM_QUAD_CAP 56
#endif
#ifndef PM_QUAD_OCC
#define PM_QUAD_OCC 5
#endif
constexpr int kQuadThCap = PM_QUAD_CAP;  // NCC task slots per batch (homographies in LDS)
constexpr int kQuadOcc = PM_QUAD_OCC;    // workgroups per CU the photometric build is compiled for

__device__ __forceinline__ uint32_t task16_pack(int c, int i, int s, int geom_only) {
  return ((uint32_t)c << 13) | ((uint32_t)geom_only << 12) | ((uint32_t)i << 9) | (uint32_t)s;
}

// NCC tasks of one phase: 4 hypotheses per distinct drawn view, or the winner against <= S views
__host__ __device__ inline int wave_max_tasks(int C, int S, int M) {
  const int ms = M < S ? M : S;
  return C * (4 * ms > S ? 4 * ms : S);
}

// LDS carve-up of a workgroup of `nw` waves, each sweeping its own column group. The read-only tables that are
// the same for every column group of a problem -- pose records, packed-image slots, tap tables: 3.7 KB at
// S = 20 -- exist once per workgroup at the start of the block; everything else is private to a wave and repeats
// with `priv_stride` (the offsets of the private items are those of wave 0). `cap` = NCC task slots per batch.
__host__ __device__ inline LdsOffsets lds_offsets_wave(int C, int S, int radius, int ntaps, int M, bool geom,
                                                       int cap, int nw) {
  LdsOffsets o;
  uint32_t off = 0;
  auto take = [&](uint32_t bytes) {
    const uint32_t at = off;
    off += (bytes + 15u) & ~15u;
    return at;
  };
  const int win = 2 * radius + 1;
  const int tw = C + 2 * radius;
  const int max_tasks = wave_max_tasks(C, S, M);
  o.ring = 0;
  o.poses = take(4u * S * lds_pose_stride(geom));
  o.fpb = take(8u * S);                       // packed images: 32-bit slots of the buffer resource (Lds::fpo) or addresses (fpb)
  o.tapg = take(4u * 512);                    // tap tables: dx, dy (NCC warp), packed tile offsets, spatial exponent (patch weights)
  const uint32_t shared = off;
  o.tile = take(4u * win * tw);
  o.wgt = take(4u * C * tap_stride(ntaps));
  o.refc = take(4u * C * tap_stride(ntaps));
  o.fm = take(4u * C * S);
  o.q = take(4u * C * S);
  o.betav = take(4u * C * S);
  o.prevv = take(4u * C * S);
  o.ncc = take(4u * C * 5 * (S + 1));         // Lds::cost5: [C][5][S + 1] cost map row (hypothesis 0), NCC of hypotheses 1..4, zero slot
  o.costv = o.ncc;
  o.geo = take(geom ? 4u * C * 5 * (S + 1) : 0u);
  o.hyp = take(4u * C * 20);
  o.colf = take(4u * C * 8);
  o.flags = take(1u * C * S);                 // filter flags
  o.us = o.flags;
  o.sv = take(4u * C * M);
  o.best = take(4u * C);
  o.csum = take(4u * C * 5);
  o.drawn = take(4u * C * ((S + 31) / 32));   // bitmap of the views drawn in this row
  o.tasks = take(2u * max_tasks + (geom ? 2u * C * S : 0u));  // 16-bit task words: NCC tasks, then
                                                              // (GEOM) the geometric-cost-only list
  o.th = take(36u * (uint32_t)cap);
  o.ntasks = 0;
  o.tin = take(4u * (uint32_t)cap);           // Lds::desc: pass-B descriptor per slot of the batch (inside-first order)
  o.priv_stride = off - shared;
  o.total = shared + (uint32_t)nw * o.priv_stride;
  return o;
}

// Tap tables of the 11 x 11 window, [4][128] in LDS, filled once per workgroup; tap t = wrow * 11 + wcol, or with
// `transpose` (odd sweep directions, patch_weights) t = wcol * 11 + wrow:
//   [0]   (float) wcol * step        window offsets of the NCC warp (ncc_front reads taps j + 16 k)
//   [1]   (float) wrow * step
//   [2]   (int)   (wrow * step) << 16 | wcol * step                  tile offsets of the patch-weight pass
//   [3]   (float) -(wr^2 + wc^2) * spatial_norm, wr = wrow * step - radius: the spatial term of the bilateral weight
//         (BilateralWeightComputer::Compute, gpu_mat_ref_image.h:70-90: the same product, taken once per kernel)
__device__ __forceinline__ void tap_tables_init(lds_f32* tapg, int tid, int nthreads, int step, int radius,
                                                float spatial_norm, bool transpose) {
  for (int t = tid; t < 128; t += nthreads) {
    const int tt = t < 121 ? t : 0;
    int wrow = tt / 11;
    int wcol = tt - wrow * 11;
    if (transpose) {
      const int sw = wrow; wrow = wcol; wcol = sw;
    }
    tapg[t] = (float)(wcol * step);
    tapg[128 + t] = (float)(wrow * step);
    ((lds_i32*)tapg)[256 + t] = ((wrow * step) << 16) | (wcol * step);
    const float wr = (float)(wrow * step - radius), wc = (float)(wcol * step - radius);
    const float sds = wr * wr + wc * wc;
    tapg[384 + t] = -sds * spatial_norm;
  }
}

// item / n for 0 <= item < 4096, n <= 512 with inv_n = 1.0f / n (one IEEE division per kernel): the quotient's
// fraction lies in [0.5 / n, 1 - 0.5 / n], far from the 2^-21 the float product can be off by.
__device__ __forceinline__ int item_div(int item, float inv_n) { return (int)(((float)item + 0.5f) * inv_n); }

// lanes of the wave below this one whose bit is set in `mask`
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Load one sweep-frame row of the reference image into ring slot `slot` of the LDS tile.
__device__ __forceinline__ void tile_load_row_slot(const PmParams& p, const Lds& L, int col0, int row, int slot,
                                                   int tid) {
  const int tw = p.C + 2 * p.radius;
  for (int lc = tid; lc < tw; lc += 64) L.tile[slot * tw + lc] = ref_texel(p, row, col0 - p.radius + lc);
}

// Bilateral weights + reference colours of the patches centred on (row, col0 + c), 11 x 11 window (patch_weights
// with the per-tap integer divisions and the spatial term taken from the tap tables). slot_c / slot_top = ring slots
// of rows `row` and `row - radius`. Taps 121..127 of every column keep the zeros written once before the row loop.
__device__ __forceinline__ void patch_weights_wave(const PmParams& p, const Lds& L, int slot_c, int slot_top,
                                                   int tid) {
  const int win = 2 * p.radius + 1;
  const int tw = p.C + 2 * p.radius;
  const lds_i32* tapi = (const lds_i32*)L.tapg + 256;
  const lds_f32* taps = L.tapg + 384;
  for (int item = tid; item < p.C * 128; item += 64) {
    const int c = item >> 7;
    const int tap = item & 127;
    if (tap >= 121) continue;
    const int ti = tapi[tap];
    int slot = slot_top + (ti >> 16);
    if (slot >= win) slot -= win;
    const float center = L.tile[slot_c * tw + c + p.radius];
    const float color = L.tile[slot * tw + c + (ti & 0xffff)];
    const float cd = center - color;
    L.wgt[item] = pm_exp(taps[tap] - cd * cd * p.color_norm);
    L.refc[item] = color;
  }
}

// Does every tap of the patch with (centred) homography Hm provably fall on texels x in [0, w - 1],
// y in [0, h - 1] of the source image? The window maps to a convex quadrilateral when the projective
// divisor is positive at its four corners, so the taps lie inside the corners' bounding box; one
// texel of margin absorbs the rounding of the per-tap evaluation. NaNs compare false.
__device__ __forceinline__ bool patch_inside(const PmParams& p, const float Hm[9]) {
  // With z > 0 at a corner, 1 <= x / z <= w - 2 is z <= x <= (w - 2) z: eight divisions saved per task. The
  // flag only selects the addressing variant of the gathers (both give the same texels for a patch that is
  // inside; the one-texel margin dwarfs the rounding of the products), it never changes a result -- so the corner
  // values are taken incrementally (two products and three sums per coordinate instead of eight and eight).
  const float e = (float)(2 * p.radius);  // window extent: taps at offsets 0 .. 2 r
  const float wx = (float)(p.src_w - 2), wy = (float)(p.src_h - 2);
  float x[4], y[4], z[4];
  {
    const float a = Hm[0] * e, b = Hm[1] * e;
    x[0] = Hm[2]; x[1] = a + x[0]; x[2] = b + x[0]; x[3] = a + x[2];
  }
  {
    const float a = Hm[3] * e, b = Hm[4] * e;
    y[0] = Hm[5]; y[1] = a + y[0]; y[2] = b + y[0]; y[3] = a + y[2];
  }
  {
    const float a = Hm[6] * e, b = Hm[7] * e;
    z[0] = Hm[8]; z[1] = a + z[0]; z[2] = b + z[0]; z[3] = a + z[2];
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    ok = ok && (z[k] > 0.0f) && x[k] >= z[k] && y[k] >= z[k] && x[k] <= wx * z[k] && y[k] <= wy * z[k];
  return ok;
}

// Synchronisation point between the lane-parallel phases of ONE wave. A single-wave workgroup's
// __syncthreads() compiles to the two fences without an s_barrier; the multi-wave workgroup spells that out so
// that its waves stay independent of each other.
template <int NW>
__device__ __forceinline__ void wave_sync() {
  if (NW == 1) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// Mailbox of a wave pair (pm_sweep_pair_kernel): the words behind Lds::best[C] -- its 16-byte slot holds C <= 2 entries,
// the pair kernel runs one column per group -- carry the size of the published batch and whether it is the phase's last.
constexpr int kPairNb = 2, kPairLast = 3;

// Pass B of a batch of NCC tasks, 16-lane group per task: one round = four tasks; all eight gathers of a lane are in
// flight before the first texel is consumed (ncc_front / ncc_back). Control flow is wave-uniform -- every group runs
// whole rounds, a group without a task in the last round recomputes the batch's last task and drops the result -- so
// that the DPP rows are always fully active. A wave runs the rounds first, first + stride, ...: stride 1 normally;
// with a helper wave (pm_sweep_pair_kernel) the two waves of a column group take alternate rounds, the helper the
// batch's last one. The sums of a task replace its homography in L.th (slots are disjoint between rounds).
template <bool MUBUF>
__device__ __forceinline__ void ncc_rounds_wave(const PmParams& p, const Lds& L, const v4i srd, const lds_f32* G, int tid,
                                                int nb, int first, int stride) {
  const int g = tid >> 4, j = tid & 15;
  // slot offset of texel (0, 0) relative to entry (0, 0): one strip (kFpRingX entries) and kFpRingY rows
  const uint32_t origin = 4u * ((uint32_t)p.fp_rows1 + 1u) + 4u * (uint32_t)kFpRingY;
  const uint32_t gorigin = fp_index(kFpRingX, kFpRingY, (unsigned)p.fp_rows1);  // the same as an entry index
  const int rounds = (nb + 3) >> 2;
  for (int r = first; r < rounds; r += stride) {
    const int tr = g + 4 * r;
    const bool own = tr < nb;
    const uint32_t d = L.desc[own ? tr : nb - 1];
    // wave-uniform: the unclamped addressing only when all four patches of the round are inside
    // (a recomputed task may already hold its sums instead of its homography: it must take the
    // clamping path, where any coordinate is safe and the result is dropped)
    const bool fast = __all(own && (d & 0x80u) != 0u) != 0;
    const uint32_t slot = MUBUF ? L.fpo[d >> 16] : 0u;
    gbl_u32* gbase = MUBUF ? nullptr : (gbl_u32*)L.fpb[d >> 16];
    const lds_f32* H = L.th + (d & 63u) * 9u;
    launder_lds(H);  // one address register for the nine reads (offsets 0..32) instead of a base + constant each
    NccStage A;
    uint32_t tex[8];
    if (fast) ncc_front<true, MUBUF>(p, srd, H, slot + origin, gbase + gorigin, G, j, A, tex);
    else ncc_front<false, MUBUF>(p, srd, H, slot, gbase, G, j, A, tex);
    __builtin_amdgcn_sched_barrier(0);
    TapRegs R;
    const int c128 = (int)((d >> 8) & 0xffu) * 128;
    tap_regs_load(R, L.wgt + c128, L.refc + c128, j);
    float s_sum, s_sq, s_ref;
    ncc_back(A, tex, R, j, s_sum, s_sq, s_ref);
    if (j == 0 && own) {
      lds_f32* Hw = L.th + (d & 63u) * 9u;
      Hw[0] = s_sum;  // the homography of this task is no longer needed
      Hw[1] = s_sq;
      Hw[2] = s_ref;
    }
  }
}

// Run the `n` queued NCC tasks (and, with GEOM, the `ng` entries of the geometric-cost-only list) of one phase.
template <bool GEOM, int NW, int CAP, bool MUBUF, bool PROF, int HELP = 1>
__device__ __forceinline__ void run_tasks_wave(const PmParams& p, const Lds& L, const v4i srd, int row, int col0,
                                               int tid, int n, int ng, unsigned& evals, unsigned long long* prof_acc,
                                               unsigned long long& prof_t, const int prof_slot0) {
  const lds_f32* G = L.tapg;
  evals += (unsigned)n;
  const LDS_AS uint16_t* tasks = (const LDS_AS uint16_t*)L.tasks;
  const int S = p.S, S1 = p.S + 1;
  if (GEOM) {
    // geometric consistency cost of hypothesis 0 against the drawn views (no NCC: cached cost map)
    const LDS_AS uint16_t* gtasks = tasks + wave_max_tasks(p.C, S, p.num_samples);
    for (int t = tid; t < ng; t += 64) {
      const uint32_t task = gtasks[t];
      const int c = task >> 13, i = (task >> 9) & 7, s = task & 0x1ff;
      const lds_f32* h = L.hyp + (c * 5 + i) * 4;
      L.geo[(c * 5 + i) * S1 + s] = geom_cost(p, L.poses + s * L.pstride, s, (float)row, (float)(col0 + c), h[0]);
    }
  }
  if (HELP > 1 && n == 0) {  // the helper waits for one batch per phase at least
    if (tid == 0) {
      L.best[kPairNb] = 0;
      L.best[kPairLast] = 1;
    }
    __syncthreads();
    __syncthreads();
  }
  for (int base = 0; base < n; base += CAP) {
    const int nb = min(CAP, n - base);
    PM_MARK("passA");
    // pass A, lane per task: homography of the (hypothesis, view) pair (+ geometric cost)
    bool inside = false;
    uint32_t desc = 0;
    if (tid < nb) {
      const uint32_t task = tasks[base + tid];
      const int c = task >> 13;
      const int i = (task >> 9) & 7;
      const int s = task & 0x1ff;
      const lds_f32* h = L.hyp + (c * 5 + i) * 4;
      const lds_f32* pose = L.poses + s * L.pstride;
      const int col = col0 + c;
      float Hm[9];
      compose_homography(p.refInvK, pose, row, col, h[0], h[1], h[2], h[3], Hm);
      centre_homography(Hm, row, col, p.radius);
      for (int k = 0; k < 9; ++k) L.th[tid * 9 + k] = Hm[k];
      inside = patch_inside(p, Hm);
      desc = (uint32_t)tid | (inside ? 0x80u : 0u) | ((uint32_t)c << 8) | ((uint32_t)s << 16);
      if (GEOM) L.geo[(c * 5 + i) * S1 + s] = geom_cost(p, pose, s, (float)row, (float)col, h[0]);
    }
    {
      // Order of the batch's tasks for pass B: the tasks whose patch is inside the source image first. A round
      // takes the cheaper unclamped addressing only when all four of its patches are inside; with the tasks in
      // list order one outside patch in four spoils the round, sorted they collect in the last rounds. Results
      // are stored per task, so the order cannot change a bit. A slot's descriptor = everything pass B needs to
      // know about its task: slot of the homography (bits 0-5), inside flag (7), column (8-15), view (16-).
      const unsigned long long m1 = __ballot(inside ? 1 : 0);
      const unsigned long long valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
      const unsigned long long m0 = valid & ~m1;
      if (tid < nb) L.desc[inside ? lanes_below(m1) : __popcll(m1) + lanes_below(m0)] = desc;
    }
    wave_sync<NW>();
    // pass B (ncc_rounds_wave); with a helper wave (HELP = 2, pm_sweep_pair_kernel) this wave takes every other round
    if (HELP > 1) {
      if (tid == 0) {
        L.best[kPairNb] = nb;
        L.best[kPairLast] = base + CAP >= n ? 1 : 0;
      }
      __syncthreads();  // the batch is published: homographies, descriptors, its size
    }
    PM_PROF_MARK(prof_slot0 - 1)
    PM_MARK("passB");
    {
      const int rounds = (nb + 3) >> 2;
      ncc_rounds_wave<MUBUF>(p, L, srd, G, tid, nb, HELP > 1 ? (rounds & 1) : 0, HELP);
    }
    if (HELP > 1) __syncthreads();  // the helper's sums are in
    wave_sync<NW>();
    PM_PROF_MARK(prof_slot0)
    PM_MARK("finish");
    // lane per task: normalisation, variances, square root, division
    if (tid < nb) {
      const uint32_t task = tasks[base + tid];
      const int c = task >> 13;
      const int i = (task >> 9) & 7;
      const int s = task & 0x1ff;
      L.cost5[(c * 5 + i) * S1 + s] = ncc_finish(L.th[tid * 9 + 0], L.th[tid * 9 + 1], L.th[tid * 9 + 2],
                                                 L.colf[c * 8 + 0], L.colf[c * 8 + 1], L.colf[c * 8 + 5]);
    }
    wave_sync<NW>();
    PM_PROF_MARK(prof_slot0 + 1)
  }
}

// Phase profile of the wave kernel (PROF instantiation, pm_enable_phase_profile): every wave accumulates shader-clock
// deltas per phase in scalar registers and adds them to p.prof[] when it retires. Slots:
[[maybe_unused]] constexpr int kProfSetup = 0, kProfP0 = 1, kProfP1 = 2, kProfP1w = 3, kProfP2 = 4, kProfP3a = 5, kProfP3b = 6,
              kProfP3c = 7, kProfP4A = 8, kProfP4B = 9, kProfP4F = 10, kProfP5a = 11, kProfP5b = 12, kProfP5c = 13,
              kProfP6A = 14, kProfP6B = 15, kProfP6F = 16, kProfP7 = 17, kProfP8 = 18, kProfSlots = kPmProfSlots;

// HELP = 2 (pm_sweep_pair_kernel, NW = 2): the two waves of a workgroup serve ONE column group. Wave 0 walks the
// rows as always; wave 1 is bound to the same LDS region and only runs pass B of the NCC phases, every other round of
// four evaluations (ncc_rounds_wave), between two workgroup barriers per batch: half the serial time of a row is
// pass B. For launches that cannot fill the GPU with one wave per column (ONE 2560 x 1920 problem = 2 560 waves for
// 5 120 slots: how the reference's controller drives the seam).
template <bool GEOM, bool FILTER_PHOTO, bool FILTER_GEOM, int NW, int CAP, bool MUBUF, bool PROF = false, int HELP = 1>
__device__ __forceinline__ void sweep_wave_body(const PmParams* __restrict__ pp) {
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
  unsigned group = lin / gridDim.y;
  unsigned prob = lin - group * gridDim.y;
  if (pp[0].xcd_map == 1) prob = (prob & 7u) * (gridDim.y >> 3) + (prob >> 3);  // see pm_sweep_kernel
  const PmParams& p = pp[prob];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lds L;
  // NW > 1: wave w of the workgroup sweeps column group NW * group + w out of its own LDS region; the pose
  // records, packed-image slots and tap tables are shared (lds_offsets_wave). After the one workgroup barrier
  // behind their initialisation the waves never meet again: every later synchronisation point is
  // wave_sync<NW>(), a memory fence without s_barrier -- exactly what __syncthreads() compiles to in the
  // single-wave workgroups.
  const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  {
    const LdsOffsets o = lds_offsets_wave(p.C, p.S, p.radius, p.ntaps, p.num_samples, GEOM, CAP, HELP > 1 ? 1 : NW);
    lds_bind(L, (lds_char*)smem + (HELP > 1 ? 0 : wave) * o.priv_stride, o);
    L.poses = (lds_f32*)((lds_char*)smem + o.poses);
    L.fpo = (lds_u32*)((lds_char*)smem + o.fpb);
    L.fpb = (lds_u64*)((lds_char*)smem + o.fpb);
    L.tapg = (lds_f32*)((lds_char*)smem + o.tapg);
  }
  if (NW > 1 && HELP == 1) group = group * NW + wave;
  const int tid_entry = threadIdx.x & 63;
  const int tid = tid_entry;
  constexpr int nt = 64;
  const int S = p.S, M = p.num_samples, C = p.C;
  const int S1 = S + 1;            // row length of cost5 / geo: S views + the zero slot a draw without a view reads
  const int DW = (S + 31) >> 5;    // words per column of the drawn-view bitmap
  const float inv_S = 1.0f / (float)S, inv_M = 1.0f / (float)M;
  const int RW = rot_width(p), RH = rot_height(p);
  const int col0 = group * C;
  const int ncols = min(C, RW - col0);
  const int win = 2 * p.radius + 1;
  const float* iK = p.refInvK;
  const v4i srd = MUBUF ? fp_resource(p) : (v4i)(0);
  gbl_f32* draws = (gbl_f32*)p.draws;
  const int dstride = pm_draw_stride(M);
  if (wave == 0) tap_tables_init(L.tapg, tid, 64, p.step, p.radius, p.spatial_norm, (p.rot & 1) != 0);
  {
    // pose records and packed-image slots, once per workgroup
    L.pstride = lds_pose_stride(GEOM);
    for (int i = threadIdx.x; i < p.S * L.pstride; i += 64 * NW) {
      const int s = i / L.pstride;
      L.poses[i] = p.poses[s * kPoseStride + (i - s * L.pstride)];
    }
    for (int i = threadIdx.x; i < p.S; i += 64 * NW) {
      if (MUBUF) L.fpo[i] = p.src_fp_off[i];
      else L.fpb[i] = (uint64_t)p.src_fp_tab[i];
    }
  }
  if (NW > 1) {
    __syncthreads();                  // the only workgroup barrier of the kernel (HELP = 1)
    if (col0 >= RW) return;           // surplus wave of the last workgroup (grid.x = ceil(groups / NW))
  }
  if (HELP > 1 && wave != 0) {
    // the helper: per row and NCC phase, the batches wave 0 publishes (two barriers each, run_tasks_wave)
    for (int row = 0; row < RH; ++row)
      for (int phase = 0; phase < 2; ++phase) {
        int last;
        do {
          __syncthreads();
          const int nb = L.best[kPairNb];
          last = L.best[kPairLast];
          const int rounds = (nb + 3) >> 2;
          if (nb > 0) ncc_rounds_wave<MUBUF>(p, L, srd, L.tapg, tid, nb, (rounds - 1) & 1, HELP);
          __syncthreads();
        } while (!last);
      }
    return;
  }

  // ---- backward messages for all rows (:976-989); stored in sel_out ----------
  for (int item = tid; item < ncols * S; item += nt) {
    const int c = item_div(item, inv_S);
    const int s = item - c * S;
    float beta = 0.5f;
    for (int row = (PM_ABLATE(p) & 4) ? -1 : RH - 1; row >= 0; --row) {
      float* rec = p.rec + (size_t)pix_index(p, row, col0 + c) * p.rec_stride;
      beta = hmm_message<false>(p, rec[4 + s], beta);
      rec[p.sel_out_off + s] = beta;
    }
    L.fm[c * S + s] = 0.5f;
  }

  // ---- per-column state kept across rows: the previous row's plane (:1022-1028) ----
  if (tid < ncols) {
    const int pix0 = pix_index(p, 0, col0 + tid);
    const float* rec = p.rec + (size_t)pix0 * p.rec_stride;
    float sx, sy;
    normal_to_sweep(p.rot, rec[1], rec[2], sx, sy);
    lds_f32* h1 = L.hyp + (tid * 5 + 1) * 4;
    h1[0] = rec[0]; h1[1] = sx; h1[2] = sy; h1[3] = rec[3];
  }
  // written once: the zero slots behind the S views of every (column, hypothesis) row and the seven padding taps
  for (int i = tid; i < C * 5; i += nt) {
    L.cost5[i * S1 + S] = 0.0f;
    if (GEOM) L.geo[i * S1 + S] = 0.0f;
  }
  for (int i = tid; i < C * 8; i += nt) {
    const int at = (i >> 3) * 128 + 120 + (i & 7);
    if ((i & 7) != 0) { L.wgt[at] = 0.0f; L.refc[at] = 0.0f; }
  }
  for (int r = -p.radius; r < p.radius; ++r) tile_load_row(p, L, col0, r, tid, nt);
  wave_sync<NW>();

  const int tid0 = tid_entry;
  unsigned long long prof_acc[kProfSlots] = {};
  unsigned long long prof_t = PROF ? __builtin_readcyclecounter() : 0ull;
  unsigned evals = 0;  // NCC evaluations of this wave (< 2^32: RH * C * (4 M + S) per sweep)
  // ring slots of the tile rows row - radius, row, row + radius (advance by one per row, modulo the window)
  int slot_top = p.radius + 1 == win ? 0 : p.radius + 1, slot_c = 0, slot_new = p.radius;
  const int step0 = 1 << (31 - __builtin_clz(S));  // largest power of two <= S (CDF search)
  for (int row = 0; row < RH; ++row) {
    // The lane id is laundered through an empty asm once per row: everything the phases derive from
    // it (item -> column / view, LDS addresses) is then recomputed per row instead of being hoisted
    // out of the row loop and held in VGPRs across the NCC loop, which needs them.
    int tid = tid0;
    launder_vgpr(tid);
    const bool col_lane = tid < ncols;
    if (p.trace && (row & 127) == 0 && tid == 0)  // debug: pm_enable_progress_trace
      p.trace[(size_t)group * p.trace_stride + (row >> 7)] = __builtin_amdgcn_s_memrealtime();
    PM_PROF_MARK(row == 0 ? kProfSetup : kProfP8)
    PM_MARK("P0");
    // ---- P0: scroll the reference tile (LocalRefImage::Read, :357-410) -------
    tile_load_row_slot(p, L, col0, row + p.radius, slot_new, tid);
    wave_sync<NW>();

    PM_PROF_MARK(kProfP0)
    PM_MARK("P1");
    // ---- P1: hypotheses, lane per column (:1047-1068); the random ones come from pm_draw_kernel ----
    if (col_lane && !(PM_ABLATE(p) & 2)) {
      const int c = tid;
      const int col = col0 + c;
      const int pix = pix_index(p, row, col);
      const float* rec = p.rec + (size_t)pix * p.rec_stride;
      gbl_f32* dr = draws + ((size_t)row * RW + col) * dstride;
      lds_f32* h = L.hyp + c * 20;
      h[4] = propagate_depth(iK, h[4], h[6], h[7], (float)(row - 1), (float)row);
      const float cd = rec[0];
      float cn0, cn1;
      normal_to_sweep(p.rot, rec[1], rec[2], cn0, cn1);
      const float cn2 = rec[3];
      const float rd = dr[0], rn0 = dr[1], rn1 = dr[2], rn2 = dr[3];
      h[0] = cd; h[1] = cn0; h[2] = cn1; h[3] = cn2;
      h[8] = rd; h[9] = rn0; h[10] = rn1; h[11] = rn2;
      h[12] = cd; h[13] = rn0; h[14] = rn1; h[15] = rn2;
      h[16] = rd; h[17] = cn0; h[18] = cn1; h[19] = cn2;
      lds_f32* cf = L.colf + c * 8;
      cf[0] = p.ref_sum[pix];
      cf[1] = p.ref_sqsum[pix];
      cf[2] = cd * (iK[0] * col + iK[1]);
      cf[3] = cd * (iK[2] * row + iK[3]);
      cf[4] = cd;
    }
    PM_PROF_MARK(kProfP1)
    PM_MARK("P1w");
    patch_weights_wave(p, L, slot_c, slot_top, tid);
    if (tid < ncols * DW) L.drawn[tid] = 0u;
    wave_sync<NW>();
    PM_PROF_MARK(kProfP1w)
    PM_MARK("P2");

    // ---- P2: per-view selection priors (:1070-1104), lane per (column, view) --
    patch_weight_sums(p, L, ncols, tid, nt);
    for (int item = tid; item < ncols * S; item += nt) {
      const int c = item_div(item, inv_S);
      const int s = item - c * S;
      const int col = col0 + c;
      const float* rec = p.rec + (size_t)pix_index(p, row, col) * p.rec_stride;
      const lds_f32* pose = L.poses + s * L.pstride;
      const lds_f32* h = L.hyp + c * 20;
      const lds_f32* cf = L.colf + c * 8;
      const float cost = rec[4 + s];
      const float beta = rec[p.sel_out_off + s];
      const float prev = rec[p.sel_in_off + s];
      L.cost5[c * 5 * S1 + s] = cost;
      L.betav[item] = beta;
      L.prevv[item] = prev;
      const float alpha = hmm_message<true>(p, cost, L.fm[item]);
      const float sp = sel_prob_fn(alpha, beta, prev, p.prev_sel_prob_weight);
      float cos_tri, cos_inc;
      viewing_angles(pose, cf[2], cf[3], cf[4], h[1], h[2], h[3], cos_tri, cos_inc);
      const float tp = tri_prob(p, cos_tri);
      const float ip = inc_prob(p, cos_inc);
      float Hm[9];
      compose_homography(iK, pose, row, col, h[0], h[1], h[2], h[3], Hm);
      const float rp = res_prob(Hm, (float)row, (float)col, p.radius);
      L.q[item] = sp * tp * ip * rp;
    }
    wave_sync<NW>();

    PM_PROF_MARK(kProfP2)
    PM_MARK("P3a");
    // ---- P3a: TransformPDFToCDF (:683-696), sequential sum order, lane per column
    if (col_lane) {
      const int c = tid;
      lds_f32* q = L.q + c * S;
      float prob_sum = 0.0f;
#pragma unroll 4
      for (int i = 0; i < S; ++i) prob_sum += q[i];
      const float inv_prob_sum = 1.0f / prob_sum;
      float cum = 0.0f;
#pragma unroll 4
      for (int i = 0; i < S; ++i) {
        cum += q[i] * inv_prob_sum;
        q[i] = cum;
      }
    }
    wave_sync<NW>();
    PM_PROF_MARK(kProfP3a)
    PM_MARK("P3b");
    // ---- P3b: Monte-Carlo view draws (:1128-1138), lane per (column, draw): the first view whose CDF value exceeds
    // the uniform. The CDF is non-decreasing unless it holds a NaN (then its last value is one: the running sum never
    // recovers), so the first such view = the number of leading views that do NOT exceed it, found by bisection; a
    // column whose CDF ends in a NaN takes the reference's linear scan. sv = S: no view (the zero slot of P5a).
    for (int item = tid; item < ncols * M; item += nt) {
      const int c = item_div(item, inv_M);
      const int m = item - c * M;
      const float u = draws[((size_t)row * RW + col0 + c) * dstride + 4 + m];
      const lds_f32* q = L.q + c * S;
      const float last = q[S - 1];
      int lo = 0;
      if (last != last) {
        lo = S;
        for (int s = 0; s < S; ++s) {
          if (q[s] > u) { lo = s; break; }
        }
      } else {
        for (int step = step0; step > 0; step >>= 1) {
          const int idx = lo + step;
          if (idx <= S && !(q[idx - 1] > u)) lo = idx;
        }
      }
      L.sv[item] = lo;
      if (lo < S)
        __hip_atomic_fetch_or(L.drawn + c * DW + (lo >> 5), 1u << (lo & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wave_sync<NW>();
    PM_PROF_MARK(kProfP3b)
    PM_MARK("P3c");
    // ---- P3c: one task set per distinct drawn view, lane per (column, view); list positions from a ballot ----
    int n4 = 0, ng = 0;
    {
      LDS_AS uint16_t* tasks = (LDS_AS uint16_t*)L.tasks;
      for (int item0 = 0; item0 < ncols * S; item0 += nt) {
        const int item = item0 + tid;
        bool drawn = false;
        int c = 0, s = 0;
        if (item < ncols * S) {
          c = item_div(item, inv_S);
          s = item - c * S;
          drawn = ((L.drawn[c * DW + (s >> 5)] >> (s & 31)) & 1u) != 0u;
        }
        const unsigned long long bal = __ballot(drawn ? 1 : 0);
        if (drawn) {
          const int below = lanes_below(bal);
          const uint32_t t1 = task16_pack(c, 1, s, 0);
          const uint32_t w01 = t1 | ((t1 + 512u) << 16);   // hypotheses 1, 2
          const uint32_t w23 = w01 + (1024u | (1024u << 16));  // hypotheses 3, 4
          LDS_AS uint32_t* tw = (LDS_AS uint32_t*)(tasks + n4 + 4 * below);
          tw[0] = w01;
          tw[1] = w23;
          if (GEOM) tasks[wave_max_tasks(C, S, M) + ng + below] = (uint16_t)task16_pack(c, 0, s, 1);
        }
        const int cnt = __popcll(bal);
        n4 += 4 * cnt;
        ng += cnt;
      }
    }
    wave_sync<NW>();

    PM_PROF_MARK(kProfP3c)
    PM_MARK("P4");
    // ---- P4: NCC of hypotheses 1..4 against the drawn views (:1157-1172) -----
    if (HELP > 1 || !(PM_ABLATE(p) & 1))
      run_tasks_wave<GEOM, NW, CAP, MUBUF, PROF, HELP>(p, L, srd, row, col0, tid, n4, ng, evals, prof_acc, prof_t, kProfP4B);

    PM_PROF_MARK(kProfP4F)
    PM_MARK("P5a");
    // ---- P5a: accumulate in draw order (:1144-1172), lane per (column, hypothesis); a draw without a view adds the
    // row's zero slot
    for (int item = tid; item < ncols * 5; item += nt) {
      const int c = item / 5;
      const lds_f32* costs = L.cost5 + item * S1;
      const lds_f32* geos = L.geo + item * S1;
      const lds_i32* sv = L.sv + c * M;
      float acc = 0.0f;
      for (int m = 0; m < M; ++m) {
        const int src = sv[m];
        acc += costs[src];
        if (GEOM) acc += p.geom_reg * geos[src];
      }
      L.csum[item] = acc;
    }
    wave_sync<NW>();
    PM_PROF_MARK(kProfP5a)
    PM_MARK("P5b");
    // ---- P5b: argmin, store, next row's previous state (:1176-1182,1279-1282) --
    if (col_lane) {
      const int c = tid;
      int min_idx = 0;
      float min_cost = L.csum[c * 5];
#pragma unroll
      for (int i = 1; i < 5; ++i) {
        const float ci = L.csum[c * 5 + i];
        if (ci <= min_cost) { min_cost = ci; min_idx = i; }
      }
      L.best[c] = min_idx;
      const lds_f32* hb = L.hyp + (c * 5 + min_idx) * 4;
      const float bd = hb[0], b0 = hb[1], b1 = hb[2], b2 = hb[3];
      float* rec = p.rec + (size_t)pix_index(p, row, col0 + c) * p.rec_stride;
      float nx, ny;
      normal_from_sweep(p.rot, b0, b1, nx, ny);
      rec[0] = bd; rec[1] = nx; rec[2] = ny; rec[3] = b2;
      lds_f32* h1 = L.hyp + (c * 5 + 1) * 4;
      h1[0] = bd; h1[1] = b0; h1[2] = b1; h1[3] = b2;
    }
    wave_sync<NW>();
    PM_PROF_MARK(kProfP5b)
    PM_MARK("P5c");
    // ---- P5c: winner vs. the views not evaluated yet (= not drawn), lane per (column, view) --
    int n1 = 0;
    {
      LDS_AS uint16_t* tasks = (LDS_AS uint16_t*)L.tasks;
      for (int item0 = 0; item0 < ncols * S; item0 += nt) {
        const int item = item0 + tid;
        bool take = false;
        uint32_t task = 0;
        if (item < ncols * S) {
          const int c = item_div(item, inv_S);
          const int s = item - c * S;
          const int k = L.best[c];
          const bool drawn = ((L.drawn[c * DW + (s >> 5)] >> (s & 31)) & 1u) != 0u;
          take = k != 0 && !drawn;
          task = task16_pack(c, k, s, 0);
        }
        const unsigned long long bal = __ballot(take ? 1 : 0);
        if (take) tasks[n1 + lanes_below(bal)] = (uint16_t)task;
        n1 += __popcll(bal);
      }
    }
    wave_sync<NW>();

    PM_PROF_MARK(kProfP5c)
    PM_MARK("P6");
    // ---- P6: NCC of the winner against the remaining views (:1188-1197) ------
    if (HELP > 1 || !(PM_ABLATE(p) & 1))
      run_tasks_wave<false, NW, CAP, MUBUF, PROF, HELP>(p, L, srd, row, col0, tid, n1, 0, evals, prof_acc, prof_t, kProfP6B);
    PM_PROF_MARK(kProfP6F)
    PM_MARK("P7");

    // ---- P7: cost map, forward message, selection probability (:1186-1207) ---
    for (int item = tid; item < ncols * S; item += nt) {
      const int c = item_div(item, inv_S);
      const int s = item - c * S;
      const int col = col0 + c;
      const int k = L.best[c];
      float* rec = p.rec + (size_t)pix_index(p, row, col) * p.rec_stride;
      const float cost = L.cost5[(c * 5 + k) * S1 + s];
      if (k != 0) rec[4 + s] = cost;
      const float alpha = hmm_message<true>(p, cost, L.fm[item]);
      const float prob = sel_prob_fn(alpha, L.betav[item], L.prevv[item], p.prev_sel_prob_weight);
      L.fm[item] = alpha;
      rec[p.sel_out_off + s] = prob;
      if (FILTER_PHOTO || FILTER_GEOM) {
        const lds_f32* hb = L.hyp + (c * 5 + 1) * 4;  // == best (stored in P5)
        const lds_f32* pose = L.poses + s * L.pstride;
        const float bp0 = hb[0] * (iK[0] * col + iK[1]);
        const float bp1 = hb[0] * (iK[2] * row + iK[3]);
        const float bp2 = hb[0];
        float cos_tri, cos_inc;
        viewing_angles(pose, bp0, bp1, bp2, hb[1], hb[2], hb[3], cos_tri, cos_inc);
        int ok = 0;
        if (!(cos_tri > p.filter_cos_min_tri || cos_inc <= 0.0f)) {
          const float min_ncc_prob = ncc_prob(p, 1.0f - p.filter_min_ncc);
          bool photo_ok = true, geom_ok = true;
          if (FILTER_PHOTO) photo_ok = prob >= min_ncc_prob;
          if (FILTER_GEOM)
            geom_ok = geom_cost(p, pose, s, (float)row, (float)col, hb[0]) <= p.filter_geom_max_cost;
          ok = (photo_ok && geom_ok) ? 1 : 0;
        }
        L.flags[item] = ok;
      }
    }
    PM_PROF_MARK(kProfP7)
    if (FILTER_PHOTO || FILTER_GEOM) {
      wave_sync<NW>();
      PM_MARK("P8");
      if (col_lane) {
        const int c = tid;
        int num = 0;
        for (int s = 0; s < S; ++s) num += L.flags[c * S + s];
        const int pix = pix_index(p, row, col0 + c);
        if (num < p.filter_min_num_consistent) {
          float* rec = p.rec + (size_t)pix * p.rec_stride;
          rec[0] = 0.0f; rec[1] = 0.0f; rec[2] = 0.0f; rec[3] = 0.0f;
        } else {
          for (int s = 0; s < S; ++s)
            if (L.flags[c * S + s]) p.mask[(size_t)s * p.W * p.H + pix] = 1;
        }
      }
    }
    wave_sync<NW>();
    PM_MARK("ROWEND");
    slot_top = slot_top + 1 == win ? 0 : slot_top + 1;
    slot_c = slot_c + 1 == win ? 0 : slot_c + 1;
    slot_new = slot_new + 1 == win ? 0 : slot_new + 1;
  }

  if (tid == 0 && p.evals) atomicAdd(p.evals, (unsigned long long)evals);
  if (PROF) {
    PM_PROF_MARK(kProfP8)
    if (tid == 0 && p.prof) {
      for (int i = 0; i < kProfSlots - 1; ++i) atomicAdd(p.prof + i, prof_acc[i]);
      atomicAdd(p.prof + kProfSlots - 1, 1ull);  // waves that reported
    }
  }
}

// ---------------------------------------------------------------------------
// ComputeInitialCost (patch_match_cuda.cu:863-912) for the 11 x 11 window in the sweep kernel's shape (round 6): a wave
// takes kInitRows consecutive rows of its column group, scrolls the reference tile by one row per step, and evaluates
// the ncols x S costs of a row as one or more batches of the sweep's own pass B (ncc_rounds_wave: inside-first order,
// unclamped addressing, gathers through the buffer resource). pm_initial_cost_kernel above -- one wave per (row, column
// group): pose records, eleven tile rows and the weights of 2 x 121 taps fetched per 40 evaluations, the evaluations
// through ncc_group's explicit addressing -- spent 41 ms per 2560 x 1920 image, as much as a sweep for half its
// evaluations. Same arithmetic, same bits (every pixel is independent; tests: initial state and cost).
// ---------------------------------------------------------------------------
constexpr int kInitRows = 32;
template <bool MUBUF>
__global__ void __launch_bounds__(64 * kQuadWaves, kQuadOcc) pm_initial_cost_wave_kernel(const PmParams* __restrict__ pp) {
  constexpr int NW = kQuadWaves, CAP = kQuadThCap;
  const PmParams& p = pp[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lds L;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  {
    const LdsOffsets o = lds_offsets_wave(p.C, p.S, p.radius, p.ntaps, p.num_samples, false, CAP, NW);
    lds_bind(L, (lds_char*)smem + wave * o.priv_stride, o);
    L.poses = (lds_f32*)((lds_char*)smem + o.poses);
    L.fpo = (lds_u32*)((lds_char*)smem + o.fpb);
    L.fpb = (lds_u64*)((lds_char*)smem + o.fpb);
    L.tapg = (lds_f32*)((lds_char*)smem + o.tapg);
  }
  const int tid0 = threadIdx.x & 63;
  const int S = p.S, C = p.C;
  const int col0 = (blockIdx.x * NW + wave) * C;
  const int ncols = min(C, p.W - col0);
  const int win = 2 * p.radius + 1;
  const int row0 = blockIdx.y * kInitRows, row1 = min(row0 + kInitRows, p.H);
  const v4i srd = MUBUF ? fp_resource(p) : (v4i)(0);
  if (wave == 0) tap_tables_init(L.tapg, tid0, 64, p.step, p.radius, p.spatial_norm, false);
  L.pstride = lds_pose_stride(false);
  for (int i = threadIdx.x; i < S * L.pstride; i += 64 * NW) {
    const int s = i / L.pstride;
    L.poses[i] = p.poses[s * kPoseStride + (i - s * L.pstride)];
  }
  for (int i = threadIdx.x; i < S; i += 64 * NW) {
    if (MUBUF) L.fpo[i] = p.src_fp_off[i];
    else L.fpb[i] = (uint64_t)p.src_fp_tab[i];
  }
  __syncthreads();
  if (col0 >= p.W) return;
  for (int i = tid0; i < C * 8; i += 64) {  // the seven padding taps of every column: written once
    const int at = (i >> 3) * 128 + 120 + (i & 7);
    if ((i & 7) != 0) { L.wgt[at] = 0.0f; L.refc[at] = 0.0f; }
  }
  for (int r = row0 - p.radius; r < row0 + p.radius; ++r) tile_load_row(p, L, col0, r, tid0, 64);
  wave_sync<NW>();
  int slot_c = row0 % win, slot_top = (row0 - p.radius) % win, slot_new = (row0 + p.radius) % win;
  if (slot_top < 0) slot_top += win;
  const int n = ncols * S;
  for (int row = row0; row < row1; ++row) {
    int tid = tid0;
    launder_vgpr(tid);  // (as in the sweep: nothing derived from the lane id lives across the NCC rounds)
    tile_load_row_slot(p, L, col0, row + p.radius, slot_new, tid);
    wave_sync<NW>();
    patch_weights_wave(p, L, slot_c, slot_top, tid);
    wave_sync<NW>();
    patch_weight_sums(p, L, ncols, tid, 64);
    for (int base = 0; base < n; base += CAP) {
      const int nb = min(CAP, n - base);
      // pass A, lane per (column, view): the homography of the pixel's initial plane
      bool inside = false;
      uint32_t desc = 0;
      if (tid < nb) {
        const int t = base + tid;
        const int c = t / S, sv = t - c * S;
        const int col = col0 + c;
        const float* rec = p.rec + (size_t)(row * p.W + col) * p.rec_stride;
        float Hm[9];
        compose_homography(p.refInvK, L.poses + sv * L.pstride, row, col, rec[0], rec[1], rec[2], rec[3], Hm);
        centre_homography(Hm, row, col, p.radius);
        for (int k = 0; k < 9; ++k) L.th[tid * 9 + k] = Hm[k];
        inside = patch_inside(p, Hm);
        desc = (uint32_t)tid | (inside ? 0x80u : 0u) | ((uint32_t)c << 8) | ((uint32_t)sv << 16);
      }
      {
        const unsigned long long m1 = __ballot(inside ? 1 : 0);
        const unsigned long long valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
        const unsigned long long m0 = valid & ~m1;
        if (tid < nb) L.desc[inside ? lanes_below(m1) : __popcll(m1) + lanes_below(m0)] = desc;
      }
      wave_sync<NW>();
      ncc_rounds_wave<MUBUF>(p, L, srd, L.tapg, tid, nb, 0, 1);
      wave_sync<NW>();
      if (tid < nb) {
        const int t = base + tid;
        const int c = t / S, sv = t - c * S;
        const int pix = row * p.W + col0 + c;
        p.rec[(size_t)pix * p.rec_stride + 4 + sv] =
            ncc_finish(L.th[tid * 9 + 0], L.th[tid * 9 + 1], L.th[tid * 9 + 2], p.ref_sum[pix], p.ref_sqsum[pix],
                       L.colf[c * 8 + 5]);
      }
      wave_sync<NW>();
    }
    slot_top = slot_top + 1 == win ? 0 : slot_top + 1;
    slot_c = slot_c + 1 == win ? 0 : slot_c + 1;
    slot_new = slot_new + 1 == win ? 0 : slot_new + 1;
  }
}

// Four waves per workgroup, each with its own column group, sharing one LDS copy of the read-only per-problem
// tables: four workgroups = 16 waves per CU with 64 task slots per batch at S = 20 (geometric pass included).
// MUBUF: packed images addressed through the problem's buffer resource (the normal case), or by explicit indices
template <bool GEOM, bool FILTER_PHOTO, bool FILTER_GEOM, bool MUBUF>
__global__ void __launch_bounds__(64 * kQuadWaves, GEOM ? 4 : kQuadOcc) pm_sweep_quad_kernel(const PmParams* __restrict__ pp) {
  sweep_wave_body<GEOM, FILTER_PHOTO, FILTER_GEOM, kQuadWaves, kQuadThCap, MUBUF>(pp);
}
// Two waves per column group (sweep_wave_body, HELP = 2): launches that one wave per column cannot fill the GPU with.
template <bool GEOM, bool FILTER_PHOTO, bool FILTER_GEOM>
__global__ void __launch_bounds__(128, GEOM ? 4 : kQuadOcc) pm_sweep_pair_kernel(const PmParams* __restrict__ pp) {
  sweep_wave_body<GEOM, FILTER_PHOTO, FILTER_GEOM, 2, kQuadThCap, true, false, 2>(pp);
}
// The same kernel with the phase clocks compiled in (pm_enable_phase_profile; photometric sweeps).
template <bool FILTER_PHOTO>
__global__ void __launch_bounds__(64 * kQuadWaves, 4) pm_sweep_quad_prof_kernel(const PmParams* __restrict__ pp) {
  sweep_wave_body<false, FILTER_PHOTO, false, kQuadWaves, kQuadThCap, true, true>(pp);
}

// ---------------------------------------------------------------------------
// The random numbers of one sweep, for every pixel, before the sweep (the 11 x 11 kernel above reads them from
// PmParams::draws). A column of the sweep frame owns one XORWOW stream (reference: one curand state per thread =
// column, :1022-1028, stored back at :1285-1287) and consumes it row by row: the perturbed depth (:1055-1058), the
// 3 .. 12 draws of PerturbNormal (:133-196, its retries depend on the pixel's current normal -- known before the sweep,
// the sweep writes a pixel's plane only when it reaches its row) and the M uniforms of the view draws (:1129). None of
// it depends on what the sweep decides in the rows above, so it is taken out of the sequential row step, where it ran
// on C of 64 lanes: here a lane is a column and a wave covers 64 of them. Per pixel: {rd, rn0, rn1, rn2, u[M]}.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) pm_draw_kernel(const PmParams* __restrict__ pp) {
  // A few hundred waves, each a long serial chain, usually beside the other sub-batch's sweep launch (20 waves per CU
  // that are never short of work): without priority a draw wave gets one issue slot in twenty and the sweep behind it
  // waits 45-100 ms instead of 7.
  __builtin_amdgcn_s_setprio(3);
  const PmParams& p = pp[blockIdx.y];
  const int RW = rot_width(p), RH = rot_height(p);
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= RW) return;
  const int M = p.num_samples;
  const int dstride = pm_draw_stride(M);
  const float* iK = p.refInvK;
  uint32_t* rng_at = p.rng + (size_t)pix_index(p, 0, col) * kRngWords;
  Rng rng = rng_load(rng_at);
  float* out = p.draws + (size_t)col * dstride;
  const float* rec = p.rec + (size_t)pix_index(p, 0, col) * p.rec_stride;
  float cd = rec[0], r1 = rec[1], r2 = rec[2], cn2 = rec[3];
  for (int row = 0; row < RH; ++row) {
    // next row's plane in flight while this row's numbers are drawn
    float nd = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
    if (row + 1 < RH) {
      const float* nrec = p.rec + (size_t)pix_index(p, row + 1, col) * p.rec_stride;
      nd = nrec[0]; n1 = nrec[1]; n2 = nrec[2]; n3 = nrec[3];
    }
    float cn0, cn1;
    normal_to_sweep(p.rot, r1, r2, cn0, cn1);
    const float dmin = (1.0f - p.perturbation) * cd;
    const float dmax = (1.0f + p.perturbation) * cd;
    const float rd = rng_uniform(rng) * (dmax - dmin) + dmin;
    float rn0, rn1, rn2;
    perturb_normal(iK, row, col, p.perturbation_pi, cn0, cn1, cn2, rng, rn0, rn1, rn2);
    float* o = out + (size_t)row * RW * dstride;
    o[0] = rd; o[1] = rn0; o[2] = rn1; o[3] = rn2;
    for (int m = 0; m < M; ++m) o[4 + m] = rng_uniform(rng) - FLT_EPSILON;  // :1129
    cd = nd; r1 = n1; r2 = n2; cn2 = n3;
  }
  rng_store(rng_at, rng);  // :1285-1287
}

// Debug: raw XORWOW streams of the generator above (seed = sequence id, as InitRandomStateKernel
// seeds it), for the bit comparison with rocRAND's rocrand_init / rocrand_uniform in the tests.
__global__ void pm_rng_streams_kernel(const unsigned long long* __restrict__ seeds, int nseeds, int ndraws,
                                      float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseeds) return;
  Rng rng;
  rng_init(rng, seeds[i]);
  for (int k = 0; k < ndraws; ++k) out[(size_t)i * ndraws + k] = rng_uniform(rng);
}

// pixel records -> API layout (Mat<float> slice-major, mat.h:107-109)
// Thread = (pixel, field of its record): four neighbouring lanes read four consecutive floats of one record (16
// contiguous bytes per quad: what the address unit coalesces), every thread issues ONE load, and a field's plane is
// written by 16 lanes per wave as 64 contiguous bytes. A lane per pixel walking its 176-byte record -- every load of
// the wave 64 lines apart, the loop over the sources rolled, one wait per trip -- took 3.9 ms per 2560 x 1920 image
// (220 GB/s) for a copy of 0.87 GB.
__global__ void __launch_bounds__(256) pm_extract_kernel(const PmParams p, int sel_off, float* __restrict__ depth,
                                                         float* __restrict__ normal, float* __restrict__ sel,
                                                         float* __restrict__ cost) {
  const int pix = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int f = 4 * blockIdx.y + (threadIdx.x & 3);
  const int n = p.W * p.H;
  if (pix >= n || f >= p.rec_stride) return;
  float* out = nullptr;
  if (f == 0) out = depth;
  else if (f < 4) out = normal ? normal + (size_t)(f - 1) * n : nullptr;
  else if (f < 4 + p.S) out = cost ? cost + (size_t)(f - 4) * n : nullptr;
  else if (f >= sel_off && f < sel_off + p.S) out = sel ? sel + (size_t)(f - sel_off) * n : nullptr;
  if (out) out[pix] = p.rec[(size_t)pix * p.rec_stride + f];
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------

size_t pm_sweep_lds_bytes(const PmParams& p, bool geom) {
  return lds_offsets(p.C, p.S, p.radius, p.ntaps, p.num_samples, geom).total;
}

// LDS budget of one four-wave workgroup when four of them share a CU: 160 KB / 4 in 1280-byte granules.
constexpr size_t kQuadLdsBudget = 40960;

// Is this shape served by the 11 x 11 four-wave kernel (else: the generic kernel)?
static bool pm_wave_shape(int ntap1d, int step, int S, int C) { return ntap1d == 11 && step >= 1 && S <= 512 && C <= 8; }

bool pm_sweep_uses_draws(const PmParams& p, bool geom) {
  return pm_wave_shape(p.ntap1d, p.step, p.S, p.C) &&
         lds_offsets_wave(p.C, p.S, p.radius, p.ntaps, p.num_samples, geom, kQuadThCap, kQuadWaves).total <= kQuadLdsBudget;
}

int pm_pick_columns(int S, int ntaps, int num_samples, bool geom, int radius, int requested) {
  const size_t budget = 60 * 1024;
  // default: 2 columns per wave of the 11 x 11 kernel (16 waves resident per CU, the lane-per-(column, view)
  // phases are one pass; measured 604 / 643 / 718 ms per 16-image launch for C = 2 / 3 / 4), 4 for the generic kernel
  const int cols_env = dev_switch_int("COLMAP_AMD_PM_COLS", 0);  // experiments / tests
  if (requested <= 0 && cols_env > 0) requested = cols_env;
  int c = requested > 0 ? requested : (ntaps == 121 ? 2 : 4);
  if (c > 64) c = 64;
  if (ntaps == 121 && requested <= 0) {
    // many source images: one column per wave while that keeps the four-wave workgroup within its LDS budget
    while (c > 1 && lds_offsets_wave(c, S, radius, ntaps, num_samples, geom, kQuadThCap, kQuadWaves).total > kQuadLdsBudget) --c;
  }
  while (c > 1 && lds_offsets(c, S, radius, ntaps, num_samples, geom).total > budget) --c;
  return c;
}

// Can the 11 x 11 sweep kernels address this problem's packed source images through one buffer resource
// (fp_resource)? The host has found all images inside a 4 GB window (pm_api.cpp) and the stride field of the
// resource holds 4 * rows < 2^14.
static bool pm_fp_resource_ok(const PmParams& p) {
  return p.fp_base != nullptr && p.src_fp_off != nullptr && 4 * (p.fp_rows1 + 1) < (1 << 14);
}

void pm_launch_build_footprint(const uint8_t* src, uint32_t* fp, int S, int w, int h, hipStream_t st) {
  dim3 block(256, 1, 1);
  const int pw = pm_fp_width(w), ph = pm_fp_height(h);
  dim3 grid((pw + 255) / 256, ph, S);
  hipLaunchKernelGGL(pm_build_footprint_kernel, grid, block, 0, st, src, fp, w, h, pw, ph);
}

void pm_launch_filter_ref(const uint8_t* gray, int W, int H, int radius, int step, float sigma_spatial,
                          float sigma_color, uint8_t* out_img, float* out_sum, float* out_sqsum,
                          hipStream_t st) {
  const float sn = 1.0f / (2.0f * sigma_spatial * sigma_spatial);
  const float cn = 1.0f / (2.0f * sigma_color * sigma_color);
  dim3 block(64, 4, 1);
  dim3 grid((W + 63) / 64, (H + 3) / 4, 1);
  hipLaunchKernelGGL(pm_filter_ref_kernel, grid, block, 0, st, gray, W, H, radius, step, sn, cn,
                     out_img, out_sum, out_sqsum);
}

void pm_launch_init_state(const PmParams& p, bool random_init, float depth_min, float depth_max,
                          const float* init_depth, const float* init_normal, hipStream_t st) {
  dim3 block(64, 4, 1);
  dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1);
  hipLaunchKernelGGL(pm_init_state_kernel, grid, block, 0, st, p, random_init ? 1 : 0, depth_min,
                     depth_max, init_depth, init_normal);
}

void pm_launch_initial_cost(const PmParams& p, const PmParams* dev_params, int batch, hipStream_t st) {
  // 11 x 11 window and a shape the four-wave LDS block holds: the sweep-shaped kernel (COLMAP_AMD_PM_WAVE=0: tests)
  if (dev_switch_int("COLMAP_AMD_PM_WAVE", 1) != 0 && pm_sweep_uses_draws(p, false)) {
    const bool mubuf = pm_fp_resource_ok(p) && dev_switch_int("COLMAP_AMD_PM_FP_GLOBAL", 0) == 0;
    const size_t qlds = lds_offsets_wave(p.C, p.S, p.radius, p.ntaps, p.num_samples, false, kQuadThCap, kQuadWaves).total;
    const unsigned groups = (unsigned)((p.W + p.C - 1) / p.C);
    const dim3 grid((groups + kQuadWaves - 1) / kQuadWaves, (unsigned)((p.H + kInitRows - 1) / kInitRows), batch);
    if (mubuf) hipLaunchKernelGGL(pm_initial_cost_wave_kernel<true>, grid, dim3(64 * kQuadWaves), qlds, st, dev_params);
    else hipLaunchKernelGGL(pm_initial_cost_wave_kernel<false>, grid, dim3(64 * kQuadWaves), qlds, st, dev_params);
    return;
  }
  const size_t lds = lds_offsets(p.C, p.S, p.radius, p.ntaps, p.num_samples, false).total;
  dim3 block(64, 1, 1);
  dim3 grid((p.W + p.C - 1) / p.C, p.H, batch);
  if (p.ntap1d == 11) hipLaunchKernelGGL(pm_initial_cost_kernel<11>, grid, block, lds, st, dev_params);
  else hipLaunchKernelGGL(pm_initial_cost_kernel<0>, grid, block, lds, st, dev_params);
}

// Which sweep kernel runs (two families):
//  * pm_sweep_quad_kernel -- 11 x 11 window, four-wave workgroups whose waves each sweep their own column group, fed
//    by pm_draw_kernel (the sweep's random numbers): whenever four workgroups fit a CU;
//  * pm_sweep_kernel      -- any window, 256-thread workgroups with barriers (round 1's design): other window
//    sizes, more source images than the four-wave LDS block holds, COLMAP_AMD_PM_WAVE=0.
// The 11 x 11 kernel exists with and without the buffer-resource addressing of the packed images (fp_resource).
// Both families produce the same bits.
// The sweep's random numbers (pm_draw_kernel), launched in front of a sweep that reads them; a launch of its own so
// that the caller's events bracket the sweep kernel alone.
static bool pm_sweep_takes_wave_kernel(const PmParams& p, bool geom) {
  return dev_switch_int("COLMAP_AMD_PM_WAVE", 1) != 0 && p.draws != nullptr && pm_sweep_uses_draws(p, geom);
}
void pm_launch_draws(const PmParams& p, const PmParams* dev_params, int batch, bool geom, hipStream_t st) {
  if (!pm_sweep_takes_wave_kernel(p, geom)) return;
  const int rw = (p.rot & 1) ? p.H : p.W;
  hipLaunchKernelGGL(pm_draw_kernel, dim3((rw + 63) / 64, batch, 1), dim3(64, 1, 1), 0, st, dev_params);
}

const char* pm_launch_sweep(const PmParams& p, const PmParams* dev_params, int batch, int threads, bool geom,
                            bool filter_photo, bool filter_geom, hipStream_t st) {
  const int rw = (p.rot & 1) ? p.H : p.W;
  const unsigned groups = (unsigned)((rw + p.C - 1) / p.C);
#define PM_LAUNCH_V4(KERNEL, MB, GRID, BLOCK, LDS)                                                   \
  do {                                                                                              \
    if (geom) {                                                                                     \
      if (filter_photo && filter_geom) hipLaunchKernelGGL((KERNEL<true, true, true, MB>), GRID, BLOCK, LDS, st, dev_params);  \
      else hipLaunchKernelGGL((KERNEL<true, false, false, MB>), GRID, BLOCK, LDS, st, dev_params);  \
    } else {                                                                                        \
      if (filter_photo) hipLaunchKernelGGL((KERNEL<false, true, false, MB>), GRID, BLOCK, LDS, st, dev_params);  \
      else hipLaunchKernelGGL((KERNEL<false, false, false, MB>), GRID, BLOCK, LDS, st, dev_params); \
    }                                                                                               \
  } while (0)
  if (pm_sweep_takes_wave_kernel(p, geom)) {   // (pm_launch_draws has run in front of this launch)
    // COLMAP_AMD_PM_FP_GLOBAL=1 (tests): explicit indices although the buffer resource would do
    const bool mubuf = pm_fp_resource_ok(p) && dev_switch_int("COLMAP_AMD_PM_FP_GLOBAL", 0) == 0;
    const size_t qlds = lds_offsets_wave(p.C, p.S, p.radius, p.ntaps, p.num_samples, geom, kQuadThCap, kQuadWaves).total;
    const dim3 qgrid((groups + kQuadWaves - 1) / kQuadWaves, batch, 1), qblock(64 * kQuadWaves, 1, 1);
    if (p.prof && mubuf && !geom) {
      // phase profile (pm_enable_phase_profile): the shipped kernel with its phase clocks compiled in
      if (filter_photo) hipLaunchKernelGGL(pm_sweep_quad_prof_kernel<true>, qgrid, qblock, qlds, st, dev_params);
      else hipLaunchKernelGGL(pm_sweep_quad_prof_kernel<false>, qgrid, qblock, qlds, st, dev_params);
      return "pm_sweep_quad_prof_kernel";
    }
    if (p.help > 1 && mubuf && p.C == 1) {
      const size_t plds = lds_offsets_wave(p.C, p.S, p.radius, p.ntaps, p.num_samples, geom, kQuadThCap, 1).total;
      const dim3 pgrid(groups, batch, 1), pblock(128, 1, 1);
      if (geom) {
        if (filter_photo && filter_geom) hipLaunchKernelGGL((pm_sweep_pair_kernel<true, true, true>), pgrid, pblock, plds, st, dev_params);
        else hipLaunchKernelGGL((pm_sweep_pair_kernel<true, false, false>), pgrid, pblock, plds, st, dev_params);
      } else {
        if (filter_photo) hipLaunchKernelGGL((pm_sweep_pair_kernel<false, true, false>), pgrid, pblock, plds, st, dev_params);
        else hipLaunchKernelGGL((pm_sweep_pair_kernel<false, false, false>), pgrid, pblock, plds, st, dev_params);
      }
      return "pm_sweep_pair_kernel";
    }
    if (mubuf) PM_LAUNCH_V4(pm_sweep_quad_kernel, true, qgrid, qblock, qlds);
    else PM_LAUNCH_V4(pm_sweep_quad_kernel, false, qgrid, qblock, qlds);
    return mubuf ? "pm_sweep_quad_kernel" : "pm_sweep_quad_kernel (explicit indices)";
  }
#undef PM_LAUNCH_V4
  const size_t lds = pm_sweep_lds_bytes(p, geom);
  dim3 block(threads, 1, 1);
  dim3 grid(groups, batch, 1);
#define PM_LAUNCH_N(N, G, FP, FG, PR) \
  hipLaunchKernelGGL((pm_sweep_kernel<N, G, FP, FG, PR>), grid, block, lds, st, dev_params)
  if (geom) {
    if (filter_photo && filter_geom) PM_LAUNCH_N(0, true, true, true, false);
    else PM_LAUNCH_N(0, true, false, false, false);
  } else {
    if (filter_photo) PM_LAUNCH_N(0, false, true, false, false);
    else PM_LAUNCH_N(0, false, false, false, false);
  }
#undef PM_LAUNCH_N
  return "pm_sweep_kernel";
}
void pm_launch_rng_streams(const unsigned long long* seeds, int nseeds, int ndraws, float* out,
                           hipStream_t st) {
  hipLaunchKernelGGL(pm_rng_streams_kernel, dim3((nseeds + 63) / 64), dim3(64), 0, st, seeds, nseeds,
                     ndraws, out);
}

void pm_launch_extract(const PmParams& p, int sel_off, float* depth, float* normal, float* sel,
                       float* cost, hipStream_t st) {
  const int n = p.W * p.H;
  hipLaunchKernelGGL(pm_extract_kernel, dim3((n + 63) / 64, (p.rec_stride + 3) / 4), dim3(256), 0, st, p, sel_off, depth,
                     normal, sel, cost);
}

}  // namespace colmap_amd
