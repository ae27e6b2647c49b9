// Internal interface between the C-ABI host code (pm_api.cpp) and the gfx950
// kernels (pm_kernels.hip). Not installed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "switches.h"

// PmParams::ablate (skip phases of the sweep to time the rest: results are garbage) is only ever non-zero in a
// profiling build (-DCOLMAP_AMD_DIAG_BUILD: pm_api.cpp sets it from a development switch); the kernels keep their three
// scalar tests on it so that the shipped instruction stream is the profiled one.
#define PM_ABLATE(p) ((p).ablate)

namespace colmap_amd {

constexpr int kPoseStride = 43;  // K4 R9 T3 C3 P12 invP12 (reference patch_match_cuda.cu:1762)
constexpr int kRngWords = 6;     // XORWOW: x[5] + d
constexpr int kPmProfSlots = 24; // phase-profile counters per handle (pm_kernels.hip: kProf*; the generic kernel uses 10)

// Packed source images ("footprints": one dword per texel position = its 2 x 2 bilinear neighbourhood) are
// stored as vertical strips of kFpStrip = 16 entries: inside a strip the rows follow each other, 64 bytes each, so
// a 128-byte cache line is a 16 x 2 block of entries and the entry index is
//   (ex / 16) * 16 * rows + 16 * ey + (ex % 16).
// Why strips: (a) the 11 x 11 sweep kernels leave that index to the address unit (swizzled buffer resource,
// pm_kernels.hip: fp_resource), which is what makes 2-D blocking free; (b) the texture-address unit serves a quad
// of lanes in one cycle only when its four addresses lie within 16 bytes (scripts/ubench/gather_rates.hip,
// profiles/r04_ubench_gather_rates.log) -- the taps of a quad are neighbours along x, so wide strip rows keep
// the quads of a warped 11 x 11 window fast (16 x 2: 17-19 cycles per gather instruction in the microbenchmark,
// 8 x 4: 24, lane-per-line: 64) while two rows per line still halve the lines a window touches against a
// row-major image. Entry (ex, ey) holds texel position (ex - kFpRingX, ey - kFpRingY); positions -2 and w (h) are
// the all-zero border ring a clamped tap reads; the ring offsets are one strip / whole lines so that texel (0, 0)
// starts a cache line.
#ifndef PM_FP_STRIP
#define PM_FP_STRIP 16  // entries per strip row: 16 (x 2 rows per 128-byte cache line); measured 8 (x 4): +4.4 %, 32 (x 1): +1.2 % launch time
#endif
constexpr int kFpStrip = PM_FP_STRIP;
constexpr int kFpRingX = kFpStrip, kFpRingY = 4;
inline int pm_fp_width(int w) { return (w + kFpRingX + 1 + kFpStrip - 1) & ~(kFpStrip - 1); }    // entries per row (whole strips)
inline int pm_fp_height(int h) { return (h + kFpRingY + 1 + 3) & ~3; }   // rows (multiple of 4)
inline size_t pm_fp_entries(int w, int h) { return (size_t)pm_fp_width(w) * pm_fp_height(h); }

// Per-sweep kernel parameters (reference SweepOptions, patch_match_cuda.cu:914-931,
// plus the geometry of the virtual rotation).
struct PmParams {
  // geometry
  int W, H;         // un-rotated reference image size
  int rot;          // number of 90-degree CCW rotations of the sweep frame (0..3)
  int S;            // number of source images
  int src_w, src_h; // source slot size (max over sources)
  float fp_xmax, fp_ymax;  // src_w + kFpRingX, src_h + kFpRingY: last used column / row of the packed image
  int fp_rows1;            // rows of the packed image (pm_fp_height), minus one (fp_index)
  int radius, step, ntap1d, ntaps;
  int num_samples;
  int rec_stride;   // floats per pixel record: 4 + 3*S
  int sel_in_off;   // record offset of prev_sel_prob (read)
  int sel_out_off;  // record offset of sel_prob (backward msgs, then written)
  int C;            // image columns per workgroup
  int help;         // waves per column group of the 11 x 11 sweep kernel: 2 = a helper wave shares pass B (pm_sweep_pair_kernel, C = 1)
  int ablate;       // profiling only (COLMAP_AMD_PM_ABLATE): bit 0 skip the NCC task passes, bit 1 skip the
                    // hypothesis generation, bit 2 skip the backward-message pre-pass; results are garbage
  int xcd_map;      // batched launch of the generic kernel: 0 problem = id % batch, 1 neighbouring problems per XCD
  float refK[4];    // rotated {fx, cx, fy, cy}
  float refInvK[4]; // rotated {1/fx, -cx/fx, 1/fy, -cy/fy}
  float perturbation;
  float perturbation_pi;
  float prev_sel_prob_weight;
  float spatial_norm, color_norm;
  float cos_min_tri, inv_inc_sigma_sq, inv_ncc_sigma_sq, ncc_norm;
  float geom_reg, geom_max_cost;
  float filter_min_ncc;
  float filter_cos_min_tri;
  float filter_geom_max_cost;
  int filter_min_num_consistent;
  // device pointers
  float* rec;               // [H*W][rec_stride]
  const uint32_t* const* src_fp_tab;  // [S] pointers to packed 2x2 footprints, pm_fp_entries(src_w, src_h) each
                                      // (separate allocations: shareable between problems)
  const uint32_t* fp_base;            // lowest address among them: base of the problem's buffer resource, or null
  const uint32_t* src_fp_off;         // [S] (address - fp_base) / kFpStrip: the images as slots of that resource
  const float* src_depth;   // [S][src_h][src_w] or null
  const uint8_t* ref_img;   // [H][W]
  const float* ref_sum;     // [H][W]
  const float* ref_sqsum;   // [H][W]
  uint32_t* rng;            // [H*W][6]
  float* draws;             // [rot H][rot W][pm_draw_stride]: the sweep's random numbers per pixel of the sweep frame
                            // (pm_draw_kernel -> 11 x 11 sweep kernel), or null: the generic kernel draws in place
  uint8_t* mask;            // [S][H][W] or null
  const float* poses;       // [S][43] for this rotation
  unsigned long long* prof; // optional phase-cycle counters [kPmProfSlots] (debug), else null
  unsigned long long* evals; // NCC evaluations executed by the sweep kernels of this run (one atomic
                             // add per workgroup at its end), always allocated
  unsigned long long* trace; // optional progress trace (debug, pm_enable_progress_trace): [column group][row / 128]
                             // device-wide clock when the group's wave reached that row, last sweep launch; else null
  int trace_stride;          // samples per column group
};

// floats per pixel of PmParams::draws: perturbed depth, perturbed normal, M uniforms (whole float4s)
__host__ __device__ inline int pm_draw_stride(int M) { return 4 + ((M + 3) & ~3); }
// does the sweep of this shape run the 11 x 11 kernel that reads PmParams::draws?
bool pm_sweep_uses_draws(const PmParams& p, bool geom);
size_t pm_sweep_lds_bytes(const PmParams& p, bool geom);
int pm_pick_columns(int S, int ntaps, int num_samples, bool geom, int radius, int requested);

void pm_launch_build_footprint(const uint8_t* src, uint32_t* fp, int S, int w, int h, hipStream_t st);
void pm_launch_filter_ref(const uint8_t* gray, int W, int H, int radius, int step, float sigma_spatial,
                          float sigma_color, uint8_t* out_img, float* out_sum, float* out_sqsum,
                          hipStream_t st);
void pm_launch_init_state(const PmParams& p, bool random_init, float depth_min, float depth_max,
                          const float* init_depth, const float* init_normal, hipStream_t st);
// `p` describes the (identical) shape of every problem of the batch; `dev_params` is the
// device array of per-problem parameter blocks the kernel indexes with its batch coordinate.
void pm_launch_initial_cost(const PmParams& p, const PmParams* dev_params, int batch, hipStream_t st);
// the random numbers of the next sweep launch (11 x 11 kernel; no-op for shapes the generic kernel serves): call before
// pm_launch_sweep with the same arguments
void pm_launch_draws(const PmParams& p, const PmParams* dev_params, int batch, bool geom, hipStream_t st);
// returns the name of the kernel it launched
const char* pm_launch_sweep(const PmParams& p, const PmParams* dev_params, int batch, int threads, bool geom,
                            bool filter_photo, bool filter_geom, hipStream_t st);
void pm_launch_rng_streams(const unsigned long long* seeds, int nseeds, int ndraws, float* out,
                           hipStream_t st);
void pm_launch_extract(const PmParams& p, int sel_off, float* depth, float* normal, float* sel,
                       float* cost, hipStream_t st);

}  // namespace colmap_amd
