/*
 * colmap_amd_pm.h -- C ABI of the MI355X-native PatchMatch multi-view stereo.
 *
 * Drop-in boundary: the private pimpl `std::unique_ptr<PatchMatchCuda>` inside
 * colmap::mvs::PatchMatch (reference src/colmap/mvs/patch_match.h:95, used at
 * patch_match.cc:128-153). Each entry point below replaces one member of
 * `class PatchMatchCuda` (reference src/colmap/mvs/patch_match_cuda.h:49-59);
 * INTEGRATION.md shows the C++ shim a COLMAP maintainer would add.
 *
 * Conventions
 *   - plain C types only, no torch/HIP types; all matrices row-major float.
 *   - every function returns 0 on success, non-zero on error; the message of
 *     the last error on the calling thread is pm_last_error() (the reference
 *     throws from THROW_CHECK / CUDA_SAFE_CALL, util/cudacc.cc:56-65; the shim
 *     re-throws).
 *   - the caller owns all input buffers for the lifetime of pm_create() only:
 *     inputs are copied to HBM inside pm_create (the reference's ctor uploads in
 *     InitRefImage/InitSourceImages, patch_match_cuda.cu:1290-1302).
 *   - handles are independent and re-entrant across host threads; one handle is
 *     bound to one GPU (options.gpu_index) and one HIP stream, so a host thread
 *     per GPU (reference PatchMatchController, patch_match.cc:177,394) or several
 *     handles per GPU both work.
 */
#ifndef COLMAP_AMD_PM_H_
#define COLMAP_AMD_PM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* colmap::mvs::PatchMatchOptions (reference mvs/patch_match_options.h:37-126),
 * same names, same units (angles in degrees), same defaults via
 * pm_options_init(). Fields that only concern the controller (cache_size,
 * max_image_size, num_threads, allow_missing_files, write_consistency_graph)
 * stay on the host side. */
typedef struct pm_options {
  double depth_min;
  double depth_max;
  double sigma_spatial;
  double sigma_color;
  double ncc_sigma;
  double min_triangulation_angle;
  double incident_angle_sigma;
  double geom_consistency_regularizer;
  double geom_consistency_max_cost;
  double filter_min_ncc;
  double filter_min_triangulation_angle;
  double filter_geom_consistency_max_cost;
  int32_t window_radius;
  int32_t window_step;
  int32_t num_samples;
  int32_t num_iterations;
  int32_t filter_min_num_consistent;
  int32_t geom_consistency; /* bool */
  int32_t filter;           /* bool */
  int32_t gpu_index;        /* single device ordinal; -1 = current device */
  /* extensions (0 = reference behaviour) */
  int32_t max_sweeps;       /* debug: >0 stop after this many sweeps; 0: all; <0: initial cost only */
  int32_t inputs_on_device; /* 1: gray/depth/normal pointers are device pointers */
  int32_t columns_per_group;/* tuning: image columns per workgroup, 0 = auto */
  int32_t threads_per_group;/* tuning: 0 = auto */
} pm_options;

/* colmap::mvs::Image (reference mvs/image.h:40-98) + the DepthMap / NormalMap of
 * the same image (PatchMatch::Problem::depth_maps / normal_maps,
 * mvs/patch_match.h:57-75). */
typedef struct pm_image {
  int32_t width, height;
  float K[9];            /* only fx, fy, cx, cy may be non-trivial (patch_match.cc:101-106) */
  float R[9];
  float T[3];
  const uint8_t* gray;   /* height*width grey bitmap, tightly packed rows */
  const float* depth_map;  /* height*width, or NULL */
  const float* normal_map; /* 3*height*width slice-major (Mat<float>, mat.h:107-109), or NULL */
} pm_image;

/* colmap::mvs::PatchMatch::Problem (reference mvs/patch_match.h:57-75) */
typedef struct pm_problem {
  int32_t ref_image_idx;
  int32_t num_src_images;
  const int32_t* src_image_idxs;
  int32_t num_images;
  const pm_image* images;
} pm_problem;

typedef struct pm_handle pm_handle;

/* PatchMatchOptions default member initialisers (patch_match_options.h:37-126). */
void pm_options_init(pm_options* options);

/* PatchMatchOptions::Check (patch_match_options.cc:73-100) + PatchMatch::Check
 * (patch_match.cc:67-126). */
int pm_check(const pm_options* options, const pm_problem* problem);

/* PatchMatchCuda::PatchMatchCuda(options, problem) (patch_match_cuda.cu:1290-1302):
 * validates, uploads, filters the reference image, builds the pose tables,
 * initialises depth/normal/PRNG state. */
int pm_create(const pm_options* options, const pm_problem* problem, pm_handle** out);

/* Device-side cache of packed source images shared by problems (the reference re-uploads every
 * source bitmap for every problem, patch_match_cuda.cu:1595-1654; its host-side equivalent is the
 * CachedWorkspace, mvs/workspace.h). Entries are keyed by the caller's bitmap pointer and sizes, so
 * the caller must keep a bitmap's address stable and unmodified while it is cached. Handles keep
 * their sources alive; destroy the cache when no problem will reuse it. A cache belongs to one GPU
 * (gpu_index, or with -1 the device of the first problem created with it): pm_create_cached fails
 * for a problem on another device. Thread-safe. */
typedef struct pm_image_cache pm_image_cache;
int pm_image_cache_create(int32_t gpu_index, pm_image_cache** out);
void pm_image_cache_destroy(pm_image_cache* cache);
/* Soft limit in bytes (default: unlimited): on insertion the oldest entries that no live problem
 * references are dropped until the cache fits. */
int pm_image_cache_set_capacity(pm_image_cache* cache, size_t max_bytes);
int pm_image_cache_stats(pm_image_cache* cache, size_t* entries, size_t* hits, size_t* misses);
int pm_create_cached(const pm_options* options, const pm_problem* problem, pm_image_cache* cache,
                     pm_handle** out);

/* PatchMatchCuda::Run() (patch_match_cuda.cu:1304-1352,1393-1546): blocking. */
int pm_run(pm_handle* h);
/* Same work, enqueued on the handle's stream; pm_synchronize() waits. */
int pm_run_async(pm_handle* h);
int pm_synchronize(pm_handle* h);

/* Solve `n` problems of identical image size / source count / options together: each
 * kernel launch covers all n reference images (the reference runs one problem per
 * GPU at a time, patch_match.cc:394; a single 2560-wide image cannot fill 256 CUs).
 * Results are bit-identical to n separate pm_run() calls. Timing of the batched
 * sweep launches is reported by pm_get_sweep_timing(handles[0]) (see pm_get_launch_shape). */
int pm_run_batch(pm_handle** handles, int32_t n);
/* How the last run of this handle was launched: reference images per sweep launch (its sub-batch: pm_run_batch
 * runs 16 or more problems as two sub-batches on two streams so that the drain of one sweep launch is filled by
 * the other's) and how many such launches were in flight together. pm_get_sweep_timing times the launches of the
 * handle's own sub-batch. */
int pm_get_launch_shape(pm_handle* h, int32_t* images_per_launch, int32_t* concurrent_launches);
int pm_run_batch_async(pm_handle** handles, int32_t n); /* then pm_synchronize() each, before destroying any */

/* PatchMatchCuda::GetDepthMap / GetNormalMap / GetSelProbMap
 * (patch_match_cuda.cu:1354-1365). out buffers are host memory:
 * depth H*W, normal 3*H*W slice-major, sel_prob S*H*W slice-major. */
int pm_get_depth_map(pm_handle* h, float* out);
int pm_get_normal_map(pm_handle* h, float* out);
int pm_get_sel_prob_map(pm_handle* h, float* out);

/* PatchMatchCuda::GetConsistentImageIdxs (patch_match_cuda.cu:1367-1391): flat
 * list [col, row, n, idx_1..idx_n]* with idx = problem.src_image_idxs[d]
 * (consumed by ConsistencyGraph, mvs/consistency_graph.cc:121-139). Call with
 * buf == NULL to obtain the required length in *count. */
int pm_get_consistent_image_idxs(pm_handle* h, int32_t* buf, size_t capacity, size_t* count);

/* Extras used by tests/bench (no reference counterpart). */
int pm_get_cost_map(pm_handle* h, float* out);            /* S*H*W */
int pm_get_consistency_mask(pm_handle* h, uint8_t* out);  /* S*H*W */
int pm_get_ref_filter(pm_handle* h, uint8_t* image, float* sum, float* sqsum); /* H*W each */
int pm_get_pose_tables(pm_handle* h, float* poses /*4*S*43*/, float* ref_K /*16*/, float* ref_inv_K /*16*/);
/* HIP-event timing of the sweep kernel launches of the last run, measured on
 * the handle's stream: total ms and launch count. */
int pm_get_sweep_timing(pm_handle* h, double* total_ms, int32_t* num_launches);
/* ... and launch by launch: ms[i] = duration of sweep launch i of the last run (i < min(capacity, *num_launches)). */
int pm_get_sweep_times(pm_handle* h, float* ms, int32_t capacity, int32_t* num_launches);
/* Name of the sweep kernel the last run launched ("pm_sweep_quad_kernel"; "pm_sweep_pair_kernel" = two waves per
 * column, what a launch that cannot fill the GPU with one wave per column runs -- a lone large problem;
 * "pm_sweep_kernel" = other window sizes; static string, valid for the life of the library; the handle that led the
 * batch). */
int pm_get_sweep_kernel_name(pm_handle* h, const char** name);
/* Bilaterally weighted NCC evaluations (PhotoConsistencyCostComputer::Compute,
 * patch_match_cuda.cu:489-593; (2 r / step + 1)^2 taps each) the last run actually executed: in the
 * sweep launches (identical hypothesis / view pairs of a pixel are evaluated once) and in
 * ComputeInitialCost (W * H * S). With geom_consistency the sweep count also includes the
 * geometric-cost-only entries when the two-wave kernel runs. bench.py turns it into taps/s. */
int pm_get_evaluation_count(pm_handle* h, unsigned long long* sweep_evals, unsigned long long* initial_evals);
/* Device pointers of the result maps in API layout (valid after pm_synchronize;
 * for consumers that stay on the GPU, e.g. the geometric pass). */
int pm_get_device_maps(pm_handle* h, const float** depth, const float** normal);
/* Device-to-device copy of the result maps into caller-owned DEVICE buffers (depth: H*W floats,
 * normal: 3*H*W floats, either may be NULL), ordered after the run on the handle's stream and
 * complete on return. Keeps the photometric maps in HBM for the geometric pass (replaces the
 * reference's write-to-disk / read-back, patch_match.cc:507-508,530-531). */
int pm_copy_maps_to_device(pm_handle* h, float* depth_dev, float* normal_dev);

/* Debug: per-phase shader-clock totals of the sweep kernel, photometric sweeps of the 11 x 11 window. With the
 * profile enabled the run launches the profiling build of the shipped four-wave kernel (every wave adds its
 * phase totals when it retires): pm_get_phase_profile_slots returns PM_PROFILE_SLOTS counters -- 0 set-up + backward
 * messages, 1 tile scroll, 2 hypotheses, 3 patch weights, 4 priors, 5 CDF, 6 draws, 7 task lists, 8-10 hypothesis
 * NCC (homographies, tap rounds, normalisation), 11-13 sums / argmin / winner tasks, 14-16 winner NCC, 17 messages and
 * record stores, 18 filter + row end, 23 = number of waves that reported. pm_get_phase_profile returns the first ten
 * (the generic kernel's coarser split when that kernel ran). */
#define PM_PROFILE_SLOTS 24
int pm_enable_phase_profile(pm_handle* h, int enable);
int pm_get_phase_profile(pm_handle* h, unsigned long long* out10);
int pm_get_phase_profile_slots(pm_handle* h, unsigned long long* out, int32_t capacity);
/* Debug: progress trace of the 11 x 11 sweep kernel. When enabled every wave stores the device-wide clock
 * (100 MHz, s_memrealtime) at rows 0, 128, 256, ... of its column group; the buffer holds the last sweep launch. out[group * samples + row / 128]; 0 = not reached. How far the waves of a launch drift apart in
 * the sweep direction decides how much source-image data is in use at a time. */
int pm_enable_progress_trace(pm_handle* h, int enable);
int pm_get_progress_trace(pm_handle* h, unsigned long long* out, size_t capacity, int32_t* groups, int32_t* samples);

/* Debug: the first `ndraws` uniforms of the kernels' XORWOW generator for each 64-bit seed
 * (= curand_init(seed, 0, 0) + curand_uniform, gpu_mat_prng.cu:36-48), computed on GPU
 * `gpu_index`; out is host memory [nseeds][ndraws]. The tests compare it bit for bit with
 * rocRAND (the library the reference's HIP build links). */
int pm_debug_rng_streams(int32_t gpu_index, const uint64_t* seeds, int32_t nseeds, int32_t ndraws,
                         float* out);

/* Development switches (colmap_amd/csrc/switches.h): selects a kernel variant kept for A/B comparisons ("COLMAP_AMD_PM_QUAD",
 * "COLMAP_AMD_BA_FORM_PAIRS", ...) for this process; value NULL restores the built-in default. The library reads no
 * environment variables; tests and bench.py's A/B legs call this. Serves all three paths (PatchMatch, BA, fusion). */
void colmap_amd_set_switch(const char* name, const char* value);

/* Debug: packed source images are allocated from slabs (3.5 GB for full-size images) and the images of one problem
 * must lie within 4 GB of each other to be read through one buffer resource; cached images that do not are re-homed
 * (copied into the slab that holds most of the problem's images). `slots` > 0 puts the allocator into a test mode --
 * `slots` images per slab, the span limit = one slab -- so that a handful of small images exercises that path; 0 =
 * the hardware's limits. Call before any image is packed. Returns the number of images re-homed so far. */
unsigned long long pm_debug_set_image_slab_slots(size_t slots);

void pm_destroy(pm_handle* h);
/* Device buffers of destroyed handles are kept (exact-size free lists, at most 64 GB and a quarter of the device's
 * memory unless pm_set_cached_memory_limit says otherwise) for the next handle of the same shape;
 * pm_release_cached_memory returns them to the driver, pm_set_cached_memory_limit(gigabytes) changes the bound from
 * then on (0 = keep nothing; what is held beyond the new bound is released at once). */
void pm_release_cached_memory(void);
int pm_set_cached_memory_limit(double gigabytes);
const char* pm_last_error(void);
/* Number of visible GPUs (controller: gpu_index == -1 -> all devices,
 * patch_match.cc:375-383). */
int pm_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* COLMAP_AMD_PM_H_ */
