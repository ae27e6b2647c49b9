/*
 * colmap_amd_fusion.h -- C ABI of the depth-map fusion step that consumes the PatchMatch output
 * (SURVEY.md section 8f row 3).
 *
 * Replaces colmap::mvs::StereoFusion::Run / Fuse (reference src/colmap/mvs/fusion.cc:135-524) for
 * inputs that are already in memory: the caller (colmap_amd/fusion.py, the `stereo_fusion` command)
 * does the workspace reading the reference does through mvs::Workspace. The inputs are host buffers
 * (the reference reads them from the workspace files); the traversal, the medians and the compaction
 * run on the GPU (colmap_amd/csrc/fusion.hip) and there is no CPU path. The pixels of an image take
 * their turns in the order of the reference's own thread-pool schedule (stripes of ten rows, fusion.cc:253-269,
 * its num_threads threads advancing in step: deterministic, where the reference's order depends on thread
 * timing unless num_threads = 1); the result is the reference's algorithm run sequentially in that order.
 * Differences: max_num_pixels above 16 384 is clamped (the reference's default 10 000 is honoured in full),
 * visibility lists are sorted (the reference copies an unordered set). No limit on the number of images
 * other than HBM (~40 B per depth-map pixel resident; 64-bit pixel offsets).
 */
#ifndef COLMAP_AMD_FUSION_H_
#define COLMAP_AMD_FUSION_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* colmap::mvs::StereoFusionOptions (mvs/fusion.h:46-94); the workspace-side fields (mask_path,
 * max_image_size, use_cache, cache_size) stay with the caller. */
typedef struct fusion_options {
  int32_t min_num_pixels;      /* 5 */
  int32_t max_num_pixels;      /* 10000 */
  int32_t max_traversal_depth; /* 100 */
  int32_t check_num_images;    /* 50 (used by the caller to build the overlap lists) */
  double max_reproj_error;     /* 2 px */
  double max_depth_error;      /* 0.01 relative */
  double max_normal_error;     /* 10 degrees */
  float bbox_min[3], bbox_max[3]; /* -FLT_MAX / FLT_MAX */
  /* StereoFusionOptions::num_threads (mvs/fusion.h:53): the size of the reference's thread pool, whose tasks are
   * stripes of ten rows walked row-major (fusion.cc:253-269, 293-300). The turn order IS that pool's schedule with
   * its threads advancing in step: thread t takes stripes t, t + T, t + 2T, ... and every tick each thread takes
   * the next pixel of its stripe. 1 = plain row-major (the reference with one thread, bit for bit);
   * <= 0 (the default -1) = one thread per stripe. */
  int32_t num_threads;
} fusion_options;

/* One workspace image: mvs::Image pose at the MODEL image size, its colour bitmap, and the depth /
 * normal maps (Mat<float>, normal slice-major) at the depth-map size. used = 0 skips the image
 * (fusion.cc:204-213). mask: optional depth-map-sized bytes, non-zero = pre-masked pixel (:374-399). */
typedef struct fusion_image {
  int32_t width, height;
  float K[9], R[9], T[3];
  const uint8_t* rgb; /* bitmap_height * bitmap_width * 3, or NULL (colour 0) */
  int32_t bitmap_width, bitmap_height;
  const float* depth_map;
  const float* normal_map;
  int32_t depth_width, depth_height;
  const uint8_t* mask;
  int32_t used;
} fusion_image;

typedef struct fusion_result fusion_result;

void fusion_options_init(fusion_options* options);
/* StereoFusionOptions::Check (fusion.cc:96-106): 0 = valid. */
int fusion_options_check(const fusion_options* options);

/* overlapping images of image i: overlap_idx[overlap_ptr[i] .. overlap_ptr[i+1])
 * (Model::GetMaxOverlappingImages(check_num_images, 0), fusion.cc:180-186). */
int fusion_run(const fusion_options* options, int32_t num_images, const fusion_image* images,
               const int32_t* overlap_ptr, const int32_t* overlap_idx, fusion_result** out);

size_t fusion_num_points(const fusion_result* r);
/* PlyPoint fields: xyz_normal [n][6] floats, rgb [n][3] */
int fusion_get_points(const fusion_result* r, float* xyz_normal, uint8_t* rgb);
/* visibility (fusion.cc:514-517): vis_ptr [n+1], vis_idx [vis_ptr[n]]; pass NULLs to query the total */
int fusion_get_visibility(const fusion_result* r, int64_t* vis_ptr, int32_t* vis_idx, size_t* total);
void fusion_free(fusion_result* r);
/* Counters of the last fusion_run on this process: images fused, their pixels, commit rounds, and
 * walks summed over the rounds (walks / seeds = average number of turns a pixel needed). */
void fusion_last_stats(int64_t* images, int64_t* seeds, int64_t* rounds, int64_t* walks);
/* ... and where its time went: host maps -> HBM plus workspace set-up, and everything after (the rounds on the
 * device, medians, compaction, read-back of the points). */
void fusion_last_timing(double* upload_seconds, double* device_seconds);
/* ... and how many of its walks were started breadth-first, met a limit of the traversal (max_traversal_depth,
 * max_num_pixels) and were repeated depth-first (colmap_amd/csrc/fusion.hip: walk_turn_wide). */
int64_t fusion_last_redone_walks(void);
const char* fusion_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* COLMAP_AMD_FUSION_H_ */
