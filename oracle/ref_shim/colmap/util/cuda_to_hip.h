// TEST INFRASTRUCTURE (oracle/ref_shim/README.md). Shadows the reference's
// src/colmap/util/cuda_to_hip.h: includes the reference's own rename table, then redirects the
// texture API to a software texture, because gfx950 has no image instructions (hipcc refuses
// tex2D / tex2DLayered for it) -- see the README for the exact semantics emulated.
#pragma once

#include_next "colmap/util/cuda_to_hip.h"

#include <cstdint>
#include <cstring>

namespace ref_shim {

// Linear-memory stand-in for a (layered) cudaArray.
struct SoftArray {
  void* data = nullptr;  // device, width * height * depth elements, slice-major
  size_t width = 0, height = 0, depth = 0;
  int elem_bytes = 0;    // 1 (uint8) or 4 (float)
};

// Device-resident texture descriptor; its device address is handed out as the texture object.
struct SoftTexture {
  const void* data;
  int width, height, depth;
  int elem_bytes;
  int normalized_float;  // cudaReadModeNormalizedFloat on a uint8 array
};

hipError_t Malloc3DArray(hipArray_t* array, const hipChannelFormatDesc* desc, hipExtent extent, unsigned int flags);
hipError_t FreeArray(hipArray_t array);
hipError_t Memcpy3D(const hipMemcpy3DParms* p);
hipError_t CreateTextureObject(hipTextureObject_t* tex, const hipResourceDesc* res, const hipTextureDesc* desc,
                               const void* view);
hipError_t DestroyTextureObject(hipTextureObject_t tex);

#if defined(__HIPCC__)
// Point filter, border address mode, unnormalised coordinates: texel index = floor(coordinate),
// 0 outside [0, size). A uint8 texel read as normalised float is b / 255.
template <typename T>
__device__ inline T Fetch(hipTextureObject_t obj, float x, float y, int layer) {
  const SoftTexture* t = reinterpret_cast<const SoftTexture*>(obj);
  const float fx = floorf(x), fy = floorf(y);
  if (!(fx >= 0.0f) || !(fy >= 0.0f) || !(fx < static_cast<float>(t->width)) ||
      !(fy < static_cast<float>(t->height)) || layer < 0 || layer >= t->depth)
    return T(0);
  const size_t i = (static_cast<size_t>(layer) * t->height + static_cast<size_t>(fy)) * t->width +
                   static_cast<size_t>(fx);
  if (t->elem_bytes == 1) {
    const uint8_t b = static_cast<const uint8_t*>(t->data)[i];
    return t->normalized_float ? static_cast<T>(static_cast<float>(b) / 255.0f) : static_cast<T>(b);
  }
  return static_cast<T>(static_cast<const float*>(t->data)[i]);
}

template <typename T, typename X, typename Y>
__device__ inline T Tex2D(hipTextureObject_t obj, X x, Y y) {
  return Fetch<T>(obj, static_cast<float>(x), static_cast<float>(y), 0);
}
template <typename T, typename X, typename Y>
__device__ inline T Tex2DLayered(hipTextureObject_t obj, X x, Y y, int layer) {
  return Fetch<T>(obj, static_cast<float>(x), static_cast<float>(y), layer);
}
#endif

}  // namespace ref_shim

#undef cudaMalloc3DArray
#undef cudaFreeArray
#undef cudaMemcpy3D
#undef cudaCreateTextureObject
#undef cudaDestroyTextureObject
#define cudaMalloc3DArray ref_shim::Malloc3DArray
#define cudaFreeArray ref_shim::FreeArray
#define cudaMemcpy3D ref_shim::Memcpy3D
#define cudaCreateTextureObject ref_shim::CreateTextureObject
#define cudaDestroyTextureObject ref_shim::DestroyTextureObject
#define tex2D ref_shim::Tex2D
#define tex2DLayered ref_shim::Tex2DLayered
