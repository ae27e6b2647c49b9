// rocrand_pin.hip -- TEST INFRASTRUCTURE: pins the XORWOW streams of the PatchMatch path to the
// library the reference's own HIP build links.
//
// The reference seeds one generator per pixel with curand_init(id, 0, 0, &state)
// (src/colmap/mvs/gpu_mat_prng.cu:36-48) and draws with curand_uniform(&state)
// (mvs/gpu_mat.h:370-387, patch_match_cuda.cu:94-196,1055-1062,1129); on ROCm these are
// hipRAND -> rocRAND (rocrand_init / rocrand_uniform on rocrand_state_xorwow). This file calls
// exactly those two functions, once through rocRAND's host path and once in a gfx950 kernel, and
// returns the raw streams; tests/test_pm_oracle.py (host) and tests/test_pm_gpu.py (device)
// compare them bit for bit with oracle/pm_oracle.c:pmo_rng_* and with the product kernels' own
// generator (pm_debug_rng_streams in the C ABI library).
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>

extern "C" __attribute__((visibility("default")))
void rocrand_pin_host(const uint64_t* seeds, int nseeds, int ndraws, float* out) {
  for (int i = 0; i < nseeds; ++i) {
    rocrand_state_xorwow st;
    rocrand_init(seeds[i], 0, 0, &st);
    for (int k = 0; k < ndraws; ++k) out[(size_t)i * ndraws + k] = rocrand_uniform(&st);
  }
}

__global__ void rocrand_pin_kernel(const uint64_t* seeds, int nseeds, int ndraws, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseeds) return;
  rocrand_state_xorwow st;
  rocrand_init(seeds[i], 0, 0, &st);
  for (int k = 0; k < ndraws; ++k) out[(size_t)i * ndraws + k] = rocrand_uniform(&st);
}

// returns 0 on success, a hipError_t otherwise
extern "C" __attribute__((visibility("default")))
int rocrand_pin_device(const uint64_t* seeds, int nseeds, int ndraws, float* out) {
  uint64_t* d_seeds = nullptr;
  float* d_out = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&d_seeds, sizeof(uint64_t) * nseeds)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&d_out, sizeof(float) * (size_t)nseeds * ndraws)) != hipSuccess) return (int)e;
  e = hipMemcpy(d_seeds, seeds, sizeof(uint64_t) * nseeds, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rocrand_pin_kernel, dim3((nseeds + 63) / 64), dim3(64), 0, 0, d_seeds, nseeds,
                       ndraws, d_out);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess)
    e = hipMemcpy(out, d_out, sizeof(float) * (size_t)nseeds * ndraws, hipMemcpyDeviceToHost);
  (void)hipFree(d_seeds);
  (void)hipFree(d_out);
  return (int)e;
}
