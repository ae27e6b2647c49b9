// What slows the serial chain of the blocked Cholesky when the bulk trailing update runs beside it? (MEASUREMENT)
// In the round-5 kernel trace a diagonal-block kernel takes 22-28 us alone and 107-162 us while chol_update128_kernel
// runs on the lookahead stream; CU masks, a third stream and s_setprio did not change that. This program times a train
// of 40 chol_diag_kernel launches (the product's kernel: the .hip file is included) alone and beside four "hogs" on a
// second stream, each isolating one shared resource:
//   mem   a streaming copy over 4 GB (HBM / fabric / L2 traffic, no arithmetic)
//   mfma  v_mfma_f64_16x16x4_f64 in registers on every CU (fp64 matrix pipe + power, no memory)
//   lds   ds_read / ds_write loops on every CU (LDS bandwidth, no global memory)
//   valu  v_fma_f64 chains on every CU (vector fp64 pipe, no memory)
//   real  chol_update128_kernel itself on an 8 192^2 matrix
// Build (container): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I colmap_amd/csrc -o scripts/ubench/chol_contention \
//        scripts/ubench/chol_contention.hip ; run on the GPU box.
#include "../../colmap_amd/csrc/ba_schur_explicit.hip"

#include <cstdio>
#include <vector>

using namespace ba_explicit;

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void hog_mem(const double4* __restrict__ a, double4* __restrict__ b, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256, 2) hog_mfma(double* out, int iters) {
  v4d acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = v4d{0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  double s = 0;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678) out[0] = s;
}
__global__ void __launch_bounds__(256, 2) hog_lds(double* out, int iters) {
  __shared__ double buf[4096];
  double s = 0;
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = i;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) s += buf[(threadIdx.x * 2 + 64 * k + i) & 4095];
  }
  if (s == 12345.678) out[0] = s;
}
__global__ void __launch_bounds__(256, 2) hog_valu(double* out, int iters) {
  double x[8];
  for (int k = 0; k < 8; ++k) x[k] = 1.0 + threadIdx.x * 1e-9 + k;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = fma(x[k], 0.999999, 1e-7);
  double s = 0;
  for (int k = 0; k < 8; ++k) s += x[k];
  if (s == 12345.678) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int n = 8192;
  double *S = nullptr, *Linv = nullptr, *big_a = nullptr, *big_b = nullptr, *sink = nullptr, *Sbig = nullptr;
  int* info = nullptr;
  CK(hipMalloc(&S, sizeof(double) * 64 * 64 * 64));           // 64 independent 64 x 64 blocks (n = 64 each)
  CK(hipMalloc(&Linv, sizeof(double) * 64 * 64 * 64));
  CK(hipMalloc(&info, sizeof(int) * 4));
  CK(hipMalloc(&sink, 64));
  const size_t big = (size_t)1 << 31;                          // 2 GB each
  CK(hipMalloc(&big_a, big));
  CK(hipMalloc(&big_b, big));
  CK(hipMalloc(&Sbig, sizeof(double) * ((size_t)n * n + n)));
  CK(hipMemset(big_a, 0, big));
  CK(hipMemset(Sbig, 0, sizeof(double) * ((size_t)n * n + n)));
  std::vector<double> h(64 * 64 * 64);
  for (int b = 0; b < 64; ++b)
    for (int i = 0; i < 64; ++i)
      for (int j = 0; j < 64; ++j) h[(size_t)b * 4096 + i * 64 + j] = i == j ? 64.0 : 0.5;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, h0, h1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
  const char* names[] = {"alone", "mem", "mfma", "lds", "valu", "real"};
  for (int rep = 0; rep < 2; ++rep)
    for (int hog = 0; hog < 6; ++hog) {
      CK(hipMemcpy(S, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
      CK(hipMemset(info, 0, sizeof(int) * 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(h0, sb));
      switch (hog) {
        case 1: hipLaunchKernelGGL(hog_mem, dim3(2048), dim3(256), 0, sb, (const double4*)big_a, (double4*)big_b, big / 32, 12); break;
        case 2: hipLaunchKernelGGL(hog_mfma, dim3(512), dim3(256), 0, sb, sink, 60000); break;
        case 3: hipLaunchKernelGGL(hog_lds, dim3(512), dim3(256), 0, sb, sink, 400000); break;
        case 4: hipLaunchKernelGGL(hog_valu, dim3(512), dim3(256), 0, sb, sink, 600000); break;
        case 5:
          for (int k = 0; k < 6; ++k)
            hipLaunchKernelGGL(chol_update128_kernel, dim3((n - 256 + 127) / 128, (n - 256 + 127) / 128), dim3(256), 0, sb, Sbig, n, n,
                               0, 256, 256, 256, n);
          break;
        default: break;
      }
      CK(hipEventRecord(h1, sb));
      // let the hog get going, then the train of diagonal-block kernels (independent blocks, back to back)
      if (hog) { hipEvent_t w; CK(hipEventCreate(&w)); (void)w; }
      CK(hipEventRecord(e0, sa));
      for (int k = 0; k < 40; ++k)
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(128), 0, sa, S + (size_t)k * 4096, 64, 0, 64, Linv + (size_t)k * 4096, info);
      CK(hipEventRecord(e1, sa));
      CK(hipDeviceSynchronize());
      float td = 0, th = 0;
      CK(hipEventElapsedTime(&td, e0, e1));
      CK(hipEventElapsedTime(&th, h0, h1));
      int hinfo = 0;
      CK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
      printf("rep %d  %-6s diag train: %8.1f us per kernel   (hog ran %8.2f ms, pivot flag %d)\n", rep, names[hog], td * 1e3 / 40, th, hinfo);
    }
  return 0;
}
