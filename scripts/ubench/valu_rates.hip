// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction, measured as
// (kernel time x clock) / (instructions per SIMD). Each kernel runs a register-only loop of 32
// independent instances of one instruction; 4 waves per SIMD keep the pipe full.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY32(INS)                                                                         \
  REP8(asm volatile(INS " %0, %0, %4, %0\n" INS " %1, %1, %4, %1\n" INS " %2, %2, %4, %2\n" \
                    INS " %3, %3, %4, %3"                                                   \
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                    \
                    : "v"(k));)

typedef float v2f __attribute__((ext_vector_type(2)));

#define KERNEL3(NAME, TYPE, INIT, INS)                                  \
  __global__ void NAME(float* out, int iters) {                         \
    TYPE a = INIT, b = INIT, c = INIT, d = INIT, k = INIT;              \
    for (int i = 0; i < iters; ++i) { BODY32(INS) }                     \
    if (threadIdx.x == 9999) out[0] = *(float*)&a + *(float*)&b + *(float*)&c + *(float*)&d; \
  }

KERNEL3(k_fma, float, 1.0f, "v_fma_f32")
KERNEL3(k_pk_fma, v2f, (v2f)(1.0f), "v_pk_fma_f32")
KERNEL3(k_med3, float, 1.0f, "v_med3_f32")
KERNEL3(k_mad_u24, unsigned, 3u, "v_mad_u32_u24")
KERNEL3(k_add3, unsigned, 3u, "v_add3_u32")

#define BODY32_2(INS)                                                                   \
  REP8(asm volatile(INS " %0, %0, %4\n" INS " %1, %1, %4\n" INS " %2, %2, %4\n" INS     \
                        " %3, %3, %4"                                                   \
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                \
                    : "v"(k));)
#define KERNEL2(NAME, TYPE, INIT, INS)                                  \
  __global__ void NAME(float* out, int iters) {                         \
    TYPE a = INIT, b = INIT, c = INIT, d = INIT, k = INIT;              \
    for (int i = 0; i < iters; ++i) { BODY32_2(INS) }                   \
    if (threadIdx.x == 9999) out[0] = *(float*)&a + *(float*)&b + *(float*)&c + *(float*)&d; \
  }
KERNEL2(k_mul, float, 1.0f, "v_mul_f32")
KERNEL2(k_pk_mul, v2f, (v2f)(1.0f), "v_pk_mul_f32")
KERNEL2(k_pk_add, v2f, (v2f)(1.0f), "v_pk_add_f32")
KERNEL2(k_mul_lo, unsigned, 3u, "v_mul_lo_u32")
KERNEL2(k_mul_u24, unsigned, 3u, "v_mul_u32_u24")

#define BODY32_1(INS)                                                                        \
  REP8(asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3"            \
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
#define KERNEL1(NAME, TYPE, INIT, INS)                                  \
  __global__ void NAME(float* out, int iters) {                         \
    TYPE a = INIT, b = INIT, c = INIT, d = INIT;                        \
    for (int i = 0; i < iters; ++i) { BODY32_1(INS) }                   \
    if (threadIdx.x == 9999) out[0] = *(float*)&a + *(float*)&b + *(float*)&c + *(float*)&d; \
  }
KERNEL1(k_floor, float, 1.5f, "v_floor_f32")
KERNEL1(k_cvt_ub, float, 1.5f, "v_cvt_f32_ubyte1")
KERNEL1(k_cvt_i32, float, 1.5f, "v_cvt_i32_f32")
KERNEL1(k_rcp, float, 1.5f, "v_rcp_f32")
KERNEL1(k_mov, float, 1.5f, "v_mov_b32")

// 64-bit shift-add (address arithmetic)
__global__ void k_lshl_add_u64(float* out, int iters) {
  unsigned long long a = 1, b = 2, c = 3, d = 4, k = 5;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %4\nv_lshl_add_u64 %1, %1, 2, %4\n"
                      "v_lshl_add_u64 %2, %2, 2, %4\nv_lshl_add_u64 %3, %3, 2, %4"
                      : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                      : "v"(k));)
  }
  if (threadIdx.x == 9999) out[0] = (float)(a + b + c + d);
}

// dependent chains: one accumulator
__global__ void k_fma_dep(float* out, int iters) {
  float a = 1.0f, k = 1.0f;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_fma_f32 %0, %0, %1, %0\nv_fma_f32 %0, %0, %1, %0\nv_fma_f32 %0, %0, %1, %0\n"
                      "v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(k));)
  }
  if (threadIdx.x == 9999) out[0] = a;
}
__global__ void k_pk_fma_dep(float* out, int iters) {
  v2f a = (v2f)(1.0f), k = (v2f)(1.0f);
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\ns_nop 0\nv_pk_fma_f32 %0, %0, %1, %0\ns_nop 0\n"
                      "v_pk_fma_f32 %0, %0, %1, %0\ns_nop 0\nv_pk_fma_f32 %0, %0, %1, %0\ns_nop 0"
                      : "+v"(a) : "v"(k));)
  }
  if (threadIdx.x == 9999) out[0] = a[0];
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, clk_khz);
  struct K { const char* name; void (*fn)(float*, int); };
  std::vector<K> ks = {{"v_fma_f32", k_fma}, {"v_pk_fma_f32", k_pk_fma}, {"v_mul_f32", k_mul},
                       {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_med3_f32", k_med3},
                       {"v_floor_f32", k_floor}, {"v_cvt_f32_ubyte1", k_cvt_ub}, {"v_cvt_i32_f32", k_cvt_i32},
                       {"v_rcp_f32", k_rcp}, {"v_mov_b32", k_mov}, {"v_mad_u32_u24", k_mad_u24},
                       {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_add3_u32", k_add3},
                       {"v_lshl_add_u64", k_lshl_add_u64}, {"v_fma_f32 dependent", k_fma_dep},
                       {"v_pk_fma_f32 dependent(+nop)", k_pk_fma_dep}};
  const int iters = 4000;
  for (int wps : {1, 4}) {  // waves per SIMD
    printf("-- %d wave(s) per SIMD\n", wps);
    for (auto& k : ks) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
      k.fn<<<blocks, 256>>>(out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k.fn<<<blocks, 256>>>(out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double insts_per_simd = (double)iters * 32 * wps;
      const double cyc = ms * 1e-3 * (clk_khz * 1e3) / insts_per_simd;
      printf("%-32s %8.3f ms  %6.2f cycles / wave64 instruction (at nominal clock)\n", k.name, ms, cyc);
    }
  }
  return 0;
}
