// hip_runtime.h of tests/hip_emul -- TEST INFRASTRUCTURE ONLY: a minimal HIP-on-CPU stand-in, just large enough to run
// colmap_amd/csrc/fusion.hip and the bundle-adjustment sources ba_kernels.hip / ba_schur_explicit.hip (host loops AND
// kernels, unmodified source) in a container without a GPU, so that their logic can be compared with the checkers
// in oracle/ by `pytest -m "not gpu"`. Never part of the product: the shipped library is built by hipcc against the
// real headers (colmap_amd/build.py).
//
// Execution model: a launch runs its blocks one after the other; the threads of a block are ucontext fibers of ONE OS
// thread, switched only inside the cross-lane primitives (ballot / shuffle / readfirstlane / wave barrier /
// __syncthreads), each of which is a barrier over the 64 lanes of the calling fiber's wave (or the block). A lane
// therefore runs ahead of its wave between two primitives -- code that hands data from lane to lane through memory
// without a barrier in between, which lock-step hardware forgives, fails here. readfirstlane asserts that the value
// really is uniform. A barrier some lanes never reach is reported instead of hanging.
//
// Fibers are switched by a dozen instructions of x86-64 assembly (callee-saved registers + stack pointer): ucontext's
// swapcontext makes a signal-mask system call per switch, and a cross-lane primitive costs 128 switches.
#pragma once
#include <time.h>

#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
// LDS: one OS thread runs every lane of every workgroup, so "per workgroup" storage is storage of that thread.
// thread_local rather than static because it also combines with `extern` (dynamic LDS: `extern __shared__ char smem[]`
// refers to the array a build script defines, see build_pm.sh); at block scope thread_local implies static.
#define __shared__ thread_local
#define __launch_bounds__(...)
#define __forceinline__ inline

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct alignas(16) double2 { double x, y; };
inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct alignas(16) int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
// HIP declares min / max for device code at global scope
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }

typedef int hipError_t;
struct hip_emul_stream {};
typedef hip_emul_stream* hipStream_t;
struct hip_emul_event { double t = 0.0; };
typedef hip_emul_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocMapped = 2 };
inline const char* hipGetErrorString(hipError_t) { return "hip_emul error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t bytes) {  // device allocations are page-aligned (the host code relies on 256 B)
  *p = nullptr;
  if (posix_memalign(p, 4096, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory;
  // HIP_EMUL_POISON=1: fresh device memory holds 0xCD bytes (NaN-like floats, huge indices) instead of whatever the
  // allocator returns -- a kernel that relies on hipMalloc'd memory being zero shows up as a mismatch or a crash
  static const bool poison = [] { const char* e = std::getenv("HIP_EMUL_POISON"); return e && e[0] != '0'; }();
  if (poison) std::memset(*p, 0xCD, bytes ? bytes : 1);
  return hipSuccess;
}
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned = 0) { *p = std::calloc(bytes ? bytes : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
// streams are synchronous (a launch has finished when hipLaunchKernelGGL returns); events carry wall-clock time
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hip_emul_stream(); return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = new hip_emul_stream(); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new hip_emul_stream(); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hip_emul_event(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hip_emul_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

namespace hip_emul {

constexpr int kWaveSize = 64;
constexpr size_t kFiberStack = 256 * 1024;

#if !defined(__x86_64__)
#error "tests/hip_emul switches fibers with x86-64 assembly"
#endif
// saves the callee-saved registers of the caller on its stack, stores that stack pointer in *save, and resumes the
// context whose stack pointer is `load` (System V ABI: rdi = save, rsi = load)
__attribute__((naked, noinline, used)) static void ctx_switch(void** save, void* load) {
  __asm__ volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret\n\t");
}

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  dim3 tid;
  bool done = true;
};

struct Barrier {
  int count = 0;
  unsigned long gen = 0;
};

struct Block {
  std::vector<Fiber> fibers;
  void* sched = nullptr;
  int cur = 0, nthreads = 0;
  Barrier block_bar;
  std::vector<Barrier> wave_bar;
  std::vector<int> wave_live;            // lanes of a wave / of the block that have not returned from the kernel
  int block_live = 0;
  std::vector<Barrier> row_bar;          // DPP rows of 16 lanes: a row operation only involves its own row (the
  std::vector<int> row_live;             // other rows of the wave may be switched off by divergent control flow)
  std::vector<unsigned long long> slot;  // one per thread: operand of the collective in flight
  unsigned long progress = 0;            // barriers completed (deadlock detection)
  const std::function<void()>* body = nullptr;
};

inline Block& blk() { static Block b; return b; }
inline dim3& block_idx() { static dim3 v; return v; }
inline dim3& block_dim() { static dim3 v; return v; }
inline dim3& grid_dim() { static dim3 v; return v; }
inline Fiber* cur_fiber() { return &blk().fibers[blk().cur]; }
inline int lane_id() { return blk().cur % kWaveSize; }
inline int wave_id() { return blk().cur / kWaveSize; }
inline int wave_lanes(int w) { const int n = blk().nthreads - w * kWaveSize; return n < kWaveSize ? n : kWaveSize; }

inline void yield() {
  Block& b = blk();
  ctx_switch(&b.fibers[b.cur].sp, b.sched);
}

// A barrier is complete when every lane that is still running has arrived: lanes that have returned from the kernel
// no longer take part (as on the hardware, where s_barrier counts the waves that have not ended and an exited lane is
// simply inactive in cross-lane operations).
inline void release_if_complete(Barrier& bar, int live) {
  if (bar.count > 0 && bar.count >= live) {
    bar.count = 0;
    ++bar.gen;
    ++blk().progress;
  }
}
inline void barrier(Barrier& bar, const int& live) {
  const unsigned long g = bar.gen;
  ++bar.count;
  release_if_complete(bar, live);
  while (bar.gen == g) yield();
}
inline void wave_barrier() { barrier(blk().wave_bar[wave_id()], blk().wave_live[wave_id()]); }
inline void block_barrier() { barrier(blk().block_bar, blk().block_live); }
inline void row_barrier() { barrier(blk().row_bar[blk().cur / 16], blk().row_live[blk().cur / 16]); }

inline void trampoline() {
  Block& b = blk();
  (*b.body)();
  b.fibers[b.cur].done = true;
  b.slot[b.cur] = 0ull;  // an exited lane contributes 0 to later ballots of its wave
  const int w = b.cur / kWaveSize;
  --b.wave_live[w];
  --b.block_live;
  --b.row_live[b.cur / 16];
  release_if_complete(b.row_bar[b.cur / 16], b.row_live[b.cur / 16]);
  release_if_complete(b.wave_bar[w], b.wave_live[w]);
  release_if_complete(b.block_bar, b.block_live);
  ctx_switch(&b.fibers[b.cur].sp, b.sched);
  std::abort();  // a finished fiber is never resumed
}

// The array a build defines for `extern __shared__` registers itself here (pm/pm_stubs.cpp) so that HIP_EMUL_POISON can
// refill it before every workgroup: LDS holds whatever the previous workgroup left on the hardware, and a kernel that
// reads a word of it before writing it is a bug that zero-initialised thread storage would hide.
inline unsigned char*& dynamic_lds_base() { static unsigned char* p = nullptr; return p; }
inline size_t& dynamic_lds_size() { static size_t n = 0; return n; }
inline void poison_dynamic_lds() {
  static const bool poison = [] { const char* e = std::getenv("HIP_EMUL_POISON"); return e && e[0] != '0'; }();
  if (poison && dynamic_lds_base()) std::memset(dynamic_lds_base(), 0xCD, dynamic_lds_size());
}

inline void launch(const char* name, dim3 grid, dim3 block, const std::function<void()>& body) {
  static const bool trace = std::getenv("HIP_EMUL_TRACE") != nullptr;
  if (trace) std::fprintf(stderr, "hip_emul: %s grid (%u, %u, %u) block %u\n", name, grid.x, grid.y, grid.z, block.x);
  Block& b = blk();
  const int nt = (int)(block.x * block.y * block.z);
  if ((int)b.fibers.size() < nt) b.fibers.resize(nt);
  b.nthreads = nt;
  b.body = &body;
  b.wave_bar.assign((nt + kWaveSize - 1) / kWaveSize, Barrier());
  b.slot.assign(nt, 0ull);
  block_dim() = block;
  grid_dim() = grid;
  // HIP_EMUL_BLOCK_ORDER=reverse runs the blocks of every launch last to first: the other extreme of the orders in
  // which concurrently resident workgroups can reach a shared word
  static const bool reverse = [] { const char* e = std::getenv("HIP_EMUL_BLOCK_ORDER"); return e && e[0] == 'r'; }();
  // HIP_EMUL_LANE_ORDER=reverse resumes the lanes of a workgroup last to first between two barriers: the last wave runs
  // ahead of the first -- a missing barrier between a producer and a consumer wave (or lane) that the forward order
  // happens to satisfy shows up under the other one
  static const bool lane_reverse = [] { const char* e = std::getenv("HIP_EMUL_LANE_ORDER"); return e && e[0] == 'r'; }();
  const unsigned nblocks = grid.x * grid.y * grid.z;
  for (unsigned bi = 0; bi < nblocks; ++bi) {
    const unsigned bid = reverse ? nblocks - 1 - bi : bi;
    block_idx() = dim3(bid % grid.x, (bid / grid.x) % grid.y, bid / (grid.x * grid.y));
    poison_dynamic_lds();
    b.block_bar = Barrier();
    for (auto& w : b.wave_bar) w = Barrier();
    b.wave_live.assign(b.wave_bar.size(), 0);
    for (int i = 0; i < nt; ++i) ++b.wave_live[i / kWaveSize];
    b.row_bar.assign((nt + 15) / 16, Barrier());
    b.row_live.assign((nt + 15) / 16, 0);
    for (int i = 0; i < nt; ++i) ++b.row_live[i / 16];
    b.block_live = nt;
    std::fill(b.slot.begin(), b.slot.end(), 0ull);
    for (int i = 0; i < nt; ++i) {
      Fiber& f = b.fibers[i];
      if (!f.stack) f.stack = (char*)std::malloc(kFiberStack);
      // first switch into the fiber "returns" into trampoline with the stack aligned as after a call
      void** top = (void**)(((uintptr_t)f.stack + kFiberStack) & ~(uintptr_t)15);
      top[-1] = nullptr;                 // return address of trampoline (it never returns)
      top[-2] = (void*)&trampoline;
      for (int r = 3; r <= 8; ++r) top[-r] = nullptr;  // rbp rbx r12 r13 r14 r15
      f.sp = (void*)(top - 8);
      f.tid = dim3((unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y));
      f.done = false;
    }
    int live = nt;
    while (live > 0) {
      const unsigned long before = b.progress;
      const int live_before = live;
      for (int k = 0; k < nt; ++k) {
        const int i = lane_reverse ? nt - 1 - k : k;
        if (b.fibers[i].done) continue;
        b.cur = i;
        ctx_switch(&b.sched, b.fibers[i].sp);
        if (b.fibers[i].done) --live;
      }
      if (live > 0 && b.progress == before && live == live_before) {
        std::fprintf(stderr, "hip_emul: %s: block %u is stuck at a barrier some of its lanes never reach (live %d; block barrier %d arrived; wave 0: %d of %d)\n",
                     name, bid, live, b.block_bar.count, b.wave_bar[0].count, b.wave_live[0]);
        std::abort();
      }
    }
  }
}

// dynamic LDS of a launch: the array a build script defines for `extern __shared__` holds a CU's 160 KB
inline void check_dynamic_lds(const char* name, size_t bytes) {
  if (bytes > 160u * 1024u) {
    std::fprintf(stderr, "hip_emul: %s asks for %zu bytes of dynamic LDS\n", name, bytes);
    std::abort();
  }
}

// collectives over the wave of the calling lane
template <typename F>
inline unsigned long long collective(unsigned long long mine, F&& f) {
  Block& b = blk();
  const int w = wave_id(), base = w * kWaveSize;
  b.slot[base + lane_id()] = mine;
  wave_barrier();
  const unsigned long long r = f(&b.slot[base], wave_lanes(w));
  wave_barrier();
  return r;
}

}  // namespace hip_emul

#define threadIdx (hip_emul::cur_fiber()->tid)
#define blockIdx (hip_emul::block_idx())
#define blockDim (hip_emul::block_dim())
#define gridDim (hip_emul::grid_dim())
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  (hip_emul::check_dynamic_lds(#kernel, (size_t)(shmem)), \
   hip_emul::launch(#kernel, (grid), (block), std::function<void()>([&]() { kernel(__VA_ARGS__); })))

inline void __syncthreads() { hip_emul::block_barrier(); }
inline void __builtin_amdgcn_wave_barrier() { hip_emul::wave_barrier(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)

inline unsigned long long __ballot(int pred) {
  return hip_emul::collective(pred ? 1ull : 0ull, [](const unsigned long long* s, int n) {
    unsigned long long m = 0ull;
    for (int i = 0; i < n; ++i) m |= (s[i] ? 1ull : 0ull) << i;
    return m;
  });
}
inline unsigned long long hip_emul_shfl_bits(unsigned long long v, int src) {
  return hip_emul::collective(v, [src](const unsigned long long* s, int n) { return s[(src % hip_emul::kWaveSize + hip_emul::kWaveSize) % hip_emul::kWaveSize < n ? (src % hip_emul::kWaveSize + hip_emul::kWaveSize) % hip_emul::kWaveSize : 0]; });
}
inline int __shfl(int v, int src, int = 64) { return (int)(unsigned)hip_emul_shfl_bits((unsigned)v, src); }
inline unsigned __shfl(unsigned v, int src, int = 64) { return (unsigned)hip_emul_shfl_bits(v, src); }
inline float __shfl(float v, int src, int = 64) {
  unsigned u;
  std::memcpy(&u, &v, 4);
  u = (unsigned)hip_emul_shfl_bits(u, src);
  std::memcpy(&v, &u, 4);
  return v;
}
inline double __shfl(double v, int src, int = 64) {
  unsigned long long u;
  std::memcpy(&u, &v, 8);
  u = hip_emul_shfl_bits(u, src);
  std::memcpy(&v, &u, 8);
  return v;
}
inline int __shfl_xor(int v, int mask, int = 64) { return __shfl(v, hip_emul::lane_id() ^ mask); }
inline float __shfl_xor(float v, int mask, int = 64) { return __shfl(v, hip_emul::lane_id() ^ mask); }
inline double __shfl_xor(double v, int mask, int = 64) { return __shfl(v, hip_emul::lane_id() ^ mask); }
inline int __shfl_down(int v, int d, int = 64) { return __shfl(v, hip_emul::lane_id() + d < 64 ? hip_emul::lane_id() + d : hip_emul::lane_id()); }
inline double __shfl_down(double v, int d, int = 64) { return __shfl(v, hip_emul::lane_id() + d < 64 ? hip_emul::lane_id() + d : hip_emul::lane_id()); }
// v_readlane_b32 with a wave-uniform lane select
inline int __builtin_amdgcn_readlane(int v, int src) { return (int)(unsigned)hip_emul_shfl_bits((unsigned)v, src); }
inline int __builtin_amdgcn_readfirstlane(int v) {
  return (int)(unsigned)hip_emul::collective((unsigned)v, [](const unsigned long long* s, int n) {
    for (int i = 1; i < n; ++i)
      if (s[i] != s[0]) {
        std::fprintf(stderr, "hip_emul: readfirstlane of a value that is not wave-uniform (lane 0: %llx, lane %d: %llx)\n", s[0], i, s[i]);
        std::abort();
      }
    return s[0];
  });
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// v_mbcnt_lo / v_mbcnt_hi: bits of the mask half that belong to lanes below the calling one, added to `base`
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) {
  const int l = hip_emul::lane_id();
  return base + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) {
  const int l = hip_emul::lane_id();
  return base + (l > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }

// atomics: one OS thread, fibers switch only inside the primitives above
#if defined(__clang__)  // clang knows __hip_atomic_load / _store as builtins on every target
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#else
#define __HIP_MEMORY_SCOPE_AGENT 0
template <typename T> inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <typename T> inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
#endif
template <typename T> inline T unsafeAtomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
inline void __threadfence() {}
inline void __threadfence_system() {}
using std::isfinite;  // device code calls the global overloads of the HIP headers
using std::isnan;
template <typename T> inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
inline long long __double2ll_rn(double v) {  // round to nearest even (default mode); the device conversion saturates, NaN -> 0
  if (v != v) return 0;
  if (v >= 9223372036854775808.0) return 0x7fffffffffffffffLL;
  if (v < -9223372036854775808.0) return (long long)0x8000000000000000ULL;
  return (long long)std::nearbyint(v);
}
inline long long __double_as_longlong(double v) { long long u; std::memcpy(&u, &v, 8); return u; }
inline double __longlong_as_double(long long u) { double v; std::memcpy(&v, &u, 8); return v; }

// v_mfma_f64_16x16x4_f64: D = A (16 x 4) * B (4 x 16) + C over one wave. Lane l supplies A[l & 15][l >> 4] and
// B[l >> 4][l & 15]; register r of lane l holds C / D [(l >> 4) + 4 * r][l & 15] (cdna_hip_programming.md section 3:
// the f64 form does NOT use the f32 row map; the sources state the same mapping where they unpack the tile). The four products of an entry are added in k order
// with separate multiply and add roundings replaced by fma, like the hardware's fused accumulation.
typedef double hip_emul_v4f64 __attribute__((ext_vector_type(4)));
inline hip_emul_v4f64 hip_emul_mfma_f64_16x16x4(double a, double b, hip_emul_v4f64 c) {
  static double As[16][16][4], Bs[16][4][16];  // per wave of the block; filled between two barriers of the calling wave
  const int l = hip_emul::lane_id();
  double (*A)[4] = As[hip_emul::wave_id()];
  double (*B)[16] = Bs[hip_emul::wave_id()];
  hip_emul::wave_barrier();          // the previous call's readers are done
  A[l & 15][l >> 4] = a;
  B[l >> 4][l & 15] = b;
  hip_emul::wave_barrier();
  const int j = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = (l >> 4) + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k) acc = std::fma(A[i][k], B[k][j], acc);
    c[r] = acc;
  }
  return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, cbsz, abid, blgp) hip_emul_mfma_f64_16x16x4((a), (b), (c))

// ---- what the PatchMatch kernels use beyond the above ----
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }   // v_rcp_f64 (the kernels refine it by Newton steps)
inline float __builtin_amdgcn_fractf(float x) { return x - std::floor(x); }  // v_fract_f32 (finite x >= 0 in the kernels)
inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {            // v_med3_f32: the median; a NaN operand gives min3
  if (a != a || b != b || c != c) return std::fmin(std::fmin(a, b), c);
  return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c));
}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline unsigned long long __builtin_amdgcn_s_memrealtime() { return 0ull; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
#ifndef __HIP_MEMORY_SCOPE_WAVEFRONT
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#endif
#if !defined(__clang__)  // (a clang builtin on every target)
template <typename T, typename U> inline T __hip_atomic_fetch_add(T* p, U v, int, int) { const T o = *p; *p = (T)(o + (T)v); return o; }
#endif
inline int __all(int pred) { return __ballot(pred) == __ballot(1); }
inline int __any(int pred) { return __ballot(pred) != 0ull; }
// v_mov_b32_dpp: lane l of a row of 16 reads lane src(l) of the same row (row_mask / bank_mask 0xf, bound_ctrl off:
// every source lane exists, `old` is never taken). Controls the kernels use: quad_perm (0x00-0xFF), row_mirror (0x140),
// row_half_mirror (0x141), row_shr:n (0x110+n) / row_shl:n (0x100+n) / row_ror:n (0x120+n).
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool bound_ctrl) {
  const int l = hip_emul::lane_id(), row = l & ~15, r = l & 15;
  int from = -1;
  if (ctrl >= 0 && ctrl <= 0xFF) from = row + (r & ~3) + ((ctrl >> (2 * (r & 3))) & 3);
  else if (ctrl == 0x140) from = row + 15 - r;
  else if (ctrl == 0x141) from = row + (r & 8) + 7 - (r & 7);
  else if (ctrl > 0x100 && ctrl <= 0x10F) from = r + (ctrl & 15) < 16 ? l + (ctrl & 15) : -1;   // row_shl
  else if (ctrl > 0x110 && ctrl <= 0x11F) from = r - (ctrl & 15) >= 0 ? l - (ctrl & 15) : -1;   // row_shr
  else if (ctrl > 0x120 && ctrl <= 0x12F) from = row + ((r - (ctrl & 15)) & 15);               // row_ror
  else { std::fprintf(stderr, "hip_emul: DPP control 0x%x not modelled\n", ctrl); std::abort(); }
  // a collective over the ROW only
  hip_emul::Block& b = hip_emul::blk();
  b.slot[b.cur] = (unsigned)src;
  hip_emul::row_barrier();
  const int base = (b.cur / hip_emul::kWaveSize) * hip_emul::kWaveSize;
  const unsigned long long got = from >= 0 && base + from < b.nthreads ? b.slot[base + from] : ~0ull;
  hip_emul::row_barrier();
  if (got == ~0ull) return bound_ctrl ? 0 : old;
  return (int)(unsigned)got;
}
// host API the PatchMatch host code uses
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* free_, size_t* total) { *free_ = (size_t)8 << 30; *total = (size_t)16 << 30; return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                                   hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) std::memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
  return hipSuccess;
}
