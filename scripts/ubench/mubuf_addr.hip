// What address does a gfx950 MUBUF load form from (index VGPR, offset VGPR) under the stride / swizzle settings of its
// buffer resource? The buffer holds its own dword indices, so the loaded value IS the address (in dwords); the host
// compares it with the candidate formulas. Used to decide whether the sweep kernel's packed-image indexing (row-major:
// y * pitch + x; tiled: 8 x 4-entry tiles) can be left to the address unit instead of 6-7 VALU instructions per tap.
// Build: hipcc --offload-arch=gfx950 -O3 -o mubuf_addr mubuf_addr.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_load(const uint32_t* base, uint32_t w1_hi, uint32_t num_records, uint32_t w3, const uint32_t* idx,
                       const uint32_t* off, uint32_t* out, int n, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t b = (uint64_t)base;
  v4i srd;
  srd[0] = (int)(uint32_t)b;
  srd[1] = (int)((uint32_t)(b >> 32) & 0xffffu) | (int)w1_hi;
  srd[2] = (int)num_records;
  srd[3] = (int)w3;
  // the resource must be wave-uniform (SGPRs)
  srd[0] = __builtin_amdgcn_readfirstlane(srd[0]);
  srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
  srd[2] = __builtin_amdgcn_readfirstlane(srd[2]);
  srd[3] = __builtin_amdgcn_readfirstlane(srd[3]);
  uint32_t vi = i < n ? idx[i] : 0u, vo = i < n ? off[i] : 0u, r = 0xdeadbeefu;
  if (mode == 0) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(vo), "s"(srd) : "memory");
  } else if (mode == 1) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 idxen\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(vi), "s"(srd) : "memory");
  } else {  // modes 2, 3
    uint64_t both = (uint64_t)vi | ((uint64_t)vo << 32);   // v[n] = index, v[n+1] = offset
    asm volatile("buffer_load_dword %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(both), "s"(srd) : "memory");
  }
  if (i < n) out[i] = r;
}

int main() {
  const size_t ndw = 64u << 20;  // 256 MB of dwords holding their own index
  std::vector<uint32_t> h(ndw);
  for (size_t i = 0; i < ndw; ++i) h[i] = (uint32_t)i;
  uint32_t *d, *d_idx, *d_off, *d_out;
  hipMalloc(&d, ndw * 4);
  hipMemcpy(d, h.data(), ndw * 4, hipMemcpyHostToDevice);
  const int n = 4096;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_off, n * 4); hipMalloc(&d_out, n * 4);
  std::vector<uint32_t> idx(n), off(n), out(n);
  uint32_t seed = 12345u;
  auto rnd = [&](uint32_t m) { seed = seed * 1664525u + 1013904223u; return (seed >> 8) % m; };
  const uint32_t W3 = 0x00020000u;  // the raw-buffer word 3 of gfx90a / gfx94x / gfx950
  struct Case { const char* name; int mode; uint32_t stride; bool swz; uint32_t istride_sel, esize_sel; uint32_t nrec; uint32_t max_idx, max_off; };
  const uint32_t pitch = 10304;  // 2576 entries of 4 bytes: a 2560-wide packed image
  Case cases[] = {
    {"raw offen, stride 0", 0, 0, false, 0, 0, (uint32_t)(ndw * 4), 0, (uint32_t)(ndw * 4)},
    {"idxen, stride 16", 1, 16, false, 0, 0, 1u << 24, 1u << 24, 0},
    {"idxen offen, stride 16, offsets beyond the stride", 2, 16, false, 0, 0, 1u << 24, 1u << 20, 1u << 26},
    {"idxen offen, stride = row pitch 10304 (index = row, offset = 4 x)", 2, pitch, false, 0, 0, 1u << 24, 20000, pitch},
    {"idxen offen, stride = row pitch, offsets beyond the stride (image base folded into the offset)", 2, pitch, false, 0, 0, 1u << 24, 2000, 1u << 27},
    {"idxen offen, stride 4 (index = x), offset = row * pitch + image base", 2, 4, false, 0, 0, 1u << 24, 2576, 1u << 27},
    {"idxen offen, stride 5152 (index = row), offset = 2 x + base: dword loads at 2-byte alignment", 3, 5152, false, 0, 0, 1u << 24, 20000, 5152},
    {"swizzle: stride 16, index_stride 8, element 4 (index = x), small offsets", 2, 16, true, 0, 1, 1u << 24, 2576, 16},
    {"swizzle: stride 16, index_stride 8, element 4 (index = x), large offsets", 2, 16, true, 0, 1, 1u << 24, 2576, 1u << 24},
    {"swizzle: stride 128, index_stride 8, element 4 (index = x), large offsets", 2, 128, true, 0, 1, 1u << 24, 2576, 1u << 24},
  };
  for (const Case& c : cases) {
    for (int i = 0; i < n; ++i) {
      idx[i] = c.max_idx ? rnd(c.max_idx) : 0u;
      off[i] = c.max_off ? (rnd(c.max_off) & (c.mode == 3 ? ~1u : ~3u)) : 0u;
    }
    hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_off, off.data(), n * 4, hipMemcpyHostToDevice);
    const uint32_t w1_hi = (c.stride << 16) | (c.swz ? 0x80000000u : 0u);
    const uint32_t w3 = W3 | (c.swz ? (c.istride_sel << 21) | (c.esize_sel << 19) : 0u);
    hipLaunchKernelGGL(k_load, dim3(n / 64), dim3(64), 0, 0, d, w1_hi, c.nrec, w3, d_idx, d_off, d_out, n, c.mode);
    hipDeviceSynchronize();
    hipMemcpy(out.data(), d_out, n * 4, hipMemcpyDeviceToHost);
    const uint32_t istr[4] = {8, 16, 32, 64}, esz[4] = {2, 4, 8, 16};
    int ok_lin = 0, ok_swz = 0, zeros = 0, inrange = 0;
    for (int i = 0; i < n; ++i) {
      const uint64_t lin = (uint64_t)idx[i] * c.stride + off[i];
      const uint32_t is = istr[c.istride_sel], es = esz[c.esize_sel];
      const uint64_t swz = ((uint64_t)(idx[i] / is) * c.stride + (uint64_t)(off[i] / es) * es) * is + (idx[i] % is) * es + off[i] % es;
      if (lin / 4 < ndw) {
        ++inrange;
        // value of the 4 bytes at byte address lin of an array of little-endian dword indices
        const uint64_t w0 = lin / 4, sh = (lin % 4) * 8;
        const uint32_t expect = sh ? (uint32_t)(((w0 | ((w0 + 1) << 32)) >> sh) & 0xffffffffu) : (uint32_t)w0;
        ok_lin += out[i] == expect;
      }
      if (swz / 4 < ndw) ok_swz += out[i] == (uint32_t)(swz / 4);
      zeros += out[i] == 0u;
    }
    printf("%-100s linear %4d / %4d   swizzled %4d   zero results %4d", c.name, ok_lin, inrange, ok_swz, zeros);
    // a few samples where neither matches
    int shown = 0;
    for (int i = 0; i < n && shown < 3; ++i) {
      const uint64_t lin = (uint64_t)idx[i] * c.stride + off[i];
      if (lin / 4 < ndw && out[i] != (uint32_t)(lin / 4) && !c.swz) { printf("  [idx %u off %u -> %u]", idx[i], off[i], out[i]); ++shown; }
    }
    if (c.swz) for (int i = 0; i < 3; ++i) printf("  [idx %u off %u -> byte %llu]", idx[i], off[i], (unsigned long long)out[i] * 4ull);
    printf("\n");
  }
  return 0;
}
