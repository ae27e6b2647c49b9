// pm_gfx950_asm.h of tests/hip_emul -- TEST INFRASTRUCTURE ONLY (see ../hip/hip_runtime.h): what the five
// inline-assembly helpers of colmap_amd/csrc/gfx950/pm_gfx950_asm.h compute, in C++, so that the unmodified
// pm_kernels.hip runs on the CPU stand-in. Put first on the include path by build_pm.sh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace colmap_amd {

// v_cvt_f32_ubyteK
inline float ubyte0(uint32_t x) { return (float)(x & 0xffu); }
inline float ubyte1(uint32_t x) { return (float)((x >> 8) & 0xffu); }
inline float ubyte2(uint32_t x) { return (float)((x >> 16) & 0xffu); }
inline float ubyte3(uint32_t x) { return (float)(x >> 24); }

// 12 v_add_f32_dpp: row_mirror, row_half_mirror, quad_perm [3,2,1,0], quad_perm [1,0,3,2] on each of the three values
inline float hip_emul_dpp_f(float v, int ctrl) {
  int i;
  std::memcpy(&i, &v, 4);
  i = __builtin_amdgcn_update_dpp(0, i, ctrl, 0xf, 0xf, false);
  std::memcpy(&v, &i, 4);
  return v;
}
inline void reduce16x3(float& a, float& b, float& c) {
  const int steps[4] = {0x140, 0x141, 0x1B, 0xB1};
  for (int s = 0; s < 4; ++s) {
    a = a + hip_emul_dpp_f(a, steps[s]);
    b = b + hip_emul_dpp_f(b, steps[s]);
    c = c + hip_emul_dpp_f(c, steps[s]);
  }
}

inline void launder_vgpr(int&) {}
inline void launder_lds(const __attribute__((address_space(3))) float*&) {}

// buffer_load_dword idxen offen through a buffer resource (gfx9 buffer addressing, the form pm_kernels.hip states and
// scripts/ubench/mubuf_addr.hip verified on gfx950): with swizzling
//   address = base + ((index / IS) * stride + (offset / 4) * 4) * IS + (index % IS) * 4,   element size 4,
// without: base + index * stride + offset.
typedef int v4i __attribute__((ext_vector_type(4)));
inline uint32_t llvm_struct_buffer_load_u32(v4i rsrc, int vindex, int voffset, int soffset, int /*aux*/) {
  const uint64_t base = (uint64_t)(uint32_t)rsrc[0] | ((uint64_t)((uint32_t)rsrc[1] & 0xffffu) << 32);
  const uint64_t stride = ((uint32_t)rsrc[1] >> 16) & 0x3fffu;
  const bool swizzle = ((uint32_t)rsrc[1] >> 31) != 0u;
  const uint64_t index = (uint32_t)vindex, offset = (uint64_t)(uint32_t)voffset + (uint32_t)soffset;
  uint64_t addr;
  if (swizzle) {
    const uint64_t IS = 8ull << (((uint32_t)rsrc[3] >> 21) & 3u);
    addr = base + ((index / IS) * stride + (offset / 4) * 4) * IS + (index % IS) * 4 + (offset % 4);
  } else {
    addr = base + index * stride + offset;
  }
  uint32_t v;
  std::memcpy(&v, (const void*)addr, 4);
  return v;
}

}  // namespace colmap_amd
