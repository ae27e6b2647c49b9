// pm_gfx950_asm.h -- the places where the PatchMatch kernels (pm_kernels.hip) speak gfx950 assembly directly:
// instruction selections the compiler does not make on its own. They sit in a header of their own so that
// tests/hip_emul -- the CPU stand-in that runs the unmodified kernels against the oracle -- can restate these five
// helpers in C++ (it compiles pm_kernels.hip through a link in a directory that holds its own gfx950/pm_gfx950_asm.h,
// the same way it stands in for <hip/hip_runtime.h>). There is one implementation in the product: this one.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace colmap_amd {

// Byte k of a packed footprint entry as float. Inline asm keeps the four conversions as four
// full-rate v_cvt_f32_ubyteK: left to itself the compiler rewrites (float)b - (float)a into an
// integer subtract + v_cvt_f32_i32 pair per difference (8 instead of 5 instructions per tap).
#define PM_UBYTE(K)                                                              \
  __device__ __forceinline__ float ubyte##K(uint32_t x) {                        \
    float f;                                                                     \
    asm("v_cvt_f32_ubyte" #K " %0, %1" : "=v"(f) : "v"(x));                      \
    return f;                                                                    \
  }
PM_UBYTE(0)
PM_UBYTE(1)
PM_UBYTE(2)
PM_UBYTE(3)
#undef PM_UBYTE

// The three 16-lane tree sums of an evaluation in one instruction block: 12 v_add_f32_dpp, each
// value's next step separated from its previous one by the other two values' steps (the DPP
// read-after-VALU-write hazard needs two wait states; inline asm is not seen by the compiler's
// hazard recogniser, hence the leading s_nop). Same tree as reduce16 (pm_kernels.hip): row_mirror,
// row_half_mirror, quad reverse, quad swap leave in every lane
//   ((q0+q7)+(q3+q4)) + ((q1+q6)+(q2+q5)),  q_l = p_l + p_(15-l).
__device__ __forceinline__ void reduce16x3(float& a, float& b, float& c) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b), "+v"(c));
}

// A value laundered through an empty asm: the compiler may not hoist what is derived from it out of the loop the
// statement sits in (the sweep kernels re-derive their per-lane indices per row instead of holding them in VGPRs
// across the NCC loop, which needs the registers).
__device__ __forceinline__ void launder_vgpr(int& v) { asm volatile("" : "+v"(v)); }
// The same for an LDS address: what follows addresses relative to this one register (small immediate offsets)
// instead of re-deriving base + constant per access.
__device__ __forceinline__ void launder_lds(const __attribute__((address_space(3))) float*& v) { asm volatile("" : "+v"(v)); }

// MUBUF load with index and offset (buffer_load_dword ... idxen offen) through a buffer resource: no builtin for
// struct buffer loads in this clang, the intrinsic is reachable by name.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ uint32_t llvm_struct_buffer_load_u32(v4i rsrc, int vindex, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.struct.buffer.load.i32");

}  // namespace colmap_amd
