// gfx950 inline-assembly helpers of the bundle-adjustment kernels (the CPU stand-in of tests/hip_emul has a twin of this
// file with the same names in plain C++).
#pragma once

namespace ba_explicit {

// Scheduling fence with a true dependence: `tok` (an address offset that is always 0) cannot be known before `dep` is
// computed, so the loads addressed through it are issued after the arithmetic that produces `dep` -- the only thing
// that keeps the compiler from hoisting every LDS read of an unrolled loop to its top and spilling
// (__builtin_amdgcn_sched_barrier and memory clobbers do not hold LDS reads back: ROUND_NOTES round 4).
__device__ __forceinline__ void order_after(int& tok, double dep) { asm volatile("" : "+v"(tok) : "v"(dep)); }

}  // namespace ba_explicit
