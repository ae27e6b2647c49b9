// CPU stand-in twin of colmap_amd/csrc/gfx950/ba_gfx950_asm.h (TEST INFRASTRUCTURE ONLY): the same names in plain C++.
#pragma once

namespace ba_explicit {

inline void order_after(int&, double) {}   // a scheduling fence: nothing to do on the CPU

}  // namespace ba_explicit
