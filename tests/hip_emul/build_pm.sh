#!/bin/sh
# Builds tests/hip_emul/libpm_emul.so: colmap_amd/csrc/pm_kernels.hip + pm_api.cpp (unmodified) against the CPU stand-in
# headers of this directory. pm_kernels.hip includes "gfx950/pm_gfx950_asm.h" relative to its own directory (the five
# inline-assembly helpers of the product): it is compiled here through a LINK in pm/, next to pm/gfx950/pm_gfx950_asm.h,
# the C++ restatement of those helpers; every other quoted include falls through to colmap_amd/csrc.
# TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
cxx=${HIP_EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
ln -sf "$root/colmap_amd/csrc/pm_kernels.hip" "$here/pm/pm_kernels.hip"
"$cxx" -O2 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility-inlines-hidden -Wl,-Bsymbolic \
    -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unknown-attributes -I "$here" -I "$root/colmap_amd/csrc" \
    -x c++ "$here/pm/pm_kernels.hip" "$root/colmap_amd/csrc/pm_api.cpp" "$here/pm/pm_stubs.cpp" \
    -o "$here/libpm_emul.so"
