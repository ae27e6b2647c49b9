#!/bin/sh
# Builds tests/hip_emul/libpm_emul.so: colmap_amd/csrc/pm_kernels.hip + pm_api.cpp (unmodified) against the CPU stand-in
# headers of this directory, with pm/pm_gfx950_asm.h standing in for the five inline-assembly helpers of
# colmap_amd/csrc/gfx950/pm_gfx950_asm.h. TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
cxx=${HIP_EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
"$cxx" -O2 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility-inlines-hidden -Wl,-Bsymbolic \
    -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unknown-attributes -I "$here/pm" -I "$here" \
    -x c++ "$root/colmap_amd/csrc/pm_kernels.hip" "$root/colmap_amd/csrc/pm_api.cpp" "$here/pm/pm_stubs.cpp" \
    -o "$here/libpm_emul.so"
