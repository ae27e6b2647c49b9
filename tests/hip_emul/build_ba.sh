#!/bin/sh
# Builds tests/hip_emul/libba_emul.so: colmap_amd/csrc/ba_kernels.hip + ba_schur_explicit.hip (unmodified) against the
# CPU stand-in headers of this directory, with ROCm's clang++ as the HOST compiler (the sources use clang vector
# types). TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
printf 'extern "C" void pm_release_cached_memory(void) {}\n' > "$here/_stubs.cpp"
cxx=${HIP_EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
# ba_schur_explicit.hip includes "gfx950/ba_gfx950_asm.h" relative to its own directory (inline-assembly helpers of the
# product): compiled here through a LINK in ba/, next to ba/gfx950/ba_gfx950_asm.h, the C++ restatement of those helpers;
# every other quoted include falls through to colmap_amd/csrc (as build_pm.sh does for pm_kernels.hip)
ln -sf "$root/colmap_amd/csrc/ba_schur_explicit.hip" "$here/ba/ba_schur_explicit.hip"
"$cxx" -O2 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility-inlines-hidden -Wl,-Bsymbolic \
    -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unknown-attributes -I "$here" -I "$root/colmap_amd/csrc" \
    -x c++ "$root/colmap_amd/csrc/ba_kernels.hip" "$here/ba/ba_schur_explicit.hip" "$here/_stubs.cpp" \
    -o "$here/libba_emul.so"
