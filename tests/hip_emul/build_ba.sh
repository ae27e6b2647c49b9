#!/bin/sh
# Builds tests/hip_emul/libba_emul.so: colmap_amd/csrc/ba_kernels.hip + ba_schur_explicit.hip (unmodified) against the
# CPU stand-in headers of this directory, with ROCm's clang++ as the HOST compiler (the sources use clang vector
# types). TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
printf 'extern "C" void pm_release_cached_memory(void) {}\n' > "$here/_stubs.cpp"
cxx=${HIP_EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
"$cxx" -O2 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility-inlines-hidden -Wl,-Bsymbolic \
    -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unknown-attributes -I "$here" -I "$root/colmap_amd/csrc" \
    -x c++ "$root/colmap_amd/csrc/ba_kernels.hip" "$root/colmap_amd/csrc/ba_schur_explicit.hip" "$here/_stubs.cpp" \
    -o "$here/libba_emul.so"
