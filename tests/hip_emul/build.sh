#!/bin/sh
# Builds tests/hip_emul/libfusion_emul.so: colmap_amd/csrc/fusion.hip (unmodified) against the CPU stand-in headers.
# TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
printf 'extern "C" void pm_release_cached_memory(void) {}\n' > "$here/_stubs.cpp"
g++ -O1 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall \
    -Wno-unknown-pragmas -Wno-unused-function -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$here/_stubs.cpp" \
    -o "$here/libfusion_emul.so"
