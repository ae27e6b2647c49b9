#!/bin/sh
# Builds colmap_amd/csrc/fusion.hip (unmodified) against the CPU stand-in headers of this directory:
#   libfusion_emul.so        the product's capacities
#   libfusion_emul_small.so  tiny capacities (record buffer, LDS stack, stack spill, median staging): the overflow
#                            paths run on inputs of a few thousand pixels
# TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
printf 'extern "C" void pm_release_cached_memory(void) {}\n' > "$here/_stubs.cpp"
flags="-O2 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unknown-pragmas -Wno-unused-function"
g++ $flags -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$here/_stubs.cpp" -o "$here/libfusion_emul.so" &
g++ $flags -DFUSION_RECORD_BUF=1024 -DFUSION_STACK_LDS=8 -DFUSION_STACK_SPILL=8 -DFUSION_MEDIAN_STAGE=4 \
    -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$here/_stubs.cpp" -o "$here/libfusion_emul_small.so" &
wait
