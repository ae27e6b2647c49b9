#!/bin/sh
# Builds colmap_amd/csrc/fusion.hip (unmodified) against the CPU stand-in headers of this directory (ROCm's clang++ as
# the HOST compiler, like build_ba.sh / build_pm.sh):
#   libfusion_emul.so        the product's capacities
#   libfusion_emul_small.so  tiny capacities (record buffer, LDS stack, stack spill, median staging): the overflow
#                            paths run on inputs of a few thousand pixels
# TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
printf 'extern "C" void pm_release_cached_memory(void) {}\n' > "$here/_stubs.cpp"
cxx=${HIP_EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
flags="-O2 -g -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unknown-attributes"
"$cxx" $flags -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$here/_stubs.cpp" -o "$here/libfusion_emul.so" &
p1=$!
"$cxx" $flags -DFUSION_RECORD_BUF=1024 -DFUSION_STACK_LDS=8 -DFUSION_STACK_SPILL=8 -DFUSION_MEDIAN_STAGE=4 \
    -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$here/_stubs.cpp" -o "$here/libfusion_emul_small.so" &
p2=$!
wait $p1
wait $p2
