#!/bin/sh
# Sanitizer audit of the kernels through the CPU stand-in (see asan_audit.py). ~5 minutes on a few cores.
#   sh asan_audit.sh                 AddressSanitizer (out-of-bounds / use-after-free accesses to global memory and LDS)
#   sh asan_audit.sh undefined       UndefinedBehaviorSanitizer (+ float-cast-overflow: shifts, signed overflow, misaligned
#                                    accesses, float -> int conversions out of range)
#   HIP_EMUL_AUDIT_ONLY="ba" sh asan_audit.sh [mode]      only the named components (fusion fusion_small ba pm)
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
cxx=${HIP_EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
mode=${1:-address}
if [ "$mode" = undefined ]; then
  rt=$(ls "$(dirname "$cxx")"/../lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
  san="-fsanitize=undefined,float-cast-overflow -fno-sanitize=vptr,function -shared-libsan"
else
  rt=$(ls "$(dirname "$cxx")"/../lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  san="-fsanitize=address -shared-libasan"
fi
out=${HIP_EMUL_ASAN_DIR:-$(mktemp -d)}
export HIP_EMUL_ASAN_DIR="$out"
printf 'extern "C" void pm_release_cached_memory(void) {}\n' > "$out/stubs.cpp"
ln -sf "$root/colmap_amd/csrc/pm_kernels.hip" "$here/pm/pm_kernels.hip"
F="-O1 -g $san -fno-omit-frame-pointer -std=c++17 -fPIC -shared -mavx2 -mfma -ffp-contract=off -Wno-unknown-attributes"
"$cxx" $F -fvisibility=hidden -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$out/stubs.cpp" -o "$out/libfusion_asan.so" &
p1=$!
"$cxx" $F -fvisibility=hidden -DFUSION_RECORD_BUF=1024 -DFUSION_STACK_LDS=8 -DFUSION_STACK_SPILL=8 -DFUSION_MEDIAN_STAGE=4 \
    -I "$here" -x c++ "$root/colmap_amd/csrc/fusion.hip" "$out/stubs.cpp" -o "$out/libfusion_small_asan.so" &
p2=$!
"$cxx" $F -fvisibility-inlines-hidden -Wl,-Bsymbolic -I "$here" -I "$root/colmap_amd/csrc" -x c++ "$root/colmap_amd/csrc/ba_kernels.hip" \
    "$root/colmap_amd/csrc/ba_schur_explicit.hip" "$out/stubs.cpp" -o "$out/libba_asan.so" &
p3=$!
"$cxx" $F -fvisibility-inlines-hidden -Wl,-Bsymbolic -I "$here" -I "$root/colmap_amd/csrc" -x c++ "$here/pm/pm_kernels.hip" \
    "$root/colmap_amd/csrc/pm_api.cpp" "$here/pm/pm_stubs.cpp" -o "$out/libpm_asan.so" &
p4=$!
wait $p1; wait $p2; wait $p3; wait $p4
rc=0
parts=${HIP_EMUL_AUDIT_ONLY:-fusion fusion_small ba pm}
for w in $parts; do
  LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=0 \
      python "$here/asan_audit.py" $w > "$out/audit_$w.log" 2>&1 &
done
wait
for w in $parts; do
  if grep -q "runtime error" "$out/audit_$w.log"; then echo "$w: undefined behaviour reported -- see $out/audit_$w.log"; grep "runtime error" "$out/audit_$w.log" | sort | uniq -c | head -20; rc=1;
  elif grep -q "AUDIT DONE $w" "$out/audit_$w.log"; then echo "$w: clean ($(grep -c ' ok' "$out/audit_$w.log") comparisons)";
  else echo "$w: FAILED -- see $out/audit_$w.log"; tail -20 "$out/audit_$w.log"; rc=1; fi
done
exit $rc
