#!/bin/bash
# What do the gathers of the sweep kernel cost? Diagnostic builds of the library (PM_DIAG_GATHER in
# pm_kernels.hip::ncc_front / tap_geom_init; PM_FP_TILED=0 = row-major packed images) timed against the product
# build with scripts/pm_probe.py. The builds change only WHICH packed-image entry a gather reads; results are
# garbage, the instruction stream is the product's.
#   1  every gather reads its image's first entry (address path and caches idle: the instruction-issue time)
#   2  entry indices wrapped into 8 KB per image (same lines per instruction, no L2 misses)
#   4  (row-major) tap rows rounded down to multiples of four: a quarter of the cache lines, same pages
#   5  entry indices wrapped into 2 MB per image: L2 misses as in the product, few pages, infinity-cache resident
#   PM_FP_FORMAT=16 / 8  not diagnostics but experimental product formats (2-byte entries / 1-byte texels): the same
#      results bit for bit, 354 / 177 MB of packed sources instead of 708 MB for the benchmark's batch
#   3  one window row per gather instruction -- MISLEADING: the garbage sums drive the hypotheses out of the images,
#      the clamped taps then all read the zero ring (fast for the wrong reason; see ROUND_NOTES.md round 3)
# Build here (no GPU needed):   bash scripts/profile_pm_gather_diag.sh build
# Run on the GPU box:           bash scripts/profile_pm_gather_diag.sh run TAG      (results: profiles/r03_pm_gather_diag.log)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/scratch_diag
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function -munsafe-fp-atomics"
SRC="pm_api.cpp pm_kernels.hip ba_kernels.hip ba_schur_explicit.hip fusion.hip"
one() { (cd $ROOT/colmap_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $2 $SRC -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/$1); }
if [ "$1" = build ]; then
  mkdir -p $D; rm -f $D/*.so
  one libdiag1.so "-DPM_DIAG_GATHER=1" &
  one libdiag2.so "-DPM_DIAG_GATHER=2" &
  one libdiag5.so "-DPM_DIAG_GATHER=5" &
  one librowmajor.so "-DPM_FP_TILED=0" &
  one libdiag4.so "-DPM_FP_TILED=0 -DPM_DIAG_GATHER=4" &
  one libfp16.so "-DPM_FP_FORMAT=16" &   # experimental: 2-byte entries (vertical texel pairs), unaligned dword gathers
  one libfp8.so "-DPM_FP_FORMAT=8" &     # experimental: 1-byte texels, two unaligned 2-byte loads per tap
  wait
  ls -la $D
else
  TAG=${2:?tag}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
  probe() { timeout 120 python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc 16 --sweeps 4 "$@" 2>&1 | grep -E "sweep kernel|rror"; }
  echo "product build (tiled packed images):              $(probe)" | tee -a $OUT/pm_gather_diag.log
  echo "product build, COLMAP_AMD_PM_QUAD=0:              $(COLMAP_AMD_PM_QUAD=0 probe)" | tee -a $OUT/pm_gather_diag.log
  echo "row-major packed images:                          $(PM_PROBE_LIB=$D/librowmajor.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_DIAG_GATHER=1 (one entry per image):           $(PM_PROBE_LIB=$D/libdiag1.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_DIAG_GATHER=2 (8 KB window per image):         $(PM_PROBE_LIB=$D/libdiag2.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_DIAG_GATHER=5 (2 MB window per image):         $(PM_PROBE_LIB=$D/libdiag5.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "row-major, PM_DIAG_GATHER=4 (rows in fours):      $(PM_PROBE_LIB=$D/libdiag4.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "product build, geometric consistency:             $(probe --geom 1)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_FP_FORMAT=16 (2-byte entries):                 $(PM_PROBE_LIB=$D/libfp16.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_FP_FORMAT=16, geometric consistency:           $(PM_PROBE_LIB=$D/libfp16.so probe --geom 1)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_FP_FORMAT=8 (1-byte texels):                   $(PM_PROBE_LIB=$D/libfp8.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "PM_FP_FORMAT=8, geometric consistency:            $(PM_PROBE_LIB=$D/libfp8.so probe --geom 1)" | tee -a $OUT/pm_gather_diag.log
  # the experimental formats must reproduce the oracle bit for bit like the product build
  for f in 16 8; do
    (cd $ROOT && COLMAP_AMD_TEST_LIB=$D/libfp$f.so timeout 600 python -m pytest tests/test_pm_gpu.py -m gpu -q -x 2>&1 | tail -3) | tee -a $OUT/pm_gather_diag.log
  done
fi
