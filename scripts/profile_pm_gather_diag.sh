#!/bin/bash
# What do the gathers of the sweep kernel cost? Two diagnostic builds of the library (PM_DIAG_GATHER in
# pm_kernels.hip::ncc_front) timed against the product build with scripts/pm_probe.py:
#   1 = every gather reads its image's first entry (address path and caches idle: the instruction-issue time)
#   2 = entry indices wrapped into 8 KB per image (same lines per instruction, no L2 misses)
# Build here (no GPU needed):   bash scripts/profile_pm_gather_diag.sh build
# Run on the GPU box:           bash scripts/profile_pm_gather_diag.sh run TAG
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/scratch_diag
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function -munsafe-fp-atomics"
SRC="pm_api.cpp pm_kernels.hip ba_kernels.hip ba_schur_explicit.hip fusion.hip"
if [ "$1" = build ]; then
  mkdir -p $D
  for v in 1 2; do
    (cd $ROOT/colmap_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DPM_DIAG_GATHER=$v $SRC -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/libdiag$v.so) &
  done
  # row-major packed images (PM_FP_TILED=0): the product arithmetic, and with whole-row gather instructions (3)
  (cd $ROOT/colmap_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DPM_FP_TILED=0 $SRC -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/librowmajor.so) &
  (cd $ROOT/colmap_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DPM_FP_TILED=0 -DPM_DIAG_GATHER=3 $SRC -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/libdiag3.so) &
  (cd $ROOT/colmap_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DPM_DIAG_GATHER=3 $SRC -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/libdiag3t.so) &
  wait
  ls -la $D
elif [ "$1" = run2 ]; then
  TAG=${2:?tag}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
  probe() { timeout 600 python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc 16 --sweeps 4 "$@" 2>&1 | grep -E "sweep kernel|rror"; }
  echo "product build (tiled packed images):              $(probe)" | tee -a $OUT/pm_gather_diag.log
  echo "row-major packed images:                          $(PM_PROBE_LIB=$D/librowmajor.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "row-major, PM_DIAG_GATHER=3 (whole-row gathers):  $(PM_PROBE_LIB=$D/libdiag3.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "tiled, PM_DIAG_GATHER=3 (whole-row gathers):      $(PM_PROBE_LIB=$D/libdiag3t.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "row-major, PM_DIAG_GATHER=4 (rows in fours):      $(PM_PROBE_LIB=$D/libdiag4.so probe)" | tee -a $OUT/pm_gather_diag.log
  echo "tiled, PM_DIAG_GATHER=2 (8 KB window per image):  $(PM_PROBE_LIB=$D/libdiag2.so probe)" | tee -a $OUT/pm_gather_diag.log
else
  TAG=${2:?tag}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
  probe() { timeout 600 python $ROOT/scripts/pm_probe.py --w 2560 --h 1920 --views 21 --arc 72 --nofilter 1 --conc 16 --sweeps 4 "$@" 2>&1 | grep -E "sweep kernel|rror"; }
  for q in 0 1; do
    echo "product build, COLMAP_AMD_PM_QUAD=$q:   $(COLMAP_AMD_PM_QUAD=$q COLMAP_AMD_PM_COLS=2 probe)" | tee -a $OUT/pm_gather_diag.log
    for v in 1 2; do
      echo "PM_DIAG_GATHER=$v, COLMAP_AMD_PM_QUAD=$q: $(COLMAP_AMD_PM_QUAD=$q COLMAP_AMD_PM_COLS=2 PM_PROBE_LIB=$D/libdiag$v.so probe)" | tee -a $OUT/pm_gather_diag.log
    done
    echo "product build, geometric consistency, COLMAP_AMD_PM_QUAD=$q: $(COLMAP_AMD_PM_QUAD=$q probe --geom 1)" | tee -a $OUT/pm_gather_diag.log
  done
fi
