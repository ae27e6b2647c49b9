"""Builds the gfx950 shared library in-tree (colmap_amd/lib/libcolmap_amd.so).

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess

_ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_ROOT, "csrc")
LIB_DIR = os.path.join(_ROOT, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcolmap_amd.so")

SOURCES = ["pm_api.cpp", "pm_kernels.hip", "ba_kernels.hip", "ba_schur_explicit.hip", "fusion.hip"]

# -ffp-contract=off: fused multiply-adds only where the source says fmaf(); the
# arithmetic is specified operation by operation (oracle/pm_oracle.c header).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function", "-munsafe-fp-atomics"]


def _sources():
    paths = [os.path.join(CSRC, s) for s in SOURCES]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:  # a renamed / deleted source must not silently produce a partial library
        raise FileNotFoundError(f"colmap_amd.build: missing sources {missing}")
    return paths


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = _sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(CSRC, "gfx950", f) for f in os.listdir(os.path.join(CSRC, "gfx950"))]
    inc = os.path.join(os.path.dirname(_ROOT), "include")
    deps += [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, dev: bool = False, out: str = "") -> str:
    """dev=True: a profiling build -- development switches also readable from the environment
    (-DCOLMAP_AMD_ENV_SWITCHES) and the garbage-producing PatchMatch diagnostics compiled in (-DCOLMAP_AMD_DIAG_BUILD);
    written to `out` (default lib/libcolmap_amd_dev.so), never the library the package loads."""
    if dev:
        out = out or os.path.join(LIB_DIR, "libcolmap_amd_dev.so")
    elif not force and not needs_build():
        return LIB_PATH
    out = out or LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-DCOLMAP_AMD_ENV_SWITCHES", "-DCOLMAP_AMD_DIAG_BUILD"] if dev else []
    cmd = [hipcc] + HIPCC_FLAGS + extra + _sources() + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, dev="--dev" in sys.argv))
