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


def _header_deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(CSRC, "gfx950", f) for f in os.listdir(os.path.join(CSRC, "gfx950"))]
    inc = os.path.join(os.path.dirname(_ROOT), "include")
    for d, _, fs in os.walk(inc):
        deps += [os.path.join(d, f) for f in fs]
    return deps


def _includes_of(src, seen=None):
    """The in-tree headers `src` includes, transitively (quoted includes resolved against the file's directory, csrc/
    and include/): an object is rebuilt only when one of THESE is newer, not when any header of the tree is."""
    import re
    seen = set() if seen is None else seen
    inc_root = os.path.join(os.path.dirname(_ROOT), "include")
    try:
        text = open(src, errors="replace").read()
    except OSError:
        return seen
    for name in re.findall(r'^\s*#\s*include\s*["<]([^">]+)[">]', text, re.M):
        for base in (os.path.dirname(src), CSRC, inc_root):
            path = os.path.normpath(os.path.join(base, name))
            if os.path.isfile(path) and path.startswith(os.path.dirname(_ROOT)):
                if path not in seen:
                    seen.add(path)
                    _includes_of(path, seen)
                break
    return seen


def build(force: bool = False, verbose: bool = False, dev: bool = False, out: str = "", incremental: bool = False) -> str:
    """Compiles every source to an object of its own, in parallel, and links them. force: recompile everything;
    incremental: recompile only the objects older than their source or any header (the default call does nothing while
    the library is newer than all of them). dev=True: a profiling build -- development switches also readable from the
    environment (-DCOLMAP_AMD_ENV_SWITCHES) and the garbage-producing PatchMatch diagnostics compiled in
    (-DCOLMAP_AMD_DIAG_BUILD); written to `out` (default lib/libcolmap_amd_dev.so), never the library the package loads."""
    if dev:
        out = out or os.path.join(LIB_DIR, "libcolmap_amd_dev.so")
    elif not force and not incremental and not needs_build():
        return LIB_PATH
    out = out or LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj_dev" if dev else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-DCOLMAP_AMD_ENV_SWITCHES", "-DCOLMAP_AMD_DIAG_BUILD"] if dev else []
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        newest = max([os.path.getmtime(src)] + [os.path.getmtime(d) for d in _includes_of(src)])
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, j in jobs if j.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, f"hipcc -c {failed}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    # default: recompile what is older than its source / headers and link; --all: recompile every object
    print(build(force="--all" in sys.argv, incremental=True, verbose=True, dev="--dev" in sys.argv))
