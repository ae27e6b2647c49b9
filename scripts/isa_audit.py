#!/usr/bin/env python3
"""Static audit of the gfx950 ISA of the product's kernels -- no GPU needed (hipcc cross-compiles).

  python scripts/isa_audit.py                      # all translation units: per-kernel table + suspicious loops
  python scripts/isa_audit.py ba_schur_explicit    # one translation unit (substring of the file name)
  python scripts/isa_audit.py ba_kernels gram tail # ... and only kernels whose mangled name contains one of the words

Per kernel: instruction count, VGPRs (next_free_vgpr; > 256 means AGPR copies), scratch bytes, LDS bytes, barriers,
matrix-core and atomic instructions. Then, for every backward branch (a loop) with 1-4 global / buffer loads and at least one
`s_waitcnt vmcnt(0)` in fewer than 120 instructions: a candidate for "one load, one full wait per trip" -- the pattern a
rolled staging loop `for (e = tid; e < N; e += blockDim) lds[e] = global[e]` or a per-element read-modify-write compiles to
(N / blockDim serial memory round trips; fix: compile-time trip count, values in registers, all loads first). Loops of
genuinely dependent loads (pointer chasing, walks) show up too: read the source before changing anything.
The product's flags are taken from colmap_amd/build.py's command line (-O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "colmap_amd", "csrc")
UNITS = ["pm_kernels.hip", "ba_kernels.hip", "ba_schur_explicit.hip", "fusion.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics"]


def compile_unit(unit, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, "-c", os.path.join(CSRC, unit), "-o",
                           os.path.join(out, unit + ".o"), "-save-temps=obj"], cwd=out, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    stem = unit.rsplit(".", 1)[0]
    return os.path.join(out, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(asm):
    for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        meta = asm[m.start():m.end() + 3000]
        get = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", meta) or [None, "?"])[1]
        lines = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
        yield m.group(1), lines, get


def audit(path, words):
    asm = open(path).read()
    for name, lines, get in kernels(asm):
        if "rocprim" in name or "hipcub" in name or (words and not any(w in name for w in words)):
            continue
        ins = [l for l in lines if not l.startswith(".")]
        cnt = lambda p: sum(1 for l in ins if l.startswith(p))
        print(f"{name[:96]}\n    {len(ins)} instrs, vgpr {get('next_free_vgpr')}, scratch {get('private_segment_fixed_size')} B, "
              f"lds {get('group_segment_fixed_size')} B, barriers {cnt('s_barrier')}, mfma {cnt('v_mfma')}, "
              f"atomics {cnt('global_atomic') + cnt('flat_atomic')}, scratch ops {cnt('scratch_')}")
        labels = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB")}
        found = []
        for i, l in enumerate(lines):
            mm = re.match(r"s_cbranch_\w+ (\.LBB\S+)", l) or re.match(r"s_branch (\.LBB\S+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                body = lines[labels[mm.group(1)]:i + 1]
                loads = sum(1 for x in body if x.startswith(("global_load", "buffer_load", "flat_load")))
                waits = sum(1 for x in body if re.match(r"s_waitcnt.*vmcnt\(0\)", x))
                if 1 <= loads <= 4 and waits >= 1 and len(body) < 120:
                    found.append((len(body), loads, waits))
        for n, loads, waits in sorted(set(found))[:3]:
            print(f"    ? loop of {n} instructions: {loads} load(s), {waits} full wait(s) per trip")


def main():
    args = sys.argv[1:]
    units = [u for u in UNITS if not args or args[0] in u] or UNITS
    words = args[1:] if args and any(args[0] in u for u in UNITS) else (args if not any(a in u for a in args[:1] for u in UNITS) else [])
    with tempfile.TemporaryDirectory() as out:
        for u in units:
            print(f"==== {u}")
            audit(compile_unit(u, out), words)


if __name__ == "__main__":
    main()
