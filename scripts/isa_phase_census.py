"""Static instruction census of the sweep kernel per row phase. MEASUREMENT INFRASTRUCTURE (no product code).

    python scripts/isa_phase_census.py [--kernel ILb0ELb0ELb0ELb1E] [--dump PHASE]

Compiles colmap_amd/csrc/pm_kernels.hip for gfx950 with -DPM_ISA_MARKERS (asm comment markers at the phase boundaries of
sweep_wave_body / run_tasks_wave), cuts the chosen pm_sweep_quad_kernel instantiation at the markers in layout order and
counts VALU / LDS / VMEM / SALU instructions per region. Static counts: a region that holds a loop counts its body once
(the backward branches of a region are listed so that the reader can weight them); code the compiler moved across a marker
is attributed to where it landed. The dynamic picture is profiles/r06_pm_phase_profile.json (shader-clock deltas on the GPU)."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "colmap_amd", "csrc", "pm_kernels.hip")


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="pm_sweep_quad_kernelILb0ELb0ELb0ELb1E")
    ap.add_argument("--asm", default="/tmp/pm_mark.s")
    ap.add_argument("--reuse", action="store_true")
    ap.add_argument("--dump", default="")
    ap.add_argument("--define", action="append", default=[])
    a = ap.parse_args()
    if not a.reuse:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DPM_ISA_MARKERS",
               "-S", "--cuda-device-only", "-o", a.asm, SRC] + ["-D" + d for d in a.define]
        subprocess.check_call(cmd, cwd="/tmp", stderr=subprocess.DEVNULL)
    lines = open(a.asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(a.kernel) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\ts_endpgm") or ".Lfunc_end" in lines[i])
    region = "entry"
    counts = collections.OrderedDict()
    order = []
    labels = {}
    cur_label_region = {}
    branches = collections.defaultdict(list)
    seq = 0
    for i in range(start + 1, end):
        l = lines[i]
        m = re.search(r"; PMARK (\w+)", l)
        if m:
            seq += 1
            region = f"{seq:02d}:{m.group(1)}"
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            labels[l.split(":")[0]] = (i, region)
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = counts.setdefault(region, collections.Counter())
        c[classify(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[1]
            if tgt in labels:  # backward branch = loop
                branches[region].append(f"loop->{labels[tgt][1]}@{i - labels[tgt][0]}l")
        if a.dump and region.endswith(":" + a.dump):
            print(l)
    print(f"{'region':<16}{'valu':>6}{'lds':>6}{'vmem':>6}{'salu':>6}{'wait':>6}{'br':>5}  loops")
    tot = collections.Counter()
    for r, c in counts.items():
        tot.update(c)
        print(f"{r:<16}{c['valu']:>6}{c['lds']:>6}{c['vmem']:>6}{c['salu']:>6}{c['wait']:>6}{c['branch']:>5}  {' '.join(branches[r])}")
    print(f"{'total':<16}{tot['valu']:>6}{tot['lds']:>6}{tot['vmem']:>6}{tot['salu']:>6}{tot['wait']:>6}{tot['branch']:>5}")
    # register / LDS footprint
    for i in range(end, min(end + 80, len(lines))):
        if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs)", lines[i]):
            print(lines[i].strip())


if __name__ == "__main__":
    main()
