"""`colmap patch_match_stereo` on MI355X (reference exe/mvs.cc:228-279, option names
controllers/option_manager.cc:932-980):

    python -m colmap_amd.patch_match_stereo --workspace_path DENSE \\
        [--workspace_format COLMAP] [--config_path CFG] [--PatchMatchStereo.geom_consistency 1] ...

reads the undistorted workspace (`images/`, `sparse/`, `stereo/patch-match.cfg`), solves every
problem (photometric pass, then the geometric pass when geom_consistency is on) and writes
`stereo/{depth_maps,normal_maps}/<image>.<photometric|geometric>.bin` (+ consistency graphs).
Under `torchrun` the problems are sharded over the ranks (one GPU each).
"""
from __future__ import annotations

import argparse
import dataclasses
import os
import sys

from . import mvs


def _parse_bool(v: str) -> bool:
    if v.lower() in ("1", "true", "yes", "on"):
        return True
    if v.lower() in ("0", "false", "no", "off"):
        return False
    raise argparse.ArgumentTypeError(f"not a boolean: {v}")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="patch_match_stereo", description=__doc__.split("\n\n")[0])
    ap.add_argument("--workspace_path", required=True, help="Path to the folder containing the undistorted images")
    ap.add_argument("--workspace_format", default="COLMAP", help="{COLMAP, PMVS}")
    ap.add_argument("--pmvs_option_name", default="option-all")
    ap.add_argument("--config_path", default="")
    ap.add_argument("--batch_size", type=int, default=8,
                    help="(MI355X) same-shaped problems solved by shared kernel launches")
    defaults = mvs.PatchMatchOptions()
    for f in dataclasses.fields(mvs.PatchMatchOptions):
        if f.name in ("max_sweeps", "columns_per_group", "threads_per_group"):
            continue
        d = getattr(defaults, f.name)
        t = _parse_bool if isinstance(d, bool) else type(d)
        ap.add_argument(f"--PatchMatchStereo.{f.name}", dest=f"pm_{f.name}", type=t, default=d)
    return ap


def options_from_args(a) -> mvs.PatchMatchOptions:
    kw = {k[3:]: v for k, v in vars(a).items() if k.startswith("pm_")}
    return mvs.PatchMatchOptions(**kw)


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    fmt = a.workspace_format.lower()
    if fmt not in ("colmap", "pmvs"):
        raise SystemExit(f"Invalid `workspace_format` {fmt} - supported values are 'COLMAP' or 'PMVS'.")
    opt = options_from_args(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        opt.gpu_index = str(local_rank)
    elif opt.gpu_index == "-1":
        opt.gpu_index = "0"  # ReadGpuIndices (:361-375): all devices; one process drives one GPU here
    ctl = mvs.PatchMatchController.FromWorkspace(opt, a.workspace_path, fmt, a.pmvs_option_name, a.config_path,
                                                 batch_size=a.batch_size, rank=rank, world_size=world)
    for w in getattr(ctl, "warnings_", []):
        print("W", w, file=sys.stderr)
    if rank == 0:
        print(f"Configuration has {len(ctl.problems_)} problems...")
    ctl.Run()
    if rank == 0:
        print("Elapsed: " + ", ".join(f"{k} {v:.1f}s" for k, v in ctl.timings.items() if isinstance(v, float)))
        if "map_exchange" in ctl.timings:
            print("Photometric maps for the geometric pass: " + str(ctl.timings["map_exchange"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
