"""`python -m colmap_amd <command> [options]`: the commands of the `colmap` executable that sit on the
two MI355X paths (exe/colmap.cc command table)."""
import sys

COMMANDS = {
    "patch_match_stereo": ("colmap_amd.patch_match_stereo", "dense stereo on an undistorted workspace (exe/mvs.cc:228-279)"),
    "stereo_fusion": ("colmap_amd.fusion", "fuse depth / normal maps into a point cloud (exe/mvs.cc:299-386)"),
    "bundle_adjuster": ("colmap_amd.bundle_adjuster", "global bundle adjustment of a sparse model (exe/sfm.cc:175-206)"),
}


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help", "help"):
        print("usage: python -m colmap_amd <command> [options]\n\ncommands:")
        for name, (_, what) in COMMANDS.items():
            print(f"  {name:20s} {what}")
        return 0
    if argv[0] not in COMMANDS:
        print(f"E Command `{argv[0]}` not recognized. To list the available commands, run `python -m colmap_amd help`.",
              file=sys.stderr)
        return 1
    import importlib
    return importlib.import_module(COMMANDS[argv[0]][0]).main(argv[1:])


if __name__ == "__main__":
    sys.exit(main())
