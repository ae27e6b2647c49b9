"""pycolmap-style pipeline functions on the MI355X paths: `from colmap_amd.pipeline import
patch_match_stereo, stereo_fusion, bundle_adjustment`."""
# ---------------------------------------------------------------------------------------------
# pycolmap-style pipeline functions (reference pycolmap/pipeline/mvs.cc:119-127,182-193,
# pycolmap/pipeline/sfm.cc:153-161,258-263): same names, argument order and defaults. Imports are
# deferred so that importing this module stays light.
# ---------------------------------------------------------------------------------------------

def patch_match_stereo(workspace_path, workspace_format="COLMAP", pmvs_option_name="option-all", options=None,
                       config_path=""):
    """pycolmap.patch_match_stereo: runs PatchMatch stereo on an undistorted workspace (needs an MI355X)."""
    from . import mvs
    opt = options or mvs.PatchMatchOptions()
    if opt.gpu_index == "-1":
        opt.gpu_index = "0"
    ctl = mvs.PatchMatchController.FromWorkspace(opt, str(workspace_path), workspace_format.lower(), pmvs_option_name,
                                                 str(config_path))
    ctl.Run()
    return ctl


def stereo_fusion(output_path, workspace_path, workspace_format="COLMAP", pmvs_option_name="option-all",
                  input_type="geometric", options=None, output_type="bin"):
    """pycolmap.stereo_fusion: fuses the depth / normal maps of a workspace and writes a model (bin / txt)
    or a PLY file + visibility; returns the fused points."""
    from . import fusion
    argv = ["--workspace_path", str(workspace_path), "--workspace_format", workspace_format,
            "--pmvs_option_name", pmvs_option_name, "--input_type", input_type, "--output_type", output_type,
            "--output_path", str(output_path)]
    opt = options or fusion.StereoFusionOptions()
    for name in ("mask_path", "num_threads", "max_image_size", "min_num_pixels", "max_num_pixels", "max_traversal_depth",
                 "max_reproj_error", "max_depth_error", "max_normal_error", "check_num_images", "use_cache", "cache_size"):
        argv += [f"--StereoFusion.{name}", str(getattr(opt, name))]
    if fusion.main(argv) != 0:
        raise RuntimeError("stereo_fusion failed")
    if output_type.lower() == "ply":
        return fusion.read_binary_ply_points(str(output_path))
    return None


def bundle_adjustment(reconstruction, options=None):
    """pycolmap.bundle_adjustment: BundleAdjustmentController on a colmap_amd.scene.Reconstruction, in place."""
    from . import bundle_adjuster, estimators
    ctl = bundle_adjuster.BundleAdjustmentController(options or estimators.BundleAdjustmentOptions(), reconstruction)
    ctl.Run()
    return ctl.summary
