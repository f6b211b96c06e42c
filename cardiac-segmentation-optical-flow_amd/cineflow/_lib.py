"""ctypes binding of libcineflow_hip.so (C ABI declared in include/cineflow.h).

The library is mandatory: there is no CPU fallback anywhere in this package.
``lib()`` raises CineflowLibraryError with build instructions when the shared
object is missing, and every wrapper in ops.py turns a non-zero return code
into CineflowError(cf_last_error()).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CINEFLOW_LIB") or os.path.join(_HERE, "libcineflow_hip.so")   # CINEFLOW_LIB: A/B builds of the same ABI


class CineflowLibraryError(RuntimeError):
    pass


class CineflowError(RuntimeError):
    pass


P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_long
F = ctypes.c_float
DBL = ctypes.c_double

# name -> argtypes, exactly the prototypes of include/cineflow.h (tests/test_abi.py checks both directions)
SIGNATURES = {
    "cf_warp_bilinear_2d": [P, P, P, I, I, I, I, P],
    "cf_vecint_2d": [P, P, P, I, I, I, I, P],
    "cf_warp_labels_2d": [P, P, P, I, I, I, I, I, P],
    "cf_memory_input": [P, P, P, P, I, I, I, P],
    "cf_jacobian_det_2d": [P, P, I, I, I, P],
    "cf_warp_trilinear_3d": [P, P, P, I, I, I, I, I, P],
    "cf_jacobian_det_3d": [P, P, I, I, I, I, P],
    "cf_corr_volume": [P, P, P, I, I, I, I, I, I, P],
    "cf_corr_mfma_enable": [I],
    "cf_corr_pyramid": [P, P, P, I, I, I, I, I, P],
    "cf_corr_lookup": [P, P, P, I, I, I, I, I, P],
    "cf_convex_upsample": [P, P, P, I, I, I, I, P],
    "cf_conv2d": [P, I, P, I, P, L, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, P],
    "cf_conv_transpose2d_k2s2": [P, P, P, P, I, I, I, I, I, I, I, P],
    "cf_conv2d_f16s": [P, I, P, I, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, P, I, P],
    "cf_conv_transpose2d_k2s2_f16s": [P, P, P, P, I, I, I, I, I, I, I, F, P, I, P],
    "cf_group_norm_apply": [P, P, P, P, P, I, I, I, I, F, I, I, P, P],
    "cf_group_norm_coef": [P, P, P, I, I, I, I, F, P, P],
    "cf_norm_head_1x1": [P, P, F, P, P, P, I, I, I, I, P],
    "cf_conv2d_f16s_prenorm_ok": [I, I, I, I, I],
    "cf_conv_stream_enable": [I],
    "cf_conv2d_f16s_prenorm": [P, I, P, F, P, P, P, I, I, I, I, F, P, I, P],
    "cf_conv_terms": [I],
    "cf_conv_wino_enable": [I],
    "cf_conv2d_wino_ok": [I, I, I, I, I, I, I],
    "cf_conv2d_wino": [P, I, P, I, P, P, P, P, I, I, I, I, I, I, I, F, P, I, P],
    "cf_conv2d_wino_prenorm": [P, I, P, F, P, P, P, I, I, I, I, F, P, I, P],
    "cf_group_norm_apply_res_norm": [P, P, P, P, P, I, I, I, I, F, I, I, P, P, P, P, P],
    "cf_group_norm": [P, P, P, P, P, I, I, I, I, F, I, I, P, P],
    "cf_layer_norm_cf": [P, P, P, P, I, I, I, F, P],
    "cf_attention_cf": [P, L, P, L, P, L, P, I, I, I, I, I, P],
    "cf_attention_cf_masked": [P, L, P, L, P, L, P, I, I, I, I, I, I, P],
    "cf_gru_reset_mul": [P, P, P, I, I, I, P],
    "cf_gru_blend": [P, P, P, P, I, I, I, P],
    "cf_binary": [I, P, P, P, L, L, P],
    "cf_count_out_of_range": [P, L, F, P, P],
    "cf_copy_channels": [P, I, I, P, I, I, I, I, I, I, P],
    "cf_coords_grid": [P, I, I, I, P],
    "cf_crop2d": [P, P, I, I, I, I, I, I, I, P],
    "cf_pad2d": [P, P, I, I, I, I, I, I, I, P],
    "cf_tta_accumulate": [P, P, I, I, I, I, I, I, F, P],
    "cf_flip2d": [P, P, I, I, I, I, I, P],
    "cf_tile_accumulate": [P, P, P, P, I, I, I, I, I, I, I, P],
    "cf_tile_finalize": [P, P, P, P, I, I, I, P],
    "cf_tta_accumulate_3d": [P, P, I, I, I, I, I, I, I, I, F, P],
    "cf_flip3d": [P, P, I, I, I, I, I, I, I, P],
    "cf_tile_accumulate_3d": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cf_argmax_channels": [P, P, I, I, I, P],
    "cf_resize3d": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cf_cc_init": [P, P, L, P, I, P],
    "cf_cc_sweep": [P, I, I, I, P, P],
    "cf_cc_count": [P, P, L, P],
    "cf_cc_remove": [P, P, P, L, I, DBL, DBL, P],
    "cf_conv2d_small_cin": [P, P, P, P, I, I, I, I, I, I, P, I, P],
    "cf_conv2d_small_cout": [P, P, P, P, P, I, I, I, I, I, P],
    "cf_nonzero_mask": [P, I, L, P, P],
    "cf_fill_holes": [P, P, P, I, I, I, I, P],
    "cf_mask_bbox": [P, I, I, I, P, P],
    "cf_spline3_resample_axis": [P, P, L, I, L, I, P],
    "cf_slab_minmax_chunks": [I, I, I, L],
    "cf_slab_minmax": [P, I, I, I, L, P, P, P],
    "cf_slab_clip_to_f32": [P, P, I, I, I, L, P, P],
    "cf_masked_moments": [P, P, L, I, F, F, P, P],
    "cf_normalize": [P, P, L, I, F, F, F, F, I, P],
    "cf_nan_to_zero": [P, L, P],
    "cf_assign_where_ge": [P, P, L, F, F, P],
    "cf_seg_outside_mask": [P, P, I, L, F, P],
    "cf_confusion_counts": [P, P, L, P, P],
    "cf_label_confusion": [P, P, L, I, P, P],
    "cf_surface_border": [P, I, I, I, I, P, P, P],
    "cf_surface_min_dist": [P, I, P, I, DBL, DBL, DBL, P, P],
    "cf_max_sum_nonneg": [P, L, P, P],
    "cf_region_stats": [P, P, L, I, P, P],
    "cf_spatial_gradient3d": [P, P, L, I, I, I, P],
    "cf_slab_abs_sum": [P, I, I, I, L, P, P],
    "cf_ssim_map": [P, P, I, I, I, DBL, DBL, DBL, P, P],
    "cf_window_attention": [P, P, P, P, I, I, I, I, I, I, I, P],
    "cf_frame_boxes": [P, I, P, I, I, I, P],
    "cf_sample_points_2d": [P, P, P, I, I, I, I, I, P],
    "cf_profile_enable": [I],
    "cf_profile_reset": [],
    "cf_profile_read": [I, P, P, P],
}

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).  Import it FIRST so that this
    # process holds exactly one HIP runtime and libcineflow_hip.so binds to the one torch's allocator and streams live in;
    # loading our library first would bring in /opt/rocm's copy and the two runtimes then fight over the device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CineflowLibraryError(
            "libcineflow_hip.so not found at %s.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cardiac-segmentation-optical-flow_amd/csrc`).  cineflow has no CPU fallback." % LIB_PATH)
    try:
        h = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise CineflowLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    h.cf_last_error.restype = ctypes.c_char_p
    h.cf_last_error.argtypes = []
    h.cf_version.restype = ctypes.c_int
    h.cf_version.argtypes = []
    _lib = h
    return h


def check(rc, name):
    if rc != 0:
        raise CineflowError("%s failed (%d): %s" % (name, rc, lib().cf_last_error().decode()))
