"""ctypes mirror of include/gh_raster.h (the C-ABI drop-in boundary). Pure declarations."""
from __future__ import annotations

import ctypes as C

GH_TILE = 16
GH_CAM_FLOATS = 40
GH_FLAG_BLEND_W_PER_GAUSSIAN = 1
GH_FLAG_BLEND_COLOR_B_RGB = 2
GH_FLAG_PER_VIEW_GAUSSIANS = 4
GH_FLAG_SPLIT_STREAMS = 8
GH_FLAG_STATIC_LISTS = 16
GH_FLAG_DEPTH24 = 32
GH_FLAG_DEFER_LOSS_SUM = 64
GH_FLAG_FRESH_ORDER = 128
GH_VERSION_MAJOR, GH_VERSION_MINOR = 0, 8        # the header this mirror was written against (checked against gh_version() on load)
GH_ABI_TAG = 0x47480000 | (GH_VERSION_MAJOR << 8) | GH_VERSION_MINOR
GH_COUNTER_ERROR_MASK = 15     # GhCounters.overflow bits 0-3: errors
GH_COUNTER_DEPTH24_OK = 16     # bit 4: information (the depth keys' top byte did not vary)

GH_OK = 0
GH_ERR_INVALID_ARG = -1
GH_ERR_WORKSPACE_SMALL = -2
GH_ERR_LAUNCH = -3
GH_ERR_UNSUPPORTED = -4
GH_ERR_ABI = -5
_STATUS = {0: "GH_OK", -1: "GH_ERR_INVALID_ARG", -2: "GH_ERR_WORKSPACE_SMALL", -3: "GH_ERR_LAUNCH",
           -4: "GH_ERR_UNSUPPORTED", -5: "GH_ERR_ABI"}

fp = C.POINTER(C.c_float)


class GhDims(C.Structure):
    """GhDims(P, n_views, H, W, sh_degree, M, scale_modifier, flags, max_instances): the ABI tag (first field of the C struct) is
    filled in here, so that every struct this mirror builds carries the version this mirror was written against."""
    _fields_ = [("abi", C.c_uint32), ("P", C.c_int32), ("n_views", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("sh_degree", C.c_int32), ("M", C.c_int32), ("scale_modifier", C.c_float),
                ("flags", C.c_uint32), ("max_instances", C.c_int64)]

    def __init__(self, *args, **kw):
        super().__init__(GH_ABI_TAG, *args, **kw)


class GhInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "cams", "means3D", "opacities", "scales", "rotations", "shs", "colors_precomp",
        "blend_xyz_b", "blend_opacity_b", "blend_color_w", "blend_color_b", "tile_depth_bound", "cov3D_precomp")]


class GhFitLoss(C.Structure):
    _fields_ = [("gt_rgb", C.c_void_p), ("gt_mask", C.c_void_p), ("bbox", C.c_void_p), ("lambda_l1", C.c_float),
                ("lambda_mask", C.c_float), ("scale", C.c_float), ("dL_dimage", C.c_void_p), ("dL_dalpha", C.c_void_p),
                ("loss", C.c_void_p)]


class GhOutputs(C.Structure):
    _fields_ = [("image", C.c_void_p), ("radii", C.c_void_p), ("alpha", C.c_void_p), ("tile_depth_seen", C.c_void_p),
                ("tile_depth_seen_scale", C.c_float), ("tile_depth_seen_slack", C.c_uint32),
                ("l1_target", C.c_void_p), ("l1_dL_dimage", C.c_void_p), ("l1_loss", C.c_void_p),
                ("fit_loss", C.POINTER(GhFitLoss))]


class GhCounters(C.Structure):
    _fields_ = [("num_rendered", C.c_uint32), ("overflow", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class GhGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dimage", "dL_dalpha", "dL_dmeans3D", "dL_dmeans2D", "dL_dopacities", "dL_dscales", "dL_drotations",
        "dL_dshs", "dL_dcolors", "dL_dblend_xyz_b", "dL_dblend_opacity_b", "dL_dblend_color_w",
        "dL_dblend_color_b", "upstream_scale", "dL_dcov3D", "deferred_loss")]


class GhAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_size_t),
                ("reg_l1", C.c_float), ("reg_l2", C.c_float), ("partials", C.c_void_p), ("n_partials", C.c_int),
                ("step_state", C.c_void_p)]


LAYOUT_FIELDS = ("total_bytes", "counters", "geom", "clamped",
                 "tiles_touched", "slot_begin", "depth_keys_a", "depth_keys_b", "depth_vals_a", "depth_vals_b",
                 "block_sums", "keys_a", "keys_b", "vals_a", "vals_b", "sorted_slot", "inst_r0", "inst_r1", "inst_r2",
                 "sort_tables", "ranges", "tile_walk", "tile_order", "bwd_items", "ckpt_rgb", "final_C", "final_T", "n_contrib", "inst_grad", "inst_flag", "sh_rgb", "dmean_sh", "sh_scratch", "grad_sums", "bwd_scratch", "cull_bound", "inst_c", "attr", "half_counters", "key_bits", "tile_bound", "block_tiles", "render_guard", "loss_partials", "view_start")


class GhLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in LAYOUT_FIELDS]


def status_name(code: int) -> str:
    return _STATUS.get(code, f"GH_ERR({code})")


def declare(lib: C.CDLL) -> None:
    """Attach argtypes/restypes for every symbol include/gh_raster.h declares."""
    lib.gh_version.restype = C.c_int
    lib.gh_version.argtypes = []
    lib.gh_workspace_layout.restype = C.c_int
    lib.gh_workspace_layout.argtypes = [C.POINTER(GhDims), C.POINTER(GhLayout)]
    lib.gh_workspace_bytes.restype = C.c_size_t
    lib.gh_workspace_bytes.argtypes = [C.POINTER(GhDims)]
    lib.gh_partition_is_per_view.restype = C.c_int
    lib.gh_partition_is_per_view.argtypes = [C.POINTER(GhDims)]
    lib.gh_forward.restype = C.c_int
    lib.gh_forward.argtypes = [C.POINTER(GhDims), C.POINTER(GhInputs), C.POINTER(GhOutputs),
                               C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gh_backward.restype = C.c_int
    lib.gh_backward.argtypes = [C.POINTER(GhDims), C.POINTER(GhInputs), C.POINTER(GhGrads),
                                C.c_void_p, C.c_size_t, C.c_void_p]


    lib.gh_uv_sample_forward.restype = C.c_int
    lib.gh_uv_sample_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.gh_uv_sample_backward.restype = C.c_int
    lib.gh_uv_sample_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.gh_uv_gather_forward.restype = C.c_int
    lib.gh_uv_gather_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.gh_uv_gather_backward.restype = C.c_int
    lib.gh_uv_gather_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.gh_uv_scatter_sorted.restype = C.c_int
    lib.gh_uv_scatter_sorted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.gh_adam_reg_step.restype = C.c_int
    lib.gh_adam_reg_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    lib.gh_adam_reg_step_group.restype = C.c_int
    lib.gh_adam_reg_step_group.argtypes = [C.POINTER(GhAdamTensor), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p]
    lib.gh_uv_gather_forward2.restype = C.c_int
    lib.gh_uv_gather_forward2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p]
    lib.gh_uv_scatter_sorted2.restype = C.c_int
    lib.gh_uv_scatter_sorted2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_int, C.c_void_p]
    lib.gh_reg_total.restype = C.c_int
    lib.gh_reg_total.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
    lib.gh_l1_loss.restype = C.c_int
    lib.gh_l1_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.gh_fit_loss.restype = C.c_int
    lib.gh_fit_loss.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 4 + \
        [C.c_int, C.c_void_p, C.c_void_p]
    lib.gh_knn_workspace_bytes.restype = C.c_size_t
    lib.gh_knn_workspace_bytes.argtypes = [C.c_int]
    lib.gh_knn_indices.restype = C.c_int
    lib.gh_knn_indices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gh_knn_mismatch_mask.restype = C.c_int
    lib.gh_knn_mismatch_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.gh_forward_shared.restype = C.c_int
    lib.gh_forward_shared.argtypes = [C.POINTER(GhDims), C.POINTER(GhInputs), C.POINTER(GhOutputs), C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]
    lib.gh_backward_shared.restype = C.c_int
    lib.gh_backward_shared.argtypes = [C.POINTER(GhDims), C.POINTER(GhInputs), C.POINTER(GhGrads), C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]
    lib.gh_forward_refresh.restype = C.c_int
    lib.gh_forward_refresh.argtypes = lib.gh_forward_shared.argtypes
    lib.gh_backward_refresh.restype = C.c_int
    lib.gh_backward_refresh.argtypes = lib.gh_backward_shared.argtypes
    lib.gh_forward_stages.restype = C.c_int
    lib.gh_forward_stages.argtypes = lib.gh_forward.argtypes + [C.c_uint32]
    lib.gh_backward_stages.restype = C.c_int
    lib.gh_backward_stages.argtypes = lib.gh_backward.argtypes + [C.c_uint32]
    lib.gh_select_workspace_bytes.restype = C.c_size_t
    lib.gh_select_workspace_bytes.argtypes = [C.c_int]
    lib.gh_select_rows.restype = C.c_int
    lib.gh_select_rows.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + \
                                  [C.c_void_p, C.c_size_t, C.c_void_p]


GH_FWD_PREPROCESS, GH_FWD_BINNING, GH_FWD_RENDER, GH_FWD_ALL = 1, 2, 4, 7
GH_BWD_RENDER, GH_BWD_PREPROCESS, GH_BWD_ALL = 1, 2, 3

EXPORTED_SYMBOLS = ("gh_version", "gh_workspace_layout", "gh_workspace_bytes", "gh_partition_is_per_view", "gh_forward", "gh_backward",
                    "gh_forward_stages", "gh_backward_stages", "gh_forward_shared", "gh_backward_shared", "gh_forward_refresh", "gh_backward_refresh", "gh_uv_sample_forward", "gh_uv_sample_backward",
                    "gh_uv_gather_forward", "gh_uv_gather_backward", "gh_uv_scatter_sorted", "gh_adam_reg_step", "gh_reg_total", "gh_adam_reg_step_group", "gh_uv_gather_forward2", "gh_uv_scatter_sorted2",
                    "gh_knn_workspace_bytes", "gh_knn_indices", "gh_knn_mismatch_mask", "gh_l1_loss", "gh_fit_loss",
                    "gh_select_workspace_bytes", "gh_select_rows")
