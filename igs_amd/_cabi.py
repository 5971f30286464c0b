"""ctypes binding of the C ABI declared in include/igs_rast.h (libigs_rast.so, gfx950).

There is NO CPU fallback: if the HIP library cannot be built or loaded this module raises.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None
E_RETRY = -6      # IGS_RAST_E_RETRY

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

class RefineStepArgs(C.Structure):
    """igs_refine_step_args (include/igs_rast.h)."""
    _fields_ = [("stream", C.c_void_p),
                ("geometry_buffer", ALLOC_FN), ("geometry_user", C.c_void_p),
                ("binning_buffer", ALLOC_FN), ("binning_user", C.c_void_p),
                ("image_buffer", ALLOC_FN), ("image_user", C.c_void_p),
                ("workspace", C.c_void_p),
                ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("background", C.c_void_p),
                ("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("grad_out", C.c_void_p),
                ("off_xyz", C.c_size_t), ("off_rot", C.c_size_t), ("off_sh", C.c_size_t), ("off_opacity", C.c_size_t),
                ("off_scale", C.c_size_t),
                ("lr_xyz", C.c_float), ("lr_rot", C.c_float), ("lr_sh", C.c_float), ("lr_opacity", C.c_float), ("lr_scale", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("gt", C.c_void_p), ("loss_weight", C.c_float), ("lambda_dssim", C.c_float), ("lambda_depth_normal", C.c_float), ("depth_ratio", C.c_float),
                ("loss_scratch", C.c_void_p),
                ("out_images", C.c_void_p), ("radii", C.c_void_p), ("dL_dmean2D", C.c_void_p), ("loss_out", C.c_void_p),
                ("require_coord", C.c_int), ("require_depth", C.c_int), ("clamp_grads", C.c_float),
                ("color_grad_out", C.c_void_p), ("scratch_clean", C.c_int),
                ("gt_stats", C.c_void_p), ("gt_stats_valid", C.c_int), ("color_ready_event", C.c_void_p)]


EXPORTS = ["igs_rast_version", "igs_rast_last_error", "igs_rast_forward", "igs_rast_backward_workspace_bytes",
           "igs_rast_forward_async", "igs_rast_forward_finish", "igs_rast_forward_nowait", "igs_rast_last_status", "igs_rast_last_posted_status", "igs_rast_hint_scratch_clean", "igs_rast_set_slab_hint", "igs_rast_get_slab_hint", "igs_rast_backward", "igs_rast_mark_visible", "igs_rast_debug_dump",
           "igs_rast_profile_enable", "igs_rast_profile_read", "igs_adam_step", "igs_adam_step_groups", "igs_adam_step_multi", "igs_adam_step_multi_dev", "igs_adam_step_multi_dev_scratch_words", "igs_densify_stats", "igs_densify_remap", "igs_refine_step", "igs_refine_loss_scratch_bytes", "igs_ssim_l1_scratch_bytes", "igs_ssim_l1_loss_fwd_bwd", "igs_ssim_l1_loss_fwd_bwd_cached", "igs_ssim_mean_fwd_bwd", "igs_ssim_gt_stats_bytes", "igs_depth_normal_loss_fwd_bwd", "igs_l1_loss_fwd_bwd", "igs_l1_mean_fwd_bwd", "igs_activate_fwd", "igs_activate_bwd",
           "igs_sh_grad_from_view_colors", "igs_adam_sh_from_view_colors", "igs_rast_last_backward_instance", "igs_rast_next_backward_options", "igs_rast_nan_report_wait", "igs_rast_nan_report_handle", "igs_rast_nan_report_wait_at", "igs_refine_step_args_size", "igs_rast_debug_poison_lds", "igs_adam_exchange_step", "igs_morton_order", "igs_morton_order_scratch_bytes", "igs_ply_to_params", "igs_params_to_ply", "igs_debug_tile_sort"]

VERSION = 4       # IGS_RAST_VERSION this binding was written against (include/igs_rast.h)

STAGES = ["preprocess", "depth_sort", "scan", "emit", "tile_sort", "ranges", "blend_fwd", "memset", "blend_bwd", "geom_bwd"]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        path = _build.build()
    except Exception as e:  # noqa: BLE001
        # no hipcc on this box (or the build failed): the library that travelled with the tree is used only if it was built from
        # exactly these sources -- a stale one with another igs_refine_step_args layout would corrupt memory instead of failing
        if os.path.exists(_build.LIB) and not _build.needs_build():
            path = _build.LIB
        elif os.path.exists(_build.LIB):
            raise RuntimeError("igs_amd: libigs_rast.so is present but was built from other sources (build.stamp does not match) "
                               "and rebuilding it failed: %s" % e)
        else:
            raise RuntimeError("igs_amd: the HIP extension libigs_rast.so is missing and could not be built: %s" % e)
    # Load order matters: the library needs libamdhip64, and so does PyTorch, which ships its own copy.  Whichever is mapped
    # first serves both; if ours (from /opt/rocm) came first, torch would bring a SECOND runtime and the two would not share
    # the device (symptom: hipGetDevice -> "no ROCm-capable device is detected").  Device memory and streams come from torch
    # on this path, so torch's runtime goes first.
    import torch  # noqa: F401
    L = C.CDLL(path)
    L.igs_rast_version.restype = _i
    if L.igs_rast_version() != VERSION:
        raise RuntimeError("igs_amd: libigs_rast.so reports C-ABI version %d, this binding needs %d" % (L.igs_rast_version(), VERSION))
    L.igs_refine_step_args_size.restype = C.c_size_t
    if L.igs_refine_step_args_size() != C.sizeof(RefineStepArgs):
        raise RuntimeError("igs_amd: igs_refine_step_args is %d bytes in the library, %d in the binding"
                           % (L.igs_refine_step_args_size(), C.sizeof(RefineStepArgs)))
    L.igs_rast_last_backward_instance.restype = _i
    L.igs_rast_last_backward_instance.argtypes = []
    L.igs_rast_next_backward_options.restype = None
    L.igs_rast_next_backward_options.argtypes = [_i, _f]
    L.igs_rast_nan_report_wait.restype = _i
    L.igs_rast_nan_report_wait.argtypes = []
    L.igs_rast_debug_poison_lds.restype = _i
    L.igs_rast_debug_poison_lds.argtypes = [_vp]
    L.igs_rast_last_error.restype = C.c_char_p
    L.igs_rast_forward.restype = _i
    L.igs_rast_forward.argtypes = ([_vp, ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i]
                                   + [_vp] * 5 + [_f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i] + [_vp] * 8 + [_i, _i, _i])
    L.igs_rast_forward_async.restype = _i
    L.igs_rast_forward_async.argtypes = L.igs_rast_forward.argtypes
    L.igs_rast_forward_nowait.restype = _i
    L.igs_rast_forward_nowait.argtypes = L.igs_rast_forward.argtypes
    L.igs_rast_hint_scratch_clean.restype = None
    L.igs_rast_hint_scratch_clean.argtypes = [_i]
    L.igs_rast_last_status.restype = _i
    L.igs_rast_last_status.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.igs_rast_last_posted_status.restype = _i
    L.igs_rast_last_posted_status.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.igs_rast_forward_finish.restype = _i
    L.igs_rast_forward_finish.argtypes = []
    L.igs_rast_set_slab_hint.restype = None
    L.igs_rast_set_slab_hint.argtypes = [C.c_uint]
    L.igs_rast_get_slab_hint.restype = C.c_uint
    L.igs_rast_get_slab_hint.argtypes = []
    L.igs_rast_backward_workspace_bytes.restype = C.c_size_t
    L.igs_rast_backward_workspace_bytes.argtypes = [_i]
    L.igs_rast_backward.restype = _i
    L.igs_rast_backward.argtypes = ([_vp, _i, _i, _i, _i, _vp, _i, _i] + [_vp] * 5 + [_f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f]
                                    + [_vp] * 5 + [_vp] * 7 + [_vp] + [_vp] * 8 + [_i, _i, _i])
    L.igs_rast_mark_visible.restype = _i
    L.igs_rast_mark_visible.argtypes = [_vp, _i, _vp, _vp, _vp, _vp]
    L.igs_rast_debug_dump.restype = _i
    L.igs_rast_debug_dump.argtypes = [_vp, _i, _i, _i, _i] + [_vp] * 8
    L.igs_rast_profile_enable.restype = _i
    L.igs_rast_profile_enable.argtypes = [_i]
    L.igs_rast_profile_read.restype = _i
    L.igs_rast_profile_read.argtypes = [_vp, _vp, _vp, _vp, _i]
    if hasattr(L, "igs_adam_step"):
        L.igs_adam_step.restype = _i
        L.igs_adam_step.argtypes = [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f]
        L.igs_adam_step_groups.restype = _i
        L.igs_adam_step_groups.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f]
        L.igs_adam_step_multi.restype = _i
        L.igs_adam_step_multi.argtypes = [_vp, _i] + [_vp] * 8 + [_f, _f, _f]
        L.igs_adam_step_multi_dev.restype = _i
        L.igs_adam_step_multi_dev.argtypes = [_vp, _i] + [_vp] * 8 + [_f, _f, _f]
        L.igs_adam_step_multi_dev_scratch_words.restype = C.c_size_t
        L.igs_adam_step_multi_dev_scratch_words.argtypes = []
        L.igs_densify_stats.restype = _i
        L.igs_densify_stats.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp]
        L.igs_densify_remap.restype = _i
        L.igs_densify_remap.argtypes = [_vp, _i, _i] + [_vp] * 13
        L.igs_refine_step.restype = _i
        L.igs_refine_step.argtypes = [C.POINTER(RefineStepArgs)]
        L.igs_refine_loss_scratch_bytes.restype = C.c_size_t
        L.igs_refine_loss_scratch_bytes.argtypes = [_i, _i]
        L.igs_ssim_l1_scratch_bytes.restype = C.c_size_t
        L.igs_ssim_l1_scratch_bytes.argtypes = [_i, _i]
        L.igs_ssim_l1_loss_fwd_bwd.restype = _i
        L.igs_ssim_l1_loss_fwd_bwd.argtypes = [_vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp]
        L.igs_ssim_l1_loss_fwd_bwd_cached.restype = _i
        L.igs_ssim_l1_loss_fwd_bwd_cached.argtypes = [_vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i]
        L.igs_ssim_gt_stats_bytes.restype = C.c_size_t
        L.igs_ssim_gt_stats_bytes.argtypes = [_i, _i]
        L.igs_morton_order_scratch_bytes.restype = C.c_size_t
        L.igs_debug_tile_sort.restype = _i
        L.igs_debug_tile_sort.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i]
        L.igs_morton_order_scratch_bytes.argtypes = [_i]
        L.igs_morton_order.restype = _i
        L.igs_morton_order.argtypes = [_vp, _i, _vp, _vp, _i, _vp, _vp]
        L.igs_depth_normal_loss_fwd_bwd.restype = _i
        L.igs_depth_normal_loss_fwd_bwd.argtypes = [_vp, _i, _i, _f, _f, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]
        L.igs_l1_loss_fwd_bwd.restype = _i
        L.igs_l1_loss_fwd_bwd.argtypes = [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _f]
        L.igs_l1_mean_fwd_bwd.restype = _i
        L.igs_l1_mean_fwd_bwd.argtypes = [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp]
        L.igs_ply_to_params.restype = _i
        L.igs_ply_to_params.argtypes = [_vp, _i, _vp, _i, _vp, _i, _i] + [_vp] * 5
        L.igs_params_to_ply.restype = _i
        L.igs_params_to_ply.argtypes = [_vp, _i, _i] + [_vp] * 6
        L.igs_activate_fwd.restype = _i
        L.igs_activate_fwd.argtypes = [_vp, _i] + [_vp] * 6
        L.igs_activate_bwd.restype = _i
        L.igs_activate_bwd.argtypes = [_vp, _i] + [_vp] * 9
        L.igs_sh_grad_from_view_colors.restype = _i
        L.igs_sh_grad_from_view_colors.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp]
        L.igs_adam_sh_from_view_colors.restype = _i
        L.igs_adam_sh_from_view_colors.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f]
        L.igs_adam_exchange_step.restype = _i
        L.igs_adam_exchange_step.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp] + [C.c_size_t] * 5 + [_f] * 10
    _LIB = L
    return L


_EXT = None


def ext():
    """The compiled `_C` module (igs_amd/_C.*.so, igs_amd/csrc_torch/igs_torch_ext.cpp), built in-tree on first use.  No fallback:
    raises if it cannot be built or does not match the C-ABI library."""
    global _EXT
    if _EXT is not None:
        return _EXT
    lib()                                   # libigs_rast.so first (build / stamp / version checks; torch's HIP runtime mapped before ours)
    from . import build_ext as _bx
    try:
        _bx.build()
    except Exception as e:  # noqa: BLE001
        if not (os.path.exists(_bx.TARGET) and not _bx.needs_build()):
            raise RuntimeError("igs_amd: the compiled _C module is missing or stale and could not be built: %s" % e)
    import importlib
    m = importlib.import_module("igs_amd._C")
    if m.abi_version() != VERSION:
        raise RuntimeError("igs_amd: the compiled _C module was built against C-ABI version %d, this binding needs %d" % (m.abi_version(), VERSION))
    _EXT = m
    return m


def last_backward_instance():
    """{coord, depth, normal, absgrad} of the blend-backward instance the last backward on this thread launched, or None."""
    b = lib().igs_rast_last_backward_instance()
    if b < 0:
        return None
    return dict(coord=bool(b & 1), depth=bool(b & 2), normal=bool(b & 4), absgrad=bool(b & 8))


def last_error():
    return lib().igs_rast_last_error().decode("utf-8", "replace")


def profile_enable(on=True, every=1):
    """Stage marks on every `every`-th frame (an event record costs a few microseconds of stream time)."""
    lib().igs_rast_profile_enable(int(every) if on else 0)


def profile_read(reset=True):
    """{stage: (ms_sum, count)}, sum of num_rendered, number of forward calls."""
    n = len(STAGES)
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    r = C.c_double(0)
    calls = C.c_longlong(0)
    lib().igs_rast_profile_read(ms, cnt, C.byref(r), C.byref(calls), int(bool(reset)))
    return {STAGES[i]: (ms[i], cnt[i]) for i in range(n)}, r.value, calls.value
