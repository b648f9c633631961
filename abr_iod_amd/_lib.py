"""ctypes binding of libabr_iod_hip.so (C ABI declared in include/abr_iod_hip.h).

PyTorch is plumbing here: it owns device memory and streams; every op on the hot path is a HIP kernel
behind the C ABI.  There is NO fallback: if the library is missing or a call fails this raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ABR_IOD_HIP_LIB: another build of the same C ABI (A/B measurements of two library versions inside one GPU session)
LIB_PATH = os.environ.get("ABR_IOD_HIP_LIB") or os.path.join(_HERE, "libabr_iod_hip.so")

NCHW, NHWC = 0, 1

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64


class ConvDesc(C.Structure):
    """abr_conv_desc (include/abr_iod_hip.h section 3)."""

    _fields_ = [
        ("B", _i), ("H", _i), ("W", _i), ("Cin", _i),
        ("Cout", _i), ("R", _i), ("S", _i),
        ("stride", _i), ("pad", _i),
        ("Ho", _i), ("Wo", _i),
        ("scale", _vp), ("bias", _vp), ("residual", _vp), ("mask", _vp),
        ("relu", _i),
        ("out_H", _i), ("out_W", _i), ("out_sh", _i), ("out_sw", _i),
        ("math", _i),
        ("wino_v", _vp),
        ("w_planes", _vp),
        ("w_version", _i64),
        ("x_amax", _vp), ("x_amax_epoch", C.c_uint32),
        ("gy_amax", _vp), ("gy_amax_epoch", C.c_uint32),
        ("out_amax", _vp), ("out_amax_epoch", C.c_uint32),
    ]


class ConvOp(C.Structure):
    """abr_conv_op (include/abr_iod_hip.h): one entry of abr_conv_run's table."""

    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("desc", ConvDesc), ("a", _vp), ("b", _vp), ("out", _vp), ("stream", _vp), ("other", _vp)]


OP_FORWARD, OP_WGRAD, OP_STREAM_WAIT = 0, 1, 2


class PrepItem(C.Structure):
    """abr_prep_item (include/abr_iod_hip.h section 3): one tensor of abr_conv_prepare_batch."""

    _fields_ = [("w", _vp), ("scale", _vp), ("wt", _vp),
                ("Cout", C.c_int32), ("R", C.c_int32), ("S", C.c_int32), ("Cin", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("math", C.c_int32),
                ("w_version", _i64)]


_SIGS = {
    "abr_version": (_i, []),
    "abr_device_info": (_i, [_vp]),
    "abr_prof_begin": (_i, []),
    "abr_prof_mark_overlap": (_i, [_i]),
    "abr_prof_set_mask": (_i, [C.c_uint32, _i]),
    "abr_prof_end": (_i, [_vp, _i]),
    "abr_prof_totals": (_i, [_vp, _i]),
    "abr_prof_bytes": (_i, [_vp, _i]),
    "abr_prof_clocks": (_i, [_vp, _i]),
    "abr_prof_event_overhead_ms": (_i, [_vp, _vp]),
    "abr_prof_step_begin": (_i, []),
    "abr_conv_prepare_weights": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i64, _vp]),
    "abr_conv_prepare_batch": (_i, [_vp, _i, _vp]),
    "abr_conv_run": (_i, [_vp, _i]),
    "abr_conv_cache_clear": (_i, []),
    "abr_conv_cache_drop_range": (_i, [_vp, _i64]),
    "abr_conv_cache_bytes": (_i64, []),
    "abr_roi_head_targets": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _f, _i, _i, C.c_uint64,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "abr_rpn_targets_batched": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _i64, _vp]),
    "abr_rpn_loss_indices": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "abr_loss_sum": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "abr_loss_sum_backward": (_i, [_vp, _i, _vp, _vp, _vp]),
    "abr_gather_proposals": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "abr_h3_amax_alloc": (_i, [_vp, _vp]),
    "abr_h3_amax_alloc_block": (_i, [_i, _vp, _vp]),
    "abr_stream_wait_stream": (_i, [_vp, _vp]),
    "abr_h3_amax": (_i, [_vp, _i64, _vp, C.c_uint32, _vp]),
    "abr_h3_range_stats": (_i, [_vp, _i, _vp]),
    "abr_h3_range_stats_to_device": (_i, [_vp, _i, _vp]),
    "abr_x6_range_flags": (_i, [_vp, _i, _vp]),
    "abr_x6_range_flags_async": (_i, [_vp, _vp]),
    "abr_x6_range_flags_to_device": (_i, [_vp, _vp]),
    "abr_roi_align_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _vp, _vp]),
    "abr_roi_align_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "abr_roi_align_backward_ws_bytes": (_i64, [_i, _i, _i, _i, _i, _i, _i]),
    "abr_roi_align_backward_gather": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp]),
    "abr_roi_align_taps": (_i, [_vp, _i, _i, _i, _f, _i, _i, _i, _i, _vp, _vp, _vp]),
    "abr_nms_workspace_bytes": (_i64, [_i, _i]),
    "abr_nms_unsorted_workspace_bytes": (_i64, [_i]),
    "abr_nms": (_i, [_vp, _vp, _i, _f, _i, _vp, _vp, _vp, _i64, _vp]),
    "abr_sort_scores_max_n": (_i64, []),
    "abr_sort_scores_desc": (_i, [_vp, _i, _vp, _vp]),
    "abr_nms_sorted_batched": (_i, [_vp, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    "abr_roi_align_forward_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.c_double, _i, _i, _i, _vp, _vp]),
    "abr_roi_align_backward_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.c_double, _i, _i, _i, _vp, _vp]),
    "abr_sigmoid_focal_forward_f64": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "abr_sigmoid_focal_backward_f64": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "abr_sigmoid_focal_forward": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "abr_sigmoid_focal_backward": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "abr_ard_forward": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "abr_ard_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp]),
    "abr_smooth_l1": (_i, [_vp, _vp, _i64, _f, _f, _vp, _f, _vp, _vp]),
    "abr_smooth_l1_rows": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _f, _vp, _vp]),
    "abr_softmax_ce": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _f, _vp, _i, _vp]),
    "abr_feat_distill": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _vp, _vp]),
    "abr_rpn_distill": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i64, _i, _f, _i, _vp, _f, _vp, _vp, _i, _i, _vp]),
    "abr_roi_distill": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _f, _vp, _vp, _vp]),
    "abr_bce_logits_gather": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp]),
    "abr_img_resample_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "abr_img_blend_paste_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.c_double, _vp]),
    "abr_img_copy_rect_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "abr_img_fill_u8": (_i, [_vp, _i64, _i, _vp]),
    "abr_img_color_jitter_u8": (_i, [_vp, _i, _i, _i, C.c_double, _vp, _vp]),
    "abr_img_normalize_to_batch": (_i, [_vp, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _i, _i, _vp]),
    "abr_conv_forward": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "abr_conv_tail64_forward": (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "abr_conv_wgrad": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "abr_conv_wino_v_floats": (_i64, [C.POINTER(ConvDesc)]),
    "abr_conv_packed_bytes": (_i64, [_i64, _i64]),
    "abr_conv_pack_weights": (_i, [_vp, _i64, _i, _vp, _vp]),
    "abr_conv_dgrad_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "abr_bias_grad": (_i, [_vp, _i64, _i, _vp, _vp]),
    "abr_nchw_to_nhwc_pad": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "abr_nchw_to_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "abr_nhwc_to_nchw": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "abr_maxpool3x3s2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "abr_avgpool_forward": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "abr_avgpool_backward": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "abr_avgpool_relu_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "abr_avgpool_relu_backward_amax": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, C.c_uint32, _vp]),
    "abr_channel_mean": (_i, [_vp, _i64, _i, _vp, _vp]),
    "abr_relu_backward": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "abr_add_inplace": (_i, [_vp, _vp, _i64, _vp]),
    "abr_scale_inplace": (_i, [_vp, _i64, _f, _vp, _vp]),
    "abr_grid_anchors": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "abr_topk_sigmoid": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "abr_det_softmax_decode": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _f, _f, _f, _f, _vp, _vp, _vp]),
    "abr_det_select_workspace_bytes": (_i64, [_i, _i, _i]),
    "abr_det_select": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "abr_rpn_decode_clip": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _f, _f, _f, _f, _vp, _vp]),
    "abr_box_encode": (_i, [_vp, _vp, _i, _f, _f, _f, _f, _vp, _vp]),
    "abr_match_workspace_bytes": (_i64, [_i, _i]),
    "abr_match_encode": (_i, [_vp, _i, _vp, _vp, _i, _vp, _f, _f, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "abr_sample_pos_neg": (_i, [_vp, _i, _i, _i, _i64, _i, _i, C.c_uint64, _i, _i64, _vp, _vp, _vp, _vp]),
    "abr_sgd_momentum": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _f, _f, _i, _vp]),
    "abr_comm_rccl_version": (_i, []),
    "abr_comm_unique_id": (_i, [_vp]),
    "abr_comm_init": (_i, [_i, _i, _vp, C.POINTER(_vp)]),
    "abr_comm_info": (_i, [_vp, _vp]),
    "abr_comm_destroy": (_i, [_vp]),
    "abr_allreduce_flat": (_i, [_vp, _vp, _vp, _i, _vp]),
}

# every symbol include/abr_iod_hip.h declares (tests/test_abi.py checks the library exports them all)
EXPORTS = sorted(list(_SIGS) + ["abr_last_error"])

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  abr_iod_amd has no CPU/eager fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        L.abr_last_error.restype = C.c_char_p
        L.abr_last_error.argtypes = []
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().abr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"abr_iod_hip {what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    """raw handle of torch's CURRENT stream on the current device.  torch.cuda.current_stream().cuda_stream builds a Stream object through
    three layers of Python (~9 us: 1.3 ms of host time per training step over ~150 launches); the two C calls below are ~0.3 us"""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except (AttributeError, RuntimeError):      # (private names moved, or the runtime is not initialised yet)
        return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("abr_iod_amd ops need device tensors (MI355X); there is no CPU path — "
                               "the CPU oracle lives in oracle/ and is test infrastructure only")


def f32c(t):
    """contiguous fp32 view/copy (the reference kernels call .contiguous() too, ROIAlign_cuda.cu:286)."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def fpc(t, like=None):
    """contiguous float32 / float64 tensor (the two types AT_DISPATCH_FLOATING_TYPES gives the reference's _C entry points); `like`: the
    tensor whose dtype it must share"""
    if t.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"expected float32 or float64, got {t.dtype}")
    if like is not None and t.dtype != like.dtype:
        raise RuntimeError(f"expected {like.dtype} like the first argument, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()
