"""Drop-in replacement for the reference's pybind11 module `maskrcnn_benchmark._C`
(maskrcnn_benchmark/csrc/vision.cpp:10-16) on the hot path: same callable names, argument order,
tensor layouts (NCHW fp32), return types and error behaviour — backed by libabr_iod_hip.so.

    nms(dets[n,4], scores[n], threshold) -> int64[k]                     csrc/nms.h:10-27
    roi_align_forward(input, rois, spatial_scale, ph, pw, sampling_ratio) csrc/ROIAlign.h:11-25
    roi_align_backward(grad, rois, spatial_scale, ph, pw, B, C, H, W, sr) csrc/ROIAlign.h:27-45
    sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha) csrc/SigmoidFocalLoss.h:10-23
    sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha)   :25-41

CPU tensors raise RuntimeError (the reference raises "Not compiled with GPU support" the other way round):
this library is MI355X-only by design; the CPU restatement lives in oracle/ and is test infrastructure.
"""
import torch

from . import _lib as L

# NMS comparison rule.  The reference is inconsistent: CPU suppresses at IoU >= thr (nms_cpu.cpp:60),
# CUDA at IoU > thr (nms.cu:60).  The CPU path is the parity oracle, so '>=' is the default here.
NMS_STRICT_GT = False


def nms(dets, scores, threshold, strict_gt=None):
    L.require_cuda(dets, scores)
    if dets.numel() == 0:  # nms.h:15-16
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    if dets.dtype != torch.float32 or scores.dtype != torch.float32:
        raise RuntimeError("nms: dets and scores must be float32 (nms.cu:71 is float-only)")
    strict = NMS_STRICT_GT if strict_gt is None else strict_gt
    n = dets.shape[0]
    if n <= int(L.lib().abr_sort_scores_max_n()):
        # the whole call inside the library (round 5): score ranking on the proposal ranking's own kernels, gather, mask + sweep, and the
        # survivors' original indices compacted in ascending order -- no ATen sort on either side of the suppression
        d, s = dets.contiguous(), scores.contiguous()
        nbytes = int(L.lib().abr_nms_unsorted_workspace_bytes(n))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dets.device)
        keep_out = torch.empty((n,), dtype=torch.int64, device=dets.device)
        n_keep = torch.empty((1,), dtype=torch.int32, device=dets.device)
        L.check(L.lib().abr_nms(L.ptr(d), L.ptr(s), n, float(threshold), int(strict), L.ptr(keep_out), L.ptr(n_keep), L.ptr(ws), nbytes, L.stream()), "nms")
        return keep_out[: int(n_keep.item())]   # the reference blocks here too (nms.cu:100 cudaMemcpy D2H)
    # more boxes than the in-LDS merge of the score sort takes (15360; the reference's call sites stay at or below 12000): ATen's sort
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    boxes = dets.index_select(0, order).contiguous()
    counts = torch.tensor([n], dtype=torch.int32, device=dets.device)
    keep = torch.empty((1, n), dtype=torch.int32, device=dets.device)
    n_keep = torch.empty((1,), dtype=torch.int32, device=dets.device)
    ws_bytes = L.lib().abr_nms_workspace_bytes(1, n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dets.device)
    L.check(L.lib().abr_nms_sorted_batched(L.ptr(boxes), L.ptr(counts), 1, n, float(threshold), int(strict), n,
                                           L.ptr(keep), L.ptr(n_keep), L.ptr(ws), ws_bytes, L.stream()), "nms")
    k = int(n_keep.item())  # the reference blocks here too (nms.cu:100 cudaMemcpy D2H)
    # nms.cu:127-130 / nms_cpu.cpp:66: ascending ORIGINAL indices of the survivors
    return order.index_select(0, keep[0, :k].long()).sort()[0]


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """float32 or float64 (AT_DISPATCH_FLOATING_TYPES, ROIAlign_cuda.cu:283)"""
    L.require_cuda(input, rois)
    x = L.fpc(input)
    r = L.fpc(rois, like=x)
    B, Ch, H, W = x.shape
    K = r.shape[0]
    out = torch.empty((K, Ch, pooled_height, pooled_width), dtype=x.dtype, device=x.device)
    if out.numel() == 0:
        return out
    if x.dtype == torch.float64:
        L.check(L.lib().abr_roi_align_forward_f64(L.ptr(x), L.ptr(r), K, B, Ch, H, W, float(spatial_scale), pooled_height, pooled_width,
                                                  sampling_ratio, L.ptr(out), L.stream()), "roi_align_forward (float64)")
        return out
    L.check(L.lib().abr_roi_align_forward(L.ptr(x), L.ptr(r), K, B, Ch, H, W, float(spatial_scale), pooled_height,
                                          pooled_width, sampling_ratio, 1, L.NCHW, L.ptr(out), L.stream()),
            "roi_align_forward")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                       sampling_ratio):
    L.require_cuda(grad, rois)
    g = L.fpc(grad)
    r = L.fpc(rois, like=g)
    K = r.shape[0]
    out = torch.empty((batch_size, channels, height, width), dtype=g.dtype, device=g.device)
    if g.dtype == torch.float64:   # ROIAlign_cuda.cu:329
        L.check(L.lib().abr_roi_align_backward_f64(L.ptr(g), L.ptr(r), K, batch_size, channels, height, width, float(spatial_scale), pooled_height,
                                                   pooled_width, sampling_ratio, L.ptr(out), L.stream()), "roi_align_backward (float64)")
        return out
    L.check(L.lib().abr_roi_align_backward(L.ptr(g), L.ptr(r), K, batch_size, channels, height, width,
                                           float(spatial_scale), pooled_height, pooled_width, sampling_ratio, 1, L.NCHW,
                                           0, L.ptr(out), L.stream()), "roi_align_backward")
    return out


def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    L.require_cuda(logits, targets)
    if logits.dim() != 2:
        raise RuntimeError("logits should be NxClass")  # SigmoidFocalLoss_cuda.cu:112
    x = L.fpc(logits)
    t = targets.to(torch.int32).contiguous()
    out = torch.empty_like(x)
    if x.dtype == torch.float64:   # SigmoidFocalLoss_cuda.cu:128
        L.check(L.lib().abr_sigmoid_focal_forward_f64(L.ptr(x), L.ptr(t), x.shape[0], num_classes, float(gamma), float(alpha), L.ptr(out), L.stream()),
                "sigmoid_focalloss_forward (float64)")
        return out
    L.check(L.lib().abr_sigmoid_focal_forward(L.ptr(x), L.ptr(t), x.shape[0], num_classes, float(gamma), float(alpha),
                                              L.ptr(out), L.stream()), "sigmoid_focalloss_forward")
    return out


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    L.require_cuda(logits, targets, d_losses)
    x = L.fpc(logits)
    d = L.fpc(d_losses, like=x)
    t = targets.to(torch.int32).contiguous()
    out = torch.empty_like(x)
    if x.dtype == torch.float64:   # SigmoidFocalLoss_cuda.cu:172
        L.check(L.lib().abr_sigmoid_focal_backward_f64(L.ptr(x), L.ptr(t), L.ptr(d), x.shape[0], num_classes, float(gamma), float(alpha), L.ptr(out),
                                                       L.stream()), "sigmoid_focalloss_backward (float64)")
        return out
    L.check(L.lib().abr_sigmoid_focal_backward(L.ptr(x), L.ptr(t), L.ptr(d), x.shape[0], num_classes, float(gamma),
                                               float(alpha), L.ptr(out), L.stream()), "sigmoid_focalloss_backward")
    return out


def _not_on_path(name):
    def f(*a, **k):
        raise RuntimeError(f"_C.{name} is outside the hot path (never instantiated by any configs/voc YAML; SURVEY.md §2 rows 17-18)")
    return f


roi_pool_forward = _not_on_path("roi_pool_forward")
roi_pool_backward = _not_on_path("roi_pool_backward")
