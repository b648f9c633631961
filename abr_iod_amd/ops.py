"""Functional host API over the C ABI in the library's NATIVE layout (NHWC activations, OHWI weights).

Thin: shape bookkeeping + output allocation (torch = device-memory plumbing) + one C-ABI call each.
The reference-shaped wrappers (autograd Functions, nn.Modules with the reference's names) live in
abr_iod_amd/layers and abr_iod_amd/modeling and are built from these.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib as L

_f32 = torch.float32


def _empty(shape, like, dtype=_f32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ----------------------------------------------------------------------------------------------- ROIAlign
def roi_align_forward(feat, rois, spatial_scale, ph, pw, sampling_ratio, bin_step=1, out=None):
    """feat [B,H,W,C], rois [K,5] -> [K, ceil(ph/step), ceil(pw/step), C] (written into `out`, a contiguous tensor of that shape, when given)"""
    L.require_cuda(feat, rois)
    feat, rois = L.f32c(feat), L.f32c(rois)
    B, H, W, Ch = feat.shape
    K = rois.shape[0]
    pho, pwo = -(-ph // bin_step), -(-pw // bin_step)
    if out is None:
        out = _empty((K, pho, pwo, Ch), feat)
    else:
        assert tuple(out.shape) == (K, pho, pwo, Ch) and out.is_contiguous() and out.dtype == _f32
    L.check(L.lib().abr_roi_align_forward(L.ptr(feat), L.ptr(rois), K, B, Ch, H, W, float(spatial_scale), ph, pw,
                                          sampling_ratio, bin_step, L.NHWC, L.ptr(out), L.stream()), "roi_align_forward")
    return out


_roi_bwd_ws = {}


def roi_align_backward(grad, rois, spatial_scale, ph, pw, sampling_ratio, B, H, W, Ch, bin_step=1, out=None, method="gather"):
    """grad [K,pho,pwo,C] -> grad_feat [B,H,W,C]; accumulates into `out` when given.
    method 'gather' = atomic-free two-kernel form (default), 'scatter' = per-RoI atomics (abr_roi_align_backward)."""
    L.require_cuda(grad, rois)
    grad, rois = L.f32c(grad), L.f32c(rois)
    K = rois.shape[0]
    acc = out is not None
    if out is None:
        out = _empty((B, H, W, Ch), grad)
    if method == "gather" and Ch % 4 == 0 and -(-ph // bin_step) <= 8 and -(-pw // bin_step) <= 8:
        nbytes = L.lib().abr_roi_align_backward_ws_bytes(K, B, H, W, ph, pw, bin_step)
        key = (grad.device,)
        ws = _roi_bwd_ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _roi_bwd_ws[key] = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=grad.device)
        L.check(L.lib().abr_roi_align_backward_gather(L.ptr(grad), L.ptr(rois), K, B, Ch, H, W, float(spatial_scale), ph, pw,
                                                      sampling_ratio, bin_step, int(acc), L.ptr(out), L.ptr(ws), ws.numel(),
                                                      L.stream()), "roi_align_backward_gather")
        amax_drop(out)
        return out
    L.check(L.lib().abr_roi_align_backward(L.ptr(grad), L.ptr(rois), K, B, Ch, H, W, float(spatial_scale), ph, pw,
                                           sampling_ratio, bin_step, L.NHWC, int(acc), L.ptr(out), L.stream()),
            "roi_align_backward")
    amax_drop(out)
    return out


def roi_align_taps(rois, H, W, spatial_scale, ph, pw, sampling_ratio, max_s):
    rois = L.f32c(rois)
    K = rois.shape[0]
    idx = torch.empty((K, ph * pw, max_s, 4), dtype=torch.int32, device=rois.device)
    grid = torch.empty((K, 2), dtype=torch.int32, device=rois.device)
    L.check(L.lib().abr_roi_align_taps(L.ptr(rois), K, H, W, float(spatial_scale), ph, pw, sampling_ratio, max_s,
                                       L.ptr(idx), L.ptr(grid), L.stream()), "roi_align_taps")
    return idx, grid


# ----------------------------------------------------------------------------------------------- NMS
_nms_ws = {}


def nms_sorted_batched(boxes, counts, thr, max_keep, strict_gt=False):
    """boxes [N,n,4] sorted by descending score per image, counts [N] int32 -> keep [N,max_keep] int32, n_keep [N]"""
    L.require_cuda(boxes, counts)
    boxes = L.f32c(boxes)
    N, n = boxes.shape[0], boxes.shape[1]
    keep = torch.empty((N, max(max_keep, 1)), dtype=torch.int32, device=boxes.device)
    n_keep = torch.empty((N,), dtype=torch.int32, device=boxes.device)
    ws_bytes = L.lib().abr_nms_workspace_bytes(N, n)
    key = (boxes.device, L.stream())  # one workspace per stream: the source and target models may run on different streams
    ws = _nms_ws.get(key)
    if ws is None or ws.numel() < ws_bytes:
        ws = _nms_ws[key] = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=boxes.device)
    L.check(L.lib().abr_nms_sorted_batched(L.ptr(boxes), L.ptr(counts), N, n, float(thr), int(strict_gt), max_keep,
                                           L.ptr(keep), L.ptr(n_keep), L.ptr(ws), ws_bytes, L.stream()), "nms")
    return keep, n_keep


# ----------------------------------------------------------------------------------------------- losses
def ard_forward(f_src, f_tgt, gamma, layout=L.NHWC):
    """f_* [N,HW,C] (NHWC) -> (loss[4] = total, afd, pad, -), coef [N,2,HW])"""
    L.require_cuda(f_src, f_tgt)
    f_src, f_tgt = L.f32c(f_src), L.f32c(f_tgt)
    if layout == L.NHWC:
        N, HW, Ch = f_src.shape[0], f_src.shape[1:-1].numel(), f_src.shape[-1]
    else:
        N, Ch, HW = f_src.shape[0], f_src.shape[1], f_src.shape[2:].numel()
    loss = _empty((4,), f_src)
    coef = _empty((max(N, 1), 2, HW), f_src)
    L.check(L.lib().abr_ard_forward(L.ptr(f_src), L.ptr(f_tgt), N, Ch, HW, float(gamma), layout, L.ptr(coef),
                                    L.ptr(loss), L.stream()), "ard_forward")
    return loss, coef


def ard_backward(f_src, f_tgt, coef, gamma, gscale=1.0, gscale_dev=None, layout=L.NHWC):
    f_src, f_tgt = L.f32c(f_src), L.f32c(f_tgt)
    if layout == L.NHWC:
        N, HW, Ch = f_src.shape[0], f_src.shape[1:-1].numel(), f_src.shape[-1]
    else:
        N, Ch, HW = f_src.shape[0], f_src.shape[1], f_src.shape[2:].numel()
    grad = torch.empty_like(f_tgt)
    L.check(L.lib().abr_ard_backward(L.ptr(f_src), L.ptr(f_tgt), L.ptr(coef), N, Ch, HW, float(gamma), layout,
                                     float(gscale), L.ptr(gscale_dev), L.ptr(grad), L.stream()), "ard_backward")
    return grad


def smooth_l1(x, t, beta, scale=1.0, gscale=1.0, want_grad=False):
    L.require_cuda(x, t)
    x, t = L.f32c(x), L.f32c(t)
    loss = _empty((4,), x)
    grad = torch.empty_like(x) if want_grad else None
    L.check(L.lib().abr_smooth_l1(L.ptr(x), L.ptr(t), x.numel(), float(beta), float(scale), L.ptr(loss), float(gscale),
                                  L.ptr(grad), L.stream()), "smooth_l1")
    return loss, grad


def smooth_l1_rows(x, t, rows, col0, beta, scale=1.0, gscale=1.0, want_grad=False, trows=None, denom_dev=None):
    """sum over i of smoothL1(x[rows[i], col0[i]:col0[i]+4] - t[trows[i] (default rows[i]), :4]) * scale"""
    L.require_cuda(x, t, rows)
    x, t = L.f32c(x), L.f32c(t)
    loss = _empty((4,), x)
    grad = torch.zeros_like(x) if want_grad else None
    L.check(L.lib().abr_smooth_l1_rows(L.ptr(x), x.shape[1], L.ptr(t), L.ptr(rows), L.ptr(col0), L.ptr(trows), rows.numel(), float(beta),
                                       float(scale), L.ptr(denom_dev), L.ptr(loss), float(gscale), L.ptr(grad), L.stream()), "smooth_l1_rows")
    return loss, grad


def _rows2d(t):
    """2-D fp32 tensor whose rows are contiguous (column slices of a fused buffer are fine) -> (tensor, row pitch)"""
    if t.dtype != _f32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    if t.dim() != 2 or t.stride(1) != 1:
        t = t.reshape(t.shape[0], -1).contiguous()
    return t, t.stride(0) if t.shape[0] > 1 else t.shape[1]


def softmax_ce(logits, labels, inclusive=False, n_old=0, gscale=1.0, want_grad=False, grad_out=None):
    """grad_out: optional [n,>=K]-pitched view to receive d_logits (e.g. a column slice of the fused grad buffer)"""
    L.require_cuda(logits, labels)
    logits, ldz = _rows2d(logits)
    labels = labels.contiguous()
    loss = _empty((4,), logits)
    grad = None
    if want_grad:
        grad = grad_out if grad_out is not None else torch.empty((logits.shape[0], logits.shape[1]), dtype=_f32, device=logits.device)
    ldg = grad.stride(0) if grad is not None and grad.shape[0] > 1 else logits.shape[1]
    L.check(L.lib().abr_softmax_ce(L.ptr(logits), ldz, L.ptr(labels), logits.shape[0], logits.shape[1], int(inclusive), n_old,
                                   L.ptr(loss), float(gscale), L.ptr(grad), ldg, L.stream()), "softmax_ce")
    return loss, grad


def roi_distill(z_s, b_s, z_t, b_t, dist_id=True, gscale=1.0, want_grad=False, d_zt=None, d_bt=None):
    """z_s [n,K_old], b_s [n,K_old,4], z_t [n,K_all], b_t [n,K_all,4]; any of them may be a column slice of a fused
    [n,ld] buffer (rows contiguous).  d_zt / d_bt: optional pre-made (possibly sliced) gradient destinations."""
    L.require_cuda(z_s, b_s, z_t, b_t)
    n, K_old, K_all = z_t.shape[0], z_s.shape[1], z_t.shape[1]
    z_s, l0 = _rows2d(z_s)
    b_s, l1 = _rows2d(b_s.reshape(n, K_old * 4) if b_s.dim() == 3 else b_s)
    z_t, l2 = _rows2d(z_t)
    b_t, l3 = _rows2d(b_t.reshape(n, K_all * 4) if b_t.dim() == 3 else b_t)
    loss = _empty((4,), z_t)
    if want_grad:
        d_zt = d_zt if d_zt is not None else torch.empty((n, K_all), dtype=_f32, device=z_t.device)
        d_bt = d_bt if d_bt is not None else torch.empty((n, K_all * 4), dtype=_f32, device=z_t.device)
    else:
        d_zt = d_bt = None
    l4 = d_zt.stride(0) if d_zt is not None and n > 1 else K_all
    l5 = d_bt.stride(0) if d_bt is not None and n > 1 else K_all * 4
    ld = (C.c_int32 * 6)(l0, l1, l2, l3, l4, l5)
    L.check(L.lib().abr_roi_distill(L.ptr(z_s), L.ptr(b_s), L.ptr(z_t), L.ptr(b_t), n, K_old, K_all, C.cast(ld, C.c_void_p),
                                    int(dist_id), L.ptr(loss), float(gscale), L.ptr(d_zt), L.ptr(d_bt), L.stream()), "roi_distill")
    return loss, d_zt, d_bt


def bce_logits_gather(x, y, idx, gscale=1.0, want_grad=False, yidx=None, denom_dev=None):
    L.require_cuda(x, y, idx)
    x, y = L.f32c(x), L.f32c(y)
    loss = _empty((4,), x)
    grad = torch.zeros_like(x) if want_grad else None
    L.check(L.lib().abr_bce_logits_gather(L.ptr(x), L.ptr(y), L.ptr(idx), L.ptr(yidx), idx.numel(), L.ptr(denom_dev), L.ptr(loss), float(gscale),
                                          L.ptr(grad), L.stream()), "bce_logits_gather")
    return loss, grad


# ----------------------------------------------------------------------------------------------- conv
MATH_F32, MATH_BF16, MATH_BF16X6, MATH_F16X3 = 0, 1, 2, 3   # abr_conv_desc::math (include/abr_iod_hip.h)
X6_FLAG_TINY, X6_FLAG_NONFINITE = 1, 2       # ABR_X6_FLAG_*
H3_FLAG_STALE = 8                            # ABR_H3_FLAG_STALE

# ---- f16x3 (MATH_F16X3): amax words.  A tensor's amax word (include/abr_iod_hip.h, abr_conv_desc) rides on the torch.Tensor OBJECT the producing
# op returned, as `_abr_amax` = (word address, epoch, data_ptr, tensor version, allocation count): valid only while that object still names the
# same bytes (same storage address, no in-place write since) and the library's ring has not wrapped past it.  A consumer that finds no valid tag
# lets the library reduce the tensor itself (one extra pass over it: always correct).  ABR_H3_TAGS=0: never tag (every consumer reduces).
H3_TAGS = os.environ.get("ABR_H3_TAGS", "1") != "0"
_AMAX_RING = 65536
_amax_count = [0]
amax_reductions = [0, 0]   # (calls, bytes) of amax_compute: operands whose producer did not emit an amax word (bench.py reports them per step)


_AMAX_BLOCK = 512
_amax_block = {"it": iter(()), "base": 0}
_amax_lock = threading.Lock()


def amax_new():
    """a fresh amax word of the library's ring: (address, epoch).  Words come in blocks of _AMAX_BLOCK consecutive allocations
    (abr_h3_amax_alloc_block) handed out here: one library call per block instead of one per tensor (~200 per training step)."""
    try:
        c = next(_amax_block["it"])       # (atomic under the GIL: the forward pass and autograd's thread both allocate)
    except StopIteration:
        with _amax_lock:
            try:
                c = next(_amax_block["it"])
            except StopIteration:
                base, first = C.c_void_p(0), C.c_uint64(0)
                L.check(L.lib().abr_h3_amax_alloc_block(_AMAX_BLOCK, C.byref(base), C.byref(first)), "h3_amax_alloc_block")
                _amax_block["base"] = int(base.value)
                it = iter(range(int(first.value), int(first.value) + _AMAX_BLOCK))
                c = next(it)
                _amax_block["it"] = it
    _amax_count[0] += 1
    return _amax_block["base"] + (c % _AMAX_RING) * 8, (c + 1) & 0xFFFFFFFF


def amax_tag(t, word, epoch, stream=None):
    """remember that `word` (epoch) holds max |t| -- called by the op that just produced t with that word as its out_amax.  stream: the word
    was written by a reduction queued on that stream AFTER t was produced (amax_compute): only consumers on the same stream are ordered
    behind it; a producer's own word (stream=None) is as ordered as the tensor's bytes."""
    t._abr_amax = (word, epoch, t.data_ptr(), t._version, _amax_count[0], stream)
    return t


def amax_of(t):
    """(word, epoch) of t's amax if t still carries a valid tag, else (None, 0)"""
    tag = getattr(t, "_abr_amax", None)
    if (tag is not None and tag[2] == t.data_ptr() and tag[3] == t._version and _amax_count[0] - tag[4] < _AMAX_RING // 4
            and (tag[5] is None or tag[5] == L.stream())):
        return tag[0], tag[1]
    return None, 0


def amax_carry(dst, src):
    """dst is another view of exactly src's elements (permute / reshape of the whole tensor): it inherits the tag"""
    tag = getattr(src, "_abr_amax", None)
    if tag is not None and tag[2] == dst.data_ptr() and dst.numel() == src.numel():
        dst._abr_amax = (tag[0], tag[1], tag[2], dst._version, tag[4], tag[5])
    return dst


def amax_carry_bound(dst, src):
    """dst holds a subset / masked copy of src's values in fresh memory: src's amax is an upper bound of dst's"""
    w, e = amax_of(src)
    if w is not None:
        dst._abr_amax = (w, e, dst.data_ptr(), dst._version, src._abr_amax[4], src._abr_amax[5])
    return dst


def amax_drop(t):
    """t was written in place by a raw kernel: whatever tag it carried no longer describes it (nor does the "zero except the pixels of a
    strided scatter" mark: add_ fills the holes, and scale_ by inf / nan would too)"""
    if getattr(t, "_abr_amax", None) is not None:
        t._abr_amax = None
    if getattr(t, "_abr_scatter", None) is not None:
        t._abr_scatter = None
    return t


def amax_compute(t):
    """reduce max |t| into a fresh word on the current stream and tag t with it"""
    t = L.f32c(t)
    w, e = amax_new()
    st = L.stream()
    L.check(L.lib().abr_h3_amax(L.ptr(t), t.numel(), w, e, st), "h3_amax")
    amax_reductions[0] += 1
    amax_reductions[1] += t.numel() * 4
    return amax_tag(t, w, e, st)


def h3_range_stats(reset=True):
    """(operand elements more than 18 binades below their tensor's amax, operand elements inspected) by the f16x3 kernels since the last
    reset (abr_h3_range_stats).  Synchronises the current stream."""
    import ctypes
    v = (ctypes.c_uint64 * 2)(0, 0)
    L.check(L.lib().abr_h3_range_stats(ctypes.cast(v, ctypes.c_void_p), int(bool(reset)), L.stream()), "h3_range_stats")
    return int(v[0]), int(v[1])


def x6_range_flags(reset=True):
    """Range guard of the bf16x6 arithmetic (include/abr_iod_hip.h, abr_x6_range_flags): OR of X6_FLAG_* raised by any bf16x6 kernel
    since the last reset -- an operand left the domain in which the three-way bf16 split is exact (non-zero |x| < 2^-110, inf, nan).
    Synchronises the current stream."""
    import ctypes
    v = ctypes.c_uint32(0)
    L.check(L.lib().abr_x6_range_flags(ctypes.cast(ctypes.pointer(v), ctypes.c_void_p), int(bool(reset)), L.stream()), "x6_range_flags")
    return int(v.value)


class X6RangeWatch(object):
    """Non-blocking poll of the range guard for a training loop: `poll()` at the end of every step enqueues an asynchronous copy of the
    flag word into pinned host memory and returns the value of the PREVIOUS copy once its event has completed (no host stall)."""

    def __init__(self):
        self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._dev = None
        self._event = None

    def poll(self, reduce_over_ranks=False):
        """reduce_over_ranks (data-parallel training): the flag word is MAX-reduced over the process group ON THE DEVICE before it is
        copied out (4 bytes per step on the collective's stream, still no host stall), so every rank sees a trip at the same poll."""
        seen = 0
        if self._event is not None and reduce_over_ranks:
            # ranks must issue the SAME collective sequence and read the SAME poll: wait for the previous copy instead of skipping a round
            # (free in the training step: its one read-back during the forward pass already passed the previous step's end)
            self._event.synchronize()
        if self._event is not None and self._event.query():
            seen = int(self._host[0].item()) & 0xFFFFFFFF
            self._event = None
        if self._event is None:
            if reduce_over_ranks:
                import torch.distributed as dist
                if self._dev is None:
                    self._dev = torch.zeros(1, dtype=torch.int32, device="cuda")
                L.check(L.lib().abr_x6_range_flags_to_device(self._dev.data_ptr(), L.stream()), "x6_range_flags_to_device")
                dist.all_reduce(self._dev, op=dist.ReduceOp.MAX)   # TINY = 1, NONFINITE = 2, both = 3: MAX keeps "tripped" alive
                self._host.copy_(self._dev, non_blocking=True)
            else:
                L.check(L.lib().abr_x6_range_flags_async(self._host.data_ptr(), L.stream()), "x6_range_flags_async")
            self._event = torch.cuda.Event()
            self._event.record()
        return seen

    def reset(self):
        """clear the device word AND the copy in flight (its value predates the reset and would re-trip the next poll)"""
        x6_range_flags(reset=True)    # synchronises the stream: the pending copy has landed
        self._event = None
        self._host.zero_()

    def poll_h3_stats(self, reduce_over_ranks=False):
        """f16x3: enqueue an asynchronous read of (small operand elements, inspected operand elements) since the previous call (the device
        counters are reset by the read; SUM over the ranks first under data parallelism) and return the PREVIOUS read's pair, or None while
        none has landed.  No host stall."""
        seen = None
        ev = getattr(self, "_h3_event", None)
        if ev is not None and reduce_over_ranks:
            ev.synchronize()   # (every rank reads the same poll: see poll())
        if ev is not None and ev.query():
            seen = (int(self._h3_host[0].item()), int(self._h3_host[1].item()))
            self._h3_event = ev = None
        if ev is None:
            if getattr(self, "_h3_dev", None) is None:
                self._h3_dev = torch.zeros(2, dtype=torch.int64, device="cuda")
                self._h3_host = torch.zeros(2, dtype=torch.int64).pin_memory()
            L.check(L.lib().abr_h3_range_stats_to_device(self._h3_dev.data_ptr(), 1, L.stream()), "h3_range_stats_to_device")
            if reduce_over_ranks:
                import torch.distributed as dist
                dist.all_reduce(self._h3_dev, op=dist.ReduceOp.SUM)
            self._h3_host.copy_(self._h3_dev, non_blocking=True)
            self._h3_event = torch.cuda.Event()
            self._h3_event.record()
        return seen


_desc_templates = {}


def conv_desc(x_shape, w_shape, stride, pad, scale=None, bias=None, residual=None, mask=None, relu=False,
              out_hw=None, out_stride=(1, 1), math=MATH_F32):
    """abr_conv_desc of a conv.  The geometry part is built once per distinct (shapes, stride, pad, epilogue flags, math) and kept as a
    byte image: a call copies it and fills in the pointers (the field-by-field build costs 4.7 us, ~300 descriptors per training step)."""
    key = (x_shape, w_shape, stride, pad, relu, out_hw, out_stride, math)
    tmpl = _desc_templates.get(key)
    if tmpl is None:
        B, H, W, Cin = x_shape
        Cout, R, S, Cin2 = w_shape
        if Cin != Cin2:
            raise RuntimeError(f"conv: input has {Cin} channels, weight expects {Cin2}")
        Ho = (H + 2 * pad - R) // stride + 1
        Wo = (W + 2 * pad - S) // stride + 1
        d = L.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.R, d.S = B, H, W, Cin, Cout, R, S
        d.stride, d.pad, d.Ho, d.Wo = stride, pad, Ho, Wo
        d.relu = int(relu)
        if out_hw is None:
            d.out_H, d.out_W, d.out_sh, d.out_sw = Ho, Wo, 1, 1
        else:
            d.out_H, d.out_W = out_hw
            d.out_sh, d.out_sw = out_stride
        d.math = int(math)
        if len(_desc_templates) > 8192:     # (ragged batches: every image size brings its own set)
            _desc_templates.clear()
        tmpl = _desc_templates[key] = bytes(d)
    d = L.ConvDesc.from_buffer_copy(tmpl)
    if scale is not None:
        d.scale = scale.data_ptr()
    if bias is not None:
        d.bias = bias.data_ptr()
    if residual is not None:
        d.residual = residual.data_ptr()
    if mask is not None:
        d.mask = mask.data_ptr()
    return d


KEEP_WINO_V = os.environ.get("ABR_WINOGRAD_KEEP_V", "1") != "0"


def wino_v_alloc(x, w, stride, pad, math=MATH_F32):
    """Buffer for the Winograd-domain input V of conv(x, w) if both its forward pass and its weight gradient take the Winograd path
    (abr_conv_wino_v_floats), else None.  Passed to conv_forward(wino_v=) to be filled and to conv_wgrad(wino_v=) to be reused:
    the gradient then skips its own input transform (the same B^T d B over the same x)."""
    if not KEEP_WINO_V:
        return None
    d = conv_desc(x.shape, w.shape, stride, pad, math=math)
    n = int(L.lib().abr_conv_wino_v_floats(C.byref(d)))
    return torch.empty(n, dtype=_f32, device=x.device) if n > 0 else None


def conv_forward(x, w, stride=1, pad=0, scale=None, bias=None, residual=None, mask=None, relu=False,
                 out=None, out_hw=None, out_stride=(1, 1), math=MATH_F32, wino_v=None, w_planes=None, w_version=0, emit_amax=None):
    """x [B,H,W,Cin] NHWC, w [Cout,R,S,Cin] OHWI -> [B,Ho,Wo,Cout] (or scattered into `out` [B,out_H,out_W,Cout]).
    math=MATH_BF16: operands rounded to bf16 inside the kernel, bf16 MFMA, fp32 accumulate (fp32 tensors in and out).
    math=MATH_F16X3: x's amax word is taken from its tag when it has one (else the library reduces x first); emit_amax (default: under
    MATH_F16X3) makes the kernel's epilogue write the output's amax word, and the result is tagged with it -- not when the conv accumulates
    into a caller's `out` that already holds other pixels (a scattered second pass), whose amax the epilogue cannot know."""
    L.require_cuda(x, w)
    xc, w = L.f32c(x), L.f32c(w)
    d = conv_desc(xc.shape, w.shape, stride, pad, scale, bias, residual, mask, relu, out_hw, out_stride, math)
    d.wino_v = L.ptr(wino_v)
    d.w_planes = L.ptr(w_planes)   # MATH_BF16X6: the caller's own pack_weights(w) planes (else the library packs (w, w_version) itself)
    d.w_version = int(w_version)   # non-zero: the library may keep data derived from (w, w_version): Winograd-domain weights, packed bf16x3 planes
    if math == MATH_F16X3:
        aw, ae = amax_of(x) if xc is x else (None, 0)
        if aw is None and H3_TAGS:
            aw, ae = amax_of(amax_compute(xc))
        d.x_amax, d.x_amax_epoch = aw, ae
    fresh = out is None
    if out is None:
        if out_hw is not None:
            out = torch.zeros((d.B, d.out_H, d.out_W, d.Cout), dtype=_f32, device=x.device)
        else:
            out = _empty((d.B, d.Ho, d.Wo, d.Cout), xc)
    if emit_amax is None:
        emit_amax = math == MATH_F16X3
    # a strided scatter (the dgrad of a stride-2 1x1 conv: rows land on every out_sh-th pixel of a zeroed tensor) writes every non-zero element of
    # its result, so its epilogue's amax IS the tensor's -- in the fresh pass, and in a second pass that adds into exactly that tensor
    strided = d.out_sh != 1 or d.out_sw != 1
    if strided:
        covers = fresh or (out is residual and getattr(out, "_abr_scatter", None) == (out.data_ptr(), d.out_sh, d.out_sw))
    else:
        covers = fresh or out is residual
    emit_amax = bool(emit_amax) and H3_TAGS and covers
    if emit_amax:
        ow, oe = amax_new()
        d.out_amax, d.out_amax_epoch = ow, oe
    L.check(L.lib().abr_conv_forward(C.byref(d), L.ptr(xc), L.ptr(w), L.ptr(out), L.stream()), "conv_forward")
    if emit_amax:
        amax_tag(out, ow, oe)
    elif not fresh:
        amax_drop(out)     # a raw kernel moves neither data_ptr nor _version: whatever word `out` carried no longer bounds its values
    if strided and fresh:
        out._abr_scatter = (out.data_ptr(), d.out_sh, d.out_sw)   # (zero everywhere but the pixels this geometry writes)
    return out


# Main-stream timeline of the UN-PROFILED step (tools/step_marks.py): with ABR_STEP_MARKS=1 `mark(name)` records an event on the current stream at
# a handful of points of the step (a kernel trace slows the host enough to change what the device waits for; events do not).  Off: a no-op.
STEP_MARKS = os.environ.get("ABR_STEP_MARKS", "0") != "0"
_marks = []


def mark(name):
    if STEP_MARKS and torch.cuda.is_available():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _marks.append((name, ev))


def take_marks():
    """[(name, ms since the first mark)] of the marks recorded since the last call (synchronises)"""
    torch.cuda.synchronize()
    out = [(n, _marks[0][1].elapsed_time(e)) for n, e in _marks] if _marks else []
    del _marks[:]
    return out


FUSE_TAIL64 = os.environ.get("ABR_FUSE_TAIL64", "1") != "0"


def bottleneck_tail64_applies(o1, w2, w3, math):
    """the fused tail takes the 64-wide bottleneck of the frozen layer1 in the bf16x6 arithmetic (abr_conv_tail64_forward)"""
    return (FUSE_TAIL64 and math == MATH_BF16X6 and tuple(w2.shape) == (64, 3, 3, 64) and tuple(w3.shape) == (256, 1, 1, 64)
            and o1.shape[-1] == 64 and o1.numel() * 4 * 4 < 0x7FFFFFF0)


def bottleneck_tail64(o1, w2, w3, scale2, bias2, scale3, bias3, residual, w2_version, w3_version, out=None):
    """relu(bn3(conv1x1(relu(bn2(conv3x3(o1)))) + residual) in ONE launch (resnet.py:327-346 without a backward pass): o1 [B,H,W,64] NHWC,
    w2 [64,3,3,64], w3 [256,1,1,64] OHWI -> [B,H,W,256]; bit-identical to the two conv_forward calls it replaces."""
    L.require_cuda(o1, w2, w3)
    o1, w2, w3 = L.f32c(o1), L.f32c(w2), L.f32c(w3)
    d2 = conv_desc(o1.shape, w2.shape, 1, 1, scale2, bias2, None, None, True, math=MATH_BF16X6)
    B, H, W, _ = o1.shape
    d3 = conv_desc((B, H, W, 64), w3.shape, 1, 0, scale3, bias3, residual, None, True, math=MATH_BF16X6)
    d2.w_version, d3.w_version = int(w2_version), int(w3_version)
    if out is None:
        out = _empty((B, H, W, 256), o1)
    L.check(L.lib().abr_conv_tail64_forward(C.byref(d2), C.byref(d3), L.ptr(o1), L.ptr(w2), L.ptr(w3), L.ptr(out), L.stream()), "conv_tail64_forward")
    return out


def conv_prepare_weights(w, stride, pad, math, w_version):
    """Derive on the CURRENT stream what conv_forward(.., w, stride, pad, math=, w_version=) would derive from w (the Winograd-domain
    weights of a wide 3x3 conv) so that the call itself finds it cached; consumers on other streams are ordered behind it."""
    Cout, R, S, Cin = w.shape
    L.check(L.lib().abr_conv_prepare_weights(L.ptr(w), Cout, R, S, Cin, stride, pad, int(math), int(w_version), L.stream()), "conv_prepare_weights")


def conv_prepare_batch(entries):
    """entries: [(w [Cout,R,S,Cin], scale or None, wt or None, stride, pad, math, w_version)] -- what conv_prepare_weights(w, ...) derives,
    and for wt != None the dgrad copy wt = conv_dgrad_weights(w, scale) plus what conv_prepare_weights(wt, 1, R-1-pad, ...) derives from
    it, for ALL entries in three launches on the current stream (abr_conv_prepare_batch)."""
    if not entries:
        return
    arr = (L.PrepItem * len(entries))()
    keep = []
    for a, (w, scale, wt, stride, pad, math, ver) in zip(arr, entries):
        Cout, R, S, Cin = w.shape
        if scale is not None:
            scale = L.f32c(scale)
            keep.append(scale)
        a.w, a.scale, a.wt = L.ptr(w), L.ptr(scale), L.ptr(wt)
        a.Cout, a.R, a.S, a.Cin, a.stride, a.pad, a.math, a.w_version = Cout, R, S, Cin, int(stride), int(pad), int(math), int(ver)
    L.check(L.lib().abr_conv_prepare_batch(C.cast(arr, C.c_void_p), len(entries), L.stream()), "conv_prepare_batch")


class PreparedBatch(object):
    """conv_prepare_batch with its table built ONCE: the weights live in the model's flat storage, the dgrad copies and the folded FrozenBN
    scales in buffers their modules keep, so every pointer of an entry is fixed from step to step and only w_version moves.  `entries` as for
    conv_prepare_batch, with a CALLABLE in the version slot (read at every run).  `still_valid()` re-reads the pointers (a re-homed weight, a
    re-fused scale or another math mode means: rebuild)."""

    def __init__(self, entries):
        self.n = len(entries)
        self.arr = (L.PrepItem * max(self.n, 1))()
        self.keep, self.vers, self.sig_src = [], [], []
        for a, (w, scale, wt, stride, pad, math, ver) in zip(self.arr, entries):
            Cout, R, S, Cin = w.shape
            if scale is not None:
                scale = L.f32c(scale)
            self.keep.append((w, scale, wt))
            a.w, a.scale, a.wt = L.ptr(w), L.ptr(scale), L.ptr(wt)
            a.Cout, a.R, a.S, a.Cin, a.stride, a.pad, a.math = Cout, R, S, Cin, int(stride), int(pad), int(math)
            self.vers.append(ver)
        self.ptr = C.cast(self.arr, C.c_void_p)

    def run(self):
        if not self.n:
            return
        for a, ver in zip(self.arr, self.vers):
            a.w_version = int(ver())
        L.check(L.lib().abr_conv_prepare_batch(self.ptr, self.n, L.stream()), "conv_prepare_batch")


def conv_cache_clear():
    """Drop the library's per-weight derived data (Winograd-domain weights); call when parameter storage is released or rebuilt."""
    L.check(L.lib().abr_conv_cache_clear(), "conv_cache_clear")


def conv_cache_drop_range(base, nbytes):
    """Drop the derived data of the weights stored inside [base, base + nbytes) only (abr_conv_cache_drop_range)."""
    L.check(L.lib().abr_conv_cache_drop_range(int(base), int(nbytes)), "conv_cache_drop_range")


def conv_cache_bytes():
    return int(L.lib().abr_conv_cache_bytes())


def conv_wgrad(x, gy, dw, stride=1, pad=0, scale=None, math=MATH_F32, wino_v=None, amax_refs=None, stream=None):
    """dw [Cout,R,S,Cin] += scale * gy^T im2col(x) (fp32 atomics; caller zeroes dw once per step).  MATH_F16X3: the amax words of x and
    gy come from their tags (else the library reduces the operand first); amax_refs = ((word, epoch) of x, of gy) taken by the caller on
    the stream that produced the operands (conv_wgrad_async)."""
    L.require_cuda(x, gy, dw)
    xc, gyc = L.f32c(x), L.f32c(gy)
    d = conv_desc(xc.shape, dw.shape, stride, pad, scale=scale, math=math)
    d.wino_v = L.ptr(wino_v)
    if math == MATH_F16X3:
        if xc is x:
            d.x_amax, d.x_amax_epoch = amax_refs[0] if amax_refs else amax_of(x)
        if gyc is gy:
            d.gy_amax, d.gy_amax_epoch = amax_refs[1] if amax_refs else amax_of(gy)
        if H3_TAGS:   # (reduced here rather than inside the call so that a second consumer of the same tensor finds the tag)
            if not d.x_amax and wino_v is None:
                d.x_amax, d.x_amax_epoch = amax_of(amax_compute(xc))
            if not d.gy_amax:
                d.gy_amax, d.gy_amax_epoch = amax_of(amax_compute(gyc))
    L.check(L.lib().abr_conv_wgrad(C.byref(d), L.ptr(xc), L.ptr(gyc), L.ptr(dw), L.stream() if stream is None else stream), "conv_wgrad")
    return dw


# Weight gradients do not feed anything else in backward (they accumulate atomically into the flat gradient buffer), so they
# run on a SIDE HIP stream next to the dgrad chain: their workgroups fill the tail rounds of the main stream's kernels (and
# vice versa) instead of each kernel draining the chip alone.  ABR_WGRAD_STREAM=0 keeps everything on one stream.
WGRAD_SIDE_STREAM = os.environ.get("ABR_WGRAD_STREAM", "1") != "0"
_side_streams = {}
_join_pending = [False]


def h2d(values, dtype, device):
    """Small host list -> device tensor WITHOUT stalling the host: a pageable-memory copy (`torch.tensor(..., device=)`) waits for
    everything queued on the stream (measured: ~3 ms per call inside the training step, six calls per step); staging through the
    caching pinned allocator makes it a true async copy."""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


# Weight preparation off the critical stream (solver/build.py::FusedSGD.step): after the SGD kernel the data derived from the new weights
# (flipped dgrad copies, Winograd-domain weights) is rebuilt on its own stream, next to the following step's source-model forward; the
# first consumer on another stream waits for the event recorded behind it.
_prep = {"event": None, "waited": set()}


def prep_done(stream):
    ev = torch.cuda.Event()
    ev.record(stream)
    _prep["event"], _prep["waited"] = ev, set()


def prep_wait():
    """make the current stream see the last weight preparation (no-op once per stream and preparation)"""
    ev = _prep["event"]
    if ev is not None:
        raw = L.stream()      # (torch.cuda.current_stream() builds a Stream object: ~9 us, and every dgrad call comes through here)
        if raw not in _prep["waited"]:
            torch.cuda.current_stream().wait_event(ev)
            _prep["waited"].add(raw)


def side_stream(key):
    """key = device index (the wgrad stream) or (device index, tag)"""
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=key[0] if isinstance(key, tuple) else key)
    return s


def mark_overlap(on):
    """Tell the per-launch profiler (abr_prof_*) that main-stream launches now share the device with another stream."""
    L.lib().abr_prof_mark_overlap(1 if on else 0)


# Weight gradients go to this many side streams.  Two are worth 0.7 ms per step now that the small gradients are short (three: +0.5, four:
# +0.8 ms over two).  A weight's gradient always goes to the SAME stream (first-seen round-robin over the gradient buffers): a weight
# that is used twice in a step (layer4 serves the detection RoIs and the distillation RoIs) gets `dw +=` from two launches, and the
# Winograd inverse transform adds with a plain read-modify-write -- stream order keeps those two apart.
WGRAD_STREAMS = max(1, int(os.environ.get("ABR_WGRAD_STREAMS", "2")))
_wg_owner = {}
_wg_keep = {}    # {raw handle of the stream that produced them: [operands of the weight gradients queued on side streams since that stream's last join]}
_wg_lock = threading.Lock()


def _wg_hold(main, item):
    lst = _wg_keep.get(main)
    if lst is None:
        with _wg_lock:
            lst = _wg_keep.setdefault(main, [])
    lst.append(item)


_wgrad_keys = {}


def _wgrad_stream(dev, i):
    key = _wgrad_keys.get((dev, i))
    if key is None:
        key = _wgrad_keys[(dev, i)] = dev if i == 0 else (dev, "wgrad%d" % i)
    return side_stream(key)


def join_side_stream():
    """Make the current stream wait for everything queued on the side stream(s) (queued automatically as an end-of-backward
    callback by conv_wgrad_async; FusedSGD.step calls it again before touching the gradients)."""
    if _join_pending[0]:
        L.lib().abr_prof_mark_overlap(0)
    _join_pending[0] = False
    if torch.cuda.is_available():
        dev = torch.cuda.current_device()
        for i in range(WGRAD_STREAMS):
            s = _side_streams.get(dev if i == 0 else (dev, "wgrad%d" % i))
            if s is not None:
                torch.cuda.current_stream().wait_stream(s)
    # (after the waits: see conv_wgrad_async.)  Only the operands that belong to THIS stream go: another host thread's backward pass on another
    # stream has not waited for its side kernels yet, and its blocks would return to a pool whose stream is not ordered behind them.
    lst = _wg_keep.get(L.stream()) if torch.cuda.is_available() else None
    if lst:
        del lst[:]


def conv_wgrad_async(x, gy, dw, stride=1, pad=0, scale=None, math=MATH_F32, wino_v=None):
    """conv_wgrad on the side stream.  Only for use inside an autograd backward (the join is an engine callback)."""
    if not WGRAD_SIDE_STREAM:
        return conv_wgrad(x, gy, dw, stride, pad, scale, math, wino_v)
    owner = _wg_owner.get(dw.data_ptr())
    if owner is None:
        owner = _wg_owner[dw.data_ptr()] = len(_wg_owner) % WGRAD_STREAMS
    side = _wgrad_stream(x.device.index, owner)
    if not _join_pending[0]:
        _join_pending[0] = True
        L.lib().abr_prof_mark_overlap(1)  # from here to the join, main-stream launches share the device with the side stream
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
    # The operands are made contiguous, and their amax words reduced, HERE -- on the stream that produced them and before the side stream's wait is
    # recorded: a copy or a reduction issued after that event would not be ordered before the side stream's kernel.  The copies are what the
    # kernel reads, so they are what is kept alive until the join.
    x, gy = L.f32c(x), L.f32c(gy)
    refs = None
    if math == MATH_F16X3 and H3_TAGS:
        # operands without an amax word get one on this stream: the dgrad that follows here shares gy's, and the side stream (ordered behind
        # this point by the wait below) is handed both -- a word reduced on the side stream would not be ordered before this stream's later readers
        if wino_v is None and amax_of(x)[0] is None:
            amax_compute(x)
        if amax_of(gy)[0] is None:
            amax_compute(gy)
        refs = (amax_of(x), amax_of(gy))
    # x, gy (and the zeroed gradient buffer) are produced on the current stream: the side stream is ordered behind it, and the launch names the
    # side stream itself (side.wait_stream(cur) + `with torch.cuda.stream(side)` cost ~30 us of Python per gradient, ~60 gradients per step)
    raw = side.cuda_stream
    main = L.stream()
    L.check(L.lib().abr_stream_wait_stream(raw, main), "stream_wait_stream")
    conv_wgrad(x, gy, dw, stride, pad, scale, math, wino_v, amax_refs=refs, stream=raw)
    # The caching allocator must not recycle the operands for the main stream while the side kernel still reads them: they are kept alive until
    # the join (join_side_stream: the main stream waits for the side streams, THEN the references go), so their blocks return to the main
    # stream's pool ordered behind the gradients.  (tensor.record_stream did the same at ~2 us per call plus an event per block at free time,
    # three tensors per gradient.)
    _wg_hold(main, (x, gy, wino_v))
    return dw


# ---- one library call for a conv's whole backward pass (round 5): the weight gradient on its side stream and the input gradient on the current
# stream are the per-conv calls above (conv_wgrad_async + conv_forward on the dgrad copy) -- ~40 us of interpreter / binding time together,
# ~40 convs per step.  The three-entry abr_conv_run table [side stream waits for this one, weight gradient, input gradient] of a (weight, shapes,
# geometry) is kept with everything constant filled in; a call writes this step's pointers and amax words.  Same kernels, same arguments,
# same order on each stream: bit-identical to the two calls.
_bwd_plans = {}


def conv_backward(x, gy, dw, wt, stride, pad, scale=None, math=MATH_F32, w_version=0, wino_v=None, dgrad=True, mask=None, residual=None,
                  out=None, out_hw=None, out_stride=(1, 1)):
    """dw += scale * gy^T im2col(x) on dw's side stream, and (dgrad) returns dL/dx = conv(gy, wt, stride 1, pad R-1-pad) with the epilogue
    options of conv_forward (mask / residual / out / the strided scatter of a stride-2 conv).  x, gy contiguous NHWC fp32; wt = the flipped,
    scaled dgrad copy [Cin,R,S,Cout] of the weight whose gradient buffer dw is.  Only inside an autograd backward (see conv_wgrad_async)."""
    if not (WGRAD_SIDE_STREAM and H3_TAGS and x.is_contiguous() and gy.is_contiguous() and x.dtype == _f32 and gy.dtype == _f32):
        conv_wgrad_async(x, gy, dw, stride, pad, scale=scale, math=math, wino_v=wino_v)
        if not dgrad:
            return None
        return conv_forward(gy, wt, 1, dw.shape[1] - 1 - pad, mask=mask, residual=residual, out=out, out_hw=out_hw, out_stride=out_stride,
                            math=math, w_version=w_version)
    key = (dw.data_ptr(), x.shape, gy.shape, stride, pad, math, dgrad, out_hw, out_stride, wino_v is not None)
    plan = _bwd_plans.get(key)
    wtp = wt.data_ptr() if dgrad else 0
    scp = scale.data_ptr() if scale is not None else 0
    if plan is None or plan[2] != wtp or plan[3] != scp:
        if len(_bwd_plans) > 4096:
            _bwd_plans.clear()
        arr = (L.ConvOp * 3)()
        arr[0].kind = L.OP_STREAM_WAIT
        arr[1].kind = L.OP_WGRAD
        arr[1].desc = conv_desc(x.shape, dw.shape, stride, pad, scale=scale, math=math)
        arr[1].out = dw.data_ptr()
        if dgrad:
            arr[2].kind = L.OP_FORWARD
            arr[2].desc = conv_desc(gy.shape, wt.shape, 1, dw.shape[1] - 1 - pad, out_hw=out_hw, out_stride=out_stride, math=math)
            arr[2].b = wtp
        owner = _wg_owner.get(dw.data_ptr())
        if owner is None:
            owner = _wg_owner[dw.data_ptr()] = len(_wg_owner) % WGRAD_STREAMS
        plan = _bwd_plans[key] = (arr, C.cast(arr, C.c_void_p), wtp, scp, owner, (scale, dw), threading.Lock())
    arr, ptr, _, _, owner, _, lock = plan
    with lock:     # (the table is shared by every host thread that runs this conv's backward pass; abr_conv_run releases the GIL)
        return _conv_backward_run(arr, ptr, owner, x, gy, dw, math, wino_v, dgrad, mask, residual, out, out_hw, w_version)


def _conv_backward_run(arr, ptr, owner, x, gy, dw, math, wino_v, dgrad, mask, residual, out, out_hw, w_version):
    side = _wgrad_stream(x.device.index, owner).cuda_stream
    if not _join_pending[0]:
        _join_pending[0] = True
        L.lib().abr_prof_mark_overlap(1)
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
    main = L.stream()
    h3 = math == MATH_F16X3
    xp, gp = x.data_ptr(), gy.data_ptr()
    a0, a1 = arr[0], arr[1]
    a0.stream, a0.other = side, main
    a1.a, a1.b, a1.stream = xp, gp, side
    d1 = a1.desc
    d1.wino_v = wino_v.data_ptr() if wino_v is not None else None
    if h3:
        # operands without an amax word get one here, on the stream that produced them (see conv_wgrad_async)
        gw, ge = amax_of(gy)
        if gw is None:
            gw, ge = amax_of(amax_compute(gy))
        if wino_v is None:
            xw, xe = amax_of(x)
            if xw is None:
                xw, xe = amax_of(amax_compute(x))
        else:
            xw, xe = amax_of(x)     # (the kept V carries its own word inside the library; x's, when there, is passed as the per-conv call does)
        d1.x_amax, d1.x_amax_epoch, d1.gy_amax, d1.gy_amax_epoch = xw, (xe if xw is not None else 0), gw, ge
    n_ops = 2
    res = None
    if dgrad:
        n_ops = 3
        a2 = arr[2]
        d2 = a2.desc
        fresh = out is None
        if fresh:
            if out_hw is not None:
                out = torch.zeros((d2.B, d2.out_H, d2.out_W, d2.Cout), dtype=_f32, device=x.device)
            else:
                out = torch.empty((d2.B, d2.Ho, d2.Wo, d2.Cout), dtype=_f32, device=x.device)
        a2.a, a2.out, a2.stream = gp, out.data_ptr(), main
        d2.residual = residual.data_ptr() if residual is not None else None
        d2.mask = mask.data_ptr() if mask is not None else None
        d2.w_version = int(w_version)
        emit = False
        if h3:
            strided = d2.out_sh != 1 or d2.out_sw != 1
            if strided:
                emit = fresh or (out is residual and getattr(out, "_abr_scatter", None) == (out.data_ptr(), d2.out_sh, d2.out_sw))
            else:
                emit = fresh or out is residual
            d2.x_amax, d2.x_amax_epoch = gw, ge
            if emit:
                ow, oe = amax_new()
                d2.out_amax, d2.out_amax_epoch = ow, oe
            else:
                d2.out_amax, d2.out_amax_epoch = None, 0
        res = out
    L.check(L.lib().abr_conv_run(ptr, n_ops), "conv_run (conv backward)")
    if dgrad:
        if emit:
            amax_tag(out, ow, oe)
        elif not fresh:
            amax_drop(out)
        if fresh and out_hw is not None and (d2.out_sh != 1 or d2.out_sw != 1):
            out._abr_scatter = (out.data_ptr(), d2.out_sh, d2.out_sw)
    _wg_hold(main, (x, gy, wino_v))
    return res


def pack_weights(w, out=None):
    """w [Cout,R,S,Cin] (or any [rows, K] fp32 matrix, K % 16 == 0) -> its fragment-packed exact bf16x3 planes (uint8 buffer of
    abr_conv_packed_bytes bytes) for conv_forward(..., math=MATH_BF16X6, w_planes=): the weights-direct bf16x6 kernel loads weight
    fragments straight from them (include/abr_iod_hip.h, abr_conv_pack_weights)."""
    L.require_cuda(w)
    w = L.f32c(w)
    rows, K = w.shape[0], w.numel() // w.shape[0]
    n = int(L.lib().abr_conv_packed_bytes(rows, K))
    if n <= 0:
        raise RuntimeError("pack_weights: K = {} is not a positive multiple of 16".format(K))
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=w.device)
    L.check(L.lib().abr_conv_pack_weights(L.ptr(w), rows, K, L.ptr(out), L.stream()), "conv_pack_weights")
    return out


def conv_dgrad_weights(w, scale=None, out=None):
    """w [Cout,R,S,Cin] -> [Cin,R,S,Cout] flipped, scaled by scale[Cout]"""
    w = L.f32c(w)
    Cout, R, S, Cin = w.shape
    if out is None:
        out = _empty((Cin, R, S, Cout), w)
    L.check(L.lib().abr_conv_dgrad_weights(L.ptr(w), L.ptr(scale), Cout, R, S, Cin, L.ptr(out), L.stream()), "dgrad_weights")
    return out


def bias_grad(gy, db):
    gy = L.f32c(gy)
    Ch = gy.shape[-1]
    L.check(L.lib().abr_bias_grad(L.ptr(gy), gy.numel() // Ch, Ch, L.ptr(db), L.stream()), "bias_grad")
    return db


# ----------------------------------------------------------------------------------------------- pointwise
def nchw_to_nhwc(x, cpad=None):
    x = L.f32c(x)
    B, Ch, H, W = x.shape
    cpad = cpad or Ch
    out = _empty((B, H, W, cpad), x)
    L.check(L.lib().abr_nchw_to_nhwc_pad(L.ptr(x), B, Ch, H, W, cpad, L.ptr(out), L.stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x):
    x = L.f32c(x)
    B, H, W, Ch = x.shape
    out = _empty((B, Ch, H, W), x)
    L.check(L.lib().abr_nhwc_to_nchw(L.ptr(x), B, Ch, H, W, L.ptr(out), L.stream()), "nhwc_to_nchw")
    return out


def maxpool3x3s2(x):
    x = L.f32c(x)
    B, H, W, Ch = x.shape
    out = _empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Ch), x)
    L.check(L.lib().abr_maxpool3x3s2(L.ptr(x), B, H, W, Ch, L.ptr(out), L.stream()), "maxpool")
    return amax_carry_bound(out, x)   # (every input pixel lies in some window: max |pooled| <= max |x|, equal for a ReLU's output)


def avgpool_forward(x):
    """x [N,h,w,C] -> [N,C]"""
    x = L.f32c(x)
    N, Ch = x.shape[0], x.shape[-1]
    HW = x.shape[1:-1].numel()
    out = _empty((N, Ch), x)
    L.check(L.lib().abr_avgpool_forward(L.ptr(x), N, HW, Ch, L.ptr(out), L.stream()), "avgpool_forward")
    return out


def avgpool_backward(g, shape, relu_of=None, emit_amax=False):
    """g [N, C] -> [N, H, W, C] / (H*W); relu_of = the pooled tensor itself when it is a ReLU's output: that ReLU's backward rides along
    (gx = relu_of > 0 ? g / HW : 0) and the caller's gradient is already masked"""
    g = L.f32c(g)
    N, Ch = g.shape
    gx = _empty(shape, g)
    HW = gx.numel() // (N * Ch) if N else 1
    if relu_of is not None:
        assert tuple(relu_of.shape) == tuple(shape) and relu_of.is_contiguous()
        if emit_amax and H3_TAGS:
            aw, ae = amax_new()
            L.check(L.lib().abr_avgpool_relu_backward_amax(L.ptr(g), L.ptr(relu_of), N, HW, Ch, L.ptr(gx), aw, ae, L.stream()), "avgpool_relu_backward")
            amax_tag(gx, aw, ae)
        else:
            L.check(L.lib().abr_avgpool_relu_backward(L.ptr(g), L.ptr(relu_of), N, HW, Ch, L.ptr(gx), L.stream()), "avgpool_relu_backward")
    else:
        L.check(L.lib().abr_avgpool_backward(L.ptr(g), N, HW, Ch, L.ptr(gx), L.stream()), "avgpool_backward")
    return gx


def channel_mean(x):
    """x [..., C] contiguous -> mean over the last (channel) axis, shape x.shape[:-1]"""
    L.require_cuda(x)
    x = L.f32c(x)
    out = torch.empty(x.shape[:-1], dtype=_f32, device=x.device)
    L.check(L.lib().abr_channel_mean(L.ptr(x), out.numel(), x.shape[-1], L.ptr(out), L.stream()), "channel_mean")
    return out


def relu_backward(g, y, inplace=False):
    """g * (y > 0); a new tensor unless inplace (one pass either way: no clone + in-place pair)"""
    g = L.f32c(g)
    out = g if inplace else torch.empty_like(g)
    L.check(L.lib().abr_relu_backward(L.ptr(g), L.ptr(y), g.numel(), L.ptr(out), L.stream()), "relu_backward")
    if not inplace:
        amax_carry_bound(out, g)   # (masking only removes elements: g's amax stays an upper bound, which is all a split scale needs)
    return out


def relu_backward_(g, y):
    return relu_backward(g, y, inplace=True)


def add_(a, b):
    L.check(L.lib().abr_add_inplace(L.ptr(a), L.ptr(b), a.numel(), L.stream()), "add_inplace")
    amax_drop(a)
    return a


class _LossSum(torch.autograd.Function):
    """total = sum_i w_i * loss_i (plus the two group sums, detached) in ONE launch, and one launch for all the terms' gradients: the
    reference's chain of python adds / muls on device scalars (train_incremental.py:91,101-128) is ~10 tiny kernels forward and as many
    autograd nodes backward"""

    @staticmethod
    def forward(ctx, weights, groups, *terms):
        n = len(terms)
        ts = [t.detach().reshape(()).contiguous() for t in terms]
        total = torch.empty((), dtype=_f32, device=ts[0].device)
        out = torch.empty((3,), dtype=_f32, device=ts[0].device)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        w = (C.c_float * n)(*[float(v) for v in weights])
        g = (C.c_int32 * n)(*[int(v) for v in groups])
        L.check(L.lib().abr_loss_sum(C.cast(ptrs, C.c_void_p), C.cast(w, C.c_void_p), C.cast(g, C.c_void_p), n, L.ptr(total), L.ptr(out), L.stream()), "loss_sum")
        ctx.weights, ctx.n = [float(v) for v in weights], n
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        n = ctx.n
        grads = torch.empty((n,), dtype=_f32, device=g_total.device)
        w = (C.c_float * n)(*ctx.weights)
        g_c = g_total.contiguous()   # (a local: a temporary inside the argument list would be released before the launch)
        L.check(L.lib().abr_loss_sum_backward(C.cast(w, C.c_void_p), n, L.ptr(g_c), L.ptr(grads), L.stream()), "loss_sum_backward")
        return (None, None) + tuple(grads[i] for i in range(n))


def loss_sum(terms, weights, groups):
    """-> (total (differentiable 0-dim), parts [3] = (total, group-0 sum, group-1 sum), detached)"""
    return _LossSum.apply(tuple(weights), tuple(groups), *terms)


def scale_(x, s=1.0, s_dev=None):
    """x *= s * s_dev[0] (s_dev: device scalar, e.g. the upstream gradient of a loss) -- no host sync"""
    if x is None:
        return None
    L.check(L.lib().abr_scale_inplace(L.ptr(x), x.numel(), float(s), L.ptr(s_dev), L.stream()), "scale_inplace")
    amax_drop(x)
    return x


# ----------------------------------------------------------------------------------------------- RPN glue
def grid_anchors(cell, H, W, stride, img_h, img_w, straddle=0):
    A = cell.shape[0]
    out = torch.empty((H * W * A, 4), dtype=_f32, device=cell.device)
    vis = torch.empty((H * W * A,), dtype=torch.uint8, device=cell.device)
    L.check(L.lib().abr_grid_anchors(L.ptr(cell), A, H, W, stride, img_h, img_w, straddle, L.ptr(out), L.ptr(vis),
                                     L.stream()), "grid_anchors")
    return out, vis


def topk_sigmoid(y, A, k):
    """y [N, nloc, ld] fp32 contiguous (columns 0..A-1 of every row are logits of anchors loc*A+a) -> (scores [N,k], idx [N,k] int64):
    the k largest sigmoid(logit) per image, sorted descending, ties by ascending anchor index."""
    y = L.f32c(y)
    N, nloc, ld = y.shape
    scores = torch.empty((N, k), dtype=_f32, device=y.device)
    idx = torch.empty((N, k), dtype=torch.int64, device=y.device)
    L.check(L.lib().abr_topk_sigmoid(L.ptr(y), nloc * ld, N, nloc * A, A, ld, k, L.ptr(scores), L.ptr(idx), L.stream()), "topk_sigmoid")
    return scores, idx


def box_encode_rows(gt, ex, weights, out=None):
    gt, ex = L.f32c(gt), L.f32c(ex)
    if out is None:
        out = torch.empty_like(ex)
    L.check(L.lib().abr_box_encode(L.ptr(gt), L.ptr(ex), ex.shape[0], *[float(v) for v in weights], L.ptr(out), L.stream()), "box_encode")
    return out


def rpn_decode_clip(reg, reg_col0, anchors, idx, img_hw, weights=(1.0, 1.0, 1.0, 1.0), A=1, clip=True):
    """reg [N,n_anchor/A,stride] ; idx [N,k] int64 anchor ids ; img_hw [N,2] int32 -> [N,k,4]"""
    N, nloc, rs = reg.shape
    n_anchor = nloc * A
    k = idx.shape[1]
    out = torch.empty((N, k, 4), dtype=_f32, device=reg.device)
    L.check(L.lib().abr_rpn_decode_clip(L.ptr(reg), rs, reg_col0, A, L.ptr(anchors), L.ptr(idx), N, n_anchor, k,
                                        L.ptr(img_hw) if clip else None, *[float(v) for v in weights], L.ptr(out), L.stream()), "rpn_decode_clip")
    return out


def match_encode(boxes, gt, gt_labels, vis, hi, lo, allow_low_quality, weights, rpn_labels):
    """-> matched int64 [n], labels (fp32 RPN / int64 head) [n], reg_targets [n,4]"""
    boxes, gt = L.f32c(boxes), L.f32c(gt)
    n, G = boxes.shape[0], gt.shape[0]
    dev = boxes.device
    matched = torch.empty((n,), dtype=torch.int64, device=dev)
    lab_f = torch.empty((n,), dtype=_f32, device=dev) if rpn_labels else None
    lab_i = None if rpn_labels else torch.empty((n,), dtype=torch.int64, device=dev)
    tgt = torch.empty((n, 4), dtype=_f32, device=dev)
    ws = torch.empty((max(G, 1),), dtype=torch.int32, device=dev)
    L.check(L.lib().abr_match_encode(L.ptr(boxes), n, L.ptr(gt), L.ptr(gt_labels), G, L.ptr(vis), float(hi), float(lo),
                                     int(allow_low_quality), *[float(v) for v in weights], L.ptr(matched), L.ptr(lab_f),
                                     L.ptr(lab_i), L.ptr(tgt), L.ptr(ws), ws.numel() * 4, L.stream()), "match_encode")
    return matched, (lab_f if rpn_labels else lab_i), tgt


_sample_calls = [0]
_ptr_tables = {}


_small_tables = {}


def _evict_oldest(cache, limit):
    """Drop the oldest half of a table cache that has reached `limit` entries.  NOT `cache.clear()`: a caller builds several tables for ONE
    launch, and clearing while it builds the second would free the first -- whose memory the second upload then reuses before the kernel
    has read it (seen once as a training step with garbage ground-truth boxes).  Callers also keep every table in a local until the
    launch is enqueued: after that a free is ordered behind the kernel on the same stream."""
    if len(cache) >= limit:
        for k in list(cache.keys())[:limit // 2]:
            del cache[k]


def _small_table(values, dtype, dev):
    """device copy of a short list of host integers, cached while the same list repeats (per-image GT counts, ...)"""
    key = (tuple(values), dtype, torch.device(dev))
    t = _small_tables.get(key)
    if t is None:
        _evict_oldest(_small_tables, 256)
        t = _small_tables[key] = h2d(list(values), dtype, dev)
    return t


def _pointer_table(tensors, dev):
    """device int64 array of the tensors' device addresses (one pinned asynchronous upload; cached while the addresses repeat)"""
    key = (tuple(t.data_ptr() for t in tensors), torch.device(dev))
    tab = _ptr_tables.get(key)
    if tab is None:
        _evict_oldest(_ptr_tables, 64)
        tab = _ptr_tables[key] = h2d(list(key[0]), torch.int64, dev)
    return tab


def roi_head_targets(props, scores, keep, n_keep, gt_boxes, gt_labels, hi, lo, weights, batch_size, max_pos, num_classes, cls_agnostic=False,
                     seed=None):
    """Box-head training targets for the whole batch, on device, no host round trip (include/abr_iod_hip.h, abr_roi_head_targets).
    props [N,k,4] / scores [N,k] / keep [N,post] int32 / n_keep [N] int32: the RPN selector's raw output; gt_boxes / gt_labels: per-image
    tensors.  Returns a dict of fixed-size tensors (batch_size rows per image)."""
    N, k_pre, _ = props.shape
    post = keep.shape[1]
    dev = props.device
    gtb = [L.f32c(b) for b in gt_boxes]
    gtl = [l.to(torch.int64).contiguous() for l in gt_labels]
    n_gt = [int(b.shape[0]) for b in gtb]
    if min(n_gt) == 0:   # matcher.py:53-57
        raise ValueError("No ground-truth boxes available for one of the images during training")
    g_max = max(n_gt)
    Pmax = post + g_max
    R = batch_size
    i64, i32 = torch.int64, torch.int32
    out = dict(
        cand=torch.empty((N, Pmax, 4), dtype=_f32, device=dev), labels_all=torch.empty((N, Pmax), dtype=i64, device=dev),
        regt_all=torch.empty((N, Pmax, 4), dtype=_f32, device=dev), obj_all=torch.empty((N, Pmax), dtype=_f32, device=dev),
        n_cand=torch.empty((N,), dtype=i32, device=dev), pos_idx=torch.empty((N, max(max_pos, 1)), dtype=i64, device=dev),
        neg_idx=torch.empty((N, R), dtype=i64, device=dev), counts=torch.empty((N, 2), dtype=i32, device=dev),
        rois=torch.empty((N * R, 5), dtype=_f32, device=dev), labels=torch.empty((N * R,), dtype=i64, device=dev),
        reg_targets=torch.empty((N * R, 4), dtype=_f32, device=dev), sampled_idx=torch.empty((N, R), dtype=i64, device=dev),
        n_valid=torch.empty((1,), dtype=_f32, device=dev), obj=torch.empty((N * R,), dtype=_f32, device=dev),
        pos_rows=torch.empty((N * R,), dtype=i64, device=dev), col0=torch.empty((N * R,), dtype=i64, device=dev),
        n_gt=n_gt, g_max=g_max, keepalive=(gtb, gtl))
    if seed is None:
        _sample_calls[0] += 1
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
    ngt_dev = _small_table(n_gt, i32, dev)
    gtb_tab, gtl_tab = _pointer_table(gtb, dev), _pointer_table(gtl, dev)   # (held in locals until the launch is enqueued: _evict_oldest)
    L.check(L.lib().abr_roi_head_targets(
        L.ptr(props), L.ptr(keep), L.ptr(n_keep), N, k_pre, post, L.ptr(gtb_tab), L.ptr(gtl_tab), L.ptr(ngt_dev),
        g_max, float(hi), float(lo), *[float(v) for v in weights], R, max_pos, seed, L.ptr(out["cand"]), L.ptr(out["labels_all"]),
        L.ptr(out["regt_all"]), L.ptr(out["n_cand"]), L.ptr(out["pos_idx"]), L.ptr(out["neg_idx"]), L.ptr(out["counts"]), L.ptr(out["rois"]),
        L.ptr(out["labels"]), L.ptr(out["reg_targets"]), L.ptr(out["sampled_idx"]), L.ptr(out["n_valid"]), L.ptr(scores), L.ptr(out["obj_all"]),
        L.ptr(out["obj"]), L.ptr(out["pos_rows"]), L.ptr(out["col0"]), int(num_classes), int(bool(cls_agnostic)), L.stream()), "roi_head_targets")
    return out


def rpn_targets_batched(anchors, vis_list, gt_boxes, hi, lo, weights):
    """RPN labels [N,n] (fp32 1 / 0 / -1) and regression targets [N,n,4] for the whole batch in two launches (abr_rpn_targets_batched);
    anchors [n,4] shared, vis_list / gt_boxes per image"""
    n, N, dev = anchors.shape[0], len(gt_boxes), anchors.device
    gtb = [L.f32c(b) for b in gt_boxes]
    n_gt = [int(b.shape[0]) for b in gtb]
    if min(n_gt) == 0:   # matcher.py:53-57
        raise ValueError("No ground-truth boxes available for one of the images during training")
    g_max = max(n_gt)
    labels = torch.empty((N, n), dtype=_f32, device=dev)
    tgt = torch.empty((N, n, 4), dtype=_f32, device=dev)
    ws = torch.empty((N * g_max,), dtype=torch.int32, device=dev)
    gtb_tab, ngt_tab, vis_tab = _pointer_table(gtb, dev), _small_table(n_gt, torch.int32, dev), _pointer_table(vis_list, dev)
    L.check(L.lib().abr_rpn_targets_batched(L.ptr(anchors), n, N, L.ptr(gtb_tab), L.ptr(ngt_tab), g_max,
                                            L.ptr(vis_tab), float(hi), float(lo), *[float(v) for v in weights], L.ptr(labels),
                                            L.ptr(tgt), L.ptr(ws), ws.numel() * 4, L.stream()), "rpn_targets_batched")
    return labels, tgt, (gtb,)


def rpn_loss_indices(pos, neg, counts, A, Cf):
    """-> (samp [n_pos+n_neg], obj_flat, pos_row [n_pos], pos_col [n_pos], denom [1] fp32) from the sampler's padded lists (one launch)"""
    pos, neg = pos.reshape(-1), neg.reshape(-1)
    n_pos, n_neg, dev = pos.numel(), neg.numel(), pos.device
    samp = torch.empty((n_pos + n_neg,), dtype=torch.int64, device=dev)
    obj_flat = torch.empty_like(samp)
    pos_row = torch.empty((n_pos,), dtype=torch.int64, device=dev)
    pos_col = torch.empty_like(pos_row)
    denom = torch.empty((1,), dtype=_f32, device=dev)
    L.check(L.lib().abr_rpn_loss_indices(L.ptr(pos), n_pos, L.ptr(neg), n_neg, L.ptr(counts), counts.shape[0], int(A), int(Cf), L.ptr(samp),
                                         L.ptr(obj_flat), L.ptr(pos_row), L.ptr(pos_col), L.ptr(denom), L.stream()), "rpn_loss_indices")
    return samp, obj_flat, pos_row, pos_col, denom


def gather_proposals(props, scores, keep, picks, P):
    """rois [N*P,5] = (i, props[i, keep[i, picks[i*P+j]]]), obj [N*P]: P picked rows per image of the post-NMS lists"""
    N, k_pre, _ = props.shape
    rois = torch.empty((N * P, 5), dtype=_f32, device=props.device)
    obj = torch.empty((N * P,), dtype=_f32, device=props.device)
    L.check(L.lib().abr_gather_proposals(L.ptr(props), L.ptr(scores), L.ptr(keep), N, k_pre, keep.shape[1], L.ptr(picks), P, L.ptr(rois), L.ptr(obj),
                                         L.stream()), "gather_proposals")
    return rois, obj


def sample_pos_neg(labels, batch_size, max_pos, index_offset_per_image=0, seed=None):
    """labels [N,n] (fp32 or int64, rows contiguous) -> pos_idx [N,max_pos] int64, neg_idx [N,batch_size] int64 (-1 padded,
    ascending), counts [N,2] int32 -- all on device, no sync."""
    if labels.dim() == 1:
        labels = labels.view(1, -1)
    N, n = labels.shape
    dev = labels.device
    pos = torch.empty((N, max(max_pos, 1)), dtype=torch.int64, device=dev)
    neg = torch.empty((N, batch_size), dtype=torch.int64, device=dev)
    counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
    if seed is None:
        # one draw from torch's default CPU generator per launch (host-side, no device sync): torch.manual_seed() therefore replays the
        # sampler's choices, as it replays torch.randperm in the reference (balanced_positive_negative_sampler.py:44-49)
        _sample_calls[0] += 1
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF
    is64 = labels.dtype == torch.int64
    if not is64 and labels.dtype != _f32:
        raise RuntimeError("sample_pos_neg: labels must be float32 or int64")
    L.check(L.lib().abr_sample_pos_neg(L.ptr(labels), int(is64), N, n, labels.stride(0), batch_size, max_pos, seed, 0,
                                       index_offset_per_image, L.ptr(pos), L.ptr(neg), L.ptr(counts), L.stream()), "sample_pos_neg")
    return pos, neg, counts


# ----------------------------------------------------------------------------------------------- optimiser
def sgd_momentum_(p, g, m, seg_end, lr, wd, momentum, gscale=1.0, first_step=False):
    L.check(L.lib().abr_sgd_momentum(L.ptr(p), L.ptr(g), L.ptr(m), p.numel(), L.ptr(seg_end), L.ptr(lr), L.ptr(wd),
                                     seg_end.numel(), float(momentum), float(gscale), int(first_step), L.stream()), "sgd")


# ----------------------------------------------------------------------------------------------- test-time detections (F4)
def det_softmax_decode(logits, deltas, rois, C, img_hw, weights, cls_agnostic=False):
    """logits [K,>=C] / deltas [K,4C] (row-strided views of the fused predictor output are fine), rois [K,5], img_hw [N,2] int32
    -> prob [K,C], boxes [K,C,4]   (roi_heads/box_head/inference.py:55-70)"""
    L.require_cuda(logits, deltas, rois, img_hw)
    K = logits.shape[0]
    if logits.stride(-1) != 1 or logits.dtype != _f32:
        logits = L.f32c(logits)
    if deltas.stride(-1) != 1 or deltas.dtype != _f32:
        deltas = L.f32c(deltas)
    rois = L.f32c(rois)
    prob = torch.empty((K, C), dtype=_f32, device=logits.device)
    boxes = torch.empty((K, C, 4), dtype=_f32, device=logits.device)
    ld_l = logits.stride(0) if K > 1 else max(logits.shape[1], C)
    ld_d = deltas.stride(0) if K > 1 else max(deltas.shape[1], 4)
    agn = deltas.shape[1] - 4 if cls_agnostic else -1
    L.check(L.lib().abr_det_softmax_decode(L.ptr(logits), ld_l, L.ptr(deltas), ld_d, 0, agn, L.ptr(rois), K, C, L.ptr(img_hw),
                                           *[float(w) for w in weights], L.ptr(prob), L.ptr(boxes), L.stream()), "det_softmax_decode")
    return prob, boxes


def det_select(prob, boxes, row_off, N, C, r_max, score_thresh, nms_thresh, detections_per_img, background=True):
    """filter_results for the batch (inference.py:106-151) -> out_boxes [N,cap,4], out_scores [N,cap], out_labels [N,cap] int64,
    out_count [N] int32, bg_boxes [N,r_max,4], bg_scores [N,r_max], bg_count [N]"""
    L.require_cuda(prob, boxes, row_off)
    dev = prob.device
    cap = (C - 1) * r_max
    ob = torch.empty((N, cap, 4), dtype=_f32, device=dev)
    os_ = torch.empty((N, cap), dtype=_f32, device=dev)
    ol = torch.empty((N, cap), dtype=torch.int64, device=dev)
    oc = torch.empty((N,), dtype=torch.int32, device=dev)
    bb = torch.empty((N, r_max, 4), dtype=_f32, device=dev) if background else None
    bs = torch.empty((N, r_max), dtype=_f32, device=dev) if background else None
    bc = torch.empty((N,), dtype=torch.int32, device=dev) if background else None
    ws_bytes = L.lib().abr_det_select_workspace_bytes(N, C, r_max)
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=dev)
    L.check(L.lib().abr_det_select(L.ptr(L.f32c(prob)), L.ptr(L.f32c(boxes)), L.ptr(row_off), N, C, r_max, float(score_thresh),
                                   float(nms_thresh), int(detections_per_img), cap, L.ptr(ob), L.ptr(os_), L.ptr(ol), L.ptr(oc),
                                   L.ptr(bb), L.ptr(bs), L.ptr(bc), L.ptr(ws), ws_bytes, L.stream()), "det_select")
    return ob, os_, ol, oc, bb, bs, bc


# ----------------------------------------------------------------------------------------------- ablation distillation losses
def feat_distill(src, tgt, want_grad=False):
    """normalized_filtered_l1 between two feature maps of equal size (any layout, both the same) -> (loss[1], d_tgt or None)"""
    L.require_cuda(src, tgt)
    assert src.numel() == tgt.numel()
    loss = torch.empty((1,), dtype=_f32, device=tgt.device)
    stats = torch.empty((3,), dtype=_f32, device=tgt.device)
    d = torch.empty_like(tgt, memory_format=torch.preserve_format) if want_grad else None
    L.check(L.lib().abr_feat_distill(L.ptr(src), L.ptr(tgt), tgt.numel(), L.ptr(loss), L.ptr(stats), 1.0, L.ptr(d), L.stream()), "feat_distill")
    return loss, d


def rpn_distill(obj_s, reg_s, obj_t, reg_t, thr, use_bbox, want_grad=False):
    """obj_* [N,H,W,A] / reg_* [N,H,W,4A] NHWC views (row-strided slices of the fused head outputs are fine) ->
    (loss[1], d_obj_t [N,H,W,A] or None, d_reg_t [N,H,W,4A] or None)"""
    L.require_cuda(obj_s, reg_s, obj_t, reg_t)
    N, H, W, A = obj_t.shape
    for t in (obj_s, reg_s, obj_t, reg_t):
        assert t.stride(-1) == 1 and t.stride(0) == H * W * t.stride(2) and t.stride(1) == W * t.stride(2), "expected row-strided NHWC views"
    loss = torch.empty((1,), dtype=_f32, device=obj_t.device)
    d_o = torch.empty((N, H, W, A), dtype=_f32, device=obj_t.device) if want_grad else None
    d_r = torch.empty((N, H, W, 4 * A), dtype=_f32, device=obj_t.device) if want_grad else None
    L.check(L.lib().abr_rpn_distill(L.ptr(obj_s), L.ptr(reg_s), obj_s.stride(2), reg_s.stride(2), L.ptr(obj_t), L.ptr(reg_t), obj_t.stride(2),
                                    reg_t.stride(2), N * H * W, A, float(thr), int(bool(use_bbox)), L.ptr(loss), 1.0, L.ptr(d_o), L.ptr(d_r),
                                    A, 4 * A, L.stream()), "rpn_distill")
    return loss, d_o, d_r
