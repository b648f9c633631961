"""Region Proposal Network on the HIP library.

Mirrors, for the single-level C4 case every configs/voc YAML uses:
    AnchorGenerator          maskrcnn_benchmark/modeling/rpn/anchor_generator.py:34-123, 215-284
    RPNHead                  modeling/rpn/rpn.py:70-121  (conv3x3+ReLU, cls_logits 1x1, bbox_pred 1x1)
    RPNPostProcessor         modeling/rpn/inference.py:14-147
    RPNLossComputation       modeling/rpn/loss.py:21-148
    RPNModule / build_rpn    modeling/rpn/rpn.py:124-230

MI355X-first differences (results identical):
  * cls_logits and bbox_pred are ONE 1x1 conv with 15+60(+1 pad)=76 output channels over the shared 3x3 feature: the
    1024-channel tensor `t` is read once.  The NHWC output [N,H,W,76] already IS the reference's
    permute_and_flatten order (location-major, anchor-minor; rpn/utils.py:10-14), so no permute kernels exist.
  * proposals for all images are decoded / clipped / NMS-ed in batched launches; the greedy NMS sweep stays on device.
  * anchors are cached per (H, W, image size) instead of being rebuilt with numpy every step.
  * IoU + Matcher + labels + BoxCoder.encode for 35 910 anchors is one kernel per image.
"""
import math

import numpy as np
import os

import torch
from torch import nn
from torch.autograd import Function

from ... import ops
from ...layers._layout import as_nhwc, from_nhwc
from ...structures.bounding_box import BoxList
from ...structures.image_list import ImageList
from ..backbone.resnet import Conv2d, _grad_buf
from ..balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from ..box_coder import BoxCoder
from ..matcher import Matcher


# ------------------------------------------------------------------------------------------------ anchors
def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """Cell anchors (anchor_generator.py:215-284): float64, numpy round (half-to-even) on the ratio enumeration."""
    base = float(stride)
    xc = yc = 0.5 * (base - 1)
    area = base * base
    out = []
    for r in aspect_ratios:
        ws = np.round(np.sqrt(area / r))
        hs = np.round(ws * r)
        x1, y1, x2, y2 = xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)
        w0, h0 = x2 - x1 + 1, y2 - y1 + 1
        cx, cy = x1 + 0.5 * (w0 - 1), y1 + 0.5 * (h0 - 1)
        for s in sizes:
            sc = float(s) / stride
            w, h = w0 * sc, h0 * sc
            out.append([cx - 0.5 * (w - 1), cy - 0.5 * (h - 1), cx + 0.5 * (w - 1), cy + 0.5 * (h - 1)])
    return torch.tensor(np.array(out, dtype=np.float64)).float()


PROPOSALS_SIDE_STREAM = os.environ.get("ABR_PROPOSAL_STREAM", "1") != "0"
# training: hand the box head the selector's RAW device output (LazyProposals) instead of per-image BoxLists cut on the host
FUSED_ROI_TARGETS = os.environ.get("ABR_FUSED_ROI_TARGETS", "1") != "0"


_DBG_SLEEP_PROPOSALS = int(os.environ.get("ABR_DBG_SLEEP_PROPOSALS", "0"))   # spin cycles in front of the training selection (probe only)


class LazyProposals(object):
    """The training selector's output before anything has been read back: the decoded score-sorted boxes, the NMS keep lists and
    their device-resident counts (RPNPostProcessor.launch).  The box head consumes it as is (ROIBoxHead.forward ->
    ops.roi_head_targets: GT append, matching, sampling and the RoI table in three launches, no host round trip).  Any other use --
    indexing, iteration, len() -- materialises the reference's list of per-image BoxLists (RPNPostProcessor.collect: one read-back
    of the counts), so it can stand wherever that list is expected."""

    def __init__(self, selector, pending, targets):
        self.selector, self.pending, self.targets = selector, pending, targets
        self._list = None

    def raw(self):
        """(props [N,k,4], scores [N,k], keep [N,post] int32, n_keep [N] int32, image sizes) visible to the current stream"""
        p = self.selector.join(self.pending)
        return p["props"], p["scores"], p["keep"], p["n_keep"], p["sizes"]

    def materialize(self):
        if self._list is None:
            self._list = self.selector.collect(self.pending, self.targets)
        return self._list

    def __len__(self):
        return len(self.pending["sizes"])

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, i):
        return self.materialize()[i]


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32), straddle_thresh=0):
        super().__init__()
        assert len(anchor_strides) == 1, "single feature level (C4) only: FPN is out of scope (no voc config uses it)"
        self.strides = anchor_strides
        self.straddle_thresh = straddle_thresh
        self.register_buffer("cell_anchors", generate_anchors(anchor_strides[0], sizes, aspect_ratios))
        self._cache = {}

    def num_anchors_per_location(self):
        return [self.cell_anchors.shape[0]]

    def grid(self, H, W, image_hw):
        """([H*W*A,4] anchors, [H*W*A] uint8 visibility) for one image size; cached on device."""
        key = (H, W, int(image_hw[0]), int(image_hw[1]), self.cell_anchors.device)
        hit = self._cache.get(key)
        if hit is None:
            a, vis = ops.grid_anchors(self.cell_anchors, H, W, self.strides[0], int(image_hw[0]), int(image_hw[1]), self.straddle_thresh)
            # the boxes depend on (H, W) only, the visibility also on the image size: share ONE box tensor per grid
            a = self._cache.setdefault((H, W, self.cell_anchors.device), a)
            hit = self._cache[key] = (a, vis, vis.bool())   # uint8 for the kernels, bool for the reference-typed BoxList field: made once
        return hit

    def forward(self, image_list, feature_maps):
        H, W = feature_maps[0].shape[-2:]
        anchors = []
        for (ih, iw) in image_list.image_sizes:
            a, vis, vis_bool = self.grid(H, W, (ih, iw))
            bl = BoxList(a, (iw, ih), mode="xyxy")
            bl.add_field("visibility", vis_bool)
            bl._visibility_u8 = vis                   # the same mask as the kernels read it (no per-step dtype round trip)
            anchors.append([bl])
        return anchors


def make_anchor_generator(cfg):
    return AnchorGenerator(cfg.MODEL.RPN.ANCHOR_SIZES, cfg.MODEL.RPN.ASPECT_RATIOS, cfg.MODEL.RPN.ANCHOR_STRIDE, cfg.MODEL.RPN.STRADDLE_THRESH)


# ------------------------------------------------------------------------------------------------ head
class _RPNHeadFn(Function):
    """t = relu(conv3x3(x)+b) ; y = conv1x1_fused(t)+b  as one autograd node with a hand-scheduled backward."""

    @staticmethod
    def forward(ctx, x, head, *params):
        xh = as_nhwc(x)
        v = ops.wino_v_alloc(xh, head.conv.weight, 1, 1, head.math) if head.conv.weight.requires_grad else None
        t = ops.conv_forward(xh, head.conv.weight, 1, 1, bias=head.conv.bias, relu=True, math=head.math, wino_v=v, w_version=head.conv.version())
        y = ops.conv_forward(t, head.fused_weight, 1, 0, bias=head.fused_bias, math=head.math, w_version=head.fused_version())
        ctx.head, ctx.saved = head, (xh, t, v)
        ctx.need_dx = x.requires_grad
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, gy):
        head = ctx.head
        xh, t, v = ctx.saved
        g = as_nhwc(gy)
        if not g.is_contiguous():
            g = g.contiguous()
        ops.conv_wgrad_async(t, g, head.fused_weight_grad, 1, 0, math=head.math)
        ops.bias_grad(g, head.fused_bias_grad)
        gt = ops.conv_forward(g, head.fused_dgrad_weight(), 1, 0, mask=t, math=head.math)   # (reduction width 76: exact fp32 either way)
        ops.conv_wgrad_async(xh, gt, _grad_buf(head.conv.weight), 1, 1, math=head.math, wino_v=v)
        ops.bias_grad(gt, _grad_buf(head.conv.bias))
        gx = from_nhwc(ops.conv_forward(gt, head.conv.dgrad_weight(), 1, 1, math=head.math, w_version=head.conv.version())) if ctx.need_dx else None
        ctx.saved = None
        return (gx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class RPNHead(nn.Module):
    """rpn.py:70-121.  `cls_logits` / `bbox_pred` keep their reference names and shapes; their storage is two
    row-slices of one fused [76,1,1,C] buffer (rows 0..14, 15..74, row 75 = zero pad so that Cout % 4 == 0)."""

    def __init__(self, cfg, in_channels, num_anchors):
        super().__init__()
        self.num_anchors = num_anchors
        self.math = ops.MATH_F32   # see backbone.resnet.set_conv_math
        self.conv = Conv2d(in_channels, in_channels, 3, stride=1, padding=1)
        self.cls_logits = Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = Conv2d(in_channels, num_anchors * 4, 1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):  # rpn.py:87-89
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        if cfg.MODEL.RPN.CONV_FREEZE:
            for p in self.conv.parameters():
                p.requires_grad = False
        if cfg.MODEL.RPN.CLS_FREEZE:
            for p in self.cls_logits.parameters():
                p.requires_grad = False
        if cfg.MODEL.RPN.BBS_FREEZE:
            for p in self.bbox_pred.parameters():
                p.requires_grad = False
        self.n_out = num_anchors * 5
        self.n_out_pad = (self.n_out + 3) // 4 * 4
        self._fuse()

    def _fuse(self):
        """(re)build the fused storage and re-point the two reference-named parameters into it"""
        A, C_, dev = self.num_anchors, self.conv.in_channels, self.cls_logits.weight.device
        fw = torch.zeros(self.n_out_pad, 1, 1, C_, device=dev)
        fb = torch.zeros(self.n_out_pad, device=dev)
        with torch.no_grad():
            fw[:A].copy_(self.cls_logits.weight)
            fw[A:5 * A].copy_(self.bbox_pred.weight)
            fb[:A].copy_(self.cls_logits.bias)
            fb[A:5 * A].copy_(self.bbox_pred.bias)
        self._set_fused(fw, fb, torch.zeros_like(fw), torch.zeros_like(fb))

    def _set_fused(self, fw, fb, gw, gb):
        A = self.num_anchors
        self.fused_weight, self.fused_bias = fw, fb
        self.fused_weight_grad, self.fused_bias_grad = gw, gb
        for mod, lo, hi in ((self.cls_logits, 0, A), (self.bbox_pred, A, 5 * A)):
            mod.weight.data, mod.bias.data = fw[lo:hi], fb[lo:hi]
            mod.weight.grad, mod.bias.grad = gw[lo:hi], gb[lo:hi]
        self._wt, self._wt_version = None, -1

    def rehome(self, which, view, grad_view):
        """called by flatten_parameters: adopt flat-buffer storage for the fused weight / bias"""
        if which == "weight":
            shp = self.fused_weight.shape
            self._set_fused(view.view(shp), self.fused_bias, grad_view.view(shp), self.fused_bias_grad)
        else:
            self._set_fused(self.fused_weight, view, self.fused_weight_grad, grad_view)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._fuse()
        return out

    def flat_groups(self):
        """tensors that must stay contiguous when the model's parameters are re-homed into one flat buffer"""
        return [("weight", [self.cls_logits.weight, self.bbox_pred.weight], self.n_out_pad - self.n_out),
                ("bias", [self.cls_logits.bias, self.bbox_pred.bias], self.n_out_pad - self.n_out)]

    def prepare_derived(self):
        """see Bottleneck.prepare_derived"""
        if not self.fused_weight.is_cuda:
            return
        if self.fused_weight_grad is not None and (self.cls_logits.weight.requires_grad or self.bbox_pred.weight.requires_grad):
            self.fused_dgrad_weight()
            ops.conv_prepare_weights(self.fused_weight, 1, 0, self.math, self.fused_version())
        if self.conv.weight.requires_grad:
            wt = self.conv.dgrad_weight()
            ops.conv_prepare_weights(self.conv.weight, 1, 1, self.math, self.conv.version())
            ops.conv_prepare_weights(wt, 1, 1, self.math, self.conv.version())

    def prep_entries(self):
        """the 3x3 head conv goes with FusedSGD's batched preparation; prepare_rest() does the fused 1x1 heads"""
        return [(self.conv, None, 1, 1, self.math)] if (self.conv.weight.requires_grad and self.conv.weight.is_cuda) else []

    def prepare_rest(self):
        if self.fused_weight.is_cuda and self.fused_weight_grad is not None and (self.cls_logits.weight.requires_grad or self.bbox_pred.weight.requires_grad):
            self.fused_dgrad_weight()
            ops.conv_prepare_weights(self.fused_weight, 1, 0, self.math, self.fused_version())

    def fused_version(self):
        """abr_conv_desc::w_version of the fused cls | bbox weight (Conv2d.version's rule: it moves with optimiser steps iff an optimiser
        owns one of its two halves)"""
        from ..backbone.resnet import _PARAM_VERSION, _STATIC_VERSION
        opt = self.cls_logits._optimised or self.bbox_pred._optimised
        return 2 * _PARAM_VERSION[0] + 1 if opt else 2 * _STATIC_VERSION[0] + 2

    def fused_dgrad_weight(self):
        from ..backbone.resnet import _PARAM_VERSION
        ops.prep_wait()
        if self._wt is None or self._wt_version != _PARAM_VERSION[0]:
            self._wt = ops.conv_dgrad_weights(self.fused_weight, None, out=self._wt)
            self._wt_version = _PARAM_VERSION[0]
        return self._wt

    def forward_fused(self, x):
        """x logical [N,C,H,W] -> logical [N,76,H,W] whose NHWC memory is [N,H,W,(15 obj | 60 reg | pad)]"""
        params = list(self.parameters())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _RPNHeadFn.apply(x, self, *params)
        xh = as_nhwc(x)
        t = ops.conv_forward(xh, self.conv.weight, 1, 1, bias=self.conv.bias, relu=True, math=self.math, w_version=self.conv.version())
        return from_nhwc(ops.conv_forward(t, self.fused_weight, 1, 0, bias=self.fused_bias, math=self.math, w_version=self.fused_version()))

    def forward(self, x):
        logits, bbox_reg = [], []
        for feature in x:
            y = self.forward_fused(feature)
            logits.append(y[:, : self.num_anchors])
            bbox_reg.append(y[:, self.num_anchors: 5 * self.num_anchors])
        return logits, bbox_reg


# ------------------------------------------------------------------------------------------------ proposals
class RPNPostProcessor(nn.Module):
    """inference.py:14-147: sigmoid -> top-k (sorted) -> decode -> clip -> remove small -> NMS -> first post_nms (+GT in training)."""

    def __init__(self, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_coder=None, fpn_post_nms_top_n=None,
                 fpn_post_nms_per_batch=True):
        super().__init__()
        self._hw_cache = {}
        self.pre_nms_top_n, self.post_nms_top_n = pre_nms_top_n, post_nms_top_n
        self.nms_thresh, self.min_size = nms_thresh, min_size
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))

    def add_gt_proposals(self, proposals, targets):
        out = []
        for p, t in zip(proposals, targets):
            gt = BoxList(t.bbox, t.size, t.mode)
            gt.add_field("objectness", torch.ones(len(gt), device=t.bbox.device))
            b = BoxList(torch.cat((p.bbox, gt.bbox), 0), p.size, p.mode)
            b.add_field("objectness", torch.cat((p.get_field("objectness"), gt.get_field("objectness")), 0))
            out.append(b)
        return out

    def launch(self, anchors, fused, num_anchors):
        """Device half of forward_fused: top-k -> decode -> clip -> NMS, all enqueued on the CURRENT stream, nothing read back.
        Returns the pending state for `collect`."""
        y = as_nhwc(fused)
        N, H, W, Cf = y.shape
        A = num_anchors
        n_anchor = H * W * A
        yf = y.reshape(N, H * W, Cf)
        k = min(self.pre_nms_top_n, n_anchor)
        if k <= 15360:   # one fused kernel per batch: sigmoid + radix select + in-LDS bitonic sort (inference.py:87-96)
            scores, topk_idx = ops.topk_sigmoid(yf, A, k)
        else:            # beyond the in-LDS sort capacity (no voc config gets here: PRE_NMS_TOP_N is 12000 / 6000)
            scores, topk_idx = yf[:, :, :A].reshape(N, n_anchor).sigmoid().topk(k, dim=1, sorted=True)
        same = all(a[0].bbox.data_ptr() == anchors[0][0].bbox.data_ptr() for a in anchors)
        hw_key = (tuple((a[0].size[1], a[0].size[0]) for a in anchors), y.device)
        img_hw = self._hw_cache.get(hw_key)   # image sizes repeat from step to step: keep the device copy
        if img_hw is None:
            if len(self._hw_cache) > 64:
                self._hw_cache.clear()
            img_hw = self._hw_cache[hw_key] = ops.h2d([list(v) for v in hw_key[0]], torch.int32, y.device)
        assert same, "per-image anchor grids of one batch share (H,W): they differ only in the visibility field"
        props = ops.rpn_decode_clip(yf, A, anchors[0][0].bbox, topk_idx, img_hw, self.box_coder.weights, A=A)  # :101-112
        counts = torch.full((N,), k, dtype=torch.int32, device=y.device)
        if self.min_size > 0:  # remove_small_boxes (boxlist_ops.py:34-48): with MIN_SIZE=0 (every voc config) nothing is dropped
            raise NotImplementedError("RPN.MIN_SIZE > 0 is not used by any configs/voc YAML")
        keep, n_keep = ops.nms_sorted_batched(props, counts, self.nms_thresh, self.post_nms_top_n)     # :113-116
        return dict(props=props, scores=scores, keep=keep, n_keep=n_keep, sizes=[a[0].size for a in anchors], fused=yf)

    def collect(self, pending, targets=None):
        """Host half: read the per-image keep counts (the only host sync of the proposal path; the reference syncs inside every
        nms call) and cut the BoxLists."""
        side = pending.get("stream")
        if side is not None:
            # read the counts ON THE SIDE STREAM: the copy then waits for the selection only, not for whatever the caller has queued
            # on the main stream in the meantime (a main-stream read-back would stall the host behind all of it)
            with torch.cuda.stream(side):
                nk = pending["n_keep"].tolist()
            self.join(pending)
        else:
            nk = pending["n_keep"].tolist()
        props, scores, keep = pending["props"], pending["scores"], pending["keep"]
        result = []
        for i, size in enumerate(pending["sizes"]):
            ki = keep[i, : nk[i]].long()
            b = BoxList(props[i].index_select(0, ki), size, mode="xyxy")
            b.add_field("objectness", scores[i].index_select(0, ki))
            result.append(b)
        if self.training and targets is not None:
            result = self.add_gt_proposals(result, targets)                               # :144-145
        return result

    def launch_on_side_stream(self, anchors, fused, num_anchors, tag="proposals"):
        """`launch` on a side stream: the selection is a chain of latency-bound kernels (a 1024-thread top-k and a one-workgroup-per-
        image NMS sweep, ~0.9 ms with a handful of CUs busy) that has no consumer until `collect`, so it runs NEXT to whatever the
        caller enqueues on the current stream meanwhile (the RPN loss; for the frozen source model, the target's whole backbone)."""
        cur = torch.cuda.current_stream()
        side = ops.side_stream((fused.device.index, tag))
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if _DBG_SLEEP_PROPOSALS and self.training:
                torch.cuda._sleep(_DBG_SLEEP_PROPOSALS)      # (criticality probe: tools/dbg/critical_probe.sh)
            pending = self.launch(anchors, fused, num_anchors)
        fused.record_stream(side)
        pending["stream"] = side
        return pending

    def join(self, pending):
        """Make the current stream (and the caching allocator) see the side stream's results."""
        side = pending.pop("stream", None)
        if side is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(side)
            for k in ("props", "scores", "keep", "n_keep"):
                pending[k].record_stream(cur)
        return pending

    def forward_fused(self, anchors, fused, num_anchors, targets=None):
        """fused: logical [N,5A(+pad),H,W] head output.  anchors: list (per image) of [BoxList] as AnchorGenerator returns."""
        return self.collect(self.launch(anchors, fused, num_anchors), targets)

    def forward(self, anchors, objectness, box_regression, targets=None):
        """reference signature (lists of logical [N,A,H,W] / [N,4A,H,W]); re-fuses the two tensors (compat path)."""
        fused = torch.cat((objectness[0], box_regression[0]), 1)
        return self.forward_fused(anchors, fused, objectness[0].shape[1], targets)


def make_rpn_postprocessor(config, rpn_box_coder, is_train):
    pre = config.MODEL.RPN.PRE_NMS_TOP_N_TRAIN if is_train else config.MODEL.RPN.PRE_NMS_TOP_N_TEST
    post = config.MODEL.RPN.POST_NMS_TOP_N_TRAIN if is_train else config.MODEL.RPN.POST_NMS_TOP_N_TEST
    return RPNPostProcessor(pre, post, config.MODEL.RPN.NMS_THRESH, config.MODEL.RPN.MIN_SIZE, rpn_box_coder)


# ------------------------------------------------------------------------------------------------ loss
class _RPNLossFn(Function):
    """BCE-with-logits over the sampled anchors + smooth-L1(beta=1/9) over the sampled positives / #sampled
    (rpn/loss.py:136,145-146), reading objectness and deltas straight out of the fused NHWC head output."""

    @staticmethod
    def forward(ctx, fused, A, labels, reg_targets, pos_idx, samp_idx, denom=None, prepared=None):
        y = as_nhwc(fused)
        N, H, W, Cf = y.shape
        y2 = y.reshape(N * H * W, Cf)
        # index lists may be fixed-size and -1 padded (fused sampler): negative entries stay negative below and are skipped
        # by the kernels; `denom` is then the device-resident number of sampled anchors
        n_samp = samp_idx.numel()
        # anchor j of the flattened batch lives in row j // A; objectness at column j % A, deltas at A + 4*(j % A)
        if prepared is not None:   # (obj_flat_idx, pos rows, pos columns) straight from ops.rpn_loss_indices
            obj_flat_idx, prow, pcol = prepared
        else:
            row = torch.div(samp_idx, A, rounding_mode="floor")
            obj_flat_idx = row * Cf + (samp_idx - row * A)
            prow = torch.div(pos_idx, A, rounding_mode="floor")
            pcol = A + 4 * (pos_idx - prow * A)
        want = fused.requires_grad
        lo, g_obj = ops.bce_logits_gather(y2, labels, obj_flat_idx, want_grad=want, yidx=samp_idx, denom_dev=denom)
        lb, g_reg = ops.smooth_l1_rows(y2, reg_targets, prow, pcol, 1.0 / 9,
                                       scale=1.0 if denom is not None else 1.0 / max(n_samp, 1), want_grad=want, trows=pos_idx,
                                       denom_dev=denom)
        ctx.shape = (N, H, W, Cf)
        ctx.save_for_backward(g_obj, g_reg)
        return lo[0], lb[0]

    @staticmethod
    def backward(ctx, g_lo, g_lb):
        g_obj, g_reg = ctx.saved_tensors
        ops.scale_(g_obj, 1.0, g_lo.contiguous())
        ops.scale_(g_reg, 1.0, g_lb.contiguous())
        ops.add_(g_obj, g_reg)  # disjoint columns of the same [N*H*W, Cf] buffer
        return from_nhwc(g_obj.view(ctx.shape)), None, None, None, None, None, None, None


class RPNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, generate_labels_func=None):
        self.proposal_matcher, self.fg_bg_sampler, self.box_coder = proposal_matcher, fg_bg_sampler, box_coder
        self.discard_cases = ["not_visibility", "between_thresholds"]

    def prepare_targets(self, anchors, targets):
        """rpn/loss.py:66-102 per image -> labels fp32 {1,0,-1}, regression targets, matched idxs"""
        labels, regression_targets, matched = [], [], []
        for a, t in zip(anchors, targets):
            vis = getattr(a, "_visibility_u8", None)
            if vis is None:
                vis = a.get_field("visibility")
                vis = vis.to(torch.uint8) if vis.dtype != torch.uint8 else vis
            m, lab, tgt = self.proposal_matcher.match_boxes(t.bbox, a.bbox, None, vis, self.box_coder.weights, rpn_labels=True)
            labels.append(lab)
            regression_targets.append(tgt)
            matched.append(m)
        return labels, regression_targets, matched

    def sample(self, labels):
        """One fused sampler launch for the whole batch, nothing leaves the device.  Returns batch-flattened index lists of
        FIXED length (-1 padded): positives [N*128], all sampled (positives then negatives) [N*128 + N*256], and the device
        scalar #sampled that normalises both losses (rpn/loss.py:119-123,136,146)."""
        lab2d = torch.stack(labels)
        n = lab2d.shape[1]
        pos, neg, counts = self.fg_bg_sampler.sample_padded(lab2d, index_offset_per_image=n)
        pos = pos.reshape(-1)
        return pos, torch.cat([pos, neg.reshape(-1)]), counts.sum().to(torch.float32).reshape(1)

    def __call__(self, anchors, objectness, box_regression, targets, rpn_output_source=None, fused=None, sampled=None):
        """Returns (objectness_loss, box_loss).  `rpn_output_source` is accepted and ignored as in the reference (:129-143).
        `sampled=(pos_idx, samp_idx)` injects the sampler's choice (parity tests)."""
        anchors = [a[0] if isinstance(a, (list, tuple)) else a for a in anchors]
        if sampled is None:
            sampled = getattr(self, "inject_sampled", None)  # parity tests pin the sampler's draw here
        if fused is None:
            fused = torch.cat((objectness[0], box_regression[0]), 1)
            A = objectness[0].shape[1]
        else:
            A = anchors[0].bbox.shape[0] // (fused.shape[-1] * fused.shape[-2])
        shared = fused.is_cuda and all(a.bbox.data_ptr() == anchors[0].bbox.data_ptr() and getattr(a, "_visibility_u8", None) is not None for a in anchors)
        if shared:
            # the batch's targets in two launches (one anchor grid for all images), the sampler in one, the loss bookkeeping in one
            lab2d, tgt3d, _keep = ops.rpn_targets_batched(anchors[0].bbox, [a._visibility_u8 for a in anchors], [t.bbox for t in targets],
                                                          self.proposal_matcher.high_threshold, self.proposal_matcher.low_threshold,
                                                          self.box_coder.weights)
            labels, regression_targets = list(lab2d.unbind(0)), list(tgt3d.unbind(0))      # views (introspection, API)
            lab_flat, tgt_flat = lab2d.view(-1), tgt3d.view(-1, 4)
        else:
            labels, regression_targets, _ = self.prepare_targets(anchors, targets)
            lab2d = None
            lab_flat, tgt_flat = torch.cat(labels), torch.cat(regression_targets)
        denom = prepared = None
        if sampled is not None:
            pos_idx, samp_idx = sampled
        elif lab2d is not None:
            n = lab2d.shape[1]
            pos, neg, counts = self.fg_bg_sampler.sample_padded(lab2d, index_offset_per_image=n)
            samp_idx, obj_flat, prow, pcol, denom = ops.rpn_loss_indices(pos, neg, counts, A, fused.shape[1])
            pos_idx = pos.reshape(-1)
            prepared = (obj_flat, prow, pcol)
        else:
            pos_idx, samp_idx, denom = self.sample(labels)
        self.last_sampled, self.last_targets = (pos_idx, samp_idx), (labels, regression_targets)  # introspection for parity tests
        return _RPNLossFn.apply(fused, A, lab_flat, tgt_flat, pos_idx, samp_idx, denom, prepared)


def make_rpn_loss_evaluator(cfg, box_coder):
    matcher = Matcher(cfg.MODEL.RPN.FG_IOU_THRESHOLD, cfg.MODEL.RPN.BG_IOU_THRESHOLD, allow_low_quality_matches=True)
    sampler = BalancedPositiveNegativeSampler(cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE, cfg.MODEL.RPN.POSITIVE_FRACTION)
    return RPNLossComputation(matcher, sampler, box_coder)


# ------------------------------------------------------------------------------------------------ module
class RPNModule(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.cfg = cfg.clone()
        self.anchor_generator = make_anchor_generator(cfg)
        self.head = RPNHead(cfg, in_channels, self.anchor_generator.num_anchors_per_location()[0])
        rpn_box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.box_selector_train = make_rpn_postprocessor(cfg, rpn_box_coder, is_train=True)
        self.box_selector_test = make_rpn_postprocessor(cfg, rpn_box_coder, is_train=False)
        self.loss_evaluator = make_rpn_loss_evaluator(cfg, rpn_box_coder)

    def forward(self, images, features, targets=None, rpn_output_source=None, defer_proposals=False):
        """-> ((boxes, losses), anchors, rpn_output) exactly as rpn.py:161-183."""
        if self.training:
            return self.forward_finish(self.forward_begin(images, features, targets, rpn_output_source))
        fused = self.head.forward_fused(features[0])
        A = self.head.num_anchors
        anchors = self.anchor_generator(images, features)
        rpn_output = ([fused[:, :A]], [fused[:, A:5 * A]])
        self.box_selector_test.eval()
        if defer_proposals:   # the caller collects later (GeneralizedRCNN.soften_begin / soften_finish)
            pending = self.box_selector_test.launch_on_side_stream(anchors, fused, A, tag="source-proposals")
            return (pending, {}), anchors, rpn_output
        boxes = self.box_selector_test.forward_fused(anchors, fused, A)
        return (boxes, {}), anchors, rpn_output

    def forward_begin(self, images, features, targets, rpn_output_source=None):
        """Training forward up to (not including) the read-back of the proposal counts: head conv, anchors, the proposal selection
        launched on a side stream, the loss on the current stream."""
        fused = self.head.forward_fused(features[0])
        A = self.head.num_anchors
        anchors = self.anchor_generator(images, features)
        rpn_output = ([fused[:, :A]], [fused[:, A:5 * A]])
        with torch.no_grad():
            self.box_selector_train.train()
            if PROPOSALS_SIDE_STREAM and fused.is_cuda:   # selection runs next to the loss kernels below
                pending = self.box_selector_train.launch_on_side_stream(anchors, fused.detach(), A)
            else:
                pending = self.box_selector_train.launch(anchors, fused.detach(), A)
        loss_objectness, loss_rpn_box_reg = self.loss_evaluator(anchors, None, None, targets, rpn_output_source, fused=fused)
        return dict(pending=pending, targets=targets, anchors=anchors, rpn_output=rpn_output,
                    losses={"loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg})

    def forward_finish(self, state):
        with torch.no_grad():
            if FUSED_ROI_TARGETS and state["pending"]["props"].is_cuda and state["targets"] is not None:
                boxes = LazyProposals(self.box_selector_train, state["pending"], state["targets"])
            else:
                boxes = self.box_selector_train.collect(state["pending"], state["targets"])
        return (boxes, state["losses"]), state["anchors"], state["rpn_output"]


def build_rpn(cfg, in_channels):
    return RPNModule(cfg, in_channels)
