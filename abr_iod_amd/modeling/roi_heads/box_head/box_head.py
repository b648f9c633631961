"""RoI box head on the HIP library.

Mirrors (C4 / ResNet50Conv5 variant, the only one configs/voc uses):
    Pooler                              maskrcnn_benchmark/modeling/poolers.py:45-105 (single level -> ROIAlign)
    ResNet50Conv5ROIFeatureExtractor    modeling/roi_heads/box_head/roi_box_feature_extractors.py:14-55
    FastRCNNPredictor                   modeling/roi_heads/box_head/roi_box_predictors.py:8-33
    FastRCNNLossComputation             modeling/roi_heads/box_head/loss.py:15-181
    ROIBoxHead                          modeling/roi_heads/box_head/box_head.py:12-87

MI355X-first differences (results identical):
  * layer4's first 1x1 convs have stride 2 over the 7x7 pooled map, i.e. they read bins (0,2,4,6)^2 only.  When the
    caller does not consume the pooled features themselves (the 512-RoI detection pass: train_incremental.py:89 binds
    them to `_`) ROIAlign computes just those 16 bins (bin_step=2) and layer4 runs stride-1 on the 4x4 map: 3x less
    ROIAlign traffic, forward and backward.  The 64-RoI distillation passes need all 49 bins for ARD and get them.
  * avgpool + cls_score + bbox_pred is one pooled vector and ONE GEMM with K_all + 4*K_all (+pad) output columns.
  * IoU/Matcher/labels/encode are one kernel per image; both loss terms and their gradients are one kernel each.
"""
import torch
import os
from torch import nn
from torch.autograd import Function

from .... import ops
from ....layers import ROIAlign
from ....layers._layout import as_nhwc, from_nhwc
from ....structures.bounding_box import BoxList
from ...backbone.resnet import Conv2d, ResNetHead, _grad_buf, _PARAM_VERSION
from ...balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from ...box_coder import BoxCoder
from ...matcher import Matcher


# ------------------------------------------------------------------------------------------------ pooler
_roi_id_cache = {}


def convert_to_roi_format(boxes):
    """poolers.py:73-86: [K,5] = (batch index, x1, y1, x2, y2).  The batch-index column only depends on the per-image box counts,
    which repeat from step to step (512 sampled RoIs / 64 distillation RoIs per image): it is cached instead of being rebuilt from
    2N tiny kernels each time."""
    tab = getattr(boxes[0], "_roi_table", None) if len(boxes) else None
    if tab is not None and all(getattr(b, "_roi_table", (None, -1))[0] is tab[0] and b._roi_table[1] == i for i, b in enumerate(boxes)) \
            and tab[0].shape[0] == sum(len(b) for b in boxes):
        return tab[0]   # these BoxLists ARE consecutive views of one ready-made [K,5] table (fused proposal paths)
    concat = torch.cat([b.bbox for b in boxes], dim=0)
    key = (tuple(len(b) for b in boxes), concat.device, concat.dtype)
    ids = _roi_id_cache.get(key)
    if ids is None:
        if len(_roi_id_cache) > 256:
            _roi_id_cache.clear()
        ids = _roi_id_cache[key] = torch.cat([torch.full((len(b), 1), i, dtype=concat.dtype, device=concat.device) for i, b in enumerate(boxes)], dim=0)
    return torch.cat([ids, concat], dim=1)


_DBG_SLEEP_ROI_TARGETS = int(os.environ.get("ABR_DBG_SLEEP_ROI_TARGETS", "0"))   # spin cycles in front of the RoI targets (probe only)


class _JointPoolFn(Function):
    """ROIAlign of the detection RoIs (even bins: all layer4's stride-2 1x1 convs read) and of the distillation RoIs (all 7x7 bins: ARD reads
    them) with the rows layer4 sees written side by side into ONE [Kd + Ks, 4, 4, C] tensor -- no concatenation copy of the detection part.
    -> (joint logical [Kd+Ks,C,4,4], soft logical [Ks,C,7,7])."""

    @staticmethod
    def forward(ctx, feat, det_rois, soft_rois, output_size, spatial_scale, sampling_ratio, soft_ready=None):
        """soft_ready: the distillation RoIs' all-bin pooling [Ks,ph,pw,C] when the caller computed it already (it depends on the features and the
        SOURCE's proposals only: ROIBoxHead.forward_joint issues it before the main stream starts waiting for the target's own proposals)"""
        ph, pw = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        fh = as_nhwc(feat)
        Kd, Ks, C_ = det_rois.shape[0], soft_rois.shape[0], fh.shape[-1]
        pho, pwo = -(-ph // 2), -(-pw // 2)
        joint = torch.empty((Kd + Ks, pho, pwo, C_), dtype=fh.dtype, device=fh.device)
        ops.roi_align_forward(fh, det_rois, spatial_scale, ph, pw, sampling_ratio, 2, out=joint[:Kd])
        soft = soft_ready if soft_ready is not None else ops.roi_align_forward(fh, soft_rois, spatial_scale, ph, pw, sampling_ratio, 1)
        joint[Kd:].copy_(soft[:, ::2, ::2, :])
        ops.amax_carry_bound(joint, fh)   # pooled values are averages of bilinear samples: bounded by the feature map's amax (f16x3 scales)
        ops.amax_carry_bound(soft, fh)
        ctx.save_for_backward(det_rois, soft_rois)
        ctx.geom = (ph, pw, spatial_scale, sampling_ratio, tuple(fh.shape), Kd)
        return from_nhwc(joint), from_nhwc(soft)

    @staticmethod
    def backward(ctx, g_joint, g_soft):
        det_rois, soft_rois = ctx.saved_tensors
        ph, pw, scale, sr, (B, H, W, C_), Kd = ctx.geom
        gj = as_nhwc(g_joint).contiguous()
        if g_soft is None:
            gs = torch.zeros((soft_rois.shape[0], ph, pw, C_), dtype=gj.dtype, device=gj.device)
        else:
            gs = as_nhwc(g_soft)
            gs = gs.clone() if gs.is_contiguous() else gs.contiguous()     # (never write into autograd's own gradient tensor)
        gs[:, ::2, ::2, :] += gj[Kd:]            # layer4's gradient reaches the even bins of the distillation RoIs
        g = ops.roi_align_backward(gj[:Kd], det_rois, scale, ph, pw, sr, B, H, W, C_, 2)
        g = ops.roi_align_backward(gs, soft_rois, scale, ph, pw, sr, B, H, W, C_, 1, out=g)
        return from_nhwc(g), None, None, None, None, None, None


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio):
        super().__init__()
        assert len(scales) == 1, "single-level pooling only (C4); FPN level mapping is out of scope"
        self.poolers = nn.ModuleList([ROIAlign(output_size, spatial_scale=scales[0], sampling_ratio=sampling_ratio)])
        self.output_size = output_size

    def forward(self, x, boxes, bin_step=1):
        """boxes: list of BoxLists (poolers.py:88-105) or an [K,5] RoI table (batch index, x1, y1, x2, y2) already on the device"""
        rois = boxes if torch.is_tensor(boxes) else convert_to_roi_format(boxes)
        return self.poolers[0](x[0], rois, bin_step)


class ResNet50Conv5ROIFeatureExtractor(nn.Module):
    def __init__(self, config, in_channels):
        super().__init__()
        res = config.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler = Pooler((res, res), config.MODEL.ROI_BOX_HEAD.POOLER_SCALES, config.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO)
        self.head = ResNetHead(stage_index=4, block_count=3, width_per_group=config.MODEL.RESNETS.WIDTH_PER_GROUP,
                               res2_out_channels=config.MODEL.RESNETS.RES2_OUT_CHANNELS)
        self.out_channels = self.head.out_channels
        self.resolution = res

    def forward(self, x, proposals, need_roi_features=True):
        """-> (head features logical [K,2048,4,4], roi_align_features).  need_roi_features=False lets ROIAlign skip the bins
        layer4 never reads; roi_align_features is then the [K,1024,4,4] even-bin sub-grid instead of [K,1024,7,7]."""
        sparse = (not need_roi_features) and self.resolution % 2 == 1 and list(self.head.layer4)[0].stride == 2
        roi_align_features = self.pooler(x, proposals, bin_step=2 if sparse else 1)
        x = self.head(roi_align_features, first_stride=1 if sparse else None)
        return x, roi_align_features


    def pool_soft_early(self, x, soft_proposals):
        """the distillation RoIs' pooling, ahead of the detection RoIs' (see _JointPoolFn.forward): -> (RoI table, pooled [Ks,7,7,C] NHWC)"""
        soft_rois = soft_proposals if torch.is_tensor(soft_proposals) else convert_to_roi_format(soft_proposals)
        al = self.pooler.poolers[0]
        ph, pw = al.output_size if isinstance(al.output_size, (tuple, list)) else (al.output_size, al.output_size)
        with torch.no_grad():
            pooled = ops.roi_align_forward(as_nhwc(x[0]), soft_rois, al.spatial_scale, ph, pw, al.sampling_ratio, 1)
        return soft_rois, pooled

    def forward_joint(self, x, det_proposals, soft_proposals, soft_early=None):
        """Detection RoIs and the source model's distillation RoIs through ONE layer4 pass (they share weights and are independent
        rows of every GEMM): the detection RoIs are pooled on the even bins only, the distillation RoIs on all 7x7 bins (ARD reads
        them) and then sub-sampled the way layer4's stride-2 1x1 convs would.  -> (head features [Kd+Ks,2048,4,4], detection
        pooled [Kd,1024,4,4], distillation pooled [Ks,1024,7,7])"""
        assert self.resolution % 2 == 1 and list(self.head.layer4)[0].stride == 2
        det_rois = det_proposals if torch.is_tensor(det_proposals) else convert_to_roi_format(det_proposals)
        soft_rois, soft_ready = soft_early if soft_early is not None else (
            soft_proposals if torch.is_tensor(soft_proposals) else convert_to_roi_format(soft_proposals), None)
        al = self.pooler.poolers[0]
        ops.mark("distillation RoIs pooled, proposal selection joined, RoI targets done (detection ROIAlign starts)")
        joint, soft = _JointPoolFn.apply(x[0], det_rois, soft_rois, al.output_size, al.spatial_scale, al.sampling_ratio, soft_ready)
        ops.mark("ROIAlign done (layer4 starts)")
        # (second value: the tensor whose gradient marks "layer4's backward is queued" for the gradient exchange hooks, engine/trainer.py::_arm_overlap:
        #  with the joint pass that is `joint` itself -- `soft` receives ARD's gradient long before layer4 has run its backward)
        joint._abr_joint_pool = True
        return self.head(joint, first_stride=1), joint, soft


# ------------------------------------------------------------------------------------------------ predictor
FUSE_POOL_RELU_BWD = os.environ.get("ABR_FUSE_POOL_RELU_BWD", "1") != "0"


class _PredictorFn(Function):
    @staticmethod
    def forward(ctx, x, pred, *params):
        xh = as_nhwc(x)                                   # [K,4,4,2048]
        pooled = ops.avgpool_forward(xh)                  # AdaptiveAvgPool2d(1), roi_box_predictors.py:28
        K_ = pooled.shape[0]
        y = ops.conv_forward(pooled.view(K_, 1, 1, -1), pred.fused_weight, 1, 0, bias=pred.fused_bias).view(K_, -1)
        ctx.pred, ctx.saved, ctx.xshape = pred, pooled, tuple(xh.shape)
        ctx.need_dx = x.requires_grad
        # x is layer4's output, i.e. a ReLU's: its backward is fused into the pooling's (one pass over the [K,4,4,2048] tensor instead of
        # three), and the gradient handed to layer4 says so (_StageFn.backward skips its own mask; masking twice would be harmless)
        ctx.relu_of = xh if (FUSE_POOL_RELU_BWD and x.requires_grad and getattr(x, "_abr_relu_output", False) and xh.is_contiguous()) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        pred, pooled = ctx.pred, ctx.saved
        K_ = pooled.shape[0]
        g = gy.contiguous().view(K_, 1, 1, -1)
        ops.conv_wgrad_async(pooled.view(K_, 1, 1, -1), g, pred.fused_weight_grad, 1, 0)
        ops.bias_grad(g, pred.fused_bias_grad)
        gx = None
        if ctx.need_dx:
            gp = ops.conv_forward(g, pred.fused_dgrad_weight(), 1, 0).view(K_, -1)
            gx = from_nhwc(ops.avgpool_backward(gp, ctx.xshape, relu_of=ctx.relu_of, emit_amax=True))   # (amax word for layer4's f16x3 backward)
            if ctx.relu_of is not None:
                # "layer4's final ReLU mask is already applied" travels as a TOKEN bound to this very storage and version: if the autograd
                # engine accumulates another consumer's gradient into the tensor (in place: the version moves; out of place: a new tensor
                # without the attribute), _StageFn.backward no longer finds a matching token and applies the mask itself (idempotent on
                # this part, required for the other) -- the flag can never outlive the values it describes.
                gx._abr_relu_masked = (gx.data_ptr(), gx._version)
        ctx.saved = ctx.relu_of = None
        return (gx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _Linear(nn.Module):
    """nn.Linear stand-in: weight [out,in], bias [out] (reference names cls_score / bbox_pred)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))


class FastRCNNPredictor(nn.Module):
    def __init__(self, config, in_channels):
        super().__init__()
        num_inputs = in_channels
        num_classes = config.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        self.num_classes = num_classes
        self.num_bbox_reg_classes = 2 if config.MODEL.CLS_AGNOSTIC_BBOX_REG else num_classes
        self.cls_score = _Linear(num_inputs, num_classes)
        self.bbox_pred = _Linear(num_inputs, self.num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, mean=0, std=0.01)     # roi_box_predictors.py:21-25
        nn.init.normal_(self.bbox_pred.weight, mean=0, std=0.001)
        self.n_out = num_classes + self.num_bbox_reg_classes * 4
        self.n_out_pad = (self.n_out + 3) // 4 * 4
        self._fuse()

    def _fuse(self):
        K, R4, dev = self.num_classes, self.num_bbox_reg_classes * 4, self.cls_score.weight.device
        C_ = self.cls_score.weight.shape[1]
        fw = torch.zeros(self.n_out_pad, C_, device=dev)
        fb = torch.zeros(self.n_out_pad, device=dev)
        with torch.no_grad():
            fw[:K].copy_(self.cls_score.weight)
            fw[K:K + R4].copy_(self.bbox_pred.weight)
            fb[:K].copy_(self.cls_score.bias)
            fb[K:K + R4].copy_(self.bbox_pred.bias)
        self._set_fused(fw, fb, torch.zeros_like(fw), torch.zeros_like(fb))

    def _set_fused(self, fw, fb, gw, gb):
        K, R4 = self.num_classes, self.num_bbox_reg_classes * 4
        self._fw2d, self._gw2d = fw, gw
        self.fused_weight = fw.view(self.n_out_pad, 1, 1, -1)
        self.fused_bias = fb
        self.fused_weight_grad, self.fused_bias_grad = gw.view(self.n_out_pad, 1, 1, -1), gb
        for mod, lo, hi in ((self.cls_score, 0, K), (self.bbox_pred, K, K + R4)):
            mod.weight.data, mod.bias.data = fw[lo:hi], fb[lo:hi]
            mod.weight.grad, mod.bias.grad = gw[lo:hi], gb[lo:hi]
        self._wt, self._wt_version = None, -1

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._fuse()
        return out

    def flat_groups(self):
        return [("weight", [self.cls_score.weight, self.bbox_pred.weight], self.n_out_pad - self.n_out),
                ("bias", [self.cls_score.bias, self.bbox_pred.bias], self.n_out_pad - self.n_out)]

    def rehome(self, which, view, grad_view):
        """called by flatten_parameters: adopt flat-buffer storage for the fused weight / bias"""
        if which == "weight":
            self._set_fused(view.view(self.n_out_pad, -1), self.fused_bias, grad_view.view(self.n_out_pad, -1), self.fused_bias_grad)
        else:
            self._set_fused(self._fw2d, view, self._gw2d, grad_view)

    def prepare_derived(self):
        """see Bottleneck.prepare_derived"""
        if self.fused_weight.is_cuda and self.cls_score.weight.requires_grad:
            self.fused_dgrad_weight()

    def fused_dgrad_weight(self):
        ops.prep_wait()
        if self._wt is None or self._wt_version != _PARAM_VERSION[0]:
            self._wt = ops.conv_dgrad_weights(self.fused_weight, None, out=self._wt)
            self._wt_version = _PARAM_VERSION[0]
        return self._wt

    def forward_fused(self, x):
        params = list(self.parameters())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _PredictorFn.apply(x, self, *params)
        pooled = ops.avgpool_forward(as_nhwc(x))
        K_ = pooled.shape[0]
        return ops.conv_forward(pooled.view(K_, 1, 1, -1), self.fused_weight, 1, 0, bias=self.fused_bias).view(K_, -1)

    def forward(self, x):
        """-> (cls_logit [K,K_all], bbox_pred [K,4*K_all]) as column slices of the fused output"""
        y = self.forward_fused(x)
        K = self.num_classes
        return y[:, :K], y[:, K:K + self.num_bbox_reg_classes * 4]


# ------------------------------------------------------------------------------------------------ loss
class _BoxHeadLossFn(Function):
    """classification (plain CE or the inclusive loss of dist_type=='id') + class-specific smooth-L1(beta=1)/N, both reading
    the fused predictor output and writing ONE fused gradient buffer (box_head/loss.py:151-179)."""

    @staticmethod
    def forward(ctx, fused, K, labels, regression_targets, inclusive, n_old, cls_agnostic, prepared=None):
        """prepared = (pos_rows, col0, n_valid) from ops.roi_head_targets: the regression rows / columns and the device-resident
        number of sampled RoIs (rows with label -1 are padding: both loss kernels skip them)"""
        n = fused.shape[0]
        want = fused.requires_grad
        grad = torch.zeros_like(fused) if want else None
        lc, _ = ops.softmax_ce(fused[:, :K], labels, inclusive, n_old, want_grad=want, grad_out=grad[:, :K] if want else None)
        if prepared is not None:
            pos, col0, n_valid = prepared
            lb, gb = ops.smooth_l1_rows(fused, regression_targets, pos, col0, 1.0, scale=1.0, want_grad=want, denom_dev=n_valid)
        else:
            # rows with label > 0 (:166) as a fixed-size list: non-positive rows become -1 and are skipped by the kernel (no nonzero() sync)
            ar = torch.arange(n, device=labels.device)
            pos = torch.where(labels > 0, ar, torch.full_like(ar, -1))
            col0 = K + (4 * labels.clamp(min=0) if not cls_agnostic else torch.full_like(ar, 4))  # :168-171
            lb, gb = ops.smooth_l1_rows(fused, regression_targets, pos, col0, 1.0, scale=1.0 / max(n, 1), want_grad=want)
        if want:
            ops.add_(grad, gb)
        ctx.save_for_backward(grad)
        ctx.K = K
        ctx.g_cls_cols = K
        return lc[0], lb[0]

    @staticmethod
    def backward(ctx, g_lc, g_lb):
        (grad,) = ctx.saved_tensors
        # the two losses enter the total with the same weight in every caller; honour distinct upstream scales anyway
        K = ctx.K
        gc = grad[:, :K].contiguous()
        ops.scale_(gc, 1.0, g_lc.contiguous())
        gr = grad[:, K:].contiguous()
        ops.scale_(gr, 1.0, g_lb.contiguous())
        return torch.cat((gc, gr), 1), None, None, None, None, None, None, None


class FastRCNNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, cls_agnostic_bbox_reg=False, dist_type=None, old_classes=()):
        self.proposal_matcher, self.fg_bg_sampler, self.box_coder = proposal_matcher, fg_bg_sampler, box_coder
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg
        self.dist_type = dist_type
        self.n_old_cl = len(old_classes)

    def prepare_targets(self, proposals, targets):
        """box_head/loss.py:56-84: labels int64 (class / 0 background / -1 ignore) + encoded regression targets"""
        labels, regression_targets = [], []
        for p, t in zip(proposals, targets):
            _, lab, tgt = self.proposal_matcher.match_boxes(t.bbox, p.bbox, t.get_field("labels").to(torch.int64), None,
                                                            self.box_coder.weights, rpn_labels=False)
            labels.append(lab)
            regression_targets.append(tgt)
        return labels, regression_targets

    def subsample_fused(self, lazy, targets, num_classes):
        """`subsample` for the training selector's raw output (rpn.LazyProposals): GT append, matching, labels, encoding, sampling and
        the RoI table for the whole batch in three launches without a host round trip (ops.roi_head_targets).  Every image gets
        BATCH_SIZE_PER_IMAGE rows; rows past the number actually drawn (only when an image has fewer candidates than that) are
        padding with label -1, which the loss kernels skip.  Returns the dict of device tensors; `self._proposals` are BoxList VIEWS of it."""
        props, scores, keep, n_keep, sizes = lazy.raw()
        if _DBG_SLEEP_ROI_TARGETS:
            torch.cuda._sleep(_DBG_SLEEP_ROI_TARGETS)        # (criticality probe: tools/dbg/critical_probe.sh)
        R = self.fg_bg_sampler.batch_size_per_image
        t = ops.roi_head_targets(props, scores, keep, n_keep, [g.bbox for g in targets], [g.get_field("labels") for g in targets],
                                 self.proposal_matcher.high_threshold, self.proposal_matcher.low_threshold, self.box_coder.weights, R,
                                 int(R * self.fg_bg_sampler.positive_fraction), num_classes, self.cls_agnostic_bbox_reg)
        out = []
        for i, size in enumerate(sizes):
            lo, hi = i * R, (i + 1) * R
            b = BoxList(t["rois"][lo:hi, 1:5], size, mode="xyxy")
            b.add_field("objectness", t["obj"][lo:hi])
            b.add_field("labels", t["labels"][lo:hi])
            b.add_field("regression_targets", t["reg_targets"][lo:hi])
            out.append(b)
        self._proposals, self._fused_targets, self._lazy = out, t, lazy
        self.last_sampled_inds = [t["sampled_idx"][i] for i in range(len(sizes))]
        return t

    @property
    def last_input_proposals(self):
        """the post-NMS + GT lists the last `subsample` drew from (introspection for parity tests; reads counts back)"""
        lazy = getattr(self, "_lazy", None)
        return lazy.materialize() if lazy is not None else getattr(self, "_last_input_proposals", None)

    @last_input_proposals.setter
    def last_input_proposals(self, v):
        self._last_input_proposals, self._lazy = v, None

    def subsample(self, proposals, targets, sampled_inds=None):
        """:86-120.  Keeps state (self._proposals).  `sampled_inds` (list of index tensors) injects the sampler's choice."""
        labels, regression_targets = self.prepare_targets(proposals, targets)
        if sampled_inds is None:
            sampled_inds = getattr(self, "inject_sampled_inds", None)  # parity tests pin the sampler's draw here
        proposals = list(proposals)
        self.last_input_proposals = list(proposals)   # introspection for parity tests (the post-NMS + GT lists, before sampling)
        if sampled_inds is None:
            # fused sampler: one launch per image (ragged proposal counts), all counts fetched with ONE host sync
            drawn = [self.fg_bg_sampler.sample_padded(lab) for lab in labels]
            cnt = torch.cat([d[2] for d in drawn]).tolist()
            sampled_inds = [torch.cat((d[0][0, :c[0]], d[1][0, :c[1]])).sort()[0] for d, c in zip(drawn, cnt)]  # ascending (:114)
        self.last_sampled_inds = sampled_inds  # (tests replay a draw through inject_sampled_inds)
        for i, (lab, tgt, p) in enumerate(zip(labels, regression_targets, proposals)):
            p.add_field("labels", lab)
            p.add_field("regression_targets", tgt)
            proposals[i] = p[sampled_inds[i]]
        self._proposals, self._fused_targets = proposals, None
        return proposals

    def __call__(self, class_logits, box_regression, fused=None):
        if not hasattr(self, "_proposals"):
            raise RuntimeError("subsample needs to be called before")
        t = getattr(self, "_fused_targets", None)
        if t is not None:    # subsample_fused: the batch's labels / targets / loss rows already are single tensors
            labels, regression_targets, prepared = t["labels"], t["reg_targets"], (t["pos_rows"], t["col0"], t["n_valid"])
        else:
            proposals = self._proposals
            labels = torch.cat([p.get_field("labels") for p in proposals], dim=0)
            regression_targets = torch.cat([p.get_field("regression_targets") for p in proposals], dim=0)
            prepared = None
        if fused is None:
            fused = torch.cat((torch.cat(class_logits, 0), torch.cat(box_regression, 0)), 1)
            K = class_logits[0].shape[1]
        else:
            K = class_logits
        return _BoxHeadLossFn.apply(fused, K, labels, regression_targets, self.dist_type == "id", self.n_old_cl, self.cls_agnostic_bbox_reg,
                                    prepared)


def make_roi_box_loss_evaluator(cfg):
    matcher = Matcher(cfg.MODEL.ROI_HEADS.FG_IOU_THRESHOLD, cfg.MODEL.ROI_HEADS.BG_IOU_THRESHOLD, allow_low_quality_matches=False)
    box_coder = BoxCoder(weights=cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS)
    sampler = BalancedPositiveNegativeSampler(cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION)
    return FastRCNNLossComputation(matcher, sampler, box_coder, cfg.MODEL.CLS_AGNOSTIC_BBOX_REG, cfg.DIST.TYPE,
                                   cfg.MODEL.ROI_BOX_HEAD.NAME_OLD_CLASSES)


# ------------------------------------------------------------------------------------------------ head
from .inference import make_roi_box_post_processor  # noqa: E402


class ROIBoxHead(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.feature_extractor = ResNet50Conv5ROIFeatureExtractor(cfg, in_channels)
        self.predictor = FastRCNNPredictor(cfg, self.feature_extractor.out_channels)
        self.loss_evaluator = make_roi_box_loss_evaluator(cfg)
        self.post_processor = make_roi_box_post_processor(cfg)
        self.need_roi_features_in_training = False

    def forward(self, features, proposals, targets=None):
        """training: -> (x, proposals, (class_logits, box_regression[K,K_all,4]), loss dict, roi_align_features)
        eval: -> (x, detections, results_background)  (box_head.py:24-58)"""
        if not self.training:
            x, _ = self.feature_extractor(features, proposals, need_roi_features=False)
            fused = self.predictor.forward_fused(x)
            K = self.predictor.num_classes
            result, results_background = self.post_processor(
                (fused[:, :K], fused[:, K:K + 4 * self.predictor.num_bbox_reg_classes]), proposals)
            return x, result, results_background
        K = self.predictor.num_classes
        ev = self.loss_evaluator
        with torch.no_grad():
            if hasattr(proposals, "raw") and getattr(ev, "inject_sampled_inds", None) is None:
                # the training selector's raw output: everything between NMS and ROIAlign stays on the device
                rois = ev.subsample_fused(proposals, targets, K)["rois"]
                proposals = ev._proposals
            else:
                proposals = ev.subsample(proposals, targets)
                rois = proposals
        x, roi_align_features = self.feature_extractor(features, rois, need_roi_features=self.need_roi_features_in_training)
        fused = self.predictor.forward_fused(x)
        loss_classifier, loss_box_reg = self.loss_evaluator(K, None, fused=fused)
        class_logits, box_regression = fused[:, :K], fused[:, K:K + 4 * self.predictor.num_bbox_reg_classes]
        return (x, proposals, (class_logits, box_regression.reshape(-1, K, 4)),
                dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg), roi_align_features)

    def forward_joint(self, features, proposals, targets, soften_proposals):
        """training forward (as `forward`) AND the second RoI pass on `soften_proposals` (as `calculate_soften_label`) in one
        trip through the head: -> (forward's 5-tuple, (soften_scores, soften_bboxes, roi_align_features [Ks,1024,7,7]))."""
        K = self.predictor.num_classes
        ev = self.loss_evaluator
        # the distillation RoIs' pooling needs the features and the SOURCE's proposals only: issued before the main stream starts waiting for the
        # target's own proposal selection (inside subsample_fused), it runs in that wait instead of behind it
        soft_early = self.feature_extractor.pool_soft_early(features, soften_proposals) if features[0].is_cuda else None
        with torch.no_grad():
            if hasattr(proposals, "raw") and getattr(ev, "inject_sampled_inds", None) is None:
                rois = ev.subsample_fused(proposals, targets, K)["rois"]    # (as `forward`: everything between NMS and ROIAlign stays on the device)
                proposals = ev._proposals
            else:
                proposals = ev.subsample(proposals, targets)
                rois = proposals
        x, raf_det, raf_soft = self.feature_extractor.forward_joint(features, rois, soften_proposals, soft_early=soft_early)
        fused = self.predictor.forward_fused(x)
        K, R4 = self.predictor.num_classes, 4 * self.predictor.num_bbox_reg_classes
        kd = sum(len(p) for p in proposals)
        det, soft = fused[:kd], fused[kd:]
        loss_classifier, loss_box_reg = self.loss_evaluator(K, None, fused=det)
        first = (x[:kd], proposals, (det[:, :K], det[:, K:K + R4].reshape(-1, K, 4)),
                 dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg), raf_det)
        return first, (soft[:, :K], soft[:, K:K + R4].reshape(-1, K, 4), raf_soft)

    def calculate_soften_label(self, features, proposals, targets=None):
        """box_head.py:60-78 -> (soften_scores [K,Kc], soften_bboxes [K,Kc,4], x, roi_align_features [K,1024,7,7])"""
        x, roi_align_features = self.feature_extractor(features, proposals, need_roi_features=True)
        class_logits, box_regression = self.predictor(x)
        return class_logits, box_regression.reshape(-1, class_logits.shape[1], 4), x, roi_align_features


def build_roi_box_head(cfg, in_channels):
    return ROIBoxHead(cfg, in_channels)
