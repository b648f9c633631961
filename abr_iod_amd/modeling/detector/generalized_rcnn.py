"""GeneralizedRCNN (mirror of maskrcnn_benchmark/modeling/detector/generalized_rcnn.py:22-175):
backbone -> rpn -> roi_heads orchestration with the reference's three entry points, argument lists and return tuples:

    forward(images, targets=None, rpn_output_source=None, features=None, proposals=None)                       (:50-95)
        training, full pass  -> (losses, features, backbone_features, anchors, rpn_output, proposals,
                                 roi_align_features, soften_results)                                            (:93)
        features+proposals   -> ((target_scores, target_bboxes), mask_logits, roi_align_features)               (:66-68)
    generate_soften_proposal(images, targets=None)
                             -> ((soften_scores, soften_bboxes), mask_logits, all_selected_proposals, features,
                                 backbone_features, anchors, rpn_output, roi_align_features)                    (:121-167)
    generate_feature_logits_by_targets(images, targets=None)
                             -> ((target_scores, target_bboxes), mask_logits, features, backbone_features,
                                 roi_align_features)                                                            (:169-175)
"""
import os
import random

import torch
from torch import nn

from ... import ops

from ...structures.bounding_box import BoxList
from ...structures.image_list import to_image_list
from ..backbone.backbone import build_backbone
from ..roi_heads.roi_heads import build_roi_heads
from ..rpn.rpn import build_rpn
from .._flat import flatten_parameters


# Default contraction arithmetic (round 5): f16x3 -- fp32-accurate contractions as three fp16 products per multiply-add, operands scaled by their
# amax (admitted by tests/test_gpu_f16x3_admission.py at the bounds bf16x6 was admitted with; guarded: engine/trainer.py::_x6_guard moves the
# models to bf16x6 when a large share of the operands leaves its domain and to the fp32 MFMA kernels on inf / nan).  ABR_CONV_MATH=bf16x6 selects
# rounds 2-4's default (exact three-term bf16 split, six products), ABR_CONV_MATH=f32 the fp32 MFMA kernels.
DEFAULT_CONV_MATH = "f16x3"
# source model's distillation proposals gathered from the selector's raw output by one kernel (GeneralizedRCNN._soften_fused)
FUSED_SOFTEN = os.environ.get("ABR_FUSED_SOFTEN", "1") != "0"


def _drop_derived_cache(base, nbytes):
    """weakref finalizer of a model's flat parameter storage (see flatten_parameters): only what was derived from THIS storage goes -- the
    garbage collector may run this on any thread while another model's conv call is in flight"""
    try:
        from ... import ops
        ops.conv_cache_drop_range(base, nbytes)
    except Exception:   # never raise from a finalizer
        pass


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.incremental = cfg.INCREMENTAL
        self.n_old_cl = len(cfg.MODEL.ROI_BOX_HEAD.NAME_OLD_CLASSES)
        self.n_new_cl = len(cfg.MODEL.ROI_BOX_HEAD.NAME_NEW_CLASSES)
        assert not cfg.MODEL.RPN.EXTERNAL_PROPOSAL, "external (EdgeBoxes) proposals are outside the hot path"
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.roi_heads = build_roi_heads(cfg, self.backbone.out_channels)
        # cfg.DTYPE == "bfloat16" = BASELINE.json configs[4] "bf16 MFMA backbone" (backbone/resnet.py).  ABR_BF16_SCOPE=all extends the
        # bf16 contractions to the RPN head and layer4 as well (the predictor FCs and every loss stay fp32).
        if cfg.DTYPE == "bfloat16" and os.environ.get("ABR_BF16_SCOPE", "backbone") == "all":
            from ..backbone.resnet import set_conv_math
            set_conv_math(self.rpn, ops.MATH_BF16)
            set_conv_math(self.roi_heads, ops.MATH_BF16)
        # Contraction arithmetic of every bottleneck / RPN conv (cfg.DTYPE float32): "bf16x6" = fp32-ACCURATE contractions on the bf16
        # matrix cores -- each operand split exactly into three bf16 terms, six cross products, fp32 accumulate (csrc/conv_igemm.hip),
        # guarded by a hardware range check (ops.x6_range_flags; engine/trainer.py falls back to "f32" when it trips) -- or "f32" =
        # v_mfma_f32_32x32x2_f32.  ABR_CONV_MATH overrides the default.
        # "f16x3" (round 5) = the same on a two-term fp16 split with three products, operands scaled by their amax (csrc/common.h).
        self.conv_math = "f32"
        self.bf16_backbone = False     # cfg.DTYPE "bfloat16" with the backbone only in bf16: set_conv_math leaves the backbone's arithmetic alone
        env_math = os.environ.get("ABR_CONV_MATH", DEFAULT_CONV_MATH)
        if env_math not in ("f32", "bf16x6", "f16x3"):
            raise ValueError("ABR_CONV_MATH must be f32, bf16x6 or f16x3, got {!r}".format(env_math))
        want_x6 = env_math in ("bf16x6", "f16x3")
        if cfg.DTYPE == "float32" and want_x6:
            self.set_conv_math(env_math)
        elif cfg.DTYPE == "bfloat16" and want_x6 and os.environ.get("ABR_BF16_SCOPE", "backbone") != "all":
            # "bf16 MFMA backbone" (configs[4]): layer1-3 contract in bf16, EVERYTHING ELSE in the default arithmetic -- until round 4 the RPN head and
            # layer4 of this mode were left on the fp32 MFMA kernels (round 1's default), which is what made the mode 20 % slower than bf16x6
            from ..backbone.resnet import set_conv_math
            set_conv_math(self.rpn, ops.MATH_F16X3 if env_math == "f16x3" else ops.MATH_BF16X6)
            set_conv_math(self.roi_heads, ops.MATH_F16X3 if env_math == "f16x3" else ops.MATH_BF16X6)
            self.conv_math = env_math      # (the range guard of engine/trainer.py watches the fp32-accurate part)
            self.bf16_backbone = True
        self.flat = None

    def set_conv_math(self, name):
        """'f32', 'bf16x6' or 'f16x3' for every conv of the backbone, RPN head and layer4 head (takes effect at the next call)"""
        from ..backbone.resnet import set_conv_math
        math = {"f32": ops.MATH_F32, "bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[name]
        # (a "bf16 MFMA backbone" -- configs[4] -- keeps its own arithmetic when the guard moves the rest: rounding to bf16 is defined for every
        #  finite value, and the mode would otherwise silently stop being what its name says)
        for m in ((self.rpn, self.roi_heads) if self.bf16_backbone else (self.backbone, self.rpn, self.roi_heads)):
            set_conv_math(m, math)
        self.conv_math = name

    # --- storage: one flat fp32 buffer for parameters, one for gradients (RCCL all-reduce + fused SGD work on them)
    def flatten_parameters(self):
        old = getattr(self, "flat", None)
        old_range = (old.params.data_ptr(), old.params.numel() * old.params.element_size()) if (old is not None and old.params.is_cuda) else None
        self.flat = flatten_parameters(self)
        if old_range is not None:
            from ... import ops
            ops.conv_cache_drop_range(*old_range)   # the old storage is gone: entries keyed by its addresses would outlive it (and may alias new tensors)
        if self.flat.params.is_cuda:
            # the library keeps packed planes / Winograd-domain copies per weight ADDRESS (raw hipMalloc, outside torch's caching allocator): drop
            # them when this storage dies (a model that is deleted, a long pytest session building model after model), not only when it is rebuilt
            import weakref
            fin = weakref.finalize(self.flat, _drop_derived_cache, self.flat.params.data_ptr(), self.flat.params.numel() * self.flat.params.element_size())
            fin.atexit = False   # not during interpreter shutdown (the runtime may already be gone)
        from ..backbone.resnet import Conv2d, bump_param_version
        bump_param_version()   # new weight storage: nothing derived from an earlier tensor at the same address may be reused
        for m in self.modules():
            if isinstance(m, Conv2d):
                m._flat = self.flat
        return self.flat

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if self.flat is not None and next(self.parameters()).device != self.flat.params.device:
            # a device move: NEW flat buffers (an optimiser built on the old ones refuses to step: FusedSGD.step), convs re-pointed
            self.flatten_parameters()
        return out

    def forward(self, images, targets=None, rpn_output_source=None, features=None, proposals=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        if features is not None and proposals is not None:
            target_scores, target_bboxes, mask_logits, roi_align_features = self.roi_heads.calculate_soften_label(features, proposals)
            return (target_scores, target_bboxes), mask_logits, roi_align_features
        if self.training:
            return self.forward_finish(self.forward_begin(images, targets, rpn_output_source))
        images = to_image_list(images)
        features, backbone_features = self.backbone(images.tensors)
        (proposals, proposal_losses), anchors, rpn_output = self.rpn(images, features, targets, rpn_output_source)
        # generalized_rcnn.py:76-78 -> (detections, features, background detections)
        x, result, results_background, _ = self.roi_heads(features, proposals, targets)
        return result, features, results_background

    def prefetch_frozen(self, images):
        """The frozen stem + leading frozen stages on `images`, on the current stream (see ResNet.frozen_prefix); hand the result to
        forward_begin(..., prefix=) for the same batch."""
        images = to_image_list(images)
        return self.backbone.frozen_prefix(images.tensors)

    def forward_begin(self, images, targets, rpn_output_source=None, prefix=None):
        """Training forward up to the point where the host needs the proposal counts: backbone, RPN head, RPN loss, and the
        proposal selection in flight on its side stream.  The trainer slots the source model's head pass between `forward_begin`
        and `forward_finish`, so the selection (and the host's read-back of its counts) hides behind real work."""
        images = to_image_list(images)
        features, backbone_features = self.backbone(images.tensors, prefix)
        return dict(features=features, backbone_features=backbone_features, targets=targets,
                    rpn=self.rpn.forward_begin(images, features, targets, rpn_output_source))

    def forward_finish(self, state, soften_proposals=None):
        """`soften_proposals` (the source model's distillation RoIs): they ride along with the detection RoIs through ONE layer4 / predictor pass
        (train_incremental.py:89-95 as forward_joint does it); the result is then (the 8-tuple, the second call's 3-tuple)."""
        features, targets = state["features"], state["targets"]
        (proposals, proposal_losses), anchors, rpn_output = self.rpn.forward_finish(state["rpn"])
        ops.mark("RPN finish returned (the selection is still in flight on its stream: the RoI targets join it)")
        second = None
        if soften_proposals is not None:
            (x, result, soften_results, detector_losses, roi_align_features), (t_scores, t_bboxes, mask_logits, t_raf) = \
                self.roi_heads.forward_joint(features, proposals, targets, soften_proposals)
            second = ((t_scores, t_bboxes), mask_logits, t_raf)
        else:
            x, result, soften_results, detector_losses, roi_align_features = self.roi_heads(features, proposals, targets)
        ops.mark("RoI targets + ROIAlign + layer4 + predictor + box losses done")
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        first = (losses, features, state["backbone_features"], anchors, rpn_output, result, roi_align_features, soften_results)
        return first if second is None else (first, second)

    def forward_joint(self, images, targets, soften_proposals, rpn_output_source=None):
        """`forward(images, targets)` followed by `forward(images, targets, features=..., proposals=soften_proposals)`
        (train_incremental.py:89-95) as ONE pass: the 64 distillation RoIs per image ride along with the 512 detection RoIs through
        layer4 and the predictor.  Returns (the training 8-tuple, the second call's 3-tuple); same values as the two calls."""
        if targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        features, backbone_features = self.backbone(images.tensors)
        (proposals, proposal_losses), anchors, rpn_output = self.rpn(images, features, targets, rpn_output_source)
        (x, result, soften_results, detector_losses, roi_align_features), (t_scores, t_bboxes, mask_logits, t_raf) = \
            self.roi_heads.forward_joint(features, proposals, targets, soften_proposals)
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return ((losses, features, backbone_features, anchors, rpn_output, result, roi_align_features, soften_results),
                ((t_scores, t_bboxes), mask_logits, t_raf))

    def generate_soften_proposal(self, images, targets=None, selected_indices=None):
        """Source-model pass (model.eval(), under no_grad in the trainer): top-128 by objectness, python `random.sample`
        picks 64 (:140-149).  `selected_indices` (list of index lists) injects that choice for parity tests."""
        if not self.training and targets is None:
            return self.soften_finish(self.soften_begin(images, defer=False), selected_indices)
        images = to_image_list(images)
        features, backbone_features = self.backbone(images.tensors)
        (all_proposals, _), anchors, rpn_output = self.rpn(images, features, targets)
        return self._soften_from_proposals(all_proposals, features, backbone_features, anchors, rpn_output, selected_indices)

    def soften_begin(self, images, defer=True, prefix=None):
        """First half of generate_soften_proposal for the frozen source model: backbone + RPN head on the current stream, the
        proposal selection (top-k, decode, NMS) on a side stream with nothing read back.  The trainer enqueues the target model's
        forward between `soften_begin` and `soften_finish`, so the selection's latency-bound kernels hide behind the target's
        backbone convolutions."""
        assert not self.training, "the soften pass runs the source model in eval mode (train_incremental.py:80)"
        images = to_image_list(images)
        # `prefix`: frozen_prefix() of a backbone whose frozen stem / stages were verified identical to this one's (trainer.frozen_prefix_shareable)
        features, backbone_features = self.backbone(images.tensors, prefix) if prefix is not None else self.backbone(images.tensors)
        (pending, _), anchors, rpn_output = self.rpn(images, features, None, defer_proposals=defer and images.tensors.is_cuda)
        return dict(features=features, backbone_features=backbone_features, pending=pending, anchors=anchors, rpn_output=rpn_output)

    def soften_finish(self, state, selected_indices=None):
        pending = state["pending"]
        if isinstance(pending, dict):   # deferred: join the side stream, read the keep counts
            sel = self.rpn.box_selector_test
            if FUSED_SOFTEN and pending["props"].is_cuda:
                return self._soften_fused(sel, pending, state, selected_indices)
            pending = sel.collect(pending)
        return self._soften_from_proposals(pending, state["features"], state["backbone_features"], state["anchors"],
                                           state["rpn_output"], selected_indices)

    def _pick_soften(self, n, k, selected_indices):
        """generalized_rcnn.py:140-149: python's random.sample over the top-128 (or all n < 128) of the ranked list"""
        if selected_indices is None and getattr(self, "inject_soften_indices", None) is not None:
            selected_indices = self.inject_soften_indices     # parity tests pin python's random.sample here
        if selected_indices is not None:
            return list(selected_indices[k])
        if n < 64:
            return random.sample(range(0, n, 1), n)
        if n < 128:
            return random.sample(range(0, n, 1), 64)
        return random.sample(range(0, 128, 1), 64)

    def _soften_fused(self, sel, pending, state, selected_indices):
        """The 64 distillation proposals per image straight from the selector's raw output: the keep counts are read on the selection's
        own stream (it finished long ago: the target's backbone ran in between), the picks go up in one pinned asynchronous copy and ONE
        gather kernel builds the RoI table -- instead of cutting per-image BoxLists, sorting them (the post-NMS list already is in
        descending objectness order) and indexing them field by field.  Falls back to the BoxList path for ragged pick counts."""
        side = pending.get("stream")
        if side is not None:
            with torch.cuda.stream(side):
                nk = pending["n_keep"].tolist()
        else:
            nk = pending["n_keep"].tolist()
        picks = [self._pick_soften(n, k, selected_indices) for k, n in enumerate(nk)]
        P = len(picks[0]) if picks else 0
        if not picks or any(len(p) != P for p in picks) or P == 0:
            return self._soften_from_proposals(sel.collect(pending), state["features"], state["backbone_features"], state["anchors"],
                                               state["rpn_output"], picks)
        sel.join(pending)
        self.last_soften_indices = picks
        dev = pending["props"].device
        rois, obj = ops.gather_proposals(pending["props"], pending["scores"], pending["keep"], ops.h2d([i for p in picks for i in p], torch.int64, dev), P)
        ready = torch.cuda.Event()
        ready.record()          # the RoI table exists from here on: the trainer starts the target's distillation pass behind this event,
        all_selected = []       # not behind the source model's whole head pass
        for k, size in enumerate(pending["sizes"]):
            b = BoxList(rois[k * P:(k + 1) * P, 1:5], size, mode="xyxy")
            b.add_field("objectness", obj[k * P:(k + 1) * P])
            b._roi_table = (rois, k)     # lets Pooler.convert_to_roi_format hand the table back instead of re-assembling it
            b._roi_ready = ready
            all_selected.append(b)
        soften_scores, soften_bboxes, mask_logits, roi_align_features = self.roi_heads.calculate_soften_label(state["features"], rois)
        return ((soften_scores, soften_bboxes), mask_logits, all_selected, state["features"], state["backbone_features"], state["anchors"],
                state["rpn_output"], roi_align_features)

    def _soften_from_proposals(self, all_proposals, features, backbone_features, anchors, rpn_output, selected_indices=None):
        picks = [self._pick_soften(len(props), k, selected_indices) for k, props in enumerate(all_proposals)]
        # one pinned, asynchronous upload for the whole batch's picks (a pageable torch.tensor(..., device=) per image would make the
        # host wait for everything queued on the stream -- here, the target's entire forward)
        dev = all_proposals[0].bbox.device if all_proposals else None
        flat_idx = ops.h2d([i for sel in picks for i in sel], torch.int64, dev) if picks else None
        self.last_soften_indices = picks   # (tests replay a draw through inject_soften_indices)
        all_selected, off = [], 0
        for props, sel in zip(all_proposals, picks):
            order = props.get_field("objectness").sort(descending=True)[1]
            props = props[order]
            idx = flat_idx[off:off + len(sel)]
            off += len(sel)
            chosen = BoxList(props.bbox.index_select(0, idx).view(-1, 4), props.size, props.mode)
            chosen.add_field("objectness", props.get_field("objectness").index_select(0, idx).view(-1))
            all_selected.append(chosen)
        soften_scores, soften_bboxes, mask_logits, roi_align_features = self.roi_heads.calculate_soften_label(features, all_selected)
        return (soften_scores, soften_bboxes), mask_logits, all_selected, features, backbone_features, anchors, rpn_output, roi_align_features

    def generate_feature_logits_by_targets(self, images, targets=None):
        images = to_image_list(images)
        features, backbone_features = self.backbone(images.tensors)
        target_scores, target_bboxes, mask_logits, roi_align_features = self.roi_heads.calculate_soften_label(features, targets)
        return (target_scores, target_bboxes), mask_logits, features, backbone_features, roi_align_features


def build_detection_model(cfg):
    """modeling/detector/detectors.py: META_ARCHITECTURE 'GeneralizedRCNN'.  The model is created on cfg.MODEL.DEVICE and its
    parameters are re-homed into flat buffers (see modeling/_flat.py)."""
    assert cfg.MODEL.META_ARCHITECTURE == "GeneralizedRCNN"
    model = GeneralizedRCNN(cfg)
    dev = torch.device(cfg.MODEL.DEVICE)
    if dev.type == "cuda":
        model.to(dev)
        model.flatten_parameters()
    return model
