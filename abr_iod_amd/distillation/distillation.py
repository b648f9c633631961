"""Distillation losses (mirror of maskrcnn_benchmark/distillation/distillation.py) for the hot path:

    calculate_attentive_roi_feature_distillation(f_map_s, f_map_t, gamma=1.0)                        (:86-130)
    calculate_roi_distillation_losses(soften_results, target_results, dist='l2', soften_proposal=None) (:223-240)
    calculate_roi_distillation_loss(...)                                                              (:164-220)

Same signatures, same quirks ((source, target) call order with swapped names; `temp` unused in the
softmax; ID loss divides by K_old), each loss ONE fused HIP kernel for the value and one for the gradient.
The ablation-only variants (`calculate_rpn_distillation_loss`, `calculate_feature_distillation_loss`) hard-code 'cuda'
tensors in the reference (:36,:149) and are next-tier.
"""
import torch
from torch.autograd import Function

from .. import _lib, ops
from ..layers._layout import as_nhwc, from_nhwc


class _ARDFn(Function):
    @staticmethod
    def forward(ctx, f_map_s, f_map_t, gamma):
        # the reference is called as (source, target): train_incremental.py:115.  Gradient flows into the 2nd argument only
        # when the 1st is detached (it is: the source pass runs under no_grad).
        nhwc = f_map_t.permute(0, 2, 3, 1).is_contiguous() and f_map_s.permute(0, 2, 3, 1).is_contiguous()
        if nhwc:
            fs, ft, layout = f_map_s.permute(0, 2, 3, 1), f_map_t.permute(0, 2, 3, 1), _lib.NHWC
        else:
            fs, ft, layout = f_map_s.contiguous(), f_map_t.contiguous(), _lib.NCHW
        loss, coef = ops.ard_forward(fs, ft, gamma, layout)
        ctx.save_for_backward(fs, ft, coef)
        ctx.gamma, ctx.layout, ctx.nhwc = gamma, layout, nhwc
        ctx.need_s = f_map_s.requires_grad
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        fs, ft, coef = ctx.saved_tensors
        if ctx.need_s:
            raise RuntimeError("ARD: gradient w.r.t. the first (source) feature map is not on the hot path (source is frozen)")
        grad = ops.ard_backward(fs, ft, coef, ctx.gamma, 1.0, g.contiguous(), ctx.layout)
        return None, (grad.permute(0, 3, 1, 2) if ctx.nhwc else grad), None


def calculate_attentive_roi_feature_distillation(f_map_s, f_map_t, gamma=1.0):
    """f_map_s, f_map_t: [N,C,H,W].  loss = afd + gamma*pad with the attention mask taken from f_map_s (the SOURCE)."""
    return _ARDFn.apply(f_map_s, f_map_t, float(gamma))


class _RoiDistillFn(Function):
    @staticmethod
    def forward(ctx, soften_scores, soften_bboxes, target_scores, target_bboxes, dist_id):
        want = target_scores.requires_grad or target_bboxes.requires_grad
        loss, d_zt, d_bt = ops.roi_distill(soften_scores, soften_bboxes, target_scores, target_bboxes, dist_id, want_grad=want)
        ctx.save_for_backward(d_zt, d_bt)
        ctx.bshape = target_bboxes.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d_zt, d_bt = ctx.saved_tensors
        g = g.contiguous()
        ops.scale_(d_zt, 1.0, g)
        ops.scale_(d_bt, 1.0, g)
        return None, None, d_zt, d_bt.view(ctx.bshape), None


def calculate_roi_distillation_loss(soften_results, target_results, cls_preprocess=None, cls_loss=None, bbs_loss=None,
                                    temperature=1, soften_proposal=None):
    soften_scores, soften_bboxes = soften_results
    target_scores, target_bboxes = target_results
    if cls_loss == "unbiased-cross-entropy" and cls_preprocess == "inclusive_distillation" and bbs_loss == "l2":
        dist_id = True
    elif cls_loss == "l2" and cls_preprocess == "normalization" and bbs_loss == "l2":
        dist_id = False
    else:
        raise ValueError("Wrong preprocessing / loss combination for RoI distillation (hot path: 'id' or 'l2')")
    return _RoiDistillFn.apply(soften_scores.detach(), soften_bboxes.detach(), target_scores, target_bboxes, dist_id)


def calculate_roi_distillation_losses(soften_results, target_results, dist="l2", soften_proposal=None):
    if dist == "id":
        if soften_proposal is not None:
            # distillation.py:225-229: `cls_preprocess` is undefined on this branch -> UnboundLocalError in the reference
            raise UnboundLocalError("local variable 'cls_preprocess' referenced before assignment")
        return calculate_roi_distillation_loss(soften_results, target_results, "inclusive_distillation", "unbiased-cross-entropy",
                                               "l2", 1, soften_proposal)
    return calculate_roi_distillation_loss(soften_results, target_results, "normalization", "l2", "l2", 1, soften_proposal)


# ------------------------------------------------------------------------------------------------ ablation-only losses
class _FeatDistillFn(Function):
    @staticmethod
    def forward(ctx, src, tgt):
        loss, d = ops.feat_distill(src, tgt, want_grad=tgt.requires_grad)
        ctx.save_for_backward(d)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        ops.scale_(d, 1.0, g.contiguous())
        return None, d


def calculate_feature_distillation_loss(source_features, target_features, loss=None):
    """distillation.py:133-161 (DIST.FEAT == 'std'): per feature level mean(max((s - mean s) - (t - mean t), 0)), summed over levels.
    Both maps of a level must share one memory layout (they do: both are the models' channels-last C4 features)."""
    if len(source_features) != len(target_features):
        raise ValueError("Number of source features must equal to number of target features")
    if loss != "normalized_filtered_l1":
        raise ValueError("Wrong loss function for feature distillation")
    total = 0
    for s, t in zip(source_features, target_features):
        if s.stride() != t.stride():
            s = s.contiguous(); t = t.contiguous()
        total = total + _FeatDistillFn.apply(s.detach(), t)
    return total


class _RPNDistillFn(Function):
    @staticmethod
    def forward(ctx, obj_s, reg_s, obj_t, reg_t, thr, use_bbox):
        want = obj_t.requires_grad or reg_t.requires_grad
        loss, d_o, d_r = ops.rpn_distill(as_nhwc(obj_s), as_nhwc(reg_s), as_nhwc(obj_t), as_nhwc(reg_t), thr, use_bbox, want_grad=want)
        ctx.save_for_backward(d_o, d_r)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d_o, d_r = ctx.saved_tensors
        g = g.contiguous()
        ops.scale_(d_o, 1.0, g)
        ops.scale_(d_r, 1.0, g)
        return None, None, from_nhwc(d_o), from_nhwc(d_r), None, None


def calculate_rpn_distillation_loss(rpn_output_source, rpn_output_target, cls_loss=None, bbox_loss=None, bbox_threshold=None):
    """distillation.py:18-84 (DIST.RPN): filtered-L2 on the objectness logits + L2 on the box deltas of the anchors whose source
    objectness exceeds the target's by more than `bbox_threshold`; both averaged over anchors, divided by the number of levels."""
    obj_s, reg_s = rpn_output_source
    obj_t, reg_t = rpn_output_target
    if len(obj_s) != len(obj_t):
        raise ValueError("Wrong rpn objectness output")
    if len(reg_s) != len(reg_t):
        raise ValueError("Wrong RPN bounding box regression output")
    if cls_loss != "filtered_l2":
        raise ValueError("Wrong loss function for rpn classification distillation")
    if bbox_loss not in ("l2", "None"):
        raise ValueError("Wrong loss function for rpn bounding box regression distillation")
    total = 0
    for os_, rs_, ot_, rt_ in zip(obj_s, reg_s, obj_t, reg_t):
        total = total + _RPNDistillFn.apply(os_.detach(), rs_.detach(), ot_, rt_, float(bbox_threshold), bbox_loss == "l2")
    return total / len(obj_s)
