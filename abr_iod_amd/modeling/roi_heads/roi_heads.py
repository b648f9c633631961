"""CombinedROIHeads (mirror of maskrcnn_benchmark/modeling/roi_heads/roi_heads.py:9-77) with the box head only:
MASK_ON / KEYPOINT_ON are False in every configs/voc YAML (defaults.py:25,27)."""
import torch

from .box_head.box_head import build_roi_box_head


class CombinedROIHeads(torch.nn.ModuleDict):
    def __init__(self, cfg, heads):
        super().__init__(heads)
        self.cfg = cfg.clone()

    def forward(self, features, proposals, targets=None):
        """training -> (x, detections, soften_results, losses, roi_align_features); eval -> (x, detections, results_background, [])
        (roi_heads.py:23-63)"""
        losses = {}
        if not self.training:
            x, detections, results_background = self.box(features, proposals, targets)
            return x, detections, results_background, []
        x, detections, soft_res, loss_box, roi_align_features = self.box(features, proposals, targets)
        losses.update(loss_box)
        return x, detections, soft_res, losses, roi_align_features

    def forward_joint(self, features, proposals, targets, soften_proposals):
        """training forward + calculate_soften_label(features, soften_proposals) sharing one head pass"""
        (x, detections, soft_res, loss_box, raf), (s_score, s_bbox, s_raf) = self.box.forward_joint(features, proposals, targets, soften_proposals)
        return (x, detections, soft_res, dict(loss_box), raf), (s_score, s_bbox, None, s_raf)

    def calculate_soften_label(self, features, proposals, targets=None):
        """-> (soften_score, soften_bbox, mask_logits=None, roi_align_features)  (roi_heads.py:65-72)"""
        soften_score, soften_bbox, _, roi_align_features = self.box.calculate_soften_label(features, proposals, targets)
        return soften_score, soften_bbox, None, roi_align_features


def build_roi_heads(cfg, in_channels):
    if cfg.MODEL.RETINANET_ON or cfg.MODEL.MASK_ON or cfg.MODEL.KEYPOINT_ON:
        raise NotImplementedError("only the box head is on the hot path (SURVEY.md §2 rows 6b/7b)")
    if cfg.MODEL.RPN_ONLY:
        return []
    return CombinedROIHeads(cfg, [("box", build_roi_box_head(cfg, in_channels))])
