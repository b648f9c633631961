"""Test-time PostProcessor (mirror of maskrcnn_benchmark/modeling/roi_heads/box_head/inference.py:12-175).

Same constructor, same `forward(x, boxes) -> (results, results_background)`.  The reference loops over images and classes on the
host (nonzero / gather / _C.nms / kthvalue(.cpu()) per class); here the whole batch is four launches (csrc/detect.hip) and a
single read-back of the per-image detection counts to size the returned BoxLists."""
import torch
from torch import nn

from .... import _lib as L
from .... import ops
from ....structures.bounding_box import BoxList
from ...box_coder import BoxCoder


class PostProcessor(nn.Module):
    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None, cls_agnostic_bbox_reg=False):
        super().__init__()
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg

    def forward(self, x, boxes):
        """x = (class_logits [K,C], box_regression [K,4C] or [K,C,4]); boxes: list[BoxList] proposals per image.
        -> (list[BoxList] with fields scores / labels, BoxList of the background class of the LAST image -- the reference's
        loop variable, inference.py:75-82)."""
        class_logits, box_regression = x
        K, C = class_logits.shape
        box_regression = box_regression.reshape(K, -1)
        dev = class_logits.device
        counts = [len(b) for b in boxes]
        if sum(counts) != K:
            raise ValueError("PostProcessor: {} logits rows for {} proposals".format(K, sum(counts)))
        N = len(boxes)
        rois = torch.cat([torch.cat([torch.full((len(b), 1), i, dtype=torch.float32, device=dev), b.bbox], 1)
                          for i, b in enumerate(boxes)], 0) if K else torch.zeros((0, 5), device=dev)
        img_hw = torch.tensor([[b.size[1], b.size[0]] for b in boxes], dtype=torch.int32).to(dev, non_blocking=True)
        off = [0]
        for c in counts:
            off.append(off[-1] + c)
        row_off = torch.tensor(off, dtype=torch.int32).to(dev, non_blocking=True)
        prob, dec = ops.det_softmax_decode(class_logits, box_regression, rois, C, img_hw, self.box_coder.weights,
                                           self.cls_agnostic_bbox_reg)
        ob, os_, ol, oc, bb, bs, bc = ops.det_select(prob, dec, row_off, N, C, max(counts) if counts else 0, self.score_thresh,
                                                     self.nms, self.detections_per_img)
        n_out = torch.cat([oc, bc]).tolist()  # the one device->host read of the eval step
        results = []
        for i, b in enumerate(boxes):
            r = BoxList(ob[i, : n_out[i]], b.size, mode="xyxy")
            r.add_field("scores", os_[i, : n_out[i]])
            r.add_field("labels", ol[i, : n_out[i]])
            results.append(r)
        results_background = None
        if N:
            nb = n_out[N + N - 1]
            results_background = BoxList(bb[N - 1, :nb], boxes[-1].size, mode="xyxy")
            results_background.add_field("scores", bs[N - 1, :nb])
            results_background.add_field("labels", torch.zeros((nb,), dtype=torch.int64, device=dev))
        return results, results_background


def make_roi_box_post_processor(cfg):
    """inference.py:154-175"""
    return PostProcessor(cfg.MODEL.ROI_HEADS.SCORE_THRESH, cfg.MODEL.ROI_HEADS.NMS, cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG,
                         BoxCoder(weights=cfg.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS), cfg.MODEL.CLS_AGNOSTIC_BBOX_REG)
