"""boxlist_nms / remove_small_boxes / boxlist_iou / cat_boxlist
(mirror of maskrcnn_benchmark/structures/boxlist_ops.py:9-128), NMS and IoU+match on the HIP library."""
import torch

from ..layers import nms as _box_nms
from .bounding_box import BoxList


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep].convert(mode)


def remove_small_boxes(boxlist, min_size):
    wh = boxlist.convert("xywh").bbox
    keep = ((wh[:, 2] >= min_size) & (wh[:, 3] >= min_size)).nonzero().squeeze(1)
    return boxlist[keep]


def boxlist_iou(boxlist1, boxlist2):
    """[N,M] IoU with the +1 convention (:53-88).  Used by tests / tools; the training path fuses IoU + Matcher +
    labels + BoxCoder.encode in one kernel (abr_match_encode) and never materialises this matrix."""
    if boxlist1.size != boxlist2.size:
        raise RuntimeError("boxlists should have same image size, got {}, {}".format(boxlist1, boxlist2))
    a1, a2 = boxlist1.area(), boxlist2.area()
    b1, b2 = boxlist1.bbox, boxlist2.bbox
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def cat_boxlist(bboxes):
    assert isinstance(bboxes, (list, tuple)) and all(isinstance(b, BoxList) for b in bboxes)
    size, mode = bboxes[0].size, bboxes[0].mode
    assert all(b.size == size and b.mode == mode for b in bboxes)
    fields = set(bboxes[0].fields())
    assert all(set(b.fields()) == fields for b in bboxes)
    cat = (lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 0))
    out = BoxList(cat([b.bbox for b in bboxes]), size, mode)
    for f in fields:
        out.add_field(f, cat([b.get_field(f) for b in bboxes]))
    return out
