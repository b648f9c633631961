"""PASCAL VOC detection metric (mAP) on the detections of the test-time path -- SURVEY.md §8f row F4.

Mirror of maskrcnn_benchmark/data/datasets/evaluation/voc/voc_eval.py:11-228 (same function names and return values).  This is
host-side numpy bookkeeping in the reference too (a few thousand boxes per class); the per-image greedy assignment is written
here without the python loop over detections.  Reference quirks that change the numbers are kept and marked (Q1..Q3)."""
import os

import numpy as np


def _iou_voc(det, gt):
    """IoU matrix with the reference's double +1: voc_eval.py:121-124 first moves x2,y2 by +1 ("integer typed boxes"), then
    boxlist_iou adds its own TO_REMOVE = 1 (structures/boxlist_ops.py:66-86).  fp32 like the torch original.  (Q1)"""
    d = det.astype(np.float32).copy()
    g = gt.astype(np.float32).copy()
    d[:, 2:] += 1
    g[:, 2:] += 1
    one = np.float32(1)
    area_d = (d[:, 2] - d[:, 0] + one) * (d[:, 3] - d[:, 1] + one)
    area_g = (g[:, 2] - g[:, 0] + one) * (g[:, 3] - g[:, 1] + one)
    lt = np.maximum(d[:, None, :2], g[None, :, :2])
    rb = np.minimum(d[:, None, 2:], g[None, :, 2:])
    wh = np.clip(rb - lt + one, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_d[:, None] + area_g[None, :] - inter)


def _match_image_class(det_boxes, gt_boxes, gt_difficult, iou_thresh):
    """Detections (already in descending score order) of one class in one image -> +1 true positive, 0 false positive,
    -1 ignored (hit a `difficult` box).  Each GT is credited to the FIRST detection whose best-IoU GT it is; a difficult GT
    ignores every detection assigned to it (voc_eval.py:126-146)."""
    if len(gt_boxes) == 0:
        return np.zeros(len(det_boxes), np.int8)
    iou = _iou_voc(det_boxes, gt_boxes)
    g = iou.argmax(axis=1)
    hit = iou.max(axis=1) >= iou_thresh
    first = np.zeros(len(det_boxes), bool)
    rows = np.nonzero(hit)[0]
    _, where = np.unique(g[rows], return_index=True)  # first detection per claimed GT
    first[rows[where]] = True
    out = np.where(first, 1, 0).astype(np.int8)
    out[hit & gt_difficult[g].astype(bool)] = -1
    out[~hit] = 0
    return out


def calc_detection_voc_prec_rec(gt_boxlists, pred_boxlists, iou_thresh=0.5):
    """-> (prec, rec): lists indexed by class id; entries None for ids never seen (rec also None when a class has no
    non-difficult GT).  voc_eval.py:79-168."""
    n_pos, scores, matches = {}, {}, {}
    for gt, pred in zip(gt_boxlists, pred_boxlists):
        pb = pred.bbox.cpu().numpy()
        pl = pred.get_field("labels").cpu().numpy()
        ps = pred.get_field("scores").cpu().numpy()
        gb = gt.bbox.cpu().numpy()
        gl = gt.get_field("labels").cpu().numpy()
        gd = gt.get_field("difficult").cpu().numpy()
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            sel = pl == l
            s = ps[sel]
            order = s.argsort()[::-1]  # the reference's (unstable) descending order, kept verbatim for equal scores
            boxes, s = pb[sel][order], s[order]
            gsel = gl == l
            n_pos[l] = n_pos.get(l, 0) + int(np.logical_not(gd[gsel]).sum())
            scores.setdefault(l, []).append(s)
            matches.setdefault(l, [])
            if len(boxes):
                matches[l].append(_match_image_class(boxes, gb[gsel], gd[gsel], iou_thresh))
    n_class = max(n_pos.keys()) + 1
    prec, rec = [None] * n_class, [None] * n_class
    for l in n_pos:
        s = np.concatenate(scores[l]) if scores[l] else np.zeros(0)
        m = np.concatenate(matches[l]) if matches[l] else np.zeros(0, np.int8)
        m = m[s.argsort()[::-1]]
        tp, fp = np.cumsum(m == 1), np.cumsum(m == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            prec[l] = tp / (fp + tp)  # nan while only ignored detections have been seen (Q2)
        if n_pos[l] > 0:
            rec[l] = tp / n_pos[l]
    return prec, rec


def calc_detection_voc_ap(prec, rec, use_07_metric=False):
    """Area under the precision envelope (VOC2010+) or the 11-point VOC2007 average; nan for classes without prec/rec
    (voc_eval.py:171-228)."""
    ap = np.full(len(prec), np.nan)
    for l, (p, r) in enumerate(zip(prec, rec)):
        if p is None or r is None:
            continue
        p = np.nan_to_num(p)
        if use_07_metric:
            ap[l] = sum((p[r >= t].max() if (r >= t).any() else 0.0) for t in np.arange(0.0, 1.1, 0.1)) / 11
        else:
            env = np.maximum.accumulate(np.concatenate(([0.0], p, [0.0]))[::-1])[::-1]
            rr = np.concatenate(([0.0], r, [1.0]))
            step = np.nonzero(rr[1:] != rr[:-1])[0]
            ap[l] = np.sum((rr[step + 1] - rr[step]) * env[step + 1])
    return ap


def eval_detection_voc(pred_boxlists, gt_boxlists, iou_thresh=0.5, use_07_metric=False):
    """-> {"ap": per-class array (index 0 = background, nan), "map": nanmean}  (voc_eval.py:57-76)"""
    assert len(gt_boxlists) == len(pred_boxlists), "Length of gt and pred lists need to be same."
    prec, rec = calc_detection_voc_prec_rec(gt_boxlists, pred_boxlists, iou_thresh)
    ap = calc_detection_voc_ap(prec, rec, use_07_metric)
    return {"ap": ap, "map": np.nanmean(ap)}


def do_voc_evaluation(dataset, predictions, output_folder, logger):
    """Predictions are resized back to the original image size, scored with the VOC2010 area metric at IoU 0.5, and the
    table is printed and written to <output_folder>/result.txt (voc_eval.py:11-54).  Class 0 (background) is skipped in the
    table but its nan stays in the comma-separated line, as in the reference (Q3)."""
    preds, gts = [], []
    for image_id, prediction in enumerate(predictions):
        info = dataset.get_img_info(image_id)
        preds.append(prediction.resize((info["width"], info["height"])))
        gts.append(dataset.get_groundtruth(image_id))
    result = eval_detection_voc(pred_boxlists=preds, gt_boxlists=gts, iou_thresh=0.5, use_07_metric=False)
    lines = ["mAP: {:.4f}".format(result["map"])]
    lines += ["{:<16}: {:.4f}".format(dataset.map_class_id_to_class_name(i), ap) for i, ap in enumerate(result["ap"]) if i > 0]
    text = "\n".join(lines) + "\n"
    csv = ",".join(str(x) for x in result["ap"])
    print(text)
    print(csv)
    if output_folder:
        with open(os.path.join(output_folder, "result.txt"), "w") as f:
            f.write(text)
            f.write(csv)
    return result
