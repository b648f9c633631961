"""TEST INFRASTRUCTURE ONLY -- the reference's CPU path for one batch: Faster R-CNN forward + the 4 detector losses
(BASELINE.json configs[0]: "CPU-only forward+loss"; the reference cannot run backward on CPU, csrc/ROIAlign.h:44).
torch-CPU convs (the reference's own substrate) + oracle.c for ROIAlign / NMS / IoU / matching.  Used by bench.py's
`cpu_baseline` leg (kind "port") and by tests; never by the product path."""
import time

import numpy as np
import torch

from . import ops as cops
from . import torch_ref as R
from .model_ref import RefModel


def cpu_forward_loss(sd_target, images, gt_boxes, gt_labels, n_old, dist_type="id", pre_nms=12000, post_nms=2000,
                     rpn_batch=256, roi_batch=512, seed=0, timings=None):
    """images [B,3,H,W] torch CPU; gt_boxes/gt_labels lists of numpy arrays.  Returns the loss dict (python floats)."""
    g = torch.Generator().manual_seed(seed)
    t0 = time.time()
    m = RefModel(sd_target, trainable_prefixes=())
    B, _, H, W = images.shape
    with torch.no_grad():
        feat = m.backbone(images)
        t1 = time.time()
        obj, reg = m.rpn_head(feat)
        t2 = time.time()
        fh, fw = feat.shape[-2:]
        cell = cops.cell_anchors()
        anchors, vis = cops.grid_anchors(cell, fh, fw, 16, (H, W))
        props = R.rpn_post_process(obj, reg, [anchors] * B, [(H, W)] * B, pre_nms, post_nms, gt_boxes=gt_boxes)
        t3 = time.time()
        # RPN loss
        labs, tgts, posm, negm = [], [], [], []
        for i in range(B):
            lab, tgt, _ = R.rpn_prepare_targets(anchors, vis, gt_boxes[i])
            lab_t = torch.from_numpy(lab)
            pos = torch.nonzero(lab_t >= 1).squeeze(1); neg = torch.nonzero(lab_t == 0).squeeze(1)
            npos = min(pos.numel(), rpn_batch // 2); nneg = min(neg.numel(), rpn_batch - npos)
            pm = torch.zeros_like(lab_t, dtype=torch.bool); nm = torch.zeros_like(lab_t, dtype=torch.bool)
            pm[pos[torch.randperm(pos.numel(), generator=g)[:npos]]] = True
            nm[neg[torch.randperm(neg.numel(), generator=g)[:nneg]]] = True
            labs.append(lab_t); tgts.append(torch.from_numpy(tgt)); posm.append(pm); negm.append(nm)
        lo, lb = R.rpn_loss(obj, reg, torch.stack(labs), torch.stack(tgts), torch.stack(posm), torch.stack(negm))
        t4 = time.time()
        # RoI subsample
        rois, labels_h, rt_h = [], [], []
        for i in range(B):
            boxes = props[i][0]
            iou = cops.box_iou(gt_boxes[i], boxes)
            mt = cops.matcher(iou, 0.5, 0.5, False)
            lab = gt_labels[i][np.clip(mt, 0, None)].astype(np.int64)
            lab[mt == -1] = 0; lab[mt == -2] = -1
            tgt = cops.box_encode(gt_boxes[i][np.clip(mt, 0, None)], boxes, (10.0, 10.0, 5.0, 5.0))
            lab_t = torch.from_numpy(lab)
            pos = torch.nonzero(lab_t >= 1).squeeze(1); neg = torch.nonzero(lab_t == 0).squeeze(1)
            npos = min(pos.numel(), roi_batch // 4); nneg = min(neg.numel(), roi_batch - npos)
            sel = torch.cat([pos[torch.randperm(pos.numel(), generator=g)[:npos]], neg[torch.randperm(neg.numel(), generator=g)[:nneg]]]).sort()[0].numpy()
            rois.append(np.concatenate([np.full((len(sel), 1), i, np.float32), boxes[sel]], 1))
            labels_h.append(lab[sel]); rt_h.append(tgt[sel])
        rois = torch.from_numpy(np.concatenate(rois, 0))
        t5 = time.time()
        pooled = torch.from_numpy(cops.roi_align_forward(feat.numpy(), rois.numpy(), 0.0625, 7, 7, 0))
        t6 = time.time()
        x = pooled
        for i in range(3):
            x = m._block(x, f"roi_heads.box.feature_extractor.head.layer4.{i}", 2 if i == 0 else 1)
        v = torch.nn.functional.adaptive_avg_pool2d(x, 1).flatten(1)
        pr = "roi_heads.box.predictor"
        logits = torch.nn.functional.linear(v, m.p[f"{pr}.cls_score.weight"], m.p[f"{pr}.cls_score.bias"])
        boxreg = torch.nn.functional.linear(v, m.p[f"{pr}.bbox_pred.weight"], m.p[f"{pr}.bbox_pred.bias"])
        lc, lbox = R.box_head_loss(logits, boxreg, torch.from_numpy(np.concatenate(labels_h)), torch.from_numpy(np.concatenate(rt_h)), dist_type, n_old)
        t7 = time.time()
    if timings is not None:
        timings.update(backbone=t1 - t0, rpn_head=t2 - t1, rpn_postproc=t3 - t2, rpn_loss=t4 - t3, roi_subsample=t5 - t4,
                       roi_align=t6 - t5, layer4_predictor_loss=t7 - t6, total=t7 - t0)
    return dict(loss_classifier=float(lc), loss_box_reg=float(lbox), loss_objectness=float(lo), loss_rpn_box_reg=float(lb))
