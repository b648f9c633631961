"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatements of the reference's floating-point ops.

These re-derive each formula from the reference (file:line cited per function, paths under
/root/reference/maskrcnn_benchmark/) in plain torch so that autograd supplies reference gradients.
Pinned against tests/golden/*.npz (generated from the reference itself) in tests/test_oracle_golden.py.
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as cops


# ------------------------------------------------------------------ A16  ARD
def attention_map(f):
    """distillation/distillation.py:121-130  A(F) = HW * softmax_hw(mean_c |F|^2); `temp` unused in the softmax."""
    n, c, h, w = f.shape
    m = f.abs().pow(2).mean(dim=1)
    return (h * w * F.softmax(m.view(n, -1), dim=1)).view(n, h, w)


def ard_loss(f_src, f_tgt, gamma=1.0):
    """distillation.py:86-118, called as (source, target) at tools/train_incremental.py:115.
    mask = attention of the SOURCE; afd = mean((F_src*sqrt(A_src) - F_tgt*sqrt(A_src))^2); pad = mean|A_tgt - A_src|."""
    a_src = attention_map(f_src)
    a_tgt = attention_map(f_tgt)
    pad = (a_tgt - a_src).abs().mean()
    s = torch.sqrt(a_src).unsqueeze(1)
    afd = ((f_src * s - f_tgt * s) ** 2).mean()
    return afd + gamma * pad


# ------------------------------------------------------------------ A17  RoI distillation
def roi_distillation_loss(z_s, b_s, z_t, b_t, dist="id"):
    """distillation.py:164-240.  z_s [n,K_old], b_s [n,K_old,4], z_t [n,K_all], b_t [n,K_all,4]."""
    k_old, k_all = z_s.shape[1], z_t.shape[1]
    if dist == "id":
        den = torch.logsumexp(z_t, dim=1)
        out_no_bg = z_t[:, 1:k_old] - den[:, None]                              # :195 (slice 1:-(K_all-K_old))
        bg_idx = torch.tensor([0] + list(range(k_old, k_all)))
        out_bg = torch.logsumexp(z_t[:, bg_idx], dim=1) - den                   # :196
        lab = torch.softmax(z_s, dim=1)
        loss = (lab[:, 0] * out_bg + (lab[:, 1:] * out_no_bg).sum(dim=1)) / k_old   # :198 divides by K_old
        cls = -loss.mean()
    else:  # 'l2' : mean-centred logits (:171-177), MSE averaged over classes then proposals (:185-188)
        zs_n = z_s - z_s.mean(dim=1, keepdim=True)
        zt_n = z_t - z_t.mean(dim=1, keepdim=True)
        cls = ((zt_n[:, :k_old] - zs_n[:, :k_old]) ** 2).mean(dim=1).mean(dim=0)
    bbox = ((b_t[:, 1:k_old, :] - b_s[:, 1:, :]) ** 2).sum(dim=2).mean(dim=1).mean(dim=0)   # :204-209
    return cls + bbox


# ------------------------------------------------------------------ A10  box-head loss
def box_head_loss(logits, reg, labels, reg_targets, dist_type="l2", n_old=0):
    """modeling/roi_heads/box_head/loss.py:122-181."""
    if dist_type == "id":
        out = torch.zeros_like(logits)
        den = torch.logsumexp(logits, dim=1)
        out[:, 0] = torch.logsumexp(logits[:, 0:n_old + 1], dim=1) - den        # :155
        out[:, n_old + 1:] = logits[:, n_old + 1:] - den[:, None]               # :156 ; cols 1..n_old stay 0
        cls = F.nll_loss(out, labels)
    else:
        cls = F.cross_entropy(logits, labels)
    pos = torch.nonzero(labels > 0).squeeze(1)
    cols = 4 * labels[pos][:, None] + torch.arange(4)[None]
    d = reg[pos[:, None], cols] - reg_targets[pos]
    a = d.abs()
    box = torch.where(a < 1.0, 0.5 * a * a, a - 0.5).sum() / labels.numel()     # beta=1, :173-179
    return cls, box


# ------------------------------------------------------------------ A9  RPN loss
def permute_and_flatten(t, N, A, Cc, H, W):
    """modeling/rpn/utils.py:10-14 : [N, A*C, H, W] -> [N, H*W*A, C]."""
    return t.view(N, A, Cc, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, Cc)


def rpn_loss(objectness, box_regression, labels, reg_targets, sampled_pos_mask, sampled_neg_mask):
    """modeling/rpn/loss.py:104-148.  objectness [N,A,H,W], box_regression [N,4A,H,W];
    labels/reg_targets/masks are per-image stacked [N, HWA(,4)] (sampling is injected, not drawn)."""
    N, A, H, W = objectness.shape
    obj = permute_and_flatten(objectness, N, A, 1, H, W).reshape(-1)
    reg = permute_and_flatten(box_regression, N, A, 4, H, W).reshape(-1, 4)
    lab = labels.reshape(-1)
    rt = reg_targets.reshape(-1, 4)
    pos = torch.nonzero(sampled_pos_mask.reshape(-1)).squeeze(1)
    neg = torch.nonzero(sampled_neg_mask.reshape(-1)).squeeze(1)
    samp = torch.cat([pos, neg])
    d = (reg[pos] - rt[pos]).abs()
    beta = 1.0 / 9
    box = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).sum() / samp.numel()   # :136
    objl = F.binary_cross_entropy_with_logits(obj[samp], lab[samp])                          # :145-146
    return objl, box


def rpn_prepare_targets(anchors, vis, gt, hi=0.7, lo=0.3):
    """modeling/rpn/loss.py:66-102 for one image (numpy in, numpy out): labels fp32 {1,0,-1}, encoded targets, matches."""
    iou = cops.box_iou(gt, anchors)
    m = cops.matcher(iou, hi, lo, True)
    labels = (m >= 0).astype(np.float32)
    labels[m == -1] = 0
    labels[~vis] = -1
    labels[m == -2] = -1
    tgt = cops.box_encode(gt[np.clip(m, 0, None)], anchors, (1.0, 1.0, 1.0, 1.0))
    return labels, tgt, m


# ------------------------------------------------------------------ A5  RPN post-processing
def rpn_post_process(objectness, box_regression, anchors, image_sizes, pre_nms_top_n, post_nms_top_n,
                     nms_thresh=0.7, min_size=0, gt_boxes=None, strict_gt=False):
    """modeling/rpn/inference.py:76-147 + structures/boxlist_ops.py:9-48 + bounding_box.py:214-225.
    objectness [N,A,H,W] / box_regression [N,4A,H,W] torch; anchors list of [HWA,4] numpy; image_sizes (h,w).
    Returns per-image (boxes, scores) numpy; training callers pass gt_boxes to append (inference.py:53-74)."""
    N, A, H, W = objectness.shape
    obj = permute_and_flatten(objectness, N, A, 1, H, W).view(N, -1).sigmoid()
    reg = permute_and_flatten(box_regression, N, A, 4, H, W)
    k = min(pre_nms_top_n, A * H * W)
    top, idx = obj.topk(k, dim=1, sorted=True)
    out = []
    for i in range(N):
        ii = idx[i].numpy()
        props = cops.box_decode(reg[i].numpy()[ii], anchors[i][ii], (1.0, 1.0, 1.0, 1.0))
        h, w = image_sizes[i]
        props[:, 0] = np.clip(props[:, 0], 0, w - 1); props[:, 1] = np.clip(props[:, 1], 0, h - 1)
        props[:, 2] = np.clip(props[:, 2], 0, w - 1); props[:, 3] = np.clip(props[:, 3], 0, h - 1)
        sc = top[i].numpy()
        ws = props[:, 2] - props[:, 0] + 1; hs = props[:, 3] - props[:, 1] + 1
        keep = np.nonzero((ws >= min_size) & (hs >= min_size))[0]
        props, sc = props[keep], sc[keep]
        kept = cops.nms(props, sc, nms_thresh, strict_gt=strict_gt)[:post_nms_top_n]
        props, sc = props[kept], sc[kept]
        if gt_boxes is not None:
            props = np.concatenate([props, gt_boxes[i].astype(np.float32)], 0)
            sc = np.concatenate([sc, np.ones(len(gt_boxes[i]), np.float32)], 0)
        out.append((props, sc))
    return out


# ------------------------------------------------------------------ ablation distillation losses
def feature_distillation_loss(source_features, target_features):
    """distillation/distillation.py:133-161, loss='normalized_filtered_l1': sum over levels of mean(max((s - mean s) - (t - mean t), 0))"""
    total = 0
    for s, t in zip(source_features, target_features):
        diff = (s - s.mean()) - (t - t.mean())
        total = total + torch.max(diff, torch.zeros_like(diff)).mean()
    return total


def rpn_distillation_loss(rpn_output_source, rpn_output_target, bbox_threshold=0.1, bbox_loss="l2"):
    """distillation/distillation.py:18-84 with cls_loss='filtered_l2': objectness lists of [N,A,H,W], delta lists of [N,4A,H,W]."""
    (obj_s, reg_s), (obj_t, reg_t) = rpn_output_source, rpn_output_target
    cls, box = [], []
    for os_, rs_, ot_, rt_ in zip(obj_s, reg_s, obj_t, reg_t):
        diff = os_ - ot_
        cls.append((torch.max(diff, torch.zeros_like(diff)) ** 2).mean())
        N, A, H, W = diff.shape
        mask = (permute_and_flatten(diff, N, A, 1, H, W) > bbox_threshold).to(diff.dtype).detach()
        ms = permute_and_flatten(rs_, N, A, 4, H, W) * mask
        mt = permute_and_flatten(rt_, N, A, 4, H, W) * mask
        box.append(((ms - mt) ** 2).sum(dim=2).mean(dim=1).mean(dim=0) if bbox_loss == "l2" else 0)
    return sum(cls) / len(obj_s) + sum(box) / len(reg_s)


# ------------------------------------------------------------------ F4  test-time PostProcessor
def det_softmax_decode(class_logits, box_regression, proposals, image_sizes_wh, weights=(10.0, 10.0, 5.0, 5.0),
                       cls_agnostic=False):
    """roi_heads/box_head/inference.py:55-70,84-105: softmax, per-class BoxCoder.decode, clip_to_image.
    class_logits [K,C], box_regression [K,4C] torch; proposals list of [n_i,4] numpy; image_sizes_wh (w,h) per image.
    -> prob [K,C], boxes [K,C,4] numpy."""
    prob = F.softmax(class_logits, -1).numpy()
    K, C = prob.shape
    reg = box_regression.numpy()
    if cls_agnostic:
        reg = np.tile(reg[:, -4:], (1, C))
    cat = np.concatenate(proposals, 0).astype(np.float32)
    boxes = np.empty((K, C, 4), np.float32)
    for j in range(C):
        boxes[:, j] = cops.box_decode(reg[:, 4 * j:4 * j + 4], cat, weights)
    r0 = 0
    for p, (w, h) in zip(proposals, image_sizes_wh):
        b = boxes[r0:r0 + len(p)]
        b[..., 0] = np.clip(b[..., 0], 0, w - 1); b[..., 1] = np.clip(b[..., 1], 0, h - 1)
        b[..., 2] = np.clip(b[..., 2], 0, w - 1); b[..., 3] = np.clip(b[..., 3], 0, h - 1)
        r0 += len(p)
    return prob, boxes


def det_filter_results(prob, boxes, score_thresh=0.05, nms_thresh=0.5, detections_per_img=100):
    """filter_results (inference.py:106-151) for ONE image: prob [n,C], boxes [n,C,4] numpy ->
    (boxes, scores, labels) of classes 1..C-1 and (boxes, scores) of the background class 0.
    Equal scores are ordered by ascending proposal index (the sort the reference leaves unspecified)."""
    n, C = prob.shape
    rb, rs, rl = [], [], []
    bg = None
    for j in range(C):
        inds = np.nonzero(prob[:, j] > score_thresh)[0]
        sj, bj = prob[inds, j], boxes[inds, j]
        keep = cops.nms(bj, sj, nms_thresh)
        if j > 0:
            rb.append(bj[keep]); rs.append(sj[keep]); rl.append(np.full(len(keep), j, np.int64))
        else:
            bg = (bj[keep], sj[keep])
    rb, rs, rl = np.concatenate(rb, 0), np.concatenate(rs, 0), np.concatenate(rl, 0)
    if len(rs) > detections_per_img > 0:
        thresh = np.sort(rs)[len(rs) - detections_per_img]      # kthvalue(n - D + 1)
        keep = np.nonzero(rs >= thresh)[0]
        rb, rs, rl = rb[keep], rs[keep], rl[keep]
    return (rb, rs, rl), bg


def post_process(class_logits, box_regression, proposals, image_sizes_wh, score_thresh=0.05, nms_thresh=0.5,
                 detections_per_img=100, weights=(10.0, 10.0, 5.0, 5.0)):
    """PostProcessor.forward (inference.py:43-82): -> list of (boxes, scores, labels), background of the LAST image."""
    prob, boxes = det_softmax_decode(class_logits, box_regression, proposals, image_sizes_wh, weights)
    out, bg, r0 = [], None, 0
    for p in proposals:
        res, bg = det_filter_results(prob[r0:r0 + len(p)], boxes[r0:r0 + len(p)], score_thresh, nms_thresh, detections_per_img)
        out.append(res)
        r0 += len(p)
    return out, bg


# ------------------------------------------------------------------ A2/A3/A12/A13  conv blocks
def frozen_bn(x, w, b, rm, rv):
    """layers/batch_norm.py:19-31 : scale = w * rsqrt(var) (NO eps), bias = b - mean*scale."""
    scale = w * rv.rsqrt()
    bias = b - rm * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def bf16_round(t):
    """round-to-nearest-even to bfloat16 and back: what the bf16 math mode does to both conv operands (BASELINE.json configs[4])"""
    return t.bfloat16().float()


def bottleneck(x, p, stride, has_ds, bf16=False):
    """modeling/backbone/resnet.py:327-346 with STRIDE_IN_1X1=True (:278): stride sits in conv1 and the downsample.
    bf16=True restates the "bf16 MFMA backbone" configuration: every conv sees bf16-rounded activations and weights, products and
    sums in fp32 (a bf16 x bf16 product is exact in fp32), everything else unchanged."""
    r = bf16_round if bf16 else (lambda t: t)
    idt = x
    o = F.relu(frozen_bn(F.conv2d(r(x), r(p["conv1.weight"]), stride=stride), *p["bn1"]))
    o = F.relu(frozen_bn(F.conv2d(r(o), r(p["conv2.weight"]), padding=1), *p["bn2"]))
    o = frozen_bn(F.conv2d(r(o), r(p["conv3.weight"])), *p["bn3"])
    if has_ds:
        idt = frozen_bn(F.conv2d(r(x), r(p["downsample.0.weight"]), stride=stride), *p["ds_bn"])
    return F.relu(o + idt)


LOG_CLIP = math.log(1000.0 / 16)
