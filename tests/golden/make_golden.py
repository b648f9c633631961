#!/usr/bin/env python3
"""Generate the op-level golden vectors under tests/golden/*.npz FROM THE REFERENCE ITSELF.

Run by hand, in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py
Every fixture holds inputs and the reference's outputs for one SURVEY.md §8(a) row; the committed
.npz files are data only.  Gradients that the reference cannot produce on CPU (ROIAlign backward:
csrc/ROIAlign.h:44 "Not implemented on the CPU") are NOT produced here; see make_golden_grads.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.setup()

from maskrcnn_benchmark import _C  # noqa: E402
from maskrcnn_benchmark.distillation.distillation import (  # noqa: E402
    activation_at,
    calculate_attentive_roi_feature_distillation,
    calculate_roi_distillation_losses,
)
from maskrcnn_benchmark.layers import smooth_l1_loss  # noqa: E402
from maskrcnn_benchmark.layers.sigmoid_focal_loss import sigmoid_focal_loss_cpu  # noqa: E402
from maskrcnn_benchmark.modeling.box_coder import BoxCoder  # noqa: E402
from maskrcnn_benchmark.modeling.matcher import Matcher  # noqa: E402
from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import FastRCNNLossComputation  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator, generate_anchors  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.loss import RPNLossComputation, generate_rpn_labels  # noqa: E402
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler  # noqa: E402
from maskrcnn_benchmark.structures.bounding_box import BoxList  # noqa: E402
from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou  # noqa: E402
from maskrcnn_benchmark.structures.image_list import ImageList  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"wrote {name}.npz: " + ", ".join(f"{k}{list(v.shape)}" for k, v in out.items()))


def rand_boxes(g, n, W, H, min_wh=1.0, max_wh=None):
    max_wh = max_wh or min(W, H)
    x1 = torch.rand(n, generator=g) * (W - 2)
    y1 = torch.rand(n, generator=g) * (H - 2)
    w = min_wh + torch.rand(n, generator=g) * (max_wh - min_wh)
    h = min_wh + torch.rand(n, generator=g) * (max_wh - min_wh)
    return torch.stack([x1, y1, (x1 + w).clamp(max=W - 1), (y1 + h).clamp(max=H - 1)], 1)


# ---------------------------------------------------------------- A4 anchors
def gold_anchors():
    cell = generate_anchors(16, (32, 64, 128, 256, 512), (0.5, 1.0, 2.0))  # anchor_generator.py:215
    ag = AnchorGenerator(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0),
                         anchor_strides=(16,), straddle_thresh=0)
    H, W = 38, 63
    img = ImageList(torch.zeros(2, 3, 600, 1000), [(600, 1000), (560, 900)])
    anchors = ag(img, [torch.zeros(2, 1024, H, W)])
    a0, a1 = anchors[0][0], anchors[1][0]
    # full arrays are integer-valued fp32: store all of them, they compress well
    save("anchors", cell=cell, bbox0=a0.bbox, vis0=a0.get_field("visibility"),
         bbox1=a1.bbox, vis1=a1.get_field("visibility"), hw=np.array([H, W]),
         image_sizes=np.array([[600, 1000], [560, 900]]))


# ---------------------------------------------------------------- A7 box coder
def gold_box_coder():
    g = torch.Generator().manual_seed(1)
    ex = rand_boxes(g, 64, 1000, 600, 4, 400)
    gt = rand_boxes(g, 64, 1000, 600, 4, 400)
    out = {}
    for name, w in (("rpn", (1.0, 1.0, 1.0, 1.0)), ("head", (10.0, 10.0, 5.0, 5.0))):
        bc = BoxCoder(weights=w)
        enc = bc.encode(gt, ex)
        deltas = torch.randn(64, 4 * 3, generator=g) * 2.0
        deltas[0, 2] = 50.0  # forces the log(1000/16) clamp (box_coder.py:71-72)
        deltas[1, 7] = 9.0
        dec = bc.decode(deltas, ex)
        out.update({f"{name}_enc": enc, f"{name}_deltas": deltas, f"{name}_dec": dec})
    save("box_coder", ex=ex, gt=gt, **out)


# ---------------------------------------------------------------- A8 IoU + matcher
def gold_matcher():
    g = torch.Generator().manual_seed(2)
    n = 500
    gt = torch.tensor([[100., 100., 400., 300.], [500., 200., 900., 550.], [120., 90., 380., 310.]])
    prop = rand_boxes(g, n, 1000, 600, 8, 500)
    prop[0] = gt[0]                       # exact match
    prop[1] = torch.tensor([100., 100., 400., 250.])  # high overlap with 0 and 2
    prop[2] = prop[3] = torch.tensor([480., 180., 880., 560.])  # tie for gt1's best
    t = BoxList(gt, (1000, 600)); p = BoxList(prop, (1000, 600))
    iou = boxlist_iou(t, p)
    out = {}
    for tag, (hi, lo, lq) in {"rpn": (0.7, 0.3, True), "head": (0.5, 0.5, False)}.items():
        out[f"{tag}_matched"] = Matcher(hi, lo, allow_low_quality_matches=lq)(iou)
    save("matcher", gt=gt, prop=prop, iou=iou, **out)


# ---------------------------------------------------------------- A6 NMS
def gold_nms():
    g = torch.Generator().manual_seed(3)
    n = 300
    centers = rand_boxes(g, 30, 900, 500, 40, 200)
    boxes = centers.repeat(10, 1) + torch.randn(n, 4, generator=g) * 6.0
    boxes[:, 2:] = torch.max(boxes[:, 2:], boxes[:, :2] + 1)
    # exact-threshold pair: IoU == 0.5 exactly (100/200): exposes CPU '>=' vs CUDA '>' (nms_cpu.cpp:60)
    boxes[0] = torch.tensor([0., 0., 9., 9.]); boxes[1] = torch.tensor([0., 0., 9., 19.])
    scores = torch.rand(n, generator=g)
    scores[0], scores[1] = 0.999, 0.998
    out = {}
    for thr in (0.5, 0.7):
        out[f"keep_{int(thr * 10)}"] = _C.nms(boxes, scores, thr)
    save("nms", boxes=boxes, scores=scores, **out)


# ---------------------------------------------------------------- A11 ROIAlign forward
def gold_roi_align():
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(2, 8, 10, 14, generator=g)
    rois = torch.tensor([
        [0, 16., 16., 120., 100.], [1, 0., 0., 223., 159.], [0, -40., -30., 20., 30.],
        [1, 200., 140., 260., 200.], [0, 50.2, 60.7, 50.9, 61.1], [1, 7., 5., 7.4, 5.2],
        [0, 100.5, 20.25, 180.75, 140.5], [1, 3.3, 4.4, 215.5, 150.1], [0, 215., 150., 300., 260.],
        [1, 64., 64., 128., 128.], [0, 0., 0., 15.99, 15.99], [1, 111.1, 22.2, 133.3, 155.5],
    ])
    out = {}
    for sr in (0, 2):
        out[f"out_sr{sr}"] = _C.roi_align_forward(feat, rois, 1.0 / 16, 7, 7, sr)
    out["out_sr0_3x5"] = _C.roi_align_forward(feat, rois, 0.0625, 3, 5, 0)
    save("roi_align", feat=feat, rois=rois, **out)
    # a second, C4-shaped case: many channels, realistic proposals, sr=0 (target YAMLs) — small K
    feat2 = torch.randn(2, 24, 38, 63, generator=g)
    r = rand_boxes(g, 40, 1000, 600, 12, 700)
    rois2 = torch.cat([torch.randint(0, 2, (40, 1), generator=g).float(), r], 1)
    save("roi_align_c4", feat=feat2, rois=rois2,
         out_sr0=_C.roi_align_forward(feat2, rois2, 0.0625, 7, 7, 0),
         out_sr2=_C.roi_align_forward(feat2, rois2, 0.0625, 7, 7, 2))


# ---------------------------------------------------------------- A14 / A15 elementwise losses
def gold_elementwise():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 4, generator=g); t = torch.randn(37, 4, generator=g) * 0.5
    x.requires_grad_(True)
    out = {"x": x.detach(), "t": t}
    for tag, beta, avg in (("b19_sum", 1.0 / 9, False), ("b1_sum", 1.0, False), ("b19_mean", 1.0 / 9, True)):
        x.grad = None
        l = smooth_l1_loss(x, t, beta=beta, size_average=avg)
        l.backward()
        out[f"{tag}_loss"] = l.detach(); out[f"{tag}_grad"] = x.grad.clone()
    save("smooth_l1", **out)

    logits = (torch.randn(40, 6, generator=g) * 3).requires_grad_(True)
    targets = torch.randint(-1, 7, (40,), generator=g).int()  # -1 ignore, 0 bg, 1..6 classes
    loss = sigmoid_focal_loss_cpu(logits, targets, (2.0,), (0.25,))  # layers/sigmoid_focal_loss.py:40-52
    d_loss = torch.randn(40, 6, generator=g)
    (loss * d_loss).sum().backward()
    save("sigmoid_focal", logits=logits.detach(), targets=targets, gamma=2.0, alpha=0.25,
         loss=loss.detach(), d_loss=d_loss, d_logits=logits.grad)


# ---------------------------------------------------------------- A10 box-head losses
def gold_box_head_loss():
    g = torch.Generator().manual_seed(6)
    out = {}
    for (k_old, k_all) in ((16, 21), (11, 21), (11, 16)):
        n = 96
        n_old = k_old - 1
        logits = (torch.randn(n, k_all, generator=g) * 2).requires_grad_(True)
        reg = torch.randn(n, 4 * k_all, generator=g).requires_grad_(True)
        labels = torch.zeros(n, dtype=torch.int64)
        labels[:20] = torch.randint(n_old + 1, k_all, (20,), generator=g)  # new-class fg
        labels[20:24] = torch.randint(1, n_old + 1, (4,), generator=g)      # replayed old-class boxes (quirk 4)
        rt = torch.randn(n, 4, generator=g) * 0.3
        for dist in ("id", "l2"):
            ev = FastRCNNLossComputation(None, None, None, False, dist, ["c"] * n_old)
            p = BoxList(torch.zeros(n, 4), (10, 10)); p.add_field("labels", labels); p.add_field("regression_targets", rt)
            ev._proposals = [p]
            logits.grad = None; reg.grad = None
            lc, lb = ev([logits], [reg])
            (lc + lb).backward()
            tag = f"k{k_old}_{k_all}_{dist}"
            out.update({f"{tag}_cls": lc.detach(), f"{tag}_box": lb.detach(),
                        f"{tag}_dlogits": logits.grad.clone(), f"{tag}_dreg": reg.grad.clone()})
        out.update({f"k{k_old}_{k_all}_logits": logits.detach(), f"k{k_old}_{k_all}_reg": reg.detach(),
                    f"k{k_old}_{k_all}_labels": labels, f"k{k_old}_{k_all}_rt": rt})
    save("box_head_loss", **out)


# ---------------------------------------------------------------- A17 RoI distillation (ID / L2)
def gold_roi_distill():
    g = torch.Generator().manual_seed(7)
    out = {}
    for (k_old, k_all) in ((16, 21), (11, 21), (11, 16), (21, 21)):
        n = 128
        zs = torch.randn(n, k_old, generator=g) * 2; bs = torch.randn(n, k_old, 4, generator=g)
        zt = (torch.randn(n, k_all, generator=g) * 2).requires_grad_(True)
        bt = torch.randn(n, k_all, 4, generator=g).requires_grad_(True)
        tag = f"k{k_old}_{k_all}"
        out.update({f"{tag}_zs": zs, f"{tag}_bs": bs, f"{tag}_zt": zt.detach(), f"{tag}_bt": bt.detach()})
        for dist in ("id", "l2"):
            if dist == "id" and k_old == k_all:
                continue  # quirk 3: empty slice -> shape error in the reference
            zt.grad = None; bt.grad = None
            l = calculate_roi_distillation_losses((zs, bs), (zt, bt), dist=dist)
            l.backward()
            out.update({f"{tag}_{dist}_loss": l.detach(), f"{tag}_{dist}_dzt": zt.grad.clone(),
                        f"{tag}_{dist}_dbt": bt.grad.clone()})
    save("roi_distill", **out)


# ---------------------------------------------------------------- A16 ARD
def gold_ard():
    g = torch.Generator().manual_seed(8)
    out = {}
    for tag, shape in (("s", (6, 16, 7, 7)), ("m", (2, 1024, 7, 7))):
        fs = torch.randn(*shape, generator=g)
        ft = (fs + 0.5 * torch.randn(*shape, generator=g)).requires_grad_(True)
        out.update({f"{tag}_fs": fs if tag == "s" else fs[:, :, :, :].clone(), f"{tag}_ft": ft.detach()})
        out[f"{tag}_att_s"] = activation_at(fs)
        for gamma in (0.0, 1.0, 5.0):
            ft.grad = None
            l = calculate_attentive_roi_feature_distillation(fs, ft, gamma=gamma)  # (source, target) order!
            l.backward()
            out[f"{tag}_loss_g{int(gamma)}"] = l.detach()
            if tag == "s" or gamma == 1.0:
                out[f"{tag}_dft_g{int(gamma)}"] = ft.grad.clone()
    # keep the file small: the 'm' case stores fp32 [8,1024,7,7] x3 = 4.8 MB raw -> store as float32 anyway (compresses ~10%)
    save("ard", **out)


# ---------------------------------------------------------------- A5 RPN post-processor, A9 RPN loss
def gold_rpn():
    g = torch.Generator().manual_seed(9)
    N, A, H, W = 2, 15, 10, 14
    ag = AnchorGenerator(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0),
                         anchor_strides=(16,), straddle_thresh=0)
    sizes = [(160, 224), (150, 200)]
    img = ImageList(torch.zeros(N, 3, 160, 224), sizes)
    anchors = ag(img, [torch.zeros(N, 8, H, W)])
    obj = torch.randn(N, A, H, W, generator=g) * 2
    reg = torch.randn(N, 4 * A, H, W, generator=g) * 0.5
    out = {"objectness": obj, "box_regression": reg, "image_sizes": np.array(sizes)}
    for tag, (pre, post, train) in {"train": (600, 100, True), "test": (300, 50, False)}.items():
        pp = RPNPostProcessor(pre_nms_top_n=pre, post_nms_top_n=post, nms_thresh=0.7, min_size=0,
                              box_coder=BoxCoder((1., 1., 1., 1.)))
        pp.train(train)
        targets = [BoxList(torch.tensor([[20., 30., 120., 140.]]), (224, 160)),
                   BoxList(torch.tensor([[5., 5., 60., 70.], [90., 40., 190., 140.]]), (200, 150))]
        res = pp(anchors, [obj], [reg], targets if train else None)
        for i, b in enumerate(res):
            out[f"{tag}_boxes{i}"] = b.bbox; out[f"{tag}_scores{i}"] = b.get_field("objectness")
    # RPN loss with the sampler's random choice captured (A9): monkeypatch randperm -> recorded
    gt = [torch.tensor([[20., 30., 120., 140.]]), torch.tensor([[5., 5., 60., 70.], [90., 40., 190., 140.]])]
    targets = [BoxList(b, (s[1], s[0])) for b, s in zip(gt, sizes)]
    ev = RPNLossComputation(Matcher(0.7, 0.3, allow_low_quality_matches=True),
                            BalancedPositiveNegativeSampler(64, 0.5), BoxCoder((1., 1., 1., 1.)), generate_rpn_labels)
    obj_r = obj.clone().requires_grad_(True); reg_r = reg.clone().requires_grad_(True)
    labels, reg_t, _, matched = ev.prepare_targets([a[0] for a in anchors], targets)
    torch.manual_seed(123)
    pos, neg = ev.fg_bg_sampler(labels)
    torch.manual_seed(123)
    lo, lb = ev(anchors, [obj_r], [reg_r], targets)
    (lo + lb).backward()
    out.update(gt0=gt[0], gt1=gt[1], rpn_labels=torch.stack(labels), rpn_reg_targets=torch.stack(reg_t),
               rpn_matched=torch.stack(matched), sampled_pos=torch.stack(pos), sampled_neg=torch.stack(neg),
               loss_objectness=lo.detach(), loss_rpn_box_reg=lb.detach(),
               d_objectness=obj_r.grad, d_box_regression=reg_r.grad,
               anchors0=anchors[0][0].bbox, vis0=anchors[0][0].get_field("visibility"),
               anchors1=anchors[1][0].bbox, vis1=anchors[1][0].get_field("visibility"))
    save("rpn", **out)


def gold_post_processor():
    """F4: test-time PostProcessor (roi_heads/box_head/inference.py:43-151), ragged proposal counts, > 100 candidates."""
    from maskrcnn_benchmark.modeling.roi_heads.box_head.inference import PostProcessor
    g = torch.Generator().manual_seed(21)
    C = 21
    sizes = [(224, 160), (200, 150)]  # BoxList.size = (w, h)
    counts = [120, 90]
    props = []
    for (w, h), n in zip(sizes, counts):
        b = BoxList(rand_boxes(g, n, w, h, 16.0, 110.0), (w, h), mode="xyxy")
        props.append(b)
    K = sum(counts)
    logits = torch.randn(K, C, generator=g) * 2.0
    reg = torch.randn(K, 4 * C, generator=g) * 0.5
    out = {"logits": logits, "box_regression": reg, "counts": np.array(counts), "sizes_wh": np.array(sizes),
           "boxes0": props[0].bbox, "boxes1": props[1].bbox}
    for tag, (thr, nms_t, det) in {"std": (0.05, 0.5, 100), "tight": (0.2, 0.3, 7), "all": (0.05, 0.5, 0)}.items():
        pp = PostProcessor(thr, nms_t, det, BoxCoder(weights=(10., 10., 5., 5.)), False)
        res, bg = pp((logits, reg), props)
        for i, r in enumerate(res):
            out[f"{tag}_boxes{i}"] = r.bbox; out[f"{tag}_scores{i}"] = r.get_field("scores")
            out[f"{tag}_labels{i}"] = r.get_field("labels")
        out[f"{tag}_bg_boxes"] = bg.bbox; out[f"{tag}_bg_scores"] = bg.get_field("scores")
        print(tag, [len(r) for r in res], len(bg))
    save("post_processor", **out)


def gold_voc_eval():
    """F4: VOC mAP on random detections vs random GT (data/datasets/evaluation/voc/voc_eval.py:57-228): 40 images,
    classes 1..20, difficult flags, duplicate detections of one GT, equal scores, images without detections / without GT."""
    import importlib.util  # the package __init__ chain needs torchvision (absent here); the file itself only needs structures/
    spec = importlib.util.spec_from_file_location(
        "ref_voc_eval", "/root/reference/maskrcnn_benchmark/data/datasets/evaluation/voc/voc_eval.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    eval_detection_voc = mod.eval_detection_voc
    rs = np.random.RandomState(13)
    out = {"n_images": 40}
    preds, gts = [], []
    for i in range(40):
        W, H = int(rs.randint(200, 500)), int(rs.randint(200, 400))
        ng = int(rs.randint(0, 5)) if i % 9 else 0
        x1, y1 = rs.rand(ng) * (W - 60), rs.rand(ng) * (H - 60)
        gb = np.stack([x1, y1, x1 + 20 + rs.rand(ng) * 100, y1 + 20 + rs.rand(ng) * 100], 1).astype(np.float32).reshape(-1, 4)
        gl = rs.randint(1, 21, ng).astype(np.int64)
        gd = (rs.rand(ng) < 0.2).astype(np.uint8)
        nd = int(rs.randint(0, 30)) if i % 7 else 0
        db, dl, ds = [], [], []
        for _ in range(nd):
            if ng and rs.rand() < 0.6:   # jittered copy of a GT, right or wrong label
                k = rs.randint(ng)
                db.append(gb[k] + rs.randn(4) * 8); dl.append(gl[k] if rs.rand() < 0.8 else rs.randint(1, 21))
            else:
                a, b = rs.rand() * (W - 60), rs.rand() * (H - 60)
                db.append([a, b, a + 20 + rs.rand() * 100, b + 20 + rs.rand() * 100]); dl.append(rs.randint(1, 21))
            ds.append(np.round(rs.rand(), 2))  # 2 decimals -> equal scores occur
        db = np.array(db, np.float32).reshape(-1, 4); dl = np.array(dl, np.int64); ds = np.array(ds, np.float32)
        p = BoxList(torch.from_numpy(db), (W, H)); p.add_field("labels", torch.from_numpy(dl)); p.add_field("scores", torch.from_numpy(ds))
        t = BoxList(torch.from_numpy(gb), (W, H)); t.add_field("labels", torch.from_numpy(gl)); t.add_field("difficult", torch.from_numpy(gd))
        preds.append(p); gts.append(t)
        out.update({f"size{i}": np.array([W, H]), f"gb{i}": gb, f"gl{i}": gl, f"gd{i}": gd, f"db{i}": db, f"dl{i}": dl, f"ds{i}": ds})
    for tag, m07 in (("area", False), ("voc07", True)):
        r = eval_detection_voc(preds, gts, iou_thresh=0.5, use_07_metric=m07)
        out[f"ap_{tag}"] = r["ap"]; out[f"map_{tag}"] = r["map"]
        print(tag, r["map"], np.round(r["ap"], 3))
    save("voc_eval", **out)


def gold_rehearsal():
    """F2: prototype box selection (tools/extract_memory.py Mem): which records the REFERENCE picks, and in which order, for the
    'mean' and 'random' strategies (herding crashes in the reference: UnboundLocalError, extract_memory.py:203).  The image
    writer is replaced by a recorder; 3 new classes with 40 / 9 / 3 candidates and 5 slots per class (the last is topped up)."""
    import random
    import tempfile
    import types
    sys.path.insert(0, "/root/reference")
    from tools.extract_memory import Mem
    rs = np.random.RandomState(5)
    counts = [40, 9, 3]
    feats = [np.abs(rs.randn(n, 7, 7)).astype(np.float32) * (1 + c) for c, n in enumerate(counts)]
    out = {"counts": np.array(counts), "mem_size": 18}
    for c, f in enumerate(feats):
        out[f"feat{c}"] = f
    for mem_type in ("mean", "random"):
        cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(ROI_BOX_HEAD=types.SimpleNamespace(
            NAME_OLD_CLASSES=["a"], NAME_NEW_CLASSES=["b", "c", "d"]), SOURCE_WEIGHT=""), MEM_TYPE=mem_type, MEM_BUFF=18,
            TASK="t", NAME="n")
        with tempfile.TemporaryDirectory() as d:
            m = Mem(cfg, 0, d)
            picked = []

            def rec(self, info, ind, picked=picked, d=d):
                picked.append((info["box_class"], ind, info["rid"]))
                open(os.path.join(d, "{0}_{1:05d}_{2}.jpg".format(info["box_class"], ind, len(picked))), "w").close()
            m.creat_and_save_box_image = types.MethodType(rec, m)
            info = [[{"feature": feats[c][j].tolist(), "logits": None, "image_path": ["x"], "box_class": 2 + c, "box": [0, 0, 99, 99],
                      "mode": "xyxy", "rid": j} for j in range(n)] for c, n in enumerate(counts)]
            random.seed(3)
            try:
                m.update_memory(info)
            except AssertionError:   # 4 classes x ceil(18/4)=5 slots, only the 3 new ones are written here: 15 < 18
                pass
            out[f"{mem_type}_picked"] = np.array(picked)
            print(mem_type, picked[:6], len(picked))
    save("rehearsal", **out)


def gold_ablation_distill():
    """The two ablation losses (distillation.py:18-84, :133-161).  The reference functions allocate their zero filters with
    `.to('cuda')`; on this CPU-only container that one device move is neutralised (Tensor.to ignoring the 'cuda' target) so that the
    reference's own arithmetic runs and produces the golden numbers."""
    from maskrcnn_benchmark.distillation.distillation import calculate_feature_distillation_loss, calculate_rpn_distillation_loss
    g = torch.Generator().manual_seed(31)
    fs = [torch.randn(2, 24, 10, 14, generator=g)]
    ft = [(fs[0] + 0.3 * torch.randn(2, 24, 10, 14, generator=g)).requires_grad_(True)]
    N, A, H, W = 2, 15, 10, 14
    obj_s = [torch.randn(N, A, H, W, generator=g)]; reg_s = [torch.randn(N, 4 * A, H, W, generator=g) * 0.5]
    obj_t = [(obj_s[0] + 0.4 * torch.randn(N, A, H, W, generator=g)).requires_grad_(True)]
    reg_t = [(reg_s[0] + 0.2 * torch.randn(N, 4 * A, H, W, generator=g)).requires_grad_(True)]
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        if a and a[0] == "cuda":
            return self
        return orig_to(self, *a, **k)
    torch.Tensor.to = to_cpu
    orig_empty_cache = torch.cuda.empty_cache
    torch.cuda.empty_cache = lambda: None
    try:
        lf = calculate_feature_distillation_loss(fs, ft, loss="normalized_filtered_l1")
        lf.backward()
        lr = calculate_rpn_distillation_loss((obj_s, reg_s), (obj_t, reg_t), cls_loss="filtered_l2", bbox_loss="l2", bbox_threshold=0.1)
        lr.backward()
    finally:
        torch.Tensor.to = orig_to
        torch.cuda.empty_cache = orig_empty_cache
    save("ablation_distill", feat_s=fs[0], feat_t=ft[0], loss_feat=lf.detach(), d_feat_t=ft[0].grad,
         obj_s=obj_s[0], reg_s=reg_s[0], obj_t=obj_t[0], reg_t=reg_t[0], loss_rpn=lr.detach(), d_obj_t=obj_t[0].grad, d_reg_t=reg_t[0].grad)
    print(float(lf), float(lr))


if __name__ == "__main__":
    torch.set_num_threads(4)
    gold_anchors(); gold_box_coder(); gold_matcher(); gold_nms(); gold_roi_align()
    gold_elementwise(); gold_box_head_loss(); gold_roi_distill(); gold_ard(); gold_rpn(); gold_post_processor(); gold_voc_eval(); gold_rehearsal(); gold_ablation_distill()
    print("done")
