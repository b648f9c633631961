#!/usr/bin/env python3
"""Where does a training step spend its time?  Per stage: host enqueue time (no sync) vs wall time with a device sync."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
from abr_iod_amd.distillation.distillation import calculate_attentive_roi_feature_distillation, calculate_roi_distillation_losses
from abr_iod_amd.structures.image_list import to_image_list

cfg_s, cfg_t = make_cfgs("15-5")
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(4)

def run(sync):
    T = {}
    def mark(name, t0):
        if sync: torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        il = to_image_list(images)
        fs, _ = ms.backbone(il.tensors); t = mark("src.backbone", t)
        (props_s, _), anc, rpn_s = ms.rpn(il, fs, None); t = mark("src.rpn(head+proposals)", t)
        import random
        sel = []
        from abr_iod_amd.structures.bounding_box import BoxList
        allsel = []
        for p in props_s:
            order = p.get_field("objectness").sort(descending=True)[1]
            p = p[order]
            idx = torch.tensor(random.sample(range(0, 128), 64), device="cuda")
            b = BoxList(p.bbox.index_select(0, idx), p.size, p.mode); b.add_field("objectness", p.get_field("objectness").index_select(0, idx)); allsel.append(b)
        t = mark("src.select64", t)
        zs, bs, _, raf_s = ms.roi_heads.calculate_soften_label(fs, allsel); t = mark("src.head64", t)
    ft, _ = mt.backbone(il.tensors); t = mark("tgt.backbone", t)
    fused = mt.rpn.head.forward_fused(ft[0]); t = mark("tgt.rpn.head", t)
    anchors = mt.rpn.anchor_generator(il, ft)
    with torch.no_grad():
        mt.rpn.box_selector_train.train()
        boxes = mt.rpn.box_selector_train.forward_fused(anchors, fused.detach(), 15, targets)
    t = mark("tgt.rpn.proposals", t)
    lo, lb = mt.rpn.loss_evaluator(anchors, None, None, targets, fused=fused); t = mark("tgt.rpn.loss", t)
    with torch.no_grad():
        props = mt.roi_heads.box.loss_evaluator.subsample(boxes, targets)
    t = mark("tgt.head.subsample", t)
    x, raf = mt.roi_heads.box.feature_extractor(ft, props, need_roi_features=False); t = mark("tgt.head.roialign+layer4", t)
    fz = mt.roi_heads.box.predictor.forward_fused(x)
    lc, lbox = mt.roi_heads.box.loss_evaluator(21, None, fused=fz); t = mark("tgt.head.pred+loss", t)
    tr, _, raf_t = mt.forward(images, targets, features=ft, proposals=allsel); t = mark("tgt.head64", t)
    l_id = calculate_roi_distillation_losses((zs, bs), tr, dist="id")
    l_ard = calculate_attentive_roi_feature_distillation(raf_s, raf_t, 1.0); t = mark("distill losses", t)
    total = lo + lb + lc + lbox + 0.5 * l_id + l_ard
    opt.zero_grad(); t = mark("zero_grad", t)
    total.backward(); t = mark("backward", t)
    opt.step(); sch.step(); t = mark("sgd", t)
    torch.cuda.synchronize()
    return T

for _ in range(3): run(False)
torch.cuda.synchronize(); t0 = time.perf_counter(); A = run(False); wall_nosync = (time.perf_counter() - t0) * 1e3
B = run(True)
print(f"{'stage':28s} {'host-only ms':>12s} {'with sync ms':>12s}")
for k in A:
    print(f"{k:28s} {A[k]:12.2f} {B[k]:12.2f}")
print(f"{'TOTAL':28s} {sum(A.values()):12.2f} {sum(B.values()):12.2f}   wall(no per-stage sync) {wall_nosync:.2f}")
