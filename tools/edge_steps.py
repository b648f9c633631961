#!/usr/bin/env python3
"""Unusual-but-legal batches through the default (overlapped, prefetching) training step at full width: odd image extents, one image, seven
images, 60 ground-truth boxes per image, a box that covers the whole image, 8-pixel boxes, boxes on the borders, very small and very large
images, batches of changing shape back to back.  Every case: finite losses and gradients, and the same losses as the step with every stream
folded into one (a missing dependency or a shape assumption shows as a mismatch or a launch error).  GPU box: python tools/edge_steps.py"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402  (fold_streams)
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402
from abr_iod_amd.structures.bounding_box import BoxList  # noqa: E402


def with_boxes(batch, boxes_per_image, labels=(16, 21), seed=0):
    """replace the targets of a synthetic batch by explicit box lists [[(x1, y1, x2, y2), ...], ...]"""
    images, targets = batch
    g = torch.Generator().manual_seed(seed)
    out = []
    for t, boxes in zip(targets, boxes_per_image):
        b = BoxList(torch.tensor(boxes, dtype=torch.float32, device="cuda").view(-1, 4), t.size, mode="xyxy")
        b.add_field("labels", torch.randint(labels[0], labels[1], (len(boxes),), generator=g).cuda())
        out.append(b)
    return images, out


def many(n, w, h, seed):
    g = random.Random(seed)
    res = []
    for _ in range(n):
        x1, y1 = g.uniform(0, w - 40), g.uniform(0, h - 40)
        res.append((x1, y1, min(w - 1, x1 + g.uniform(10, 300)), min(h - 1, y1 + g.uniform(10, 300))))
    return res


cases = []
cases.append(("odd extents 601x1001, B = 3", synthetic_batch(3, 601, 1001, seed=1)))
cases.append(("one image", synthetic_batch(1, 600, 1000, seed=2)))
cases.append(("seven images 480x640", synthetic_batch(7, 480, 640, seed=3)))
cases.append(("60 boxes per image", with_boxes(synthetic_batch(2, 600, 1000, seed=4), [many(60, 1000, 600, 1), many(60, 1000, 600, 2)])))
cases.append(("a box covering the whole image + 8-pixel boxes", with_boxes(synthetic_batch(2, 600, 1000, seed=5),
                                                                         [[(0, 0, 999, 599)], [(10, 10, 18, 18), (500, 300, 508, 308), (991, 591, 999, 599)]])))
cases.append(("boxes on the borders", with_boxes(synthetic_batch(2, 600, 1000, seed=6), [[(0, 0, 50, 599), (949, 0, 999, 599)], [(0, 0, 999, 40), (0, 559, 999, 599)]])))
cases.append(("small images 224x320, B = 4", synthetic_batch(4, 224, 320, seed=7, max_boxes=2)))
cases.append(("large image 1000x1666, B = 1", synthetic_batch(1, 1000, 1666, seed=8)))
cases.append(("portrait 1000x600, B = 2", synthetic_batch(2, 1000, 600, seed=9)))


def run(fold):
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=4, base_lr=0.0)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    if fold:
        opt._folded_saved = bench.fold_streams(True, opt)
    res = []
    try:
        for i, (name, (images, targets)) in enumerate(cases):
            torch.manual_seed(100 + i)
            random.seed(100 + i)
            nxt = cases[(i + 1) % len(cases)][1][0]
            ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=None if fold else nxt)
            g = mt.flat.grads
            res.append((name, {k: float(v.detach()) for k, v in ld.items()}, float(total.detach()), bool(torch.isfinite(g).all()), float(g.norm())))
    finally:
        if fold:
            bench.fold_streams(False, opt)
    torch.cuda.synchronize()
    return res


over, ser = run(False), run(True)
bad = 0
for (name, ld, total, fin, gn), (_, ld2, total2, fin2, gn2) in zip(over, ser):
    dl = max(abs(ld[k] - ld2[k]) / max(1.0, abs(ld2[k])) for k in ld)
    dg = abs(gn - gn2) / max(gn2, 1e-30)
    ok = fin and fin2 and dl <= 1e-4 and dg <= 1e-4 and all(v == v and abs(v) < 1e6 for v in ld.values())
    bad += not ok
    print("%-48s total loss %9.5f (folded %9.5f)  worst loss rel. diff %.1e  |grad| rel. diff %.1e  %s" % (name, total, total2, dl, dg, "ok" if ok else "MISMATCH"))
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
