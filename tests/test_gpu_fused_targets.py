"""GPU: the device-resident proposal -> box-head-target path (rpn.LazyProposals -> ops.roi_head_targets, abr_roi_head_targets) and the
fused gather of the source model's distillation proposals (abr_gather_proposals) against the reference-shaped path they replace:
RPNPostProcessor.collect + add_gt_proposals (rpn/inference.py:53-74,113-147), FastRCNNLossComputation.subsample
(box_head/loss.py:56-120), Pooler.convert_to_roi_format (poolers.py:73-86), generalized_rcnn.py:140-158.

The sampler's draw is device RNG, so the generic path replays the fused path's draw (inject_sampled_inds); everything else -- the
candidate lists, labels, regression targets, the RoI table, both losses and their gradients -- must then be IDENTICAL (same
arithmetic, same order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TINY = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
        "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128, "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 600, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300,
        "MODEL.RPN.POST_NMS_TOP_N_TEST", 150, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64]


def _setup(post_nms, rois_per_image, extra=()):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2e_common import clamp_targets
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    cfg_s, cfg_t = make_cfgs("15-5", overrides=TINY + ["MODEL.RPN.POST_NMS_TOP_N_TRAIN", post_nms, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", rois_per_image] + list(extra))
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    images, targets = synthetic_batch(3, 160, 224, seed=5, max_boxes=3)
    clamp_targets(targets, 224, 160)
    return cfg_t, ms, mt, images, targets


@pytest.mark.parametrize("post_nms,R", [(100, 48), (20, 48)])   # the second case has fewer candidates than RoIs per image: padded rows
def test_fused_box_head_targets_equal_generic_path(post_nms, R):
    from abr_iod_amd.modeling.roi_heads.box_head.box_head import convert_to_roi_format
    from abr_iod_amd.modeling.rpn.rpn import LazyProposals
    cfg, ms, mt, images, targets = _setup(post_nms, R)
    ev = mt.roi_heads.box.loss_evaluator
    K = mt.roi_heads.box.predictor.num_classes
    begun = mt.forward_begin(images, targets)
    (boxes, _), _, _ = mt.rpn.forward_finish(begun["rpn"])
    assert isinstance(boxes, LazyProposals)
    feats = begun["features"]

    # ---- fused
    torch.manual_seed(3)
    x, result, soft, losses, raf = mt.roi_heads(feats, boxes, targets)
    t = ev._fused_targets
    assert t is not None
    gf, = torch.autograd.grad(losses["loss_classifier"] + losses["loss_box_reg"], feats[0], retain_graph=False)
    lists = boxes.materialize()                                  # the reference-shaped candidate lists (post-NMS + GT)
    n_cand = t["n_cand"].tolist()
    counts = t["counts"].tolist()
    N = len(lists)
    for i in range(N):
        n = len(lists[i])
        assert n_cand[i] == n
        assert torch.equal(t["cand"][i, :n], lists[i].bbox)
        assert torch.equal(t["obj_all"][i, :n], lists[i].get_field("objectness"))
        assert bool((t["labels_all"][i, n:] == -1).all())
        # labels / targets of EVERY candidate equal the per-image matcher's (index-exact / bit-exact)
        _, lab, tgt = ev.proposal_matcher.match_boxes(targets[i].bbox, lists[i].bbox, targets[i].get_field("labels").to(torch.int64), None,
                                                      ev.box_coder.weights, rpn_labels=False)
        assert torch.equal(t["labels_all"][i, :n], lab)
        assert torch.equal(t["regt_all"][i, :n], tgt)
        # sampler quotas (balanced_positive_negative_sampler.py:44-60) and the ascending, duplicate-free merged list
        cp, cn = counts[i]
        npos_avail, nneg_avail = int((lab >= 1).sum()), int((lab == 0).sum())
        assert cp == min(npos_avail, int(R * 0.25)) and cn == min(nneg_avail, R - cp)
        idx = t["sampled_idx"][i]
        drawn = idx[: cp + cn]
        assert bool((idx[cp + cn:] == -1).all()) and bool((drawn[1:] > drawn[:-1]).all())
        assert int((lab[drawn] >= 1).sum()) == cp and int((lab[drawn] == 0).sum()) == cn
    total = sum(c[0] + c[1] for c in counts)
    assert float(t["n_valid"]) == total
    if post_nms == 20:
        assert total < N * R                                    # the padded case really is exercised
        pad = t["labels"] == -1
        assert int(pad.sum()) == N * R - total and bool((t["pos_rows"][pad] == -1).all())
    else:
        assert total == N * R

    # ---- generic path, the same draw replayed
    ev.inject_sampled_inds = [t["sampled_idx"][i][t["sampled_idx"][i] >= 0] for i in range(N)]
    try:
        x2, result2, soft2, losses2, raf2 = mt.roi_heads(feats, boxes, targets)
    finally:
        ev.inject_sampled_inds = None
    assert ev._fused_targets is None
    rois2 = convert_to_roi_format(ev._proposals)
    valid = t["labels"] >= 0 if post_nms == 20 else torch.ones_like(t["labels"], dtype=torch.bool)
    assert torch.equal(t["rois"][valid], rois2)
    assert torch.equal(t["labels"][valid], torch.cat([p.get_field("labels") for p in ev._proposals]))
    assert torch.equal(t["reg_targets"][valid], torch.cat([p.get_field("regression_targets") for p in ev._proposals]))
    assert torch.equal(t["obj"][valid], torch.cat([p.get_field("objectness") for p in ev._proposals]))
    for k in ("loss_classifier", "loss_box_reg"):
        a, b = float(losses[k]), float(losses2[k])
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (k, a, b)
    gf2, = torch.autograd.grad(losses2["loss_classifier"] + losses2["loss_box_reg"], feats[0])
    assert float((gf - gf2).abs().max()) <= 1e-6 * max(1e-12, float(gf2.abs().max()))
    # the BoxLists the API returns are views of the fused tables
    for i in range(N):
        assert result[i].bbox.shape == (R, 4) and result[i].has_field("labels") and result[i].has_field("regression_targets")


def test_fused_soften_gather_equals_boxlist_path():
    cfg, ms, mt, images, targets = _setup(100, 48, ["MODEL.RPN.PRE_NMS_TOP_N_TEST", 2000, "MODEL.RPN.POST_NMS_TOP_N_TEST", 400])
    import random
    random.seed(7)
    with torch.no_grad():
        fused = ms.soften_finish(ms.soften_begin(images))                       # deferred selection + one gather kernel
        picks = ms.last_soften_indices
        assert all(len(p) == 64 for p in picks) and all(hasattr(b, "_roi_table") for b in fused[2])   # the fused path really ran
        generic = ms.generate_soften_proposal(images, selected_indices=picks)   # reference-shaped: cut, sort, index per image
    (zs, bs), _, sel, _, _, _, _, raf = fused
    (zs2, bs2), _, sel2, _, _, _, _, raf2 = generic
    for a, b in zip(sel, sel2):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("objectness"), b.get_field("objectness"))
    assert torch.equal(raf, raf2) and torch.equal(zs, zs2) and torch.equal(bs, bs2)
    # the target's second RoI pass takes the source's RoI table as is
    from abr_iod_amd.modeling.roi_heads.box_head.box_head import convert_to_roi_format
    tab = convert_to_roi_format(sel)
    assert tab.data_ptr() == sel[0]._roi_table[0].data_ptr() and torch.equal(tab, convert_to_roi_format(sel2))


def test_table_caches_survive_eviction_inside_one_launch():
    """ops._pointer_table / ops._small_table cache the small device tables a launch needs (GT pointers, GT counts, visibility pointers).
    A launch builds several of them; when the cache was full, building the second used to CLEAR the cache, which freed the first --
    and the second upload reused its memory before the kernel had read it (one training step with garbage ground truth, seen once in a
    long test session).  Fill the caches to the brim and check that the targets still equal the ones computed with empty caches."""
    from abr_iod_amd import ops
    rng = np.random.default_rng(0)
    N, n = 2, 4000
    anchors = torch.tensor(np.stack([rng.uniform(0, 200, n), rng.uniform(0, 150, n), rng.uniform(210, 400, n), rng.uniform(160, 300, n)], -1), dtype=torch.float32, device="cuda")
    vis = [torch.ones(n, dtype=torch.uint8, device="cuda") for _ in range(N)]
    gts = [torch.tensor([[20, 30, 300, 250], [100, 80, 390, 280]], dtype=torch.float32, device="cuda"),
           torch.tensor([[5, 5, 220, 170]], dtype=torch.float32, device="cuda")]
    ops._ptr_tables.clear(); ops._small_tables.clear()
    want = ops.rpn_targets_batched(anchors, vis, gts, 0.7, 0.3, (1.0, 1.0, 1.0, 1.0))[:2]
    want = [t.clone() for t in want]
    for fill in (62, 63, 64, 65):                        # the eviction lands on the first / second / third table of the launch
        ops._ptr_tables.clear(); ops._small_tables.clear()
        for i in range(fill):
            ops._ptr_tables[((i,), torch.device("cuda", 0))] = torch.zeros(1, dtype=torch.int64, device="cuda")
        for i in range(255):
            ops._small_tables[((i, -1), torch.int32, torch.device("cuda", 0))] = torch.zeros(1, dtype=torch.int32, device="cuda")
        labels, tgt = ops.rpn_targets_batched(anchors, vis, gts, 0.7, 0.3, (1.0, 1.0, 1.0, 1.0))[:2]
        torch.cuda.synchronize()
        assert torch.equal(labels, want[0]) and torch.equal(tgt, want[1]), fill
    ops._ptr_tables.clear(); ops._small_tables.clear()
