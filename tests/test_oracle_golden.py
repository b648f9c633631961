"""Pin the oracle (oracle/oracle.c + oracle/torch_ref.py) against golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import ops as O
from oracle import torch_ref as R


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_anchors_bit_exact(gold):
    g = gold("anchors")
    cell = O.cell_anchors()
    assert np.array_equal(cell, g["cell"].astype(np.float32))
    H, W = g["hw"]
    for i in (0, 1):
        a, v = O.grid_anchors(cell, int(H), int(W), 16, g["image_sizes"][i])
        assert np.array_equal(a, g[f"bbox{i}"])
        assert np.array_equal(v, g[f"vis{i}"].astype(bool))


def test_box_coder(gold):
    g = gold("box_coder")
    for tag, w in (("rpn", (1, 1, 1, 1)), ("head", (10, 10, 5, 5))):
        np.testing.assert_allclose(O.box_encode(g["gt"], g["ex"], w), g[f"{tag}_enc"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(O.box_decode(g[f"{tag}_deltas"], g["ex"], w), g[f"{tag}_dec"], rtol=1e-5, atol=1e-3)


def test_iou_matcher_index_exact(gold):
    g = gold("matcher")
    iou = O.box_iou(g["gt"], g["prop"])
    np.testing.assert_array_equal(iou, g["iou"])  # same fp32 op order -> bit-exact
    assert np.array_equal(O.matcher(iou, 0.7, 0.3, True), g["rpn_matched"])
    assert np.array_equal(O.matcher(iou, 0.5, 0.5, False), g["head_matched"])


def test_nms_index_exact(gold):
    g = gold("nms")
    assert np.array_equal(O.nms(g["boxes"], g["scores"], 0.5), g["keep_5"])
    assert np.array_equal(O.nms(g["boxes"], g["scores"], 0.7), g["keep_7"])
    # the IoU==thr pair: CPU '>=' drops box 1, CUDA '>' keeps it (nms_cpu.cpp:60 vs nms.cu:60)
    assert 1 not in O.nms(g["boxes"], g["scores"], 0.5)
    assert 1 in O.nms(g["boxes"], g["scores"], 0.5, strict_gt=True)
    # Appendix-B known answer
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 19], [100, 100, 109, 109]], np.float32)
    s = np.array([.9, .8, .7], np.float32)
    assert O.nms(b, s, 0.5).tolist() == [0, 2] and O.nms(b, s, 0.51).tolist() == [0, 1, 2]
    assert O.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).size == 0


def test_roi_align_forward_bit_exact(gold):
    g = gold("roi_align")
    for sr in (0, 2):
        np.testing.assert_array_equal(O.roi_align_forward(g["feat"], g["rois"], 1 / 16, 7, 7, sr), g[f"out_sr{sr}"])
    np.testing.assert_array_equal(O.roi_align_forward(g["feat"], g["rois"], 0.0625, 3, 5, 0), g["out_sr0_3x5"])
    g = gold("roi_align_c4")
    for sr in (0, 2):
        np.testing.assert_array_equal(O.roi_align_forward(g["feat"], g["rois"], 0.0625, 7, 7, sr), g[f"out_sr{sr}"])
    # SURVEY Appendix B known answers
    feat = np.arange(20, dtype=np.float32).reshape(1, 1, 4, 5)
    out = O.roi_align_forward(feat, np.array([[0, 1, 1, 10, 8]], np.float32), 0.5, 2, 2, 2).ravel()
    np.testing.assert_allclose(out, [8.5, 10.53125, 15.84375, 17.875])
    out = O.roi_align_forward(feat, np.array([[0, -4, -4, 2, 2], [0, 7, 5, 7.4, 5.2]], np.float32), 0.5, 2, 2, 0).ravel()
    np.testing.assert_allclose(out, [0, 0.15625, 0.78125, 1.875, 17.5, 17.75, 18.75, 19.0], rtol=1e-6)


def test_roi_align_backward_is_adjoint_of_forward(gold):
    """The reference has no CPU backward; the CUDA formula is the transpose of the forward's linear map:
    <fwd(x), g> == <x, bwd(g)> for all x, g, and finite differences agree."""
    g = gold("roi_align")
    rng = np.random.default_rng(0)
    feat = g["feat"]; rois = g["rois"]
    for sr in (0, 2):
        gy = rng.standard_normal((rois.shape[0], 8, 7, 7)).astype(np.float32)
        y = O.roi_align_forward(feat, rois, 1 / 16, 7, 7, sr)
        gx = O.roi_align_backward(gy, rois, 1 / 16, 7, 7, 2, 8, 10, 14, sr)
        lhs = float((y.astype(np.float64) * gy).sum()); rhs = float((feat.astype(np.float64) * gx).sum())
        assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_smooth_l1_and_focal(gold):
    g = gold("smooth_l1")
    for tag, beta, avg in (("b19_sum", 1 / 9, False), ("b1_sum", 1.0, False), ("b19_mean", 1 / 9, True)):
        l, gr = O.smooth_l1(g["x"], g["t"], beta, avg)
        np.testing.assert_allclose(l, g[f"{tag}_loss"], rtol=2e-6)
        np.testing.assert_allclose(gr, g[f"{tag}_grad"], rtol=1e-5, atol=1e-7)
    g = gold("sigmoid_focal")
    l = O.sigmoid_focal_forward(g["logits"], g["targets"], 2.0, 0.25)
    np.testing.assert_allclose(l, g["loss"], rtol=2e-5, atol=1e-7)   # python-CPU fallback uses log(1-p): ~1e-6 apart
    d = O.sigmoid_focal_backward(g["logits"], g["targets"], g["d_loss"], 2.0, 0.25)
    np.testing.assert_allclose(d, g["d_logits"], rtol=2e-4, atol=2e-6)


def test_box_head_loss(gold):
    g = gold("box_head_loss")
    for (ko, ka) in ((16, 21), (11, 21), (11, 16)):
        p = f"k{ko}_{ka}"
        for dist in ("id", "l2"):
            lg = T(g[f"{p}_logits"]).requires_grad_(True); rg = T(g[f"{p}_reg"]).requires_grad_(True)
            c, b = R.box_head_loss(lg, rg, T(g[f"{p}_labels"]), T(g[f"{p}_rt"]), dist, ko - 1)
            (c + b).backward()
            np.testing.assert_allclose(c.item(), g[f"{p}_{dist}_cls"], rtol=1e-6)
            np.testing.assert_allclose(b.item(), g[f"{p}_{dist}_box"], rtol=1e-6)
            np.testing.assert_allclose(lg.grad.numpy(), g[f"{p}_{dist}_dlogits"], rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(rg.grad.numpy(), g[f"{p}_{dist}_dreg"], rtol=1e-5, atol=1e-8)


def test_roi_distillation(gold):
    g = gold("roi_distill")
    for (ko, ka) in ((16, 21), (11, 21), (11, 16), (21, 21)):
        p = f"k{ko}_{ka}"
        for dist in ("id", "l2"):
            if f"{p}_{dist}_loss" not in g.files:
                continue
            zt = T(g[f"{p}_zt"]).requires_grad_(True); bt = T(g[f"{p}_bt"]).requires_grad_(True)
            l = R.roi_distillation_loss(T(g[f"{p}_zs"]), T(g[f"{p}_bs"]), zt, bt, dist)
            l.backward()
            np.testing.assert_allclose(l.item(), g[f"{p}_{dist}_loss"], rtol=2e-6)
            np.testing.assert_allclose(zt.grad.numpy(), g[f"{p}_{dist}_dzt"], rtol=1e-4, atol=1e-9)
            np.testing.assert_allclose(bt.grad.numpy(), g[f"{p}_{dist}_dbt"], rtol=1e-5, atol=1e-9)
    # Appendix B closed-form answers
    ar = lambda n: torch.arange(n, dtype=torch.float32)
    zs = torch.sin(0.5 * ar(12)).view(3, 4); bs = torch.sin(0.2 * ar(48)).view(3, 4, 4)
    zt = torch.cos(0.3 * ar(18)).view(3, 6); bt = torch.cos(0.1 * ar(72)).view(3, 6, 4)
    assert abs(R.roi_distillation_loss(zs, bs, zt, bt, "id").item() - 0.8528258) < 2e-6
    assert abs(R.roi_distillation_loss(zs, bs, zt, bt, "l2").item() - 0.6494453) < 2e-6


def test_ard(gold):
    g = gold("ard")
    for tag in ("s", "m"):
        fs = T(g[f"{tag}_fs"])
        np.testing.assert_allclose(R.attention_map(fs).numpy(), g[f"{tag}_att_s"], rtol=1e-5)
        for gamma in (0, 1, 5):
            ft = T(g[f"{tag}_ft"]).requires_grad_(True)
            l = R.ard_loss(fs, ft, float(gamma))
            np.testing.assert_allclose(l.item(), g[f"{tag}_loss_g{gamma}"], rtol=1e-6)
            if f"{tag}_dft_g{gamma}" in g.files:
                l.backward()
                np.testing.assert_allclose(ft.grad.numpy(), g[f"{tag}_dft_g{gamma}"], rtol=1e-4, atol=1e-9)
    ar = lambda n: torch.arange(n, dtype=torch.float32)
    f_s = torch.sin(0.37 * ar(72)).view(2, 4, 3, 3); f_t = torch.cos(0.11 * ar(72)).view(2, 4, 3, 3)
    for gamma, want in ((0, 1.0373448), (1, 1.3012114), (5, 2.3566775)):
        assert abs(R.ard_loss(f_s, f_t, gamma).item() - want) < 3e-6


def test_rpn_targets_loss_and_proposals(gold):
    g = gold("rpn")
    anchors = [g["anchors0"], g["anchors1"]]; vis = [g["vis0"].astype(bool), g["vis1"].astype(bool)]
    gts = [g["gt0"], g["gt1"]]
    labs, tgts = [], []
    for i in range(2):
        lab, tgt, m = R.rpn_prepare_targets(anchors[i], vis[i], gts[i])
        assert np.array_equal(m, g["rpn_matched"][i])
        assert np.array_equal(lab, g["rpn_labels"][i])
        np.testing.assert_allclose(tgt, g["rpn_reg_targets"][i], rtol=1e-5, atol=1e-6)
        labs.append(lab); tgts.append(tgt)
    obj = T(g["objectness"]).requires_grad_(True); reg = T(g["box_regression"]).requires_grad_(True)
    lo, lb = R.rpn_loss(obj, reg, T(np.stack(labs)), T(np.stack(tgts)), T(g["sampled_pos"]), T(g["sampled_neg"]))
    (lo + lb).backward()
    np.testing.assert_allclose(lo.item(), g["loss_objectness"], rtol=1e-6)
    np.testing.assert_allclose(lb.item(), g["loss_rpn_box_reg"], rtol=1e-6)
    np.testing.assert_allclose(obj.grad.numpy(), g["d_objectness"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(reg.grad.numpy(), g["d_box_regression"], rtol=1e-5, atol=1e-9)
    for tag, pre, post, train in (("train", 600, 100, True), ("test", 300, 50, False)):
        res = R.rpn_post_process(T(g["objectness"]), T(g["box_regression"]), anchors, g["image_sizes"], pre, post,
                                 gt_boxes=gts if train else None)
        for i, (b, s) in enumerate(res):
            assert b.shape == g[f"{tag}_boxes{i}"].shape
            np.testing.assert_allclose(b, g[f"{tag}_boxes{i}"], rtol=1e-5, atol=2e-4)
            np.testing.assert_allclose(s, g[f"{tag}_scores{i}"], rtol=1e-6)


def test_post_processor_against_reference(gold):
    """F4: the oracle's PostProcessor restatement reproduces the reference's detections (boxes, scores, labels, order) on a
    ragged 2-image batch for three (score_thresh, nms, detections_per_img) settings."""
    g = gold("post_processor")
    props = [g["boxes0"], g["boxes1"]]
    for tag, (thr, nms_t, det) in {"std": (0.05, 0.5, 100), "tight": (0.2, 0.3, 7), "all": (0.05, 0.5, 0)}.items():
        res, bg = R.post_process(T(g["logits"]), T(g["box_regression"]), props, g["sizes_wh"], thr, nms_t, det)
        for i, (b, s, l) in enumerate(res):
            assert np.array_equal(l, g[f"{tag}_labels{i}"]), tag
            np.testing.assert_allclose(s, g[f"{tag}_scores{i}"], rtol=1e-6)
            np.testing.assert_allclose(b, g[f"{tag}_boxes{i}"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(bg[1], g[f"{tag}_bg_scores"], rtol=1e-6)
        np.testing.assert_allclose(bg[0], g[f"{tag}_bg_boxes"], rtol=1e-5, atol=2e-4)


def test_ablation_distillation_losses_against_reference(gold):
    """DIST.FEAT='std' / DIST.RPN ablation losses: the oracle restatement reproduces the reference's values and gradients."""
    g = gold("ablation_distill")
    ft = T(g["feat_t"]).requires_grad_(True)
    lf = R.feature_distillation_loss([T(g["feat_s"])], [ft])
    lf.backward()
    assert abs(float(lf) - float(g["loss_feat"])) < 1e-6
    np.testing.assert_allclose(ft.grad.numpy(), g["d_feat_t"], rtol=1e-5, atol=1e-9)
    ot, rt = T(g["obj_t"]).requires_grad_(True), T(g["reg_t"]).requires_grad_(True)
    lr = R.rpn_distillation_loss(([T(g["obj_s"])], [T(g["reg_s"])]), ([ot], [rt]), 0.1)
    lr.backward()
    assert abs(float(lr) - float(g["loss_rpn"])) < 1e-6
    np.testing.assert_allclose(ot.grad.numpy(), g["d_obj_t"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(rt.grad.numpy(), g["d_reg_t"], rtol=1e-5, atol=1e-9)
