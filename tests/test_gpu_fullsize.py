"""GPU: size-independent properties at BASELINE.json's full configuration (configs[2]: task 15-5, ID + ARD, batch 4, 600x1000 images,
12000 -> 2000 proposals, 512 sampled RoIs per image), where the CPU oracle would take minutes per step:
  * convolutions of the full-size layer shapes: exact linearity under power-of-two scaling, and float64 dot products at sampled
    output positions (direct, split-K and Winograd kernels all take part at these shapes);
  * RPN selection: counts, ordering, clipping, min size and the NMS invariant (no kept pair above the IoU threshold);
  * the samplers' quotas (rpn/loss.py + balanced_positive_negative_sampler.py:20-68);
  * the attentive feature term vanishes at step 0, where the target's backbone still equals the source's (train_incremental.py:113-116);
  * the forward's index work is bit-reproducible from a seed, frozen tensors stay frozen, and trainable ones move."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, H, W = 4, 600, 1000


@pytest.fixture(scope="module")
def full():
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    images, targets = synthetic_batch(B, H, W, seed=7)
    return cfg_s, cfg_t, ms, mt, images, targets


FULL_CONVS = [
    # B, Cin, H, W, Cout, k, stride, pad
    (4, 4, 600, 1000, 64, 7, 2, 3),      # stem (3 channels padded to 4)
    (4, 64, 150, 250, 256, 1, 1, 0),     # layer1 expand: short K, store-bound
    (4, 128, 75, 125, 128, 3, 1, 1),     # layer2 3x3: Winograd F(4x4,3x3)
    (4, 256, 38, 63, 256, 3, 1, 1),      # layer3 3x3: Winograd, ragged tiles (38 and 63 are not multiples of 4)
    (4, 512, 75, 125, 1024, 1, 2, 0),    # layer3 stride-2 projection
    (4, 1024, 38, 63, 1024, 3, 1, 1),    # RPN 3x3
    (4, 1024, 38, 63, 75, 1, 1, 0),      # fused RPN heads (15 objectness + 60 deltas)
    (256, 512, 7, 7, 512, 3, 1, 1),      # layer4 3x3 on the 64-RoI distillation pass
    # layer4 on the step's 4 x 512 detection RoIs (M = 32768 rows: the 128x128 instance, split-M weight gradients with parked partials)
    (2048, 1024, 7, 7, 512, 1, 2, 0),    # conv1 (stride in the 1x1, resnet.py:278)
    (2048, 512, 4, 4, 512, 3, 1, 1),     # conv2: one F(4x4,3x3) tile per RoI
    (2048, 512, 4, 4, 2048, 1, 1, 0),    # conv3
    (2048, 1024, 7, 7, 2048, 1, 2, 0),   # stride-2 projection
]
# the reference's own INPUT defaults are 800 x 1333 (config/defaults.py:44-46; no configs/voc YAML overrides them): stem 400x667 -> layer1
# 200x334 -> layer2 100x167 -> layer3 / C4 50x84 (63 000 anchors).  None of these extents is a multiple of the 128-row GEMM tile or of the
# 4x4 Winograd tile in both directions: the same kernels, other tail cases.
SCALE_CONVS = [
    (2, 4, 800, 1333, 64, 7, 2, 3),      # stem
    (2, 64, 200, 334, 256, 1, 1, 0),     # layer1 expand (M = 133 600)
    (2, 128, 100, 167, 128, 3, 1, 1),    # layer2 3x3: Winograd, 167 = 41 tiles + 3 columns
    (2, 256, 200, 334, 512, 1, 2, 0),    # layer2 stride-2 projection
    (2, 256, 50, 84, 256, 3, 1, 1),      # layer3 3x3: Winograd, 50 = 12 tiles + 2 rows
    (2, 1024, 50, 84, 1024, 3, 1, 1),    # RPN 3x3
]
MATHS = ["bf16x6", "f16x3", "f32"]   # bf16x6 = the default arithmetic, the one bench.py reports; f32 = the fp32 MFMA kernels


def _math(name):
    from abr_iod_amd import ops
    return {"bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3, "f32": ops.MATH_F32}[name]


def _case_tensors(case):
    Bc, Cin, Hc, Wc, Cout, k, s, p = case
    g = torch.Generator(device="cuda").manual_seed(Cin * 7 + Cout)
    x = torch.randn(Bc, Hc, Wc, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    return x, w, g


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", FULL_CONVS + SCALE_CONVS)
def test_full_size_conv_linearity_and_sampled_dot_products(case, math):
    from abr_iod_amd import ops
    Bc, Cin, Hc, Wc, Cout, k, s, p = case
    m = _math(math)
    x, w, _ = _case_tensors(case)
    y = ops.conv_forward(x, w, s, p, math=m)
    Ho, Wo = (Hc + 2 * p - k) // s + 1, (Wc + 2 * p - k) // s + 1
    assert tuple(y.shape) == (Bc, Ho, Wo, Cout) and bool(torch.isfinite(y).all())
    # scaling by a power of two commutes with every fp32 rounding step, whatever the kernel's summation order (and with the exact
    # three-way bf16 split of the bf16x6 kernels)
    assert torch.equal(ops.conv_forward(x * 4.0, w, s, p, math=m), y * 4.0)
    assert torch.equal(ops.conv_forward(x, w * 0.5, s, p, math=m), y * 0.5)
    assert torch.equal(ops.conv_forward(x, w, s, p, math=m), y)   # and the kernel (split-K included) is run-to-run deterministic
    # float64 dot products at 256 sampled output positions, corners and edges included
    rng = np.random.default_rng(Cin + Cout)
    pos = [(0, 0, 0), (Bc - 1, Ho - 1, Wo - 1), (0, Ho - 1, 0), (Bc - 1, 0, Wo - 1)]
    pos += [(int(rng.integers(Bc)), int(rng.integers(Ho)), int(rng.integers(Wo))) for _ in range(252)]
    xp = torch.nn.functional.pad(x, (0, 0, p, p, p, p))
    w64 = w.double().reshape(Cout, -1)
    patches = torch.stack([xp[b, i * s:i * s + k, j * s:j * s + k, :].reshape(-1) for b, i, j in pos]).double()
    want = patches @ w64.t()
    got = torch.stack([y[b, i, j] for b, i, j in pos]).double()
    # 1e-4 of the output scale (north_star tolerance); the Winograd path itself stays below 5e-5
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    assert ops.x6_range_flags() == 0


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("case", FULL_CONVS[1:] + SCALE_CONVS[2:])   # the stem is frozen (FREEZE_CONV_BODY_AT = 2) and has no backward
def test_full_size_conv_backward_sampled_vs_float64(case, math):
    """dgrad and wgrad of the BASELINE layer shapes, as Bottleneck.bwd issues them (resnet.py here; the reference: cuDNN dgrad / wgrad
    behind resnet.py:261-323): the input gradient = the forward kernel on the flipped, FrozenBN-scaled weight copy (a scatter to the
    even pixels for the stride-2 1x1 convs), the weight gradient = the split-M kernel (parked partials + reduce) or, for the wide 3x3
    convs, the Winograd-domain gradient, also with the forward's kept V.  Checked against float64 sums at sampled positions."""
    from abr_iod_amd import ops
    if case[4] == 75:   # the model's fused RPN head weight is 76 rows (15 + 60 + 1 pad): its dgrad reduces over 76 channels
        case = case[:4] + (76,) + case[5:]
    Bc, Cin, Hc, Wc, Cout, k, s, p = case
    m = _math(math)
    x, w, g = _case_tensors(case)
    Ho, Wo = (Hc + 2 * p - k) // s + 1, (Wc + 2 * p - k) // s + 1
    gy = torch.randn(Bc, Ho, Wo, Cout, device="cuda", generator=g)
    scale = torch.rand(Cout, device="cuda", generator=g) + 0.5          # FrozenBN scale folded into both gradients
    rng = np.random.default_rng(Cin * 3 + Cout)
    # ---- dgrad: dx[b,h,w,:] = sum_{r,t,co} scale[co] gy[b,(h+p-r)/s,(w+p-t)/s,co] w[co,r,t,:]
    wt = ops.conv_dgrad_weights(w, scale)
    if s == 1:
        dx = ops.conv_forward(gy, wt, 1, k - 1 - p, math=m)
    else:
        assert k == 1 and p == 0
        dx = ops.conv_forward(gy, wt, 1, 0, out_hw=(Hc, Wc), out_stride=(s, s), math=m)
    assert tuple(dx.shape) == (Bc, Hc, Wc, Cin)
    assert torch.equal(dx, ops.conv_forward(gy, wt, 1, k - 1 - p, math=m) if s == 1 else
                       ops.conv_forward(gy, wt, 1, 0, out_hw=(Hc, Wc), out_stride=(s, s), math=m))
    pos = [(0, 0, 0), (Bc - 1, Hc - 1, Wc - 1), (0, Hc - 1, 0), (Bc - 1, 0, Wc - 1)]
    pos += [(int(rng.integers(Bc)), int(rng.integers(Hc)), int(rng.integers(Wc))) for _ in range(124)]
    ws64 = (w * scale.view(-1, 1, 1, 1)).double()
    want = torch.zeros(len(pos), Cin, dtype=torch.float64, device="cuda")
    for i, (b, h, ww) in enumerate(pos):
        for r in range(k):
            for t in range(k):
                hn, wn = h + p - r, ww + p - t
                if hn % s or wn % s or not (0 <= hn // s < Ho and 0 <= wn // s < Wo):
                    continue
                want[i] += gy[b, hn // s, wn // s].double() @ ws64[:, r, t, :]
    got = torch.stack([dx[b, h, ww] for b, h, ww in pos]).double()
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), "dgrad"
    # ---- wgrad: dw[co,r,t,:] += scale[co] sum_{b,ho,wo} gy[b,ho,wo,co] x[b,ho*s+r-p,wo*s+t-p,:]
    dw = torch.zeros_like(w)
    ops.conv_wgrad(x, gy, dw, s, p, scale=scale, math=m)
    dw2 = torch.zeros_like(w)
    ops.conv_wgrad(x, gy, dw2, s, p, scale=scale, math=m)
    assert torch.equal(dw, dw2)   # parked partials are added in a fixed order: deterministic
    v = ops.wino_v_alloc(x, w, s, p, m)
    if v is not None:             # the step's form: the forward keeps its Winograd-domain input for the weight gradient
        ops.conv_forward(x, w, s, p, math=m, wino_v=v)
        dw3 = torch.zeros_like(w)
        ops.conv_wgrad(x, gy, dw3, s, p, scale=scale, math=m, wino_v=v)
        assert torch.equal(dw3, dw)
    ops.conv_wgrad(x, gy, dw2, s, p, scale=scale, math=m)   # accumulates: a weight used twice per step (layer4) receives two launches
    xp = torch.nn.functional.pad(x, (0, 0, p, p, p, p))
    taps = [(0, 0, 0), (Cout - 1, k - 1, k - 1)] + [(int(rng.integers(Cout)), int(rng.integers(k)), int(rng.integers(k))) for _ in range(10)]
    worst = 0.0
    for co, r, t in taps:
        xs = xp[:, r:r + (Ho - 1) * s + 1:s, t:t + (Wo - 1) * s + 1:s, :].reshape(-1, Cin)
        gcol = gy[..., co].reshape(-1)
        want_v = torch.zeros(Cin, dtype=torch.float64, device="cuda")
        for lo in range(0, xs.shape[0], 1 << 16):   # float64 in slices (keeps the temporaries small)
            want_v += gcol[lo:lo + (1 << 16)].double() @ xs[lo:lo + (1 << 16)].double()
        want_v *= float(scale[co])
        got_v = dw[co, r, t].double()
        tol = 1e-4 * max(1.0, want_v.abs().max().item())
        worst = max(worst, (got_v - want_v).abs().max().item() / tol)
        assert (got_v - want_v).abs().max().item() < tol, ("wgrad", co, r, t)
        assert (dw2[co, r, t].double() - 2 * want_v).abs().max().item() < 2 * tol, ("wgrad accumulate", co, r, t)
    assert ops.x6_range_flags() == 0


def _iou(a, b):
    # boxlist_ops.py:53-76 convention (+1 widths)
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    lt = torch.max(a[:, None, :2], b[None, :, :2]); rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter)


def test_full_size_rpn_selection_invariants(full):
    from abr_iod_amd.structures.image_list import to_image_list
    cfg_s, cfg_t, ms, mt, images, targets = full
    il = to_image_list(images)
    with torch.no_grad():
        feats, _ = ms.backbone(il.tensors)
        (boxes, _), anchors, (obj, reg) = ms.rpn(il, feats, None)
    A = 15
    assert tuple(obj[0].shape) == (B, A, 38, 63) and tuple(reg[0].shape) == (B, 4 * A, 38, 63)
    for bl in boxes:   # inference.py:74-110: top-6000 -> clip -> remove_small(0) -> NMS 0.7 -> top-300 (TEST settings)
        n = len(bl)
        assert 0 < n <= cfg_s.MODEL.RPN.POST_NMS_TOP_N_TEST
        bb, sc = bl.bbox.double(), bl.get_field("objectness")
        assert bool((sc[:-1] >= sc[1:]).all()) and bool(((sc >= 0) & (sc <= 1)).all())
        assert bool((bb[:, 0] >= 0).all()) and bool((bb[:, 1] >= 0).all())
        assert bool((bb[:, 2] <= W - 1).all()) and bool((bb[:, 3] <= H - 1).all())
        assert bool((bb[:, 2] >= bb[:, 0]).all()) and bool((bb[:, 3] >= bb[:, 1]).all())
        iou = _iou(bb, bb); iou.fill_diagonal_(0)
        assert iou.max().item() <= cfg_s.MODEL.RPN.NMS_THRESH + 1e-9
    mt.train()
    with torch.no_grad():
        feats, _ = mt.backbone(il.tensors)
        (props, losses), _, _ = mt.rpn(il, feats, targets)
    assert set(losses) == {"loss_objectness", "loss_rpn_box_reg"}
    for bl, tg in zip(props, targets):   # training: top-12000 -> NMS -> top-2000, then the GT boxes appended (inference.py:51-72)
        n_gt = len(tg)
        assert n_gt < len(bl) <= cfg_t.MODEL.RPN.POST_NMS_TOP_N_TRAIN + n_gt
        assert torch.equal(bl.bbox[-n_gt:], tg.bbox)
        kept = bl.bbox[:-n_gt].double()
        iou = _iou(kept, kept); iou.fill_diagonal_(0)
        assert iou.max().item() <= cfg_t.MODEL.RPN.NMS_THRESH + 1e-9


def test_full_size_step0_properties(full):
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine.synthetic import build_models
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    cfg_s, cfg_t, _, _, images, targets = full
    runs = []
    import random
    from abr_iod_amd import ops
    for _ in range(2):
        ms, mt = build_models(cfg_s, cfg_t, seed=0)   # torch.manual_seed(0): the samplers derive their draws from it and a call counter
        ops._sample_calls[0] = 0
        random.seed(0)                                # the 64-of-128 soften picks use python `random` as the reference does
        opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
        before = {n: p.detach().clone() for n, p in mt.named_parameters()}
        ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        runs.append(({k: float(v.detach()) for k, v in ld.items()}, float(total.detach()), before, mt))
    ld, total, before, mt = runs[0]
    assert set(ld) == {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg", "distillation_loss"}
    assert all(np.isfinite(v) for v in ld.values())
    # inclusive distillation folds the (still random) new-class logits into the background term (distillation.py:225-229), so it
    # is positive from the start; see test_full_size_ard_term_vanishes_at_step0 for the term that must start at zero
    assert 0.0 < ld["distillation_loss"] < 10.0
    assert abs(total - sum(ld.values())) <= 1e-5 * abs(total)
    # a fresh build from the same seed repeats the forward: the index work (proposal selection, both samplers) bit for bit, the
    # scalar losses to the last ulp or two (their final reduction is one float atomicAdd per workgroup, as in the reference's CUDA)
    for k, v in ld.items():
        assert abs(runs[1][0][k] - v) <= 1e-6 * max(abs(v), 1e-3), k
    ev0, ev1 = runs[0][3].roi_heads.box.loss_evaluator, runs[1][3].roi_heads.box.loss_evaluator
    for a, b in zip(ev0.last_sampled_inds, ev1.last_sampled_inds):
        assert torch.equal(a, b)
    for a, b in zip(ev0._proposals, ev1._proposals):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("labels"), b.get_field("labels"))
    # sampler quotas on the last pass (balanced_positive_negative_sampler.py:20-68)
    ev = mt.roi_heads.box.loss_evaluator
    per_img, pos_frac = cfg_t.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, cfg_t.MODEL.ROI_HEADS.POSITIVE_FRACTION
    for inds, props in zip(ev.last_sampled_inds, ev._proposals):
        assert inds.numel() == per_img == len(props)            # 2000+ proposals: the quota is always filled
        assert bool((inds[1:] > inds[:-1]).all())               # distinct, ascending (loss.py:114)
        lab = props.get_field("labels")
        assert 0 < int((lab > 0).sum()) <= int(per_img * pos_frac)   # the appended GT boxes guarantee a positive
        assert bool(((lab == 0) | ((lab >= 16) & (lab <= 20))).all())   # task 15-5: only the new classes are annotated
    # FREEZE_CONV_BODY_AT=2: stem + layer1 never move; something trainable in every other group does
    moved = {n: not torch.equal(before[n], p.detach()) for n, p in mt.named_parameters()}
    for n, m in moved.items():
        if ".stem." in n or ".layer1." in n:
            assert not m, n
    for group in ("layer2", "layer3", "rpn.head", "roi_heads.box.feature_extractor", "roi_heads.box.predictor"):
        assert any(m for n, m in moved.items() if group in n), group


def test_full_size_ard_term_vanishes_at_step0(full):
    """The target is initialised from the source (model_serialization.py:47-55): identical pooled features on the 64 soften
    proposals, so the attentive RoI feature term (distillation.py:291-333) is exactly zero before the first update and positive
    after it.  (Both RoI-output terms normalise the target's logits over ALL its classes, the random new rows included, and
    start above zero.)"""
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    _, _, _, _, images, targets = full
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="l2", feat="ard", alpha=0.0, beta=1.0)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    ld, _ = train_step(ms, mt, images, targets, opt, sch, cfg_t)
    assert float(ld["distillation_loss"]) == 0.0
    ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
    d = float(ld["distillation_loss"])
    assert 0.0 < d < 1.0 and np.isfinite(float(total.detach()))
    for p in ms.parameters():   # the source model is frozen and runs under no_grad (train_incremental.py:80-86)
        assert p.grad is None or not bool(p.grad.any())


def test_reference_scale_800x1333_step():
    """The reference trains at INPUT.MIN_SIZE_TRAIN 800 / MAX_SIZE_TRAIN 1333 unless a YAML says otherwise, and no configs/voc YAML does
    (config/defaults.py:44-46): one ARD + ID step of the full-width model at that geometry -- C4 map 50x84, 63 000 anchors per image,
    12000 -> 2000 proposals, 512 + 64 RoIs per image.  Checks the geometry, the selection invariants, finite losses that equal a second
    run bit for bit in their index work, and that the update moves what it should."""
    import random
    from abr_iod_amd import ops
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    from abr_iod_amd.structures.image_list import to_image_list
    Hs, Ws, Bs = 800, 1333, 2
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0)
    images, targets = synthetic_batch(Bs, Hs, Ws, seed=11)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    il = to_image_list(images)
    with torch.no_grad():
        feats, _ = mt.backbone(il.tensors)
        assert tuple(feats[0].shape) == (Bs, 1024, 50, 84)
        (props, _), anchors, (obj, reg) = mt.rpn(il, feats, targets)
    assert anchors[0][0].bbox.shape[0] == 50 * 84 * 15 == 63000
    assert tuple(obj[0].shape) == (Bs, 15, 50, 84) and tuple(reg[0].shape) == (Bs, 60, 50, 84)
    for bl, tg in zip(props, targets):
        n_gt = len(tg)
        assert n_gt < len(bl) <= cfg_t.MODEL.RPN.POST_NMS_TOP_N_TRAIN + n_gt
        bb = bl.bbox[:-n_gt].double()
        assert bool((bb[:, 0] >= 0).all()) and bool((bb[:, 1] >= 0).all()) and bool((bb[:, 2] <= Ws - 1).all()) and bool((bb[:, 3] <= Hs - 1).all())
        iou = _iou(bb, bb); iou.fill_diagonal_(0)
        assert iou.max().item() <= cfg_t.MODEL.RPN.NMS_THRESH + 1e-9
    runs = []
    for _ in range(2):
        ms, mt = build_models(cfg_s, cfg_t, seed=0)
        ops._sample_calls[0] = 0
        random.seed(0)
        opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
        before = mt.flat.params.clone()
        ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        runs.append(({k: float(v.detach()) for k, v in ld.items()}, float(total.detach()), before, mt))
    ld, total, before, mt = runs[0]
    assert set(ld) == {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg", "distillation_loss"}
    assert all(np.isfinite(v) for v in ld.values()) and np.isfinite(total)
    for k, v in ld.items():
        assert abs(runs[1][0][k] - v) <= 1e-6 * max(abs(v), 1e-3), k
    ev = mt.roi_heads.box.loss_evaluator
    for inds, pr in zip(ev.last_sampled_inds, ev._proposals):
        assert inds.numel() == cfg_t.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE == len(pr)
    assert torch.isfinite(mt.flat.params).all() and not torch.equal(mt.flat.params, before)
    assert ops.x6_range_flags() == 0
