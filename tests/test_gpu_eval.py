"""GPU: test-time detection path through the C-ABI: PostProcessor = csrc/detect.hip + the training NMS.
Checked against (a) the REFERENCE's PostProcessor outputs (tests/golden/post_processor.npz) and (b) the oracle's
filter_results restatement on identical inputs, where the index work must be bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SETTINGS = {"std": (0.05, 0.5, 100), "tight": (0.2, 0.3, 7), "all": (0.05, 0.5, 0)}


def _boxlists(g):
    from abr_iod_amd.structures.bounding_box import BoxList
    return [BoxList(torch.from_numpy(g[f"boxes{i}"]).cuda(), tuple(int(v) for v in g["sizes_wh"][i]), mode="xyxy") for i in range(2)]


def test_post_processor_matches_reference(gold):
    from abr_iod_amd.modeling.box_coder import BoxCoder
    from abr_iod_amd.modeling.roi_heads.box_head.inference import PostProcessor
    g = gold("post_processor")
    logits, reg = torch.from_numpy(g["logits"]).cuda(), torch.from_numpy(g["box_regression"]).cuda()
    for tag, (thr, nms_t, det) in SETTINGS.items():
        pp = PostProcessor(thr, nms_t, det, BoxCoder(weights=(10., 10., 5., 5.)), False)
        res, bg = pp((logits, reg), _boxlists(g))
        for i, r in enumerate(res):
            assert r.size == tuple(int(v) for v in g["sizes_wh"][i]) and r.mode == "xyxy"
            assert np.array_equal(r.get_field("labels").cpu().numpy(), g[f"{tag}_labels{i}"]), (tag, i)
            np.testing.assert_allclose(r.get_field("scores").cpu().numpy(), g[f"{tag}_scores{i}"], rtol=1e-5)
            np.testing.assert_allclose(r.bbox.cpu().numpy(), g[f"{tag}_boxes{i}"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(bg.get_field("scores").cpu().numpy(), g[f"{tag}_bg_scores"], rtol=1e-5)
        np.testing.assert_allclose(bg.bbox.cpu().numpy(), g[f"{tag}_bg_boxes"], rtol=1e-5, atol=2e-4)
    # [K,C,4] regression (the training-path shape) is accepted too; a wrong row count is an error like the reference's view()
    pp = PostProcessor()
    res2, _ = pp((logits, reg.view(-1, 21, 4)), _boxlists(g))
    assert len(res2[0]) == len(g["std_labels0"])
    with pytest.raises(ValueError):
        pp((logits[:-1], reg[:-1]), _boxlists(g))


def test_softmax_decode_vs_oracle():
    from abr_iod_amd import ops
    from oracle import torch_ref as R
    g = torch.Generator().manual_seed(0)
    C, counts, sizes = 21, [300, 0, 517], [(1000, 600), (800, 600), (640, 480)]
    K = sum(counts)
    props = []
    for (w, h), n in zip(sizes, counts):
        x1, y1 = torch.rand(n, generator=g) * (w - 40), torch.rand(n, generator=g) * (h - 40)
        props.append(torch.stack([x1, y1, x1 + 8 + torch.rand(n, generator=g) * 300, y1 + 8 + torch.rand(n, generator=g) * 300], 1).numpy())
    logits = torch.randn(K, C, generator=g) * 3
    reg = torch.randn(K, 4 * C, generator=g) * 0.7
    reg[::7, 2::4] = 60.0  # dw far above the log(1000/16) clamp
    prob_o, box_o = R.det_softmax_decode(logits, reg, props, sizes)
    rois = torch.cat([torch.cat([torch.full((len(p), 1), float(i)), torch.from_numpy(p)], 1) for i, p in enumerate(props)], 0).cuda()
    hw = torch.tensor([[h, w] for w, h in sizes], dtype=torch.int32).cuda()
    # read logits / deltas as column slices of one fused [K,108] matrix, as the predictor hands them over
    fused = torch.cat([logits, reg, torch.zeros(K, 3)], 1).cuda()
    prob, box = ops.det_softmax_decode(fused[:, :C], fused[:, C:C + 4 * C], rois, C, hw, (10., 10., 5., 5.))
    np.testing.assert_allclose(prob.cpu().numpy(), prob_o, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(box.cpu().numpy(), box_o, rtol=1e-5, atol=2e-4)
    prob_a, box_a = R.det_softmax_decode(logits, reg, props, sizes, cls_agnostic=True)
    _, box2 = ops.det_softmax_decode(fused[:, :C], fused[:, C:C + 4 * C], rois, C, hw, (10., 10., 5., 5.), cls_agnostic=True)
    np.testing.assert_allclose(box2.cpu().numpy(), box_a, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("quantize", [False, True])
def test_det_select_bit_exact_vs_oracle(quantize):
    """Identical prob/boxes into the HIP selection and the oracle's filter_results: labels, scores, boxes and their ORDER are
    equal bit for bit -- ragged batch incl. an image without proposals, ties in the scores (quantised), ties at the top-D cut."""
    from abr_iod_amd import ops
    from oracle import torch_ref as R
    g = torch.Generator().manual_seed(5)
    C, counts, D = 21, [1000, 0, 333, 64], 100
    K, N = sum(counts), 4
    prob = torch.softmax(torch.randn(K, C, generator=g) * 2.5, -1)
    if quantize:
        prob = (prob * 50).round() / 50  # many equal scores -> stable order + `>= kth` ties
    cx, cy = torch.rand(K, C, generator=g) * 900, torch.rand(K, C, generator=g) * 500
    bw, bh = 20 + torch.rand(K, C, generator=g) * 250, 20 + torch.rand(K, C, generator=g) * 250
    boxes = torch.stack([cx, cy, (cx + bw).clamp(max=999), (cy + bh).clamp(max=599)], -1)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    out = ops.det_select(prob.cuda(), boxes.cuda(), torch.from_numpy(off).cuda(), N, C, max(counts), 0.05, 0.5, D)
    ob, os_, ol, oc, bb, bs, bc = [t.cpu().numpy() for t in out]
    for i in range(N):
        p, b = prob[off[i]:off[i + 1]].numpy(), boxes[off[i]:off[i + 1]].numpy()
        if counts[i] == 0:
            assert oc[i] == 0 and bc[i] == 0
            continue
        (rb, rs, rl), (gb, gs) = R.det_filter_results(p, b, 0.05, 0.5, D)
        assert oc[i] == len(rs), (i, oc[i], len(rs))
        assert np.array_equal(ol[i, :oc[i]], rl) and np.array_equal(os_[i, :oc[i]], rs) and np.array_equal(ob[i, :oc[i]], rb)
        assert bc[i] == len(gs) and np.array_equal(bs[i, :bc[i]], gs) and np.array_equal(bb[i, :bc[i]], gb)
        if quantize:
            assert oc[i] >= D  # ties at the cut are all kept (inference.py:147 `cls_scores >= image_thresh`)


def test_eval_forward_end_to_end():
    """model.eval()(images) -> (detections, features, background) (generalized_rcnn.py:76-78): equals the oracle's PostProcessor
    applied to the model's own logits for the RPN's test-time proposals."""
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from oracle import torch_ref as R
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300, "MODEL.RPN.POST_NMS_TOP_N_TEST", 150]
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, overrides=tiny)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    images, targets = synthetic_batch(2, 160, 224, seed=1)
    mt.eval()
    with torch.no_grad():
        result, features, bg = mt(images)
        from abr_iod_amd.structures.image_list import to_image_list
        (props, _), _, _ = mt.rpn(to_image_list(images), features, None)
        logits, reg, _, _ = mt.roi_heads.box.calculate_soften_label(features, props)
    assert len(result) == 2 and all(len(r) <= 100 for r in result)
    ref, ref_bg = R.post_process(logits.cpu(), reg.reshape(len(logits), -1).cpu(), [p.bbox.cpu().numpy() for p in props],
                                 [p.size for p in props])
    for r, (b, s, l) in zip(result, ref):
        assert np.array_equal(r.get_field("labels").cpu().numpy(), l)
        np.testing.assert_allclose(r.get_field("scores").cpu().numpy(), s, rtol=1e-5)
        np.testing.assert_allclose(r.bbox.cpu().numpy(), b, rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(bg.get_field("scores").cpu().numpy(), ref_bg[1], rtol=1e-5)


def test_inference_loop_to_map(tmp_path):
    """engine.inference.inference(): eval forward over a loader -> per-image BoxLists on the host -> VOC mAP dict."""
    from abr_iod_amd.engine.inference import inference
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300, "MODEL.RPN.POST_NMS_TOP_N_TEST", 150]
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, overrides=tiny)
    _, mt = build_models(cfg_s, cfg_t, seed=0, need_source=False)

    class DS(object):
        def __init__(self):
            self.items = []
            for b in range(3):
                images, targets = synthetic_batch(2, 160, 224, seed=10 + b, label_range=(1, 21))
                for t in targets:
                    t = t.to("cpu")
                    t.add_field("difficult", torch.zeros(len(t), dtype=torch.uint8))
                    self.items.append(t)
                self.batches = getattr(self, "batches", []) + [(images, targets, (2 * b, 2 * b + 1))]

        def __len__(self):
            return len(self.items)

        def get_img_info(self, i):
            return {"width": 448, "height": 320}  # "original" images are 2x the network input

        def get_groundtruth(self, i):
            return self.items[i].resize((448, 320))

        def map_class_id_to_class_name(self, i):
            return str(i)

    ds = DS()

    class Loader(object):
        dataset = ds

        def __iter__(self):
            return iter(ds.batches)

    r = inference(mt, Loader(), "synthetic_voc", output_folder=str(tmp_path), save_predictions=True)
    assert set(r.keys()) == {"ap", "map"} and len(r["ap"]) <= 21
    preds = torch.load(str(tmp_path / "predictions.pth"), weights_only=False)
    assert len(preds) == 6 and all(p.bbox.device.type == "cpu" and p.has_field("scores") and p.has_field("labels") for p in preds)
    assert (tmp_path / "result.txt").exists()


def _tiny_eval_model():
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300, "MODEL.RPN.POST_NMS_TOP_N_TEST", 150]
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, overrides=tiny)
    _, mt = build_models(cfg_s, cfg_t, seed=0, need_source=False)
    mt.eval()
    return mt


def test_eval_range_guard_in_domain_batch_stays_on_f16x3():
    """The test loop polls the split arithmetic's range words per batch (engine/inference.py::EvalRangeGuard): an ordinary batch is inside
    f16x3's domain -- no flag, a small-element share far below the limit, one forward per batch, same detections as the bare model call."""
    from abr_iod_amd.engine.inference import EvalRangeGuard
    from abr_iod_amd.engine.synthetic import synthetic_batch
    mt = _tiny_eval_model()
    if mt.conv_math != "f16x3":
        pytest.skip("default arithmetic is not f16x3 in this environment")
    images, _ = synthetic_batch(2, 160, 224, seed=3)
    g = EvalRangeGuard(mt)
    with torch.no_grad():
        want, _, _ = mt(images)
        got, _, _ = g.forward(images)
    assert mt.conv_math == "f16x3" and g.stats["reruns"] == 0 and g.stats["batches"] == 1 and g.stats["flags"] == 0
    assert g.stats["seen"] > 0 and g.stats["max_small_fraction"] < 0.05
    for a, b in zip(want, got):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores"))


def test_eval_range_guard_reruns_out_of_domain_batches():
    """(a) a batch whose small-element share exceeds the limit (forced here by a limit below zero: the share of a seeded Gaussian-like batch is
    data, not a constant) is re-run in bf16x6 (exact split, no amax domain) and its detections equal a bf16x6 model's; (b) an inf pixel -> the
    fp32 MFMA kernels.  A warning each time."""
    from abr_iod_amd.engine.inference import EvalRangeGuard
    from abr_iod_amd.engine.synthetic import synthetic_batch
    mt = _tiny_eval_model()
    if mt.conv_math != "f16x3":
        pytest.skip("default arithmetic is not f16x3 in this environment")
    images, _ = synthetic_batch(2, 160, 224, seed=4)
    g = EvalRangeGuard(mt)
    g.limit = -1.0
    with torch.no_grad():
        got, _, _ = g.forward(images)
    assert g.stats["reruns"] == 1 and g.stats["batches"] == 2 and mt.conv_math == "bf16x6", g.stats
    ref = _tiny_eval_model()
    ref.set_conv_math("bf16x6")
    with torch.no_grad():
        want, _, _ = ref(images)
    for a, b in zip(want, got):
        assert torch.equal(a.bbox, b.bbox)
    bad = images.clone()
    bad[0, 0, 5, 5] = float("inf")
    with torch.no_grad():
        g.forward(bad)
    assert mt.conv_math == "f32" and g.stats["flags"] & 2
    # from here on the guard has nothing left to watch
    assert not g.active()
