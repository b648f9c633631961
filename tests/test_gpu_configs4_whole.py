"""GPU, FULL SIZE: BASELINE.json configs[4] as ONE workload -- task 10-5 (K_old 11, K_all 16; scripts/run_MI.sh:11-21), batches fed by the
box-rehearsal data path (mixup + mosaic from a rehearsal memory, voc_abr.py:555-838 -> abr_iod_amd/data/abr.py on the device), hence RAGGED
(a mosaic canvas is mean(w,h)^2 -> 600x600 after the resize, voc_abr.py:712-714, next to 600x800 / 600x1000 images, zero-padded by
to_image_list as image_list.py:50-68 does), through the full-width R50-C4 at the benchmark's RoI counts, in

  * the default arithmetic (fp32 tensors, f16x3 contractions since round 5): C4 features against the torch-CPU oracle on the same padded batch at
    the north-star tolerance, a whole training step with finite losses and a moving update, and
  * cfg.DTYPE = "bfloat16" (the "bf16 MFMA backbone" the config names): every backbone stage, teacher-forced on the oracle's own stage
    input, against oracle/torch_ref.py::bottleneck(bf16=True), and a whole step whose losses track the default arithmetic's.

The pieces have their own pins elsewhere (tests/test_gpu_data.py: pixels / boxes of mixup + mosaic equal the reference's; tests/
test_gpu_e2e_full_golden.py: the 10-5 step at 600x1000 equals the reference; tests/test_gpu_configs.py: the bf16 mode at small size)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rehearsal_memory(tmp_path, n=24, seed=5):
    """a box-rehearsal memory as tools/prototype_box_selection.py writes it: '<class>_<idx:05d>.jpg' crops (PNG content: lossless)"""
    from PIL import Image
    rs = np.random.RandomState(seed)
    names = []
    for i in range(n):
        h, w = int(rs.randint(40, 220)), int(rs.randint(40, 260))
        name = "{}_{:05d}.jpg".format(1 + i % 10, i)          # old classes 1..10 of task 10-5
        Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(str(tmp_path), name), format="PNG")
        names.append(name)
    return names


@pytest.fixture(scope="module")
def abr_batch(tmp_path_factory):
    """three VOC-sized images through mixup / mosaic / untouched (forced, one of each), the train transform and the collate"""
    import types
    from abr_iod_amd.data.abr import BoxRehearsalABR, GPUTransform
    from abr_iod_amd.data.gpu_transforms import to_device_u8
    from abr_iod_amd.structures.bounding_box import BoxList
    tmp = tmp_path_factory.mktemp("mem")
    names = _rehearsal_memory(tmp)
    abr = BoxRehearsalABR(str(tmp), names, batch_size=3, shuffle=False)
    cfg_in = types.SimpleNamespace(INPUT=types.SimpleNamespace(MIN_SIZE_TRAIN=(600,), MAX_SIZE_TRAIN=1000, MIN_SIZE_TEST=600, MAX_SIZE_TEST=1000,
                                                                FLIP_PROB_TRAIN=0.5, PIXEL_MEAN=[102.9801, 115.9465, 122.7717],
                                                                PIXEL_STD=[1.0, 1.0, 1.0], TO_BGR255=True, BRIGHTNESS=0.0, CONTRAST=0.0,
                                                                SATURATION=0.0, HUE=0.0))
    tf = GPUTransform(cfg_in, is_train=True)
    rs = np.random.RandomState(9)
    random.seed(3); torch.manual_seed(3)
    samples, kinds = [], []
    for kind, (H, W) in (("mixup", (375, 500)), ("mosaic", (375, 500)), ("new", (300, 500))):
        img = to_device_u8(rs.randint(0, 256, (H, W, 3), dtype=np.uint8))
        t = BoxList(torch.tensor([[30.0, 40.0, 130.0, 160.0], [250.0, 100.0, 420.0, 290.0]]), (W, H), mode="xyxy")
        t.add_field("labels", torch.tensor([12, 15]))          # new classes 11..15 of task 10-5
        if kind == "mixup":
            img, t = abr._start_mixup(img, t)
        elif kind == "mosaic":
            img, t = abr._start_boxes_mosaic((W, H))
        samples.append(tf(img, t))
        kinds.append(kind)
    images, targets = tf.collate(samples)
    targets = [t.to("cuda") for t in targets]
    return images, targets, kinds


def test_abr_batch_is_ragged_at_full_size(abr_batch):
    images, targets, kinds = abr_batch
    sizes = [tuple(s) for s in images.image_sizes]
    assert sizes[kinds.index("mixup")] == (600, 800)            # 375x500 -> min side 600
    assert sizes[kinds.index("mosaic")] == (600, 600)           # canvas mean(500, 375)^2 = 437^2 -> 600x600 (voc_abr.py:712-714)
    assert sizes[kinds.index("new")] == (600, 1000)             # 300x500 -> 600x1000
    assert tuple(images.tensors.shape) == (3, 3, 600, 1000)     # zero-padded to the per-batch maximum (SIZE_DIVISIBILITY 0)
    i = kinds.index("mosaic")
    assert not bool(images.tensors[i, :, :, 600:].any())
    lab = targets[i].get_field("labels")
    assert len(targets[i]) >= 1 and bool(((lab >= 1) & (lab <= 10)).all())      # a mosaic holds replayed OLD-class boxes only
    for t, (h, w) in zip(targets, sizes):
        assert tuple(t.size) == (w, h)
        assert bool((t.bbox[:, 2] <= w + 1e-3).all()) and bool((t.bbox[:, 3] <= h + 1e-3).all())   # (a mosaic box may end ON the canvas edge, as the reference's)


def _models(dtype):
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    cfg_s, cfg_t = make_cfgs("10-5", dist_type="id", feat="ard", alpha=1.0, beta=1.0, gamma=1.0, overrides=["DTYPE", dtype])
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    return cfg_s, cfg_t, ms, mt


def _step(cfg_t, ms, mt, images, targets):
    from abr_iod_amd import ops
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    ops._sample_calls[0] = 0
    random.seed(0)
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    before = mt.flat.params.clone()
    ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    return {k: float(v.detach()) for k, v in ld.items()}, float(total.detach()), (mt.flat.params - before)


def test_configs4_default_arithmetic_features_and_step(abr_batch):
    from abr_iod_amd import ops
    from abr_iod_amd.utils.checkpoint import reference_state_dict
    from oracle.model_ref import RefModel
    images, targets, kinds = abr_batch
    cfg_s, cfg_t, ms, mt = _models("float32")
    assert ms.roi_heads.box.predictor.num_classes == 11 and mt.roi_heads.box.predictor.num_classes == 16
    assert all(m.math == ops.MATH_F16X3 for m in mt.modules() if hasattr(m, "math"))
    with torch.no_grad():
        feats, _ = mt.backbone(images.tensors)
    assert tuple(feats[0].shape) == (3, 1024, 38, 63)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 8)))
    with torch.no_grad():
        want = RefModel(reference_state_dict(mt), trainable_prefixes=()).backbone(images.tensors.cpu())
    got = feats[0].cpu()
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()          # incl. the zero-padded regions of the short images
    ld, total, delta = _step(cfg_t, ms, mt, images, targets)
    assert set(ld) == {"loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg", "distillation_loss"}
    assert all(np.isfinite(v) for v in ld.values()) and np.isfinite(total) and float(delta.norm()) > 0
    n = 38 * 63 * 15
    for a in mt.rpn.loss_evaluator.last_targets[0]:
        assert a.numel() == n
    # anchors of a short image that straddle ITS border (not the padded batch's) are ignored: visibility is per image (anchor_generator.py:97-110)
    i = kinds.index("mosaic")
    lab = mt.rpn.loss_evaluator.last_targets[0][i].view(38, 63, 15)
    assert bool((lab[:, 38:, :] == -1).all())          # x >= 608 > 599: outside the 600x600 image
    # a zero-padded image's input gradients fall to ~1e-36 in places (tools/x6_flag_hunt.py): the guard may REPORT tiny operands (their
    # cost is an absolute error below 2^-119 of the other operand, engine/trainer.py::_x6_guard); it must never see inf / nan
    assert not (ops.x6_range_flags() & ops.X6_FLAG_NONFINITE)


def test_configs4_bf16_backbone_stages_and_step(abr_batch):
    """cfg.DTYPE = bfloat16 at full size on the ragged ABR batch: each stage of the backbone, fed the ORACLE's input to that stage, against
    oracle/torch_ref.py::bottleneck(bf16=True) stacked over the stage (operands rounded to bf16, fp32 accumulate); then a whole step."""
    from abr_iod_amd import ops
    from abr_iod_amd.modeling.backbone.resnet import run_stage
    from abr_iod_amd.utils.checkpoint import reference_state_dict
    from oracle.model_ref import RefModel
    images, targets, kinds = abr_batch
    cfg_s, cfg_t, ms, mt = _models("bfloat16")
    assert all(m.math == ops.MATH_BF16 for m in mt.backbone.modules() if hasattr(m, "math"))
    ref = RefModel(reference_state_dict(mt), trainable_prefixes=(), bf16_backbone=True)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 8)))
    body = mt.backbone.body
    with torch.no_grad():
        x_ref = body.stem(images.tensors).cpu()          # the stem is exact fp32 in both (Cin = 3): start both from the same tensor
        for name, n_blocks in (("layer1", 3), ("layer2", 4), ("layer3", 6)):
            got = run_stage(x_ref.cuda(), list(getattr(body, name))).cpu()
            want = x_ref
            for i in range(n_blocks):
                want = ref._block(want, "backbone.body.{}.{}".format(name, i), (2 if name != "layer1" else 1) if i == 0 else 1, bf16=True)
            rel = ((got - want).norm() / want.norm()).item()
            print("bf16 {}: rel. L2 distance to the oracle's stage on the same input: {:.2e}".format(name, rel))
            # one block agrees to summation order (6e-6); over a stage, activations within that distance of a bf16 rounding boundary flip
            # and feed the next blocks: bounded well below one bf16 ulp (3.9e-3)
            assert rel < 1.5e-3, (name, rel)
            x_ref = want
    ld16, total16, d16 = _step(cfg_t, ms, mt, images, targets)
    assert all(np.isfinite(v) for v in ld16.values()) and np.isfinite(total16) and float(d16.norm()) > 0
    cfg_s2, cfg_t2, ms2, mt2 = _models("float32")
    ld32, total32, d32 = _step(cfg_t2, ms2, mt2, images, targets)
    print("bf16 backbone step:", ld16)
    print("default arithmetic:", ld32)
    for k in ld32:   # same seeds; bf16-rounded backbone operands shift the proposal scores, hence which 2000 boxes survive NMS and which 512 are
        # sampled: the classification / objectness losses move by a few per cent, the box-regression losses (a handful of positives) by more
        tol = 0.25 if "box" in k else 0.08
        assert abs(ld16[k] - ld32[k]) <= tol * max(abs(ld32[k]), 0.02), (k, ld32[k], ld16[k])
    cos = float((d32 * d16).sum() / (d32.norm() * d16.norm()))
    assert cos > 0.95, cos
