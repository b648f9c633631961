"""GPU: every stream-level overlap of the training step -- weight gradients on a side stream, proposal selections on side streams, the
source model's forward and head pass on a stream of their own (for the NEXT batch pipelined into the current backward pass, together with
the target's frozen stem + layer1), the target's distillation-RoI pass issued ahead of its detection pass, two weight-gradient streams, weight preparation (dgrad copies, Winograd-domain weights) on its own
stream after the SGD kernel -- changes WHEN kernels run, never what they compute: four training steps with all of them on must
leave the same parameters as four steps with all of them off (same seeds, so the same sampler draws and soften picks; fp32 atomic
accumulation order is the only difference).  A missing stream dependency (a kernel reading a buffer another stream has not finished
writing, a cached derived weight of the wrong version) shows up here as a parameter mismatch."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

TINY = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
        "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128, "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 600, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 100,
        "MODEL.RPN.PRE_NMS_TOP_N_TEST", 2000, "MODEL.RPN.POST_NMS_TOP_N_TEST", 400, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32,
        "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64]


SWITCHES = ("wgrad_stream", "proposal_stream", "source_overlap", "source_stream", "source_head_stream", "early_second_pass",
            "pipeline_target_frozen", "pipeline_source", "early_prefetch", "prep_stream")


def _run(overlap, width_overrides, steps=4, off=(), joint=False):
    """overlap: every stream / prefetch switch on (True) or off (False); `off`: names of SWITCHES forced off in an otherwise-on run"""
    from abr_iod_amd import ops
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.modeling.rpn import rpn
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    where = {"wgrad_stream": (ops, "WGRAD_SIDE_STREAM"), "proposal_stream": (rpn, "PROPOSALS_SIDE_STREAM"), "source_overlap": (trainer, "SOURCE_OVERLAP"),
             "source_stream": (trainer, "SOURCE_STREAM"), "source_head_stream": (trainer, "SOURCE_HEAD_STREAM"),
             "early_second_pass": (trainer, "EARLY_SECOND_PASS"), "pipeline_target_frozen": (trainer, "PIPELINE_TARGET_FROZEN"),
             "pipeline_source": (trainer, "PIPELINE_SOURCE"), "early_prefetch": (trainer, "EARLY_PREFETCH")}
    where["joint_roi"] = (trainer, "JOINT_ROI_PASS")
    saved = {k: getattr(m, a) for k, (m, a) in where.items()}
    val = {k: bool(overlap) and k not in off for k in SWITCHES}
    val["joint_roi"] = bool(joint)
    for k, (m, a) in where.items():
        setattr(m, a, val[k])
    try:
        cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, overrides=width_overrides)
        ms, mt = build_models(cfg_s, cfg_t, seed=0)
        opt = make_optimizer(cfg_t, mt)
        opt._prep_stream = val["prep_stream"]
        sch = make_lr_scheduler(cfg_t, opt)
        batches = [synthetic_batch(2, 192, 256, seed=30 + i, max_boxes=2) for i in range(2)]
        torch.manual_seed(9); random.seed(9)
        losses = []
        for it in range(steps):
            images, targets = batches[it % 2]
            # overlap mode also names the NEXT batch: the source model's forward for it is pipelined into this step's backward pass
            nxt = batches[(it + 1) % 2][0] if (overlap and it + 1 < steps) else None
            ld, total = trainer.train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=nxt)
            losses.append(float(total.detach()))
        torch.cuda.synchronize()
        return mt.flat.params.detach().clone(), losses
    finally:
        for k, (m, a) in where.items():
            setattr(m, a, saved[k])


@pytest.mark.parametrize("width", ["tiny", "full"])
def test_overlapped_step_equals_serial_step(width):
    ov = TINY if width == "tiny" else TINY[8:]       # "full": the real channel widths (Winograd paths, cached U, split-K, ...)
    p_on, l_on = _run(True, ov)
    p_off, l_off = _run(False, ov)
    for a, b in zip(l_on, l_off):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (l_on, l_off)
    rel = float((p_on - p_off).norm() / p_off.norm())
    assert rel < 1e-5, rel
    assert torch.isfinite(p_on).all()


@pytest.mark.timeout(900)
def test_every_pair_of_switches_off_equals_serial_step():
    """The intermediate settings a user can select: with every SINGLE switch and every PAIR of switches forced off in an
    otherwise fully overlapped run, four training steps leave the parameters of the serial run (tiny widths: 55 runs of ~1 s)."""
    import itertools
    p_off, l_off = _run(False, TINY)
    combos = [(a,) for a in SWITCHES] + list(itertools.combinations(SWITCHES, 2))
    bad = []
    for off in combos:
        p, l = _run(True, TINY, off=off)
        rel = float((p - p_off).norm() / p_off.norm())
        dl = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(l, l_off))
        if not (rel < 1e-5 and dl <= 1e-4 and bool(torch.isfinite(p).all())):
            bad.append((off, rel, dl))
    assert not bad, bad


@pytest.mark.parametrize("width", ["tiny", "full"])
def test_joint_roi_pass_equals_two_passes(width):
    """ABR_JOINT_ROI in the overlapped step: the distillation RoIs through layer4 / the predictor TOGETHER with the detection RoIs (rows of the
    same GEMMs) leave the parameters of the two-pass step (train_incremental.py:89-95); only the accumulation order of the weight gradients
    differs (one launch over all rows instead of two)."""
    ov = TINY if width == "tiny" else TINY[8:]
    p_j, l_j = _run(True, ov, joint=True)
    p_2, l_2 = _run(True, ov)
    for a, b in zip(l_j, l_2):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (l_j, l_2)
    rel = float((p_j - p_2).norm() / p_2.norm())
    assert rel < 1e-5, rel


@pytest.mark.parametrize("width", ["tiny", "full"])
def test_one_call_per_bottleneck_pass_equals_the_per_conv_calls(width, monkeypatch):
    """ABR_BLOCK_PLANS (round 5): the bottlenecks' no-backward forward passes as one abr_conv_run each and every conv's backward pass (weight
    gradient on its side stream + input gradient) as one abr_conv_run -- the same launches with the same arguments, so four overlapped training
    steps leave BIT-IDENTICAL parameters and losses with the tables on and off"""
    from abr_iod_amd.modeling.backbone import resnet
    ov = TINY if width == "tiny" else TINY[8:]
    monkeypatch.setattr(resnet, "BLOCK_PLANS", True)
    p_on, l_on = _run(True, ov, joint=True)
    monkeypatch.setattr(resnet, "BLOCK_PLANS", False)
    p_off, l_off = _run(True, ov, joint=True)
    assert l_on == l_off, (l_on, l_off)
    assert torch.equal(p_on, p_off), float((p_on - p_off).abs().max())
