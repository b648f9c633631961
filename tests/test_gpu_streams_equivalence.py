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


def _run(overlap, width_overrides, steps=4):
    from abr_iod_amd import ops
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.modeling.rpn import rpn
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    saved = (ops.WGRAD_SIDE_STREAM, rpn.PROPOSALS_SIDE_STREAM, trainer.SOURCE_OVERLAP, trainer.SOURCE_STREAM, trainer.SOURCE_HEAD_STREAM)
    ops.WGRAD_SIDE_STREAM = rpn.PROPOSALS_SIDE_STREAM = trainer.SOURCE_OVERLAP = trainer.SOURCE_STREAM = trainer.SOURCE_HEAD_STREAM = overlap
    try:
        cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, overrides=width_overrides)
        ms, mt = build_models(cfg_s, cfg_t, seed=0)
        opt = make_optimizer(cfg_t, mt)
        opt._prep_stream = overlap
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
        ops.WGRAD_SIDE_STREAM, rpn.PROPOSALS_SIDE_STREAM, trainer.SOURCE_OVERLAP, trainer.SOURCE_STREAM, trainer.SOURCE_HEAD_STREAM = saved


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
