"""smoke_step(): one tiny incremental training step (ARD + ID on) on cuda:0; asserts finite losses and non-zero gradients."""
import torch

from .synthetic import build_models, make_cfgs, synthetic_batch
from .trainer import train_step


def smoke_step():
    from ..solver.build import make_lr_scheduler, make_optimizer
    overrides = ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 600, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300,
                 "MODEL.RPN.POST_NMS_TOP_N_TEST", 150, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64]
    cfg_s, cfg_t = make_cfgs("15-5", overrides=overrides)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    images, targets = synthetic_batch(2, 192, 256, seed=1, max_boxes=2)
    for t in targets:
        t.bbox[:, 0::2].clamp_(max=255); t.bbox[:, 1::2].clamp_(max=191)
        t.bbox[:, 2] = torch.max(t.bbox[:, 2], t.bbox[:, 0] + 8).clamp(max=255); t.bbox[:, 3] = torch.max(t.bbox[:, 3], t.bbox[:, 1] + 8).clamp(max=191)
    before = mt.flat.params[: mt.flat.n_trainable].clone()
    loss_dict, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in loss_dict.items()}
    assert all(v == v and abs(v) < 1e4 for v in vals.values()), vals
    assert float(mt.flat.grads.abs().sum()) > 0, "no gradient reached the flat buffer"
    assert not torch.equal(before, mt.flat.params[: mt.flat.n_trainable]), "SGD step did not update the parameters"
    print("smoke step losses:", {k: round(v, 4) for k, v in vals.items()})
