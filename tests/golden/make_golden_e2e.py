#!/usr/bin/env python3
"""End-to-end golden FROM THE REFERENCE (in-container only): one incremental training step's forward
(tools/train_incremental.py:82-116) on a tiny-width R50-C4 (SURVEY.md §8c fixture 9):
source model (K_old+1 = 16 classes, eval) + target model (21 classes, train), 2 x 160x224 images.

What is stored (tests/golden/e2e_tiny.npz, ~5 MB): both reference state_dicts, inputs, GT, everything random the
reference drew (RPN / box-head sampler masks, the 64 soften proposals) and the reference's outputs: C4 features, RPN head
outputs, post-NMS proposals, detection-pass logits, the 4 detector losses, soften results, target-on-soften results,
ID and ARD losses.  The reference cannot run backward on CPU (csrc/ROIAlign.h:44), so no gradients here.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.setup()

from maskrcnn_benchmark.distillation.distillation import (  # noqa: E402
    calculate_attentive_roi_feature_distillation, calculate_roi_distillation_losses)
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler  # noqa: E402
from maskrcnn_benchmark.modeling.detector import build_detection_model  # noqa: E402
from maskrcnn_benchmark.structures.bounding_box import BoxList  # noqa: E402
from maskrcnn_benchmark.structures.image_list import to_image_list  # noqa: E402

TINY = ["MODEL.DEVICE", "cpu", "MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32,
        "MODEL.RESNETS.WIDTH_PER_GROUP", 8, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128,
        "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 600, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 100,
        "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300, "MODEL.RPN.POST_NMS_TOP_N_TEST", 150,
        "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64, "DIST.TYPE", "id"]
YAML = "configs/voc/15-5/e2e_faster_rcnn_R_50_C4_4x_RB_Target_model.yaml"


def randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if m.__class__.__name__ == "FrozenBatchNorm2d":
            n = m.weight.numel()
            damp = 1.0 / 64 if name.endswith("stem.bn1") else (0.25 if name.endswith("bn3") else 1.0)
            m.weight.copy_((torch.rand(n, generator=g) + 0.5) * damp)
            m.bias.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=g) + 0.5)


def main():
    torch.set_num_threads(4)
    cfg_t = rh.default_cfg(YAML, TINY)
    cfg_s = rh.default_cfg(YAML, TINY + ["MODEL.ROI_BOX_HEAD.NUM_CLASSES", 16])
    torch.manual_seed(0)
    mt = build_detection_model(cfg_t)
    ms = build_detection_model(cfg_s)
    with torch.no_grad():
        randomize_bn(mt, 1)
        sd_t = mt.state_dict()
        g = torch.Generator().manual_seed(7)
        for k, v in ms.state_dict().items():  # source = perturbed copy of the target (old-class rows of the grown head)
            src = sd_t[k][: v.shape[0]] if v.shape != sd_t[k].shape else sd_t[k]
            if v.dtype.is_floating_point and "bn" not in k and "downsample.1" not in k and "anchor_generator" not in k:
                v.copy_(src * (1.0 + 0.05 * torch.randn(src.shape, generator=g)))
            else:
                v.copy_(src)
    mt.train()
    ms.eval()

    g = torch.Generator().manual_seed(3)
    images = torch.randint(0, 256, (2, 3, 160, 224), generator=g).float() - torch.tensor([102.9801, 115.9465, 122.7717]).view(1, 3, 1, 1)
    gt = [torch.tensor([[20., 30., 120., 140.], [130., 10., 210., 90.]]), torch.tensor([[5., 5., 60., 70.], [90., 40., 190., 140.]])]
    gl = [torch.tensor([16, 18]), torch.tensor([20, 17])]
    targets = []
    for b, l in zip(gt, gl):
        t = BoxList(b, (224, 160), mode="xyxy"); t.add_field("labels", l); targets.append(t)

    # record every sampler draw (RPN first, then box head)
    draws = []
    orig = BalancedPositiveNegativeSampler.__call__

    def rec(self, matched_idxs, objectness=None):
        pos, neg = orig(self, matched_idxs, objectness)
        draws.append((torch.stack(pos), torch.stack(neg)) if len({p.numel() for p in pos}) == 1 else (pos, neg))
        return pos, neg
    BalancedPositiveNegativeSampler.__call__ = rec

    out = {}
    torch.manual_seed(11)
    random.seed(5)
    with torch.no_grad():
        soften_result, _, soften_proposal, feat_s, _, _, rpn_out_s, raf_s = ms.generate_soften_proposal(images)
        # full ranked proposal list of the source's test selector, to let a re-implementation reproduce the python random.sample
        il = to_image_list(images)
        (all_props, _), _, _ = ms.rpn(il, feat_s, None)
    for i, p in enumerate(all_props):
        order = p.get_field("objectness").sort(descending=True)[1]
        ranked = p[order]
        sel = []
        for bx in soften_proposal[i].bbox:
            sel.append(int(torch.nonzero((ranked.bbox == bx).all(dim=1))[0, 0]))
        out[f"soften_sel{i}"] = np.array(sel)
        out[f"src_props{i}"] = ranked.bbox.numpy(); out[f"src_scores{i}"] = ranked.get_field("objectness").numpy()
        out[f"soften_boxes{i}"] = soften_proposal[i].bbox.numpy()

    with torch.no_grad():  # forward only: ROIAlign has no CPU backward
        loss_dict, feat_t, _, anchors, rpn_out_t, props_t, raf_det, soft_res = mt(images, targets)
        target_result, _, raf_t = mt.forward(images, targets, features=feat_t, proposals=soften_proposal)
        l_id = calculate_roi_distillation_losses(soften_result, target_result, dist="id")
        l_ard = calculate_attentive_roi_feature_distillation(raf_s, raf_t, gamma=1.0)
        # the post-NMS (+GT) proposal list the box head sampled from
        mt.rpn.box_selector_train.train()
        pre = mt.rpn.box_selector_train(anchors, rpn_out_t[0], rpn_out_t[1], targets)
    assert len(draws) == 2, len(draws)
    out["rpn_pos"], out["rpn_neg"] = draws[0][0].numpy(), draws[0][1].numpy()
    for i in range(2):
        out[f"head_pos{i}"], out[f"head_neg{i}"] = draws[1][0][i].numpy(), draws[1][1][i].numpy()
        out[f"tgt_props{i}"] = pre[i].bbox.numpy(); out[f"tgt_scores{i}"] = pre[i].get_field("objectness").numpy()
        out[f"det_boxes{i}"] = props_t[i].bbox.numpy()
        out[f"det_labels{i}"] = props_t[i].get_field("labels").numpy()
        out[f"det_reg_targets{i}"] = props_t[i].get_field("regression_targets").numpy()
        out[f"gt{i}"] = gt[i].numpy(); out[f"gt_labels{i}"] = gl[i].numpy()
    out.update(images=images.numpy(), feat_s=feat_s[0].numpy(), feat_t=feat_t[0].numpy(),
               rpn_obj_t=rpn_out_t[0][0].numpy(), rpn_reg_t=rpn_out_t[1][0].numpy(),
               det_logits=soft_res[0].numpy(), det_boxreg=soft_res[1].numpy(),
               soften_scores=soften_result[0].numpy(), soften_bboxes=soften_result[1].numpy(),
               target_scores=target_result[0].numpy(), target_bboxes=target_result[1].numpy(),
               raf_s=raf_s.numpy()[:8], raf_t=raf_t.numpy()[:8],
               loss_id=float(l_id), loss_ard=float(l_ard), **{k: float(v) for k, v in loss_dict.items()})
    for k, v in mt.state_dict().items():
        out["T/" + k] = v.numpy()
    for k, v in ms.state_dict().items():
        out["S/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "e2e_tiny.npz"), **out)
    print({k: out[k] for k in ("loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg", "loss_id", "loss_ard")})
    print("params", sum(v.size for k, v in out.items() if k.startswith("T/")), "file MB",
          os.path.getsize(os.path.join(HERE, "e2e_tiny.npz")) / 1e6)


if __name__ == "__main__":
    main()
