#!/usr/bin/env python3
"""Full-size golden FROM THE REFERENCE (in-container only; SURVEY.md §8c fixture 10 = BASELINE.json configs[0]):
one incremental training step's forward (tools/train_incremental.py:82-116) of the FULL-WIDTH R50-C4 on two synthetic
600x1000 images, MODEL.DEVICE cpu, run by /root/reference's own code.

Weights are NOT stored (33 M floats): both sides regenerate them -- this package's seeded initialisation
(`engine/synthetic.py::build_models(seed=0)`, CPU generator, identical here and on the GPU box) exported through
`reference_state_dict`, the target's trainable tensors perturbed by `tests/e2e_common.py::perturb_trainable` -- and this script
loads that state_dict into the REFERENCE's models with the reference's `load_state_dict`.

Stored (tests/golden/e2e_full_<config>.npz, < 0.3 MB): everything random the reference drew (RPN sampler index lists, box-head
sampler index lists, the 64 soften picks per image), the reference's proposal lists (post-NMS + GT; top-128 of the source), and
its outputs: the 4 detector losses, ID and ARD losses, and spot values of the C4 features / RPN logits / detection logits.
The reference cannot run backward on CPU (csrc/ROIAlign.h:44), so no gradients.

    python tests/golden/make_golden_e2e_full.py [15-5|10-10|10-5|finetune ...]
    python tests/golden/make_golden_e2e_full.py 15-5@4      (B = 4, the benchmarked batch -> e2e_full_15-5_b4.npz)
"""
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402

rh.setup()

from e2e_common import CONFIGS, needs_source, perturb_trainable  # noqa: E402
from maskrcnn_benchmark.distillation.distillation import (  # noqa: E402
    calculate_attentive_roi_feature_distillation, calculate_roi_distillation_losses)
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler  # noqa: E402
from maskrcnn_benchmark.modeling.detector import build_detection_model  # noqa: E402
from maskrcnn_benchmark.structures.bounding_box import BoxList  # noqa: E402

H, W, B = 600, 1000, 2


def our_state_dicts(name):
    """this package's seeded weights in the reference layout (built on CPU: no kernels run)"""
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    from abr_iod_amd.utils.checkpoint import reference_state_dict
    task, dist_type, feat, alpha, beta, gamma, label_range, n_old = CONFIGS[name]
    cfg_s, cfg_t = make_cfgs(task, dist_type=dist_type, feat=feat, alpha=alpha, beta=beta, gamma=gamma, overrides=["MODEL.DEVICE", "cpu"])
    ms, mt = build_models(cfg_s, cfg_t, seed=0, need_source=needs_source(name))
    sd_t = reference_state_dict(mt)
    perturb_trainable(sd_t, [n for n, p in mt.named_parameters() if p.requires_grad])
    return (reference_state_dict(ms) if ms is not None else None), sd_t


def load(model, sd):
    own = model.state_dict()
    missing = [k for k in own if k not in sd and "anchor_generator" not in k]
    assert not missing, missing
    for k, v in sd.items():
        assert own[k].shape == v.shape, (k, own[k].shape, v.shape)
    model.load_state_dict(sd, strict=False)


def main(name, B=B):
    task, dist_type, feat, alpha, beta, gamma, label_range, n_old = CONFIGS[name]
    torch.set_num_threads(8)
    yaml = f"configs/voc/{task}/e2e_faster_rcnn_R_50_C4_4x_{'RB_' if needs_source(name) else ''}Target_model.yaml"
    n_all = {"15-5": 21, "10-10": 21, "10-5": 16}[task]      # train_incremental.py:445-447 at step 1
    cfg_t = rh.default_cfg(yaml, ["MODEL.DEVICE", "cpu", "DIST.TYPE", dist_type, "MODEL.ROI_BOX_HEAD.NUM_CLASSES", n_all])
    cfg_s = rh.default_cfg(yaml, ["MODEL.DEVICE", "cpu", "DIST.TYPE", dist_type, "MODEL.ROI_BOX_HEAD.NUM_CLASSES", n_old + 1])
    sd_s, sd_t = our_state_dicts(name)
    mt = build_detection_model(cfg_t)
    load(mt, sd_t)
    mt.train()
    ms = None
    if sd_s is not None:
        ms = build_detection_model(cfg_s)
        load(ms, sd_s)
        ms.eval()

    from abr_iod_amd.engine.synthetic import synthetic_batch
    images, tg = synthetic_batch(B, H, W, seed=42, label_range=label_range, device="cpu")
    targets = []
    for t in tg:
        bl = BoxList(t.bbox.clone(), (W, H), mode="xyxy")
        bl.add_field("labels", t.get_field("labels").clone())
        targets.append(bl)

    draws = []
    orig = BalancedPositiveNegativeSampler.__call__

    def rec(self, matched_idxs, objectness=None):
        pos, neg = orig(self, matched_idxs, objectness)
        draws.append((pos, neg))
        return pos, neg
    BalancedPositiveNegativeSampler.__call__ = rec

    out = {"config": np.array(name), "image_seed": np.array(42), "batch": np.array(B)}
    torch.manual_seed(11)
    random.seed(5)
    t0 = time.time()
    with torch.no_grad():
        if ms is not None:
            soften_result, _, soften_proposal, feat_s, _, _, rpn_out_s, raf_s = ms.generate_soften_proposal(images)       # :82-85
            from maskrcnn_benchmark.structures.image_list import to_image_list
            (all_props, _), _, _ = ms.rpn(to_image_list(images), feat_s, None)
            for i, p in enumerate(all_props):
                order = p.get_field("objectness").sort(descending=True)[1]
                ranked = p[order][:128]
                sel = [int(torch.nonzero((ranked.bbox == bx).all(dim=1))[0, 0]) for bx in soften_proposal[i].bbox]
                out[f"soften_sel{i}"] = np.array(sel, np.int64)
                out[f"src_top128_{i}"] = ranked.bbox.numpy()
                out[f"src_n_props{i}"] = np.array(len(p))
        n_draws0 = len(draws)
        loss_dict, feat_t, _, anchors, rpn_out_t, props_t, raf_det, soft_res = mt(images, targets)                      # :89-90
        if ms is not None:
            target_result, _, raf_t = mt.forward(images, targets, features=feat_t, proposals=soften_proposal)           # :93-95
            l_id = calculate_roi_distillation_losses(soften_result, target_result, dist=dist_type)                      # :101
            l_ard = calculate_attentive_roi_feature_distillation(raf_s, raf_t, gamma=gamma)                             # :113-116
            out["loss_id"], out["loss_ard"] = np.array(float(l_id)), np.array(float(l_ard))
            out["target_scores_head"] = target_result[0][:8].numpy()
            out["soften_scores_head"] = soften_result[0][:8].numpy()
        mt.rpn.box_selector_train.train()
        pre = mt.rpn.box_selector_train(anchors, rpn_out_t[0], rpn_out_t[1], targets)   # the post-NMS (+GT) lists the box head sampled from
    print(name, "reference forward: %.1f s" % (time.time() - t0))
    tdraws = draws[n_draws0:]
    assert len(tdraws) == 2, len(tdraws)   # (one RPN draw, one box-head draw; each a per-image list)
    (rpn_pos, rpn_neg), (head_pos, head_neg) = tdraws
    for i in range(B):
        out[f"rpn_pos{i}"] = torch.nonzero(rpn_pos[i]).squeeze(1).numpy().astype(np.int32)
        out[f"rpn_neg{i}"] = torch.nonzero(rpn_neg[i]).squeeze(1).numpy().astype(np.int32)
        out[f"head_sel{i}"] = torch.nonzero(head_pos[i] | head_neg[i]).squeeze(1).numpy().astype(np.int32)
        out[f"tgt_props{i}"] = pre[i].bbox.numpy()
        out[f"det_labels{i}"] = props_t[i].get_field("labels").numpy().astype(np.int16)
        out[f"gt{i}"] = targets[i].bbox.numpy()
        out[f"gt_labels{i}"] = targets[i].get_field("labels").numpy()
    f = feat_t[0]
    out["feat_t_absmax"] = np.array(float(f.abs().max()))
    out["feat_t_spot"] = f[:, ::97, ::7, ::11].numpy()                    # [B, 11, 6, 6]
    out["rpn_obj_spot"] = rpn_out_t[0][0][:, :, ::5, ::9].numpy()         # [2, 15, 8, 7]
    out["rpn_obj_absmax"] = np.array(float(rpn_out_t[0][0].abs().max()))
    out["det_logits_head"] = soft_res[0][:16].numpy()
    for k, v in loss_dict.items():
        out[k] = np.array(float(v))
    path = os.path.join(HERE, f"e2e_full_{name}.npz" if B == 2 else f"e2e_full_{name}_b{B}.npz")
    np.savez_compressed(path, **out)
    print({k: float(out[k]) for k in out if k.startswith("loss")}, "file KB", os.path.getsize(path) / 1e3)
    BalancedPositiveNegativeSampler.__call__ = orig


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["15-5"]):
        if "@" in nm:
            nm, b = nm.split("@")
            main(nm, int(b))
        else:
            main(nm)
