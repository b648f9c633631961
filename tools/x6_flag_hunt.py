#!/usr/bin/env python3
"""Where does the bf16x6 range guard trip?  Runs the configs[4] ragged ABR step of tests/test_gpu_configs4_whole.py piece by piece and
polls abr_x6_range_flags (synchronising) after each piece and at every autograd node boundary of the backward pass."""
import os, sys, random, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tempfile
from test_gpu_configs4_whole import _rehearsal_memory, _models
from abr_iod_amd import ops
from abr_iod_amd.data.abr import BoxRehearsalABR, GPUTransform
from abr_iod_amd.data.gpu_transforms import to_device_u8
from abr_iod_amd.structures.bounding_box import BoxList

tmp = tempfile.mkdtemp()
names = _rehearsal_memory(tmp)
abr = BoxRehearsalABR(tmp, names, batch_size=3, shuffle=False)
cfg_in = types.SimpleNamespace(INPUT=types.SimpleNamespace(MIN_SIZE_TRAIN=(600,), MAX_SIZE_TRAIN=1000, MIN_SIZE_TEST=600, MAX_SIZE_TEST=1000, FLIP_PROB_TRAIN=0.5,
                               PIXEL_MEAN=[102.9801, 115.9465, 122.7717], PIXEL_STD=[1.0, 1.0, 1.0], TO_BGR255=True, BRIGHTNESS=0.0, CONTRAST=0.0, SATURATION=0.0, HUE=0.0))
tf = GPUTransform(cfg_in, is_train=True)
rs = np.random.RandomState(9); random.seed(3); torch.manual_seed(3)
samples = []
for kind, (H, W) in (("mixup", (375, 500)), ("mosaic", (375, 500)), ("new", (300, 500))):
    img = to_device_u8(rs.randint(0, 256, (H, W, 3), dtype=np.uint8))
    t = BoxList(torch.tensor([[30.0, 40.0, 130.0, 160.0], [250.0, 100.0, 420.0, 290.0]]), (W, H), mode="xyxy"); t.add_field("labels", torch.tensor([12, 15]))
    if kind == "mixup": img, t = abr._start_mixup(img, t)
    elif kind == "mosaic": img, t = abr._start_boxes_mosaic((W, H))
    samples.append(tf(img, t))
images, targets = tf.collate(samples)
targets = [t.to("cuda") for t in targets]
cfg_s, cfg_t, ms, mt = _models("float32")

def poll(tag):
    f = ops.x6_range_flags()
    print("%-60s flags=%d" % (tag, f), flush=True)
    return f

poll("start")
from abr_iod_amd.distillation.distillation import calculate_attentive_roi_feature_distillation, calculate_roi_distillation_losses
with torch.no_grad():
    soften_result, _, soften_proposal, feat_s, _, _, _, raf_s = ms.generate_soften_proposal(images)
poll("source model: generate_soften_proposal")
ops._sample_calls[0] = 0; random.seed(0)
loss_dict, feat_t, bb_feats, anchors, rpn_out, props, raf_det, _ = mt(images, targets)
poll("target forward")
target_result, _, raf_t = mt.forward(images, targets, features=feat_t, proposals=soften_proposal)
poll("target second RoI pass")
l_id = calculate_roi_distillation_losses(soften_result, target_result, dist="id")
l_ard = calculate_attentive_roi_feature_distillation(raf_s, raf_t, gamma=1.0)
total = sum(loss_dict.values()) + l_id + l_ard
print({k: float(v.detach()) for k, v in loss_dict.items()}, float(l_id.detach()), float(l_ard.detach()))
for name, t in (("raf_t (second-pass pooled)", raf_t), ("raf_det (detection pooled)", raf_det), ("features C4", feat_t[0]), ("rpn_out obj", rpn_out[0][0])):
    if torch.is_tensor(t) and t.requires_grad:
        t.register_hook(lambda g, name=name: (poll("backward reached grad of " + name), print("    |g| min nonzero %.3e max %.3e" % (float(g[g != 0].abs().min()) if bool((g != 0).any()) else 0.0, float(g.abs().max()))), None)[2])
for i, bf in enumerate(bb_feats):
    if bf.requires_grad:
        bf.register_hook(lambda g, i=i: (poll("backward reached grad of backbone stage output %d" % i), print("    |g| min nonzero %.3e max %.3e" % (float(g[g != 0].abs().min()) if bool((g != 0).any()) else 0.0, float(g.abs().max()))), None)[2])
mt.flat.zero_grad()
total.backward()
ops.join_side_stream()
poll("backward done")
