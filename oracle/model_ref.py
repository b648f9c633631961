"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the reference's R50-C4 Faster R-CNN forward
(maskrcnn_benchmark/modeling/backbone/resnet.py, rpn/rpn.py:114-121, roi_heads/box_head/*) over a reference-layout
state_dict, with ROIAlign from oracle.c wrapped as an autograd Function (the reference's own CPU ROIAlign has no backward,
csrc/ROIAlign.h:44; oracle.c's backward restates the CUDA formula and is pinned by the adjoint test).
Autograd on this model supplies the reference gradients the GPU path is compared with."""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops as cops
from . import torch_ref as R


class _RoiAlignRef(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, scale, ph, pw, sr):
        ctx.save_for_backward(rois)
        ctx.meta = (scale, ph, pw, sr, feat.shape)
        return torch.from_numpy(cops.roi_align_forward(feat.detach().numpy(), rois.numpy(), scale, ph, pw, sr))

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        scale, ph, pw, sr, shp = ctx.meta
        gx = cops.roi_align_backward(g.contiguous().numpy(), rois.numpy(), scale, ph, pw, *shp, sr)
        return torch.from_numpy(gx), None, None, None, None, None


class RefModel(object):
    BLOCKS = {"layer1": 3, "layer2": 4, "layer3": 6}

    def __init__(self, sd, trainable_prefixes=("backbone.body.layer2", "backbone.body.layer3", "rpn.", "roi_heads."), bf16_backbone=False):
        self.bf16_backbone = bf16_backbone   # cfg.DTYPE == "bfloat16": layer1-3 convs on bf16-rounded operands (the stem stays fp32)
        self.p = {}
        for k, v in sd.items():
            t = v.detach().cpu().float().clone()
            if any(k.startswith(pre) for pre in trainable_prefixes) and ("bn" not in k.split(".")[-2] and "downsample.1" not in k):
                t.requires_grad_(True)
            self.p[k] = t

    def _bn(self, prefix):
        return tuple(self.p[f"{prefix}.{n}"] for n in ("weight", "bias", "running_mean", "running_var"))

    def _block(self, x, prefix, stride, bf16=False):
        has_ds = f"{prefix}.downsample.0.weight" in self.p
        q = {"conv1.weight": self.p[f"{prefix}.conv1.weight"], "conv2.weight": self.p[f"{prefix}.conv2.weight"],
             "conv3.weight": self.p[f"{prefix}.conv3.weight"], "bn1": self._bn(f"{prefix}.bn1"), "bn2": self._bn(f"{prefix}.bn2"),
             "bn3": self._bn(f"{prefix}.bn3")}
        if has_ds:
            q["downsample.0.weight"] = self.p[f"{prefix}.downsample.0.weight"]
            q["ds_bn"] = self._bn(f"{prefix}.downsample.1")
        return R.bottleneck(x, q, stride, has_ds, bf16=bf16)

    def backbone(self, images):
        b = "backbone.body"
        x = F.relu(R.frozen_bn(F.conv2d(images, self.p[f"{b}.stem.conv1.weight"][:, :3], stride=2, padding=3), *self._bn(f"{b}.stem.bn1")))
        x = F.max_pool2d(x, 3, 2, 1)
        for name, n in self.BLOCKS.items():
            for i in range(n):
                x = self._block(x, f"{b}.{name}.{i}", (2 if name != "layer1" else 1) if i == 0 else 1, bf16=self.bf16_backbone)
        return x

    def rpn_head(self, feat):
        h = "rpn.head"
        t = F.relu(F.conv2d(feat, self.p[f"{h}.conv.weight"], self.p[f"{h}.conv.bias"], padding=1))
        return (F.conv2d(t, self.p[f"{h}.cls_logits.weight"], self.p[f"{h}.cls_logits.bias"]),
                F.conv2d(t, self.p[f"{h}.bbox_pred.weight"], self.p[f"{h}.bbox_pred.bias"]))

    def box_head(self, feat, rois, sr=0, res=7, scale=0.0625):
        pooled = _RoiAlignRef.apply(feat, rois, scale, res, res, sr)
        x = pooled
        pre = "roi_heads.box.feature_extractor.head.layer4"
        for i in range(3):
            x = self._block(x, f"{pre}.{i}", 2 if i == 0 else 1)
        v = F.adaptive_avg_pool2d(x, 1).flatten(1)
        pr = "roi_heads.box.predictor"
        logits = F.linear(v, self.p[f"{pr}.cls_score.weight"], self.p[f"{pr}.cls_score.bias"])
        reg = F.linear(v, self.p[f"{pr}.bbox_pred.weight"], self.p[f"{pr}.bbox_pred.bias"])
        return pooled, logits, reg

    def grads(self):
        return {k: v.grad for k, v in self.p.items() if v.requires_grad and v.grad is not None}
