"""Mirror of maskrcnn_benchmark/layers/__init__.py:4-20 for the hot path: same public names
(ROIAlign, roi_align, nms, smooth_l1_loss, SigmoidFocalLoss, FrozenBatchNorm2d, Conv2d), backed by the HIP library."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C, ops
from .._lib import require_cuda
from ._layout import as_nhwc, from_nhwc

nms = _C.nms  # layers/nms.py:8 (amp.float_function is a no-op at O0)


# ------------------------------------------------------------------------------------------- ROIAlign
class _ROIAlign(Function):
    """layers/roi_align.py:12-48.  input: logical [B,C,H,W]; output: logical [K,C,ph,pw] (channels-last memory)."""

    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, bin_step=1):
        require_cuda(input, roi)
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale, ctx.sampling_ratio, ctx.bin_step = spatial_scale, sampling_ratio, bin_step
        ctx.input_shape = input.size()
        xin = as_nhwc(input)
        out = ops.roi_align_forward(xin, roi, spatial_scale, ctx.output_size[0], ctx.output_size[1],
                                    sampling_ratio, bin_step)
        ops.amax_carry_bound(out, xin)   # pooled values are averages of bilinear samples: bounded by the feature map's amax (f16x3 scales)
        return from_nhwc(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        g = ops.roi_align_backward(as_nhwc(grad_output), rois, ctx.spatial_scale, ctx.output_size[0], ctx.output_size[1],
                                   ctx.sampling_ratio, bs, h, w, ch, ctx.bin_step)
        return from_nhwc(g), None, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    """layers/roi_align.py:51-70"""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois, bin_step=1):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, bin_step)

    def __repr__(self):
        return "{}(output_size={}, spatial_scale={}, sampling_ratio={})".format(
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)


# ------------------------------------------------------------------------------------------- smooth L1
class _SmoothL1(Function):
    @staticmethod
    def forward(ctx, input, target, beta, size_average):
        scale = 1.0 / max(input.numel(), 1) if size_average else 1.0
        loss, grad = ops.smooth_l1(input, target, beta, scale=scale, want_grad=input.requires_grad)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return ops.scale_(grad, 1.0, g.contiguous()), None, None, None


def smooth_l1_loss(input, target, beta=1.0 / 9, size_average=True):
    """layers/smooth_l1_loss.py:6-17"""
    if input.numel() == 0:
        return input.sum() * 0.0
    return _SmoothL1.apply(input, target.detach(), beta, size_average)


# ------------------------------------------------------------------------------------------- focal loss
class _SigmoidFocalLoss(Function):
    """layers/sigmoid_focal_loss.py:9-35"""

    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.num_classes, ctx.gamma, ctx.alpha = logits.shape[1], gamma, alpha
        return _C.sigmoid_focalloss_forward(logits, targets, ctx.num_classes, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        d = _C.sigmoid_focalloss_backward(logits, targets, d_loss.contiguous(), ctx.num_classes, ctx.gamma, ctx.alpha)
        return d, None, None, None, None


sigmoid_focal_loss_cuda = _SigmoidFocalLoss.apply


class SigmoidFocalLoss(nn.Module):
    """layers/sigmoid_focal_loss.py:55-76 (returns the SUM)."""

    def __init__(self, gamma, alpha):
        super().__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        return sigmoid_focal_loss_cuda(logits, targets, self.gamma, self.alpha).sum()

    def __repr__(self):
        return "{}(gamma={}, alpha={})".format(self.__class__.__name__, self.gamma, self.alpha)


# ------------------------------------------------------------------------------------------- FrozenBN
class FrozenBatchNorm2d(nn.Module):
    """layers/batch_norm.py:6-31: fixed statistics and affine, NO epsilon.  On the hot path it never runs as its
    own kernel: `scale_bias()` feeds the conv kernel's epilogue (y = acc*scale + bias)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._fused = None

    def scale_bias(self):
        if self._fused is None or self._fused[0].device != self.weight.device:
            scale = self.weight * self.running_var.rsqrt()      # batch_norm.py:27
            bias = self.bias - self.running_mean * scale        # :28
            self._fused = (scale.contiguous(), bias.contiguous())
        return self._fused

    def invalidate(self):
        self._fused = None

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._fused = None

    def forward(self, x):
        scale, bias = self.scale_bias()
        eye = torch.eye(scale.numel(), device=x.device).view(scale.numel(), 1, 1, scale.numel())
        return from_nhwc(ops.conv_forward(as_nhwc(x), eye, scale=scale, bias=bias))
