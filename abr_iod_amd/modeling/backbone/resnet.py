"""ResNet-50 C4 body and C5 head on the HIP conv engine.

Mirror of maskrcnn_benchmark/modeling/backbone/resnet.py for the one variant every configs/voc YAML uses
(CONV_BODY "R-50-C4", BottleneckWithFixedBatchNorm, StemWithFixedBatchNorm, STRIDE_IN_1X1=True, no DCN, groups=1):
    ResNet (:81-155)       stem + layer1..layer3, FREEZE_CONV_BODY_AT semantics of _freeze_backbone (:134-143)
    ResNetHead (:158-207)  layer4, used by ResNet50Conv5ROIFeatureExtractor
    Bottleneck (:242-346), BaseStem (:349-368)
Module / parameter / buffer NAMES are the reference's (state_dict keys match; conv weights are stored OHWI
instead of OIHW, converted at the checkpoint boundary).

Execution model: every conv is ONE launch of the implicit-GEMM MFMA kernel with FrozenBN scale/bias, the residual
add and the ReLU fused in its epilogue.  A whole stage (3-6 bottlenecks) is a single autograd node whose backward is
hand-scheduled: dgrad is the same kernel on a flipped/transposed weight copy with the ReLU mask of the producer
fused in the epilogue, wgrad accumulates atomically into the flat gradient buffer.  Activations are NHWC.
"""
import os
import threading
from collections import namedtuple

import torch
from torch import nn
from torch.autograd import Function

from ... import ops
from ...layers import FrozenBatchNorm2d
from ...layers._layout import as_nhwc, from_nhwc

StageSpec = namedtuple("StageSpec", ["index", "block_count", "return_features"])
ResNet50StagesTo4 = tuple(StageSpec(index=i, block_count=c, return_features=r) for (i, c, r) in ((1, 3, False), (2, 4, False), (3, 6, True)))
ResNet50StagesTo5 = tuple(StageSpec(index=i, block_count=c, return_features=r) for (i, c, r) in ((1, 3, False), (2, 4, False), (3, 6, False), (4, 3, True)))

# Weight versions: data derived from a weight tensor (the flipped dgrad copies; inside the library, under abr_conv_desc::w_version, the
# Winograd-domain weights and the fragment-packed bf16x3 planes the bf16x6 weights-direct kernel reads) is rebuilt lazily when its version is stale.  _PARAM_VERSION moves with EVERY change (optimiser steps, loads, in-place
# surgery), _STATIC_VERSION only with changes that can touch weights no optimiser owns (loads, surgery, model construction): the
# frozen source model's derived data therefore survives the target's optimiser steps.  Code that writes weights in place must call
# bump_param_version().
_PARAM_VERSION = [0]
_STATIC_VERSION = [0]


def bump_param_version():
    _PARAM_VERSION[0] += 1
    _STATIC_VERSION[0] += 1


def bump_trained_version():
    """after an optimiser step: only tensors an optimiser owns (Conv2d._optimised) have changed"""
    _PARAM_VERSION[0] += 1


class Conv2d(nn.Module):
    """nn.Conv2d stand-in (layers/misc.py:30-43) holding an OHWI weight [Cout,R,S,Cin]; `cin_pad` zero-pads input
    channels (the 3-channel stem runs with Cin=4 so that every gather is a 16 B load)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, cin_pad=None):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        cin = cin_pad or in_channels
        self.weight = nn.Parameter(torch.zeros(out_channels, kernel_size, kernel_size, cin))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self._wt = None
        self._wt_version = -1
        self._optimised = False   # set by FusedSGD for the convs it updates: their version moves with every optimiser step
        self._flat = None         # the model's FlatParams (set by GeneralizedRCNN.flatten_parameters): source of weight planes

    def kaiming_uniform_(self, a=1):
        """nn.init.kaiming_uniform_(w, a=1) of the reference (resnet.py:270,315,325,361), fan_in = Cin*R*S of the REAL channels."""
        fan_in = self.in_channels * self.kernel_size * self.kernel_size
        bound = (6.0 / ((1 + a * a) * fan_in)) ** 0.5
        with torch.no_grad():
            self.weight.zero_()
            self.weight[..., : self.in_channels].uniform_(-bound, bound)

    def load_oihw(self, w):
        """copy a reference-layout [Cout,Cin,R,S] tensor in"""
        with torch.no_grad():
            self.weight.zero_()
            self.weight[..., : w.shape[1]].copy_(w.permute(0, 2, 3, 1))
        self._wt_version = -1

    def oihw(self):
        return self.weight.detach()[..., : self.in_channels].permute(0, 3, 1, 2).contiguous()

    def dgrad_weight(self, scale=None):
        """[Cin,R,S,Cout] flipped copy with the FrozenBN scale folded in; rebuilt only after an optimiser step."""
        ops.prep_wait()
        if self._wt is None or self._wt_version != _PARAM_VERSION[0] or self._wt.device != self.weight.device:
            same = self._wt is not None and self._wt.device == self.weight.device
            self._wt = ops.conv_dgrad_weights(self.weight.detach(), scale, out=self._wt if same else None)
            self._wt_version = _PARAM_VERSION[0]
        return self._wt

    def dgrad_buffer(self):
        """the (possibly stale) buffer of the dgrad copy, allocated on first use: FusedSGD's batched weight preparation fills it"""
        if self._wt is None or self._wt.device != self.weight.device:
            Cout, R, S, Cin = self.weight.shape
            self._wt = torch.empty((Cin, R, S, Cout), dtype=self.weight.dtype, device=self.weight.device)
            self._wt_version = -1
        return self._wt

    def version(self):
        """abr_conv_desc::w_version for this conv's weight and for its dgrad copy: non-zero, changes whenever the values may have"""
        return 2 * _PARAM_VERSION[0] + 1 if self._optimised else 2 * _STATIC_VERSION[0] + 2


def _grad_buf(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# Round 5: one library call per bottleneck pass.  The four convs of a block's forward pass cost ~12 us of interpreter / binding time EACH
# (tools/dbg/host_segments.py: at B = 1 the step is the host's issue time); their descriptors, weights and FrozenBN vectors do not change from
# step to step, so a block keeps, per (input shape, stride, arithmetic), a table of abr_conv_op with all of that filled in (`_FwdPlan`) and
# writes only this call's pointers and amax words before ONE abr_conv_run.  Same kernels, same arguments: bit-identical to the per-conv path,
# which stays for everything the tables do not cover (ABR_BLOCK_PLANS=0: always).
BLOCK_PLANS = os.environ.get("ABR_BLOCK_PLANS", "1") != "0"


class _FwdPlan(object):
    """the no-backward forward pass of one Bottleneck (conv1 [+ downsample] -> conv2 -> conv3 + identity) as an abr_conv_run table"""

    def __init__(self, blk, x_shape, s, save=False):
        L = ops.L
        self.save = save
        B, H, W, Cin = x_shape
        self.blk, self.math = blk, blk.math
        ds = blk.downsample
        convs = [(blk.conv1, blk.bn1, s, 0, True)] + ([(ds[0], ds[1], s, 0, False)] if ds is not None else []) + \
                [(blk.conv2, blk.bn2, 1, 1, True), (blk.conv3, blk.bn3, 1, 0, True)]
        self.n = len(convs)
        self.arr = (L.ConvOp * self.n)()
        self.ptr = ops.C.cast(self.arr, ops.C.c_void_p)
        self.has_ds = ds is not None
        self.fused, self.wptr, self.convs = [], [], []
        shape = tuple(x_shape)
        shapes = []
        for i, (conv, bn, st, pad, relu) in enumerate(convs):
            if conv is blk.conv2:
                in_shape = shapes[0]          # conv1's output
            elif conv is blk.conv3:
                in_shape = shapes[-1]         # conv2's output
            else:
                in_shape = shape              # conv1 and the downsample branch read x
            sb = bn.scale_bias()
            d = ops.conv_desc(in_shape, conv.weight.shape, st, pad, sb[0], sb[1], None, None, relu, math=self.math)
            op = self.arr[i]
            op.kind = L.OP_FORWARD
            op.desc = d
            op.b = conv.weight.data_ptr()
            self.fused.append((bn, sb))
            self.wptr.append((conv, conv.weight.data_ptr()))
            self.convs.append(conv)
            shapes.append((d.B, d.Ho, d.Wo, d.Cout))
        self.shapes = shapes
        i1, i2, i3 = 0, self.n - 2, self.n - 1
        n1, n2 = _numel(shapes[i1]), _numel(shapes[i2])
        nd = _numel(shapes[1]) if self.has_ds else 0
        if save:   # o1, o2 are kept for the backward pass (tensors of their own); only the identity branch is a temporary
            self.off = (0, 0, 0)
            self.tmp_floats = nd
            self.o1_shape, self.o2_shape = shapes[i1], shapes[i2]
            # the Winograd-domain input of conv2, kept for its weight gradient (ops.wino_v_alloc)
            self.v_floats = 0
            if ops.KEEP_WINO_V and blk.conv2.weight.requires_grad:
                self.v_floats = int(L.lib().abr_conv_wino_v_floats(ops.C.byref(self.arr[i2].desc)))
        else:
            self.off = (0, n1, n1 + n2)        # o1, o2, identity branch inside the temporaries' buffer (floats)
            self.tmp_floats = n1 + n2 + nd
        self.out_shape = shapes[i3]
        self.h3 = self.math == ops.MATH_F16X3
        self.s = s
        # the table is mutable: two host threads running the same model (threaded inference) must not rewrite each other's pointers while
        # abr_conv_run -- which releases the GIL -- still walks them
        self.lock = threading.Lock()

    def valid(self):
        blk = self.blk
        if blk.math != self.math:
            return False
        for bn, sb in self.fused:
            if bn._fused is not sb:
                return False
        for conv, p in self.wptr:
            if conv.weight.data_ptr() != p:
                return False
        return True

    def run(self, x):
        with self.lock:
            return self._run(x)

    def _run(self, x):
        arr, n = self.arr, self.n
        st = ops.L.stream()
        f32, dev = torch.float32, x.device
        tmp = torch.empty(self.tmp_floats, dtype=f32, device=dev) if self.tmp_floats else None
        out = torch.empty(self.out_shape, dtype=f32, device=dev)
        base, xp = (tmp.data_ptr() if tmp is not None else 0), x.data_ptr()
        i2, i3 = n - 2, n - 1
        v2 = None
        if self.save:
            t1, t2 = torch.empty(self.o1_shape, dtype=f32, device=dev), torch.empty(self.o2_shape, dtype=f32, device=dev)
            o1, o2 = t1.data_ptr(), t2.data_ptr()
            if self.v_floats:
                v2 = torch.empty(self.v_floats, dtype=f32, device=dev)
            arr[i2].desc.wino_v = v2.data_ptr() if v2 is not None else None
        else:
            o1, o2 = base + 4 * self.off[0], base + 4 * self.off[1]
        idt = base + 4 * self.off[2] if self.has_ds else xp
        a0 = arr[0]
        a0.a, a0.out, a0.stream = xp, o1, st
        a0.desc.w_version = self.convs[0].version()
        if self.has_ds:
            a1 = arr[1]
            a1.a, a1.out, a1.stream = xp, idt, st
            a1.desc.w_version = self.convs[1].version()
        a2, a3 = arr[i2], arr[i3]
        a2.a, a2.out, a2.stream = o1, o2, st
        a2.desc.w_version = self.convs[i2].version()
        a3.a, a3.out, a3.stream = o2, out.data_ptr(), st
        a3.desc.residual = idt
        a3.desc.w_version = self.convs[i3].version()
        if self.h3:
            xw, xe = ops.amax_of(x)
            if xw is None:
                xw, xe = ops.amax_of(ops.amax_compute(x))
            w1, e1 = ops.amax_new()
            w2, e2 = ops.amax_new()
            w3, e3 = ops.amax_new()
            d0, d2, d3 = a0.desc, a2.desc, a3.desc
            d0.x_amax, d0.x_amax_epoch, d0.out_amax, d0.out_amax_epoch = xw, xe, w1, e1
            if self.has_ds:
                d1 = arr[1].desc
                d1.x_amax, d1.x_amax_epoch = xw, xe
            d2.x_amax, d2.x_amax_epoch, d2.out_amax, d2.out_amax_epoch = w1, e1, w2, e2
            d3.x_amax, d3.x_amax_epoch, d3.out_amax, d3.out_amax_epoch = w2, e2, w3, e3
        ops.L.check(ops.L.lib().abr_conv_run(self.ptr, n), "conv_run (bottleneck forward)")
        if self.h3:
            ops.amax_tag(out, w3, e3)
        if not self.save:
            return out, None
        if self.h3:
            ops.amax_tag(t1, w1, e1)
            ops.amax_tag(t2, w2, e2)
        return out, (x, t1, t2, out, self.s, v2)


def _numel(shape):
    n = 1
    for v in shape:
        n *= v
    return n


class Bottleneck(nn.Module):
    """resnet.py:242-346 (BottleneckWithFixedBatchNorm :371-395): 1x1(stride) -> 3x3 -> 1x1 (+identity / 1x1(stride) downsample) -> ReLU."""

    def __init__(self, in_channels, bottleneck_channels, out_channels, stride):
        super().__init__()
        self.downsample = None
        if in_channels != out_channels:
            self.downsample = nn.Sequential(Conv2d(in_channels, out_channels, 1, stride=stride, bias=False), FrozenBatchNorm2d(out_channels))
            self.downsample[0].kaiming_uniform_()
        self.conv1 = Conv2d(in_channels, bottleneck_channels, 1, stride=stride, bias=False)  # STRIDE_IN_1X1
        self.bn1 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, 3, stride=1, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(out_channels)
        for l in (self.conv1, self.conv2, self.conv3):
            l.kaiming_uniform_()
        self.stride = stride
        self.math = ops.MATH_F32   # ops.MATH_BF16: bf16 MFMA contractions (cfg.DTYPE == "bfloat16", see set_conv_math)

    def _conv(self, x, conv, stride, pad, **kw):
        """conv_forward in this block's arithmetic.  w_version lets the library keep what it derives from the weights (Winograd-domain
        weights; under bf16x6 the fragment-packed planes its weights-direct kernel reads)."""
        return ops.conv_forward(x, conv.weight, stride, pad, math=self.math, w_version=conv.version(), **kw)

    def _dgrad(self, g, conv, scale, pad, **kw):
        return ops.conv_forward(g, conv.dgrad_weight(scale), 1, pad, math=self.math, w_version=conv.version(), **kw)

    def prepare_derived(self):
        """Rebuild, on the current stream, everything this block derives from its trainable weights: the flipped / BN-scaled dgrad
        copies and, inside the library, what abr_conv_forward derives from the weight and from its dgrad copy (Winograd-domain weights of
        the 3x3 conv, packed bf16x3 planes under bf16x6).  FusedSGD.step runs this on the weight-preparation stream right after the update."""
        pairs = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        if self.downsample is not None:
            pairs.append((self.downsample[0], self.downsample[1]))
        for conv, bn in pairs:
            if not (conv.weight.requires_grad and conv.weight.is_cuda):
                continue
            wt = conv.dgrad_weight(bn.scale_bias()[0])
            ops.conv_prepare_weights(conv.weight, conv.stride, conv.padding, self.math, conv.version())
            # the dgrad conv is stride 1 with pad k-1-p (a scatter for the stride-2 1x1 convs): same derived data either way
            ops.conv_prepare_weights(wt, 1, conv.kernel_size - 1 - conv.padding, self.math, conv.version())

    def prep_entries(self):
        """(conv, FrozenBN scale, stride, pad, math) of the trainable convs: FusedSGD prepares them all in one batched call instead of
        prepare_derived()'s four launches per conv"""
        pairs = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        if self.downsample is not None:
            pairs.append((self.downsample[0], self.downsample[1]))
        return [(conv, bn.scale_bias()[0], conv.stride, conv.padding, self.math) for conv, bn in pairs
                if conv.weight.requires_grad and conv.weight.is_cuda]

    # x, returns NHWC tensors.  `stride` may be overridden to 1 when the caller already sub-sampled (bin_step=2 ROIAlign)
    def fwd(self, x, save, stride=None):
        s = self.stride if stride is None else stride
        if (BLOCK_PLANS and ops.H3_TAGS and x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
                and (save or not (self.math == ops.MATH_BF16X6 and ops.FUSE_TAIL64 and self.conv2.weight.shape[0] == 64))):
            plans = self.__dict__.setdefault("_fwd_plans", {})
            keep_v = bool(save and ops.KEEP_WINO_V and self.conv2.weight.requires_grad)
            key = (x.shape, s, self.math, save, keep_v)
            plan = plans.get(key)
            if plan is None or not plan.valid():
                if len(plans) > 32:       # (ragged batches: every image size brings its own tables)
                    plans.clear()
                plan = plans[key] = _FwdPlan(self, x.shape, s, save)
            return plan.run(x)
        s1, b1 = self.bn1.scale_bias()
        s2, b2 = self.bn2.scale_bias()
        s3, b3 = self.bn3.scale_bias()
        o1 = self._conv(x, self.conv1, s, 0, scale=s1, bias=b1, relu=True)
        # the weight gradient of conv2 transforms the same o1 with the same B^T d B: keep the forward's V for it when training
        v2 = ops.wino_v_alloc(o1, self.conv2.weight, 1, 1, self.math) if (save and self.conv2.weight.requires_grad) else None
        if self.downsample is not None:
            sd, bd = self.downsample[1].scale_bias()
            idt = self._conv(x, self.downsample[0], s, 0, scale=sd, bias=bd)
        else:
            idt = x
        if not save and ops.bottleneck_tail64_applies(o1, self.conv2.weight, self.conv3.weight, self.math):
            # no backward pass (the frozen layer1 of both models): conv2 -> bn2 -> relu -> conv3 -> bn3 -> += identity -> relu in one launch,
            # o2 stays on the compute unit (same products in the same order: bit-identical to the two launches below)
            out = ops.bottleneck_tail64(o1, self.conv2.weight, self.conv3.weight, s2, b2, s3, b3, idt, self.conv2.version(), self.conv3.version())
            return out, None
        o2 = self._conv(o1, self.conv2, 1, 1, scale=s2, bias=b2, relu=True, wino_v=v2)
        out = self._conv(o2, self.conv3, 1, 0, scale=s3, bias=b3, residual=idt, relu=True)
        return out, ((x, o1, o2, out, s, v2) if save else None)

    def bwd(self, saved, gout, need_dx, g_owned, g_masked=False, mask_dx=None):
        """gout = dL/d(out), not yet masked by out's ReLU unless g_masked.  Writes weight grads into .grad; returns dL/dx or None.
        mask_dx (the producer block's output, = this block's x): fuse THAT block's ReLU backward into the last dgrad launch."""
        x, o1, o2, out, s, v2 = saved
        s1, _ = self.bn1.scale_bias()
        s2, _ = self.bn2.scale_bias()
        s3, _ = self.bn3.scale_bias()
        g = gout if g_masked else ops.relu_backward(gout, out, inplace=g_owned)   # through the block's final ReLU
        if BLOCK_PLANS:
            return self._bwd_pairs(x, o1, o2, s, v2, g, need_dx, mask_dx, s1, s2, s3)
        ops.conv_wgrad_async(o2, g, _grad_buf(self.conv3.weight), 1, 0, scale=s3, math=self.math)
        g2 = self._dgrad(g, self.conv3, s3, 0, mask=o2)    # dgrad + ReLU mask of o2
        ops.conv_wgrad_async(o1, g2, _grad_buf(self.conv2.weight), 1, 1, scale=s2, math=self.math, wino_v=v2)
        g1 = self._dgrad(g2, self.conv2, s2, 1, mask=o1)   # 3x3 dgrad: pad = 3-1-1
        ops.conv_wgrad_async(x, g1, _grad_buf(self.conv1.weight), s, 0, scale=s1, math=self.math)
        ds = self.downsample
        if ds is not None:
            sd, _ = ds[1].scale_bias()
            ops.conv_wgrad_async(x, g, _grad_buf(ds[0].weight), s, 0, scale=sd, math=self.math)
        if not need_dx:
            return None
        if s == 1:
            if ds is not None:
                gx = self._dgrad(g, ds[0], sd, 0)
                return self._dgrad(g1, self.conv1, s1, 0, residual=gx, out=gx, mask=mask_dx)
            return self._dgrad(g1, self.conv1, s1, 0, residual=g, mask=mask_dx)
        # stride-2 1x1 convs: gradient rows land on the even pixels of a zeroed tensor
        B, H, W, _ = x.shape
        assert mask_dx is None, "stride-2 blocks open a stage: their input is not a block output of the same stage"
        gx = self._dgrad(g1, self.conv1, s1, 0, out_hw=(H, W), out_stride=(s, s))
        if ds is not None:
            self._dgrad(g, ds[0], sd, 0, residual=gx, out=gx, out_hw=(H, W), out_stride=(s, s))
        return gx


def _bwd_pairs(self, x, o1, o2, s, v2, g, need_dx, mask_dx, s1, s2, s3):
    """Bottleneck.bwd with ONE library call per conv (ops.conv_backward: the weight gradient on its side stream + the input gradient on this one):
    the same launches with the same arguments as the per-call form; on every stream the same order except that a downsample branch's weight
    gradient now precedes conv1's (independent buffers)."""
    m = self.math
    c1, c2, c3 = self.conv1, self.conv2, self.conv3
    g2 = ops.conv_backward(o2, g, _grad_buf(c3.weight), c3.dgrad_weight(s3), 1, 0, scale=s3, math=m, w_version=c3.version(), mask=o2)
    g1 = ops.conv_backward(o1, g2, _grad_buf(c2.weight), c2.dgrad_weight(s2), 1, 1, scale=s2, math=m, w_version=c2.version(), wino_v=v2, mask=o1)
    ds = self.downsample
    sd = ds[1].scale_bias()[0] if ds is not None else None
    if not need_dx:
        ops.conv_backward(x, g1, _grad_buf(c1.weight), None, s, 0, scale=s1, math=m, dgrad=False)
        if ds is not None:
            ops.conv_backward(x, g, _grad_buf(ds[0].weight), None, s, 0, scale=sd, math=m, dgrad=False)
        return None
    if s == 1:
        if ds is not None:
            gx = ops.conv_backward(x, g, _grad_buf(ds[0].weight), ds[0].dgrad_weight(sd), s, 0, scale=sd, math=m, w_version=ds[0].version())
            return ops.conv_backward(x, g1, _grad_buf(c1.weight), c1.dgrad_weight(s1), s, 0, scale=s1, math=m, w_version=c1.version(),
                                     residual=gx, out=gx, mask=mask_dx)
        return ops.conv_backward(x, g1, _grad_buf(c1.weight), c1.dgrad_weight(s1), s, 0, scale=s1, math=m, w_version=c1.version(),
                                 residual=g, mask=mask_dx)
    # stride-2 1x1 convs: gradient rows land on the even pixels of a zeroed tensor
    B, H, W, _ = x.shape
    assert mask_dx is None, "stride-2 blocks open a stage: their input is not a block output of the same stage"
    gx = ops.conv_backward(x, g1, _grad_buf(c1.weight), c1.dgrad_weight(s1), s, 0, scale=s1, math=m, w_version=c1.version(),
                           out_hw=(H, W), out_stride=(s, s))
    if ds is not None:
        ops.conv_backward(x, g, _grad_buf(ds[0].weight), ds[0].dgrad_weight(sd), s, 0, scale=sd, math=m, w_version=ds[0].version(),
                          residual=gx, out=gx, out_hw=(H, W), out_stride=(s, s))
    return gx


Bottleneck._bwd_pairs = _bwd_pairs


class _StageFn(Function):
    """A run of bottlenecks as one autograd node.  Weight gradients are accumulated straight into p.grad
    (views of the flat gradient buffer) by the wgrad kernel, so backward returns None for them."""

    @staticmethod
    def forward(ctx, x, blocks, first_stride, need_dx, *params):
        ctx.blocks, ctx.need_dx = blocks, need_dx
        ctx.saved = []
        save = True  # run_stage only takes this path when a gradient is wanted (grad mode is off INSIDE Function.forward)
        h = as_nhwc(x)
        for i, blk in enumerate(blocks):
            h, s = blk.fwd(h, save, stride=first_stride if i == 0 else None)
            ctx.saved.append(s)
        return from_nhwc(h)

    @staticmethod
    def backward(ctx, gout):
        g = as_nhwc(gout)
        owned = False
        # the producer of gout (the box predictor's fused pooling + ReLU backward) already applied this stage's final ReLU mask -- valid only while
        # gout is still exactly the tensor that producer wrote (same storage, same version: nothing was accumulated into it since)
        masked = getattr(gout, "_abr_relu_masked", None) == (gout.data_ptr(), gout._version)
        n = len(ctx.blocks)
        for i in range(n - 1, -1, -1):
            need = ctx.need_dx or i > 0
            # block i's input x is block i-1's (post-ReLU) output: its ReLU backward rides in block i's last dgrad epilogue
            fuse = i > 0 and ctx.saved[i][4] == 1
            g = ctx.blocks[i].bwd(ctx.saved[i], g, need, owned, masked, ctx.saved[i][0] if fuse else None)
            ctx.saved[i] = None
            owned, masked = True, fuse
        return (from_nhwc(g) if g is not None else None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


def _stage_params(blocks):
    """the parameters of a stage's blocks, in module order.  Walking the module tree (nn.Module.parameters) costs ~40 us per stage call and the
    step makes ~25 of them; a block's parameter OBJECTS never change (flatten_parameters re-homes their .data), so the list is kept on the
    first block, keyed by the identity of the blocks it was built for"""
    if not blocks:
        return []
    key = tuple(id(b) for b in blocks)
    cached = blocks[0].__dict__.get("_stage_params_cache")
    if cached is not None and cached[0] == key:
        # a replaced parameter OBJECT (module.weight = nn.Parameter(...), a swapped sub-module) must not leave a stale list behind: the blocks' direct
        # conv weights are checked by identity (cheap: 3-4 per block), anything else invalidates through flatten_parameters / _apply
        ws = cached[2]
        if all(c.weight is w for c, w in ws):
            return cached[1]
    plist = [p for b in blocks for p in b.parameters()]
    ws = [(m, m.weight) for b in blocks for m in b.modules() if isinstance(m, Conv2d)]
    blocks[0].__dict__["_stage_params_cache"] = (key, plist, ws)
    return plist


def run_stage(x, blocks, first_stride=None, need_dx=True):
    """x logical [B,C,H,W] -> logical output; differentiable when any block parameter requires grad."""
    params = _stage_params(blocks)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        out = _StageFn.apply(x, blocks, first_stride, need_dx and x.requires_grad, *params)
        out._abr_relu_output = True      # (a bottleneck ends in a ReLU: consumers may fuse its backward into theirs, see _PredictorFn)
        return out
    h = as_nhwc(x)
    for i, blk in enumerate(blocks):
        h, _ = blk.fwd(h, False, stride=first_stride if i == 0 else None)
    return from_nhwc(h)


def set_conv_math(module, math):
    """Select the contraction arithmetic (ops.MATH_F32 / ops.MATH_BF16) of every bottleneck / conv head under `module`."""
    n = 0
    for m in module.modules():
        if hasattr(m, "math"):
            m.math = math
            n += 1
    return n


def _make_stage(in_channels, bottleneck_channels, out_channels, block_count, first_stride):
    blocks, stride = [], first_stride
    for _ in range(block_count):
        blocks.append(Bottleneck(in_channels, bottleneck_channels, out_channels, stride))
        stride, in_channels = 1, out_channels
    return nn.Sequential(*blocks)


class StemWithFixedBatchNorm(nn.Module):
    """BaseStem (resnet.py:349-368): conv7x7 s2 p3 (3->64) -> FrozenBN -> ReLU -> maxpool 3x3 s2 p1."""

    def __init__(self, cfg):
        super().__init__()
        out_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        self.conv1 = Conv2d(3, out_channels, 7, stride=2, padding=3, bias=False, cin_pad=4)
        self.bn1 = FrozenBatchNorm2d(out_channels)
        self.conv1.kaiming_uniform_()

    def forward(self, x):
        """x: [B,3,H,W] plain NCHW image batch -> logical [B,64,H/4,W/4]"""
        xh = ops.nchw_to_nhwc(x.contiguous(), cpad=4) if x.shape[1] == 3 else as_nhwc(x)
        s, b = self.bn1.scale_bias()
        # (the stem runs on the fp32 MFMA kernel; its epilogue still writes the output's amax word for layer1's f16x3 convs: the pooled tensor
        #  inherits it as a bound instead of being reduced again)
        y = ops.conv_forward(xh, self.conv1.weight, 2, 3, scale=s, bias=b, relu=True, emit_amax=True)
        return from_nhwc(ops.maxpool3x3s2(y))


class ResNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        assert cfg.MODEL.BACKBONE.CONV_BODY == "R-50-C4", "only the R-50-C4 body is on the hot path (every configs/voc YAML)"
        assert cfg.MODEL.RESNETS.STRIDE_IN_1X1 and cfg.MODEL.RESNETS.NUM_GROUPS == 1
        self.stem = StemWithFixedBatchNorm(cfg)
        width = cfg.MODEL.RESNETS.NUM_GROUPS * cfg.MODEL.RESNETS.WIDTH_PER_GROUP
        in_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        out2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
        self.stages, self.return_features = [], {}
        for spec in ResNet50StagesTo4:
            name = "layer" + str(spec.index)
            f = 2 ** (spec.index - 1)
            self.add_module(name, _make_stage(in_channels, width * f, out2 * f, spec.block_count, int(spec.index > 1) + 1))
            in_channels = out2 * f
            self.stages.append(name)
            self.return_features[name] = spec.return_features
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)
        if cfg.DTYPE == "bfloat16":   # BASELINE.json configs[4]: "bf16 MFMA backbone" (fp32 tensors, fp32 accumulate; the stem stays fp32)
            set_conv_math(self, ops.MATH_BF16)
        elif cfg.DTYPE != "float32":
            raise NotImplementedError("DTYPE {!r}: this build computes in float32 or with a bfloat16 MFMA backbone "
                                      "(the reference's float16 = apex amp O1, train_incremental.py:194)".format(cfg.DTYPE))

    def _freeze_backbone(self, freeze_at):
        for stage_index in range(max(freeze_at, 0)):
            m = self.stem if stage_index == 0 else getattr(self, "layer" + str(stage_index))
            for p in m.parameters():
                p.requires_grad = False

    def _trains(self, name):
        """does stage `name` ("stem", "layer1", ...) hold a trainable parameter?  (parameter lists kept: see _stage_params)"""
        cache = self.__dict__.setdefault("_stage_param_lists", {})
        ent = cache.get(name)
        mod = getattr(self, name)
        if ent is None or ent[0] is not mod or not all(c.weight is w for c, w in ent[2]):   # (rebuilt when a stage module or a conv weight object is replaced)
            ent = cache[name] = (mod, list(mod.parameters()), [(m, m.weight) for m in mod.modules() if isinstance(m, Conv2d)])
        return any(p.requires_grad for p in ent[1])

    def frozen_stage_names(self):
        """names of the leading stages without trainable parameters (what frozen_prefix computes after the stem), or None when the stem trains"""
        if self._trains("stem"):
            return None
        names = []
        for name in self.stages:
            if self._trains(name):
                break
            names.append(name)
        return names

    def frozen_prefix(self, x):
        """The stem and the leading stages whose parameters are all frozen (FREEZE_CONV_BODY_AT = 2: stem + layer1), run without autograd:
        their outputs do not depend on the optimiser, so the trainer may compute them for the NEXT batch while the current backward pass
        runs.  Returns (x, stage outputs so far) or None when the stem itself trains."""
        if self._trains("stem"):
            return None
        with torch.no_grad():
            x = self.stem(x)
            done = []
            for name in self.stages:
                if self._trains(name):
                    break
                x = run_stage(x, list(getattr(self, name)))
                done.append(x)
        return x, done

    def forward(self, x, prefix=None):
        """`prefix`: the result of frozen_prefix() on the same input (the same values the inline path computes)"""
        if torch.is_grad_enabled() and x.requires_grad and self.frozen_prefix is not None and not self._trains("stem"):
            # the frozen stem / stages run without autograd (their kernels have no backward): a gradient w.r.t. the IMAGE would silently be
            # dropped.  The reference's training loop never asks for one (images come from the data loader); fail loudly instead.
            raise NotImplementedError("gradients with respect to the input image are not implemented: the frozen stem (FREEZE_CONV_BODY_AT >= 1) "
                                      "has no backward pass")
        outputs, backbone_features = [], []
        if prefix is None:
            prefix = self.frozen_prefix(x)
        if prefix is None:
            x, done = self.stem(x), []
        else:
            x, done = prefix
        for i, name in enumerate(self.stages):
            x = done[i] if i < len(done) else run_stage(x, list(getattr(self, name)))
            if i >= len(done) and torch.is_grad_enabled():
                ops.mark("target " + name + " forward done")
            if self.return_features[name]:
                outputs.append(x)
            backbone_features.append(x)
        return outputs, backbone_features


class ResNetHead(nn.Module):
    """layer4 on pooled RoI features (resnet.py:158-207).  `first_stride=1` is used when ROIAlign already produced only
    the even bins (bin_step=2): a stride-2 1x1 conv over a 7x7 map reads exactly bins (0,2,4,6)^2."""

    def __init__(self, stage_index=4, block_count=3, width_per_group=64, res2_out_channels=256, stride_init=None):
        super().__init__()
        f = 2 ** (stage_index - 1)
        out_channels = res2_out_channels * f
        stride = stride_init or (int(stage_index > 1) + 1)
        self.layer4 = _make_stage(out_channels // 2, width_per_group * f, out_channels, block_count, stride)
        self.stages = ["layer4"]
        self.out_channels = out_channels

    def forward(self, x, first_stride=None):
        return run_stage(x, list(self.layer4), first_stride=first_stride)
