#!/usr/bin/env python3
"""f16x3's domain statistic on a TRAINED-LIKE weight set: every test and the benchmark run seeded random-init weights whose
FrozenBN scales sit in [0.5, 1.5]; a trained detector's do not.  Here the FrozenBN scales (weight / sqrt(var)) of both models are redrawn log-normally
over ~1e-3 ... 1e2 (sigma = 1.6 in natural-log units, clipped, then normalised to RMS 1 per layer so that the network stays finite), 2 % of the channels of every layer are made near-dead (scale 1e-6), conv weights get
heavy-tailed per-output-channel norms (log-normal, sigma = 0.7), and a few full-size ARD steps are run with the per-conv calls (ABR_BLOCK_PLANS=0) so that the
share of operand elements more than 18 binades below their tensor's amax can be read back PER LAUNCH SITE: which layers, forward or backward, approach the
trainer's 5 % limit (engine/trainer.py::H3_MAX_SMALL_FRACTION), and what the whole step's fraction is -- the number the trainer's guard acts on.

GPU box:  python tools/h3_trained_like_stats.py [--steps 3] [--sigma 1.6] [--dead 0.02]"""
import argparse
import collections
import os
import sys

os.environ["ABR_BLOCK_PLANS"] = "0"     # per-conv library calls: one statistic per conv launch site
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from abr_iod_amd import ops  # noqa: E402
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.layers import FrozenBatchNorm2d  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--sigma", type=float, default=1.6, help="log-normal spread of the FrozenBN scales (natural-log units); 0 = the benchmark's own weights")
ap.add_argument("--dead", type=float, default=0.02, help="share of near-dead channels (scale 1e-6) per FrozenBN layer")
ap.add_argument("--wsigma", type=float, default=0.7, help="log-normal spread of the conv weights' per-output-channel norms")
a = ap.parse_args()

cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=4)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
g = torch.Generator().manual_seed(1234)


def trained_like(model, gen):
    """the same draw for the layers the two models share (the target starts from the source's checkpoint)"""
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, FrozenBatchNorm2d) and a.sigma > 0:
                n = m.weight.numel()
                ln = torch.randn(n, generator=gen) * a.sigma
                sc = torch.exp(ln - ln.mean()).clamp(1e-3, 1e2)
                dead = torch.rand(n, generator=gen) < a.dead
                sc[dead] = 1e-6
                sc = sc / sc.pow(2).mean().sqrt()                    # RMS 1: the layer's gain stays O(1) (a geometric mean of 1 multiplies the
                                                                     # activations by e^(sigma^2) per conv: inf after a dozen blocks)
                cur = (m.weight * m.running_var.rsqrt()).cpu()            # the fused scale as it is (random init: U[.5, 1.5] x damping)
                m.weight.copy_((sc * float(cur.abs().mean()) * m.running_var.cpu().sqrt()).to(m.weight.device))   # fused scale := sc x its mean
                m.invalidate()
        if a.wsigma > 0:
            for name, p in model.named_parameters():
                if p.dim() == 4 and p.shape[0] >= 16:
                    f = torch.exp(torch.randn(p.shape[0], generator=gen) * a.wsigma)
                    p.mul_((f / f.pow(2).mean().sqrt()).view(-1, 1, 1, 1).to(p.device))


gs, gt = torch.Generator().manual_seed(1234), torch.Generator().manual_seed(1234)
trained_like(ms, gs)
trained_like(mt, gt)
from abr_iod_amd.modeling.backbone.resnet import bump_param_version  # noqa: E402
bump_param_version()
opt = make_optimizer(cfg_t, mt)
sch = make_lr_scheduler(cfg_t, opt)
for gr in opt.param_groups:
    gr["lr"] = 0.0          # statistics of ONE weight set: no update between the steps
images, targets = synthetic_batch(4, 600, 1000, seed=42, label_range=(16, 21))

site = collections.OrderedDict()
_fwd, _bwd, _wg = ops.conv_forward, ops.conv_backward, ops.conv_wgrad


def _note(kind, x_shape, w_shape, stride):
    small, seen = ops.h3_range_stats(reset=True)      # (synchronises: the launch has reported)
    if seen:
        k = (kind, tuple(x_shape), tuple(w_shape), stride)
        s = site.setdefault(k, [0, 0, 0])
        s[0] += small
        s[1] += seen
        s[2] += 1


def fwd(*args, **kw):
    ops.h3_range_stats(reset=True)
    out = _fwd(*args, **kw)
    if kw.get("math") == ops.MATH_F16X3:
        _note("fwd/dgrad", args[0].shape, args[1].shape, args[2] if len(args) > 2 else kw.get("stride", 1))
    return out


def wg(*args, **kw):        # conv_wgrad(x, gy, dw, stride, pad, scale, math, wino_v, ...)
    ops.h3_range_stats(reset=True)
    out = _wg(*args, **kw)
    math = args[6] if len(args) > 6 else kw.get("math")
    if math == ops.MATH_F16X3:
        _note("wgrad", args[0].shape, args[2].shape, args[3] if len(args) > 3 else kw.get("stride", 1))
    return out


ops.conv_forward, ops.conv_wgrad = fwd, wg
ops.WGRAD_SIDE_STREAM = False       # weight gradients on the calling stream, through conv_wgrad: attributable
tot = [0, 0]
for it in range(a.steps):
    ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
torch.cuda.synchronize()
print("trained-like weights: FrozenBN scale sigma %.2f (clipped to 1e-3..1e2), %.0f %% near-dead channels, conv row-norm sigma %.2f; %d steps, losses %s" % (
    a.sigma, 100 * a.dead, a.wsigma, a.steps, {k: round(float(v), 4) for k, v in ld.items()}))
print("range-guard flags of the run: %d (2 = an inf / nan operand was seen: the trainer then leaves the split arithmetics, and the rows below cover the steps before it)" % ops.x6_range_flags(reset=False))
print("conv math at the end: %s (the trainer's guard switches at %.0f %%)" % (getattr(mt, "conv_math", None), 100 * float(os.environ.get("ABR_H3_MAX_SMALL_FRACTION", "0.05"))))
print("%-10s %-26s %-22s %3s %8s %14s %10s" % ("kind", "input", "weight", "s", "launches", "inspected", "small %"))
rows = sorted(site.items(), key=lambda kv: -(kv[1][0] / max(kv[1][1], 1)))
for (kind, xs, ws, st), (small, seen, n) in rows:
    tot[0] += small
    tot[1] += seen
    print("%-10s %-26s %-22s %3d %8d %14d %10.4f" % (kind, "x".join(map(str, xs)), "x".join(map(str, ws)), st, n, seen, 100.0 * small / max(seen, 1)))
print("whole run: %.4f %% of %d inspected operand elements are more than 18 binades below their tensor's amax" % (100.0 * tot[0] / max(tot[1], 1), tot[1]))
