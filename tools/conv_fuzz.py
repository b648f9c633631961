#!/usr/bin/env python3
"""Random-shape sweep of the conv engine against float64: forward (with a random choice of the fused epilogue: BN scale / bias / residual /
ReLU), input gradient and weight gradient, in the three arithmetics, at shapes the model never uses (odd extents, channel counts that are
not tile multiples, 1-pixel maps, strides).  A configuration the library does not support must raise (counted, listed), never return wrong
values.  GPU box: python tools/conv_fuzz.py [--cases 300] [--seed 0]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from abr_iod_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
MODES = [("f32", ops.MATH_F32, 1e-4), ("bf16x6", ops.MATH_BF16X6, 1e-4), ("f16x3", ops.MATH_F16X3, 1e-4), ("bf16", ops.MATH_BF16, 3e-2)]
CH = [4, 8, 12, 16, 20, 32, 36, 48, 64, 76, 96, 100, 128, 160, 192, 256, 320, 512]
fails, refused, ran = [], {}, 0


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def check(tag, case, got, want, tol):
    global ran
    ran += 1
    err = (got.detach().cpu().double() - want).abs().max().item()
    lim = tol * max(1.0, want.abs().max().item())
    if not (err <= lim) or not bool(torch.isfinite(got).all()):
        fails.append((tag, case, err, lim))


def attempt(tag, case, fn):
    try:
        return fn()
    except RuntimeError as e:
        refused.setdefault((tag, str(e)[:90]), []).append(case)
        return None


for ci in range(a.cases):
    k = rng.choice([1, 1, 3, 3, 7])
    Cin = 4 if k == 7 else rng.choice(CH)
    Cout = rng.choice(CH + [1, 3, 5, 21, 33, 108])
    s = rng.choice([1, 1, 2])
    p = rng.choice([0, (k - 1) // 2]) if k > 1 else 0
    B = rng.randint(1, 5)
    H, W = rng.randint(max(1, k - 2 * p), 41), rng.randint(max(1, k - 2 * p), 41)
    if (Cin * k * k * Cout * H * W * B) > 6e9:      # keep the float64 CPU reference in seconds
        B, H, W = 1, min(H, 16), min(W, 16)
        H, W = max(H, k - 2 * p), max(W, k - 2 * p)
    case = (B, Cin, H, W, Cout, k, s, p)
    g = torch.Generator().manual_seed(a.seed * 100003 + ci)
    x = torch.randn(B, Cin, H, W, generator=g).float().double().requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).float().double().requires_grad_(True)
    scale = (torch.rand(Cout, generator=g) + 0.5).float()
    bias = (torch.randn(Cout, generator=g) * 0.1).float()
    y = F.conv2d(x, w, stride=s, padding=p)
    res = torch.randn(y.shape, generator=g).float()
    gy = torch.randn(y.shape, generator=g).float().double()
    (y * scale.double().view(1, -1, 1, 1)).backward(gy)
    use = dict(scale=rng.random() < 0.7, bias=rng.random() < 0.5, residual=rng.random() < 0.4, relu=rng.random() < 0.6)
    yr = y.detach()
    if use["scale"]:
        yr = yr * scale.double().view(1, -1, 1, 1)
    if use["bias"]:
        yr = yr + bias.double().view(1, -1, 1, 1)
    if use["residual"]:
        yr = yr + res.double()
    if use["relu"]:
        yr = torch.relu(yr)
    xg, wg, gyg = nhwc(x.detach()), nhwc(w.detach()), nhwc(gy)
    kw = dict(scale=scale.cuda() if use["scale"] else None, bias=bias.cuda() if use["bias"] else None,
              residual=nhwc(res) if use["residual"] else None, relu=use["relu"])
    for name, m, tol in MODES:
        if m != ops.MATH_F32 and k == 7:
            continue
        got = attempt("fwd/" + name, case, lambda: ops.conv_forward(xg, wg, s, p, math=m, **kw))
        if got is not None:
            check("fwd/" + name, case + (tuple(sorted(k_ for k_, v in use.items() if v)),), got.permute(0, 3, 1, 2), yr, tol)
        if k == 7:
            continue      # the stem is frozen: no backward on the path
        dw = torch.zeros_like(wg)
        if attempt("wgrad/" + name, case, lambda: (ops.conv_wgrad(xg, gyg, dw, s, p, scale=scale.cuda(), math=m), True)[1]):
            check("wgrad/" + name, case, dw.permute(0, 3, 1, 2), w.grad, tol * 2)
        if s == 1 or k == 1:
            def dgrad():
                wt = ops.conv_dgrad_weights(wg, scale.cuda())
                if s == 1:
                    return ops.conv_forward(gyg, wt, 1, k - 1 - p, math=m)
                return ops.conv_forward(gyg, wt, 1, 0, out_hw=(H, W), out_stride=(s, s), math=m)
            dx = attempt("dgrad/" + name, case, dgrad)
            if dx is not None:
                check("dgrad/" + name, case, dx.permute(0, 3, 1, 2), x.grad, tol * 2)
    if (ci + 1) % 50 == 0:
        print("%d cases, %d comparisons, %d failures, %d refusals" % (ci + 1, ran, len(fails), sum(len(v) for v in refused.values())), flush=True)

print("\n%d cases (B, Cin, H, W, Cout, k, stride, pad), %d comparisons against float64; tolerances: f32 / bf16x6 1e-4 (2e-4 gradients), bf16 3e-2 of the output scale" % (a.cases, ran))
print("refused (RuntimeError) configurations:")
for (tag, msg), cs in sorted(refused.items()):
    print("  %-14s %4d x  %s   e.g. %s" % (tag, len(cs), msg, cs[0]))
print("FAILURES: %d" % len(fails))
for f in fails[:40]:
    print("  ", f)
sys.exit(1 if fails else 0)
