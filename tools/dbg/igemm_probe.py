#!/usr/bin/env python3
"""Start-up skew of the weights-direct GEMM launches (ConvP::stagger_*, ABR_IGEMM_STAGGER = cycles per k-tile and slot): per-shape time of the step's large
launches with the epilogue they carry in the step, for a list of skews.  GPU box: python tools/dbg/igemm_probe.py [name=ENV:VALUE,ENV:VALUE ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from abr_iod_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


# (M as B x H x W, Cin, Cout, epilogue): layer4 conv3 (residual + ReLU), conv1 (ReLU), downsample (plain), conv1's dgrad (residual + mask);
# layer3 / layer2 / layer1 expand + reduce convs
SHAPES = [((2304, 4, 4), 512, 2048, "res+relu"), ((2304, 4, 4), 2048, 512, "relu"), ((2304, 4, 4), 1024, 2048, ""), ((2304, 4, 4), 512, 2048, "res+mask"),
          ((2304, 4, 4), 512, 1024, "res+mask"), ((2304, 4, 4), 1024, 512, "relu"),
          ((4, 38, 63), 256, 1024, "res+relu"), ((4, 38, 63), 1024, 256, "relu"), ((4, 75, 125), 128, 512, "res+relu"), ((4, 75, 125), 512, 128, "relu"),
          ((4, 150, 250), 64, 256, "res+relu"), ((4, 150, 250), 256, 64, "relu")]
# variants: name=ENV:VALUE,ENV:VALUE ...   (the library reads these switches at every launch)
VARIANTS = [("base", {})]
if len(sys.argv) > 1:
    VARIANTS = [("base", {})] + [(a.split("=")[0], dict(kv.split(":") for kv in a.split("=")[1].split(","))) for a in sys.argv[1:]]
KEYS = sorted({k for _, e in VARIANTS for k in e})
MATH = ops.MATH_F16X3
g = torch.Generator(device="cuda").manual_seed(0)
cases = []
for (B, H, W), Cin, Cout, ep in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 1, 1, Cin, device="cuda", generator=g) * 0.05
    sc = torch.rand(Cout, device="cuda", generator=g) + 0.5
    bi = torch.randn(Cout, device="cuda", generator=g)
    res = torch.randn(B, H, W, Cout, device="cuda", generator=g) if "res" in ep else None
    mask = torch.randn(B, H, W, Cout, device="cuda", generator=g) if "mask" in ep else None
    out = torch.empty(B, H, W, Cout, device="cuda")
    ops.amax_compute(x)
    cases.append((x, w, sc, bi, res, mask, out, "relu" in ep, "%dx%dx%d %s" % (B * H * W, Cout, Cin, ep)))
print("%-34s" % "shape (M x N x K, epilogue)" + "".join("%14s" % n for n, _ in VARIANTS) + "   us per launch (best of 3 interleaved rounds); results identical: checked")
for x, w, sc, bi, res, mask, out, relu, name in cases:
    best = [1e9] * len(VARIANTS)
    ref = None
    fn = lambda: ops.conv_forward(x, w, 1, 0, scale=sc, bias=bi, relu=relu, residual=res, mask=mask, math=MATH, w_version=7, out=out, emit_amax=False)
    for rnd in range(3):
        for vi, (_, env) in enumerate(VARIANTS):
            for k in KEYS:
                os.environ[k] = env.get(k, "0")
            t = timeit(fn)
            best[vi] = min(best[vi], t * 1e3)
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(out, ref), (name, VARIANTS[vi][0])
    print("%-34s" % name + "".join("%14.1f" % v for v in best))
for k in KEYS:
    os.environ[k] = "0"
