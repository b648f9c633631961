#!/usr/bin/env python3
"""Which tile instance suits a conv shape?  Run under ABR_X6_TILE=1|2|3 (forces 128x128 | 128x64 | 64x64 for K <= ABR_X6_TILE_MAXK) -- GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd import ops

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n

SHAPES = [(4, 150, 250, 64, 256, "res"), (4, 150, 250, 256, 64, ""), (4, 150, 250, 64, 64, ""), (4, 75, 125, 128, 512, "res"), (4, 75, 125, 512, 128, ""),
          (4, 75, 125, 256, 128, ""), (4, 38, 63, 256, 1024, "res"), (4, 38, 63, 1024, 256, ""), (4, 38, 63, 512, 256, ""), (4, 38, 63, 1024, 76, "")]
MATH = {"bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[os.environ.get("ABR_PROBE_MATH", "f16x3")]
row = []
for B, H, W, Cin, Cout, ep in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; bi = torch.randn(Cout, device="cuda")
    res = torch.randn(B, H, W, Cout, device="cuda") if ep else None
    out = torch.empty(B, H, W, Cout, device="cuda")
    t = timeit(lambda: ops.conv_forward(x, w, 1, 0, scale=sc, bias=bi, relu=True, residual=res, math=MATH, w_version=5, out=out))
    row.append(f"{B*H*W}x{Cout}x{Cin}:{t*1e3:.0f}us")
print(os.environ.get("ABR_X6_TILE", "auto"), " ".join(row))
