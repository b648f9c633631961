#!/usr/bin/env python3
"""Fixed cost vs per-stage cost of the weight-gradient kernel on the small-output shapes of layer2/3: time(M) at fixed (Cout, K).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from abr_iod_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

mth = {"f32": ops.MATH_F32, "bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[os.environ.get("PROBE_MATH", "bf16x6")]
print(f"{'M':>7s} {'Cout':>5s} {'K':>5s} | {'us':>7s} {'TF/s':>6s}")
for Cout, K in ((1024, 256), (256, 1024), (512, 128), (128, 512), (2048, 512)):
    for M in (2048, 4096, 8192, 9576, 16384, 32768, 65536, 131072):
        x = torch.randn(1, 1, M, K, device="cuda")
        gy = torch.randn(1, 1, M, Cout, device="cuda")
        dw = torch.zeros(Cout, 1, 1, K, device="cuda")
        t = timeit(lambda: ops.conv_wgrad(x, gy, dw, 1, 0, math=mth), iters=20) * 1e3
        print(f"{M:7d} {Cout:5d} {K:5d} | {t:7.1f} {2.0 * M * Cout * K / t / 1e6:6.1f}", flush=True)
