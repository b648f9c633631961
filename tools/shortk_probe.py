#!/usr/bin/env python3
"""Fixed cost vs per-stage cost of the 1x1 forward kernel: time(M, N, K) at grids of exactly r rounds of workgroups.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from abr_iod_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

mth = {"f32": ops.MATH_F32, "bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[os.environ.get("PROBE_MATH", "bf16x6")]
print(f"{'M':>7s} {'N':>5s} {'K':>5s} {'WGs':>5s} | {'us':>7s} {'TF/s':>6s} {'GB/s':>6s}")
for N in (64, 256, 1024):
    for K in (64, 128, 256, 512, 1024):
        for wgs in (128, 256, 512, 1024, 2048):
            mt = wgs // max(1, N // 128)
            M = mt * 128
            x = torch.randn(1, 1, M, K, device="cuda")
            w = torch.randn(N, 1, 1, K, device="cuda") * 0.05
            sc = torch.rand(N, device="cuda") + 0.5
            bi = torch.randn(N, device="cuda")
            t = timeit(lambda: ops.conv_forward(x, w, 1, 0, scale=sc, bias=bi, relu=True, math=mth), iters=20) * 1e3
            by = 4.0 * (M * K + M * N + N * K)
            print(f"{M:7d} {N:5d} {K:5d} {wgs:5d} | {t:7.1f} {2.0 * M * N * K / t / 1e6:6.1f} {by / t / 1e3:6.0f}", flush=True)
