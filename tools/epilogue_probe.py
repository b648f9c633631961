#!/usr/bin/env python3
"""How much of a layer4 1x1 GEMM is its epilogue?  The same M x N GEMM at several K, with and without the fused epilogue inputs (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n

M = 32768
for N, Ks in ((2048, (128, 256, 512, 1024, 2048)), (512, (512, 1024, 2048, 4096))):
    for K in Ks:
        x = torch.randn(M // 16, 4, 4, K, device="cuda"); w = torch.randn(N, 1, 1, K, device="cuda") * 0.05
        sc = torch.rand(N, device="cuda") + 0.5; bi = torch.randn(N, device="cuda"); res = torch.randn(M // 16, 4, 4, N, device="cuda")
        out = torch.empty(M // 16, 4, 4, N, device="cuda")
        fl = 2.0 * M * N * K
        row = []
        for name, kw in (("plain", {}), ("bn+relu", dict(scale=sc, bias=bi, relu=True)), ("bn+res+relu", dict(scale=sc, bias=bi, relu=True, residual=res)),
                         ("mask", dict(mask=res))):
            t = timeit(lambda: ops.conv_forward(x, w, 1, 0, math=ops.MATH_BF16X6, w_version=77, out=out, **kw))
            row.append(f"{name} {t:.3f} ms {fl / t / 1e9:6.1f} TF")
        print(f"M {M} N {N:5d} K {K:5d} | " + " | ".join(row), flush=True)
