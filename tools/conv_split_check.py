#!/usr/bin/env python3
"""Compute a few convolutions whose grids trigger the split-K plan of abr_conv_forward and save the outputs (tests/test_gpu_ops.py
runs this twice, with ABR_IGEMM_SPLIT=0 and =1, and compares).  GPU box only.  usage: conv_split_check.py out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd import ops  # noqa: E402

CASES = [  # B, H, W, Cin, Cout, k, pad, with residual/mask
    (4, 38, 63, 1024, 1024, 3, 1, "bias"),      # RPN 3x3: 600 tiles of 128x128 -> the last 88 split by 2
    (4, 38, 63, 1024, 1024, 3, 1, "relu"),
    (4, 40, 63, 1024, 768, 3, 1, "mask"),       # 79 x 6 = 474 tiles -> the last 218 split
    (5, 38, 63, 1024, 512, 3, 1, "residual"),   # 94 x 4 = 376 tiles -> the last 120 split by 2
]


def main():
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for n, (B, H, W, Cin, Cout, k, pad, mode) in enumerate(CASES):
        x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * 0.05
        sc = torch.rand(Cout, device="cuda", generator=g) + 0.5
        bi = torch.randn(Cout, device="cuda", generator=g)
        kw = dict(scale=sc, bias=bi)
        if mode == "relu":
            kw["relu"] = True
        elif mode == "mask":
            kw["mask"] = torch.randn(B, H, W, Cout, device="cuda", generator=g)
        elif mode == "residual":
            kw["residual"] = torch.randn(B, H, W, Cout, device="cuda", generator=g)
            kw["relu"] = True
        y1 = ops.conv_forward(x, w, 1, pad, **kw)
        for _ in range(6):   # stale partials (a missed coherence bit) or an order-dependent sum would show up as run-to-run drift
            y2 = ops.conv_forward(x, w, 1, pad, **kw)
            assert torch.equal(y1, y2), "split-K must be deterministic run to run"
        torch.cuda.synchronize()
        out["y%d" % n] = y1.cpu().numpy()
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
