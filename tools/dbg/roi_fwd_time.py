#!/usr/bin/env python3
"""ROIAlign forward: the register-reuse kernel (roi_align_fwd_nhwc_reuse, default) against the plain tap loop (ABR_ROIALIGN_REUSE=0), same process,
interleaved; outputs compared bit for bit.  RoI sets: the microbenchmark's (tools/microbench.py: widths / heights 32 + U(0,400) px), small RoIs
(<= 7 cells: one sample per bin -- nothing to reuse), large RoIs (21-38 cells).  GPU box: python tools/dbg/roi_fwd_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from abr_iod_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
B, H, W, C = 4, 38, 63, 1024
feat = torch.randn(B, H, W, C, device=dev, generator=g)


def rois_of(K, lo, hi):
    x1 = torch.rand(K, device=dev, generator=g) * 600
    y1 = torch.rand(K, device=dev, generator=g) * 300
    w = lo + torch.rand(K, device=dev, generator=g) * (hi - lo)
    h = lo + torch.rand(K, device=dev, generator=g) * (hi - lo)
    return torch.stack([torch.arange(K, device=dev).float() // (K // B), x1, y1, (x1 + w).clamp(max=999), (y1 + h).clamp(max=599)], 1).contiguous()


SETS = [("microbench 32..432 px", rois_of(2048, 32, 432)), ("small 16..110 px", rois_of(2048, 16, 110)), ("medium 112..224 px", rois_of(2048, 112, 224)),
        ("large 340..600 px", rois_of(2048, 340, 600)), ("256 RoIs 32..432 px", rois_of(256, 32, 432))]
print("%-26s %5s %12s %12s %8s   algorithmic TB/s (plain -> reuse)" % ("RoI set", "step", "plain us", "reuse us", "ratio"))
for name, rois in SETS:
    for step in (2, 1):
        res = {}
        best = {"0": 1e9, "1": 1e9}
        for rnd in range(3):
            for v in ("0", "1"):
                os.environ["ABR_ROIALIGN_REUSE"] = v
                fn = lambda: ops.roi_align_forward(feat, rois, 0.0625, 7, 7, 0, bin_step=step)
                res[v] = fn()
                best[v] = min(best[v], timeit(fn))
        assert torch.equal(res["0"], res["1"]), (name, step, float((res["0"] - res["1"]).abs().max()))
        by = res["1"].numel() * 4 + feat.numel() * 4
        print("%-26s %5d %12.1f %12.1f %8.2f   %.2f -> %.2f" % (name, step, best["0"] * 1e3, best["1"] * 1e3, best["0"] / best["1"],
                                                                  by / best["0"] / 1e9, by / best["1"] / 1e9))
os.environ.pop("ABR_ROIALIGN_REUSE", None)
