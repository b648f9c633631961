#!/usr/bin/env python3
"""Per-kernel micro-benchmark at BASELINE shapes (B images of 600x1000, C4 map 38x63).  GPU box only.
Prints achieved TFLOP/s (MFMA-bound kernels) or GB/s of ALGORITHMIC bytes (HBM-bound kernels)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd import ops  # noqa: E402


_side = []


def _capture_stream():
    if not _side:
        _side.append(torch.cuda.Stream())
    return _side[0]


def timeit(fn, iters=20, warmup=3):
    """ms per call.  A Python call into the library costs ~40-50 us of host time, more than most of these kernels run: the calls
    are captured ONCE into a HIP graph (the launches land on torch's capturing stream) and the replay is timed, so the figure is
    device time; MICROBENCH_EAGER=1 times the host loop instead (launch-bound below ~60 us)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("MICROBENCH_EAGER", "0") != "1":
        side = _capture_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()                      # per-stream scratch of the library (Winograd workspace) is allocated outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(side)
            g.replay()
            e.record(side)
            torch.cuda.synchronize()
        return s.elapsed_time(e) / iters
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def conv_cases(B, R_):
    M2, M3, MH = (75, 125), (38, 63), (4, 4)
    return [
        # name, N, H, W, Cin, Cout, k, stride, pad
        ("stem7x7", B, 600, 1000, 4, 64, 7, 2, 3),
        ("l1.conv1", B, 150, 250, 64, 64, 1, 1, 0),
        ("l1.conv2", B, 150, 250, 64, 64, 3, 1, 1),
        ("l1.conv3", B, 150, 250, 64, 256, 1, 1, 0),
        ("l1.conv1b", B, 150, 250, 256, 64, 1, 1, 0),
        ("l2.conv1", B, 75, 125, 512, 128, 1, 1, 0),
        ("l2.conv2", B, 75, 125, 128, 128, 3, 1, 1),
        ("l2.conv3", B, 75, 125, 128, 512, 1, 1, 0),
        ("l3.conv1", B, 38, 63, 1024, 256, 1, 1, 0),
        ("l3.conv2", B, 38, 63, 256, 256, 3, 1, 1),
        ("l3.conv3", B, 38, 63, 256, 1024, 1, 1, 0),
        ("rpn3x3", B, 38, 63, 1024, 1024, 3, 1, 1),
        ("l4.conv1", B * R_, 4, 4, 1024, 512, 1, 1, 0),
        ("l4.conv2", B * R_, 4, 4, 512, 512, 3, 1, 1),
        ("l4.conv3", B * R_, 4, 4, 512, 2048, 1, 1, 0),
        ("l4.ds", B * R_, 4, 4, 1024, 2048, 1, 1, 0),
        ("l4.conv1b", B * R_, 4, 4, 2048, 512, 1, 1, 0),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rois", type=int, default=512)
    ap.add_argument("--only", default="")
    ap.add_argument("--math", choices=["f32", "bf16", "bf16x6", "f16x3"], default="f32")
    ap.add_argument("--convs-only", action="store_true")
    a = ap.parse_args()
    mth = {"f32": ops.MATH_F32, "bf16": ops.MATH_BF16, "bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[a.math]
    dev = "cuda"
    print(f"{'layer':10s} {'M':>7s} {'N':>5s} {'K':>5s} | {'fwd ms':>8s} {'TF/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
    for name, N, H, W, Cin, Cout, k, s, p in conv_cases(a.batch, a.rois):
        if a.only and a.only not in name:
            continue
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
        sc = torch.rand(Cout, device=dev) + 0.5
        bi = torch.randn(Cout, device=dev)
        y = ops.conv_forward(x, w, s, p, scale=sc, bias=bi, relu=True, math=mth)
        M = y.numel() // Cout
        K = k * k * Cin
        fl = 2.0 * M * Cout * K
        t_f = timeit(lambda: ops.conv_forward(x, w, s, p, scale=sc, bias=bi, relu=True, math=mth))
        line = f"{name:10s} {M:7d} {Cout:5d} {K:5d} | {t_f:8.3f} {fl / t_f / 1e9:6.1f}"
        if k != 7 and s == 1:
            gy = torch.randn_like(y)
            wt = ops.conv_dgrad_weights(w, sc)
            t_d = timeit(lambda: ops.conv_forward(gy, wt, 1, k - 1 - p, mask=x, math=mth))
            dw = torch.zeros_like(w)
            t_w = timeit(lambda: ops.conv_wgrad(x, gy, dw, s, p, scale=sc, math=mth))
            line += f" | {t_d:8.3f} {fl / t_d / 1e9:6.1f} | {t_w:8.3f} {fl / t_w / 1e9:6.1f}"
        print(line, flush=True)
    if a.convs_only:
        return
    # HBM-bound kernels
    B, Rr = a.batch, a.rois
    feat = torch.randn(B, 38, 63, 1024, device=dev)
    K_ = B * Rr
    x1 = torch.rand(K_, device=dev) * 800; y1 = torch.rand(K_, device=dev) * 450
    rois = torch.stack([torch.arange(K_, device=dev).float() // Rr, x1, y1, (x1 + 32 + torch.rand(K_, device=dev) * 400).clamp(max=999),
                        (y1 + 32 + torch.rand(K_, device=dev) * 300).clamp(max=599)], 1)
    for step in (1, 2):
        out = ops.roi_align_forward(feat, rois, 0.0625, 7, 7, 0, bin_step=step)
        t = timeit(lambda: ops.roi_align_forward(feat, rois, 0.0625, 7, 7, 0, bin_step=step))
        by = feat.numel() * 4 + out.numel() * 4
        print(f"roi_align_fwd step={step} K={K_}: {t:.3f} ms  {by / t / 1e6:.0f} GB/s algorithmic")
        g = torch.randn_like(out)
        t = timeit(lambda: ops.roi_align_backward(g, rois, 0.0625, 7, 7, 0, B, 38, 63, 1024, bin_step=step))
        by = 2 * feat.numel() * 4 + out.numel() * 4
        print(f"roi_align_bwd step={step} K={K_}: {t:.3f} ms  {by / t / 1e6:.0f} GB/s algorithmic")
    fs = torch.randn(64 * B, 7, 7, 1024, device=dev); ft = fs + 0.1 * torch.randn_like(fs)
    loss, coef = ops.ard_forward(fs, ft, 1.0)
    t = timeit(lambda: ops.ard_forward(fs, ft, 1.0))
    print(f"ard_fwd N={64 * B}: {t:.3f} ms  {2 * fs.numel() * 4 / t / 1e6:.0f} GB/s")
    t = timeit(lambda: ops.ard_backward(fs, ft, coef, 1.0))
    print(f"ard_bwd N={64 * B}: {t:.3f} ms  {3 * fs.numel() * 4 / t / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
