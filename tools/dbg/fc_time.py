#!/usr/bin/env python3
"""Predictor FCs (box_head.py: one fused [n_out_pad, 2048] weight): forward, dgrad, wgrad times and the forward's error vs float64 -- GPU box.
Run under ABR_IGEMM_FC_SPLIT=0 for the unsplit kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from abr_iod_amd import ops

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.manual_seed(0)
for M, N, K in ((2304, 108, 2048), (256, 80, 2048), (2048, 108, 2048), (512, 108, 2048), (1000, 84, 2048), (64, 108, 4096)):
    x = torch.randn(M, 1, 1, K, device="cuda"); w = torch.randn(N, 1, 1, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    y = ops.conv_forward(x, w, 1, 0, bias=b).view(M, N)
    y2 = ops.conv_forward(x, w, 1, 0, bias=b).view(M, N)
    ref = x.view(M, K).double() @ w.view(N, K).double().t() + b.double()
    bound = (x.view(M, K).abs().double() @ w.view(N, K).abs().double().t())
    err = ((y.double() - ref).abs() / bound).max().item() * 2 ** 24
    t_f = timeit(lambda: ops.conv_forward(x, w, 1, 0, bias=b))
    g = torch.randn(M, 1, 1, N, device="cuda")
    wt = ops.conv_dgrad_weights(w, None)
    t_d = timeit(lambda: ops.conv_forward(g, wt, 1, 0))
    dw = torch.zeros_like(w)
    t_w = timeit(lambda: ops.conv_wgrad(x, g, dw, 1, 0))
    print("%5d x %4d x %5d: fwd %6.1f us (err %.2f units of 2^-24 sum|x||w|, rerun equal %s)  dgrad %6.1f us  wgrad %6.1f us" % (
        M, N, K, t_f, err, torch.equal(y, y2), t_d, t_w), flush=True)
