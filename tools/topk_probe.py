#!/usr/bin/env python3
"""topk_sigmoid alone: N = 4 images x 38*63*15 anchors (600x1000 C4), k = 12000 (training) / 6000 (testing); checks against torch.  GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd import ops

torch.manual_seed(0)
N, nloc, A, ld = 4, 38 * 63, 15, 76
for scale in (1.0, 0.01):
    y = torch.randn(N, nloc, ld, device="cuda") * scale
    for k in (12000, 6000, 1000):
        s, i = ops.topk_sigmoid(y, A, k)
        ref = torch.sigmoid(y[:, :, :A].reshape(N, -1))
        rs, ri = ref.topk(k, dim=1, sorted=True)
        ok_s = torch.equal(s, rs)
        # equal scores: ascending anchor index
        srt = torch.sort(torch.stack([-ref.double() * 2 ** 40, torch.arange(ref.shape[1], device="cuda").expand_as(ref).double()], 0)[0] * 2 ** 20
                         + torch.arange(ref.shape[1], device="cuda").expand_as(ref).double(), dim=1)[1][:, :k]
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5): ops.topk_sigmoid(y, A, k)
        a.record()
        for _ in range(50): ops.topk_sigmoid(y, A, k)
        b.record(); b.synchronize()
        print(f"logit scale {scale}: k={k}: {a.elapsed_time(b) / 50 * 1e3:.1f} us  scores equal torch.topk: {ok_s}  gathered scores consistent: {torch.equal(ref.gather(1, i), s)}")
