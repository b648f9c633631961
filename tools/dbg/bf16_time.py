import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
torch.manual_seed(0)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [  # B,H,W,Cin,Cout,k,stride,pad
    (4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 1024, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1),
    (4, 75, 125, 512, 128, 1, 1, 0), (4, 75, 125, 128, 128, 3, 1, 1), (4, 150, 250, 256, 64, 1, 1, 0), (4, 150, 250, 64, 64, 3, 1, 1)]
ver = 1
for (B, H, W, Ci, Co, k, s, p) in shapes:
    x = torch.randn(B, H, W, Ci, device="cuda"); w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
    gy = torch.randn(B, H, W, Co, device="cuda"); dw = torch.zeros_like(w)
    row = []
    for name, m, env in (("x6", ops.MATH_BF16X6, None), ("bf16", ops.MATH_BF16, None)):
        ver += 1
        v = ver
        f = t(lambda: ops.conv_forward(x, w, s, p, relu=True, math=m, w_version=v))
        g = t(lambda: ops.conv_wgrad(x, gy, dw, s, p, math=m))
        row.append("%s fwd %6.1f us wgrad %6.1f us" % (name, f, g))
    print("%4dx%4dx%4d k%d: " % (B * H * W, Co, Ci * k * k, k) + " | ".join(row))
