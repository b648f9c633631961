#!/usr/bin/env python3
"""Per-kernel times of the Winograd path for ONE 3x3 shape of the step (run under rocprofv3 --kernel-trace --stats, one shape per run):
python tools/dbg/wino_time.py layer2|layer3|layer4|rpn [fwd|dgrad|wgrad]   -- prints the algorithmic bytes of each transform for the roofline."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from abr_iod_amd import ops

SH = {"layer2": (4, 75, 125, 128, 128), "layer3": (4, 38, 63, 256, 256), "layer4": (2304, 4, 4, 512, 512), "rpn": (4, 38, 63, 1024, 1024)}
name = sys.argv[1]
what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
B, H, W, Cin, Cout = SH[name]
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.02
sc = torch.rand(Cout, device="cuda") + 0.5; bi = torch.randn(Cout, device="cuda")
T = B * ((H + 3) // 4) * ((W + 3) // 4)
print("%s: x %.1f MB, V %.1f MB, M %.1f MB, out %.1f MB, tiles %d" % (name, x.numel() * 4e-6, 36 * T * Cin * 4e-6, 36 * T * Cout * 4e-6, B * H * W * Cout * 4e-6, T))
ops.amax_compute(x)
if what == "fwd":
    for _ in range(20):
        y = ops.conv_forward(x, w, 1, 1, scale=sc, bias=bi, relu=True, math=ops.MATH_F16X3, w_version=3)
else:
    gy = torch.randn(B, H, W, Cout, device="cuda"); ops.amax_compute(gy)
    dw = torch.zeros_like(w)
    v = ops.wino_v_alloc(x, w, 1, 1, ops.MATH_F16X3)
    y = ops.conv_forward(x, w, 1, 1, scale=sc, bias=bi, relu=True, math=ops.MATH_F16X3, w_version=3, wino_v=v)
    for _ in range(20):
        ops.conv_wgrad(x, gy, dw, 1, 1, scale=sc, math=ops.MATH_F16X3, wino_v=v)
torch.cuda.synchronize()
