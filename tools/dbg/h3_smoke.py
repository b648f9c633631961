"""quick f16x3 check: conv forward (1x1, strided, direct 3x3, Winograd 3x3, scatter dgrad), wgrad (plain, Winograd) against float64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
EPS = 2.0 ** -24
torch.manual_seed(0)
dev = "cuda"

def err(y, y64, s64):
    return float(((y.double() - y64).abs() / s64.clamp_min(1e-300)).max()) / EPS

def conv64(x, w, stride, pad):
    f = torch.nn.functional.conv2d
    return (f(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1),
            f(x.double().abs().permute(0, 3, 1, 2), w.double().abs().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1))

for name, (B, H, W, C, Co, R, stride, pad) in {"1x1": (2, 38, 63, 1024, 256, 1, 1, 0), "1x1 s2": (2, 38, 64, 256, 512, 1, 2, 0), "3x3 direct 64": (1, 40, 40, 64, 64, 3, 1, 1),
                                             "3x3 wino": (2, 38, 63, 256, 256, 3, 1, 1), "1x1 wide": (1, 128, 128, 512, 2048, 1, 1, 0)}.items():
    x = torch.randn(B, H, W, C, device=dev) * 3
    w = torch.randn(Co, R, R, C, device=dev) / (R * R * C) ** 0.5
    sc = torch.rand(Co, device=dev) + 0.5
    bi = torch.randn(Co, device=dev) * 0.1
    y64, s64 = conv64(x, w, stride, pad)
    y64 = torch.relu(y64 * sc.double() + bi.double())
    s64 = s64 * sc.double() + bi.double().abs()
    for ver in (0, 7):
        r = {}
        for m, tag in ((ops.MATH_F32, "f32"), (ops.MATH_BF16X6, "x6"), (ops.MATH_F16X3, "h3")):
            y = ops.conv_forward(x, w, stride, pad, scale=sc, bias=bi, relu=True, math=m, w_version=ver)
            r[tag] = err(y, y64, s64)
            if m == ops.MATH_F16X3:
                aw, ae = ops.amax_of(y)
                assert aw is not None
        print(f"{name:14s} ver {ver}: err/ulp f32 {r['f32']:.2f} x6 {r['x6']:.2f} h3 {r['h3']:.2f}   flags {ops.x6_range_flags(True)}")

# wgrad
for name, (B, H, W, C, Co, R, pad) in {"wgrad 1x1": (2, 38, 63, 512, 256, 1, 0), "wgrad 3x3 wino": (2, 38, 63, 256, 256, 3, 1), "wgrad 3x3 direct": (1, 40, 40, 64, 64, 3, 1)}.items():
    x = torch.randn(B, H, W, C, device=dev)
    gy = torch.randn(B, H, W, Co, device=dev) * 1e-4
    d64 = torch.zeros(Co, R, R, C, dtype=torch.float64, device=dev)
    s64 = torch.zeros_like(d64)
    xp = torch.nn.functional.pad(x.double(), (0, 0, pad, pad, pad, pad))
    for r in range(R):
        for s in range(R):
            xs = xp[:, r:r + H, s:s + W, :].reshape(-1, C)
            d64[:, r, s, :] = gy.double().reshape(-1, Co).t() @ xs
            s64[:, r, s, :] = gy.double().abs().reshape(-1, Co).t() @ xs.abs()
    r_ = {}
    for m, tag in ((ops.MATH_F32, "f32"), (ops.MATH_BF16X6, "x6"), (ops.MATH_F16X3, "h3")):
        dw = torch.zeros(Co, R, R, C, device=dev)
        ops.conv_wgrad(x, gy, dw, 1, pad, math=m)
        r_[tag] = err(dw, d64, s64)
    print(f"{name:16s}: err/ulp f32 {r_['f32']:.2f} x6 {r_['x6']:.2f} h3 {r_['h3']:.2f}   flags {ops.x6_range_flags(True)}")

# scatter dgrad (stride-2 1x1) + residual accumulate
gy = torch.randn(2, 19, 32, 512, device=dev)
w1 = torch.randn(512, 1, 1, 256, device=dev) / 16
wt = ops.conv_dgrad_weights(w1, None)
for m in (ops.MATH_F32, ops.MATH_F16X3):
    gx = ops.conv_forward(gy, wt, 1, 0, out_hw=(38, 64), out_stride=(2, 2), math=m, w_version=3)
    ref = torch.zeros(2, 38, 64, 256, dtype=torch.float64, device=dev)
    ref[:, ::2, ::2, :] = (gy.double().reshape(-1, 512) @ w1.double().view(512, 256)).view(2, 19, 32, 256)
    print("scatter dgrad math", m, "max abs err", float((gx.double() - ref).abs().max()), "flags", ops.x6_range_flags(True))
print("done")
