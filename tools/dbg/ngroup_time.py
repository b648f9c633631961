import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
torch.manual_seed(0)
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("ABR_X6_NGROUP =", os.environ.get("ABR_X6_NGROUP", "(rule)"))
ver = 10
for (M, N, K) in [(36864, 2048, 512), (36864, 2048, 1024), (36864, 1024, 2048), (36864, 512, 2048), (36864, 1024, 512), (18432, 2048, 512), (9576, 1024, 1024)]:
    x = torch.randn(1, 1, M, K, device="cuda"); w = torch.randn(N, 1, 1, K, device="cuda") * 0.05
    res = torch.randn(1, 1, M, N, device="cuda")
    ver += 1
    v = ver
    us = t(lambda: ops.conv_forward(x, w, 1, 0, residual=res, relu=True, math=ops.MATH_BF16X6, w_version=v))
    print("%6d x %4d x %4d: %6.1f us  %6.1f TF-eq" % (M, N, K, us, 2.0 * M * N * K / us * 1e-6))
# RPN-sized Winograd 3x3 (1024 -> 1024 at 38x63, B = 4) and layer4's (512 -> 512 on 2304 4x4 maps)
for (B, H, W, C, N) in [(4, 38, 63, 1024, 1024), (2304, 4, 4, 512, 512)]:
    x = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(N, 3, 3, C, device="cuda") * 0.02
    ver += 1
    v = ver
    us = t(lambda: ops.conv_forward(x, w, 1, 1, relu=True, math=ops.MATH_BF16X6, w_version=v))
    print("3x3 %d x %dx%d x %d -> %d: %6.1f us" % (B, H, W, C, N, us))
