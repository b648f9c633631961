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
print("ABR_X6_SPLITK =", os.environ.get("ABR_X6_SPLITK", "1"), " ABR_X6_SPLITK_MINK =", os.environ.get("ABR_X6_SPLITK_MINK", "512"))
ver = 10
for (M, N, K) in [(9576, 256, 1024), (4788, 256, 1024), (9576, 76, 1024), (9576, 512, 1024), (37500, 128, 512), (18750, 128, 512), (9576, 256, 512), (4096, 512, 2048), (4096, 2048, 512), (2048, 512, 2048), (9576, 1024, 256)]:
    x = torch.randn(1, 1, M, K, device="cuda"); w = torch.randn(N, 1, 1, K, device="cuda") * 0.05
    res = torch.randn(1, 1, M, N, device="cuda")
    ver += 1
    v = ver
    us = t(lambda: ops.conv_forward(x, w, 1, 0, residual=res, relu=True, math=ops.MATH_BF16X6, w_version=v))
    print("%6d x %4d x %4d: %6.1f us  %6.1f TF-eq" % (M, N, K, us, 2.0 * M * N * K / us * 1e-6))
