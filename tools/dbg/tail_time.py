import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
torch.manual_seed(0)
B, H, W = 4, 150, 250
o1 = torch.relu(torch.randn(B, H, W, 64, device="cuda"))
x = torch.relu(torch.randn(B, H, W, 256, device="cuda"))
w1 = torch.randn(64, 1, 1, 256, device="cuda") * 0.06
w2 = torch.randn(64, 3, 3, 64, device="cuda") * 0.06
w3 = torch.randn(256, 1, 1, 64, device="cuda") * 0.15
s2, b2 = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.2
s3, b3 = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.2
X6 = ops.MATH_BF16X6
def t(fn, n=int(os.environ.get("NIT", "50"))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
c1 = lambda: ops.conv_forward(x, w1, 1, 0, scale=s2, bias=b2, relu=True, math=X6, w_version=7)
c2 = lambda: ops.conv_forward(o1, w2, 1, 1, scale=s2, bias=b2, relu=True, math=X6, w_version=7)
o2 = c2()
c3 = lambda: ops.conv_forward(o2, w3, 1, 0, scale=s3, bias=b3, residual=x, relu=True, math=X6, w_version=7)
fu = lambda: ops.bottleneck_tail64(o1, w2, w3, s2, b2, s3, b3, x, 7, 7)
print("conv1 1x1 256->64      %.1f us" % t(c1))
print("conv2 3x3 64->64       %.1f us" % t(c2))
print("conv3 1x1 64->256 +res %.1f us" % t(c3))
print("fused conv2+conv3      %.1f us" % t(fu))
