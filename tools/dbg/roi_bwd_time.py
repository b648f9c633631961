import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
torch.manual_seed(0)
dev = "cuda"
B, Rr = 4, 512
K_ = B * Rr
x1 = torch.rand(K_, device=dev) * 500; y1 = torch.rand(K_, device=dev) * 300
rois = torch.stack([torch.arange(K_, device=dev).float() // Rr, x1, y1, (x1 + 32 + torch.rand(K_, device=dev) * 400).clamp(max=999),
                    (y1 + 32 + torch.rand(K_, device=dev) * 250).clamp(max=599)], 1).contiguous()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for step in (2, 1):
    po = 4 if step == 2 else 7
    g = torch.randn(K_, po, po, 1024, device=dev)
    out = ops.roi_align_backward(g, rois, 0.0625, 7, 7, 0, B, 38, 63, 1024, bin_step=step)
    print("bin_step", step, "K", K_, "%.1f us" % t(lambda: ops.roi_align_backward(g, rois, 0.0625, 7, 7, 0, B, 38, 63, 1024, bin_step=step)), "checksum %.6e" % float(out.double().sum()), "absmax %.4e" % float(out.abs().max()))
    gs = g[:256].contiguous(); rs = rois[:256].contiguous(); rs[:, 0] = torch.arange(256, device=dev) // 64
    print("   K 256: %.1f us" % t(lambda: ops.roi_align_backward(gs, rs, 0.0625, 7, 7, 0, B, 38, 63, 1024, bin_step=step)))
