import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
for (N, n, batch, mp, dt) in [(4, 35910, 256, 128, torch.float32), (2, 35910, 256, 128, torch.float32), (4, 63000, 256, 128, torch.float32), (4, 2005, 512, 128, torch.int64), (2, 2005, 512, 128, torch.int64)]:
    lab = torch.zeros(N, n)
    lab[torch.rand(N, n) < 0.002] = 1
    lab[torch.rand(N, n) < 0.05] = -1
    lab = lab.to(dt).cuda()
    print("N = %d, n = %6d, %s: %.1f us" % (N, n, str(dt).split('.')[-1], t(lambda: ops.sample_pos_neg(lab, batch, mp, seed=5))))
print("-- n = 35910, N = 4: what the passes cost")
n, N = 35910, 4
def mk(p_pos, p_neg):
    r = torch.rand(N, n)
    lab = torch.full((N, n), -1.0)
    lab[r < p_neg] = 0
    lab[r < p_pos] = 1
    return lab.cuda()
for name, lab in [("everything ignored", mk(0, 0)), ("100 negatives, 20 positives (no radix select)", mk(0.0005, 0.003)),
                  ("all negatives but 70 positives (radix select over 35 800)", mk(0.002, 1.1)), ("half ignored", mk(0.002, 0.5))]:
    print("%-60s %.1f us" % (name, t(lambda: ops.sample_pos_neg(lab, 256, 128, seed=5))))
