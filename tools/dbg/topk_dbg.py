import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
torch.manual_seed(0)
A, ld = 15, 76
for (N, hw, k) in ((1, 16 * 16, 1000), (1, 38 * 63, 12000), (3, 38 * 63, 6000)):
    y = torch.randn(N, hw, ld, device="cuda") * 3
    sc, idx = ops.topk_sigmoid(y, A, k)
    torch.cuda.synchronize()
    s_all = torch.sigmoid(y[:, :, :A].reshape(N, -1))
    ref_s, ref_i = s_all.topk(k, dim=1, sorted=True)
    print("case", N, hw, k, "scores equal", torch.equal(sc, ref_s), "idx range", int(idx.min()), int(idx.max()), "n", hw * A)
    bad = (sc != ref_s)
    print("  mismatches", int(bad.sum()), "first", bad.nonzero()[:5].tolist())
    print("  sorted desc", bool((sc[:, 1:] <= sc[:, :-1]).all()), "zeros", int((sc == 0).sum()))
    print("  sc[:8]", sc[0, :8].tolist(), "ref", ref_s[0, :8].tolist())
    print("  sc[-4:]", sc[0, -4:].tolist(), "ref", ref_s[0, -4:].tolist())
