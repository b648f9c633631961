#!/usr/bin/env python3
"""The batched NMS on the bench workload's own proposals: keep counts and stand-alone time against n and max_keep.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from abr_iod_amd import ops  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from microbench import timeit  # noqa: E402

cfg_s, cfg_t = make_cfgs("15-5")
ms, mt = build_models(cfg_s, cfg_t, seed=0)
images, targets = synthetic_batch(4)
captured = {}
orig = ops.nms_sorted_batched


def spy(boxes, counts, thr, max_keep, strict_gt=False):
    captured.setdefault("args", (boxes.clone(), counts.clone(), thr, max_keep, strict_gt))
    return orig(boxes, counts, thr, max_keep, strict_gt)


ops.nms_sorted_batched = spy
import abr_iod_amd.modeling.rpn.rpn as R  # noqa: E402
R.ops.nms_sorted_batched = spy
mt.train()
with torch.no_grad():
    mt(images, targets)
torch.cuda.synchronize()
boxes, counts, thr, max_keep, sg = captured["args"]
print("boxes", tuple(boxes.shape), "thr", thr, "max_keep", max_keep)
keep, nk = orig(boxes, counts, thr, max_keep, sg)
print("n_keep", nk.tolist(), "last kept index per image", [int(keep[i, int(nk[i]) - 1]) for i in range(boxes.shape[0])])
for n in (3000, 6000, 12000):
    for mk in (500, 1000, 2000, 12000):
        b = boxes[:, :n].contiguous()
        c = torch.full_like(counts, n)
        t = timeit(lambda: orig(b, c, thr, mk, sg), iters=10) * 1e3
        _, k2 = orig(b, c, thr, mk, sg)
        print(f"n {n:6d} max_keep {mk:6d}: {t:7.1f} us  n_keep {k2.tolist()}")
