#!/usr/bin/env python3
"""Is the host the bottleneck?  Host-side time to ISSUE n free-running steps against the device time to finish them, per batch size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

for B in (4, 2, 1):
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=B)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    images, targets = synthetic_batch(B, 600, 1000, seed=42, label_range=(16, 21))
    for _ in range(6):
        train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B = %d: host issued %d steps in %.2f ms / step; device finished them %.2f ms / step (host ahead by %.1f ms at the end)" % (
        B, n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n, 1e3 * (t2 - t1)), flush=True)
    del ms, mt, opt
    torch.cuda.empty_cache()
