#!/usr/bin/env python3
"""Host floor of the training step: B = 1 on 192 x 256 images, where every kernel is short, so the step time IS the host's issue time
(plus the waits of the read-backs on a nearly idle device).  Prints ms / step, then a cProfile of the main thread sorted by cumulative time."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

H, W = int(os.environ.get("FLOOR_H", 192)), int(os.environ.get("FLOOR_W", 256))
cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=1)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt)
sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(1, H, W, seed=42, label_range=(16, 21))
for _ in range(8):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host floor (B = 1, %d x %d): %.2f ms / step issued, %.2f ms / step finished" % (H, W, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n), flush=True)
if os.environ.get("FLOOR_PROFILE", "1") != "0":
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
