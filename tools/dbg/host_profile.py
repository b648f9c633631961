#!/usr/bin/env python3
"""cProfile of the host side of the training step (B = 2: the per-rank workload of the 8-GPU configurations, where host and device time are
close): where the ~11 ms of Python / ctypes / launch time per step go."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=B)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt)
sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(B, 600, 1000, seed=42, label_range=(16, 21))
for _ in range(6):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
