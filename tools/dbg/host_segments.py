#!/usr/bin/env python3
"""Where the host's issue time goes at the floor (B = 1, 192 x 256 images: every kernel short): wall time of the step's segments on the main
thread (time.perf_counter at the trainer's own marks) -- forward up to the read-back, the read-back's wait, the rest of the forward, backward
(autograd's thread), optimizer."""
import os, sys, time, collections
os.environ["ABR_STEP_MARKS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
import abr_iod_amd.engine.trainer as T
from abr_iod_amd.engine import train_step
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer

H, W = int(os.environ.get("FLOOR_H", 192)), int(os.environ.get("FLOOR_W", 256))
B = int(os.environ.get("FLOOR_B", 1))
cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=B)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(B, H, W, seed=42, label_range=(16, 21))
acc = collections.OrderedDict()
last = [0.0]
def mark(name):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - last[0])
    last[0] = t
T._ops.mark = mark          # the trainer's own marks (ops.mark) become host timestamps
orig_step, orig_bwd = opt.step, torch.Tensor.backward
for _ in range(8):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
acc.clear()
n = 40
t0 = time.perf_counter(); last[0] = t0
for _ in range(n):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    mark("(scheduler, guard, python between steps)")
t1 = time.perf_counter()
torch.cuda.synchronize()
print("B = %d, %d x %d: %.2f ms / step issued" % (B, H, W, 1e3 * (t1 - t0) / n))
for k, v in acc.items():
    print("  %6.2f ms  up to: %s" % (1e3 * v / n, k))
