#!/usr/bin/env python3
"""Main-stream timeline of the UN-PROFILED training step: ABR_STEP_MARKS=1 makes the trainer record events at a handful of points; this tool runs
the bench workload, averages the marks over the timed steps and prints when the main stream reaches each point (ms after the step's first mark)
and how long each span took.  GPU box:  python tools/step_marks.py [--batch-per-gpu 4] [--steps 20]"""
import argparse
import collections
import os
import sys

os.environ["ABR_STEP_MARKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from abr_iod_amd import ops  # noqa: E402
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-per-gpu", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=a.batch_per_gpu)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt)
sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(a.batch_per_gpu, 600, 1000, seed=42, label_range=(16, 21))
for _ in range(5):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
ops.take_marks()
acc, order = collections.defaultdict(list), []
FIRST = "step: target forward_begin issued from here"
for _ in range(a.steps):      # FREE-RUNNING: no synchronisation between the steps (the host runs ahead of the device as in bench.py)
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
marks = list(ops._marks)
del ops._marks[:]
step_len = []
start = None
prev_start = None
for n, e in marks:
    if n == FIRST:
        if start is not None:
            step_len.append(start.elapsed_time(e))
        start = e
    t = start.elapsed_time(e)
    if n not in acc:
        order.append(n)
    acc[n].append(t)
print("B = %d, %d free-running steps; step = %.3f ms (first mark to first mark)" % (a.batch_per_gpu, a.steps, sum(step_len) / max(len(step_len), 1)))
last = 0.0
for n in order:
    t = sum(acc[n]) / len(acc[n])
    print("  %8.3f ms  (+%6.3f)  %s" % (t, t - last, n))
    last = t
