#!/usr/bin/env python3
"""Where the step's device-to-device copies (`__amd_rocclr_copyBuffer` in a kernel trace) and small torch kernels come from: one step under
torch.profiler with Python stacks, grouped by op and by the innermost abr_iod_amd frame.  GPU box: python tools/dbg/copy_sources.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=4)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt)
sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(4, 600, 1000, seed=42, label_range=(16, 21))
for _ in range(4):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    torch.cuda.synchronize()
ops = collections.Counter()
where = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue      # only the outermost aten op of a call
    ops[e.name] += 1
    frame = next((s for s in (e.stack or []) if "abr_iod_amd" in s or "bench" in s), "(autograd engine / no python frame)")
    where[(e.name, frame.split("abr_iod_amd/")[-1][:110], str(e.input_shapes)[:60])] += 1
print("outermost aten ops of one step:", sum(ops.values()))
for k, v in ops.most_common(25):
    print("  %5d  %s" % (v, k))
print()
for (n, f, sh), v in where.most_common(70):
    print("  %4d  %-22s %-112s %s" % (v, n, f, sh))
kern = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:80]] += 1
print()
for k, v in kern.most_common(12):
    print("  %5d  %s" % (v, k))
