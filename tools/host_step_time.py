#!/usr/bin/env python3
"""Host enqueue time of a training step vs its wall time: how far ahead of the GPU does the host run?  (GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine import train_step
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer

cfg_s, cfg_t = make_cfgs("15-5")
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(4)
for _ in range(5):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
n = 20
host = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print(f"wall {wall * 1e3:.2f} ms/step; host time inside train_step: mean {sum(host) / n * 1e3:.2f} ms, min {min(host) * 1e3:.2f}, max {max(host) * 1e3:.2f}")
# host-only cost: run the same loop but let the GPU drain first each step (host time then excludes waiting on syncs that hide behind GPU work)
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(22); print(buf.getvalue()[:6000])
