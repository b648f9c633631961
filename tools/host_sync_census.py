#!/usr/bin/env python3
"""Where does the host block during a training step?  Wraps the read-back entry points (Tensor.tolist / item / cpu, stream and event
synchronize) with timers and prints calls and blocked ms per step by call site, next to host time and wall time per step.  (GPU box)"""
import collections
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

cfg_s, cfg_t = make_cfgs("15-5")
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt)
sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(4)
for _ in range(5):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()

blocked = collections.defaultdict(lambda: [0, 0.0])


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        t = time.perf_counter()
        r = orig(*a, **k)
        dt = time.perf_counter() - t
        fr = [s for s in traceback.extract_stack(limit=6)[:-1] if "abr_iod_amd" in s.filename]
        site = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
        b = blocked[(name, site)]
        b[0] += 1
        b[1] += dt
        return r
    setattr(owner, name, f)


for owner, name in ((torch.Tensor, "tolist"), (torch.Tensor, "item"), (torch.Tensor, "cpu"), (torch.cuda.Stream, "synchronize"),
                    (torch.cuda.Event, "synchronize")):
    wrap(owner, name)

n = 20
host = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
tot_b = sum(v[1] for v in blocked.values()) / n
print(f"wall {wall * 1e3:.2f} ms/step; host inside train_step {sum(host) / n * 1e3:.2f} ms/step of which blocked in read-backs {tot_b * 1e3:.2f} ms; "
      f"host done {1e3 * (wall * n - t_enq):.2f} ms before the device at the end")
for (name, site), (c, t) in sorted(blocked.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:12s} {site:32s} {c / n:5.1f} calls/step  {t / n * 1e3:7.3f} ms/step")
