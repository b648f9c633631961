"""which per-tensor weight preparations (ops.conv_prepare_weights) and cache-miss packs does one training step make?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
from abr_iod_amd.engine import train_step
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=4)
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
im, tg = synthetic_batch(4, 600, 1000, seed=42, label_range=(16, 21))
for _ in range(3):
    train_step(ms, mt, im, tg, opt, sch, cfg_t, next_images=im)
calls = collections.Counter()
orig = ops.conv_prepare_weights
def spy(w, stride, pad, math, ver):
    calls[("prepare_weights", tuple(w.shape), stride, pad, math)] += 1
    return orig(w, stride, pad, math, ver)
ops.conv_prepare_weights = spy
origb = ops.PreparedBatch.run
def spyb(self):
    calls[("PreparedBatch.run", self.n)] += 1
    return origb(self)
ops.PreparedBatch.run = spyb
for _ in range(2):
    train_step(ms, mt, im, tg, opt, sch, cfg_t, next_images=im)
torch.cuda.synchronize()
for k, v in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(v / 2, k)
