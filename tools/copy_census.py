#!/usr/bin/env python3
"""Which python call sites still issue small ATen copy / index / fill ops inside one training step?  (TorchDispatchMode + stack)"""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine import train_step
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer

SKIP = ("aten.view", "aten.as_strided", "aten.detach", "aten.permute", "aten.select", "aten.slice", "aten.expand", "aten.alias", "aten._unsafe_view",
        "aten.reshape", "aten.t.", "aten.unsqueeze", "aten.squeeze", "aten.empty", "aten.record_stream", "aten.is_pinned", "aten._local_scalar", "aten.sym_",
        "aten.lift_fresh", "aten.transpose", "aten.unbind", "aten.split", "aten._reshape_alias", "aten.resize_", "aten.set_", "aten.stride", "prim.", "aten.is_", "aten.numel")
sites = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if "abr_iod_amd" in f.filename and "/ops.py" not in f.filename]
            where = "{}:{}".format(fr[-1].filename.split("abr_iod_amd/")[-1], fr[-1].lineno) if fr else "?"
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


cfg_s, cfg_t = make_cfgs("15-5")
ms, mt = build_models(cfg_s, cfg_t, seed=0)
opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
images, targets = synthetic_batch(4)
for _ in range(3):
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
with Census():
    train_step(ms, mt, images, targets, opt, sch, cfg_t, next_images=images)
torch.cuda.synchronize()
tot = sum(sites.values())
print("ATen ops that launch something (forward part; autograd's backward runs outside the dispatch mode):", tot)
for (name, where), n in sites.most_common(60):
    print(f"{n:4d}  {name:38s} {where}")
