#!/usr/bin/env python3
"""LR schedule golden FROM THE REFERENCE (in-container only): maskrcnn_benchmark/solver/lr_scheduler.py:10-52
(WarmupMultiStepLR) driven the way tools/train_incremental.py does -- scheduler.step() after every optimizer.step() (:146-147) --
on the reference's own per-tensor param groups (solver/build.py:7-21: weights BASE_LR / WEIGHT_DECAY, biases
BASE_LR*BIAS_LR_FACTOR / WEIGHT_DECAY_BIAS).  Stored: the lr of a weight group and of a bias group at every iteration, for
  * "short":  BASE_LR 0.01, linear warm-up 1/3 over 10 iterations, milestones (30, 40), 50 iterations, gamma 0.1
  * "const":  same with warmup_method 'constant'
  * "voc":    the configs/voc schedule (BASE_LR 0.002, WARMUP 500, STEPS (12500,), MAX_ITER 15000) at iterations 0..520 and 12490..12510
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.setup()
from maskrcnn_benchmark.solver.lr_scheduler import WarmupMultiStepLR  # noqa: E402


def run(base_lr, milestones, gamma, wf, wi, method, iters, keep):
    w, b = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.SGD([{"params": [w], "lr": base_lr, "weight_decay": 1e-4}, {"params": [b], "lr": base_lr * 2, "weight_decay": 0.0}],
                          lr=base_lr, momentum=0.9)
    sch = WarmupMultiStepLR(opt, milestones, gamma, warmup_factor=wf, warmup_iters=wi, warmup_method=method)
    out = {}
    for it in range(iters):
        if it in keep:
            out[str(it)] = [opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]]   # the lr optimizer.step() of iteration `it` uses
        opt.step()
        sch.step()
    return out


def main():
    g = {"short": dict(args=[0.01, [30, 40], 0.1, 1.0 / 3, 10, "linear"], lr=run(0.01, (30, 40), 0.1, 1.0 / 3, 10, "linear", 50, set(range(50)))),
         "const": dict(args=[0.01, [30, 40], 0.1, 1.0 / 3, 10, "constant"], lr=run(0.01, (30, 40), 0.1, 1.0 / 3, 10, "constant", 50, set(range(50)))),
         "voc": dict(args=[0.002, [12500], 0.1, 1.0 / 3, 500, "linear"],
                     lr=run(0.002, (12500,), 0.1, 1.0 / 3, 500, "linear", 12511, set(range(0, 521)) | set(range(12490, 12511))))}
    with open(os.path.join(HERE, "lr_schedule.json"), "w") as f:
        json.dump(g, f)
    print({k: len(v["lr"]) for k, v in g.items()}, g["voc"]["lr"]["0"], g["voc"]["lr"]["500"], g["voc"]["lr"]["12500"])


if __name__ == "__main__":
    main()
