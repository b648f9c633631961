#!/usr/bin/env python3
"""cProfile of the host side of the training step (GPU box): where the Python / ctypes time goes, and how much is spent waiting in
the three device->host reads."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402


def main():
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    images, targets = synthetic_batch(4)
    for _ in range(3):
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
