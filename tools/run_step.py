#!/usr/bin/env python3
"""Run a few incremental training steps on synthetic data and print losses + step time (GPU box)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--feat", default="ard")
    ap.add_argument("--alpha", type=float, default=0.5)
    a = ap.parse_args()
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id" if a.alpha > 0 else "l2", feat=a.feat, alpha=a.alpha)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    print("trainable elems", mt.flat.n_trainable, "segments", len(mt.flat.segments), flush=True)
    images, targets = synthetic_batch(a.batch, a.height, a.width)
    for it in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.time()
        ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"step {it}: {dt * 1e3:.1f} ms  total={float(total):.4f}  " + "  ".join(f"{k}={float(v):.4f}" for k, v in ld.items()), flush=True)


if __name__ == "__main__":
    main()
