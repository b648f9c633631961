#!/usr/bin/env python3
"""Census of the ATen ops (and their python call sites) issued by one training step: what is left of torch glue on the hot path.
GPU box: python tools/op_census.py [--stack]"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stack", action="store_true")
    ap.add_argument("--ops", default="aten::copy_,aten::clone,aten::cat,aten::to,aten::_to_copy,aten::contiguous,aten::index_select,aten::fill_,aten::zero_,aten::add_,aten::add,aten::mul,aten::sum,aten::index,aten::full,aten::arange,aten::sort,aten::zeros,aten::div,aten::where")
    a = ap.parse_args()
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    images, targets = synthetic_batch(4)
    for _ in range(2):
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=a.stack) as prof:
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
    if not a.stack:
        rows = sorted(prof.key_averages(), key=lambda e: -e.count)
        for e in rows[:60]:
            print(f"{e.count:5d}  {e.key:45s} self_cpu {e.self_cpu_time_total / 1e3:8.2f} ms")
        return
    want = set(a.ops.split(","))
    import collections
    sites = collections.Counter()
    for e in prof.events():
        if e.name in want and e.stack:
            fr = [f for f in e.stack if "abr_iod_amd" in f and "ops.py" not in f.split(":")[0][-8:]]
            key = (e.name, fr[0].split("abr_iod_amd/")[-1] if fr else e.stack[0][-60:])
            sites[key] += 1
    for (name, site), n in sites.most_common(70):
        print(f"{n:5d}  {name:22s} {site}")


if __name__ == "__main__":
    main()
