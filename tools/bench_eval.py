#!/usr/bin/env python3
"""Throughput of the test-time path (reference engine/inference.py:43-109): the target detector in eval mode on batches of
TEST.IMS_PER_BATCH = 8 synthetic 600x1000 images (6000 -> 1000 proposals per image, 100 detections per image), detections moved to the host per batch as
compute_on_dataset does.  Prints one JSON line: images/s, ms per batch, and where the time goes (events).   GPU box: python tools/bench_eval.py [--batch 8]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from abr_iod_amd.engine.inference import EvalRangeGuard  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--math", default=None, choices=[None, "f32", "bf16x6", "f16x3"], help="contraction arithmetic (default: the library's default, f16x3)")
a = ap.parse_args()
if a.math:
    os.environ["ABR_CONV_MATH"] = a.math
cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=4)
_, model = build_models(cfg_s, cfg_t, seed=0)
model.eval()
batches = [synthetic_batch(a.batch, 600, 1000, seed=50 + i)[0] for i in range(3)]


guard = EvalRangeGuard(model)     # the test loop's range guard (engine/inference.py): inside the timed region, as compute_on_dataset runs it


def one(images):
    with torch.no_grad():
        output, _features, background = guard.forward(images)
    out = [o.to("cpu") for o in output]
    bg = background.to("cpu") if background is not None else None
    return out, bg


for i in range(4):
    out, _ = one(batches[i % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
n_det = 0
for i in range(a.iters):
    out, _ = one(batches[i % 3])
    n_det += sum(len(o) for o in out)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# device-only time of the forward (no host copies): events around model(images)
ev = []
for i in range(a.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        model(batches[i % 3])
    e1.record()
    ev.append((e0, e1))
torch.cuda.synchronize()
dev_ms = sum(x.elapsed_time(y) for x, y in ev) / len(ev)
print(json.dumps({"metric": "test-time images/sec (R50-C4 Faster R-CNN, eval mode)", "value": round(a.batch * a.iters / dt, 2), "unit": "img/s",
                  "ms_per_batch": round(1e3 * dt / a.iters, 3), "batch": a.batch, "device_ms_per_batch_forward_only": round(dev_ms, 3),
                  "detections_per_image": round(n_det / (a.batch * a.iters), 1), "math": getattr(model, "conv_math", a.math),
                  "range_guard": dict(guard.stats, small_fraction=(guard.stats["small"] / guard.stats["seen"] if guard.stats["seen"] else None),
                                      note="polled per batch inside the timed region; a batch outside the arithmetic's domain is re-run one arithmetic down"), "data": "synthetic 600x1000, random-init weights",
                  "config": {"workload": "TEST.IMS_PER_BATCH 8, PRE/POST_NMS_TOP_N_TEST 6000/1000, DETECTIONS_PER_IMG 100 (configs/voc/15-5/*RB_Target_model.yaml)"}}))
