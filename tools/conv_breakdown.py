#!/usr/bin/env python3
"""Per-shape breakdown of the conv kernels inside ONE real training step (GPU box): every abr_conv_forward / abr_conv_wgrad
call is timed with HIP events (serialised), grouped by (kind, M, N, K, geometry) and ranked by the time LOST against a
125 TFLOP/s target -- the list of layers worth tuning next."""
import argparse
import collections
import os
import sys

import torch

os.environ["ABR_BLOCK_PLANS"] = "0"   # this tool times the per-conv host calls: the bottlenecks' op tables (abr_conv_run) would bypass its hooks
os.environ["ABR_WGRAD_STREAM"] = "0"  # ... and brackets every call with events on the CURRENT stream: weight gradients must run there too
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd import ops  # noqa: E402
from abr_iod_amd.engine import train_step  # noqa: E402
from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch  # noqa: E402
from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer  # noqa: E402

REC = collections.OrderedDict()
BYTES = {}
ON = [False]


SEQ = {"igemm": [], "wgrad": []}
SEQ_ON = [False]


def _timed(kind, fn, key_fn):
    def wrapper(*a, **k):
        if SEQ_ON[0]:
            key, flops = key_fn(*a, **k)
            SEQ[kind].append([list(key), flops])
            return fn(*a, **k)
        if not ON[0]:
            return fn(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **k)
        e.record()
        e.synchronize()
        key, flops = key_fn(*a, **k)
        r = REC.setdefault((kind,) + key, [0, 0.0, flops])
        r[0] += 1
        r[1] += s.elapsed_time(e)
        return out
    return wrapper


def _fwd_key(x, w, stride=1, pad=0, scale=None, bias=None, residual=None, mask=None, relu=False, out=None, out_hw=None,
             out_stride=(1, 1), *_a, **_):
    Cout, R, S, Cin = w.shape
    B, H, W, _ = x.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    M = B * Ho * Wo
    tag = "dgrad" if mask is not None or out is not None or out_hw is not None else "fwd"
    # algorithmic HBM bytes of the call: the input pixels the conv reads (a stride-2 1x1 reads a quarter), weights, output, + the residual / mask tensors
    n_out = M * Cout * (out_stride[0] * out_stride[1] if out_hw is not None else 1)
    nbytes = 4.0 * (x.numel() / (stride * stride if R == 1 else 1) + w.numel() + n_out + (M * Cout if residual is not None else 0) + (M * Cout if mask is not None else 0))
    BYTES[(tag, M, Cout, R * S * Cin, f"{R}x{S}s{stride} {H}x{W}")] = nbytes
    return (tag, M, Cout, R * S * Cin, f"{R}x{S}s{stride} {H}x{W}"), 2.0 * M * Cout * R * S * Cin


def _wg_key(x, gy, dw, stride=1, pad=0, scale=None, *_a, **_):
    Cout, R, S, Cin = dw.shape
    M = gy.numel() // Cout
    BYTES[("wgrad", M, Cout, R * S * Cin, f"{R}x{S}s{stride} {x.shape[1]}x{x.shape[2]}")] = 4.0 * (x.numel() / (stride * stride if R == 1 else 1) + gy.numel() + dw.numel())
    return ("wgrad", M, Cout, R * S * Cin, f"{R}x{S}s{stride} {x.shape[1]}x{x.shape[2]}"), 2.0 * M * Cout * R * S * Cin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--target-tf", type=float, default=125.0)
    ap.add_argument("--seq", default="", help="only record the call sequence of the last step to this JSON (run under rocprofv3 "
                    "--kernel-trace and join with tools/conv_breakdown_join.py: no per-call event overhead)")
    a = ap.parse_args()
    ops.conv_forward = _timed("igemm", ops.conv_forward, _fwd_key)
    ops.conv_wgrad = _timed("wgrad", ops.conv_wgrad, _wg_key)
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt)
    sch = make_lr_scheduler(cfg_t, opt)
    images, targets = synthetic_batch(a.batch)
    for _ in range(2):
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    if a.seq:
        import json
        SEQ_ON[0] = True
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        with open(a.seq, "w") as f:
            json.dump(SEQ, f)
        return
    ON[0] = True
    steps = 3
    for _ in range(steps):
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
    rows = []
    for key, (n, ms_, fl) in REC.items():
        t = ms_ / steps
        cnt = n / steps
        ideal = cnt * fl / a.target_tf / 1e9
        rows.append((t - ideal, key, cnt, t, fl * cnt / t / 1e9))
    rows.sort(reverse=True)
    tot = sum(r[3] for r in rows)
    print(f"total conv ms/step {tot:.2f}; lost vs {a.target_tf:.0f} TF: {sum(r[0] for r in rows):.2f} ms")
    print("(TF/s on ALGORITHMIC flops: a Winograd 3x3 executes 1/4 of them; TB/s = algorithmic HBM bytes of the call -- input, weights, output, residual / mask -- / its time)")
    print(f"{'lost ms':>8s} {'ms':>7s} {'calls':>5s} {'TF/s':>6s} {'TB/s':>5s}  kind   M       N     K     geometry")
    for lost, key, cnt, t, tf in rows:
        tbs = BYTES.get(key[1:], 0.0) * cnt / (t * 1e-3) / 1e12 if t > 0 else 0.0
        print(f"{lost:8.3f} {t:7.3f} {cnt:5.0f} {tf:6.1f} {tbs:5.2f}  {key[1]:6s} {key[2]:7d} {key[3]:5d} {key[4]:5d} {key[5]}")


if __name__ == "__main__":
    main()
