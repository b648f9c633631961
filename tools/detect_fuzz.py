#!/usr/bin/env python3
"""Random / degenerate-input sweep of the detection kernels against the oracle (test infrastructure: oracle/ is the checker here, never the
product): ROIAlign forward (bit-exact), backward in both forms, NMS keep lists (index-exact) and the fused sigmoid + top-k ranking -- with RoIs
that are empty, inverted, far outside the map or larger than it, pooled sizes 1..9, sampling ratios 0..3, box lists full of duplicates and
exact score / IoU ties, empty images.  GPU box: python tools/detect_fuzz.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from abr_iod_amd import _C, ops  # noqa: E402
from oracle import ops as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
fails = []
count = {"roi_align_forward": 0, "roi_align_backward": 0, "nms": 0, "topk": 0}


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def nhwc(x):
    return T(np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1))))


def random_rois(K, B, H, W, scale):
    """image-space boxes of every kind: ordinary, tiny, empty, inverted, partly / wholly outside, huge"""
    iw, ih = W / scale, H / scale
    kind = rng.integers(0, 8, K)
    x1 = rng.uniform(-0.2 * iw, 1.1 * iw, K); y1 = rng.uniform(-0.2 * ih, 1.1 * ih, K)
    w = np.exp(rng.uniform(np.log(0.5), np.log(1.5 * iw + 1), K)); h = np.exp(rng.uniform(np.log(0.5), np.log(1.5 * ih + 1), K))
    w[kind == 1] = 0.0; h[kind == 2] = 0.0                       # empty
    w[kind == 3] *= -1.0                                          # inverted
    x1[kind == 4] = 3.0 * iw; y1[kind == 5] = -3.0 * ih           # wholly outside
    snap = kind == 6                                              # integer-aligned (samples land on pixel centres / borders)
    x1[snap] = np.round(x1[snap] * scale) / scale; y1[snap] = np.round(y1[snap] * scale) / scale
    w[snap] = np.round(w[snap] * scale) / scale; h[snap] = np.round(h[snap] * scale) / scale
    return np.stack([rng.integers(0, B, K).astype(np.float64), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


for ci in range(a.cases):
    # ---------------------------------------------------------------- ROIAlign
    B, Ch = int(rng.integers(1, 5)), int(rng.choice([1, 3, 4, 8, 24, 64, 100]))
    H, W = int(rng.integers(1, 51)), int(rng.integers(1, 51))
    ph, pw = int(rng.integers(1, 10)), int(rng.integers(1, 10))
    sr = int(rng.integers(0, 4))
    scale = float(rng.choice([1 / 16, 1 / 8, 1.0, 0.3]))
    K = int(rng.choice([0, 1, 2, 17, 64, 200]))
    feat = rng.standard_normal((B, Ch, H, W)).astype(np.float32)
    rois = random_rois(K, B, H, W, scale)
    case = ("roi", B, Ch, H, W, ph, pw, sr, scale, K)
    if K > 0:
        want = O.roi_align_forward(feat, rois, scale, ph, pw, sr)
        got = ops.roi_align_forward(nhwc(feat), T(rois), scale, ph, pw, sr).permute(0, 3, 1, 2).cpu().numpy()
        count["roi_align_forward"] += 1
        if not np.array_equal(got, want):
            fails.append(("roi_align_forward NHWC", case, float(np.abs(got - want).max())))
        got = _C.roi_align_forward(T(feat), T(rois), scale, ph, pw, sr).cpu().numpy()
        if not np.array_equal(got, want):
            fails.append(("roi_align_forward NCHW", case, float(np.abs(got - want).max())))
        for step in ([1, 2] if min(ph, pw) >= 2 else [1]):
            gy = rng.standard_normal((K, Ch, ph, pw)).astype(np.float32)
            gz = np.zeros_like(gy); gz[:, :, ::step, ::step] = gy[:, :, ::step, ::step]
            want = O.roi_align_backward(gz, rois, scale, ph, pw, B, Ch, H, W, sr)
            ge = nhwc(gy)[:, ::step, ::step, :].contiguous()
            tol = 1e-5 * max(1.0, float(np.abs(want).max()))
            for method in ("gather", "scatter"):
                got = ops.roi_align_backward(ge, T(rois), scale, ph, pw, sr, B, H, W, Ch, bin_step=step, method=method).permute(0, 3, 1, 2).cpu().numpy()
                count["roi_align_backward"] += 1
                if not (np.abs(got - want).max() <= tol) or not np.isfinite(got).all():
                    fails.append(("roi_align_backward " + method, case + (step,), float(np.abs(got - want).max())))
    # ---------------------------------------------------------------- NMS
    N = int(rng.integers(1, 4))
    n = int(rng.choice([1, 2, 63, 64, 65, 300, 2047, 2048, 2049, 3500]))
    style = int(rng.integers(0, 4))
    cx = rng.uniform(0, 400, (N, n)); cy = rng.uniform(0, 300, (N, n))
    w = np.exp(rng.uniform(np.log(4), np.log(300), (N, n))); h = np.exp(rng.uniform(np.log(4), np.log(200), (N, n)))
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)
    if style == 1:        # integer grid: many exact duplicates and exact IoU ties
        boxes = np.round(boxes / 40.0) * 40.0
    elif style == 2:      # a few distinct boxes repeated
        boxes = boxes[:, rng.integers(0, min(n, 7), n)]
    elif style == 3:      # degenerate boxes among ordinary ones (zero width / height)
        z = rng.random((N, n)) < 0.2
        boxes[..., 2] = np.where(z, boxes[..., 0], boxes[..., 2])
    boxes = boxes.astype(np.float32)
    counts = rng.integers(0, n + 1, N).astype(np.int32)
    counts[rng.integers(0, N)] = n
    thr = float(rng.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
    mk = int(rng.choice([1, 64, 300, 1000, n]))
    strict = bool(rng.integers(0, 2))
    keep, nk = ops.nms_sorted_batched(T(boxes), T(counts), thr, mk, strict_gt=strict)
    keep, nk = keep.cpu().numpy(), nk.cpu().numpy()
    scores = -np.arange(n, dtype=np.float32)
    for i in range(N):
        want = O.nms(boxes[i, :counts[i]], scores[:counts[i]], thr, strict_gt=strict)[:mk]
        count["nms"] += 1
        if nk[i] != len(want) or not np.array_equal(keep[i, :nk[i]], want):
            fails.append(("nms", (N, n, style, int(counts[i]), thr, mk, strict), int(nk[i]), len(want)))
    # ---------------------------------------------------------------- sigmoid + top-k ranking
    Nn, A = int(rng.integers(1, 5)), int(rng.choice([1, 3, 15]))
    hw = int(rng.choice([1, 7, 64, 1000, 38 * 63]))
    ld = A * 5 + int(rng.integers(0, 3))
    y = torch.randn(Nn, hw, ld, device="cuda") * float(rng.choice([0.1, 3.0, 30.0]))
    if rng.integers(0, 2):
        y = torch.round(y * 2) / 2          # heavy exact ties
    k = int(min(rng.choice([1, 10, 300, 1000, 6000, 12000]), hw * A))
    sc, idx = ops.topk_sigmoid(y, A, k)
    s_all = torch.sigmoid(y[:, :, :A].reshape(Nn, -1))
    ref_s, _ = s_all.topk(k, dim=1, sorted=True)
    count["topk"] += 1
    ok = torch.allclose(sc, ref_s, rtol=2e-7, atol=0) and int(idx.min()) >= 0 and int(idx.max()) < hw * A
    ok = ok and torch.allclose(s_all.gather(1, idx.long()), sc, rtol=2e-7, atol=0)
    ok = ok and all(len(set(idx[i].tolist())) == k for i in range(Nn))
    if ok:   # ties by ascending index: within a run of equal scores the indices ascend
        eq = sc[:, 1:] == sc[:, :-1]
        ok = bool(((idx[:, 1:] > idx[:, :-1]) | ~eq).all())
    if not ok:
        fails.append(("topk", (Nn, hw, A, ld, k)))
    if (ci + 1) % 50 == 0:
        print("%d cases: %s, %d failures" % (ci + 1, count, len(fails)), flush=True)

print("\n%d cases; comparisons: %s" % (a.cases, count))
print("FAILURES: %d" % len(fails))
for f in fails[:40]:
    print("  ", f)
sys.exit(1 if fails else 0)
