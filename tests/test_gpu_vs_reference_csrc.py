"""GPU: the HIP library against the REFERENCE'S OWN compiled C++ at full size, on the GPU box.

oracle/_ref/_C.so is /root/reference/maskrcnn_benchmark/csrc (vision.cpp + cpu/*.cpp) compiled unmodified by oracle/Makefile in
the build container; being a built artefact it travels with the snapshot (the sources and the Python reference cannot).  Here it
is the checker -- never the product: `abr_iod_amd._C` has the same call signatures (csrc/vision.cpp:10-16), so both are called
with the same arguments on BASELINE.json's geometry (38x63x1024 C4 map, 4 x 512 RoIs; 12000 ranked proposals per NMS call):
    * ROIAlign forward (ROIAlign_cpu.cpp:112-257): values bit-identical
    * NMS (nms_cpu.cpp:5-67, IoU >= thr suppresses): keep lists index-identical
Skipped when the file is absent (a checkout that never built the checker)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "_C.so")


@pytest.fixture(scope="module")
def ref_C():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/_C.so not built (make -C oracle ref, build container only)")
    spec = importlib.util.spec_from_file_location("_C", REF_SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rois(rng, n_img, per_img, W=1000, H=600):
    out = []
    for b in range(n_img):
        x1 = rng.uniform(-20, W - 10, per_img); y1 = rng.uniform(-20, H - 10, per_img)
        w = np.exp(rng.uniform(np.log(4), np.log(700), per_img)); h = np.exp(rng.uniform(np.log(4), np.log(500), per_img))
        out.append(np.stack([np.full(per_img, b), x1, y1, np.minimum(x1 + w, W + 15), np.minimum(y1 + h, H + 15)], 1))
    return np.concatenate(out).astype(np.float32)


@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_forward_full_size_bit_exact_vs_reference_build(ref_C, sr):
    from abr_iod_amd import _C
    rng = np.random.default_rng(7 + sr)
    B, Ch, H, W = 4, 1024, 38, 63
    feat = torch.from_numpy(rng.standard_normal((B, Ch, H, W)).astype(np.float32))
    rois = torch.from_numpy(_rois(rng, B, 512))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    want = ref_C.roi_align_forward(feat, rois, 0.0625, 7, 7, sr)
    got = _C.roi_align_forward(feat.cuda(), rois.cuda(), 0.0625, 7, 7, sr).cpu()
    assert got.shape == want.shape == (2048, Ch, 7, 7)
    assert torch.equal(got, want), float((got - want).abs().max())


def test_nms_full_size_index_exact_vs_reference_build(ref_C):
    from abr_iod_amd import _C
    rng = np.random.default_rng(11)
    for n, thr in ((12000, 0.7), (6000, 0.7), (1000, 0.5)):
        # clustered boxes (heavy overlap, as decoded RPN proposals are) with distinct scores
        centers = rng.uniform([50, 50], [950, 550], (n // 20 + 1, 2)).repeat(20, 0)[:n]
        c = centers + rng.normal(0, 12, (n, 2))
        wh = np.exp(rng.normal(np.log(120), 0.5, (n, 2)))
        boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        boxes[:, 0::2] = boxes[:, 0::2].clip(0, 999); boxes[:, 1::2] = boxes[:, 1::2].clip(0, 599)
        scores = rng.permutation(n).astype(np.float32) / n
        b, s = torch.from_numpy(boxes), torch.from_numpy(scores)
        want = ref_C.nms(b, s, thr)
        got = _C.nms(b.cuda(), s.cuda(), thr).cpu()
        assert 0 < want.numel() < n
        assert torch.equal(got, want), (n, thr, got.numel(), want.numel())
