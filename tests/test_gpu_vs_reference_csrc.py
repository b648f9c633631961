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


def test_nms_ties_negative_scores_and_the_sort_itself(ref_C):
    """_C.nms ranks the scores with the library's own sort (abr_sort_scores_desc): equal scores must come out in ascending index order
    (torch.sort(stable=True, descending=True), what nms_cpu.cpp:24 relies on), negative scores and zeros of both signs must order like floats."""
    import ctypes as C
    from abr_iod_amd import _C, _lib as L
    rng = np.random.default_rng(5)
    n = 9000
    scores = np.round(rng.standard_normal(n).astype(np.float32), 1)          # ~80 distinct values: long runs of ties, both signs
    scores[:7] = [0.0, -0.0, 0.0, 3.5, 3.5, -7.25, -7.25]
    s = torch.from_numpy(scores).cuda()
    order = torch.empty(n, dtype=torch.int64, device="cuda")
    L.check(L.lib().abr_sort_scores_desc(L.ptr(s), n, L.ptr(order), L.stream()), "sort_scores_desc")
    want = torch.sort(s, descending=True, stable=True)[1]
    got_scores = s[order]
    assert bool((got_scores[:-1] >= got_scores[1:]).all())
    same = got_scores == s[want]
    assert bool(same.all())
    # inside a run of equal scores the indices ascend (zeros: -0.0 == 0.0 compare equal but sort apart here: skip them)
    nz = got_scores != 0
    run = (got_scores[:-1] == got_scores[1:]) & nz[:-1] & nz[1:]
    assert bool((order[1:][run] > order[:-1][run]).all())
    # and NMS on tied scores equals the reference build
    centers = rng.uniform([50, 50], [950, 550], (n // 20 + 1, 2)).repeat(20, 0)[:n]
    c = centers + rng.normal(0, 12, (n, 2))
    wh = np.exp(rng.normal(np.log(120), 0.5, (n, 2)))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    # (the reference's own sort is not stable -- nms_cpu.cpp:24 -- so with ties ITS order is unspecified: the boxes are handed to the reference build
    #  already in the stable order, with strictly decreasing stand-in scores, and its keep list is mapped back to the original indices)
    sc = torch.from_numpy(np.where(scores == 0, np.float32(0.05), scores))     # (no signed zeros: they compare equal but sort apart)
    order_ref = torch.sort(sc, descending=True, stable=True)[1]
    keep_sorted = ref_C.nms(torch.from_numpy(boxes)[order_ref], torch.arange(n, 0, -1, dtype=torch.float32), 0.6)
    want = order_ref[keep_sorted].sort()[0]
    got = _C.nms(torch.from_numpy(boxes).cuda(), sc.cuda(), 0.6).cpu()
    assert torch.equal(got, want), (got.numel(), want.numel())


@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_float64_instantiation_vs_reference_build(ref_C, sr):
    """AT_DISPATCH_FLOATING_TYPES gives the reference double kernels too (ROIAlign_cpu.cpp:242, ROIAlign_cuda.cu:283,329): the float64 forward
    must equal the reference's own compiled double forward bit for bit; the float64 backward (CUDA-only in the reference) is the forward's
    adjoint to double precision and agrees with the float32 backward to float precision."""
    from abr_iod_amd import _C
    rng = np.random.default_rng(3 + sr)
    B, Ch, H, W = 2, 32, 38, 63
    feat = torch.from_numpy(rng.standard_normal((B, Ch, H, W)))
    rois = torch.from_numpy(_rois(rng, B, 96).astype(np.float64))
    want = ref_C.roi_align_forward(feat, rois, 0.0625, 7, 7, sr)
    got = _C.roi_align_forward(feat.cuda(), rois.cuda(), 0.0625, 7, 7, sr)
    assert got.dtype == torch.float64 and torch.equal(got.cpu(), want), float((got.cpu() - want).abs().max())
    g = torch.from_numpy(rng.standard_normal(tuple(want.shape)))
    gx = _C.roi_align_backward(g.cuda(), rois.cuda(), 0.0625, 7, 7, B, Ch, H, W, sr)
    assert gx.dtype == torch.float64
    lhs, rhs = float((got.cpu() * g).sum()), float((gx.cpu() * feat).sum())        # <A x, g> == <x, A^T g>
    assert abs(lhs - rhs) <= 1e-11 * max(1.0, abs(lhs)), (lhs, rhs)
    gx32 = _C.roi_align_backward(g.float().cuda(), rois.float().cuda(), 0.0625, 7, 7, B, Ch, H, W, sr)
    assert float((gx32.double() - gx).abs().max()) <= 2e-5 * float(gx.abs().max())
    with pytest.raises(RuntimeError):
        _C.roi_align_forward(feat.cuda(), rois.float().cuda(), 0.0625, 7, 7, sr)      # mixed dtypes are refused, not converted


def test_sigmoid_focal_loss_float64_instantiation():
    """SigmoidFocalLoss_cuda.cu:128,172 with T = double (the template keeps its float gamma / alpha and its expf / powf / logf): restated with
    torch on the host, float32 transcendentals inside float64 arithmetic"""
    from abr_iod_amd import _C
    rng = np.random.default_rng(9)
    N, Cn, gamma, alpha = 300, 20, 2.0, 0.25
    x = torch.from_numpy(rng.standard_normal((N, Cn)) * 3)
    t = torch.from_numpy(rng.integers(-1, Cn + 1, N).astype(np.int32))
    d = torch.from_numpy(rng.standard_normal((N, Cn)))
    f32 = lambda v: v.float()
    cls = torch.arange(1, Cn + 1)[None, :]
    c1 = (t[:, None] == cls).double()
    c2 = ((t[:, None] >= 0) & (t[:, None] != cls)).double()
    zn, zp = 1.0 - float(np.float32(alpha)), float(np.float32(alpha))
    p = 1.0 / (1.0 + torch.exp(f32(-x)).double())
    ge = (x >= 0).double()
    lg = torch.log(f32(torch.clamp(p, min=float(np.finfo(np.float32).tiny)))).double()
    sp = torch.log(f32(1.0 + torch.exp(f32(x - 2.0 * x * ge)).double())).double()
    pw1 = torch.pow(f32(1.0 - p), np.float32(gamma)).double()
    pw2 = torch.pow(f32(p), np.float32(gamma)).double()
    want_f = -c1 * (pw1 * lg) * zp - c2 * (pw2 * (-1.0 * x * ge - sp)) * zn
    want_b = (-c1 * (pw1 * (1.0 - p - p * gamma * lg)) * zp - c2 * (pw2 * ((-1.0 * x * ge - sp) * (1.0 - p) * gamma - p)) * zn) * d
    got_f = _C.sigmoid_focalloss_forward(x.cuda(), t.cuda(), Cn, gamma, alpha).cpu()
    got_b = _C.sigmoid_focalloss_backward(x.cuda(), t.cuda(), d.cuda(), Cn, gamma, alpha).cpu()
    assert got_f.dtype == got_b.dtype == torch.float64
    # float transcendentals differ by an ulp between libm and the device: 1e-6 relative; the double arithmetic around them is exact
    assert float((got_f - want_f).abs().max()) <= 2e-6 * float(want_f.abs().max())
    assert float((got_b - want_b).abs().max()) <= 2e-6 * float(want_b.abs().max())
    f32r = _C.sigmoid_focalloss_forward(x.float().cuda(), t.cuda(), Cn, gamma, alpha).cpu()
    assert float((f32r.double() - got_f).abs().max()) <= 1e-5 * float(got_f.abs().max())


def test_integration_option_b_stub_verbatim(ref_C):
    """The ctypes stub INTEGRATION.md shows a maintainer (option B) is executed AS PRINTED -- the code block is cut out of the document -- against
    the built library, and its roi_align_forward must equal the reference build's."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"## Option B.*?```python\n(.*?)```", text, re.S)
    assert m, "INTEGRATION.md: option B code block not found"
    code = m.group(1).replace('C.CDLL("libabr_iod_hip.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "abr_iod_amd", "libabr_iod_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md:option-B", "exec"), ns)
    rng = np.random.default_rng(21)
    feat = torch.from_numpy(rng.standard_normal((2, 64, 38, 63)).astype(np.float32))
    rois = torch.from_numpy(_rois(rng, 2, 64))
    want = ref_C.roi_align_forward(feat, rois, 0.0625, 7, 7, 0)
    got = ns["roi_align_forward"](feat.cuda(), rois.cuda(), 0.0625, 7, 7, 0).cpu()
    assert torch.equal(got, want)
    # the other four entry points of the stub's comment: the packaged binding (abr_iod_amd/_C.py) exports them with the reference's signatures
    from abr_iod_amd import _C
    import inspect
    for name, nargs in (("nms", 3), ("roi_align_backward", 10), ("sigmoid_focalloss_forward", 5), ("sigmoid_focalloss_backward", 6)):
        params = [p for p in inspect.signature(getattr(_C, name)).parameters.values() if p.default is inspect._empty]
        assert len(params) == nargs, (name, len(params))
