"""GPU: abr_conv_prepare_batch (all of a model's per-step weight preparation in three launches) leaves exactly what the per-tensor calls leave:
the dgrad copies equal abr_conv_dgrad_weights', and convolutions that find the batched derived data (packed bf16x3 planes, packed Winograd-domain
weights) cached give bit for bit the results of convolutions that derive everything themselves (w_version = 0)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (Cout, R, Cin, stride, pad, with scale)
    (256, 1, 64, 1, 0, True), (128, 3, 128, 1, 1, True), (512, 1, 256, 2, 0, False), (1024, 3, 1024, 1, 1, False),
    (64, 3, 64, 1, 1, True), (76, 1, 1024, 1, 0, False), (512, 3, 512, 1, 1, True), (256, 3, 256, 1, 1, True),
]


@pytest.mark.parametrize("math_name", ["bf16x6", "f16x3", "f32"])
def test_batched_weight_preparation_equals_per_tensor_preparation(math_name):
    from abr_iod_amd import ops
    math = {"bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3, "f32": ops.MATH_F32}[math_name]
    torch.manual_seed(0)
    ws, entries, wts, scales = [], [], [], []
    for i, (Cout, R, Cin, stride, pad, sc) in enumerate(SHAPES):
        w = torch.randn(Cout, R, R, Cin, device="cuda") * 0.05
        scale = (torch.rand(Cout, device="cuda") + 0.5) if sc else None
        wt = torch.empty(Cin, R, R, Cout, device="cuda")
        ws.append(w); wts.append(wt); scales.append(scale)
        entries.append((w, scale, wt, stride, pad, math, 1001 + 2 * i))
    ops.conv_cache_clear()
    ops.conv_prepare_batch(entries)
    torch.cuda.synchronize()
    for (Cout, R, Cin, stride, pad, sc), w, wt, scale, e in zip(SHAPES, ws, wts, scales, entries):
        assert torch.equal(wt, ops.conv_dgrad_weights(w, scale)), (Cout, R, Cin)
        H = W = 20
        x = torch.randn(2, H, W, Cin, device="cuda")
        got = ops.conv_forward(x, w, stride, pad, math=math, w_version=e[6])          # finds the batched derived data
        want = ops.conv_forward(x, w, stride, pad, math=math, w_version=0)            # derives everything itself
        assert torch.equal(got, want), ("fwd", Cout, R, Cin)
        Ho = got.shape[1]
        g = torch.randn(2, Ho, Ho, Cout, device="cuda")
        if stride == 1:   # the dgrad conv on the copy: stride 1, pad R-1-pad
            got = ops.conv_forward(g, wt, 1, R - 1 - pad, math=math, w_version=e[6])
            want = ops.conv_forward(g, wt, 1, R - 1 - pad, math=math, w_version=0)
            assert torch.equal(got, want), ("dgrad", Cout, R, Cin)
    # a second call with the same versions changes nothing and launches nothing new; a new version refills in place
    before = ops.conv_cache_bytes()
    ops.conv_prepare_batch(entries)
    ws[0].mul_(2.0)
    entries[0] = entries[0][:6] + (9001,)
    ops.conv_prepare_batch(entries)
    torch.cuda.synchronize()
    assert ops.conv_cache_bytes() == before
    x = torch.randn(2, 20, 20, SHAPES[0][2], device="cuda")
    assert torch.equal(ops.conv_forward(x, ws[0], 1, 0, math=math, w_version=9001), ops.conv_forward(x, ws[0], 1, 0, math=math, w_version=0))
    assert torch.equal(wts[0], ops.conv_dgrad_weights(ws[0], scales[0]))


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_batched_winograd_preparation_reports_non_finite_weights(bad):
    """f16x3, 3x3 stride 1: the batch goes from w straight to packed Winograd-domain planes (conv_winograd.hip: wino_h3_scales / wino_h3_pack); a
    non-finite weight must raise the range guard's flag there as it does on the per-tensor way, and poison its output channel."""
    from abr_iod_amd import ops
    torch.manual_seed(1)
    w = torch.randn(256, 3, 3, 256, device="cuda") * 0.05
    w[5, 1, 1, 7] = bad
    wt = torch.empty(256, 3, 3, 256, device="cuda")
    ops.conv_cache_clear()
    ops.x6_range_flags(reset=True)
    ops.conv_prepare_batch([(w, None, wt, 1, 1, ops.MATH_F16X3, 4242)])
    torch.cuda.synchronize()
    assert ops.x6_range_flags(reset=True) & ops.X6_FLAG_NONFINITE
    y = ops.conv_forward(torch.randn(1, 12, 12, 256, device="cuda"), w, 1, 1, math=ops.MATH_F16X3, w_version=4242)
    assert bool((~torch.isfinite(y))[..., 5].all())
    assert bool(torch.isfinite(y[..., :5]).all()) and bool(torch.isfinite(y[..., 6:]).all())
