"""GPU: admission tests of the bf16x6 arithmetic (fp32 operands split exactly into three bf16 terms, six cross products on the bf16
matrix cores, fp32 accumulate: csrc/conv_igemm.hip, csrc/conv_wgrad.hip) -- what it must show before it may produce the benchmark
number.

Domain.  The split x = x0 + x1 + x2 is exact for x = 0 and for finite 2^-110 <= |x| (include/abr_iod_hip.h, abr_x6_range_flags).
  * INSIDE the domain -- operands whose exponents spread over 2^+-60 inside one reduction, magnitudes from 2^-105 to 2^120, sums
    that cancel to 2^-12 of their terms -- the error against float64, measured against the dot product's natural scale sum|x||w|,
    must stay within 2x the fp32 MFMA kernel's on the same data (the two are statistically the same size: x6 is the smaller one on
    most cases, the larger by up to 1.7x under extreme exponent spread) and below 32 ulp, and the range guard must stay silent;
  * OUTSIDE it -- non-zero magnitudes below 2^-110 (incl. fp32 subnormals), inf, nan -- the hardware range guard must raise its
    flag (every operand element is inspected once per GEMM); non-finite operands must make exactly the outputs non-finite that the
    fp32 kernel makes non-finite, and the trainer must switch the models to the fp32 MFMA kernels; tiny operands sprinkled among
    ordinary ones (what real training produces: the input gradients of padded image regions) must leave the result within the SAME
    bound as inside the domain -- each costs an absolute error below 2^-119 of the other operand -- which is why the trainer only logs
    them unless ABR_X6_STRICT=1.
Covered contractions: 1x1 conv forward (= plain GEMM), 3x3 through the Winograd domain, the input gradient, the weight gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -24
M, N, K = 384, 256, 1024


def _gen(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (lambda *s: torch.randn(*s, device="cuda", generator=g)), (lambda *s: torch.rand(*s, device="cuda", generator=g))


def _gemm(x, w, math):
    from abr_iod_amd import ops
    return ops.conv_forward(x.view(1, x.shape[0], 1, x.shape[1]).contiguous(), w.view(w.shape[0], 1, 1, w.shape[1]).contiguous(), 1, 0, math=math).view(x.shape[0], w.shape[0])


def _wgrad(x, gy, math):
    """dW[n,k] = sum_m gy[m,n] x[m,k]"""
    from abr_iod_amd import ops
    dw = torch.zeros(gy.shape[1], 1, 1, x.shape[1], device="cuda")
    ops.conv_wgrad(x.view(1, x.shape[0], 1, x.shape[1]).contiguous(), gy.view(1, gy.shape[0], 1, gy.shape[1]).contiguous(), dw, 1, 0, math=math)
    return dw.view(gy.shape[1], x.shape[1])


def _rel_err(y, y64, scale):
    ok = scale > 0
    return float(((y.double() - y64).abs()[ok] / scale[ok]).max())


CASES = ["N(0,1)", "exponent spread 2^+-20", "exponent spread 2^+-40", "exponent spread 2^+-60", "x ~ 2^-60", "x ~ 2^-100", "x ~ 2^-105",
         "x ~ 2^100", "x ~ 2^120", "x, w ~ 2^-60", "cancellation to 2^-12", "70 % exact zeros"]


def _case(name):
    rn, ru = _gen(CASES.index(name))
    if name == "N(0,1)":
        return rn(M, K), rn(N, K)
    if name.startswith("exponent spread"):   # exponents spread over 2^+-sp INSIDE every reduction, both operands
        sp = int(name.split("+-")[1])
        return rn(M, K) * torch.exp2((ru(M, K) * 2 - 1) * sp), rn(N, K) * torch.exp2((ru(N, K) * 2 - 1) * sp)
    if name.startswith("x ~ 2^"):             # magnitudes in [0.5, 1.5) * 2^e, random signs: every element inside the domain
        e = int(name.split("^")[1])
        return torch.sign(rn(M, K)) * (0.5 + ru(M, K)) * 2.0 ** e, rn(N, K) * (2.0 ** -10 if e > 0 else 1.0)
    if name == "x, w ~ 2^-60":
        return torch.sign(rn(M, K)) * (0.5 + ru(M, K)) * 2.0 ** -60, torch.sign(rn(N, K)) * (0.5 + ru(N, K)) * 2.0 ** -60
    if name == "cancellation to 2^-12":       # pairs (u, -u(1 + 2^-12 r)) against equal x
        v, u = rn(M, K // 2), rn(N, K // 2)
        return torch.stack([v, v], 2).reshape(M, K), torch.stack([u, -u * (1 + 2.0 ** -12 * rn(N, K // 2))], 2).reshape(N, K)
    if name == "70 % exact zeros":            # ReLU-like: zeros are in the domain
        xz = rn(M, K)
        xz[xz < 0.5] = 0.0
        return xz, rn(N, K)
    raise KeyError(name)


@pytest.mark.parametrize("name", CASES)
def test_in_domain_error_matches_fp32_kernel_and_guard_is_silent(name):
    from abr_iod_amd import ops
    x, w = _case(name)
    ops.x6_range_flags(reset=True)
    y64 = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    e32 = _rel_err(_gemm(x, w, ops.MATH_F32), y64, scale)
    e6 = _rel_err(_gemm(x, w, ops.MATH_BF16X6), y64, scale)
    print(f"{name}: forward  f32 {e32 / EPS:.1f} ulp   x6 {e6 / EPS:.1f} ulp")
    assert e6 <= max(2.0 * e32, 8 * EPS), (name, e6, e32)
    assert e6 <= 32 * EPS, (name, e6)
    # the weight gradient reduces over the ROW axis: dW[n,k] = sum_m G[m,n] X[m,k]; X = columns of x, G = w's values re-shaped to [M, N]
    G, X = w.t()[:M].contiguous(), x[:, :N].contiguous()
    d64 = G.double().t() @ X.double()
    dscale = G.double().abs().t() @ X.double().abs()
    w32 = _rel_err(_wgrad(X, G, ops.MATH_F32), d64, dscale)
    w6 = _rel_err(_wgrad(X, G, ops.MATH_BF16X6), d64, dscale)
    print(f"{name}: wgrad    f32 {w32 / EPS:.1f} ulp   x6 {w6 / EPS:.1f} ulp")
    assert w6 <= max(2.0 * w32, 8 * EPS), (name, w6, w32)
    assert ops.x6_range_flags(reset=True) == 0, name


def test_in_domain_winograd_and_dgrad_paths():
    """3x3 stride-1 conv (Winograd-domain GEMMs: the operands of the split are B^T d B and G g G^T) and its input gradient, wide spread."""
    from abr_iod_amd import ops
    rn, ru = _gen(1)
    B, H, W, C, Co = 2, 20, 24, 128, 128
    for sp in (0, 30):
        x = rn(B, H, W, C) * torch.exp2((ru(B, H, W, C) * 2 - 1) * sp)
        w = rn(Co, 3, 3, C) / (9 * C) ** 0.5
        y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1)
        s64 = torch.nn.functional.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs().permute(0, 3, 1, 2), padding=1)
        ops.x6_range_flags(reset=True)
        e = {}
        for m, tag in ((ops.MATH_F32, "f32"), (ops.MATH_BF16X6, "x6")):
            y = ops.conv_forward(x, w, 1, 1, math=m).permute(0, 3, 1, 2)
            e[tag] = float(((y.double() - y64).abs() / s64).max())
        print(f"winograd 3x3, spread 2^+-{sp}: f32 {e['f32'] / EPS:.1f} ulp, x6 {e['x6'] / EPS:.1f} ulp (of sum|x||w|; includes the transforms' own fp32 rounding)")
        assert e["x6"] <= max(2.0 * e["f32"], 8 * EPS)
        assert ops.x6_range_flags(reset=True) == 0
        # input gradient of a 1x1 conv = forward with the transposed weight copy
        w1 = rn(Co, 1, 1, C)
        wt = ops.conv_dgrad_weights(w1, None)
        gy = rn(B, H, W, Co) * torch.exp2((ru(B, H, W, Co) * 2 - 1) * sp)
        g64 = gy.double().reshape(-1, Co) @ w1.double().view(Co, C)
        gs = gy.double().abs().reshape(-1, Co) @ w1.double().abs().view(Co, C)
        d = {}
        for m, tag in ((ops.MATH_F32, "f32"), (ops.MATH_BF16X6, "x6")):
            gx = ops.conv_forward(gy, wt, 1, 0, math=m).reshape(-1, C)
            d[tag] = float(((gx.double() - g64).abs() / gs).max())
        assert d["x6"] <= max(2.0 * d["f32"], 8 * EPS), d
        assert ops.x6_range_flags(reset=True) == 0


@pytest.mark.parametrize("which", ["tiny 2^-120", "fp32 subnormal", "one tiny element", "tiny weight"])
def test_out_of_domain_magnitudes_raise_the_flag(which):
    from abr_iod_amd import ops
    rn, _ = _gen(2)
    x, w = rn(M, K), rn(N, K)
    if which == "tiny 2^-120":
        x = x * 2.0 ** -120
    elif which == "fp32 subnormal":
        x = x * 2.0 ** -130
    elif which == "one tiny element":
        x[M - 1, K - 1] = 2.0 ** -115          # a single element anywhere in the operand is found
    else:
        w[N - 1, 3] = -(2.0 ** -118)
    ops.x6_range_flags(reset=True)
    _gemm(x, w, ops.MATH_F32)
    assert ops.x6_range_flags(reset=False) == 0            # the fp32 kernels never touch the guard
    _gemm(x, w, ops.MATH_BF16X6)
    assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_TINY
    assert ops.x6_range_flags(reset=True) == 0             # reset worked
    if which != "tiny weight":
        _wgrad(x[:, K - N:].contiguous(), rn(M, N), ops.MATH_BF16X6)
        assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_TINY
        _wgrad(rn(M, N), x[:, K - N:].contiguous(), ops.MATH_BF16X6)   # the tiny operand in the OTHER role (gy)
        assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_TINY


@pytest.mark.parametrize("frac", [0.01, 0.5])
def test_tiny_operands_among_ordinary_ones_cost_nothing_measurable(frac):
    """Out-of-domain magnitudes (2^-112 ... fp32 subnormals) mixed into ordinary operands, in both operands and in every role: the guard
    reports them, and the error against float64 -- relative to the reduction's natural scale sum|x||w| -- stays within the in-domain
    bound (2 x the fp32 kernel's, 8 ulp).  The all-tiny reduction (the only case with bf16-like RELATIVE accuracy) is bounded absolutely."""
    from abr_iod_amd import ops
    rn, ru = _gen(7)
    x, w = rn(M, K), rn(N, K)
    tiny_x = ru(M, K) < frac
    x[tiny_x] = (torch.sign(rn(M, K)) * (0.5 + ru(M, K)) * torch.exp2(-112 - 30 * ru(M, K)))[tiny_x]      # 2^-112 ... 2^-142 (subnormal)
    tiny_w = ru(N, K) < frac / 4
    w[tiny_w] = (torch.sign(rn(N, K)) * (0.5 + ru(N, K)) * 2.0 ** -115)[tiny_w]
    ops.x6_range_flags(reset=True)
    y64 = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    e32 = _rel_err(_gemm(x, w, ops.MATH_F32), y64, scale)
    e6 = _rel_err(_gemm(x, w, ops.MATH_BF16X6), y64, scale)
    assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_TINY
    print(f"{frac:.0%} tiny elements: forward f32 {e32 / EPS:.1f} ulp, x6 {e6 / EPS:.1f} ulp")
    assert e6 <= max(2.0 * e32, 8 * EPS), (e6, e32)
    G, X = w.t()[:M].contiguous(), x[:, :N].contiguous()
    d64 = G.double().t() @ X.double()
    dscale = G.double().abs().t() @ X.double().abs()
    w32 = _rel_err(_wgrad(X, G, ops.MATH_F32), d64, dscale)
    w6 = _rel_err(_wgrad(X, G, ops.MATH_BF16X6), d64, dscale)
    assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_TINY
    assert w6 <= max(2.0 * w32, 8 * EPS), (w6, w32)
    # a reduction made of tiny operands ONLY: bf16-like relative accuracy on a result that is itself ~2^-115 of the operand scale --
    # bounded absolutely by 2^-9 |x| |w| per product
    xt = torch.sign(rn(M, K)) * (0.5 + ru(M, K)) * 2.0 ** -115
    wn = rn(N, K)
    yt = _gemm(xt, wn, ops.MATH_BF16X6).double()
    bound = (2.0 ** -9) * (xt.double().abs() @ wn.double().abs().t())
    assert bool(((yt - xt.double() @ wn.double().t()).abs() <= bound).all())
    ops.x6_range_flags(reset=True)


def test_non_finite_operands_propagate_and_raise_the_flag():
    from abr_iod_amd import ops
    rn, _ = _gen(3)
    x, w = rn(M, K), rn(N, K)
    x[3, 5] = float("inf"); x[7, 900] = float("nan"); x[11, 2] = -float("inf")
    ops.x6_range_flags(reset=True)
    y32 = _gemm(x, w, ops.MATH_F32)
    y6 = _gemm(x, w, ops.MATH_BF16X6)
    assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_NONFINITE
    bad = ~torch.isfinite(y32)
    assert bad.any(dim=1).nonzero().flatten().tolist() == [3, 7, 11]
    assert torch.equal(~torch.isfinite(y6), bad)                       # non-finiteness reaches exactly the same outputs
    assert torch.allclose(y6[~bad], y32[~bad], rtol=1e-5, atol=2e-4)    # and nothing else is disturbed (|y| ~ 32, K = 1024)
    w2 = w.clone(); w2[5, 7] = float("nan")
    y6 = _gemm(rn(M, K), w2, ops.MATH_BF16X6)
    assert ops.x6_range_flags(reset=True) == ops.X6_FLAG_NONFINITE
    assert (~torch.isfinite(y6)).any(dim=0).nonzero().flatten().tolist() == [5]


@pytest.mark.parametrize("trip", ["nonfinite", "tiny-strict", "tiny"])
def test_trainer_falls_back_to_fp32_when_the_guard_trips(trip, monkeypatch):
    """inf / nan operands: both models leave the bf16x6 arithmetic.  Tiny operands: logged once, no switch (see _x6_guard for why) --
    unless ABR_X6_STRICT=1."""
    import logging
    import os
    from abr_iod_amd import ops
    from abr_iod_amd.engine import trainer
    monkeypatch.setattr(trainer, "X6_STRICT", trip == "tiny-strict")
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128]
    os.environ["ABR_CONV_MATH"] = "bf16x6"
    try:
        cfg_s, cfg_t = make_cfgs("15-5", overrides=tiny)
        ms, mt = build_models(cfg_s, cfg_t, seed=0)
    finally:
        os.environ.pop("ABR_CONV_MATH", None)
    assert mt.conv_math == ms.conv_math == "bf16x6"
    assert all(m.math == ops.MATH_BF16X6 for m in mt.modules() if hasattr(m, "math"))
    ops.x6_range_flags(reset=True)
    trainer.trainer_state(mt).x6_watch = None
    for _ in range(3):                       # clean steps: nothing happens
        trainer._x6_guard(ms, mt)
        torch.cuda.synchronize()
    assert mt.conv_math == "bf16x6"
    rn, _ = _gen(4)
    bad = rn(M, K)
    if trip == "nonfinite":
        bad[5, 7] = float("inf")
    else:
        bad = bad * 2.0 ** -125
    _gemm(bad, rn(N, K), ops.MATH_BF16X6)          # some bf16x6 kernel of the step sees an out-of-domain operand
    records = []
    h = logging.Handler(); h.emit = records.append
    log = logging.getLogger("x6test." + trip); log.addHandler(h); log.setLevel(logging.INFO)
    for _ in range(4):                       # the poll is asynchronous: the flag is seen one or two steps later
        trainer._x6_guard(ms, mt, log)
        torch.cuda.synchronize()
    if trip == "tiny":
        assert mt.conv_math == ms.conv_math == "bf16x6"
        assert all(m.math == ops.MATH_BF16X6 for m in mt.modules() if hasattr(m, "math"))
        assert len(records) == 1 and "below 2^-110" in records[0].getMessage() and records[0].levelno == logging.INFO   # once, not per step
        ops.x6_range_flags(reset=True)
        return
    assert mt.conv_math == ms.conv_math == "f32"
    assert all(m.math == ops.MATH_F32 for m in mt.modules() if hasattr(m, "math"))
    assert len(records) == 1 and "range guard tripped" in records[0].getMessage()
    assert ops.x6_range_flags(reset=True) == 0
